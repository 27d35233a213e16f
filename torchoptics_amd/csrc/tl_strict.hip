// Strict arithmetic mode: build this TU with -ffp-contract=off (no FMA contraction) and the
// HIP default correctly-rounded fp32 sqrt/divide, so the forward is operation-for-operation
// the reference's eager fp32 graph (ray_tracing_lite.py:525-571, 594-675).
#include "tl_common.h"
#define TL_NS tl_strict_impl
#define TL_FAST 0
#include "tl_kernels.inc"
namespace tl_strict {
int api_fwd(const tl_problem &p, float *x, float *y, float *cx, float *cy, uint8_t *ok, uint8_t *back,
            float *opd, float *stacks, double *part, int nbx, int R, hipStream_t st)
{ return tl_strict_impl::launch_fwd(p, x, y, cx, cy, ok, back, opd, stacks, part, nbx, R, st); }
int api_bwd(const tl_problem &p, const float *gx, const float *gy, const float *gcx, const float *gcy,
            const double *gmom, float *gxin, float *gyin, double *part, int nbx, int R, hipStream_t st,
            const float *gopd)
{ return tl_strict_impl::launch_bwd(p, gx, gy, gcx, gcy, gmom, gxin, gyin, part, nbx, R, st, nullptr, nullptr, 0u, gopd); }
int api_bwd_inv(const tl_problem &p, const float *gx, const float *gy, const float *gcx, const float *gcy,
                const double *gmom, const float *fx, const float *fy, const float *fcx, const float *fcy,
                const uint8_t *fok, const double *fmom, float *gxin, float *gyin, double *part_inv, double *part_ck,
                unsigned *poison, unsigned token, int nbx, int R, int nbx_ck, int R_ck, hipStream_t st)
{ return tl_strict_impl::launch_bwd_inv(p, gx, gy, gcx, gcy, gmom, fx, fy, fcx, fcy, fok, fmom, gxin, gyin, part_inv, part_ck, poison, token, nbx, R, nbx_ck, R_ck, st); }
int api_selftest_arith(const float *a, const float *b, int64_t n, float *quot, float *root, hipStream_t st)
{ return tl_strict_impl::launch_selftest_arith(a, b, n, quot, root, st); }
}
