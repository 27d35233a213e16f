// The walk-back backward kernel of STRICT mode, built with FMA contraction allowed (-ffp-contract=fast) but
// the strict math policy otherwise (Newton-refined v_rcp / v_rsq, TL_FAST = 0).
// "Strict" promises the reference's fp32 operation order for the FORWARD (ray_tracing_lite.py:525-571,
// 594-675), where every rounding is pinned by the oracle.  The walk-back has no reference operation order to
// follow -- the reference differentiates by autograd over a different graph -- so unfused multiplies and adds
// only cost instructions there: contraction removes a fifth of them (-21 % kernel time) and moves the
// gradients by 1e-8 relative (fewer roundings, not more).  The checkpoint kernel stays in tl_strict.hip: its
// forward sweep has to reproduce the forward kernel bit for bit.
#include "tl_common.h"
#define TL_NS tl_strict_wb_impl
#define TL_FAST 0
#define TL_ONLY_WALKBACK 1
#include "tl_kernels.inc"
namespace tl_strict_wb {
int api_walk_back(const tl_problem &p, const float *gx, const float *gy, const float *gcx, const float *gcy,
                  const double *gmom, const float *fx, const float *fy, const float *fcx, const float *fcy,
                  const uint8_t *fok, const double *fmom, float *gxin, float *gyin, double *part_inv,
                  unsigned *poison, unsigned token, int nbx, int R, hipStream_t st)
{ return tl_strict_wb_impl::launch_walk_back(p, gx, gy, gcx, gcy, gmom, fx, fy, fcx, fcy, fok, fmom, gxin, gyin, part_inv, poison, token, nbx, R, st); }
}
