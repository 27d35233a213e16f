// tl_common.h -- shared by the per-mode kernel TUs and the C-ABI TU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>
#include "../../include/tl_trace.h"

// compile-time surface-row buckets of the backward kernel
static inline int tl_bwd_bucket(int S)
{
    static const int b[] = {4, 8, 12, 16, 20, 24, 32};
    for (int v : b) if (S <= v) return v;
    return -1;
}

// doubles per block-partial row of the backward kernel (see trace_bwd_kernel)
// g_c | g_t | g_mu [NS each] | g_z g_cx g_cy | [g_kappa[NS] | g_poly[NS][4]] | g_n[NS+1]
__host__ __device__ static inline int tl_bwd_row(int ns, bool asph) { return (asph ? 8 : 3) * ns + 3 + ns + 1; }

// Row counts the walk-back kernel is instantiated for with its row loop unrolled (trace_bwd_inv_unrolled_kernel), and
// the smallest pupil it takes (no skip there for waves past the end of a small pupil).
#define TL_INVU_MIN 3
#define TL_INVU_MAX 20
// bytes of the penalty walk-back's scan map (tl_kernels.inc: tl_scanmap): one per (row of the grid, 256-ray chunk, wave of
// the 256-thread block), rounded up to 256
__host__ __device__ static inline size_t tl_scanmap_bytes(int rows, int64_t nchunks)
{
    return (((size_t)rows * (size_t)nchunks * 4u) + 255u) & ~(size_t)255u;
}

static inline bool tl_walk_unrolled(int S, int P) { return S >= TL_INVU_MIN && S <= TL_INVU_MAX && P >= 256; }

// per-mode launchers (defined in tl_strict.hip / tl_fast.hip); return hipError_t as int
#define TL_DECLARE_MODE(NS)                                                                          \
    namespace NS {                                                                                   \
    int api_fwd(const tl_problem &p, float *x, float *y, float *cx, float *cy, uint8_t *ok,           \
                uint8_t *back, float *opd, float *stacks, double *part, int nbx, int R,             \
                hipStream_t st);                                                                     \
    int api_bwd(const tl_problem &p, const float *gx, const float *gy, const float *gcx,              \
                const float *gcy, const double *gmom, float *gxin, float *gyin, double *part,         \
                int nbx, int R, hipStream_t st, const float *gopd);                                  \
    int api_bwd_inv(const tl_problem &p, const float *gx, const float *gy, const float *gcx,          \
                    const float *gcy, const double *gmom, const float *fx, const float *fy,           \
                    const float *fcx, const float *fcy, const uint8_t *fok, const double *fmom,       \
                    float *gxin, float *gyin, double *part_inv, double *part_ck, unsigned *poison,    \
                    unsigned token, int nbx, int R, int nbx_ck, int R_ck, hipStream_t st);           \
    int api_selftest_arith(const float *a, const float *b, int64_t n, float *quot, float *root, hipStream_t st); \
    }
TL_DECLARE_MODE(tl_strict)
TL_DECLARE_MODE(tl_fast)

// double-precision twin (tl_f64.hip): generic, untuned kernels behind tl_trace_fwd_f64 / tl_trace_bwd_f64
namespace tl_f64 {
int launch_fwd(const tl_problem &p, double *x, double *y, double *cx, double *cy, uint8_t *ok, uint8_t *back, double *part, int nbx,
               hipStream_t st);
int launch_reduce_moments(const tl_problem &p, const double *part, double *mom, int nbx, hipStream_t st);
int launch_bwd(const tl_problem &p, const double *gx, const double *gy, const double *gcx, const double *gcy, const double *gmom,
               double *gxin, double *gyin, double *part, int nbx, double *g_c, double *g_t, double *g_mu, double *g_z, double *g_cx,
               double *g_cy, double *g_kappa, double *g_poly, hipStream_t st);
}
