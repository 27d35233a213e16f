// tl_api.hip -- the C ABI of libtltrace.so (declared in include/tl_trace.h): argument
// checks, launch planning, the fixed-order reduction kernels and the spot kernels.
#include <atomic>
#include "tl_common.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, const char *detail = "")
{
    snprintf(g_err, sizeof(g_err), fmt, detail);
    return code;
}

int hip_fail(int herr, const char *where)
{
    snprintf(g_err, sizeof(g_err), "%s: %s", where, hipGetErrorString((hipError_t)herr));
    return TL_ELAUNCH;
}

constexpr int kBlock = 256;

int env_int(const char *name, int dflt)
{
    const char *v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

// Launch plan: gridDim.x blocks of kBlock threads per (f, w), each thread tracing R rays.
// Every block pays a fixed prologue + reduction epilogue worth ~0.3-1 ray of work, so rays per lane want to be
// several; on the other hand the VALU-bound trace kernels need >= ~4 waves per SIMD in flight (>= ~1024 blocks, see
// DESIGN.md "What bounds the trace kernels") and an even share per block.  Measured on cfg2, cfg3 at 2^20 / 2^22 / 2^24
// points and a small cfg5 (profiles/r02_ab_launch_plan.txt): aim for ~`few` blocks until a lane has `rwant` rays, then
// grow the grid up to `cap` blocks, then let R grow.  Small workloads gain 10-20 % per kernel over a fixed 8192-block
// target (which gave them R = 1..2 with an uneven share).
struct Plan { int nbx, R; };

Plan make_plan(int P, int FW, int cap, int rmax, int few = 2048, int rwant = 8)
{
    // each block takes an equal share (to within one) of the kBlock-point chunks of the pupil (TL_BLOCK_CHUNKS in
    // tl_kernels.inc); R = the largest share
    const int64_t chunks = ((int64_t)P + kBlock - 1) / kBlock;
    int64_t r = chunks * FW / few;                            // rays per lane at ~`few` blocks
    if (r < 1) r = 1;
    if (r > rwant) r = rwant;
    int64_t nbx = (chunks + r - 1) / r;
    if (nbx * FW > cap) nbx = (cap + FW - 1) / FW;
    if (nbx > chunks) nbx = chunks;
    if (nbx * rmax < chunks) nbx = (chunks + rmax - 1) / rmax;
    if (nbx < 1) nbx = 1;
    Plan pl;
    pl.nbx = (int)nbx;
    pl.R = (int)((chunks + nbx - 1) / nbx);
    return pl;
}

// lenses in the launch (tl_problem.B, 0 read as 1) and rows of the grid's y dimension = (b, f, w) triples
inline int lenses(const tl_problem *p) { return p->B > 0 ? p->B : 1; }
inline int rows_bfw(const tl_problem *p) { return lenses(p) * p->F * p->W; }

Plan plan_fwd(const tl_problem *p)
{
    static const int cap = env_int("TL_FWD_BLOCKS", 8192), rmax = env_int("TL_FWD_RMAX", 32);
    static const int few = env_int("TL_PLAN_FEW", 2048), rwant = env_int("TL_PLAN_R", 8);
    return make_plan(p->P, rows_bfw(p), cap, rmax, few, rwant);
}

Plan plan_bwd(const tl_problem *p)
{
    // 16 rays per lane once there is work for >= 4096 such blocks (8 per lane and 2048 blocks in the forward): the
    // walk-back's per-block prologue (LDS rows, accumulator slots) and partial row are paid half as often: -1...2 % at
    // 16.8 M rays on every workload; at 8.4 M rays (2048 blocks of 16: a 1.3-round tail at 6 waves per SIMD) it stays at 8.
    // A/B tools/sessions/r3_s32.txt, r3_s33.txt
    static const int cap = env_int("TL_BWD_BLOCKS", 8192), rmax = env_int("TL_BWD_RMAX", 64);
    static const int few = env_int("TL_PLAN_FEW", 2048), rwant = env_int("TL_PLAN_R", 8);
    static const int rbig = env_int("TL_BWD_PLAN_R", 16), big = env_int("TL_BWD_PLAN_BIG", 4096);
    const int64_t chunks = ((int64_t)p->P + kBlock - 1) / kBlock;
    return make_plan(p->P, rows_bfw(p), cap, rmax, few, chunks * rows_bfw(p) >= (int64_t)rbig * big ? rbig : rwant);
}

int check_problem(const tl_problem *p)
{
    if (!p) return fail(TL_EINVAL, "tl_problem is NULL");
    if (p->F < 1 || p->W < 1 || p->S < 1 || p->P < 0) return fail(TL_EINVAL, "F, W, S must be >= 1 and P >= 0");
    if (p->S > TL_MAX_SURFACES) return fail(TL_EINVAL, "S exceeds TL_MAX_SURFACES (32)");
    if (p->B < 0) return fail(TL_EINVAL, "B must be >= 0 (0 is read as 1)");
    if ((int64_t)lenses(p) * p->F * p->W > 65535) return fail(TL_EINVAL, "B*F*W exceeds 65535");
    // (an empty shard, P = 0, has nothing to point x_in / y_in at: an empty tensor's data pointer is NULL)
    if ((p->P > 0 && (!p->x_in || !p->y_in)) || !p->z || !p->cx || !p->cy || !p->c || !p->t || !p->mu || !p->mask)
        return fail(TL_EINVAL, "a required device pointer of tl_problem is NULL");
    if (p->mode != TL_MODE_STRICT && p->mode != TL_MODE_FAST) return fail(TL_EINVAL, "unknown mode");
    if ((p->cx_stride | 1) != 1 || (p->cy_stride | 1) != 1) return fail(TL_EINVAL, "cx/cy stride must be 0 or 1");
    if ((p->surf_kind != nullptr) != (p->kappa != nullptr) || (p->surf_kind != nullptr) != (p->poly != nullptr))
        return fail(TL_EINVAL, "surf_kind, kappa and poly must be given together (or all NULL)");
    if (p->aggregate && p->S > 31) return fail(TL_EINVAL, "aggregate needs S <= 31");
    if (p->asph_hit_slots < 0 || p->asph_hit_slots > TL_MAX_HIT_SLOTS) return fail(TL_EINVAL, "asph_hit_slots must be 0..TL_MAX_HIT_SLOTS");
    if (p->asph_hits && p->asph_hit_slots == 0) return fail(TL_EINVAL, "asph_hits given with asph_hit_slots = 0");
    return TL_OK;
}

// ---------------------------------------------------------------- fixed-order reductions
// 256 values -> one: wave sums by shuffles, then the four wave sums in a fixed order (bitwise reproducible)
__device__ __forceinline__ double block_sum_256(double v, double *sm)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    return ((sm[0] + sm[1]) + sm[2]) + sm[3];
}

// sum part[col][row] over rows = {((f*W + w)*nbx + bx)} for f in [f0,f0+nf), w in [w0,w0+nw), nw == W or nw == 1.
// The partials are stored column-major ([column][F*W*nbx blocks]) and the rows of one f (all w, or one w) are one
// contiguous run: coalesced reads, no index arithmetic in the loop (the first version spent its 18-20 us per
// launch on two 64-bit divisions per element).
__device__ __forceinline__ double sum_rows(const double *part, int64_t nrows, int col, int W, int nbx,
                                           int f0, int nf, int w0, int nw, double *sm)
{
    const double *colp = part + (int64_t)col * nrows;
    const int count = nw * nbx;
    double acc = 0.0;
    for (int f = f0; f < f0 + nf; ++f) {
        const double *run = colp + ((int64_t)f * W + w0) * nbx;
        // eight independent loads and four partial sums per trip: with one load per trip the kernel spent its ~12 us on 32
        // dependent load latencies per thread (8192 partial rows at 16.8 M rays).  Still one fixed summation order.
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        int i = threadIdx.x;
        for (; i + 7 * kBlock < count; i += 8 * kBlock) {
            const double v0 = run[i], v1 = run[i + kBlock], v2 = run[i + 2 * kBlock], v3 = run[i + 3 * kBlock];
            const double v4 = run[i + 4 * kBlock], v5 = run[i + 5 * kBlock], v6 = run[i + 6 * kBlock], v7 = run[i + 7 * kBlock];
            a0 += v0 + v4; a1 += v1 + v5; a2 += v2 + v6; a3 += v3 + v7;
        }
        for (; i < count; i += kBlock) a0 += run[i];
        acc += (a0 + a1) + (a2 + a3);
    }
    return block_sum_256(acc, sm);
}

// clear_word (nullable): the walk-back's poison word of this problem's workspace (poison_word below), cleared here
// so that a step replayed from a HIP graph -- same token every replay -- starts clean without a memset node
__global__ __launch_bounds__(kBlock) void reduce_moments_kernel(const double *__restrict__ part,
                                                                double *__restrict__ mom, int W, int nbx,
                                                                unsigned *__restrict__ clear_word)
{
    __shared__ double sm[kBlock];
    const int f = blockIdx.x / TL_NMOM, j = blockIdx.x % TL_NMOM;
    const double s = sum_rows(part, (int64_t)(gridDim.x / TL_NMOM) * W * nbx, j, W, nbx, f, 1, 0, W, sm);
    if (threadIdx.x == 0) mom[blockIdx.x] = s;
    if (clear_word && blockIdx.x == 0 && threadIdx.x == 0) *clear_word = 0u;
}

__global__ __launch_bounds__(kBlock) void reduce_bwd_kernel(const double *__restrict__ part, int NS, int F,
                                                            int W, int S, int nbx, float *__restrict__ g_c,
                                                            float *__restrict__ g_t, float *__restrict__ g_mu,
                                                            float *__restrict__ g_z, float *__restrict__ g_cx,
                                                            float *__restrict__ g_cy, int ncol,
                                                            float *__restrict__ g_kappa, float *__restrict__ g_poly,
                                                            const double *__restrict__ alt_part, int alt_NS,
                                                            const double *__restrict__ fmom,
                                                            const unsigned *__restrict__ poison,
                                                            unsigned token, int alt_nbx, float *__restrict__ g_n,
                                                            int add_alt)
{
    __shared__ double sm[kBlock];
    // two candidate partial arrays (walk-back kernel / checkpoint fallback): the forward's conditioning count
    // and the walk-back's poison word (== this call's token) say which of the two launches did the work (same rule as
    // fallback_needed in tl_kernels.inc).  add_alt (penalty term): when the walk-back did its work, the checkpoint
    // launch behind it took the rays that died on the way -- the result is the SUM of the two arrays.
    // The count and the word are read FIRST and looked at LAST: the sum over the walk-back's array -- nearly always the
    // one that counts -- is formed while those loads are in flight (two dependent round trips off this tiny kernel).
    double n_ill = 0.0;
    unsigned pz = 0u;
    if (alt_part) {
        if (fmom)
            for (int f = 0; f < F * (int)gridDim.y; ++f) n_ill += fmom[(size_t)f * TL_NMOM + 9];
        if (poison) pz = *poison;
    }
    // one block per output scalar: g_c[S] | g_t[S] | g_mu[W,S] | g_z | g_cx[F] | g_cy[F] [| g_kappa[S] | g_poly[S,4]] [| g_n[W,S+1]]
    // of lens blockIdx.y, whose partial rows are those of the fields [lens F, (lens + 1) F)
    const int lens = blockIdx.y;
    g_c += lens * S; g_t += lens * S; g_mu += lens * W * S; g_z += lens; g_cx += lens * F; g_cy += lens * F;
    if (g_kappa) { g_kappa += lens * S; g_poly += lens * S * TL_MAX_POLY; }
    if (g_n) g_n += lens * W * (S + 1);
    int b = blockIdx.x;
    float *out;
    // column = a * NS + c0 in a partial array whose rows were written for NS surface rows
    int ca, c0, f0 = 0, nf = F, w0 = 0, nw = W;
    const int n_asph = g_kappa ? 5 * S : 0;
    if (b < S) { ca = 0; c0 = b; out = g_c + b; }
    else if ((b -= S) < S) { ca = 1; c0 = b; out = g_t + b; }
    else if ((b -= S) < W * S) { const int w = b / S, k = b % S; ca = 2; c0 = k; w0 = w; nw = 1; out = g_mu + b; }
    else if ((b -= W * S) < 1) { ca = 3; c0 = 0; out = g_z; }
    else if ((b -= 1) < F) { ca = 3; c0 = 1; f0 = b; nf = 1; out = g_cx + b; }
    else if ((b -= F) < F) { ca = 3; c0 = 2; f0 = b; nf = 1; out = g_cy + b; }
    else if ((b -= F) < n_asph) {                                                        // aspheric rows only
        if (b < S) { ca = 3; c0 = 3 + b; out = g_kappa + b; }
        else { ca = 4; c0 = 3 + (b - S); out = g_poly + (b - S); }                       // b - S = 4k + j
    } else {                                                                             // g_n[w][k], k <= S (OPD gradient only)
        b -= n_asph;
        const int w = b / (S + 1), k = b % (S + 1);
        ca = g_kappa ? 8 : 3; c0 = 3 + k; w0 = w; nw = 1; out = g_n + b;
    }
    (void)ncol;
    double s = sum_rows(part, (int64_t)gridDim.y * F * W * nbx, ca * NS + c0, W, nbx, lens * F + f0, nf, w0, nw, sm);
    if (alt_part) {
        // add_alt bit 1: the forward left per-ray conditioning flags (tl_problem.cond_flags), so ill-conditioned rays
        // (n_ill > 0) were split between the two launches as well, instead of sending the whole launch to the checkpoint kernel
        const bool alt_only = (n_ill > 0.0 && !(add_alt & 2)) || (poison && pz == token);
        if (alt_only || (add_alt & 1) || n_ill > 0.0) {
            __syncthreads();                          // sm is reused
            const double s2 = sum_rows(alt_part, (int64_t)gridDim.y * F * W * alt_nbx, ca * alt_NS + c0, W, alt_nbx,
                                       lens * F + f0, nf, w0, nw, sm);
            s = alt_only ? s2 : s + s2;
        }
    }
    if (threadIdx.x == 0) *out = (float)s;        // summed in fp64, rounded once
}

// rms = mean_f sqrt((M2 - 2 m M1 + m^2 M3)/n), m = M0/n  (compute_rms2d on the moments)
// and d rms / d moments, in one block: replaces ~25 tiny elementwise kernels of the autograd graph.
// One block per lens (blockIdx.x): rms[b] from the moments rows [b F, (b + 1) F).
__global__ __launch_bounds__(64) void spot_rms_kernel(const double *__restrict__ mom, int F, double n,
                                                      float *__restrict__ rms, double *__restrict__ dmom)
{
    mom += (size_t)blockIdx.x * F * TL_NMOM;
    dmom += (size_t)blockIdx.x * F * TL_NMOM;
    rms += blockIdx.x;
    double acc = 0.0;
    for (int f = threadIdx.x; f < F; f += 64) {
        const double *M = mom + (size_t)f * TL_NMOM;
        const double m = M[0] / n;
        const double var = (M[2] - 2.0 * m * M[1] + m * m * M[3]) / n;
        const double sd = sqrt(var);
        acc += sd;
        const double k = var > 0.0 ? 1.0 / (2.0 * (double)F * n * sd) : 0.0;
        double *D = dmom + (size_t)f * TL_NMOM;
        for (int j = 0; j < TL_NMOM; ++j) D[j] = 0.0;
        D[0] = k * (-2.0 * M[1] + 2.0 * m * M[3]) / n;
        D[1] = -2.0 * m * k;
        D[2] = k;
        D[3] = m * m * k;
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (threadIdx.x == 0) *rms = (float)(acc / (double)F);
}

// loss_dict of compute_loss_out (optics_simulator_lite.py:430-450) per lens, on the moments: rms as spot_rms_kernel,
// penalty = (sum_f M8) / n_sequence rounded to fp32, loss_unsup = rms + rate * penalty in fp32 (the rounding points of the
// op sequence  rms + penalty_rate * (q / n_seq).to(float32)  it replaces).  One block per lens.
__global__ __launch_bounds__(64) void unsup_loss_kernel(const double *__restrict__ mom, int F, double n,
                                                        const double *__restrict__ n_seq, double n_seq_all, float rate,
                                                        float *__restrict__ loss, float *__restrict__ rms,
                                                        float *__restrict__ pen, double *__restrict__ dmom)
{
#pragma clang fp contract(off)
    const int b = blockIdx.x;
    mom += (size_t)b * F * TL_NMOM;
    dmom += (size_t)b * F * TL_NMOM;
    double acc = 0.0, q = 0.0;
    for (int f = threadIdx.x; f < F; f += 64) {
        const double *M = mom + (size_t)f * TL_NMOM;
        const double m = M[0] / n;
        const double var = (M[2] - 2.0 * m * M[1] + m * m * M[3]) / n;
        const double sd = sqrt(var);
        acc += sd;
        q += M[8];
        const double k = var > 0.0 ? 1.0 / (2.0 * (double)F * n * sd) : 0.0;
        double *D = dmom + (size_t)f * TL_NMOM;
        for (int j = 0; j < TL_NMOM; ++j) D[j] = 0.0;
        D[0] = k * (-2.0 * M[1] + 2.0 * m * M[3]) / n;
        D[1] = -2.0 * m * k;
        D[2] = k;
        D[3] = m * m * k;
    }
    for (int o = 32; o > 0; o >>= 1) { acc += __shfl_xor(acc, o, 64); q += __shfl_xor(q, o, 64); }
    if (threadIdx.x == 0) {
        const float r = (float)(acc / (double)F);
        // (a tensor divided by a host scalar is a multiplication by its reciprocal in PyTorch's kernels: same here)
        const float pn = (float)(n_seq ? q / n_seq[b] : q * (1.0 / n_seq_all));
        const float prod = rate * pn;
        rms[b] = r;
        pen[b] = pn;
        loss[b] = r + prod;
    }
}

// d(sum_b g_loss[b] loss[b] + g_rms[b] rms[b] + g_pen[b] penalty[b]) / d moments, with the same rounding points as
// autograd over the op sequence: the fp32 sums g_loss + g_rms and rate * g_loss + g_pen, then fp64
__global__ __launch_bounds__(kBlock) void unsup_loss_bwd_kernel(const double *__restrict__ dmom, int F, int total,
                                                                const float *__restrict__ g_loss, const float *__restrict__ g_rms,
                                                                const float *__restrict__ g_pen, int g_stride,
                                                                const double *__restrict__ n_seq, double n_seq_all, float rate,
                                                                double *__restrict__ gmom)
{
#pragma clang fp contract(off)
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= total) return;
    const int b = i / (F * TL_NMOM), j = i % TL_NMOM;
    const int gb = b * g_stride;
    float a = 0.0f, c = 0.0f;
    if (g_loss) { a = g_loss[gb]; c = rate * g_loss[gb]; }
    if (g_rms) a = g_loss ? a + g_rms[gb] : g_rms[gb];
    if (g_pen) c = g_loss ? c + g_pen[gb] : g_pen[gb];
    double v = dmom[i] * (double)a;
    if (j == 8) v += n_seq ? (double)c / n_seq[b] : (double)c * (1.0 / n_seq_all);
    gmom[i] = v;
}

// ---------------------------------------------------------------- spot kernels (strided tensors)
__global__ __launch_bounds__(kBlock) void spot_moments_kernel(int P, int W, const float *__restrict__ x,
                                                              const float *__restrict__ y,
                                                              const uint8_t *__restrict__ ok, int64_t s_f,
                                                              int64_t s_p, int64_t s_w, double *__restrict__ part,
                                                              int R)
{
    const int fw = blockIdx.y;
    const int f = fw / W, w = fw - f * W;
    const int64_t off = f * s_f + w * s_w;
    double m[TL_NMOM];
#pragma unroll
    for (int j = 0; j < TL_NMOM; ++j) m[j] = 0.0;
    for (int r = 0; r < R; ++r) {
        const int64_t ip = ((int64_t)blockIdx.x * R + r) * kBlock + threadIdx.x;
        if (ip < P) {
            const int64_t o = off + ip * s_p;
            const double yd = (double)y[o], xd = x ? (double)x[o] : 0.0, okd = ok[o] ? 1.0 : 0.0;
            m[0] += yd; m[1] += okd * yd; m[2] += okd * yd * yd; m[3] += okd;
            m[4] += xd; m[5] += okd * xd; m[6] += okd * xd * xd;
        }
    }
    __shared__ double red[kBlock / 64][TL_NMOM];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < TL_NMOM; ++j) {
        double v = m[j];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) red[wv][j] = v;
    }
    __syncthreads();
    if (threadIdx.x < TL_NMOM) {
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < kBlock / 64; ++q) s += red[q][threadIdx.x];
        part[(size_t)threadIdx.x * ((size_t)gridDim.y * gridDim.x) + ((size_t)fw * gridDim.x + blockIdx.x)] = s;   // [column][block]
    }
}

__global__ __launch_bounds__(kBlock) void spot_seed_kernel(int P, int W, const float *__restrict__ x,
                                                           const float *__restrict__ y,
                                                           const uint8_t *__restrict__ ok, int64_t s_f, int64_t s_p,
                                                           int64_t s_w, const double *__restrict__ gmom,
                                                           float *__restrict__ gx, float *__restrict__ gy)
{
    const int fw = blockIdx.y;
    const int f = fw / W, w = fw - f * W;
    const double *q = gmom + (size_t)f * TL_NMOM;
    const double g0 = q[0], g1 = q[1], g2 = q[2], g4 = q[4], g5 = q[5], g6 = q[6];
    const int64_t ip = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (ip >= P) return;
    const int64_t o = f * s_f + w * s_w + ip * s_p;
    const double okd = ok[o] ? 1.0 : 0.0;
    if (gy) gy[o] = (float)(g0 + okd * (g1 + 2.0 * (double)y[o] * g2));
    if (gx) gx[o] = x ? (float)(g4 + okd * (g5 + 2.0 * (double)x[o] * g6)) : 0.0f;
}

}  // namespace

// =================================================================== C ABI
// Paraxial entrance-pupil position (the `z` argument of the trace): z = B/A of the ordered product
// M = M_{K-1} ... M_0 of the rows in front of the stop, M_k = [[1 + P t, r t], [P, r]], r = n_k / n_{k+1},
// P = c (r - 1)  (ray_tracing_lite.py:301-350, lens_modeling.py ABCD).  K <= 32 rows: one thread per lens, fp64 inside.
// Replaces ~25 tiny elementwise / 2x2-matmul launches of the host chain and ~60 of its autograd backward.
// A batch of B lenses: c, t [B,K], n [B,K+1], z, g_z [B]; rows behind a lens' own stop are padded (c = 0, t = 0,
// n = 1): identity matrices.
// strict: the VALUE of z is the reference's fp32 arithmetic, operation for operation -- 2x2 matrices of fp32 entries
// [[1 + P t, r t], [P, r]], multiplied pairwise in a halving tree (reduce_abcd, ray_tracing_lite.py:301-318: rows
// (1,0), (3,2) ... then the products again, an odd one carried), every product and sum rounded separately (the
// reference's tiny batched matmul does not fuse; checked bit for bit against the z of fixtures G1-G11 on the Cooke
// triplet, the doublet and the Tessar).  The gradient stays the fp64 adjoint below: the derivative of the exact
// function, which the derivative of the rounded one equals to rounding.
__device__ float pupil_position_fp32_tree(int K, const float *__restrict__ c, const float *__restrict__ t,
                                          const float *__restrict__ n)
{
    // Plain operators under contract(off): every product and sum is rounded on its own.  (HIP's __fmul_rn / __fadd_rn are
    // inline functions compiled under the translation unit's default, contract(fast): inlined here they would still fuse.)
#pragma clang fp contract(off)
    float M[TL_MAX_SURFACES][4];
    for (int k = 0; k < K; ++k) {
        const float r = n[k] / n[k + 1];
        const float P = c[k] * (r - 1.0f);
        const float Pt = P * t[k];
        M[k][0] = 1.0f + Pt; M[k][1] = r * t[k]; M[k][2] = P; M[k][3] = r;
    }
    int cnt = K;
    while (cnt > 1) {
        const int even = cnt - (cnt & 1);
        for (int i = 0; i < even; i += 2) {          // out[i/2] = M[i+1] @ M[i]
            const float a0 = M[i + 1][0], a1 = M[i + 1][1], a2 = M[i + 1][2], a3 = M[i + 1][3];
            const float b0 = M[i][0], b1 = M[i][1], b2 = M[i][2], b3 = M[i][3];
            float *o = M[i >> 1];
            const float p00 = a0 * b0, q00 = a1 * b2, p01 = a0 * b1, q01 = a1 * b3;
            const float p10 = a2 * b0, q10 = a3 * b2, p11 = a2 * b1, q11 = a3 * b3;
            o[0] = p00 + q00; o[1] = p01 + q01; o[2] = p10 + q10; o[3] = p11 + q11;
        }
        if (cnt & 1) { for (int j = 0; j < 4; ++j) M[even >> 1][j] = M[cnt - 1][j]; }
        cnt = (even >> 1) + (cnt & 1);
    }
    return M[0][1] / M[0][0];
}

__global__ __launch_bounds__(64) void pupil_position_kernel(int B, int K, const float *__restrict__ c,
                                                            const float *__restrict__ t, const float *__restrict__ n,
                                                            float *__restrict__ z, const float *__restrict__ g_z,
                                                            float *__restrict__ g_c, float *__restrict__ g_t,
                                                            float *__restrict__ g_n, int strict)
{
    const int lens = blockIdx.x * 64 + threadIdx.x;
    if (lens >= B) return;
    c += (size_t)lens * K; t += (size_t)lens * K; n += (size_t)lens * (K + 1);
    if (z) z += lens;
    if (g_z) { g_z += lens; g_c += (size_t)lens * K; g_t += (size_t)lens * K; g_n += (size_t)lens * (K + 1); }
    double R[TL_MAX_SURFACES][4];            // R[k] = M_{k-1} ... M_0 (row-major a b / c d), R[0] = I
    double a = 1.0, b = 0.0, cc = 0.0, d = 1.0;
    for (int k = 0; k < K; ++k) {
        R[k][0] = a; R[k][1] = b; R[k][2] = cc; R[k][3] = d;
        const double r = (double)n[k] / (double)n[k + 1], P = (double)c[k] * (r - 1.0), tk = (double)t[k];
        const double m00 = 1.0 + P * tk, m01 = r * tk, m10 = P, m11 = r;
        const double na = m00 * a + m01 * cc, nb = m00 * b + m01 * d, nc = m10 * a + m11 * cc, nd = m10 * b + m11 * d;
        a = na; b = nb; cc = nc; d = nd;
    }
    if (z) z[0] = strict ? pupil_position_fp32_tree(K, c, t, n) : (float)(b / a);
    if (!g_z) return;
    // reverse mode: G = d z / d M_full, carried down as G_k = L_k^T G with L_k = M_{K-1} ... M_{k+1}
    const double gz = (double)g_z[0];
    double G00 = -gz * b / (a * a), G01 = gz / a, G10 = 0.0, G11 = 0.0;
    for (int k = 0; k <= K; ++k) g_n[k] = 0.0f;
    double carry_n = 0.0;                     // d/d n[k+1] collected from row k+1 (the "n_k" side of that row)
    for (int k = K - 1; k >= 0; --k) {
        const double nk = (double)n[k], nk1 = (double)n[k + 1];
        const double r = nk / nk1, ck = (double)c[k], tk = (double)t[k], P = ck * (r - 1.0);
        // dM_k = G_k R_k^T
        const double d00 = G00 * R[k][0] + G01 * R[k][1], d01 = G00 * R[k][2] + G01 * R[k][3];
        const double d10 = G10 * R[k][0] + G11 * R[k][1], d11 = G10 * R[k][2] + G11 * R[k][3];
        const double dP = d00 * tk + d10;
        const double dt = d00 * P + d01 * r;
        const double dr = d01 * tk + d11 + dP * ck;
        g_c[k] = (float)(dP * (r - 1.0));
        g_t[k] = (float)dt;
        // r = n_k / n_{k+1}
        g_n[k + 1] = (float)(carry_n - dr * nk / (nk1 * nk1));
        carry_n = dr / nk1;
        // G_{k-1} = M_k^T G_k
        const double m00 = 1.0 + P * tk, m01 = r * tk, m10 = P, m11 = r;
        const double h00 = m00 * G00 + m10 * G10, h01 = m00 * G01 + m10 * G11;
        const double h10 = m01 * G00 + m11 * G10, h11 = m01 * G01 + m11 * G11;
        G00 = h00; G01 = h01; G10 = h10; G11 = h11;
    }
    g_n[0] = (float)carry_n;
}

// ---------------------------------------------------------------- ray aiming (ray_tracing_lite.py:129-208, 834-844)
// One thread per (lens, field, wavelength): everything the reference's RayTracer.ray_aiming does with ~60 tensor ops, two
// traces and an autograd pass over 3 rays per thread -- in fp64, in registers:
//   rs       height of the marginal ray (pupil y = 1, on axis, d line) at the stop                 (compute_pupil_radius :834-844)
//   tee rays bottom / top meridional and +x sagittal ray (pupil (0,-1), (0,1), (1,0)) traced to the stop through the
//            rows in front of it, positions over rs                                               (:169-176)
//   J        d(xs_rel + ys_rel)/d(pupil x) of the sagittal ray, d(.)/d(pupil y) of the two meridional ones -- the
//            reference backpropagates ones through xs_rel and ys_rel into the same .grad (:177-181); here central
//            differences in fp64 (h = 1e-4 of the pupil radius: ~1e-9 relative)
//   remap    one Newton step per ray, non-finite steps dropped, then the affine map of the pupil          (:183-208)
// A dead ray (miss / total internal reflection on the way) takes no step, as in the reference (its zeroed outputs and
// zero Jacobian give a non-finite step, which is dropped).
struct AimLens {
    int K;
    const float *c, *t, *n;          // c, t [K]; n [K][W] indices behind each row for the tracer's wavelengths
    const float *n_d;                // [K] d-line indices (marginal ray)
    const uint8_t *mask, *kind;      // [K]; kind nullable
    const float *kappa, *poly;       // [K], [K][4]; nullable with kind
    int W;
    bool allow_back;
};

// fp64 restatement of one ray through the rows (SURVEY Appendix A op for op, aspheric rows by Newton to convergence).
// w < 0: the d line.  Returns false for a ray the forward would have flagged not ok.
__device__ bool aim_trace(const AimLens &L, int w, double x, double y, double z, double cx, double cy, double &xo, double &yo)
{
    const double eps = 1e-6;
    double cz = sqrt((1.0 - cx * cx) - cy * cy);
    double n_prev = 1.0;
    for (int k = 0; k < L.K; ++k) {
        const double c = (double)L.c[k], t = (double)L.t[k];
        const double n_k = (double)(w < 0 ? L.n_d[k] : L.n[(size_t)k * L.W + w]);
        const double mu = (double)((float)n_prev / (float)n_k);          // the host chain forms mu in fp32 (:123)
        n_prev = n_k;
        double X, Y, Z, dz, cos_i, nx, ny, nz;
        const bool asph = L.kind && L.kind[k];
        if (!asph) {
            const double e = -((x * cx + y * cy) + z * cz);
            const double mz = z + e * cz;
            const double m2 = ((x * x + y * y) + z * z) - e * e;
            const double tmp = c * m2 - 2.0 * mz;
            const double cos2 = cz * cz - c * tmp;
            if (cos2 - eps < 0.0) return false;
            cos_i = sqrt(cos2);
            const double d = e + tmp / (cz + cos_i);
            dz = d * cz;
            X = x + d * cx; Y = y + d * cy; Z = z + dz;
            nx = -c * X; ny = -c * Y; nz = 1.0 - c * Z;                  // unit normal of the sphere at the hit
        } else {
            const double kap = (double)L.kappa[k];
            const double a0 = (double)L.poly[4 * k], a1 = (double)L.poly[4 * k + 1], a2 = (double)L.poly[4 * k + 2], a3 = (double)L.poly[4 * k + 3];
            const double Kc = 1.0 + kap;
            const double e = -((x * cx + y * cy) + Kc * (z * cz));
            const double dd = (cx * cx + cy * cy) + Kc * (cz * cz);
            const double rr = (x * x + y * y) + Kc * (z * z);
            const double bq = c * e + cz, cq = c * rr - 2.0 * z;
            const double disc = bq * bq - (c * dd) * cq;
            if (disc - eps < 0.0) return false;
            double s = cq / (bq + sqrt(disc)), dsag = 0.0, rho = 0.0, F = 0.0;
            bool bad = false;
            for (int it = 0; it < 12; ++it) {
                X = x + s * cx; Y = y + s * cy; Z = z + s * cz;
                rho = X * X + Y * Y;
                const double q2 = 1.0 - Kc * c * c * rho;
                bad = q2 - eps < 0.0;
                const double q = sqrt(bad ? 1.0 : q2);
                const double sag = c * rho / (1.0 + q) + rho * rho * (a0 + rho * (a1 + rho * (a2 + rho * a3)));
                dsag = c / (2.0 * q) + rho * (2.0 * a0 + rho * (3.0 * a1 + rho * (4.0 * a2 + rho * (5.0 * a3))));
                F = Z - sag;
                if (fabs(F) <= 1e-13 * (1.0 + fabs(Z))) break;
                s -= F / (cz - dsag * (2.0 * (X * cx + Y * cy)));
            }
            if (bad || !(fabs(F) <= 1e-9 * (1.0 + fabs(Z)))) return false;
            dz = s * cz;
            const double m = 2.0 * dsag, inv_n = 1.0 / sqrt(1.0 + m * m * rho);
            nx = -(m * X) * inv_n; ny = -(m * Y) * inv_n; nz = inv_n;
            cos_i = (cx * nx + cy * ny) + cz * nz;
        }
        // refraction (vector form; for a sphere it is apply_snell_spherical :554-568 written with the normal)
        const double cos2_t = 1.0 - (mu * mu) * (1.0 - cos_i * cos_i);
        if (cos2_t - eps < 0.0) return false;
        const double g = sqrt(cos2_t) - mu * cos_i;
        const double cx3 = mu * cx + g * nx, cy3 = mu * cy + g * ny;
        const double czsq = 1.0 - (cx3 * cx3 + cy3 * cy3);
        if (k > 0 && L.mask[k - 1] && dz < 0.0 && !L.allow_back) return false;       // :626-632
        if (czsq - eps < 0.0) return false;
        x = X; y = Y; z = Z - t;
        cx = cx3; cy = cy3; cz = sqrt(czsq);
    }
    const double dzi = -z, dist = dzi / cz;                                          // :660-663
    if (L.mask[L.K - 1] && dzi < 0.0 && !L.allow_back) return false;
    xo = x + dist * cx; yo = y + dist * cy;
    return true;
}

__global__ __launch_bounds__(64) void ray_aim_kernel(int B, int F, int W, int K, const float *__restrict__ c,
                                                     const float *__restrict__ t, const float *__restrict__ n,
                                                     const float *__restrict__ n_d, const uint8_t *__restrict__ mask,
                                                     const float *__restrict__ kappa, const float *__restrict__ poly,
                                                     const uint8_t *__restrict__ kind, const float *__restrict__ z,
                                                     const float *__restrict__ hfov, const float *__restrict__ fields,
                                                     const float *__restrict__ epd, int allow_back,
                                                     float *__restrict__ x_scale, float *__restrict__ y_scale,
                                                     float *__restrict__ y_offset)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= B * F * W) return;
    const int w = i % W, f = (i / W) % F, b = i / (W * F);
    AimLens L;
    L.K = K; L.W = W; L.allow_back = allow_back != 0;
    L.c = c + (size_t)b * K; L.t = t + (size_t)b * K; L.n = n + (size_t)b * K * W; L.n_d = n_d + (size_t)b * K;
    L.mask = mask + (size_t)b * K;
    L.kind = kind ? kind + (size_t)b * K : nullptr;
    L.kappa = kind ? kappa + (size_t)b * K : nullptr;
    L.poly = kind ? poly + (size_t)b * K * 4 : nullptr;
    const double z0 = (double)z[b], half = 0.5 * (double)epd[b];
    const double cy0 = sin((double)(hfov[b] * fields[f]));               // the correctly rounded sine of the fp32 angle (assemble)
    const double cyf = (double)(float)cy0;
    double xs, rs = 0.0;
    const bool ok_m = aim_trace(L, -1, 0.0, half, z0, 0.0, 0.0, xs, rs);  // compute_pupil_radius: marginal ray, on axis, d line
    const double h = 1e-4;
    double step[3] = {0.0, 0.0, 0.0};                                     // Newton step of (bottom y, top y, sagittal x)
    const double px[3] = {0.0, 0.0, 1.0}, py[3] = {-1.0, 1.0, 0.0};
    for (int r = 0; r < 3 && ok_m; ++r) {
        double x0, y0, xa, ya, xb, yb;
        const double ex = (r == 2) ? h : 0.0, ey = (r == 2) ? 0.0 : h;
        const bool ok = aim_trace(L, w, px[r] * half, py[r] * half, z0, 0.0, cyf, x0, y0) &&
                        aim_trace(L, w, (px[r] + ex) * half, (py[r] + ey) * half, z0, 0.0, cyf, xa, ya) &&
                        aim_trace(L, w, (px[r] - ex) * half, (py[r] - ey) * half, z0, 0.0, cyf, xb, yb);
        if (!ok) continue;
        const double j = ((xa + ya) - (xb + yb)) / (2.0 * h * rs);        // d(xs_rel + ys_rel) / d(pupil coordinate)
        const double pos = ((r == 2) ? x0 : y0) / rs, want = (r == 2) ? 1.0 : py[r];
        const double d = -(pos - want) / j;
        step[r] = (d - d == 0.0) ? d : 0.0;                               // non-finite -> no step (:185-186)
    }
    // affine pupil map through the corrected tee rays (:187-208)
    x_scale[i] = (float)(1.0 + step[2]);
    y_scale[i] = (float)(((1.0 + step[1]) - (-1.0 + step[0])) / 2.0);
    y_offset[i] = (float)(((-1.0) * step[1] - (1.0) * step[0]) / (-2.0));
}

extern "C" {

int tl_version(void) { return TL_ABI_VERSION; }

const char *tl_last_error(void) { return g_err; }

size_t tl_problem_size(void) { return sizeof(tl_problem); }

// The walk-back kernel's poison word lives in the slack at the END of the workspace this problem asks for (the
// last 256 bytes belong to no partial array), at an address tl_trace_fwd can compute as well.
static unsigned *poison_word(const tl_problem *p, void *workspace)
{
    return (unsigned *)((char *)workspace + tl_workspace_bytes(p) - 64);
}

size_t tl_workspace_bytes(const tl_problem *p)
{
    if (!p || p->F < 1 || p->W < 1 || p->S < 1 || p->P < 0) return 0;
    const Plan pf = plan_fwd(p), pb = plan_bwd(p);
    const int ns = tl_bwd_bucket(p->S);
    const size_t fw = (size_t)rows_bfw(p);
    const size_t a = fw * pf.nbx * TL_NMOM * sizeof(double);
    const size_t b = fw * pb.nbx * (size_t)tl_bwd_row(ns < 0 ? TL_MAX_SURFACES : ns, p->surf_kind != nullptr) * sizeof(double);
    const size_t c = fw * pb.nbx * (size_t)((p->surf_kind ? 8 : 3) * p->S + 3) * sizeof(double);   // walk-back kernel next to its fallback
    // + the penalty walk-back's scan map, right below the 256-byte slack at the end (the poison word lives in that slack)
    const size_t scan = tl_scanmap_bytes(rows_bfw(p), ((int64_t)p->P + kBlock - 1) / kBlock);
    return (a > b + c ? a : b + c) + scan + 256;
}

int tl_trace_fwd(const tl_problem *p, float *x, float *y, float *cx, float *cy, uint8_t *ok, uint8_t *back,
                 float *opd, float *stacks, double *moments, void *workspace, size_t workspace_bytes, void *stream)
{
    int rc = check_problem(p);
    if (rc) return rc;
    if (stacks && !p->aggregate) return fail(TL_EINVAL, "the stacks output needs tl_problem.aggregate");
    if (opd && !p->n_index) return fail(TL_EINVAL, "the opd output needs tl_problem.n_index");
    if (opd && p->aggregate) return fail(TL_EINVAL, "the opd output and aggregate (penalty term) cannot be combined in one call");
    if (p->P == 0) {
        if (moments) {
            hipError_t e = hipMemsetAsync(moments, 0, (size_t)lenses(p) * p->F * TL_NMOM * sizeof(double), (hipStream_t)stream);
            if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(moments)");
        }
        return TL_OK;
    }
    hipError_t e = hipSetDevice(p->device);
    if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
    const Plan pl = plan_fwd(p);
    double *part = nullptr;
    if (moments) {
        const size_t need = (size_t)rows_bfw(p) * pl.nbx * TL_NMOM * sizeof(double);
        if (!workspace || workspace_bytes < need) return fail(TL_EWORKSPACE, "workspace too small for tl_trace_fwd");
        part = (double *)workspace;
    }
    hipStream_t st = (hipStream_t)stream;
    tl_problem q = *p;
    if (!opd) q.n_index = nullptr;          // the kernel accumulates the path length only when asked
    int herr = (p->mode == TL_MODE_FAST) ? tl_fast::api_fwd(q, x, y, cx, cy, ok, back, opd, stacks, part, pl.nbx, pl.R, st)
                                         : tl_strict::api_fwd(q, x, y, cx, cy, ok, back, opd, stacks, part, pl.nbx, pl.R, st);
    if (herr) return hip_fail(herr, "trace_fwd_kernel launch");
    if (moments) {
        unsigned *clear = (workspace_bytes >= tl_workspace_bytes(p)) ? poison_word(p, workspace) : nullptr;
        hipLaunchKernelGGL(reduce_moments_kernel, dim3(lenses(p) * p->F * TL_NMOM), dim3(kBlock), 0, st, part, moments, p->W, pl.nbx, clear);
        herr = (int)hipGetLastError();
        if (herr) return hip_fail(herr, "reduce_moments_kernel launch");
    } else if (workspace && workspace_bytes >= tl_workspace_bytes(p)) {
        // no moments, so no reduction kernel to clear the walk-back's poison word for a replayed step: clear it here
        hipError_t e2 = hipMemsetAsync(poison_word(p, workspace), 0, sizeof(unsigned), st);
        if (e2 != hipSuccess) return hip_fail(e2, "hipMemsetAsync(poison word)");
    }
    return TL_OK;
}

int tl_trace_bwd(const tl_problem *p, const float *gx, const float *gy, const float *gcx, const float *gcy,
                 const double *g_moments, const float *g_opd, float *g_c, float *g_t, float *g_mu, float *g_z, float *g_cx,
                 float *g_cy, float *g_kappa, float *g_poly, float *g_n_index, float *g_x_in, float *g_y_in, void *workspace,
                 size_t workspace_bytes, void *stream)
{
    int rc = check_problem(p);
    if (rc) return rc;
    if ((g_opd != nullptr) != (g_n_index != nullptr)) return fail(TL_EINVAL, "g_opd and g_n_index must be given together");
    if (g_opd && !p->n_index) return fail(TL_EINVAL, "the gradient of the optical path length needs tl_problem.n_index");
    if (!g_c || !g_t || !g_mu || !g_z || !g_cx || !g_cy) return fail(TL_EINVAL, "a parameter-gradient output is NULL");
    if ((g_kappa || g_poly) && !p->surf_kind) return fail(TL_EINVAL, "g_kappa / g_poly need aspheric rows (surf_kind)");
    if (p->surf_kind && (!g_kappa || !g_poly)) return fail(TL_EINVAL, "aspheric rows need g_kappa and g_poly outputs");
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipSetDevice(p->device);
    if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
    if (p->P == 0) {
        const size_t S = p->S, B = (size_t)lenses(p);
        if ((e = hipMemsetAsync(g_c, 0, B * S * 4, st)) || (e = hipMemsetAsync(g_t, 0, B * S * 4, st)) ||
            (e = hipMemsetAsync(g_mu, 0, B * S * p->W * 4, st)) || (e = hipMemsetAsync(g_z, 0, B * 4, st)) ||
            (e = hipMemsetAsync(g_cx, 0, B * p->F * 4, st)) || (e = hipMemsetAsync(g_cy, 0, B * p->F * 4, st)) ||
            (g_kappa && (e = hipMemsetAsync(g_kappa, 0, B * S * 4, st))) || (g_poly && (e = hipMemsetAsync(g_poly, 0, B * S * 16, st))) ||
            (g_n_index && (e = hipMemsetAsync(g_n_index, 0, B * (S + 1) * p->W * 4, st))))
            return hip_fail(e, "hipMemsetAsync(grads)");
        return TL_OK;
    }
    const Plan pl = plan_bwd(p);
    const int ns = tl_bwd_bucket(p->S);
    const bool asph = p->surf_kind != nullptr;
    const int ncol = tl_bwd_row(ns, asph);
    const size_t need = (size_t)rows_bfw(p) * pl.nbx * (size_t)ncol * sizeof(double);
    if (!workspace || workspace_bytes < need) return fail(TL_EWORKSPACE, "workspace too small for tl_trace_bwd");
    double *part = (double *)workspace;
    int herr = (p->mode == TL_MODE_FAST)
                   ? tl_fast::api_bwd(*p, gx, gy, gcx, gcy, g_moments, g_x_in, g_y_in, part, pl.nbx, pl.R, st, g_opd)
                   : tl_strict::api_bwd(*p, gx, gy, gcx, gcy, g_moments, g_x_in, g_y_in, part, pl.nbx, pl.R, st, g_opd);
    if (herr) return hip_fail(herr, "trace_bwd_kernel launch");
    const int nout = 2 * p->S + p->W * p->S + 1 + 2 * p->F + (asph ? 5 * p->S : 0) + (g_opd ? p->W * (p->S + 1) : 0);
    hipLaunchKernelGGL(reduce_bwd_kernel, dim3(nout, lenses(p)), dim3(kBlock), 0, st, part, ns, p->F, p->W, p->S, pl.nbx, g_c,
                       g_t, g_mu, g_z, g_cx, g_cy, ncol, g_kappa, g_poly, (const double *)nullptr, 0,
                       (const double *)nullptr, (const unsigned *)nullptr, 0u, 0, g_n_index, 0);
    herr = (int)hipGetLastError();
    if (herr) return hip_fail(herr, "reduce_bwd_kernel launch");
    return TL_OK;
}

int tl_trace_bwd_from_outputs(const tl_problem *p, const float *gx, const float *gy, const float *gcx, const float *gcy,
                              const double *g_moments, const float *x_fwd, const float *y_fwd, const float *cx_fwd,
                              const float *cy_fwd, const uint8_t *ok_fwd, const double *moments_fwd, float *g_c,
                              float *g_t, float *g_mu, float *g_z, float *g_cx, float *g_cy, float *g_kappa,
                              float *g_poly, float *g_x_in, float *g_y_in, void *workspace, size_t workspace_bytes,
                              void *stream)
{
    int rc = check_problem(p);
    if (rc) return rc;
    if (!g_c || !g_t || !g_mu || !g_z || !g_cx || !g_cy) return fail(TL_EINVAL, "a parameter-gradient output is NULL");
    // p->cond_flags (written by the forward) is the ok bytes with the conditioning flag: it is read instead of ok_fwd
    if (p->cond_flags) ok_fwd = p->cond_flags;
    if (p->P > 0 && (!x_fwd || !y_fwd || !cx_fwd || !cy_fwd || !ok_fwd))
        return fail(TL_EINVAL, "the forward outputs x, y, cx, cy, ok are required");
    if (!p->allow_backward)
        return fail(TL_EINVAL, "tl_trace_bwd_from_outputs: allow_backward_rays only");
    const bool hits_ok = p->surf_kind == nullptr || (p->asph_hits != nullptr && p->asph_hit_slots > 0);
    // the penalty term is walked back by the unrolled kernels only (3..20 rows, aspheric rows with stored hits);
    // other lenses take the checkpoint kernel for every ray
    if (p->aggregate && !(tl_walk_unrolled(p->S, p->P) && hits_ok))
        return tl_trace_bwd(p, gx, gy, gcx, gcy, g_moments, nullptr, g_c, g_t, g_mu, g_z, g_cx, g_cy, g_kappa, g_poly,
                            nullptr, g_x_in, g_y_in, workspace, workspace_bytes, stream);
    if ((g_kappa || g_poly) && !p->surf_kind) return fail(TL_EINVAL, "g_kappa / g_poly need aspheric rows (surf_kind)");
    if (p->surf_kind && (!g_kappa || !g_poly)) return fail(TL_EINVAL, "aspheric rows need g_kappa and g_poly outputs");
    if (p->P == 0) return tl_trace_bwd(p, gx, gy, gcx, gcy, g_moments, nullptr, g_c, g_t, g_mu, g_z, g_cx, g_cy, g_kappa, g_poly,
                                       nullptr, g_x_in, g_y_in, workspace, workspace_bytes, stream);
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipSetDevice(p->device);
    if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
    const Plan pl = plan_bwd(p);
    // the fallback is launched too and normally retires at once: a quarter of the blocks keeps that idle launch cheap
    const Plan pk = make_plan(p->P, rows_bfw(p), 1024, 256);
    const bool asph = p->surf_kind != nullptr;
    const int ncol = (asph ? 8 : 3) * p->S + 3, ns = tl_bwd_bucket(p->S), ncol_ck = tl_bwd_row(ns, asph);
    const size_t rows = (size_t)rows_bfw(p) * pl.nbx, rows_ck = (size_t)rows_bfw(p) * pk.nbx;
    const size_t need_inv = rows * ncol * sizeof(double), need_ck = rows_ck * ncol_ck * sizeof(double);
    const size_t need_scan = tl_scanmap_bytes(rows_bfw(p), ((int64_t)p->P + kBlock - 1) / kBlock);
    if (!workspace || workspace_bytes < tl_workspace_bytes(p) || tl_workspace_bytes(p) < need_inv + need_ck + need_scan + 256)
        return fail(TL_EWORKSPACE, "workspace too small for tl_trace_bwd_from_outputs");
    double *part = (double *)workspace, *part_ck = part + rows * ncol;
    unsigned *poison = poison_word(p, workspace);      // the walk-back writes `token` here on a non-finite adjoint
    static std::atomic<uint32_t> calls{0};
    uint32_t token = (calls.fetch_add(1u) + 1u) * 0x9E3779B1u;         // unique per call, nothing like stale data
    if (token == 0u) token = 1u;
    // (Recorded into a HIP graph the call is replayed with the SAME token every time; the word is cleared by the
    //  forward call of the same step, see reduce_moments_kernel, so a poisoned replay does not stick.)
    int herr = (p->mode == TL_MODE_FAST)
                   ? tl_fast::api_bwd_inv(*p, gx, gy, gcx, gcy, g_moments, x_fwd, y_fwd, cx_fwd, cy_fwd, ok_fwd,
                                          moments_fwd, g_x_in, g_y_in, part, part_ck, poison, token, pl.nbx, pl.R, pk.nbx, pk.R, st)
                   : tl_strict::api_bwd_inv(*p, gx, gy, gcx, gcy, g_moments, x_fwd, y_fwd, cx_fwd, cy_fwd, ok_fwd,
                                            moments_fwd, g_x_in, g_y_in, part, part_ck, poison, token, pl.nbx, pl.R, pk.nbx, pk.R, st);
    if (herr) return hip_fail(herr, "trace_bwd_inv_kernel launch");
    const int nout = 2 * p->S + p->W * p->S + 1 + 2 * p->F + (asph ? 5 * p->S : 0);
    hipLaunchKernelGGL(reduce_bwd_kernel, dim3(nout, lenses(p)), dim3(kBlock), 0, st, part, p->S, p->F, p->W, p->S, pl.nbx, g_c, g_t,
                       g_mu, g_z, g_cx, g_cy, ncol, g_kappa, g_poly, (const double *)part_ck, ns,
                       moments_fwd, (const unsigned *)poison, token, pk.nbx, (float *)nullptr,
                       (p->aggregate ? 1 : 0) | (p->cond_flags ? 2 : 0));
    herr = (int)hipGetLastError();
    if (herr) return hip_fail(herr, "reduce_bwd_kernel launch");
    return TL_OK;
}

int tl_spot_moments(int32_t device, int32_t F, int32_t P, int32_t W, const float *x, const float *y,
                    const uint8_t *ok, int64_t s_f, int64_t s_p, int64_t s_w, double *moments, void *workspace,
                    size_t workspace_bytes, void *stream)
{
    if (F < 1 || W < 1 || P < 1 || !y || !ok || !moments) return fail(TL_EINVAL, "tl_spot_moments: bad argument");
    if ((int64_t)F * W > 65535) return fail(TL_EINVAL, "F*W exceeds 65535");
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
    const Plan pl = make_plan(P, F * W, 4096, 64);
    const size_t need = (size_t)F * W * pl.nbx * TL_NMOM * sizeof(double);
    if (!workspace || workspace_bytes < need) return fail(TL_EWORKSPACE, "workspace too small for tl_spot_moments");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(spot_moments_kernel, dim3(pl.nbx, F * W), dim3(kBlock), 0, st, P, W, x, y, ok, s_f, s_p, s_w,
                       (double *)workspace, pl.R);
    int herr = (int)hipGetLastError();
    if (herr) return hip_fail(herr, "spot_moments_kernel launch");
    hipLaunchKernelGGL(reduce_moments_kernel, dim3(F * TL_NMOM), dim3(kBlock), 0, st, (const double *)workspace,
                       moments, W, pl.nbx, (unsigned *)nullptr);
    herr = (int)hipGetLastError();
    if (herr) return hip_fail(herr, "reduce_moments_kernel launch");
    return TL_OK;
}

int tl_spot_rms(int32_t device, int32_t B, int32_t F, double n_per_field, const double *moments, float *rms,
                double *d_moments, void *stream)
{
    if (B < 1 || F < 1 || !(n_per_field > 0.0) || !moments || !rms || !d_moments)
        return fail(TL_EINVAL, "tl_spot_rms: bad argument");
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
    hipLaunchKernelGGL(spot_rms_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, moments, F, n_per_field, rms, d_moments);
    const int herr = (int)hipGetLastError();
    if (herr) return hip_fail(herr, "spot_rms_kernel launch");
    return TL_OK;
}

int tl_unsup_loss(int32_t device, int32_t B, int32_t F, double n_per_field, const double *moments, const double *n_sequence,
                  double n_sequence_all, float penalty_rate, float *loss, float *rms, float *penalty, double *d_rms,
                  void *stream)
{
    if (B < 1 || F < 1 || !(n_per_field > 0.0) || !moments || !loss || !rms || !penalty || !d_rms ||
        (!n_sequence && !(n_sequence_all > 0.0)))
        return fail(TL_EINVAL, "tl_unsup_loss: bad argument");
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
    hipLaunchKernelGGL(unsup_loss_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, moments, F, n_per_field, n_sequence,
                       n_sequence_all, penalty_rate, loss, rms, penalty, d_rms);
    const int herr = (int)hipGetLastError();
    if (herr) return hip_fail(herr, "unsup_loss_kernel launch");
    return TL_OK;
}

int tl_unsup_loss_bwd(int32_t device, int32_t B, int32_t F, const double *d_rms, const float *g_loss, const float *g_rms,
                      const float *g_penalty, int32_t g_stride, const double *n_sequence, double n_sequence_all,
                      float penalty_rate, double *g_moments, void *stream)
{
    if (B < 1 || F < 1 || !d_rms || !g_moments || (!g_loss && !g_rms && !g_penalty) || (g_stride != 0 && g_stride != 1) ||
        (!n_sequence && !(n_sequence_all > 0.0)))
        return fail(TL_EINVAL, "tl_unsup_loss_bwd: bad argument");
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
    const int total = B * F * TL_NMOM;
    hipLaunchKernelGGL(unsup_loss_bwd_kernel, dim3((total + kBlock - 1) / kBlock), dim3(kBlock), 0, (hipStream_t)stream, d_rms, F,
                       total, g_loss, g_rms, g_penalty, g_stride, n_sequence, n_sequence_all, penalty_rate, g_moments);
    const int herr = (int)hipGetLastError();
    if (herr) return hip_fail(herr, "unsup_loss_bwd_kernel launch");
    return TL_OK;
}

int tl_spot_seed(int32_t device, int32_t F, int32_t P, int32_t W, const float *x, const float *y, const uint8_t *ok,
                 int64_t s_f, int64_t s_p, int64_t s_w, const double *g_moments, float *gx, float *gy, void *stream)
{
    if (F < 1 || W < 1 || P < 1 || !y || !ok || !g_moments) return fail(TL_EINVAL, "tl_spot_seed: bad argument");
    if ((int64_t)F * W > 65535) return fail(TL_EINVAL, "F*W exceeds 65535");
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
    hipLaunchKernelGGL(spot_seed_kernel, dim3((P + kBlock - 1) / kBlock, F * W), dim3(kBlock), 0, (hipStream_t)stream,
                       P, W, x, y, ok, s_f, s_p, s_w, g_moments, gx, gy);
    const int herr = (int)hipGetLastError();
    if (herr) return hip_fail(herr, "spot_seed_kernel launch");
    return TL_OK;
}

int tl_pupil_position(int32_t device, int32_t B, int32_t K, const float *c, const float *t, const float *n, float *z,
                      const float *g_z, float *g_c, float *g_t, float *g_n, int32_t mode, void *stream)
{
    if (mode != TL_MODE_STRICT && mode != TL_MODE_FAST) return fail(TL_EINVAL, "unknown mode");
    if (B < 1 || K < 1 || K > TL_MAX_SURFACES || !c || !t || !n) return fail(TL_EINVAL, "tl_pupil_position: bad argument");
    if (!z && !g_z) return fail(TL_EINVAL, "tl_pupil_position: neither z nor g_z given");
    if (g_z && (!g_c || !g_t || !g_n)) return fail(TL_EINVAL, "tl_pupil_position: g_z needs g_c, g_t and g_n");
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
    hipLaunchKernelGGL(pupil_position_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, B, K, c, t, n, z, g_z, g_c, g_t, g_n,
                       mode == TL_MODE_STRICT ? 1 : 0);
    const int herr = (int)hipGetLastError();
    if (herr) return hip_fail(herr, "pupil_position_kernel launch");
    return TL_OK;
}

// The aimed fan in one launch: x = clamp(xp x_scale, -2, 2) (epd / 2), y = clamp(yp y_scale + y_offset, -2, 2) (epd / 2) --
// the op sequence of RayTracer.assemble after ray_aiming (remap, clamp, scale_to_epd: 7 elementwise launches), with its
// rounding points (no contraction).  Output laid out [B,F,W,P]: the trace kernels read consecutive pupil points.
__global__ __launch_bounds__(kBlock) void aim_fan_kernel(int P, int F, int W, const float *__restrict__ xp,
                                                         const float *__restrict__ yp, const float *__restrict__ x_scale,
                                                         const float *__restrict__ y_scale, const float *__restrict__ y_offset,
                                                         const float *__restrict__ epd, float *__restrict__ x_out,
                                                         float *__restrict__ y_out)
{
#pragma clang fp contract(off)
    const int bfw = blockIdx.y, b = bfw / (F * W);
    const int ip = blockIdx.x * kBlock + threadIdx.x;
    if (ip >= P) return;
    const float xs = x_scale[bfw], ys = y_scale[bfw], yo = y_offset[bfw], half = epd[b] / 2.0f;
    float x = xp[ip] * xs;
    float y = yp[ip] * ys;
    y = y + yo;
    x = fminf(fmaxf(x, -2.0f), 2.0f);
    y = fminf(fmaxf(y, -2.0f), 2.0f);
    x_out[(size_t)bfw * P + ip] = x * half;
    y_out[(size_t)bfw * P + ip] = y * half;
}

int tl_aim_fan(int32_t device, int32_t B, int32_t F, int32_t W, int32_t P, const float *xp, const float *yp,
               const float *x_scale, const float *y_scale, const float *y_offset, const float *epd, float *x_out, float *y_out,
               void *stream)
{
    if (B < 1 || F < 1 || W < 1 || P < 1 || !xp || !yp || !x_scale || !y_scale || !y_offset || !epd || !x_out || !y_out)
        return fail(TL_EINVAL, "tl_aim_fan: bad argument");
    if ((int64_t)B * F * W > 65535) return fail(TL_EINVAL, "B*F*W exceeds 65535");
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
    hipLaunchKernelGGL(aim_fan_kernel, dim3((P + kBlock - 1) / kBlock, B * F * W), dim3(kBlock), 0, (hipStream_t)stream, P, F, W, xp,
                       yp, x_scale, y_scale, y_offset, epd, x_out, y_out);
    const int herr = (int)hipGetLastError();
    if (herr) return hip_fail(herr, "aim_fan_kernel launch");
    return TL_OK;
}

int tl_ray_aim(int32_t device, int32_t B, int32_t F, int32_t W, int32_t K, const float *c, const float *t, const float *n,
               const float *n_d, const uint8_t *mask, const float *kappa, const float *poly, const uint8_t *surf_kind,
               const float *z, const float *hfov, const float *fields, const float *epd, int32_t allow_backward,
               float *x_scale, float *y_scale, float *y_offset, void *stream)
{
    if (B < 1 || F < 1 || W < 1 || K < 1 || K > TL_MAX_SURFACES || !c || !t || !n || !n_d || !mask || !z || !hfov || !fields ||
        !epd || !x_scale || !y_scale || !y_offset)
        return fail(TL_EINVAL, "tl_ray_aim: bad argument");
    if ((surf_kind != nullptr) != (kappa != nullptr) || (surf_kind != nullptr) != (poly != nullptr))
        return fail(TL_EINVAL, "tl_ray_aim: surf_kind, kappa and poly must be given together (or all NULL)");
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
    const int nthr = B * F * W;
    hipLaunchKernelGGL(ray_aim_kernel, dim3((nthr + 63) / 64), dim3(64), 0, (hipStream_t)stream, B, F, W, K, c, t, n, n_d, mask,
                       kappa, poly, surf_kind, z, hfov, fields, epd, allow_backward, x_scale, y_scale, y_offset);
    const int herr = (int)hipGetLastError();
    if (herr) return hip_fail(herr, "ray_aim_kernel launch");
    return TL_OK;
}

// ---------------------------------------------------------------- double precision (RayTracer(double_precision=True))
static int check_f64(const tl_problem *p)
{
    int rc = check_problem(p);
    if (rc) return rc;
    if (p->aggregate) return fail(TL_EINVAL, "the double-precision trace has no penalty term (aggregate)");
    if (p->P < 1) return fail(TL_EINVAL, "the double-precision trace needs P >= 1");
    return TL_OK;
}

size_t tl_workspace_bytes_f64(const tl_problem *p)
{
    if (!p || p->F < 1 || p->W < 1 || p->S < 1 || p->P < 0) return 0;
    const Plan pb = plan_bwd(p), pf = plan_fwd(p);
    const size_t rows = (size_t)rows_bfw(p);
    const size_t a = rows * pf.nbx * TL_NMOM * sizeof(double), b = rows * pb.nbx * (size_t)(8 * p->S + 3) * sizeof(double);
    return (a > b ? a : b) + 256;
}

int tl_trace_fwd_f64(const tl_problem *p, double *x, double *y, double *cx, double *cy, uint8_t *ok, uint8_t *back,
                     double *moments, void *workspace, size_t workspace_bytes, void *stream)
{
    int rc = check_f64(p);
    if (rc) return rc;
    hipError_t e = hipSetDevice(p->device);
    if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
    const Plan pl = plan_fwd(p);
    double *part = nullptr;
    if (moments) {
        const size_t need = (size_t)rows_bfw(p) * pl.nbx * TL_NMOM * sizeof(double);
        if (!workspace || workspace_bytes < need) return fail(TL_EWORKSPACE, "workspace too small for tl_trace_fwd_f64");
        part = (double *)workspace;
    }
    int herr = tl_f64::launch_fwd(*p, x, y, cx, cy, ok, back, part, pl.nbx, (hipStream_t)stream);
    if (herr) return hip_fail(herr, "tl_f64::fwd_kernel launch");
    if (moments) {
        herr = tl_f64::launch_reduce_moments(*p, part, moments, pl.nbx, (hipStream_t)stream);
        if (herr) return hip_fail(herr, "tl_f64::reduce_moments_kernel launch");
    }
    return TL_OK;
}

int tl_trace_bwd_f64(const tl_problem *p, const double *gx, const double *gy, const double *gcx, const double *gcy,
                     const double *g_moments, double *g_c, double *g_t, double *g_mu, double *g_z, double *g_cx, double *g_cy,
                     double *g_kappa, double *g_poly, double *g_x_in, double *g_y_in, void *workspace, size_t workspace_bytes,
                     void *stream)
{
    int rc = check_f64(p);
    if (rc) return rc;
    if (!g_c || !g_t || !g_mu || !g_z || !g_cx || !g_cy) return fail(TL_EINVAL, "a parameter-gradient output is NULL");
    if (p->surf_kind && (!g_kappa || !g_poly)) return fail(TL_EINVAL, "aspheric rows need g_kappa and g_poly outputs");
    hipError_t e = hipSetDevice(p->device);
    if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
    const Plan pl = plan_bwd(p);
    const size_t need = (size_t)rows_bfw(p) * pl.nbx * (size_t)(8 * p->S + 3) * sizeof(double);
    if (!workspace || workspace_bytes < need) return fail(TL_EWORKSPACE, "workspace too small for tl_trace_bwd_f64");
    const int herr = tl_f64::launch_bwd(*p, gx, gy, gcx, gcy, g_moments, g_x_in, g_y_in, (double *)workspace, pl.nbx, g_c, g_t, g_mu,
                                        g_z, g_cx, g_cy, g_kappa, g_poly, (hipStream_t)stream);
    if (herr) return hip_fail(herr, "tl_f64::bwd_kernel launch");
    return TL_OK;
}

int tl_selftest_arith(int32_t device, int32_t mode, const float *a, const float *b, int64_t n, float *quot, float *root,
                      void *stream)
{
    if (!a || !b || !quot || !root || n < 1) return fail(TL_EINVAL, "tl_selftest_arith: bad argument");
    if (mode != TL_MODE_STRICT && mode != TL_MODE_FAST) return fail(TL_EINVAL, "unknown mode");
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
    const int herr = mode == TL_MODE_FAST ? tl_fast::api_selftest_arith(a, b, n, quot, root, (hipStream_t)stream)
                                          : tl_strict::api_selftest_arith(a, b, n, quot, root, (hipStream_t)stream);
    if (herr) return hip_fail(herr, "selftest_arith_kernel launch");
    return TL_OK;
}

}  // extern "C"
