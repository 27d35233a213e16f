// tl_f64.hip -- the double-precision twin of the trace: RayTracer(double_precision=True) (ray_tracing_lite.py:82-84, which
// crashes in the reference: Specs / Lens have no .double(); SURVEY Appendix B3 "implement properly").
//
// One generic forward kernel and one checkpoint backward kernel, everything in fp64, one ray per lane, rolled loops, the
// per-surface states of the backward in a per-thread array (scratch).  NOT tuned: this path exists for reference-quality
// numbers on the GPU (fp64 gradients at full fan size, the fp64 column of the gradient checks), not for speed -- the fp32
// kernels of tl_kernels.inc are the product's hot path.  Arithmetic: SURVEY Appendix A op for op (:525-571, 594-675) for
// spherical rows, Newton to convergence from the closed-form conic hit for aspheric rows (the oracle's definition), adjoint
// = the same chain rule as step_vals / step_adjoint / asph_vals / asph_adjoint of the fp32 kernels.  No penalty term, no
// optical path length here (the C entry points refuse them).
#include "tl_common.h"

namespace tl_f64 {

constexpr int kBlock = 256;
constexpr int kWaves = kBlock / 64;
constexpr double kEps = 1e-6;

struct Ray { double x, y, z, cx, cy, cz; };
struct Adj { double x, y, z, cx, cy, cz; };

// the float-typed pointers of tl_problem point at doubles when the problem is handed to the *_f64 entry points
struct View {
    const double *x_in, *y_in, *z, *cx, *cy, *c, *t, *mu, *kappa, *poly;
    const uint8_t *mask, *kind;
};

__device__ __forceinline__ View lens_view(const tl_problem &p, const int b)
{
    View v;
    const int64_t S = p.S, W = p.W;
    v.x_in = (const double *)p.x_in + b * p.xs_b;
    v.y_in = (const double *)p.y_in + b * p.ys_b;
    v.z = (const double *)p.z + b;
    v.cx = (const double *)p.cx + b * p.cx_stride_b;
    v.cy = (const double *)p.cy + b * p.cy_stride_b;
    v.c = (const double *)p.c + b * S;
    v.t = (const double *)p.t + b * S;
    v.mu = (const double *)p.mu + b * W * S;
    v.mask = p.mask + b * S;
    v.kind = p.surf_kind ? p.surf_kind + b * S : nullptr;
    v.kappa = p.surf_kind ? (const double *)p.kappa + b * S : nullptr;
    v.poly = p.surf_kind ? (const double *)p.poly + b * S * TL_MAX_POLY : nullptr;
    return v;
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ------------------------------------------------------------------ one row forwards (lean form: `ok` is sticky, dead
// lanes compute garbage that is masked at the image plane, exactly as in trace_fwd_kernel)
struct Asph { double c, kappa, a0, a1, a2, a3; };

__device__ __forceinline__ void sag_terms(const Asph &A, const double rho, double &sag, double &dsag, bool &bad)
{
    const double q2 = 1.0 - (1.0 + A.kappa) * A.c * A.c * rho;
    bad = q2 - kEps < 0.0;
    const double q = sqrt(bad ? 1.0 : q2);
    sag = A.c * rho / (1.0 + q) + rho * rho * (A.a0 + rho * (A.a1 + rho * (A.a2 + rho * A.a3)));
    dsag = A.c / (2.0 * q) + rho * (2.0 * A.a0 + rho * (3.0 * A.a1 + rho * (4.0 * A.a2 + rho * (5.0 * A.a3))));
}

__device__ void step_fwd(Ray &r, bool &ok, bool &back, const double c, const double t, const double mu, const bool asph,
                         const Asph &A, const bool chk, const bool allow_back, double &min_cos2)
{
    double X, Y, Z, dz, cos_i, nx, ny, nz;
    bool miss;
    if (!asph) {
        const double e = -((r.x * r.cx + r.y * r.cy) + r.z * r.cz);
        const double mz = r.z + e * r.cz;
        const double m2 = ((r.x * r.x + r.y * r.y) + r.z * r.z) - e * e;
        const double tmp = c * m2 - 2.0 * mz;
        const double cos2 = r.cz * r.cz - c * tmp;
        miss = cos2 - kEps < 0.0;
        cos_i = sqrt(miss ? 1.0 : cos2);
        const double d = e + tmp / (r.cz + cos_i);
        dz = d * r.cz;
        X = r.x + d * r.cx; Y = r.y + d * r.cy; Z = r.z + dz;
        nx = -c * X; ny = -c * Y; nz = 1.0 - c * Z;
        min_cos2 = fmin(min_cos2, cos2);
    } else {
        const double K = 1.0 + A.kappa;
        const double e = -((r.x * r.cx + r.y * r.cy) + K * (r.z * r.cz));
        const double dd = (r.cx * r.cx + r.cy * r.cy) + K * (r.cz * r.cz);
        const double rr = (r.x * r.x + r.y * r.y) + K * (r.z * r.z);
        const double bq = A.c * e + r.cz, cq = A.c * rr - 2.0 * r.z;
        const double disc = bq * bq - (A.c * dd) * cq;
        const bool miss0 = disc - kEps < 0.0;
        double s = miss0 ? 0.0 : cq / (bq + sqrt(miss0 ? 1.0 : disc));
        double sag = 0.0, dsag = 0.0, F = 0.0, rho = 0.0;
        bool bad = false;
        for (int it = 0; it < 12; ++it) {                       // (per-lane exit: this path is not wave-tuned)
            X = r.x + s * r.cx; Y = r.y + s * r.cy; Z = r.z + s * r.cz;
            rho = X * X + Y * Y;
            sag_terms(A, rho, sag, dsag, bad);
            F = Z - sag;
            if (miss0 || bad || fabs(F) <= 1e-14 * (1.0 + fabs(Z))) break;
            s -= F / (r.cz - dsag * (2.0 * (X * r.cx + Y * r.cy)));
        }
        miss = miss0 || bad || !(fabs(F) <= 1e-12 * (1.0 + fabs(Z)));
        dz = s * r.cz;
        const double m = 2.0 * dsag, inv_n = 1.0 / sqrt(1.0 + m * m * rho);
        nx = -(m * X) * inv_n; ny = -(m * Y) * inv_n; nz = inv_n;
        cos_i = (r.cx * nx + r.cy * ny) + r.cz * nz;
        min_cos2 = fmin(min_cos2, cos_i * cos_i);
    }
    ok = ok && !miss;
    const double cos2_t = 1.0 - (mu * mu) * (1.0 - cos_i * cos_i);
    const bool tir = cos2_t - kEps < 0.0;
    const double cos_t = sqrt(tir ? 1.0 : cos2_t);
    const double g = cos_t - mu * cos_i;
    const double cx3 = mu * r.cx + g * nx, cy3 = mu * r.cy + g * ny;
    const double czsq = 1.0 - (cx3 * cx3 + cy3 * cy3);
    const bool fail = tir || (czsq - kEps < 0.0);
    min_cos2 = fmin(min_cos2, cos2_t);
    {
        const bool hit = chk && dz < 0.0 && ok;                  // :626-632 (ok after the miss test, before the refraction test)
        back = back || (allow_back && hit);
        ok = ok && !(hit && !allow_back);
    }
    ok = ok && !fail;
    r.x = X; r.y = Y; r.z = Z - t;
    r.cx = cx3; r.cy = cy3; r.cz = sqrt(fail ? 1.0 : czsq);
}

__device__ __forceinline__ Asph load_asph(const View &v, const int k)
{
    Asph A;
    A.c = v.c[k]; A.kappa = 0.0; A.a0 = A.a1 = A.a2 = A.a3 = 0.0;
    if (v.kind && v.kind[k]) {
        A.kappa = v.kappa[k];
        A.a0 = v.poly[4 * k]; A.a1 = v.poly[4 * k + 1]; A.a2 = v.poly[4 * k + 2]; A.a3 = v.poly[4 * k + 3];
    }
    return A;
}

#define TLF_ROW(PA_)                                                              \
    const int fw = blockIdx.y;                                                    \
    const int gf = fw / (PA_).W, w = fw - gf * (PA_).W;                           \
    const int B_ = (PA_).B > 0 ? (PA_).B : 1;                                     \
    (void)B_;                                                                     \
    const int lens_ = gf / (PA_).F, f = gf - lens_ * (PA_).F;                     \
    const View v = lens_view(PA_, lens_)

#define TLF_CHUNKS(P_)                                                            \
    const int64_t nchunk_ = ((int64_t)(P_) + kBlock - 1) / kBlock;                \
    const int64_t c0 = (int64_t)blockIdx.x * nchunk_ / gridDim.x;                 \
    const int R = (int)((int64_t)(blockIdx.x + 1) * nchunk_ / gridDim.x - c0)

// ------------------------------------------------------------------ forward
__global__ __launch_bounds__(kBlock) void fwd_kernel(const tl_problem p, double *__restrict__ ox, double *__restrict__ oy,
                                                     double *__restrict__ ocx, double *__restrict__ ocy,
                                                     uint8_t *__restrict__ ook, uint8_t *__restrict__ oback,
                                                     double *__restrict__ part)
{
    TLF_ROW(p);
    TLF_CHUNKS(p.P);
    const int S = p.S;
    const double z0 = v.z[0], cx0 = v.cx[f * p.cx_stride], cy0 = v.cy[f * p.cy_stride];
    const double cz0 = sqrt((1.0 - cx0 * cx0) - cy0 * cy0);
    const double *mu_w = v.mu + (size_t)w * S;
    const bool allow_back = p.allow_backward != 0;
    const int64_t xoff = f * p.xs_f + w * p.xs_w, yoff = f * p.ys_f + w * p.ys_w;
    double m[TL_NMOM];
    for (int j = 0; j < TL_NMOM; ++j) m[j] = 0.0;
    for (int r = 0; r < R; ++r) {
        const int64_t ip = (c0 + r) * kBlock + threadIdx.x;
        if (ip >= p.P) continue;
        Ray ray;
        ray.x = v.x_in[xoff + ip * p.xs_p]; ray.y = v.y_in[yoff + ip * p.ys_p];
        ray.z = z0; ray.cx = cx0; ray.cy = cy0; ray.cz = cz0;
        bool ok = true, back = false;
        double min_cos2 = 1.0;
        for (int k = 0; k < S; ++k) {
            const bool asph = v.kind && v.kind[k];
            step_fwd(ray, ok, back, v.c[k], v.t[k], mu_w[k], asph, load_asph(v, k), k > 0 && v.mask[k - 1], allow_back, min_cos2);
        }
        const double dz = -ray.z, dist = dz / ray.cz;
        const bool alive = ok;
        const double xo = ok ? ray.x + dist * ray.cx : 0.0, yo = ok ? ray.y + dist * ray.cy : 0.0;
        {
            const bool hit = dz < 0.0 && ok && v.mask[S - 1];
            if (allow_back) back = back || hit; else ok = ok && !hit;
        }
        const int64_t o = (int64_t)fw * p.P + ip;
        if (ox) ox[o] = xo;
        if (oy) oy[o] = yo;
        if (ocx) ocx[o] = alive ? ray.cx : 0.0;
        if (ocy) ocy[o] = alive ? ray.cy : 0.0;
        if (ook) ook[o] = ok ? 1 : 0;
        if (oback) oback[o] = back ? 1 : 0;
        const double yk = ok ? yo : 0.0, xk = ok ? xo : 0.0;
        m[0] += yo; m[1] += yk; m[2] += yk * yk; m[3] += ok ? 1.0 : 0.0;
        m[4] += xo; m[5] += xk; m[6] += xk * xk; m[7] += back ? 1.0 : 0.0;
        m[9] += (alive && min_cos2 < 0.01) ? 1.0 : 0.0;
    }
    if (part) {
        __shared__ double red[kWaves][TL_NMOM];
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        for (int j = 0; j < TL_NMOM; ++j) {
            const double s = wave_sum(m[j]);
            if (lane == 0) red[wv][j] = s;
        }
        __syncthreads();
        if (threadIdx.x < TL_NMOM) {
            double s = 0.0;
            for (int q = 0; q < kWaves; ++q) s += red[q][threadIdx.x];
            part[(size_t)threadIdx.x * ((size_t)gridDim.y * gridDim.x) + ((size_t)fw * gridDim.x + blockIdx.x)] = s;
        }
    }
}

// ------------------------------------------------------------------ adjoint of one row (state entering it, adjoint of
// the state leaving it in / out); g5 = d/d(kappa, a4, a6, a8, a10) for an aspheric row
__device__ void step_bwd(const Ray &in, const double zout, const double c, const double t, const double mu, const bool asph,
                         const Asph &A, Adj &a, double &g_c, double &g_t, double &g_mu, double g5[5])
{
    g5[0] = g5[1] = g5[2] = g5[3] = g5[4] = 0.0;
    if (!asph) {
        // ---- forward intermediates (step_vals)
        const double e = -((in.x * in.cx + in.y * in.cy) + in.z * in.cz);
        const double mz = in.z + e * in.cz;
        const double m2 = ((in.x * in.x + in.y * in.y) + in.z * in.z) - e * e;
        const double tmp = c * m2 - 2.0 * mz;
        const double cos2 = in.cz * in.cz - c * tmp;
        const double cos_i = sqrt(cos2), r_i = 1.0 / cos_i;
        const double inv_den = 1.0 / (in.cz + cos_i);
        const double q = tmp * inv_den, d = e + q;
        const double x1 = in.x + d * in.cx, y1 = in.y + d * in.cy;
        const double s2 = 1.0 - cos_i * cos_i;
        const double cos2_t = 1.0 - (mu * mu) * s2;
        const double cos_t = sqrt(cos2_t), r_t = 1.0 / cos_t;
        const double g = cos_t - mu * cos_i, gc = g * c;
        const double cx3 = mu * in.cx - gc * x1, cy3 = mu * in.cy - gc * y1;
        const double r_z = 1.0 / sqrt(1.0 - (cx3 * cx3 + cy3 * cy3));
        // ---- reverse (step_adjoint<false>)
        g_t = -a.z;
        const double b_cz_r = a.cz * r_z;
        const double b_cx3 = a.cx - cx3 * b_cz_r, b_cy3 = a.cy - cy3 * b_cz_r;
        double b_mu = b_cx3 * in.cx + b_cy3 * in.cy;
        const double b_gc = -(b_cx3 * x1 + b_cy3 * y1);
        const double b_x1 = a.x - gc * b_cx3, b_y1 = a.y - gc * b_cy3;
        double b_c = b_gc * g;
        const double b_g = b_gc * c;
        const double wv = b_g * r_t;                                   // 2 * adjoint of cos2_t
        const double mw = mu * wv;
        b_mu -= b_g * cos_i;
        b_mu -= s2 * mw;
        double b_cos_i = mu * (mw * cos_i - b_g);
        const double b_d = (b_x1 * in.cx + b_y1 * in.cy) + a.z * in.cz;
        double bx = b_x1, by = b_y1, bz = a.z;
        double bcx = mu * b_cx3 + b_x1 * d, bcy = mu * b_cy3 + b_y1 * d, bcz = a.z * d;
        double b_tmp = b_d * inv_den;
        const double b_den = -(b_tmp * q);
        bcz += b_den;
        b_cos_i += b_den;
        const double b2 = b_cos_i * r_i;                               // 2 * adjoint of cos2
        const double b_cos2 = 0.5 * b2;
        bcz += in.cz * b2;
        b_c -= tmp * b_cos2;
        b_tmp -= c * b_cos2;
        b_c += m2 * b_tmp;
        const double b_mz = -2.0 * b_tmp;
        const double b_m2_x2 = -(c * b_mz);
        bx += in.x * b_m2_x2; by += in.y * b_m2_x2; bz += in.z * b_m2_x2 + b_mz;
        const double b_e = (b_d + b_mz * in.cz) - e * b_m2_x2;
        bcz += b_mz * e;
        a.x = bx - b_e * in.cx; a.y = by - b_e * in.cy; a.z = bz - b_e * in.cz;
        a.cx = bcx - b_e * in.x; a.cy = bcy - b_e * in.y; a.cz = bcz - b_e * in.z;
        g_c = b_c; g_mu = b_mu;
        return;
    }
    // ---- aspheric row (asph_vals + asph_adjoint<false>)
    const double s = (zout + t - in.z) / in.cz;
    const double X = in.x + s * in.cx, Y = in.y + s * in.cy;
    const double rho = X * X + Y * Y;
    const double K = 1.0 + A.kappa, c2 = A.c * A.c, c3 = c2 * A.c;
    const double q2 = 1.0 - (K * c2) * rho;
    const double q = sqrt(q2), rq = 1.0 / q, rq3 = rq * rq * rq;
    const double rho2 = rho * rho, rho3 = rho2 * rho, rho4 = rho2 * rho2;
    const double dsag = 0.5 * A.c * rq + rho * (2.0 * A.a0 + rho * (3.0 * A.a1 + rho * (4.0 * A.a2 + rho * (5.0 * A.a3))));
    const double sag2 = 0.25 * K * c3 * rq3 + (2.0 * A.a0 + rho * (6.0 * A.a1 + rho * (12.0 * A.a2 + rho * (20.0 * A.a3))));
    const double m = 2.0 * dsag, inv_n = 1.0 / sqrt(1.0 + (m * m) * rho);
    const double nx = -(m * X) * inv_n, ny = -(m * Y) * inv_n, nz = inv_n;
    const double cos_i = (in.cx * nx + in.cy * ny) + in.cz * nz;
    const double s2 = 1.0 - cos_i * cos_i;
    const double cos2_t = 1.0 - (mu * mu) * s2;
    const double cos_t = sqrt(cos2_t), r_t = 1.0 / cos_t;
    const double g = cos_t - mu * cos_i;
    const double cx3 = mu * in.cx + g * nx, cy3 = mu * in.cy + g * ny;
    const double r_z = 1.0 / sqrt(1.0 - (cx3 * cx3 + cy3 * cy3));
    g_t = -a.z;
    const double b_czsq = a.cz * (0.5 * r_z);
    const double b_cx3 = a.cx - 2.0 * cx3 * b_czsq, b_cy3 = a.cy - 2.0 * cy3 * b_czsq;
    double b_mu = b_cx3 * in.cx + b_cy3 * in.cy;
    const double b_g = b_cx3 * nx + b_cy3 * ny;
    double b_nx = g * b_cx3, b_ny = g * b_cy3;
    double bcx = mu * b_cx3, bcy = mu * b_cy3;
    const double b_cos2_t = b_g * (0.5 * r_t);
    b_mu -= b_g * cos_i + 2.0 * mu * s2 * b_cos2_t;
    const double b_cos_i = -b_g * mu + 2.0 * (mu * mu) * cos_i * b_cos2_t;
    bcx += b_cos_i * nx; bcy += b_cos_i * ny;
    double bcz = b_cos_i * nz;
    b_nx += b_cos_i * in.cx; b_ny += b_cos_i * in.cy;
    const double b_nz = b_cos_i * in.cz;
    const double b_invn = -m * (b_nx * X + b_ny * Y) + b_nz;
    double b_m = -inv_n * (b_nx * X + b_ny * Y);
    double bX = a.x - (m * inv_n) * b_nx, bY = a.y - (m * inv_n) * b_ny;
    const double b_n2 = -0.5 * (inv_n * inv_n * inv_n) * b_invn;
    b_m += 2.0 * m * rho * b_n2;
    double b_rho = (m * m) * b_n2;
    const double b_dsag = 2.0 * b_m;
    b_rho += sag2 * b_dsag;
    double b_c = b_dsag * (0.5 * rq3);
    double b_kap = b_dsag * (0.25 * c3 * rho * rq3);
    double b_a0 = b_dsag * (2.0 * rho), b_a1 = b_dsag * (3.0 * rho2), b_a2 = b_dsag * (4.0 * rho3), b_a3 = b_dsag * (5.0 * rho4);
    bX += 2.0 * X * b_rho; bY += 2.0 * Y * b_rho;
    const double bZ = a.z;
    double bx = bX, by = bY, bz = bZ;
    bcx += bX * s; bcy += bY * s; bcz += bZ * s;
    const double b_s = (bX * in.cx + bY * in.cy) + bZ * in.cz;
    const double Fp = in.cz - dsag * (2.0 * (X * in.cx + Y * in.cy));
    const double lam = -b_s / Fp;
    const double lx = -lam * dsag * (2.0 * X), ly = -lam * dsag * (2.0 * Y);
    bz += lam; bcz += lam * s;
    bx += lx; by += ly; bcx += lx * s; bcy += ly * s;
    const double r1q = 1.0 / (1.0 + q);
    b_c -= lam * rho * rq * r1q;
    b_kap -= lam * (0.5 * c3 * rho2 * rq * r1q * r1q);
    b_a0 -= lam * rho2; b_a1 -= lam * rho3; b_a2 -= lam * rho4; b_a3 -= lam * (rho4 * rho);
    a.x = bx; a.y = by; a.z = bz; a.cx = bcx; a.cy = bcy; a.cz = bcz;
    g_c = b_c; g_mu = b_mu;
    g5[0] = b_kap; g5[1] = b_a0; g5[2] = b_a1; g5[3] = b_a2; g5[4] = b_a3;
}

// ------------------------------------------------------------------ backward (checkpoint algorithm)
// part: [column][block] doubles, columns g_c[S] | g_t[S] | g_mu[S] | g_z g_cx g_cy | g_kappa[S] | g_poly[S][4]
__global__ __launch_bounds__(kBlock) void bwd_kernel(const tl_problem p, const double *__restrict__ gx,
                                                     const double *__restrict__ gy, const double *__restrict__ gcx,
                                                     const double *__restrict__ gcy, const double *__restrict__ gmom,
                                                     double *__restrict__ gxin, double *__restrict__ gyin,
                                                     double *__restrict__ part)
{
    TLF_ROW(p);
    TLF_CHUNKS(p.P);
    const int S = p.S;
    const double z0 = v.z[0], cx0 = v.cx[f * p.cx_stride], cy0 = v.cy[f * p.cy_stride];
    const double cz0 = sqrt((1.0 - cx0 * cx0) - cy0 * cy0);
    const double *mu_w = v.mu + (size_t)w * S;
    const bool allow_back = p.allow_backward != 0;
    const int64_t xoff = f * p.xs_f + w * p.xs_w, yoff = f * p.ys_f + w * p.ys_w;
    double gm0 = 0, gm1 = 0, gm2 = 0, gm4 = 0, gm5 = 0, gm6 = 0;
    if (gmom) {
        const double *q = gmom + (size_t)gf * TL_NMOM;
        gm0 = q[0]; gm1 = q[1]; gm2 = q[2]; gm4 = q[4]; gm5 = q[5]; gm6 = q[6];
    }
    double acc[8 * TL_MAX_SURFACES + 4];                      // g_c | g_t | g_mu | g_z g_cx g_cy g_cz | g_kappa | g_poly
    const int ncol = 8 * S + 3;
    for (int j = 0; j < 8 * S + 4; ++j) acc[j] = 0.0;
    double *a_c = acc, *a_t = acc + S, *a_mu = acc + 2 * S, *a_e = acc + 3 * S, *a_k = acc + 3 * S + 4, *a_p = acc + 4 * S + 4;
    for (int r = 0; r < R; ++r) {
        const int64_t ip = (c0 + r) * kBlock + threadIdx.x;
        if (ip >= p.P) continue;
        Ray ray;
        ray.x = v.x_in[xoff + ip * p.xs_p]; ray.y = v.y_in[yoff + ip * p.ys_p];
        ray.z = z0; ray.cx = cx0; ray.cy = cy0; ray.cz = cz0;
        bool ok = true, back = false;
        double mc = 1.0;
        Ray ck[TL_MAX_SURFACES];
        for (int k = 0; k < S; ++k) {
            ck[k] = ray;
            step_fwd(ray, ok, back, v.c[k], v.t[k], mu_w[k], v.kind && v.kind[k], load_asph(v, k), k > 0 && v.mask[k - 1], allow_back, mc);
        }
        const int64_t o = (int64_t)fw * p.P + ip;
        const bool alive = ok;
        const double dz = -ray.z, dist = dz / ray.cz;
        if (!allow_back) ok = ok && !(dz < 0.0 && ok && v.mask[S - 1]);
        if (!alive) {                                          // failed anywhere: exactly zero gradient
            if (gxin) gxin[o] = 0.0;
            if (gyin) gyin[o] = 0.0;
            continue;
        }
        const double xo = ray.x + dist * ray.cx, yo = ray.y + dist * ray.cy;
        double sx = gx ? gx[o] : 0.0, sy = gy ? gy[o] : 0.0;
        const double scx = gcx ? gcx[o] : 0.0, scy = gcy ? gcy[o] : 0.0;
        if (gmom) {
            const double okd = ok ? 1.0 : 0.0;
            sy += gm0 + okd * (gm1 + 2.0 * yo * gm2);
            sx += gm4 + okd * (gm5 + 2.0 * xo * gm6);
        }
        Adj a;
        const double b_dist = sx * ray.cx + sy * ray.cy;
        a.x = sx; a.y = sy;
        a.cx = scx + sx * dist; a.cy = scy + sy * dist;
        a.z = -b_dist / ray.cz;
        a.cz = -b_dist * dist / ray.cz;
        const double z_last = ray.z;
        for (int k = S - 1; k >= 0; --k) {
            const double zout = (k + 1 < S) ? ck[k + 1].z : z_last;
            double g_c, g_t, g_mu, g5[5];
            step_bwd(ck[k], zout, v.c[k], v.t[k], mu_w[k], v.kind && v.kind[k], load_asph(v, k), a, g_c, g_t, g_mu, g5);
            a_c[k] += g_c; a_t[k] += g_t; a_mu[k] += g_mu;
            a_k[k] += g5[0];
            a_p[4 * k] += g5[1]; a_p[4 * k + 1] += g5[2]; a_p[4 * k + 2] += g5[3]; a_p[4 * k + 3] += g5[4];
        }
        a_e[0] += a.z; a_e[1] += a.cx; a_e[2] += a.cy; a_e[3] += a.cz;
        if (gxin) gxin[o] = a.x;
        if (gyin) gyin[o] = a.y;
    }
    // cz0 = sqrt(1 - cx0^2 - cy0^2): fold its adjoint into cx, cy
    a_e[1] -= a_e[3] * cx0 / cz0;
    a_e[2] -= a_e[3] * cy0 / cz0;
    // block reduction in a fixed order; columns: [0,3S) c|t|mu, 3S..3S+2 z|cx|cy, then kappa[S], poly[S][4]
    __shared__ double red[kWaves];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const size_t nrows = (size_t)gridDim.y * gridDim.x;
    double *row = part + ((size_t)fw * gridDim.x + blockIdx.x);
    for (int j = 0; j < ncol; ++j) {
        const int src = j < 3 * S + 3 ? j : j + 1;               // skip the g_cz slot
        const double s = wave_sum(acc[src]);
        __syncthreads();
        if (lane == 0) red[wv] = s;
        __syncthreads();
        if (threadIdx.x == 0) row[(size_t)j * nrows] = ((red[0] + red[1]) + red[2]) + red[3];
    }
}

// sum part[col][rows of lens, fields f0.., wavelengths w0..] -> one double (fixed order)
__global__ __launch_bounds__(kBlock) void reduce_kernel(const double *__restrict__ part, int n_rows_all, int ncol_unused, int F, int W,
                                                        int S, int nbx, int asph, double *__restrict__ g_c, double *__restrict__ g_t,
                                                        double *__restrict__ g_mu, double *__restrict__ g_z, double *__restrict__ g_cx,
                                                        double *__restrict__ g_cy, double *__restrict__ g_kappa,
                                                        double *__restrict__ g_poly)
{
    (void)ncol_unused;
    __shared__ double sm[kWaves];
    const int lens = blockIdx.y;
    int b = blockIdx.x, col, f0 = 0, nf = F, w0 = 0, nw = W;
    double *out;
    if (b < S) { col = b; out = g_c + lens * S + b; }
    else if ((b -= S) < S) { col = S + b; out = g_t + lens * S + b; }
    else if ((b -= S) < W * S) { const int w = b / S, k = b % S; col = 2 * S + k; w0 = w; nw = 1; out = g_mu + lens * W * S + b; }
    else if ((b -= W * S) < 1) { col = 3 * S; out = g_z + lens; }
    else if ((b -= 1) < F) { col = 3 * S + 1; f0 = b; nf = 1; out = g_cx + lens * F + b; }
    else if ((b -= F) < F) { col = 3 * S + 2; f0 = b; nf = 1; out = g_cy + lens * F + b; }
    else { b -= F; if (!asph) return; if (b < S) { col = 3 * S + 3 + b; out = g_kappa + lens * S + b; } else { col = 4 * S + 3 + (b - S); out = g_poly + lens * S * 4 + (b - S); } }
    const double *colp = part + (size_t)col * n_rows_all;
    double acc = 0.0;
    for (int f = lens * F + f0; f < lens * F + f0 + nf; ++f) {
        const double *run = colp + ((size_t)f * W + w0) * nbx;
        for (int i = threadIdx.x; i < nw * nbx; i += kBlock) acc += run[i];
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) *out = ((sm[0] + sm[1]) + sm[2]) + sm[3];
}

__global__ __launch_bounds__(kBlock) void reduce_moments_kernel(const double *__restrict__ part, double *__restrict__ mom, int W, int nbx)
{
    __shared__ double sm[kWaves];
    const int f = blockIdx.x / TL_NMOM, j = blockIdx.x % TL_NMOM;
    const size_t nrows = (size_t)(gridDim.x / TL_NMOM) * W * nbx;
    const double *run = part + (size_t)j * nrows + (size_t)f * W * nbx;
    double acc = 0.0;
    for (int i = threadIdx.x; i < W * nbx; i += kBlock) acc += run[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) mom[blockIdx.x] = ((sm[0] + sm[1]) + sm[2]) + sm[3];
}

int launch_fwd(const tl_problem &p, double *x, double *y, double *cx, double *cy, uint8_t *ok, uint8_t *back, double *part, int nbx,
               hipStream_t st)
{
    const int B = p.B > 0 ? p.B : 1;
    hipLaunchKernelGGL(fwd_kernel, dim3(nbx, B * p.F * p.W), dim3(kBlock), 0, st, p, x, y, cx, cy, ok, back, part);
    return (int)hipGetLastError();
}

int launch_reduce_moments(const tl_problem &p, const double *part, double *mom, int nbx, hipStream_t st)
{
    const int B = p.B > 0 ? p.B : 1;
    hipLaunchKernelGGL(reduce_moments_kernel, dim3(B * p.F * TL_NMOM), dim3(kBlock), 0, st, part, mom, p.W, nbx);
    return (int)hipGetLastError();
}

int launch_bwd(const tl_problem &p, const double *gx, const double *gy, const double *gcx, const double *gcy, const double *gmom,
               double *gxin, double *gyin, double *part, int nbx, double *g_c, double *g_t, double *g_mu, double *g_z, double *g_cx,
               double *g_cy, double *g_kappa, double *g_poly, hipStream_t st)
{
    const int B = p.B > 0 ? p.B : 1;
    hipLaunchKernelGGL(bwd_kernel, dim3(nbx, B * p.F * p.W), dim3(kBlock), 0, st, p, gx, gy, gcx, gcy, gmom, gxin, gyin, part);
    int herr = (int)hipGetLastError();
    if (herr) return herr;
    const int asph = p.surf_kind != nullptr;
    const int nout = 2 * p.S + p.W * p.S + 1 + 2 * p.F + (asph ? 5 * p.S : 0);
    hipLaunchKernelGGL(reduce_kernel, dim3(nout, B), dim3(kBlock), 0, st, (const double *)part, B * p.F * p.W * nbx, 0, p.F, p.W, p.S, nbx,
                       asph, g_c, g_t, g_mu, g_z, g_cx, g_cy, g_kappa, g_poly);
    return (int)hipGetLastError();
}

}  // namespace tl_f64
