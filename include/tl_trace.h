/*
 * tl_trace.h -- C ABI of the MI355X (gfx950) sequential ray-trace library, libtltrace.so
 *
 * The reference (OceanT-shirt/TorchOptics) has no FFI: its hot path is a chain of
 * in-process Python calls.  Each entry point below states which reference function it
 * stands in for (file:line under torchlens/); the Python host in torchoptics_amd/
 * mirrors the reference signatures on top of these.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless marked "host";
 *   - nothing is allocated here: the caller owns all buffers (stream-ordered borrow);
 *   - functions are re-entrant and thread-safe (autograd calls backward from its own
 *     thread; ray aiming re-enters the tracer inside a forward);
 *   - return 0 on success, a negative TL_E* code otherwise; tl_last_error() gives the
 *     message for the calling thread;
 *   - the ray batch is [B lenses][F fields][W wavelengths][P pupil points] with P contiguous
 *     ("FWP" layout): one wavefront = 64 consecutive pupil points of one (b, f, w).  B = 1 is the
 *     reference's callers' case; every "[F,...]" / "[S]" shape below gains a leading B when B > 1.
 */
#ifndef TL_TRACE_H
#define TL_TRACE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TL_ABI_VERSION 13
#define TL_MAX_SURFACES 32       /* rows per lens the backward kernels are built for */
#define TL_NMOM 10               /* per-field sums, see tl_trace_fwd */
#define TL_MAX_POLY 4            /* even aspheric terms a4,a6,a8,a10 */
#define TL_MAX_HIT_SLOTS 8       /* aspheric rows per lens whose hit points tl_trace_fwd can hand to the walk-back */

enum {
    TL_OK = 0,
    TL_EINVAL = -1,      /* bad argument (null pointer, bad size, S > TL_MAX_SURFACES ...) */
    TL_ELAUNCH = -2,     /* HIP launch / runtime failure; message has hipGetErrorString */
    TL_EWORKSPACE = -3   /* workspace too small, see tl_workspace_bytes */
};

enum { TL_MODE_STRICT = 0,   /* op-order-faithful fp32, no FMA contraction, IEEE sqrt and divide:
                                forward bit-exact with the reference's eager fp32 ops        */
       TL_MODE_FAST = 1 };   /* FMA contraction + hardware rcp/rsq; a few ulp from strict   */

/*
 * One trace problem.  Mirrors the argument list of
 *   trace_skew(x, y, z, cx, cy, c, t, mu, mask, aggregate, allow_backward_rays)
 *   (ray_tracing_lite.py:594), whose tensors broadcast over a leading lens axis [B, ...]: a batch of B padded
 *   lenses (Structure / Lens of lens_modeling.py:151-386 hold B rows) is ONE launch here, blockIdx.y = (b F + f) W + w.
 *   Padded rows (c = 0, t = 0, mu = 1, mask 0) are traced as the identity rows they are in the reference.
 */
typedef struct tl_problem {
    int32_t F, P, W, S;          /* fields, pupil points in this shard, wavelengths, surface rows (per lens) */
    int32_t device;              /* HIP device ordinal the pointers live on */
    int32_t mode;                /* TL_MODE_* */
    int32_t allow_backward;      /* allow_backward_rays (ray_tracing_lite.py:629) */
    int32_t aggregate;           /* aggregate=True of trace_skew (:641-657): evaluate the penalty-term quantities
                                    (aspheric rows included: theta from the cosine at the aspheric normal) */
    /* entrance-pupil ray coordinates; element strides (floats) over (b, f, p, w), 0 = broadcast.
       Reference shapes [1|B, 1|F, P, 1|W] (ray_tracing_lite.py:112-113). */
    const float *x_in, *y_in;
    int64_t xs_f, xs_p, xs_w;
    int64_t ys_f, ys_p, ys_w;
    const float *z;              /* [B]   pupil position (ray_tracing_lite.py:91)            */
    const float *cx, *cy;        /* initial direction cosines (:116-118), element (b, f) at
                                    cx[b * cx_stride_b + f * cx_stride]                      */
    int32_t cx_stride, cy_stride;/* 1 or 0 (broadcast over fields)                           */
    const float *c, *t;          /* [B,S]   curvature, thickness (:121-122)                  */
    const float *mu;             /* [B,W,S] n_before/n_after per wavelength (:123)           */
    const uint8_t *mask;         /* [B,S]   non-padding rows (:124)                          */
    /* ---- aspheric extension (not in the reference; all three NULL = all-spherical) ----
       sag(rho) = c rho/(1+sqrt(1-(1+kappa) c^2 rho)) + a4 rho^2 + a6 rho^3 + a8 rho^4 + a10 rho^5,
       rho = x^2+y^2; rows with surf_kind 1 are intersected by Newton iteration from the closed-form
       sphere hit and refracted at the aspheric normal; rows with surf_kind 0 take the reference's
       closed form and ignore kappa / poly. */
    const float *kappa;          /* [B,S]   conic constant                                   */
    const float *poly;           /* [B,S,TL_MAX_POLY] a4,a6,a8,a10                           */
    const uint8_t *surf_kind;    /* [B,S]   0 = closed-form sphere, 1 = Newton asphere       */
    const float *n_index;        /* [B,W,S+1] refractive indices (entry 0 = object space); only
                                    read when tl_trace_fwd is asked for `opd`                 */
    /* ---- lens batch (ABI 12) ---- */
    int32_t B;                   /* lenses in this launch; 0 is read as 1.  B * F * W <= 65535 */
    int32_t cx_stride_b, cy_stride_b;   /* element stride of cx / cy over the lens index (0 = shared) */
    int64_t xs_b, ys_b;          /* element stride of x_in / y_in over the lens index (0 = shared) */
    /* ---- ABI 13 ---- */
    float *asph_hits;            /* [asph_hit_slots][2][B,F,W,P] float, nullable: the hit point (X, Y) of every ray on the
                                    j-th aspheric row of its lens, j < asph_hit_slots -- WRITTEN by tl_trace_fwd, READ by
                                    tl_trace_bwd_from_outputs, which then needs no Newton iteration on the reversed ray (and
                                    re-anchors the reconstruction at every aspheric row).  8 bytes per ray and aspheric row.
                                    A lens with more aspheric rows than slots is still traced correctly: its backward falls
                                    back, on the device, to the checkpoint kernel.  NULL: Newton on the reversed ray. */
    int32_t asph_hit_slots;      /* 0..TL_MAX_HIT_SLOTS */
    int32_t moments_x;           /* 1: tl_trace_fwd also accumulates the x-moments 4..6 (compute_rms2d reads y only:
                                    ray_tracing_lite.py:684-701); 0: they are returned as 0 */
    uint8_t *cond_flags;         /* [B,F,W,P] bytes, nullable: the `ok` output with the conditioning flag -- 0 dead, 1 live, 2 live
                                    and ill-conditioned (smallest cos^2 of incidence or refraction below 0.01: the rays moment 9
                                    counts) -- WRITTEN by tl_trace_fwd (next to `ok`, which stays 0/1), READ by
                                    tl_trace_bwd_from_outputs in place of ok_fwd.  With it, a launch that holds such rays is
                                    no longer handed to the checkpoint kernel as a whole: the walk-back differentiates the
                                    rays marked 1 and the checkpoint kernel exactly those marked 2 (waves without one skip
                                    their chunk); the two partial sums are added.  NULL: one ill-conditioned ray sends the
                                    whole launch to the checkpoint kernel. */
} tl_problem;

int         tl_version(void);            /* == TL_ABI_VERSION */
const char *tl_last_error(void);         /* host string, thread-local, never NULL */
size_t      tl_problem_size(void);       /* sizeof(tl_problem): lets a foreign-language binding check its layout */

/* Bytes of scratch (device) the calls below need for this problem. */
size_t tl_workspace_bytes(const tl_problem *p);

/*
 * Forward trace: replaces the whole Python loop trace_skew (ray_tracing_lite.py:594-675 =
 * find_marching_distance_spherical :525-545, update_ray_coordinates :514-522,
 * reset_bad_rays :574-591, apply_snell_spherical :548-571, image-plane transfer :659-663).
 *   (shapes for B = 1; with a lens batch every output gains a leading B: [B,F,W,P], moments [B,F,TL_NMOM],
 *    stacks [3][S][B,F,W,P])
 *   x,y,cx,cy : [F,W,P] float  (any may be NULL = not wanted)
 *   ok,back   : [F,W,P] uint8  (nullable)
 *   opd       : [F,W,P] float  optical path length sum_k n_k d_k + n_S d_image from the pupil plane
 *               to the image plane, 0 for failed rays; needs p->n_index (nullable; extension; its gradient:
 *               `g_opd` of tl_trace_bwd)
 *   moments   : [F,TL_NMOM] double (nullable), per field over (w,p):
 *               0 sum y | 1 sum ok*y | 2 sum ok*y^2 | 3 sum ok | 4 sum x | 5 sum ok*x |
 *               6 sum ok*x^2 | 7 sum back | 8 sum q (p->aggregate only) |
 *               9 number of ILL-CONDITIONED live rays: smallest cos^2 of incidence / refraction along the
 *                 path below 0.01 (fp32 rounding is amplified ~1/cos^2 there)
 *               -- 0..3 are the sufficient statistics of compute_rms2d (ray_tracing_lite.py:678-702);
 *               q = per-ray sum over the surfaces of theta_norm + theta_prime_norm + z_RELU
 *               (ray_tracing_lite.py:641-657), NaN -> 0, i.e. sumQ * n_sequence of
 *               optics_simulator_lite.py:441-448.  All reduced in a fixed order (bitwise reproducible).
 *   stacks    : [3][S][F,W,P] float (nullable, p->aggregate only): the per-surface stacks
 *               z_RELU | theta_norm | theta_prime_norm that trace_skew(aggregate=True) returns.
 */
int tl_trace_fwd(const tl_problem *p,
                 float *x, float *y, float *cx, float *cy, uint8_t *ok, uint8_t *back,
                 float *opd, float *stacks, double *moments,
                 void *workspace, size_t workspace_bytes, void *stream);

/*
 * Backward trace: replaces PyTorch autograd's replay of the recorded graph of trace_skew
 * (SURVEY 3.4).  Recomputes the forward per ray in registers, then sweeps the surfaces in
 * reverse.
 *   gx,gy,gcx,gcy : [F,W,P] upstream gradients of the per-ray outputs (each nullable)
 *   g_opd         : [F,W,P] upstream gradient of the optical path length output (nullable; needs p->n_index and
 *                   g_n_index).  OPD = sum_k n_k d_k + n_S d_image: it enters the adjoint of every marching distance
 *                   and so reaches c, t, mu, z, cx, cy, kappa, poly, x_in, y_in; g_n_index [W,S+1] = d/d n_index
 *   g_moments     : [F,TL_NMOM] double upstream gradient of `moments` (nullable); the per-ray
 *                   seed  gM0 + ok*(gM1 + 2*y*gM2)  (and the x analogue) is formed in-kernel;
 *                   entry 8 seeds the penalty term when p->aggregate
 *   g_c,g_t [S], g_mu [W,S], g_z [1], g_cx,g_cy [F] : float, OVERWRITTEN (not accumulated); summed over the
 *                   rays in fp64 in a fixed order and rounded once.  Lens batch: per lens, [B,S], [B,W,S], [B],
 *                   [B,F] (and g_kappa [B,S], g_poly [B,S,4], g_n_index [B,W,S+1]); a lens never sees another's rays
 *   g_kappa [S], g_poly [S,TL_MAX_POLY] : float, nullable (aspheric extension)
 *   g_x_in,g_y_in : [F,W,P] float per-ray input gradients (nullable; used by ray aiming,
 *                   ray_tracing_lite.py:169-181)
 */
int tl_trace_bwd(const tl_problem *p,
                 const float *gx, const float *gy, const float *gcx, const float *gcy,
                 const double *g_moments, const float *g_opd,
                 float *g_c, float *g_t, float *g_mu, float *g_z, float *g_cx, float *g_cy,
                 float *g_kappa, float *g_poly, float *g_n_index,
                 float *g_x_in, float *g_y_in,
                 void *workspace, size_t workspace_bytes, void *stream);

/*
 * Backward trace WITHOUT re-tracing forwards: same gradients as tl_trace_bwd, computed by walking each ray
 * back from the forward kernel's own output (x, y, cx, cy, ok of tl_trace_fwd for this problem) -- undo
 * the refraction at surface k, intersect the incoming line with surface k-1, apply the adjoint step.
 * No per-surface state is kept, so the kernel runs at twice the occupancy and needs no bit-exact
 * re-derivation of the forward; the reconstructed states differ from the forward's by rounding only,
 * which perturbs the gradients at the 1e-6 level (tested) -- except for ill-conditioned rays (moment 9 of
 * tl_trace_fwd), where it would reach 1e-4.  The call therefore always enqueues the checkpoint kernel of
 * tl_trace_bwd behind the walk-back kernel; both decide ON THE DEVICE which of them does the work (no host
 * synchronisation; the idle launch retires in microseconds).  The checkpoint kernel takes over when
 *   - `moments_fwd` (the forward's moments, nullable) counts an ill-conditioned live ray -- for exactly those rays when
 *     the forward left their flags in p->cond_flags (read in place of ok_fwd, which may then be NULL; the ok bytes given
 *     otherwise must be 0 / 1 as tl_trace_fwd writes them), for the whole launch when it did not -- or
 *   - the walk-back met a non-finite adjoint (it then flags a word at the end of the workspace): the whole launch.
 * Pass `moments_fwd` whenever it is available: without it an ill-conditioned fan is walked back anyway.
 * allow_backward = 1 only, no OPD gradient (TL_EINVAL otherwise: use tl_trace_bwd).  Aspheric rows are walked
 * back too (g_kappa, g_poly as in tl_trace_bwd, required iff p->surf_kind).
 * Workspace: tl_workspace_bytes(p); its contents need not be initialised.
 * Recorded into a HIP graph, the call is replayed with the same internal token: give the tl_trace_fwd of the same
 * step the SAME workspace with its full tl_workspace_bytes(p) (with or without `moments`) -- it clears the flag word, so
 * that one replay that had to fall back does not pin every later replay to the (exact, slower) checkpoint kernel.
 * Penalty term (p->aggregate, ABI 13): walked back too for lenses of 3..20 rows -- the rays alive at the image plane by
 * the walk-back kernel, the rays that died on the way (they keep the gradient of the rows they passed alive) by the
 * checkpoint kernel in the same call; other lenses are handed to tl_trace_bwd as a whole.  Aspheric rows (ABI 13): with
 * p->asph_hits filled by tl_trace_fwd the hit on an aspheric row is read instead of searched.
 */
int tl_trace_bwd_from_outputs(const tl_problem *p,
                              const float *gx, const float *gy, const float *gcx, const float *gcy,
                              const double *g_moments,
                              const float *x_fwd, const float *y_fwd, const float *cx_fwd, const float *cy_fwd,
                              const uint8_t *ok_fwd, const double *moments_fwd,
                              float *g_c, float *g_t, float *g_mu, float *g_z, float *g_cx, float *g_cy,
                              float *g_kappa, float *g_poly,
                              float *g_x_in, float *g_y_in,
                              void *workspace, size_t workspace_bytes, void *stream);

/*
 * Spot moments of arbitrary per-ray tensors (same TL_NMOM layout): the reduction inside
 * compute_rms2d(x, y, ray_ok) (ray_tracing_lite.py:678-702) when its inputs did not come
 * from tl_trace_fwd.  Element strides over (f, p, w); x may be NULL.
 */
int tl_spot_moments(int32_t device, int32_t F, int32_t P, int32_t W,
                    const float *x, const float *y, const uint8_t *ok,
                    int64_t s_f, int64_t s_p, int64_t s_w,
                    double *moments, void *workspace, size_t workspace_bytes, void *stream);

/*
 * The closed form of compute_rms2d (ray_tracing_lite.py:684-701) on the moments, and its derivative:
 *   rms = mean_f sqrt((M2 - 2 m M1 + m^2 M3) / n),  m = M0 / n,  n = P*W rays per field (of the WHOLE
 *   pupil when it is sharded over GPUs: pass the all-reduced moments).
 *   rms [B] float (one value per lens: moments [B,F,TL_NMOM]; compute_rms2d itself reads lens 0 only);
 *   d_moments [B,F,TL_NMOM] double = d rms[b] / d moments[b] (zero where the variance is 0).
 */
int tl_spot_rms(int32_t device, int32_t B, int32_t F, double n_per_field, const double *moments, float *rms,
                double *d_moments, void *stream);

/*
 * loss_dict of RaytracedOptics.compute_loss_out (optics_simulator_lite.py:430-450) per lens, on the moments of an
 * aggregate trace:  rms[b] as tl_spot_rms,  penalty[b] = (sum_f moments[b,f,8]) / n_sequence  (sumQ :441-448, the
 * per-ray sums fused into the trace kernel; rounded to float),  loss[b] = rms[b] + penalty_rate * penalty[b]  (:449).
 * One launch instead of ~10 elementwise ones; same rounding points as that op sequence.
 *   n_sequence: [B] doubles on the device (padded batches: rows of each lens' sequence), or NULL: n_sequence_all for all;
 *   d_rms [B,F,TL_NMOM] double = d rms[b] / d moments[b], for tl_unsup_loss_bwd.
 * tl_unsup_loss_bwd: g_moments [B,F,TL_NMOM] = d(sum_b g_loss[b] loss[b] + g_rms[b] rms[b] + g_penalty[b] penalty[b]) /
 *   d moments; each upstream gradient nullable (not all three), element b at g[b * g_stride] (g_stride 0: one value for all).
 */
int tl_unsup_loss(int32_t device, int32_t B, int32_t F, double n_per_field, const double *moments, const double *n_sequence,
                  double n_sequence_all, float penalty_rate, float *loss, float *rms, float *penalty, double *d_rms,
                  void *stream);
int tl_unsup_loss_bwd(int32_t device, int32_t B, int32_t F, const double *d_rms, const float *g_loss, const float *g_rms,
                      const float *g_penalty, int32_t g_stride, const double *n_sequence, double n_sequence_all,
                      float penalty_rate, double *g_moments, void *stream);

/* d(loss)/dy, d(loss)/dx per ray from d(loss)/d(moments); outputs [F,P,W]-strided like y. */
int tl_spot_seed(int32_t device, int32_t F, int32_t P, int32_t W,
                 const float *x, const float *y, const uint8_t *ok,
                 int64_t s_f, int64_t s_p, int64_t s_w,
                 const double *g_moments, float *gx, float *gy, void *stream);

/*
 * Paraxial entrance-pupil position w.r.t. the first vertex -- the `z` argument of the trace
 * (compute_pupil_position, ray_tracing_lite.py:301-350): z = B/A of the ABCD product of the K rows in
 * front of the stop, row k = refraction at curvature c[k] from index n[k] to n[k+1], then a gap t[k].
 *   c, t [B,K], n [B,K+1] (n[.,0] = object space), z [B] (nullable): float; B lenses, one thread each (rows behind a
 *   lens' own stop padded with c = 0, t = 0, n = 1: identity).
 *   g_z [B] (nullable): upstream gradient; then g_c, g_t [B,K] and g_n [B,K+1] are OVERWRITTEN with
 *   d(loss)/d(c, t, n).  One tiny launch instead of the ~25 (+ ~60 in autograd) of the elementwise / 2x2-matmul chain.
 *   mode (ABI 13): TL_MODE_STRICT -- z is the reference's fp32 value bit for bit (its pairwise product tree of fp32 2x2
 *   matrices, reduce_abcd :301-318, every operation rounded separately); TL_MODE_FAST -- the product in fp64, rounded
 *   once.  The gradient is the fp64 adjoint in both modes.
 */
int tl_pupil_position(int32_t device, int32_t B, int32_t K, const float *c, const float *t, const float *n, float *z,
                      const float *g_z, float *g_c, float *g_t, float *g_n, int32_t mode, void *stream);

/*
 * Ray aiming, one iteration (RayTracer.ray_aiming, ray_tracing_lite.py:129-208, with compute_pupil_radius :834-844,
 * ray_aiming_mode 'real'): per (lens, field, wavelength) the three 'tee' rays -- bottom and top meridional, +x sagittal --
 * are traced to the stop through the K rows in front of it, one Newton step each brings them to where an ideal pupil
 * would put them (stop radius = height of the on-axis d-line marginal ray), and the affine pupil map through the
 * corrected rays is returned:   x_pupil' = x_pupil * x_scale,   y_pupil' = y_pupil * y_scale + y_offset.
 * One thread per (lens, field, wavelength), fp64 in registers, Jacobian by central differences (the reference:
 * two eager traces + an autograd pass, ~60 tensor ops per call).  No gradient: the reference aims a detached lens (:108).
 *   c, t [B,K]; n [B,K,W] refractive index behind each row at the tracer's wavelengths, n_d [B,K] at the d line (rows
 *   behind a lens' own stop padded: c = 0, t = 0, n = 1); mask [B,K] (backward-ray test, :626-632); kappa [B,K],
 *   poly [B,K,TL_MAX_POLY], surf_kind [B,K] nullable (aspheric rows); z [B] pupil position; hfov [B] half field of view
 *   (rad), fields [F] relative field heights (cy = sin(hfov * field), :116-118); epd [B];
 *   x_scale, y_scale, y_offset [B,F,W] float outputs.
 */
int tl_ray_aim(int32_t device, int32_t B, int32_t F, int32_t W, int32_t K, const float *c, const float *t, const float *n,
               const float *n_d, const uint8_t *mask, const float *kappa, const float *poly, const uint8_t *surf_kind,
               const float *z, const float *hfov, const float *fields, const float *epd, int32_t allow_backward,
               float *x_scale, float *y_scale, float *y_offset, void *stream);

/*
 * The aimed fan, [B,F,W,P] floats (consecutive pupil points contiguous), from one shared relative pupil grid xp, yp [P] and
 * tl_ray_aim's map:  x = clamp(xp x_scale, -2, 2) epd / 2,  y = clamp(yp y_scale + y_offset, -2, 2) epd / 2
 * (RayTracer.trace_rays, ray_tracing_lite.py:104-113: remap, torch.clamp(-2, 2), scale_to_epd) -- one launch for seven,
 * same rounding points.  x_scale, y_scale, y_offset [B,F,W]; epd [B].
 */
int tl_aim_fan(int32_t device, int32_t B, int32_t F, int32_t W, int32_t P, const float *xp, const float *yp,
               const float *x_scale, const float *y_scale, const float *y_offset, const float *epd, float *x_out, float *y_out,
               void *stream);

/*
 * Double precision -- RayTracer(double_precision=True) (ray_tracing_lite.py:82-84; the reference crashes there: Specs and
 * Lens have no .double()).  The same trace, forward and checkpoint backward, entirely in fp64: generic,
 * untuned kernels (one ray per lane, rolled loops) for reference-quality numbers on the GPU, not for speed.
 * `p` is a tl_problem whose float-typed pointers (x_in, y_in, z, cx, cy, c, t, mu, kappa, poly) POINT AT DOUBLES, same
 * shapes and strides (in elements); mask / surf_kind stay uint8; aggregate must be 0, n_index / asph_hits are ignored.
 * Outputs and gradients as in tl_trace_fwd / tl_trace_bwd with double instead of float (moments [B,F,TL_NMOM] as there,
 * entry 8 = 0).  Workspace: tl_workspace_bytes_f64(p).
 */
size_t tl_workspace_bytes_f64(const tl_problem *p);
int tl_trace_fwd_f64(const tl_problem *p, double *x, double *y, double *cx, double *cy, uint8_t *ok, uint8_t *back,
                     double *moments, void *workspace, size_t workspace_bytes, void *stream);
int tl_trace_bwd_f64(const tl_problem *p, const double *gx, const double *gy, const double *gcx, const double *gcy,
                     const double *g_moments, double *g_c, double *g_t, double *g_mu, double *g_z, double *g_cx, double *g_cy,
                     double *g_kappa, double *g_poly, double *g_x_in, double *g_y_in, void *workspace, size_t workspace_bytes,
                     void *stream);

/*
 * Diagnostic: quot[i] = a[i] / b[i] and root[i] = sqrt(b[i]) evaluated by the division and square root the trace kernels of
 * `mode` use.  Strict mode promises the correctly rounded (IEEE) results on the operand ranges of the trace, from shorter
 * instruction sequences than the compiler's general ones: tests/test_gpu_arith.py holds it to that, bit for bit.
 */
int tl_selftest_arith(int32_t device, int32_t mode, const float *a, const float *b, int64_t n, float *quot, float *root,
                      void *stream);

#ifdef __cplusplus
}
#endif
#endif /* TL_TRACE_H */
