// tl_torch.cpp -- the eager host chain of the ray tracer as C++ autograd functions (torchoptics_amd/_tlx.so).
//
// What the Python wrappers in ops.py do per call -- normalise the broadcast shapes of trace_skew's arguments
// (ray_tracing_lite.py:594-612), allocate the outputs, fill a tl_problem, call the C ABI of libtltrace.so
// (include/tl_trace.h), hand the gradients back in the callers' shapes -- costs 0.25-0.55 ms of interpreter and
// autograd-bookkeeping time per optimisation step there (one Python autograd.Function each way, ~15 view ops that each
// leave a node for the engine to walk, ctypes marshalling).  Here the same work is one C++ torch::autograd::Function
// per direction: no view nodes (gradients are reduced to the callers' shapes by hand), no interpreter in the backward
// (the engine calls straight into C++), the tl_problem rebuilt from the saved tensors (so saved-tensor hooks are safe).
// This file contains NO device code and no arithmetic of the trace: every number comes from libtltrace.so.
#include <torch/extension.h>
// (PyTorch-ROCm presents its HIP devices under the device type "cuda": the guard and the stream accessor that go with
//  that are the ...MasqueradingAsCUDA ones)
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <hip/hip_runtime_api.h>

#include <map>
#include <mutex>
#include <vector>

#include "../../include/tl_trace.h"

namespace {

using at::Tensor;
using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

void check(int rc, const char *what)
{
    TORCH_CHECK(rc == 0, what, " failed (code ", rc, "): ", tl_last_error());
}

void require_device(const Tensor &t, const char *name)
{
    TORCH_CHECK(t.is_cuda(), "torchoptics_amd: `", name, "` lives on ", t.device(),
                "; the ray tracer runs only as HIP kernels on an AMD GPU (there is no CPU fallback).  Move the lens and "
                "rays to device='cuda'.");
}

// ------------------------------------------------------------------ per-(device, stream) scratch for the block partials
std::mutex g_ws_mutex;
std::map<std::pair<int, void *>, Tensor> g_ws;

Tensor workspace(size_t nbytes, const at::Device &dev, void *stream)
{
    std::lock_guard<std::mutex> lock(g_ws_mutex);
    Tensor &ws = g_ws[{dev.index(), stream}];
    if (!ws.defined() || (size_t)ws.numel() < nbytes)
        ws = at::empty({(int64_t)std::max<size_t>(nbytes, (size_t)1 << 20)}, at::TensorOptions().dtype(at::kByte).device(dev));
    return ws;
}

// ------------------------------------------------------------------ optional live timing of the C-ABI calls (bench.py)
struct Timing {
    std::mutex m;
    bool on = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev[2];      // 0 = fwd, 1 = bwd
} g_timing;

struct Timed {
    int key;
    hipStream_t st;
    hipEvent_t a = nullptr, b = nullptr;
    Timed(int key_, hipStream_t st_) : key(key_), st(st_)
    {
        if (!g_timing.on) return;
        (void)hipEventCreate(&a);
        (void)hipEventCreate(&b);
        (void)hipEventRecord(a, st);
    }
    ~Timed()
    {
        if (!a) return;
        (void)hipEventRecord(b, st);
        std::lock_guard<std::mutex> lock(g_timing.m);
        g_timing.ev[key].emplace_back(a, b);
    }
};

void enable_timing(bool on)
{
    std::lock_guard<std::mutex> lock(g_timing.m);
    for (auto &v : g_timing.ev) {
        for (auto &p : v) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
        v.clear();
    }
    g_timing.on = on;
}

// mean GPU milliseconds per call since enable_timing(true): (fwd, bwd), -1 where nothing was recorded
std::pair<double, double> timing_ms()
{
    (void)hipDeviceSynchronize();
    std::lock_guard<std::mutex> lock(g_timing.m);
    double out[2] = {-1.0, -1.0};
    for (int k = 0; k < 2; ++k) {
        double s = 0.0;
        for (auto &p : g_timing.ev[k]) {
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, p.first, p.second);
            s += ms;
        }
        if (!g_timing.ev[k].empty()) out[k] = s / (double)g_timing.ev[k].size();
    }
    return {out[0], out[1]};
}

// C-ABI calls recorded since enable_timing(true): (fwd, bwd)
std::pair<int64_t, int64_t> timing_counts()
{
    std::lock_guard<std::mutex> lock(g_timing.m);
    return {(int64_t)g_timing.ev[0].size(), (int64_t)g_timing.ev[1].size()};
}

// ------------------------------------------------------------------ tl_problem from the normalised tensors
struct Norm {                       // trace_skew's arguments brought to the kernel's shapes (no copies where they fit)
    Tensor x_e, y_e;                // [B,F,P,W] expanded views
    Tensor z, cx, cy;               // [B], [1|B, 1|F], [1|B, 1|F]
    Tensor c, t, mu, mask;          // [B,S], [B,S], [B,W,S], [B,S] u8
    Tensor kappa, poly, kind;       // [B,S], [B,S,4], [B,S] u8 or undefined
    Tensor n_index;                 // [B,W,S+1] or undefined
    int64_t B, F, P, W, S;
};

const void *ptr(const Tensor &t) { return t.defined() ? t.data_ptr() : nullptr; }

tl_problem make_problem(const Norm &n, bool allow_back, int mode, bool aggregate, const Tensor &hits, bool moments_x,
                        bool with_n_index, const Tensor &cond)
{
    tl_problem p{};
    p.F = (int32_t)n.F; p.P = (int32_t)n.P; p.W = (int32_t)n.W; p.S = (int32_t)n.S; p.B = (int32_t)n.B;
    p.device = n.x_e.device().index();
    p.mode = mode;
    p.allow_backward = allow_back ? 1 : 0;
    p.aggregate = aggregate ? 1 : 0;
    p.x_in = (const float *)n.x_e.data_ptr(); p.y_in = (const float *)n.y_e.data_ptr();
    p.xs_b = n.x_e.stride(0); p.xs_f = n.x_e.stride(1); p.xs_p = n.x_e.stride(2); p.xs_w = n.x_e.stride(3);
    p.ys_b = n.y_e.stride(0); p.ys_f = n.y_e.stride(1); p.ys_p = n.y_e.stride(2); p.ys_w = n.y_e.stride(3);
    p.z = (const float *)n.z.data_ptr();
    p.cx = (const float *)n.cx.data_ptr(); p.cy = (const float *)n.cy.data_ptr();
    p.cx_stride = n.cx.size(1) == 1 ? 0 : 1; p.cx_stride_b = n.cx.size(0) == 1 ? 0 : (int32_t)n.cx.size(1);
    p.cy_stride = n.cy.size(1) == 1 ? 0 : 1; p.cy_stride_b = n.cy.size(0) == 1 ? 0 : (int32_t)n.cy.size(1);
    p.c = (const float *)n.c.data_ptr(); p.t = (const float *)n.t.data_ptr(); p.mu = (const float *)n.mu.data_ptr();
    p.mask = (const uint8_t *)n.mask.data_ptr();
    if (n.kind.defined()) {
        p.kappa = (const float *)n.kappa.data_ptr(); p.poly = (const float *)n.poly.data_ptr();
        p.surf_kind = (const uint8_t *)n.kind.data_ptr();
    }
    p.n_index = (with_n_index && n.n_index.defined()) ? (const float *)n.n_index.data_ptr() : nullptr;
    p.asph_hits = hits.defined() ? (float *)hits.data_ptr() : nullptr;
    p.asph_hit_slots = hits.defined() ? (int32_t)hits.size(0) : 0;
    p.moments_x = moments_x ? 1 : 0;
    p.cond_flags = cond.defined() ? (uint8_t *)cond.data_ptr() : nullptr;
    return p;
}

Tensor f32(const Tensor &t) { return t.scalar_type() == at::kFloat ? t : t.to(at::kFloat); }
Tensor dense(const Tensor &t) { return t.is_contiguous() ? t : t.contiguous(); }

// [n_lens, tail...] contiguous from an argument that holds one set of values per lens, or one set for all
Tensor rows(const Tensor &a, int64_t n_lens, at::IntArrayRef tail)
{
    int64_t n_tail = 1;
    for (auto d : tail) n_tail *= d;
    std::vector<int64_t> shape{n_lens};
    shape.insert(shape.end(), tail.begin(), tail.end());
    if (a.numel() == n_lens * n_tail) return dense(a.reshape(shape));
    std::vector<int64_t> one{1};
    one.insert(one.end(), tail.begin(), tail.end());
    return dense(a.reshape(one).expand(shape));
}

// gradient of an argument that was broadcast into `g` (any shape with g.numel() = prod(full)): reduce to `orig`
Tensor to_shape(const Tensor &g, at::IntArrayRef full, at::IntArrayRef orig)
{
    return at::sum_to(g.reshape(full), orig);
}

thread_local bool g_last_use_inv = false;

// flag bits of the int argument of TraceFn (kept in one int: every extra non-tensor argument of a custom function is
// one more slot in the gradient list)
enum { kAllowBack = 1, kWantRays = 2, kWantOpd = 4, kAggregate = 8, kWantStacks = 16, kMomentsX = 32, kInverse = 64, kFuseRms = 128 };

class TraceFn : public torch::autograd::Function<TraceFn> {
public:
    // inputs 0..12 are tensors (some optional), then three ints
    static variable_list forward(AutogradContext *ctx, Tensor x, Tensor y, Tensor z, Tensor cx, Tensor cy, Tensor c, Tensor t,
                                 Tensor mu, Tensor mask, c10::optional<Tensor> kappa, c10::optional<Tensor> poly,
                                 c10::optional<Tensor> kind, c10::optional<Tensor> n_index, int64_t flags, int64_t mode,
                                 int64_t hit_slots)
    {
        const bool allow_back = flags & kAllowBack, want_rays = flags & kWantRays, want_opd = flags & kWantOpd;
        const bool aggregate = flags & kAggregate, want_stacks = flags & kWantStacks, moments_x = flags & kMomentsX;
        require_device(x, "x"); require_device(y, "y"); require_device(z, "z"); require_device(cx, "cx");
        require_device(cy, "cy"); require_device(c, "c"); require_device(t, "t"); require_device(mu, "mu");
        require_device(mask, "mask");
        const bool need_xy = x.requires_grad() || y.requires_grad();
        bool need_any = z.requires_grad() || cx.requires_grad() || cy.requires_grad() || c.requires_grad() ||
                        t.requires_grad() || mu.requires_grad();
        if (kappa && kappa->defined()) need_any = need_any || kappa->requires_grad();
        if (poly && poly->defined()) need_any = need_any || poly->requires_grad();
        // ---- shapes (ray_tracing_lite.py:594-612: everything broadcasts to [B, F, P, W(, S)])
        Norm n;
        int64_t B = 1;
        for (const Tensor *a : {&x, &y, &z, &cx, &cy, &c, &t, &mu, &mask})
            if (a->dim() >= 4) B = std::max(B, a->size(0));
        for (const Tensor *a : {&x, &y, &cx, &cy, &z})
            TORCH_CHECK(a->dim() == 4 && (a->size(0) == 1 || a->size(0) == B), "x, y, z, cx, cy must be 4-D with 1 or ", B,
                        " lenses in dim 0, got ", a->sizes());
        TORCH_CHECK(c.dim() == 5 && t.dim() == 5 && mu.dim() == 5, "c, t, mu must be 5-D [1|B,1,1,1|W,S]");
        for (const Tensor *a : {&c, &t, &mu})
            TORCH_CHECK(a->size(0) == 1 || a->size(0) == B, "c, t, mu must be 5-D [1|B,1,1,1|W,S]");
        const int64_t S = c.size(-1);
        TORCH_CHECK(S <= TL_MAX_SURFACES, "lens has ", S, " rows; this build supports at most ", TL_MAX_SURFACES);
        const int64_t F = std::max(std::max(x.size(1), y.size(1)), std::max(cx.size(1), cy.size(1)));
        const int64_t P = std::max(x.size(2), y.size(2));
        const int64_t W = std::max(std::max(x.size(3), y.size(3)), mu.size(3));
        TORCH_CHECK(z.numel() == 1 || z.numel() == B, "z must hold one pupil position per lens");
        for (const Tensor *a : {&cx, &cy})
            TORCH_CHECK((a->size(1) == 1 || a->size(1) == F) && a->size(2) == 1 && a->size(3) == 1,
                        "cx, cy must be per-field [1|B,1|F,1,1]");
        n.B = B; n.F = F; n.P = P; n.W = W; n.S = S;
        n.x_e = f32(x).expand({B, F, P, W});
        n.y_e = f32(y).expand({B, F, P, W});
        n.z = rows(f32(z), B, {});
        n.cx = dense(f32(cx).reshape({cx.size(0), cx.size(1)}));
        n.cy = dense(f32(cy).reshape({cy.size(0), cy.size(1)}));
        n.c = rows(f32(c), B, {S});
        n.t = rows(f32(t), B, {S});
        n.mu = mu.size(3) == W ? rows(f32(mu), B, {W, S}) : dense(f32(mu).reshape({mu.size(0), 1, S}).expand({B, W, S}));
        {
            Tensor m = mask.reshape({-1, S});
            m = m.scalar_type() == at::kBool ? m.view(at::kByte) : m.to(at::kByte);
            n.mask = dense(m.expand({B, S}));
        }
        const bool asph = kind && kind->defined();
        if (asph) {
            TORCH_CHECK(kappa && kappa->defined() && poly && poly->defined(), "aspheric rows need kappa and poly");
            n.kappa = dense(f32(*kappa).reshape({-1, S}).expand({B, S}));
            n.poly = dense(f32(*poly).reshape({-1, S, 4}).expand({B, S, 4}));
            n.kind = dense(kind->reshape({-1, S}).to(at::kByte).expand({B, S}));
        }
        if (want_opd) {
            TORCH_CHECK(n_index && n_index->defined(), "want_opd=True needs n_index [1|B,1,1,W,S+1]");
            const Tensor ni = f32(*n_index);
            n.n_index = dense(ni.reshape({ni.dim() == 5 ? ni.size(0) : 1, -1, S + 1}).expand({B, W, S + 1}));
            need_any = need_any || n_index->requires_grad();
        }
        const at::Device dev = n.x_e.device();
        c10::hip::HIPGuardMasqueradingAsCUDA guard(dev);
        const hipStream_t st = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(dev.index()).stream();
        // per-ray input gradients (ray aiming: a handful of rays) and the gradient through the optical path length keep
        // the checkpoint algorithm (see ops.py)
        const bool use_inv = (flags & kInverse) && want_rays && allow_back && !want_opd && !need_xy;
        g_last_use_inv = use_inv;
        const auto fopt = at::TensorOptions().dtype(at::kFloat).device(dev);
        Tensor hits;
        if (use_inv && asph && hit_slots > 0 && need_any)
            hits = at::empty({std::min<int64_t>(hit_slots, S), 2, B, F, W, P}, fopt);
        // one byte per ray: the ill-conditioned live rays, which the backward then leaves to the checkpoint kernel (ops.py)
        Tensor cond;
        if (use_inv && need_any) cond = at::empty({B, F, W, P}, fopt.dtype(at::kByte));
        tl_problem prob = make_problem(n, allow_back, (int)mode, aggregate, hits, moments_x, want_opd, cond);
        const size_t nbytes = tl_workspace_bytes(&prob);
        Tensor ws = workspace(nbytes, dev, (void *)st);
        Tensor fp[4], bp[2], opd, stacks;
        if (want_rays) {
            for (auto &b : fp) b = at::empty({B, F, W, P}, fopt);
            for (auto &b : bp) b = at::empty({B, F, W, P}, fopt.dtype(at::kByte));
        }
        if (want_opd) opd = at::empty({B, F, W, P}, fopt);
        if (aggregate && want_stacks) stacks = at::empty({3, S, B, F, W, P}, fopt);
        Tensor moments = at::empty({B * F, TL_NMOM}, fopt.dtype(at::kDouble));
        {
            Timed timed(0, st);
            check(tl_trace_fwd(&prob, (float *)ptr(fp[0]), (float *)ptr(fp[1]), (float *)ptr(fp[2]), (float *)ptr(fp[3]),
                               (uint8_t *)ptr(bp[0]), (uint8_t *)ptr(bp[1]), (float *)ptr(opd), (float *)ptr(stacks),
                               (double *)moments.data_ptr(), ws.data_ptr(), (size_t)ws.numel(), (void *)st),
                  "tl_trace_fwd");
        }
        // kFuseRms: compute_rms2d of these rays (every lens of the batch; n = P W rays per field) as a tenth output of THIS
        // node -- one tl_spot_rms launch here instead of a second extension call and a second autograd node for the loss
        // every caller of the spot metric forms next (ray_tracing.compute_rms2d picks it up)
        Tensor rms, d_rms;
        if ((flags & kFuseRms) && want_rays && P > 0) {
            rms = B == 1 ? at::empty({}, fopt) : at::empty({B}, fopt);
            d_rms = at::empty_like(moments);
            check(tl_spot_rms(dev.index(), (int32_t)B, (int32_t)F, (double)(P * W), (const double *)moments.data_ptr(),
                              (float *)rms.data_ptr(), (double *)d_rms.data_ptr(), (void *)st),
                  "tl_spot_rms");
        }
        const bool inv = use_inv;
        ctx->save_for_backward({n.x_e, n.y_e, n.z, n.cx, n.cy, n.c, n.t, n.mu, n.mask, n.kappa, n.poly, n.kind,
                                inv ? fp[0] : Tensor(), inv ? fp[1] : Tensor(), inv ? fp[2] : Tensor(), inv ? fp[3] : Tensor(),
                                inv ? bp[0] : Tensor(), inv ? moments : Tensor(), want_opd ? n.n_index : Tensor(), hits, cond,
                                d_rms});
        // needs_input_grad() of the backward counts the tensor arguments that are PRESENT: an absent optional has no edge
        {
            int64_t e = 9;
            ctx->saved_data["e_kappa"] = (kappa && kappa->defined()) ? e++ : (int64_t)-1;
            ctx->saved_data["e_poly"] = (poly && poly->defined()) ? e++ : (int64_t)-1;
            if (kind && kind->defined()) ++e;
            ctx->saved_data["e_n"] = (n_index && n_index->defined()) ? e++ : (int64_t)-1;
        }
        ctx->saved_data["flags"] = flags;
        ctx->saved_data["mode"] = mode;
        ctx->saved_data["use_inv"] = use_inv;
        ctx->saved_data["x_shape"] = x.sizes().vec(); ctx->saved_data["y_shape"] = y.sizes().vec();
        ctx->saved_data["z_shape"] = z.sizes().vec(); ctx->saved_data["cx_shape"] = cx.sizes().vec();
        ctx->saved_data["cy_shape"] = cy.sizes().vec(); ctx->saved_data["c_shape"] = c.sizes().vec();
        ctx->saved_data["t_shape"] = t.sizes().vec(); ctx->saved_data["mu_shape"] = mu.sizes().vec();
        ctx->saved_data["kappa_shape"] = asph ? kappa->sizes().vec() : std::vector<int64_t>{};
        ctx->saved_data["poly_shape"] = asph ? poly->sizes().vec() : std::vector<int64_t>{};
        ctx->saved_data["n_shape"] = want_opd ? n_index->sizes().vec() : std::vector<int64_t>{};
        ctx->set_materialize_grads(false);
        const auto bopt = fopt.dtype(at::kBool);
        Tensor xo, yo, cxo, cyo, ok, back;
        if (want_rays) {
            xo = fp[0].permute({0, 1, 3, 2}); yo = fp[1].permute({0, 1, 3, 2});
            cxo = fp[2].permute({0, 1, 3, 2}); cyo = fp[3].permute({0, 1, 3, 2});
            ok = bp[0].view(at::kBool).permute({0, 1, 3, 2}); back = bp[1].view(at::kBool).permute({0, 1, 3, 2});
        } else {
            xo = at::empty({0}, fopt); yo = at::empty({0}, fopt); cxo = at::empty({0}, fopt); cyo = at::empty({0}, fopt);
            ok = at::empty({0}, bopt); back = at::empty({0}, bopt);
        }
        Tensor opd_out = want_opd ? opd.permute({0, 1, 3, 2}) : at::empty({0}, fopt);
        Tensor stk_out = stacks.defined() ? stacks.permute({0, 1, 2, 3, 5, 4}) : at::empty({0}, fopt);
        Tensor rms_out = rms.defined() ? rms : at::empty({0}, fopt);
        if (want_opd) ctx->mark_non_differentiable({ok, back, stk_out});
        else ctx->mark_non_differentiable({ok, back, stk_out, opd_out});
        if (!rms.defined()) ctx->mark_non_differentiable({rms_out});
        return {xo, yo, cxo, cyo, ok, back, moments, opd_out, stk_out, rms_out};
    }

    static variable_list backward(AutogradContext *ctx, variable_list g)
    {
        const auto sv = ctx->get_saved_variables();
        Norm n;
        n.x_e = sv[0]; n.y_e = sv[1]; n.z = sv[2]; n.cx = sv[3]; n.cy = sv[4]; n.c = sv[5]; n.t = sv[6]; n.mu = sv[7];
        n.mask = sv[8]; n.kappa = sv[9]; n.poly = sv[10]; n.kind = sv[11];
        const Tensor fx = sv[12], fy = sv[13], fcx = sv[14], fcy = sv[15], fok = sv[16], fmom = sv[17];
        n.n_index = sv[18];
        const Tensor hits = sv[19], cond = sv[20], d_rms = sv[21];
        n.B = n.x_e.size(0); n.F = n.x_e.size(1); n.P = n.x_e.size(2); n.W = n.x_e.size(3); n.S = n.c.size(-1);
        const int64_t B = n.B, F = n.F, P = n.P, W = n.W, S = n.S;
        const int64_t flags = ctx->saved_data["flags"].toInt();
        const bool use_inv = ctx->saved_data["use_inv"].toBool();
        const bool allow_back = flags & kAllowBack, aggregate = flags & kAggregate;
        variable_list out(16);
        Tensor gx = g[0], gy = g[1], gcx = g[2], gcy = g[3], gmom = g[6], gopd = g[7];
        if (g.size() > 9 && g[9].defined() && d_rms.defined()) {
            // the fused spot metric: d rms / d moments times its upstream gradient (SpotRmsFn's backward), added to whatever
            // reached the moments directly -- what autograd's accumulation of the two would give
            const Tensor via_rms = B == 1 ? d_rms * g[9] : (d_rms.view({B, -1, TL_NMOM}) * g[9].view({-1, 1, 1})).view_as(d_rms);
            gmom = gmom.defined() ? gmom + via_rms : via_rms;
        }
        if (gopd.defined() && (!n.n_index.defined() || gopd.numel() == 0)) gopd = Tensor();
        if (!gx.defined() && !gy.defined() && !gcx.defined() && !gcy.defined() && !gmom.defined() && !gopd.defined()) return out;
        const bool asph = n.kind.defined();
        const at::Device dev = n.x_e.device();
        c10::hip::HIPGuardMasqueradingAsCUDA guard(dev);
        const hipStream_t st = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(dev.index()).stream();
        tl_problem prob = make_problem(n, allow_back, (int)ctx->saved_data["mode"].toInt(), aggregate, hits, false,
                                       gopd.defined(), cond);
        Tensor ws = workspace(tl_workspace_bytes(&prob), dev, (void *)st);
        auto fwp = [](const Tensor &t) {               // [B,F,P,W] logical -> memory [B,F,W,P] contiguous
            if (!t.defined() || t.numel() == 0) return Tensor();
            return f32(t).permute({0, 1, 3, 2}).contiguous();
        };
        const Tensor gxd = fwp(gx), gyd = fwp(gy), gcxd = fwp(gcx), gcyd = fwp(gcy), gopdd = fwp(gopd);
        const Tensor gmd = gmom.defined() ? gmom.to(at::kDouble).contiguous() : Tensor();
        const bool need_xin = ctx->needs_input_grad(0), need_yin = ctx->needs_input_grad(1);
        const auto fopt = at::TensorOptions().dtype(at::kFloat).device(dev);
        Tensor gxin = need_xin ? at::empty({B, F, W, P}, fopt) : Tensor();
        Tensor gyin = need_yin ? at::empty({B, F, W, P}, fopt) : Tensor();
        // one fp32 tensor per parameter group, written by the reduction kernel (fp64 sums rounded once)
        Tensor g_c = at::empty({B, S}, fopt), g_t = at::empty({B, S}, fopt), g_mu = at::empty({B, W, S}, fopt);
        Tensor g_z = at::empty({B}, fopt), g_cx = at::empty({B, F}, fopt), g_cy = at::empty({B, F}, fopt);
        Tensor g_kappa, g_poly, g_n;
        if (asph) { g_kappa = at::empty({B, S}, fopt); g_poly = at::empty({B, S, 4}, fopt); }
        if (gopdd.defined()) g_n = at::empty({B, W, S + 1}, fopt);
        {
            Timed timed(1, st);
            if (use_inv)
                check(tl_trace_bwd_from_outputs(
                          &prob, (const float *)ptr(gxd), (const float *)ptr(gyd), (const float *)ptr(gcxd),
                          (const float *)ptr(gcyd), (const double *)ptr(gmd), (const float *)ptr(fx), (const float *)ptr(fy),
                          (const float *)ptr(fcx), (const float *)ptr(fcy), (const uint8_t *)ptr(fok), (const double *)ptr(fmom),
                          (float *)g_c.data_ptr(), (float *)g_t.data_ptr(), (float *)g_mu.data_ptr(), (float *)g_z.data_ptr(),
                          (float *)g_cx.data_ptr(), (float *)g_cy.data_ptr(), (float *)ptr(g_kappa), (float *)ptr(g_poly),
                          (float *)ptr(gxin), (float *)ptr(gyin), ws.data_ptr(), (size_t)ws.numel(), (void *)st),
                      "tl_trace_bwd_from_outputs");
            else
                check(tl_trace_bwd(&prob, (const float *)ptr(gxd), (const float *)ptr(gyd), (const float *)ptr(gcxd),
                                   (const float *)ptr(gcyd), (const double *)ptr(gmd), (const float *)ptr(gopdd),
                                   (float *)g_c.data_ptr(), (float *)g_t.data_ptr(), (float *)g_mu.data_ptr(),
                                   (float *)g_z.data_ptr(), (float *)g_cx.data_ptr(), (float *)g_cy.data_ptr(),
                                   (float *)ptr(g_kappa), (float *)ptr(g_poly), (float *)ptr(g_n), (float *)ptr(gxin),
                                   (float *)ptr(gyin), ws.data_ptr(), (size_t)ws.numel(), (void *)st),
                      "tl_trace_bwd");
        }
        auto shape = [&](const char *key) { return ctx->saved_data[key].toIntVector(); };
        // gradients in the callers' shapes: what was broadcast over lenses / fields / wavelengths is summed back
        if (need_xin) out[0] = to_shape(gxin.permute({0, 1, 3, 2}), {B, F, P, W}, shape("x_shape"));
        if (need_yin) out[1] = to_shape(gyin.permute({0, 1, 3, 2}), {B, F, P, W}, shape("y_shape"));
        if (ctx->needs_input_grad(2)) out[2] = to_shape(g_z, {B, 1, 1, 1}, shape("z_shape"));
        if (ctx->needs_input_grad(3)) out[3] = to_shape(g_cx, {B, F, 1, 1}, shape("cx_shape"));
        if (ctx->needs_input_grad(4)) out[4] = to_shape(g_cy, {B, F, 1, 1}, shape("cy_shape"));
        if (ctx->needs_input_grad(5)) out[5] = to_shape(g_c, {B, 1, 1, 1, S}, shape("c_shape"));
        if (ctx->needs_input_grad(6)) out[6] = to_shape(g_t, {B, 1, 1, 1, S}, shape("t_shape"));
        if (ctx->needs_input_grad(7)) out[7] = to_shape(g_mu, {B, 1, 1, W, S}, shape("mu_shape"));
        auto needs = [&](const char *key) {
            const int64_t e = ctx->saved_data[key].toInt();
            return e >= 0 && ctx->needs_input_grad((size_t)e);
        };
        if (asph && needs("e_kappa")) out[9] = at::sum_to(g_kappa, shape("kappa_shape"));      // [S]: summed over lenses
        if (asph && needs("e_poly")) out[10] = at::sum_to(g_poly, shape("poly_shape"));
        if (g_n.defined() && needs("e_n")) {
            const auto ns = shape("n_shape");
            out[12] = ns.size() == 5 ? to_shape(g_n, {B, 1, 1, W, S + 1}, ns) : at::sum_to(g_n, ns);
        }
        return out;
    }
};

// rms = compute_rms2d on the [F, TL_NMOM] moments (closed form) with its derivative: one tiny kernel
class SpotRmsFn : public torch::autograd::Function<SpotRmsFn> {
public:
    static Tensor forward(AutogradContext *ctx, Tensor moments, double n_per_field, int64_t n_lens)
    {
        require_device(moments, "moments");
        const at::Device dev = moments.device();
        c10::hip::HIPGuardMasqueradingAsCUDA guard(dev);
        const hipStream_t st = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(dev.index()).stream();
        const Tensor m = moments.to(at::kDouble).contiguous();
        Tensor rms = n_lens == 1 ? at::empty({}, m.options().dtype(at::kFloat)) : at::empty({n_lens}, m.options().dtype(at::kFloat));
        Tensor dm = at::empty_like(m);
        check(tl_spot_rms(dev.index(), (int32_t)n_lens, (int32_t)(m.size(0) / n_lens), n_per_field, (const double *)m.data_ptr(),
                          (float *)rms.data_ptr(), (double *)dm.data_ptr(), (void *)st),
              "tl_spot_rms");
        ctx->save_for_backward({dm});
        ctx->saved_data["n_lens"] = n_lens;
        return rms;
    }

    static variable_list backward(AutogradContext *ctx, variable_list g)
    {
        const Tensor dm = ctx->get_saved_variables()[0];
        const int64_t n_lens = ctx->saved_data["n_lens"].toInt();
        if (!g[0].defined()) return {Tensor(), Tensor(), Tensor()};
        if (n_lens == 1) return {dm * g[0], Tensor(), Tensor()};       // [F,10] fp64 * 0-dim fp32 -> fp64 in one launch
        return {(dm.view({n_lens, -1, TL_NMOM}) * g[0].view({-1, 1, 1})).view_as(dm), Tensor(), Tensor()};
    }
};

// z [B] = paraxial entrance-pupil position of the rows in front of the stop (tl_pupil_position forward / backward): the
// counterpart of ops.PupilPositionFunction
class PupilPositionFn : public torch::autograd::Function<PupilPositionFn> {
public:
    static Tensor forward(AutogradContext *ctx, const Tensor &c_in, const Tensor &t_in, const Tensor &n_in, int64_t mode)
    {
        require_device(c_in, "c"); require_device(t_in, "t"); require_device(n_in, "n");
        TORCH_CHECK(c_in.dim() == 2, "pupil position: c must be [B,K]");
        const int64_t B = c_in.size(0), K = c_in.size(1);
        TORCH_CHECK(t_in.dim() == 2 && t_in.size(0) == B && t_in.size(1) == K && n_in.dim() == 2 && n_in.size(0) == B &&
                        n_in.size(1) == K + 1,
                    "pupil position: c, t must hold [B,K] rows and n [B,K+1] indices");
        const Tensor c = dense(f32(c_in.detach())), t = dense(f32(t_in.detach())), n = dense(f32(n_in.detach()));
        const at::Device dev = c.device();
        c10::hip::HIPGuardMasqueradingAsCUDA guard(dev);
        const hipStream_t st = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(dev.index()).stream();
        Tensor z = at::empty({B}, c.options());
        check(tl_pupil_position(dev.index(), (int32_t)B, (int32_t)K, (const float *)c.data_ptr(), (const float *)t.data_ptr(),
                                (const float *)n.data_ptr(), (float *)z.data_ptr(), nullptr, nullptr, nullptr, nullptr,
                                (int32_t)mode, (void *)st),
              "tl_pupil_position");
        ctx->save_for_backward({c, t, n});
        ctx->saved_data["mode"] = mode;
        return z;
    }

    static variable_list backward(AutogradContext *ctx, variable_list g)
    {
        if (!g[0].defined()) return {Tensor(), Tensor(), Tensor(), Tensor()};
        const auto sv = ctx->get_saved_variables();
        const Tensor c = sv[0], t = sv[1], n = sv[2];
        const int64_t B = c.size(0), K = c.size(1);
        const Tensor gz = dense(f32(g[0]).reshape({B}));
        const at::Device dev = c.device();
        c10::hip::HIPGuardMasqueradingAsCUDA guard(dev);
        const hipStream_t st = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(dev.index()).stream();
        Tensor g_c = at::empty_like(c), g_t = at::empty_like(t), g_n = at::empty_like(n);
        check(tl_pupil_position(dev.index(), (int32_t)B, (int32_t)K, (const float *)c.data_ptr(), (const float *)t.data_ptr(),
                                (const float *)n.data_ptr(), nullptr, (const float *)gz.data_ptr(), (float *)g_c.data_ptr(),
                                (float *)g_t.data_ptr(), (float *)g_n.data_ptr(), (int32_t)ctx->saved_data["mode"].toInt(),
                                (void *)st),
              "tl_pupil_position (backward)");
        return {g_c, g_t, g_n, Tensor()};
    }
};

Tensor pupil_position(const Tensor &c, const Tensor &t, const Tensor &n, int64_t mode) { return PupilPositionFn::apply(c, t, n, mode); }

// loss_dict of compute_loss_out per lens on the moments of an aggregate trace (tl_unsup_loss): (loss_unsup, rms, penalty),
// each [n_lens] (0-dim for n_lens = 1, like the op sequence of ray_tracing.unsupervised_loss it replaces)
class UnsupLossFn : public torch::autograd::Function<UnsupLossFn> {
public:
    static variable_list forward(AutogradContext *ctx, Tensor moments, double n_per_field, int64_t n_lens,
                                 const c10::optional<Tensor> &n_seq, double n_seq_all, double rate)
    {
        require_device(moments, "moments");
        const at::Device dev = moments.device();
        c10::hip::HIPGuardMasqueradingAsCUDA guard(dev);
        const hipStream_t st = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(dev.index()).stream();
        const Tensor m = moments.to(at::kDouble).contiguous();
        TORCH_CHECK(m.dim() == 2 && m.size(1) == TL_NMOM && n_lens >= 1 && m.size(0) % n_lens == 0,
                    "unsup_loss: moments must be [n_lens * F, TL_NMOM]");
        Tensor ns;
        if (n_seq && n_seq->defined()) {
            ns = n_seq->to(dev, at::kDouble).reshape({-1}).contiguous();
            TORCH_CHECK(ns.numel() == n_lens, "n_sequence must hold one entry per lens");
        }
        const auto fopt = m.options().dtype(at::kFloat);
        const auto shape = n_lens == 1 ? std::vector<int64_t>{} : std::vector<int64_t>{n_lens};
        Tensor loss = at::empty(shape, fopt), rms = at::empty(shape, fopt), pen = at::empty(shape, fopt);
        Tensor dm = at::empty_like(m);
        check(tl_unsup_loss(dev.index(), (int32_t)n_lens, (int32_t)(m.size(0) / n_lens), n_per_field, (const double *)m.data_ptr(),
                            (const double *)ptr(ns), n_seq_all, (float)rate, (float *)loss.data_ptr(), (float *)rms.data_ptr(),
                            (float *)pen.data_ptr(), (double *)dm.data_ptr(), (void *)st),
              "tl_unsup_loss");
        ctx->save_for_backward({dm, ns});
        ctx->saved_data["n_lens"] = n_lens;
        ctx->saved_data["n_seq_all"] = n_seq_all;
        ctx->saved_data["rate"] = rate;
        return {loss, rms, pen};
    }

    static variable_list backward(AutogradContext *ctx, variable_list g)
    {
        variable_list out(6);
        if (!g[0].defined() && !g[1].defined() && !g[2].defined()) return out;
        const auto sv = ctx->get_saved_variables();
        const Tensor dm = sv[0], ns = sv[1];
        const int64_t n_lens = ctx->saved_data["n_lens"].toInt();
        const at::Device dev = dm.device();
        c10::hip::HIPGuardMasqueradingAsCUDA guard(dev);
        const hipStream_t st = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(dev.index()).stream();
        // upstream gradients: [n_lens] dense, or one value for every lens (an expanded scalar: stride 0)
        Tensor gg[3];
        bool uniform = n_lens > 1;
        for (int i = 0; i < 3; ++i) {
            if (!g[i].defined()) continue;
            gg[i] = f32(g[i]).reshape({-1});
            if (n_lens > 1 && gg[i].stride(0) != 0) uniform = false;
        }
        if (!uniform)
            for (auto &t : gg) if (t.defined()) t = t.contiguous();
        const int64_t stride = uniform ? 0 : 1;
        Tensor gm = at::empty_like(dm);
        check(tl_unsup_loss_bwd(dev.index(), (int32_t)n_lens, (int32_t)(dm.size(0) / n_lens), (const double *)dm.data_ptr(),
                                (const float *)ptr(gg[0]), (const float *)ptr(gg[1]), (const float *)ptr(gg[2]), (int32_t)stride,
                                (const double *)ptr(ns), ctx->saved_data["n_seq_all"].toDouble(),
                                (float)ctx->saved_data["rate"].toDouble(), (double *)gm.data_ptr(), (void *)st),
              "tl_unsup_loss_bwd");
        out[0] = gm;
        return out;
    }
};

std::vector<Tensor> unsup_loss(const Tensor &moments, double n_per_field, int64_t n_lens, const c10::optional<Tensor> &n_seq,
                               double n_seq_all, double rate)
{
    return UnsupLossFn::apply(moments, n_per_field, n_lens, n_seq, n_seq_all, rate);
}

std::vector<Tensor> trace(const Tensor &x, const Tensor &y, const Tensor &z, const Tensor &cx, const Tensor &cy, const Tensor &c,
                          const Tensor &t, const Tensor &mu, const Tensor &mask, const c10::optional<Tensor> &kappa,
                          const c10::optional<Tensor> &poly, const c10::optional<Tensor> &kind,
                          const c10::optional<Tensor> &n_index, int64_t flags, int64_t mode, int64_t hit_slots)
{
    return TraceFn::apply(x, y, z, cx, cy, c, t, mu, mask, kappa, poly, kind, n_index, flags, mode, hit_slots);
}

Tensor spot_rms(const Tensor &moments, double n_per_field, int64_t n_lens)
{
    return SpotRmsFn::apply(moments, n_per_field, n_lens);
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
    m.doc() = "C++ autograd wrappers of libtltrace.so's C ABI (the eager host chain of torchoptics_amd)";
    m.def("trace", &trace);
    m.def("spot_rms", &spot_rms);
    m.def("pupil_position", &pupil_position);
    m.def("unsup_loss", &unsup_loss);
    m.def("last_use_inv", [] { return g_last_use_inv; });
    m.def("enable_timing", &enable_timing);
    m.def("timing_ms", &timing_ms);
    m.def("timing_counts", &timing_counts);
    m.def("abi_version", [] { return tl_version(); });
    m.attr("ALLOW_BACK") = (int)kAllowBack; m.attr("WANT_RAYS") = (int)kWantRays; m.attr("WANT_OPD") = (int)kWantOpd;
    m.attr("AGGREGATE") = (int)kAggregate; m.attr("WANT_STACKS") = (int)kWantStacks; m.attr("MOMENTS_X") = (int)kMomentsX;
    m.attr("INVERSE") = (int)kInverse; m.attr("FUSE_RMS") = (int)kFuseRms;
}
