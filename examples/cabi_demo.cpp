// cabi_demo.cpp -- drives libtltrace.so through its C ABI only (include/tl_trace.h): no Python, no torch.
// Traces a circular fan through the reference's singlet (data/singlet_lens.yml values), forms the RMS spot
// from the fused moments, back-propagates it, and prints the numbers as one JSON line.
//
//   hipcc --offload-arch=gfx950 -O2 examples/cabi_demo.cpp -Iinclude -Ltorchoptics_amd -ltltrace \
//         -Wl,-rpath,'$ORIGIN/../torchoptics_amd' -o examples/cabi_demo.bin
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#include "tl_trace.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define TL_OK_(x) do { int r_ = (x); if (r_ != 0) { fprintf(stderr, "%s -> %d: %s\n", #x, r_, tl_last_error()); return 3; } } while (0)

template <class T> static T *to_device(const std::vector<T> &h)
{
    T *d = nullptr;
    if (hipMalloc((void **)&d, h.size() * sizeof(T)) != hipSuccess) return nullptr;
    if (hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    return d;
}

int main(int argc, char **argv)
{
    const int n_r = argc > 1 ? atoi(argv[1]) : 64, n_t = argc > 2 ? atoi(argv[2]) : 64;
    const int P = n_r * n_t, F = 1, W = 1, S = 3;
    if (tl_version() != TL_ABI_VERSION || tl_problem_size() != sizeof(tl_problem)) { fprintf(stderr, "ABI mismatch\n"); return 1; }
    // singlet: rows A G A (stop first), d line; n = nd
    const float nd = 1.916499376296997f, epd = 8.57803f;
    std::vector<float> c = {0.0f, 0.01867167465388775f, -0.04616425931453705f};
    std::vector<float> t = {6.715000152587891f, 3.0007503032684326f, 15.0230131149292f};
    std::vector<float> mu = {1.0f / 1.0f, 1.0f / nd, nd / 1.0f};          // n_before / n_after per row
    std::vector<uint8_t> mask = {1, 1, 1};
    std::vector<float> z = {0.0f}, cx = {0.0f}, cy = {0.0f};
    std::vector<float> x(P), y(P);
    for (int i = 0; i < n_r; ++i)
        for (int j = 0; j < n_t; ++j) {
            const float r = (float)i / n_r, th = 2.0f * (float)M_PI * j / n_t;
            x[i * n_t + j] = r * cosf(th) * epd / 2;
            y[i * n_t + j] = r * sinf(th) * epd / 2;
        }
    tl_problem p = {};
    p.F = F; p.P = P; p.W = W; p.S = S; p.device = 0; p.mode = TL_MODE_STRICT; p.allow_backward = 1;
    p.x_in = to_device(x); p.y_in = to_device(y); p.xs_p = p.ys_p = 1;
    p.z = to_device(z); p.cx = to_device(cx); p.cy = to_device(cy); p.cx_stride = p.cy_stride = 0;
    p.c = to_device(c); p.t = to_device(t); p.mu = to_device(mu); p.mask = to_device(mask);
    float *ox, *oy, *ocx, *ocy, *rms, *g_c, *g_t, *g_mu, *g_z, *g_cx, *g_cy;
    uint8_t *ok, *back;
    double *mom, *dmom;
    void *ws;
    const size_t wsz = tl_workspace_bytes(&p);
    HIP_OK(hipMalloc((void **)&ox, P * 4)); HIP_OK(hipMalloc((void **)&oy, P * 4));
    HIP_OK(hipMalloc((void **)&ocx, P * 4)); HIP_OK(hipMalloc((void **)&ocy, P * 4));
    HIP_OK(hipMalloc((void **)&ok, P)); HIP_OK(hipMalloc((void **)&back, P));
    HIP_OK(hipMalloc((void **)&mom, F * TL_NMOM * 8)); HIP_OK(hipMalloc((void **)&dmom, F * TL_NMOM * 8));
    HIP_OK(hipMalloc((void **)&rms, 4)); HIP_OK(hipMalloc(&ws, wsz));
    HIP_OK(hipMalloc((void **)&g_c, S * 4)); HIP_OK(hipMalloc((void **)&g_t, S * 4)); HIP_OK(hipMalloc((void **)&g_mu, W * S * 4));
    HIP_OK(hipMalloc((void **)&g_z, 4)); HIP_OK(hipMalloc((void **)&g_cx, F * 4)); HIP_OK(hipMalloc((void **)&g_cy, F * 4));
    hipStream_t st;
    HIP_OK(hipStreamCreate(&st));
    TL_OK_(tl_trace_fwd(&p, ox, oy, ocx, ocy, ok, back, nullptr, nullptr, mom, ws, wsz, st));
    TL_OK_(tl_spot_rms(0, 1, F, (double)P * W, mom, rms, dmom, st));
    TL_OK_(tl_trace_bwd(&p, nullptr, nullptr, nullptr, nullptr, dmom, nullptr, g_c, g_t, g_mu, g_z, g_cx, g_cy, nullptr, nullptr,
                        nullptr, nullptr, nullptr, ws, wsz, st));
    HIP_OK(hipStreamSynchronize(st));
    float h_rms, h_gc[3], h_gt[3], h_gmu[3];
    double h_mom[TL_NMOM];
    HIP_OK(hipMemcpy(&h_rms, rms, 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(h_gc, g_c, 12, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(h_gt, g_t, 12, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(h_gmu, g_mu, 12, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(h_mom, mom, sizeof(h_mom), hipMemcpyDeviceToHost));
    printf("{\"rays\": %d, \"ok\": %.0f, \"rms\": %.9g, \"g_c\": [%.9g, %.9g, %.9g], \"g_t\": [%.9g, %.9g, %.9g], "
           "\"g_mu\": [%.9g, %.9g, %.9g]}\n", P, h_mom[3], h_rms, h_gc[0], h_gc[1], h_gc[2], h_gt[0], h_gt[1], h_gt[2],
           h_gmu[0], h_gmu[1], h_gmu[2]);
    return 0;
}
