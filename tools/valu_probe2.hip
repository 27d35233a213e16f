// valu_probe2.hip -- second round: what limits a wave's VALU issue on gfx950 -- dependency distance (ILP),
// instruction encoding size (4-byte VOP2 vs 8-byte VOP3), the length of the loop body (instruction fetch), and
// how both scale with the waves per SIMD.  Every instruction is an `asm volatile`, so the order is exactly the
// one written: chain c of CH independent chains is touched every CH-th instruction.
//
//   hipcc --offload-arch=gfx950 -O3 -o build/valu_probe2 tools/valu_probe2.hip && build/valu_probe2
//
// Output: SIMD cycles per wave-instruction from the wall time of the launch (HIP events) and the in-kernel clock
// (s_memtime / s_memrealtime); 2.0 = the VALU peak.  Development tool, see DESIGN.md section 4.
#include <hip/hip_runtime.h>
#include <stdio.h>

enum { VOP2_FMAC = 0, VOP3_FMA = 1, VOP2_MUL = 2, VOP3_FMA_SGPR = 3 };

template <int KIND>
__device__ __forceinline__ void op(float &v, float a, float b, float sa)
{
    if (KIND == VOP2_FMAC) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v) : "v"(a), "v"(b));             // 4 bytes
    else if (KIND == VOP3_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(a), "v"(b));     // 8 bytes
    else if (KIND == VOP2_MUL) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(v) : "v"(a));                 // 4 bytes
    else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v) : "s"(sa), "v"(b));                          // 8 bytes, SGPR operand
}

// BODY instructions per loop iteration, CH independent chains
template <int KIND, int CH, int BODY>
__global__ __launch_bounds__(256) void probe(float *out, int iters, float a, float b, unsigned long long *clk)
{
    extern __shared__ float lds[];
    float v[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) v[c] = 1.0f + 0.001f * (threadIdx.x + c);
    float av = a + 1e-9f * threadIdx.x, bv = b;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < BODY / CH; ++u) {
#pragma unroll
            for (int c = 0; c < CH; ++c) op<KIND>(v[c], av, bv, a);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.0f;
#pragma unroll
    for (int c = 0; c < CH; ++c) s += v[c];
    if (s == 123.456f) out[0] = s + lds[0];
    if (threadIdx.x == 0 && blockIdx.x == gridDim.x / 2) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int KIND, int CH, int BODY>
static void run(const char *name, float *out, unsigned long long *clk)
{
    for (int waves : {1, 2, 4, 8}) {
        const size_t lds = (160 * 1024) / waves - 1024;
        (void)hipFuncSetAttribute((const void *)probe<KIND, CH, BODY>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        const int grid = 256 * waves;
        const int iters = (1 << 21) / BODY;                       // 2 M instructions per wave
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        probe<KIND, CH, BODY><<<grid, 256, lds>>>(out, 16, 0.999f, 0.001f, clk);
        (void)hipEventRecord(e0);
        probe<KIND, CH, BODY><<<grid, 256, lds>>>(out, iters, 0.999f, 0.001f, clk);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[2];
        (void)hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
        const double ghz = (double)h[0] / ((double)h[1] * 10.0);
        const double winstr = (double)iters * BODY * waves;        // wave-instructions per SIMD
        printf("%-14s chains %d body %4d waves/SIMD %d : %5.2f cycles per wave-instruction (%.3f ms, %.2f GHz)\n", name, CH, BODY,
               waves, ms * 1e-3 * ghz * 1e9 / winstr, ms, ghz);
    }
}

int main()
{
    float *out;
    unsigned long long *clk;
    (void)hipMalloc(&out, 1024);
    (void)hipMalloc(&clk, 64);
    run<VOP2_FMAC, 1, 32>("vop2 fmac", out, clk);
    run<VOP2_FMAC, 2, 32>("vop2 fmac", out, clk);
    run<VOP2_FMAC, 4, 32>("vop2 fmac", out, clk);
    run<VOP2_FMAC, 8, 32>("vop2 fmac", out, clk);
    run<VOP3_FMA, 1, 32>("vop3 fma", out, clk);
    run<VOP3_FMA, 2, 32>("vop3 fma", out, clk);
    run<VOP3_FMA, 4, 32>("vop3 fma", out, clk);
    run<VOP3_FMA, 8, 32>("vop3 fma", out, clk);
    run<VOP3_FMA_SGPR, 2, 32>("vop3 fma sgpr", out, clk);
    run<VOP3_FMA_SGPR, 4, 32>("vop3 fma sgpr", out, clk);
    run<VOP2_FMAC, 2, 1024>("vop2 fmac", out, clk);
    run<VOP2_FMAC, 4, 1024>("vop2 fmac", out, clk);
    run<VOP3_FMA, 2, 1024>("vop3 fma", out, clk);
    run<VOP3_FMA, 4, 1024>("vop3 fma", out, clk);
    run<VOP3_FMA, 2, 4096>("vop3 fma", out, clk);
    run<VOP3_FMA, 4, 4096>("vop3 fma", out, clk);
    run<VOP2_FMAC, 4, 4096>("vop2 fmac", out, clk);
    return 0;
}
