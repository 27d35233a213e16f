// valu_probe3.hip -- third round: are the "light" VALU instructions of the trace kernels (register moves, selects)
// priced like the arithmetic ones?  A body of NF dependent-chain FMAs (two chains, all-VGPR operands) is timed alone and
// with NM extra instructions of one kind interleaved one-to-one: if the extra instructions were free the time would
// not move, if they cost a full issue slot it grows by NM/NF.
//
//   hipcc --offload-arch=gfx950 -O3 -o build/valu_probe3 tools/valu_probe3.hip && build/valu_probe3
#include <hip/hip_runtime.h>
#include <stdio.h>

enum { NONE = 0, MOV = 1, CNDMASK = 2, ADD = 3, FMA_SGPR = 4, MUL_OTHER = 5 };

template <int KIND>
__device__ __forceinline__ void extra(float &w, float a, float sa)
{
    if (KIND == MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(w) : "v"(a));
    else if (KIND == CNDMASK) asm volatile("v_cndmask_b32 %0, %1, %0, vcc" : "+v"(w) : "v"(a));
    else if (KIND == ADD) asm volatile("v_add_f32 %0, %1, %0" : "+v"(w) : "v"(a));
    else if (KIND == FMA_SGPR) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(w) : "s"(sa), "v"(a));
    else if (KIND == MUL_OTHER) asm volatile("v_mul_f32 %0, %1, %1" : "=v"(w) : "v"(a));
}

template <int KIND>
__global__ __launch_bounds__(256) void probe(float *out, int iters, float a, float b, unsigned long long *clk)
{
    extern __shared__ float lds[];
    float v0 = 1.0f + 0.001f * threadIdx.x, v1 = 1.5f + 0.001f * threadIdx.x, w = 0.25f;
    float av = a + 1e-9f * threadIdx.x, bv = b;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v0) : "v"(av), "v"(bv));
            if (KIND != NONE) extra<KIND>(w, av, a);
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v1) : "v"(av), "v"(bv));
            if (KIND != NONE) extra<KIND>(w, bv, a);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (v0 + v1 + w == 123.456f) out[0] = v0 + lds[0];
    if (threadIdx.x == 0 && blockIdx.x == gridDim.x / 2) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int KIND>
static void run(const char *name, float *out, unsigned long long *clk)
{
    for (int waves : {2, 4, 6, 8}) {
        const size_t lds = (160 * 1024) / waves - 1024;
        (void)hipFuncSetAttribute((const void *)probe<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        const int iters = 1 << 14;                                 // 2^20 FMAs per wave
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        probe<KIND><<<256 * waves, 256, lds>>>(out, 16, 0.999f, 0.001f, clk);
        (void)hipEventRecord(e0);
        probe<KIND><<<256 * waves, 256, lds>>>(out, iters, 0.999f, 0.001f, clk);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[2];
        (void)hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
        const double ghz = (double)h[0] / ((double)h[1] * 10.0);
        const double fmas = (double)iters * 64 * waves;            // FMAs per SIMD
        printf("64 FMAs + 64 x %-10s waves/SIMD %d : %5.2f cycles per FMA (%.3f ms, %.2f GHz)\n", name, waves,
               ms * 1e-3 * ghz * 1e9 / fmas, ms, ghz);
    }
}

int main()
{
    float *out;
    unsigned long long *clk;
    (void)hipMalloc(&out, 1024);
    (void)hipMalloc(&clk, 64);
    run<NONE>("nothing", out, clk);
    run<MOV>("v_mov", out, clk);
    run<CNDMASK>("v_cndmask", out, clk);
    run<ADD>("v_add", out, clk);
    run<MUL_OTHER>("v_mul", out, clk);
    run<FMA_SGPR>("fma(sgpr)", out, clk);
    return 0;
}
