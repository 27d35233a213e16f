// valu_probe.hip -- what one gfx950 SIMD sustains on the instruction mixes of the ray-trace kernels.
//
//   hipcc --offload-arch=gfx950 -O3 -o build/valu_probe tools/valu_probe.hip && build/valu_probe
//
// Every lane runs `chains` independent dependency chains of one operation for `iters` rounds; occupancy (waves per
// SIMD) is set by the dynamic LDS size of the 256-thread blocks (one wave per SIMD each) on a grid of exactly
// 256 CUs x waves blocks.  Prints SIMD cycles per wave-instruction = wall time of the launch (HIP events) x the
// in-kernel clock (s_memtime / s_memrealtime) / instructions per SIMD: 2.0 = the VALU peak (one wave64 instruction
// per 2 cycles).  The chains are written in C, so which operands sit in SGPRs is the compiler's choice (the MIX_ROW
// FMAs read one: that is the half-rate case of valu_probe2.hip).
// Development tool (DESIGN.md section 4 "what bounds the walk-back kernel"), not part of the product.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

enum Op { FMA = 0, RCP = 1, RSQ = 2, SQRT = 3, RSQ_NEWTON = 4, DPP_ADD = 5, MIX_ROW = 6 };

template <int OP>
__device__ __forceinline__ float step(float v, float a, float b)
{
    if (OP == FMA) return __builtin_fmaf(v, a, b);
    if (OP == RCP) return __builtin_amdgcn_rcpf(v);
    if (OP == RSQ) return __builtin_amdgcn_rsqf(v);
    if (OP == SQRT) return __builtin_amdgcn_sqrtf(v);
    if (OP == RSQ_NEWTON) {      // tl_rsq of the strict kernels: v_rsq + 4 plain ops, all dependent
        const float r = __builtin_amdgcn_rsqf(v);
        const float e = __builtin_fmaf(-(v * r), r, 1.0f);
        return __builtin_fmaf(0.5f * r, e, r) + a;
    }
    if (OP == DPP_ADD)
        return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));
    // MIX_ROW: 28 dependent FMAs and one v_rsq, the mix of one surface row (5 transcendentals in ~150 instructions)
    float x = v;
#pragma unroll
    for (int j = 0; j < 28; ++j) x = __builtin_fmaf(x, a, b);
    return __builtin_amdgcn_rsqf(x * x + 1.0f);
}

template <int OP, int CHAINS>
__global__ __launch_bounds__(256) void probe(float *out, int iters, float a, float b, unsigned long long *clk)
{
    extern __shared__ float lds[];
    float v[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) v[c] = 1.0f + 0.001f * (threadIdx.x + c);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) v[c] = step<OP>(v[c], a, b);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.0f;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) s += v[c];
    if (s == 123.456f) out[0] = s + lds[0];     // never true: keeps the chains alive
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int OP, int CHAINS>
static void run(const char *name, int instr_per_step, float *out, unsigned long long *clk)
{
    const int iters = 20000;
    for (int waves : {1, 2, 3, 4, 5, 6, 8}) {
        const size_t lds = (160 * 1024) / waves - 1024;
        hipFuncSetAttribute((const void *)probe<OP, CHAINS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        const int grid = 256 * waves;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        probe<OP, CHAINS><<<grid, 256, lds>>>(out, 10, 0.999f, 0.001f, clk);       // warm-up
        hipEventRecord(e0);
        probe<OP, CHAINS><<<grid, 256, lds>>>(out, iters, 0.999f, 0.001f, clk);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[2];
        hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
        const double ghz = (double)h[0] / ((double)h[1] * 10.0);       // s_memrealtime ticks at 100 MHz
        const double winstr = (double)iters * 8 * CHAINS * instr_per_step * waves;     // wave-instructions per SIMD
        const double cyc = ms * 1e-3 * ghz * 1e9 / winstr;                             // wall time x in-kernel clock
        printf("%-12s chains %d waves/SIMD %d : %6.2f cycles per wave-instruction  (%.3f ms, in-kernel clock %.2f GHz)\n", name,
               CHAINS, waves, cyc, ms, ghz);
    }
}

int main()
{
    float *out;
    unsigned long long *clk;
    hipMalloc(&out, 1024);
    hipMalloc(&clk, 64);
    run<FMA, 1>("fma", 1, out, clk);
    run<FMA, 2>("fma", 1, out, clk);
    run<FMA, 4>("fma", 1, out, clk);
    run<RCP, 1>("rcp", 1, out, clk);
    run<RCP, 4>("rcp", 1, out, clk);
    run<RSQ, 4>("rsq", 1, out, clk);
    run<SQRT, 4>("sqrt", 1, out, clk);
    run<RSQ_NEWTON, 1>("rsq+newton", 6, out, clk);
    run<DPP_ADD, 1>("dpp_add", 1, out, clk);
    run<DPP_ADD, 4>("dpp_add", 1, out, clk);
    run<MIX_ROW, 1>("row mix", 30, out, clk);
    run<MIX_ROW, 2>("row mix", 30, out, clk);
    return 0;
}
