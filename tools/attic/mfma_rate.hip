// Microbenchmark: issue rate of v_mfma_f32_32x32x2_f32 under different dependency patterns (diagnostic only).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));
template <int NACC, int RUN>   // NACC accumulators, RUN consecutive MFMAs per accumulator before switching
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0)
{
    floatx16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x, b = b0 + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i)
#pragma unroll
            for (int r = 0; r < RUN; ++r) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC, int RUN> void run(const char* name, int waves_per_simd)
{
    float* out; hipMalloc(&out, 256 * 2048 * 4);
    const int iters = 4096 / (NACC * RUN) * 8;
    const int blocks = 256 * waves_per_simd;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NACC, RUN>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, RUN>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mf = (double)iters * NACC * RUN;            // MFMAs per wave
    const double flops = mf * 4096.0 * blocks * 4;
    printf("%-28s waves/SIMD %d: %.3f ms, %.1f TFLOP/s, %.1f ns per MFMA per SIMD (=%.1f cyc @2.4GHz)\n", name, waves_per_simd, ms,
           flops / ms / 1e9, ms * 1e6 / (mf * waves_per_simd), ms * 1e6 / (mf * waves_per_simd) * 2.4);
    hipFree(out);
}
int main()
{
    run<4, 1>("4 acc round-robin", 1);
    run<3, 1>("3 acc round-robin", 1);
    run<1, 1>("1 acc dependent chain", 1);
    run<3, 4>("3 acc, runs of 4 dependent", 1);
    run<4, 1>("4 acc round-robin", 2);
    run<1, 1>("1 acc dependent chain", 2);
    return 0;
}
