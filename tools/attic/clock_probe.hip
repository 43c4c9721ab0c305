// clock_probe.hip -- what s_memtime counts and how fast the matrix pipe really runs: a kernel of back-to-back bf16 MFMAs (one to four
// waves per SIMD, every CU) and one that only sleeps, each timed by HIP events and by s_memtime inside.
//   hipcc --offload-arch=gfx950 -O2 tools/attic/clock_probe.hip -o build/clock_probe && build/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void mfma_loop(float* out, unsigned long long* ticks, int iters, int nacc)
{
    floatx16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (i + 1)); }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (nacc == 4)
        for (int i = 0; i < iters; ++i) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, a2, 0, 0, 0); a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, a3, 0, 0, 0);
        }
    else
        for (int i = 0; i < iters; ++i) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, a0, 0, 0, 0); a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, a0, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, a0, 0, 0, 0); a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, a0, 0, 0, 0);
        }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
    if (blockIdx.x == 0 && threadIdx.x == 0) ticks[0] = t1 - t0;
}
__global__ __launch_bounds__(256) void sleep_loop(unsigned long long* ticks, int iters)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) __builtin_amdgcn_s_sleep(64);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (blockIdx.x == 0 && threadIdx.x == 0) ticks[0] = t1 - t0;
}

int main()
{
    float* out; unsigned long long* ticks;
    CK(hipMalloc(&out, 4096 * 256 * 4)); CK(hipMalloc(&ticks, 8));
    int cus = 0; CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto report = [&](const char* what, double mfmas_per_wave) {
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long t; CK(hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost));
        printf("%-58s %8.1f us, %10llu s_memtime ticks = %7.1f ticks/us", what, ms * 1e3, t, t / (ms * 1e3));
        if (mfmas_per_wave > 0) printf(", %.1f ns per MFMA per wave", ms * 1e6 / mfmas_per_wave);
        printf("\n");
    };
    const int iters = 100000;
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(sleep_loop, dim3(cus), dim3(256), 0, 0, ticks, 20000); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        report("sleep only, 1 workgroup per CU", 0);
        for (int wg = 1; wg <= 2; ++wg)
            for (int nacc : {4, 1}) {
                CK(hipEventRecord(e0)); hipLaunchKernelGGL(mfma_loop, dim3(cus * wg), dim3(256), 0, 0, out, ticks, iters, nacc); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                char buf[128]; snprintf(buf, sizeof buf, "MFMA 32x32x16 bf16, %d wave(s) per SIMD, %s", wg, nacc == 4 ? "4 accumulators" : "1 accumulator (dependent)");
                report(buf, 4.0 * iters);
            }
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(mfma_loop, dim3(cus / 8), dim3(256), 0, 0, out, ticks, iters, 4); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        report("MFMA, 1 wave per SIMD, 4 accumulators, 32 CUs only", 4.0 * iters);
    }
    return 0;
}
