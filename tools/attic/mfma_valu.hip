// Microbenchmark (diagnostic): do packed fp32 FMAs on the vector ALU run in the shadow of fp32 MFMAs?
// Per loop iteration a wave issues G v_mfma_f32_32x32x2_f32 and P v_pk_fma_f32 (independent accumulators; the packed
// FMA takes its "weight" pair from SGPRs or VGPRs). Reported: SIMD cycles per iteration (s_memtime), the shader clock
// (s_memtime / s_memrealtime) and the combined fp32 rate against the MFMA-only rate.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_valu tools/mfma_valu.hip && /tmp/mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));

template <int G, int P, int SG>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* stamps, const float* sw, int iters)
{
    floatx16 acc[2];
    for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    v2f va[16];
    for (int i = 0; i < 16; ++i) va[i] = v2f{0.f, 0.f};
    float a = threadIdx.x * 1e-3f, b = 1.0f;
    v2f x = {a, a};
    unsigned sacc = 0;
    // uniform weight pairs (SGPRs when SG)
    v2f w0 = {sw[0], sw[1]}, w1 = {sw[2], sw[3]};
    if (!SG) { w0.x += threadIdx.x * 1e-9f; w1.x += threadIdx.x * 1e-9f; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < (G > P ? G : P); ++g) {
            if (g < G) acc[g & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[g & 1], 0, 0, 0);
#pragma unroll
            for (int p = 0; p < (G ? (P + G - 1) / G : 1); ++p) {
                const int q = g * (G ? (P + G - 1) / G : 1) + p;
                if (q < P) {
                    if (SG == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(va[q & 15]) : "v"(x), "s"((q & 1) ? w1 : w0));
                    else if (SG == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(va[q & 15]) : "v"(x), "v"((q & 1) ? w1 : w0));
                    else if (SG == 2) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(va[q & 15].x) : "v"(x.x), "v"(w0.x));
                    else if (SG == 3) asm volatile("v_add_u32 %0, %1, %0" : "+v"(va[q & 15].x) : "v"(x.x));
                    else if (SG == 4) asm volatile("v_exp_f32 %0, %1" : "=v"(va[q & 15].x) : "v"(x.x));
                    else if (SG == 5) asm volatile("v_mov_b32 %0, %1" : "=v"(va[q & 15].x) : "v"(x.x));
                    else if (SG == 6) asm volatile("v_max_f32 %0, %1, %0" : "+v"(va[q & 15].x) : "v"(x.x));
                    else if (SG == 7) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(va[q & 15]) : "v"(x));
                    else if (SG == 8) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(va[q & 15].x));
                    else if (SG == 9) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sacc));
                }
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = (float)sacc;
    for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 16; ++i) s += va[i].x + va[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { stamps[0] = t1 - t0; stamps[1] = r1 - r0; }
}

static const char* names[] = {"pk_fma vgpr", "pk_fma sgpr", "v_fma_f32", "v_add_u32", "v_exp_f32", "v_mov_b32", "v_max_f32", "v_pk_mul_f32", "v_lshlrev", "s_add_u32"};
template <int G, int P, int SG> void run(int wps, float* out, unsigned long long* st, float* sw)
{
    const int iters = 20000;
    hipLaunchKernelGGL((k<G, P, SG>), dim3(256 * wps), dim3(256), 0, 0, out, st, sw, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<G, P, SG>), dim3(256 * wps), dim3(256), 0, 0, out, st, sw, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; hipMemcpy(h, st, 16, hipMemcpyDeviceToHost);
    const double cyc = (double)h[0] / iters / wps;          // SIMD cycles per iteration (per wave slot)
    const double ghz = (double)h[0] / ((double)h[1] * 10.0);  // realtime counter: 100 MHz
    const double flop_iter = G * 4096.0 + P * 256.0;         // per wave
    const double tflops = flop_iter * iters * 1024.0 * wps / (ms * 1e-3) / 1e12;
    printf("G=%2d mfma  P=%2d x %-12s  waves/SIMD %d: %7.1f cyc/iter  clock %.2f GHz  %.3f ms  %6.1f TFLOP/s (mfma part %6.1f)\n", G, P,
           names[SG], wps, cyc, ghz, ms, tflops, G * 4096.0 * iters * 1024.0 * wps / (ms * 1e-3) / 1e12);
}

int main()
{
    float *out, *sw; unsigned long long* st;
    hipMalloc(&out, 256 * 4096 * 4); hipMalloc(&st, 64); hipMalloc(&sw, 64);
    float hw[4] = {1e-3f, 2e-3f, 3e-3f, 4e-3f};
    hipMemcpy(sw, hw, 16, hipMemcpyHostToDevice);
    for (int wps = 1; wps <= 2; ++wps) {
        run<4, 0, 1>(wps, out, st, sw);
        run<0, 32, 1>(wps, out, st, sw);
        run<4, 32, 1>(wps, out, st, sw);
        run<0, 32, 2>(wps, out, st, sw); run<4, 32, 2>(wps, out, st, sw);
        run<0, 32, 3>(wps, out, st, sw); run<4, 32, 3>(wps, out, st, sw);
        run<0, 32, 4>(wps, out, st, sw); run<4, 32, 4>(wps, out, st, sw);
        run<0, 32, 5>(wps, out, st, sw); run<4, 32, 5>(wps, out, st, sw);
        run<0, 32, 6>(wps, out, st, sw); run<4, 32, 6>(wps, out, st, sw);
        run<0, 32, 7>(wps, out, st, sw); run<4, 32, 7>(wps, out, st, sw);
        run<0, 32, 8>(wps, out, st, sw); run<4, 32, 8>(wps, out, st, sw);
        run<0, 32, 9>(wps, out, st, sw); run<4, 32, 9>(wps, out, st, sw);
    }
    return 0;
}
