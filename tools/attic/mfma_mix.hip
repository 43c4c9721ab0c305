// Microbenchmark (diagnostic): how much MFMA-pipe time does one extra memory/LDS instruction cost
// when it is slotted between fp32 MFMAs, at 1 and 2 waves per SIMD?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) v4f* gptr4;

// MODE 0: MFMA only; 1: + 1 global_load_dwordx4 per G MFMAs (L2-resident 1 MiB panel per wave... shared);
// 2: + 1 ds_read_b128 per G; 3: + 1 ds_write_b128 per G; 4: + 4 v_cndmask per G
template <int MODE, int G>
__global__ __launch_bounds__(256) void k(float* out, const float* src, int iters)
{
    __shared__ __attribute__((aligned(16))) float lds[256 * 4 * 2];
    floatx16 acc[3];
    for (int i = 0; i < 3; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const float* p = src + ((size_t)(blockIdx.x % 64) * 65536 + threadIdx.x * 4);
    v4f a = {1.f, 2.f, 3.f, 4.f};
    v4f ld = a;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 1) { ld = *(gptr4)(p); p += 1024; if ((it & 63) == 63) p -= 65536; }
        if (MODE == 2) ld = *(v4f*)&lds[threadIdx.x * 4 + (it & 1) * 1024];
        if (MODE == 3) *(v4f*)&lds[threadIdx.x * 4 + (it & 1) * 1024] = a;
        if (MODE == 4) { a.x = (it & 1) ? a.x : 0.f; a.y = (it & 2) ? a.y : 0.f; a.z = (it & 4) ? a.z : 0.f; a.w = (it & 8) ? a.w : 0.f; }
#pragma unroll
        for (int g = 0; g < G; ++g)
            acc[g % 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, a.y, acc[g % 3], 0, 0, 0);
        if (MODE == 1 || MODE == 2) a.x += ld.x * 1e-30f;   // consume one iteration later-ish (forces the wait here)
    }
    float s = a.x;
    for (int i = 0; i < 3; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE, int G> void run(const char* name, int wps, float* out, float* src, double base[3])
{
    const int iters = 4096 / G * 4;
    const int blocks = 256 * wps;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, G>), dim3(blocks), dim3(256), 0, 0, out, src, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, G>), dim3(blocks), dim3(256), 0, 0, out, src, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double cyc_per_iter = ms * 1e-3 * 2.4e9 / iters / wps;      // SIMD cycles per loop iteration
    if (MODE == 0) base[wps] = cyc_per_iter;
    printf("%-26s G=%2d waves/SIMD %d: %7.1f cyc/iter (MFMA %4d) extra vs mfma-only %+6.1f cyc\n", name, G, wps, cyc_per_iter,
           64 * G, cyc_per_iter - base[wps]);
}
int main()
{
    float *out, *src;
    hipMalloc(&out, 256 * 2048 * 4);
    hipMalloc(&src, (size_t)64 * 65536 * 4 + 65536);
    hipMemset(src, 0, (size_t)64 * 65536 * 4 + 65536);
    double b4[3] = {0, 0, 0}, b12[3] = {0, 0, 0};
    for (int wps = 1; wps <= 2; ++wps) {
        run<0, 4>("mfma only", wps, out, src, b4);
        run<1, 4>("+1 global_load_dwordx4", wps, out, src, b4);
        run<2, 4>("+1 ds_read_b128", wps, out, src, b4);
        run<3, 4>("+1 ds_write_b128", wps, out, src, b4);
        run<4, 4>("+4 v_cndmask", wps, out, src, b4);
        run<0, 12>("mfma only", wps, out, src, b12);
        run<1, 12>("+1 global_load_dwordx4", wps, out, src, b12);
    }
    return 0;
}
