// l2_stream.hip -- what a CU takes in when the workgroups of an XCD walk SHARED operand panels that are not L2-resident
// (the split dense: 32 workgroups of an XCD walk 4 activation panels and 8 weight panels in step; every byte is new to the L2 when
// the first of its sharers asks for it). Each workgroup (256 threads) walks one panel of 1 KiB fragments once, front to back,
// `inflight` fragments per wave ahead, by plain loads (mode 0) or by LDS-DMA into a ring with counted vmcnt (mode 1).
//   share = workgroups of one XCD that walk the same panel; panels are `frags` KiB long; footprint = panels x frags KiB.
//   hipcc --offload-arch=gfx950 -O2 tools/attic/l2_stream.hip -o build/l2_stream && build/l2_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst)
{
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(gsrc), "s"(lds_dst) : "memory");
}

// panel of workgroup b: XCD x = b % 8 owns panels [x * ppx, (x + 1) * ppx) (xcd_private) or all XCDs walk the same panels
template <int INFLIGHT, int MODE>
__global__ __launch_bounds__(256) void walk(const uint4* __restrict__ src, int frags, int share, int ppx, int xcd_private, int reps, unsigned* sink)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char ring[];      // MODE 1: [4 waves][INFLIGHT][1 KiB]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int p = (xcd_private ? x * ppx : 0) + (j / share) % ppx;
    const uint4* base = src + (size_t)p * frags * 64 + lane;
    unsigned acc = 0;
    const int per_wave = frags / 4;                             // wave w walks fragments w, w + 4, ...
    for (int r = 0; r < reps; ++r) {
    if (MODE == 0) {
        for (int f0 = 0; f0 < per_wave; f0 += INFLIGHT) {
            uint4 v[INFLIGHT];
#pragma unroll
            for (int u = 0; u < INFLIGHT; ++u) v[u] = base[(size_t)((f0 + u) * 4 + wave) * 64];
#pragma unroll
            for (int u = 0; u < INFLIGHT; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;      // (all four words: hipcc narrows the load to the words used)
        }
    } else {
        const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)ring + wave * INFLIGHT * 1024);
#pragma unroll
        for (int u = 0; u < INFLIGHT - 1; ++u) glds16(base + (size_t)(u * 4 + wave) * 64, lds0 + u * 1024);
        for (int f0 = 0; f0 < per_wave; f0 += INFLIGHT) {
#pragma unroll
            for (int u = 0; u < INFLIGHT; ++u) {
                const int f = f0 + u;
                asm volatile("s_waitcnt vmcnt(%0)" :: "n"(INFLIGHT - 2) : "memory");      // fragment f has landed
                acc += reinterpret_cast<const unsigned*>(ring + wave * INFLIGHT * 1024 + u * 1024)[lane * 4];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const int nf = f + INFLIGHT - 1;
                glds16(base + (size_t)((nf < per_wave ? nf : per_wave - 1) * 4 + wave) * 64, lds0 + ((u + INFLIGHT - 1) % INFLIGHT) * 1024);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int INFLIGHT, int MODE>
static void run(const uint4* src, unsigned* sink, int cus, int wgs_per_cu, int frags, int share, int ppx, int xcd_private, int reps, const char* what)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = cus * wgs_per_cu;
    float best = 1e9f;
    for (int it = 0; it < 3; ++it) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((walk<INFLIGHT, MODE>), dim3(grid), dim3(256), MODE ? 4 * INFLIGHT * 1024 : 0, 0, src, frags, share, ppx, xcd_private, reps, sink);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    const double b = (double)grid * reps * frags * 1024.0;
    const double foot = (double)(xcd_private ? 8 : 1) * ppx * frags / 1024.0;
    printf("%-10s %s in flight/wave %d, %d wg/CU, share %2d, footprint %7.1f MiB (%s): %6.1f GB/s per CU = %4.1f B/clk, %5.2f TB/s chip, unique %5.2f TB/s\n", what,
           MODE ? "LDS-DMA" : "plain  ", INFLIGHT, wgs_per_cu, share, foot, xcd_private ? "per-XCD panels" : "panels shared by all XCDs",
           b / best / 1e6 / cus, b / best / 1e6 / cus / 2.4, b / best / 1e9, b / best / 1e9 / share / (xcd_private ? 1 : 8));
}

int main()
{
    const size_t bytes = (size_t)1 << 30;
    uint4* src; unsigned* sink;
    CK(hipMalloc(&src, bytes)); CK(hipMemset(src, 1, bytes)); CK(hipMalloc(&sink, 64));
    int cus = 0; CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    // L2-resident: 32 panels of 64 KiB per XCD walked 100 times
    run<4, 0>(src, sink, cus, 1, 64, 1, 32, 1, 100, "L2-hot");
    run<4, 1>(src, sink, cus, 1, 64, 1, 32, 1, 100, "L2-hot");
    run<8, 1>(src, sink, cus, 1, 64, 1, 32, 1, 100, "L2-hot");
    run<8, 1>(src, sink, cus, 2, 64, 1, 32, 1, 100, "L2-hot");
    // streamed once: per-XCD panels of 4 MiB, shared by 1 / 4 / 8 / 32 workgroups of the XCD
    for (int share : {1, 4, 8, 32}) {
        const int ppx = 32 / share;
        run<4, 0>(src, sink, cus, 1, 4096, share, ppx, 1, 1, "stream");
        run<4, 1>(src, sink, cus, 1, 4096, share, ppx, 1, 1, "stream");
        run<8, 1>(src, sink, cus, 1, 4096, share, ppx, 1, 1, "stream");
        run<16, 1>(src, sink, cus, 1, 4096, share, ppx, 1, 1, "stream");
    }
    // two workgroups per CU
    for (int share : {4, 8}) {
        run<8, 1>(src, sink, cus, 2, 4096, share, 64 / share, 1, 1, "stream");
        run<8, 0>(src, sink, cus, 2, 4096, share, 64 / share, 1, 1, "stream");
    }
    // panels shared by all XCDs (the activations of the dense: MALL-resident after the first XCD's pass)
    for (int share : {8, 32}) {
        run<8, 1>(src, sink, cus, 1, 4096, share, 32 / share, 0, 1, "all-XCD");
        run<8, 0>(src, sink, cus, 1, 4096, share, 32 / share, 0, 1, "all-XCD");
    }
    return 0;
}
