// l2_channels.hip -- does the stride between the operand panels that concurrently running workgroups walk decide what the L2 delivers?
// Every workgroup walks "its" panel (blockIdx % panels) of 1 KiB fragments front to back, `reps` times; panels lie `stride` bytes apart.
// All data is L2-resident (panels x frags KiB <= 2 MB). Reports GB/s per CU and chip-wide for power-of-two and padded strides.
//   hipcc --offload-arch=gfx950 -O2 tools/attic/l2_channels.hip -o build/l2_channels && build/l2_channels
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

__global__ __launch_bounds__(256) void walk(const uint4* __restrict__ src, size_t stride16, int panels, int frags, int reps, int rot, unsigned* sink)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int p = (blockIdx.x >> 3) % panels;                 // the 32 workgroups of an XCD (blockIdx % 8) take different panels
    const uint4* base = src + (size_t)p * stride16 + lane;
    const int start = rot ? (p * rot) % frags : 0;          // rot: workgroups start at different fragments of their panel
    unsigned acc = 0;
    for (int r = 0; r < reps; ++r)
        for (int f0 = 0; f0 < frags; f0 += 16) {              // 4 waves x 4 loads in flight each = 16 fragments per round
            uint4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                int f = f0 + u * 4 + wave + start; if (f >= frags) f -= frags;
                v[u] = base[(size_t)f * 64];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
        }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main()
{
    const size_t bytes = 64u << 20;
    uint4* src; unsigned* sink;
    CK(hipMalloc(&src, bytes)); CK(hipMemset(src, 1, bytes)); CK(hipMalloc(&sink, 64));
    int cus = 0; CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int frags = 64, reps = 200;
    struct { int panels; size_t stride; int rot; int wgs_per_cu; } cfg[] = {
        {32, 64 << 10, 0, 1}, {32, 65 << 10, 0, 1}, {32, (64 << 10) + 256, 0, 1}, {32, 68 << 10, 0, 1}, {32, 64 << 10, 5, 1},
        {1, 64 << 10, 0, 1}, {1, 64 << 10, 5, 1},
        {32, 64 << 10, 0, 4}, {32, 65 << 10, 0, 4}, {32, (64 << 10) + 256, 0, 4}, {32, 64 << 10, 5, 4}, {1, 64 << 10, 0, 4},
        {16, 32 << 10, 0, 4}, {16, 33 << 10, 0, 4}, {8, 128 << 10, 0, 4}, {8, 129 << 10, 0, 4},
    };
    for (auto& c : cfg) {
        const int grid = cus * c.wgs_per_cu;
        for (int it = 0; it < 3; ++it) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(walk, dim3(grid), dim3(256), 0, 0, src, c.stride / 16, c.panels, frags, reps, c.rot, sink);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (it == 2) {
                const double b = (double)grid * reps * frags * 1024.0;
                printf("panels %2d stride %7zu B rot %d, %d workgroups/CU: %.1f GB/s per CU, %.2f TB/s chip\n", c.panels, c.stride, c.rot, c.wgs_per_cu,
                       b / ms / 1e6 / cus, b / ms / 1e9);
            }
        }
    }
    return 0;
}
