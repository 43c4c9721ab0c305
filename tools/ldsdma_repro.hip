// ldsdma_repro.hip -- reduced reproducer for the fault round 4's persistent BiLSTM experiment showed (VERDICT r04
// item 3b): "operands wrong on chip" -- an LDS ring stage filled by LDS-DMA read back wrong although `s_waitcnt vmcnt(0)` and the
// workgroup barrier had passed, only with other kernels co-running. Two hypotheses:
//   H1  an LDS-DMA transfer (global_load_lds_dwordx4) is counted complete by vmcnt before its data has landed in LDS;
//   H2  state of a long-resident wave (LDS contents, in-flight transfers) is lost when the wave is preempted / context-switched.
// The kernel: long-resident workgroups (4 waves) loop over a three-stage LDS ring. Every iteration fills a stage with a pattern that
// is a function of (source row, byte offset) -- by LDS-DMA (mode 0) or by plain loads + ds_write (mode 1, the control) --, waits
// `vmcnt(0)`, passes the barrier, optionally sleeps (mode bit 2: a late-landing transfer would have arrived by then), reads the stage
// back and compares it with the pattern computed arithmetically. A second kernel (MFMA loop) can co-run on another stream, and
// further streams can be loaded so that the scheduler has to share or preempt CUs.
//   hipcc --offload-arch=gfx950 -O2 tools/ldsdma_repro.hip -o build/ldsdma_repro && build/ldsdma_repro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
constexpr int STAGE = 4096;           // bytes per ring stage: 256 threads x 16 B
constexpr int ROWS = 4096;            // source rows of STAGE bytes (16 MB: not L2-resident across workgroups)
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__host__ __device__ inline unsigned pat(unsigned row, unsigned word) { return (row * 2654435761u) ^ (word * 40503u + 0x9e3779b9u); }

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// mode bit 0: 0 = LDS-DMA, 1 = plain load + ds_write (control); bit 1: sleep between the barrier and the read-back;
// bit 2: two stages of lookahead behind a COUNTED vmcnt (the shipped cell kernels' form) instead of vmcnt(0) per stage
__global__ __launch_bounds__(256) void ring_kernel(const uint4* __restrict__ src, int iters, int mode, unsigned long long* bad, unsigned* first)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char ring[];      // [3][STAGE]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    unsigned row = (blockIdx.x * 977u + 13u) % ROWS;
    const unsigned ring_lds = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)ring;      // LDS byte address
    auto fill = [&](int stage, unsigned r) {
        const uint4* g = src + (size_t)r * (STAGE / 16) + tid;
        if (mode & 1) reinterpret_cast<uint4*>(ring + stage * STAGE)[tid] = *g;
        else glds16(g, __builtin_amdgcn_readfirstlane(ring_lds + stage * STAGE + wave * 1024));       // a wave's 64 x 16 B land contiguously
    };
    unsigned long long nbad = 0;
    if (mode & 4) { fill(0, row); fill(1, (row + 1) % ROWS); }
    for (int it = 0; it < iters; ++it) {
        const int st = it % 3;
        if (mode & 4) {
            fill((it + 2) % 3, (row + 2) % ROWS);
            asm volatile("s_waitcnt vmcnt(2)\n\ts_barrier" ::: "memory");             // (plain-load control: hipcc's own waits apply too)
        } else {
            fill(st, row);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        if (mode & 2) __builtin_amdgcn_s_sleep(64);
        // read back ANOTHER wave's quarter of the stage (what a consumer of a shared operand ring does)
        const int q = (wave + 1 + (it & 1)) & 3;
        const uint4 v = reinterpret_cast<const uint4*>(ring + st * STAGE + q * 1024)[lane];
        const unsigned w0 = (q * 64 + lane) * 4;
        const bool ok = v.x == pat(row, w0) && v.y == pat(row, w0 + 1) && v.z == pat(row, w0 + 2) && v.w == pat(row, w0 + 3);
        if (!ok) {
            ++nbad;
            if (atomicCAS(first, 0u, 1u) == 0u) { first[1] = blockIdx.x; first[2] = it; first[3] = tid; first[4] = v.x; first[5] = pat(row, w0); }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");              // everybody has read the stage before it is refilled
        row = (row + 1) % ROWS;
    }
    if (nbad) atomicAdd(bad, nbad);
}

// the co-runner: a register-resident bf16 MFMA loop (no memory traffic beyond one store), grid and length set by the host
__global__ __launch_bounds__(256) void mfma_kernel(float* out, int iters)
{
    floatx16 acc = {0};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (i + 1)); }
    for (int i = 0; i < iters; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[7];
}

int main(int argc, char** argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;      // ~20 ms per ring launch
    const int rounds = argc > 2 ? atoi(argv[2]) : 40;
    uint4* src; unsigned long long* bad; unsigned* first; float* sink;
    std::vector<unsigned> h((size_t)ROWS * STAGE / 4);
    for (unsigned r = 0; r < ROWS; ++r) for (unsigned w = 0; w < STAGE / 4; ++w) h[(size_t)r * (STAGE / 4) + w] = pat(r, w);
    CK(hipMalloc(&src, h.size() * 4)); CK(hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&bad, 8)); CK(hipMalloc(&first, 32)); CK(hipMalloc(&sink, 4096 * 256 * 4));
    hipStream_t s[4];
    for (auto& x : s) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
    int cus = 0; CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    // configurations: (co-running streams, ring workgroups per CU); 5 ring workgroups of 12 KB per CU = the cell kernels' occupancy
    const int cfgs[5][2] = {{0, 2}, {1, 2}, {3, 2}, {3, 5}, {3, 8}};
    int any = 0;
    for (int mode = 0; mode < 8; ++mode)
        for (auto& c : cfgs) {
            CK(hipMemset(bad, 0, 8)); CK(hipMemset(first, 0, 32));
            for (int r = 0; r < rounds; ++r) {
                hipLaunchKernelGGL(ring_kernel, dim3(cus * c[1]), dim3(256), 3 * STAGE, s[0], src, iters, mode, bad, first);
                for (int k = 0; k < c[0]; ++k)      // co-runners sized to want the whole chip for about as long as the ring launch
                    for (int j = 0; j < 4; ++j) hipLaunchKernelGGL(mfma_kernel, dim3(cus * 4), dim3(256), 0, s[1 + k], sink, iters * 4);
            }
            CK(hipDeviceSynchronize());
            unsigned long long nb; unsigned f[8];
            CK(hipMemcpy(&nb, bad, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(f, first, 32, hipMemcpyDeviceToHost));
            const double checks = (double)rounds * cus * c[1] * 256.0 * iters;
            printf("mode %d (%s%s%s) co-running streams %d, ring workgroups/CU %d: %.3g fragment checks, %llu wrong", mode,
                   mode & 1 ? "plain loads + ds_write" : "LDS-DMA", mode & 2 ? ", sleep before read-back" : "", mode & 4 ? ", counted vmcnt(2), two stages ahead" : ", vmcnt(0)",
                   c[0], c[1], checks, nb);
            if (nb) { printf("  [first: block %u iter %u thread %u got %08x want %08x]", f[1], f[2], f[3], f[4], f[5]); any = 1; }
            printf("\n"); fflush(stdout);
        }
    return any;
}
