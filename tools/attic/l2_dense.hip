// l2_dense.hip -- the operand traffic of dense_split_kernel<1,3,4,1> at 512 sites WITHOUT its arithmetic: 252 workgroups (one per CU),
// 377 k-steps, per k-step 12 KiB of activations (4 m-tiles x 3 terms, image [16][377][3 KiB]) + 9 KiB of weights (3 n-tiles x 3
// terms, image [189][377][3 KiB]) by LDS-DMA into a ring, one barrier per k-step. What does the delivery cost, and what changes it?
//   DEAL 0: fragment q of the 21 goes to wave q % 4 (the kernel's dealing)   1: wave w takes whole 3 KiB chunks (chunk c -> wave c % 4)
//   READS: also read the stage back with 12 ds_read_b128 per wave (as the MFMAs' operand reads do)
//   hipcc --offload-arch=gfx950 -O2 tools/attic/l2_dense.hip -o build/l2_dense && build/l2_dense
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
constexpr int KS = 377, MT = 16, NT = 189;

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst)
{
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(gsrc), "s"(lds_dst) : "memory");
}

template <int NSLOT, int DEAL, int READS, int XCDMAP>
__global__ __launch_bounds__(256) void dense_traffic(const char* __restrict__ A, const char* __restrict__ B, int ksteps, unsigned* sink)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char ring[];      // [NSLOT][24 KiB]
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int total = 252;
    int b = blockIdx.x;
    if (XCDMAP) { const int bx = b & 7, bq = b >> 3, per = total >> 3, rem8 = total & 7; b = bx * per + (bx < rem8 ? bx : rem8) + bq; }
    const int nb = b / 4, mb = b % 4;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)ring;
    // this wave's fragments of a k-step: up to 6 (source pointer at k-step 0, LDS offset within the stage)
    const char* src[6]; unsigned dst[6]; int nf = 0;
    for (int q = 0; q < 21; ++q) {
        const int chunk = q / 3, term = q % 3;                       // chunks 0..3: A m-tiles, 4..6: B n-tiles
        const int owner = DEAL ? chunk % 4 : q % 4;
        if (owner != wave) continue;
        const char* base = chunk < 4 ? A + (size_t)(mb * 4 + chunk) * KS * 3072 : B + (size_t)(nb * 3 + chunk - 4) * KS * 3072;
        src[nf] = base + term * 1024 + lane * 16; dst[nf] = q * 1024; ++nf;
    }
    auto request = [&](int k, int slot) {
#pragma unroll
        for (int j = 0; j < 6; ++j)
            if (j < nf) glds16(src[j] + (size_t)k * 3072, __builtin_amdgcn_readfirstlane(lds0 + slot * 24576 + dst[j]));
    };
    for (int s = 0; s < NSLOT - 1; ++s) request(s, s);
    unsigned acc = 0;
    int slot = 0;
    for (int k = 0; k < ksteps; ++k) {
        // stage k landed: everything but the later requested stages; (uniform count needs equal nf: use vmcnt(0)-style conservative bound per wave)
        if (k + NSLOT - 2 < ksteps) {
            if (nf == 6) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NSLOT - 2) * 6) : "memory");
            else if (nf == 5) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NSLOT - 2) * 5) : "memory");
            else if (nf == 3) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NSLOT - 2) * 3) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const int pslot = slot == 0 ? NSLOT - 1 : slot - 1;
        if (k + NSLOT - 1 < ksteps) request(k + NSLOT - 1, pslot);
        if (READS) {
#pragma unroll
            for (int r = 0; r < 12; ++r) {
                const uint4 v = *reinterpret_cast<const uint4*>(ring + slot * 24576 + ((wave * 3 + r) % 21) * 1024 + lane * 16);
                acc += v.x ^ v.w;
            }
        }
        slot = slot + 1 == NSLOT ? 0 : slot + 1;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int NSLOT, int DEAL, int READS, int XCDMAP>
static void run(const char* A, const char* B, unsigned* sink, const char* what)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto k = dense_traffic<NSLOT, DEAL, READS, XCDMAP>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    float best = 1e9f;
    for (int it = 0; it < 4; ++it) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(252), dim3(256), NSLOT * 24576, 0, A, B, KS, sink);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    const double cyc = best * 1e-3 * 2.4e9 / KS;
    printf("%-34s slots %d: %6.1f us, %5.0f cycles per k-step, %4.1f B/clk per CU\n", what, NSLOT, best * 1e3, cyc, 21 * 1024.0 / cyc);
}

int main()
{
    char *A, *B; unsigned* sink;
    CK(hipMalloc(&A, (size_t)MT * KS * 3072)); CK(hipMalloc(&B, (size_t)NT * KS * 3072)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(A, 1, (size_t)MT * KS * 3072)); CK(hipMemset(B, 1, (size_t)NT * KS * 3072));
    run<3, 0, 0, 1>(A, B, sink, "round-robin deal, no reads");
    run<3, 0, 1, 1>(A, B, sink, "round-robin deal, reads");
    run<5, 0, 1, 1>(A, B, sink, "round-robin deal, reads");
    run<3, 1, 0, 1>(A, B, sink, "chunk deal, no reads");
    run<3, 1, 1, 1>(A, B, sink, "chunk deal, reads");
    run<5, 1, 1, 1>(A, B, sink, "chunk deal, reads");
    run<3, 0, 1, 0>(A, B, sink, "round-robin, reads, no XCD map");
    run<5, 0, 0, 1>(A, B, sink, "round-robin deal, no reads");
    return 0;
}
