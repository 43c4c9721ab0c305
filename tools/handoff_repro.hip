// handoff_repro.hip -- a cross-workgroup hand-off of 1 KiB fragments INSIDE one launch, checked word by word.
// Round 4's persistent BiLSTM handed h from one workgroup to another with `sc1` stores / `sc1` LDS-DMA requests and
// polled counters, and produced a few wrong 32-site tiles per hundred forwards under co-running load; round 5 showed that plain
// LDS-DMA behind vmcnt + barrier is sound (tools/ldsdma_repro.hip), which leaves the hand-off. This tool runs the protocol of
// cdna_hip_programming.md Guideline 16 in isolation:
//   producer workgroup (4 waves): writes a 16 KiB image of sixteen 1 KiB fragments (what a cell's epilogue writes of h), every word a
//     function of (pair, epoch, word) -- `sc1` write-through 16-byte stores (mode bit 0 = 0) or plain stores behind an agent-scope
//     release fence (bit 0 = 1) --, every storing wave drains vmcnt, barrier, ONE lane stores flag = epoch (relaxed, agent scope);
//   consumer workgroup on ANOTHER CU (the next workgroup: another XCD; bit 3: eight further: the same XCD): optionally pre-reads the
//     image with plain loads (bit 2: its L1 then holds the previous epoch's lines -- Guideline 16 pitfall 3), ONE lane polls the flag
//     (relaxed, s_sleep), ONE agent-scope acquire, vmcnt(0), barrier, then every wave fetches the image by PLAIN LDS-DMA
//     (global_load_lds_dwordx4, bit 1 = 0: what a ring kernel would do) or by plain loads to registers (bit 1 = 1), vmcnt(0), barrier,
//     reads ANOTHER wave's fragments back from LDS and compares; then acks (flag the producer waits on before it reuses the buffer:
//     two buffers, epoch parity).
// Every spin is bounded (~2 s of s_memrealtime -> abort word -> everybody leaves, exit code 3): a hand-off between workgroups that
// are not co-resident cannot hang the box. MFMA kernels co-run on up to three more streams as in ldsdma_repro.
//   hipcc --offload-arch=gfx950 -O2 tools/handoff_repro.hip -o build/handoff_repro && build/handoff_repro [seconds per configuration]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
constexpr int IMG = 16384;            // bytes per hand-off: sixteen 1 KiB fragments
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(1))) unsigned gu32;
typedef unsigned int u4v __attribute__((ext_vector_type(4)));

__host__ __device__ inline unsigned pat(unsigned pair, unsigned epoch, unsigned word) { return (pair * 2654435761u) ^ (epoch * 40503u + 0x9e3779b9u) ^ (word * 2246822519u); }

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// one lane waits until *w >= want (relaxed agent-scope loads, s_sleep between them); false: timed out or somebody else did
__device__ __forceinline__ bool wait_ge(gu32* w, unsigned want, gu32* abort_word)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();      // 100 MHz
    for (;;) {
        if (__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) return true;
        if (__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
        if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) { __hip_atomic_store(abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return false; }
        __builtin_amdgcn_s_sleep(2);
    }
}

// flags: [pair][0] = ready epoch, [pair][16] = acked epoch (64-byte apart); stats: [0] checks, [1] wrong, [2] first wrong (pair << 40 | epoch << 16 | word)
__global__ __launch_bounds__(256) void handoff_kernel(unsigned char* img, gu32* flags, gu32* abort_word, int epochs, int mode, int npairs, unsigned long long* stats)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[IMG];
    __shared__ int ok_s;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int stride = (mode & 8) ? 8 : 1;       // consumer = producer + stride: +1 = the next XCD, +8 = the same XCD (workgroup b runs on XCD b % 8)
    const int b = blockIdx.x, grp = b / (2 * stride), in = b % (2 * stride);
    const int pair = grp * stride + in % stride;
    const bool producer = in < stride;
    if (pair >= npairs) return;
    gu32* const ready = flags + (size_t)pair * 32, *ack = ready + 16;
    unsigned long long checks = 0, wrong = 0, first = 0;
    for (int e = 1; e <= epochs; ++e) {
        unsigned char* const buf = img + ((size_t)pair * 2 + ((mode & 16) ? 0 : (e & 1))) * IMG;      // bit 4: ONE buffer -- a strict ping-pong (hand-off + ack per step)
        if (producer) {
            const int lag = (mode & 16) ? 1 : 2;
            if (e > lag) {      // the consumer has finished with this buffer's previous image
                if (tid == 0) ok_s = wait_ge(ack, (unsigned)(e - lag), abort_word);
                __syncthreads();
                if (!ok_s) break;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned q = (k * 256 + tid) * 4;      // first word of this lane's 16 bytes: wave w writes fragments 4 k + w
                const u4v v = {pat(pair, e, q), pat(pair, e, q + 1), pat(pair, e, q + 2), pat(pair, e, q + 3)};
                u4v* p = reinterpret_cast<u4v*>(buf) + k * 256 + tid;
                if (mode & 1) *(__attribute__((address_space(1))) u4v*)p = v;
                else asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");      // write-through to the memory side
            }
            if (mode & 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // EVERY storing wave drains (Guideline 16 pitfall 14), behind the fence (pitfall 12)
            __syncthreads();
            if (tid == 0) __hip_atomic_store(ready, (unsigned)e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (mode & 4) {      // stale lines into this CU's L1 on purpose
                const u4v s = *reinterpret_cast<const u4v*>(buf + tid * 16);
                asm volatile("" ::"v"(s));
            }
            if (tid == 0) ok_s = wait_ge(ready, (unsigned)e, abort_word);
            if (tid == 0) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }      // ONE acquire, behind the match
            __syncthreads();
            if (!ok_s) break;
            const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const u4v* g = reinterpret_cast<const u4v*>(buf) + k * 256 + tid;
                if (mode & 2) reinterpret_cast<u4v*>(lds)[k * 256 + tid] = *(const __attribute__((address_space(1))) u4v*)g;
                else glds16(g, __builtin_amdgcn_readfirstlane(lds0 + (k * 4 + wave) * 1024));
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
            for (int k = 0; k < 4; ++k) {      // another wave's fragments
                const int f = k * 4 + ((wave + 1 + (e & 1)) & 3);
                const u4v v = reinterpret_cast<const u4v*>(lds + f * 1024)[lane];
                const unsigned q = (f * 64 + lane) * 4;
                const bool ok = v.x == pat(pair, e, q) && v.y == pat(pair, e, q + 1) && v.z == pat(pair, e, q + 2) && v.w == pat(pair, e, q + 3);
                ++checks;
                if (!ok) { if (!wrong) first = ((unsigned long long)pair << 40) | ((unsigned long long)e << 16) | q; ++wrong; }
            }
            __syncthreads();      // every wave has read the LDS image (and the global one) before the ack lets the producer overwrite it
            if (tid == 0) __hip_atomic_store(ack, (unsigned)e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (checks) atomicAdd(&stats[0], checks);
    if (wrong) { atomicAdd(&stats[1], wrong); atomicCAS(&stats[2], 0ull, first); }
}

__global__ __launch_bounds__(256) void mfma_noise(float* out, int iters)
{
    floatx16 acc = {};
    bf16x8 a, bb;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.01f * (threadIdx.x + i)); bb[i] = (__bf16)(0.02f * i); }
    for (int it = 0; it < iters; ++it) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bb, acc, 0, 0, 0);
    if (acc[0] == 12345.f) out[threadIdx.x] = acc[3];
}

int main(int argc, char** argv)
{
    const double seconds = argc > 1 ? atof(argv[1]) : 2.0;
    int ncu = 0;
    CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0));
    const int npairs = ncu / 2;      // one workgroup per CU at most: every producer and consumer is resident
    unsigned char* img; gu32* flags; gu32* abort_word; unsigned long long* stats; float* sink;
    CK(hipMalloc(&img, (size_t)npairs * 2 * IMG));
    CK(hipMalloc((void**)&flags, (size_t)npairs * 128 + 64));
    abort_word = flags + (size_t)npairs * 32;
    CK(hipMalloc(&stats, 32)); CK(hipMalloc(&sink, 1024));
    hipStream_t s[4];
    for (auto& x : s) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    unsigned long long total = 0, total_wrong = 0;
    int rc = 0;
    printf("%d CUs, %d producer / consumer pairs, %.1f s per configuration; mode bits: 1 plain stores + release fence (0: sc1 stores), 2 plain loads (0: LDS-DMA), 4 consumer pre-reads the lines, 8 same XCD\n", ncu, npairs, seconds);
    for (int noise = 0; noise <= 3; noise += 3)
        for (int mode = 0; mode < 20; ++mode) {      // 16 .. 19: the ping-pong form of modes 0 .. 3
            const int epochs = 20000;
            unsigned long long checks = 0, wrong = 0; double us = 0, launches = 0;
            hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
            CK(hipEventRecord(t0, s[0]));
            double elapsed = 0;
            while (elapsed < seconds * 1e3) {
                CK(hipMemsetAsync((void*)flags, 0, (size_t)npairs * 128 + 64, s[0]));      // every polled word, every launch
                CK(hipMemsetAsync(stats, 0, 32, s[0]));
                for (int k = 1; k <= noise; ++k) hipLaunchKernelGGL(mfma_noise, dim3(ncu), dim3(256), 0, s[k], sink, 3000000);
                CK(hipEventRecord(e0, s[0]));
                hipLaunchKernelGGL(handoff_kernel, dim3(npairs * 2), dim3(256), 0, s[0], img, flags, abort_word, epochs, mode, npairs, stats);
                CK(hipEventRecord(e1, s[0]));
                CK(hipDeviceSynchronize());
                unsigned long long st[3]; unsigned ab = 0; float ms = 0;
                CK(hipMemcpy(st, stats, 24, hipMemcpyDeviceToHost));
                CK(hipMemcpy(&ab, (void*)abort_word, 4, hipMemcpyDeviceToHost));
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (ab) { printf("mode %2d noise %d: a spin timed out (abort word set): workgroups not co-resident or a flag never seen\n", mode, noise); rc = 3; break; }
                checks += st[0]; wrong += st[1]; us += ms * 1e3; launches += 1;
                if (st[1]) printf("   first wrong: pair %llu epoch %llu word %llu\n", st[2] >> 40, (st[2] >> 16) & 0xffffff, st[2] & 0xffff);
                CK(hipEventRecord(t1, s[0])); CK(hipEventSynchronize(t1));
                float el = 0; CK(hipEventElapsedTime(&el, t0, t1)); elapsed = el;
            }
            printf("mode %2d (%s stores, %s, %s, %s XCD) noise streams %d: %.3e 16-byte checks, %llu wrong, %.2f us per %s (%d in a chain per pair)\n", mode, (mode & 1) ? "plain + release" : "sc1",
                   (mode & 2) ? "plain loads" : "LDS-DMA", (mode & 4) ? "pre-read" : "cold", (mode & 8) ? "same" : "next", noise, (double)checks, wrong, launches ? us / launches / epochs : 0.0, (mode & 16) ? "hand-off + ack round trip" : "hand-off (two buffers in flight)", epochs);
            total += checks; total_wrong += wrong;
            if (rc) break;
        }
    printf("TOTAL %.3e 16-byte checks (%.3e KiB fragments), %llu wrong\n", (double)total, (double)total / 64, total_wrong);
    return rc ? rc : (total_wrong ? 1 : 0);
}
