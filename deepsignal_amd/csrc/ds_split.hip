// ds_split.hip -- split-operand kernels (DS_PRECISION_BF16X3): fp32-class arithmetic on the bf16 matrix pipe of gfx950.
//
// On gfx950 the fp32 MFMA (v_mfma_f32_32x32x2_f32) runs at 1/16 of the bf16 MFMA rate and there is no xf32/tf32 form
// (MI355X_MICROARCH.md, matrix-core table). An fp32 operand x is therefore carried as three bf16 TERMS
//     t0 = bf16(x), t1 = bf16(x - t0), t2 = bf16(x - t0 - t1)        (round to nearest even; both differences are exact in fp32)
// which together hold all 24 significant bits of x, and a product a * b is the fp32-accumulated sum of the six
// bf16 x bf16 products (each exact in fp32) a0 b0, a0 b1, a1 b0, a0 b2, a1 b1, a2 b0 -- the three dropped ones lie below
// 2^-24 of the product. Six v_mfma_f32_32x32x16_bf16 (6 x 32 cycles) replace the eight v_mfma_f32_32x32x2_f32 (8 x 64 cycles)
// of a 32 x 32 x 16 block: 0.375 of the matrix-pipe time at fp32-class accuracy (CPU statement:
// oracle/torch_statement.py::forward_split; tools/split_emulation.py: 1.3 - 1.9e-5 from the float64 oracle on the
// trained-regime sets where native fp32 is 2.8 - 3.3e-5). Weights are split ONCE at load (ds_engine.cpp pack_b_split),
// activations ONCE where they are produced or staged (never inside a K loop); everything between the matrix products -- bias,
// ReLU, residual add, pooling -- is fp32, and the activations that travel through global memory are the fp32 engine's fp32 rows.
//
// Reference arithmetic: deepsignal/layers.py:87-139 (inception_layer), 205-232 (the eleven modules of incept_net).
#include "ds_device.h"

namespace ds {

// ---- term helpers ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float bf_lo(unsigned p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float bf_hi(unsigned p) { return __uint_as_float(p & 0xffff0000u); }
// the three bf16 terms of four fp32 values, as three 8-byte groups of four bf16 (channel order kept)
__device__ __forceinline__ void split3x4(float x0, float x1, float x2, float x3, uint2& t0, uint2& t1, uint2& t2)
{
    unsigned a = pack_bf2(x0, x1), b = pack_bf2(x2, x3);
    t0 = make_uint2(a, b);
    float r0 = x0 - bf_lo(a), r1 = x1 - bf_hi(a), r2 = x2 - bf_lo(b), r3 = x3 - bf_hi(b);
    a = pack_bf2(r0, r1); b = pack_bf2(r2, r3);
    t1 = make_uint2(a, b);
    r0 -= bf_lo(a); r1 -= bf_hi(a); r2 -= bf_lo(b); r3 -= bf_hi(b);
    t2 = make_uint2(pack_bf2(r0, r1), pack_bf2(r2, r3));
}
// the first / last three of the six term products of one 32 x 32 x 16 block (w = the weight fragment's terms, a = the
// activation fragment's terms), small products first. Issued TRANSPOSED like every MFMA of the fused modules: mfma(weights,
// activations) gives (X W)^T, so a lane holds 4 x 4 consecutive output channels of ONE activation row.
__device__ __forceinline__ floatx16 mfma3_lo(const float4 (&w)[3], float4 a0, float4 a1, float4 a2, floatx16 c)
{
    c = mfma_bf(w[2], a0, c);
    c = mfma_bf(w[0], a2, c);
    c = mfma_bf(w[1], a1, c);
    return c;
}
__device__ __forceinline__ floatx16 mfma3_hi(const float4 (&w)[3], float4 a0, float4 a1, floatx16 c)
{
    c = mfma_bf(w[1], a0, c);
    c = mfma_bf(w[0], a1, c);
    c = mfma_bf(w[0], a0, c);
    return c;
}

// ---------------------------------------------------------------------------------------------------------------------
// Fused inception modules, split operands. Tile, chaining and phases are those of inception_fused_kernel (ds_kernels.hip): one
// workgroup (8 waves) owns a tile of whole sites (<= 96 rows) and takes it through the modules of one width class; only a chain's
// first module may take the stride-2 max pool of its input while staging;
//   P1  [rows x cin] x [cin x 256]: wave w owns n-tile w for all m-tiles, K in chunks of 16 channels = ONE bf16 k-step.
//       The staging waves (0 .. 2 TM - 1) load the chunk's fp32 rows ONCE, park a raw copy in LDS (T2's region, idle during P1), and
//       one step later read the row and its two neighbours back from it, take the 3-tap max (branch 1's maxpool(3, 1, SAME), padded
//       taps ignored = the own row), split both the plain and the pooled values into terms and write ONE staged row
//       [plain t0 | t1 | t2 | pooled t0 | t1 | t2] x 32 B: every wave's fragment reads are three ds_read_b128 at immediate
//       offsets from one address (waves 6, 7: the pooled half), no wave pools or splits inside its MFMA stream. Per chunk and wave:
//       3 weight fragments (one per term, global -> VGPR), 3 x TM activation fragments, 6 x TM MFMAs; one barrier per chunk.
//   P2  the 1x3 / 1x5 convs from the 32-channel intermediates, which the P1 epilogue wrote to LDS as terms (T1, zero halo rows =
//       SAME padding), dealt by (conv, n-tile): a wave takes its weight fragments through all the m-tiles it owns. Branch 5's
//       64-channel intermediate goes to LDS as terms as well (T2); its last 1x1 accumulates on top of the stem accumulators.
// LDS (bytes): staged chunks 2 x rows x 208 (later the fp32 b1|b2 output tile) | T2 rows x 400 (raw chunks during P1) | T1
// (spt (W + 4) + 5) x 592: 143 - 147 KB for 96-row tiles, one workgroup per CU. Every row stride is an odd number of 16-byte slots.
// What bounds it (round 6, profiles/r06_chain_p1_experiments.md): a P1 step takes ~1,660 cycles for 1,152 of MFMA; barrier, transform
// arithmetic and staged writes each cost nothing when removed alone, the 30 KiB of weight fragments and rows a step asks the L2 for do
// (18 B/clk per CU against 26 at full matrix rate), and eight re-orderings of the step's work all took the same time. Per module 47 - 53 k
// cycles: P1 25 k in its loop + 4.5 - 8 k of start-up, P1 epilogue + barrier 6 - 7.7 k, P2 11.6 k (7.7 k of MFMA).
constexpr int S_LDP = 208;      // staged chunk row: 6 x 32 B + 16
constexpr int S_LD1 = 592;      // T1 row: 3 terms x 96 channels x 2 B + 16
constexpr int S_LD2 = 400;      // T2 row: 3 terms x 64 channels x 2 B + 16
constexpr int S_LDY = 400;      // b1|b2 output tile row: 96 floats + 4
constexpr int S_LDR = 80;       // raw fp32 chunk row: 16 floats + 16 B
constexpr int S_T1P = 192;      // bytes between the terms of a T1 row
constexpr int S_T2P = 128;      // ... of a T2 row

size_t inception_fused_split_lds_bytes(int tm, int W, int spt)
{
    const int tr32 = tm * 32;
    return (size_t)2 * tr32 * S_LDP + (size_t)tr32 * S_LD2 + (size_t)(spt * (W + 4) + 5) * S_LD1 + (size_t)tr32 * 4 + 192 * 4;
}

// weights of one second-stage conv unit: ntaps x 2 k-steps x 3 terms (<= 30 fragments); the packed panel of an n-tile holds
// ks k-steps of three 1 KiB fragments each (pack_b_split)
__device__ __forceinline__ void fuseds_unit_prefetch(const float* __restrict__ Bp, int ntaps, int nt, int lane, float4 (&ub)[30])
{
    const int ks = (ntaps * 32 + 63) / 64 * 4;
    const float* bsrc = Bp + ((size_t)(nt * ks) * 3 * 64 + lane) * 4;
#pragma unroll
    for (int g = 0; g < 30; ++g)
        if (g < ntaps * 6) ub[g] = gload4(bsrc + g * 256);
}

// one second-stage conv on NM m-tiles at once: one set of weights, NM accumulators, TWO fragment buffers: the three reads of the next
// (tap, k-step, m-tile) group are issued in front of the six MFMAs of the current one and land while those run (192 cycles of MFMA
// per group against ~100 of LDS latency); left to itself hipcc issues a group's reads directly in front of its MFMAs and every group
// waits for the LDS (6.8 k cycles for 108 MFMAs with one wave per SIMD active).
template <int NTAPS, int NM>
__device__ __forceinline__ void fuseds_conv_units(const char* T1, const int (&rm)[NM], int cbyte, int lane, const float4 (&ub)[30], floatx16 (&acc)[NM])
{
    const char* base[NM];
#pragma unroll
    for (int m = 0; m < NM; ++m) base[m] = T1 + (rm[m] - NTAPS / 2) * S_LD1 + cbyte + (lane >> 5) * 16;
    float4 fr[2][3];
#pragma unroll
    for (int p = 0; p < 3; ++p) fr[0][p] = *reinterpret_cast<const float4*>(base[0] + p * S_T1P);
    int cur = 0;
#pragma unroll
    for (int t = 0; t < NTAPS; ++t)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int q = (t * 2 + j) * 3;
            const float4 w[3] = {ub[q], ub[q + 1], ub[q + 2]};
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                // next group: (t, j, m + 1), or m-tile 0 of the next k-step / tap
                const int mn = m + 1 < NM ? m + 1 : 0;
                const int jn = m + 1 < NM ? j : (j + 1) & 1;
                const int tn = (m + 1 < NM || j == 0) ? t : t + 1;
                if (tn < NTAPS) {
                    const char* arow = base[mn] + tn * S_LD1 + jn * 32;
#pragma unroll
                    for (int p = 0; p < 3; ++p) fr[cur ^ 1][p] = *reinterpret_cast<const float4*>(arow + p * S_T1P);
                }
                acc[m] = mfma3_lo(w, fr[cur][0], fr[cur][1], fr[cur][2], acc[m]);
                acc[m] = mfma3_hi(w, fr[cur][0], fr[cur][1], acc[m]);
                cur ^= 1;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
}

#ifndef DS_SPLIT_VD
#define DS_SPLIT_VD 4       // register stages of the stagers' row loads (chunk c + VD is requested at step c)
#endif
#ifndef DS_SPLIT_BD
#define DS_SPLIT_BD 2       // register stages of a wave's P1 weight fragments (chunk c + BD - 1 is requested at step c)
#endif
#ifndef DS_SPLIT_SW0
#define DS_SPLIT_SW0 0          // the two waves whose phase stamps a debug build records (tools/stamps.py prints them as "wave 0" / "wave 7")
#define DS_SPLIT_SW1 7
#endif
#ifndef DS_SPLIT_BISECT
#define DS_SPLIT_BISECT 0       // timing-only bisect builds of P1 (results WRONG): 1 no transform / LDS writes, 2 no row loads, 4 no barrier in the
#endif                          // chunk loop, 8 no weight loads, 16 no fragment reads, 32 no transform arithmetic, 64 no staged writes, 128 no raw reads
#ifndef DS_SPLIT_WNT
#define DS_SPLIT_WNT 0          // 1: P1 weight fragments by non-temporal loads
#endif
#if DS_SPLIT_WNT
#define DS_SPLIT_WLOAD(p) gload4_nt(p)
#else
#define DS_SPLIT_WLOAD(p) gload4(p)
#endif
#ifndef DS_SPLIT_READS_FIRST
#define DS_SPLIT_READS_FIRST 1
#endif
#ifndef DS_P1_CLOCK
#define DS_P1_CLOCK 0           // 1: s_memtime inside every P1 step (lo half | LDS drain | barrier | hi half), printed by two workgroups for the chain's
#endif                          // second module (diagnostic build: tools/attic/r06_p1clk.sh)
template <int R> struct SplitRole { static constexpr int value = R; };

template <int TM>
__global__ __launch_bounds__(512, 2) void inception_fused_split_kernel(const FusedChain c)
{
    constexpr int TR32 = TM * 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* const sm = reinterpret_cast<char*>(smem);
    char* const Pst = sm;                                // [2][TR32][S_LDP] staged chunks
    char* const Ys = sm;                                 // [TR32][S_LDY] b1|b2 output tile, aliases the staged chunks once P1 is done
    char* const T2 = sm + 2 * TR32 * S_LDP;              // [TR32][S_LD2]
    char* const Raw = T2;                                // [2][TR32][S_LDR] raw fp32 chunks during P1 (T2 is written in P2a)
    char* const T1 = T2 + TR32 * S_LD2;                  // [spt*(W+4)+5][S_LD1] (5 spare rows: a dump row for padding rows + its halo)
    const int W = c.m[0].W, spt = c.m[0].spt;            // the same for every module of a chain (ds_internal.h FusedChain)
    int* const rowmap = reinterpret_cast<int*>(T1 + (spt * (W + 4) + 5) * S_LD1);      // [TR32] tile row -> T1 row
    float* const Bs = reinterpret_cast<float*>(rowmap + TR32);                         // [3][64] biases of b5b | b3b | b4b
    const int site0 = blockIdx.x * spt;
    const int nhere = min(spt, c.m[0].n_sites - site0);
    const int TRv = nhere * W;                           // valid rows of this tile
    const size_t grow0 = (size_t)site0 * W;
    auto lds_barrier = []() __attribute__((always_inline)) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    // once per workgroup: T1 zeroed (every module rewrites the interior rows completely and never touches the halos) and the
    // tile's row map
    for (int i = threadIdx.x; i < (spt * (W + 4) + 5) * (S_LD1 / 16); i += 512)
        reinterpret_cast<float4*>(T1)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (threadIdx.x < TR32)   // rows past the tile's last site map to the dump row, so LDS writes need no predicate
        rowmap[threadIdx.x] = (int)threadIdx.x < TRv ? ((int)threadIdx.x / W) * (W + 4) + 2 + (int)threadIdx.x % W : spt * (W + 4) + 2;

    for (int mi = 0; mi < c.nmod; ++mi) {
    const FusedArgs& a = c.m[mi];
    // every per-lane quantity derives from this opaque copy of the thread index (hipcc otherwise hoists loop-invariant per-lane
    // addresses out of the module loop and holds them in registers for the whole kernel)
    int tid_opaque = threadIdx.x, wave_opaque = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    asm volatile("" : "+v"(tid_opaque), "+s"(wave_opaque));
    const int tid = tid_opaque, lane = tid & 63, wave = wave_opaque;
    const int cin = a.cin;
    const gptr1w Yg = (gptr1w)(a.Y + grow0 * 240);     // wave-uniform base; per-lane offsets stay 32-bit
    const bool stamp = a.dbg != nullptr && lane == 0 && (wave == DS_SPLIT_SW0 || wave == DS_SPLIT_SW1) && blockIdx.x < DBG_MAX_WGS;
#define DS_STAMP(i) do { if (stamp) (a.dbg + ((size_t)blockIdx.x * 2 + (wave == DS_SPLIT_SW1)) * 8)[i] = __builtin_amdgcn_s_memtime(); } while (0)
    DS_STAMP(0);
    if (tid >= 256 && tid < 448) {
        const int q = tid - 256;
        Bs[q] = gload((q < 64 ? a.bias5b : q < 128 ? a.bias3b : a.bias4b) + (q & 63));
    }

    // ---- P1 staging cursor: thread -> (row, 16-byte slot of the chunk's 64 B); rows past the tile end re-read row TRv-1
    // waves 0 .. 2 TM - 1 stage TR32 x 4 sixteen-byte slots, four lanes a row (a row's 64 B of a chunk are one contiguous request).
    // (Sixteen consecutive lanes on sixteen rows at one slot makes the 8-byte term writes and the raw copy's accesses tile the LDS
    // banks exactly -- SQ_LDS_BANK_CONFLICT is 36 % of the kernel's LDS-active cycles with this mapping -- but scatters every
    // row load over sixteen lines per sixteen lanes: measured 373 against 345 us per step. The conflicts are the cheaper evil.)
    const int sr = tid >> 2, sq = tid & 3;
    const int rr = sr < TRv ? sr : TRv - 1;
    const int wr = rr % W;
    // plain input: one row per staged row. Pooled input (the module right after maxpool_layer2 / 3, layers.py:211-213,224-226): the
    // staged row (site s, w) is the max of the input rows 2 w - pad + {0, 1, 2} of site s that exist (padded taps ignored), taken
    // while loading -- the stride-2 pool costs no launch and no buffer, as in the fp32 kernel
    const bool pooled_in = a.pool_win > 0;        // wave-uniform per module
    const float* pc;
    int oq = 0, orr = 0;                          // float offsets of the second / third tap from the first (0 = the same row again)
    if (!pooled_in) {
        pc = a.X + (grow0 + rr) * cin + sq * 4;
    } else {
        const int s_ = rr / W;
        const int i0 = 2 * wr - a.pool_pad;
        const int ia = i0 < 0 ? i0 + 1 : i0;                               // first existing tap
        const int ib = i0 + 1 < a.pool_win ? (i0 + 1 < 0 ? ia : i0 + 1) : ia;
        const int ic = i0 + 2 < a.pool_win ? i0 + 2 : ib;
        pc = a.X + ((size_t)(site0 + s_) * a.pool_win + ia) * cin + sq * 4;
        oq = (ib - ia) * cin; orr = (ic - ia) * cin;
    }
    const int om = wr > 0 ? -S_LDR : 0;           // previous / next row of the same site in the raw LDS copy, or the own row at a site edge
    const int op = wr < W - 1 ? S_LDR : 0;
    // this wave's P1 weights: n-tile `wave`, 16 k-steps x 3 terms of 1 KiB (K padded to 256 in the pack)
    const float* bp = a.Bp1 + ((size_t)wave * 16 * 3 * 64 + lane) * 4;

    const int h4 = 4 * (lane >> 5), rlane = lane & 31;
    floatx16 acc[TM];
    {
        float4 bv[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            bv[g] = gload4(a.bias1 + wave * 32 + 8 * g + h4);                  // bias1 is zero-padded to 256
            if (wave < 2) {                                                     // b5 stem columns also carry the tail's BN shift
                const float4 t = gload4(a.bias5c + wave * 32 + 8 * g + h4);     // (zero-padded to 64)
                bv[g].x += t.x; bv[g].y += t.y; bv[g].z += t.z; bv[g].w += t.w;
            }
        }
#pragma unroll
        for (int mt = 0; mt < TM; ++mt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                acc[mt][4 * g + 0] = bv[g].x; acc[mt][4 * g + 1] = bv[g].y;
                acc[mt][4 * g + 2] = bv[g].z; acc[mt][4 * g + 3] = bv[g].w;
            }
    }

    const int nchunks = cin / KC;      // 15 or 16
    __builtin_assume(nchunks >= 15 && nchunks <= 16);
    // The staging waves (the first 2 TM: one 16-byte slot per thread) and the others run two instantiations of the loop, so that
    // inside either no vector-memory instruction sits under a per-thread condition: hipcc counts s_waitcnt vmcnt conservatively
    // across such a branch (as if the loads had not been issued), which made every step wait for the row loads it had just
    // requested -- 46 k cycles of P1 instead of ~20 k.
    auto run_p1 = [&](auto role_tag) __attribute__((always_inline)) {
        // ROLE 1: a staging wave (waves 0 .. 2 TM - 1: TR32 x 4 sixteen-byte slots, one per thread). ROLE 0: the others.
        // A CU's vector-memory path delivers ~32 B/clk and a step has 1,152 cycles of MFMA: the 24 KB of weight
        // fragments per step already take two thirds of that, so a chunk's rows are loaded from global memory ONCE (6 KB) -- the
        // two neighbour rows branch 1's pool needs come out of a raw fp32 copy of the chunk in LDS (Raw, which lives in T2's
        // region: T2 is not in use during P1). Chunk k: requested at step k - VD, raw copy written at step k - 2, transformed
        // (3-tap max, terms) at step k - 1 into the staged buffer, consumed by the MFMAs of step k; every hand-over crosses one of
        // the per-step barriers.
        constexpr int ROLE = decltype(role_tag)::value;
        constexpr bool STG = ROLE != 0;
        constexpr int VD = DS_SPLIT_VD;
        constexpr int BD = DS_SPLIT_BD;
        float4 vo[VD];                      // the stager's 16 B of its row, chunks in flight
        float4 bq[BD][3];
        float4 af[2][3][TM];
#if DS_SPLIT_BISECT
        {
            const float4 z = make_float4(__uint_as_float(tid), 1.f, 2.f, 3.f);
            for (int i = 0; i < VD; ++i) vo[i] = z;
            for (int i = 0; i < BD; ++i) bq[i][0] = bq[i][1] = bq[i][2] = z;
            for (int i = 0; i < 2; ++i) for (int p = 0; p < 3; ++p) for (int mt = 0; mt < TM; ++mt) af[i][p][mt] = z;
        }
#endif
        // waves 6, 7 (branch 1) read the pooled half of a staged row
        const char* const fsrc = Pst + rlane * S_LDP + (lane >> 5) * 16 + (wave >= 6 ? 96 : 0);
        char* const sdst = Pst + sr * S_LDP + sq * 8;
        char* const rdst = Raw + sr * S_LDR + sq * 16;
        auto load_a = [&](int V) __attribute__((always_inline)) {
            if (STG && !(DS_SPLIT_BISECT & 2)) {
                vo[V] = gload4(pc);
                if (pooled_in) vo[V] = f4max(f4max(vo[V], gload4(pc + oq)), gload4(pc + orr));      // wave-uniform branch
                pc += KC;
            }
        };
        auto raw_a = [&](int X, int V) __attribute__((always_inline)) {
            if (STG && !(DS_SPLIT_BISECT & 1)) *reinterpret_cast<float4*>(rdst + X * TR32 * S_LDR) = vo[V];
        };
        float4 rv[3];                                   // (DS_P1_CLOCK 2 only: the raw rows read apart from the transform)
        auto raw_read = [&](int slot) __attribute__((always_inline)) {
            if (STG && !(DS_SPLIT_BISECT & 1)) {
                const char* r = rdst + slot * TR32 * S_LDR;
                rv[0] = *reinterpret_cast<const float4*>(r);
                rv[1] = *reinterpret_cast<const float4*>(r + om);
                rv[2] = *reinterpret_cast<const float4*>(r + op);
            }
        };
        auto store_a = [&](int X) __attribute__((always_inline)) {
            if (STG && !(DS_SPLIT_BISECT & 1)) {
                const char* r = rdst + X * TR32 * S_LDR;
                float4 o, m, n;
                if (DS_SPLIT_BISECT & 128) { o = vo[0]; m = vo[1 % VD]; n = vo[2 % VD]; }      // (timing only: no raw reads)
                else if (DS_P1_CLOCK == 2) { o = rv[0]; m = rv[1]; n = rv[2]; }
                else {
                    o = *reinterpret_cast<const float4*>(r);
                    m = *reinterpret_cast<const float4*>(r + om);
                    n = *reinterpret_cast<const float4*>(r + op);
                }
                char* d = sdst + X * TR32 * S_LDP;
                uint2 t0, t1, t2;
                constexpr bool NOARITH = (DS_SPLIT_BISECT & 32) != 0, NOWRITE = (DS_SPLIT_BISECT & 64) != 0;      // (timing-only builds)
                if (NOARITH) { t0 = make_uint2(__float_as_uint(o.x), __float_as_uint(o.y)); t1 = make_uint2(__float_as_uint(o.z), __float_as_uint(o.w)); t2 = make_uint2(__float_as_uint(m.x), __float_as_uint(m.y)); }
                else split3x4(o.x, o.y, o.z, o.w, t0, t1, t2);
                if (NOWRITE) asm volatile("" ::"v"(t0.x), "v"(t0.y), "v"(t1.x), "v"(t1.y), "v"(t2.x), "v"(t2.y));
                else { *reinterpret_cast<uint2*>(d) = t0; *reinterpret_cast<uint2*>(d + 32) = t1; *reinterpret_cast<uint2*>(d + 64) = t2; }
                if (NOARITH) { t0 = make_uint2(__float_as_uint(m.z), __float_as_uint(m.w)); t1 = make_uint2(__float_as_uint(n.x), __float_as_uint(n.y)); t2 = make_uint2(__float_as_uint(n.z), __float_as_uint(n.w)); }
                else split3x4(fmaxf(fmaxf(o.x, m.x), n.x), fmaxf(fmaxf(o.y, m.y), n.y), fmaxf(fmaxf(o.z, m.z), n.z), fmaxf(fmaxf(o.w, m.w), n.w), t0, t1, t2);
                if (NOWRITE) asm volatile("" ::"v"(t0.x), "v"(t0.y), "v"(t1.x), "v"(t1.y), "v"(t2.x), "v"(t2.y));
                else { *reinterpret_cast<uint2*>(d + 96) = t0; *reinterpret_cast<uint2*>(d + 128) = t1; *reinterpret_cast<uint2*>(d + 160) = t2; }
            }
        };
        auto load_b = [&](int Bi) __attribute__((always_inline)) {
            if (DS_SPLIT_BISECT & 8) return;
            bq[Bi][0] = DS_SPLIT_WLOAD(bp); bq[Bi][1] = DS_SPLIT_WLOAD(bp + 256); bq[Bi][2] = DS_SPLIT_WLOAD(bp + 512);
            bp += 768;
        };
        auto read_frags = [&](int X) __attribute__((always_inline)) {
            if (DS_SPLIT_BISECT & 16) return;
#pragma unroll
            for (int mt = 0; mt < TM; ++mt) {
                const char* q = fsrc + X * TR32 * S_LDP + mt * 32 * S_LDP;
#pragma unroll
                for (int p = 0; p < 3; ++p) af[X][p][mt] = *reinterpret_cast<const float4*>(q + 32 * p);
            }
        };
#pragma unroll
        for (int i = 0; i < VD; ++i) load_a(i);                  // chunks 0 .. VD - 1
#pragma unroll
        for (int i = 0; i < BD - 1; ++i) load_b(i);              // weight fragments of chunks 0 .. BD - 2
        raw_a(0, 0);
        raw_a(1, 1 % VD);
        lds_barrier();
        if (DS_P1_CLOCK == 2) raw_read(0);
        store_a(0);
        lds_barrier();
        read_frags(0);
#if DS_P1_CLOCK
        unsigned long long pk[4] = {0, 0, 0, 0}, pl[4] = {0, 0, 0, 0}, plast = 0, pm = 0;
        unsigned long long ptop = __builtin_amdgcn_s_memtime();
        const unsigned long long pbeg = ptop;
#endif
#pragma unroll
        for (int cc = 0; cc < 16; ++cc) {
            if (cc < nchunks) {                                   // wave-uniform; only cc = 15 is really conditional
                const int X = cc & 1;
                const bool has1 = cc + 1 < nchunks, has2 = cc + 2 < nchunks;
                if (has2) raw_a(X, (cc + 2) % VD);                // the raw copy chunk cc lived in was last read at step cc - 1
                if (cc + VD < nchunks) load_a(cc % VD);           // chunk cc + VD into the register stage chunk cc left two steps ago
                if (cc + BD - 1 < nchunks) load_b((cc + BD - 1) % BD);
#if DS_P1_CLOCK == 2
                // (diagnostic: the lo half taken apart -- loads / raw write | raw reads until they have landed | arithmetic + staged writes issued | MFMAs)
                __builtin_amdgcn_sched_barrier(0);
                const unsigned long long l0 = __builtin_amdgcn_s_memtime();
                if (has1 && STG) { raw_read((cc + 1) & 1); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
                __builtin_amdgcn_sched_barrier(0);
                const unsigned long long l1 = __builtin_amdgcn_s_memtime();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                if (has1) store_a(X ^ 1);
                __builtin_amdgcn_sched_barrier(0);
                const unsigned long long l2 = __builtin_amdgcn_s_memtime();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const unsigned long long l3 = __builtin_amdgcn_s_memtime();
                __builtin_amdgcn_sched_barrier(0);
                pl[0] += l0 - ptop; pl[1] += l1 - l0; pl[2] += l2 - l1; pl[3] += l3 - l2; plast = l3;
#else
                if (has1) store_a(X ^ 1);
#endif
#pragma unroll
                for (int mt = 0; mt < TM; ++mt) acc[mt] = mfma3_lo(bq[cc % BD], af[X][0][mt], af[X][1][mt], af[X][2][mt], acc[mt]);
                __builtin_amdgcn_sched_barrier(0);
#if DS_P1_CLOCK
                const unsigned long long q1 = __builtin_amdgcn_s_memtime();
                pm += q1 - plast;
                asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory");      // (everything but the stamp just requested)
                const unsigned long long q2 = __builtin_amdgcn_s_memtime();
                asm volatile("s_barrier" ::: "memory");
                const unsigned long long q3 = __builtin_amdgcn_s_memtime();
#else
                if (!(DS_SPLIT_BISECT & 4)) lds_barrier();
#endif
                if (has1) read_frags(X ^ 1);
#pragma unroll
                for (int mt = 0; mt < TM; ++mt) acc[mt] = mfma3_hi(bq[cc % BD], af[X][0][mt], af[X][1][mt], acc[mt]);
#if DS_SPLIT_READS_FIRST
                if (has1) __builtin_amdgcn_sched_group_barrier(0x100, 3 * TM, 0);      // the next chunk's fragment reads lead the half-step
#endif
                __builtin_amdgcn_sched_barrier(0);
#if DS_P1_CLOCK
                const unsigned long long q4 = __builtin_amdgcn_s_memtime();
                pk[0] += q1 - ptop; pk[1] += q2 - q1; pk[2] += q3 - q2; pk[3] += q4 - q3; ptop = q4;
#endif
            }
        }
#if DS_P1_CLOCK
        if (mi == 1 && (blockIdx.x == 0 || blockIdx.x == 37) && lane == 0)
            printf("P1CLK block %d wave %d role %d W %d: cycles per step lo %d drain %d barrier %d hi %d | P1 total %d | lo: top %d rawread %d transform %d writes-land %d mfma %d\n", (int)blockIdx.x, wave, ROLE, W,
                   (int)(pk[0] / nchunks), (int)(pk[1] / nchunks), (int)(pk[2] / nchunks), (int)(pk[3] / nchunks), (int)(ptop - pbeg),
                   (int)(pl[0] / nchunks), (int)(pl[1] / nchunks), (int)(pl[2] / nchunks), (int)(pl[3] / nchunks), (int)(pm / nchunks));
#endif
    };
    if (wave < 2 * TM) run_p1(SplitRole<1>{}); else run_p1(SplitRole<0>{});      // wave-uniform
    DS_STAMP(1);
    lds_barrier();   // all fragment reads of the staging area are done before the output tile aliases it
    DS_STAMP(2);

    // ---- P2 jobs (wave-uniform). A CU's vector-memory path is the scarce resource of this kernel (~32 B/clk against 24 KB of P1
    // weight fragments per 1,152-cycle step), so the second-stage convs are dealt by (conv, n-tile) and a wave takes ITS weight
    // fragments (18 or 30 KiB) through all the m-tiles it owns -- the unit table of inception_fused_kernel, one (conv, m-tile,
    // n-tile) per wave, re-loaded them per m-tile: 420 KB per module and tile against 216 KB here:
    //   waves 0, 1   P2a: b5b (1x3, 32 -> 64), n-tile = wave, all m-tiles -> T2;      P2b: the residual tail on the stem accumulators
    //   waves 2, 3   P2a: b3b (1x3, 32 -> 48), n-tile = wave - 2, all m-tiles
    //   waves 4, 5   P2a: the b1|b2 tile's stores (with waves 6, 7);                   P2b: b4b (1x5, 32 -> 48), n-tile = wave - 4, m-tile 0
    //   waves 6, 7   P2a: the b1|b2 tile's stores;                                     P2b: b4b, n-tile = wave - 6, m-tiles 1 .. TM - 1
    // MFMAs per SIMD and phase at TM = 3: P2a 108 | 108 | 108 | 108, P2b 72 + 60 | 72 + 60 | 120 | 120.
    float4 pf[30];                                   // this wave's unit weights
    // (defined on every path: a register array that is only loaded under a wave-uniform condition is "undefined on some
    // paths", and hipcc keeps such a value live around the whole module loop)
#pragma unroll
    for (int g = 0; g < 30; ++g) pf[g] = make_float4(0.f, 0.f, 0.f, 0.f);
    // Vector-memory operations retire in issue order (one vmcnt counter for loads and stores) and hipcc counts conservatively
    // across control flow, so a weight fragment requested BEHIND a store is only usable once that store is acknowledged: every
    // wave requests its weights here, in front of all its stores.
    if (wave < 2) fuseds_unit_prefetch(a.Bp5b, 3, wave, lane, pf);
    else if (wave < 4) fuseds_unit_prefetch(a.Bp3b, 3, wave - 2, lane, pf);
    else fuseds_unit_prefetch(a.Bp4b, 5, (wave - 4) & 1, lane, pf);

    // ---- P1 epilogue (bias already inside acc): route the 256 columns. b3a | b4a | b5a go to T1 as terms, b1|b2 through an
    // fp32 LDS tile and leave as whole 384-B row segments; the b5 stem stays in the accumulators of waves 0, 1.
    if (wave >= 3 && wave <= 5) {            // wave-uniform: n-tiles 3,4,5 -> T1, through the row map
#pragma unroll
        for (int mt = 0; mt < TM; ++mt) {
            char* d = T1 + rowmap[mt * 32 + rlane] * S_LD1 + ((wave * 32 - 96) + h4) * 2;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint2 t0, t1, t2;
                split3x4(relu_f(acc[mt][4 * g]), relu_f(acc[mt][4 * g + 1]), relu_f(acc[mt][4 * g + 2]), relu_f(acc[mt][4 * g + 3]), t0, t1, t2);
                *reinterpret_cast<uint2*>(d + 16 * g) = t0;
                *reinterpret_cast<uint2*>(d + 16 * g + S_T1P) = t1;
                *reinterpret_cast<uint2*>(d + 16 * g + 2 * S_T1P) = t2;
            }
        }
    } else if (wave != 0) {                  // n-tiles 1,2 (b5s tail | b2) and 6,7 (b1 | padding) -> output tile
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int col = wave * 32 + 8 * g + h4;          // groups of 4 never straddle 48 / 240
            if (col >= 48 && col < 240) {
                const int ycol = col < 96 ? col : col - 192;        // position inside Y[:, 0:96) (b1 first, then b2)
#pragma unroll
                for (int mt = 0; mt < TM; ++mt)
                    *reinterpret_cast<float4*>(Ys + (mt * 32 + rlane) * S_LDY + ycol * 4) =
                        make_float4(relu_f(acc[mt][4 * g]), relu_f(acc[mt][4 * g + 1]), relu_f(acc[mt][4 * g + 2]),
                                    relu_f(acc[mt][4 * g + 3]));
            }
        }
    }
    DS_STAMP(3);
    lds_barrier();   // T1 and the b1|b2 tile complete
    DS_STAMP(4);
    // a job: conv KIND (1 b5b, 2 b3b, 3 b4b) on NM m-tiles from m0, n-tile nt, the weights in pf; the NM accumulator chains are
    // interleaved (a single unit is a chain of dependent LDS reads and MFMAs, i.e. latency)
    auto run_job = [&](auto kind_tag, auto nm_tag, int m0, int nt, const float4 (&wts)[30]) __attribute__((always_inline)) {
        constexpr int KIND = decltype(kind_tag)::value, NM = decltype(nm_tag)::value;
        if constexpr (NM > 0) {
            floatx16 u[NM];
            const float* bsrc = Bs + (KIND == 1 ? 0 : KIND == 2 ? 64 : 128) + nt * 32 + h4;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 t = *reinterpret_cast<const float4*>(bsrc + 8 * g);
#pragma unroll
                for (int m = 0; m < NM; ++m) { u[m][4 * g] = t.x; u[m][4 * g + 1] = t.y; u[m][4 * g + 2] = t.z; u[m][4 * g + 3] = t.w; }
            }
            int rm[NM];
#pragma unroll
            for (int m = 0; m < NM; ++m) rm[m] = rowmap[(m0 + m) * 32 + rlane];
            // T1 channels: b3a 0..31, b4a 32..63, b5a 64..95
            fuseds_conv_units<KIND == 3 ? 5 : 3, NM>(T1, rm, KIND == 1 ? 128 : KIND == 2 ? 0 : 64, lane, wts, u);
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                const int row = (m0 + m) * 32 + rlane;
                if (KIND == 1) {          // 1x3, 32 -> 64, ReLU, to T2 as terms                  layers.py:127-131
                    char* d = T2 + row * S_LD2 + (nt * 32 + h4) * 2;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        uint2 t0, t1, t2;
                        split3x4(relu_f(u[m][4 * g]), relu_f(u[m][4 * g + 1]), relu_f(u[m][4 * g + 2]), relu_f(u[m][4 * g + 3]), t0, t1, t2);
                        *reinterpret_cast<uint2*>(d + 16 * g) = t0;
                        *reinterpret_cast<uint2*>(d + 16 * g + S_T2P) = t1;
                        *reinterpret_cast<uint2*>(d + 16 * g + 2 * S_T2P) = t2;
                    }
                } else {
                    // KIND 2: 1x3, 32 -> 48, ReLU, to Y[96,144)   layers.py:106-110
                    // KIND 3: 1x5, 32 -> 48, ReLU, to Y[144,192)  layers.py:115-119
                    const int ybase = KIND == 2 ? 96 : 144;
                    if (row < TRv) {
#pragma unroll
                        for (int g = 0; g < 4; ++g)
                            if (nt * 32 + 8 * g < 48) {      // wave-uniform: 48 output channels = n-tile 0 and half of n-tile 1
                                v4f o = {relu_f(u[m][4 * g]), relu_f(u[m][4 * g + 1]), relu_f(u[m][4 * g + 2]), relu_f(u[m][4 * g + 3])};
                                *(__attribute__((address_space(1))) v4f*)(Yg + (unsigned)(row * 240 + ybase + nt * 32 + 8 * g + h4)) = o;
                            }
                    }
                }
            }
        }
    };

    // ---- P2a | barrier (T2 complete) | P2b, one code path per wave role with the barrier INSIDE the paths (every path passes exactly
    // one): the stem accumulators are then live in the path of waves 0, 1 only and each role's registers are its own (round 5: 250 ->
    // fewer VGPRs, which pays for the second fragment buffer of fuseds_conv_units)
    auto tail_and_store = [&]() __attribute__((always_inline)) {
        // branch 5 tail: 1x1 64 -> 48 (BN, no ReLU) accumulated on top of the stem conv held in acc,
        // then relu(stem + tail)                                                         layers.py:132-138
        const char* const tb = T2 + rlane * S_LD2 + (lane >> 5) * 16;
        float4 fr[2][3];
#pragma unroll
        for (int p = 0; p < 3; ++p) fr[0][p] = *reinterpret_cast<const float4*>(tb + p * S_T2P);
        int cur = 0;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 w[3] = {pf[3 * g], pf[3 * g + 1], pf[3 * g + 2]};
#pragma unroll
            for (int mt = 0; mt < TM; ++mt) {
                const int mn = mt + 1 < TM ? mt + 1 : 0, gn = mt + 1 < TM ? g : g + 1;
                if (gn < 4) {      // the next group's fragments, in front of this group's MFMAs
                    const char* q = tb + mn * 32 * S_LD2 + gn * 32;
#pragma unroll
                    for (int p = 0; p < 3; ++p) fr[cur ^ 1][p] = *reinterpret_cast<const float4*>(q + p * S_T2P);
                }
                acc[mt] = mfma3_lo(w, fr[cur][0], fr[cur][1], fr[cur][2], acc[mt]);
                acc[mt] = mfma3_hi(w, fr[cur][0], fr[cur][1], acc[mt]);
                cur ^= 1;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int mt = 0; mt < TM; ++mt) {
            const int row = mt * 32 + rlane;
            if (row < TRv) {
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    if (wave * 32 + 8 * g < 48) {
                        v4f o = {relu_f(acc[mt][4 * g]), relu_f(acc[mt][4 * g + 1]), relu_f(acc[mt][4 * g + 2]),
                                 relu_f(acc[mt][4 * g + 3])};
                        *(__attribute__((address_space(1))) v4f*)(Yg + (unsigned)(row * 240 + 192 + wave * 32 + 8 * g + h4)) = o;
                    }
            }
        }
    };
    if (wave < 2) {
        run_job(SplitRole<1>{}, SplitRole<TM>{}, 0, wave, pf);
        // the tail's twelve weight fragments travel in the unit registers (behind the job's MFMAs; its results are LDS writes)
#pragma unroll
        for (int g = 0; g < 12; ++g) pf[g] = gload4(a.Bp5c + ((size_t)(wave * 12 + g) * 64 + lane) * 4);
        DS_STAMP(5);
        lds_barrier();   // T2 complete
        DS_STAMP(6);
        tail_and_store();
    } else if (wave < 4) {
        run_job(SplitRole<2>{}, SplitRole<TM>{}, 0, wave - 2, pf);
        lds_barrier();
    } else {
        // the b1|b2 tile leaves as whole 384-byte row segments, by the four waves whose second-stage job waits for P2b; every
        // thread issues the same number of stores (slots past the tile's last one repeat it)
        constexpr int NFL = (TR32 * 24 + 255) / 256;
        const int last = TRv * 24 - 1, t4 = tid - 256;
#pragma unroll
        for (int i = 0; i < NFL; ++i) {
            const int idx = min(t4 + i * 256, last);
            const int row = idx / 24, q = idx - row * 24;
            const float4 v = *reinterpret_cast<const float4*>(Ys + row * S_LDY + q * 16);
            v4f o = {v.x, v.y, v.z, v.w};
            *(__attribute__((address_space(1))) v4f*)(Yg + (unsigned)(row * 240 + q * 4)) = o;
        }
        DS_STAMP(5);
        lds_barrier();
        DS_STAMP(6);
        if (wave >= 6) run_job(SplitRole<3>{}, SplitRole<TM - 1>{}, 1, wave - 6, pf);
        else run_job(SplitRole<3>{}, SplitRole<1>{}, 0, wave - 4, pf);
    }
    DS_STAMP(7);
#undef DS_STAMP
    // the next module reads the rows this one has stored (other waves' stores included: __syncthreads drains vmcnt) and
    // re-uses every LDS region
    if (mi + 1 < c.nmod) __syncthreads();
    }   // modules of the chain
}

// ---------------------------------------------------------------------------------------------------------------------
// conv_layer2 (1x1, 64 -> 128) + conv_layer3 (1x3, 128 -> 256), BN folded, ReLU (layers.py:192-203), split operands: the
// structure of stem23_kernel (ds_kernels.hip) -- a tile of whole sites (<= 96 rows), 8 waves, conv2's rows never leave LDS (zero halo
// row on either side of every site = conv3's SAME padding), conv3 with wave w = output channels [32 w, 32 w + 32) of all three
// m-tiles and its weights streamed global -> VGPR -- with the input rows split into terms while they are staged and conv2's rows
// split in its epilogue. LDS: input terms 96 x 400 B + conv2 terms (spt (W + 2) + 3) x 784 B = 117 KB, one workgroup per CU.
// Per tile conv3 streams 576 KB of weight fragments (24 k-steps x 3 terms x 8 n-tiles) for 27.6 k cycles of MFMA per SIMD (matrix
// pipe 0.51 busy).
constexpr int S23_SX = 400;      // input term row: 3 x 64 channels x 2 B + 16
constexpr int S23_ST = 784;      // conv2 term row: 3 x 128 channels x 2 B + 16 (49 x 16 B)
size_t stem23_split_lds_bytes(int W, int spt) { return (size_t)96 * S23_SX + (size_t)(spt * (W + 2) + 3) * S23_ST + 96 * 4; }

__global__ __launch_bounds__(512, 2) void stem23_split_kernel(const Stem23Args a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* const Xs = reinterpret_cast<char*>(smem);                  // [96][S23_SX]
    char* const T = Xs + 96 * S23_SX;                               // [spt * (W + 2) + 3][S23_ST]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int W = a.W, spt = a.spt;
    const int trows = spt * (W + 2) + 3;                            // last three rows: halo | dump row (padding rows of the tile) | halo
    int* const rowmap = reinterpret_cast<int*>(T + trows * S23_ST);
    const int site0 = blockIdx.x * spt;
    const int nhere = min(spt, a.n_sites - site0);
    const int TRv = nhere * W;
    const size_t grow0 = (size_t)site0 * W;
    const int h4 = 4 * (lane >> 5), rlane = lane & 31, hb = (lane >> 5) * 16;
    auto lds_barrier = []() __attribute__((always_inline)) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    // ---- stage: zero T (halo rows must be zero; the rest is overwritten), row map, input rows as terms
    for (int i = tid; i < trows * (S23_ST / 16); i += 512) reinterpret_cast<float4*>(T)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < 96) rowmap[tid] = tid < TRv ? (tid / W) * (W + 2) + 1 + tid % W : spt * (W + 2) + 1;
    {
        float4 v[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {                          // 96 rows x 16 float4
            const int idx = tid + 512 * i, row = idx >> 4, q = idx & 15;
            const int rr = row < TRv ? row : TRv - 1;
            v[i] = gload4(a.X + (grow0 + rr) * 64 + q * 4);
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int idx = tid + 512 * i, row = idx >> 4, q = idx & 15;
            uint2 t0, t1, t2;
            split3x4(v[i].x, v[i].y, v[i].z, v[i].w, t0, t1, t2);
            char* d = Xs + row * S23_SX + q * 8;
            *reinterpret_cast<uint2*>(d) = t0; *reinterpret_cast<uint2*>(d + 128) = t1; *reinterpret_cast<uint2*>(d + 256) = t2;
        }
    }
    // conv3 weights of the first two k-steps and both bias vectors are requested before the barrier
    const float* const b3 = a.Bp3 + ((size_t)wave * 24 * 3 * 64 + lane) * 4;       // n-tile `wave`: 24 k-steps x 3 terms of 1 KiB
    float4 bq[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p) bq[i][p] = gload4(b3 + (i * 3 + p) * 256);
    float4 bias3[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bias3[g] = gload4(a.bias3 + wave * 32 + 8 * g + h4);
    // conv2: unit u = (m, n); waves 0..3 take (0, w) and (2, w), waves 4..7 take (1, w - 4): three units per SIMD
    const int n2 = wave & 3;
    float4 w2[12];
#pragma unroll
    for (int g = 0; g < 12; ++g) w2[g] = gload4(a.Bp2 + ((size_t)(n2 * 12 + g) * 64 + lane) * 4);      // 4 k-steps x 3 terms
    float4 bias2[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bias2[g] = gload4(a.bias2 + n2 * 32 + 8 * g + h4);
    lds_barrier();
    {
        const int nunits = wave < 4 ? 2 : 1;
        for (int ui = 0; ui < nunits; ++ui) {
            const int m = wave < 4 ? 2 * ui : 1;
            floatx16 u;
#pragma unroll
            for (int g = 0; g < 4; ++g) { u[4 * g] = bias2[g].x; u[4 * g + 1] = bias2[g].y; u[4 * g + 2] = bias2[g].z; u[4 * g + 3] = bias2[g].w; }
            const char* xr = Xs + (m * 32 + rlane) * S23_SX + hb;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 x0 = *reinterpret_cast<const float4*>(xr + g * 32);
                const float4 x1 = *reinterpret_cast<const float4*>(xr + 128 + g * 32);
                const float4 x2 = *reinterpret_cast<const float4*>(xr + 256 + g * 32);
                const float4 w[3] = {w2[3 * g], w2[3 * g + 1], w2[3 * g + 2]};
                u = mfma3_lo(w, x0, x1, x2, u);
                u = mfma3_hi(w, x0, x1, u);
            }
            const int row = m * 32 + rlane;
            char* const td = T + rowmap[row] * S23_ST + (n2 * 32 + h4) * 2;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 o = make_float4(relu_f(u[4 * g]), relu_f(u[4 * g + 1]), relu_f(u[4 * g + 2]), relu_f(u[4 * g + 3]));
                uint2 t0, t1, t2;
                split3x4(o.x, o.y, o.z, o.w, t0, t1, t2);
                *reinterpret_cast<uint2*>(td + 16 * g) = t0;
                *reinterpret_cast<uint2*>(td + 16 * g + 256) = t1;
                *reinterpret_cast<uint2*>(td + 16 * g + 512) = t2;
                if (a.C2 && row < TRv) {                       // diagnostic tap (debug mode): conv_layer2's output rows
                    const v4f ov = {o.x, o.y, o.z, o.w};
                    *(__attribute__((address_space(1))) v4f*)(a.C2 + (grow0 + row) * 128 + n2 * 32 + 8 * g + h4) = ov;
                }
            }
        }
    }
    lds_barrier();

    // ---- conv3: 24 k-steps (3 taps x 8) without a barrier: the whole activation tile is resident; weights through a register ring
    // two k-steps deep, activation fragments one k-step ahead
    floatx16 acc[3];
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int g = 0; g < 4; ++g) { acc[m][4 * g] = bias3[g].x; acc[m][4 * g + 1] = bias3[g].y; acc[m][4 * g + 2] = bias3[g].z; acc[m][4 * g + 3] = bias3[g].w; }
    const char* tb[3];
#pragma unroll
    for (int m = 0; m < 3; ++m) tb[m] = T + (rowmap[m * 32 + rlane] - 1) * S23_ST + hb;     // tap t reads row + t - 1
    float4 af[2][3][3];
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int p = 0; p < 3; ++p) af[0][m][p] = *reinterpret_cast<const float4*>(tb[m] + p * 256);
#pragma unroll
    for (int ks = 0; ks < 24; ++ks) {
        const int cur = ks & 1;
        if (ks + 1 < 24) {
            const int t1 = (ks + 1) >> 3, g1 = (ks + 1) & 7;
#pragma unroll
            for (int m = 0; m < 3; ++m)
#pragma unroll
                for (int p = 0; p < 3; ++p) af[cur ^ 1][m][p] = *reinterpret_cast<const float4*>(tb[m] + t1 * S23_ST + g1 * 32 + p * 256);
        }
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            acc[m] = mfma3_lo(bq[cur], af[cur][m][0], af[cur][m][1], af[cur][m][2], acc[m]);
            acc[m] = mfma3_hi(bq[cur], af[cur][m][0], af[cur][m][1], acc[m]);
        }
        if (ks + 2 < 24) {
#pragma unroll
            for (int p = 0; p < 3; ++p) bq[cur][p] = gload4(b3 + ((ks + 2) * 3 + p) * 256);
        }
        __builtin_amdgcn_sched_barrier(0);       // pins the ring: requests stay two k-steps ahead of their MFMAs
    }

    // ---- ReLU, rows out (each lane: 4 x 16 B of one row)
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        const int row = m * 32 + rlane;
        if (row < TRv) {
            float* const yd = a.Y + (grow0 + row) * 256 + wave * 32 + h4;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const v4f o = {relu_f(acc[m][4 * g]), relu_f(acc[m][4 * g + 1]), relu_f(acc[m][4 * g + 2]), relu_f(acc[m][4 * g + 3])};
                *(__attribute__((address_space(1))) v4f*)(yd + 8 * g) = o;
            }
        }
    }
}

hipError_t launch_stem23_split(const Stem23Args& a, hipStream_t s)
{
    if (a.n_sites <= 0) return hipSuccess;
    if (a.spt * a.W > 96 || stem23_split_lds_bytes(a.W, a.spt) > 160 * 1024) return hipErrorInvalidValue;
    const int grid = (a.n_sites + a.spt - 1) / a.spt;
    hipLaunchKernelGGL(stem23_split_kernel, dim3(grid), dim3(512), stem23_split_lds_bytes(a.W, a.spt), s, a);
    return hipGetLastError();
}

// =====================================================================================================================
// BiLSTM cells and the joint model's dense(J, J), split operands (stage 2 of the split-operand engine).
//
// Both are plain GEMMs whose operands already lie in global memory as 1 KiB MFMA-fragment images, so they share one main loop:
// the structure of lstm_cell_bf16_kernel (ds_kernels.hip) -- workgroup tile 64 MTW rows x 64 NTW columns, 4 waves as 2 x 2,
// fragments global -> LDS by LDS-DMA into a ring of three stages, requests two stages ahead, ONE barrier per stage behind a
// counted vmcnt -- with every fragment in three terms: per k-step (16 k) a workgroup takes 3 (2 MTW + 2 NTW) KiB and issues
// 6 MTW NTW MFMAs per wave. The activation operand is fragment-major with the three terms of a k-step next to each other,
//     [m-tile of 32 rows][k-step of 16][term][64 lanes][8 bf16]      (lane (r, half) holds k = 16 s + 8 half .. + 7 of row 32 m + r),
// which is what a cell's epilogue writes for h (split once, where it is produced) and what pack_joint_split_kernel writes for the
// joint row; the weights are pack_b_split's panels [n-tile][k-step][term][64][8].
// One workgroup per CU means one wave per SIMD: hipcc's read -> wait -> MFMA schedule and the requests' issue time in front of the
// MFMAs cost these kernels 12 - 25 % until SplitRing::run_piped pinned the order (round 5). What bounds them since is what a CU takes in
// from the L2 (round 6): at full matrix rate a tile asks for 2048 (M + N) / (M N) B/clk -- 32 at 128 x 128, where
// the cells move 23; 18.7 at 256 x 192, where the dense is bound by the matrix pipe -- and the same 128 x 128 tile by EIGHT waves (two per
// SIMD, WM x WN = 4 x 2) takes the same time. The split cells use the 128 x 128 tile: slower alone than 64 x 64, faster in the pipelined step;
// 128 x 256 (24 B/clk, half the workgroups, 108 KB rings) is 5 % slower there.
constexpr int SPLIT_KSTEP_BYTES = 3 * 1024;               // the three term fragments of one k-step
constexpr int SPLIT_MT_BYTES = 16 * SPLIT_KSTEP_BYTES;    // one m-tile of a split h buffer (256 units = 16 k-steps)

#ifndef DS_SPLIT_PIPED
#define DS_SPLIT_PIPED 1        // SplitRing::run: 1 = the register-piped K loop with the pinned instruction order (shipped), 0 = one stage at a
#endif                          // time (run_simple: the loop of mid-round 5, kept for same-box A/Bs through tools/build_variant.sh)
#ifndef DS_SPLIT_KGS11
#define DS_SPLIT_KGS11 1
#endif
#ifndef DS_SPLIT_KGS22
#define DS_SPLIT_KGS22 1
#endif
#ifndef DS_RING_SCHED
#define DS_RING_SCHED 1
#endif
#ifndef DS_RING_BISECT
#define DS_RING_BISECT 0      // timing experiments only (results WRONG): 4 = no K loop at all; with DS_SPLIT_PIPED=0 also 1 = no MFMAs, 2 = no LDS-DMA requests, 3 = no barrier
#endif
#ifndef DS_SPLIT_LSTM_SLOTS
#define DS_SPLIT_LSTM_SLOTS 3
#endif
#ifndef DS_SPLIT_DENSE_SLOTS
#define DS_SPLIT_DENSE_SLOTS 4
#endif
// WM x WN waves (= 4), each MTW x NTW tiles of 32 x 32; NSLOT ring slots: stage st + NSLOT - 1 is requested while stage st is consumed
template <int MTW, int NTW, int WM = 2, int WN = 2, int NSLOT = 3>
struct SplitRing {
    static constexpr int NW = WM * WN;                       // waves per workgroup: 4, or 8 (two per SIMD: one wave's waits and requests under the other's MFMAs)
    static_assert(NW == 4 || NW == 8, "four or eight waves per workgroup");
    static_assert(NSLOT >= 3 && (NSLOT - 2) * ((3 * (WM * MTW + WN * NTW) + NW - 1) / NW) <= 63, "vmcnt holds six bits");
    static constexpr int FRA = WM * MTW, FRB = WN * NTW;
    static constexpr int NF1 = 3 * (FRA + FRB);              // 1 KiB fragments per k-step
    // k-steps per ring stage: one barrier per KGS k-steps. DS_SPLIT_KGS11 / DS_SPLIT_KGS22: measured choices for the 64 x 64 and the
    // 128 x 128 tile. The fragments of a stage are dealt to the four waves round robin; where NF is not a multiple of 4 (the 64 x 128
    // tile: 18) waves 0 .. NF % NW - 1 request one more than the others and wait with a count of their own.
    static constexpr int KGS = (MTW == 1 && NTW == 1) ? DS_SPLIT_KGS11 : (MTW == 2 && NTW == 2) ? DS_SPLIT_KGS22 : 1;
    static constexpr int NF = KGS * NF1;
    static constexpr int LPS = (NF + NW - 1) / NW;                 // LDS-DMA requests per wave and stage (waves >= NF % NW: one fewer if NF % NW)
    static constexpr int STAGE = NF * 256;                   // floats
    // (piped loop: waves with one fragment fewer send a filler request into a pad, so that every wave counts the same vmcnt)
    static constexpr int PAD_BYTES = (DS_SPLIT_PIPED && NF % NW != 0) ? NW * 1024 : 0;
    static constexpr size_t LDS_BYTES = (size_t)NSLOT * STAGE * 4 + PAD_BYTES;
#ifndef DS_RING_CLOCK
#define DS_RING_CLOCK 0       // 1: s_memtime around the piped loop's wait / barrier / requests / rest, printed by two workgroups of the dense kernel
#endif
#if DS_RING_CLOCK
    mutable unsigned long long clk[5] = {0, 0, 0, 0, 0};
#endif
    const char* src[LPS];
    mutable const char* rs[LPS];          // piped loop: request sources of stage 0, moved to the second A segment when the stages reach it
    mutable bool in_seg1 = false;
    int kgi_[LPS];
    bool is_a[LPS];
    long dseg;
    int s0;
    unsigned ring_lds, lane16;
    int wave;

    // A operand: k-steps [0, s0) from a0, the rest from a1 (both [m-tile][k-step][term] images, m-tile stride a_mt bytes);
    // this workgroup's m-tiles mt0 .. mt0 + FRA - 1 (clamped to mtiles - 1), n-tiles nt0 .. nt0 + FRB - 1 of panel B
    __device__ __forceinline__ void init(float* ring, int wave_, int lane, const char* a0, const char* a1, int s0_, long a_mt, int mt0, int mtiles,
                                         const char* B, int kg_stride, int nt0)
    {
        wave = wave_;
        s0 = s0_;
        lane16 = (unsigned)lane * 16;
        ring_lds = (unsigned)(size_t)(__attribute__((address_space(3))) float*)ring;
        dseg = (a1 - a0) - (long)s0_ * SPLIT_KSTEP_BYTES;
#pragma unroll
        for (int j = 0; j < LPS; ++j) {
            const int q = min(wave_ + NW * j, NF - 1), kgi = q / NF1, f = q - kgi * NF1;       // (q >= NF: never requested)
            kgi_[j] = kgi;
            is_a[j] = f < 3 * FRA;
            if (f < 3 * FRA) {
                const int m = min(mt0 + f / 3, mtiles - 1);
                src[j] = a0 + (size_t)m * a_mt + (size_t)kgi * SPLIT_KSTEP_BYTES + (f % 3) * 1024;
            } else {
                const int g = f - 3 * FRA;
                src[j] = B + ((size_t)(nt0 + g / 3) * kg_stride + kgi) * SPLIT_KSTEP_BYTES + (g % 3) * 1024;
            }
            rs[j] = src[j];
        }
    }
    __device__ __forceinline__ void request(int st, int slot) const
    {
        const unsigned dst = __builtin_amdgcn_readfirstlane(ring_lds + (slot * STAGE + wave * 256) * 4);
#pragma unroll
        for (int j = 0; j < LPS; ++j) {
            if (NF % NW != 0 && j == LPS - 1 && wave >= NF % NW) break;      // wave-uniform: this wave has no fragment 4 j + wave
            const int ks = st * KGS + kgi_[j];
            const long off = (long)st * (KGS * SPLIT_KSTEP_BYTES) + ((is_a[j] && ks >= s0) ? dseg : 0);
            glds16s(src[j] + off, lane16, dst + j * (NW * 1024));      // fragment q = wave + 4 j of the stage
        }
    }
    // ---- piped loop, requests: the source of request J is (wave-uniform base rs[J]) + (lane offset + stage offset, ONE vector add per
    // stage): no scalar address arithmetic per request; no branch either -- past the last stage the requests repeat it into the slot
    // that has just been freed, and a wave without a fragment 4 J + wave refetches its last one into a pad
    __device__ __forceinline__ unsigned stage_voff(int sreq) const
    {
        if (!in_seg1 && sreq >= s0) {
#pragma unroll
            for (int j = 0; j < LPS; ++j)
                if (is_a[j]) rs[j] += dseg;
            in_seg1 = true;
        }
        return lane16 + (unsigned)sreq * SPLIT_KSTEP_BYTES;
    }
    template <int J>
    __device__ __forceinline__ void request3(unsigned voff, unsigned rdst) const
    {
        unsigned dst = rdst + J * (NW * 1024);
        if (NF % NW != 0 && J == LPS - 1 && wave >= NF % NW) dst = ring_lds + NSLOT * STAGE * 4 + wave * 1024;
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(rs[J]), "s"(dst) : "memory");      // (m0: saved and restored around the run of requests by the caller; hipcc rejects m0 as a clobber -- "reserved register" -- so tests/test_kernel_resources.py holds what the save/restore relies on: no SGPR spill, hence no compiler use of m0, in these kernels)
    }
    template <int J>
    __device__ __forceinline__ void request3_all(unsigned voff, unsigned rdst) const
    {
        if constexpr (J < LPS) { request3<J>(voff, rdst); request3_all<J + 1>(voff, rdst); }
    }
    __device__ __forceinline__ void prologue(int nstages) const
    {
#if DS_SPLIT_PIPED
        static_assert(KGS == 1 && (NSLOT - 1) * LPS <= 63, "one k-step per stage; vmcnt holds six bits");
        if (nstages <= 0 || DS_RING_BISECT == 4) return;
        unsigned keep_m0;
        asm volatile("s_mov_b32 %0, m0" : "=s"(keep_m0));
#pragma unroll
        for (int s = 0; s < NSLOT; ++s) {
            const unsigned voff = stage_voff(min(s, nstages - 1));
            request3_all<0>(voff, __builtin_amdgcn_readfirstlane(ring_lds + (s * STAGE + wave * 256) * 4));
        }
        asm volatile("s_mov_b32 m0, %0" ::"s"(keep_m0));
        return;
#endif
#pragma unroll
        for (int s = 0; s < NSLOT - 1; ++s)
            if (DS_RING_BISECT != 2 && DS_RING_BISECT != 4 && s < nstages) request(s, s);
    }
    // wait until all but the last K stages requested have landed (this wave's share; waves >= NF % NW request one fragment fewer)
    template <int K>
    __device__ __forceinline__ void wait_but(int later) const
    {
        if (later >= K) {
            if (NF % NW == 0 || wave < NF % NW) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K * LPS) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K * (LPS - 1)) : "memory");
        } else if constexpr (K > 0) wait_but<K - 1>(later);
    }
    // stage st has landed once every wave's requests for it are done: counted wait (stages st + 1 .. st + NSLOT - 2 stay in flight),
    // barrier -- which also frees ring slot (st - 1) % NSLOT, read during stage st - 1 --, request stage st + NSLOT - 1 into it, then
    // this stage's MFMAs
    template <int SLOT>
    __device__ __forceinline__ void stage(int st, int nstages, const float* fa0, const float* fb0, floatx16 (&acc)[MTW][NTW]) const
    {
        wait_but<NSLOT - 2>(nstages - 1 - st);
        if (DS_RING_BISECT != 3) __builtin_amdgcn_s_barrier();
        if (DS_RING_BISECT != 2 && st + NSLOT - 1 < nstages) request(st + NSLOT - 1, (SLOT + NSLOT - 1) % NSLOT);
#pragma unroll
        for (int kgi = 0; kgi < KGS; ++kgi) {
            float4 a[MTW][3], b[NTW][3];
#pragma unroll
            for (int i = 0; i < MTW; ++i)
#pragma unroll
                for (int p = 0; p < 3; ++p) a[i][p] = *reinterpret_cast<const float4*>(fa0 + SLOT * STAGE + (kgi * NF1 + i * 3 + p) * 256);
#pragma unroll
            for (int j = 0; j < NTW; ++j)
#pragma unroll
                for (int p = 0; p < 3; ++p) b[j][p] = *reinterpret_cast<const float4*>(fb0 + SLOT * STAGE + (kgi * NF1 + j * 3 + p) * 256);
#pragma unroll
            for (int i = 0; i < MTW; ++i)
#pragma unroll
                for (int j = 0; j < NTW; ++j) {
                    if (DS_RING_BISECT == 1) { acc[i][j][0] += b[j][0].x + b[j][1].y + b[j][2].z + a[i][0].x + a[i][1].y + a[i][2].z; continue; }
                    acc[i][j] = mfma3_lo(b[j], a[i][0], a[i][1], a[i][2], acc[i][j]);      // transposed: (A B)^T
                    acc[i][j] = mfma3_hi(b[j], a[i][0], a[i][1], acc[i][j]);
                }
#if DS_RING_SCHED == 1
            // hipcc's own schedule reads a fragment, waits for it and issues its MFMA, one after the other, in 20 registers: every LDS
            // latency is exposed (one wave per SIMD). All of a k-step's reads first, then its MFMAs behind counted waits.
            __builtin_amdgcn_sched_group_barrier(0x100, 3 * (MTW + NTW), 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 6 * MTW * NTW, 0);
#endif
        }
    }
    template <int SLOT>
    __device__ __forceinline__ void run_from(int& st, int nstages, const float* fa0, const float* fb0, floatx16 (&acc)[MTW][NTW]) const
    {
        stage<SLOT>(st, nstages, fa0, fb0, acc);
        if (++st >= nstages) return;
        if constexpr (SLOT + 1 < NSLOT) run_from<SLOT + 1>(st, nstages, fa0, fb0, acc);
    }
    __device__ __forceinline__ void run_simple(int nstages, const float* fa0, const float* fb0, floatx16 (&acc)[MTW][NTW]) const
    {
        if (DS_RING_BISECT == 4) return;
        for (int st = 0; st < nstages;) run_from<0>(st, nstages, fa0, fb0, acc);
    }

    // The same ring with a stage's fragments DOUBLE-BUFFERED IN REGISTERS. With one wave per SIMD (one workgroup per CU: the 128 x 96 dense
    // tile, the 128 x 128 cell tile) nothing else issues while a wave waits, so run_simple pays per k-step, one after the other: the
    // barrier, the requests, the LDS latency of the first fragments, then the MFMAs (dense(6032, 6032), 512 sites: 1,080 cycles per
    // k-step WITHOUT any operand traffic, for 576 of MFMA). Here iteration st holds stage st in registers; it waits until stage st + 1
    // has landed, passes the barrier (everybody's share of st + 1 has landed, everybody has READ stage st: its slot is free), requests
    // stage st + NSLOT into that slot and then issues the LDS reads of stage st + 1 BETWEEN the MFMAs of stage st -- the interleave is
    // pinned instruction by instruction (sched_barrier), hipcc's own schedule serialises read -> wait -> MFMA. NSLOT stages are requested ahead.
    // The MFMAs of one accumulator keep their order: bit-identical to run_simple.
    static_assert(!DS_SPLIT_PIPED || KGS == 1, "the piped loop takes one k-step per stage");
    typedef float4 Frag[MTW + NTW][3];
    __device__ __forceinline__ void read_frags(const float* fa0, const float* fb0, int slot, Frag& f) const
    {
        const float* fa = fa0 + slot * STAGE;
        const float* fb = fb0 + slot * STAGE;
#pragma unroll
        for (int i = 0; i < MTW; ++i)
#pragma unroll
            for (int p = 0; p < 3; ++p) f[i][p] = *reinterpret_cast<const float4*>(fa + (i * 3 + p) * 256);
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
            for (int p = 0; p < 3; ++p) f[MTW + j][p] = *reinterpret_cast<const float4*>(fb + (j * 3 + p) * 256);
    }
    static constexpr int req_pos(int j)        // request j of a stage goes behind MFMA (j NM) / LPS + 1 of the pinned sequence
    {
        const int nm = 6 * MTW * NTW, p = (j * nm + LPS - 1) / LPS + 1;
        return p < nm ? p : nm - 1;
    }
    template <int M, int J>
    __device__ __forceinline__ void piped_req(unsigned voff, unsigned rdst) const
    {
        if constexpr (J < LPS) {
            if constexpr (req_pos(J) == M) request3<J>(voff, rdst);
            piped_req<M, J + 1>(voff, rdst);
        }
    }
    static constexpr bool ROLL = MTW >= 3;
    static constexpr int read_frag(int m)
    {
        if (!ROLL) return m < 3 * (MTW + NTW) ? m : -1;
        // the weight fragments and the LAST m-tile's activations first (that m-tile's MFMAs close the sequence), then m-tile i's
        // activations behind its last MFMA
        if (m < 3 * NTW) return (MTW + m / 3) * 3 + m % 3;
        if (m < 3 * NTW + 3) return (MTW - 1) * 3 + (m - 3 * NTW);
        for (int i = 0; i + 1 < MTW; ++i) {
            const int base = (i + 1) * 6 * NTW;
            if (m >= base && m < base + 3) return i * 3 + (m - base);
        }
        return -1;
    }
    static_assert(!ROLL || 3 * NTW + 3 <= 6 * NTW, "the early reads end before the first m-tile's MFMAs do");
    // element M of the pinned sequence
    template <int M>
    __device__ __forceinline__ void piped_seq(unsigned voff, unsigned rdst, const float* fa, const float* fb, const Frag& cur, Frag& nxt,
                                              floatx16 (&acc)[MTW][NTW]) const
    {
        constexpr int NT_ = MTW * NTW, NM = 6 * NT_;
        if constexpr (M < NM) {
            // small tiles: product-major (every MFMA of a round goes to another accumulator). Tiles of three and more m-tiles per wave:
            // m-tile-major, so that an activation fragment is dead after its 6 NTW MFMAs and the next stage's copy can take its
            // registers (both stages' fragments live at once are 168 registers beside 192 of accumulators: hipcc spilled inside the loop)
            constexpr int P = ROLL ? (M % (6 * NTW)) / NTW : M / NT_, I = ROLL ? M / (6 * NTW) : (M % NT_) / NTW, J = M % NTW;
            // products in mfma3_lo / mfma3_hi order: (w2, a0) (w0, a2) (w1, a1) | (w1, a0) (w0, a1) (w0, a0)
            constexpr int WI = P == 0 ? 2 : P == 1 ? 0 : P == 2 ? 1 : P == 3 ? 1 : 0;
            constexpr int AI = P == 0 ? 0 : P == 1 ? 2 : P == 2 ? 1 : P == 3 ? 0 : P == 4 ? 1 : 0;
            acc[I][J] = mfma_bf(cur[MTW + J][WI], cur[I][AI], acc[I][J]);
            constexpr int RF = read_frag(M);          // 3 fragment + term of the next stage's LDS read behind this MFMA, or -1
            if constexpr (RF >= 0) {
                constexpr int F = RF / 3, TERM = RF % 3;
                nxt[F][TERM] = *reinterpret_cast<const float4*>((F < MTW ? fa + (F * 3 + TERM) * 256 : fb + ((F - MTW) * 3 + TERM) * 256));
            }
            piped_req<M, 0>(voff, rdst);
            __builtin_amdgcn_sched_barrier(0);
            piped_seq<M + 1>(voff, rdst, fa, fb, cur, nxt, acc);
        }
    }
    // one iteration: registers `cur` hold stage st (slot = its ring slot), `nxt` take stage st + 1
    __device__ __forceinline__ void piped(int st, int slot, int nstages, const float* fa0, const float* fb0, const Frag& cur, Frag& nxt, floatx16 (&acc)[MTW][NTW]) const
    {
#if DS_RING_CLOCK
        const unsigned long long c0 = __builtin_amdgcn_s_memtime();
        if (clk[4]) clk[3] += c0 - clk[4];
#endif
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NSLOT - 2) * LPS) : "memory");
#if DS_RING_CLOCK
        const unsigned long long c1 = __builtin_amdgcn_s_memtime();
#endif
        __builtin_amdgcn_s_barrier();
#if DS_RING_CLOCK
        const unsigned long long c2 = __builtin_amdgcn_s_memtime();
#endif
        // the order is pinned instruction by instruction: MFMA m (round robin over the accumulators, the six products of one accumulator
        // in mfma3_lo / mfma3_hi order), behind it LDS read m of the next stage, and every NM / LPS MFMAs one LDS-DMA request. In one
        // burst behind the barrier the four waves' 21 requests held every wave for 350 cycles per k-step in front of its MFMAs
        // (s_memtime: wait 150, barrier 20 - 50, requests 350, reads + MFMAs 600 of 1,150)
        const unsigned voff = stage_voff(min(st + NSLOT, nstages - 1));
        const unsigned rdst = __builtin_amdgcn_readfirstlane(ring_lds + (slot * STAGE + wave * 256) * 4);
        const int nslot = slot + 1 == NSLOT ? 0 : slot + 1;
        const float* fa = fa0 + nslot * STAGE;
        const float* fb = fb0 + nslot * STAGE;
#if DS_RING_CLOCK
        const unsigned long long c3 = __builtin_amdgcn_s_memtime();
        clk[0] += c1 - c0; clk[1] += c2 - c1; clk[2] += c3 - c2; clk[4] = c3;
#endif
        unsigned keep_m0;
        asm volatile("s_mov_b32 %0, m0" : "=s"(keep_m0));
        __builtin_amdgcn_sched_barrier(0);
        piped_seq<0>(voff, rdst, fa, fb, cur, nxt, acc);
        asm volatile("s_mov_b32 m0, %0" ::"s"(keep_m0));
        __builtin_amdgcn_sched_barrier(0);      // (the next iteration's lgkmcnt(0) stays behind this iteration's last MFMAs)
    }
    __device__ __forceinline__ void run_piped(int nstages, const float* fa0, const float* fb0, floatx16 (&acc)[MTW][NTW]) const
    {
        if (nstages <= 0) return;
        // (prologue() has requested stages 0 .. NSLOT - 1)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSLOT - 1) * LPS) : "memory");
        __builtin_amdgcn_s_barrier();
        Frag f0, f1;
        read_frags(fa0, fb0, 0, f0);
        int slot = 0, st = 0;
        // every stage through the same pinned sequence, two per trip (the register sets swap roles) -- the last one reads a "next stage"
        // nobody uses: a separate tail of plain MFMAs made hipcc keep the 256 x 192 tile's accumulators in two places and spill inside the loop
        for (int p = 0; p < (nstages >> 1); ++p) {
            piped(st, slot, nstages, fa0, fb0, f0, f1, acc);
            ++st; slot = slot + 1 == NSLOT ? 0 : slot + 1;
            piped(st, slot, nstages, fa0, fb0, f1, f0, acc);
            ++st; slot = slot + 1 == NSLOT ? 0 : slot + 1;
        }
        if (nstages & 1) piped(st, slot, nstages, fa0, fb0, f0, f1, acc);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // (the filler requests and reads of the last iterations)
    }
    __device__ __forceinline__ void run(int nstages, const float* fa0, const float* fb0, floatx16 (&acc)[MTW][NTW]) const
    {
#if DS_SPLIT_PIPED
        if (DS_RING_BISECT == 4) return;
        run_piped(nstages, fa0, fb0, acc);
#else
        run_simple(nstages, fa0, fb0, acc);
#endif
    }
};

// ---- the BiLSTM cells of one wavefront diagonal (layers.py:45-72), split operands: h and the weights in three bf16 terms, six
// products per MAC, fp32 accumulate; the layer-0 table row / rank-1 terms, the gates and the cell state are fp32 as in the fp32
// kernel (lstm_acc_init / lstm_gates are shared with it); h is split once, in the epilogue that produces it.
template <int MTW, int NTW, int WM = 2, int WN = 2>
__global__ __launch_bounds__(64 * WM * WN, (SplitRing<MTW, NTW, WM, WN, DS_SPLIT_LSTM_SLOTS>::LDS_BYTES > 80 * 1024) ? 1 : 2) void lstm_cell_split_kernel(const LstmLaunch L_)
{
    const LstmLaunch* const Lp = &L_;
    typedef SplitRing<MTW, NTW, WM, WN, DS_SPLIT_LSTM_SLOTS> R;
    extern __shared__ __attribute__((aligned(16))) float ring[];    // [NSLOT * STAGE]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int mi = wave % WM, nj = wave / WM;
    const int bid = lstm_logical_tile(blockIdx.x, gridDim.x, Lp->cls_tiles[0], Lp->cls_tiles[1]);
    const int mtiles = Lp->mtiles;
    const int mblocks = (mtiles + R::FRA - 1) / R::FRA;
    constexpr int NGROUPS = 32 / R::FRB;
    const int per_cell = mblocks * NGROUPS;
    const int ci = bid / per_cell, rem = bid - ci * per_cell;
    const int ng = rem % NGROUPS, mb = rem / NGROUPS;          // the n-groups of one m-block are neighbours (lstm_cell_bf16_kernel)
    const LstmCell& C = Lp->cell[ci];
    const int half = lane >> 5, r31 = lane & 31;
    const int n = Lp->n, T = Lp->T;
    const unsigned lane4 = (unsigned)lane * 4;
    const bool has_x = C.ax != nullptr, has_h = C.ah != nullptr;
    const int KS = (has_x ? 16 : 0) + (has_h ? 16 : 0);       // k-steps: x rows first, then h rows (TF kernel order)
    const int nstages = KS / R::KGS;
#if DS_RING_CLOCK
    const unsigned long long k0 = __builtin_amdgcn_s_memtime();
#endif
    R rg;
    rg.init(ring, wave, lane, reinterpret_cast<const char*>(has_x ? C.ax : C.ah), reinterpret_cast<const char*>(has_h ? C.ah : C.ax),
            has_x ? 16 : 0, SPLIT_MT_BYTES, mb * R::FRA, mtiles, reinterpret_cast<const char*>(C.Bp), C.kg_stride, ng * R::FRB);
    rg.prologue(nstages);

    int mt[MTW];
    bool valid[MTW];
#pragma unroll
    for (int i = 0; i < MTW; ++i) {
        const int raw = mb * R::FRA + mi * MTW + i;
        valid[i] = raw < mtiles;
        mt[i] = valid[i] ? raw : mtiles - 1;
    }
    floatx16 acc[MTW][NTW];
    float4 cp[MTW][NTW];
    {
        int ntl[NTW], rowc[MTW];
#pragma unroll
        for (int j = 0; j < NTW; ++j) ntl[j] = ng * R::FRB + nj * NTW + j;
#pragma unroll
        for (int i = 0; i < MTW; ++i) rowc[i] = min(mt[i] * 32 + r31, n - 1);
        if (C.xinit) {                                     // wave-uniform: layer 0 behind lstm_xproj_kernel's image (DS_TUNE_LSTM_XPROJ_ALL)
#pragma unroll
            for (int i = 0; i < MTW; ++i)
#pragma unroll
                for (int j = 0; j < NTW; ++j) lstm_acc_load(C.xinit, mt[i], ntl[j], lane4, acc[i][j]);
        } else {
            // (bias and the rank-1 rows on the SCALAR path -- s_load_dwordx8 + selects instead of ~80 L2-hot float4 per lane -- was built and
            // measured in round 6, git 755c1c9: 383 against 373 us per step alone, pipelined +- 0: eight dependent scalar round trips)
#pragma unroll
            for (int i = 0; i < MTW; ++i)
#pragma unroll
                for (int j = 0; j < NTW; ++j) lstm_acc_init(C, ntl[j] * 8 + 4 * half, rowc[i], T, acc[i][j]);
        }
    }
    const bool c_zero = C.c_zero != 0;
#pragma unroll
    for (int i = 0; i < MTW; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            cp[i][j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!c_zero) cp[i][j] = gload4(C.c + (size_t)mt[i] * LSTM_MT_FLOATS + (unsigned)(ng * R::FRB + nj * NTW + j) * 256 + lane4);
        }
    // every compiler-visible load is retired here: the loop's vmcnt waits count the LDS-DMA requests only (lstm_cell_bf16_kernel)
#pragma unroll
    for (int i = 0; i < MTW; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            asm volatile("" : "+v"(cp[i][j].x), "+v"(cp[i][j].y), "+v"(cp[i][j].z), "+v"(cp[i][j].w));
            asm volatile("" : "+v"(acc[i][j]));
        }
#if DS_RING_CLOCK
    const unsigned long long k1 = __builtin_amdgcn_s_memtime();
#endif
    rg.run(nstages, ring + (mi * MTW * 3) * 256 + lane4, ring + (3 * R::FRA + nj * NTW * 3) * 256 + lane4, acc);
#if DS_RING_CLOCK
    const unsigned long long k2 = __builtin_amdgcn_s_memtime();
#endif

    // ---- gates (fp32), new state; c fragment-major fp32, h fragment-major in three terms (8 bytes per lane and term), optional
    // row-major fp32 h for the joint model
#pragma unroll
    for (int i = 0; i < MTW; ++i) {
        if (!valid[i]) continue;
        const int row = mt[i] * 32 + r31;
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const int ntile = ng * R::FRB + nj * NTW + j;
            float4 cn, hn;
            lstm_gates(acc[i][j], cp[i][j], cn, hn);
            const v4f co = {cn.x, cn.y, cn.z, cn.w};
            *(__attribute__((address_space(1))) v4f*)(C.c + (size_t)mt[i] * LSTM_MT_FLOATS + (unsigned)ntile * 256 + lane4) = co;
            uint2 t0, t1, t2;
            split3x4(hn.x, hn.y, hn.z, hn.w, t0, t1, t2);
            char* const hb = reinterpret_cast<char*>(C.h_out) + (size_t)mt[i] * SPLIT_MT_BYTES + (unsigned)(ntile >> 1) * SPLIT_KSTEP_BYTES +
                             (unsigned)(((ntile & 1) * 32 + r31) * 16 + half * 8);
            typedef unsigned int u2v __attribute__((ext_vector_type(2)));
            *(__attribute__((address_space(1))) u2v*)hb = u2v{t0.x, t0.y};
            *(__attribute__((address_space(1))) u2v*)(hb + 1024) = u2v{t1.x, t1.y};
            *(__attribute__((address_space(1))) u2v*)(hb + 2048) = u2v{t2.x, t2.y};
            if (C.h_row && row < n) {
                const v4f hr = {hn.x, hn.y, hn.z, hn.w};
                *(__attribute__((address_space(1))) v4f*)(C.h_row + (size_t)row * 256 + ntile * 8 + 4 * half) = hr;
            }
        }
    }
#if DS_RING_CLOCK
    const unsigned long long k3 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long k4 = __builtin_amdgcn_s_memtime();
    if ((blockIdx.x == 0 || blockIdx.x == gridDim.x - 1 || blockIdx.x == 77) && threadIdx.x == 0 && C.t == 8)
        printf("CELL block %d of %d cell %d ksteps %d: cycles init %d loop %d gates+stores issued %d stores done %d\n", (int)blockIdx.x, (int)gridDim.x, ci, KS,
               (int)(k1 - k0), (int)(k2 - k1), (int)(k3 - k2), (int)(k4 - k3));
#endif
}

// ---- layer 0's accumulator-initial values for every step of both directions, and the first step's cells (ds_internal.h LstmXproj).
// One wave = one 32-site x 8-unit tile of one (direction, step): lstm_acc_init's arithmetic (the bits a cell would compute itself),
// stored as the four 1 KiB fragments lstm_acc_load reads back; step 0 (h = 0, c = 0: no matrix product) goes through the gates here
// and leaves c and the split h image exactly as lstm_cell_split_kernel's epilogue does.
__global__ __launch_bounds__(256) void lstm_xproj_kernel(const LstmXproj X)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int dir = blockIdx.z, sidx = blockIdx.y;
    const int mtile = blockIdx.x >> 3, ntile = (blockIdx.x & 7) * 4 + wave;
    const int half = lane >> 5, r31 = lane & 31;
    const unsigned lane4 = (unsigned)lane * 4;
    const int t = dir == 0 ? sidx : X.T - 1 - sidx;
    LstmCell C = X.cell[dir];
    C.t = t;
    const int row = mtile * 32 + r31;
    floatx16 acc;
    lstm_acc_init(C, ntile * 8 + 4 * half, row < X.n ? row : X.n - 1, X.T, acc);
    if (sidx > 0) {
        float* const p = X.xinit[dir] + (size_t)t * X.x_step + ((size_t)(mtile * 32 + ntile) * 4) * 256 + lane4;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const v4f o = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
            *(__attribute__((address_space(1))) v4f*)(p + g * 256) = o;
        }
        return;
    }
    float4 cn, hn;
    lstm_gates(acc, make_float4(0.f, 0.f, 0.f, 0.f), cn, hn);
    const v4f co = {cn.x, cn.y, cn.z, cn.w};
    *(__attribute__((address_space(1))) v4f*)(C.c + (size_t)mtile * LSTM_MT_FLOATS + (unsigned)ntile * 256 + lane4) = co;
    uint2 t0, t1, t2;
    split3x4(hn.x, hn.y, hn.z, hn.w, t0, t1, t2);
    char* const hb = reinterpret_cast<char*>(C.h_out + (size_t)t * X.h_step) + (size_t)mtile * SPLIT_MT_BYTES + (unsigned)(ntile >> 1) * SPLIT_KSTEP_BYTES +
                     (unsigned)(((ntile & 1) * 32 + r31) * 16 + half * 8);
    typedef unsigned int u2v __attribute__((ext_vector_type(2)));
    *(__attribute__((address_space(1))) u2v*)hb = u2v{t0.x, t0.y};
    *(__attribute__((address_space(1))) u2v*)(hb + 1024) = u2v{t1.x, t1.y};
    *(__attribute__((address_space(1))) u2v*)(hb + 2048) = u2v{t2.x, t2.y};
}

hipError_t launch_lstm_xproj(const LstmXproj& X, hipStream_t s)
{
    if (X.n <= 0 || X.mtiles <= 0) return hipSuccess;
    hipLaunchKernelGGL(lstm_xproj_kernel, dim3(X.mtiles * 8, X.nsteps, 2), dim3(256), 0, s, X);
    return hipGetLastError();
}

hipError_t launch_lstm_cells_split(int tile, const LstmLaunch& L, hipStream_t s)
{
    const int ncell = L.ncell, mtiles = L.mtiles;
    if (ncell <= 0 || mtiles <= 0) return hipSuccess;
    switch (tile) {
    case 11: hipLaunchKernelGGL((lstm_cell_split_kernel<1, 1>), dim3(ncell * ((mtiles + 1) / 2) * 16), dim3(256), (SplitRing<1, 1, 2, 2, DS_SPLIT_LSTM_SLOTS>::LDS_BYTES), s, L); break;
    case 12: hipLaunchKernelGGL((lstm_cell_split_kernel<1, 2>), dim3(ncell * ((mtiles + 1) / 2) * 8), dim3(256), (SplitRing<1, 2, 2, 2, DS_SPLIT_LSTM_SLOTS>::LDS_BYTES), s, L); break;
    case 22: hipLaunchKernelGGL((lstm_cell_split_kernel<2, 2>), dim3(ncell * ((mtiles + 3) / 4) * 8), dim3(256), (SplitRing<2, 2, 2, 2, DS_SPLIT_LSTM_SLOTS>::LDS_BYTES), s, L); break;
    // the same 128 x 128 workgroup tile by EIGHT waves (4 x 2, each 32 x 64): two waves per SIMD of the one workgroup a CU holds
    case 28: hipLaunchKernelGGL((lstm_cell_split_kernel<1, 2, 4, 2>), dim3(ncell * ((mtiles + 3) / 4) * 8), dim3(512), (SplitRing<1, 2, 4, 2, DS_SPLIT_LSTM_SLOTS>::LDS_BYTES), s, L); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ---- dense(J, J) of the joint model (layers.py:257-259: no bias, no activation), split operands. A = the joint rows
// [h_fw(T-1) | h_bw(0) | signal features] as a fragment-major term image (pack_joint_split_kernel), B = W1's panels.
__global__ __launch_bounds__(256) void pack_joint_split_kernel(const SplitDense d)
{
    // one thread per (row, k-step half): 8 consecutive k of one row -> the 16 bytes of its lane slot in each of the three term fragments.
    // A wave = the 64 lane slots of ONE (m-tile, k-step): its three stores are whole 1 KiB fragments (with the k-step half fastest over
    // the threads every store was a lone 16 bytes: 38 MB written for the image's 18.5 MB, PMC)
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const int ksteps = d.ksteps;
    const long total = (long)d.mtiles * 32 * ksteps * 2;
    if (idx >= total) return;
    const int half = (int)((idx >> 5) & 1);
    const long t = idx >> 6;
    const int ks = (int)(t % ksteps);
    const int row = (int)(t / ksteps) * 32 + (int)(idx & 31);
    const int k0 = ks * 16 + half * 8;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.0f;
    if (row < d.n) {
        // the joint row's three segments (lengths are multiples of 8)
        const float* p = nullptr;
        int k = k0;
        if (k < d.len[0]) p = d.seg[0] + (size_t)row * d.len[0] + k;
        else if ((k -= d.len[0]) < d.len[1]) p = d.seg[1] + (size_t)row * d.len[1] + k;
        else { k -= d.len[1]; p = d.seg[2] + (size_t)row * d.len[2] + k; }
        const float4 x = gload4(p), y = gload4(p + 4);
        v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w; v[4] = y.x; v[5] = y.y; v[6] = y.z; v[7] = y.w;
    }
    uint2 a0, a1, a2, b0, b1, b2;
    split3x4(v[0], v[1], v[2], v[3], a0, a1, a2);
    split3x4(v[4], v[5], v[6], v[7], b0, b1, b2);
    char* dst = d.A + ((size_t)(row >> 5) * ksteps + ks) * SPLIT_KSTEP_BYTES + (size_t)(half * 32 + (row & 31)) * 16;
    *reinterpret_cast<uint4*>(dst) = make_uint4(a0.x, a0.y, b0.x, b0.y);
    *reinterpret_cast<uint4*>(dst + 1024) = make_uint4(a1.x, a1.y, b1.x, b1.y);
    *reinterpret_cast<uint4*>(dst + 2048) = make_uint4(a2.x, a2.y, b2.x, b2.y);
}

// Workgroup tile = (32 WM MTW) rows x (32 WN NTW) columns; the launcher picks it by forward size.
template <int MTW, int NTW, int WM, int WN>
struct DenseRing {           // (the 256 x 192 tile's stage is 42 KiB: three slots)
    typedef SplitRing<MTW, NTW, WM, WN, (MTW * NTW > 4 ? 3 : DS_SPLIT_DENSE_SLOTS)> R;
};
template <int MTW, int NTW, int WM, int WN>
__global__ __launch_bounds__(256, (DenseRing<MTW, NTW, WM, WN>::R::LDS_BYTES > 80 * 1024) ? 1 : 2) void dense_split_kernel(const SplitDense d)
{
    typedef typename DenseRing<MTW, NTW, WM, WN>::R R;
    extern __shared__ __attribute__((aligned(16))) float ring[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int mi = wave % WM, nj = wave / WM;
    const int mblocks = (d.mtiles + R::FRA - 1) / R::FRA;
    const int nblocks = (d.ntiles + R::FRB - 1) / R::FRB;
    const int tiles = mblocks * nblocks, total = tiles * d.splits;      // logical workgroup = (K range, n-block, m-block), m fastest
    // XCD-aware order (workgroup b runs on XCD b % 8): every XCD takes a contiguous run of logical tiles; the m-blocks of one weight
    // panel are neighbours, so a panel streams into one L2 once
    // (workgroup b runs on XCD b % 8; XCD x takes total / 8 tiles, the first total % 8 XCDs one more; PMC: with the m-blocks of a panel on
    // different XCDs the 128 x 96 launch fetched 913 MB from the memory side for 218 MB of weights)
    const int bx = blockIdx.x & 7, bq = blockIdx.x >> 3, per = total >> 3, rem8 = total & 7;
    const int b = bx * per + min(bx, rem8) + bq;
    const int ks = b / tiles, bt = b - ks * tiles;
    const int nb = bt / mblocks, mb = bt - nb * mblocks;
    const int k0 = (int)((long)ks * d.ksteps / d.splits), k1 = (int)((long)(ks + 1) * d.ksteps / d.splits);
    const int half = lane >> 5, r31 = lane & 31;
    const unsigned lane4 = (unsigned)lane * 4;
    const int nstages = (k1 - k0) / R::KGS;         // (ksteps is even wherever KGS = 2 is instantiated: checked by the launcher)
    R rg;
    rg.init(ring, wave, lane, d.A + (size_t)k0 * SPLIT_KSTEP_BYTES, d.A + (size_t)k0 * SPLIT_KSTEP_BYTES, nstages, (long)d.ksteps * SPLIT_KSTEP_BYTES, mb * R::FRA,
            d.mtiles, d.Bp + (size_t)k0 * SPLIT_KSTEP_BYTES, d.kg_stride, min(nb * R::FRB, d.ntiles_alloc - R::FRB));
    float* const Cp = d.C + (size_t)ks * d.part_stride;
    rg.prologue(nstages);
    floatx16 acc[MTW][NTW];
#pragma unroll
    for (int i = 0; i < MTW; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
    rg.run(nstages, ring + (mi * MTW * 3) * 256 + lane4, ring + (3 * R::FRA + nj * NTW * 3) * 256 + lane4, acc);
#if DS_RING_CLOCK
    if ((blockIdx.x == 0 || blockIdx.x == 101) && lane == 0 && (wave == 0 || wave == 3))
        printf("RING block %d wave %d: per k-step (100 MHz ticks x 1000) wait %d barrier %d request %d rest %d\n", (int)blockIdx.x, wave,
               (int)(rg.clk[0] * 1000 / nstages), (int)(rg.clk[1] * 1000 / nstages), (int)(rg.clk[2] * 1000 / nstages), (int)(rg.clk[3] * 1000 / nstages));
#endif
    const int nt_base = min(nb * R::FRB, d.ntiles_alloc - R::FRB);
#pragma unroll
    for (int i = 0; i < MTW; ++i) {
        const int mtile = mb * R::FRA + mi * MTW + i;
        const int row = mtile * 32 + r31;
        if (mtile >= d.mtiles || row >= d.n) continue;
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const int ntile = nt_base + nj * NTW + j;
            if (ntile < nb * R::FRB) continue;             // (a clamped last block recomputes columns its neighbour owns: not stored twice)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = ntile * 32 + 8 * g + 4 * half;
                if (col < d.N) {
                    const v4f o = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                    *(__attribute__((address_space(1))) v4f*)(Cp + (size_t)row * d.N + col) = o;
                }
            }
        }
    }
}

hipError_t launch_dense_split(const SplitDense& d, hipStream_t s)
{
    if (d.n <= 0) return hipSuccess;
    if (d.ksteps <= 0 || (d.N & 3) || d.ntiles_alloc < 4) return hipErrorInvalidValue;
    const long total = (long)d.mtiles * 32 * d.ksteps * 2;
    hipLaunchKernelGGL(pack_joint_split_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, d);
    if (d.splits < 1 || d.splits > d.ksteps || (!d.wide && d.splits != 1)) return hipErrorInvalidValue;
    if (d.wide) {
        // 256 x 192 (waves 2 x 2, each 128 x 96), K in d.splits ranges: half the operand bytes per MFMA of the 128 x 96 tile; the partial
        // products go to d.C + range * d.part_stride and their reader adds them up (launch_head)
        typedef DenseRing<4, 3, 2, 2>::R R;
        if (d.ntiles_alloc < R::FRB) return hipErrorInvalidValue;
        const int mblocks = (d.mtiles + R::FRA - 1) / R::FRA, nblocks = (d.ntiles + R::FRB - 1) / R::FRB;
        hipLaunchKernelGGL((dense_split_kernel<4, 3, 2, 2>), dim3(mblocks * nblocks * d.splits), dim3(256), R::LDS_BYTES, s, d);
    } else {
        // 128 x 96 (waves 4 x 1): 252 workgroups at 512 sites (mid-round's tile; narrow output widths)
        typedef DenseRing<1, 3, 4, 1>::R R;
        const int mblocks = (d.mtiles + R::FRA - 1) / R::FRA, nblocks = (d.ntiles + R::FRB - 1) / R::FRB;
        hipLaunchKernelGGL((dense_split_kernel<1, 3, 4, 1>), dim3(mblocks * nblocks), dim3(256), R::LDS_BYTES, s, d);
    }
    return hipGetLastError();
}

static bool split_chain_ok(const FusedChain& c)
{
    if (c.nmod <= 0 || c.nmod > FUSED_CHAIN_MAX) return false;
    for (int i = 0; i < c.nmod; ++i) {
        if (i > 0 && c.m[i].pool_win != 0) return false;                     // only a chain's first module may pool its input
        if (c.m[i].cin != 240 && c.m[i].cin != 256) return false;            // P1 is unrolled over 15 or 16 chunks (every module of the model)
        if (i > 0 && (c.m[i].W != c.m[0].W || c.m[i].spt != c.m[0].spt || c.m[i].n_sites != c.m[0].n_sites || c.m[i].X != c.m[i - 1].Y))
            return false;
    }
    return true;
}

hipError_t configure_split_kernels()
{
    const void* fns[10] = {(const void*)dense_split_kernel<4, 3, 2, 2>, (const void*)stem23_split_kernel, (const void*)inception_fused_split_kernel<1>, (const void*)inception_fused_split_kernel<2>,
                          (const void*)inception_fused_split_kernel<3>, (const void*)lstm_cell_split_kernel<1, 1>,
                          (const void*)lstm_cell_split_kernel<1, 2>, (const void*)lstm_cell_split_kernel<2, 2>, (const void*)lstm_cell_split_kernel<1, 2, 4, 2>,
                          (const void*)dense_split_kernel<1, 3, 4, 1>};
    for (const void* f : fns) {
        const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t launch_inception_fused_split(int tm, const FusedChain& c, hipStream_t s)
{
    if (!split_chain_ok(c)) return hipErrorInvalidValue;
    const FusedArgs& a = c.m[0];
    if (a.n_sites <= 0) return hipSuccess;
    const size_t lds = inception_fused_split_lds_bytes(tm, a.W, a.spt);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    const int grid = (a.n_sites + a.spt - 1) / a.spt;
    switch (tm) {
    case 1: hipLaunchKernelGGL(inception_fused_split_kernel<1>, dim3(grid), dim3(512), lds, s, c); break;
    case 2: hipLaunchKernelGGL(inception_fused_split_kernel<2>, dim3(grid), dim3(512), lds, s, c); break;
    case 3: hipLaunchKernelGGL(inception_fused_split_kernel<3>, dim3(grid), dim3(512), lds, s, c); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace ds
