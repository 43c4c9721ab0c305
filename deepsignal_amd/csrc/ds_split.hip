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
// Fused inception modules, split operands. Phases, wave roles and the static wave -> unit table are those of
// inception_fused_kernel (ds_kernels.hip): one workgroup (8 waves) owns a tile of whole sites (<= 96 rows) and takes it through
// the modules of one width class;
//   P1  [rows x cin] x [cin x 256]: wave w owns n-tile w for all m-tiles, K in chunks of 16 channels = ONE bf16 k-step.
//       The stagers (waves 0..5) load the chunk's fp32 rows AND their two neighbour rows from global memory, take the 3-tap
//       max (branch 1's maxpool(3, 1, SAME), padded taps ignored = the own row), split both the plain and the pooled values
//       into terms and write ONE staged row [plain t0 | t1 | t2 | pooled t0 | t1 | t2] x 32 B: every wave's fragment reads
//       are three ds_read_b128 at immediate offsets from one address (waves 6, 7: the pooled half), no wave pools or splits
//       inside its MFMA stream. Per chunk and wave: 3 weight fragments (one per term, global -> VGPR), 3 x TM activation
//       fragments, 6 x TM MFMAs.
//   P2a / P2b: the 1x3 / 1x5 convs from the 32-channel intermediates, which the P1 epilogue wrote to LDS as terms (T1, zero
//       halo rows = SAME padding); branch 5's 64-channel intermediate likewise (T2); its last 1x1 accumulates on top of the
//       stem accumulators.
// LDS (bytes): staged chunks 2 x rows x 208 (later the fp32 b1|b2 output tile) | T2 rows x 400 | T1 (spt (W + 4) + 5) x 592:
// 143 - 147 KB for 96-row tiles, one workgroup per CU. Every row stride is an odd number of 16-byte slots.
// Roofline: the bf16 matrix pipe at six products per MAC.
constexpr int S_LDP = 208;      // staged chunk row: 6 x 32 B + 16
constexpr int S_LD1 = 592;      // T1 row: 3 terms x 96 channels x 2 B + 16
constexpr int S_LD2 = 400;      // T2 row: 3 terms x 64 channels x 2 B + 16
constexpr int S_LDY = 400;      // b1|b2 output tile row: 96 floats + 4
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

template <int NTAPS>
__device__ __forceinline__ void fuseds_conv_unit(const char* T1, int rm, int cbyte, int lane, const float4 (&ub)[30], floatx16& acc)
{
    const char* base = T1 + rm * S_LD1 + cbyte + (lane >> 5) * 16;
#pragma unroll
    for (int t = 0; t < NTAPS; ++t) {
        const char* arow = base + (t - NTAPS / 2) * S_LD1;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float4 a0 = *reinterpret_cast<const float4*>(arow + j * 32);
            const float4 a1 = *reinterpret_cast<const float4*>(arow + S_T1P + j * 32);
            const float4 a2 = *reinterpret_cast<const float4*>(arow + 2 * S_T1P + j * 32);
            const int q = (t * 2 + j) * 3;
            const float4 w[3] = {ub[q], ub[q + 1], ub[q + 2]};
            acc = mfma3_lo(w, a0, a1, a2, acc);
            acc = mfma3_hi(w, a0, a1, acc);
        }
    }
}

#ifndef DS_SPLIT_VD
#define DS_SPLIT_VD 2       // register stages of the stagers' row loads (chunk c + VD is requested at step c)
#endif

template <int TM>
__global__ __launch_bounds__(512, 2) void inception_fused_split_kernel(const FusedChain c)
{
    constexpr int TR32 = TM * 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* const sm = reinterpret_cast<char*>(smem);
    char* const Pst = sm;                                // [2][TR32][S_LDP] staged chunks
    char* const Ys = sm;                                 // [TR32][S_LDY] b1|b2 output tile, aliases the staged chunks once P1 is done
    char* const T2 = sm + 2 * TR32 * S_LDP;              // [TR32][S_LD2]
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
    const bool stamp = a.dbg != nullptr && lane == 0 && (wave == 0 || wave == 7) && blockIdx.x < DBG_MAX_WGS;
#define DS_STAMP(i) do { if (stamp) (a.dbg + ((size_t)blockIdx.x * 2 + (wave == 7)) * 8)[i] = __builtin_amdgcn_s_memtime(); } while (0)
    DS_STAMP(0);
    if (tid >= 256 && tid < 448) {
        const int q = tid - 256;
        Bs[q] = gload((q < 64 ? a.bias5b : q < 128 ? a.bias3b : a.bias4b) + (q & 63));
    }

    // ---- P1 staging cursor: thread -> (row, 16-byte slot of the chunk's 64 B); rows past the tile end re-read row TRv-1
    const bool stager = tid < TR32 * 4;
    const int sr = tid >> 2, sq = tid & 3;
    const int rr = sr < TRv ? sr : TRv - 1;
    const int wr = rr % W;
    const float* pc = a.X + (grow0 + rr) * cin + sq * 4;
    const int om = wr > 0 ? -cin : 0;             // previous / next row of the same site, or the own row at a site edge
    const int op = wr < W - 1 ? cin : 0;
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
    {
        constexpr int VD = DS_SPLIT_VD;
        float4 vo[VD], vm[VD], vn[VD];      // own / previous / next row, 16 B each
        float4 bq[2][3];
        float4 af[2][3][TM];
        // waves 6, 7 (branch 1) read the pooled half of a staged row
        const char* const fsrc = Pst + rlane * S_LDP + (lane >> 5) * 16 + (wave >= 6 ? 96 : 0);
        char* const sdst = Pst + sr * S_LDP + sq * 8;
        auto load_a = [&](int V) __attribute__((always_inline)) {
            if (stager) {
                vo[V] = gload4(pc); vm[V] = gload4(pc + om); vn[V] = gload4(pc + op);
                pc += KC;
            }
        };
        auto store_a = [&](int X, int V) __attribute__((always_inline)) {
            if (stager) {
                char* d = sdst + X * TR32 * S_LDP;
                uint2 t0, t1, t2;
                split3x4(vo[V].x, vo[V].y, vo[V].z, vo[V].w, t0, t1, t2);
                *reinterpret_cast<uint2*>(d) = t0; *reinterpret_cast<uint2*>(d + 32) = t1; *reinterpret_cast<uint2*>(d + 64) = t2;
                split3x4(fmaxf(fmaxf(vo[V].x, vm[V].x), vn[V].x), fmaxf(fmaxf(vo[V].y, vm[V].y), vn[V].y),
                         fmaxf(fmaxf(vo[V].z, vm[V].z), vn[V].z), fmaxf(fmaxf(vo[V].w, vm[V].w), vn[V].w), t0, t1, t2);
                *reinterpret_cast<uint2*>(d + 96) = t0; *reinterpret_cast<uint2*>(d + 128) = t1; *reinterpret_cast<uint2*>(d + 160) = t2;
            }
        };
        auto load_b = [&](int Bi) __attribute__((always_inline)) {
            bq[Bi][0] = gload4(bp); bq[Bi][1] = gload4(bp + 256); bq[Bi][2] = gload4(bp + 512);
            bp += 768;
        };
        auto read_frags = [&](int X) __attribute__((always_inline)) {
#pragma unroll
            for (int mt = 0; mt < TM; ++mt) {
                const char* q = fsrc + X * TR32 * S_LDP + mt * 32 * S_LDP;
#pragma unroll
                for (int p = 0; p < 3; ++p) af[X][p][mt] = *reinterpret_cast<const float4*>(q + 32 * p);
            }
        };
#pragma unroll
        for (int i = 0; i < VD; ++i) load_a(i);                  // chunks 0 .. VD - 1
        load_b(0);
        store_a(0, 0);
        lds_barrier();
        read_frags(0);
#pragma unroll
        for (int cc = 0; cc < 16; ++cc) {
            if (cc < nchunks) {                                   // wave-uniform; only cc = 15 is really conditional
                const int X = cc & 1;
                const bool has1 = cc + 1 < nchunks;
                if (cc + VD < nchunks) load_a(cc % VD);           // chunk cc + VD into the stage chunk cc left (consumed at step cc - 1)
                if (has1) load_b(X ^ 1);
                if (has1) store_a(X ^ 1, (cc + 1) % VD);
#pragma unroll
                for (int mt = 0; mt < TM; ++mt) acc[mt] = mfma3_lo(bq[X], af[X][0][mt], af[X][1][mt], af[X][2][mt], acc[mt]);
                __builtin_amdgcn_sched_barrier(0);
                lds_barrier();
                if (has1) read_frags(X ^ 1);
#pragma unroll
                for (int mt = 0; mt < TM; ++mt) acc[mt] = mfma3_hi(bq[X], af[X][0][mt], af[X][1][mt], acc[mt]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    DS_STAMP(1);
    lds_barrier();   // all fragment reads of the staging area are done before the output tile aliases it
    DS_STAMP(2);

    // ---- static wave -> unit assignment of P2 (wave-uniform), as inception_fused_kernel. kind: 0 none, 1 b5b, 2 b3b, 3 b4b.
    int a1k = 0, a1m = 0, a1n = 0, a2k = 0, a2m = 0, a2n = 0;      // P2a units
    int b1k = 0, b1m = 0, b1n = 0, b2k = 0, b2m = 0, b2n = 0;      // P2b units (waves 0,1 run the residual tail first)
    if (TM == 3) {
        if (wave < 6) { a1k = 1; a1m = wave % 3; a1n = wave / 3; }
        else { a1k = 2; a1m = 0; a1n = wave - 6; }
        if (wave >= 2) { b1k = 3; b1m = (wave - 2) % 3; b1n = (wave - 2) / 3; }
        if (wave >= 4) { b2k = 2; b2m = 1 + ((wave - 4) >> 1); b2n = (wave - 4) & 1; }
    } else if (TM == 2) {
        if (wave < 4) { a1k = 1; a1m = wave & 1; a1n = wave >> 1; }
        else { a1k = 2; a1m = wave & 1; a1n = (wave - 4) >> 1; }
        if (wave >= 2 && wave < 6) { b1k = 3; b1m = (wave - 2) & 1; b1n = (wave - 2) >> 1; }
    } else {
        if (wave < 2) { a1k = 1; a1n = wave; }
        else if (wave < 4) { a1k = 2; a1n = wave - 2; }
        else if (wave < 6) { a1k = 3; a1n = wave - 4; }
    }
    auto unit_Bp = [&](int k) { return k == 1 ? a.Bp5b : k == 2 ? a.Bp3b : a.Bp4b; };
    auto unit_taps = [&](int k) { return k == 3 ? 5 : 3; };
    float4 pf[30];                                   // prefetched weights of the next unit
    // (defined on every path: a register array that is only loaded under a wave-uniform condition is "undefined on some
    // paths", and hipcc keeps such a value live around the whole module loop)
#pragma unroll
    for (int g = 0; g < 30; ++g) pf[g] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a1k) fuseds_unit_prefetch(unit_Bp(a1k), unit_taps(a1k), a1n, lane, pf);

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
    for (int idx = tid; idx < TR32 * 24; idx += 512) {
        const int row = idx / 24, q = idx - row * 24;
        if (row < TRv) {
            const float4 v = *reinterpret_cast<const float4*>(Ys + row * S_LDY + q * 16);
            v4f o = {v.x, v.y, v.z, v.w};
            *(__attribute__((address_space(1))) v4f*)(Yg + (unsigned)(row * 240 + q * 4)) = o;
        }
    }

    auto run_unit = [&](int kind, int mt, int nt) __attribute__((always_inline)) {
        floatx16 u;
        const float* bsrc = Bs + (kind == 1 ? 0 : kind == 2 ? 64 : 128) + nt * 32 + h4;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 t = *reinterpret_cast<const float4*>(bsrc + 8 * g);
            u[4 * g] = t.x; u[4 * g + 1] = t.y; u[4 * g + 2] = t.z; u[4 * g + 3] = t.w;
        }
        const int row = mt * 32 + rlane;
        const int rm = rowmap[row];
        if (kind == 1) {          // 1x3, 32 -> 64, ReLU, to T2 as terms                  layers.py:127-131
            fuseds_conv_unit<3>(T1, rm, 128, lane, pf, u);
            char* d = T2 + row * S_LD2 + (nt * 32 + h4) * 2;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint2 t0, t1, t2;
                split3x4(relu_f(u[4 * g]), relu_f(u[4 * g + 1]), relu_f(u[4 * g + 2]), relu_f(u[4 * g + 3]), t0, t1, t2);
                *reinterpret_cast<uint2*>(d + 16 * g) = t0;
                *reinterpret_cast<uint2*>(d + 16 * g + S_T2P) = t1;
                *reinterpret_cast<uint2*>(d + 16 * g + 2 * S_T2P) = t2;
            }
        } else {
            // kind 2: 1x3, 32 -> 48, ReLU, to Y[96,144)   layers.py:106-110
            // kind 3: 1x5, 32 -> 48, ReLU, to Y[144,192)  layers.py:115-119
            if (kind == 2) fuseds_conv_unit<3>(T1, rm, 0, lane, pf, u);
            else fuseds_conv_unit<5>(T1, rm, 64, lane, pf, u);
            const int ybase = kind == 2 ? 96 : 144;
            if (row < TRv) {
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    if (nt * 32 + 8 * g < 48) {      // wave-uniform: 48 output channels = n-tile 0 and half of n-tile 1
                        v4f o = {relu_f(u[4 * g]), relu_f(u[4 * g + 1]), relu_f(u[4 * g + 2]), relu_f(u[4 * g + 3])};
                        *(__attribute__((address_space(1))) v4f*)(Yg + (unsigned)(row * 240 + ybase + nt * 32 + 8 * g + h4)) = o;
                    }
            }
        }
    };

    // ---- P2a
    if (a1k) run_unit(a1k, a1m, a1n);
    if (a2k) {
        fuseds_unit_prefetch(unit_Bp(a2k), unit_taps(a2k), a2n, lane, pf);
        run_unit(a2k, a2m, a2n);
    }
    // weights of the first P2b job are requested before the barrier (the tail's twelve fragments travel in the unit registers)
    if (wave < 2) {
#pragma unroll
        for (int g = 0; g < 12; ++g) pf[g] = gload4(a.Bp5c + ((size_t)(wave * 12 + g) * 64 + lane) * 4);
    } else if (b1k) {
        fuseds_unit_prefetch(unit_Bp(b1k), unit_taps(b1k), b1n, lane, pf);
    }
    DS_STAMP(5);
    lds_barrier();   // T2 complete
    DS_STAMP(6);

    // ---- P2b
    if (wave < 2) {
        // branch 5 tail: 1x1 64 -> 48 (BN, no ReLU) accumulated on top of the stem conv held in acc,
        // then relu(stem + tail)                                                         layers.py:132-138
#pragma unroll
        for (int mt = 0; mt < TM; ++mt) {
            const char* base = T2 + (mt * 32 + rlane) * S_LD2 + (lane >> 5) * 16;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 a0 = *reinterpret_cast<const float4*>(base + g * 32);
                const float4 a1 = *reinterpret_cast<const float4*>(base + S_T2P + g * 32);
                const float4 a2 = *reinterpret_cast<const float4*>(base + 2 * S_T2P + g * 32);
                const float4 w[3] = {pf[3 * g], pf[3 * g + 1], pf[3 * g + 2]};
                acc[mt] = mfma3_lo(w, a0, a1, a2, acc[mt]);
                acc[mt] = mfma3_hi(w, a0, a1, acc[mt]);
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
    } else {
        if (b1k) run_unit(b1k, b1m, b1n);
        if (b2k) {
            fuseds_unit_prefetch(unit_Bp(b2k), unit_taps(b2k), b2n, lane, pf);
            run_unit(b2k, b2m, b2n);
        }
    }
    DS_STAMP(7);
#undef DS_STAMP
    // the next module reads the rows this one has stored (other waves' stores included: __syncthreads drains vmcnt) and
    // re-uses every LDS region
    if (mi + 1 < c.nmod) __syncthreads();
    }   // modules of the chain
}

static bool split_chain_ok(const FusedChain& c)
{
    if (c.nmod <= 0 || c.nmod > FUSED_CHAIN_MAX) return false;
    for (int i = 0; i < c.nmod; ++i) {
        if (c.m[i].pool_win != 0) return false;                              // the stride-2 pools run as their own launches in this mode
        if (c.m[i].cin != 240 && c.m[i].cin != 256) return false;            // P1 is unrolled over 15 or 16 chunks (every module of the model)
        if (i > 0 && (c.m[i].W != c.m[0].W || c.m[i].spt != c.m[0].spt || c.m[i].n_sites != c.m[0].n_sites || c.m[i].X != c.m[i - 1].Y))
            return false;
    }
    return true;
}

hipError_t configure_split_kernels()
{
    const void* fns[3] = {(const void*)inception_fused_split_kernel<1>, (const void*)inception_fused_split_kernel<2>,
                          (const void*)inception_fused_split_kernel<3>};
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
