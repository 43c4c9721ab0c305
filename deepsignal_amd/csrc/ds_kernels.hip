// ds_kernels.hip — gfx950 (CDNA4 / MI355X) kernels of the call_mods forward pass.
//
// Kernels, by role (DESIGN.md section 4 has the table): a whole-module fused inception kernel (fp32 and bf16 operand
// forms), a conv_layer2 + conv_layer3 kernel, dedicated BiLSTM cell kernels of one anti-diagonal per launch (operands
// direct to registers, or shared through an LDS-DMA ring), an implicit-GEMM template (the 6032 x 6032 dense layer of
// the three-step joint model, the bf16-mode stem convolutions, the layer-granular diagnostic path), and small kernels
// for the Cin = 1 stem conv, the pools, the folded / two-class head and the input gather.
//
// Design notes (MI355X_MICROARCH.md / cdna_hip_programming.md):
//  * v_mfma_f32_32x32x2_f32 is exact fp32 (an fmaf chain) at 64 FLOP/clk/SIMD and needs only ONE
//    VGPR per operand per lane, so the weight (B) operand is pre-packed on the host in fragment
//    order and streamed global->VGPR with 1 KiB coalesced wave loads (no LDS round trip), while the
//    activation (A) tile is staged through LDS once per K-chunk and shared by all waves.
//  * K is consumed in groups of 8 with a lane-local permutation (lane half h owns k = 4h..4h+3 of
//    the group) so one ds_read_b128 / one global_load_dwordx4 feeds four consecutive MFMAs.
//  * LDS rows are padded to KC+4 floats: the 16-lane ds_read_b128 groups hit 16 distinct 16-B slots.
//  * wave64 everywhere; no CUDA-isms, no compatibility shims.
#include "ds_device.h"

namespace ds {


// BF = bf16 operands. The byte geometry is the same as fp32: a "unit" is 4 bytes (one float or two bf16), a
// K chunk is 16 units = 64 B per row (16 floats / 32 bf16), one 16-B fragment per lane feeds four
// v_mfma_f32_32x32x2_f32 (fp32) or ONE v_mfma_f32_32x32x16_bf16 (lane half h owns k = 8h..8h+7 of the 16).
// The host passes ld / klen / K of bf16 operands in units, so the staging code is shared.
// KS = in-workgroup K split: KS wave groups ("K-lanes") own the same output tile and take alternate
// 16-wide K chunks (lane g: chunks g, g+KS, ...), each with its own LDS staging area, and the partial
// accumulators are exchanged through LDS at the end. It doubles the waves per SIMD for grids that
// only have ~one workgroup per CU and halves the serial chunk chain of short-K problems.
template <int MT, int NT, int WM, int WN, int EPI, int AMODE, int BD, int KS, bool BF = false>
__global__ __launch_bounds__(64 * WM * WN * KS, 1) void gemm_kernel(const GemmLaunch* __restrict__ L)
{
    constexpr int BM = WM * MT * 32;
    constexpr int LDA = KC + 4;
    constexpr int NTHR = 64 * WM * WN;          // threads per K-lane
    constexpr int SLOTS = (BM * 4 + NTHR - 1) / NTHR;
    constexpr int AS_FLOATS = 2 * KS * BM * LDA;
    constexpr int RED_FLOATS = KS > 1 ? WM * WN * MT * NT * 16 * 64 : 0;
    __shared__ __attribute__((aligned(16))) float smem_[AS_FLOATS > RED_FLOATS ? AS_FLOATS : RED_FLOATS];

    // the wave index is wave-uniform, but hipcc cannot prove it from threadIdx: readfirstlane keeps the
    // K-lane / segment-cursor logic on the scalar unit (otherwise it becomes exec-masked vector code)
    const int lane = threadIdx.x & 63, wave_all = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kl = wave_all / (WM * WN);        // K-lane of this wave
    const int wave = wave_all % (WM * WN);
    const int tid = threadIdx.x - kl * NTHR;    // thread index inside the K-lane
    float* const As0 = smem_ + (0 * KS + kl) * BM * LDA;
    float* const As1 = smem_ + (1 * KS + kl) * BM * LDA;
    const int wm = wave % WM, wn = wave / WM;
    // XCD-aware tile order: hardware deals workgroup b to XCD b % 8, each with a private L2. Give every XCD a
    // contiguous run of logical tiles (m-tiles of one weight panel are neighbours), so a weight panel is pulled
    // into ONE L2 instead of up to eight (PMC: FC1 fetched its 146 MB of weights 4x before this remap).
    int bid;
    {
        const int total = L->total_tiles, q = total >> 3, r = total & 7;
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        bid = xcd * q + (xcd < r ? xcd : r) + idx;
    }
    int pi = 0;
    for (int i = 1; i < L->nprob; ++i)
        if (bid >= L->prob[i].tile_start) pi = i;
    const GemmProblem& P = L->prob[pi];
    const int local = bid - P.tile_start;
    // neighbours in the logical order share the bigger operand: the weight panel (m fastest) for FC / LSTM
    // shapes, the activation rows (n fastest) for the convolutions
    const int tm = P.n_fast ? local / P.tiles_n : local % P.tiles_m;
    const int tn = P.n_fast ? local % P.tiles_n : local / P.tiles_m;
    const int m0 = tm * BM;
    const int M = P.M, W = P.W;
    const int nchunks = P.K / KC / KS;          // chunks per K-lane (planner pads K to a multiple of KC*KS)
    const int nseg = P.nseg;

    // per-slot row bookkeeping (a slot = one float4 of the staged A chunk)
    int sm[SLOTS], sw[SLOTS];
    bool sin_[SLOTS];
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
        const int idx = tid + i * NTHR;
        sm[i] = m0 + (idx >> 2);
        sin_[i] = (idx < BM * 4) && (sm[i] < M);
        sw[i] = sm[i] % W;
    }

    int ntile[NT];
    bool nvalid[NT];
    const float* bp[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        ntile[nt] = (tn * WN + wn) * NT + nt;
        nvalid[nt] = ntile[nt] < P.ntiles32;
        const int tcl = nvalid[nt] ? ntile[nt] : P.ntiles32 - 1;   // clamp: loads stay in bounds, result unused
        bp[nt] = P.Bp + ((size_t)tcl * P.kgroups_stride + kl * 2) * 256 + lane * 4;
    }

    floatx16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.0f;

    // A cursor: branch-free loads. Every slot always loads from a legal address (its own row when
    // valid, the segment base otherwise) and invalid slots are zeroed by a select afterwards.
    const float* ap[SLOTS];
    const float* apm[SLOTS];   // AMODE 1: previous / next row (clamped to the own row at site edges)
    const float* app[SLOTS];
    bool aok[SLOTS];
    int seg_i = 0, seg_left = 0;
    auto seg_begin = [&](int si) {
        const float* base = P.seg[si].base;
        const int ld = P.seg[si].ld;
        const int shift = P.seg[si].row_shift;
        seg_left = P.seg[si].klen / KC;
#pragma unroll
        for (int i = 0; i < SLOTS; ++i) {
            const int idx = tid + i * NTHR;
            const int ws = sw[i] + shift;
            aok[i] = sin_[i] && (unsigned)ws < (unsigned)W;
            const float* q = base + (size_t)(sm[i] + shift) * ld + (idx & 3) * 4;
            ap[i] = aok[i] ? q : base;
            if (AMODE == 1) {
                apm[i] = (aok[i] && sw[i] > 0) ? q - ld : ap[i];
                app[i] = (aok[i] && sw[i] < W - 1) ? q + ld : ap[i];
            }
        }
    };
    auto advance = [&]() {
        if (--seg_left == 0 && ++seg_i < nseg) seg_begin(seg_i);
    };
    auto skip_chunk = [&]() {     // step the cursor over a chunk that belongs to another K-lane
#pragma unroll
        for (int i = 0; i < SLOTS; ++i) {
            ap[i] += KC;
            if (AMODE == 1) { apm[i] += KC; app[i] += KC; }
        }
        advance();
    };
    // Software pipeline (per chunk c, parity X = c&1; chunk k lives in R[k&1] / b[k&1] / As[k&1] / a[k&1]):
    //   (1) issue global loads: A rows of chunk c+2 -> R[X], weights of chunk c+BD -> b[(c+BD) % (BD+1)]
    //       (BD = weight prefetch distance in chunks: 1 for L2-resident weights, 2 when they stream from HBM)
    //   (2) write chunk c+1 (loaded one step ago) to LDS As[X^1] (SAME-padding select / 3-tap max applied here)
    //   (3) MFMAs of read-step 0 of chunk c
    //   (4) barrier                       -- everyone's (2) is done
    //   (5) ds_read chunk c+1 fragments -> a[X^1]   (latency hidden by (6))
    //   (6) MFMAs of read-step 1 of chunk c
    // One barrier per chunk, no load / LDS latency on the MFMA critical path.
    // BD == 3 ("deep"): activations AND weights are requested two full chunks ahead through rings of three register
    // buffers (chunk k in R[k % 3] / bq[k % 3]; four would not fit 256 VGPRs next to a 1x4 accumulator tile). For
    // bf16 operands a chunk is only ~256 matrix-pipe cycles, so one step of look-ahead cannot cover an L2 / HBM
    // round trip.
    constexpr int RA = BD == 3 ? 3 : 2;
    float4 R[RA][SLOTS], Rm[RA][SLOTS], Rp[RA][SLOTS];
    bool lok[RA][SLOTS];
    float4 bq[BD == 3 ? 3 : 2 * BD][NT][2];   // weight ring: 2 buffers (BD=1), 4 (BD=2) or 3 (BD=3), statically indexed
    float4 af[2][2][MT];
    auto load_a = [&](int X) {
#pragma unroll
        for (int i = 0; i < SLOTS; ++i) {
            R[X][i] = gload4(ap[i]);
            ap[i] += KC;
            if (AMODE == 1) {
                Rm[X][i] = gload4(apm[i]);
                Rp[X][i] = gload4(app[i]);
                apm[i] += KC;
                app[i] += KC;
            }
            if (AMODE != 2) lok[X][i] = aok[i];
        }
        advance();
#pragma unroll
        for (int j = 1; j < KS; ++j) skip_chunk();
    };
    auto store_a = [&](int X, int P = -1) {      // ring slot X -> LDS buffer P (default: the slot's own parity)
        float* dst = (P < 0 ? X : P) ? As1 : As0;
#pragma unroll
        for (int i = 0; i < SLOTS; ++i) {
            const int idx = tid + i * NTHR;
            float4 v = R[X][i];
            if (AMODE == 1) v = BF ? bf8max_nn(bf8max_nn(v, Rm[X][i]), Rp[X][i])
                                   : f4max(f4max(v, Rm[X][i]), Rp[X][i]);   // maxpool(3, s1, SAME): padded taps ignored
            if (AMODE != 2) {     // AMODE 2 = dense: every row of every tile is valid, no taps -> no select
                v.x = lok[X][i] ? v.x : 0.0f;
                v.y = lok[X][i] ? v.y : 0.0f;
                v.z = lok[X][i] ? v.z : 0.0f;
                v.w = lok[X][i] ? v.w : 0.0f;
            }
            if (SLOTS * NTHR == BM * 4 || idx < BM * 4)
                *reinterpret_cast<float4*>(dst + (idx >> 2) * LDA + (idx & 3) * 4) = v;
        }
    };
    auto load_b = [&](int X) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            bq[X][nt][0] = gload4(bp[nt]);
            bq[X][nt][1] = gload4(bp[nt] + 256);
            bp[nt] += 512 * KS;
        }
    };
    auto read_frags = [&](int X) {
#pragma unroll
        for (int rs = 0; rs < 2; ++rs)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                af[X][rs][mt] = *reinterpret_cast<const float4*>(
                    (X ? As1 : As0) + ((wm * MT + mt) * 32 + (lane & 31)) * LDA + rs * 8 + (lane >> 5) * 4);
    };
    // X / JC / JN are literals at every call site, so all register arrays are statically indexed
    auto mfma_rs2 = [&](int X, int J, int rs) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                if (BF) {
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[X][rs][mt]),
                                                                          __builtin_bit_cast(bf16x8, bq[J][nt][rs]), acc[mt][nt], 0, 0, 0);
                    continue;
                }
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[X][rs][mt].x, bq[J][nt][rs].x, acc[mt][nt], 0, 0, 0);
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[X][rs][mt].y, bq[J][nt][rs].y, acc[mt][nt], 0, 0, 0);
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[X][rs][mt].z, bq[J][nt][rs].z, acc[mt][nt], 0, 0, 0);
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[X][rs][mt].w, bq[J][nt][rs].w, acc[mt][nt], 0, 0, 0);
            }
    };
    // Instruction interleave inside a step (sched_group_barrier, cdna_hip_programming.md T19): with one
    // wave per SIMD a burst of loads / LDS ops in front of the MFMA block leaves the matrix pipe idle
    // while they issue, so each non-MFMA instruction is slotted behind one MFMA instead.
    //   masks: VALU 0x2, MFMA 0x8, VMEM read 0x20, DS read 0x100, DS write 0x200
    constexpr int N_MFMA_HALF = MT * NT * (BF ? 1 : 4);
    constexpr int N_ALOADS = SLOTS * (AMODE == 1 ? 3 : 1);
    constexpr int N_BLOADS = 2 * NT;
    constexpr int N_VALU_PER_STORE = (AMODE == 1 ? 12 : AMODE == 2 ? 0 : 4);
    // deep variant: chunk c sits in ring slot Q = c % 3 and LDS buffer X = c & 1
#define DS_STEPD(X, Q, HAS1, HAS3)                                                        \
    do {                                                                                  \
        if (HAS3) { load_a(((Q) + 2) % 3); load_b(((Q) + 2) % 3); }                       \
        if (HAS1) store_a(((Q) + 1) % 3, (X) ^ 1);                                        \
        mfma_rs2(X, Q, 0);                                                                \
        if (HAS1) {                                                                       \
            _Pragma("unroll") for (int i_ = 0; i_ < SLOTS; ++i_) {                        \
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);                          \
                if (N_VALU_PER_STORE) __builtin_amdgcn_sched_group_barrier(0x2, N_VALU_PER_STORE, 0); \
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                        \
            }                                                                             \
        }                                                                                 \
        if (HAS3) {                                                                       \
            _Pragma("unroll") for (int i_ = 0; i_ < N_ALOADS + N_BLOADS; ++i_) {          \
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);                          \
                __builtin_amdgcn_sched_group_barrier(0x20, 1, 0);                         \
            }                                                                             \
        }                                                                                 \
        __builtin_amdgcn_sched_group_barrier(0x8, N_MFMA_HALF, 0);                        \
        __builtin_amdgcn_sched_barrier(0);                                                \
        __syncthreads();                                                                  \
        if (HAS1) read_frags((X) ^ 1);                                                    \
        mfma_rs2(X, Q, 1);                                                                \
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);                                  \
        if (HAS1) __builtin_amdgcn_sched_group_barrier(0x100, 2 * MT, 0);                 \
        __builtin_amdgcn_sched_group_barrier(0x8, N_MFMA_HALF, 0);                        \
        __builtin_amdgcn_sched_barrier(0);                                                \
    } while (0)
#define DS_STEP(X, JC, JN, HAS1, HAS2, HASB)                                              \
    do {                                                                                  \
        if (HAS2) load_a(X);                                                              \
        if (HASB) load_b(JN);                                                             \
        if (HAS1) store_a((X) ^ 1);                                                       \
        mfma_rs2(X, JC, 0);                                                               \
        if (HAS1) {                                                                       \
            _Pragma("unroll") for (int i_ = 0; i_ < SLOTS; ++i_) {                        \
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);                          \
                if (N_VALU_PER_STORE) __builtin_amdgcn_sched_group_barrier(0x2, N_VALU_PER_STORE, 0); \
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                        \
            }                                                                             \
        }                                                                                 \
        if (HAS2) {                                                                       \
            _Pragma("unroll") for (int i_ = 0; i_ < N_ALOADS; ++i_) {                     \
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);                          \
                __builtin_amdgcn_sched_group_barrier(0x20, 1, 0);                         \
            }                                                                             \
        }                                                                                 \
        if (HASB) {                                                                       \
            _Pragma("unroll") for (int i_ = 0; i_ < N_BLOADS; ++i_) {                     \
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);                          \
                __builtin_amdgcn_sched_group_barrier(0x20, 1, 0);                         \
            }                                                                             \
        }                                                                                 \
        __builtin_amdgcn_sched_group_barrier(0x8, N_MFMA_HALF, 0);                        \
        __builtin_amdgcn_sched_barrier(0);                                                \
        __syncthreads();                                                                  \
        if (HAS1) read_frags((X) ^ 1);                                                    \
        mfma_rs2(X, JC, 1);                                                               \
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);                                  \
        if (HAS1) __builtin_amdgcn_sched_group_barrier(0x100, 2 * MT, 0);                 \
        __builtin_amdgcn_sched_group_barrier(0x8, N_MFMA_HALF, 0);                        \
        __builtin_amdgcn_sched_barrier(0);                                                \
    } while (0)

    if (nchunks > 0) {
        seg_begin(0);
#pragma unroll
        for (int j = 0; j < KS - 1; ++j)
            if (j < kl) skip_chunk();
        load_a(0);
        load_b(0);
        store_a(0);
        if (nchunks > 1) load_a(1);
        if (BD >= 2 && nchunks > 1) load_b(1);
        __syncthreads();
        read_frags(0);
        int c = 0;
        if (BD == 3) {
            // six steps per trip: LDS parity has period 2, the register rings period 3
            while (c + 7 < nchunks) {      // chunks c .. c+5 with every look-ahead (up to c+7) in range
                DS_STEPD(0, 0, true, true);
                DS_STEPD(1, 1, true, true);
                DS_STEPD(0, 2, true, true);
                DS_STEPD(1, 0, true, true);
                DS_STEPD(0, 1, true, true);
                DS_STEPD(1, 2, true, true);
                c += 6;
            }
            // tail: 1..7 chunks left, runtime (wave-uniform) look-ahead flags
            DS_STEPD(0, 0, c + 1 < nchunks, c + 2 < nchunks);
            if (++c < nchunks) {
                DS_STEPD(1, 1, c + 1 < nchunks, c + 2 < nchunks);
                if (++c < nchunks) {
                    DS_STEPD(0, 2, c + 1 < nchunks, c + 2 < nchunks);
                    if (++c < nchunks) {
                        DS_STEPD(1, 0, c + 1 < nchunks, c + 2 < nchunks);
                        if (++c < nchunks) {
                            DS_STEPD(0, 1, c + 1 < nchunks, c + 2 < nchunks);
                            if (++c < nchunks) {
                                DS_STEPD(1, 2, c + 1 < nchunks, false);
                                if (++c < nchunks) DS_STEPD(0, 0, false, false);
                            }
                        }
                    }
                }
            }
        } else if (BD == 1) {
            while (c + 3 < nchunks) {      // steady state: every look-ahead exists, no conditionals
                DS_STEP(0, 0, 1, true, true, true);
                DS_STEP(1, 1, 0, true, true, true);
                c += 2;
            }
            const int left = nchunks - c;  // 1..3
            if (left == 3) {
                DS_STEP(0, 0, 1, true, true, true);
                DS_STEP(1, 1, 0, true, false, true);
                DS_STEP(0, 0, 1, false, false, false);
            } else if (left == 2) {
                DS_STEP(0, 0, 1, true, false, true);
                DS_STEP(1, 1, 0, false, false, false);
            } else {
                DS_STEP(0, 0, 1, false, false, false);
            }
        } else {
            while (c + 5 < nchunks) {
                DS_STEP(0, 0, 2, true, true, true);
                DS_STEP(1, 1, 3, true, true, true);
                DS_STEP(0, 2, 0, true, true, true);
                DS_STEP(1, 3, 1, true, true, true);
                c += 4;
            }
            // tail: 1..5 chunks, runtime (wave-uniform) look-ahead flags
            DS_STEP(0, 0, 2, c + 1 < nchunks, c + 2 < nchunks, c + 2 < nchunks);
            if (++c < nchunks) {
                DS_STEP(1, 1, 3, c + 1 < nchunks, c + 2 < nchunks, c + 2 < nchunks);
                if (++c < nchunks) {
                    DS_STEP(0, 2, 0, c + 1 < nchunks, c + 2 < nchunks, c + 2 < nchunks);
                    if (++c < nchunks) {
                        DS_STEP(1, 3, 1, c + 1 < nchunks, c + 2 < nchunks, c + 2 < nchunks);
                        if (++c < nchunks) DS_STEP(0, 0, 2, false, false, false);
                    }
                }
            }
        }
    }
#undef DS_STEP
#undef DS_STEPD

    // ---------------- K-lane exchange: each lane ends up owning half of the accumulator rows ----------------
    constexpr int R0 = 0;
    int r_lo = 0, r_hi = 16;
    if (KS == 2) {
        __syncthreads();                       // staging buffers are dead; reuse them for the exchange
        float* red = smem_ + (size_t)wave * MT * NT * 16 * 64;
        const int give_lo = kl == 0 ? 8 : 0;   // registers this lane hands to the other lane
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const float v = kl == 0 ? acc[mt][nt][8 + r] : acc[mt][nt][r];
                    red[((mt * NT + nt) * 16 + give_lo + r) * 64 + lane] = v;
                }
        __syncthreads();
        const int take_lo = kl == 0 ? 0 : 8;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const float v = red[((mt * NT + nt) * 16 + take_lo + r) * 64 + lane];
                    if (kl == 0) acc[mt][nt][r] += v; else acc[mt][nt][8 + r] += v;
                }
        r_lo = take_lo; r_hi = take_lo + 8;
    }
    (void)R0;
    // ---------------- epilogue (rows r_lo..r_hi-1 of every 16-register accumulator) ----------------
    const int rbase = m0 + wm * MT * 32 + 4 * (lane >> 5);
    if (EPI == 0) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int col = ntile[nt] * 32 + (lane & 31);
            if (!nvalid[nt] || col >= P.N) continue;
            int sel = 0;
            for (int o = 1; o < P.nout; ++o)
                if (col >= P.out[o].col0) sel = o;
            const OSeg os = P.out[sel];
            const int cc = col - os.col0;
            const float bias = P.bias ? gload(P.bias + col) : 0.0f;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rbase + mt * 32 + (r & 3) + 8 * (r >> 2);
                    if (row < M && r >= r_lo && r < r_hi) {
                        float v = acc[mt][nt][r] + bias;
                        if (os.add) v += gload(os.add + (size_t)row * os.add_ld + cc);
                        if (os.relu) v = relu_f(v);
                        if (BF && os.bf16)
                            *(__attribute__((address_space(1))) unsigned short*)((unsigned short*)os.base + (size_t)row * os.ld + cc) = f2bf(v);
                        else
                            gstore(os.base + (size_t)row * os.ld + cc, v);
                    }
                }
        }
    }
    static_assert(EPI == 0, "the BiLSTM epilogues left this template in round 3 (lstm_cell_*kernel)");
}

TileGeom gemm_geom(GemmCfg cfg)
{
    switch (cfg) {
    case CFG_CONV: return {128, 64, 256, 1};        // MT1 NT2 WM4 WN1
    case CFG_FC: return {128, 96, 256, 1};          // MT1 NT3 WM4 WN1, weights prefetched 2 chunks ahead
    case CFG_CONV_WIDE: return {128, 128, 256, 1};  // MT2 NT2 WM2 WN2
    case CFG_CONV_POOL: return {128, 64, 256, 1};   // CFG_CONV with maxpool(3,s1) fused into the A load
    case CFG_FC_DENSE: return {128, 96, 256, 1};    // CFG_FC for M % 128 == 0 (no row masks)
    case CFG_BCONV: return {128, 64, 256, 1};       // bf16 operands, same staging as CFG_CONV
    case CFG_BCONV_POOL: return {128, 64, 256, 1};
    case CFG_BFC: return {128, 256, 256, 1};        // MT4 NT2 WM1 WN4: every wave owns all 128 rows x 64 columns, so a
    case CFG_BFC_DENSE: return {128, 256, 256, 1};  // weight fragment is loaded by exactly one wave of the workgroup
    }
    return {0, 0, 0, 1};
}

hipError_t launch_gemm(GemmCfg cfg, const GemmLaunch* d_launch, int total_tiles, hipStream_t s)
{
    if (total_tiles <= 0) return hipSuccess;
    switch (cfg) {
    case CFG_CONV: hipLaunchKernelGGL((gemm_kernel<1, 2, 4, 1, 0, 0, 1, 1>), dim3(total_tiles), dim3(256), 0, s, d_launch); break;
    case CFG_FC: hipLaunchKernelGGL((gemm_kernel<1, 3, 4, 1, 0, 0, 2, 1>), dim3(total_tiles), dim3(256), 0, s, d_launch); break;
    case CFG_CONV_WIDE: hipLaunchKernelGGL((gemm_kernel<2, 2, 2, 2, 0, 0, 1, 1>), dim3(total_tiles), dim3(256), 0, s, d_launch); break;
    case CFG_CONV_POOL: hipLaunchKernelGGL((gemm_kernel<1, 2, 4, 1, 0, 1, 1, 1>), dim3(total_tiles), dim3(256), 0, s, d_launch); break;
    case CFG_FC_DENSE: hipLaunchKernelGGL((gemm_kernel<1, 3, 4, 1, 0, 2, 2, 1>), dim3(total_tiles), dim3(256), 0, s, d_launch); break;
    case CFG_BCONV: hipLaunchKernelGGL((gemm_kernel<1, 2, 4, 1, 0, 0, 1, 1, true>), dim3(total_tiles), dim3(256), 0, s, d_launch); break;
    case CFG_BCONV_POOL: hipLaunchKernelGGL((gemm_kernel<1, 2, 4, 1, 0, 1, 1, 1, true>), dim3(total_tiles), dim3(256), 0, s, d_launch); break;
    case CFG_BFC: hipLaunchKernelGGL((gemm_kernel<4, 2, 1, 4, 0, 0, 2, 1, true>), dim3(total_tiles), dim3(256), 0, s, d_launch); break;
    case CFG_BFC_DENSE: hipLaunchKernelGGL((gemm_kernel<4, 2, 1, 4, 0, 2, 2, 1, true>), dim3(total_tiles), dim3(256), 0, s, d_launch); break;
    default: return hipErrorInvalidValue;      // the BiLSTM cells run in their own kernels (lstm_cell_*), not through this template
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// fp32 BiLSTM cells of one wavefront diagonal (layers.py:45-72; TF LSTMCell: gate order i, j, f, o, forget_bias 1.0
// added at run time; c' = sigmoid(f + 1) c + sigmoid(i) tanh(j); h' = sigmoid(o) tanh(c')).
//
// A workgroup = 4 INDEPENDENT waves (no LDS, no barrier): wave w owns the 32 sites of m-tile 4*mb + w and NT n-tiles of
// 32 gate columns ([gate][8 units] order, so a lane ends up with the four gates of four neighbouring units of ONE
// site). Both operands stream global -> VGPR in MFMA-fragment order with 1 KiB coalesced wave loads, four k-groups
// ahead of the matrix pipe: the weights as packed by the host, the activations because the previous cell's epilogue
// wrote h in exactly that order (ds_internal.h, LstmCell). The product is issued transposed -- mfma(weights, h) --
// as in the fused inception kernel. Bias (+1 on the forget gate) and, for layer 0, the embedding-table row and the
// (mean, std, len) rank-1 terms are the accumulator's INITIAL value, computed while the first fragments are in
// flight; the previous cell state is requested up front as well, so the epilogue is pure gate arithmetic + two
// coalesced 1 KiB stores.
// Roofline: MFMA (fp32, 64 FLOP/clk/SIMD). Algorithmic FLOPs per launch = sum over cells of 2 * n * 1024 * K.
template <int NT>
__global__ __launch_bounds__(256, NT == 1 ? 4 : 2) void lstm_cell_kernel(const LstmLaunch L_)
{
    const LstmLaunch* const Lp = &L_;      // by-value kernel argument: the descriptor arrives with the dispatch packet (kernarg
                                           // segment) instead of behind a cold pointer chase at the start of every workgroup
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // diagnostic stamps (wave 0, lane 0 of every workgroup; only when a debug buffer is attached):
    // [0] s_memrealtime at entry (100 MHz, comparable across CUs)  [1..4] s_memtime (shader clock) at entry / before the
    // K loop / after it / at exit  [5] s_memrealtime at exit  [6] HW_ID  [7] XCC_ID
    unsigned long long* const sdst = (Lp->dbg && blockIdx.x < DBG_MAX_WGS) ? Lp->dbg + (size_t)blockIdx.x * 8 : nullptr;
    const bool stamp = sdst != nullptr && threadIdx.x == 0;
#define DS_LSTAMP(i, v) do { if (stamp) sdst[i] = (v); } while (0)
    DS_LSTAMP(0, __builtin_amdgcn_s_memrealtime());
    DS_LSTAMP(1, __builtin_amdgcn_s_memtime());
    DS_LSTAMP(6, (unsigned long long)__builtin_amdgcn_s_getreg(63492));     // HW_REG_HW_ID, all 32 bits
    DS_LSTAMP(7, (unsigned long long)__builtin_amdgcn_s_getreg(63508));     // HW_REG_XCC_ID
    const int bid = lstm_logical_tile(blockIdx.x, gridDim.x, Lp->cls_tiles[0], Lp->cls_tiles[1]);   // work-balanced, XCD-aware order
    const int mtiles = Lp->mtiles;
    const int mblocks = (mtiles + 3) >> 2;
    constexpr int NGROUPS = 32 / NT;
    const int per_cell = mblocks * NGROUPS;
    const int ci = bid / per_cell, rem = bid - ci * per_cell;
    const int mb = rem % mblocks, ng = rem / mblocks;
    const int mt = mb * 4 + wave;
    if (mt >= mtiles) return;                       // waves are independent: a ragged last m-block just has fewer of them
    const LstmCell& C = Lp->cell[ci];
    const int half = lane >> 5, r31 = lane & 31;
    const int n = Lp->n, T = Lp->T;
    const int row = mt * 32 + r31;
    const int rowc = row < n ? row : n - 1;         // padding rows of the last m-tile compute on a copy of the last site

    // ---- requests that the accumulator's initial value needs
    floatx16 acc[NT];
    float4 cp[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) lstm_acc_init(C, (ng * NT + nt) * 8 + 4 * half, rowc, T, acc[nt]);
    const unsigned lane4 = (unsigned)lane * 4;
    const size_t mt_off = (size_t)mt * LSTM_MT_FLOATS;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        cp[nt] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!C.c_zero) cp[nt] = gload4(C.c + mt_off + (unsigned)(ng * NT + nt) * 256 + lane4);
    }

    // ---- K loop: k-groups of 8 (one 16-byte fragment per lane and operand = four MFMAs), a ring of four register
    // stages; x rows first, then h rows, as in the TF kernel (and in the packed weights)
    const bool has_x = C.ax != nullptr, has_h = C.ah != nullptr;
    const int KG = (has_x ? 32 : 0) + (has_h ? 32 : 0);
    // one linear k-group index for both segments: group kg lives at a0 + kg KiB (+ the distance between the two
    // buffers from group 32 on when both are present) -- scalar arithmetic only, no branch in the loop
    const char* const a0 = reinterpret_cast<const char*>(has_x ? C.ax : C.ah) + mt_off * 4;
    const long dseg = (has_x && has_h) ? (reinterpret_cast<const char*>(C.ah) - reinterpret_cast<const char*>(C.ax)) - 32 * 1024 : 0;
    const char* bpn[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bpn[nt] = reinterpret_cast<const char*>(C.Bp + (size_t)(ng * NT + nt) * C.kg_stride * 256);
    const unsigned lane16 = (unsigned)lane * 16;
    DS_LSTAMP(2, __builtin_amdgcn_s_memtime());
    if (KG > 0) {
        float4 a[4], b[4][NT];
        auto request = [&](int slot, int kg) {           // slot is a literal at every call site
            const int kc = min(kg, KG - 1);              // past the end: re-request the last group (never consumed)
            const long aoff = (long)kc * 1024 + (kc >= 32 ? dseg : 0);                               // wave-uniform
            a[slot] = gload4(reinterpret_cast<const float*>(a0 + aoff + lane16));
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) b[slot][nt] = gload4(reinterpret_cast<const float*>(bpn[nt] + (long)kc * 1024 + lane16));
        };
        auto consume = [&](int slot) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[slot][nt].x, a[slot].x, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[slot][nt].y, a[slot].y, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[slot][nt].z, a[slot].z, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[slot][nt].w, a[slot].w, acc[nt], 0, 0, 0);
            }
        };
        request(0, 0); request(1, 1); request(2, 2); request(3, 3);
        for (int kg = 0; kg < KG; kg += 4) {             // KG is 32 or 64
            // sched_barrier pins the ring: without it hipcc sinks all eight requests to the bottom of the trip, and a
            // stage's operands are then only ~1 stage ahead of their MFMAs instead of 4
            consume(0); request(0, kg + 4); __builtin_amdgcn_sched_barrier(0);
            consume(1); request(1, kg + 5); __builtin_amdgcn_sched_barrier(0);
            consume(2); request(2, kg + 6); __builtin_amdgcn_sched_barrier(0);
            consume(3); request(3, kg + 7); __builtin_amdgcn_sched_barrier(0);
        }
    }
    DS_LSTAMP(3, __builtin_amdgcn_s_memtime());

    // ---- gates, new state, coalesced fragment-major stores
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int ntile = ng * NT + nt;
        float4 cn, hn;
        lstm_gates(acc[nt], cp[nt], cn, hn);
        const size_t off = mt_off + (unsigned)ntile * 256 + lane4;
        const v4f co = {cn.x, cn.y, cn.z, cn.w}, ho = {hn.x, hn.y, hn.z, hn.w};
        *(__attribute__((address_space(1))) v4f*)(C.c + off) = co;
        *(__attribute__((address_space(1))) v4f*)(C.h_out + off) = ho;
        if (C.h_row && row < n)
            *(__attribute__((address_space(1))) v4f*)(C.h_row + (size_t)row * 256 + ntile * 8 + 4 * half) = ho;
    }
    DS_LSTAMP(4, __builtin_amdgcn_s_memtime());
    DS_LSTAMP(5, __builtin_amdgcn_s_memrealtime());
#undef DS_LSTAMP
}

// The same cells with the operands SHARED through LDS. A CU's vector-memory path moves ~64 B/clk at best; the
// direct-to-register kernel above asks it for 32 B/clk/CU (every wave fetches its own 1 KiB activation and weight
// fragment per four MFMAs), which is what bounds it. Here a workgroup owns a 64-site x (64 * NT)-column block: waves
// (mi, nj) = (wave & 1, wave >> 1) take m-tile mi and the NT n-tiles of column group nj, so every activation
// fragment feeds two waves and every weight fragment two waves -- half the global traffic per MFMA. Fragments are
// 1 KiB images in global memory already, so they go global -> LDS by LDS-DMA (no VGPR round trip): a ring of three
// stages of KGS k-groups (1 at NT = 1, 2 at NT = 2), requests two stages ahead, ONE barrier per stage, a counted vmcnt in
// front of it so the next stage's requests stay in flight across the barrier.
// (glds16 / glds16s, the LDS-DMA requests, live in ds_device.h)
__device__ __forceinline__ floatx16 mfma_bf_early(float4 a, float4 b, floatx16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <int NT>
__global__ __launch_bounds__(256, NT == 1 ? 5 : 2) void lstm_cell_lds_kernel(const LstmLaunch L_)
{
    const LstmLaunch* const Lp = &L_;      // by-value kernel argument (see lstm_cell_kernel)
    constexpr int FR = 2 + 2 * NT;            // fragments per k-group: 2 m-tiles of h, 2 * NT n-tiles of weights
    constexpr int KGS = NT == 1 ? 1 : 2;      // k-groups per stage (KGS * FR fragments, a multiple of the four waves)
    constexpr int LPS = KGS * FR / 4;         // DMA requests per wave and stage
    constexpr int STAGE = KGS * FR * 256;     // floats
    // Dynamic on purpose: with a static array hipcc derives the register budget from the LDS-limited occupancy and ignores
    // __launch_bounds__. Footprint per workgroup (NT = 1): 12 KB of LDS and 88 VGPRs per wave -- up to five workgroups per
    // CU, and one of them fits next to a fused-module workgroup (2 x 184 + 88 VGPRs per SIMD, 105 + 12 KB). Measured at 512
    // sites per forward (ring = 3 stages of KGS k-groups): KGS 4 / 136 VGPRs 545 k sites/s, KGS 2 / 88 VGPRs 554 k,
    // KGS 1 / 88 VGPRs 561 k (one barrier per four MFMAs of a wave, but more waves per SIMD to hide it).
    extern __shared__ __attribute__((aligned(16))) float ring[];      // [3 * STAGE]

    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int mi = wave & 1, nj = wave >> 1;
    // diagnostic stamps (wave 0, lane 0 of every workgroup; only when a debug buffer is attached):
    // [0] s_memrealtime at entry (100 MHz, comparable across CUs)  [1..4] s_memtime (shader clock) at entry / before the
    // K loop / after it / at exit  [5] s_memrealtime at exit  [6] HW_ID  [7] XCC_ID
    unsigned long long* const sdst = (Lp->dbg && blockIdx.x < DBG_MAX_WGS) ? Lp->dbg + (size_t)blockIdx.x * 8 : nullptr;
    const bool stamp = sdst != nullptr && threadIdx.x == 0;
#define DS_LSTAMP(i, v) do { if (stamp) sdst[i] = (v); } while (0)
    DS_LSTAMP(0, __builtin_amdgcn_s_memrealtime());
    DS_LSTAMP(1, __builtin_amdgcn_s_memtime());
    DS_LSTAMP(6, (unsigned long long)__builtin_amdgcn_s_getreg(63492));     // HW_REG_HW_ID, all 32 bits
    DS_LSTAMP(7, (unsigned long long)__builtin_amdgcn_s_getreg(63508));     // HW_REG_XCC_ID
    const int bid = lstm_logical_tile(blockIdx.x, gridDim.x, Lp->cls_tiles[0], Lp->cls_tiles[1]);   // work-balanced, XCD-aware order
    const int mtiles = Lp->mtiles;
    const int mblocks = (mtiles + 1) >> 1;
    constexpr int NGROUPS = 16 / NT;
    const int per_cell = mblocks * NGROUPS;
    const int ci = bid / per_cell, rem = bid - ci * per_cell;
    const int mb = rem % mblocks, ng = rem / mblocks;
    const LstmCell& C = Lp->cell[ci];
    const int mt_raw = mb * 2 + mi;
    const bool valid = mt_raw < mtiles;                       // odd m-tile count: the last block's second half idles along
    const int mt = valid ? mt_raw : mtiles - 1;
    const int half = lane >> 5, r31 = lane & 31;
    const int n = Lp->n, T = Lp->T;
    const int row = mt * 32 + r31;
    const int rowc = row < n ? row : n - 1;
    const unsigned lane16 = (unsigned)lane * 16;

    const bool has_x = C.ax != nullptr, has_h = C.ah != nullptr;
    const int KG = (has_x ? 32 : 0) + (has_h ? 32 : 0);
    const int nstages = KG / KGS;
    const char* const a0 = reinterpret_cast<const char*>(has_x ? C.ax : C.ah);
    const long dseg = (has_x && has_h) ? (reinterpret_cast<const char*>(C.ah) - reinterpret_cast<const char*>(C.ax)) - 32 * 1024 : 0;
    // request j of this wave: linear fragment index q = wave + 4 j of the stage -> (k-group kgi, fragment f). The
    // source of a request is WAVE-UNIFORM up to the lane's 16 bytes, so it is kept as a scalar base (advanced with
    // scalar adds) plus one loop-invariant VGPR offset: on gfx950 a VALU instruction does not run in the shadow of an
    // fp32 MFMA (tools/attic/mfma_valu.hip) -- per-lane 64-bit pointer arithmetic in the K loop is matrix-pipe time lost.
    const char* src[LPS];
    int kgi_[LPS];
    bool is_a[LPS];
#pragma unroll
    for (int j = 0; j < LPS; ++j) {
        const int q = wave + 4 * j, kgi = q / FR, f = q - kgi * FR;
        kgi_[j] = kgi;
        is_a[j] = f < 2;
        if (f < 2) {
            const int m = min(mb * 2 + f, mtiles - 1);
            src[j] = a0 + (size_t)m * LSTM_MT_FLOATS * 4 + (size_t)kgi * 1024;
        } else {
            const int ntile = ng * 2 * NT + (f - 2);
            src[j] = reinterpret_cast<const char*>(C.Bp) + ((size_t)ntile * C.kg_stride + kgi) * 1024;
        }
    }
    const unsigned ring_lds = (unsigned)(size_t)(__attribute__((address_space(3))) float*)ring;    // LDS byte address
    auto request = [&](int st, int slot) __attribute__((always_inline)) {      // stage st -> ring slot st % 3
        const unsigned dst = __builtin_amdgcn_readfirstlane(ring_lds + (slot * STAGE + wave * 256) * 4);
#pragma unroll
        for (int j = 0; j < LPS; ++j) {
            const int kg = st * KGS + kgi_[j];
            const long off = (long)st * (KGS * 1024) + ((is_a[j] && kg >= 32) ? dseg : 0);
            glds16s(src[j] + off, lane16, dst + j * 4096);     // fragment q = wave + 4 j of the stage, 1 KiB each
        }
    };
    if (nstages > 0) request(0, 0);
    if (nstages > 1) request(1, 1);

    // ---- accumulator's initial value: bias (+1 on f), layer 0: table row + (mean, std, len) terms
    floatx16 acc[NT];
    float4 cp[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) lstm_acc_init(C, ((ng * 2 + nj) * NT + nt) * 8 + 4 * half, rowc, T, acc[nt]);
    const size_t mt_off = (size_t)mt * LSTM_MT_FLOATS;
    const unsigned lane4 = (unsigned)lane * 4;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        cp[nt] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!C.c_zero) cp[nt] = gload4(C.c + mt_off + (unsigned)(((ng * 2 + nj) * NT + nt)) * 256 + lane4);
    }

    // every compiler-visible load is retired here, so hipcc has no reason to put a vmcnt wait into the loop below (it
    // cannot see the LDS-DMA requests, which are counted by hand)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        asm volatile("" : "+v"(cp[nt].x), "+v"(cp[nt].y), "+v"(cp[nt].z), "+v"(cp[nt].w));
        asm volatile("" : "+v"(acc[nt]));          // bias loads land directly in the accumulator registers
    }
    DS_LSTAMP(2, __builtin_amdgcn_s_memtime());
    // stage st has landed in LDS once every wave's requests for it are done: counted wait (stage st + 1 stays in
    // flight), then the barrier; it also tells everybody that ring slot (st + 2) % 3 -- read during stage st - 1 -- is
    // free again. The ring index is a compile-time constant of each of the three unrolled bodies, so every LDS address
    // is a loop-invariant VGPR plus an immediate.
    const float* const fa0 = ring + mi * 256 + lane4;
    const float* const fb0 = ring + (2 + nj * NT) * 256 + lane4;
    auto stage = [&](int st, auto slot_c) __attribute__((always_inline)) {
        constexpr int SLOT = decltype(slot_c)::value;
        if (st + 1 < nstages) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (st + 2 < nstages) request(st + 2, (SLOT + 2) % 3);
#pragma unroll
        for (int kgi = 0; kgi < KGS; ++kgi) {
            const float4 a = *reinterpret_cast<const float4*>(fa0 + SLOT * STAGE + kgi * FR * 256);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float4 b = *reinterpret_cast<const float4*>(fb0 + SLOT * STAGE + (kgi * FR + nt) * 256);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(b.x, a.x, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(b.y, a.y, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(b.z, a.z, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(b.w, a.w, acc[nt], 0, 0, 0);
            }
        }
    };
    for (int st = 0; st < nstages;) {
        stage(st, LdsSlot<0>{}); if (++st >= nstages) break;
        stage(st, LdsSlot<1>{}); if (++st >= nstages) break;
        stage(st, LdsSlot<2>{}); ++st;
    }

    DS_LSTAMP(3, __builtin_amdgcn_s_memtime());
    if (!valid) return;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int ntile = (ng * 2 + nj) * NT + nt;
        float4 cn, hn;
        lstm_gates(acc[nt], cp[nt], cn, hn);
        const size_t off = mt_off + (unsigned)ntile * 256 + lane4;
        const v4f co = {cn.x, cn.y, cn.z, cn.w}, ho = {hn.x, hn.y, hn.z, hn.w};
        *(__attribute__((address_space(1))) v4f*)(C.c + off) = co;
        *(__attribute__((address_space(1))) v4f*)(C.h_out + off) = ho;
        if (C.h_row && row < n)
            *(__attribute__((address_space(1))) v4f*)(C.h_row + (size_t)row * 256 + ntile * 8 + 4 * half) = ho;
    }
    DS_LSTAMP(4, __builtin_amdgcn_s_memtime());
    DS_LSTAMP(5, __builtin_amdgcn_s_memrealtime());
#undef DS_LSTAMP
}

// ---------------------------------------------------------------------------------------------
// The same cells with bf16 OPERANDS (DS_PRECISION_BF16_ALL): h and the weights are bf16, products accumulate in fp32
// (v_mfma_f32_32x32x16_bf16), the layer-0 table row / rank-1 terms, the gates and the cell state stay fp32.
//
// A bf16 MFMA does 16x the work of an fp32 one per cycle, so what bounds this kernel is operand delivery, not the matrix
// pipe: a diagonal at 4096 sites per forward moves ~75 MB (c: 8 MB per cell read + written in fp32, h: 4 + 2 MB, weights
// 1 MB) for 21 GFLOP -- ~20 us of memory against ~9 us of MFMA at peak. Design for bytes per MFMA:
//   * workgroup tile = 64 MTW sites x 64 NTW columns, 4 waves as 2 x 2, each wave MTW x NTW tiles of 32 x 32: at 2 x 2 a
//     k-step (16 k) needs 8 KB from L2 for 16 MFMAs -- 0.5 KB per MFMA (the round-1 template re-loaded every weight
//     fragment in each of its four waves: 1.25 KB per MFMA) -- and 1 KB of LDS reads per MFMA;
//   * operands are 1 KiB fragment images in global memory (weights packed so by the host; h because the previous cell's
//     epilogue wrote it so) and go global -> LDS by LDS-DMA, ring of three stages of two k-steps, one barrier per stage
//     behind a counted vmcnt -- the structure of lstm_cell_lds_kernel;
//   * h lives FRAGMENT-MAJOR in bf16: [m-tile of 32 sites][k-step of 16 units][64 lanes][8 bf16], lane (r, half) holding
//     units 16 s + 8 half .. + 7 of site 32 m + r. A lane of the epilogue owns four neighbouring units of one site
//     (8 bytes): lanes (r, 0) and (r, 1) of an n-tile fill the two halves of one 16-byte slot, a wave store covers 512
//     contiguous bytes. c keeps the fp32 kernels' fragment-major layout.
// Roofline: HBM / L2 (operand delivery). Algorithmic FLOPs per launch = sum over cells of 2 * n * 1024 * K.
constexpr int LSTM_MT_BYTES_BF16 = 32 * 256 * 2;      // one m-tile of a bf16 fragment-major h buffer
template <int MTW, int NTW>
__global__ __launch_bounds__(256, MTW * NTW >= 4 ? 2 : 3) void lstm_cell_bf16_kernel(const LstmLaunch L_)
{
    const LstmLaunch* const Lp = &L_;
    constexpr int FRA = 2 * MTW, FRB = 2 * NTW, FR = FRA + FRB;     // 1 KiB fragments per k-step: m-tiles of h, n-tiles of weights
    constexpr int KGS = 2;                                          // k-steps per ring stage
    constexpr int NF = KGS * FR;                                    // fragments per stage (a multiple of the four waves)
    static_assert(NF % 4 == 0, "fragments of a stage are dealt to four waves");
    constexpr int LPS = NF / 4;                                     // DMA requests per wave and stage
    constexpr int STAGE = NF * 256;                                 // floats
    extern __shared__ __attribute__((aligned(16))) float ring[];    // [3 * STAGE]

    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int mi = wave & 1, nj = wave >> 1;
    unsigned long long* const sdst = (Lp->dbg && blockIdx.x < DBG_MAX_WGS) ? Lp->dbg + (size_t)blockIdx.x * 8 : nullptr;
    const bool stamp = sdst != nullptr && threadIdx.x == 0;
#define DS_LSTAMP(i, v) do { if (stamp) sdst[i] = (v); } while (0)
    DS_LSTAMP(0, __builtin_amdgcn_s_memrealtime());
    DS_LSTAMP(1, __builtin_amdgcn_s_memtime());
    DS_LSTAMP(6, (unsigned long long)__builtin_amdgcn_s_getreg(63492));
    DS_LSTAMP(7, (unsigned long long)__builtin_amdgcn_s_getreg(63508));
    const int bid = lstm_logical_tile(blockIdx.x, gridDim.x, Lp->cls_tiles[0], Lp->cls_tiles[1]);
    const int mtiles = Lp->mtiles;
    const int mblocks = (mtiles + FRA - 1) / FRA;
    constexpr int NGROUPS = 32 / FRB;
    const int per_cell = mblocks * NGROUPS;
    const int ci = bid / per_cell, rem = bid - ci * per_cell;
    // n-groups of one m-block are neighbours (the fp32 kernels walk the m-blocks of one weight panel instead): the workgroups
    // an XCD has resident at one time then cover a few m-blocks x ALL n-groups -- every h fragment is fetched into that L2
    // once and used by eight workgroups, and the cell's whole weight matrix (1 MB in bf16) stays resident beside it
    const int ng = rem % NGROUPS, mb = rem / NGROUPS;
    const LstmCell& C = Lp->cell[ci];
    const int half = lane >> 5, r31 = lane & 31;
    const int n = Lp->n, T = Lp->T;
    const unsigned lane16 = (unsigned)lane * 16, lane4 = (unsigned)lane * 4;

    const bool has_x = C.ax != nullptr, has_h = C.ah != nullptr;
    const int KS = (has_x ? 16 : 0) + (has_h ? 16 : 0);             // k-steps of 16: x rows first, then h rows (TF kernel order)
    const int nstages = KS / KGS;
    const char* const a0 = reinterpret_cast<const char*>(has_x ? C.ax : C.ah);
    const long dseg = (has_x && has_h) ? (reinterpret_cast<const char*>(C.ah) - reinterpret_cast<const char*>(C.ax)) - 16 * 1024 : 0;
    // request j of this wave = fragment q = wave + 4 j of a stage -> (k-step kgi inside the stage, fragment f of the k-step);
    // sources are wave-uniform bases (scalar adds in the loop) + one loop-invariant lane offset
    const char* src[LPS];
    int kgi_[LPS];
    bool is_a[LPS];
#pragma unroll
    for (int j = 0; j < LPS; ++j) {
        const int q = wave + 4 * j, kgi = q / FR, f = q - kgi * FR;
        kgi_[j] = kgi;
        is_a[j] = f < FRA;
        if (f < FRA) {
            const int m = min(mb * FRA + f, mtiles - 1);
            src[j] = a0 + (size_t)m * LSTM_MT_BYTES_BF16 + (size_t)kgi * 1024;
        } else {
            const int ntile = ng * FRB + (f - FRA);
            src[j] = reinterpret_cast<const char*>(C.Bp) + ((size_t)ntile * C.kg_stride + kgi) * 1024;      // kg_stride: k-steps per n-tile panel
        }
    }
    const unsigned ring_lds = (unsigned)(size_t)(__attribute__((address_space(3))) float*)ring;
    auto request = [&](int st, int slot) __attribute__((always_inline)) {
        const unsigned dst = __builtin_amdgcn_readfirstlane(ring_lds + (slot * STAGE + wave * 256) * 4);
#pragma unroll
        for (int j = 0; j < LPS; ++j) {
            const int ks = st * KGS + kgi_[j];
            const long off = (long)st * (KGS * 1024) + ((is_a[j] && ks >= 16) ? dseg : 0);
            glds16s(src[j] + off, lane16, dst + j * 4096);
        }
    };
    if (nstages > 0) request(0, 0);
    if (nstages > 1) request(1, 1);

    // ---- this wave's tiles: m-tiles mb * FRA + mi * MTW + i, n-tiles ng * FRB + nj * NTW + j
    int mt[MTW];
    bool valid[MTW];
#pragma unroll
    for (int i = 0; i < MTW; ++i) {
        const int raw = mb * FRA + mi * MTW + i;
        valid[i] = raw < mtiles;
        mt[i] = valid[i] ? raw : mtiles - 1;
    }
    floatx16 acc[MTW][NTW];
    float4 cp[MTW][NTW];
#pragma unroll
    for (int i = 0; i < MTW; ++i) {
        const int row = mt[i] * 32 + r31;
        const int rowc = row < n ? row : n - 1;
#pragma unroll
        for (int j = 0; j < NTW; ++j) lstm_acc_init(C, (ng * FRB + nj * NTW + j) * 8 + 4 * half, rowc, T, acc[i][j]);
    }
    const bool c_zero = C.c_zero != 0;
#pragma unroll
    for (int i = 0; i < MTW; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            cp[i][j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!c_zero) cp[i][j] = gload4(C.c + (size_t)mt[i] * LSTM_MT_FLOATS + (unsigned)(ng * FRB + nj * NTW + j) * 256 + lane4);
        }
    // every compiler-visible load is retired here: the loop's vmcnt waits count the LDS-DMA requests only. (Leaving the
    // cell-state loads in flight across the first stages, with the waits widened by their number, was tried: on MI355X two
    // engines then disagreed in a few bits -- LDS-DMA requests and loads to registers do not retire strictly in issue order
    // with respect to each other, so a count cannot tell which of the two kinds is still outstanding. Requesting c late instead -- behind
    // the last ring request, the following stages waiting with vmcnt(0) -- is correct and was measured: 672 -> 662 us per step
    // at 4096 sites, 165 -> 168 us at 512: the prologue is not the c loads.)
#pragma unroll
    for (int i = 0; i < MTW; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            asm volatile("" : "+v"(cp[i][j].x), "+v"(cp[i][j].y), "+v"(cp[i][j].z), "+v"(cp[i][j].w));
            asm volatile("" : "+v"(acc[i][j]));
        }
    DS_LSTAMP(2, __builtin_amdgcn_s_memtime());

    const float* const fa0 = ring + (mi * MTW) * 256 + lane4;
    const float* const fb0 = ring + (FRA + nj * NTW) * 256 + lane4;
    // (SplitRing::run_piped's register-piped, pinned K loop -- ds_split.hip -- was built into this kernel in round 6, git 5dac46d: bit-identical,
    // 715 - 722 against 689 - 693 us per 4,096-site step, 159 against 163 us at 512; at 2 - 3 workgroups per CU the other waves already fill the
    // gaps that order closes for the one-wave-per-SIMD split kernels: profiles/r06_bf16_lstm_piped.json)
    auto stage = [&](int st, auto slot_c) __attribute__((always_inline)) {
        constexpr int SLOT = decltype(slot_c)::value;
        if (st + 1 < nstages) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (st + 2 < nstages) request(st + 2, (SLOT + 2) % 3);
#pragma unroll
        for (int kgi = 0; kgi < KGS; ++kgi) {
            float4 a[MTW], b[NTW];
#pragma unroll
            for (int i = 0; i < MTW; ++i) a[i] = *reinterpret_cast<const float4*>(fa0 + SLOT * STAGE + (kgi * FR + i) * 256);
#pragma unroll
            for (int j = 0; j < NTW; ++j) b[j] = *reinterpret_cast<const float4*>(fb0 + SLOT * STAGE + (kgi * FR + j) * 256);
#pragma unroll
            for (int i = 0; i < MTW; ++i)
#pragma unroll
                for (int j = 0; j < NTW; ++j) acc[i][j] = mfma_bf_early(b[j], a[i], acc[i][j]);     // transposed: (h W)^T
        }
    };
    for (int st = 0; st < nstages;) {
        stage(st, LdsSlot<0>{}); if (++st >= nstages) break;
        stage(st, LdsSlot<1>{}); if (++st >= nstages) break;
        stage(st, LdsSlot<2>{}); ++st;
    }
    DS_LSTAMP(3, __builtin_amdgcn_s_memtime());

    // ---- gates (fp32), new state; c fragment-major fp32, h fragment-major bf16 (8 bytes per lane), optional row-major fp32 h
    typedef unsigned int u2v __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int i = 0; i < MTW; ++i) {
        if (!valid[i]) continue;
        const int row = mt[i] * 32 + r31;
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const int ntile = ng * FRB + nj * NTW + j;
            float4 cn, hn;
            lstm_gates(acc[i][j], cp[i][j], cn, hn);
            const v4f co = {cn.x, cn.y, cn.z, cn.w};
            *(__attribute__((address_space(1))) v4f*)(C.c + (size_t)mt[i] * LSTM_MT_FLOATS + (unsigned)ntile * 256 + lane4) = co;
            const u2v ho = {pack_bf2(hn.x, hn.y), pack_bf2(hn.z, hn.w)};
            char* const hb = reinterpret_cast<char*>(C.h_out) + (size_t)mt[i] * LSTM_MT_BYTES_BF16 + (unsigned)(ntile >> 1) * 1024 +
                             (unsigned)(((ntile & 1) * 32 + r31) * 16 + half * 8);
            *(__attribute__((address_space(1))) u2v*)hb = ho;
            if (C.h_row && row < n) {
                const v4f hr = {hn.x, hn.y, hn.z, hn.w};
                *(__attribute__((address_space(1))) v4f*)(C.h_row + (size_t)row * 256 + ntile * 8 + 4 * half) = hr;
            }
        }
    }
    DS_LSTAMP(4, __builtin_amdgcn_s_memtime());
    DS_LSTAMP(5, __builtin_amdgcn_s_memrealtime());
#undef DS_LSTAMP
}

hipError_t launch_lstm_cells(int nt, const LstmLaunch& L, hipStream_t s)
{
    const int ncell = L.ncell, mtiles = L.mtiles;
    if (ncell <= 0 || mtiles <= 0) return hipSuccess;
    const int mblocks = (mtiles + 3) / 4, mblocks2 = (mtiles + 1) / 2;
    switch (nt) {
    case 101: hipLaunchKernelGGL(lstm_cell_lds_kernel<1>, dim3(ncell * mblocks2 * 16), dim3(256), 3 * 1 * 4 * 1024, s, L); break;      // 3 stages x KGS x FR KiB
    case 102: hipLaunchKernelGGL(lstm_cell_lds_kernel<2>, dim3(ncell * mblocks2 * 8), dim3(256), 3 * 2 * 6 * 1024, s, L); break;
    // bf16-operand cells (DS_PRECISION_BF16_ALL): 2MN = workgroup tile of 64 M sites x 64 N columns
    case 211: hipLaunchKernelGGL((lstm_cell_bf16_kernel<1, 1>), dim3(ncell * ((mtiles + 1) / 2) * 16), dim3(256), 3 * 2 * 4 * 1024, s, L); break;
    case 212: hipLaunchKernelGGL((lstm_cell_bf16_kernel<1, 2>), dim3(ncell * ((mtiles + 1) / 2) * 8), dim3(256), 3 * 2 * 6 * 1024, s, L); break;
    case 222: hipLaunchKernelGGL((lstm_cell_bf16_kernel<2, 2>), dim3(ncell * ((mtiles + 3) / 4) * 8), dim3(256), 3 * 2 * 8 * 1024, s, L); break;
    // (a 128 x 256 tile -- a wave owning 64 sites x 128 columns, 0.09 KiB of operand fragments per MFMA against 0.125 -- was measured in
    // round 5: 756 against 673 us per 4096-site step; 768 workgroups of 72 KB rings fill the GPU in 1.5 rounds)
    case 1: hipLaunchKernelGGL(lstm_cell_kernel<1>, dim3(ncell * mblocks * 32), dim3(256), 0, s, L); break;
    case 2: hipLaunchKernelGGL(lstm_cell_kernel<2>, dim3(ncell * mblocks * 16), dim3(256), 0, s, L); break;
    case 4: hipLaunchKernelGGL(lstm_cell_kernel<4>, dim3(ncell * mblocks * 8), dim3(256), 0, s, L); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Fused inception module. One workgroup (8 waves) owns a tile of whole sites (<= 96 rows):
//   P1  [rows x cin] x [cin x 256]: the six 1x1 convs that read the module input in ONE pass over
//       it (branch 1 reads the 3-tap max-pooled rows, staged next to the plain rows). Wave w owns
//       n-tile w for all m-tiles. b2/b1 go straight to HBM, the three 32-channel intermediates go
//       to LDS (T1, with zero halo rows = SAME padding), the residual stem stays in accumulators.
//   P2a 1x3 64-ch conv of branch 5 (T1 -> T2 in LDS) + part of branch 3.
//   P2b branch 5's last 1x1 accumulates ON TOP of the stem accumulators (waves 0,1) while the other
//       waves finish the 1x3 / 1x5 convs of branches 3 and 4.
// HBM traffic per module = read input once + write output once (module-granular bytes).
#ifndef DS_FUSED_WPS
#define DS_FUSED_WPS 2
#endif
// Register stages of the fused module's P1 operand prefetch (weight fragments / input rows): 2 = one chunk ahead. Deeper
// rings were measured (3, 4 stages each): the same P1 time -- the operands are not late -- and more registers. The
// register count matters beyond spills: at 183 VGPRs two module waves leave room for one BiLSTM cell wave (136) on the
// same SIMD, and a cell workgroup's 48 KB of LDS fits next to the module's 105 - 111 KB, so cell workgroups run in the
// shadow of module workgroups (512-site bench 527 k -> 540 k sites/s; 532 k at 188, 529 k at 196 VGPRs).
#ifndef DS_FUSED_BD
#define DS_FUSED_BD 2
#endif
// (s_setprio 2 for the fp32 pooling waves, as in the bf16 kernel, was measured: 625 against 586 us per step)
#ifndef DS_FUSED_POOLPRIO
#define DS_FUSED_POOLPRIO 0
#endif
#ifndef DS_FUSED_VD
#define DS_FUSED_VD 2
#endif
constexpr int F_LDA = KC + 4;   // staged chunk row stride (floats)
constexpr int F_LD1 = 100;      // T1 row stride: 96 channels + 4 pad (25 x 16 B: odd -> conflict-free b128)
constexpr int F_LD2 = 68;       // T2 row stride: 64 channels + 4 pad

size_t inception_fused_lds_bytes(int tm, int W, int spt)
{
    const int tr32 = tm * 32;
    // region A (chunk staging, later the b1|b2 output tile) | T1 | T2 | rowmap
    return (size_t)(tr32 * F_LD1 + (spt * (W + 4) + 5) * F_LD1 + tr32 * F_LD2 + tr32 + 192) * sizeof(float);
}

// All weights of a conv unit (ntaps x 4 k-groups, <= 20 float4): requested EARLY — before the barrier or
// the epilogue in front of the unit — so their L2 latency is off the unit's critical path.
__device__ __forceinline__ void fused_unit_prefetch(const float* __restrict__ Bp, int ntaps, int nt, int lane, float4 (&ub)[20])
{
    const float* bsrc = Bp + ((size_t)(nt * ntaps * 4) * 64 + lane) * 4;
#pragma unroll
    for (int g = 0; g < 20; ++g)
        if (g < ntaps * 4) ub[g] = gload4(bsrc + g * 256);
}

template <int NTAPS>
__device__ __forceinline__ void fused_conv_unit(const float* T1, int rm, int coloff, int lane, const float4 (&ub)[20], floatx16& acc)
{
    const float* base = T1 + rm * F_LD1 + coloff + (lane >> 5) * 4;
#pragma unroll
    for (int t = 0; t < NTAPS; ++t) {
        const float* arow = base + (t - NTAPS / 2) * F_LD1;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 a = *reinterpret_cast<const float4*>(arow + g * 8);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ub[t * 4 + g].x, a.x, acc, 0, 0, 0);     // transposed: see the kernel
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ub[t * 4 + g].y, a.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ub[t * 4 + g].z, a.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ub[t * 4 + g].w, a.w, acc, 0, 0, 0);
        }
    }
}

struct FusedTagT { static constexpr bool value = true; };
struct FusedTagF { static constexpr bool value = false; };

template <int TM>
__device__ __forceinline__ void inception_fused_body(const FusedChain& c)
{
    constexpr int TR32 = TM * 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ad = smem;                         // [2][TR32*F_LDA] staged input rows
    float* Ys = smem;                         // [TR32][F_LD1] b1|b2 output tile, aliases Ad once P1 is done
    float* T1 = smem + TR32 * F_LD1;          // [spt*(W+4)][F_LD1]   (TR32*F_LD1 >= 4*TR32*F_LDA)
    const int W = c.m[0].W, spt = c.m[0].spt;          // the same for every module of a chain (ds_internal.h FusedChain)
    float* T2 = T1 + (spt * (W + 4) + 5) * F_LD1;   // [TR32*F_LD2]  (T1 has 5 spare rows: a dump row for padding rows + its halo)
    int* rowmap = reinterpret_cast<int*>(T2 + TR32 * F_LD2);   // [TR32] tile row -> T1 row
    float* const Bs = reinterpret_cast<float*>(rowmap + TR32); // [3][64] biases of b5b | b3b | b4b
    const int site0 = blockIdx.x * spt;
    const int nhere = min(spt, c.m[0].n_sites - site0);
    const int TRv = nhere * W;                // valid rows of this tile
    const size_t grow0 = (size_t)site0 * W;
    // once per workgroup: zero halo rows of T1 (every module rewrites the interior rows completely and never touches the
    // halos) and the tile's row map
    for (int i = threadIdx.x; i < (spt * (W + 4) + 5) * (F_LD1 / 4); i += 512)
        reinterpret_cast<float4*>(T1)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (threadIdx.x < TR32)   // rows past the tile's last site map to the dump row, so LDS writes need no predicate
        rowmap[threadIdx.x] = (int)threadIdx.x < TRv ? ((int)threadIdx.x / W) * (W + 4) + 2 + (int)threadIdx.x % W : spt * (W + 4) + 2;

    // ---- the modules of the chain, one after the other on THIS tile (one module per launch in the diagnostic modes): module
    // k + 1 reads the rows module k has just written -- same workgroup, so they come back from this XCD's L2 and no launch
    // boundary (with its drain and its cold start) sits between the modules
    for (int mi = 0; mi < c.nmod; ++mi) {
    const FusedArgs& a = c.m[mi];
    // every per-lane quantity derives from this opaque copy of the thread index: hipcc otherwise hoists loop-invariant
    // per-lane addresses out of the module loop, and the kernel must stay within 184 VGPRs (co-tenancy)
    int tid_opaque = threadIdx.x, wave_opaque = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    asm volatile("" : "+v"(tid_opaque), "+s"(wave_opaque));
    const int tid = tid_opaque, lane = tid & 63, wave = wave_opaque;
    const int cin = a.cin;
    const gptr1w Yg = (gptr1w)(a.Y + grow0 * 240);     // wave-uniform base; per-lane offsets stay 32-bit

    // diagnostic phase stamps (wave 0 and wave 7, lane 0): only when a debug buffer is attached
    const bool stamp = a.dbg != nullptr && lane == 0 && (wave == 0 || wave == 7) && blockIdx.x < DBG_MAX_WGS;
    // (the destination is recomputed at every stamp: a pointer held in VGPRs for the whole kernel costs two registers)
#define DS_STAMP(i) do { if (stamp) (a.dbg + ((size_t)blockIdx.x * 2 + (wave == 7)) * 8)[i] = __builtin_amdgcn_s_memtime(); } while (0)
    DS_STAMP(0);
    if (tid >= 256 && tid < 448) {
        const int q = tid - 256;
        Bs[q] = gload((q < 64 ? a.bias5b : q < 128 ? a.bias3b : a.bias4b) + (q & 63));
    }

    // ---- P1 staging cursor: thread -> (row, 16-byte slot); rows past the tile end re-read row TRv-1
    const bool stager = tid < TR32 * 4;
    const int sr = tid >> 2, sq = tid & 3;
    const int rr = sr < TRv ? sr : TRv - 1;
    // plain input: one row per staged row. Pooled input (modules right after maxpool_layer2/3): the staged
    // row (site s, w) is the max of input rows 2w - pad + {0,1,2} of site s that exist (padded taps ignored).
    const bool pooled_in = a.pool_win > 0;
    const float *pc, *pq = nullptr, *pr = nullptr;
    if (!pooled_in) {
        pc = a.X + (grow0 + rr) * cin + sq * 4;
    } else {
        const int s_ = rr / W, w_ = rr % W;
        const int i0 = 2 * w_ - a.pool_pad;
        const int ia = i0 < 0 ? i0 + 1 : i0;                               // first existing tap
        const int ib = i0 + 1 < a.pool_win ? (i0 + 1 < 0 ? ia : i0 + 1) : ia;
        const int ic = i0 + 2 < a.pool_win ? i0 + 2 : ib;
        const float* sb = a.X + ((size_t)(site0 + s_) * a.pool_win) * cin + sq * 4;
        pc = sb + (size_t)ia * cin;
        pq = sb + (size_t)ib * cin;
        pr = sb + (size_t)ic * cin;
    }
    const float* bp = a.Bp1 + ((size_t)wave * ((cin + 31) / 32 * 4) * 64 + lane) * 4;   // K padded to 32 in the pack

    // Every MFMA of this kernel is issued TRANSPOSED -- mfma(weight fragment, activation fragment) computes (X W)^T --
    // so a lane ends up holding, for ONE activation row (lane & 31), 4 x 4 consecutive output channels
    // (register 4g+e <-> channel 32*ntile + 8g + 4*(lane >> 5) + e). Results leave as float4 groups (LDS and HBM)
    // instead of scalars, and the bias is simply the accumulator's initial value.
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

    // branch 1's maxpool(3, stride 1, SAME): waves 6,7 build their A fragments as the max of three staged
    // rows (previous / own / next; at a site edge the missing neighbour is the own row = "padded taps
    // ignored"), so the input is read from HBM once and staged once.
    int offm[TM], offp[TM];
#pragma unroll
    for (int mt = 0; mt < TM; ++mt) {
        const int row = mt * 32 + (lane & 31);
        const int w = row % W;
        offm[mt] = (row < TRv && w > 0) ? -F_LDA : 0;
        offp[mt] = (row < TRv && w < W - 1) ? F_LDA : 0;
    }

    const int nchunks = cin / KC;      // >= 15

    // P1 main loop: one barrier per 16-channel chunk; fragments of chunk c+1 are read into registers behind the second
    // half of chunk c's MFMAs; input rows and weight fragments are requested DS_FUSED_VD - 1 / DS_FUSED_BD - 1 chunks
    // ahead. Fully unrolled over the 15 or 16 chunks, so every register-ring and LDS-buffer index is a literal (183
    // instead of 247 VGPRs: see DS_FUSED_BD above).
    __builtin_assume(nchunks >= 15 && nchunks <= 16);
    auto run_p1 = [&](auto pool_tag) {
        constexpr bool POOL = decltype(pool_tag)::value;
        constexpr int VD = DS_FUSED_VD;
        float4 vc[VD];
        constexpr int BD = DS_FUSED_BD;                  // register stages of the weight fragments (chunk c's are requested BD - 1 steps ahead)
        float4 bq[BD][2];
        float4 af[2][2][TM];
        // (the pooling waves 6, 7 hold threads 384 .. 511 and a tile has at most 96 x 4 = 384 staging slots: they never stage,
        // and their variant carries neither the staging registers nor the row pointers)
        auto load_a = [&](int V) __attribute__((always_inline)) {
            if (!POOL && stager) {
                vc[V] = gload4(pc); pc += KC;
                if (pooled_in) {       // wave-uniform per launch
                    vc[V] = f4max(f4max(vc[V], gload4(pq)), gload4(pr));
                    pq += KC; pr += KC;
                }
            }
        };
        auto store_a = [&](int X, int V) __attribute__((always_inline)) {
            if (!POOL && stager) *reinterpret_cast<float4*>(Ad + X * TR32 * F_LDA + sr * F_LDA + sq * 4) = vc[V];
        };
        auto load_b = [&](int Bi) __attribute__((always_inline)) {
            bq[Bi][0] = gload4(bp);
            bq[Bi][1] = gload4(bp + 256);
            bp += 512;
        };
        auto read_frags = [&](int X) __attribute__((always_inline)) {
#pragma unroll
            for (int rs = 0; rs < 2; ++rs)
#pragma unroll
                for (int mt = 0; mt < TM; ++mt) {
                    const float* q = Ad + X * TR32 * F_LDA + (mt * 32 + (lane & 31)) * F_LDA + rs * 8 + (lane >> 5) * 4;
                    float4 v = *reinterpret_cast<const float4*>(q);
                    if (POOL) v = f4max(f4max(v, *reinterpret_cast<const float4*>(q + offm[mt])), *reinterpret_cast<const float4*>(q + offp[mt]));
                    af[X][rs][mt] = v;
                }
        };
        auto mfma_rs = [&](int X, int Bi, int rs) __attribute__((always_inline)) {
#pragma unroll
            for (int mt = 0; mt < TM; ++mt) {
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(bq[Bi][rs].x, af[X][rs][mt].x, acc[mt], 0, 0, 0);
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(bq[Bi][rs].y, af[X][rs][mt].y, acc[mt], 0, 0, 0);
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(bq[Bi][rs].z, af[X][rs][mt].z, acc[mt], 0, 0, 0);
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(bq[Bi][rs].w, af[X][rs][mt].w, acc[mt], 0, 0, 0);
            }
        };
#pragma unroll
        for (int i = 0; i < VD; ++i) load_a(i);                  // chunks 0 .. VD - 1 (nchunks >= 15)
#pragma unroll
        for (int i = 0; i < BD - 1; ++i) load_b(i);              // weight fragments of chunks 0 .. BD - 2
        store_a(0, 0);
        __syncthreads();
        read_frags(0);
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            if (c < nchunks) {                                    // wave-uniform; only c = 15 is really conditional
                const int X = c & 1;
                const bool has1 = c + 1 < nchunks;
                if (c + VD < nchunks) load_a(c % VD);             // chunk c + VD into the stage chunk c left (written at step c - 1)
                if (c + BD - 1 < nchunks) load_b((c + BD - 1) % BD);
                if (has1) store_a(X ^ 1, (c + 1) % VD);
                mfma_rs(X, c % BD, 0);
                __builtin_amdgcn_sched_barrier(0);
                __syncthreads();
                if (has1) read_frags(X ^ 1);
                mfma_rs(X, c % BD, 1);
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                if (has1) __builtin_amdgcn_sched_group_barrier(0x100, (POOL ? 6 : 2) * TM, 0);
                __builtin_amdgcn_sched_group_barrier(0x8, 4 * TM, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    // the two pooling waves carry three LDS reads + eight v_max per fragment where the others carry one read: every chunk's
    // barrier waits for them, so they get the SIMD's issue priority while P1 runs (DS_FUSED_POOLPRIO, measured)
    if (wave >= 6) { __builtin_amdgcn_s_setprio(DS_FUSED_POOLPRIO); run_p1(FusedTagT{}); __builtin_amdgcn_s_setprio(0); }
    else run_p1(FusedTagF{});     // wave-uniform
    DS_STAMP(1);
    __syncthreads();   // all fragment reads of the staging area are done before T2 aliases it
    DS_STAMP(2);

    // ---- static wave -> unit assignment of P2 (wave-uniform). kind: 0 none, 1 b5b, 2 b3b, 3 b4b. Waves w and w+4 share
    // a SIMD; the table keeps the MFMA count per SIMD within 5 % across P2a + P2b.
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
    float4 pf[20];                                   // prefetched weights of the next unit
    // (defined on every path: a register array that is only loaded under a wave-uniform condition is "undefined on some
    // paths", and hipcc keeps such a value live around the whole module loop -- 80 + 32 VGPRs of phantom live range here)
#pragma unroll
    for (int g = 0; g < 20; ++g) pf[g] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a1k) fused_unit_prefetch(unit_Bp(a1k), unit_taps(a1k), a1n, lane, pf);

    // ---- P1 epilogue (bias already inside acc): route the 256 columns. b1|b2 go through an LDS tile and leave as
    // whole 384-B row segments.
    if (wave >= 3 && wave <= 5) {            // wave-uniform: n-tiles 3,4,5 = b3a | b4a | b5a -> T1, through the row map
#pragma unroll
        for (int mt = 0; mt < TM; ++mt) {
            const int rm = rowmap[mt * 32 + rlane];
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(T1 + rm * F_LD1 + (wave * 32 - 96) + 8 * g + h4) =
                    make_float4(relu_f(acc[mt][4 * g]), relu_f(acc[mt][4 * g + 1]), relu_f(acc[mt][4 * g + 2]),
                                relu_f(acc[mt][4 * g + 3]));
        }
    } else if (wave != 0) {                  // n-tiles 1,2 (b5s tail | b2) and 6,7 (b1 | padding) -> output tile
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int col = wave * 32 + 8 * g + h4;          // groups of 4 never straddle 48 / 240
            if (col >= 48 && col < 240) {
                const int ycol = col < 96 ? col : col - 192;        // position inside Y[:, 0:96) (b1 first, then b2)
#pragma unroll
                for (int mt = 0; mt < TM; ++mt)
                    *reinterpret_cast<float4*>(Ys + (mt * 32 + rlane) * F_LD1 + ycol) =
                        make_float4(relu_f(acc[mt][4 * g]), relu_f(acc[mt][4 * g + 1]), relu_f(acc[mt][4 * g + 2]),
                                    relu_f(acc[mt][4 * g + 3]));
            }
        }
    }
    DS_STAMP(3);
    __syncthreads();   // T1 and the b1|b2 tile complete
    DS_STAMP(4);
    for (int idx = tid; idx < TR32 * 24; idx += 512) {
        const int row = idx / 24, q = idx - row * 24;
        if (row < TRv) {
            const float4 v = *reinterpret_cast<const float4*>(Ys + row * F_LD1 + q * 4);
            v4f o = {v.x, v.y, v.z, v.w};
            *(__attribute__((address_space(1))) v4f*)(Yg + (unsigned)(row * 240 + q * 4)) = o;
        }
    }

    auto run_unit = [&](int kind, int mt, int nt) {
        floatx16 u;
        const float* bsrc = Bs + (kind == 1 ? 0 : kind == 2 ? 64 : 128) + nt * 32 + h4;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 t = *reinterpret_cast<const float4*>(bsrc + 8 * g);
            u[4 * g] = t.x; u[4 * g + 1] = t.y; u[4 * g + 2] = t.z; u[4 * g + 3] = t.w;
        }
        const int row = mt * 32 + rlane;
        const int rm = rowmap[row];
        if (kind == 1) {          // 1x3, 32 -> 64, ReLU, to T2                            layers.py:127-131
            fused_conv_unit<3>(T1, rm, 64, lane, pf, u);
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(T2 + row * F_LD2 + nt * 32 + 8 * g + h4) =
                    make_float4(relu_f(u[4 * g]), relu_f(u[4 * g + 1]), relu_f(u[4 * g + 2]), relu_f(u[4 * g + 3]));
        } else {
            // kind 2: 1x3, 32 -> 48, ReLU, to Y[96,144)   layers.py:106-110
            // kind 3: 1x5, 32 -> 48, ReLU, to Y[144,192)  layers.py:115-119
            if (kind == 2) fused_conv_unit<3>(T1, rm, 0, lane, pf, u);
            else fused_conv_unit<5>(T1, rm, 32, lane, pf, u);
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
        fused_unit_prefetch(unit_Bp(a2k), unit_taps(a2k), a2n, lane, pf);
        run_unit(a2k, a2m, a2n);
    }
    // weights of the first P2b job are requested before the barrier
    float4 b5c[8];
#pragma unroll
    for (int g = 0; g < 8; ++g) b5c[g] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (wave < 2) {
#pragma unroll
        for (int g = 0; g < 8; ++g) b5c[g] = gload4(a.Bp5c + ((size_t)(wave * 8 + g) * 64 + lane) * 4);
    } else if (b1k) {
        fused_unit_prefetch(unit_Bp(b1k), unit_taps(b1k), b1n, lane, pf);
    }
    DS_STAMP(5);
    __syncthreads();   // T2 complete
    DS_STAMP(6);

    // ---- P2b
    if (wave < 2) {
        // branch 5 tail: 1x1 64 -> 48 (BN, no ReLU) accumulated on top of the stem conv held in acc,
        // then relu(stem + tail)                                                         layers.py:132-138
#pragma unroll
        for (int mt = 0; mt < TM; ++mt) {
            const float* base = T2 + (mt * 32 + rlane) * F_LD2 + h4;
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const float4 av = *reinterpret_cast<const float4*>(base + g * 8);
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(b5c[g].x, av.x, acc[mt], 0, 0, 0);
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(b5c[g].y, av.y, acc[mt], 0, 0, 0);
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(b5c[g].z, av.z, acc[mt], 0, 0, 0);
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(b5c[g].w, av.w, acc[mt], 0, 0, 0);
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
            fused_unit_prefetch(unit_Bp(b2k), unit_taps(b2k), b2n, lane, pf);
            run_unit(b2k, b2m, b2n);
        }
    }
    DS_STAMP(7);
#undef DS_STAMP
    // the next module reads the rows this one has stored (other waves' stores included) and re-uses every LDS region
    if (mi + 1 < c.nmod) __syncthreads();
    }   // modules of the chain
}

template <int TM>
__global__ __launch_bounds__(512, DS_FUSED_WPS) void inception_fused_kernel(const FusedChain c) { inception_fused_body<TM>(c); }
// one tiling for the whole chain; only its first module may pool its input (ds_internal.h FusedChain)
static bool fused_chain_ok(const FusedChain& c)
{
    if (c.nmod <= 0 || c.nmod > FUSED_CHAIN_MAX) return false;
    for (int i = 1; i < c.nmod; ++i)
        if (c.m[i].W != c.m[0].W || c.m[i].spt != c.m[0].spt || c.m[i].n_sites != c.m[0].n_sites || c.m[i].pool_win != 0 ||
            c.m[i].X != c.m[i - 1].Y) return false;
    return true;
}

hipError_t launch_inception_fused(int tm, const FusedChain& c, hipStream_t s)
{
    if (!fused_chain_ok(c)) return hipErrorInvalidValue;
    const FusedArgs& a = c.m[0];
    if (a.n_sites <= 0) return hipSuccess;
    for (int i = 0; i < c.nmod; ++i)
        if (c.m[i].cin != 240 && c.m[i].cin != 256) return hipErrorInvalidValue;      // P1 is unrolled over 15 or 16 chunks (every module of the model)
    const size_t lds = inception_fused_lds_bytes(tm, a.W, a.spt);
    const int grid = (a.n_sites + a.spt - 1) / a.spt;
    switch (tm) {
    case 1: hipLaunchKernelGGL(inception_fused_kernel<1>, dim3(grid), dim3(512), lds, s, c); break;
    case 2: hipLaunchKernelGGL(inception_fused_kernel<2>, dim3(grid), dim3(512), lds, s, c); break;
    case 3: hipLaunchKernelGGL(inception_fused_kernel<3>, dim3(grid), dim3(512), lds, s, c); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Fused inception modules, bf16 operands (DS_PRECISION_BF16*). Same phases and wave roles as
// inception_fused_kernel; what changes is the data: activations are bf16 rows of 256-channel pitch (channels
// 240..255 are zero in global memory), a K chunk is 16 four-byte units = 32 channels = two v_mfma_f32_32x32x16_bf16 steps,
// the 32-channel intermediates live in LDS as bf16 (T1 row = 104 elements: an odd number of 16-B slots, conflict-free
// ds_read_b128), accumulation / bias / ReLU / residual are fp32 and every value is rounded to bf16 (nearest even) exactly
// once, when it is stored.
//
// Roofline: HBM (module-granular bytes: 512 B in + 480 B out per module row). A 96-row tile is ~2 us of matrix-pipe time, so
// what matters is the memory traffic and how much of it a CU keeps in flight. Round 3 form:
//   * TWO workgroups per CU (<= 128 VGPRs, <= 76 KB of LDS each): one tile's load / store phases run under the other's MFMAs;
//   * the 3-tap max-pooled copy of the input tile (branch 1's operand, 50 KB) is gone: waves 6, 7 pool their fragments
//     on the fly (three ds_read_b128 + v_pk_max_i16 per fragment), as the fp32 kernel does;
//   * the P1 weights stream through a ring of two fragments per wave instead of sixteen resident ones;
//   * a launch carries the modules of one width class and the tile's rows STAY IN LDS from module to module: every result is
//     written in place over the (dead) input rows in the layout the next module's P1 reads. Only the chain's first module
//     loads rows and only its last one stores them (whole 480-byte rows): eight of the eleven module outputs never reach
//     HBM. Bisect builds had priced the parts of the 755 us the global-memory chain took at 4096 sites: re-loading the rows
//     55 us, issuing their stores 30 us, the write traffic itself 65 us; this form takes 620 us.
#ifndef DS_FUSEDB_WPS
#define DS_FUSEDB_WPS 4
#endif
// input rows are read exactly once: a non-temporal load does not keep them in the XCD's L2, which then holds the rows the
// workgroups have just WRITTEN (the next module of the chain reads those)
#ifndef DS_FUSEDB_NT
#define DS_FUSEDB_NT 1
#endif
#if DS_FUSEDB_NT
#define DS_FUSEDB_TILE_LOAD(p) gload4_nt(p)
#else
#define DS_FUSEDB_TILE_LOAD(p) gload4(p)
#endif
#ifndef DS_FUSEDB_POOLPRIO
#define DS_FUSEDB_POOLPRIO 2
#endif
#ifndef DS_FUSEDB_RING
#define DS_FUSEDB_RING 2      // register stages of a wave's P1 weight fragments (2, 3: same time on MI355X; 3 and 4 spill at 128 VGPRs)
#endif
constexpr int B_LDA = 132;      // staged input row stride in units (256 channels + 8 pad: 33 x 16 B, odd)
constexpr int B_LD1 = 52;       // T1 row stride in units (96 channels + 8 pad)

size_t inception_fused_bf16_lds_bytes(int tm, int W, int spt)
{
    const int tr32 = tm * 32;
    return (size_t)(tr32 * B_LDA + (spt * (W + 4) + 5) * B_LD1 + tr32 + 192) * sizeof(float);    // the tile's rows (input, then output in place; T2 in their last 64 channels) | T1 | rowmap | 3x64 biases
}

// weights of one conv unit: ntaps x 2 k-steps (<= 10 fragments); the packed panel of an n-tile holds KS k-steps
__device__ __forceinline__ void fusedb_unit_prefetch(const float* __restrict__ Bp, int ntaps, int nt, int lane, float4 (&ub)[10])
{
    const int ks = (ntaps * 32 + 63) / 64 * 4;      // K padded to 64 elements by pack_b_bf16
    const float* bsrc = Bp + ((size_t)(nt * ks) * 64 + lane) * 4;
#pragma unroll
    for (int g = 0; g < 10; ++g)
        if (g < ntaps * 2) ub[g] = gload4(bsrc + g * 256);
}

template <int NTAPS>
__device__ __forceinline__ void fusedb_conv_unit(const float* T1, int rm, int coloff_units, int lane, const float4 (&ub)[10], floatx16& acc)
{
    const float* base = T1 + rm * B_LD1 + coloff_units + (lane >> 5) * 4;
#pragma unroll
    for (int t = 0; t < NTAPS; ++t) {
        const float* arow = base + (t - NTAPS / 2) * B_LD1;
#pragma unroll
        for (int j = 0; j < 2; ++j)
            acc = mfma_bf(ub[t * 2 + j], *reinterpret_cast<const float4*>(arow + j * 8), acc);     // transposed: see the kernel
    }
}

// the same unit on NM m-tiles at once (one set of weights, NM independent accumulator chains: a single unit is a chain of
// dependent MFMAs behind dependent LDS reads, i.e. latency)
template <int NTAPS, int NM>
__device__ __forceinline__ void fusedb_conv_units(const float* T1, const int (&rm)[NM], int coloff_units, int lane, const float4 (&ub)[10], floatx16 (&acc)[NM])
{
#pragma unroll
    for (int t = 0; t < NTAPS; ++t)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int m = 0; m < NM; ++m)
                acc[m] = mfma_bf(ub[t * 2 + j], *reinterpret_cast<const float4*>(T1 + (rm[m] + t - NTAPS / 2) * B_LD1 + coloff_units + (lane >> 5) * 4 + j * 8), acc[m]);
}

template <int TM>
__global__ __launch_bounds__(512, TM >= 2 ? DS_FUSEDB_WPS : 2) void inception_fused_bf16_kernel(const FusedChain c)
{
    constexpr int TR32 = TM * 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // [TR32][B_LDA] the tile's rows, 256 channels each. A module's INPUT during its P1; behind P1 the same rows receive the
    // module's OUTPUT in place (b1 | b2 | b3 | b4 | b5 = channels [0, 240)), which is the next module's input: inside a chain
    // the rows never leave the CU. Channels [192, 256) of a row hold branch 5's 64-channel intermediate (T2) between P2a and
    // the tail; what T2 leaves in the pad channels [240, 256) is finite and meets zero weight rows in the next module's P1.
    float* const As = smem;
    float* const T1 = smem + TR32 * B_LDA;          // [spt*(W+4)+5][B_LD1] b3a | b4a | b5a with zero halo rows
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int W = c.m[0].W, spt = c.m[0].spt, cinu = c.m[0].cin;          // cin in units (128); the same for every module of a chain
    int* const rowmap = reinterpret_cast<int*>(T1 + (spt * (W + 4) + 5) * B_LD1);
    float* const Bs = reinterpret_cast<float*>(rowmap + TR32);                    // [3][64] biases of b5b | b3b | b4b
    unsigned short* const T1h = reinterpret_cast<unsigned short*>(T1);
    unsigned short* const Ash = reinterpret_cast<unsigned short*>(As);
    // workgroup barrier for LDS traffic only: __syncthreads() also drains vmcnt, i.e. waits for prefetched weights and for the
    // row stores that are meant to leave in the background
    auto lds_barrier = []() __attribute__((always_inline)) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    constexpr int NSLOT = TR32 * 32 / 512;    // 16-B slots of the input tile per thread: 2 * TM

    const int site0 = blockIdx.x * spt;
    const int TRv = min(spt, c.m[0].n_sites - site0) * W;          // valid rows of this tile
    typedef __attribute__((address_space(1))) unsigned short* gbf16w;

    // relu + round four consecutive channels to bf16 -> one 8-byte group
    auto pack4 = [](float x0, float x1, float x2, float x3) -> uint2 {
        return make_uint2(pack_bf2(relu_f(x0), relu_f(x1)), pack_bf2(relu_f(x2), relu_f(x3)));
    };

    // static wave -> unit assignment of P2 (wave-uniform). kind: 0 none, 1 b5b, 2 b3b, 3 b4b; with three m-tiles waves 6, 7 run
    // branch 3's n-tile on all of them in phase a (one set of weights)
    int ak = 0, am = 0, an = 0;
    int bk = 0, bm = 0, bn = 0;
    if (TM == 3) {
        if (wave < 6) { ak = 1; am = wave % 3; an = wave / 3; }
        else { ak = 2; am = 0; an = wave - 6; }
        if (wave >= 2) { bk = 3; bm = (wave - 2) % 3; bn = (wave - 2) / 3; }
    } else if (TM == 2) {
        if (wave < 4) { ak = 1; am = wave & 1; an = wave >> 1; }
        else { ak = 2; am = wave & 1; an = (wave - 4) >> 1; }
        if (wave >= 2 && wave < 6) { bk = 3; bm = (wave - 2) & 1; bn = (wave - 2) >> 1; }
    } else {
        if (wave < 2) { ak = 1; an = wave; }
        else if (wave < 4) { ak = 2; an = wave - 2; }
        else if (wave < 6) { ak = 3; an = wave - 4; }
    }

    // once per workgroup: the tile's row map and the zero halo rows of T1 (every module of the chain rewrites T1's interior
    // rows completely and never touches the halos)
    for (int i = tid; i < (spt * (W + 4) + 5) * (B_LD1 / 4); i += 512)
        reinterpret_cast<float4*>(T1)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < TR32)
        rowmap[tid] = tid < TRv ? (tid / W) * (W + 4) + 2 + tid % W : spt * (W + 4) + 2;

    // ---- the modules of the chain, one after the other on THIS tile. Only the first module reads rows from global memory
    // and only the last one writes rows back: in between a module's output rows are the next module's input, in LDS.
    for (int mi = 0; mi < c.nmod; ++mi) {
    const FusedArgs& a = c.m[mi];
    // Every per-lane quantity of the body derives from this opaque copy of the thread index: hipcc otherwise hoists ~100
    // loop-invariant per-lane addresses out of the module loop and spills them (the kernel lives on 128 VGPRs).
    int tid_opaque = threadIdx.x, wave_opaque = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    asm volatile("" : "+v"(tid_opaque), "+s"(wave_opaque));
    const int tid = tid_opaque, lane = tid & 63, h4 = 4 * (lane >> 5), rlane = lane & 31, wave = wave_opaque;
    const bool stamp = a.dbg != nullptr && lane == 0 && (wave == 0 || wave == 7) && blockIdx.x < DBG_MAX_WGS;
    // (the destination is recomputed at every stamp: a pointer held in VGPRs for the whole kernel costs two of the 128)
#define DS_STAMP(i) do { if (stamp) (a.dbg + ((size_t)blockIdx.x * 2 + (wave == 7)) * 8)[i] = __builtin_amdgcn_s_memtime(); } while (0)
    DS_STAMP(0);
    auto unit_Bp = [&](int k) { return k == 1 ? a.Bp5b : k == 2 ? a.Bp3b : a.Bp4b; };
    auto unit_taps = [&](int k) { return k == 3 ? 5 : 3; };

    // the small global requests of a module's prologue, all issued before anything waits: the unit biases, this wave's P1 bias
    // (+ the tail's BN shift on the b5 stem columns)
    float bsv;
    float4 bv[4], tv[4];
    auto request_biases = [&]() __attribute__((always_inline)) {
        bsv = tid < 192 ? gload((tid < 64 ? a.bias5b : tid < 128 ? a.bias3b : a.bias4b) + (tid & 63)) : 0.0f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            bv[g] = gload4(a.bias1 + wave * 32 + 8 * g + h4);                               // bias1 is zero-padded to 256
            tv[g] = gload4(a.bias5c + (wave < 2 ? wave : 0) * 32 + 8 * g + h4);             // zero-padded to 64; used by waves 0, 1 only
        }
    };
    if (mi == 0) {
        // ---- chain head: the WHOLE input tile is requested at once (a bf16 K chunk is ~200 matrix-pipe cycles: nothing
        // hides behind it; the other workgroup of the CU computes meanwhile)
        const bool pooled_in = a.pool_win > 0;
        float4 st[NSLOT];
        const size_t grow0 = (size_t)site0 * W;
        if (!pooled_in) {
#pragma unroll
            for (int i = 0; i < NSLOT; ++i) {
                const int id = tid + i * 512, row = id >> 5, q = id & 31;
                const int rr = row < TRv ? row : TRv - 1;
                st[i] = DS_FUSEDB_TILE_LOAD(a.X + (grow0 + rr) * cinu + q * 4);      // read once: see DS_FUSEDB_NT
            }
        } else {
            // module right after maxpool_layer2/3: staged row (site s, w) = max of the input rows 2w - pad + {0,1,2} that exist
#pragma unroll
            for (int i = 0; i < NSLOT; ++i) {
                const int id = tid + i * 512, row = id >> 5, q = id & 31;
                const int rr = row < TRv ? row : TRv - 1;
                const int s_ = rr / W, w_ = rr % W;
                const int i0 = 2 * w_ - a.pool_pad;
                const int ia = i0 < 0 ? i0 + 1 : i0;
                const int ib = i0 + 1 < a.pool_win ? (i0 + 1 < 0 ? ia : i0 + 1) : ia;
                const int ic = i0 + 2 < a.pool_win ? i0 + 2 : ib;
                const float* sb = a.X + ((size_t)(site0 + s_) * a.pool_win) * cinu + q * 4;
                st[i] = bf8max_nn(bf8max_nn(gload4(sb + (size_t)ia * cinu), gload4(sb + (size_t)ib * cinu)), gload4(sb + (size_t)ic * cinu));
            }
        }
        request_biases();
#pragma unroll
        for (int i = 0; i < NSLOT; ++i) {
            const int id = tid + i * 512, row = id >> 5, q = id & 31;
            *reinterpret_cast<float4*>(As + row * B_LDA + q * 4) = st[i];
        }
    } else {
        request_biases();
    }
    if (tid < 192) Bs[tid] = bsv;        // read behind the barriers of P1; the previous module's last use is behind its barriers
    // P1 weights of this wave's n-tile (16 fragments, L2-resident) through a ring of DS_FUSEDB_RING
    // (wave-uniform base + one 32-bit lane offset: scalar address arithmetic, no 64-bit per-lane pointers in the loop)
    const char* const bp_base = reinterpret_cast<const char*>(a.Bp1) + (size_t)wave * ((cinu + 31) / 32 * 4) * 1024;
    const unsigned lane16 = (unsigned)lane * 16;
    auto bfrag = [&](int g) __attribute__((always_inline)) { return gload4(reinterpret_cast<const float*>(bp_base + g * 1024 + lane16)); };
    float4 bw[DS_FUSEDB_RING];
#pragma unroll
    for (int g = 0; g < DS_FUSEDB_RING; ++g) bw[g] = bfrag(g);
    // Every MFMA of this kernel is issued TRANSPOSED: mfma(weight fragment, activation fragment) gives (X W)^T, so
    // a lane ends up holding, for ONE activation row (lane & 31), 4 x 4 consecutive output channels
    // (register 4g+e <-> channel 32*ntile + 8g + 4*(lane >> 5) + e). Results leave as packed 8-byte groups of four
    // bf16 instead of 2-byte scalars, and the bias is simply the accumulator's initial value.
    floatx16 acc[TM];
    {
        const float tsel = wave < 2 ? 1.0f : 0.0f;      // b5 stem columns also carry the tail's BN shift (exact: x + 1 * t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 b = make_float4(fmaf(tsel, tv[g].x, bv[g].x), fmaf(tsel, tv[g].y, bv[g].y), fmaf(tsel, tv[g].z, bv[g].z),
                                         fmaf(tsel, tv[g].w, bv[g].w));
#pragma unroll
            for (int mt = 0; mt < TM; ++mt) {
                acc[mt][4 * g + 0] = b.x; acc[mt][4 * g + 1] = b.y;
                acc[mt][4 * g + 2] = b.z; acc[mt][4 * g + 3] = b.w;
            }
        }
    }
    if (mi == 0) lds_barrier();   // input tile, zeroed T1, rowmap in place (later modules: behind the previous module's last barrier)
    DS_STAMP(1);

    // ---- P1: [rows x 256] x [256 x 256], 16 k-steps back to back out of LDS. Waves 6, 7
    // (branch 1) take the 3-tap max of their fragment and its two neighbour rows -- maxpool(3, stride 1, SAME), a missing
    // neighbour at a site edge = the own row ("padded taps ignored")        layers.py:90-91
    auto run_p1 = [&](auto pool_tag) __attribute__((always_inline)) {
        constexpr bool POOL = decltype(pool_tag)::value;
        const float* const src = As + rlane * B_LDA + h4;
        if (!POOL) {
            // ONE fragment buffer: the reads of k-step g + 1 are issued right behind the MFMAs of k-step g (which have
            // taken their operands by then) and land while those run -- the look-ahead of a double buffer without its registers
            float4 af[TM];
#pragma unroll
            for (int mt = 0; mt < TM; ++mt) af[mt] = *reinterpret_cast<const float4*>(src + mt * 32 * B_LDA);
#pragma unroll
            for (int g = 0; g < 16; ++g) {
#pragma unroll
                for (int mt = 0; mt < TM; ++mt) acc[mt] = mfma_bf(bw[g % DS_FUSEDB_RING], af[mt], acc[mt]);
                if (g + DS_FUSEDB_RING < 16) bw[g % DS_FUSEDB_RING] = bfrag(g + DS_FUSEDB_RING);
                if (g + 1 < 16) {
#pragma unroll
                    for (int mt = 0; mt < TM; ++mt) af[mt] = *reinterpret_cast<const float4*>(src + mt * 32 * B_LDA + (g + 1) * 8);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            // pooled operand: raw[mt] = (own, previous, next) row fragments of k-step g, requested one k-step ahead; per
            // m-tile: 3-tap max (8 x v_pk_max_i16) -> MFMA -> request the next k-step's three fragments into the registers
            // just consumed. The max of m-tile mt + 1 issues under the MFMA of m-tile mt.
            int om[TM], op[TM];
#pragma unroll
            for (int mt = 0; mt < TM; ++mt) {
                const int row = mt * 32 + rlane, w = row % W;
                om[mt] = (row < TRv && w > 0) ? -B_LDA : 0;
                op[mt] = (row < TRv && w < W - 1) ? B_LDA : 0;
            }
            float4 raw[TM][3];
#pragma unroll
            for (int mt = 0; mt < TM; ++mt) {
                const float* q = src + mt * 32 * B_LDA;
                raw[mt][0] = *reinterpret_cast<const float4*>(q);
                raw[mt][1] = *reinterpret_cast<const float4*>(q + om[mt]);
                raw[mt][2] = *reinterpret_cast<const float4*>(q + op[mt]);
            }
#pragma unroll
            for (int g = 0; g < 16; ++g) {
#pragma unroll
                for (int mt = 0; mt < TM; ++mt) {
                    const float4 af = bf8max_nn(bf8max_nn(raw[mt][0], raw[mt][1]), raw[mt][2]);
                    if (g + 1 < 16) {
                        const float* q = src + mt * 32 * B_LDA + (g + 1) * 8;
                        raw[mt][0] = *reinterpret_cast<const float4*>(q);
                        raw[mt][1] = *reinterpret_cast<const float4*>(q + om[mt]);
                        raw[mt][2] = *reinterpret_cast<const float4*>(q + op[mt]);
                    }
                    acc[mt] = mfma_bf(bw[g % DS_FUSEDB_RING], af, acc[mt]);
                }
                if (g + DS_FUSEDB_RING < 16) bw[g % DS_FUSEDB_RING] = bfrag(g + DS_FUSEDB_RING);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    // the two pooling waves carry ~2x the instructions of the others per k-step and everybody waits for them at the barrier
    // below: they get the SIMD's issue priority while they are in P1
    if (wave >= 6) { __builtin_amdgcn_s_setprio(DS_FUSEDB_POOLPRIO); run_p1(FusedTagT{}); __builtin_amdgcn_s_setprio(0); }
    else run_p1(FusedTagF{});     // wave-uniform
    DS_STAMP(2);
    lds_barrier();   // all fragment reads of the input rows are done: the output may overwrite them
    float4 pf[10];
    // (fragments 6..9 are only loaded for the five-tap unit; defined here so that their live range starts here and not, as an
    // "undefined on some paths" value, in front of the module loop -- sixteen registers held across P1 otherwise)
#pragma unroll
    for (int g = 6; g < 10; ++g) pf[g] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ak) fusedb_unit_prefetch(unit_Bp(ak), unit_taps(ak), an, lane, pf);

    // ---- P1 epilogue (bias already inside acc)
    if (wave >= 3 && wave <= 5) {            // b3a | b4a | b5a -> T1 (bf16), through the row map (SAME-padding halos)
#pragma unroll
        for (int mt = 0; mt < TM; ++mt) {
            const int rm = rowmap[mt * 32 + rlane];
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<uint2*>(T1h + rm * (2 * B_LD1) + (wave * 32 - 96) + 8 * g + h4) =
                    pack4(acc[mt][4 * g], acc[mt][4 * g + 1], acc[mt][4 * g + 2], acc[mt][4 * g + 3]);
        }
    } else if (wave != 0) {                  // b2 -> channels [48, 96), b1 -> channels [0, 48) of the tile's rows (b5 stem | padding stay behind)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int col = wave * 32 + 8 * g + h4;          // groups of 4 never straddle 48 / 240
            if (col >= 48 && col < 240) {
                const int ycol = col < 96 ? col : col - 192;
#pragma unroll
                for (int mt = 0; mt < TM; ++mt)
                    *reinterpret_cast<uint2*>(Ash + (mt * 32 + rlane) * (2 * B_LDA) + ycol) =
                        pack4(acc[mt][4 * g], acc[mt][4 * g + 1], acc[mt][4 * g + 2], acc[mt][4 * g + 3]);
            }
        }
    }
    lds_barrier();   // T1 complete
    DS_STAMP(3);

    // unit (kind, m-tile, n-tile) of the second-stage convs: 1 = 1x3 32 -> 64 of branch 5 (-> T2 = channels [192, 256) of the
    // rows, layers.py:127-131), 2 = 1x3 32 -> 48 of branch 3 (channels [96, 144), layers.py:106-110), 3 = 1x5 32 -> 48 of
    // branch 4 (channels [144, 192), layers.py:115-119)
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
        if (kind == 1) fusedb_conv_unit<3>(T1, rm, 32, lane, pf, u);          // b5a = channels 64..95 = units 32..47
        else if (kind == 2) fusedb_conv_unit<3>(T1, rm, 0, lane, pf, u);      // b3a = channels 0..31
        else fusedb_conv_unit<5>(T1, rm, 16, lane, pf, u);                    // b4a = channels 32..63
        const int cbase = kind == 1 ? 192 : kind == 2 ? 96 : 144;
        const int ncol = kind == 1 ? 64 : 48;                                 // wave-uniform: 48 output channels = n-tile 0 and half of n-tile 1
#pragma unroll
        for (int g = 0; g < 4; ++g)
            if (nt * 32 + 8 * g < ncol)
                *reinterpret_cast<uint2*>(Ash + row * (2 * B_LDA) + cbase + nt * 32 + 8 * g + h4) = pack4(u[4 * g], u[4 * g + 1], u[4 * g + 2], u[4 * g + 3]);
    };

    // ---- P2a: branch 5's 1x3 units -> T2, branch 3 (/ 4) units -> their output channels
    if (TM == 3 && wave >= 6) {
        // branch 3's n-tile on all three m-tiles (same weights): m-tiles 0, 1 interleaved, then m-tile 2 (one after the other
        // the three units took 4.4 k cycles against 2.4 k for the single unit of the other waves: a unit is a chain of
        // dependent LDS reads and MFMAs; three accumulators at once do not fit the 128 registers next to the stem's)
        {
            floatx16 u2[2];
            const float* bsrc = Bs + 64 + an * 32 + h4;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 t = *reinterpret_cast<const float4*>(bsrc + 8 * g);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) { u2[mt][4 * g] = t.x; u2[mt][4 * g + 1] = t.y; u2[mt][4 * g + 2] = t.z; u2[mt][4 * g + 3] = t.w; }
            }
            int rm[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) rm[mt] = rowmap[mt * 32 + rlane];
            fusedb_conv_units<3, 2>(T1, rm, 0, lane, pf, u2);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    if (an * 32 + 8 * g < 48)
                        *reinterpret_cast<uint2*>(Ash + (mt * 32 + rlane) * (2 * B_LDA) + 96 + an * 32 + 8 * g + h4) =
                            pack4(u2[mt][4 * g], u2[mt][4 * g + 1], u2[mt][4 * g + 2], u2[mt][4 * g + 3]);
        }
        run_unit(2, 2, an);
    } else if (ak) {
        run_unit(ak, am, an);
    }
    if (wave < 2) {       // the tail's four weight fragments travel in the (idle) unit-weight registers
#pragma unroll
        for (int g = 0; g < 4; ++g) pf[g] = gload4(a.Bp5c + ((size_t)(wave * 4 + g) * 64 + lane) * 4);
    } else if (bk) {
        fusedb_unit_prefetch(unit_Bp(bk), unit_taps(bk), bn, lane, pf);
    }
    DS_STAMP(4);
    lds_barrier();     // T2 complete
    DS_STAMP(5);

    // ---- P2b: branch 4's 1x5 units; waves 0, 1: branch 5's tail
    if (wave < 2) {
        // 1x1 64 -> 48 (BN, no ReLU) accumulated on top of the stem conv held in acc, then relu(stem + tail)   layers.py:132-138
#pragma unroll
        for (int mt = 0; mt < TM; ++mt) {
            const float* base = As + (mt * 32 + rlane) * B_LDA + 96 + h4;
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[mt] = mfma_bf(pf[g], *reinterpret_cast<const float4*>(base + g * 8), acc[mt]);
        }
    } else if (bk) {
        run_unit(bk, bm, bn);
    }
    DS_STAMP(6);
    lds_barrier();     // every T2 fragment has been read: branch 5's output may take its place
    if (wave < 2) {
#pragma unroll
        for (int mt = 0; mt < TM; ++mt)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                if (wave * 32 + 8 * g < 48)
                    *reinterpret_cast<uint2*>(Ash + (mt * 32 + rlane) * (2 * B_LDA) + 192 + wave * 32 + 8 * g + h4) =
                        pack4(acc[mt][4 * g], acc[mt][4 * g + 1], acc[mt][4 * g + 2], acc[mt][4 * g + 3]);
    }
    lds_barrier();     // the module's output rows are complete: the next module's input
    if (a.write_rows) {
        // ---- the chain's last module (every module when taps are on): rows out, 480 contiguous bytes each (30 x 16 B)
        const gbf16w Yg = (gbf16w)(reinterpret_cast<unsigned short*>(a.Y) + (size_t)__builtin_amdgcn_readfirstlane(site0) * W * 256);
        for (int idx = tid; idx < TR32 * 30; idx += 512) {
            const int row = idx / 30, q = idx - row * 30;
            if (row < TRv) {
                const float4 v = *reinterpret_cast<const float4*>(As + row * B_LDA + q * 4);
                v4f o = {v.x, v.y, v.z, v.w};
                *(__attribute__((address_space(1))) v4f*)(Yg + (unsigned)(row * 256 + q * 8)) = o;
            }
        }
    }
    DS_STAMP(7);
#undef DS_STAMP
    }   // modules of the chain
}

// ---------------------------------------------------------------------------------------------
// conv_layer2 (1x1, 64 -> 128) + conv_layer3 (1x3, 128 -> 256), BN folded, ReLU            layers.py:192-203
// One workgroup (8 waves) owns a tile of whole sites (<= 96 rows), two workgroups per CU (<= 128 VGPRs, ~76 KB of LDS),
// so one workgroup's staging / epilogue runs under the other's MFMAs.
//   stage  the 64-channel input rows (stem_pool) -> Xs
//   conv2  12 units of (32 rows x 32 channels, K = 64), three per SIMD -> T (ReLU) with one zero halo row on either
//          side of every site = conv3's SAME padding
//   conv3  wave w = output channels [32 w, 32 w + 32) of all three m-tiles. The WHOLE activation tile is in LDS, so the
//          48 k-groups (3 taps x 16) run without a barrier: weights global -> VGPR through a ring of four register
//          stages, activation fragments from T one k-group ahead. Every LDS / global offset is an immediate on a
//          loop-invariant base (fully unrolled): no vector ALU work between the MFMAs (tools/attic/mfma_valu.hip).
// MFMAs are issued transposed (weights, activations) as in the fused module: a lane holds 4 x 4 consecutive channels of
// one row, bias = accumulator init, float4 stores.
// Roofline: MFMA. Algorithmic FLOPs per site = 2 * W * (64 * 128 + 3 * 128 * 256); HBM bytes per row 256 in, 1024 out.
constexpr int S23_LDX = 68;     // Xs row stride (floats): 17 x 16 B, odd -> conflict-free b128 fragment reads
constexpr int S23_LDT = 132;    // T row stride: 33 x 16 B
size_t stem23_lds_bytes(int W, int spt)
{
    return (size_t)(96 * S23_LDX + (spt * (W + 2) + 3) * S23_LDT + 96) * sizeof(float);
}

__global__ __launch_bounds__(512, 2) void stem23_kernel(const Stem23Args a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const Xs = smem;                                   // [96][S23_LDX]
    float* const T = smem + 96 * S23_LDX;                     // [spt * (W + 2) + 3][S23_LDT]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int W = a.W, spt = a.spt;
    const int trows = spt * (W + 2) + 3;                      // last three rows: halo | dump row (padding rows of the tile) | halo
    int* const rowmap = reinterpret_cast<int*>(T + trows * S23_LDT);     // [96] tile row -> T row
    const int site0 = blockIdx.x * spt;
    const int nhere = min(spt, a.n_sites - site0);
    const int TRv = nhere * W;
    const size_t grow0 = (size_t)site0 * W;
    const int h4 = 4 * (lane >> 5), rlane = lane & 31;

    // ---- stage: zero T (halo rows must be zero; the rest is overwritten), row map, input rows
    for (int i = tid; i < trows * (S23_LDT / 4); i += 512) reinterpret_cast<float4*>(T)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
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
            *reinterpret_cast<float4*>(Xs + row * S23_LDX + q * 4) = v[i];
        }
    }
    // conv3 weights of the first ring stages and both bias vectors are requested before the barrier
    const char* const b3 = reinterpret_cast<const char*>(a.Bp3) + (size_t)wave * 48 * 1024;
    const unsigned lane16 = (unsigned)lane * 16;
    float4 bq[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) bq[i] = gload4(reinterpret_cast<const float*>(b3 + i * 1024 + lane16));
    float4 bias3[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bias3[g] = gload4(a.bias3 + wave * 32 + 8 * g + h4);
    __syncthreads();

    // ---- conv2: unit u = (m, n); waves 0..3 take (0, w) and (2, w), waves 4..7 take (1, w - 4): three units per SIMD
    {
        const int n2 = wave & 3;
        float4 w2[8];
#pragma unroll
        for (int g = 0; g < 8; ++g) w2[g] = gload4(a.Bp2 + ((size_t)(n2 * 8 + g) * 64 + lane) * 4);
        float4 bias2[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) bias2[g] = gload4(a.bias2 + n2 * 32 + 8 * g + h4);
        const int nunits = wave < 4 ? 2 : 1;
        for (int ui = 0; ui < nunits; ++ui) {
            const int m = wave < 4 ? 2 * ui : 1;
            floatx16 u;
#pragma unroll
            for (int g = 0; g < 4; ++g) { u[4 * g] = bias2[g].x; u[4 * g + 1] = bias2[g].y; u[4 * g + 2] = bias2[g].z; u[4 * g + 3] = bias2[g].w; }
            const float* xr = Xs + (m * 32 + rlane) * S23_LDX + h4;
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const float4 x = *reinterpret_cast<const float4*>(xr + g * 8);
                u = __builtin_amdgcn_mfma_f32_32x32x2f32(w2[g].x, x.x, u, 0, 0, 0);
                u = __builtin_amdgcn_mfma_f32_32x32x2f32(w2[g].y, x.y, u, 0, 0, 0);
                u = __builtin_amdgcn_mfma_f32_32x32x2f32(w2[g].z, x.z, u, 0, 0, 0);
                u = __builtin_amdgcn_mfma_f32_32x32x2f32(w2[g].w, x.w, u, 0, 0, 0);
            }
            const int row = m * 32 + rlane;
            float* const td = T + rowmap[row] * S23_LDT + n2 * 32 + h4;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 o = make_float4(relu_f(u[4 * g]), relu_f(u[4 * g + 1]), relu_f(u[4 * g + 2]), relu_f(u[4 * g + 3]));
                *reinterpret_cast<float4*>(td + 8 * g) = o;
                if (a.C2 && row < TRv) {                       // diagnostic tap (debug mode): conv_layer2's output rows
                    const v4f ov = {o.x, o.y, o.z, o.w};
                    *(__attribute__((address_space(1))) v4f*)(a.C2 + (grow0 + row) * 128 + n2 * 32 + 8 * g + h4) = ov;
                }
            }
        }
    }
    __syncthreads();

    // ---- conv3
    floatx16 acc[3];
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int g = 0; g < 4; ++g) { acc[m][4 * g] = bias3[g].x; acc[m][4 * g + 1] = bias3[g].y; acc[m][4 * g + 2] = bias3[g].z; acc[m][4 * g + 3] = bias3[g].w; }
    const float* tb[3];
#pragma unroll
    for (int m = 0; m < 3; ++m) tb[m] = T + (rowmap[m * 32 + rlane] - 1) * S23_LDT + h4;     // tap t reads row + t - 1
    float4 af[2][3];
#pragma unroll
    for (int m = 0; m < 3; ++m) af[0][m] = *reinterpret_cast<const float4*>(tb[m]);
#pragma unroll
    for (int kgl = 0; kgl < 48; ++kgl) {
        const int cur = kgl & 1, slot = kgl & 3;
        if (kgl + 1 < 48) {
            const int t1 = (kgl + 1) >> 4, g1 = (kgl + 1) & 15;
#pragma unroll
            for (int m = 0; m < 3; ++m) af[cur ^ 1][m] = *reinterpret_cast<const float4*>(tb[m] + t1 * S23_LDT + g1 * 8);
        }
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(bq[slot].x, af[cur][m].x, acc[m], 0, 0, 0);
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(bq[slot].y, af[cur][m].y, acc[m], 0, 0, 0);
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(bq[slot].z, af[cur][m].z, acc[m], 0, 0, 0);
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(bq[slot].w, af[cur][m].w, acc[m], 0, 0, 0);
        }
        if (kgl + 4 < 48) bq[slot] = gload4(reinterpret_cast<const float*>(b3 + (kgl + 4) * 1024 + lane16));
        __builtin_amdgcn_sched_barrier(0);       // pins the ring: requests stay four k-groups ahead of their MFMAs
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

hipError_t launch_stem23(const Stem23Args& a, hipStream_t s)
{
    if (a.n_sites <= 0) return hipSuccess;
    const int grid = (a.n_sites + a.spt - 1) / a.spt;
    hipLaunchKernelGGL(stem23_kernel, dim3(grid), dim3(512), stem23_lds_bytes(a.W, a.spt), s, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// conv_layer2 + conv_layer3 with bf16 operands (DS_PRECISION_BF16*): the bf16 form of stem23_kernel.   layers.py:192-203
// Rows are bf16 (64 channels in, 128 in LDS only, 256 out); v_mfma_f32_32x32x16_bf16, fp32 accumulate, bias = accumulator
// init, ReLU and one rounding to bf16 when a value is stored. Two workgroups per CU (<= 128 VGPRs, 52 KB of LDS); the
// 256-channel output rows leave through an LDS tile (over the dead input / conv2 tiles) as whole 512-byte rows.
// Roofline: HBM. Algorithmic bytes per row: 128 in + 512 out (the two-launch GEMM form also wrote and re-read 256 + 768).
constexpr int SB_LDX = 36;      // Xs row stride in 4-byte units: 64 channels + 8 pad (9 x 16 B, odd)
constexpr int SB_LDT = 68;      // T row stride: 128 channels + 8 pad (17 x 16 B)
constexpr int SB_LDO = 132;     // output tile row stride: 256 channels + 8 pad (33 x 16 B)
size_t stem23_bf16_lds_bytes(int W, int spt)
{
    const size_t work = (size_t)96 * SB_LDX + (size_t)(spt * (W + 2) + 3) * SB_LDT;
    const size_t outt = (size_t)96 * SB_LDO;
    return ((work > outt ? work : outt) + 96) * sizeof(float);
}

__global__ __launch_bounds__(512, 4) void stem23_bf16_kernel(const Stem23Args a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const Xs = smem;                                   // [96][SB_LDX]
    float* const T = smem + 96 * SB_LDX;                      // [spt * (W + 2) + 3][SB_LDT]
    float* const Ot = smem;                                   // [96][SB_LDO] output tile, over Xs / T once conv3 has read them
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int W = a.W, spt = a.spt;
    const int trows = spt * (W + 2) + 3;
    const size_t work = (size_t)96 * SB_LDX + (size_t)trows * SB_LDT, outt = (size_t)96 * SB_LDO;
    int* const rowmap = reinterpret_cast<int*>(smem + (work > outt ? work : outt));     // [96] tile row -> T row
    const int site0 = blockIdx.x * spt;
    const int TRv = min(spt, a.n_sites - site0) * W;
    const size_t grow0 = (size_t)site0 * W;
    const int h4 = 4 * (lane >> 5), rlane = lane & 31;
    unsigned short* const Th = reinterpret_cast<unsigned short*>(T);
    unsigned short* const Oh = reinterpret_cast<unsigned short*>(Ot);
    auto pack4 = [](float x0, float x1, float x2, float x3) -> uint2 {
        return make_uint2(pack_bf2(relu_f(x0), relu_f(x1)), pack_bf2(relu_f(x2), relu_f(x3)));
    };

    // ---- stage: zero T (halo rows), row map, input rows (96 rows x 8 slots of 16 B: 768 slots)
    for (int i = tid; i < trows * (SB_LDT / 4); i += 512) reinterpret_cast<float4*>(T)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < 96) rowmap[tid] = tid < TRv ? (tid / W) * (W + 2) + 1 + tid % W : spt * (W + 2) + 1;
    {
        float4 v[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + 512 * i, row = idx >> 3, q = idx & 7;
            const int rr = row < TRv ? row : TRv - 1;
            if (idx < 768) v[i] = gload4(a.X + (grow0 + rr) * 32 + q * 4);      // 64 bf16 = 32 units per row
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + 512 * i, row = idx >> 3, q = idx & 7;
            if (idx < 768) *reinterpret_cast<float4*>(Xs + row * SB_LDX + q * 4) = v[i];
        }
    }
    // conv3 weights of this wave's n-tile: 24 k-steps (3 taps x 8), ring of four; both bias vectors
    const char* const b3 = reinterpret_cast<const char*>(a.Bp3) + (size_t)wave * 24 * 1024;
    const unsigned lane16 = (unsigned)lane * 16;
    float4 bq[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) bq[i] = gload4(reinterpret_cast<const float*>(b3 + i * 1024 + lane16));
    float4 bias3[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bias3[g] = gload4(a.bias3 + wave * 32 + 8 * g + h4);
    __syncthreads();

    // ---- conv2: K = 64 (4 k-steps); unit (m, n): waves 0..3 take (0, w) and (2, w), waves 4..7 take (1, w - 4)
    {
        const int n2 = wave & 3;
        float4 w2[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) w2[g] = gload4(a.Bp2 + ((size_t)(n2 * 4 + g) * 64 + lane) * 4);
        float4 bias2[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) bias2[g] = gload4(a.bias2 + n2 * 32 + 8 * g + h4);
        const int nunits = wave < 4 ? 2 : 1;
        for (int ui = 0; ui < nunits; ++ui) {
            const int m = wave < 4 ? 2 * ui : 1;
            floatx16 u;
#pragma unroll
            for (int g = 0; g < 4; ++g) { u[4 * g] = bias2[g].x; u[4 * g + 1] = bias2[g].y; u[4 * g + 2] = bias2[g].z; u[4 * g + 3] = bias2[g].w; }
            const float* xr = Xs + (m * 32 + rlane) * SB_LDX + h4;
#pragma unroll
            for (int g = 0; g < 4; ++g) u = mfma_bf(w2[g], *reinterpret_cast<const float4*>(xr + g * 8), u);
            const int row = m * 32 + rlane;
            unsigned short* const td = Th + rowmap[row] * (2 * SB_LDT) + n2 * 32 + h4;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const uint2 o = pack4(u[4 * g], u[4 * g + 1], u[4 * g + 2], u[4 * g + 3]);
                *reinterpret_cast<uint2*>(td + 8 * g) = o;
                if (a.C2 && row < TRv) {                       // diagnostic tap (debug mode): conv_layer2's output rows, bf16
                    typedef unsigned int u2v __attribute__((ext_vector_type(2)));
                    const u2v ov = {o.x, o.y};
                    *(__attribute__((address_space(1))) u2v*)(reinterpret_cast<unsigned short*>(a.C2) + (grow0 + row) * 128 + n2 * 32 + 8 * g + h4) = ov;
                }
            }
        }
    }
    __syncthreads();

    // ---- conv3: wave w = output channels [32 w, 32 w + 32) of all three m-tiles; 24 k-steps out of T without a barrier
    floatx16 acc[3];
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int g = 0; g < 4; ++g) { acc[m][4 * g] = bias3[g].x; acc[m][4 * g + 1] = bias3[g].y; acc[m][4 * g + 2] = bias3[g].z; acc[m][4 * g + 3] = bias3[g].w; }
    const float* tb[3];
#pragma unroll
    for (int m = 0; m < 3; ++m) tb[m] = T + (rowmap[m * 32 + rlane] - 1) * SB_LDT + h4;     // tap t reads row + t - 1
    float4 af[3];
#pragma unroll
    for (int m = 0; m < 3; ++m) af[m] = *reinterpret_cast<const float4*>(tb[m]);
#pragma unroll
    for (int ks = 0; ks < 24; ++ks) {
        const int slot = ks & 3;
#pragma unroll
        for (int m = 0; m < 3; ++m) acc[m] = mfma_bf(bq[slot], af[m], acc[m]);
        if (ks + 4 < 24) bq[slot] = gload4(reinterpret_cast<const float*>(b3 + (ks + 4) * 1024 + lane16));
        if (ks + 1 < 24) {
            const int t1 = (ks + 1) >> 3, g1 = (ks + 1) & 7;
#pragma unroll
            for (int m = 0; m < 3; ++m) af[m] = *reinterpret_cast<const float4*>(tb[m] + t1 * SB_LDT + g1 * 8);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();   // everybody has read T: the output tile may overwrite it

    // ---- ReLU, bf16, output tile, whole rows out
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<uint2*>(Oh + (m * 32 + rlane) * (2 * SB_LDO) + wave * 32 + 8 * g + h4) =
                pack4(acc[m][4 * g], acc[m][4 * g + 1], acc[m][4 * g + 2], acc[m][4 * g + 3]);
    __syncthreads();
    __attribute__((address_space(1))) unsigned short* const Yg =
        (__attribute__((address_space(1))) unsigned short*)(reinterpret_cast<unsigned short*>(a.Y) + grow0 * 256);
    for (int idx = tid; idx < 96 * 32; idx += 512) {           // 256 channels = 32 x 16 B per row
        const int row = idx >> 5, q = idx & 31;
        if (row < TRv) {
            const float4 v = *reinterpret_cast<const float4*>(Ot + row * SB_LDO + q * 4);
            v4f o = {v.x, v.y, v.z, v.w};
            *(__attribute__((address_space(1))) v4f*)(Yg + (unsigned)(row * 256 + q * 8)) = o;
        }
    }
}

hipError_t launch_stem23_bf16(const Stem23Args& a, hipStream_t s)
{
    if (a.n_sites <= 0) return hipSuccess;
    const int grid = (a.n_sites + a.spt - 1) / a.spt;
    hipLaunchKernelGGL(stem23_bf16_kernel, dim3(grid), dim3(512), stem23_bf16_lds_bytes(a.W, a.spt), s, a);
    return hipGetLastError();
}

// The fused kernels need more than the default 64 KB of dynamic LDS: opt in once per device (ds_create calls this
// after hipSetDevice; function attributes are per device and must not be changed during stream capture).
hipError_t configure_fused_kernels()
{
    const void* fns[6] = {(const void*)inception_fused_kernel<1>, (const void*)inception_fused_kernel<2>,
                          (const void*)inception_fused_kernel<3>, (const void*)inception_fused_bf16_kernel<1>,
                          (const void*)inception_fused_bf16_kernel<2>, (const void*)inception_fused_bf16_kernel<3>};
    for (const void* f : fns) {
        const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
    }
    const hipError_t e2 = hipFuncSetAttribute((const void*)stem23_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)STEM23_MAX_LDS);
    if (e2 != hipSuccess) return e2;
    return hipFuncSetAttribute((const void*)stem23_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)STEM23_MAX_LDS);
}

hipError_t launch_inception_fused_bf16(int tm, const FusedChain& c, hipStream_t s)
{
    if (!fused_chain_ok(c)) return hipErrorInvalidValue;
    const FusedArgs& a = c.m[0];
    if (a.n_sites <= 0) return hipSuccess;
    for (int i = 1; i < c.nmod; ++i)
        if (c.m[i].cin != a.cin) return hipErrorInvalidValue;
    const size_t lds = inception_fused_bf16_lds_bytes(tm, a.W, a.spt);
    const int grid = (a.n_sites + a.spt - 1) / a.spt;
    switch (tm) {
    case 1: hipLaunchKernelGGL(inception_fused_bf16_kernel<1>, dim3(grid), dim3(512), lds, s, c); break;
    case 2: hipLaunchKernelGGL(inception_fused_bf16_kernel<2>, dim3(grid), dim3(512), lds, s, c); break;
    case 3: hipLaunchKernelGGL(inception_fused_bf16_kernel<3>, dim3(grid), dim3(512), lds, s, c); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// stem conv1: conv(K=7, stride 2, Cin=1 -> 64, SAME) + folded BN + ReLU + maxpool(3, stride 2, SAME)
// One block per site; the window sits in LDS with a zero halo (SAME padding, no bounds checks in the
// tap loop); lane = output channel, so the 64-float output rows are written as whole 256-B lines.
constexpr int STEM_HALO = 8;
template <bool OUT_BF>
__global__ __launch_bounds__(256) void stem1_kernel(const float* __restrict__ signals, const float* __restrict__ w,
                                                     const float* __restrict__ bias, float* __restrict__ out,
                                                     int signal_len, int w1, int pad_l_conv, int wa, int pad_l_pool)
{
    extern __shared__ __attribute__((aligned(16))) float sig[];   // [STEM_HALO + signal_len + STEM_HALO + 8]
    const int site = blockIdx.x;
    const int total = signal_len + 2 * STEM_HALO + 8;
    for (int i = threadIdx.x; i < total; i += blockDim.x) {
        const int k = i - STEM_HALO;
        sig[i] = (k >= 0 && k < signal_len) ? signals[(size_t)site * signal_len + k] : 0.0f;
    }
    __syncthreads();
    const int c = threadIdx.x & 63;
    float wk[7];
#pragma unroll
    for (int t = 0; t < 7; ++t) wk[t] = w[t * 64 + c];
    const float b = bias[c];
    for (int p = threadIdx.x >> 6; p < wa; p += blockDim.x >> 6) {
        // conv positions wc = 2p - pad_l_pool + {0,1,2}; each reads sig[2*wc - pad_l_conv + 0..6]
        const int wc0 = 2 * p - pad_l_pool;
        const float* s0 = sig + STEM_HALO + 2 * wc0 - pad_l_conv;
        float v[11];
#pragma unroll
        for (int i = 0; i < 11; ++i) v[i] = s0[i];
        float best = -INFINITY;
#pragma unroll
        for (int pt = 0; pt < 3; ++pt) {
            float a = 0.0f;
#pragma unroll
            for (int t = 0; t < 7; ++t) a = fmaf(v[2 * pt + t], wk[t], a);
            const int wc = wc0 + pt;
            best = (wc >= 0 && wc < w1) ? fmaxf(best, a) : best;     // padded pool taps are ignored
        }
        const float y = relu_f(best + b);
        if (OUT_BF) reinterpret_cast<unsigned short*>(out)[((size_t)site * wa + p) * 64 + c] = f2bf(y);
        else out[((size_t)site * wa + p) * 64 + c] = y;
    }
}

hipError_t launch_stem1(const float* signals, const float* w7x64, const float* bias64, float* out, int n,
                        int signal_len, int w1, int pad_l_conv, int wa, int pad_l_pool, int out_bf16, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    const size_t lds = (signal_len + 2 * STEM_HALO + 8) * sizeof(float);
    if (out_bf16)
        hipLaunchKernelGGL(stem1_kernel<true>, dim3(n), dim3(256), lds, s, signals, w7x64, bias64, out,
                           signal_len, w1, pad_l_conv, wa, pad_l_pool);
    else
        hipLaunchKernelGGL(stem1_kernel<false>, dim3(n), dim3(256), lds, s, signals, w7x64, bias64, out,
                           signal_len, w1, pad_l_conv, wa, pad_l_pool);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void maxpool_s2_kernel(const float4* __restrict__ in, float4* __restrict__ out,
                                                          long total, int win, int wout, int pad_l, int ch4)
{
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % ch4);
        const long r = i / ch4;
        const int wo = (int)(r % wout);
        const long site = r / wout;
        float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int wi = 2 * wo + t - pad_l;
            if (wi >= 0 && wi < win) m = f4max(m, in[(site * win + wi) * ch4 + c]);
        }
        out[i] = m;
    }
}

hipError_t launch_maxpool_s2(const float* in, float* out, int n, int win, int wout, int pad_l, int ch, hipStream_t s)
{
    const long total = (long)n * wout * (ch / 4);
    if (total <= 0) return hipSuccess;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(maxpool_s2_kernel, dim3(blocks), dim3(256), 0, s, (const float4*)in, (float4*)out, total, win, wout,
                       pad_l, ch / 4);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void avgpool7_kernel(const float4* __restrict__ in, float4* __restrict__ out, long total,
                                                        int w, int ch4)
{
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % ch4);
        const long r = i / ch4;
        const int wo = (int)(r % w);
        const long site = r / w;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        int cnt = 0;
        for (int t = -3; t <= 3; ++t) {
            const int wi = wo + t;
            if (wi < 0 || wi >= w) continue;
            const float4 v = in[(site * w + wi) * ch4 + c];
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
            ++cnt;
        }
        const float d = (float)cnt;
        out[i] = make_float4(a.x / d, a.y / d, a.z / d, a.w / d);
    }
}

hipError_t launch_avgpool7(const float* in, float* out, int n, int w, int ch, hipStream_t s)
{
    const long total = (long)n * w * (ch / 4);
    if (total <= 0) return hipSuccess;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(avgpool7_kernel, dim3(blocks), dim3(256), 0, s, (const float4*)in, (float4*)out, total, w, ch / 4);
    return hipGetLastError();
}

// fc2 + sigmoid + argmax: one 256-thread block per site, float4 sweeps of the fc1 row, block reduce.
__global__ __launch_bounds__(256) void head_kernel(const float* __restrict__ fc1, const float* __restrict__ w2,
                                                    float* __restrict__ logits, float* __restrict__ act,
                                                    int* __restrict__ pred, int n, int J, int C, int nparts, size_t part_stride)
{
    __shared__ float part[4][16];
    const int site = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float4* x4 = reinterpret_cast<const float4*>(fc1 + (size_t)site * J);
    float accv[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) accv[c] = 0.0f;
    const int J4 = J >> 2;
    for (int k4 = tid; k4 < J4; k4 += 256) {
        float4 x = x4[k4];
        for (int p = 1; p < nparts; ++p) {            // fc1 in partial products over ranges of K (the split dense at small forwards): summed in order
            const float4 y = *reinterpret_cast<const float4*>(fc1 + p * part_stride + (size_t)site * J + 4 * (size_t)k4);
            x.x += y.x; x.y += y.y; x.z += y.z; x.w += y.w;
        }
        const float* w = w2 + (size_t)k4 * 4 * C;
        if (C == 2) {
            const float4 wa = *reinterpret_cast<const float4*>(w);
            const float4 wb = *reinterpret_cast<const float4*>(w + 4);
            accv[0] = fmaf(x.x, wa.x, fmaf(x.y, wa.z, fmaf(x.z, wb.x, fmaf(x.w, wb.z, accv[0]))));
            accv[1] = fmaf(x.x, wa.y, fmaf(x.y, wa.w, fmaf(x.z, wb.y, fmaf(x.w, wb.w, accv[1]))));
        } else {
#pragma unroll
            for (int c = 0; c < 16; ++c)
                if (c < C) accv[c] += x.x * w[c] + x.y * w[C + c] + x.z * w[2 * C + c] + x.w * w[3 * C + c];
        }
    }
    for (int k = (J4 << 2) + tid; k < J; k += 256) {    // J not a multiple of 4
        float xk = fc1[(size_t)site * J + k];
        for (int p = 1; p < nparts; ++p) xk += fc1[p * part_stride + (size_t)site * J + k];
        for (int c = 0; c < C; ++c) accv[c] += xk * w2[(size_t)k * C + c];
    }
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        if (c >= C) break;
        float a = accv[c];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off);
        if (lane == 0) part[wave][c] = a;
    }
    __syncthreads();
    if (tid == 0) {
        float best = 0.0f;
        int bi = 0;
        for (int c = 0; c < C; ++c) {
            const float a = (part[0][c] + part[1][c]) + (part[2][c] + part[3][c]);
            const float sg = sigmoidf_(a);
            logits[(size_t)site * C + c] = a;
            act[(size_t)site * C + c] = sg;
            if (c == 0 || sg > best) { best = sg; bi = c; }
        }
        pred[site] = bi;
    }
}

hipError_t launch_head(const float* fc1, const float* w2, float* logits, float* act, int* pred, int n, int J,
                       int class_num, hipStream_t s, int nparts, size_t part_stride)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(head_kernel, dim3(n), dim3(256), 0, s, fc1, w2, logits, act, pred, n, J, class_num, nparts, part_stride);
    return hipGetLastError();
}

// Folded joint model: one workgroup per site, x = concatenation of up to three contiguous row segments.
__global__ __launch_bounds__(256) void head_folded_kernel(const HeadFoldedArgs a)
{
    __shared__ float part[4][16];
    const int site = blockIdx.x, C = a.C;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float accv[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) accv[c] = 0.0f;
    int koff = 0;
    if (a.bf16) {                                         // bf16 modes: bf16 rows, 8 values per 16 bytes
        for (int sg = 0; sg < a.nseg; ++sg) {             // uniform
        const float4* x8 = reinterpret_cast<const float4*>(reinterpret_cast<const unsigned short*>(a.seg[sg]) + (size_t)site * a.pitch[sg]);
        for (int k8 = tid; k8 < (a.len[sg] >> 3); k8 += 256) {
            const float4 raw = x8[k8];
            const unsigned u[4] = {__float_as_uint(raw.x), __float_as_uint(raw.y), __float_as_uint(raw.z), __float_as_uint(raw.w)};
            const float* w = a.w + (size_t)(koff + k8 * 8) * C;
            if (C == 2) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x0 = __uint_as_float(u[e] << 16), x1 = __uint_as_float(u[e] & 0xffff0000u);
                    const float4 wv = *reinterpret_cast<const float4*>(w + 4 * e);      // rows 2e, 2e + 1 of [k][2]
                    accv[0] = fmaf(x0, wv.x, fmaf(x1, wv.z, accv[0]));
                    accv[1] = fmaf(x0, wv.y, fmaf(x1, wv.w, accv[1]));
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x0 = __uint_as_float(u[e] << 16), x1 = __uint_as_float(u[e] & 0xffff0000u);
#pragma unroll
                    for (int c = 0; c < 16; ++c)
                        if (c < C) accv[c] += x0 * w[(2 * e) * C + c] + x1 * w[(2 * e + 1) * C + c];
                }
            }
        }
        koff += a.len[sg];
        }
    } else
    for (int sg = 0; sg < a.nseg; ++sg) {                 // uniform
        const int len = a.len[sg];
        const float4* x4 = reinterpret_cast<const float4*>(a.seg[sg] + (size_t)site * len);
        for (int k4 = tid; k4 < (len >> 2); k4 += 256) {
            const float4 x = x4[k4];
            const float* w = a.w + (size_t)(koff + k4 * 4) * C;
            if (C == 2) {
                const float4 wa = *reinterpret_cast<const float4*>(w);
                const float4 wb = *reinterpret_cast<const float4*>(w + 4);
                accv[0] = fmaf(x.x, wa.x, fmaf(x.y, wa.z, fmaf(x.z, wb.x, fmaf(x.w, wb.z, accv[0]))));
                accv[1] = fmaf(x.x, wa.y, fmaf(x.y, wa.w, fmaf(x.z, wb.y, fmaf(x.w, wb.w, accv[1]))));
            } else {
#pragma unroll
                for (int c = 0; c < 16; ++c)
                    if (c < C) accv[c] += x.x * w[c] + x.y * w[C + c] + x.z * w[2 * C + c] + x.w * w[3 * C + c];
            }
        }
        koff += len;
    }
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        if (c >= C) break;
        float v = accv[c];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if (lane == 0) part[wave][c] = v;
    }
    __syncthreads();
    if (tid == 0) {
        float best = 0.0f;
        int bi = 0;
        for (int c = 0; c < C; ++c) {
            const float v = (part[0][c] + part[1][c]) + (part[2][c] + part[3][c]);
            const float sgm = sigmoidf_(v);
            a.logits[(size_t)site * C + c] = v;
            a.act[(size_t)site * C + c] = sgm;
            if (c == 0 || sgm > best) { best = sgm; bi = c; }
        }
        a.pred[site] = bi;
    }
}

hipError_t launch_head_folded(const HeadFoldedArgs& a, hipStream_t s)
{
    if (a.n <= 0) return hipSuccess;
    hipLaunchKernelGGL(head_folded_kernel, dim3(a.n), dim3(256), 0, s, a);
    return hipGetLastError();
}

// ds_forward_device: the caller's five device arrays into the slot's contiguous input block and the slot's outputs
// back to the caller -- ONE small launch each way instead of five + two copy dispatches per forward (the captured graph
// reads and writes the slot's own buffers).
__global__ __launch_bounds__(256) void gather_inputs_kernel(const int* __restrict__ kmer, const float* __restrict__ means,
                                                            const float* __restrict__ stds, const float* __restrict__ lens,
                                                            const float* __restrict__ signals, float* __restrict__ block,
                                                            int n, int T, int S, int B)
{
    const long nt = (long)n * T, ns = (long)n * S, total = 4 * nt + ns;
    const long bt = (long)B * T;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        if (i < nt) reinterpret_cast<int*>(block)[i] = kmer[i];
        else if (i < 2 * nt) block[bt + (i - nt)] = means[i - nt];
        else if (i < 3 * nt) block[2 * bt + (i - 2 * nt)] = stds[i - 2 * nt];
        else if (i < 4 * nt) block[3 * bt + (i - 3 * nt)] = lens[i - 3 * nt];
        else block[4 * bt + (i - 4 * nt)] = signals[i - 4 * nt];
    }
}

__global__ __launch_bounds__(256) void scatter_outputs_kernel(const float* __restrict__ act, const int* __restrict__ pred,
                                                              float* __restrict__ act_out, int* __restrict__ pred_out, int n, int C)
{
    const int total = n * C + n;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        if (i < n * C) act_out[i] = act[i];
        else pred_out[i - n * C] = pred[i - n * C];
    }
}

hipError_t launch_gather_inputs(const int* kmer, const float* means, const float* stds, const float* lens, const float* signals,
                                float* block, int n, int T, int S, int B, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    const long total = (long)n * (4 * T + S);
    const int grid = (int)std::min<long>((total + 1023) / 1024, 1024);
    hipLaunchKernelGGL(gather_inputs_kernel, dim3(grid), dim3(256), 0, s, kmer, means, stds, lens, signals, block, n, T, S, B);
    return hipGetLastError();
}

hipError_t launch_scatter_outputs(const float* act, const int* pred, float* act_out, int* pred_out, int n, int C, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(scatter_outputs_kernel, dim3((n * (C + 1) + 255) / 256), dim3(256), 0, s, act, pred, act_out, pred_out, n, C);
    return hipGetLastError();
}

// ---- bf16-mode elementwise kernels: 8 channels (16 B) per thread ----
__global__ __launch_bounds__(256) void maxpool_s2_bf16_kernel(const float4* __restrict__ in, float4* __restrict__ out,
                                                               long total, int win, int wout, int pad_l, int ch8)
{
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % ch8);
        const long r = i / ch8;
        const int wo = (int)(r % wout);
        const long site = r / wout;
        const float ninf2 = __uint_as_float(0xff80ff80u);   // two bf16 -inf
        float4 m = make_float4(ninf2, ninf2, ninf2, ninf2);
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int wi = 2 * wo + t - pad_l;
            if (wi >= 0 && wi < win) m = bf8max(m, in[(site * win + wi) * ch8 + c]);
        }
        out[i] = m;
    }
}

hipError_t launch_maxpool_s2_bf16(const float* in, float* out, int n, int win, int wout, int pad_l, int ch_ld, hipStream_t s)
{
    const long total = (long)n * wout * (ch_ld / 8);
    if (total <= 0) return hipSuccess;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(maxpool_s2_bf16_kernel, dim3(blocks), dim3(256), 0, s, (const float4*)in, (float4*)out, total, win,
                       wout, pad_l, ch_ld / 8);
    return hipGetLastError();
}

// avgpool(7, s1, SAME, divisor = valid taps) of the bf16 module output [n][w][ld_in] -> joint[n][out_off + w*ch + c] (bf16)
__global__ __launch_bounds__(256) void avgpool7_bf16_kernel(const float4* __restrict__ in, unsigned short* __restrict__ out,
                                                             long total, int w, int ch8, int ld_in8, long out_ld, int out_off)
{
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % ch8);
        const long r = i / ch8;
        const int wo = (int)(r % w);
        const long site = r / w;
        float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int cnt = 0;
        for (int t = -3; t <= 3; ++t) {
            const int wi = wo + t;
            if (wi < 0 || wi >= w) continue;
            const float4 v = in[(site * w + wi) * ld_in8 + c];
            const unsigned u[4] = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                a[2 * q] += __uint_as_float(u[q] << 16);
                a[2 * q + 1] += __uint_as_float(u[q] & 0xffff0000u);
            }
            ++cnt;
        }
        const float d = (float)cnt;
        unsigned o[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = pack_bf2(a[2 * q] / d, a[2 * q + 1] / d);
        *reinterpret_cast<uint4*>(out + site * out_ld + out_off + ((long)wo * ch8 + c) * 8) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

hipError_t launch_avgpool7_bf16(const float* in, float* joint, int n, int w, int ch, int ld_in, int joint_ld, int joint_off,
                                hipStream_t s)
{
    const long total = (long)n * w * (ch / 8);
    if (total <= 0) return hipSuccess;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(avgpool7_bf16_kernel, dim3(blocks), dim3(256), 0, s, (const float4*)in, (unsigned short*)joint, total, w,
                       ch / 8, ld_in / 8, (long)joint_ld, joint_off);
    return hipGetLastError();
}

// joint[:, 0:256] = bf16(h_fw), joint[:, 256:512] = bf16(h_bw)     (layers.py:171-172, bf16 FC operand)
template <bool SRC_BF>
__global__ __launch_bounds__(256) void pack_event_feat_bf16_kernel(const float* __restrict__ hfw, const float* __restrict__ hbw,
                                                                    unsigned short* __restrict__ joint, int n, long joint_ld)
{
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;      // one thread per 4 values
    if (i >= (long)n * 128) return;
    const int site = (int)(i / 128), q = (int)(i % 128);
    uint2 o;
    if (SRC_BF) {        // h already bf16 ([n][256] bf16): plain copy
        const unsigned short* src = reinterpret_cast<const unsigned short*>(q < 64 ? hfw : hbw) + (size_t)site * 256 + (q & 63) * 4;
        o = *reinterpret_cast<const uint2*>(src);
    } else {
        const float* src = (q < 64 ? hfw : hbw) + (size_t)site * 256 + (q & 63) * 4;
        const float4 v = *reinterpret_cast<const float4*>(src);
        o = make_uint2(pack_bf2(v.x, v.y), pack_bf2(v.z, v.w));
    }
    *reinterpret_cast<uint2*>(joint + site * joint_ld + q * 4) = o;
}

hipError_t launch_pack_event_feat_bf16(const float* hfw, const float* hbw, float* joint, int n, int joint_ld, int src_bf16,
                                       hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    const dim3 grid((unsigned)(((long)n * 128 + 255) / 256));
    if (src_bf16)
        hipLaunchKernelGGL(pack_event_feat_bf16_kernel<true>, grid, dim3(256), 0, s, hfw, hbw, (unsigned short*)joint, n, (long)joint_ld);
    else
        hipLaunchKernelGGL(pack_event_feat_bf16_kernel<false>, grid, dim3(256), 0, s, hfw, hbw, (unsigned short*)joint, n, (long)joint_ld);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void embed_table_kernel(const float* __restrict__ emb, const float* __restrict__ kernel,
                                                           float* __restrict__ table, int vocab, int esize, int ncol)
{
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= (long)vocab * ncol) return;
    const int col = (int)(i % ncol);
    const int v = (int)(i / ncol);
    float a = 0.0f;
    for (int e = 0; e < esize; ++e) a = fmaf(emb[(size_t)v * esize + e], kernel[(size_t)e * ncol + col], a);
    table[i] = a;
}

hipError_t launch_embed_table(const float* emb, const float* kernel, float* table, int vocab, int esize, int ncol,
                              hipStream_t s)
{
    const long total = (long)vocab * ncol;
    hipLaunchKernelGGL(embed_table_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, emb, kernel, table, vocab,
                       esize, ncol);
    return hipGetLastError();
}

}  // namespace ds
