// ds_device.h -- device-side helpers shared by the kernel translation units (ds_kernels.hip, ds_split.hip).
#pragma once
#include "ds_internal.h"

namespace ds {

typedef float floatx16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float4 f4max(float4 a, float4 b)
{
    return make_float4(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w));
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
// LSTM gate non-linearities on the hardware exp (v_exp_f32, ~1 ulp): abs error < 2e-7, far inside the
// 2e-5 parity budget, at a fraction of the ocml expf/tanhf instruction count.
// v_rcp_f32 (1 ulp) instead of a division: hipcc expands __fdividef / `1.0f / x` into the full IEEE sequence
// (v_div_scale, v_rcp, four fma, v_div_fmas, v_div_fixup -- ~12 instructions); ten of those per LSTM unit were most
// of the cell epilogue's VALU time.
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float fast_tanh(float x) { return fmaf(2.0f, fast_sigmoid(2.0f * x), -1.0f); }

// Pointers reach the kernels through descriptor structs, so the compiler only knows them as
// generic; these casts make every access a global_* instruction (flat_* would tie vmcnt and
// lgkmcnt together and force full drains before each MFMA block).
typedef float v4f __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) v4f* gptr4;
typedef const __attribute__((address_space(1))) float* gptr1;
typedef __attribute__((address_space(1))) float* gptr1w;
__device__ __forceinline__ float4 gload4(const float* p)
{
    const v4f v = *(gptr4)(p);
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float gload(const float* p) { return *(gptr1)(p); }
__device__ __forceinline__ float4 gload4_nt(const float* p)      // global_load_dwordx4 ... nt
{
    const v4f v = __builtin_nontemporal_load((gptr4)(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void gstore(float* p, float v) { *(gptr1w)(p) = v; }

// ---- bf16 helpers (mixed-precision mode: bf16 operands, fp32 accumulate) ----
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// max of two packed bf16 pairs; widening a bf16 to fp32 is a shift, and max returns one of its inputs, so
// narrowing back by truncation is exact
__device__ __forceinline__ float bf2max(float a, float b)
{
    const unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
    const float lo = fmaxf(__uint_as_float(ua << 16), __uint_as_float(ub << 16));
    const float hi = fmaxf(__uint_as_float(ua & 0xffff0000u), __uint_as_float(ub & 0xffff0000u));
    return __uint_as_float((__float_as_uint(hi) & 0xffff0000u) | (__float_as_uint(lo) >> 16));
}
__device__ __forceinline__ float4 bf8max(float4 a, float4 b)
{
    return make_float4(bf2max(a.x, b.x), bf2max(a.y, b.y), bf2max(a.z, b.z), bf2max(a.w, b.w));
}
// Max of packed bf16 values that are all >= 0 (or -0): every pooled tensor of this network is a ReLU output, and for
// non-negative floats the bit patterns order like signed 16-bit integers (-0 = 0x8000 is the smallest, so
// max(-0, x) = x as it should). One v_pk_max_i16 per pair instead of ~9 VALU ops.
typedef short short2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float bf2max_nn(float a, float b)
{
    return __builtin_bit_cast(float, __builtin_elementwise_max(__builtin_bit_cast(short2v, a), __builtin_bit_cast(short2v, b)));
}
__device__ __forceinline__ float4 bf8max_nn(float4 a, float4 b)
{
    return make_float4(bf2max_nn(a.x, b.x), bf2max_nn(a.y, b.y), bf2max_nn(a.z, b.z), bf2max_nn(a.w, b.w));
}
__device__ __forceinline__ unsigned short f2bf(float v)     // round to nearest even (v_cvt_pk_bf16_f32)
{
    return __builtin_bit_cast(unsigned short, (__bf16)v);
}
__device__ __forceinline__ float bf2f(unsigned short b) { return __uint_as_float((unsigned)b << 16); }
// relu on the bits: max(int(x), 0). One v_max_i32; relu_f(x) costs two instructions in IEEE mode (hipcc first quiets
// a possible signalling NaN with v_max_f32 x, x). Same result for every non-NaN x (-0.0 -> +0.0 either way).
__device__ __forceinline__ float relu_f(float x)
{
    const int b = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, b > 0 ? b : 0);
}
// two floats -> one dword of two bf16 (round to nearest even): ONE v_cvt_pk_bf16_f32, low half = a
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float float2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf2(float a, float b)
{
    const float2_t f = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2_t));
}

__device__ __forceinline__ floatx16 mfma_bf(float4 a, float4 b, floatx16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// LDS-DMA of one 1 KiB wave fragment: lane i's 16 bytes at gsrc land at LDS byte address lds_dst + 16 i (lds_dst is
// wave-uniform and goes through M0). Written as inline asm on purpose: with the builtin, hipcc (ROCm 7.2) tracks the
// transfer as an LDS store and puts `s_waitcnt vmcnt(0)` in front of the next ds_read, which would drain the
// requests this kernel keeps in flight across its barrier; the kernel counts its own vmcnt instead (see the loop).
// M0 is compiler-reserved: saved and restored inside the statement (cdna_hip_programming.md 5.7).
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

// The same with the source as (wave-uniform 64-bit base in SGPRs) + (per-lane 32-bit byte offset): no vector address
// arithmetic when the base moves.
__device__ __forceinline__ void glds16s(const void* gbase, unsigned lane_off, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(lane_off), "s"(gbase), "s"(lds_dst)
                 : "memory");
}
template <int I> struct LdsSlot { static constexpr int value = I; };

// Accumulator's initial value of an LSTM tile (transposed product: register 4g + e = gate g of unit p8 + e for one
// site): bias (+1.0 on the forget gate: TF adds forget_bias at run time) and, for layer 0, the folded embedding-table
// row plus the (mean, std, len) rank-1 terms. Every load of a group is issued before anything waits: one L2 round
// trip for the biases, two (code -> table row) for layer 0 -- written with per-gate branches this was four to eight
// SERIALIZED round trips in front of every tile.
__device__ __forceinline__ void lstm_acc_init(const LstmCell& C, int p8, int rowc, int T, floatx16& acc)
{
    float4 z[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) z[g] = gload4(C.bias + g * 256 + p8);
    if (C.use_feat) {                                    // wave-uniform: the two layer-0 cells of a diagonal
        const float* wf = C.wfeat;
        const float* tab = C.table;
        const unsigned it = (unsigned)rowc * T + C.t;
        const float f0 = gload(C.means + it), f1 = gload(C.stds + it), f2 = gload(C.lens + it);
        // codes index the folded [vocab = 1024][1024] table; a non-Python client may pass anything: clamp
        const int code = tab ? min(max(*(const __attribute__((address_space(1))) int*)(C.codes + it), 0), 1023) : 0;
        float4 w0[4], w1[4], w2[4], tb[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            w0[g] = gload4(wf + g * 256 + p8);
            w1[g] = gload4(wf + 1024 + g * 256 + p8);
            w2[g] = gload4(wf + 2048 + g * 256 + p8);
        }
        const float* trow = tab ? tab + (size_t)code * 1024 : C.bias;      // no table (is_base = no): any valid address
#pragma unroll
        for (int g = 0; g < 4; ++g) tb[g] = gload4(trow + g * 256 + p8);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            // explicit fma chains: every instantiation / tiling must round identically
            float4 x = make_float4(fmaf(f2, w2[g].x, fmaf(f1, w1[g].x, f0 * w0[g].x)), fmaf(f2, w2[g].y, fmaf(f1, w1[g].y, f0 * w0[g].y)),
                                   fmaf(f2, w2[g].z, fmaf(f1, w1[g].z, f0 * w0[g].z)), fmaf(f2, w2[g].w, fmaf(f1, w1[g].w, f0 * w0[g].w)));
            if (tab) { x.x += tb[g].x; x.y += tb[g].y; x.z += tb[g].z; x.w += tb[g].w; }
            if (g == 2) { z[g].x += 1.0f; z[g].y += 1.0f; z[g].z += 1.0f; z[g].w += 1.0f; }
            z[g].x += x.x; z[g].y += x.y; z[g].z += x.z; z[g].w += x.w;
        }
    } else {
        z[2].x += 1.0f; z[2].y += 1.0f; z[2].z += 1.0f; z[2].w += 1.0f;
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        acc[4 * g] = z[g].x; acc[4 * g + 1] = z[g].y; acc[4 * g + 2] = z[g].z; acc[4 * g + 3] = z[g].w;
    }
}

// The same initial value from lstm_xproj_kernel's image: tile (mtile, ntile) = four 1 KiB wave loads
__device__ __forceinline__ void lstm_acc_load(const float* xinit, int mtile, int ntile, unsigned lane4, floatx16& acc)
{
    const float* p = xinit + ((size_t)(mtile * 32 + ntile) * 4) * 256 + lane4;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 z = gload4(p + g * 256);
        acc[4 * g] = z.x; acc[4 * g + 1] = z.y; acc[4 * g + 2] = z.z; acc[4 * g + 3] = z.w;
    }
}

// Gate arithmetic of an LSTM tile (layers.py:49-50 / TF LSTMCell): registers 0-3 = i, 4-7 = j, 8-11 = f (+1 already
// inside), 12-15 = o of four neighbouring units; returns the new c and h as float4.
__device__ __forceinline__ void lstm_gates(const floatx16& acc, const float4& cp, float4& cn, float4& hn)
{
    cn.x = fmaf(fast_sigmoid(acc[8]), cp.x, fast_sigmoid(acc[0]) * fast_tanh(acc[4]));
    cn.y = fmaf(fast_sigmoid(acc[9]), cp.y, fast_sigmoid(acc[1]) * fast_tanh(acc[5]));
    cn.z = fmaf(fast_sigmoid(acc[10]), cp.z, fast_sigmoid(acc[2]) * fast_tanh(acc[6]));
    cn.w = fmaf(fast_sigmoid(acc[11]), cp.w, fast_sigmoid(acc[3]) * fast_tanh(acc[7]));
    hn.x = fast_sigmoid(acc[12]) * fast_tanh(cn.x);
    hn.y = fast_sigmoid(acc[13]) * fast_tanh(cn.y);
    hn.z = fast_sigmoid(acc[14]) * fast_tanh(cn.z);
    hn.w = fast_sigmoid(acc[15]) * fast_tanh(cn.w);
}

// Workgroup -> logical tile of an LSTM launch. Cells of a diagonal differ in K (layer 0 multiplies h only, first steps
// x only): a tile of a K = 256 cell is half the matrix work of a K = 512 one, and a launch is as slow as its busiest
// CU. The host sorts the cells by descending K and passes the tile counts c0, c1 of the two heaviest classes; logical
// tiles are cell-major (heaviest first), inside a cell n-group-major with the m-blocks of one weight panel adjacent.
// The map uses the dispatch pattern observed on MI355X -- workgroup b runs on XCD b % 8, CU slot (b >> 3) % 32 of that
// XCD, so blocks b, b + 256, b + 512 share a CU (tools/attic/lstm_rawstamps.py):
//   * every XCD takes a CONTIGUOUS eighth of each class (half a K = 512 cell and a quarter of a K = 256 cell on a full
//     diagonal), so a cell's activation rows are fetched by two L2s, not by four, and a weight panel by one;
//   * inside the XCD the heavier classes come first in dispatch order, so CU slot s gets the XCD's tiles s, s + 32,
//     s + 64: two K = 512 tiles and one K = 256 tile instead of three of a kind.
// Placement is a speed assumption only: any other dispatch order gives the same results, just less balance.
__device__ __forceinline__ int lstm_logical_tile(int b, int total, int c0, int c1)
{
    if (((c0 | c1 | total) & 7) == 0) {
        const int x = b & 7, j = b >> 3, p0 = c0 >> 3, p1 = c1 >> 3;
        if (j < p0) return x * p0 + j;
        if (j < p0 + p1) return c0 + x * p1 + (j - p0);
        return c0 + c1 + x * ((total - c0 - c1) >> 3) + (j - p0 - p1);
    }
    // class sizes not divisible by the XCD count (ragged batches): rounds of one tile per CU in sorted order
    const int full = total & ~255;
    if (b < full) return (b & ~255) + (b & 7) * 32 + ((b & 255) >> 3);
    const int R = total - full, p = b - full, q = R >> 3, r = R & 7, xcd = p & 7;
    return full + xcd * q + (xcd < r ? xcd : r) + (p >> 3);
}

}  // namespace ds
