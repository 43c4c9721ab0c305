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

}  // namespace ds
