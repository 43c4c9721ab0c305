// Plain decimal tokens of the feature TSV -- [-]digits[.digits], closed by ',' or by the end of the column -- parsed two
// ways. ONE definition, included by the reader (ds_io.cpp) and by its checker / micro-benchmark (tools/parse_bench.cpp,
// run by tests/test_fastio.py): what the test compares against strtod on 3 M tokens is the code the reader runs.
// Host-only code (the file travels through hipcc -x hip with the kernels; nothing of it runs on the device).
#pragma once
#include <immintrin.h>
#include <cstdint>

namespace ds_dec {

static const double kPow10[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15,
                                  1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};

// Byte loop. What feature files hold ("%.6f" text, extract_features.py:289-303): [-]digits[.digits] with at most 15
// significant digits and no exponent. The digits form an integer m < 2^53 and the value is m / 10^k with 10^k exact in
// double (k <= 22), so ONE IEEE division gives the correctly rounded double -- the number float() / strtod return
// (Clinger's fast path). Returns the token's end, or nullptr when the form is anything else (exponents, inf / nan, long
// mantissas: the general parser takes over).
static inline const char* scalar_token(const char* p, const char* e, float* out)
{
    const char* q = p;
    const bool neg = q < e && *q == '-';
    if (neg) ++q;
    uint64_t m = 0;
    int nd = 0, frac = 0;
    bool any = false, dot = false;
    for (; q < e; ++q) {
        const unsigned c = (unsigned char)*q;
        if (c - '0' <= 9u) {
            any = true;
            if (nd == 0 && c == '0') { if (dot) ++frac; continue; }     // leading zeros carry no significance
            if (++nd > 15) return nullptr;
            m = m * 10 + (c - '0');
            if (dot) ++frac;
        } else if (c == '.' && !dot) {
            dot = true;
        } else {
            break;
        }
    }
    if (!(any && frac <= 22 && (q == e || *q == ','))) return nullptr;
    const double v = (double)m / kPow10[frac];
    *out = (float)(neg ? -v : v);
    return q;
}

// One plain decimal token -- [-]digits[.digits], at most 15 digits, closed by ',' or by the end of the column -- with ONE
// 16-byte load: the terminator by compare + tzcnt, the dot squeezed out with pshufb, the digits -> integer with three
// multiply-adds, then the same single division as the byte loop below (tools/parse_bench.cpp checks both against strtod
// on 3 M tokens: 34 -> 20 ns per token). The caller guarantees 16 readable bytes at p + 1. Returns the token's end, or
// nullptr when the form is anything else (the byte loop / from_chars take over).
#ifdef __HIP_DEVICE_COMPILE__
static inline const char* simd_token(const char*, const char*, float*) { return nullptr; }
#else
__attribute__((target("ssse3,sse4.1"))) static inline const char* simd_token(const char* p, const char* e, float* out)
{
    const bool neg = *p == '-';
    const char* q = p + neg;
    const __m128i x = _mm_loadu_si128(reinterpret_cast<const __m128i*>(q));
    const unsigned comma = (unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(x, _mm_set1_epi8(',')));
    int len = __builtin_ctz(comma | 0x10000u);
    if (q + len > e) len = (int)(e - q);
    if (len <= 0 || len >= 16) return nullptr;
    const unsigned lenmask = (1u << len) - 1;
    const unsigned dotm = (unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(x, _mm_set1_epi8('.'))) & lenmask;
    if (dotm & (dotm - 1)) return nullptr;                                         // two dots
    const __m128i d = _mm_sub_epi8(x, _mm_set1_epi8('0'));
    const unsigned digm = (unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_min_epu8(d, _mm_set1_epi8(9)), d)) & lenmask;
    if ((digm | dotm) != lenmask || digm == 0) return nullptr;                     // neither digit nor dot somewhere, or no digit at all
    const int dotpos = dotm ? __builtin_ctz(dotm) : len;
    const int n = len - (dotm ? 1 : 0);                                            // digits: m < 10^15 < 2^53
    if (n > 15) return nullptr;
    const int frac = dotm ? len - dotpos - 1 : 0;
    // right-align the digits in 16 bytes without the dot: output byte j takes digit t = n - 16 + j (negative: zero), which sits at
    // token byte t (before the dot) or t + 1 (behind it)
    const __m128i j = _mm_setr_epi8(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
    const __m128i t = _mm_add_epi8(j, _mm_set1_epi8((char)(n - 16)));
    const __m128i behind = _mm_cmpgt_epi8(t, _mm_set1_epi8((char)(dotpos - 1)));  // t >= dotpos -> 0xFF
    const __m128i ctl = _mm_sub_epi8(t, behind);                                    // t + 1 there; a negative t keeps bit 7: pshufb gives zero
    const __m128i dg = _mm_shuffle_epi8(d, ctl);
    const __m128i t2 = _mm_maddubs_epi16(dg, _mm_setr_epi8(10, 1, 10, 1, 10, 1, 10, 1, 10, 1, 10, 1, 10, 1, 10, 1));
    const __m128i t4 = _mm_madd_epi16(t2, _mm_setr_epi16(100, 1, 100, 1, 100, 1, 100, 1));
    const __m128i t4p = _mm_packus_epi32(t4, t4);
    const __m128i t8 = _mm_madd_epi16(t4p, _mm_setr_epi16(10000, 1, 10000, 1, 10000, 1, 10000, 1));
    const uint64_t hi = (uint32_t)_mm_cvtsi128_si32(t8), lo = (uint32_t)_mm_extract_epi32(t8, 1);
    const double v = (double)(hi * 100000000ull + lo) / kPow10[frac];
    *out = (float)(neg ? -v : v);
    return q + len;
}
#endif

}  // namespace ds_dec
