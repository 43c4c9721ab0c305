// Micro-benchmark + exhaustive-ish check of the decimal token parsers of the feature-TSV reader (host only):
//   g++ -O3 -std=c++17 tools/parse_bench.cpp -o /tmp/parse_bench && /tmp/parse_bench
// `scalar` is the byte loop of ds_io.cpp's fast path, `simd` the SSSE3 / SSE4.1 form (one 16-byte load per token: the
// terminator by compare + tzcnt, the dot squeezed out with pshufb, sixteen digits -> integer with three multiply-adds).
// Both must return the bits of strtod for every token either of them accepts.
#include <immintrin.h>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

static const double kPow10[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15,
                                  1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};

// returns the end of the token, or nullptr if the fast path does not apply
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
            if (nd == 0 && c == '0') { if (dot) ++frac; continue; }
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

__attribute__((target("ssse3,sse4.1"))) static inline const char* simd_token(const char* p, const char* e, float* out)
{
    const bool neg = *p == '-';
    const char* q = p + neg;
    const __m128i x = _mm_loadu_si128(reinterpret_cast<const __m128i*>(q));      // caller guarantees 16 readable bytes
    const unsigned comma = (unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(x, _mm_set1_epi8(',')));
    int len = __builtin_ctz(comma | 0x10000u);
    if (q + len > e) len = (int)(e - q);
    if (len <= 0 || len >= 16) return nullptr;
    const unsigned lenmask = (1u << len) - 1;
    const unsigned dotm = (unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(x, _mm_set1_epi8('.'))) & lenmask;
    if (dotm & (dotm - 1)) return nullptr;                                         // two dots
    const __m128i d = _mm_sub_epi8(x, _mm_set1_epi8('0'));
    const unsigned digm = (unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_min_epu8(d, _mm_set1_epi8(9)), d)) & lenmask;
    if ((digm | dotm) != lenmask || digm == 0) return nullptr;                     // something that is neither digit nor dot, or no digit
    const int dotpos = dotm ? __builtin_ctz(dotm) : len;
    const int n = len - (dotm ? 1 : 0);                                            // digits
    if (n > 15) return nullptr;
    const int frac = dotm ? len - dotpos - 1 : 0;
    // right-align the digits in 16 bytes without the dot: output byte j takes digit t = n - 16 + j (negative: zero),
    // which sits at token byte t (before the dot) or t + 1 (behind it)
    const __m128i j = _mm_setr_epi8(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
    const __m128i t = _mm_add_epi8(j, _mm_set1_epi8((char)(n - 16)));
    const __m128i behind = _mm_cmpgt_epi8(t, _mm_set1_epi8((char)(dotpos - 1)));  // t >= dotpos -> 0xFF
    const __m128i ctl = _mm_sub_epi8(t, behind);                                    // t + 1 there; negative t keeps bit 7: pshufb zero
    const __m128i dg = _mm_shuffle_epi8(d, ctl);
    const __m128i t2 = _mm_maddubs_epi16(dg, _mm_setr_epi8(10, 1, 10, 1, 10, 1, 10, 1, 10, 1, 10, 1, 10, 1, 10, 1));
    const __m128i t4 = _mm_madd_epi16(t2, _mm_setr_epi16(100, 1, 100, 1, 100, 1, 100, 1));
    const __m128i t4p = _mm_packus_epi32(t4, t4);
    const __m128i t8 = _mm_madd_epi16(t4p, _mm_setr_epi16(10000, 1, 10000, 1, 10000, 1, 10000, 1));
    const uint64_t hi = (uint32_t)_mm_cvtsi128_si32(t8), lo = (uint32_t)_mm_extract_epi32(t8, 1);
    const uint64_t m = hi * 100000000ull + lo;
    const double v = (double)m / kPow10[frac];
    *out = (float)(neg ? -v : v);
    return q + len;
}

int main()
{
    std::mt19937_64 rng(1);
    // realistic tokens: "%.6f" of N(0,1) with trailing zeros stripped as str(round(x, 6)) does, plus edge forms
    std::vector<std::string> toks;
    std::normal_distribution<double> nd(0.0, 1.5);
    for (int i = 0; i < 2000000; ++i) {
        char b[64];
        const double x = std::round(nd(rng) * 1e6) / 1e6;
        snprintf(b, sizeof b, "%.6f", x);
        std::string s(b);
        if (i & 1) { while (s.back() == '0') s.pop_back(); if (s.back() == '.') s.push_back('0'); }
        toks.push_back(s);
    }
    const char* edge[] = {"0", "-0", "0.0", "-0.0", ".5", "-.5", "5.", "123456789012345", "0.000001", "-123.456789", "000.5", "1e5", "nan",
                          "inf", "1..2", "-", ".", "", "1234567890123456", "12a", "+5", "99999999.999999", "0.1234567890123"};
    for (const char* s : edge) toks.push_back(s);
    std::uniform_int_distribution<int> ld(1, 15), dd(0, 9), pd(0, 15);
    for (int i = 0; i < 1000000; ++i) {            // random digit strings with a dot somewhere
        std::string s;
        if (i & 1) s.push_back('-');
        const int L = ld(rng), dp = pd(rng);
        for (int k = 0; k < L; ++k) { if (k == dp) s.push_back('.'); s.push_back((char)('0' + dd(rng))); }
        toks.push_back(s);
    }
    std::string buf;
    std::vector<size_t> off;
    for (auto& s : toks) { off.push_back(buf.size()); buf += s; buf.push_back(','); }
    off.push_back(buf.size());
    buf.append(32, ',');
    size_t bad = 0, simd_ok = 0, sc_ok = 0;
    for (size_t i = 0; i + 1 < off.size(); ++i) {
        const char* p = buf.data() + off[i];
        const char* e = buf.data() + off[i + 1] - 1;
        float a = 0, b = 0;
        const char* ra = scalar_token(p, e, &a);
        const char* rb = p < e ? simd_token(p, e, &b) : nullptr;
        char tmp[64]; const size_t n = (size_t)(e - p); memcpy(tmp, p, n); tmp[n] = 0;
        char* endp = nullptr;
        const float ref = (float)strtod(tmp, &endp);
        if (ra) { ++sc_ok; if (ra != e || memcmp(&a, &ref, 4)) { ++bad; if (bad < 10) printf("scalar mismatch on '%s'\n", tmp); } }
        if (rb) { ++simd_ok; if (rb != e || memcmp(&b, &ref, 4)) { ++bad; if (bad < 10) printf("simd mismatch on '%s': %g vs %g\n", tmp, b, ref); } }
        if (rb && !ra) { ++bad; if (bad < 10) printf("simd accepts '%s', scalar does not\n", tmp); }
    }
    printf("tokens %zu  scalar accepts %zu  simd accepts %zu  mismatches %zu\n", off.size() - 1, sc_ok, simd_ok, bad);
    // timing over the realistic part
    const size_t nreal = 2000000;
    for (int rep = 0; rep < 2; ++rep) {
        float acc = 0;
        auto t0 = std::chrono::steady_clock::now();
        for (size_t i = 0; i < nreal; ++i) { float v; const char* p = buf.data() + off[i]; scalar_token(p, buf.data() + off[i + 1] - 1, &v); acc += v; }
        auto t1 = std::chrono::steady_clock::now();
        for (size_t i = 0; i < nreal; ++i) { float v; const char* p = buf.data() + off[i]; simd_token(p, buf.data() + off[i + 1] - 1, &v); acc += v; }
        auto t2 = std::chrono::steady_clock::now();
        printf("scalar %.1f ns/token   simd %.1f ns/token   (%g)\n", std::chrono::duration<double, std::nano>(t1 - t0).count() / nreal,
               std::chrono::duration<double, std::nano>(t2 - t1).count() / nreal, acc);
    }
    return bad != 0;
}
