// Micro-benchmark + exhaustive-ish check of the decimal token parsers of the feature-TSV reader (host only):
//   g++ -O3 -std=c++17 tools/parse_bench.cpp -o /tmp/parse_bench && /tmp/parse_bench
// BOTH ARE THE READER'S OWN CODE (deepsignal_amd/csrc/ds_dec_token.h, included here and by ds_io.cpp).
// `scalar` is the byte loop of the reader's fast path, `simd` the SSSE3 / SSE4.1 form (one 16-byte load per token: the
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

#include "../deepsignal_amd/csrc/ds_dec_token.h"
using ds_dec::scalar_token;
using ds_dec::simd_token;

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
