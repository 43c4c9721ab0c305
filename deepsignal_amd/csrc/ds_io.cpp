// ds_io.cpp — native feature-TSV reader + result-row formatter (scope row f1 of SURVEY.md section 8).
//
// At GPU rates the reference's Python reader (~400 float() calls per row,
// /root/reference/deepsignal/call_modifications.py:35-91) and its per-site formatting loop
// (:183-190) are >100x too slow; this file restates both natively, behind the same C ABI:
//   * reader: 12 tab-separated columns (extract_features.py:289-303), rows grouped by read id
//     (column 5), one batch every `max_reads` reads, exactly as _read_features_file does;
//     numbers are parsed as double and narrowed to float (= Python float() then the float32 feed);
//   * formatter: "sampleinfo \t p0/(p0+p1) \t p1/(p0+p1) \t label \t kmer" with the float32
//     shortest round-trip text numpy's str(np.float32) prints.
// Host-only code (no HIP); compiled into libdeepsignal_hip.so.
#include "../../include/deepsignal_hip.h"
#include "ds_dec_token.h"

#include <algorithm>
#include <atomic>
#include <charconv>
#include <immintrin.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

// A team of parser threads that lives as long as the reader: a queue item is ~1,000 rows (a millisecond of parsing on
// 16 threads), so starting and joining 16 std::threads per item cost about as much as the parsing itself.
struct ds_team {
    std::vector<std::thread> th;
    std::mutex m;
    std::condition_variable cv_go, cv_done;
    const std::function<void()>* job = nullptr;
    uint64_t gen = 0;
    int want = 0, running = 0;       // helpers asked to run this generation / still running it
    bool stop = false;
    explicit ds_team(int helpers)
    {
        for (int i = 0; i < helpers; ++i)
            th.emplace_back([this, i] {
                uint64_t seen = 0;
                for (;;) {
                    const std::function<void()>* j;
                    {
                        std::unique_lock<std::mutex> lk(m);
                        cv_go.wait(lk, [&] { return stop || (gen != seen && i < want); });
                        if (stop) return;
                        seen = gen;
                        j = job;
                    }
                    (*j)();
                    {
                        std::lock_guard<std::mutex> lk(m);
                        if (--running == 0) cv_done.notify_one();
                    }
                }
            });
    }
    // run `f` on the caller and on up to `helpers` team threads; returns when all of them are done
    void run(int helpers, const std::function<void()>& f)
    {
        helpers = std::min<int>(helpers, (int)th.size());
        if (helpers > 0) {
            std::lock_guard<std::mutex> lk(m);
            job = &f; want = helpers; running = helpers; ++gen;
        }
        if (helpers > 0) cv_go.notify_all();
        f();
        if (helpers > 0) {
            std::unique_lock<std::mutex> lk(m);
            cv_done.wait(lk, [&] { return running == 0; });
            want = 0;
        }
    }
    ~ds_team()
    {
        {
            std::lock_guard<std::mutex> lk(m);
            stop = true;
        }
        cv_go.notify_all();
        for (auto& t : th) t.join();
    }
};

struct ds_tsv {
    int fd = -1;
    const char* data = nullptr;
    size_t size = 0, pos = 0;
    size_t limit = 0;          // rows are read from [pos, limit): the whole file, or one rank's byte range (ds_tsv_set_range)
    int kmer_len = 17, signal_len = 360, nthreads = 1;
    int64_t line_no = 0;       // rows handed out since the last rewind (ds_tsv_open / ds_tsv_set_range)
    size_t range_begin = 0;    // first byte of the current range (0 for the whole file)
    std::string err;
    // current batch
    std::vector<int32_t> kmer, labels;
    std::vector<float> means, stds, lens, signals;
    std::vector<char> info;
    std::vector<int64_t> info_off;
    std::vector<std::pair<const char*, const char*>> lines;
    // lines located ahead of the consumer by the parallel newline scan (non-blank, '\r'-trimmed, file order): [pend_head, size)
    // are not handed out yet; `scanned` = first byte behind the last located line (always a line start, or `limit`)
    std::vector<std::pair<const char*, const char*>> pend;
    size_t pend_head = 0, scanned = 0, scan_bytes = 2u << 20;
    ds_team* team = nullptr;   // nthreads - 1 helpers, started with the first item that needs them
};

namespace {

inline const char* find_tab(const char* p, const char* e)
{
    const void* q = memchr(p, '\t', (size_t)(e - p));
    return q ? (const char*)q : e;
}

#ifdef __HIP_DEVICE_COMPILE__          // (the file goes through hipcc -x hip with the kernels: nothing of it runs on the device)
static const bool kHaveSimdParse = false;
#else
static const bool kHaveSimdParse = (__builtin_cpu_init(), __builtin_cpu_supports("ssse3") && __builtin_cpu_supports("sse4.1"));
#endif

// comma-separated list of exactly `count` numbers in [p, e); `safe` = last address a 17-byte look-ahead may start at
template <class Out>
bool parse_list(const char* p, const char* e, int count, Out* out, bool as_int, const char* safe = nullptr)
{
    for (int i = 0; i < count; ++i) {
        if (p >= e) return false;
        if (as_int) {
            long v = 0;
            if (*p == '+') { ++p; if (p >= e || *p < '0' || *p > '9') return false; }     // int("+-5") raises in Python
            auto r = std::from_chars(p, e, v);
            if (r.ec != std::errc()) return false;
            out[i] = (Out)v;
            p = r.ptr;
        } else {
            double v = 0;
            if (*p == '+') { ++p; if (p >= e || *p == '-' || *p == '+') return false; }   // float("+-5") raises in Python
            if (kHaveSimdParse && p <= safe) {
                float fv;
                if (const char* q = ds_dec::simd_token(p, e, &fv)) {
                    out[i] = (Out)fv;
                    p = q;
                    if (i + 1 < count) {
                        if (p >= e || *p != ',') return false;
                        ++p;
                    }
                    continue;
                }
            }
            {
                // byte-loop fast path (ds_dec_token.h); libstdc++ 11's from_chars<double> below goes through strtod and a
                // locale switch: ~10x its cost, and ~400 numbers per row make it the reader's whole budget
                float fv;
                if (const char* q = ds_dec::scalar_token(p, e, &fv)) {
                    out[i] = (Out)fv;
                    p = q;
                    if (i + 1 < count) {
                        if (p >= e || *p != ',') return false;
                        ++p;
                    }
                    continue;
                }
            }
            auto r = std::from_chars(p, e, v);
            if (r.ec == std::errc::result_out_of_range) {
                // Python's float() (the reference reader, call_modifications.py:78-85) gives +-inf for 1e400 and
                // 0.0 for 1e-400 instead of failing: same here (strtod on a bounded copy of the token)
                char tmp[64];
                const size_t len = (size_t)(r.ptr - p);
                if (len >= sizeof tmp) return false;
                memcpy(tmp, p, len);
                tmp[len] = 0;
                v = strtod(tmp, nullptr);
            } else if (r.ec != std::errc()) {
                return false;
            }
            out[i] = (Out)(float)v;                       // float64 -> float32, as the TF feed does
            p = r.ptr;
        }
        if (i + 1 < count) {
            if (p >= e || *p != ',') return false;
            ++p;
        }
    }
    return p == e;
}

int base_code(char c)
{
    switch (c) {     // process_utils.py:21
    case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; case 'N': return 4;
    default: return -1;
    }
}

struct RowDest { int32_t* kmer; float* means; float* stds; float* lens; float* signals; int32_t* labels; };

// one row -> slot i of the destination arrays; returns false on a malformed row
bool parse_row(const ds_tsv* t, const RowDest& d, size_t i, const char* b, const char* e, int64_t* info_len)
{
    // col[c] = start of column c; col[12] = one past the tab (or line end) that closes column 11. Columns beyond the
    // 12th are ignored, as the reference reader ignores words[12:] (call_modifications.py:47-57).
    const char* col[13];
    col[0] = b;
    int nc = 1;
    for (const char* p = b; p < e && nc < 13;) {
        const char* q = find_tab(p, e);
        if (q == e) break;
        col[nc++] = q + 1;
        p = q + 1;
    }
    if (nc < 12) return false;
    if (nc == 12) col[12] = e + 1;
    auto cb = [&](int c) { return col[c]; };
    auto ce = [&](int c) { return col[c + 1] - 1; };
    const int K = t->kmer_len, S = t->signal_len;
    if (ce(6) - cb(6) != K) return false;
    for (int k = 0; k < K; ++k) {
        const int code = base_code(cb(6)[k]);
        if (code < 0) return false;
        d.kmer[i * K + k] = code;
    }
    const char* const safe = t->size >= 17 ? t->data + t->size - 17 : nullptr;      // a token may be looked at 17 bytes at a time up to here
    if (!parse_list(cb(7), ce(7), K, &d.means[i * K], false, safe)) return false;
    if (!parse_list(cb(8), ce(8), K, &d.stds[i * K], false, safe)) return false;
    if (!parse_list(cb(9), ce(9), K, &d.lens[i * K], true)) return false;
    if (!parse_list(cb(10), ce(10), S, &d.signals[i * S], false, safe)) return false;
    const char* le = ce(11);
    while (le > cb(11) && (le[-1] == '\r' || le[-1] == ' ')) --le;
    int lab = 0;
    auto r = std::from_chars(cb(11), le, lab);
    if (r.ec != std::errc() || r.ptr != le) return false;
    d.labels[i] = lab;
    *info_len = ce(5) - b;       // columns 0..5 joined by tabs, verbatim
    return true;
}

// str(np.float32(x)): shortest digits that round-trip float32; positional for 1e-4 <= |x| < 1e16,
// scientific otherwise (numpy's default float printing)
int format_f32(float x, char* out)
{
    if (std::isnan(x)) { memcpy(out, "nan", 3); return 3; }
    if (std::isinf(x)) { if (x < 0) { memcpy(out, "-inf", 4); return 4; } memcpy(out, "inf", 3); return 3; }
    char buf[48];
    auto r = std::to_chars(buf, buf + sizeof buf, x, std::chars_format::scientific);   // shortest round-trip
    *r.ptr = 0;
    // buf = [-]d[.ddd]e[+-]XX
    char* p = buf;
    int n = 0;
    if (*p == '-') out[n++] = *p++;
    char digits[24];
    int nd = 0;
    digits[nd++] = *p++;
    if (*p == '.') { ++p; while (*p != 'e') digits[nd++] = *p++; }
    ++p;   // 'e'
    const int ex = atoi(p);
    if (x == 0.0f) { memcpy(out + n, "0.0", 3); return n + 3; }
    if (ex >= -4 && ex < 16) {
        if (ex < 0) {
            out[n++] = '0'; out[n++] = '.';
            for (int i = 0; i < -ex - 1; ++i) out[n++] = '0';
            for (int i = 0; i < nd; ++i) out[n++] = digits[i];
        } else {
            for (int i = 0; i <= ex; ++i) out[n++] = i < nd ? digits[i] : '0';
            out[n++] = '.';
            if (nd > ex + 1) for (int i = ex + 1; i < nd; ++i) out[n++] = digits[i];
            else out[n++] = '0';
        }
        return n;
    }
    out[n++] = digits[0];
    if (nd > 1) {                       // numpy prints '1e-05', not '1.0e-05'
        out[n++] = '.';
        for (int i = 1; i < nd; ++i) out[n++] = digits[i];
    }
    out[n++] = 'e';
    out[n++] = ex < 0 ? '-' : '+';
    const int ax = ex < 0 ? -ex : ex;
    if (ax < 10) out[n++] = '0';
    n += snprintf(out + n, 8, "%d", ax);
    return n;
}

}  // namespace

// CPUs this process may actually use: the affinity mask and the cgroup-v2 CPU quota (cpu.max) both count; the GPU
// boxes expose 256 logical CPUs under a 16-CPU quota, and 256 parser threads on 16 CPUs only thrash.
static int usable_cpus()
{
    int n = (int)std::max(1u, std::thread::hardware_concurrency());
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::min(n, std::max(1, CPU_COUNT(&set)));
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char quota[32] = {0};
        long period = 0;
        if (fscanf(f, "%31s %ld", quota, &period) == 2 && strcmp(quota, "max") != 0 && period > 0) {
            const long q = atol(quota);
            if (q > 0) n = std::min<long>(n, std::max<long>(1, (q + period - 1) / period));
        }
        fclose(f);
    }
    return n;
}

namespace {
// "row N" for the whole file; inside a byte range (sharded call_mods: ds_tsv_set_range) the row count is relative to the
// range, so the message names the range and the absolute byte offset of the line -- `tail -c +$((OFFSET+1)) file | head -1` (tail -c +K is 1-based)
std::string row_where(const ds_tsv* t, int64_t index0, const char* line)
{
    std::string w = "row " + std::to_string(index0 + 1);
    if (t->range_begin != 0 || t->limit != t->size)
        w += " of the byte range [" + std::to_string(t->range_begin) + ", " + std::to_string(t->limit) + ")";
    return w + " (line at byte offset " + std::to_string((long long)(line - t->data)) + ")";
}
}  // namespace

namespace {
// [begin, end) of column 5 (the read id) of the line starting at p; false when the line has fewer than 5 columns
bool line_read_id(const char* p, const char* le, const char** ib, const char** ie)
{
    const char* c = p;
    int tabs = 0;
    while (tabs < 4 && c < le) { c = find_tab(c, le); if (c < le) { ++c; ++tabs; } }
    if (tabs < 4) return false;
    *ib = c; *ie = find_tab(c, le);
    return true;
}

// Locate the lines of the next window of the file ON THE WHOLE TEAM and append them to t->pend. A row is ~3.8 KB of text
// and the calling thread used to walk all of it (memchr for the newline, and -- first touch of a freshly mapped file --
// one page fault per row): 0.6 us per row that no number of parser threads could shrink, against 4 us / nthreads for the
// parsing itself (tools/e2e_profile.py: 1.18 M rows/s on 16 threads). Now the window is cut into one chunk per thread, every
// thread finds the newlines of its chunk (and takes its faults), and the caller only strings the offsets together.
void scan_lines(ds_tsv* t)
{
    if (t->pend_head) {                         // drop what has been handed out
        t->pend.erase(t->pend.begin(), t->pend.begin() + (ptrdiff_t)t->pend_head);
        t->pend_head = 0;
    }
    const char* data = t->data;
    size_t target = t->scan_bytes;
    for (;;) {
        const size_t a = t->scanned, e0 = std::min(t->limit, a + target);
        const int nt = (int)std::max<size_t>(1, std::min<size_t>((size_t)t->nthreads, (e0 - a) >> 18));      // >= 256 KB per thread
        std::vector<std::vector<size_t>> found((size_t)nt);
        std::atomic<int> next(0);
        auto work = [&]() {
            for (;;) {
                const int c = next.fetch_add(1);
                if (c >= nt) break;
                const size_t b0 = a + (e0 - a) * (size_t)c / (size_t)nt, b1 = a + (e0 - a) * (size_t)(c + 1) / (size_t)nt;
                std::vector<size_t>& v = found[(size_t)c];
                v.reserve((b1 - b0) / 2048 + 16);
                const char* p = data + b0;
                const char* const e = data + b1;
                while (p < e) {
                    const char* nl = (const char*)memchr(p, '\n', (size_t)(e - p));
                    if (!nl) break;
                    v.push_back((size_t)(nl - data));
                    p = nl + 1;
                }
            }
        };
        if (nt <= 1) work();
        else {
            if (!t->team) t->team = new ds_team(t->nthreads - 1);
            const std::function<void()> f = work;
            t->team->run(nt - 1, f);
        }
        size_t prev = a;
        auto add = [&](size_t b, size_t e) {
            while (e > b && data[e - 1] == '\r') --e;
            if (e > b) t->pend.emplace_back(data + b, data + e);        // blank lines are skipped
        };
        bool any = false;
        for (const auto& v : found)
            for (size_t nl : v) { add(prev, nl); prev = nl + 1; any = true; }
        if (e0 == t->limit) {                   // the last line may lack its newline
            if (prev < e0) add(prev, e0);
            t->scanned = t->limit;
            return;
        }
        if (any) { t->scanned = prev; return; }
        target *= 2;                            // one line longer than the window: look further
    }
}
}  // namespace

extern "C" {

int ds_tsv_open(const char* path, int32_t kmer_len, int32_t signal_len, int32_t nthreads, ds_tsv** out)
{
    if (!path || !out || kmer_len < 1 || signal_len < 1) return DS_ERR_INVALID;
    *out = nullptr;
    ds_tsv* t = new ds_tsv();
    t->kmer_len = kmer_len; t->signal_len = signal_len;
    t->nthreads = nthreads > 0 ? nthreads : usable_cpus();
    t->fd = open(path, O_RDONLY);
    if (t->fd < 0) { delete t; return DS_ERR_IO; }
    struct stat st;
    if (fstat(t->fd, &st) != 0) { close(t->fd); delete t; return DS_ERR_IO; }
    t->size = (size_t)st.st_size;
    if (t->size) {
        void* m = mmap(nullptr, t->size, PROT_READ, MAP_PRIVATE, t->fd, 0);
        if (m == MAP_FAILED) { close(t->fd); delete t; return DS_ERR_IO; }
        madvise(m, t->size, MADV_SEQUENTIAL);
        t->data = (const char*)m;
    }
    t->limit = t->size;
    *out = t;
    return DS_OK;
}

void ds_tsv_close(ds_tsv* t)
{
    if (!t) return;
    delete t->team;
    if (t->data) munmap((void*)t->data, t->size);
    if (t->fd >= 0) close(t->fd);
    delete t;
}

const char* ds_tsv_error(const ds_tsv* t) { return t ? t->err.c_str() : "null reader"; }

// Locate (not parse) the rows of the next queue item: all rows of the next `max_reads` reads (a read = maximal run of
// consecutive rows with the same column 5). Returns their number (0 at end of file, negative on a row with fewer than five
// columns); ds_tsv_parse_into() parses them.
int64_t ds_tsv_locate(ds_tsv* t, int32_t max_reads)
{
    if (!t || max_reads < 1) return DS_ERR_INVALID;
    if (!t->lines.empty()) {      // exactly one ds_tsv_parse_into() per successful ds_tsv_locate(): a second locate would drop the item
        t->err = "ds_tsv_locate: the rows of the previous ds_tsv_locate() have not been parsed (ds_tsv_parse_into), or their parse failed";
        return DS_ERR_INVALID;
    }
    if (t->scanned < t->pos || t->pend_head > t->pend.size()) { t->scanned = t->pos; t->pend.clear(); t->pend_head = 0; }
    const char* prev_id = nullptr;
    size_t prev_len = 0;
    int reads = 0;
    bool full = false;
    const size_t item_begin = t->pend_head < t->pend.size() ? (size_t)(t->pend[t->pend_head].first - t->data) : t->scanned;
    for (;;) {
        // group the located lines into reads (a few bytes per row: the newline search itself ran on the whole team)
        while (t->pend_head < t->pend.size()) {
            const char* p = t->pend[t->pend_head].first;
            const char* ltrim = t->pend[t->pend_head].second;
            const char *c, *ide;
            if (!line_read_id(p, ltrim, &c, &ide)) {
                t->err = row_where(t, t->line_no + (int64_t)t->lines.size(), p) + ": fewer than 5 columns";
                return DS_ERR_IO;
            }
            const size_t idl = (size_t)(ide - c);
            if (!prev_id || idl != prev_len || memcmp(c, prev_id, idl) != 0) {
                if (prev_id) {
                    ++reads;
                    if (reads % max_reads == 0) { full = true; break; }     // this row starts the next item
                }
                prev_id = c; prev_len = idl;
            }
            t->lines.emplace_back(p, ltrim);
            ++t->pend_head;
        }
        if (full || t->scanned >= t->limit) break;
        scan_lines(t);
    }
    t->pos = t->pend_head < t->pend.size() ? (size_t)(t->pend[t->pend_head].first - t->data) : t->scanned;
    if (t->pos > item_begin) t->scan_bytes = std::max<size_t>(1u << 20, (t->pos - item_begin) + (t->pos - item_begin) / 4);
    return (int64_t)t->lines.size();
}

// Parse the rows ds_tsv_locate() found into the CALLER's arrays (kmer int32[n,kmer_len], means / stds / lens
// float[n,kmer_len], signals float[n,signal_len], labels int32[n]) on the team; the six sampleinfo columns go to the
// reader's own buffer (ds_tsv_info / ds_tsv_info_offsets). Returns n, or a negative code on a malformed row.
int64_t ds_tsv_parse_into(ds_tsv* t, int64_t capacity_rows, int32_t* kmer, float* means, float* stds, float* lens, float* signals,
                          int32_t* labels)
{
    if (!t) return DS_ERR_INVALID;
    const size_t n = t->lines.size();
    if (n == 0) return 0;
    if (capacity_rows < (int64_t)n) {
        t->err = "ds_tsv_parse_into: the caller's arrays hold " + std::to_string(capacity_rows) + " rows, the located item has " + std::to_string(n);
        return DS_ERR_INVALID;
    }
    if (!kmer || !means || !stds || !lens || !signals || !labels) return DS_ERR_INVALID;
    const RowDest dest{kmer, means, stds, lens, signals, labels};
    // rows are dealt 16 at a time (~60 us of parsing): with 64-row grains a 1,000-row item was 16 grains, and 9 .. 15
    // threads finished no sooner than 8 (somebody always had two grains)
    constexpr size_t GRAIN = 16;
    std::vector<int64_t> ilen(n);
    std::atomic<size_t> next(0);
    std::atomic<int64_t> bad(-1);
    auto work = [&]() {
        for (;;) {
            const size_t i0 = next.fetch_add(GRAIN);
            if (i0 >= n) break;
            const size_t i1 = std::min(n, i0 + GRAIN);
            for (size_t i = i0; i < i1; ++i)
                if (!parse_row(t, dest, i, t->lines[i].first, t->lines[i].second, &ilen[i])) {
                    int64_t exp = -1;
                    bad.compare_exchange_strong(exp, (int64_t)i);
                }
        }
    };
    const int nt = (int)std::min<size_t>((size_t)t->nthreads, (n + GRAIN - 1) / GRAIN);
    if (nt <= 1) work();
    else {
        if (!t->team) t->team = new ds_team(t->nthreads - 1);
        const std::function<void()> f = work;
        t->team->run(nt - 1, f);
    }
    if (bad.load() >= 0) {
        t->err = row_where(t, t->line_no + bad.load(), t->lines[(size_t)bad.load()].first) + ": malformed feature row";
        return DS_ERR_IO;
    }
    t->info_off.resize(n + 1);
    int64_t tot = 0;
    for (size_t i = 0; i < n; ++i) { t->info_off[i] = tot; tot += ilen[i]; }
    t->info_off[n] = tot;
    t->info.resize((size_t)tot);
    for (size_t i = 0; i < n; ++i) memcpy(t->info.data() + t->info_off[i], t->lines[i].first, (size_t)ilen[i]);
    t->line_no += (int64_t)n;
    t->lines.clear();
    return (int64_t)n;
}

// Next queue item: all rows of the next `max_reads` reads (a read = maximal run of consecutive rows with the
// same column 5), parsed into the reader's own arrays (the ds_tsv_kmer ... accessors). Returns the number of sites, 0 at
// end of file, negative on a malformed row.
int64_t ds_tsv_next(ds_tsv* t, int32_t max_reads)
{
    const int64_t n = ds_tsv_locate(t, max_reads);
    if (n <= 0) return n;
    const size_t K = (size_t)t->kmer_len, S = (size_t)t->signal_len, m = (size_t)n;
    t->kmer.resize(m * K); t->means.resize(m * K); t->stds.resize(m * K); t->lens.resize(m * K);
    t->signals.resize(m * S); t->labels.resize(m);
    return ds_tsv_parse_into(t, n, t->kmer.data(), t->means.data(), t->stds.data(), t->lens.data(), t->signals.data(), t->labels.data());
}

int64_t ds_tsv_size(const ds_tsv* t) { return t ? (int64_t)t->size : DS_ERR_INVALID; }

// First read boundary at or after byte `pos`: the start of the first line that begins at or after `pos` and whose
// read id (column 5) differs from the line before it -- so [align(a), align(b)) always holds whole reads, and the
// ranges of consecutive nominal offsets tile the file exactly. 0 stays 0; returns the file size when no boundary
// follows. A function of the file alone: every rank computes the same cut points without talking to the others.
int64_t ds_tsv_align(const ds_tsv* t, int64_t pos)
{
    if (!t || pos < 0) return DS_ERR_INVALID;
    if (pos == 0) return 0;
    if ((size_t)pos >= t->size) return (int64_t)t->size;
    const char* data = t->data;
    const char* end = data + t->size;
    // start of the last non-blank line that begins before pos (the "previous" line of the first candidate)
    auto line_start = [&](const char* x) { const char* r = x - 1; while (r > data && r[-1] != '\n') --r; return r; };   // x > data
    auto is_blank = [&](const char* b) { const char* e = b; while (e < end && *e != '\n') { if (*e != '\r') return false; ++e; } return true; };
    const char* q = line_start(data + pos);
    while (q > data && is_blank(q)) q = line_start(q);
    const char* prev_b = nullptr; const char* prev_e = nullptr;
    bool have_prev = false;
    const char* p = q;
    bool first = true;
    while (p < end) {
        const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
        const char* le = nl ? nl : end;
        const char* ltrim = le;
        while (ltrim > p && ltrim[-1] == '\r') --ltrim;
        if (ltrim > p) {                                    // blank lines neither start nor end a read
            const char *ib, *ie;
            if (!line_read_id(p, ltrim, &ib, &ie)) return (int64_t)(p - data);      // malformed: the parser reports it
            if (!first && p >= data + pos) {
                if (!have_prev || (size_t)(ie - ib) != (size_t)(prev_e - prev_b) || memcmp(ib, prev_b, (size_t)(ie - ib)) != 0)
                    return (int64_t)(p - data);
            }
            prev_b = ib; prev_e = ie; have_prev = true;
            first = false;
        }
        p = nl ? nl + 1 : end;
    }
    return (int64_t)t->size;
}

// Restrict the reader to the byte range [begin, end) (both from ds_tsv_align) and rewind to its start.
int ds_tsv_set_range(ds_tsv* t, int64_t begin, int64_t end)
{
    if (!t || begin < 0 || end < begin || (size_t)end > t->size) return DS_ERR_INVALID;
    t->pos = (size_t)begin;
    t->limit = (size_t)end;
    t->scanned = t->pos; t->pend.clear(); t->pend_head = 0;
    t->lines.clear();              // a rewind also drops located-but-unparsed rows (and the error state a failed parse left)
    t->range_begin = (size_t)begin;
    t->line_no = 0;
    t->err.clear();
    return DS_OK;
}

const int32_t* ds_tsv_kmer(const ds_tsv* t) { return t->kmer.data(); }
const float* ds_tsv_means(const ds_tsv* t) { return t->means.data(); }
const float* ds_tsv_stds(const ds_tsv* t) { return t->stds.data(); }
const float* ds_tsv_lens(const ds_tsv* t) { return t->lens.data(); }
const float* ds_tsv_signals(const ds_tsv* t) { return t->signals.data(); }
const int32_t* ds_tsv_labels(const ds_tsv* t) { return t->labels.data(); }
const char* ds_tsv_info(const ds_tsv* t) { return t->info.data(); }
const int64_t* ds_tsv_info_offsets(const ds_tsv* t) { return t->info_off.data(); }

// Result rows of one batch (call_modifications.py:183-190). Returns bytes written (each row ends with '\n'),
// or -(needed bytes) when `cap` is too small.
int64_t ds_format_rows(int64_t n, const char* info, const int64_t* info_off, const float* act, int32_t class_num,
                       const int32_t* pred, const int32_t* kmer, int32_t kmer_len, char* out, int64_t cap)
{
    if (n < 0 || !info || !info_off || !act || !pred || !kmer || !out || class_num < 2) return DS_ERR_INVALID;
    const int64_t need = info_off[n] + n * (int64_t)(2 * 20 + 16 + kmer_len + 8);
    if (need > cap) return -need;
    static const char bases[5] = {'A', 'C', 'G', 'T', 'N'};
    char* p = out;
    for (int64_t i = 0; i < n; ++i) {
        const int64_t il = info_off[i + 1] - info_off[i];
        memcpy(p, info + info_off[i], (size_t)il);
        p += il;
        const float p0 = act[i * class_num], p1 = act[i * class_num + 1];
        const float s = p0 + p1;                      // float32 arithmetic, as the reference's np.float32 scalars
        *p++ = '\t'; p += format_f32(p0 / s, p);
        *p++ = '\t'; p += format_f32(p1 / s, p);
        *p++ = '\t'; p += snprintf(p, 16, "%d", pred[i]);
        *p++ = '\t';
        for (int k = 0; k < kmer_len; ++k) { const int c = kmer[i * kmer_len + k]; *p++ = (c >= 0 && c < 5) ? bases[c] : 'N'; }
        *p++ = '\n';
    }
    return (int64_t)(p - out);
}


// CRC-32C, slicing-by-8 (tables built on first use). TF checkpoint tensors / table blocks carry it masked.
uint32_t ds_crc32c(const void* data, size_t n, uint32_t crc)
{
    static uint32_t T[8][256];
    static bool ready = false;
    if (!ready) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
            T[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = T[0][i];
            for (int k = 1; k < 8; ++k) { c = T[0][c & 0xFF] ^ (c >> 8); T[k][i] = c; }
        }
        ready = true;
    }
    const unsigned char* p = static_cast<const unsigned char*>(data);
    crc = ~crc;
    while (n >= 8) {
        uint32_t lo, hi;
        memcpy(&lo, p, 4);
        memcpy(&hi, p + 4, 4);
        lo ^= crc;
        crc = T[7][lo & 0xFF] ^ T[6][(lo >> 8) & 0xFF] ^ T[5][(lo >> 16) & 0xFF] ^ T[4][lo >> 24] ^
              T[3][hi & 0xFF] ^ T[2][(hi >> 8) & 0xFF] ^ T[1][(hi >> 16) & 0xFF] ^ T[0][hi >> 24];
        p += 8; n -= 8;
    }
    while (n--) crc = T[0][(crc ^ *p++) & 0xFF] ^ (crc >> 8);
    return ~crc;
}

}  // extern "C"
