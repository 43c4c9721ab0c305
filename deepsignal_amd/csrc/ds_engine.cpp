// ds_engine.cpp — host side of libdeepsignal_hip.so: handle, weight folding / MFMA operand
// packing, launch planning (two HIP streams + optional hipGraph) and the extern "C" ABI declared
// in include/deepsignal_hip.h. The math it schedules restates /root/reference/deepsignal/model.py
// and layers.py (cited per stage below); nothing here falls back to a CPU path.
#include "../../include/deepsignal_hip.h"
#include "ds_internal.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <string>
#include <vector>

using namespace ds;

namespace {

constexpr int VOCAB = 1024, EMB = 128, HID = 256, NLAYER = 3, NMOD = 11, INC_OUT = 240;
constexpr double BN_EPS = 1e-3;

thread_local std::string g_create_error;

void same_pad(int in, int k, int s, int* out, int* pl)
{
    int o = (in + s - 1) / s;
    int tot = std::max((o - 1) * s + k - in, 0);
    *out = o;
    *pl = tot / 2;
}

struct HostTensor {
    std::vector<int64_t> shape;
    std::vector<float> data;
};

struct PackedGemm {      // device-resident packed weights of one GEMM
    float* Bp = nullptr;
    float* bias = nullptr;
    int K = 0, N = 0;
    float* Bps = nullptr;   // DS_PRECISION_BF16X3: the same matrix as three bf16 term panels (pack_b_split)
};

enum OpKind { OP_GEMM, OP_STEM1, OP_MAXPOOL, OP_AVGPOOL, OP_HEAD, OP_FUSED, OP_PACKEV, OP_LSTM, OP_STEM23, OP_HEADF, OP_DENSES, OP_XPROJ };

struct Op {
    OpKind kind;
    int stream;          // 0 = signal/joint stream, 1 = event (BiLSTM) stream
    int stage;
    GemmCfg cfg;
    int launch_index;    // index into the plan's GemmLaunch array
    int total_tiles;
    // elementwise params
    const float* in = nullptr;
    float* out = nullptr;
    int a = 0, b = 0, c = 0, d = 0;
    FusedArgs fa{};                       // OP_FUSED (bf16 modes: the first module of the chain)
    FusedChain fc{};                      // OP_FUSED, bf16 modes: consecutive modules of one width class in ONE launch
    Stem23Args sa{};                      // OP_STEM23
    HeadFoldedArgs ha{};                  // OP_HEADF
    SplitDense sd{};                      // OP_DENSES
    LstmXproj xp{};                       // OP_XPROJ
    int tm = 0;
    double flops = 0;                     // algorithmic FLOPs of this launch
    hipEvent_t ev0 = nullptr, ev1 = nullptr;   // profiling mode only
    bool pending = false;
    int run_launches = 0;                 // profiling mode 1: launches / FLOPs bracketed by this op's event pair
    double run_flops = 0;
};

// per-kernel accumulators (one entry per __global__ function / template instantiation)
enum KernelClass { K_GEMM_CONV = 0, K_GEMM_FC, K_GEMM_LSTM, K_GEMM_CONV_WIDE, K_GEMM_CONV_POOL, K_GEMM_FC_DENSE, K_GEMM_LSTM_DENSE, K_FUSED1, K_FUSED2, K_FUSED3, K_STEM1, K_MAXPOOL, K_AVGPOOL, K_HEAD,
                   K_GEMM_BCONV, K_GEMM_BCONV_POOL, K_GEMM_BFC, K_GEMM_BFC_DENSE, K_PACKEV, K_GEMM_BLSTM, K_GEMM_BLSTM_DENSE, K_FUSEDB1, K_FUSEDB2, K_FUSEDB3, K_GEMM_LSTM_T, K_GEMM_LSTM_T_DENSE, K_GEMM_BLSTM_T, K_GEMM_BLSTM_T_DENSE, K_LSTM_CELL1, K_LSTM_CELL2, K_LSTM_CELL4, K_LSTM_LDS1, K_LSTM_LDS2, K_STEM23, K_HEADF, K_LSTM_B11, K_LSTM_B12, K_LSTM_B22, K_STEM23B, K_FUSEDS1, K_FUSEDS2, K_FUSEDS3, K_LSTM_S11, K_LSTM_S12, K_LSTM_S22, K_DENSE_SPLIT, K_STEM23S, K_LSTM_XPROJ, K_LSTM_S28, K_COUNT };
const char* const kKernelNames[K_COUNT] = {"gemm_kernel<1,2,4,1,0,0,1,1>", "gemm_kernel<1,3,4,1,0,0,2,1>", "gemm_kernel<1,4,4,1,1,0,1,1>",
                                           "gemm_kernel<2,2,2,2,0,0,1,1>", "gemm_kernel<1,2,4,1,0,1,1,1>", "gemm_kernel<1,3,4,1,0,2,2,1>",
                                           "gemm_kernel<1,4,4,1,1,2,1,1>", "inception_fused_kernel<1>",
                                           "inception_fused_kernel<2>", "inception_fused_kernel<3>", "stem1_kernel",
                                           "maxpool_s2_kernel", "avgpool7_kernel", "head_kernel",
                                           "gemm_kernel<1,2,4,1,0,0,1,1,bf16>", "gemm_kernel<1,2,4,1,0,1,1,1,bf16>",
                                           "gemm_kernel<4,2,1,4,0,0,2,1,bf16>", "gemm_kernel<4,2,1,4,0,2,2,1,bf16>",
                                           "pack_event_feat_bf16_kernel", "gemm_kernel<1,4,4,1,1,0,3,1,bf16>",
                                           "gemm_kernel<1,4,4,1,1,2,3,1,bf16>", "inception_fused_bf16_kernel<1>",
                                           "inception_fused_bf16_kernel<2>", "inception_fused_bf16_kernel<3>",
                                           "gemm_kernel<1,1,4,1,2,0,1,1>", "gemm_kernel<1,1,4,1,2,2,1,1>",
                                           "gemm_kernel<1,1,4,1,2,0,3,1,bf16>", "gemm_kernel<1,1,4,1,2,2,3,1,bf16>",
                                           "lstm_cell_kernel<1>", "lstm_cell_kernel<2>", "lstm_cell_kernel<4>",
                                           "lstm_cell_lds_kernel<1>", "lstm_cell_lds_kernel<2>", "stem23_kernel", "head_folded_kernel",
                                           "lstm_cell_bf16_kernel<1,1>", "lstm_cell_bf16_kernel<1,2>", "lstm_cell_bf16_kernel<2,2>", "stem23_bf16_kernel",
                                           "inception_fused_split_kernel<1>", "inception_fused_split_kernel<2>", "inception_fused_split_kernel<3>",
                                           "lstm_cell_split_kernel<1,1>", "lstm_cell_split_kernel<1,2>", "lstm_cell_split_kernel<2,2>",
                                           "dense_split_kernel (+ pack_joint_split_kernel)", "stem23_split_kernel", "lstm_xproj_kernel", "lstm_cell_split_kernel<1,2,4,2>"};
struct KernelStat {
    int64_t launches = 0;
    double total_ms = 0;
    double flops = 0;
};

struct Stage {
    std::string name;
    int stream = 0;
    int launches = 0;
    double flops_per_site = 0;
    double total_ms = 0;
    int64_t calls = 0;
};

#define DS_SPLIT_DENSE_PARTS 4       // K ranges of the split dense at small forwards (fc1o holds that many partial products)

struct Plan {
    int n = 0;
    std::vector<GemmLaunch> launches;     // host copy
    GemmLaunch* d_launches = nullptr;
    std::vector<LstmLaunch> lstm_launches;   // fp32 BiLSTM diagonals (lstm_cell_*kernel): passed by value at launch
    std::vector<Op> ops;                  // merged issue order
    int fc1_parts = 1;                    // partial products the split dense leaves in fc1o (launch_head adds them up)
    hipGraphExec_t graph = nullptr;
    int64_t uses = 0, last_use = 0;       // ragged tails produce many one-off sizes: graphs are captured for sizes that
                                          // recur, and the per-slot plan cache is bounded (get_plan)
};

}  // namespace

// One in-flight forward: private workspace, stream pair, fork/join events and captured graphs.
struct Slot {
    hipStream_t s0 = nullptr, s1 = nullptr;
    bool owns_s1 = true;                 // false: DS_TUNE_SHARED_EVENT_STREAM -- s1 is slot 0's event-model stream
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    float* d_in = nullptr;               // [kmer | means | stds | sanums | signals] (the five pointers below point into it)
    int* d_kmer = nullptr;
    float *d_means = nullptr, *d_stds = nullptr, *d_sanums = nullptr, *d_signals = nullptr;
    float *stem_pool = nullptr, *conv2o = nullptr, *conv3o = nullptr;
    float* modout[NMOD] = {nullptr};
    float *pool2 = nullptr, *pool3 = nullptr;
    float *tmpA = nullptr, *tmpS = nullptr, *tmpB = nullptr;
    float* sigfeat = nullptr;
    float* H[2][NLAYER] = {{nullptr}};   // h of every step: fp32 cells [T][m-tiles][32 k-groups][64][4] (MFMA-fragment-major,
                                         // ds_internal.h LstmCell); bf16-operand cells [T][B][256] bf16 row-major
    float* Cst[2][NLAYER] = {{nullptr}}; // cell state, same layout as one step of H (fp32)
    float* xproj[2] = {nullptr, nullptr};   // split cells: layer 0's accumulator-initial values of every step (lstm_xproj_kernel), [T][Bp32][1024]
    float* hlast[2] = {nullptr, nullptr};   // fp32 cells: row-major [B][256] copy of the top layer's final h (fw t = T-1, bw t = 0)
    float *fc1o = nullptr, *logits = nullptr, *act = nullptr;
    int* pred = nullptr;
    float* joint = nullptr;              // bf16 mode: [B][JP] bf16 FC operand (event features | signal features | zero pad)
    char* jsplit = nullptr;              // DS_PRECISION_BF16X3, three-step joint model: the joint rows as a fragment-major term image

    // ds_submit / ds_wait: pinned host staging of one batch (inputs in, 12 B/site out), allocated on first use
    char* pin_in = nullptr;
    float* pin_act = nullptr;
    int* pin_pred = nullptr;
    int submitted_n = -1;                 // sites of the forward in flight on this slot (-1: none)

    std::map<int, Plan> plans;
    int last_n = 0;
    int last_fc1_parts = 1;               // partial products the last forward's dense left in fc1o (kept here: its plan may be evicted)
};

// dense(J, J) with split operands moves 1.5x the operand bytes of the fp32 GEMM for 0.375x its matrix time and is bound by operand
// delivery; measured on one box (us per forward at 512 / 2,048 sites): split 245 / 725, native fp32 GEMM 292 / 1,148. The planner uses
// it from DS_SPLIT_DENSE_MIN_N sites per forward (ds_config.reserved[6] overrides).
#ifndef DS_SPLIT_DENSE_MIN_N
#define DS_SPLIT_DENSE_MIN_N 1
#endif
struct ds_handle {
    ds_config cfg{};
    std::string err;
    int T = 17, S = 360, C = 2;
    int w1 = 0, wa = 0, wb = 0, wc = 0, J = 0, SF = 0;
    int pl_conv1 = 0, pl_pool1 = 0, pl_pool2 = 0, pl_pool3 = 0;
    int B = 512;
    bool is_cnn = true, is_rnn = true, is_base = true;   // model.py:28-29,59-75,89-95
    bool bf16 = false;    // DS_PRECISION_BF16: bf16 conv + FC operands (fp32 accumulate), fp32 BiLSTM
    bool split_dense_narrow = false;      // DS_TUNE_SPLIT_DENSE_NARROW
    bool lstm_xproj = false;              // split cells: the first step's layer-0 cells (no matrix product) by lstm_xproj_kernel instead of a diagonal of their own
    bool lstm_xproj_all = false;          // ... and layer 0's accumulator-initial values of ALL steps as an image (DS_TUNE_LSTM_XPROJ_ALL)
    int split_dense_min_n = DS_SPLIT_DENSE_MIN_N;   // sites per forward from which dense(J, J) runs split (below: the native fp32 GEMM)
    bool split = false;   // DS_PRECISION_BF16X3: fp32 activations / weights carried as three bf16 terms through the bf16 matrix pipe
                          // (six products per MAC, fp32 accumulate) in the fused inception chains; everything else as fp32
    int lstm_variant = 0;     // ds_config.reserved[3] as given (DS_LSTM_TILING_*)
    // per-handle tuning / diagnostic knobs, all from ds_config.reserved[2..5] (include/deepsignal_hip.h)
    bool no_fused = false;    // DS_TUNE_NO_FUSED: layer-granular inception modules instead of the fused kernel
    bool fold_fc = false;     // joint model folded into one J x class_num matrix (fp32, not DS_TUNE_NO_FOLD_FC, not debug)
    bool serial = false;      // DS_TUNE_SERIAL: every launch of a forward on ONE stream (stand-alone kernel times)
    bool serial_modules = false;   // DS_TUNE_NO_CHAIN: one launch per inception module instead of one per width class
    bool shared_s1 = false;        // DS_TUNE_SHARED_EVENT_STREAM: every slot's BiLSTM on slot 0's event-model stream, eager issue
    int fuse_max_spt = 8;     // sites per fused-module tile, upper bound
    int fuse_min_tiles = 128; // fused-module grids keep at least this many workgroups when the batch allows it
    bool lstm_bf16 = false;   // DS_PRECISION_BF16_ALL: additionally bf16 h / weight operands in the LSTM matmuls (fp32 accumulate,
                              // gates and cell state; the layer-0 input projection stays an fp32 table lookup)
    bool lstm_frag = false;   // fp32 BiLSTM cells: lstm_cell_kernel on fragment-major h / c (every mode but DS_PRECISION_BF16_ALL)
    int Bp32 = 0;             // max_batch rounded up to whole 32-site m-tiles (rows of the fragment-major buffers)
    int JP = 0;           // J rounded up to a whole K chunk (32 bf16)
    bool finalized = false;
    bool debug = false;
    int profiling = 0;    // 0 off | 1 one event pair per run of same-kernel launches on a stream | 2 per launch
                          // | 3 like 1 with every launch on ONE stream (stand-alone kernel times, nothing co-resident)
    bool use_graph = true;
    std::map<std::string, HostTensor> host;
    std::vector<void*> allocs;

    // packed weights
    float *stem1_w = nullptr, *stem1_b = nullptr;
    PackedGemm conv2, conv3;
    PackedGemm m_f1[NMOD];   // fused-module stage 1: [b5s|b2|b3a|b4a|b5a|b1]
    PackedGemm m_s1[NMOD], m_b1[NMOD], m_b3b[NMOD], m_b4b[NMOD], m_b5b[NMOD], m_b5c[NMOD];
    PackedGemm lstm_n[2][NLAYER];   // LSTM kernels packed [gate][8 units] per n-tile (fp32 or bf16 fragments)
    float* lstm_table[2] = {nullptr, nullptr};
    float* lstm_wfeat[2] = {nullptr, nullptr};
    PackedGemm fc1;
    float* fc2 = nullptr;
    float* w12f = nullptr;    // fold_fc: [J][C] = (avgpool^T on the signal rows) (W1 W2), float64 product rounded once

    bool debug_keep_pool = false;   // keep the stand-alone maxpool kernels (diagnostic)
    const float* zero_seg = nullptr;
    unsigned long long* dbg_stamps = nullptr;   // [NMOD][1024 wgs][2 waves][8] when DS_TUNE_DEBUG_STAMPS is set
    unsigned long long* dbg_lstm = nullptr;     // [32 diagonals][1024 wgs][8] stamps of the fp32 BiLSTM cell launches
    std::vector<Stage> stages;
    KernelStat kstat[K_COUNT];
    // pipelining: consecutive forwards rotate over independent slots (own workspace, streams, graphs), so the
    // dependency chain of one 512-site forward overlaps the next ones'; weights are shared
    std::vector<Slot> slots;
    Slot* cur = nullptr;
    unsigned next_slot = 0;
    int64_t plan_tick = 0;    // LRU clock of the per-slot plan caches
    bool stages_done = false;
};

namespace {

int fail(ds_handle* h, int code, const std::string& msg)
{
    if (h) h->err = msg; else g_create_error = msg;
    return code;
}

#define HIPCHK(h, expr)                                                                                  \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess) {                                                                          \
            (void)hipGetLastError(); /* the thread's sticky error: the launchers end with hipGetLastError() */ \
            return fail(h, DS_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));              \
        }                                                                                                \
    } while (0)

template <class T>
int dalloc(ds_handle* h, T** p, size_t count)
{
    void* q = nullptr;
    size_t bytes = std::max<size_t>(count * sizeof(T), 256);
    hipError_t e = hipMalloc(&q, bytes);
    if (e != hipSuccess) {
        // hipGetLastError() returns (and clears) the thread's last error; every launcher ends with it. Left in place, a failed
        // hipMalloc would be reported by the first launch of the NEXT handle this thread creates (call_modifications.make_engine's
        // fall-back to the user's batch size after an out-of-memory ds_create)
        (void)hipGetLastError();
        return fail(h, DS_ERR_NOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
    }
    h->allocs.push_back(q);
    *p = static_cast<T*>(q);
    return DS_OK;
}

int upload(ds_handle* h, float** dst, const std::vector<float>& v)
{
    int rc = dalloc(h, dst, v.size());
    if (rc) return rc;
    HIPCHK(h, hipMemcpy(*dst, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
    return DS_OK;
}

// Pack a logical [K][N] matrix into MFMA-fragment order: [ntile][kgroup][lane][4] where
// lane (j = lane&31, half = lane>>5) element s holds W[kgroup*8 + 4*half + s][ntile*32 + j].
std::vector<float> pack_b(int K, int N, const std::function<float(int, int)>& w_in)
{
    // K is padded to a multiple of 32 with zero rows so K-split kernels may run one extra (zero) chunk
    const int Kp = (K + 31) / 32 * 32;
    auto w = [&](int k, int col) { return k < K ? w_in(k, col) : 0.0f; };
    const int ntiles = (N + 31) / 32, kg = Kp / 8;
    std::vector<float> out((size_t)ntiles * kg * 256, 0.0f);
    for (int nt = 0; nt < ntiles; ++nt)
        for (int g = 0; g < kg; ++g)
            for (int lane = 0; lane < 64; ++lane) {
                const int col = nt * 32 + (lane & 31);
                if (col >= N) continue;
                const int k0 = g * 8 + 4 * (lane >> 5);
                float* o = &out[(((size_t)nt * kg + g) * 64 + lane) * 4];
                for (int s = 0; s < 4; ++s) o[s] = w(k0 + s, col);
            }
    return out;
}

uint16_t f32_to_bf16(float f)      // round to nearest even, like v_cvt_pk_bf16_f32
{
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // quiet NaN
    u += 0x7fffu + ((u >> 16) & 1);
    return (uint16_t)(u >> 16);
}

// bf16 operand packing, same byte geometry as pack_b: [ntile][kstep of 16][lane][8 bf16], lane (j, half) holds
// W[kstep*16 + 8*half + s][ntile*32 + j], s = 0..7 (the v_mfma_f32_32x32x16_bf16 B fragment). K is padded with
// zero rows to a multiple of 64 elements (= 32 four-byte units, like the fp32 packer).
std::vector<float> pack_b_bf16(int K, int N, const std::function<float(int, int)>& w_in)
{
    const int Kp = (K + 63) / 64 * 64;
    const int ntiles = (N + 31) / 32, ks = Kp / 16;
    std::vector<uint16_t> out((size_t)ntiles * ks * 64 * 8, 0);
    for (int nt = 0; nt < ntiles; ++nt)
        for (int g = 0; g < ks; ++g)
            for (int lane = 0; lane < 64; ++lane) {
                const int col = nt * 32 + (lane & 31);
                if (col >= N) continue;
                const int k0 = g * 16 + 8 * (lane >> 5);
                uint16_t* o = &out[(((size_t)nt * ks + g) * 64 + lane) * 8];
                for (int q = 0; q < 8; ++q) o[q] = k0 + q < K ? f32_to_bf16(w_in(k0 + q, col)) : 0;
            }
    std::vector<float> raw(out.size() / 2);
    memcpy(raw.data(), out.data(), out.size() * 2);
    return raw;
}

// Split-operand packing (DS_PRECISION_BF16X3, ds_split.hip): every weight w as three bf16 terms t0 = bf16(w), t1 = bf16(w - t0),
// t2 = bf16(w - t0 - t1) (the differences are exact in fp32, the terms sum to w exactly); layout [ntile][kstep of 16][term][lane][8 bf16],
// i.e. pack_b_bf16's fragments with the three terms of a k-step next to each other (3 KiB per wave and k-step, contiguous).
std::vector<float> pack_b_split(int K, int N, const std::function<float(int, int)>& w_in)
{
    const int Kp = (K + 63) / 64 * 64;
    const int ntiles = (N + 31) / 32, ks = Kp / 16;
    std::vector<uint16_t> out((size_t)ntiles * ks * 3 * 64 * 8, 0);
    for (int nt = 0; nt < ntiles; ++nt)
        for (int g = 0; g < ks; ++g)
            for (int lane = 0; lane < 64; ++lane) {
                const int col = nt * 32 + (lane & 31);
                if (col >= N) continue;
                const int k0 = g * 16 + 8 * (lane >> 5);
                for (int q = 0; q < 8; ++q) {
                    if (k0 + q >= K) continue;
                    float r = w_in(k0 + q, col);
                    for (int t = 0; t < 3; ++t) {
                        const uint16_t b = f32_to_bf16(r);
                        out[((((size_t)nt * ks + g) * 3 + t) * 64 + lane) * 8 + q] = b;
                        uint32_t u = (uint32_t)b << 16;
                        float f;
                        memcpy(&f, &u, 4);
                        r -= f;
                    }
                }
            }
    std::vector<float> raw(out.size() / 2);
    memcpy(raw.data(), out.data(), out.size() * 2);
    return raw;
}

struct FoldedConv {     // BN folded into the kernel: y = conv(x, w') + b'   (layers.py:80-84)
    int k, cin, cout;
    std::vector<float> w;   // [k*cin][cout]
    std::vector<float> b;   // [cout]
};

const HostTensor* find(ds_handle* h, const std::string& name)
{
    auto it = h->host.find(name);
    return it == h->host.end() ? nullptr : &it->second;
}

int fold_conv(ds_handle* h, const std::string& scope, const std::string& conv, const std::string& bn, int k, int cin,
              int cout, FoldedConv* out)
{
    const HostTensor* ker = find(h, scope + "/" + conv + "/kernel");
    const HostTensor* beta = find(h, scope + "/" + bn + "/beta");
    const HostTensor* gamma = find(h, scope + "/" + bn + "/gamma");
    const HostTensor* mean = find(h, scope + "/" + bn + "/moving_mean");
    const HostTensor* var = find(h, scope + "/" + bn + "/moving_variance");
    if (!ker || !beta || !gamma || !mean || !var)
        return fail(h, DS_ERR_INVALID, "missing tensor(s) under " + scope + "/" + conv);
    if ((int64_t)ker->data.size() != (int64_t)k * cin * cout || (int)beta->data.size() != cout ||
        (int)gamma->data.size() != cout || (int)mean->data.size() != cout || (int)var->data.size() != cout)
        return fail(h, DS_ERR_INVALID, "bad shape for " + scope + "/" + conv + " (kernel or one of beta/gamma/moving_mean/moving_variance)");
    out->k = k; out->cin = cin; out->cout = cout;
    out->w.resize((size_t)k * cin * cout);
    out->b.resize(cout);
    std::vector<double> sc(cout);
    for (int c = 0; c < cout; ++c) {
        sc[c] = (double)gamma->data[c] / std::sqrt((double)var->data[c] + BN_EPS);
        out->b[c] = (float)((double)beta->data[c] - (double)mean->data[c] * sc[c]);
    }
    for (size_t r = 0; r < (size_t)k * cin; ++r)
        for (int c = 0; c < cout; ++c) out->w[r * cout + c] = (float)((double)ker->data[r * cout + c] * sc[c]);
    return DS_OK;
}

// upload a GEMM whose columns are the concatenation of several folded convs with identical K
int upload_concat(ds_handle* h, const std::vector<const FoldedConv*>& parts, PackedGemm* pg, bool split_panel = false)
{
    const int K = parts[0]->k * parts[0]->cin;
    int N = 0;
    for (auto* p : parts) N += p->cout;
    std::vector<int> owner(N), local(N);
    int c0 = 0;
    for (size_t i = 0; i < parts.size(); ++i) {
        for (int c = 0; c < parts[i]->cout; ++c) { owner[c0 + c] = (int)i; local[c0 + c] = c; }
        c0 += parts[i]->cout;
    }
    auto wfun = [&](int k, int col) {
        const FoldedConv* p = parts[owner[col]];
        return p->w[(size_t)k * p->cout + local[col]];
    };
    // bf16 mode: the module input rows are stored with 256-channel pitch, so a cin = 240 operand is padded to 256
    // zero rows by the packer; K is then counted in 4-byte units (two bf16), see ds_kernels.hip
    std::vector<float> packed = h->bf16 ? pack_b_bf16(K, N, wfun) : pack_b(K, N, wfun);
    std::vector<float> bias((N + 31) / 32 * 32, 0.0f);
    for (int col = 0; col < N; ++col) bias[col] = parts[owner[col]]->b[local[col]];
    pg->K = h->bf16 ? (K + 1) / 2 : K; pg->N = N;
    int rc = upload(h, &pg->Bp, packed);
    if (rc) return rc;
    if (split_panel && (rc = upload(h, &pg->Bps, pack_b_split(K, N, wfun)))) return rc;
    return upload(h, &pg->bias, bias);
}

std::string mod_root(int n)
{
    char buf[128];
    snprintf(buf, sizeof buf, "modelsignalmincp_layer%d/modelsignalm%d", n, n);
    return buf;
}

int finalize_weights(ds_handle* h)
{
    int rc;
    if (h->is_cnn) {
    // ---- stem (layers.py:183-203) ----
    FoldedConv c1, c2, c3;
    if ((rc = fold_conv(h, "modelsignalmconv_layer1", "conv", "bn", 7, 1, 64, &c1))) return rc;
    if ((rc = fold_conv(h, "modelsignalmconv_layer2", "conv", "bn", 1, 64, 128, &c2))) return rc;
    if ((rc = fold_conv(h, "modelsignalmconv_layer3", "conv", "bn", 3, 128, 256, &c3))) return rc;
    if ((rc = upload(h, &h->stem1_w, c1.w))) return rc;
    if ((rc = upload(h, &h->stem1_b, c1.b))) return rc;
    if ((rc = upload_concat(h, {&c2}, &h->conv2, h->split))) return rc;
    if ((rc = upload_concat(h, {&c3}, &h->conv3, h->split))) return rc;
    // ---- inception modules (layers.py:87-139) ----
    for (int m = 0; m < NMOD; ++m) {
        const int cin = m == 0 ? 256 : INC_OUT;
        const std::string r = mod_root(m + 1);
        FoldedConv b1, b2, b3a, b3b, b4a, b4b, b5s, b5a, b5b, b5c;
        if ((rc = fold_conv(h, r + "branch1_maxpooling", "conv1a_1x1", "bn", 1, cin, 48, &b1))) return rc;
        if ((rc = fold_conv(h, r + "branch2_1x1", "conv0b_1x1", "bn", 1, cin, 48, &b2))) return rc;
        if ((rc = fold_conv(h, r + "branch3_1x3", "conv0c_1x1", "bn1", 1, cin, 32, &b3a))) return rc;
        if ((rc = fold_conv(h, r + "branch3_1x3", "conv1c_1x3", "bn2", 3, 32, 48, &b3b))) return rc;
        if ((rc = fold_conv(h, r + "branch4_1x5", "conv0d_1x1", "bn1", 1, cin, 32, &b4a))) return rc;
        if ((rc = fold_conv(h, r + "branch4_1x5", "conv1d_1x5", "bn2", 5, 32, 48, &b4b))) return rc;
        if ((rc = fold_conv(h, r + "branch5_residual_1x3", "convstem_1x1", "bn0", 1, cin, 48, &b5s))) return rc;
        if ((rc = fold_conv(h, r + "branch5_residual_1x3", "conv0e_1x1", "bn1", 1, cin, 32, &b5a))) return rc;
        if ((rc = fold_conv(h, r + "branch5_residual_1x3", "conv1e_1x3", "bn2", 3, 32, 64, &b5b))) return rc;
        if ((rc = fold_conv(h, r + "branch5_residual_1x3", "conv2e_1x1", "bn3", 1, 64, 48, &b5c))) return rc;
        // the five 1x1 convs that read the module input share one GEMM: [b2 | b5s | b3a | b4a | b5a]
        if ((rc = upload_concat(h, {&b2, &b5s, &b3a, &b4a, &b5a}, &h->m_s1[m]))) return rc;
        if ((rc = upload_concat(h, {&b1}, &h->m_b1[m]))) return rc;
        if ((rc = upload_concat(h, {&b5s, &b2, &b3a, &b4a, &b5a, &b1}, &h->m_f1[m], h->split))) return rc;
        if ((rc = upload_concat(h, {&b3b}, &h->m_b3b[m], h->split))) return rc;
        if ((rc = upload_concat(h, {&b4b}, &h->m_b4b[m], h->split))) return rc;
        if ((rc = upload_concat(h, {&b5b}, &h->m_b5b[m], h->split))) return rc;
        if ((rc = upload_concat(h, {&b5c}, &h->m_b5c[m], h->split))) return rc;
    }
    }
    if (h->is_rnn) {
    // ---- BiLSTM (layers.py:45-72; TF LSTMCell kernel rows = [input ; h], columns = [i j f o]) ----
    float* d_emb = nullptr;
    if (h->is_base) {
        const HostTensor* emb = find(h, "modelembedding");
        if (!emb || (int64_t)emb->data.size() != (int64_t)VOCAB * EMB) return fail(h, DS_ERR_INVALID, "missing modelembedding");
        if ((rc = upload(h, &d_emb, emb->data))) return rc;
    }
    const int in0 = h->is_base ? EMB + 3 : 3;      // layer-0 input width (model.py:63-75)
    const char* dirs[2] = {"fw", "bw"};
    for (int d = 0; d < 2; ++d)
        for (int l = 0; l < NLAYER; ++l) {
            char nm[160];
            snprintf(nm, sizeof nm, "modelem/%s/multi_rnn_cell/cell_%d/lstm_cell/kernel", dirs[d], l);
            const HostTensor* ker = find(h, nm);
            snprintf(nm, sizeof nm, "modelem/%s/multi_rnn_cell/cell_%d/lstm_cell/bias", dirs[d], l);
            const HostTensor* bias = find(h, nm);
            const int in = l == 0 ? in0 : HID;
            if (!ker || !bias || (int64_t)ker->data.size() != (int64_t)(in + HID) * 4 * HID || bias->data.size() != 4 * HID)
                return fail(h, DS_ERR_INVALID, std::string("missing/bad LSTM tensor ") + nm);
            const float* kd = ker->data.data();
            // rows fed through the MFMA GEMM: layer 0 -> only the h rows (x part is table + 3 rank-1 terms)
            const int row0 = l == 0 ? in0 : 0;
            const int K = l == 0 ? HID : 2 * HID;
            // CFG_LSTM_T: n-tile p holds units 8p..8p+7, column i of the tile = gate (i >> 3) of unit 8p + (i & 7)
            auto wfun_t = [&](int k, int pc) {
                const int p = pc / 32, i = pc % 32;
                return kd[(size_t)(row0 + k) * 4 * HID + (i >> 3) * HID + p * 8 + (i & 7)];
            };
            // every cell kernel (fp32 and bf16 operands) reads the [gate][8 units] layout
            {
                PackedGemm& pg = h->lstm_n[d][l];
                std::function<float(int, int)> wfun = wfun_t;
                std::vector<float> packed = h->lstm_bf16 ? pack_b_bf16(K, 4 * HID, wfun) : pack_b(K, 4 * HID, wfun);
                pg.K = h->lstm_bf16 ? K / 2 : K; pg.N = 4 * HID;
                if ((rc = upload(h, &pg.Bp, packed))) return rc;
                if (h->split && (rc = upload(h, &pg.Bps, pack_b_split(K, 4 * HID, wfun)))) return rc;
                if ((rc = upload(h, &pg.bias, bias->data))) return rc;
            }
            if (l == 0) {
                if (h->is_base) {
                    // embedding folded into W_x: table[v] = emb[v] @ kernel[0:128]   (model.py:61-69)
                    float* d_k0 = nullptr;
                    std::vector<float> k0(kd, kd + (size_t)EMB * 4 * HID);
                    if ((rc = upload(h, &d_k0, k0))) return rc;
                    if ((rc = dalloc(h, &h->lstm_table[d], (size_t)VOCAB * 4 * HID))) return rc;
                    HIPCHK(h, launch_embed_table(d_emb, d_k0, h->lstm_table[d], VOCAB, EMB, 4 * HID, h->cur->s0));
                }
                std::vector<float> wf(kd + (size_t)(in0 - 3) * 4 * HID, kd + (size_t)in0 * 4 * HID);
                if ((rc = upload(h, &h->lstm_wfeat[d], wf))) return rc;
            }
        }
    }
    // ---- joint FC (layers.py:247-264) ----
    const HostTensor* w1 = find(h, "dense/kernel");
    const HostTensor* w2 = find(h, "dense_1/kernel");
    const int J = h->J;
    if (!w1 || !w2 || (int64_t)w1->data.size() != (int64_t)J * J || (int64_t)w2->data.size() != (int64_t)J * h->C)
        return fail(h, DS_ERR_INVALID, "missing/bad dense kernels");
    if (h->fold_fc) {
        // W12 = W1 W2 in float64 (no bias, no activation, identity dropout between the two dense layers), then the
        // transpose of avgpool(7, SAME, divisor = in-bounds taps) applied to the signal rows: the head reads module 11's
        // rows directly
        const int C = h->C;
        std::vector<double> w12((size_t)J * C, 0.0);
        const float* a1 = w1->data.data();
        const float* a2 = w2->data.data();
        for (int k = 0; k < J; ++k) {
            double* o = &w12[(size_t)k * C];
            const float* r = a1 + (size_t)k * J;
            for (int m = 0; m < J; ++m) {
                const double v = r[m];
                for (int c = 0; c < C; ++c) o[c] += v * (double)a2[(size_t)m * C + c];
            }
        }
        // row pitch of module 11's rows as the head reads them: 240 channels (fp32) or the bf16 rows' 256 (pad channels: zero rows)
        const int rp = h->bf16 ? 256 : INC_OUT;
        const int ev = h->is_rnn ? 2 * HID : 0;
        std::vector<float> wf(((size_t)ev + (h->is_cnn ? (size_t)h->wc * rp : 0)) * C, 0.0f);
        for (int k = 0; k < ev; ++k)
            for (int c = 0; c < C; ++c) wf[(size_t)k * C + c] = (float)w12[(size_t)k * C + c];
        if (h->is_cnn) {
            const int wc = h->wc;
            for (int w = 0; w < wc; ++w)
                for (int ch = 0; ch < INC_OUT; ++ch)
                    for (int c = 0; c < C; ++c) {
                        double sacc = 0.0;
                        for (int wp = std::max(0, w - 3); wp <= std::min(wc - 1, w + 3); ++wp) {
                            const int cnt = std::min(wc - 1, wp + 3) - std::max(0, wp - 3) + 1;
                            sacc += w12[(size_t)(ev + wp * INC_OUT + ch) * C + c] / cnt;
                        }
                        wf[(size_t)(ev + w * rp + ch) * C + c] = (float)sacc;
                    }
        }
        if ((rc = upload(h, &h->w12f, wf))) return rc;
    } else {
        const float* wd = w1->data.data();
        auto wfun = [&](int k, int col) { return wd[(size_t)k * J + col]; };
        std::vector<float> packed = h->bf16 ? pack_b_bf16(J, J, wfun) : pack_b(J, J, wfun);
        h->fc1.K = h->bf16 ? h->JP / 2 : J; h->fc1.N = J;
        if ((rc = upload(h, &h->fc1.Bp, packed))) return rc;
        if (h->split && J % 16 == 0 && (rc = upload(h, &h->fc1.Bps, pack_b_split(J, J, wfun)))) return rc;
        h->fc1.bias = nullptr;
        if ((rc = upload(h, &h->fc2, w2->data))) return rc;
    }
    HIPCHK(h, hipStreamSynchronize(h->cur->s0));
    h->host.clear();
    h->finalized = true;
    return DS_OK;
}

int alloc_workspace(ds_handle* h)
{
    const size_t B = h->B;
    int rc = 0;
    auto A = [&](auto** p, size_t count) { if (!rc) rc = dalloc(h, p, count); };
    // inputs of a forward: ONE block [kmer | means | stds | sanums | signals], every region sized for max_batch -- the image
    // of the pinned staging buffer of ds_submit, so a full batch arrives with one H2D copy
    A(&h->cur->d_in, B * (4 * h->T + h->S));
    if (!rc) {
        h->cur->d_kmer = reinterpret_cast<int*>(h->cur->d_in);
        h->cur->d_means = h->cur->d_in + B * h->T; h->cur->d_stds = h->cur->d_in + 2 * B * h->T;
        h->cur->d_sanums = h->cur->d_in + 3 * B * h->T; h->cur->d_signals = h->cur->d_in + 4 * B * h->T;
    }
    A(&h->cur->stem_pool, B * h->wa * 64); A(&h->cur->conv2o, B * h->wa * 128); A(&h->cur->conv3o, B * h->wa * 256);
    A(&h->cur->pool2, B * h->wb * INC_OUT); A(&h->cur->pool3, B * h->wc * INC_OUT);
    A(&h->cur->tmpA, B * h->wa * 96); A(&h->cur->tmpS, B * h->wa * 48); A(&h->cur->tmpB, B * h->wa * 64);
    A(&h->cur->sigfeat, B * h->SF);
    for (int d = 0; d < 2; ++d)
        for (int l = 0; l < NLAYER; ++l) {
            // split cells keep h as three bf16 terms (6 bytes per unit) where the fp32 cells keep a float
            A(&h->cur->H[d][l], (size_t)h->T * h->Bp32 * HID * (h->split ? 3 : 2) / 2); A(&h->cur->Cst[d][l], (size_t)h->Bp32 * HID);
            if (l == 0 && h->lstm_xproj_all) A(&h->cur->xproj[d], (size_t)h->T * h->Bp32 * 4 * HID);
        }
    if (h->is_rnn)
        for (int d = 0; d < 2; ++d) A(&h->cur->hlast[d], B * HID);
    A(&h->cur->fc1o, B * h->J * (h->split && !h->fold_fc ? DS_SPLIT_DENSE_PARTS : 1)); A(&h->cur->logits, B * h->C);
    A(&h->cur->act, B * h->C + B);            // [act | pred]: one block, one D2H copy
    if (!rc) h->cur->pred = reinterpret_cast<int*>(h->cur->act + B * h->C);
    if (h->bf16) A(&h->cur->joint, B * (size_t)h->JP / 2);
    if (h->split && !h->fold_fc && h->J % 16 == 0) A(&h->cur->jsplit, (size_t)(h->Bp32 / 32) * (h->J / 16) * 3 * 1024);
    // module outputs: ping-pong pair normally; one buffer per module in debug mode (for taps)
    // the modules of a width class run as a chain inside one launch (ds_internal.h FusedChain): a workgroup that is already
    // in module 5 must not write into the buffer a slower workgroup still reads module 4's (stride-2 pooled, differently
    // tiled) input from, so a chain alternates between the two buffers that do NOT hold its first module's input: three buffers
    const int nbuf = h->debug ? NMOD : 3;
    static const int kChainBuf[NMOD] = {0, 1, 0, /* reads 0 */ 1, 2, 1, 2, 1, /* reads 1 */ 0, 2, 0};
    float* bufs[NMOD] = {nullptr};
    for (int i = 0; i < nbuf; ++i) A(&bufs[i], B * h->wa * INC_OUT);
    for (int m = 0; m < NMOD; ++m) h->cur->modout[m] = bufs[h->debug ? m : kChainBuf[m]];
    if (!rc && h->bf16) {
        // bf16 rows are [.., 256] with channels 240..255 (and the joint's tail) never written: they must read as zero
        hipError_t e = hipSuccess;
        for (int i = 0; i < nbuf && e == hipSuccess; ++i) e = hipMemset(bufs[i], 0, B * h->wa * INC_OUT * 4);
        if (e == hipSuccess) e = hipMemset(h->cur->pool2, 0, B * h->wb * INC_OUT * 4);
        if (e == hipSuccess) e = hipMemset(h->cur->pool3, 0, B * h->wc * INC_OUT * 4);
        if (e == hipSuccess) e = hipMemset(h->cur->joint, 0, B * (size_t)h->JP * 2);
        if (e != hipSuccess) rc = fail(h, DS_ERR_HIP, std::string("hipMemset: ") + hipGetErrorString(e));
    }
    return rc;
}

// workgroup tile of the split-operand BiLSTM cells by sites per forward (measured on MI355X). Stand-alone the
// 64 x 64 tile is the fastest below 2,048 sites (333 against 420 us per 512-site step: 768 small workgroups hide latency), but in the
// PIPELINED step -- where the cells share the CUs and the power budget with the other forwards' kernels -- the 128 x 128 tile's halved
// bytes per MFMA win: 647 k against 623 k sites/s (three-step), 849 k against 802 k (folded) at 512 sites, 8 slots. Tiny forwards keep
// the small tile (too few 128 x 128 workgroups to fill anything).
#ifndef DS_SPLIT_LSTM_TILE
#define DS_SPLIT_LSTM_TILE(n) ((n) >= 256 ? 322 : 311)
#endif

int module_width(const ds_handle* h, int m) { return m < 3 ? h->wa : (m < 8 ? h->wb : h->wc); }

// zero16: 16 zero floats on the handle's device (the A operand of the zero-padding segment K-split kernels append)
void add_tiles(GemmLaunch& L, GemmProblem& P, GemmCfg cfg, const float* zero16)
{
    const TileGeom g = gemm_geom(cfg);
    // K-split kernels need a chunk count divisible by the split: append a zero A segment (ld = 0, so
    // every row reads the same 16 zeros); the packed weights are zero-padded past K as well.
    while ((P.K / KC) % g.ksplit != 0) {
        ASeg& z = P.seg[P.nseg++];
        z.base = zero16; z.ld = 0; z.row_shift = 0; z.klen = KC;
        P.K += KC;
    }
    P.tiles_m = (P.M + g.bm - 1) / g.bm;
    P.tiles_n = (P.N + g.bn - 1) / g.bn;
    P.ntiles32 = (P.N + 31) / 32;
    P.n_fast = (double)P.M > (double)P.N ? 1 : 0;      // A bytes (M*K) vs B bytes (K*N)
    P.tile_start = L.total_tiles;
    L.total_tiles += P.tiles_m * P.tiles_n;
    L.prob[L.nprob++] = P;
}

GemmProblem base_problem(int M, int N, int W, const PackedGemm& pg)
{
    GemmProblem P;
    memset(&P, 0, sizeof P);
    P.M = M; P.N = N; P.W = W;
    P.Bp = pg.Bp; P.bias = pg.bias;
    P.kgroups_stride = (pg.K + 31) / 32 * 32 / 8;
    return P;
}

void add_seg(GemmProblem& P, const float* base, int ld, int shift, int klen)
{
    ASeg& s = P.seg[P.nseg++];
    s.base = base; s.ld = ld; s.row_shift = shift; s.klen = klen;
    P.K += klen;
}

void add_out(GemmProblem& P, float* base, int ld, int col0, int ncols, int relu, const float* add = nullptr, int add_ld = 0,
             int bf16 = 0)
{
    OSeg& o = P.out[P.nout++];
    o.base = base; o.ld = ld; o.col0 = col0; o.ncols = ncols; o.relu = relu; o.add = add; o.add_ld = add_ld; o.bf16 = bf16;
}

int stage_id(ds_handle* h, const std::string& name, int stream)
{
    for (size_t i = 0; i < h->stages.size(); ++i)
        if (h->stages[i].name == name) return (int)i;
    Stage s;
    s.name = name; s.stream = stream;
    h->stages.push_back(s);
    return (int)h->stages.size() - 1;
}

int build_plan(ds_handle* h, int n, Plan* plan)
{
    plan->n = n;
    std::vector<Op> cnn, rnn;
    auto& LS = plan->launches;
    const bool first_plan = !h->stages_done;
    const bool bf = h->bf16;
    auto U = [&](int elems) { return bf ? elems / 2 : elems; };                // elements -> 4-byte units of the A operand
    auto eoff = [&](float* p, size_t elems) { return bf ? reinterpret_cast<float*>(reinterpret_cast<uint16_t*>(p) + elems) : p + elems; };
    const int CL = bf ? 256 : INC_OUT;                                         // channel pitch of module outputs
    auto add_gemm_op = [&](std::vector<Op>& list, int stream, int stage, GemmCfg cfg, const GemmLaunch& L, double kscale = 1.0) {
        Op op{};
        op.kind = OP_GEMM; op.stream = stream; op.stage = stage; op.cfg = cfg;
        op.launch_index = (int)LS.size(); op.total_tiles = L.total_tiles;
        const bool bf_cfg = cfg >= CFG_BCONV && cfg <= CFG_BFC_DENSE;
        const double kelems = (bf_cfg ? 2.0 : 1.0) * kscale;   // bf16 problems count K in units; kscale removes zero pad
        for (int i = 0; i < L.nprob; ++i) op.flops += 2.0 * L.prob[i].M * (double)L.prob[i].N * L.prob[i].K * kelems;
        LS.push_back(L);
        list.push_back(op);
        if (first_plan) {
            h->stages[stage].launches += 1;
            h->stages[stage].flops_per_site += op.flops / n;
        }
    };
    auto add_ew_op = [&](std::vector<Op>& list, Op op) {
        list.push_back(op);
        if (first_plan) h->stages[op.stage].launches += 1;
    };

    // ================= signal model (stream 0) — layers.py:181-239 =================
    int st = 0;
    const float* sig_rows = nullptr;      // module 11's output rows (what the folded head reads)
    if (h->is_cnn) {
    st = stage_id(h, "stem", 0);
    {
        Op op{};
        op.kind = OP_STEM1; op.stream = 0; op.stage = st; op.in = h->cur->d_signals; op.out = h->cur->stem_pool; op.a = bf;
        op.flops = 2.0 * h->w1 * 7 * 64 * n;
        add_ew_op(cnn, op);
        if (first_plan) h->stages[st].flops_per_site += 2.0 * h->w1 * 7 * 64;
        const int M = n * h->wa;
        if (!h->no_fused && h->wa <= 96 && (bf ? stem23_bf16_lds_bytes(h->wa, 1) : stem23_lds_bytes(h->wa, 1)) <= STEM23_MAX_LDS) {
            // conv_layer2 + conv_layer3 in one kernel (stem23_kernel): tiles of whole sites, conv2's rows never leave LDS
            Op o2{};
            o2.kind = OP_STEM23; o2.stream = 0; o2.stage = st; o2.a = bf;
            o2.sa.X = h->cur->stem_pool; o2.sa.Y = h->cur->conv3o; o2.sa.C2 = h->debug ? h->cur->conv2o : nullptr;
            o2.sa.Bp2 = h->conv2.Bp; o2.sa.bias2 = h->conv2.bias; o2.sa.Bp3 = h->conv3.Bp; o2.sa.bias3 = h->conv3.bias;
            o2.sa.n_sites = n; o2.sa.W = h->wa;
            // sites per tile: as full as 96 rows allow, as long as every CU still gets a tile
            int spt = std::max(1, 96 / h->wa);
            while (spt > 1 && (n + spt - 1) / spt < 256) --spt;
            // the T tile carries two halo rows per site: short sites (signal_len <= 128) at the full 96 rows pass the 80 KB the
            // kernel may ask for at two workgroups per CU (configure_fused_kernels)
            while (spt > 1 && (bf ? stem23_bf16_lds_bytes(h->wa, spt) : stem23_lds_bytes(h->wa, spt)) > STEM23_MAX_LDS) --spt;
            o2.sa.spt = spt;
            if (h->split && stem23_split_lds_bytes(h->wa, spt) <= 160 * 1024) {      // DS_PRECISION_BF16X3: split operands (ds_split.hip)
                o2.a = 2;
                o2.sa.Bp2 = h->conv2.Bps; o2.sa.Bp3 = h->conv3.Bps;
            }
            o2.flops = 2.0 * M * (64.0 * 128 + 384.0 * 256);
            add_ew_op(cnn, o2);
            if (first_plan) h->stages[st].flops_per_site += 2.0 * h->wa * (64.0 * 128 + 384.0 * 256);
        } else {
        GemmLaunch L{};
        const GemmCfg ccfg = bf ? CFG_BCONV : CFG_CONV;
        GemmProblem P = base_problem(M, 128, h->wa, h->conv2);                 // conv_layer2 1x1 (layers.py:192-197)
        add_seg(P, h->cur->stem_pool, U(64), 0, U(64));
        add_out(P, h->cur->conv2o, 128, 0, 128, 1, nullptr, 0, bf);
        add_tiles(L, P, ccfg, h->zero_seg);
        add_gemm_op(cnn, 0, st, ccfg, L);
        GemmLaunch L3{};
        GemmProblem P3 = base_problem(M, 256, h->wa, h->conv3);                // conv_layer3 1x3 (layers.py:198-203)
        for (int t = 0; t < 3; ++t) add_seg(P3, h->cur->conv2o, U(128), t - 1, U(128));
        add_out(P3, h->cur->conv3o, 256, 0, 256, 1, nullptr, 0, bf);
        add_tiles(L3, P3, ccfg, h->zero_seg);
        add_gemm_op(cnn, 0, st, ccfg, L3);
        }
    }
    const float* x = h->cur->conv3o;
    int cin = 256;
    int pend_pool_win = 0, pend_pool_pad = 0;      // a stride-2 maxpool waiting to be folded into the next fused module
    for (int m = 0; m < NMOD; ++m) {
        char nm[32];
        snprintf(nm, sizeof nm, "module%d", m + 1);
        st = stage_id(h, nm, 0);
        const int W = module_width(h, m), M = n * W;
        float* y = h->cur->modout[m];
        if (!h->no_fused && W <= 96) {
            // one fused launch per module; tile = spt whole sites (<= 96 rows). Pick the spt that minimises padded
            // rows (= matrix-pipe time) while keeping >= 128 workgroups when the batch allows it: with several
            // forwards in flight the CUs a short grid leaves idle are taken by other kernels, so fewer, fuller
            // tiles win (W = 23 at 512 sites: 128 tiles of 92/96 rows instead of 512 tiles of 23/32 rows, +2 %).
            int best_spt = 1; long best_rows = -1;
            for (int spt = 1; spt * W <= 96 && spt <= h->fuse_max_spt; ++spt) {
                const int tiles = (n + spt - 1) / spt;
                if (spt > 1 && tiles < std::min(h->fuse_min_tiles, n)) break;
                const long rows = (long)tiles * ((spt * W + 31) / 32) * 32;
                if (best_rows < 0 || rows < best_rows) { best_rows = rows; best_spt = spt; }
            }
            Op op{};
            op.kind = OP_FUSED; op.stream = 0; op.stage = st;
            op.tm = (best_spt * W + 31) / 32;
            op.fa.X = x; op.fa.Y = y; op.fa.n_sites = n; op.fa.W = W; op.fa.cin = bf ? 128 : cin; op.fa.spt = best_spt;   // bf16: row pitch in units
            op.fa.pool_win = pend_pool_win; op.fa.pool_pad = pend_pool_pad;
            pend_pool_win = 0;
            // DS_PRECISION_BF16X3: the split-operand kernel (ds_split.hip) on the three-term panels; a module whose input is
            // not a whole number of 16-channel chunks of 240 / 256 channels does not exist in this network
            const bool sp = h->split && (cin == 240 || cin == 256) && inception_fused_split_lds_bytes(op.tm, W, best_spt) <= 160 * 1024;
            op.d = sp ? 3 : 0;
            op.fa.Bp1 = sp ? h->m_f1[m].Bps : h->m_f1[m].Bp; op.fa.bias1 = h->m_f1[m].bias;
            op.fa.Bp3b = sp ? h->m_b3b[m].Bps : h->m_b3b[m].Bp; op.fa.bias3b = h->m_b3b[m].bias;
            op.fa.Bp4b = sp ? h->m_b4b[m].Bps : h->m_b4b[m].Bp; op.fa.bias4b = h->m_b4b[m].bias;
            op.fa.Bp5b = sp ? h->m_b5b[m].Bps : h->m_b5b[m].Bp; op.fa.bias5b = h->m_b5b[m].bias;
            op.fa.Bp5c = sp ? h->m_b5c[m].Bps : h->m_b5c[m].Bp; op.fa.bias5c = h->m_b5c[m].bias;
            op.fa.dbg = h->dbg_stamps ? h->dbg_stamps + (size_t)m * 1024 * 16 : nullptr;
            op.fa.write_rows = 1;
            op.flops = 2.0 * M * ((double)cin * 240 + 96 * 48 + 160 * 48 + 96 * 64 + 64 * 48);
            if (first_plan) h->stages[st].flops_per_site += op.flops / n;
            // a module joins the launch of the module before it when both tile the batch alike and it reads that
            // module's rows as they are (no stride-2 pool in between): every workgroup then takes its sites through the whole
            // chain (ds_internal.h FusedChain). Stage times of a chain are booked on its first module.
            Op* prev = (!cnn.empty() && cnn.back().kind == OP_FUSED) ? &cnn.back() : nullptr;
            if (prev && prev->fc.nmod < FUSED_CHAIN_MAX && op.fa.pool_win == 0 && prev->fa.W == W && prev->fa.spt == best_spt &&
                prev->tm == op.tm && prev->d == op.d && prev->fc.m[prev->fc.nmod - 1].Y == op.fa.X && op.fa.Y != prev->fc.m[0].X && !h->serial_modules) {
                // bf16: rows of a chain's inner modules never leave the CU (unless the taps of debug mode want them)
                if (bf && !h->debug) prev->fc.m[prev->fc.nmod - 1].write_rows = 0;
                prev->fc.m[prev->fc.nmod++] = op.fa;
                prev->flops += op.flops;
            } else {
                op.fc.m[0] = op.fa; op.fc.nmod = 1;
                add_ew_op(cnn, op);
            }
        } else {
        {   // five 1x1 convs on the module input + branch1 (maxpool on load)   layers.py:90-101,103,112,121-126
            // bf16 mode: rows have a 256-channel pitch (cin = 240 inputs carry 16 zero channels, matched by zero weight rows)
            const int xld = bf ? 256 : cin;
            const double ks = (double)cin / xld;
            const GemmCfg ccfg = bf ? CFG_BCONV : CFG_CONV, pcfg = bf ? CFG_BCONV_POOL : CFG_CONV_POOL;
            GemmLaunch L{};
            GemmProblem P = base_problem(M, 192, W, h->m_s1[m]);
            add_seg(P, x, U(xld), 0, U(xld));
            add_out(P, eoff(y, 48), CL, 0, 48, 1, nullptr, 0, bf);          // branch2
            add_out(P, h->cur->tmpS, 48, 48, 48, 0);                        // branch5 stem (BN, no ReLU), kept fp32
            add_out(P, h->cur->tmpA, 96, 96, 96, 1, nullptr, 0, bf);        // b3a | b4a | b5a
            add_tiles(L, P, ccfg, h->zero_seg);
            add_gemm_op(cnn, 0, st, ccfg, L, ks);
            GemmLaunch L1{};
            GemmProblem Q = base_problem(M, 48, W, h->m_b1[m]);
            Q.a_mode = 1;                                   // maxpool(3, s1) fused into the A load (layers.py:90-91)
            add_seg(Q, x, U(xld), 0, U(xld));
            add_out(Q, y, CL, 0, 48, 1, nullptr, 0, bf);    // branch1
            add_tiles(L1, Q, pcfg, h->zero_seg);
            add_gemm_op(cnn, 0, st, pcfg, L1, ks);
        }
        {   // second-stage convs from the 32-channel intermediates             layers.py:106-110,115-119,127-131
            const GemmCfg ccfg = bf ? CFG_BCONV : CFG_CONV;
            GemmLaunch L{};
            GemmProblem P = base_problem(M, 48, W, h->m_b3b[m]);
            for (int t = 0; t < 3; ++t) add_seg(P, eoff(h->cur->tmpA, 0), U(96), t - 1, U(32));
            add_out(P, eoff(y, 96), CL, 0, 48, 1, nullptr, 0, bf);
            add_tiles(L, P, ccfg, h->zero_seg);
            GemmProblem Q = base_problem(M, 48, W, h->m_b4b[m]);
            for (int t = 0; t < 5; ++t) add_seg(Q, eoff(h->cur->tmpA, 32), U(96), t - 2, U(32));
            add_out(Q, eoff(y, 144), CL, 0, 48, 1, nullptr, 0, bf);
            add_tiles(L, Q, ccfg, h->zero_seg);
            GemmProblem R = base_problem(M, 64, W, h->m_b5b[m]);
            for (int t = 0; t < 3; ++t) add_seg(R, eoff(h->cur->tmpA, 64), U(96), t - 1, U(32));
            add_out(R, h->cur->tmpB, 64, 0, 64, 1, nullptr, 0, bf);
            add_tiles(L, R, ccfg, h->zero_seg);
            add_gemm_op(cnn, 0, st, ccfg, L);
        }
        {   // residual tail: relu(stem + BN(1x1 48 of tmpB))                    layers.py:132-138
            const GemmCfg ccfg = bf ? CFG_BCONV : CFG_CONV;
            GemmLaunch L{};
            GemmProblem P = base_problem(M, 48, W, h->m_b5c[m]);
            add_seg(P, h->cur->tmpB, U(64), 0, U(64));
            add_out(P, eoff(y, 192), CL, 0, 48, 1, h->cur->tmpS, 48, bf);
            add_tiles(L, P, ccfg, h->zero_seg);
            add_gemm_op(cnn, 0, st, ccfg, L);
        }
        }
        x = y; cin = INC_OUT;
        if (m == 2 || m == 7) {   // maxpool_layer2/3                            layers.py:211-213,224-226
            const int wout = m == 2 ? h->wb : h->wc, pad = m == 2 ? h->pl_pool2 : h->pl_pool3;
            if (!h->no_fused && wout <= 96 && !h->debug_keep_pool) {
                pend_pool_win = W; pend_pool_pad = pad;      // folded into module m+2's staging: no launch, no buffer
            } else {
                Op op{};
                op.kind = OP_MAXPOOL; op.stream = 0; op.stage = stage_id(h, "pools", 0);
                op.in = y; op.out = m == 2 ? h->cur->pool2 : h->cur->pool3;
                op.a = W; op.b = wout; op.c = pad; op.d = CL;
                add_ew_op(cnn, op);
                x = op.out;
            }
        }
    }
    sig_rows = x;
    if (!h->fold_fc) {   // avgpool_layer1 + flatten (the folded head's matrix carries the pool: it reads module 11's rows)   layers.py:233-238
        Op op{};
        op.kind = OP_AVGPOOL; op.stream = 0; op.stage = stage_id(h, "pools", 0);
        op.in = x; op.out = bf ? h->cur->joint : h->cur->sigfeat; op.a = h->wc; op.d = INC_OUT;
        add_ew_op(cnn, op);
    }
    }   // is_cnn

    // ================= event model (stream 1) — layers.py:20-72,161-173 =================
    // Anti-diagonal wavefront: diagonal d runs cells (layer l, step d-l) of both directions in ONE
    // grouped launch; cell (l,s) depends only on (l-1,s) and (l,s-1), both on diagonal d-1.
    st = stage_id(h, "bilstm", 1);
    const int T = h->T;
    const bool lbf = h->lstm_bf16;
    const int HU = lbf ? HID / 2 : HID;                 // floats per site of one h vector (bf16 h: two units per float)
    const GemmCfg fc_cfg = bf ? (n % 128 == 0 ? CFG_BFC_DENSE : CFG_BFC) : (n % 128 == 0 ? CFG_FC_DENSE : CFG_FC);
    if (h->is_rnn) {
        // dedicated cell kernels, h / c in MFMA-fragment-major buffers (ds_internal.h LstmCell). fp32 cells: n-tiles per wave
        // 1 fills the GPU at <= 768 sites per forward (768 workgroups per full diagonal at 512), wider tiles re-read the
        // activation fragments less at bigger batches; every width gives the same bits (same K order per element).
        // bf16-operand cells (DS_PRECISION_BF16_ALL): lstm_cell_bf16_kernel, workgroup tile 64 x 64 .. 128 x 128 by batch.
        const int mtiles = (n + 31) / 32;
        const int nt = lbf ? (h->lstm_variant == DS_LSTM_TILING_NARROW ? 211 : h->lstm_variant == DS_LSTM_TILING_LDS1 ? 212
                              : h->lstm_variant == DS_LSTM_TILING_WIDE ? 222
                              : n >= 2048 ? 222 : n > 768 ? 212 : 211)
                       : h->lstm_variant == DS_LSTM_TILING_NARROW ? 1 : h->lstm_variant == DS_LSTM_TILING_WIDE ? 4
                       : h->lstm_variant == DS_LSTM_TILING_LDS1 ? 101 : h->lstm_variant == DS_LSTM_TILING_LDS2 ? 102
                       : (n <= 1024 ? 101 : 102);
        const bool lsp = h->split;                                         // split cells (ds_split.hip): tile code 311 | 312 | 322
        const int nt_split = h->lstm_variant == DS_LSTM_TILING_NARROW ? 311 : h->lstm_variant == DS_LSTM_TILING_LDS1 ? 312
                             : h->lstm_variant == DS_LSTM_TILING_WIDE ? 322 : h->lstm_variant == DS_LSTM_TILING_WIDE8 ? 328 : DS_SPLIT_LSTM_TILE(n);
        const size_t step = lsp ? (size_t)h->Bp32 * HID * 3 / 2 : (size_t)h->Bp32 * HU;      // floats of one time step in H (bf16 h: half; split h: 3/2)
        const bool xpj = h->lstm_xproj && lsp;
        const size_t xstep = (size_t)h->Bp32 * 4 * HID;
        if (xpj) {
            Op op{};
            op.kind = OP_XPROJ; op.stream = 1; op.stage = st;
            LstmXproj& X = op.xp;
            for (int dir = 0; dir < 2; ++dir) {
                LstmCell& C = X.cell[dir];
                C.bias = h->lstm_n[dir][0].bias; C.table = h->lstm_table[dir]; C.wfeat = h->lstm_wfeat[dir];
                C.codes = h->cur->d_kmer; C.means = h->cur->d_means; C.stds = h->cur->d_stds; C.lens = h->cur->d_sanums;
                C.c = h->cur->Cst[dir][0]; C.h_out = h->cur->H[dir][0]; C.use_feat = 1; C.c_zero = 1;
                X.xinit[dir] = h->cur->xproj[dir];
            }
            X.h_step = step; X.x_step = xstep; X.n = n; X.mtiles = mtiles; X.T = T; X.nsteps = h->lstm_xproj_all ? T : 1;
            rnn.push_back(op);
            if (first_plan) h->stages[st].launches += 1;
        }
        for (int d = xpj ? 1 : 0; d < T + NLAYER - 1; ++d) {      // (diagonal 0 = the two first-step cells of layer 0: done by lstm_xproj_kernel)
            LstmLaunch L;
            memset(&L, 0, sizeof L);
            L.n = n; L.mtiles = mtiles; L.T = T;
            L.dbg = (h->dbg_lstm && d < 32) ? h->dbg_lstm + (size_t)d * 1024 * 8 : nullptr;
            double flops = 0;
            for (int dir = 0; dir < 2; ++dir)
                for (int l = 0; l < NLAYER; ++l) {
                    const int sidx = d - l;
                    if (sidx < 0 || sidx >= T) continue;
                    const int t = dir == 0 ? sidx : T - 1 - sidx;
                    const int tprev = dir == 0 ? t - 1 : t + 1;
                    LstmCell& C = L.cell[L.ncell++];
                    C.ax = l > 0 ? h->cur->H[dir][l - 1] + (size_t)t * step : nullptr;
                    C.ah = sidx > 0 ? h->cur->H[dir][l] + (size_t)tprev * step : nullptr;
                    C.Bp = lsp ? h->lstm_n[dir][l].Bps : h->lstm_n[dir][l].Bp;
                    // fp32: k-groups of 8 per n-tile panel (K padded to 32); bf16 / split: k-steps of 16 (K padded to 64 elements)
                    C.kg_stride = lsp ? (h->lstm_n[dir][l].K + 63) / 64 * 64 / 16
                                  : lbf ? (2 * h->lstm_n[dir][l].K + 63) / 64 * 64 / 16 : (h->lstm_n[dir][l].K + 31) / 32 * 32 / 8;
                    C.bias = h->lstm_n[dir][l].bias;
                    C.table = l == 0 ? h->lstm_table[dir] : nullptr;
                    C.wfeat = h->lstm_wfeat[dir];
                    C.codes = h->cur->d_kmer; C.means = h->cur->d_means; C.stds = h->cur->d_stds; C.lens = h->cur->d_sanums;
                    C.xinit = (xpj && h->lstm_xproj_all && l == 0) ? h->cur->xproj[dir] + (size_t)t * xstep : nullptr;
                    C.c = h->cur->Cst[dir][l];
                    C.h_out = h->cur->H[dir][l] + (size_t)t * step;
                    // the joint FC reads the top layer's final h (fw: t = T-1, bw: t = 0) row-major   layers.py:171-172
                    C.h_row = (l == NLAYER - 1 && sidx == T - 1) ? h->cur->hlast[dir] : nullptr;
                    C.t = t; C.use_feat = l == 0; C.c_zero = sidx == 0;
                    flops += 2.0 * n * 4 * HID * ((l > 0 ? HID : 0) + (sidx > 0 ? HID : 0));
                }
            // heaviest cells first (K = 512, then 256, then 0): lstm_logical_tile deals tiles to the CUs in that order
            std::stable_sort(L.cell, L.cell + L.ncell, [](const LstmCell& a, const LstmCell& b) {
                return (a.ax != nullptr) + (a.ah != nullptr) > (b.ax != nullptr) + (b.ah != nullptr);
            });
            {   // workgroup tiles per work class, for lstm_logical_tile
                const int ntc = lsp ? nt_split - 100 : nt;      // the split tiles deal workgroups like the bf16 tiles of the same shape
                const int per_cell = (ntc == 222 || ntc == 228) ? ((mtiles + 3) / 4) * 8 : ntc == 212 ? ((mtiles + 1) / 2) * 8 : ntc == 211 ? ((mtiles + 1) / 2) * 16
                                     : nt > 100 ? ((mtiles + 1) / 2) * (16 / (nt - 100)) : ((mtiles + 3) / 4) * (32 / nt);
                L.cls_tiles[0] = L.cls_tiles[1] = 0;
                for (int i = 0; i < L.ncell; ++i) {
                    const int k = (L.cell[i].ax != nullptr) + (L.cell[i].ah != nullptr);
                    if (k == 2) L.cls_tiles[0] += per_cell; else if (k == 1) L.cls_tiles[1] += per_cell;
                }
            }
            Op op{};
            op.kind = OP_LSTM; op.stream = 1; op.stage = st;
            op.launch_index = (int)plan->lstm_launches.size();
            op.a = L.ncell; op.b = mtiles; op.c = lsp ? nt_split : nt;
            op.flops = flops;
            plan->lstm_launches.push_back(L);
            rnn.push_back(op);
            if (first_plan) {
                h->stages[st].launches += 1;
                h->stages[st].flops_per_site += flops / n;
            }
        }
    }
    if (bf && h->is_rnn) {     // the bf16 FC reads [bf16(h_fw(T-1)) | bf16(h_bw(0)) | signal features] from one buffer
        Op op{};
        op.kind = OP_PACKEV; op.stream = 1; op.stage = st;
        add_ew_op(rnn, op);
    }

    // ================= joint model (stream 0 after join) — layers.py:247-264 =================
    std::vector<Op> tail;
    if (h->fold_fc) {
        st = stage_id(h, "head", 0);
        Op op{};
        op.kind = OP_HEADF; op.stream = 0; op.stage = st;
        HeadFoldedArgs& a = op.ha;
        if (bf) {
            a.bf16 = 1;
            if (h->is_rnn) { a.seg[a.nseg] = h->cur->joint; a.len[a.nseg] = 2 * HID; a.pitch[a.nseg++] = h->JP; }     // [bf16 h_fw | h_bw] (pack_event_feat_bf16_kernel)
            if (h->is_cnn) { a.seg[a.nseg] = sig_rows; a.len[a.nseg] = h->wc * 256; a.pitch[a.nseg++] = h->wc * 256; }
        } else {
        if (h->is_rnn) {
            a.seg[a.nseg] = h->cur->hlast[0]; a.len[a.nseg++] = HID;
            a.seg[a.nseg] = h->cur->hlast[1]; a.len[a.nseg++] = HID;
        }
        if (h->is_cnn) { a.seg[a.nseg] = sig_rows; a.len[a.nseg++] = h->SF; }
        }
        a.w = h->w12f; a.logits = h->cur->logits; a.act = h->cur->act; a.pred = h->cur->pred; a.n = n; a.C = h->C;
        op.flops = 2.0 * h->J * h->C * n;
        add_ew_op(tail, op);
        if (first_plan) h->stages[st].flops_per_site += 2.0 * h->J * h->C;
    } else {
    st = stage_id(h, "fc1", 0);
    if (h->split && h->fc1.Bps && h->cur->jsplit && n >= h->split_dense_min_n) {
        // dense(J, J) with split operands (ds_split.hip): the joint row's segments -> term image -> LDS-DMA ring GEMM
        Op op{};
        op.kind = OP_DENSES; op.stream = 0; op.stage = st;
        SplitDense& d = op.sd;
        int ns = 0;
        if (h->is_rnn) { d.seg[ns] = h->cur->hlast[0]; d.len[ns++] = HID; d.seg[ns] = h->cur->hlast[1]; d.len[ns++] = HID; }
        if (h->is_cnn) { d.seg[ns] = h->cur->sigfeat; d.len[ns++] = h->SF; }
        for (; ns < 3; ++ns) { d.seg[ns] = h->cur->hlast[0] ? h->cur->hlast[0] : h->cur->sigfeat; d.len[ns] = 0; }
        d.A = h->cur->jsplit; d.Bp = reinterpret_cast<const char*>(h->fc1.Bps); d.C = h->cur->fc1o;
        d.n = n; d.N = h->J; d.mtiles = (n + 31) / 32; d.ntiles = (h->J + 31) / 32; d.ntiles_alloc = d.ntiles;
        d.ksteps = h->J / 16; d.kg_stride = (h->J + 63) / 64 * 64 / 16;
        // the 256 x 192 tile, K in as many ranges (<= DS_SPLIT_DENSE_PARTS) as it takes to put ~256 workgroups on the 256 CUs: 4 for engines
        // of up to 512 sites per forward, 2 up to 1,024, 1 from 2,048 (us per forward against the 128 x 96 / 128 x 128 tiles of mid-round with the same piped loop:
        // 168 / 177 at 512 sites, 326 / 405 at 1,024, 633 / 683 at 2,048, 1,262 / 1,376 at 4,096). DS_TUNE_SPLIT_DENSE_NARROW: the 128 x 96 tile
        d.wide = !h->split_dense_narrow && d.ntiles >= 6;
        d.splits = 1;
        // (the ranges follow the ENGINE's forward size, not this forward's: a site's bits then do not depend on how many sites share its
        // forward -- ragged tails run with few workgroups instead; always 4 ranges cost 8 - 12 % from 1,024 sites)
        if (d.wide) d.splits = std::max(1, std::min(DS_SPLIT_DENSE_PARTS, 256 / ((((h->B + 31) / 32 + 7) / 8) * ((d.ntiles + 5) / 6))));
        d.part_stride = (size_t)h->B * h->J;
        plan->fc1_parts = d.splits;
        op.flops = 2.0 * n * (double)h->J * h->J;
        add_ew_op(tail, op);
        if (first_plan) h->stages[st].flops_per_site += 2.0 * (double)h->J * h->J;
    } else {
        GemmLaunch L{};
        GemmProblem P = base_problem(n, h->J, n, h->fc1);
        // joint = [fw h(T-1) | bw h(0) | signal features]: three A segments, no concat buffer (layers.py:171-172,250-252)
        if (bf) {
            add_seg(P, h->cur->joint, h->JP / 2, 0, h->JP / 2);
        } else {
        if (h->is_rnn) {       // fp32 mode always runs the fp32 cells: their row-major copy of the two final h vectors
            add_seg(P, h->cur->hlast[0], HID, 0, HID);
            add_seg(P, h->cur->hlast[1], HID, 0, HID);
        }
        if (h->is_cnn) add_seg(P, h->cur->sigfeat, h->SF, 0, h->SF);
        }
        add_out(P, h->cur->fc1o, h->J, 0, h->J, 0);
        add_tiles(L, P, fc_cfg, h->zero_seg);
        add_gemm_op(tail, 0, st, fc_cfg, L, bf ? (double)h->J / h->JP : 1.0);
    }
    st = stage_id(h, "head", 0);
    {
        Op op{};
        op.kind = OP_HEAD; op.stream = 0; op.stage = st;
        op.flops = 2.0 * h->J * h->C * n;
        add_ew_op(tail, op);
        if (first_plan) h->stages[st].flops_per_site += 2.0 * h->J * h->C;
    }
    }

    // merged issue order: alternate the two independent branches, then the tail
    size_t i = 0, j = 0;
    while (i < cnn.size() || j < rnn.size()) {
        if (i < cnn.size()) plan->ops.push_back(cnn[i++]);
        if (j < rnn.size()) plan->ops.push_back(rnn[j++]);
    }
    for (auto& op : tail) plan->ops.push_back(op);

    void* p = nullptr;
    HIPCHK(h, hipMalloc(&p, LS.size() * sizeof(GemmLaunch)));
    h->allocs.push_back(p);
    plan->d_launches = static_cast<GemmLaunch*>(p);
    HIPCHK(h, hipMemcpy(p, LS.data(), LS.size() * sizeof(GemmLaunch), hipMemcpyHostToDevice));
    return DS_OK;
}

int issue_op(ds_handle* h, Plan& plan, const Op& op, hipStream_t s)
{
    const int n = plan.n;
    switch (op.kind) {
    case OP_GEMM:
        HIPCHK(h, launch_gemm(op.cfg, plan.d_launches + op.launch_index, op.total_tiles, s));
        break;
    case OP_STEM1:
        HIPCHK(h, launch_stem1(op.in, h->stem1_w, h->stem1_b, op.out, n, h->S, h->w1, h->pl_conv1, h->wa, h->pl_pool1, op.a, s));
        break;
    case OP_MAXPOOL:
        if (h->bf16) HIPCHK(h, launch_maxpool_s2_bf16(op.in, op.out, n, op.a, op.b, op.c, op.d, s));
        else HIPCHK(h, launch_maxpool_s2(op.in, op.out, n, op.a, op.b, op.c, op.d, s));
        break;
    case OP_AVGPOOL:
        if (h->bf16) HIPCHK(h, launch_avgpool7_bf16(op.in, op.out, n, op.a, op.d, 256, h->JP, h->is_rnn ? 2 * HID : 0, s));
        else HIPCHK(h, launch_avgpool7(op.in, op.out, n, op.a, op.d, s));
        break;
    case OP_PACKEV:
        HIPCHK(h, launch_pack_event_feat_bf16(h->cur->hlast[0], h->cur->hlast[1], h->cur->joint, n, h->JP, 0, s));
        break;
    case OP_LSTM:
        if (op.c >= 300) HIPCHK(h, launch_lstm_cells_split(op.c - 300, plan.lstm_launches[op.launch_index], s));
        else HIPCHK(h, launch_lstm_cells(op.c, plan.lstm_launches[op.launch_index], s));
        break;
    case OP_DENSES:
        HIPCHK(h, launch_dense_split(op.sd, s));
        break;
    case OP_XPROJ:
        HIPCHK(h, launch_lstm_xproj(op.xp, s));
        break;
    case OP_STEM23:
        if (op.a == 2) HIPCHK(h, launch_stem23_split(op.sa, s));
        else if (op.a) HIPCHK(h, launch_stem23_bf16(op.sa, s));
        else HIPCHK(h, launch_stem23(op.sa, s));
        break;
    case OP_HEADF:
        HIPCHK(h, launch_head_folded(op.ha, s));
        break;
    case OP_FUSED:
        if (h->bf16) HIPCHK(h, launch_inception_fused_bf16(op.tm, op.fc, s));
        else if (op.d == 3) HIPCHK(h, launch_inception_fused_split(op.tm, op.fc, s));
        else HIPCHK(h, launch_inception_fused(op.tm, op.fc, s));
        break;
    case OP_HEAD:
        HIPCHK(h, launch_head(h->cur->fc1o, h->fc2, h->cur->logits, h->cur->act, h->cur->pred, n, h->J, h->C, s, plan.fc1_parts, (size_t)h->B * h->J));
        break;
    }
    return DS_OK;
}

// Enqueue the whole forward on (s0, s1): fork after the inputs are in place, join before fc1.
// timed: bracket every launch with its own HIP event pair on the stream it is launched on.
int kernel_class(const Op& op);

int enqueue_forward(ds_handle* h, Plan& plan, int timed)
{
    HIPCHK(h, hipEventRecord(h->cur->ev_fork, h->cur->s0));
    HIPCHK(h, hipStreamWaitEvent(h->cur->s1, h->cur->ev_fork, 0));
    bool joined = false;
    const bool serial = h->serial || timed == 3;   // diagnostic: one stream, no overlap
    if (timed == 3) timed = 1;
    Op* head[2] = {nullptr, nullptr};       // open run per stream (timed == 1)
    auto close_run = [&](int si) -> int {
        if (head[si]) {
            HIPCHK(h, hipEventRecord(head[si]->ev1, si == 0 ? h->cur->s0 : h->cur->s1));
            head[si]->pending = true;
            head[si] = nullptr;
        }
        return DS_OK;
    };
    for (Op& op : plan.ops) {
        const int si = (op.stream == 0 || serial) ? 0 : 1;
        hipStream_t s = si == 0 ? h->cur->s0 : h->cur->s1;
        const bool is_tail = h->stages[op.stage].name == "fc1" || h->stages[op.stage].name == "head";
        if (is_tail && !joined) {
            int rc = close_run(0); if (rc) return rc;
            rc = close_run(1); if (rc) return rc;
            HIPCHK(h, hipEventRecord(h->cur->ev_join, h->cur->s1));
            HIPCHK(h, hipStreamWaitEvent(h->cur->s0, h->cur->ev_join, 0));
            joined = true;
        }
        if (timed && !op.ev0) {
            HIPCHK(h, hipEventCreate(&op.ev0));
            HIPCHK(h, hipEventCreate(&op.ev1));
        }
        if (timed == 2) {
            HIPCHK(h, hipEventRecord(op.ev0, s));
            op.run_launches = 1; op.run_flops = op.flops;
        } else if (timed == 1) {
            if (head[si] && kernel_class(*head[si]) != kernel_class(op)) { int rc = close_run(si); if (rc) return rc; }
            if (!head[si]) {
                head[si] = &op;
                op.run_launches = 0; op.run_flops = 0;
                HIPCHK(h, hipEventRecord(op.ev0, s));
            }
            head[si]->run_launches += 1;
            head[si]->run_flops += op.flops;
        }
        int rc = issue_op(h, plan, op, s);
        if (rc) return rc;
        if (timed == 2) {
            HIPCHK(h, hipEventRecord(op.ev1, s));
            op.pending = true;
        }
    }
    { int rc = close_run(0); if (rc) return rc; rc = close_run(1); if (rc) return rc; }
    if (!joined) {
        HIPCHK(h, hipEventRecord(h->cur->ev_join, h->cur->s1));
        HIPCHK(h, hipStreamWaitEvent(h->cur->s0, h->cur->ev_join, 0));
    }
    return DS_OK;
}

int kernel_class(const Op& op)
{
    switch (op.kind) {
    case OP_GEMM:
        return op.cfg == CFG_CONV ? K_GEMM_CONV : op.cfg == CFG_FC ? K_GEMM_FC
               : op.cfg == CFG_CONV_POOL ? K_GEMM_CONV_POOL : op.cfg == CFG_FC_DENSE ? K_GEMM_FC_DENSE
               : op.cfg == CFG_BCONV ? K_GEMM_BCONV : op.cfg == CFG_BCONV_POOL ? K_GEMM_BCONV_POOL : op.cfg == CFG_BFC ? K_GEMM_BFC
               : op.cfg == CFG_BFC_DENSE ? K_GEMM_BFC_DENSE : K_GEMM_CONV_WIDE;
    case OP_FUSED:
        if (op.fa.cin == 128) return op.tm == 1 ? K_FUSEDB1 : op.tm == 2 ? K_FUSEDB2 : K_FUSEDB3;   // bf16 rows: pitch in units
        if (op.d == 3) return op.tm == 1 ? K_FUSEDS1 : op.tm == 2 ? K_FUSEDS2 : K_FUSEDS3;          // split operands (three terms)
        return op.tm == 1 ? K_FUSED1 : op.tm == 2 ? K_FUSED2 : K_FUSED3;
    case OP_STEM1: return K_STEM1;
    case OP_STEM23: return op.a == 2 ? K_STEM23S : op.a ? K_STEM23B : K_STEM23;
    case OP_HEADF: return K_HEADF;
    case OP_MAXPOOL: return K_MAXPOOL;
    case OP_AVGPOOL: return K_AVGPOOL;
    case OP_HEAD: return K_HEAD;
    case OP_PACKEV: return K_PACKEV;
    case OP_DENSES: return K_DENSE_SPLIT;
    case OP_XPROJ: return K_LSTM_XPROJ;
    case OP_LSTM: if (op.c >= 300) return op.c == 311 ? K_LSTM_S11 : op.c == 312 ? K_LSTM_S12 : op.c == 328 ? K_LSTM_S28 : K_LSTM_S22;
        return op.c == 1 ? K_LSTM_CELL1 : op.c == 2 ? K_LSTM_CELL2 : op.c == 4 ? K_LSTM_CELL4 : op.c == 101 ? K_LSTM_LDS1
               : op.c == 211 ? K_LSTM_B11 : op.c == 212 ? K_LSTM_B12 : op.c == 222 ? K_LSTM_B22 : K_LSTM_LDS2;
    }
    return K_HEAD;
}

int collect_stage_times(ds_handle* h)
{
    for (Slot& sl : h->slots)
      for (auto& kv : sl.plans)
        for (Op& op : kv.second.ops) {
            if (!op.pending) continue;
            HIPCHK(h, hipEventSynchronize(op.ev1));
            float ms = 0;
            HIPCHK(h, hipEventElapsedTime(&ms, op.ev0, op.ev1));
            h->stages[op.stage].total_ms += ms;     // per-stage times are only meaningful in mode 2
            KernelStat& K = h->kstat[kernel_class(op)];
            K.launches += op.run_launches; K.total_ms += ms; K.flops += op.run_flops;
            op.pending = false;
        }
    return DS_OK;
}

void destroy_plan(ds_handle* h, Plan& p)
{
    if (p.graph) hipGraphExecDestroy(p.graph);
    for (Op& op : p.ops) { if (op.ev0) hipEventDestroy(op.ev0); if (op.ev1) hipEventDestroy(op.ev1); }
    if (p.d_launches) {
        hipFree(p.d_launches);
        auto it = std::find(h->allocs.begin(), h->allocs.end(), (void*)p.d_launches);
        if (it != h->allocs.end()) h->allocs.erase(it);
    }
    p.graph = nullptr; p.d_launches = nullptr; p.ops.clear();
}

constexpr size_t MAX_PLANS_PER_SLOT = 24;

int get_plan(ds_handle* h, int n, Plan** out)
{
    auto& plans = h->cur->plans;
    auto it = plans.find(n);
    if (it == plans.end()) {
        if (plans.size() >= MAX_PLANS_PER_SLOT && !h->profiling) {
            // a long run over ragged queue items sees up to max_batch distinct tail sizes: drop the least recently
            // used plan of this slot (its work must have drained before its graph / descriptors are freed)
            auto victim = plans.end();
            for (auto jt = plans.begin(); jt != plans.end(); ++jt)
                if (jt->first != h->B && (victim == plans.end() || jt->second.last_use < victim->second.last_use)) victim = jt;
            if (victim != plans.end()) {
                HIPCHK(h, hipStreamSynchronize(h->cur->s0));
                HIPCHK(h, hipStreamSynchronize(h->cur->s1));
                destroy_plan(h, victim->second);
                plans.erase(victim);
            }
        }
        Plan p;
        int rc = build_plan(h, n, &p);
        if (rc) return rc;
        it = plans.emplace(n, std::move(p)).first;
        h->stages_done = true;
    }
    it->second.uses += 1;
    it->second.last_use = ++h->plan_tick;
    *out = &it->second;
    return DS_OK;
}

int ensure_graph(ds_handle* h, Plan& plan)
{
    if (plan.graph) return DS_OK;
    hipGraph_t g = nullptr;
    HIPCHK(h, hipStreamBeginCapture(h->cur->s0, hipStreamCaptureModeThreadLocal));
    int rc = enqueue_forward(h, plan, 0);
    hipError_t e = hipStreamEndCapture(h->cur->s0, &g);
    if (rc) { if (g) hipGraphDestroy(g); return rc; }
    if (e != hipSuccess) return fail(h, DS_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
    e = hipGraphInstantiate(&plan.graph, g, nullptr, nullptr, 0);
    hipGraphDestroy(g);
    if (e != hipSuccess) return fail(h, DS_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e));
    return DS_OK;
}

// inputs must already be in the handle's device input buffers
int run_resident(ds_handle* h, int n)
{
    Plan* plan = nullptr;
    int rc = get_plan(h, n, &plan);
    if (rc) return rc;
    h->cur->last_n = n;
    h->cur->last_fc1_parts = plan->fc1_parts;
    if (h->profiling) {
        rc = collect_stage_times(h);      // events of a previous profiled forward are reused below
        if (rc) return rc;
        rc = enqueue_forward(h, *plan, h->profiling);
        if (rc) return rc;
        for (Stage& S : h->stages) S.calls += 1;
        return DS_OK;
    }
    // a graph is worth its capture (~ms) only for sizes that come back: the full batch, or any size seen twice
    if (h->use_graph && (plan->graph || n == h->B || plan->uses >= 2)) {
        rc = ensure_graph(h, *plan);
        if (rc) return rc;
        HIPCHK(h, hipGraphLaunch(plan->graph, h->cur->s0));
        return DS_OK;
    }
    return enqueue_forward(h, *plan, 0);
}

// Every slot's plan and captured graph for the full batch, built when the weights are finalized: the first max_batch-sized
// forward of a slot then costs what every later one costs (a caller that times its first `slots` calls -- bench.py with
// --warmup smaller than the slot count -- would otherwise see plan building and graph capture inside its window).
// An optimisation, not a requirement: the weights are finalized by then, so a plan or a capture that fails here (device memory
// for the launch descriptors, a capture error) leaves the handle usable -- the slot builds its plan lazily at its first
// forward, as every other batch size does, and THAT call reports the error if it persists. Serial / profiling handles
// (DS_TUNE_SERIAL: stand-alone kernel timing under rocprofv3) skip it: they run eagerly and a capture would only add
// activity to the trace.
void prepare_slots(ds_handle* h)
{
    if (!h->use_graph || h->debug || h->serial) return;
    Slot* keep = h->cur;
    const std::string err_before = h->err;
    for (Slot& sl : h->slots) {
        h->cur = &sl;
        Plan* plan = nullptr;
        int rc = get_plan(h, h->B, &plan);
        if (!rc) { plan->uses = 0; rc = ensure_graph(h, *plan); }
        if (rc) {
            fprintf(stderr, "deepsignal_amd: preparing a slot's full-batch plan failed (%s); plans will be built at the first forward\n",
                    h->err.c_str());
            h->err = err_before;
            break;
        }
    }
    h->cur = keep;
}

}  // namespace

// ---- exception firewall: nothing may unwind across the C ABI (std::bad_alloc from an oversized tensor, length_error
// from a corrupt header, ...): every allocating entry point runs behind this guard and reports a DS_ERR_* code instead.
namespace {
template <class F>
int guarded(ds_handle* h, F&& body)
{
    try {
        return body();
    } catch (const std::bad_alloc&) {
        return fail(h, DS_ERR_NOMEM, "out of host memory");
    } catch (const std::exception& e) {
        return fail(h, DS_ERR_INVALID, std::string("internal error: ") + e.what());
    } catch (...) {
        return fail(h, DS_ERR_INVALID, "internal error");
    }
}
}  // namespace

// ======================================= C ABI =======================================
extern "C" {

const char* ds_version(void)
{
    return "deepsignal_amd 0.4 (gfx950; fp32 MFMA, bf16 conv + FC and bf16_all operand modes; bf16x3 = fp32 operands as three bf16 "
           "terms, six products per MAC, in: conv_layer2 / 3, the eleven inception modules, the BiLSTM cells' recurrent and lower-layer products, dense(J, J) of the three-step "
           "joint model)";
}

const char* ds_last_error(const ds_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

static int ds_create_impl(const ds_config* cfg, ds_handle** out)
{
    if (!cfg || !out) return fail(nullptr, DS_ERR_INVALID, "ds_create: null argument");
    *out = nullptr;
    if (!(cfg->is_cnn || cfg->is_rnn))
        return fail(nullptr, DS_ERR_INVALID, "at least one of is_cnn/is_rnn should be True");      // model.py:28-29
    if (cfg->precision != DS_PRECISION_FP32 && cfg->precision != DS_PRECISION_BF16 && cfg->precision != DS_PRECISION_BF16_ALL &&
        cfg->precision != DS_PRECISION_BF16X3)
        return fail(nullptr, DS_ERR_UNSUPPORTED, "precision must be DS_PRECISION_FP32, DS_PRECISION_BF16, DS_PRECISION_BF16_ALL or DS_PRECISION_BF16X3");
    if (cfg->precision == DS_PRECISION_BF16X3 && (cfg->reserved[2] & DS_TUNE_NO_FUSED))
        return fail(nullptr, DS_ERR_UNSUPPORTED, "DS_PRECISION_BF16X3 runs the fused inception kernels only (DS_TUNE_NO_FUSED asks for the layer-granular fp32 path)");
    if (cfg->kmer_len < 1 || cfg->kmer_len > 255 || (cfg->kmer_len & 1) == 0)
        return fail(nullptr, DS_ERR_INVALID, "kmer_len must be odd and in [1,255]");
    if (cfg->signal_len < 16 || cfg->class_num < 1 || cfg->class_num > 16)
        return fail(nullptr, DS_ERR_INVALID, "bad signal_len/class_num");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, DS_ERR_HIP, std::string("no HIP device available: ") + hipGetErrorString(e));
    if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, DS_ERR_INVALID, "device ordinal out of range");
    ds_handle* h = new ds_handle();
    h->cfg = *cfg;
    h->slots.resize(1);
    h->cur = &h->slots[0];
    h->T = cfg->kmer_len; h->S = cfg->signal_len; h->C = cfg->class_num;
    h->B = cfg->max_batch > 0 ? cfg->max_batch : 512;
    same_pad(h->S, 7, 2, &h->w1, &h->pl_conv1);
    same_pad(h->w1, 3, 2, &h->wa, &h->pl_pool1);
    same_pad(h->wa, 3, 2, &h->wb, &h->pl_pool2);
    same_pad(h->wb, 3, 2, &h->wc, &h->pl_pool3);
    h->is_cnn = cfg->is_cnn != 0; h->is_rnn = cfg->is_rnn != 0; h->is_base = cfg->is_base != 0;
    h->SF = h->wc * INC_OUT;
    h->J = (h->is_rnn ? 2 * HID : 0) + (h->is_cnn ? h->SF : 0);     // layers.py:248-255
    h->bf16 = cfg->precision == DS_PRECISION_BF16 || cfg->precision == DS_PRECISION_BF16_ALL;
    h->lstm_bf16 = cfg->precision == DS_PRECISION_BF16_ALL && h->is_rnn;
    h->split = cfg->precision == DS_PRECISION_BF16X3;
    // tuning / diagnostic knobs live in the handle's own config (no process-global state): reserved[2] = flags,
    // reserved[3] = LSTM tiling override, reserved[4] / [5] = fused-module tile bounds
    const int32_t flags = cfg->reserved[2];
    h->no_fused = (flags & DS_TUNE_NO_FUSED) != 0;
    h->serial = (flags & DS_TUNE_SERIAL) != 0;
    h->serial_modules = (flags & DS_TUNE_NO_CHAIN) != 0;
    h->shared_s1 = (flags & DS_TUNE_SHARED_EVENT_STREAM) != 0 && !h->serial;
    if (h->shared_s1) h->use_graph = false;      // the launches must stay on the shared stream (a graph node has no stream)
    h->fold_fc = !(flags & DS_TUNE_NO_FOLD_FC) && cfg->reserved[0] == 0 && cfg->class_num <= 16;
    h->lstm_variant = cfg->reserved[3];
    if (cfg->reserved[4] > 0) h->fuse_max_spt = cfg->reserved[4];
    if (cfg->reserved[5] > 0) h->fuse_min_tiles = cfg->reserved[5];
    if (cfg->reserved[6] > 0) h->split_dense_min_n = cfg->reserved[6];
    h->split_dense_narrow = (flags & DS_TUNE_SPLIT_DENSE_NARROW) != 0;
    h->lstm_xproj = h->split && h->is_rnn && !(flags & DS_TUNE_NO_LSTM_XPROJ);
    h->lstm_xproj_all = h->lstm_xproj && (flags & DS_TUNE_LSTM_XPROJ_ALL);
    h->lstm_frag = h->is_rnn && !h->lstm_bf16;
    h->Bp32 = (h->B + 31) / 32 * 32;
    h->JP = (h->J + 31) / 32 * 32;
    h->debug = cfg->reserved[0] != 0;
#define CK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { fail(nullptr, DS_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); ds_destroy(h); return DS_ERR_HIP; } } while (0)
    CK(hipSetDevice(cfg->device));
    CK(configure_fused_kernels());
    CK(configure_split_kernels());
    // reserved[1] = forwards in flight (pipeline slots); 0 -> default
    // (the bf16 modes: a 512-site forward is ~0.2 ms, four in flight measured 2.60 M sites/s against 2.51 M with eight)
    int nslots = cfg->reserved[1] > 0 ? cfg->reserved[1] : (h->B <= 1024 && !h->bf16 ? 8 : 4);
    nslots = std::max(1, std::min(nslots, 16));
    h->slots.resize(nslots);
    int rc = DS_OK;
    for (Slot& sl : h->slots) {
        h->cur = &sl;
        CK(hipStreamCreateWithFlags(&sl.s0, hipStreamNonBlocking));
        if (h->shared_s1 && &sl != &h->slots[0]) { sl.s1 = h->slots[0].s1; sl.owns_s1 = false; }
        else CK(hipStreamCreateWithFlags(&sl.s1, hipStreamNonBlocking));
        CK(hipEventCreateWithFlags(&sl.ev_fork, hipEventDisableTiming));
        CK(hipEventCreateWithFlags(&sl.ev_join, hipEventDisableTiming));
        if (!rc) rc = alloc_workspace(h);
    }
    h->cur = &h->slots[0];
#undef CK
    if (!rc) {
        float* z = nullptr;
        rc = dalloc(h, &z, 64);
        if (!rc && hipMemset(z, 0, 256) != hipSuccess) rc = fail(h, DS_ERR_HIP, "hipMemset");
        h->zero_seg = z;
        if (!rc && (flags & DS_TUNE_DEBUG_STAMPS)) {
            rc = dalloc(h, &h->dbg_stamps, (size_t)NMOD * 1024 * 16);
            if (!rc) hipMemset(h->dbg_stamps, 0, (size_t)NMOD * 1024 * 16 * 8);
            if (!rc) rc = dalloc(h, &h->dbg_lstm, (size_t)32 * 1024 * 8);
            if (!rc) hipMemset(h->dbg_lstm, 0, (size_t)32 * 1024 * 8 * 8);
        }
    }
    if (rc) { g_create_error = h->err; ds_destroy(h); return rc; }
    *out = h;
    return DS_OK;
}

void ds_destroy(ds_handle* h)
{
    if (!h) return;
    hipSetDevice(h->cfg.device);
    // every stream is drained before any is destroyed: with DS_TUNE_SHARED_EVENT_STREAM the slots share slot 0's s1
    for (Slot& sl : h->slots) {
        if (sl.s0) hipStreamSynchronize(sl.s0);
        if (sl.s1) hipStreamSynchronize(sl.s1);
    }
    for (Slot& sl : h->slots) {
        for (auto& kv : sl.plans) {
            if (kv.second.graph) hipGraphExecDestroy(kv.second.graph);
            for (Op& op : kv.second.ops) { if (op.ev0) hipEventDestroy(op.ev0); if (op.ev1) hipEventDestroy(op.ev1); }
        }
        if (sl.pin_in) hipHostFree(sl.pin_in);
        if (sl.pin_act) hipHostFree(sl.pin_act);       // pin_pred points into it
        if (sl.ev_fork) hipEventDestroy(sl.ev_fork);
        if (sl.ev_join) hipEventDestroy(sl.ev_join);
        if (sl.s0) hipStreamDestroy(sl.s0);
        if (sl.s1 && sl.owns_s1) hipStreamDestroy(sl.s1);
    }
    for (void* p : h->allocs) hipFree(p);
    (void)hipGetLastError();      // nothing a teardown call returned may surface in a later handle's first launch
    delete h;
}

static int ds_set_tensor_impl(ds_handle* h, const char* name, const float* data, const int64_t* shape, int32_t ndim)
{
    if (!h || !name || !data || !shape || ndim < 1 || ndim > 8) return fail(h, DS_ERR_INVALID, "ds_set_tensor: bad argument");
    if (h->finalized) return fail(h, DS_ERR_INVALID, "weights already finalized");
    HostTensor t;
    int64_t cnt = 1;
    for (int i = 0; i < ndim; ++i) { t.shape.push_back(shape[i]); cnt *= shape[i]; }
    if (cnt <= 0) return fail(h, DS_ERR_INVALID, "ds_set_tensor: empty tensor");
    t.data.assign(data, data + cnt);
    h->host[name] = std::move(t);
    return DS_OK;
}

static int ds_finalize_weights_impl(ds_handle* h)
{
    if (!h) return DS_ERR_INVALID;
    if (h->finalized) return fail(h, DS_ERR_INVALID, "weights already finalized");
    hipSetDevice(h->cfg.device);
    int rc = finalize_weights(h);
    if (!rc) prepare_slots(h);
    return rc;
}

static int ds_load_weights_impl(ds_handle* h, const char* path)
{
    if (!h || !path) return fail(h, DS_ERR_INVALID, "ds_load_weights: bad argument");
    FILE* f = fopen(path, "rb");
    if (!f) return fail(h, DS_ERR_IO, std::string("cannot open ") + path);
    auto bad = [&](const char* why) { fclose(f); return fail(h, DS_ERR_IO, std::string(path) + ": " + why); };
    char magic[8];
    uint32_t nt = 0;
    if (fread(magic, 1, 8, f) != 8 || memcmp(magic, "DSAMDW01", 8) != 0) return bad("not a DSAMDW01 file");
    if (fread(&nt, 4, 1, f) != 1 || nt > 100000) return bad("bad tensor count");
    struct Meta { std::string name; std::vector<int64_t> shape; uint64_t off, nbytes; };
    std::vector<Meta> metas(nt);
    if (fseek(f, 0, SEEK_END) != 0) return bad("cannot seek");
    const long fsize_l = ftell(f);
    if (fsize_l < 12 || fseek(f, 12, SEEK_SET) != 0) return bad("cannot seek");
    const uint64_t fsize = (uint64_t)fsize_l;
    for (auto& m : metas) {
        uint16_t ln = 0; uint8_t nd = 0;
        if (fread(&ln, 2, 1, f) != 1) return bad("truncated header");
        m.name.resize(ln);
        if (ln && fread(&m.name[0], 1, ln, f) != ln) return bad("truncated header");
        if (fread(&nd, 1, 1, f) != 1 || nd > 8) return bad("truncated header");
        for (int i = 0; i < nd; ++i) { uint32_t d = 0; if (fread(&d, 4, 1, f) != 1) return bad("truncated header"); m.shape.push_back(d); }
        if (fread(&m.off, 8, 1, f) != 1 || fread(&m.nbytes, 8, 1, f) != 1) return bad("truncated header");
        // the header is untrusted: the payload must be exactly prod(shape) floats and lie inside the file
        const uint64_t max_elems = (uint64_t)1 << 32;
        uint64_t cnt = 1;
        for (int64_t d : m.shape) {       // divide before multiplying: the running product never wraps
            if (d <= 0 || (uint64_t)d > max_elems / cnt) return bad("bad tensor shape");
            cnt *= (uint64_t)d;
        }
        if (m.nbytes != cnt * 4) return bad("tensor byte count does not match its shape");
        if (m.off > fsize || m.nbytes > fsize - m.off) return bad("tensor payload lies outside the file");
    }
    for (auto& m : metas) {
        HostTensor t;
        t.shape = m.shape;
        t.data.resize(m.nbytes / 4);
        if (fseek(f, (long)m.off, SEEK_SET) != 0 || fread(t.data.data(), 1, m.nbytes, f) != m.nbytes) return bad("truncated payload");
        h->host[m.name] = std::move(t);
    }
    fclose(f);
    return ds_finalize_weights(h);
}

static int ds_forward_device_impl(ds_handle* h, int32_t n, const int32_t* d_kmer, const float* d_means, const float* d_stds,
                      const float* d_sanums, const float* d_signals, float* d_act, int32_t* d_pred)
{
    if (!h) return DS_ERR_INVALID;
    if (!h->finalized) return fail(h, DS_ERR_INVALID, "weights not loaded");
    if (n < 0 || n > h->B) return fail(h, DS_ERR_INVALID, "n exceeds max_batch");
    if (n == 0) return DS_OK;
    if (!d_kmer || !d_means || !d_stds || !d_sanums || !d_signals || !d_act || !d_pred)
        return fail(h, DS_ERR_INVALID, "null buffer");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    // next pipeline slot (profiling runs stay on slot 0 so the event statistics are coherent)
    h->cur = &h->slots[h->profiling ? 0 : (h->next_slot++ % h->slots.size())];
    // one gather launch in, one scatter launch out (seven copy dispatches per forward before)
    HIPCHK(h, launch_gather_inputs(d_kmer, d_means, d_stds, d_sanums, d_signals, h->cur->d_in, n, h->T, h->S, h->B, h->cur->s0));
    int rc = run_resident(h, n);
    if (rc) return rc;
    HIPCHK(h, launch_scatter_outputs(h->cur->act, h->cur->pred, d_act, d_pred, n, h->C, h->cur->s0));
    return DS_OK;
}

int ds_sync(ds_handle* h)
{
    if (!h) return DS_ERR_INVALID;
    for (Slot& sl : h->slots) {
        HIPCHK(h, hipStreamSynchronize(sl.s0));
        HIPCHK(h, hipStreamSynchronize(sl.s1));
    }
    if (h->profiling) return collect_stage_times(h);
    return DS_OK;
}

static int ds_forward_impl(ds_handle* h, int32_t n, const int32_t* kmer, const float* means, const float* stds, const float* sanums,
               const float* signals, float* act, int32_t* pred)
{
    if (!h) return DS_ERR_INVALID;
    if (!h->finalized) return fail(h, DS_ERR_INVALID, "weights not loaded");
    if (n < 0) return fail(h, DS_ERR_INVALID, "negative n");
    if (n == 0) return DS_OK;
    if (!kmer || !means || !stds || !sanums || !signals || !act || !pred) return fail(h, DS_ERR_INVALID, "null buffer");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    if (!h->profiling) {
        // every pass goes through the asynchronous boundary: the batch is staged in the slot's pinned block (ONE H2D copy of
        // a full batch, one D2H copy of [act | pred]; five pageable copies in and two out before), and a call of more than
        // max_batch sites keeps up to `slots` passes in flight, results copied out in order
        for (Slot& sl : h->slots)
            if (sl.submitted_n >= 0) return fail(h, DS_ERR_INVALID, "ds_forward: ds_submit tickets are still in flight");
        const int nslots = (int)h->slots.size();
        std::vector<int32_t> tickets;
        size_t tail = 0;
        int off_wait = 0, rc = DS_OK;
        for (int off = 0; off < n && !rc; off += h->B) {
            const int m = std::min(h->B, n - off);
            if ((int)(tickets.size() - tail) == nslots) {
                const int mw = std::min(h->B, n - off_wait);
                rc = ds_wait(h, tickets[tail++], act + (size_t)off_wait * h->C, pred + off_wait);
                if (rc) break;
                off_wait += mw;
            }
            int32_t t = -1;
            rc = ds_submit(h, m, kmer + (size_t)off * h->T, means + (size_t)off * h->T, stds + (size_t)off * h->T,
                           sanums + (size_t)off * h->T, signals + (size_t)off * h->S, &t);
            if (!rc) tickets.push_back(t);
        }
        while (tail < tickets.size() && !rc) {
            const int mw = std::min(h->B, n - off_wait);
            rc = ds_wait(h, tickets[tail++], act + (size_t)off_wait * h->C, pred + off_wait);
            off_wait += mw;
        }
        if (rc) {       // leave no pass in flight behind a failed call (the error message of the failing step is kept)
            const std::string msg = h->err;
            for (Slot& sl : h->slots) {
                if (sl.s0) hipStreamSynchronize(sl.s0);
                if (sl.s1) hipStreamSynchronize(sl.s1);
                sl.submitted_n = -1;
            }
            h->err = msg;
        }
        return rc;
    }
    // profiling runs stay on slot 0 with plain copies, so that the event statistics are coherent
    for (int off = 0; off < n; off += h->B) {
        const int m = std::min(h->B, n - off);
        h->cur = &h->slots[0];
        const size_t mt = (size_t)m * h->T, ot = (size_t)off * h->T;
        HIPCHK(h, hipMemcpyAsync(h->cur->d_kmer, kmer + ot, mt * 4, hipMemcpyHostToDevice, h->cur->s0));
        HIPCHK(h, hipMemcpyAsync(h->cur->d_means, means + ot, mt * 4, hipMemcpyHostToDevice, h->cur->s0));
        HIPCHK(h, hipMemcpyAsync(h->cur->d_stds, stds + ot, mt * 4, hipMemcpyHostToDevice, h->cur->s0));
        HIPCHK(h, hipMemcpyAsync(h->cur->d_sanums, sanums + ot, mt * 4, hipMemcpyHostToDevice, h->cur->s0));
        HIPCHK(h, hipMemcpyAsync(h->cur->d_signals, signals + (size_t)off * h->S, (size_t)m * h->S * 4, hipMemcpyHostToDevice, h->cur->s0));
        int rc = run_resident(h, m);
        if (rc) return rc;
        HIPCHK(h, hipMemcpyAsync(act + (size_t)off * h->C, h->cur->act, (size_t)m * h->C * 4, hipMemcpyDeviceToHost, h->cur->s0));
        HIPCHK(h, hipMemcpyAsync(pred + off, h->cur->pred, (size_t)m * 4, hipMemcpyDeviceToHost, h->cur->s0));
        rc = ds_sync(h);
        if (rc) return rc;
    }
    return DS_OK;
}

// Asynchronous host-buffer boundary: ds_submit copies one batch into the next slot's pinned staging buffer and
// enqueues H2D + forward + D2H on that slot's streams; ds_wait(ticket) blocks until that forward is done and hands
// the results out. Up to `slots` forwards are in flight, so PCIe copies, the 19-launch LSTM chain of one batch and
// the host's own work (parsing, formatting) overlap.
static int ds_submit_parts_impl(ds_handle* h, int32_t nparts, const int32_t* counts, const int32_t* const* kmer, const float* const* means,
                                const float* const* stds, const float* const* sanums, const float* const* signals, int32_t* ticket)
{
    if (!h || !ticket) return DS_ERR_INVALID;
    if (!h->finalized) return fail(h, DS_ERR_INVALID, "weights not loaded");
    if (nparts <= 0 || !counts || !kmer || !means || !stds || !sanums || !signals) return fail(h, DS_ERR_INVALID, "null buffer");
    int64_t n64 = 0;
    for (int i = 0; i < nparts; ++i) {
        if (counts[i] < 0) return fail(h, DS_ERR_INVALID, "ds_submit: negative segment length");
        if (counts[i] > 0 && (!kmer[i] || !means[i] || !stds[i] || !sanums[i] || !signals[i])) return fail(h, DS_ERR_INVALID, "null buffer");
        n64 += counts[i];
    }
    if (n64 <= 0 || n64 > h->B) return fail(h, DS_ERR_INVALID, "ds_submit: n must be in [1, max_batch]");
    const int n = (int)n64;
    if (h->profiling) return fail(h, DS_ERR_INVALID, "ds_submit is not available while profiling is on");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const int si = (int)(h->next_slot % h->slots.size());
    Slot& sl = h->slots[si];
    if (sl.submitted_n >= 0) return fail(h, DS_ERR_INVALID, "ds_submit: every slot is in flight; ds_wait the oldest ticket first");
    const size_t B = h->B, T = h->T, S = h->S;
    const size_t in_bytes = B * (4 * T * 4 + S * 4);
    if (!sl.pin_in) {
        HIPCHK(h, hipHostMalloc((void**)&sl.pin_in, in_bytes, hipHostMallocDefault));
        HIPCHK(h, hipHostMalloc((void**)&sl.pin_act, (B * h->C + B) * 4, hipHostMallocDefault));      // [act | pred]
        sl.pin_pred = reinterpret_cast<int*>(sl.pin_act + B * h->C);
    }
    h->next_slot++;
    h->cur = &sl;
    const size_t nt = (size_t)n * T * 4;
    char* p = sl.pin_in;
    size_t row = 0;
    for (int i = 0; i < nparts; ++i) {
        const size_t c = (size_t)counts[i];
        if (!c) continue;
        memcpy(p + row * T * 4, kmer[i], c * T * 4);
        memcpy(p + B * T * 4 + row * T * 4, means[i], c * T * 4);
        memcpy(p + 2 * B * T * 4 + row * T * 4, stds[i], c * T * 4);
        memcpy(p + 3 * B * T * 4 + row * T * 4, sanums[i], c * T * 4);
        memcpy(p + 4 * B * T * 4 + row * S * 4, signals[i], c * S * 4);
        row += c;
    }
    if ((size_t)n == B) {        // the staging buffer is the image of the device block: one copy
        HIPCHK(h, hipMemcpyAsync(sl.d_in, p, in_bytes, hipMemcpyHostToDevice, sl.s0));
    } else {
        HIPCHK(h, hipMemcpyAsync(sl.d_kmer, p, nt, hipMemcpyHostToDevice, sl.s0));
        HIPCHK(h, hipMemcpyAsync(sl.d_means, p + B * T * 4, nt, hipMemcpyHostToDevice, sl.s0));
        HIPCHK(h, hipMemcpyAsync(sl.d_stds, p + 2 * B * T * 4, nt, hipMemcpyHostToDevice, sl.s0));
        HIPCHK(h, hipMemcpyAsync(sl.d_sanums, p + 3 * B * T * 4, nt, hipMemcpyHostToDevice, sl.s0));
        HIPCHK(h, hipMemcpyAsync(sl.d_signals, p + 4 * B * T * 4, (size_t)n * S * 4, hipMemcpyHostToDevice, sl.s0));
    }
    int rc = run_resident(h, n);
    if (rc) return rc;
    // [act (max_batch rows) | pred]: one copy back
    HIPCHK(h, hipMemcpyAsync(sl.pin_act, sl.act, (B * h->C + (size_t)n) * 4, hipMemcpyDeviceToHost, sl.s0));
    sl.submitted_n = n;
    *ticket = si;
    return DS_OK;
}

static int ds_submit_impl(ds_handle* h, int32_t n, const int32_t* kmer, const float* means, const float* stds, const float* sanums,
              const float* signals, int32_t* ticket)
{
    if (h && (!kmer || !means || !stds || !sanums || !signals)) return fail(h, DS_ERR_INVALID, "null buffer");
    return ds_submit_parts_impl(h, 1, &n, &kmer, &means, &stds, &sanums, &signals, ticket);
}

static int ds_wait_impl(ds_handle* h, int32_t ticket, float* act, int32_t* pred)
{
    if (!h || !act || !pred) return DS_ERR_INVALID;
    if (ticket < 0 || ticket >= (int)h->slots.size() || h->slots[ticket].submitted_n < 0)
        return fail(h, DS_ERR_INVALID, "ds_wait: no forward in flight for this ticket");
    Slot& sl = h->slots[ticket];
    HIPCHK(h, hipStreamSynchronize(sl.s0));
    memcpy(act, sl.pin_act, (size_t)sl.submitted_n * h->C * 4);
    memcpy(pred, sl.pin_pred, (size_t)sl.submitted_n * 4);
    sl.submitted_n = -1;
    return DS_OK;
}

int ds_num_slots(ds_handle* h) { return h ? (int)h->slots.size() : DS_ERR_INVALID; }

int ds_alloc_host(size_t bytes, void** out)
{
    if (!out) return DS_ERR_INVALID;
    return hipHostMalloc(out, bytes, hipHostMallocDefault) == hipSuccess ? DS_OK : DS_ERR_NOMEM;
}

int ds_free_host(void* p) { return hipHostFree(p) == hipSuccess ? DS_OK : DS_ERR_HIP; }

int64_t ds_get_intermediate(ds_handle* h, const char* name, float* out, int64_t capacity)
{
    if (!h || !name || !out) return DS_ERR_INVALID;
    const int n = h->cur->last_n;
    if (n <= 0) return fail(h, DS_ERR_INVALID, "no forward has run");
    int rc = ds_sync(h);
    if (rc) return rc;
    const std::string s(name);
    auto copy = [&](const float* src, int64_t count) -> int64_t {
        if (count > capacity) return fail(h, DS_ERR_INVALID, "capacity too small for " + s);
        if (hipMemcpy(out, src, (size_t)count * 4, hipMemcpyDeviceToHost) != hipSuccess) return fail(h, DS_ERR_HIP, "hipMemcpy D2H");
        return count;
    };
    // bf16 mode: activation taps are stored as bf16 rows of pitch ld; widen the valid channels to fp32
    auto copy_bf = [&](const float* src, int64_t rows, int ch, int ld, int col0 = 0) -> int64_t {
        if (rows * ch > capacity) return fail(h, DS_ERR_INVALID, "capacity too small for " + s);
        std::vector<uint16_t> tmp((size_t)rows * ld);
        if (hipMemcpy(tmp.data(), src, tmp.size() * 2, hipMemcpyDeviceToHost) != hipSuccess) return fail(h, DS_ERR_HIP, "hipMemcpy D2H");
        for (int64_t r = 0; r < rows; ++r)
            for (int c = 0; c < ch; ++c) {
                const uint32_t u = (uint32_t)tmp[(size_t)r * ld + col0 + c] << 16;
                memcpy(out + r * ch + c, &u, 4);
            }
        return rows * ch;
    };
    if (h->fold_fc && (s == "signal_feat" || s == "fc1" || s == "joint"))
        return fail(h, DS_ERR_INVALID, "the folded joint model has no " + s + " tensor: use debug mode or DS_TUNE_NO_FOLD_FC");
    if (s == "stem_conv2" && !h->debug && !h->no_fused)
        return fail(h, DS_ERR_INVALID, "conv_layer2's rows stay in LDS (stem23 kernels): the stem_conv2 tap needs debug mode");
    if (h->bf16) {
        if (s == "stem_pool") return copy_bf(h->cur->stem_pool, (int64_t)n * h->wa, 64, 64);
        if (s == "stem_conv2") return copy_bf(h->cur->conv2o, (int64_t)n * h->wa, 128, 128);
        if (s == "stem_conv3") return copy_bf(h->cur->conv3o, (int64_t)n * h->wa, 256, 256);
        if (s == "signal_feat") return copy_bf(h->cur->joint, n, h->SF, h->JP, h->is_rnn ? 2 * HID : 0);
        if (s == "joint") return copy_bf(h->cur->joint, n, h->J, h->JP);
        if (s.rfind("module", 0) == 0) {
            const int m = atoi(s.c_str() + 6) - 1;
            if (m < 0 || m >= NMOD) return fail(h, DS_ERR_INVALID, "bad module index");
            // (outside debug mode the module buffers are shared and, in the bf16 modes, the rows of a chain's inner modules
            // never leave the CU: only the last module's rows exist)
            if (!h->debug && m < NMOD - 1) return fail(h, DS_ERR_INVALID, "module taps other than the last module need debug mode (cfg.reserved[0]=1)");
            return copy_bf(h->cur->modout[m], (int64_t)n * module_width(h, m), INC_OUT, 256);
        }
    }
    if (s == "stem_pool") return copy(h->cur->stem_pool, (int64_t)n * h->wa * 64);
    if (s == "stem_conv2") return copy(h->cur->conv2o, (int64_t)n * h->wa * 128);
    if (s == "stem_conv3") return copy(h->cur->conv3o, (int64_t)n * h->wa * 256);
    if (s == "signal_feat") return copy(h->cur->sigfeat, (int64_t)n * h->SF);
    if (s == "fc1") {
        const int parts = h->cur->last_fc1_parts;
        const int64_t got = copy(h->cur->fc1o, (int64_t)n * h->J);
        if (got < 0 || parts == 1) return got;
        std::vector<float> part((size_t)n * h->J);            // the split dense's partial products, added in launch_head's order
        for (int p = 1; p < parts; ++p) {
            if (hipMemcpy(part.data(), h->cur->fc1o + (size_t)p * h->B * h->J, part.size() * 4, hipMemcpyDeviceToHost) != hipSuccess)
                return fail(h, DS_ERR_HIP, "hipMemcpy D2H");
            for (size_t i = 0; i < part.size(); ++i) out[i] += part[i];
        }
        return got;
    }
    if (s == "logits") return copy(h->cur->logits, (int64_t)n * h->C);
    if (s.rfind("module", 0) == 0) {
        const int m = atoi(s.c_str() + 6) - 1;
        if (m < 0 || m >= NMOD) return fail(h, DS_ERR_INVALID, "bad module index");
        if (!h->debug && m < NMOD - 1) return fail(h, DS_ERR_INVALID, "module taps other than the last module need debug mode (cfg.reserved[0]=1)");
        return copy(h->cur->modout[m], (int64_t)n * module_width(h, m) * INC_OUT);
    }
    if (s.rfind("lstm_", 0) == 0 && s.size() == 10) {   // lstm_fw_l0: device layout [T][B][256] -> [n][T][256]
        const int d = s[5] == 'f' ? 0 : 1, l = s[9] - '0';
        if (l < 0 || l >= NLAYER) return fail(h, DS_ERR_INVALID, "bad lstm layer");
        const int64_t count = (int64_t)n * h->T * HID;
        if (count > capacity) return fail(h, DS_ERR_INVALID, "capacity too small");
        if (h->split) {
            // split cells keep h fragment-major in three bf16 terms: [T][m-tile][k-step s of 16 units][term][lane = 32 * half + r][8]
            // holds term p of units 16 s + 8 half .. + 7 of site 32 * mtile + r (lstm_cell_split_kernel); h = the terms' sum, exactly
            const size_t per_t = (size_t)h->Bp32 * HID * 3;
            std::vector<uint16_t> tb((size_t)h->T * per_t);
            if (hipMemcpy(tb.data(), h->cur->H[d][l], tb.size() * 2, hipMemcpyDeviceToHost) != hipSuccess) return fail(h, DS_ERR_HIP, "hipMemcpy D2H");
            for (int i = 0; i < n; ++i)
                for (int t = 0; t < h->T; ++t)
                    for (int c = 0; c < HID; ++c) {
                        float v = 0.0f;
                        for (int p = 2; p >= 0; --p) {
                            const size_t idx = (size_t)t * per_t + (size_t)(i / 32) * 32 * HID * 3 +
                                               (((size_t)(c / 16) * 3 + p) * 64 + ((c % 16) / 8) * 32 + i % 32) * 8 + c % 8;
                            const uint32_t u = (uint32_t)tb[idx] << 16;
                            float f;
                            memcpy(&f, &u, 4);
                            v += f;
                        }
                        out[((size_t)i * h->T + t) * HID + c] = v;
                    }
            return count;
        }
        if (h->lstm_bf16) {
            // bf16-operand cells keep h fragment-major in bf16: [T][m-tile][k-step s of 16 units][lane = 32 * half + r][8] holds
            // units 16 s + 8 half .. + 7 of site 32 * mtile + r (lstm_cell_bf16_kernel)
            std::vector<uint16_t> tb((size_t)h->T * h->Bp32 * HID);
            if (hipMemcpy(tb.data(), h->cur->H[d][l], tb.size() * 2, hipMemcpyDeviceToHost) != hipSuccess) return fail(h, DS_ERR_HIP, "hipMemcpy D2H");
            for (int i = 0; i < n; ++i)
                for (int t = 0; t < h->T; ++t)
                    for (int c = 0; c < HID; ++c) {
                        const size_t idx = (size_t)t * h->Bp32 * HID + (size_t)(i / 32) * 32 * HID + ((size_t)(c / 16) * 64 + ((c % 16) / 8) * 32 + i % 32) * 8 + c % 8;
                        const uint32_t u = (uint32_t)tb[idx] << 16;
                        memcpy(out + ((size_t)i * h->T + t) * HID + c, &u, 4);
                    }
            return count;
        }
        // fp32 cells keep h MFMA-fragment-major: [T][m-tile][k-group g][lane = 32*half + r][4] holds units
        // 8g + 4*half .. + 3 of site 32*mtile + r (ds_internal.h LstmCell)
        std::vector<float> tmp((size_t)h->T * h->Bp32 * HID);
        if (hipMemcpy(tmp.data(), h->cur->H[d][l], tmp.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) return fail(h, DS_ERR_HIP, "hipMemcpy D2H");
        for (int i = 0; i < n; ++i)
            for (int t = 0; t < h->T; ++t)
                for (int g = 0; g < HID / 8; ++g)
                    for (int half = 0; half < 2; ++half)
                        memcpy(out + ((size_t)i * h->T + t) * HID + 8 * g + 4 * half,
                               tmp.data() + (size_t)t * h->Bp32 * HID + (size_t)(i / 32) * LSTM_MT_FLOATS + ((size_t)g * 64 + half * 32 + i % 32) * 4, 16);
        return count;
    }
    if (s.rfind("lstm_rawstamps", 0) == 0) {
        // "lstm_rawstampsD": per workgroup of diagonal D, 8 floats: entry (10 ns ticks after the first entry), cycles
        // entry -> K loop, K loop, exit part, exit tick, CU key (xcc << 8 | se << 5 | sh << 4 | cu), 0, valid
        if (!h->dbg_lstm) return fail(h, DS_ERR_INVALID, "create the handle with DS_TUNE_DEBUG_STAMPS in ds_config.reserved[2]");
        const int d = atoi(s.c_str() + 14);
        if (d < 0 || d >= 32 || capacity < 1024 * 8) return fail(h, DS_ERR_INVALID, "bad lstm_rawstamps request");
        std::vector<unsigned long long> st(1024 * 8);
        if (hipMemcpy(st.data(), h->dbg_lstm + (size_t)d * 1024 * 8, st.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return fail(h, DS_ERR_HIP, "hipMemcpy D2H");
        unsigned long long t0 = ~0ull;
        for (int wg = 0; wg < 1024; ++wg) if (st[(size_t)wg * 8 + 5]) t0 = std::min(t0, st[(size_t)wg * 8]);
        for (int wg = 0; wg < 1024; ++wg) {
            const unsigned long long* q = &st[(size_t)wg * 8];
            float* o = out + (size_t)wg * 8;
            if (!q[5]) { for (int i = 0; i < 8; ++i) o[i] = 0.f; continue; }
            const unsigned hw = (unsigned)q[6];
            o[0] = (float)(q[0] - t0); o[1] = (float)(q[2] - q[1]); o[2] = (float)(q[3] - q[2]); o[3] = (float)(q[4] - q[3]);
            o[4] = (float)(q[5] - t0);
            o[5] = (float)((((unsigned)q[7] & 0xf) << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xf));
            o[6] = 0.f; o[7] = 1.f;
        }
        return 1024 * 8;
    }
    if (s.rfind("lstm_stamps", 0) == 0) {
        // "lstm_stampsD": diagonal D of the last forward, 12 floats: workgroups, mean / max cycles of [entry -> K loop],
        // [K loop], [exit part], kernel span (us, first entry -> last exit), entry spread (us), mean workgroup life (us),
        // distinct CUs seen, most workgroups on one CU, mean in-kernel clock (GHz)
        if (!h->dbg_lstm) return fail(h, DS_ERR_INVALID, "create the handle with DS_TUNE_DEBUG_STAMPS in ds_config.reserved[2]");
        const int d = atoi(s.c_str() + 11);
        if (d < 0 || d >= 32 || capacity < 12) return fail(h, DS_ERR_INVALID, "bad lstm_stamps request");
        std::vector<unsigned long long> st(1024 * 8);
        if (hipMemcpy(st.data(), h->dbg_lstm + (size_t)d * 1024 * 8, st.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return fail(h, DS_ERR_HIP, "hipMemcpy D2H");
        hipMemset(h->dbg_lstm + (size_t)d * 1024 * 8, 0, 1024 * 8 * 8);
        double sum[3] = {0, 0, 0}, mx[3] = {0, 0, 0}, life = 0, clk = 0;
        unsigned long long t0 = ~0ull, t0max = 0, t1 = 0;
        std::map<unsigned, int> per_cu;
        int cnt = 0;
        for (int wg = 0; wg < 1024; ++wg) {
            const unsigned long long* q = &st[(size_t)wg * 8];
            if (!q[5]) continue;
            ++cnt;
            for (int i = 0; i < 3; ++i) { const double dlt = (double)(q[2 + i] - q[1 + i]); sum[i] += dlt; mx[i] = std::max(mx[i], dlt); }
            t0 = std::min(t0, q[0]); t0max = std::max(t0max, q[0]); t1 = std::max(t1, q[5]);
            life += (double)(q[5] - q[0]) * 0.01;
            if (q[5] > q[0]) clk += (double)(q[4] - q[1]) / ((double)(q[5] - q[0]) * 10.0);
            const unsigned hw = (unsigned)q[6], key = ((unsigned)q[7] & 0xf) << 16 | ((hw >> 13) & 7) << 12 | ((hw >> 12) & 1) << 8 | ((hw >> 8) & 0xf);
            per_cu[key] += 1;
        }
        int most = 0;
        for (auto& kv : per_cu) most = std::max(most, kv.second);
        out[0] = (float)cnt;
        for (int i = 0; i < 3; ++i) { out[1 + 2 * i] = cnt ? (float)(sum[i] / cnt) : 0.f; out[2 + 2 * i] = (float)mx[i]; }
        out[7] = cnt ? (float)((double)(t1 - t0) * 0.01) : 0.f;
        out[8] = cnt ? (float)((double)(t0max - t0) * 0.01) : 0.f;
        out[9] = cnt ? (float)(life / cnt) : 0.f;
        out[10] = (float)per_cu.size();
        out[11] = (float)most;
        if (capacity >= 13) out[12] = cnt ? (float)(clk / cnt) : 0.f;
        return capacity >= 13 ? 13 : 12;
    }
    if (s.rfind("stamps", 0) == 0) {   // "stampsN": phase stamp deltas (cycles) of fused module N, wave 0 and wave 7, averaged over workgroups
        if (!h->dbg_stamps) return fail(h, DS_ERR_INVALID, "create the handle with DS_TUNE_DEBUG_STAMPS in ds_config.reserved[2]");
        const int m = atoi(s.c_str() + 6) - 1;
        if (m < 0 || m >= NMOD || capacity < 16) return fail(h, DS_ERR_INVALID, "bad stamps request");
        std::vector<unsigned long long> st(1024 * 16);
        if (hipMemcpy(st.data(), h->dbg_stamps + (size_t)m * 1024 * 16, st.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return fail(h, DS_ERR_HIP, "hipMemcpy D2H");
        double sum[16] = {0};
        int cnt = 0;
        for (int wg = 0; wg < 1024; ++wg) {
            if (!st[wg * 16 + 7]) continue;
            ++cnt;
            for (int wv = 0; wv < 2; ++wv)
                for (int i = 1; i < 8; ++i) sum[wv * 8 + i] += (double)(st[wg * 16 + wv * 8 + i] - st[wg * 16 + wv * 8 + i - 1]);
        }
        for (int i = 0; i < 16; ++i) out[i] = cnt ? (float)(sum[i] / cnt) : 0.0f;
        out[0] = (float)cnt;
        return 16;
    }
    if (s == "joint") {
        const int64_t count = (int64_t)n * h->J;
        if (count > capacity) return fail(h, DS_ERR_INVALID, "capacity too small");
        std::vector<float> fw((size_t)n * HID), bw((size_t)n * HID), sf((size_t)n * h->SF);
        if (h->is_rnn) {     // fp32 tap (the bf16 modes return above): the row-major copies of the two final h vectors
            hipMemcpy(fw.data(), h->cur->hlast[0], fw.size() * 4, hipMemcpyDeviceToHost);
            hipMemcpy(bw.data(), h->cur->hlast[1], bw.size() * 4, hipMemcpyDeviceToHost);
        }
        if (hipMemcpy(sf.data(), h->cur->sigfeat, sf.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) return fail(h, DS_ERR_HIP, "hipMemcpy D2H");
        const int ev = h->is_rnn ? 2 * HID : 0;
        for (int i = 0; i < n; ++i) {
            if (h->is_rnn) {
                memcpy(out + (size_t)i * h->J, fw.data() + (size_t)i * HID, HID * 4);
                memcpy(out + (size_t)i * h->J + HID, bw.data() + (size_t)i * HID, HID * 4);
            }
            if (h->is_cnn) memcpy(out + (size_t)i * h->J + ev, sf.data() + (size_t)i * h->SF, (size_t)h->SF * 4);
        }
        return count;
    }
    return fail(h, DS_ERR_INVALID, "unknown intermediate " + s);
}

int ds_set_profiling(ds_handle* h, int32_t enable)
{
    if (!h) return DS_ERR_INVALID;
    int rc = ds_sync(h);
    h->profiling = enable < 0 ? 0 : (enable > 3 ? 3 : enable);
    return rc;
}

int ds_num_stages(ds_handle* h) { return h ? (int)h->stages.size() : DS_ERR_INVALID; }

int ds_get_stage(ds_handle* h, int32_t index, char* name, int32_t name_cap, int32_t* launches, double* total_ms,
                 int64_t* calls, double* flops_per_site)
{
    if (!h || index < 0 || index >= (int)h->stages.size()) return DS_ERR_INVALID;
    const Stage& S = h->stages[index];
    if (name && name_cap > 0) { strncpy(name, S.name.c_str(), name_cap - 1); name[name_cap - 1] = 0; }
    if (launches) *launches = S.launches;
    if (total_ms) *total_ms = S.total_ms;
    if (calls) *calls = S.calls;
    if (flops_per_site) *flops_per_site = S.flops_per_site;
    return DS_OK;
}

int ds_reset_stage_times(ds_handle* h)
{
    if (!h) return DS_ERR_INVALID;
    for (Stage& S : h->stages) { S.total_ms = 0; S.calls = 0; }
    for (KernelStat& K : h->kstat) K = KernelStat();
    return DS_OK;
}

int ds_num_kernels(ds_handle* h) { return h ? (int)K_COUNT : DS_ERR_INVALID; }

int ds_get_kernel_stat(ds_handle* h, int32_t index, char* name, int32_t name_cap, int64_t* launches, double* total_ms,
                       double* flops)
{
    if (!h || index < 0 || index >= K_COUNT) return DS_ERR_INVALID;
    if (name && name_cap > 0) { strncpy(name, kKernelNames[index], name_cap - 1); name[name_cap - 1] = 0; }
    if (launches) *launches = h->kstat[index].launches;
    if (total_ms) *total_ms = h->kstat[index].total_ms;
    if (flops) *flops = h->kstat[index].flops;
    return DS_OK;
}

int ds_set_graph(ds_handle* h, int32_t enable)
{
    if (!h) return DS_ERR_INVALID;
    h->use_graph = enable != 0 && !h->shared_s1;      // DS_TUNE_SHARED_EVENT_STREAM handles always issue eagerly
    return DS_OK;
}


int ds_create(const ds_config* cfg, ds_handle** out) { return guarded(nullptr, [&] { return ds_create_impl(cfg, out); }); }
int ds_set_tensor(ds_handle* h, const char* name, const float* data, const int64_t* shape, int32_t ndim) { return guarded(h, [&] { return ds_set_tensor_impl(h, name, data, shape, ndim); }); }
int ds_finalize_weights(ds_handle* h) { return guarded(h, [&] { return ds_finalize_weights_impl(h); }); }
int ds_load_weights(ds_handle* h, const char* path) { return guarded(h, [&] { return ds_load_weights_impl(h, path); }); }
int ds_forward_device(ds_handle* h, int32_t n, const int32_t* d_kmer, const float* d_means, const float* d_stds, const float* d_sanums, const float* d_signals, float* d_act, int32_t* d_pred) { return guarded(h, [&] { return ds_forward_device_impl(h, n, d_kmer, d_means, d_stds, d_sanums, d_signals, d_act, d_pred); }); }
int ds_forward(ds_handle* h, int32_t n, const int32_t* kmer, const float* means, const float* stds, const float* sanums, const float* signals, float* act, int32_t* pred) { return guarded(h, [&] { return ds_forward_impl(h, n, kmer, means, stds, sanums, signals, act, pred); }); }
int ds_submit(ds_handle* h, int32_t n, const int32_t* kmer, const float* means, const float* stds, const float* sanums, const float* signals, int32_t* ticket) { return guarded(h, [&] { return ds_submit_impl(h, n, kmer, means, stds, sanums, signals, ticket); }); }
int ds_submit_parts(ds_handle* h, int32_t nparts, const int32_t* counts, const int32_t* const* kmer, const float* const* means, const float* const* stds, const float* const* sanums, const float* const* signals, int32_t* ticket) { return guarded(h, [&] { return ds_submit_parts_impl(h, nparts, counts, kmer, means, stds, sanums, signals, ticket); }); }
int ds_wait(ds_handle* h, int32_t ticket, float* act, int32_t* pred) { return guarded(h, [&] { return ds_wait_impl(h, ticket, act, pred); }); }
}  // extern "C"
