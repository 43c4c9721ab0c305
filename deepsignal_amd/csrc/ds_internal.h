// ds_internal.h — structures shared by the host planner (ds_engine.cpp) and the gfx950 kernels
// (ds_kernels.hip). Not part of the public ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ds {

constexpr int KC = 16;          // K elements staged per LDS chunk (two 8-wide MFMA read steps)
constexpr int MAX_SEG = 6;      // A-operand segments (conv taps / concatenated inputs)
constexpr int MAX_OUT = 6;      // output column segments of one GEMM
constexpr int MAX_PROB = 8;     // problems per grouped launch

// One K-segment of the implicit-GEMM A operand: rows come from `base` (row stride ld, floats),
// shifted by row_shift rows; a shifted row is read as zeros when it leaves its site
// [0, W) (TF SAME zero padding) or the matrix.
struct ASeg {
    const float* base;
    int ld;
    int row_shift;
    int klen;      // multiple of KC
    int pad_;
};

// Output column segment: GEMM columns [col0, col0+ncols) go to base[row*ld + (col-col0)].
struct OSeg {
    float* base;
    const float* add;   // optional residual addend, same indexing with add_ld
    int ld;
    int add_ld;
    int col0;
    int ncols;
    int relu;
    int bf16;           // 1: base is a bf16 buffer (ld / col offsets in bf16 elements), values rounded to nearest even
};

struct GemmProblem {
    int M, N, K, W;             // K = sum of the used segments' klen
    int a_mode;                 // informational: 1 = maxpool(3, stride 1, SAME) on load (kernel variant CFG_CONV_POOL)
    int n_fast;                 // tile order inside the problem: 1 = n-tiles of one m-tile are neighbours (A-heavy shapes)
    int nseg;
    int kgroups_stride;         // K-groups (of 8) per packed n-tile panel in Bp
    int nout;
    ASeg seg[MAX_SEG];
    OSeg out[MAX_OUT];
    const float* Bp;            // pre-packed weights [ntile][kgroup][64 lanes][4]
    const float* bias;          // [N] (folded BN shift)
    int tiles_m, tiles_n, tile_start, ntiles32;   // ntiles32 = ceil(N/32)
};

struct GemmLaunch {
    GemmProblem prob[MAX_PROB];
    int nprob;
    int total_tiles;
    int pad_[2];
};

// (the numbering keeps the gaps the BiLSTM configurations of rounds 1 - 2 left: the cells have their own kernels now)
enum GemmCfg { CFG_CONV = 0, CFG_FC = 1, CFG_CONV_WIDE = 3, CFG_CONV_POOL = 4, CFG_FC_DENSE = 5,
               // bf16-operand variants (mixed-precision mode)
               CFG_BCONV = 7, CFG_BCONV_POOL = 8, CFG_BFC = 9, CFG_BFC_DENSE = 10 };

// ---- fp32 BiLSTM cell launch (lstm_cell_kernel) ---------------------------------------------------------------
// h and c of the fp32 BiLSTM live in MFMA-FRAGMENT-MAJOR buffers: [m-tile of 32 sites][k-group of 8 units][64 lanes][4]
// where lane (site r = lane & 31, half = lane >> 5) holds units 8g + 4*half .. + 3 of site 32*mtile + r. That is
// exactly what one lane of a cell's epilogue produces (four neighbouring units of one site) AND exactly the float4 a
// lane feeds to four consecutive v_mfma_f32_32x32x2_f32 as the activation operand of the next step, so h goes
// global -> VGPR -> MFMA with fully coalesced 1 KiB wave loads: no LDS, no barrier, no staging arithmetic.
constexpr int LSTM_MAX_CELLS = 6;
constexpr int DBG_MAX_WGS = 1024;       // workgroups per launch the diagnostic stamp buffers hold (later ones do not stamp)
constexpr int LSTM_MT_FLOATS = 32 * 256;      // floats of one m-tile (32 sites x 256 units) in a fragment-major buffer
struct LstmCell {
    const float* ax;      // x operand: h of the layer below at this step (fragment-major), nullptr for layer 0
    const float* ah;      // h operand: this cell's h at the previous step (fragment-major), nullptr at the first step
    const float* Bp;      // packed weights [32 n-tiles][kg_stride][64][4]; rows = x rows then h rows (layer 0: h rows only);
                          // n-tile p, column i = gate (i >> 3) of unit 8p + (i & 7)
    const float* bias;    // [4][256], TF column order i, j, f, o
    const float* table;   // layer 0, is_base: [vocab][1024] embedding x W_x; else nullptr
    const float* wfeat;   // layer 0: [3][1024] rows of W_x for (mean, std, len)
    const int* codes;     // [n][T]
    const float* means;   // [n][T]
    const float* stds;
    const float* lens;
    const float* xinit;   // layer 0, optional: this step's accumulator-initial-value image [m-tile][32 n-tiles][4 gates][64 lanes][4] written by
                          // lstm_xproj_kernel (bias + forget bias + table row + rank-1 terms, lstm_acc_init's bits); nullptr: computed in the cell
    float* c;             // cell state, fragment-major (read-modify-write)
    float* h_out;         // fragment-major
    float* h_row;         // optional row-major [n][256] copy of h_out (what the joint FC reads), or nullptr
    int kg_stride;        // k-groups per n-tile panel in Bp (32 or 64)
    int t;                // original time index of this step (layer-0 feature gather)
    int use_feat;         // 1: layer 0 -- (mean, std, len) rank-1 terms (+ table row) enter the accumulator's initial value
    int c_zero;           // 1: previous c is zero (first step)
};
struct LstmLaunch {
    LstmCell cell[LSTM_MAX_CELLS];
    int ncell, n, mtiles, T;
    int cls_tiles[2];          // workgroup tiles of the K = 512 cells and of the K = 256 cells (cells are sorted by K)
    unsigned long long* dbg;   // diagnostic (DS_TUNE_DEBUG_STAMPS): [DBG_MAX_WGS workgroups][8] time stamps of wave 0, null in normal runs
};
// Layer 0's input projection for ALL steps of both directions in one launch, in front of the diagonals (ds_split.hip): x_t W_x is
// a table row plus three rank-1 terms plus the bias (model.py:61-69 folded at load), i.e. it depends on the forward's inputs only.
// A 128 x 128 cell tile otherwise gathers ~80 L2-hot float4 per lane through the texture path in front of its K loop (16 - 24 k
// cycles); with the image a layer-0 cell starts like any other: four coalesced loads per 32 x 32 tile. The first
// step's cells (no h yet: K = 0) are finished here as well -- gates, c, h -- so the 2-cell diagonal 0 is no launch at all.
struct LstmXproj {
    LstmCell cell[2];          // per direction: bias, table, wfeat, codes / means / stds / lens, c = the layer-0 cell state, h_out = H[dir][0] (t = 0)
    float* xinit[2];           // [T][m-tiles][32][4][64][4] floats per direction
    size_t h_step, x_step;     // floats between two time steps of h_out / xinit
    int n, mtiles, T;
    int nsteps;                // steps the launch covers: 1 = the first step's cells only (no image: the later cells compute their own initial
                               // values), T = all of them
};
hipError_t launch_lstm_xproj(const LstmXproj& X, hipStream_t s);
// nt = 32-column n-tiles per wave (1, 2 or 4): the same bits for every nt (same K order per output element)
hipError_t launch_lstm_cells(int nt, const LstmLaunch& L, hipStream_t s);      // L travels as a by-value kernel argument

// tile geometry per config (host needs it for grid sizing)
struct TileGeom { int bm, bn, threads, ksplit; };
TileGeom gemm_geom(GemmCfg cfg);

hipError_t launch_gemm(GemmCfg cfg, const GemmLaunch* d_launch, int total_tiles, hipStream_t s);

// One fused inception module (layers.py:87-139): every conv+BN(+ReLU) of the module, the
// stride-1 maxpool of branch 1 and the residual add, with all intermediates kept in LDS.
struct FusedArgs {
    const float* X;      // [n_sites*W, cin] NWC input
    float* Y;            // [n_sites*W, 240] output (concat order b1|b2|b3|b4|b5)
    int n_sites, W, cin, spt;   // spt = sites per workgroup tile (whole sites only: taps never cross tiles)
    int pool_win, pool_pad;     // pool_win > 0: X is [n_sites*pool_win, cin] and the module input is its
                                // maxpool(3, stride 2, SAME) (pad_left = pool_pad), taken while staging (layers.py:211-213,224-226)
    const float* Bp1;    // packed [cin x 256]: columns b5s(48)|b2(48)|b3a(32)|b4a(32)|b5a(32)|b1(48, pooled input)|pad
    const float* bias1;  // [256]
    const float *Bp3b, *bias3b, *Bp4b, *bias4b, *Bp5b, *bias5b, *Bp5c, *bias5c;
    unsigned long long* dbg;   // diagnostic: per-workgroup phase time stamps (null in normal runs)
    int write_rows;      // bf16 chain: 1 = this module's rows go to Y (the chain's last module; every module when taps are on), 0 = they
                         // stay in LDS as the next module's input. The fp32 kernel writes every module's rows.
};

size_t inception_fused_lds_bytes(int tm, int W, int spt);
// once per device, before the first fused launch (raises the kernels' dynamic LDS limit to 160 KB)
hipError_t configure_fused_kernels();
// bf16-operand variant: X / Y are bf16 rows of 256-channel pitch, a.cin is the row pitch in 4-byte units (128).
// A launch (either precision) carries a CHAIN of consecutive modules of one width class (same W, spt, n_sites; m[k + 1].X == m[k].Y; only m[0]
// may pool its input): every workgroup takes its tile of whole sites through all of them and no launch boundary separates the
// modules. fp32: the rows a module reads are the rows the same workgroup wrote a moment ago (XCD-local L2 hits). bf16: the rows
// stay in LDS from module to module; only m[0] reads X and only modules with write_rows store Y.
constexpr int FUSED_CHAIN_MAX = 5;
struct FusedChain {
    FusedArgs m[FUSED_CHAIN_MAX];
    int nmod;
};
hipError_t launch_inception_fused_bf16(int tm, const FusedChain& c, hipStream_t s);
// fp32 form of the same chain; tm = 32-row m-tiles per workgroup (1..3), spt*W <= 32*tm
hipError_t launch_inception_fused(int tm, const FusedChain& c, hipStream_t s);
size_t inception_fused_bf16_lds_bytes(int tm, int W, int spt);
// split-operand form (ds_split.hip, DS_PRECISION_BF16X3): X / Y are the fp32 engine's fp32 rows, Bp1 / Bp3b / Bp4b / Bp5b / Bp5c point at
// three-term panels (ds_engine.cpp pack_b_split); as in the fp32 form only the chain's first module may pool its input
hipError_t launch_inception_fused_split(int tm, const FusedChain& c, hipStream_t s);
size_t inception_fused_split_lds_bytes(int tm, int W, int spt);
hipError_t configure_split_kernels();
// BiLSTM cells of one diagonal with split operands (ds_split.hip): h buffers are fragment-major term images
// [m-tile][k-step of 16 units][term][64 lanes][8 bf16] (48 KiB per m-tile), C.Bp points at pack_b_split panels, kg_stride = k-steps per
// n-tile panel; c, bias, table, the gates and h_row as in the fp32 cells. tile = 11 | 12 | 22: workgroup tile of 64 x 64 .. 128 x 128.
hipError_t launch_lstm_cells_split(int tile, const LstmLaunch& L, hipStream_t s);
// dense(J, J) of the three-step joint model with split operands: pack_joint_split_kernel writes the joint rows (three fp32 row
// segments per site) as a fragment-major term image A, dense_split_kernel multiplies it with W1's pack_b_split panels into C [n][N]
struct SplitDense {
    const float* seg[3];   // joint row segments, row-major fp32 [n][len]; unused ones have len 0 (lengths: multiples of 8)
    int len[3];
    char* A;               // [mtiles][ksteps][3][1 KiB] workspace
    const char* Bp;        // [ntiles_alloc][kg_stride][3][1 KiB]
    float* C;              // [splits][n][N] fp32: partial sums over `splits` ranges of K (1: the product itself)
    int n, N, mtiles, ntiles, ntiles_alloc, ksteps, kg_stride;
    int splits;            // K in `splits` ranges (256 x 192 tile only); the reader sums the partial products in order (launch_head)
    size_t part_stride;    // floats between two partial products
    bool wide;             // the 256 x 192 tile (K in `splits` ranges); false: 128 x 96, one range
};
hipError_t launch_dense_split(const SplitDense& d, hipStream_t s);

// conv_layer2 (1x1, 64 -> 128) + conv_layer3 (1x3, 128 -> 256), both with folded BN + ReLU        layers.py:192-203
struct Stem23Args {
    const float* X;        // [n_sites * W][64]  stem_pool rows (NWC)
    float* Y;              // [n_sites * W][256] conv_layer3 output
    float* C2;             // optional [n_sites * W][128] copy of conv_layer2's output (debug taps), or nullptr
    const float *Bp2, *bias2;   // packed [64 x 128] (pack_b) and its bias
    const float *Bp3, *bias3;   // packed [384 x 256], K = tap-major (tap * 128 + channel)
    int n_sites, W, spt;   // spt = whole sites per workgroup tile, spt * W <= 96
};
hipError_t launch_stem23(const Stem23Args& a, hipStream_t s);
// conv_layer2 + conv_layer3 with split operands (ds_split.hip): X / Y / C2 as the fp32 kernel, Bp2 / Bp3 = pack_b_split panels
hipError_t launch_stem23_split(const Stem23Args& a, hipStream_t s);
size_t stem23_split_lds_bytes(int W, int spt);
size_t stem23_lds_bytes(int W, int spt);
// bf16 form (DS_PRECISION_BF16*): X / Y / C2 are bf16 rows (64 / 256 / 128 channels), Bp2 / Bp3 packed by pack_b_bf16
hipError_t launch_stem23_bf16(const Stem23Args& a, hipStream_t s);
size_t stem23_bf16_lds_bytes(int W, int spt);
constexpr size_t STEM23_MAX_LDS = 80 * 1024;     // dynamic LDS stem23_kernel may ask for (two workgroups per CU)
// stem conv1 (K=7, stride 2, Cin=1) + folded BN + ReLU + maxpool(3, stride 2)   layers.py:183-191
hipError_t launch_stem1(const float* signals, const float* w7x64, const float* bias64, float* out,
                        int n, int signal_len, int w1, int pad_l_conv, int wa, int pad_l_pool, int out_bf16, hipStream_t s);
// bf16-mode pools: activations are [rows][ch_ld] bf16 (ch_ld = channels padded to a multiple of 32, pad = 0)
hipError_t launch_maxpool_s2_bf16(const float* in, float* out, int n, int win, int wout, int pad_l, int ch_ld, hipStream_t s);
hipError_t launch_avgpool7_bf16(const float* in, float* joint, int n, int w, int ch, int ld_in, int joint_ld, int joint_off,
                                hipStream_t s);
hipError_t launch_pack_event_feat_bf16(const float* hfw, const float* hbw, float* joint, int n, int joint_ld, int src_bf16,
                                       hipStream_t s);
// maxpool(3, stride 2, SAME) over [n, win, ch] -> [n, wout, ch]                 layers.py:211-213,224-226
hipError_t launch_maxpool_s2(const float* in, float* out, int n, int win, int wout, int pad_l, int ch, hipStream_t s);
// avgpool(7, stride 1, SAME, divisor = valid taps) + flatten                    layers.py:233-238
hipError_t launch_avgpool7(const float* in, float* out, int n, int w, int ch, hipStream_t s);
// fc2 (J x class_num, no bias) + sigmoid + argmax                               layers.py:261-263, model.py:100,108
hipError_t launch_head(const float* fc1, const float* w2, float* logits, float* act, int* pred,
                       int n, int J, int class_num, hipStream_t s, int nparts = 1, size_t part_stride = 0);      // fc1 = sum of nparts partial products
// Folded joint model (DS_TUNE_NO_FOLD_FC off): logits = x W12', x = up to three row segments per site (h_fw(T-1), h_bw(0),
// module-11 rows), W12' [J][class_num] k-major; then sigmoid + argmax as launch_head        layers.py:233-238,247-264
struct HeadFoldedArgs {
    const float* seg[3];
    int len[3];           // floats per site in the segment (multiple of 4)
    int nseg;
    const float* w;       // [sum len][C]
    int bf16;             // 1: the segments are bf16 rows, len[s] values of a site's pitch[s] (the bf16 modes: [bf16 h_fw | h_bw] and module 11's
                          // rows of 256-channel pitch, whose pad channels meet zero weight rows)
    int pitch[3];
    float *logits, *act;
    int* pred;
    int n, C;
};
hipError_t launch_head_folded(const HeadFoldedArgs& a, hipStream_t s);
// the caller's device arrays -> the slot's contiguous input block [kmer | means | stds | lens | signals] (regions sized for
// max_batch B), and the slot's act / pred -> the caller's buffers
hipError_t launch_gather_inputs(const int* kmer, const float* means, const float* stds, const float* lens, const float* signals,
                                float* block, int n, int T, int S, int B, hipStream_t s);
hipError_t launch_scatter_outputs(const float* act, const int* pred, float* act_out, int* pred_out, int n, int C, hipStream_t s);
// table[v][c] = sum_e emb[v][e] * kernel[e][c]  (embedding folded into layer-0 W_x; model.py:61-69)
hipError_t launch_embed_table(const float* emb, const float* kernel, float* table, int vocab, int esize, int ncol, hipStream_t s);

}  // namespace ds
