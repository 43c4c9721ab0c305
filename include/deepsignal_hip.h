/*
 * deepsignal_hip.h — C ABI of the MI355X (gfx950) call_mods inference engine.
 *
 * The reference has no plugin/FFI interface; its de-facto boundary to the compute engine is
 *   Model(...)                         /root/reference/deepsignal/call_modifications.py:203-205
 *   tf.Session + Saver.restore(...)    /root/reference/deepsignal/call_modifications.py:207-212
 *   tf_sess.run([model.activation_logits, model.prediction], feed_dict)
 *                                      /root/reference/deepsignal/call_modifications.py:168-178
 * Every entry point below names the piece of that boundary it replaces. Plain pointers and
 * sizes only; no torch / TensorFlow types. All functions return 0 on success or a negative
 * DS_ERR_* code; ds_last_error() gives the message. A handle is driven by one host thread at a
 * time; different handles (one per GPU) may be driven concurrently.
 */
#ifndef DEEPSIGNAL_HIP_H
#define DEEPSIGNAL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DS_OK 0
#define DS_ERR_INVALID (-1)      /* bad argument / shape / state        */
#define DS_ERR_HIP (-2)          /* a HIP runtime call failed           */
#define DS_ERR_IO (-3)           /* weight file could not be read       */
#define DS_ERR_UNSUPPORTED (-4)  /* configuration not implemented       */
#define DS_ERR_NOMEM (-5)

#define DS_PRECISION_FP32 0
/* BASELINE.json configs[2]: bf16 operands with fp32 accumulation for the signal model's convolutions and the joint
 * FC (activations between those layers are stored as bf16); the BiLSTM, the Cin=1 stem conv, FC2, sigmoid and
 * argmax stay fp32. Same ABI, same outputs to the tolerance stated in DESIGN.md. */
#define DS_PRECISION_BF16 1
/* DS_PRECISION_BF16 plus bf16 operands in the BiLSTM matmuls: h is stored as bf16 and multiplied with bf16 weights,
 * accumulation, gate non-linearities and the cell state c stay fp32, and the layer-0 input projection stays an
 * fp32 table lookup ("fp32 LSTM state/accumulate" reading of configs[2]). */
#define DS_PRECISION_BF16_ALL 2
/* fp32-class results on the bf16 matrix pipe: fp32 activations and weights are carried as THREE bf16 terms
 * (t0 = bf16(x), t1 = bf16(x - t0), t2 = bf16(x - t0 - t1): 24 significant bits, exact) and a product is the fp32-accumulated
 * sum of six bf16 x bf16 term products (the three dropped ones lie below 2^-24 of it). gfx950 runs fp32 MFMAs at 1/16 of the
 * bf16 rate and has no tf32 form, so six products cost 0.375 of the native fp32 matrix time. Stored activations, biases, ReLU,
 * pools, the residual add, gates and the head are fp32 as in DS_PRECISION_FP32, and the mode is held to the SAME parity bars as
 * DS_PRECISION_FP32 (tests/test_gpu_split.py; CPU statement oracle/torch_statement.py::forward_split). Which layers run
 * split is listed by ds_version() / DESIGN.md section 11; the rest runs the DS_PRECISION_FP32 kernels. */
#define DS_PRECISION_BF16X3 3

typedef struct ds_handle ds_handle;

/* Mirrors Model.__init__'s arguments (model.py:26-27) plus placement. */
typedef struct ds_config {
    int32_t kmer_len;     /* base_num,   default 17  (deepsignal.py:258-259) */
    int32_t signal_len;   /* signal_num, default 360 (deepsignal.py:260-262) */
    int32_t class_num;    /* default 2 */
    int32_t is_cnn;       /* model.py:28-29,59-75,89-95 switches (at least one of is_cnn / is_rnn) */
    int32_t is_rnn;
    int32_t is_base;
    int32_t device;       /* HIP device ordinal */
    int32_t precision;    /* DS_PRECISION_FP32 | DS_PRECISION_BF16 | DS_PRECISION_BF16_ALL | DS_PRECISION_BF16X3 */
    int32_t max_batch;    /* largest n per device pass (workspaces are sized for it); larger n is looped */
    int32_t reserved[7];  /* reserved[0] != 0: debug mode — keep every module output for ds_get_intermediate;
                             reserved[1]: forwards in flight for ds_forward_device / ds_submit (pipeline slots, each
                             with its own workspace, stream pair and captured graphs; default 8 for
                             max_batch <= 1024, else 4; max 16);
                             reserved[2]: DS_TUNE_* flag bits (diagnostics, below);
                             reserved[3]: DS_LSTM_TILING_* override of the planner's BiLSTM tile choice;
                             reserved[4]: fused inception module, most sites per tile (0 = default 8);
                             reserved[5]: fused inception module, fewest workgroups a grid is shrunk to when the
                                          batch allows more (0 = default 128);
                             reserved[6]: DS_PRECISION_BF16X3 only: sites per forward from which dense(J, J) of the three-step joint
                                          model runs with split operands instead of the native fp32 GEMM (0 = default: always).
                             Every knob is per handle: the library reads no environment variable and keeps no
                             process-global tuning state, so two handles in one process never influence each other. */
} ds_config;

/* ds_config.reserved[2] bits — bits 1, 2, 4 are diagnostics: results are unchanged (same bits out) */
#define DS_TUNE_NO_FUSED 1       /* layer-granular GEMM launches for the inception modules instead of the fused kernel */
#define DS_TUNE_SERIAL 2         /* every launch of a forward on ONE stream (stand-alone kernel durations)            */
#define DS_TUNE_DEBUG_STAMPS 4   /* attach the s_memtime stamp buffer of the fused kernels (tools/stamps.py)          */
/* not a diagnostic: changes the arithmetic (within rounding). By default the fp32 engine FOLDS the joint model: the two
   dense layers have no bias, no activation and (at inference) an identity dropout between them (layers.py:75-77,257-263),
   and the average pool in front of them is linear too (layers.py:233-238), so
   logits = [h_fw | h_bw | avgpool(module 11)] W1 W2 = [h_fw | h_bw | module 11] W12' with a J x class_num matrix W12'
   computed once per weight load in float64 — like the BN fold, exact in real arithmetic. This bit keeps the reference's
   three steps (avgpool kernel, J x J GEMM, head); debug mode (reserved[0]) implies it, so the fc1 / signal_feat taps exist. */
#define DS_TUNE_NO_FOLD_FC 8
/* diagnostic (same bits out): bf16 modes launch every inception module on its own instead of chaining the modules of one
   width class (layers.py:205-232: 1-3, 4-8, 9-11) inside one launch */
#define DS_TUNE_NO_CHAIN 16
/* Diagnostic (tools/soak.py shared): every pipeline slot's event-model (BiLSTM) launches go to ONE stream shared by all slots and
   forwards are issued eagerly (no captured graphs): the configuration in which round 4's persistent-BiLSTM experiment faulted
   (DESIGN.md section 9). Results are unchanged (same bits); slower than the default. */
#define DS_TUNE_SHARED_EVENT_STREAM 32
/* Timing diagnostic, DS_PRECISION_BF16X3 with the three-step joint model only: dense(J, J) runs its 128 x 96 tile with K in one range
   instead of the 256 x 192 tile with K in 4 / 2 / 1 ranges (by max_batch) whose partial products the head adds up. Same results within
   fp32 summation order. */
#define DS_TUNE_SPLIT_DENSE_NARROW 64
/* Diagnostics (same bits out), DS_PRECISION_BF16X3. By default the first step's layer-0 BiLSTM cells (h = 0: no matrix product, only
   bias + table row + rank-1 terms through the gates; model.py:61-69, layers.py:45-72) are computed by lstm_xproj_kernel, one small
   launch, instead of a cell-kernel launch of their own (19 -> 18 dependent diagonals). NO_LSTM_XPROJ: the 19 launches of rounds 2 - 5.
   LSTM_XPROJ_ALL: the kernel also writes layer 0's accumulator-initial values of ALL steps as an image the cells load (measured:
   142 MB of traffic per 512-site forward cost more in the pipelined step than the cells' own gathers). */
#define DS_TUNE_NO_LSTM_XPROJ 128
#define DS_TUNE_LSTM_XPROJ_ALL 256
/* ds_config.reserved[3] */
#define DS_LSTM_TILING_AUTO 0    /* by forward size */
#define DS_LSTM_TILING_NARROW 1  /* one 32-column n-tile per wave  */
#define DS_LSTM_TILING_WIDE 2    /* four n-tiles per wave          */
#define DS_LSTM_TILING_LDS1 3    /* fp32 cells: operands shared through LDS, 64 x 64 workgroup tile  */
#define DS_LSTM_TILING_LDS2 4    /* fp32 cells: operands shared through LDS, 64 x 128 workgroup tile */
#define DS_LSTM_TILING_WIDE8 5   /* DS_PRECISION_BF16X3 cells: the 128 x 128 workgroup tile by eight waves (two per SIMD) instead of four */

/* Replaces Model(...) + tf.Session(): call_modifications.py:203-209. */
int ds_create(const ds_config *cfg, ds_handle **out);
void ds_destroy(ds_handle *h);
const char *ds_last_error(const ds_handle *h);   /* h may be NULL: last ds_create error */
const char *ds_version(void);

/* Replaces Saver.restore(sess, model_path): call_modifications.py:210-211.
 * Either a DSAMDW01 file (deepsignal_amd/weights.py) ... */
int ds_load_weights(ds_handle *h, const char *path);
/* ... or tensor-by-tensor under the TF variable names of SURVEY.md Appendix B.7, then finalize
 * (folds BN into the conv kernels, pre-packs MFMA operand panels, uploads). */
int ds_set_tensor(ds_handle *h, const char *name, const float *data, const int64_t *shape, int32_t ndim);
int ds_finalize_weights(ds_handle *h);

/* Replaces tf_sess.run([activation_logits, prediction], feed_dict): call_modifications.py:168-178.
 * Host buffers, row-major: kmer int32[n,kmer_len] (base codes, process_utils.py:21);
 * means/stds/sanums float[n,kmer_len]; signals float[n,signal_len]. Outputs: act float[n,class_num]
 * = sigmoid(logits) (NOT normalised — the caller normalises, call_modifications.py:185-187) and
 * pred int32[n] = argmax (ties -> lowest index). Blocking. */
int ds_forward(ds_handle *h, int32_t n, const int32_t *kmer, const float *means, const float *stds,
               const float *sanums, const float *signals, float *act, int32_t *pred);

/* Same contract with every pointer in DEVICE memory of h's GPU (n <= max_batch). Asynchronous and
 * PIPELINED: consecutive calls rotate over independent slots (own workspace, HIP streams and captured
 * graph), so several forwards are in flight at once; inputs are staged and outputs written on the slot's
 * stream, so keep both buffers untouched until ds_sync() returns. Used when features are already
 * resident in HBM. */
int ds_forward_device(ds_handle *h, int32_t n, const int32_t *d_kmer, const float *d_means,
                      const float *d_stds, const float *d_sanums, const float *d_signals,
                      float *d_act, int32_t *d_pred);
int ds_sync(ds_handle *h);

/* Asynchronous form of ds_forward for host buffers (n <= max_batch): ds_submit stages the batch in pinned memory of
 * the next pipeline slot and enqueues H2D + forward + D2H there; ds_wait blocks on that one forward and copies its
 * act[n, class_num] / pred[n] out. Tickets must be waited in submission order at the latest when all slots
 * (ds_config.reserved[1], default 8) are in flight. Replaces the blocking tf_sess.run (call_modifications.py:177-178)
 * where the caller has other work to overlap (parsing the next queue item, formatting the previous rows). */
int ds_submit(ds_handle *h, int32_t n, const int32_t *kmer, const float *means, const float *stds,
              const float *sanums, const float *signals, int32_t *ticket);
int ds_wait(ds_handle *h, int32_t ticket, float *act, int32_t *pred);
/* ds_submit with the batch given as `nparts` row segments (counts[i] rows from the i-th pointer of each array): the
 * rows are gathered straight into the slot's pinned staging buffer, sum(counts) in [1, max_batch]. For callers whose
 * batches straddle their own buffers — call_mods fills a batch from the tail of one queue item and the head of the
 * next (call_modifications.py:157-166 cuts batches inside one item) — so that they need no concatenated copy. */
int ds_submit_parts(ds_handle *h, int32_t nparts, const int32_t *counts, const int32_t *const *kmer,
                    const float *const *means, const float *const *stds, const float *const *sanums,
                    const float *const *signals, int32_t *ticket);
/* Forwards that may be in flight at once (pipeline slots of this handle). */
int ds_num_slots(ds_handle *h);

/* Pinned host allocation helpers for callers that want async H2D/D2H overlap. */
int ds_alloc_host(size_t bytes, void **out);
int ds_free_host(void *p);

/* Test/diagnostic access to intermediate tensors of the LAST forward (float32, row-major, same
 * names/shapes as the oracle taps: stem_pool, stem_conv2, stem_conv3, module1..module11,
 * signal_feat, lstm_{fw,bw}_l{0,1,2}, joint, fc1, logits). Returns the number of floats written,
 * or a negative error. A tensor the configured path does not materialise is an error, not stale data: outside debug
 * mode (reserved[0]) that is stem_conv2 and module1..module10 (rows that stay in LDS / shared buffers), and with the
 * folded joint model signal_feat, joint and fc1. */
int64_t ds_get_intermediate(ds_handle *h, const char *name, float *out, int64_t capacity);

/* Timing with HIP events on the engine's own streams (forwards run eagerly while it is on).
 *   mode 1: one event pair around every RUN of consecutive launches of the same kernel on a stream
 *           (per-kernel statistics with negligible bracketing overhead; inter-launch gaps included);
 *   mode 2: one event pair per launch (per-stage breakdown; ~10 us of bracketing per launch);
 *   mode 3: mode 1 with every launch of the forward on ONE stream (stand-alone kernel times: nothing
 *           else is resident on the GPU while a kernel runs);
 *   mode 0: off.
 * ds_get_stage reports, for stage index i, its name, launches per forward, accumulated device
 * milliseconds (mode 2) and forward count since the last reset. */
int ds_set_profiling(ds_handle *h, int32_t mode);
int ds_num_stages(ds_handle *h);
int ds_get_stage(ds_handle *h, int32_t index, char *name, int32_t name_cap, int32_t *launches,
                 double *total_ms, int64_t *calls, double *flops_per_site);
int ds_reset_stage_times(ds_handle *h);
/* Per-kernel accumulators of the same profiling mode: every launch is bracketed by its own HIP
 * event pair on the stream it runs on. name = the __global__ function (as rocprofv3 prints it),
 * launches / total_ms / flops = launch count, summed device time and summed ALGORITHMIC FLOPs
 * (2*M*N*K of the GEMMs the launch carries) since the last reset. */
int ds_num_kernels(ds_handle *h);
int ds_get_kernel_stat(ds_handle *h, int32_t index, char *name, int32_t name_cap, int64_t *launches,
                       double *total_ms, double *flops);

/* ---- scope row f1: native feature-TSV reader and result-row formatter (host code) -------------------
 * Replaces _read_features_file (call_modifications.py:35-91): 12 tab-separated columns
 * (extract_features.py:289-303), rows grouped by read id (column 5). ds_tsv_next() parses the rows of
 * the next `max_reads` reads (= one queue item, f5_batch_num) with `nthreads` host threads and
 * returns the number of sites (0 at end of file, negative on a malformed row; ds_tsv_error()). The
 * accessors return the item's arrays, valid until the next ds_tsv_next(): kmer int32[n,kmer_len]
 * (A,C,G,T,N -> 0..4), means/stds/lens float[n,kmer_len], signals float[n,signal_len], labels int32[n],
 * and the verbatim first six columns ("sampleinfo") as one char buffer + int64 offsets[n+1]. */
typedef struct ds_tsv ds_tsv;
int ds_tsv_open(const char *path, int32_t kmer_len, int32_t signal_len, int32_t nthreads, ds_tsv **out);
void ds_tsv_close(ds_tsv *t);
const char *ds_tsv_error(const ds_tsv *t);
int64_t ds_tsv_next(ds_tsv *t, int32_t max_reads);
/* The same in two steps, for a caller that owns the destination arrays (no copy out of the reader): ds_tsv_locate() finds
 * the rows of the next item and returns their number n; ds_tsv_parse_into() parses them into kmer int32[n,kmer_len],
 * means / stds / lens float[n,kmer_len], signals float[n,signal_len], labels int32[n] (the sampleinfo columns stay
 * behind ds_tsv_info / ds_tsv_info_offsets) and returns n, or a negative code on a malformed row.
 * Contract: EXACTLY ONE ds_tsv_parse_into() per successful ds_tsv_locate() with n > 0. ds_tsv_locate() advances the reader, so a
 * second locate while rows are pending is refused (DS_ERR_INVALID) instead of silently dropping the item; capacity_rows is the
 * number of rows the caller's arrays hold and must be >= n (DS_ERR_INVALID otherwise, nothing written); after a parse error
 * (malformed row) the located rows stay pending: ds_tsv_set_range() (a rewind) or ds_tsv_close() are the ways on. */
int64_t ds_tsv_locate(ds_tsv *t, int32_t max_reads);
int64_t ds_tsv_parse_into(ds_tsv *t, int64_t capacity_rows, int32_t *kmer, float *means, float *stds, float *lens, float *signals,
                          int32_t *labels);
/* Multi-GPU call_mods (sites sharded BY READ): each rank parses only its own byte ranges of the file.
 * ds_tsv_align(t, pos) = the first read boundary at or after byte pos (start of the first line beginning at or after
 * pos whose read id differs from the line before it; 0 -> 0; file size when none follows), a function of the file
 * alone, so all ranks agree on the cut points without communicating; ds_tsv_set_range(t, begin, end) restricts
 * ds_tsv_next to [begin, end) and rewinds. The reference has no counterpart (single reader process,
 * call_modifications.py:453). */
int64_t ds_tsv_size(const ds_tsv *t);
int64_t ds_tsv_align(const ds_tsv *t, int64_t pos);
int ds_tsv_set_range(ds_tsv *t, int64_t begin, int64_t end);
const int32_t *ds_tsv_kmer(const ds_tsv *t);
const float *ds_tsv_means(const ds_tsv *t);
const float *ds_tsv_stds(const ds_tsv *t);
const float *ds_tsv_lens(const ds_tsv *t);
const float *ds_tsv_signals(const ds_tsv *t);
const int32_t *ds_tsv_labels(const ds_tsv *t);
const char *ds_tsv_info(const ds_tsv *t);
const int64_t *ds_tsv_info_offsets(const ds_tsv *t);
/* Replaces the per-site formatting loop of _call_mods (call_modifications.py:183-190): rows
 * "sampleinfo \t p0/(p0+p1) \t p1/(p0+p1) \t label \t kmer \n" with float32 arithmetic and the
 * shortest round-trip float32 text str(np.float32) prints. Returns bytes written or -(bytes needed). */
int64_t ds_format_rows(int64_t n, const char *info, const int64_t *info_off, const float *act,
                       int32_t class_num, const int32_t *pred, const int32_t *kmer, int32_t kmer_len,
                       char *out, int64_t cap);

/* ---- scope row f3: TensorFlow checkpoint import (deepsignal_amd/tf_checkpoint.py) ----
 * CRC-32C (Castagnoli) of a host buffer, continuing from `crc` (0 to start): the checksum TensorFlow's
 * Saver stores (masked) for every tensor and table block of the checkpoints the reference restores with
 * tf.train.Saver().restore (call_modifications.py:210-211). Host code only. */
uint32_t ds_crc32c(const void *data, size_t n, uint32_t crc);

/* Use a captured hipGraph for the forward (default on). */
int ds_set_graph(ds_handle *h, int32_t enable);

#ifdef __cplusplus
}
#endif
#endif /* DEEPSIGNAL_HIP_H */
