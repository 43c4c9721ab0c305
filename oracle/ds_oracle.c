/*
 * ds_oracle.c — CPU restatement of deepsignal's call_mods forward pass.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE. Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it — as the checker / the timed CPU baseline, never as
 * the thing shipped. The product path (deepsignal_amd/csrc) never links or calls this file.
 *
 * PARITY UNPINNED BY THE REFERENCE: the arithmetic of this path lives in TensorFlow 1.x
 * (1.8.0 <= v <= 1.13.1, /root/reference/README.md:28), which is not vendored, cannot be installed
 * here, and the reference ships no tests / golden vectors for it. This file
 * restates the published TF-1.x op semantics at the reference's own call
 * sites, cited per function below; it is cross-validated against an independent PyTorch-CPU
 * statement (tests/torch_statement.py), which is the strongest pin available.
 *
 * Build: `make -C oracle` -> oracle/_build/libds_oracle_f32.so (REAL=float) and
 *        libds_oracle_f64.so (REAL=double; same code, used to judge which fp32 side is "right").
 *
 * Tensor order = deepsignal_amd/spec.py::tensor_table (580 tensors for the full model).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifdef ORACLE_F64
typedef double real;
#define R_EXP exp
#define R_TANH tanh
#define R_SQRT sqrt
#else
typedef float real;
#define R_EXP expf
#define R_TANH tanhf
#define R_SQRT sqrtf
#endif

#define VOCAB 1024
#define EMB 128
#define HID 256
#define NLAYER 3
#define NMOD 11
#define INC_OUT 240
#define BN_EPS ((real)1e-3)

typedef struct {
    const float *kernel, *beta, *gamma, *mean, *var;
    int k, cin, cout, stride, relu;
} convbn_t;

typedef struct {
    int kmer_len, signal_len, class_num;
    int is_cnn, is_rnn, is_base;     /* model.py:28-29,59-75,89-95 */
    int w1, wa, wb, wc, joint;
    const float *embedding;
    const float *lstm_kernel[2][NLAYER];
    const float *lstm_bias[2][NLAYER];
    convbn_t stem[3];
    convbn_t mod[NMOD][10];     /* b1 b2 b3a b3b b4a b4b b5s b5a b5b b5c */
    const float *fc1, *fc2;
} net_t;

/* Optional intermediate outputs (all float32, row-major, NULL = skip). */
typedef struct {
    float *stem_pool;     /* [n, wa, 64]   after conv1+BN+ReLU+maxpool (layers.py:183-191) */
    float *stem_conv2;    /* [n, wa, 128]  layers.py:192-197 */
    float *stem_conv3;    /* [n, wa, 256]  layers.py:198-203 */
    float *module_out[NMOD]; /* [n, W_m, 240] module outputs BEFORE the inter-group maxpools */
    float *signal_feat;   /* [n, wc*240]   after avgpool + flatten (layers.py:233-238) */
    float *lstm_h[2][NLAYER]; /* [n, kmer_len, 256] hidden outputs in ORIGINAL time order */
    float *joint;         /* [n, joint]    layers.py:250-252 */
    float *fc1;           /* [n, joint]    layers.py:257-259 */
    float *logits;        /* [n, class_num] layers.py:261-263 */
} ds_oracle_taps;

/* TF 'SAME' rule. */
static void same_pad(int in, int k, int s, int *out, int *pl)
{
    int o = (in + s - 1) / s;
    int tot = (o - 1) * s + k - in;
    if (tot < 0) tot = 0;
    *out = o;
    *pl = tot / 2;
}

static real sigmoid_r(real x) { return (real)1 / ((real)1 + R_EXP(-x)); }

/* tf.layers.conv2d(use_bias=False, padding="SAME") with H=1 (cross-correlation, HWIO kernel),
 * then tf.contrib.layers.batch_norm inference (epsilon=1e-3, moving stats), then optional ReLU.
 * Reference call sites: layers.py:80-84 (BN wrapper), :91-135, :184-203. */
static int conv_bn(const convbn_t *L, const real *in, int win, real *out)
{
    int wout, pl;
    same_pad(win, L->k, L->stride, &wout, &pl);
    const int cin = L->cin, cout = L->cout;
    for (int w = 0; w < wout; ++w) {
        real *o = out + (size_t)w * cout;
        for (int c = 0; c < cout; ++c) o[c] = 0;
        for (int t = 0; t < L->k; ++t) {
            int iw = w * L->stride + t - pl;
            if (iw < 0 || iw >= win) continue;          /* zero padding */
            const real *x = in + (size_t)iw * cin;
            const float *kt = L->kernel + (size_t)t * cin * cout;
            for (int ci = 0; ci < cin; ++ci) {
                const real xv = x[ci];
                const float *kr = kt + (size_t)ci * cout;
                for (int c = 0; c < cout; ++c) o[c] += xv * (real)kr[c];
            }
        }
        for (int c = 0; c < cout; ++c) {
            real sc = (real)L->gamma[c] * ((real)1 / R_SQRT((real)L->var[c] + BN_EPS));
            real y = (o[c] - (real)L->mean[c]) * sc + (real)L->beta[c];
            if (L->relu && y < 0) y = 0;
            o[c] = y;
        }
    }
    return wout;
}

/* tf.layers.max_pooling2d([1,3], strides=s, SAME): padded taps are ignored.
 * layers.py:90-91 (s=1), :189-191, :211-213, :224-226 (s=2). */
static int maxpool3(const real *in, int win, int ch, int stride, real *out)
{
    int wout, pl;
    same_pad(win, 3, stride, &wout, &pl);
    for (int w = 0; w < wout; ++w)
        for (int c = 0; c < ch; ++c) {
            real m = 0; int have = 0;
            for (int t = 0; t < 3; ++t) {
                int iw = w * stride + t - pl;
                if (iw < 0 || iw >= win) continue;
                real v = in[(size_t)iw * ch + c];
                if (!have || v > m) { m = v; have = 1; }
            }
            out[(size_t)w * ch + c] = m;
        }
    return wout;
}

/* tf.layers.average_pooling2d([1,7], strides=1, SAME): divisor = in-bounds taps. layers.py:233-235 */
static void avgpool7(const real *in, int win, int ch, real *out)
{
    for (int w = 0; w < win; ++w)
        for (int c = 0; c < ch; ++c) {
            real s = 0; int cnt = 0;
            for (int t = -3; t <= 3; ++t) {
                int iw = w + t;
                if (iw < 0 || iw >= win) continue;
                s += in[(size_t)iw * ch + c]; ++cnt;
            }
            out[(size_t)w * ch + c] = s / (real)cnt;
        }
}

/* inception_layer, layers.py:87-139. in [w][cin] -> out [w][240]; tmp holds >= w*(cin+48+32+64+48) reals */
static void inception(const convbn_t *M, const real *in, int w, int cin, real *out, real *tmp)
{
    real *pooled = tmp;                       /* [w][cin] */
    real *t48 = pooled + (size_t)w * cin;     /* [w][48]  */
    real *t32 = t48 + (size_t)w * 48;         /* [w][32]  */
    real *t64 = t32 + (size_t)w * 32;         /* [w][64]  */
    real *t48b = t64 + (size_t)w * 64;        /* [w][48]  */
    /* branch1: maxpool(3,s1) -> 1x1 48 -> BN -> ReLU                  layers.py:90-96 */
    maxpool3(in, w, cin, 1, pooled);
    conv_bn(&M[0], pooled, w, t48);
    for (int i = 0; i < w; ++i) memcpy(out + (size_t)i * INC_OUT + 0, t48 + (size_t)i * 48, 48 * sizeof(real));
    /* branch2: 1x1 48 -> BN -> ReLU                                    layers.py:97-101 */
    conv_bn(&M[1], in, w, t48);
    for (int i = 0; i < w; ++i) memcpy(out + (size_t)i * INC_OUT + 48, t48 + (size_t)i * 48, 48 * sizeof(real));
    /* branch3: 1x1 32 -> BN -> ReLU -> 1x3 48 -> BN -> ReLU            layers.py:102-110 */
    conv_bn(&M[2], in, w, t32);
    conv_bn(&M[3], t32, w, t48);
    for (int i = 0; i < w; ++i) memcpy(out + (size_t)i * INC_OUT + 96, t48 + (size_t)i * 48, 48 * sizeof(real));
    /* branch4: 1x1 32 -> BN -> ReLU -> 1x5 48 -> BN -> ReLU            layers.py:111-119 */
    conv_bn(&M[4], in, w, t32);
    conv_bn(&M[5], t32, w, t48);
    for (int i = 0; i < w; ++i) memcpy(out + (size_t)i * INC_OUT + 144, t48 + (size_t)i * 48, 48 * sizeof(real));
    /* branch5: relu( BN(1x1 48 of in) + BN(1x1 48 of relu(BN(1x3 64 of relu(BN(1x1 32 of in))))) )
     *                                                                   layers.py:120-138 */
    conv_bn(&M[6], in, w, t48);      /* stem, no ReLU */
    conv_bn(&M[7], in, w, t32);
    conv_bn(&M[8], t32, w, t64);
    conv_bn(&M[9], t64, w, t48b);    /* no ReLU */
    for (int i = 0; i < w; ++i)
        for (int c = 0; c < 48; ++c) {
            real v = t48[(size_t)i * 48 + c] + t48b[(size_t)i * 48 + c];
            out[(size_t)i * INC_OUT + 192 + c] = v > 0 ? v : 0;
        }
}

/* incept_net.__call__, layers.py:181-239. signals [signal_len] -> feat [wc*240] */
static void signal_model(const net_t *N, const float *signals, real *feat, real *buf0, real *buf1,
                         real *tmp, size_t site, ds_oracle_taps *taps)
{
    const int sl = N->signal_len;
    for (int i = 0; i < sl; ++i) buf0[i] = (real)signals[i];          /* model.py:76, NHWC with C=1 */
    int w = conv_bn(&N->stem[0], buf0, sl, buf1);                     /* layers.py:183-188 */
    w = maxpool3(buf1, w, 64, 2, buf0);                               /* layers.py:189-191 */
    if (taps && taps->stem_pool) for (int i = 0; i < w * 64; ++i) taps->stem_pool[site * w * 64 + i] = (float)buf0[i];
    conv_bn(&N->stem[1], buf0, w, buf1);                              /* layers.py:192-197 */
    if (taps && taps->stem_conv2) for (int i = 0; i < w * 128; ++i) taps->stem_conv2[site * w * 128 + i] = (float)buf1[i];
    conv_bn(&N->stem[2], buf1, w, buf0);                              /* layers.py:198-203 */
    if (taps && taps->stem_conv3) for (int i = 0; i < w * 256; ++i) taps->stem_conv3[site * w * 256 + i] = (float)buf0[i];
    real *cur = buf0, *nxt = buf1;
    int cin = 256;
    for (int m = 0; m < NMOD; ++m) {
        inception(N->mod[m], cur, w, cin, nxt, tmp);                  /* layers.py:205-232 */
        if (taps && taps->module_out[m])
            for (int i = 0; i < w * INC_OUT; ++i) taps->module_out[m][site * w * INC_OUT + i] = (float)nxt[i];
        { real *t = cur; cur = nxt; nxt = t; }
        cin = INC_OUT;
        if (m == 2 || m == 7) {                                       /* layers.py:211-213, 224-226 */
            w = maxpool3(cur, w, INC_OUT, 2, nxt);
            real *t = cur; cur = nxt; nxt = t;
        }
    }
    avgpool7(cur, w, INC_OUT, feat);                                  /* layers.py:233-238 (flatten = [w][c]) */
}

/* rnn_layers / Event_model, layers.py:20-72,161-173 with TF-1.x LSTMCell semantics
 * (SURVEY.md Appendix B.1-B.3): z=[x,h]K+b ; i,j,f,o=split(z) ; c'=sig(f+1)c+sig(i)tanh(j) ;
 * h'=sig(o)tanh(c'). MultiRNNCell stacks the 3 cells inside each time step; the backward stack
 * consumes the time-reversed sequence and its outputs are reversed back. out = [fw h(T-1) | bw h(0)]. */
static void event_model(const net_t *N, const int32_t *kmer, const float *means, const float *stds,
                        const float *sanums, real *out512, real *scratch, size_t site,
                        ds_oracle_taps *taps)
{
    const int T = N->kmer_len;
    const int IN0 = N->is_base ? EMB + 3 : 3;    /* model.py:63-75: without is_base only (mean, std, len) */
    real *x0 = scratch;                      /* [T][IN0] */
    real *h = x0 + (size_t)T * (EMB + 3);    /* [3][256] */
    real *c = h + NLAYER * HID;              /* [3][256] */
    real *z = c + NLAYER * HID;              /* [1024]   */
    real *xin = z + 4 * HID;                 /* [512]    */
    for (int t = 0; t < T; ++t) {            /* model.py:61-75 embedding_lookup + concat */
        int o = 0;
        if (N->is_base) {
            const float *e = N->embedding + (size_t)kmer[t] * EMB;
            for (int i = 0; i < EMB; ++i) x0[(size_t)t * IN0 + i] = (real)e[i];
            o = EMB;
        }
        x0[(size_t)t * IN0 + o + 0] = (real)means[t];
        x0[(size_t)t * IN0 + o + 1] = (real)stds[t];
        x0[(size_t)t * IN0 + o + 2] = (real)sanums[t];
    }
    for (int d = 0; d < 2; ++d) {
        for (int i = 0; i < NLAYER * HID; ++i) { h[i] = 0; c[i] = 0; }   /* zero initial state */
        for (int s = 0; s < T; ++s) {
            const int t = d == 0 ? s : T - 1 - s;
            const real *x = x0 + (size_t)t * IN0;
            int xin_n = IN0;
            for (int l = 0; l < NLAYER; ++l) {
                const float *K = N->lstm_kernel[d][l];
                const float *b = N->lstm_bias[d][l];
                real *hl = h + l * HID, *cl = c + l * HID;
                for (int i = 0; i < xin_n; ++i) xin[i] = x[i];
                for (int j = 0; j < 4 * HID; ++j) z[j] = 0;
                for (int i = 0; i < xin_n; ++i) {
                    const real xv = xin[i];
                    const float *kr = K + (size_t)i * 4 * HID;
                    for (int j = 0; j < 4 * HID; ++j) z[j] += xv * (real)kr[j];
                }
                for (int i = 0; i < HID; ++i) {
                    const real hv = hl[i];
                    const float *kr = K + (size_t)(xin_n + i) * 4 * HID;
                    for (int j = 0; j < 4 * HID; ++j) z[j] += hv * (real)kr[j];
                }
                for (int u = 0; u < HID; ++u) {
                    real gi = z[u] + (real)b[u];
                    real gj = z[HID + u] + (real)b[HID + u];
                    real gf = z[2 * HID + u] + (real)b[2 * HID + u];
                    real go = z[3 * HID + u] + (real)b[3 * HID + u];
                    real cn = sigmoid_r(gf + (real)1.0) * cl[u] + sigmoid_r(gi) * R_TANH(gj);
                    cl[u] = cn;
                    hl[u] = sigmoid_r(go) * R_TANH(cn);
                }
                if (taps && taps->lstm_h[d][l])
                    for (int u = 0; u < HID; ++u)
                        taps->lstm_h[d][l][(site * T + t) * HID + u] = (float)hl[u];
                x = hl; xin_n = HID;          /* dropout(keep_prob=1.0) is the identity */
            }
        }
        /* fw: output at original t=T-1 ; bw: output at original t=0 — both are the LAST step run */
        for (int u = 0; u < HID; ++u) out512[d * HID + u] = h[(NLAYER - 1) * HID + u];
    }
}

static int bind_net(net_t *N, int kmer_len, int signal_len, int class_num, int is_cnn, int is_rnn, int is_base,
                    const float *const *t)
{
    memset(N, 0, sizeof(*N));
    N->kmer_len = kmer_len; N->signal_len = signal_len; N->class_num = class_num;
    N->is_cnn = is_cnn; N->is_rnn = is_rnn; N->is_base = is_base;
    int pl, w;
    same_pad(signal_len, 7, 2, &N->w1, &pl);
    same_pad(N->w1, 3, 2, &N->wa, &pl);
    same_pad(N->wa, 3, 2, &N->wb, &pl);
    same_pad(N->wb, 3, 2, &N->wc, &pl);
    (void)w;
    N->joint = (is_rnn ? 2 * HID : 0) + (is_cnn ? N->wc * INC_OUT : 0);     /* layers.py:248-255 */
    int i = 0;
    if (is_rnn) {
        if (is_base) N->embedding = t[i++];
        for (int d = 0; d < 2; ++d)
            for (int l = 0; l < NLAYER; ++l) { N->lstm_kernel[d][l] = t[i++]; N->lstm_bias[d][l] = t[i++]; }
    }
#define BIND(L, K_, CIN, COUT, S, RELU) do { (L).kernel = t[i++]; (L).beta = t[i++]; (L).gamma = t[i++]; \
        (L).mean = t[i++]; (L).var = t[i++]; (L).k = K_; (L).cin = CIN; (L).cout = COUT; (L).stride = S; (L).relu = RELU; } while (0)
    if (is_cnn) {
    BIND(N->stem[0], 7, 1, 64, 2, 1);
    BIND(N->stem[1], 1, 64, 128, 1, 1);
    BIND(N->stem[2], 3, 128, 256, 1, 1);
    }
    for (int m = 0; is_cnn && m < NMOD; ++m) {
        int cin = m == 0 ? 256 : INC_OUT;
        BIND(N->mod[m][0], 1, cin, 48, 1, 1);   /* b1  */
        BIND(N->mod[m][1], 1, cin, 48, 1, 1);   /* b2  */
        BIND(N->mod[m][2], 1, cin, 32, 1, 1);   /* b3a */
        BIND(N->mod[m][3], 3, 32, 48, 1, 1);    /* b3b */
        BIND(N->mod[m][4], 1, cin, 32, 1, 1);   /* b4a */
        BIND(N->mod[m][5], 5, 32, 48, 1, 1);    /* b4b */
        BIND(N->mod[m][6], 1, cin, 48, 1, 0);   /* b5s (no ReLU) */
        BIND(N->mod[m][7], 1, cin, 32, 1, 1);   /* b5a */
        BIND(N->mod[m][8], 3, 32, 64, 1, 1);    /* b5b */
        BIND(N->mod[m][9], 1, 64, 48, 1, 0);    /* b5c (no ReLU) */
    }
#undef BIND
    N->fc1 = t[i++];
    N->fc2 = t[i++];
    return i;
}

int ds_oracle_num_tensors(int is_cnn, int is_rnn, int is_base)
{
    return (is_rnn ? 12 + (is_base ? 1 : 0) : 0) + (is_cnn ? 15 + NMOD * 50 : 0) + 2;
}

int ds_oracle_real_bytes(void) { return (int)sizeof(real); }

/* The whole call_mods forward (model.py:25-108): returns 0, fills act [n,class_num] (sigmoid of the
 * logits, model.py:100) and pred [n] (argmax, ties -> first index, model.py:107-108). */
int ds_oracle_forward(int kmer_len, int signal_len, int class_num, int is_cnn, int is_rnn, int is_base,
                      const float *const *tensors, int n, const int32_t *kmer, const float *means, const float *stds,
                      const float *sanums, const float *signals, float *act, int32_t *pred,
                      ds_oracle_taps *taps, int nthreads)
{
    net_t N;
    if (!(is_cnn || is_rnn)) return -2;               /* model.py:28-29 */
    bind_net(&N, kmer_len, signal_len, class_num, is_cnn, is_rnn, is_base, tensors);
    const int J = N.joint;
    real *joint = (real *)malloc((size_t)n * J * sizeof(real));
    real *fc1 = (real *)calloc((size_t)n * J, sizeof(real));
    if (!joint || !fc1) { free(joint); free(fc1); return -1; }
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
    const size_t act_sz = (size_t)(N.w1 > signal_len ? N.w1 : signal_len) * 256 + 1024;
#pragma omp parallel
    {
        real *buf0 = (real *)malloc(act_sz * sizeof(real));
        real *buf1 = (real *)malloc(act_sz * sizeof(real));
        real *tmp = (real *)malloc((size_t)N.wa * (256 + 48 + 32 + 64 + 48) * sizeof(real));
        real *scr = (real *)malloc(((size_t)kmer_len * (EMB + 3) + 2 * NLAYER * HID + 4 * HID + 2 * HID) * sizeof(real));
#pragma omp for schedule(dynamic, 1)
        for (int s = 0; s < n; ++s) {
            real *jr = joint + (size_t)s * J;          /* joint = [event | signal] (layers.py:248-255) */
            if (is_rnn)
                event_model(&N, kmer + (size_t)s * kmer_len, means + (size_t)s * kmer_len,
                            stds + (size_t)s * kmer_len, sanums + (size_t)s * kmer_len, jr, scr, (size_t)s, taps);
            if (is_cnn)
                signal_model(&N, signals + (size_t)s * signal_len, jr + (is_rnn ? 2 * HID : 0), buf0, buf1, tmp, (size_t)s, taps);
        }
        free(buf0); free(buf1); free(tmp); free(scr);
    }
    /* Joint_model fc1: dense(J, use_bias=False), no activation (layers.py:257-260). Blocked over
     * sites so each W1 row is reused; threads own column ranges. */
    {
        const int CB = 256, SB = 16;
        const int ncb = (J + CB - 1) / CB;
#pragma omp parallel for schedule(dynamic, 1)
        for (int cb = 0; cb < ncb; ++cb) {
            const int j0 = cb * CB, j1 = j0 + CB < J ? j0 + CB : J;
            for (int s0 = 0; s0 < n; s0 += SB) {
                const int s1 = s0 + SB < n ? s0 + SB : n;
                for (int k = 0; k < J; ++k) {
                    const float *wr = N.fc1 + (size_t)k * J;
                    for (int s = s0; s < s1; ++s) {
                        const real xv = joint[(size_t)s * J + k];
                        real *o = fc1 + (size_t)s * J;
                        for (int j = j0; j < j1; ++j) o[j] += xv * (real)wr[j];
                    }
                }
            }
        }
    }
    /* fc2: dense(class_num, use_bias=False) (layers.py:261-263); sigmoid + argmax (model.py:100,108) */
#pragma omp parallel for
    for (int s = 0; s < n; ++s) {
        real best = 0; int bi = 0;
        for (int c = 0; c < class_num; ++c) {
            real acc = 0;
            for (int k = 0; k < J; ++k) acc += fc1[(size_t)s * J + k] * (real)N.fc2[(size_t)k * class_num + c];
            if (taps && taps->logits) taps->logits[(size_t)s * class_num + c] = (float)acc;
            real a = sigmoid_r(acc);
            act[(size_t)s * class_num + c] = (float)a;
            if (c == 0 || a > best) { best = a; bi = c; }
        }
        pred[s] = bi;
    }
    if (taps && taps->joint) for (size_t i = 0; i < (size_t)n * J; ++i) taps->joint[i] = (float)joint[i];
    if (taps && taps->fc1) for (size_t i = 0; i < (size_t)n * J; ++i) taps->fc1[i] = (float)fc1[i];
    if (taps && taps->signal_feat && is_cnn) {
        const int ev = is_rnn ? 2 * HID : 0;
        for (int s = 0; s < n; ++s)
            for (int i = 0; i < J - ev; ++i)
                taps->signal_feat[(size_t)s * (J - ev) + i] = (float)joint[(size_t)s * J + ev + i];
    }
    free(joint); free(fc1);
    return 0;
}
