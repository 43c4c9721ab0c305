/* Plain C99 client of libdeepsignal_hip.so (TEST INFRASTRUCTURE): proves the boundary needs nothing but the public
 * header -- no Python, no torch, no C++. Reads a DSAMDW01 weight file and a raw feature dump, runs the forward through
 * ds_forward (blocking) and through ds_submit / ds_wait, and writes act / pred so the caller can compare bits.
 *
 *   abi_client <weights.dsw> <features.bin> <n> <out.bin> [precision]
 * features.bin = int32 kmer[n][17] | float means[n][17] | float stds[n][17] | float sanums[n][17] | float signals[n][360]
 * out.bin      = float act[n][2] | int32 pred[n] | float act2[n][2] | int32 pred2[n]   (blocking, then submit/wait)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/deepsignal_hip.h"

#define CHECK(call)                                                                      \
    do {                                                                                 \
        int rc_ = (call);                                                                \
        if (rc_ != DS_OK) {                                                              \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, ds_last_error(h));       \
            return 2;                                                                    \
        }                                                                                \
    } while (0)

int main(int argc, char **argv)
{
    if (argc < 5) {
        fprintf(stderr, "usage: %s weights.dsw features.bin n out.bin [precision]\n", argv[0]);
        return 1;
    }
    const int n = atoi(argv[3]);
    const int T = 17, S = 360, C = 2, B = 64;
    ds_handle *h = NULL;
    ds_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.kmer_len = T; cfg.signal_len = S; cfg.class_num = C;
    cfg.is_cnn = cfg.is_rnn = cfg.is_base = 1;
    cfg.device = 0; cfg.max_batch = B;
    cfg.precision = argc > 5 ? atoi(argv[5]) : DS_PRECISION_FP32;
    cfg.reserved[1] = 3;                                    /* three forwards in flight */
    if (ds_create(&cfg, &h) != DS_OK) {
        fprintf(stderr, "ds_create: %s\n", ds_last_error(NULL));
        return 2;
    }
    CHECK(ds_load_weights(h, argv[1]));

    int32_t *kmer = malloc((size_t)n * T * 4), *pred = malloc((size_t)n * 4), *pred2 = malloc((size_t)n * 4);
    float *means = malloc((size_t)n * T * 4), *stds = malloc((size_t)n * T * 4), *sanums = malloc((size_t)n * T * 4);
    float *signals = malloc((size_t)n * S * 4), *act = malloc((size_t)n * C * 4), *act2 = malloc((size_t)n * C * 4);
    FILE *f = fopen(argv[2], "rb");
    if (!f || fread(kmer, 4, (size_t)n * T, f) != (size_t)n * T || fread(means, 4, (size_t)n * T, f) != (size_t)n * T ||
        fread(stds, 4, (size_t)n * T, f) != (size_t)n * T || fread(sanums, 4, (size_t)n * T, f) != (size_t)n * T ||
        fread(signals, 4, (size_t)n * S, f) != (size_t)n * S) {
        fprintf(stderr, "cannot read %s\n", argv[2]);
        return 1;
    }
    fclose(f);

    CHECK(ds_forward(h, n, kmer, means, stds, sanums, signals, act, pred));      /* loops n > max_batch inside */

    /* asynchronous boundary: keep ds_num_slots batches in flight, wait in submission order */
    {
        const int slots = ds_num_slots(h);
        int32_t tickets[8];
        int head = 0, tail = 0, off_wait = 0, off;
        for (off = 0; off < n; off += B) {
            const int m = n - off < B ? n - off : B;
            if (head - tail == slots) {
                const int mw = n - off_wait < B ? n - off_wait : B;
                CHECK(ds_wait(h, tickets[tail % 8], act2 + (size_t)off_wait * C, pred2 + off_wait));
                off_wait += mw; ++tail;
            }
            if ((head & 1) && m >= 3) {       /* every other batch as three row segments through ds_submit_parts: same bits */
                const int32_t cnt[3] = {1, m / 2, m - 1 - m / 2};
                const int32_t o1 = off + 1, o2 = off + 1 + m / 2;
                const int32_t* pk[3] = {kmer + (size_t)off * T, kmer + (size_t)o1 * T, kmer + (size_t)o2 * T};
                const float* pm[3] = {means + (size_t)off * T, means + (size_t)o1 * T, means + (size_t)o2 * T};
                const float* ps[3] = {stds + (size_t)off * T, stds + (size_t)o1 * T, stds + (size_t)o2 * T};
                const float* pl[3] = {sanums + (size_t)off * T, sanums + (size_t)o1 * T, sanums + (size_t)o2 * T};
                const float* pg[3] = {signals + (size_t)off * S, signals + (size_t)o1 * S, signals + (size_t)o2 * S};
                CHECK(ds_submit_parts(h, 3, cnt, pk, pm, ps, pl, pg, &tickets[head % 8]));
            } else {
                CHECK(ds_submit(h, m, kmer + (size_t)off * T, means + (size_t)off * T, stds + (size_t)off * T,
                                sanums + (size_t)off * T, signals + (size_t)off * S, &tickets[head % 8]));
            }
            ++head;
        }
        while (tail < head) {
            const int mw = n - off_wait < B ? n - off_wait : B;
            CHECK(ds_wait(h, tickets[tail % 8], act2 + (size_t)off_wait * C, pred2 + off_wait));
            off_wait += mw; ++tail;
        }
    }

    f = fopen(argv[4], "wb");
    if (!f) return 1;
    fwrite(act, 4, (size_t)n * C, f); fwrite(pred, 4, (size_t)n, f);
    fwrite(act2, 4, (size_t)n * C, f); fwrite(pred2, 4, (size_t)n, f);
    fclose(f);
    printf("%s: %d sites, act[0] = %.9g %.9g\n", ds_version(), n, act[0], act[1]);
    ds_destroy(h);
    return 0;
}
