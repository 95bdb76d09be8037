/*
 * abi_client.c -- a plain C client of libdce.so: what a non-Python host (the InEKF side of the
 * reference's pipeline, README.md:65) would write against include/dce.h.  Test infrastructure:
 * it links the CPU oracle (oracle/dce_oracle.c) as the checker.
 *
 *   1. error behaviour of the ABI (order of calls, bad keys, bad shapes) -- integer codes, no aborts
 *   2. load 14 state_dict tensors, finalize, dce_infer_sequence on host buffers
 *   3. logits within |d| <= 1e-5*max|ref| + 1e-4*|ref| of the oracle; argmax / contact bits equal
 *      wherever the oracle's top-2 margin exceeds 1e-3*max|logit|
 *   4. dce_online_push row by row == dce_infer_sequence, bit for bit
 *   5. dce_forward_windows on the oracle's z-scored windows == the fused sequence path (tolerance)
 *   6. dce_infer_sequence_packed rows == logits + contact bits, dce_unpack_results inverts them
 *   7. dce_comm_* / dce_gather_results / dce_allreduce_counts in an RCCL world of one rank
 *
 * Build (tests/test_c_client.py does this):
 *   gcc -O2 -std=c11 -fopenmp -Iinclude -Ioracle tests/c/abi_client.c oracle/dce_oracle.c \
 *       -Ldeep_contact_estimator_amd -ldce -L/opt/rocm/lib -lamdhip64 -lm -o abi_client
 * Exit code 0 and a final line "abi_client: OK" on success.
 */
#define _POSIX_C_SOURCE 199309L
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "dce.h"
#include "dce_oracle.h"

/* the client's own device buffers for section 7: four runtime calls, declared here so that the file stays plain C
 * (hip_runtime_api.h wants a platform macro); libamdhip64 is on the link line */
typedef int hipError_t;
#define hipSuccess 0
#define hipMemcpyHostToDevice 1
#define hipMemcpyDeviceToHost 2
hipError_t hipMalloc(void** p, size_t n);
hipError_t hipFree(void* p);
hipError_t hipMemcpy(void* dst, const void* src, size_t n, int kind);
hipError_t hipMemset(void* p, int v, size_t n);

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static double uniform01(void)
{   /* xorshift64* */
    rng_state ^= rng_state >> 12; rng_state ^= rng_state << 25; rng_state ^= rng_state >> 27;
    return (double)((rng_state * 0x2545F4914F6CDD1Dull) >> 11) / 9007199254740992.0;
}

static const struct { const char* key; int ndim; int64_t shape[3]; } KEYS[14] = {
    {"block1.0.weight", 3, {64, 54, 3}},   {"block1.0.bias", 1, {64, 0, 0}},
    {"block1.2.weight", 3, {64, 64, 3}},   {"block1.2.bias", 1, {64, 0, 0}},
    {"block2.0.weight", 3, {128, 64, 3}},  {"block2.0.bias", 1, {128, 0, 0}},
    {"block2.2.weight", 3, {128, 128, 3}}, {"block2.2.bias", 1, {128, 0, 0}},
    {"fc.0.weight", 2, {2048, 4736, 0}},   {"fc.0.bias", 1, {2048, 0, 0}},
    {"fc.3.weight", 2, {512, 2048, 0}},    {"fc.3.bias", 1, {512, 0, 0}},
    {"fc.6.weight", 2, {16, 512, 0}},      {"fc.6.bias", 1, {16, 0, 0}},
};

#define CHECK(cond, ...) do { if (!(cond)) { fprintf(stderr, "abi_client: FAILED %s:%d: ", __FILE__, __LINE__); \
    fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); return 1; } } while (0)

int main(void)
{
    enum { N = 70, T = N + DCE_WINDOW - 1 };
    float* w[14];
    for (int k = 0; k < 14; ++k) {
        int64_t count = 1, fan_in = 1;
        for (int d = 0; d < KEYS[k].ndim; ++d) count *= KEYS[k].shape[d];
        for (int d = 1; d < KEYS[k].ndim; ++d) fan_in *= KEYS[k].shape[d];
        w[k] = (float*)malloc(sizeof(float) * (size_t)count);
        /* weights: uniform with the He variance 2/fan_in; biases: U(-0.1, 0.1) */
        const double a = KEYS[k].ndim > 1 ? sqrt(6.0 / (double)fan_in) : 0.1;
        for (int64_t e = 0; e < count; ++e) w[k][e] = (float)((2.0 * uniform01() - 1.0) * a);
    }
    /* a drifting, differently scaled signal per channel (so that the z-score matters) */
    float* seq = (float*)malloc(sizeof(float) * T * DCE_CHANNELS);
    for (int c = 0; c < DCE_CHANNELS; ++c) {
        const double scale = pow(10.0, 3.0 * uniform01() - 2.0), offset = 10.0 * uniform01() - 5.0;
        double x = 0.0;
        for (int t = 0; t < T; ++t) {
            x = 0.9 * x + (2.0 * uniform01() - 1.0);
            seq[t * DCE_CHANNELS + c] = (float)(offset + scale * x);
        }
    }

    /* ---- 1. error behaviour */
    CHECK(dce_abi_version() >= 1, "abi version");
    CHECK(dce_device_count() >= 1, "no HIP device: %s", dce_last_error(NULL));
    dce_ctx* ctx = NULL;
    CHECK(dce_create(&ctx, 0, 0) == DCE_ERR_ARG, "max_batch 0 must be rejected");
    CHECK(dce_create(&ctx, 0, 32) == DCE_OK, "dce_create: %s", dce_last_error(NULL));
    float lg1[DCE_CLASSES];
    CHECK(dce_infer_sequence(ctx, seq, T, DCE_WINDOW, 0, lg1, NULL, NULL) == DCE_ERR_STATE, "forward before finalize");
    CHECK(dce_finalize_weights(ctx, DCE_FP32) == DCE_ERR_STATE, "finalize with missing keys");
    CHECK(dce_load_weight(ctx, "fc.9.weight", w[0], KEYS[0].shape, 3) == DCE_ERR_KEY, "unknown key");
    { const int64_t bad[3] = {64, 54, 5};
      CHECK(dce_load_weight(ctx, "block1.0.weight", w[0], bad, 3) == DCE_ERR_ARG, "wrong shape"); }
    for (int k = 0; k < 14; ++k)
        CHECK(dce_load_weight(ctx, KEYS[k].key, w[k], KEYS[k].shape, KEYS[k].ndim) == DCE_OK, "%s: %s", KEYS[k].key, dce_last_error(ctx));
    CHECK(dce_finalize_weights(ctx, DCE_FP32) == DCE_OK, "finalize: %s", dce_last_error(ctx));
    CHECK(dce_infer_sequence(ctx, seq, T, 100, 0, lg1, NULL, NULL) == DCE_ERR_ARG, "window != 150");
    lg1[0] = 42.f;   /* T < window: contact_dataset.__len__ <= 0 -> nothing to do, nothing written */
    CHECK(dce_infer_sequence(ctx, seq, DCE_WINDOW - 1, DCE_WINDOW, 0, lg1, NULL, NULL) == DCE_OK && lg1[0] == 42.f, "T < window");
    CHECK(dce_infer_sequence(ctx, NULL, T, DCE_WINDOW, 0, lg1, NULL, NULL) == DCE_ERR_ARG, "NULL sequence");
    CHECK(strlen(dce_last_error(ctx)) > 0, "error message missing");

    /* ---- 2. the path (max_batch 32 < N: chunked inside the library) */
    static float logits[N * DCE_CLASSES], ref_logits[N * DCE_CLASSES];
    static int32_t pred[N], ref_pred[N];
    static uint8_t contacts[N * 4], ref_contacts[N * 4];
    CHECK(dce_infer_sequence(ctx, seq, T, DCE_WINDOW, 0, logits, pred, contacts) == DCE_OK, "infer: %s", dce_last_error(ctx));
    const oracle_weights ow = { w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7], w[8], w[9], w[10], w[11], w[12], w[13] };
    float* zwin = (float*)malloc(sizeof(float) * N * DCE_WINDOW * DCE_CHANNELS);
    CHECK(oracle_infer_sequence(&ow, seq, T, zwin, ref_logits, ref_pred, ref_contacts) == 0, "oracle");

    /* ---- 3. tolerance + argmax contract */
    double maxref = 0.0, worst = 0.0;
    for (int e = 0; e < N * DCE_CLASSES; ++e) maxref = fmax(maxref, fabs(ref_logits[e]));
    for (int e = 0; e < N * DCE_CLASSES; ++e) {
        const double bound = 1e-5 * maxref + 1e-4 * fabs(ref_logits[e]);
        worst = fmax(worst, fabs((double)logits[e] - ref_logits[e]) / bound);
    }
    CHECK(worst <= 1.0, "logits outside tolerance: err/bound = %.3f", worst);
    int distinct[DCE_CLASSES] = {0}, classes = 0;
    for (int i = 0; i < N; ++i) {
        float top = -INFINITY, second = -INFINITY;
        for (int k = 0; k < DCE_CLASSES; ++k) {
            const float v = ref_logits[i * DCE_CLASSES + k];
            if (v > top) { second = top; top = v; } else if (v > second) second = v;
        }
        if (top - second > 1e-3 * maxref) CHECK(pred[i] == ref_pred[i], "argmax of window %d: %d vs %d", i, pred[i], ref_pred[i]);
        CHECK(pred[i] == oracle_argmax16(logits + i * DCE_CLASSES), "pred is not the argmax of the returned logits (window %d)", i);
        uint8_t bits[4];
        oracle_decimal2binary(pred[i], bits);
        CHECK(memcmp(bits, contacts + 4 * i, 4) == 0, "contact bits of window %d", i);
        if (!distinct[pred[i]]++) ++classes;
    }

    /* ---- 4. online mode reproduces the sequence call bit for bit */
    CHECK(dce_online_reset(ctx) == DCE_OK, "online reset");
    for (int t = 0; t < T; ++t) {
        float lg[DCE_CLASSES]; int32_t p; uint8_t cb[4];
        const int r = dce_online_push(ctx, seq + t * DCE_CHANNELS, lg, &p, cb);
        CHECK(r == (t >= DCE_WINDOW - 1 ? 1 : 0), "online push %d returned %d: %s", t, r, dce_last_error(ctx));
        if (r == 1) {
            const int j = t - (DCE_WINDOW - 1);
            CHECK(memcmp(lg, logits + j * DCE_CLASSES, sizeof lg) == 0 && p == pred[j] && memcmp(cb, contacts + 4 * j, 4) == 0,
                  "online row %d differs from dce_infer_sequence", j);
        }
    }

    {   /* latency of the online path from C (informative) */
        for (int round = 0; round < 3; ++round) {
            struct timespec t0, t1;
            clock_gettime(CLOCK_MONOTONIC, &t0);
            for (int rep = 0; rep < 1000; ++rep) {
                float lg[DCE_CLASSES]; int32_t p; uint8_t cb[4];
                CHECK(dce_online_push(ctx, seq + (rep % T) * DCE_CHANNELS, lg, &p, cb) == 1, "online push (timing)");
            }
            clock_gettime(CLOCK_MONOTONIC, &t1);
            printf("abi_client: dce_online_push %.1f us per sample (1000 pushes, sample in -> estimate out)\n",
                   ((t1.tv_sec - t0.tv_sec) * 1e9 + (t1.tv_nsec - t0.tv_nsec)) / 1000.0 / 1e3);
        }
    }

    /* ---- 4b. the LATENCY MODE from plain C (dce_create_ex "latency=1"; csrc/latency.hip): one-window calls as one kernel, pushes through the
     *          resident service kernel and its mailbox -- every estimate within the fp32 contract of the CPU restatement and argmax-exact
     *          outside the noise margin; then what a robot's loop sees per push */
    {
        dce_ctx *lat = NULL, *none = NULL;
        CHECK(dce_create_ex(&none, 0, 32, "no_such_option=1") == DCE_ERR_ARG && none == NULL && strstr(dce_last_error(NULL), "no_such_option"), "an unknown option must fail the creation");
        CHECK(dce_create_ex(&lat, 0, 32, "latency=1,latency_idle_ms=50") == DCE_OK, "dce_create_ex: %s", dce_last_error(NULL));
        for (int k = 0; k < 14; ++k) CHECK(dce_load_weight(lat, KEYS[k].key, w[k], KEYS[k].shape, KEYS[k].ndim) == DCE_OK, "%s", dce_last_error(lat));
        CHECK(dce_finalize_weights(lat, DCE_FP32) == DCE_OK, "finalize (latency ctx): %s", dce_last_error(lat));
        double worst_l = 0.0;
        for (int t = 0; t < T; ++t) {
            float lg[DCE_CLASSES]; int32_t p; uint8_t cb[4];
            if (t == 180) { struct timespec nap = {0, 120 * 1000 * 1000}; nanosleep(&nap, NULL); }      /* the service leaves (50 ms idle) and comes back */
            const int r = dce_online_push(lat, seq + t * DCE_CHANNELS, lg, &p, cb);
            CHECK(r == (t >= DCE_WINDOW - 1 ? 1 : 0), "latency push %d returned %d: %s", t, r, dce_last_error(lat));
            if (r != 1) continue;
            const int j = t - (DCE_WINDOW - 1);
            float top = -INFINITY, second = -INFINITY;
            for (int k = 0; k < DCE_CLASSES; ++k) {
                const float v = ref_logits[j * DCE_CLASSES + k];
                worst_l = fmax(worst_l, fabs((double)lg[k] - v) / (1e-5 * maxref + 1e-4 * fabs(v)));
                if (v > top) { second = top; top = v; } else if (v > second) second = v;
            }
            if (top - second > 1e-3 * maxref) CHECK(p == ref_pred[j], "latency mode: argmax of window %d: %d vs %d", j, p, ref_pred[j]);
            uint8_t bits[4];
            oracle_decimal2binary(p, bits);
            CHECK(p == oracle_argmax16(lg) && memcmp(bits, cb, 4) == 0, "latency mode: pred / contact bits of window %d", j);
        }
        CHECK(worst_l <= 1.0, "latency mode: logits outside tolerance: err/bound = %.3f", worst_l);
        float one[DCE_CLASSES]; int32_t p1; uint8_t c1[4];                   /* a one-window call (the service leaves first) */
        CHECK(dce_forward_windows(lat, zwin, 1, 0, one, &p1, c1) == DCE_OK, "latency one-shot: %s", dce_last_error(lat));
        for (int k = 0; k < DCE_CLASSES; ++k)
            CHECK(fabs((double)one[k] - ref_logits[k]) <= 1e-5 * maxref + 1e-4 * fabs(ref_logits[k]), "latency one-shot logit %d", k);
        for (int round = 0; round < 3; ++round) {
            struct timespec t0, t1;
            clock_gettime(CLOCK_MONOTONIC, &t0);
            for (int rep = 0; rep < 1000; ++rep) {
                float lg[DCE_CLASSES]; int32_t p; uint8_t cb[4];
                CHECK(dce_online_push(lat, seq + (rep % T) * DCE_CHANNELS, lg, &p, cb) == 1, "latency push (timing): %s", dce_last_error(lat));
            }
            clock_gettime(CLOCK_MONOTONIC, &t1);
            printf("abi_client: latency mode: dce_online_push %.1f us per sample (1000 pushes, sample in -> estimate out; worst err/bound %.3f)\n",
                   ((t1.tv_sec - t0.tv_sec) * 1e9 + (t1.tv_nsec - t0.tv_nsec)) / 1000.0 / 1e3, worst_l);
        }
        dce_destroy(lat);
    }

    /* ---- 5. materialised windows through dce_forward_windows */
    static float logits_w[N * DCE_CLASSES];
    CHECK(dce_forward_windows(ctx, zwin, N, 0, logits_w, NULL, NULL) == DCE_OK, "forward_windows: %s", dce_last_error(ctx));
    worst = 0.0;
    for (int e = 0; e < N * DCE_CLASSES; ++e)
        worst = fmax(worst, fabs((double)logits_w[e] - ref_logits[e]) / (1e-5 * maxref + 1e-4 * fabs(ref_logits[e])));
    CHECK(worst <= 1.0, "forward_windows outside tolerance: err/bound = %.3f", worst);

    /* ---- 6. packed rows (16 fp32 logits + 4 contact bits) == the three arrays; unpack on the host */
    static uint8_t packed[N * DCE_PACKED_ROW];
    CHECK(dce_infer_sequence_packed(ctx, seq, T, DCE_WINDOW, 0, packed) == DCE_OK, "packed: %s", dce_last_error(ctx));
    for (int i = 0; i < N; ++i)
        CHECK(memcmp(packed + i * DCE_PACKED_ROW, logits + i * DCE_CLASSES, 64) == 0 &&
              memcmp(packed + i * DCE_PACKED_ROW + 64, contacts + 4 * i, 4) == 0, "packed row %d", i);
    {
        static float ul[N * DCE_CLASSES]; static int32_t up[N]; static uint8_t uc[N * 4];
        CHECK(dce_unpack_results(NULL, packed, N, 0, ul, up, uc) == DCE_OK, "host unpack");
        CHECK(memcmp(ul, logits, sizeof ul) == 0 && memcmp(up, pred, sizeof up) == 0 && memcmp(uc, contacts, sizeof uc) == 0, "unpack");
    }

    /* ---- 7. the multi-GPU exchange in the only form one GPU allows: an RCCL world of one rank, bootstrapped exactly
     *         as N processes would (unique id from rank 0 -> dce_comm_init on every rank -> dce_gather_results) */
    {
        uint8_t id[DCE_COMM_ID_BYTES];
        CHECK(dce_gather_results(ctx, NULL, 0, NULL, NULL, 0, 0) == DCE_ERR_STATE, "gather before comm_init");
        CHECK(dce_comm_get_unique_id(id) == DCE_OK, "unique id: %s", dce_last_error(NULL));
        CHECK(dce_comm_init(ctx, 1, 1, id) == DCE_ERR_ARG, "rank out of range");
        CHECK(dce_comm_init(ctx, 0, 1, id) == DCE_OK, "comm_init: %s", dce_last_error(ctx));
        int rk = -1, wd = -1, ver = 0; char libname[256];
        CHECK(dce_comm_info(ctx, &rk, &wd, &ver, libname, sizeof libname) == DCE_OK && rk == 0 && wd == 1 && ver > 0, "comm_info");
        void *d_seq = NULL, *d_local = NULL, *d_all = NULL;
        CHECK(hipMalloc(&d_seq, sizeof(float) * T * DCE_CHANNELS) == hipSuccess && hipMalloc(&d_local, N * DCE_PACKED_ROW) == hipSuccess &&
              hipMalloc(&d_all, N * DCE_PACKED_ROW) == hipSuccess, "hipMalloc");
        CHECK(hipMemcpy(d_seq, seq, sizeof(float) * T * DCE_CHANNELS, hipMemcpyHostToDevice) == hipSuccess, "H2D");
        CHECK(dce_infer_sequence_packed(ctx, (const float*)d_seq, T, DCE_WINDOW, 1, (uint8_t*)d_local) == DCE_OK, "packed on device");
        const int64_t rows[1] = {N};
        const int64_t wrong[1] = {N - 1};
        CHECK(dce_gather_results(ctx, (const uint8_t*)d_local, N, (uint8_t*)d_all, wrong, 0, 0) == DCE_ERR_ARG, "row-count mismatch");
        for (int mode = 0; mode < 3; ++mode) {      /* uniform (ncclGather), explicit sizes, asynchronous */
            CHECK(hipMemset(d_all, 0, N * DCE_PACKED_ROW) == hipSuccess, "memset");
            CHECK(dce_gather_results(ctx, (const uint8_t*)d_local, N, (uint8_t*)d_all, mode == 1 ? rows : NULL, 0, mode == 2) == DCE_OK,
                  "gather mode %d: %s", mode, dce_last_error(ctx));
            CHECK(dce_comm_sync(ctx) == DCE_OK && dce_sync(ctx) == DCE_OK, "comm sync");
            static uint8_t back[N * DCE_PACKED_ROW];
            CHECK(hipMemcpy(back, d_all, sizeof back, hipMemcpyDeviceToHost) == hipSuccess, "D2H");
            CHECK(memcmp(back, packed, sizeof back) == 0, "gathered rows differ (mode %d)", mode);
        }
        int64_t counts[256];
        for (int k = 0; k < 256; ++k) counts[k] = k;
        CHECK(dce_allreduce_counts(ctx, counts, 0) == DCE_OK, "allreduce: %s", dce_last_error(ctx));
        for (int k = 0; k < 256; ++k) CHECK(counts[k] == k, "allreduce over one rank changed count %d", k);
        CHECK(dce_comm_destroy(ctx) == DCE_OK, "comm destroy");
        hipFree(d_seq); hipFree(d_local); hipFree(d_all);
        printf("abi_client: RCCL %d (%s), world of 1: ncclGather / grouped send-recv / async gather / all-reduce OK\n", ver, libname);
    }

    /* ---- 8. DCE_FP32_SPLIT (fc.0 on the bf16 matrix pipe, fp32 operands as three bf16 terms): a batch large enough for
     *         the split kernel, held to the same tolerance against the same oracle as the fp32 path */
    {
        enum { N2 = 3072, T2 = N2 + DCE_WINDOW - 1 };
        dce_ctx* c2 = NULL;
        CHECK(dce_create(&c2, 0, N2) == DCE_OK, "dce_create (split): %s", dce_last_error(NULL));
        for (int k = 0; k < 14; ++k)
            CHECK(dce_load_weight(c2, KEYS[k].key, w[k], KEYS[k].shape, KEYS[k].ndim) == DCE_OK, "%s", KEYS[k].key);
        CHECK(dce_finalize_weights(c2, 7) == DCE_ERR_ARG, "unknown precision must be rejected");
        CHECK(dce_finalize_weights(c2, DCE_FP32_SPLIT) == DCE_OK, "finalize (split): %s", dce_last_error(c2));
        float* seq2 = (float*)malloc(sizeof(float) * T2 * DCE_CHANNELS);
        for (int c = 0; c < DCE_CHANNELS; ++c) {
            const double scale = pow(10.0, 3.0 * uniform01() - 2.0), offset = 10.0 * uniform01() - 5.0;
            double x = 0.0;
            for (int t = 0; t < T2; ++t) { x = 0.9 * x + (2.0 * uniform01() - 1.0); seq2[t * DCE_CHANNELS + c] = (float)(offset + scale * x); }
        }
        float* lg2 = (float*)malloc(sizeof(float) * N2 * DCE_CLASSES);
        float* rl2 = (float*)malloc(sizeof(float) * N2 * DCE_CLASSES);
        int32_t* pr2 = (int32_t*)malloc(sizeof(int32_t) * N2); int32_t* rp2 = (int32_t*)malloc(sizeof(int32_t) * N2);
        uint8_t* rc2 = (uint8_t*)malloc(4 * N2);
        float* zw2 = (float*)malloc(sizeof(float) * (size_t)N2 * DCE_WINDOW * DCE_CHANNELS);
        CHECK(dce_infer_sequence(c2, seq2, T2, DCE_WINDOW, 0, lg2, pr2, NULL) == DCE_OK, "infer (split): %s", dce_last_error(c2));
        char plan[256];
        /* (round 6: the product library runs DCE_FP32_F16X2 for this precision and says so in the plan; the three-term kernels are the experiments build's) */
        CHECK(dce_last_plan(c2, plan, sizeof plan) == DCE_OK && (strstr(plan, "fc_x3_256x128") != NULL || ((dce_build_flags() & DCE_BUILD_EXPERIMENTS) == 0 && strstr(plan, "fp32_split_is_fp32_f16x2") != NULL && strstr(plan, "fc_h2_256x128") != NULL)),
              "neither the split kernels nor the alias note in the plan: %s", plan);
        CHECK(oracle_infer_sequence(&ow, seq2, T2, zw2, rl2, rp2, rc2) == 0, "oracle (split)");
        double mr = 0.0, wst = 0.0;
        for (int e = 0; e < N2 * DCE_CLASSES; ++e) mr = fmax(mr, fabs(rl2[e]));
        for (int e = 0; e < N2 * DCE_CLASSES; ++e)
            wst = fmax(wst, fabs((double)lg2[e] - rl2[e]) / (1e-5 * mr + 1e-4 * fabs(rl2[e])));
        CHECK(wst <= 1.0, "split mode: logits outside tolerance: err/bound = %.3f", wst);
        int flips = 0;
        for (int i = 0; i < N2; ++i) flips += pr2[i] != rp2[i];
        CHECK(flips <= 6, "split mode: %d argmax differences of %d", flips, N2);
        printf("abi_client: DCE_FP32_SPLIT, %d windows: err/bound %.3f, %d sub-margin argmax differences, plan %s\n", N2, wst, flips, plan);
        /* ---- 8b. the same context finalised again with DCE_FP32_F16X2 (two fp16 terms per operand, per-window scales chosen in the kernel;
         *          csrc/conv_h2.hip, csrc/fc_gemm_h2.hip): same input, same oracle rows, same tolerance; the scaled sequence (x 2^-60) too --
         *          the z-score makes it the same windows, so the same bits must come back */
        CHECK(dce_finalize_weights(c2, DCE_FP32_F16X2) == DCE_OK, "finalize (f16x2): %s", dce_last_error(c2));
        CHECK(dce_infer_sequence(c2, seq2, T2, DCE_WINDOW, 0, lg2, pr2, NULL) == DCE_OK, "infer (f16x2): %s", dce_last_error(c2));
        CHECK(dce_last_plan(c2, plan, sizeof plan) == DCE_OK && strstr(plan, "conv_h2") != NULL && strstr(plan, "fc_h2_256x128") != NULL && strstr(plan, "fc23_fused_h2_128x64") != NULL,
              "two-term fp16 kernels not in the plan: %s", plan);
        wst = 0.0;
        for (int e = 0; e < N2 * DCE_CLASSES; ++e)
            wst = fmax(wst, fabs((double)lg2[e] - rl2[e]) / (1e-5 * mr + 1e-4 * fabs(rl2[e])));
        CHECK(wst <= 1.0, "f16x2 mode: logits outside tolerance: err/bound = %.3f", wst);
        flips = 0;
        for (int i = 0; i < N2; ++i) flips += pr2[i] != rp2[i];
        CHECK(flips <= 6, "f16x2 mode: %d argmax differences of %d", flips, N2);
        printf("abi_client: DCE_FP32_F16X2, %d windows: err/bound %.3f, %d sub-margin argmax differences, plan %s\n", N2, wst, flips, plan);
        dce_destroy(c2);
        free(seq2); free(lg2); free(rl2); free(pr2); free(rp2); free(rc2); free(zw2);
    }

    CHECK(dce_sync(ctx) == DCE_OK, "sync");
    dce_destroy(ctx);
    dce_destroy(NULL);
    printf("abi_client: OK (%d windows, %d distinct classes, max|logit| %.2f)\n", N, classes, maxref);
    for (int k = 0; k < 14; ++k) free(w[k]);
    free(seq); free(zwin);
    return 0;
}
