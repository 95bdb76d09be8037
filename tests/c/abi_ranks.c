/*
 * abi_ranks.c -- the node-of-GPUs flow of SURVEY.md 8(e) with NO torch and NO MPI in the process: one process per GPU, the
 * ncclUniqueId of rank 0 carried to the others through a file, libdce.so's own communicator, one gather of the packed rows
 * (reference src/inference_one_seq.py:137-156 is single-device; this is the boundary a C / C++ host would use for 8 GPUs).
 *
 *   abi_ranks --rank R --world W --id-file F [--device D] [--windows N] [--nonce S] [--dry-run] [--delay-ms MS]
 *
 * Rendezvous: rank 0 removes a left-over F, draws the id and writes <16-byte nonce><128-byte id> to F.tmp, renames it to F;
 * the other ranks poll F and accept it only if its nonce is theirs (a file left by an earlier job is ignored, not trusted);
 * rank 0 removes F once dce_comm_init has returned (every rank has joined by then).  The nonce comes from --nonce or
 * DCE_COMM_NONCE -- tools/launch_ranks.sh draws one per job.  --dry-run stops after the rendezvous (no GPU, no RCCL: rank 0
 * writes a pattern instead of an id, the others check it) -- the part that can be tested on a CPU-only box.
 * Then every rank runs its shard -- windows [R N, (R + 1) N) of one seeded sequence, i.e. rows [R N, (R + 1) N + 149) -- with
 * dce_infer_sequence_packed, the root gathers with dce_gather_results, recomputes all W N windows itself and compares
 * byte for byte; dce_allreduce_counts sums a count per rank.  Exit code 0 and "abi_ranks[R]: OK" on success.
 *
 * Build: gcc -O2 -std=c11 -Iinclude tests/c/abi_ranks.c -Ldeep_contact_estimator_amd -ldce -L/opt/rocm/lib -lamdhip64 -lm -o abi_ranks
 */
#define _POSIX_C_SOURCE 200809L
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include "dce.h"

typedef int hipError_t;
#define hipSuccess 0
#define hipMemcpyHostToDevice 1
#define hipMemcpyDeviceToHost 2
hipError_t hipSetDevice(int d);
hipError_t hipMalloc(void** p, size_t n);
hipError_t hipFree(void* p);
hipError_t hipMemcpy(void* dst, const void* src, size_t n, int kind);

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

static int g_rank = -1;
#define CHECK(cond, ...) do { if (!(cond)) { fprintf(stderr, "abi_ranks[%d]: FAILED %s:%d: ", g_rank, __FILE__, __LINE__); \
    fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); return 1; } } while (0)

enum { NONCE = 16 };

static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

/* rank 0: publish `id` under `nonce`; others: wait (at most `timeout_s`) for a file with OUR nonce.  0 = ok */
static int rendezvous(const char* path, int rank, const uint8_t nonce[NONCE], uint8_t id[DCE_COMM_ID_BYTES], double timeout_s)
{
    uint8_t buf[NONCE + DCE_COMM_ID_BYTES];
    if (rank == 0) {
        char tmp[4096];
        snprintf(tmp, sizeof tmp, "%s.tmp.%ld", path, (long)getpid());
        remove(path);                                         /* a left-over of an earlier job */
        memcpy(buf, nonce, NONCE); memcpy(buf + NONCE, id, DCE_COMM_ID_BYTES);
        FILE* f = fopen(tmp, "wb");
        if (!f || fwrite(buf, 1, sizeof buf, f) != sizeof buf || fclose(f) != 0) return 1;
        return rename(tmp, path) != 0;
    }
    const double t0 = now_s();
    while (now_s() - t0 < timeout_s) {
        FILE* f = fopen(path, "rb");
        if (f) {
            const size_t got = fread(buf, 1, sizeof buf, f);
            fclose(f);
            if (got == sizeof buf && memcmp(buf, nonce, NONCE) == 0) { memcpy(id, buf + NONCE, DCE_COMM_ID_BYTES); return 0; }
        }   /* absent, half-written (never: rename is atomic) or another job's: keep waiting */
        struct timespec nap = {0, 10 * 1000 * 1000};
        nanosleep(&nap, NULL);
    }
    return 2;
}

int main(int argc, char** argv)
{
    int rank = -1, world = -1, device = -1, dry = 0, delay_ms = 0;
    long nwin = 1000;
    const char* path = NULL;
    const char* nonce_s = getenv("DCE_COMM_NONCE");
    for (int a = 1; a < argc; ++a) {
        if (!strcmp(argv[a], "--rank") && a + 1 < argc) rank = atoi(argv[++a]);
        else if (!strcmp(argv[a], "--world") && a + 1 < argc) world = atoi(argv[++a]);
        else if (!strcmp(argv[a], "--device") && a + 1 < argc) device = atoi(argv[++a]);
        else if (!strcmp(argv[a], "--windows") && a + 1 < argc) nwin = atol(argv[++a]);
        else if (!strcmp(argv[a], "--id-file") && a + 1 < argc) path = argv[++a];
        else if (!strcmp(argv[a], "--nonce") && a + 1 < argc) nonce_s = argv[++a];
        else if (!strcmp(argv[a], "--dry-run")) dry = 1;
        else if (!strcmp(argv[a], "--delay-ms") && a + 1 < argc) delay_ms = atoi(argv[++a]);    /* a rank that comes late (rehearsals) */
        else { fprintf(stderr, "usage: abi_ranks --rank R --world W --id-file F [--device D] [--windows N] [--nonce S] [--dry-run]\n"); return 64; }
    }
    g_rank = rank;
    if (delay_ms > 0) { struct timespec nap = {delay_ms / 1000, (long)(delay_ms % 1000) * 1000000L}; nanosleep(&nap, NULL); }
    CHECK(rank >= 0 && world >= 1 && rank < world && path && nwin >= 1, "need --rank R --world W (R < W) --id-file F");
    uint8_t nonce[NONCE] = {0};
    if (nonce_s) { const size_t l = strlen(nonce_s); memcpy(nonce, nonce_s, l < NONCE ? l : NONCE); }
    const double timeout_s = getenv("DCE_COMM_TIMEOUT") ? atof(getenv("DCE_COMM_TIMEOUT")) : 120.0;
    uint8_t id[DCE_COMM_ID_BYTES];

    if (dry) {
        if (rank == 0) for (int k = 0; k < DCE_COMM_ID_BYTES; ++k) id[k] = (uint8_t)(3 * k + 1);
        const int rc = rendezvous(path, rank, nonce, id, timeout_s);
        CHECK(rc == 0, "rendezvous failed (%d): %s", rc, rc == 2 ? "no id with this job's nonce appeared" : "cannot write the id file");
        for (int k = 0; k < DCE_COMM_ID_BYTES; ++k) CHECK(id[k] == (uint8_t)(3 * k + 1), "id byte %d differs", k);
        {   /* the partition every rank derives by itself: contiguous ranges, sizes differ by at most one (deep_contact_estimator_amd/distributed.py shard_range) */
            const long q = nwin / world, rem = nwin % world;
            const long lo = rank * q + (rank < rem ? rank : rem), hi = lo + q + (rank < rem ? 1 : 0);
            printf("abi_ranks[%d]: shard [%ld, %ld) of %ld windows\n", rank, lo, hi, nwin);
        }
        printf("abi_ranks[%d]: OK (dry run: rendezvous of %d ranks through %s)\n", rank, world, path);
        return 0;
    }

    const int ndev = dce_device_count();
    CHECK(ndev >= 1, "no HIP device: %s", dce_last_error(NULL));
    if (device < 0) device = rank % ndev;
    CHECK(device < ndev, "--device %d: %d device(s) visible", device, ndev);
    CHECK(hipSetDevice(device) == hipSuccess, "hipSetDevice(%d)", device);

    /* the same checkpoint and the same sequence on every rank (seeded) */
    float* w[14];
    for (int k = 0; k < 14; ++k) {
        int64_t count = 1, fan_in = 1;
        for (int d = 0; d < KEYS[k].ndim; ++d) count *= KEYS[k].shape[d];
        for (int d = 1; d < KEYS[k].ndim; ++d) fan_in *= KEYS[k].shape[d];
        w[k] = (float*)malloc(sizeof(float) * (size_t)count);
        const double a = KEYS[k].ndim > 1 ? sqrt(6.0 / (double)fan_in) : 0.1;
        for (int64_t e = 0; e < count; ++e) w[k][e] = (float)((2.0 * uniform01() - 1.0) * a);
    }
    const int64_t N = nwin, total = N * world, T = total + DCE_WINDOW - 1;
    float* seq = (float*)malloc(sizeof(float) * (size_t)T * DCE_CHANNELS);
    for (int c = 0; c < DCE_CHANNELS; ++c) {
        const double scale = pow(10.0, 3.0 * uniform01() - 2.0), offset = 10.0 * uniform01() - 5.0;
        double x = 0.0;
        for (int64_t t = 0; t < T; ++t) { x = 0.9 * x + (2.0 * uniform01() - 1.0); seq[t * DCE_CHANNELS + c] = (float)(offset + scale * x); }
    }
    dce_ctx* ctx = NULL;
    CHECK(dce_create(&ctx, device, N) == DCE_OK, "dce_create: %s", dce_last_error(NULL));
    for (int k = 0; k < 14; ++k)
        CHECK(dce_load_weight(ctx, KEYS[k].key, w[k], KEYS[k].shape, KEYS[k].ndim) == DCE_OK, "%s: %s", KEYS[k].key, dce_last_error(ctx));
    CHECK(dce_finalize_weights(ctx, DCE_FP32) == DCE_OK, "finalize: %s", dce_last_error(ctx));

    /* ---- the communicator: id through the file */
    if (rank == 0) CHECK(dce_comm_get_unique_id(id) == DCE_OK, "unique id: %s", dce_last_error(NULL));
    { const int rc = rendezvous(path, rank, nonce, id, timeout_s);
      CHECK(rc == 0, "rendezvous failed (%d): %s", rc, rc == 2 ? "no id with this job's nonce appeared" : "cannot write the id file"); }
    CHECK(dce_comm_init(ctx, rank, world, id) == DCE_OK, "comm_init: %s", dce_last_error(ctx));
    if (rank == 0) remove(path);                              /* every rank has joined: nobody reads it any more */
    int rk = -1, wd = -1, ver = 0; char libname[256];
    CHECK(dce_comm_info(ctx, &rk, &wd, &ver, libname, sizeof libname) == DCE_OK && rk == rank && wd == world, "comm_info: rank %d of %d", rk, wd);

    /* ---- this rank's shard: rows [rank N, (rank + 1) N + 149), results as packed rows on the device */
    void *d_seq = NULL, *d_local = NULL, *d_all = NULL;
    const size_t shard_rows = (size_t)(N + DCE_WINDOW - 1);
    CHECK(hipMalloc(&d_seq, sizeof(float) * shard_rows * DCE_CHANNELS) == hipSuccess && hipMalloc(&d_local, (size_t)N * DCE_PACKED_ROW) == hipSuccess, "hipMalloc");
    if (rank == 0) CHECK(hipMalloc(&d_all, (size_t)total * DCE_PACKED_ROW) == hipSuccess, "hipMalloc (root)");
    CHECK(hipMemcpy(d_seq, seq + (size_t)rank * N * DCE_CHANNELS, sizeof(float) * shard_rows * DCE_CHANNELS, hipMemcpyHostToDevice) == hipSuccess, "H2D");
    CHECK(dce_infer_sequence_packed(ctx, (const float*)d_seq, (int64_t)shard_rows, DCE_WINDOW, 1, (uint8_t*)d_local) == DCE_OK, "shard: %s", dce_last_error(ctx));
    CHECK(dce_gather_results(ctx, (const uint8_t*)d_local, N, (uint8_t*)d_all, NULL, 0, 0) == DCE_OK, "gather: %s", dce_last_error(ctx));
    CHECK(dce_comm_sync(ctx) == DCE_OK && dce_sync(ctx) == DCE_OK, "comm sync: %s", dce_last_error(ctx));
    if (rank == 0) {
        /* the root computes every window itself (chunks of N rows through its own context) and compares byte for byte */
        uint8_t* got = (uint8_t*)malloc((size_t)total * DCE_PACKED_ROW);
        uint8_t* want = (uint8_t*)malloc((size_t)N * DCE_PACKED_ROW);
        CHECK(hipMemcpy(got, d_all, (size_t)total * DCE_PACKED_ROW, hipMemcpyDeviceToHost) == hipSuccess, "D2H");
        for (int r = 0; r < world; ++r) {
            CHECK(dce_infer_sequence_packed(ctx, seq + (size_t)r * N * DCE_CHANNELS, (int64_t)shard_rows, DCE_WINDOW, 0, want) == DCE_OK, "root's own run of shard %d", r);
            CHECK(memcmp(got + (size_t)r * N * DCE_PACKED_ROW, want, (size_t)N * DCE_PACKED_ROW) == 0, "rows gathered from rank %d differ from the root's own", r);
        }
        free(got); free(want);
    }
    int64_t counts[256];
    for (int k = 0; k < 256; ++k) counts[k] = (int64_t)(rank + 1) * (k + 1);
    CHECK(dce_allreduce_counts(ctx, counts, 0) == DCE_OK, "allreduce: %s", dce_last_error(ctx));
    for (int k = 0; k < 256; ++k) CHECK(counts[k] == (int64_t)world * (world + 1) / 2 * (k + 1), "all-reduced count %d", k);
    CHECK(dce_comm_destroy(ctx) == DCE_OK, "comm destroy");
    hipFree(d_seq); hipFree(d_local); if (d_all) hipFree(d_all);
    dce_destroy(ctx);
    printf("abi_ranks[%d]: OK (RCCL %d, %s; world %d, device %d, %ld windows per rank%s)\n", rank, ver, libname, world, device, (long)N,
           rank == 0 ? "; every gathered row equals the root's own" : "");
    for (int k = 0; k < 14; ++k) free(w[k]);
    free(seq);
    return 0;
}
