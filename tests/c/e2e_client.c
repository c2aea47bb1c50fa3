/* A plain C99 client that COMPUTES through the C ABI of include/hmx.h, end to end, the way a non-Python binder would:
 *
 *     hmx_create -> hmx_upload -> hmx_init_cluster -> hmx_cluster -> hmx_moe_correct_ridge -> hmx_get(Z_CORR)
 *
 * i.e. one Harmony iteration of harmony.py:419-435 (cluster(): :437-462, moe_correct_ridge(): :535-569) behind the state
 * that Harmony.__init__ / init_cluster build (:230-278, :376-392).  Everything the Python binding does on the host side of
 * the boundary is restated here in C: the group-sorted cell layout (one batch variable: group = batch), the static tile
 * list, Pr_b (:170), theta / sigma / lamb expansion (:137-166), blocks (:474-475).  It pins struct layout, argument order
 * and call order for a binder that is not ctypes; tests/test_c_abi_from_c.py compiles it with gcc (no HIP headers),
 * runs it on a 2 000-cell case and compares Z_corr with the oracle on the same update order and with the Python binding.
 *
 *     e2e_client <libhmx.so> <input.bin> <output.bin>
 *
 * input.bin : int32 {N, d, K, B, rounds, seed}, float Z[N][d], int32 batch[N], float Y0[K][d]   (Y0: centroids as rows)
 * output.bin: int32 {N, d, rounds_run}, double objective terms [rounds_run][4], float Z_corr[N][d] in the caller's row order
 */
#include <dlfcn.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "hmx.h"

#define LOAD(name) \
    name##_t name; \
    *(void**)(&name) = dlsym(lib, "hmx_" #name);      /* the POSIX idiom for dlsym -> function pointer */ \
    if (!name) { fprintf(stderr, "missing symbol hmx_%s\n", #name); return 2; }

typedef const char* (*last_error_t)(void);
typedef int (*abi_version_t)(void);
typedef int (*create_t)(const hmx_config*, hmx_engine**);
typedef void (*destroy_t)(hmx_engine*);
typedef int (*upload_t)(hmx_engine*, const float*, const int32_t*, int64_t, const int32_t*, int32_t, const int32_t*, const float*,
                        const float*, const float*, const float*, const int32_t*, const int32_t*);
typedef int (*init_cluster_t)(hmx_engine*, const float*, double*);
typedef int (*cluster_t)(hmx_engine*, uint64_t, int64_t, int, int, int, double, double*, int32_t*);
typedef int (*moe_correct_ridge_t)(hmx_engine*);
typedef int (*get_t)(hmx_engine*, int, void*, size_t);
typedef int (*counters_t)(hmx_engine*, int64_t*);

static void* xmalloc(size_t n) {
    void* p = malloc(n ? n : 1);
    if (!p) { fprintf(stderr, "out of memory\n"); exit(3); }
    return p;
}

int main(int argc, char** argv) {
    if (argc != 4) { fprintf(stderr, "usage: %s libhmx.so input.bin output.bin\n", argv[0]); return 2; }
    void* lib = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!lib) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
    LOAD(last_error) LOAD(abi_version) LOAD(create) LOAD(destroy) LOAD(upload) LOAD(init_cluster) LOAD(cluster)
    LOAD(moe_correct_ridge) LOAD(get) LOAD(counters)
    if (abi_version() != HMX_ABI_VERSION) { fprintf(stderr, "ABI %d != header %d\n", abi_version(), HMX_ABI_VERSION); return 2; }

    FILE* f = fopen(argv[2], "rb");
    if (!f) { perror(argv[2]); return 2; }
    int32_t hdr[6];
    if (fread(hdr, sizeof hdr, 1, f) != 1) { fprintf(stderr, "short input\n"); return 2; }
    const int N = hdr[0], d = hdr[1], K = hdr[2], B = hdr[3], rounds = hdr[4];
    const uint64_t seed = (uint64_t)hdr[5];
    float* Z = xmalloc((size_t)N * d * sizeof(float));
    int32_t* batch = xmalloc((size_t)N * sizeof(int32_t));
    float* Y0 = xmalloc((size_t)K * d * sizeof(float));
    if (fread(Z, sizeof(float), (size_t)N * d, f) != (size_t)N * d || fread(batch, sizeof(int32_t), N, f) != (size_t)N ||
        fread(Y0, sizeof(float), (size_t)K * d, f) != (size_t)K * d) { fprintf(stderr, "short input\n"); return 2; }
    fclose(f);

    /* ---- host side of the boundary: cells stored sorted by batch group (one variable: group g = batch g) ---------- */
    int32_t* count = calloc((size_t)B + 1, sizeof(int32_t));
    for (int i = 0; i < N; ++i) {
        if (batch[i] < 0 || batch[i] >= B) { fprintf(stderr, "batch code out of range\n"); return 2; }
        count[batch[i]]++;
    }
    for (int b = 0; b < B; ++b)
        if (count[b] == 0) { fprintf(stderr, "empty batch level %d: drop it first (hmx.h, hmx_upload)\n", b); return 2; }
    int32_t* gstart = xmalloc(((size_t)B + 1) * sizeof(int32_t));
    gstart[0] = 0;
    for (int b = 0; b < B; ++b) gstart[b + 1] = gstart[b] + count[b];
    int32_t* source_row = xmalloc((size_t)N * sizeof(int32_t));      /* internal cell -> row of Z (stable within a group) */
    int32_t* fill = calloc((size_t)B, sizeof(int32_t));
    for (int i = 0; i < N; ++i) source_row[gstart[batch[i]] + fill[batch[i]]++] = i;
    int32_t n_tiles = 0;
    for (int b = 0; b < B; ++b) n_tiles += (count[b] + HMX_TILE - 1) / HMX_TILE;
    int32_t* static_cells = xmalloc((size_t)n_tiles * HMX_TILE * sizeof(int32_t));
    int32_t* static_grp = xmalloc((size_t)n_tiles * sizeof(int32_t));
    int32_t t = 0;
    for (int b = 0; b < B; ++b)
        for (int c = gstart[b]; c < gstart[b + 1]; c += HMX_TILE, ++t) {
            static_grp[t] = b;
            for (int i = 0; i < HMX_TILE; ++i) static_cells[(size_t)t * HMX_TILE + i] = c + i < gstart[b + 1] ? c + i : -1;
        }
    int32_t* group_cols = xmalloc((size_t)B * sizeof(int32_t));
    float* Pr_b = xmalloc((size_t)B * sizeof(float));
    float* theta = xmalloc((size_t)B * sizeof(float));
    float* lamb = xmalloc(((size_t)B + 1) * sizeof(float));
    float* sigma = xmalloc((size_t)K * sizeof(float));
    lamb[0] = 0.f;                                                   /* harmony.py:160-166: no penalty on the intercept */
    for (int b = 0; b < B; ++b) {
        group_cols[b] = b;
        Pr_b[b] = (float)count[b] / (float)N;                        /* harmony.py:170 */
        theta[b] = 2.0f;                                             /* harmony.py:137-150, tau = 0 */
        lamb[b + 1] = 1.0f;
    }
    for (int k = 0; k < K; ++k) sigma[k] = 0.1f;                     /* harmony.py:128-131 */
    const double block_size = 0.05;
    const int n_blocks = (int)ceil(1.0 / block_size);                /* harmony.py:474 */
    const int64_t cells_per_block = (int64_t)(N * block_size);       /* harmony.py:475 */

    hmx_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.n_cells = N; cfg.n_cells_global = 0; cfg.n_pcs = d; cfg.n_clusters = K; cfg.n_batches = B; cfg.n_groups = B;
    cfg.n_vars = 1; cfg.n_blocks = n_blocks; cfg.device_id = 0; cfg.lambda_estimation = 0; cfg.alpha = 0.2f;
    hmx_engine* e = NULL;
#define TRY(call) do { int rc_ = (call); if (rc_ != HMX_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, last_error()); return 1; } } while (0)
    TRY(create(&cfg, &e));
    /* global_id = the cell's row in the caller's matrix: the device-side update order is keyed by it */
    TRY(upload(e, Z, static_cells, (int64_t)n_tiles * HMX_TILE, static_grp, n_tiles, group_cols, Pr_b, theta, sigma, lamb,
               source_row, source_row));
    double obj0[4];
    TRY(init_cluster(e, Y0, obj0));                                  /* harmony.py:376-392 */
    double* terms = xmalloc((size_t)(rounds > 0 ? rounds : 1) * 4 * sizeof(double));
    int32_t rounds_run = 0;
    TRY(cluster(e, seed, cells_per_block, rounds, rounds, 3, 1e-5, terms, &rounds_run));   /* harmony.py:437-462, forced */
    TRY(moe_correct_ridge(e));                                       /* harmony.py:535-569 */
    float* Zi = xmalloc((size_t)N * d * sizeof(float));
    TRY(get(e, HMX_Z_CORR, Zi, (size_t)N * d * sizeof(float)));
    float* Zc = xmalloc((size_t)N * d * sizeof(float));
    for (int i = 0; i < N; ++i) memcpy(Zc + (size_t)source_row[i] * d, Zi + (size_t)i * d, (size_t)d * sizeof(float));
    int64_t cnt[HMX_N_COUNTERS];
    TRY(counters(e, cnt));
    destroy(e);

    f = fopen(argv[3], "wb");
    if (!f) { perror(argv[3]); return 2; }
    int32_t oh[3] = {N, d, rounds_run};
    fwrite(oh, sizeof oh, 1, f);
    fwrite(terms, sizeof(double), (size_t)rounds_run * 4, f);
    fwrite(Zc, sizeof(float), (size_t)N * d, f);
    fclose(f);
    printf("e2e ok abi=%d N=%d d=%d K=%d B=%d rounds=%d seeded_rounds=%lld init_objective=%.6g\n", abi_version(), N, d, K, B,
           (int)rounds_run, (long long)cnt[2], obj0[0] + obj0[1] + obj0[2]);
    return 0;
}
