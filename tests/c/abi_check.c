/* A C99 client of the boundary: include/hmx.h compiles as plain C, libhmx.so loads with dlopen and every
 * declared entry point resolves; a few calls that need no GPU behave as documented (status codes, text).
 * Built and run by tests/test_c_abi_from_c.py. */
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>

#include "hmx.h"

static const char* kSymbols[] = {
    "hmx_last_error", "hmx_abi_version", "hmx_create", "hmx_destroy", "hmx_upload", "hmx_init_cluster",
    "hmx_cluster_round", "hmx_cluster_round_seeded", "hmx_moe_correct_ridge", "hmx_get", "hmx_get_rows", "hmx_set",
    "hmx_sync", "hmx_device_ptr", "hmx_kernel_times", "hmx_enable_timing", "hmx_counters", "hmx_comm_unique_id", "hmx_comm_init",
    "hmx_set_host_allreduce", "hmx_set_ranks", "hmx_peer_export", "hmx_peer_attach", "hmx_peer_selftest",
    "hmx_peer_enable", "hmx_kmeans_seed", "hmx_kmeans_lloyd", "hmx_can_lloyd", "hmx_compute_lisi", "hmx_build_id", "hmx_cluster", "hmx_set_timing_stride",
    "hmx_set_timing_families",
};

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: abi_check /path/to/libhmx.so\n"); return 2; }
    void* lib = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!lib) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 1; }
    size_t i;
    for (i = 0; i < sizeof kSymbols / sizeof kSymbols[0]; ++i)
        if (!dlsym(lib, kSymbols[i])) { fprintf(stderr, "missing symbol %s\n", kSymbols[i]); return 1; }

    int (*abi)(void);
    const char* (*last_error)(void);
    int (*create)(const hmx_config*, hmx_engine**);
    int (*ridge)(hmx_engine*);
    *(void**)(&abi) = dlsym(lib, "hmx_abi_version");                 /* the POSIX idiom for dlsym -> function pointer */
    *(void**)(&last_error) = dlsym(lib, "hmx_last_error");
    *(void**)(&create) = dlsym(lib, "hmx_create");
    *(void**)(&ridge) = dlsym(lib, "hmx_moe_correct_ridge");
    if (abi() != HMX_ABI_VERSION) { fprintf(stderr, "ABI %d != header %d\n", abi(), HMX_ABI_VERSION); return 1; }

    hmx_config cfg;
    hmx_engine* e = NULL;
    memset(&cfg, 0, sizeof cfg);
    cfg.n_cells = 0; cfg.n_pcs = 5; cfg.n_clusters = 3; cfg.n_batches = 2; cfg.n_groups = 2; cfg.n_vars = 1; cfg.n_blocks = 20;
    if (create(&cfg, &e) >= 0) { fprintf(stderr, "hmx_create accepted n_cells = 0\n"); return 1; }
    if (!last_error() || !strlen(last_error())) { fprintf(stderr, "no error text\n"); return 1; }
    if (create(NULL, &e) >= 0 || ridge(NULL) >= 0) { fprintf(stderr, "null arguments accepted\n"); return 1; }
    printf("ok abi=%d sizeof(hmx_config)=%zu\n", abi(), sizeof(hmx_config));
    return 0;
}
