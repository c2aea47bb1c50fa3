// hmx_capi.cpp -- host side of libhmx.so: device state, kernel sequencing, the C ABI of
// include/hmx.h.  Compiled with hipcc as HIP (-x hip).  No torch, no exceptions across the ABI.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "hmx.h"
#include "hmx_internal.h"
#ifdef RTZ_PROF
void rtz_prof_dump();
#endif

struct hmx_nccl_id { char internal[HMX_UNIQUE_ID_BYTES]; };   // layout of ncclUniqueId (rccl.h)

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess)                                                                      \
            return fail(HMX_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

#define HMX_DEFER_MAX 8   /* rounds of one cluster() call whose objective may be read back late (no decision hangs on them) */

enum Family { F_ASSIGN_BLOCK = 0, F_ASSIGN_INIT, F_RTZ_ROUND, F_RTZ_REDUCE, F_BLOCK_TABLE, F_RIDGE_STATS, F_RIDGE_SOLVE, F_RIDGE_APPLY, F_COUNT };
const char* kFamilyNames = "assign_block\0assign_init\0rtz_round\0rtz_reduce\0block_table\0ridge_stats\0ridge_solve\0ridge_apply\0";

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    int reserve(size_t count) {
        if (count <= n) return 0;
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), std::max<size_t>(count, 1) * sizeof(T));
        if (e != hipSuccess) return fail(HMX_ERR_HIP, "hipMalloc(%zu bytes) failed: %s", count * sizeof(T), hipGetErrorString(e));
        n = count;
        return 0;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
};

}  // namespace

struct hmx_engine {
    hmx_config cfg{};
    int64_t N = 0;
    int d = 0, dp = 0, K = 0, Kp = 0, K16 = 0, mt = 0, ntd = 0, ldy = 0, B = 0, G = 0, V = 0, nblk = 0;
    int max_wgs = 1024;
    int ablate = 0;          // HMX_ABLATE: timing experiments only (results become wrong)
    int tiles_per_wave = 1;  // k_assign_lds grid sizing (HMX_TILES_PER_WAVE)
    hipStream_t stream = nullptr;
    bool uploaded = false, clustered = false, timing = false;
    unsigned timing_mask = ~0u;  // kernel families that are bracketed with events while `timing` is on
    int timing_stride = 1;       // ... every timing_stride-th launch of a family
    long fam_seen[8] = {0};      // launches of a family since timing was switched on (bracketed or not)

    DevBuf<float> Zorig, Zcos, Zcorr, R, Y, Yacc, sigma, theta, Pr_b, lamb, rp, lrp, slab, W;
    DevBuf<int> group_cols, s_cells, s_tile_grp, task_t0, task_t1, task_grp;
    // Update-order lists of a round, double buffered: while round r runs, the lists of round r+1 (a function of
    // seed and round counter only) are built on a second stream -- the persistent sweep kernel leaves a few CUs idle.
    struct Lists { DevBuf<int> cells, tile_grp, blk_start, run_tiles; bool runs_ok = false; };   // run_tiles: nblk*G + 1 first tiles of the (block, group) runs (k_round's group-affine map), valid when runs_ok
    Lists lists[2];
    int cur = 0;                         // lists of the round in progress / last round
    hipStream_t stream2 = nullptr;
    hipEvent_t pre_event = nullptr;
    bool pre_valid = false;              // lists[cur ^ 1] hold the round described by pre_*
    bool pre_outstanding = false;        // a build was enqueued on stream2 and nothing has waited for it yet
    uint64_t pre_seed = 0, pre_counter = 0;
    int64_t pre_cpb = 0;
    DevBuf<int> gstart, chunk_tab, run_count, run_start, global_id;
    uint64_t seeded_rounds = 0;
    int64_t Ng = 0;              // cells of the whole job (all ranks)
    DevBuf<double> Ogrp, Tmass, Ohist, scratch;
    // Tables that are summed over ranks live in one allocation, laid out so that tables summed at the
    // same point of the algorithm are neighbours (one collective each):
    //   Sold [nblk][G][K16] | Yacc64 [K16][ldy] | Snew [nblk][G][K16] | objacc [2*SLOTS+2] | Sr [G][K16][ldy] | Oxr [G][K16]
    DevBuf<float> km_hn;         // device k-means: half squared norms of the centres
    DevBuf<double> km_sums;      // device k-means: K16 x (d+1) member sums and counts
    DevBuf<double> Sslots;       // k_round: (nblk + 1) x HMX_ROUND_SLOTS x (G + 1) x K16 (row G of a slot table: the cluster masses, group-affine map)
    DevBuf<unsigned> sync_words; // k_round: {arrival counter, error flag}
    unsigned* sync_host = nullptr;  // pinned copy of sync_words
    int n_cus = 0;
    int rtz_wgs_per_cu = 4;      // k_rtz2 grid (HMX_RTZ_WGS_PER_CU)
    int round_mode = 1;          // 1: persistent sweep kernel when the shape allows it, 0: one launch per block (HMX_ROUND_MODE=blocks)
    unsigned spin_limit = 1u << 24;  // polls a grid-wide wait may take (HMX_SPIN_LIMIT; tests shrink it to force the fall-back)
    long n_sweep_fallbacks = 0;  // rounds repeated through the per-block path after a wait timed out
    bool lists_beside_rtz = true;   // wide shapes: the next round's list build starts beside the R^T.Z pass (HMX_LISTS_BESIDE_RTZ=0: beside the sweep)
    bool wide_sweep = true;      // wide shapes on a single engine: the whole sweep in one persistent launch (k_sweep_wide3; HMX_WIDE_SWEEP=0: one launch per block)
    bool wide_sweep_launched = false;   // ... by the last blocks_loop
    long n_sweeps_wide = 0;
    long n_rtz_bf16 = 0;         // R^T.Z passes launched on the bf16-pipe instance k_rtz3c
    bool allow_round_bf16 = true;   // HMX_ROUND_F32=1 at hmx_create: the f32-input instances of k_round (A/B runs and tests)
    int rtz3_quad = 4;              // tiles a workgroup of the narrow streaming pass takes side by side (8: the tasks are cut for k_rtz3c)
    bool allow_rtz_bf16 = true;     // HMX_RTZ3_BF16=0 at hmx_create: k_rtz3 instead of k_rtz3c
    long n_sweeps_bf16 = 0;      // sweeps launched on the bf16-pipe instances of k_round (round_uses_bf16_pipe)
    long n_sweeps_ga = 0;        // sweeps launched with the group-affine tile map
    bool allow_round_ga = true;  // HMX_ROUND_GA=0 at hmx_create: the classic tile map everywhere (A/B runs and tests)
    std::vector<int> gsize;      // cells of every batch group on this rank (host copy)
    DevBuf<int> ga_map;          // k_round's group-affine map: per compute workgroup {group, rank in group, workgroups of the group}
    int ga_nwg = 0;              // workgroups of that map (0: none planned)
    int ga_per_wg = 0;           // ... planned for this many tiles per workgroup and block (14, or 16 when the grid cannot carry that)
    int64_t ga_key_block = -1;   // ... planned for this largest block size and this cap
    int ga_key_cap = -1;
    bool ga_extra = false;       // ... some group's run may exceed 16 tiles per workgroup (the extra-tile loop will run)
    DevBuf<unsigned long long> wait_stats;   // {waits, incomplete polls, most polls of one wait} of the sweep kernels' grid-wide waits
    DevBuf<double> xch;
    double *Sold = nullptr, *Yacc64 = nullptr, *Snew = nullptr, *objacc = nullptr, *Sr = nullptr, *Oxr = nullptr;
    double* obj_host = nullptr;  // pinned
    double* obj_defer = nullptr; // pinned: objective blocks of rounds whose read-back was deferred (hmx_cluster)
    hipEvent_t sync_event = nullptr;
    long clean_sweeps = 0;       // persistent sweeps that were read back at once and had no time-out
    long n_sweep_launches = 0;   // persistent sweeps launched (HMX_TEST_FAIL_SWEEP counts them)
    long test_fail_sweep = -1;   // HMX_TEST_FAIL_SWEEP=k: the k-th persistent sweep (0-based) runs with spin limit 0 (tests: a time-out in a deferred round)
    unsigned* frozen() const { return reinterpret_cast<unsigned*>(wait_stats.p + 3); }   // sticky time-out word (see hmx_cluster)
    // transport for sharded jobs (null / 1 = single engine)
    void* nccl_comm = nullptr;
    int n_ranks = 1, rank = 0;
    hmx_host_allreduce_fn host_fn = nullptr;
    void* host_ctx = nullptr;
    double* stage_host = nullptr;  // pinned staging buffer of the host transport
    size_t stage_n = 0;
    long n_collectives = 0;
    // peer boxes: the in-kernel exchange of the per-block sums (k_round, cells sharded over ranks)
    double* box = nullptr;               // this rank's box (device memory, exported to the peers)
    size_t box_doubles = 0;
    bool box_fine = false;               // the box is fine-grained device memory (hipExtMallocWithFlags)
    std::vector<void*> peer_ptrs;        // every rank's box as mapped here (own box included)
    DevBuf<double*> peer_dev;            // the same array on the device
    bool peers_attached = false, peers_enabled = false;
    unsigned long long round_epoch = 0;  // flag base of the next sweep launch (+64 per launch, same on every rank)
    unsigned long long selftest_token = 0x5EED0000ull;
    int prefetch_lists = 1;              // HMX_PREFETCH_LISTS=0: build every round's lists on the main stream
    int round_wgs_cap = 0;               // HMX_ROUND_WGS: cap of the sweep grid (tests with several engines on one GPU)
    int n_s_tiles = 0, ntasks = 0;
    std::vector<int> h_task_grp;
    // the R^T.Z pass in storage order (k_rtz3): group-pure tasks over the static tiles, block ids in static tile order
    int rtz_kernel = 3;                  // HMX_RTZ=2: the list-order kernel k_rtz2 everywhere (A/B timing, fall-back)
    bool static_contig = false;          // every static tile holds consecutive cells (what harmonypy_amd builds)
    int ntasks3 = 0;
    DevBuf<int> t3_t0, t3_t1, t3_c0, t3_cend, t3_grp, t3_stride, s_tile_start;
    DevBuf<unsigned char> tile_blk[2], tile_blk_zero;
    DevBuf<unsigned> Yf;                 // wide shapes: the round's Y as bf16 fragments for k_assign_wide3 (launch_y_planes)
    DevBuf<unsigned> Wf;                 // wide shapes: W as bf16 fragments for k_ridge_apply_wideb (launch_w_planes)
    DevBuf<unsigned> Zcf;                // wide shapes: Z_cos as bf16 planes in k_rtzw2b's B-fragment order (launch_zplanes), rebuilt when Z_cos changed
    bool zcf_valid = false;              // Zcf holds the planes of the current Z_cos
    bool fuse_block_table = true;        // HMX_FUSE_TABLE=0 at hmx_create: a k_block_table launch in front of every wide block assignment (A/B runs and tests)
    bool allow_zcf = true;               // HMX_RTZW_ZF=0 at hmx_create: k_rtzw2b splits the fp32 rows of Z_cos in every pass (A/B runs and tests)
    long n_rtz_zf = 0;                   // streaming passes that read the pre-split planes
    DevBuf<double> Opriv;                // k_sweep_wide3: every workgroup's own copies of O (2 x grid x G x K16)
    DevBuf<double> Osave;                // O at the start of the round in flight (exact replay after a time-out)

    struct Span { hipEvent_t a, b; int fam; };
    std::vector<Span> spans;
    std::vector<hipEvent_t> pool;
    double fam_ms[F_COUNT] = {0};
    long fam_n[F_COUNT] = {0};
};

namespace {

int use_device(hmx_engine* e) {
    HIP_TRY(hipSetDevice(e->cfg.device_id));
    return 0;
}


// ---- transport of sharded jobs -------------------------------------------------------------
// RCCL is bound at run time (dlopen) so that a single-GPU user never needs it: the entry points
// below are the only ones used.  The library must share this process's HIP runtime with the
// engine (stream handles are runtime objects): the ROCm installation's librccl.so.1 does.
struct RcclApi {
    void* handle = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, hmx_nccl_id, int) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
RcclApi g_rccl;

int rccl_load() {
    if (g_rccl.handle) return 0;
    const char* names[] = {getenv("HMX_RCCL_LIB"), "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
    void* h = nullptr;
    std::string tried;
    for (const char* n : names) {
        if (!n || !*n) continue;
        h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (h) break;
        tried += std::string(" ") + n + " (" + dlerror() + ")";
    }
    if (!h) return fail(HMX_ERR_COMM, "RCCL not found:%s", tried.c_str());
    RcclApi a;
    a.handle = h;
    a.GetUniqueId = reinterpret_cast<int (*)(void*)>(dlsym(h, "ncclGetUniqueId"));
    a.CommInitRank = reinterpret_cast<int (*)(void**, int, hmx_nccl_id, int)>(dlsym(h, "ncclCommInitRank"));
    a.AllReduce = reinterpret_cast<int (*)(const void*, void*, size_t, int, int, void*, hipStream_t)>(dlsym(h, "ncclAllReduce"));
    a.CommDestroy = reinterpret_cast<int (*)(void*)>(dlsym(h, "ncclCommDestroy"));
    a.GetErrorString = reinterpret_cast<const char* (*)(int)>(dlsym(h, "ncclGetErrorString"));
    if (!a.GetUniqueId || !a.CommInitRank || !a.AllReduce || !a.CommDestroy || !a.GetErrorString)
        return fail(HMX_ERR_COMM, "RCCL library lacks an expected entry point");
    g_rccl = a;
    return 0;
}

void comm_release(hmx_engine* e) {
    if (e->nccl_comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(e->nccl_comm);
    e->nccl_comm = nullptr;
}

void peer_release(hmx_engine* e) {
    for (size_t r = 0; r < e->peer_ptrs.size(); ++r)
        if (e->peer_ptrs[r] && e->peer_ptrs[r] != (void*)e->box) (void)hipIpcCloseMemHandle(e->peer_ptrs[r]);
    e->peer_ptrs.clear();
    e->peer_dev.release();
    if (e->box) (void)hipFree(e->box);
    e->box = nullptr;
    e->peers_attached = e->peers_enabled = false;
}

// Sum `count` doubles at device pointer `p` over all ranks, in place, ordered on the engine's stream.
int sum_over_ranks(hmx_engine* e, double* p, size_t count) {
    if (count == 0) return 0;
    if (e->nccl_comm) {
        const int r = g_rccl.AllReduce(p, p, count, /*ncclFloat64*/ 8, /*ncclSum*/ 0, e->nccl_comm, e->stream);
        if (r != 0) return fail(HMX_ERR_COMM, "ncclAllReduce(%zu doubles) failed: %s", count, g_rccl.GetErrorString(r));
        e->n_collectives++;
        return 0;
    }
    if (e->host_fn) {
        if (count > e->stage_n) {
            if (e->stage_host) (void)hipHostFree(e->stage_host);
            e->stage_host = nullptr;
            e->stage_n = 0;
            HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&e->stage_host), count * sizeof(double), hipHostMallocDefault));
            e->stage_n = count;
        }
        HIP_TRY(hipMemcpyAsync(e->stage_host, p, count * sizeof(double), hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
        const int r = e->host_fn(e->host_ctx, e->stage_host, count);
        if (r != 0) return fail(HMX_ERR_COMM, "host all-reduce callback returned %d", r);
        HIP_TRY(hipMemcpyAsync(p, e->stage_host, count * sizeof(double), hipMemcpyHostToDevice, e->stream));
        e->n_collectives++;
        return 0;
    }
    return 0;
}

bool sharded(const hmx_engine* e) { return e->nccl_comm != nullptr || e->host_fn != nullptr; }

struct Timed {
    hmx_engine* e;
    hipEvent_t a = nullptr, b = nullptr;
    int fam;
    Timed(hmx_engine* e_, int fam_) : e(e_), fam(fam_) {
        if (!e->timing || !((e->timing_mask >> fam_) & 1u)) return;
        if ((e->fam_seen[fam_]++ % e->timing_stride) != 0) return;
        auto get = [&]() {
            hipEvent_t ev;
            if (!e->pool.empty()) { ev = e->pool.back(); e->pool.pop_back(); }
            else (void)hipEventCreate(&ev);
            return ev;
        };
        a = get();
        b = get();
        (void)hipEventRecord(a, e->stream);
    }
    ~Timed() {
        if (!a) return;
        (void)hipEventRecord(b, e->stream);
        e->spans.push_back({a, b, fam});
    }
};

void drain_spans(hmx_engine* e) {
    for (auto& s : e->spans) {
        float ms = 0.f;
        if (hipEventSynchronize(s.b) == hipSuccess && hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) {
            e->fam_ms[s.fam] += ms;
            e->fam_n[s.fam] += 1;
        }
        e->pool.push_back(s.a);
        e->pool.push_back(s.b);
    }
    e->spans.clear();
}

// Wait for the engine's stream.  The wake-up of a blocking synchronise costs 15-25 us -- on a 500 us round that is worth
// a short spin on an event first (the round trip sits on the critical path of every round that ends in a decision).
int wait_stream(hmx_engine* e) {
    if (e->sync_event && hipEventRecord(e->sync_event, e->stream) == hipSuccess) {
        for (int spin = 0; spin < 200000; ++spin) {
            const hipError_t q = hipEventQuery(e->sync_event);
            if (q == hipSuccess) return 0;
            if (q != hipErrorNotReady) break;
        }
        (void)hipGetLastError();   // hipErrorNotReady is sticky in hipGetLastError
    }
    HIP_TRY(hipStreamSynchronize(e->stream));
    return 0;
}

void fold_objective(const double* host, double out[4]) {
    double km = 0.0, ent = 0.0;
    for (int i = 0; i < HMX_OBJ_SLOTS; ++i) { km += host[2 * i]; ent += host[2 * i + 1]; }
    out[0] = (double)(float)km;  // `.item()` of an fp32 tensor
    out[1] = (double)(float)ent;
    out[2] = (double)(float)host[2 * HMX_OBJ_SLOTS];
    out[3] = 0.0;
}

int read_objective(hmx_engine* e, double out[4]) {
    const int n = 2 * HMX_OBJ_SLOTS + 2;
    int rc;
    HIP_TRY(hipMemcpyAsync(e->obj_host, e->objacc, n * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    if ((rc = wait_stream(e))) return rc;
    HIP_TRY(hipGetLastError());
    fold_objective(e->obj_host, out);
    return 0;
}

AssignArgs assign_args(hmx_engine* e) {
    AssignArgs a{};
    a.Zcos = e->Zcos.p; a.Y = e->Y.p; a.sigma = e->sigma.p; a.rp = e->rp.p; a.lrp = e->lrp.p; a.R = e->R.p;
    a.obj = e->objacc;
    a.K = e->K; a.Kp = e->Kp; a.K16 = e->K16; a.mt = e->mt; a.dp = e->dp; a.ldy = e->ldy;
    a.G = e->G; a.tiles_per_wave = e->tiles_per_wave; a.ablate = e->ablate;
    a.bf16_pipe = e->allow_round_bf16 ? 1 : 0;
    return a;
}

TableArgs table_args(hmx_engine* e) {
    TableArgs t{};
    t.group_cols = e->group_cols.p; t.Pr_b = e->Pr_b.p; t.theta = e->theta.p; t.sigma = e->sigma.p;
    t.G = e->G; t.B = e->B; t.V = e->V; t.K16 = e->K16;
    return t;
}

}  // namespace

extern "C" {

const char* hmx_last_error(void) { return g_err.c_str(); }
int hmx_abi_version(void) { return HMX_ABI_VERSION; }

int hmx_create(const hmx_config* cfg, hmx_engine** out) {
    if (!cfg || !out) return fail(HMX_ERR_ARG, "null argument");
    *out = nullptr;
    if (cfg->n_cells <= 0 || cfg->n_pcs <= 0 || cfg->n_clusters <= 0 || cfg->n_batches <= 0 || cfg->n_groups <= 0 ||
        cfg->n_vars <= 0 || cfg->n_blocks <= 0)
        return fail(HMX_ERR_ARG, "sizes must be positive");
    if (cfg->n_clusters > HMX_MAX_CLUSTERS) return fail(HMX_ERR_ARG, "n_clusters=%d > %d not supported by this build", cfg->n_clusters, HMX_MAX_CLUSTERS);
    if (cfg->n_pcs > HMX_MAX_PCS) return fail(HMX_ERR_ARG, "n_pcs=%d > %d not supported by this build", cfg->n_pcs, HMX_MAX_PCS);
    if (cfg->n_cells > (int64_t)2000000000) return fail(HMX_ERR_ARG, "n_cells too large for 32-bit cell ids");
    if (cfg->n_blocks > HMX_MAX_BLOCKS) return fail(HMX_ERR_ARG, "n_blocks=%d > %d not supported by this build", cfg->n_blocks, HMX_MAX_BLOCKS);
    if (cfg->n_vars > HMX_MAX_VARS) return fail(HMX_ERR_ARG, "n_vars=%d > %d not supported by this build", cfg->n_vars, HMX_MAX_VARS);
    if (cfg->n_cells_global != 0 && (cfg->n_cells_global < cfg->n_cells || cfg->n_cells_global > (int64_t)2000000000))
        return fail(HMX_ERR_ARG, "n_cells_global must lie in [n_cells, 2e9]");
    int ndev = 0;
    hipError_t he = hipGetDeviceCount(&ndev);
    if (he != hipSuccess || ndev <= 0) return fail(HMX_ERR_HIP, "no HIP device available (%s)", hipGetErrorString(he));
    if (cfg->device_id < 0 || cfg->device_id >= ndev) return fail(HMX_ERR_ARG, "device_id %d out of range (%d devices)", cfg->device_id, ndev);
    hmx_engine* e = new (std::nothrow) hmx_engine();
    if (!e) return fail(HMX_ERR_ARG, "out of host memory");
    e->cfg = *cfg;
    e->N = cfg->n_cells; e->d = cfg->n_pcs; e->K = cfg->n_clusters; e->B = cfg->n_batches; e->G = cfg->n_groups;
    e->V = cfg->n_vars; e->nblk = cfg->n_blocks;
    e->Ng = cfg->n_cells_global > 0 ? cfg->n_cells_global : cfg->n_cells;
    // rows of 32 / 52 / 64 floats feed the sweep kernels; wide rows: whole 16-column k-steps.  (Rows of 256 bytes for
    // d = 50 -- whole cache lines -- were measured: k_round 391 us instead of 374, k_rtz2 228 instead of 187 at C3: the
    // kernels compute on the padding)
    e->dp = round_row_floats(e->d) ? round_row_floats(e->d) : ((e->d + 15) & ~15);
    e->Kp = (e->K + 3) & ~3;
    e->mt = (e->K + 15) / 16;
    e->K16 = 16 * e->mt;
    e->ntd = (e->dp + 15) / 16;   // PC tiles cover the padded row (padding columns hold zeros)
    e->ldy = 16 * e->ntd;
    if (const char* ab = getenv("HMX_ABLATE")) e->ablate = atoi(ab);
    if (const char* tpw = getenv("HMX_TILES_PER_WAVE")) e->tiles_per_wave = std::max(1, atoi(tpw));
    if (const char* rw = getenv("HMX_RTZ_WGS_PER_CU")) e->rtz_wgs_per_cu = std::max(1, std::min(8, atoi(rw)));
    if (const char* rc_ = getenv("HMX_ROUND_WGS")) e->round_wgs_cap = std::max(0, atoi(rc_));
    if (const char* pl = getenv("HMX_PREFETCH_LISTS")) e->prefetch_lists = atoi(pl) != 0;
    if (const char* rm = getenv("HMX_ROUND_MODE")) e->round_mode = (std::string(rm) == "blocks") ? 0 : 1;
    if (const char* rf = getenv("HMX_ROUND_F32")) e->allow_round_bf16 = atoi(rf) == 0;
    if (const char* rg = getenv("HMX_ROUND_GA")) e->allow_round_ga = atoi(rg) != 0;
    if (const char* rb = getenv("HMX_RTZ3_BF16")) e->allow_rtz_bf16 = atoi(rb) != 0;
    if (const char* zf = getenv("HMX_RTZW_ZF")) e->allow_zcf = atoi(zf) != 0;
    if (const char* ft = getenv("HMX_FUSE_TABLE")) e->fuse_block_table = atoi(ft) != 0;
    if (const char* rk = getenv("HMX_RTZ")) e->rtz_kernel = atoi(rk) == 2 ? 2 : 3;
    if (const char* fs = getenv("HMX_TEST_FAIL_SWEEP")) e->test_fail_sweep = atol(fs);
    if (const char* ws = getenv("HMX_WIDE_SWEEP")) e->wide_sweep = atoi(ws) != 0;
    if (const char* lb = getenv("HMX_LISTS_BESIDE_RTZ")) e->lists_beside_rtz = atoi(lb) != 0;
    if (const char* sl = getenv("HMX_SPIN_LIMIT")) e->spin_limit = (unsigned)std::max(0L, atol(sl));   // 0: every wait of the persistent kernels gives up at once (tests)
    int rc = 0;
    do {
        if ((rc = use_device(e))) break;
        (void)hipDeviceGetAttribute(&e->n_cus, hipDeviceAttributeMultiprocessorCount, cfg->device_id);
        if (e->n_cus <= 0) e->n_cus = 64;
        (void)hipStreamCreateWithFlags(&e->stream2, hipStreamNonBlocking);
        (void)hipEventCreateWithFlags(&e->pre_event, hipEventDisableTiming);
        hipError_t se = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking);
        if (se != hipSuccess) { rc = fail(HMX_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(se)); break; }
        const size_t N = (size_t)e->N, GK = (size_t)e->G * e->K16, GKs = GK + e->K16;
        // + 16 rows of slack behind Z_orig / Z_cos / R: the streaming pass (k_rtz3) fetches whole 16-cell tiles
        if ((rc = e->Zorig.reserve((N + 16) * e->dp)) || (rc = e->Zcos.reserve((N + 16) * e->dp)) || (rc = e->Zcorr.reserve(N * e->dp)) ||
            (rc = e->R.reserve((N + 16) * e->Kp + e->K16 + 64)) || (rc = e->Osave.reserve(GK)) || (rc = e->Y.reserve((size_t)e->K16 * e->ldy)) ||
            (rc = e->Yacc.reserve((size_t)e->K16 * e->ldy)) || (rc = e->sigma.reserve(e->K16)) ||
            (rc = e->theta.reserve(e->B)) || (rc = e->Pr_b.reserve(e->B)) || (rc = e->lamb.reserve(e->B + 1)) ||
            (rc = e->rp.reserve(2 * GK * e->nblk)) || (rc = e->lrp.reserve(GK)) || (rc = e->group_cols.reserve((size_t)e->G * e->V)) ||
            (rc = e->Ogrp.reserve(GK)) || (rc = e->Tmass.reserve(e->K16)) || (rc = e->Ohist.reserve(GK * e->nblk)) ||
            (rc = e->W.reserve(GK * e->ldy)) || (rc = e->lists[0].blk_start.reserve(e->nblk + 1)) ||
            (rc = e->lists[1].blk_start.reserve(e->nblk + 1)) ||
            (rc = e->lists[0].run_tiles.reserve((size_t)e->nblk * e->G + 1)) || (rc = e->lists[1].run_tiles.reserve((size_t)e->nblk * e->G + 1)))
            break;
        {
            const size_t n_sold = GK * e->nblk, n_y = (size_t)e->K16 * e->ldy, n_obj = 2 * HMX_OBJ_SLOTS + 2;
            if ((rc = e->xch.reserve(2 * n_sold + n_y + n_obj + GK * e->ldy + GK))) break;
            e->Sold = e->xch.p;
            e->Yacc64 = e->Sold + n_sold;
            e->Snew = e->Yacc64 + n_y;
            e->objacc = e->Snew + n_sold;
            e->Sr = e->objacc + n_obj;
            e->Oxr = e->Sr + GK * e->ldy;
        }
        if ((rc = e->Sslots.reserve(GKs * (e->nblk + 1) * HMX_ROUND_SLOTS + 1 + 160 + 256)) || (rc = e->wait_stats.reserve(8))) break;   // + slack: the per-round fill is rounded up to 1 KB
        (void)hipMemsetAsync(e->wait_stats.p, 0, 8 * sizeof(unsigned long long), e->stream);
        e->sync_words.p = reinterpret_cast<unsigned*>(e->Sslots.p + GKs * (e->nblk + 1) * HMX_ROUND_SLOTS);   // borrowed tail
        e->sync_words.n = 2;
        if (hipHostMalloc(reinterpret_cast<void**>(&e->sync_host), 2 * sizeof(unsigned), hipHostMallocDefault) != hipSuccess) {
            rc = fail(HMX_ERR_HIP, "hipHostMalloc failed");
            break;
        }
        e->sync_host[0] = e->sync_host[1] = 0;
        if (e->V > 1 && (rc = e->scratch.reserve((size_t)e->K16 * (e->B + 1) * (e->B + 1 + e->d)))) break;
        hipError_t pe = hipHostMalloc(reinterpret_cast<void**>(&e->obj_host), (2 * HMX_OBJ_SLOTS + 2) * sizeof(double), hipHostMallocDefault);
        if (pe != hipSuccess) { rc = fail(HMX_ERR_HIP, "hipHostMalloc: %s", hipGetErrorString(pe)); break; }
        pe = hipHostMalloc(reinterpret_cast<void**>(&e->obj_defer), (size_t)HMX_DEFER_MAX * (2 * HMX_OBJ_SLOTS + 2) * sizeof(double), hipHostMallocDefault);
        if (pe != hipSuccess) { rc = fail(HMX_ERR_HIP, "hipHostMalloc: %s", hipGetErrorString(pe)); break; }
        (void)hipEventCreateWithFlags(&e->sync_event, hipEventDisableTiming);
        (void)hipMemsetAsync(e->R.p, 0, N * e->Kp * sizeof(float), e->stream);
        (void)hipMemsetAsync(e->Ogrp.p, 0, GK * sizeof(double), e->stream);
        (void)hipMemsetAsync(e->Tmass.p, 0, e->K16 * sizeof(double), e->stream);
        (void)hipMemsetAsync(e->W.p, 0, GK * e->ldy * sizeof(float), e->stream);
        (void)hipMemsetAsync(e->Y.p, 0, (size_t)e->K16 * e->ldy * sizeof(float), e->stream);
        (void)hipMemsetAsync(e->objacc, 0, (2 * HMX_OBJ_SLOTS + 2) * sizeof(double), e->stream);
    } while (0);
    if (rc) {
        std::string keep = g_err;
        hmx_destroy(e);
        g_err = keep;
        return rc;
    }
    *out = e;
    return HMX_OK;
}

void hmx_destroy(hmx_engine* e) {
    if (!e) return;
    (void)hipSetDevice(e->cfg.device_id);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    drain_spans(e);
    for (auto ev : e->pool) (void)hipEventDestroy(ev);
    e->Zorig.release(); e->Zcos.release(); e->Zcorr.release(); e->R.release(); e->Y.release(); e->Yacc.release();
    e->sigma.release(); e->theta.release(); e->Pr_b.release(); e->lamb.release(); e->rp.release(); e->lrp.release();
    e->slab.release(); e->W.release(); e->group_cols.release(); e->s_cells.release(); e->s_tile_grp.release();
    for (auto& L : e->lists) { L.cells.release(); L.tile_grp.release(); L.blk_start.release(); L.run_tiles.release(); }
    e->ga_map.release();
    if (e->stream2) { (void)hipStreamSynchronize(e->stream2); (void)hipStreamDestroy(e->stream2); }
    if (e->pre_event) (void)hipEventDestroy(e->pre_event);
    e->task_t0.release(); e->task_t1.release();
    e->t3_t0.release(); e->t3_t1.release(); e->t3_stride.release(); e->t3_c0.release(); e->t3_cend.release(); e->t3_grp.release(); e->s_tile_start.release();
    e->tile_blk[0].release(); e->tile_blk[1].release(); e->tile_blk_zero.release(); e->Osave.release(); e->Opriv.release(); e->Wf.release(); e->Yf.release(); e->Zcf.release();
    e->task_grp.release(); e->gstart.release(); e->chunk_tab.release(); e->run_count.release(); e->run_start.release();
    e->Ogrp.release(); e->Tmass.release(); e->Ohist.release(); e->xch.release(); e->scratch.release();
    e->global_id.release(); e->wait_stats.release(); e->sync_words.p = nullptr; e->sync_words.n = 0; e->Sslots.release(); e->km_hn.release(); e->km_sums.release();
    if (e->sync_host) (void)hipHostFree(e->sync_host);
    comm_release(e);
    peer_release(e);
    if (e->stage_host) (void)hipHostFree(e->stage_host);
    if (e->obj_host) (void)hipHostFree(e->obj_host);
    if (e->obj_defer) (void)hipHostFree(e->obj_defer);
    if (e->sync_event) (void)hipEventDestroy(e->sync_event);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
}

int hmx_upload(hmx_engine* e, const float* Z, const int32_t* static_cells, int64_t n_static_pos,
               const int32_t* static_tile_group, int32_t n_static_tiles, const int32_t* group_cols, const float* Pr_b,
               const float* theta, const float* sigma, const float* lamb, const int32_t* global_id,
               const int32_t* source_row) {
    if (!e || !Z || !static_cells || !static_tile_group || !group_cols || !Pr_b || !theta || !sigma)
        return fail(HMX_ERR_ARG, "null argument");
    if (!e->cfg.lambda_estimation && !lamb) return fail(HMX_ERR_ARG, "lamb is required unless lambda_estimation");
    if (n_static_pos != (int64_t)n_static_tiles * HMX_TILE) return fail(HMX_ERR_ARG, "n_static_pos must be 16*n_static_tiles");
    for (int64_t i = 0; i < n_static_pos; ++i)
        if (static_cells[i] < -1 || static_cells[i] >= e->N) return fail(HMX_ERR_ARG, "static_cells[%lld] out of range", (long long)i);
    for (int i = 0; i < n_static_tiles; ++i)
        if (static_tile_group[i] < 0 || static_tile_group[i] >= e->G) return fail(HMX_ERR_ARG, "static_tile_group[%d] out of range", i);
    for (int i = 0; i < e->G * e->V; ++i)
        if (group_cols[i] < 0 || group_cols[i] >= e->B) return fail(HMX_ERR_ARG, "group_cols[%d] out of range", i);
    if (e->V == 1) {
        // one batch variable: group g IS batch g (the closed-form ridge solve and the sweep's tables index by it)
        if (e->G != e->B) return fail(HMX_ERR_ARG, "one batch variable needs n_groups == n_batches (%d != %d): drop batch levels without cells", e->G, e->B);
        for (int g = 0; g < e->G; ++g)
            if (group_cols[g] != g) return fail(HMX_ERR_ARG, "one batch variable needs group_cols[g] == g (group_cols[%d] = %d)", g, group_cols[g]);
    }
    if (!global_id && e->Ng != e->N) return fail(HMX_ERR_ARG, "global_id is required when n_cells_global != n_cells");
    if (global_id)
        for (int64_t i = 0; i < e->N; ++i)
            if (global_id[i] < 0 || global_id[i] >= e->Ng) return fail(HMX_ERR_ARG, "global_id[%lld] out of range", (long long)i);
    int rc;
    if ((rc = use_device(e))) return rc;
    if (global_id) {
        if ((rc = e->global_id.reserve(e->N))) return rc;
        HIP_TRY(hipMemcpyAsync(e->global_id.p, global_id, e->N * sizeof(int), hipMemcpyHostToDevice, e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
    } else {
        e->global_id.release();
    }
    if (source_row)
        for (int64_t i = 0; i < e->N; ++i)
            if (source_row[i] < 0 || source_row[i] >= e->N) return fail(HMX_ERR_ARG, "source_row[%lld] out of range", (long long)i);
    // Z travels as it is (N x d); the device pads the rows to dp and, with source_row, brings them
    // into the group-sorted order.  Z_corr's storage is the landing area (N x d <= N x dp floats).
    {
        DevBuf<int> srow;
        if (source_row) {
            if ((rc = srow.reserve(e->N))) return rc;
            HIP_TRY(hipMemcpyAsync(srow.p, source_row, e->N * sizeof(int), hipMemcpyHostToDevice, e->stream));
        }
        HIP_TRY(hipMemcpyAsync(e->Zcorr.p, Z, (size_t)e->N * e->d * sizeof(float), hipMemcpyHostToDevice, e->stream));
        launch_load_rows(e->Zcorr.p, e->d, srow.p, e->Zorig.p, e->dp, e->N, e->stream);
        HIP_TRY(hipMemcpyAsync(e->Zcorr.p, e->Zorig.p, (size_t)e->N * e->dp * sizeof(float), hipMemcpyDeviceToDevice, e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
        srow.release();
    }
    launch_normalize_rows(e->Zorig.p, e->Zcos.p, e->N, e->dp, e->stream);  // harmony.py:238
    e->zcf_valid = false;
    std::vector<float> sg(e->K16, 0.f);
    std::memcpy(sg.data(), sigma, sizeof(float) * e->K);
    HIP_TRY(hipMemcpyAsync(e->sigma.p, sg.data(), sg.size() * sizeof(float), hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipMemcpyAsync(e->theta.p, theta, e->B * sizeof(float), hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipMemcpyAsync(e->Pr_b.p, Pr_b, e->B * sizeof(float), hipMemcpyHostToDevice, e->stream));
    std::vector<float> lm(e->B + 1, 0.f);
    if (lamb && !e->cfg.lambda_estimation) std::memcpy(lm.data(), lamb, sizeof(float) * (e->B + 1));
    HIP_TRY(hipMemcpyAsync(e->lamb.p, lm.data(), lm.size() * sizeof(float), hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipMemcpyAsync(e->group_cols.p, group_cols, (size_t)e->G * e->V * sizeof(int), hipMemcpyHostToDevice, e->stream));
    if ((rc = e->s_cells.reserve(n_static_pos)) || (rc = e->s_tile_grp.reserve(n_static_tiles))) return rc;
    HIP_TRY(hipMemcpyAsync(e->s_cells.p, static_cells, n_static_pos * sizeof(int), hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipMemcpyAsync(e->s_tile_grp.p, static_tile_group, n_static_tiles * sizeof(int), hipMemcpyHostToDevice, e->stream));
    e->n_s_tiles = n_static_tiles;
    {   // first internal cell of every group (cells are stored group-sorted)
        std::vector<int> gcount(e->G, 0), gs(e->G + 1, 0);
        for (int t = 0; t < n_static_tiles; ++t)
            for (int i = 0; i < HMX_TILE; ++i)
                if (static_cells[(size_t)t * HMX_TILE + i] >= 0) gcount[static_tile_group[t]]++;
        for (int g = 0; g < e->G; ++g) gs[g + 1] = gs[g] + gcount[g];
        if (gs[e->G] != e->N) return fail(HMX_ERR_ARG, "static list must hold every cell exactly once");
        e->gsize = gcount;
        e->ga_nwg = 0; e->ga_key_block = -1;
        if ((rc = e->gstart.reserve(e->G + 1))) return rc;
        HIP_TRY(hipMemcpyAsync(e->gstart.p, gs.data(), gs.size() * sizeof(int), hipMemcpyHostToDevice, e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
    }
    // ridge tasks: runs of tiles of one group; with k_rtz2 a task is one workgroup's share, sized
    // for about four tasks per CU (k_rtz: one task per wave, <= 64 tiles)
    std::vector<int> t0, t1, tg;
    const int CH = rtz2_ok(e->mt, e->dp) ? std::max(16, (n_static_tiles + 2 * e->rtz_wgs_per_cu * e->n_cus - 1) / (2 * e->rtz_wgs_per_cu * e->n_cus))
                   : rtz_wide_ok(e->mt, e->dp) ? std::max(16, (n_static_tiles + 2 * e->n_cus - 1) / (2 * e->n_cus)) : 64;
    for (int i = 0; i < n_static_tiles;) {
        int j = i;
        while (j < n_static_tiles && j - i < CH && static_tile_group[j] == static_tile_group[i]) ++j;
        t0.push_back(i); t1.push_back(j); tg.push_back(static_tile_group[i]);
        i = j;
    }
    for (size_t i = 1; i < tg.size(); ++i)
        if (tg[i] < tg[i - 1]) return fail(HMX_ERR_ARG, "static tiles must be sorted by group");
    e->ntasks = (int)t0.size();
    e->h_task_grp = tg;
    if ((rc = e->task_t0.reserve(t0.size())) || (rc = e->task_t1.reserve(t0.size())) || (rc = e->task_grp.reserve(t0.size()))) return rc;
    HIP_TRY(hipMemcpyAsync(e->task_t0.p, t0.data(), t0.size() * sizeof(int), hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipMemcpyAsync(e->task_t1.p, t1.data(), t1.size() * sizeof(int), hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipMemcpyAsync(e->task_grp.p, tg.data(), tg.size() * sizeof(int), hipMemcpyHostToDevice, e->stream));
    {   // the streaming R^T.Z pass (k_rtz3) needs consecutive cells in every static tile -- what harmonypy_amd's layout is
        bool contig = true;
        std::vector<int> tstart(e->G + 1, 0);
        for (int t = 0; t < n_static_tiles && contig; ++t) {
            const int32_t* c = static_cells + (size_t)t * HMX_TILE;
            int n = 0;
            while (n < HMX_TILE && c[n] >= 0) ++n;
            if (n == 0) { contig = false; break; }
            for (int i = 0; i < HMX_TILE; ++i)
                if (c[i] != (i < n ? c[0] + i : -1)) contig = false;
            // a tile that is not full is the last of its group; a group's tiles continue each other's cells
            const bool last_of_group = t + 1 == n_static_tiles || static_tile_group[t + 1] != static_tile_group[t];
            if (!last_of_group && (n < HMX_TILE || static_cells[(size_t)(t + 1) * HMX_TILE] != c[0] + HMX_TILE)) contig = false;
        }
        for (int t = 0; t < n_static_tiles; ++t) tstart[static_tile_group[t] + 1]++;
        for (int g = 0; g < e->G; ++g) tstart[g + 1] += tstart[g];
        // groups without cells have no tile; a group's first tile starts at the group's first cell
        std::vector<int> gs2(e->G + 1, 0);
        {
            std::vector<int> gcount(e->G, 0);
            for (int t = 0; t < n_static_tiles; ++t)
                for (int i = 0; i < HMX_TILE; ++i)
                    if (static_cells[(size_t)t * HMX_TILE + i] >= 0) gcount[static_tile_group[t]]++;
            for (int g = 0; g < e->G; ++g) gs2[g + 1] = gs2[g] + gcount[g];
        }
        for (int g = 0; g < e->G && contig; ++g)
            if (tstart[g + 1] > tstart[g] && static_cells[(size_t)tstart[g] * HMX_TILE] != gs2[g]) contig = false;
        e->static_contig = contig;
        e->ntasks3 = 0;
        if (contig) {
            std::vector<int> a0, a1, ac0, acend, ag, ast;
            // k_rtz3 keeps two workgroups per CU resident, k_rtz3c (sixteen tile buffers) one: as many tasks as fit at once,
            // or a second round of workgroups pays the prologue, the slab reduction and the tail again (measured: 188 us per
            // pass with 505 tasks of 31 tiles per wave)
            // The cut is decided from the ROUND pass (e->nblk block columns): ten of the eleven passes of a Harmony iteration.
            // The ridge statistics and the device Lloyd iterations run the same tasks with ONE block column and may be served
            // by the other kernel of the family there (wide shapes: rtzw2b_ok depends on the column tiles) -- results are the
            // same on any cut, the cut is tuned for the majority pass (advisor finding, round 5).
            const bool one_per_cu = e->allow_rtz_bf16 && (rtz_wide_ok(e->mt, e->dp) ? rtzw2b_ok(e->mt, e->dp, e->d, e->nblk) : rtz3b_ok(e->mt, e->dp, e->nblk, e->Kp));
            const int target = std::max(1, (one_per_cu ? 1 : 2) * e->n_cus - e->G);
            // (one workgroup per CU: tasks of up to 4096 tiles, so that 10 M cells still make ONE round of ~240 workgroups -- with the
            // cap of 2048 of round 5 they made 305 tasks on 256 CUs: a second, mostly idle round of workgroups, k_rtz3c 1 647 us for
            // 10 M cells against 124.5 us for 1 M.  A wave's fp32 accumulators then sum 512 tiles = 8 k cells before the fp64 fold.)
            int cap3 = one_per_cu ? 4096 : 256;
            if (const char* tc = getenv("HMX_RTZ3_TASK_CAP")) cap3 = std::max(16, atoi(tc));   // (A/B runs)
            const int CH3 = std::max(16, std::min(cap3, (n_static_tiles + target - 1) / target));
            // HMX_RTZ3_TASKS=contig: a task is a contiguous run of a group's tiles; default: the m tasks of a group take
            // neighbouring quads of tiles (task j: tiles ts + 4j + w + 4m i) and sweep the group's rows together
            const char* tk = getenv("HMX_RTZ3_TASKS");
            const bool interleave = !(tk && std::string(tk) == "contig");
            // tiles a workgroup takes side by side: k_rtz3's four waves own a tile each, k_rtz3c's eight; k_rtzw's waves share one
            e->rtz3_quad = rtz3_quad(e->mt, e->dp, e->nblk, e->Kp, e->allow_rtz_bf16);
            const int quad = rtz_wide_ok(e->mt, e->dp) ? 1 : e->rtz3_quad;
            for (int g = 0; g < e->G; ++g) {
                const int ts = tstart[g], te = tstart[g + 1];
                if (te <= ts) continue;
                const int m = (te - ts + CH3 - 1) / CH3, per = (te - ts + m - 1) / m;
                if (interleave) {
                    const int mm = std::min(m, (te - ts + quad - 1) / quad);       // no task without a tile
                    for (int j = 0; j < mm; ++j) {
                        a0.push_back(ts + quad * j); a1.push_back(te); ag.push_back(g); ast.push_back(quad * mm);
                        ac0.push_back(gs2[g] + quad * j * HMX_TILE); acend.push_back(gs2[g + 1]);
                    }
                    continue;
                }
                for (int i = ts; i < te; i += per) {
                    a0.push_back(i); a1.push_back(std::min(i + per, te)); ag.push_back(g); ast.push_back(quad);
                    ac0.push_back(gs2[g] + (i - ts) * HMX_TILE); acend.push_back(gs2[g + 1]);
                }
            }
            e->ntasks3 = (int)a0.size();
            const size_t nt3 = a0.size();
            if ((rc = e->t3_t0.reserve(nt3)) || (rc = e->t3_t1.reserve(nt3)) || (rc = e->t3_c0.reserve(nt3)) || (rc = e->t3_cend.reserve(nt3)) || (rc = e->t3_stride.reserve(nt3)) ||
                (rc = e->t3_grp.reserve(nt3)) || (rc = e->s_tile_start.reserve(e->G + 1)) ||
                (rc = e->tile_blk[0].reserve((size_t)n_static_tiles * HMX_TILE)) || (rc = e->tile_blk[1].reserve((size_t)n_static_tiles * HMX_TILE)) ||
                (rc = e->tile_blk_zero.reserve((size_t)n_static_tiles * HMX_TILE)))
                return rc;
            HIP_TRY(hipMemcpyAsync(e->t3_t0.p, a0.data(), nt3 * sizeof(int), hipMemcpyHostToDevice, e->stream));
            HIP_TRY(hipMemcpyAsync(e->t3_t1.p, a1.data(), nt3 * sizeof(int), hipMemcpyHostToDevice, e->stream));
            HIP_TRY(hipMemcpyAsync(e->t3_c0.p, ac0.data(), nt3 * sizeof(int), hipMemcpyHostToDevice, e->stream));
            HIP_TRY(hipMemcpyAsync(e->t3_stride.p, ast.data(), nt3 * sizeof(int), hipMemcpyHostToDevice, e->stream));
            HIP_TRY(hipMemcpyAsync(e->t3_cend.p, acend.data(), nt3 * sizeof(int), hipMemcpyHostToDevice, e->stream));
            HIP_TRY(hipMemcpyAsync(e->t3_grp.p, ag.data(), nt3 * sizeof(int), hipMemcpyHostToDevice, e->stream));
            HIP_TRY(hipMemcpyAsync(e->s_tile_start.p, tstart.data(), tstart.size() * sizeof(int), hipMemcpyHostToDevice, e->stream));
            HIP_TRY(hipMemsetAsync(e->tile_blk[0].p, 255, (size_t)n_static_tiles * HMX_TILE, e->stream));   // 255: no block (static padding)
            HIP_TRY(hipMemsetAsync(e->tile_blk[1].p, 255, (size_t)n_static_tiles * HMX_TILE, e->stream));
            HIP_TRY(hipMemsetAsync(e->tile_blk_zero.p, 0, (size_t)n_static_tiles * HMX_TILE, e->stream));
            // the slack rows are read (and discarded) by the last tile of the last group: keep them finite
            HIP_TRY(hipMemsetAsync(e->R.p + (size_t)e->N * e->Kp, 0, (size_t)16 * e->Kp * sizeof(float), e->stream));
            HIP_TRY(hipMemsetAsync(e->Zorig.p + (size_t)e->N * e->dp, 0, (size_t)16 * e->dp * sizeof(float), e->stream));
            HIP_TRY(hipMemsetAsync(e->Zcos.p + (size_t)e->N * e->dp, 0, (size_t)16 * e->dp * sizeof(float), e->stream));
            HIP_TRY(hipStreamSynchronize(e->stream));
        }
    }
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipGetLastError());
    e->uploaded = true;
    return HMX_OK;
}

int hmx_compute_lisi(int32_t device_id, const double* X, int64_t n, int32_t d, const int32_t* label_codes, int32_t n_labels,
                     double perplexity, double* lisi_out, double* knn_dist_out, int32_t* knn_idx_out) {
    if (!X || !label_codes || !lisi_out) return fail(HMX_ERR_ARG, "null argument");
    if ((knn_dist_out == nullptr) != (knn_idx_out == nullptr)) return fail(HMX_ERR_ARG, "knn_dist_out and knn_idx_out go together");
    if (n < 1 || n > (int64_t)1 << 31) return fail(HMX_ERR_ARG, "n out of range");
    if (d < 1 || d > 208) return fail(HMX_ERR_ARG, "d must be in [1, 208]");
    if (n_labels < 1) return fail(HMX_ERR_ARG, "n_labels must be >= 1");
    const int nn = (int)(perplexity * 3);                                     // lisi.py:53
    // the float32 pass keeps the best cap / 2 candidates of a query for the exact float64 ranking: 8 of them are slack for
    // rank inversions of the approximation at the boundary; three list sizes (lisi.py:53 itself takes any perplexity)
    const int cap = (perplexity > 0 && nn >= 2) ? lisi_list_cap(nn) : 0;
    if (cap == 0)
        return fail(HMX_ERR_ARG, "perplexity: 3*perplexity must lie in [2, %d] neighbours in this build (got %d)", LISI_MAX_NEIGHBOURS, nn);
    if (nn > n) return fail(HMX_ERR_ARG, "Expected n_neighbors <= n_samples_fit, but n_neighbors = %d, n_samples_fit = %lld", nn, (long long)n);
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device_id < 0 || device_id >= ndev) return fail(HMX_ERR_ARG, "device %d not present (%d devices)", device_id, ndev);
    HIP_TRY(hipSetDevice(device_id));
    const int dp = (d + 15) & ~15;
    const int64_t npad = (n + 255) & ~(int64_t)255;
    const int M = nn - 1;
    DevBuf<double> X64, sums, out, kd;
    DevBuf<float> X32, cn;
    DevBuf<unsigned long long> lists;
    DevBuf<int> counts, labels, ki;
    struct Release {
        DevBuf<double>&a, &b, &c, &d2; DevBuf<float>&f, &g; DevBuf<unsigned long long>&h; DevBuf<int>&i, &j, &k;
        ~Release() { a.release(); b.release(); c.release(); d2.release(); f.release(); g.release(); h.release(); i.release(); j.release(); k.release(); }
    } guard{X64, sums, out, kd, X32, cn, lists, counts, labels, ki};
    int rc;
    if ((rc = X64.reserve((size_t)n * d)) || (rc = sums.reserve(d)) || (rc = out.reserve((size_t)n * n_labels)) ||
        (rc = X32.reserve((size_t)npad * dp)) || (rc = cn.reserve(npad)) || (rc = lists.reserve((size_t)npad * cap)) ||
        (rc = counts.reserve(n)) || (rc = labels.reserve((size_t)n * n_labels)))
        return rc;
    if (knn_dist_out && ((rc = kd.reserve((size_t)n * M)) || (rc = ki.reserve((size_t)n * M)))) return rc;
    hipStream_t s = nullptr;                                                  // the device's default stream
    HIP_TRY(hipMemcpyAsync(X64.p, X, (size_t)n * d * sizeof(double), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(labels.p, label_codes, (size_t)n * n_labels * sizeof(int), hipMemcpyHostToDevice, s));
    launch_lisi_prepare(X64.p, n, npad, d, dp, sums.p, X32.p, cn.p, s);
    LisiKnnArgs ka{};
    ka.X = X32.p; ka.cn = cn.p; ka.n = n; ka.npad = npad; ka.dp = dp; ka.lists = lists.p; ka.counts = counts.p; ka.cap = cap;
#ifdef LISI_PROF
    DevBuf<unsigned long long> prof;
    if ((rc = prof.reserve(8))) return rc;
    HIP_TRY(hipMemsetAsync(prof.p, 0, 8 * sizeof(unsigned long long), s));
    ka.prof = prof.p;
#endif
    if (launch_lisi_knn(ka, s)) return fail(HMX_ERR_ARG, "unsupported d");
#ifdef LISI_PROF
    {
        unsigned long long h[8];
        HIP_TRY(hipMemcpy(h, prof.p, sizeof h, hipMemcpyDeviceToHost));
        const double waves = (double)npad / 64.0, tiles = (double)((n + 15) / 16);
        fprintf(stderr, "[lisi prof] cycles per tile per wave: loads-issue %.0f, fragments+MFMA %.0f, store pieces (load wait) %.0f, append %.0f, barrier %.0f\n",
                h[0] / waves / tiles, h[1] / waves / tiles, h[2] / waves / tiles, h[3] / waves / tiles, h[4] / waves / tiles);
        prof.release();
    }
#endif
    LisiFinishArgs fa{};
    fa.X = X64.p; fa.n = n; fa.d = d; fa.nn = nn; fa.n_labels = n_labels; fa.lists = lists.p; fa.counts = counts.p; fa.cap = cap;
    fa.labels = labels.p; fa.perplexity = perplexity; fa.tol = 1e-5;         // lisi.py:75
    fa.out = out.p; fa.knn_dist = kd.p; fa.knn_idx = ki.p;
    launch_lisi_finish(fa, s);
    HIP_TRY(hipMemcpyAsync(lisi_out, out.p, (size_t)n * n_labels * sizeof(double), hipMemcpyDeviceToHost, s));
    if (knn_dist_out) {
        HIP_TRY(hipMemcpyAsync(knn_dist_out, kd.p, (size_t)n * M * sizeof(double), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(knn_idx_out, ki.p, (size_t)n * M * sizeof(int), hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(hipStreamSynchronize(s));
    HIP_TRY(hipGetLastError());
    return HMX_OK;
}

int hmx_kmeans_seed(hmx_engine* e, const float* points, int64_t n_points, uint64_t seed, float* centers_out, int32_t* chosen_out) {
    if (!e || !points || !centers_out) return fail(HMX_ERR_ARG, "null argument");
    if (n_points < 1 || n_points > (int64_t)1 << 24) return fail(HMX_ERR_ARG, "n_points must be in [1, 2^24]");
    int rc;
    if ((rc = use_device(e))) return rc;
    const int n = (int)n_points, d = e->d, K = e->K;
    const int nchunks = (n + 255) / 256;
    // one allocation for all temporaries (hipMalloc / hipFree synchronise the device: once, not nine times)
    auto up = [](size_t bytes) { return (bytes + 255) & ~(size_t)255; };
    const size_t bX = up((size_t)n * d * 4), bClosest = up((size_t)n * 4), bMin = up((size_t)HMX_SEED_SLOTS * n * 4),
                 bCenters = up((size_t)K * d * 4), bChunk = up((size_t)nchunks * 8), bPots = up((size_t)K * HMX_SEED_SLOTS * 8),
                 bCand = up((size_t)K * HMX_SEED_SLOTS * 4), bChosen = up((size_t)K * 4);
    DevBuf<unsigned char> arena;
    struct Release { DevBuf<unsigned char>& a; ~Release() { a.release(); } } guard{arena};
    if ((rc = arena.reserve(2 * bX + bClosest + bMin + bCenters + bChunk + bPots + bCand + bChosen))) return rc;
    unsigned char* cur = arena.p;
    auto take = [&](size_t bytes) { unsigned char* p = cur; cur += bytes; return p; };
    struct { float* p; } X{(float*)take(bX)}, Xt{(float*)take(bX)}, closest{(float*)take(bClosest)}, cand_min{(float*)take(bMin)},
        centers{(float*)take(bCenters)};
    struct { unsigned long long* p; } chunk_sum{(unsigned long long*)take(bChunk)}, pots{(unsigned long long*)take(bPots)};
    struct { int* p; } cand{(int*)take(bCand)}, chosen{(int*)take(bChosen)};
    HIP_TRY(hipMemcpyAsync(X.p, points, (size_t)n * d * sizeof(float), hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipMemsetAsync(pots.p, 0, bPots + bCand, e->stream));              // pots and cand are neighbours
    SeedArgs a{};
    a.X = X.p; a.Xt = Xt.p; a.n = n; a.d = d;
    a.n_trials = std::min(HMX_SEED_SLOTS, 2 + (int)std::log((double)K));       // sklearn: 2 + int(log(n_clusters))
    a.seed = seed; a.closest = closest.p; a.cand_min = cand_min.p; a.chunk_sum = chunk_sum.p; a.pots = pots.p;
    a.cand = cand.p; a.chosen = chosen.p; a.centers = centers.p;
    launch_kmeans_seed(a, K, e->stream);
    HIP_TRY(hipMemcpyAsync(centers_out, centers.p, (size_t)K * d * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    if (chosen_out) HIP_TRY(hipMemcpyAsync(chosen_out, chosen.p, (size_t)K * sizeof(int), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipGetLastError());
    return HMX_OK;
}

static int lloyd_wide(hmx_engine* e, const float* centers_in, int n_iter, float* centers_out);

// 1 when hmx_kmeans_lloyd serves this engine's shape and layout, 0 when it would refuse (wide shapes whose static tiles do not
// hold consecutive cells, or with more batch groups than the finish kernel of the streaming pass tabulates): a caller --
// every rank of a sharded job, BEFORE any of them enters the iterations' collectives -- asks first and falls back together.
int hmx_can_lloyd(hmx_engine* e) {
    if (!e) return fail(HMX_ERR_ARG, "null argument");
    if (!e->uploaded) return fail(HMX_ERR_STATE, "hmx_upload must come first");
    if (e->mt > 7 || e->dp > 64) return (e->static_contig && e->ntasks3 > 0 && rtzw_ok(e->mt, e->dp, e->d, 1, e->G)) ? 1 : 0;
    return 1;
}

int hmx_kmeans_lloyd(hmx_engine* e, const float* centers_in, int n_iter, float* centers_out) {
    if (!e || !centers_in || !centers_out) return fail(HMX_ERR_ARG, "null argument");
    if (!e->uploaded) return fail(HMX_ERR_STATE, "hmx_upload must come first");
    if (n_iter < 0) return fail(HMX_ERR_ARG, "n_iter must be >= 0");
    if (e->mt > 7 || e->dp > 64) return lloyd_wide(e, centers_in, n_iter, centers_out);
    int rc;
    if ((rc = use_device(e))) return rc;
    const size_t nsum = (size_t)e->K16 * (e->d + 1);
    const int wgs = std::min(e->n_cus, std::max(1, (e->n_s_tiles + 7) / 8));   // one workgroup of 8 waves per CU (86 KB of LDS)
    if ((rc = e->km_hn.reserve(e->K16)) || (rc = e->km_sums.reserve(nsum)) ||
        (rc = e->slab.reserve(kmeans_slab_floats(wgs, e->K16, e->dp))))
        return rc;
    std::vector<float> y((size_t)e->K16 * e->ldy, 0.f);
    for (int k = 0; k < e->K; ++k) std::memcpy(&y[(size_t)k * e->ldy], centers_in + (size_t)k * e->d, sizeof(float) * e->d);
    HIP_TRY(hipMemcpyAsync(e->Yacc.p, y.data(), y.size() * sizeof(float), hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipMemsetAsync(e->km_sums.p, 0, nsum * sizeof(double), e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    launch_kmeans_update(e->km_sums.p, e->Yacc.p, e->km_hn.p, e->K, e->K16, e->d, e->ldy, e->stream);   // counts 0: norms only
    for (int it = 0; it < n_iter; ++it) {
        if (launch_kmeans_step(e->Zcos.p, e->Yacc.p, e->km_hn.p, e->s_cells.p, e->n_s_tiles, e->slab.p, e->K, e->K16, e->dp,
                               e->ldy, wgs, e->stream))
            return fail(HMX_ERR_ARG, "unsupported shape for the device k-means");
        launch_kmeans_sums(e->slab.p, wgs, e->K16, e->dp, e->d, e->km_sums.p, e->stream);
        if ((rc = sum_over_ranks(e, e->km_sums.p, nsum))) return rc;
        launch_kmeans_update(e->km_sums.p, e->Yacc.p, e->km_hn.p, e->K, e->K16, e->d, e->ldy, e->stream);
    }
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy2D(centers_out, (size_t)e->d * 4, e->Yacc.p, (size_t)e->ldy * 4, (size_t)e->d * 4, e->K, hipMemcpyDeviceToHost));
    return HMX_OK;
}

// An error return in the middle of a cluster() call may leave the sticky time-out word set (a sweep gave up, then a HIP call
// failed before the replay cleared it): every later kernel would return at once and a "replay" would mix two rounds' state.
// Called on every error return of hmx_cluster and at the head of hmx_init_cluster: drain, clear the word, forget what was
// prepared ahead, defer no read-back before a sweep has been seen to complete.
static void thaw(hmx_engine* e) {
    (void)hipStreamSynchronize(e->stream);
    if (e->stream2) (void)hipStreamSynchronize(e->stream2);
    (void)hipMemsetAsync(e->frozen(), 0, sizeof(unsigned long long), e->stream);
    e->pre_valid = false;
    e->pre_outstanding = false;
    e->clean_sweeps = 0;
}

int hmx_init_cluster(hmx_engine* e, const float* Y0, double obj_out[4]) {
    if (!e || !Y0 || !obj_out) return fail(HMX_ERR_ARG, "null argument");
    if (!e->uploaded) return fail(HMX_ERR_STATE, "hmx_upload must come first");
    int rc;
    if ((rc = use_device(e))) return rc;
    thaw(e);
    std::vector<float> y((size_t)e->K16 * e->ldy, 0.f);
    for (int k = 0; k < e->K; ++k) std::memcpy(&y[(size_t)k * e->ldy], Y0 + (size_t)k * e->d, sizeof(float) * e->d);
    HIP_TRY(hipMemcpyAsync(e->Yacc.p, y.data(), y.size() * sizeof(float), hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    launch_y_normalize(e->Yacc.p, e->Y.p, e->K, e->K16, e->d, e->ldy, e->stream);  // :377
    const size_t GK = (size_t)e->G * e->K16;
    HIP_TRY(hipMemsetAsync(e->Ogrp.p, 0, GK * sizeof(double), e->stream));
    HIP_TRY(hipMemsetAsync(e->objacc, 0, (2 * HMX_OBJ_SLOTS + 2) * sizeof(double), e->stream));
    {
        Timed t(e, F_ASSIGN_INIT);
        AssignArgs a = assign_args(e);
        a.cells = e->s_cells.p; a.tile_grp = e->s_tile_grp.p; a.S_out = e->Ogrp.p;  // O = R Phi^T exactly (:389)
        a.tile_begin = 0; a.tile_end = e->n_s_tiles;
        if (launch_assign(a, false, e->max_wgs, e->stream) < 0) return fail(HMX_ERR_ARG, "unsupported cluster count");
    }
    if ((rc = sum_over_ranks(e, e->Ogrp.p, GK)) || (rc = sum_over_ranks(e, e->objacc, 2 * HMX_OBJ_SLOTS))) return rc;
    {
        Timed t(e, F_BLOCK_TABLE);
        TableArgs ta = table_args(e);  // E = outer(R.sum(1), Pr_b) (:388) kept as T; cross-entropy term (:405-411)
        ta.O_prev = e->Ogrp.p; ta.T_out = e->Tmass.p; ta.obj_cross = e->objacc + 2 * HMX_OBJ_SLOTS;
        launch_block_table(ta, e->K16, e->stream);
    }
    e->clustered = true;
    return read_objective(e, obj_out);
}

static bool use_rtz3(const hmx_engine* e) {
    return e->rtz_kernel == 3 && e->static_contig && e->ntasks3 > 0 && rtz3_ok(e->mt, e->dp, e->nblk, e->G);
}
// the same pass for wide shapes (K > 112 or d > 64): k_rtzw, eight waves sharing every tile
static bool use_rtzw(const hmx_engine* e) {
    return e->rtz_kernel == 3 && e->static_contig && e->ntasks3 > 0 && rtzw_ok(e->mt, e->dp, e->d, e->nblk, e->G);
}
static bool streaming_rtz(const hmx_engine* e) { return use_rtz3(e) || use_rtzw(e); }

// What the sweep kernel needs done before it starts, folded into k_rtz3_finish when a single engine runs the fused
// path: the slot tables + sync words and the objective accumulators zeroed, O at the start of the round kept aside.
struct Rtz3Duties { bool on = false; };

// The R^T.Z pass in storage order (k_rtz3 + k_rtz3_finish).
//   mode 0: Z_cos; block ids `tile_blk` with `nblk_cols` one-hot columns -> Yacc64 (centroid numerators, :443),
//           Sold (removal sums of every block, :491-492) and, with `normalize`, Y (:444);
//   mode 1: Z_orig, all ids 0 -> Sr (ridge right-hand sides, :556-563), Oxr (exact O, :550);
//   mode 2: the same statistics of Z_cos (member sums and counts of the device k-means' hard assignment).
static int rtz3_pass(hmx_engine* e, int mode, const unsigned char* tile_blk, int nblk_cols, bool normalize, bool duties) {
    int rc;
    const bool wide = !(use_rtz3(e) && rtz3_ok(e->mt, e->dp, nblk_cols, e->G));   // (mode 2 comes here for wide shapes whatever the rounds' kernel is)
    if ((rc = e->slab.reserve((size_t)e->ntasks3 * (wide ? rtzw_slab_floats(e->mt, e->dp, e->d, nblk_cols) : rtz3_slab_floats(e->mt, e->dp, nblk_cols)))))
        return rc;
    {
        Timed t(e, mode == 1 ? F_RIDGE_STATS : F_RTZ_ROUND);
        Rtz3Args r{};
        r.R = e->R.p; r.Z = mode == 1 ? e->Zorig.p : e->Zcos.p; r.tile_blk = tile_blk;
        r.task_t0 = e->t3_t0.p; r.task_t1 = e->t3_t1.p; r.task_stride = e->t3_stride.p; r.task_c0 = e->t3_c0.p; r.task_cend = e->t3_cend.p;
        r.slab = e->slab.p; r.ntasks = e->ntasks3; r.Kp = e->Kp;
        r.frozen = duties ? e->frozen() : nullptr;   // (the fused round of a single engine: the only path whose read-back is deferred)
        if (wide && mode != 1 && e->allow_zcf && e->allow_rtz_bf16 && e->static_contig && rtzw2b_ok(e->mt, e->dp, e->d, nblk_cols) && rtzw2b_zf_ok(e->mt, e->dp)) {
            // Z_cos is constant between two ridge steps: its three bf16 planes are split once, in k_rtzw2b's fragment order
            if (!e->zcf_valid) {
                if ((rc = e->Zcf.reserve((size_t)e->n_s_tiles * rtzw_zf_tile_words(e->dp)))) return rc;
                launch_zplanes(e->Zcos.p, e->dp, e->n_s_tiles, e->s_tile_grp.p, e->gstart.p, e->s_tile_start.p, e->Zcf.p, e->stream);
                e->zcf_valid = true;
            }
            r.Zf = e->Zcf.p;
            e->n_rtz_zf++;
        }
        const int lr = wide ? launch_rtzw(r, e->mt, e->dp, e->d, nblk_cols, e->stream, e->allow_rtz_bf16) : launch_rtz3(r, e->mt, e->dp, nblk_cols, e->stream, e->allow_rtz_bf16, e->rtz3_quad);
        if (lr > 0) e->n_rtz_bf16++;
        if (lr < 0)
            return fail(HMX_ERR_ARG, "unsupported shape for the streaming R^T.Z pass");
    }
    Timed t(e, mode == 1 ? F_RIDGE_STATS : F_RTZ_REDUCE);
    const size_t GK = (size_t)e->G * e->K16;
    Rtz3FinishArgs f{};
    f.slab = e->slab.p; f.task_grp = e->t3_grp.p; f.ntasks = e->ntasks3;
    f.MT = e->mt; f.KS = e->dp / 4; f.NTB = rtz3_ntb(e->dp, nblk_cols);
    f.wide = wide ? 1 : 0; f.NT = wide ? rtzw_nt(e->dp, e->d, nblk_cols) : 4 + f.NTB;
    f.K = e->K; f.K16 = e->K16; f.d = e->d; f.ld = e->ldy; f.G = e->G; f.nblk = nblk_cols; f.mode = mode == 0 ? 0 : 1;
    f.Ysum = e->Yacc64; f.Yout = normalize ? e->Y.p : nullptr; f.Sold = e->Sold; f.Sr = e->Sr; f.Oxr = e->Oxr;
    if (duties) {
        f.frozen = e->frozen();
        f.zero_p = e->Sslots.p; f.zero_n = (GK + e->K16) * (e->nblk + 1) * HMX_ROUND_SLOTS + 1 + 160;   // slot tables + the two sync words + the flag replicas behind them (k_round, group-affine map)
        f.zero2_p = e->objacc; f.zero2_n = 2 * HMX_OBJ_SLOTS + 2;
        f.copy_src = e->Ogrp.p; f.copy_dst = e->Osave.p; f.copy_n = (int)GK;
    }
    launch_rtz3_finish(f, e->stream);
    return 0;
}

// Centroid numerators sum_cells R (x) Z_cos (harmony.py:443) of this rank's cells into Yacc64, by a pass over the
// static list (no removal sums).
static int centroid_pass(hmx_engine* e) {
    if (streaming_rtz(e)) return rtz3_pass(e, 0, e->tile_blk_zero.p, 1, false, false);
    int rc, nsub, spw;
    rtz_geometry(e->mt, e->ntd, &nsub, &spw);
    const bool rtz2 = rtz2_ok(e->mt, e->dp);
    const bool rtzw = !rtz2 && rtz_wide_ok(e->mt, e->dp);
    const int n_tiles = e->n_s_tiles;
    const int wgs = rtz2 ? std::min(e->rtz_wgs_per_cu * e->n_cus, std::max(1, (n_tiles + 7) / 8))
                  : rtzw ? std::min(e->n_cus, std::max(1, (n_tiles + 7) / 8))
                         : std::min(256, std::max(1, (n_tiles + 31) / 32));
    if ((rc = e->slab.reserve(rtz2 ? (size_t)wgs * rtz2_slab_floats(e->mt, e->dp)
                              : rtzw ? (size_t)wgs * rtz_wide_slab_floats(e->mt, e->dp) : (size_t)wgs * 4 * spw)))
        return rc;
    HIP_TRY(hipMemsetAsync(e->Yacc64, 0, (size_t)e->K16 * e->ldy * sizeof(double), e->stream));
    {
        Timed t(e, F_RTZ_ROUND);
        RtzArgs r{};
        r.R = e->R.p; r.Z = e->Zcos.p; r.cells = e->s_cells.p; r.tile_grp = e->s_tile_grp.p;
        r.slab = e->slab.p; r.n_tiles = n_tiles; r.nblk = e->nblk;
        r.K = e->K; r.Kp = e->Kp; r.K16 = e->K16; r.G = e->G; r.mt = e->mt; r.dp = e->dp; r.ntd = e->ntd;
        if (rtz2) launch_rtz2(r, wgs, e->stream);
        else if (rtzw) launch_rtz_wide(r, wgs, e->stream);
        else launch_rtz(r, wgs, e->stream);
    }
    Timed t(e, F_RTZ_REDUCE);
    if (rtz2) launch_rtz2_reduce(e->slab.p, wgs, e->mt, e->dp, e->K16, e->ldy, e->Yacc64, nullptr, e->stream);
    else if (rtzw) launch_rtz_wide_reduce(e->slab.p, wgs, e->mt, e->dp, e->K16, e->ldy, e->Yacc64, nullptr, e->stream);
    else launch_rtz_reduce(e->slab.p, wgs * 4, e->mt, e->ntd, e->K16, e->ldy, e->Yacc64, nullptr, e->stream);
    return 0;
}

// Lloyd iterations for shapes beyond k_kmeans_step (K > 112 or d > 64): per iteration the hard assignment of every cell
// (k_assign_wide writes a one-hot row of R -- R is free before init_cluster), then the member sums and counts as the
// R^T.Z statistics of that assignment (the streaming pass over Z_cos, summed over ranks like the ridge statistics).
static int lloyd_wide(hmx_engine* e, const float* centers_in, int n_iter, float* centers_out) {
    int rc;
    // the member sums are ONE block column of the streaming pass (k_rtzw with all block ids 0), whatever kernel the rounds use
    // (HMX_RTZ=2) and however many update blocks they have: only the layout and the finish kernel's per-group table matter
    if (!(e->static_contig && e->ntasks3 > 0 && rtzw_ok(e->mt, e->dp, e->d, 1, e->G)))
        return fail(HMX_ERR_ARG, "device k-means for K > 112 or d > 64 needs consecutive cells in every static tile and at most %d batch groups at this width "
                                 "(the caller falls back to Lloyd iterations on a subsample)", (int)(150 * 1024 / (16 * rtzw_nt(e->dp, e->d, 1) * sizeof(double))));
    if ((rc = use_device(e))) return rc;
    e->clustered = false;   // R serves as scratch: whatever assignment the engine held is void until hmx_init_cluster runs (again)
    const size_t nsum = (size_t)e->K16 * (e->d + 1), GK = (size_t)e->G * e->K16;
    if ((rc = e->km_hn.reserve(e->K16)) || (rc = e->km_sums.reserve(nsum))) return rc;
    std::vector<float> y((size_t)e->K16 * e->ldy, 0.f);
    for (int k = 0; k < e->K; ++k) std::memcpy(&y[(size_t)k * e->ldy], centers_in + (size_t)k * e->d, sizeof(float) * e->d);
    HIP_TRY(hipMemcpyAsync(e->Yacc.p, y.data(), y.size() * sizeof(float), hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipMemsetAsync(e->km_sums.p, 0, nsum * sizeof(double), e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    launch_kmeans_update(e->km_sums.p, e->Yacc.p, e->km_hn.p, e->K, e->K16, e->d, e->ldy, e->stream);   // counts 0: norms only
    for (int it = 0; it < n_iter; ++it) {
        AssignArgs a = assign_args(e);
        a.Y = e->Yacc.p; a.hn = e->km_hn.p;
        a.cells = e->s_cells.p; a.tile_grp = e->s_tile_grp.p; a.S_out = e->Ogrp.p;
        a.tile_begin = 0; a.tile_end = e->n_s_tiles;
        if (launch_assign(a, false, e->max_wgs, e->stream) < 0) return fail(HMX_ERR_ARG, "unsupported cluster count");
        if ((rc = rtz3_pass(e, 2, e->tile_blk_zero.p, 1, false, false))) return rc;
        if ((rc = sum_over_ranks(e, e->Sr, GK * e->ldy + GK))) return rc;   // Sr and Oxr are neighbours
        launch_kmeans_sums_from_stats(e->Sr, e->Oxr, e->G, e->K16, e->ldy, e->d, e->km_sums.p, e->stream);
        launch_kmeans_update(e->km_sums.p, e->Yacc.p, e->km_hn.p, e->K, e->K16, e->d, e->ldy, e->stream);
    }
    HIP_TRY(hipMemsetAsync(e->R.p, 0, (size_t)e->N * e->Kp * sizeof(float), e->stream));   // leave R as hmx_create made it
    HIP_TRY(hipMemsetAsync(e->Ogrp.p, 0, GK * sizeof(double), e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy2D(centers_out, (size_t)e->d * 4, e->Yacc.p, (size_t)e->ldy * 4, (size_t)e->d * 4, e->K, hipMemcpyDeviceToHost));
    return HMX_OK;
}

static void note_sweep_timeout(hmx_engine* e) {
    if (e->n_sweep_fallbacks++ == 0)
        fprintf(stderr, "[hmx] rank %d: a grid-wide wait of the sweep kernel timed out; the round is repeated with one launch per "
                        "block (HMX_ROUND_MODE=blocks avoids the persistent kernel altogether)\n", e->rank);
    if (e->n_sweep_fallbacks == 2 && e->round_mode == 1) {
        // every time-out costs the whole spin budget of the launch: a GPU that cannot keep the grid resident will not
        // start to.  All ranks count the same fall-backs (the failure travels in the all-reduced objective block).
        e->round_mode = 0;
        fprintf(stderr, "[hmx] rank %d: second time-out: this engine stays on the per-block path\n", e->rank);
    }
}


// update_R block by block (harmony.py:476-507): per block the diversity table (k_block_table), the assignment of the
// block's tiles (bounded launch) and the block's new sums over all ranks; closes O, T and the cross-entropy term.
// Needs Sold (removal sums) and Y of this round; Snew and objacc zeroed by the caller.
// Will blocks_loop run the wide sweep as one persistent launch (k_sweep_wide3)?  Single engine, the fused-table instance of the wide
// bf16-pipe assignment, every block with tiles, lists with their (block, group) run offsets, at most 511 chunks of sixteen tiles
// per block (the count field of the hand-off words).
static bool wide_sweep_planned(hmx_engine* e, const std::vector<int>& tiles_upper) {
    if (!e->wide_sweep || sharded(e) || !e->lists[e->cur].runs_ok || (int)tiles_upper.size() < e->nblk) return false;
    if (!(e->allow_round_bf16 && rtz_wide_ok(e->mt, e->dp) && e->fuse_block_table && assign_wide3_fuses_table(e->mt, e->dp, e->V))) return false;
    for (int b = 0; b < e->nblk; ++b)
        if (tiles_upper[b] <= 0 || tiles_upper[b] > 511 * 16) return false;
    return sweep_wide3_ok(e->mt, e->dp, e->V, e->G, e->nblk);
}

static int blocks_loop(hmx_engine* e, int flags, const std::vector<int>& tiles_upper, bool allow_sweep = true) {
    int rc;
    const size_t GK = (size_t)e->G * e->K16;
    e->wide_sweep_launched = false;
    int bf16_blocks = 0;                                            // blocks assigned by the bf16-pipe instance of the wide kernel
    const bool y_frags = e->allow_round_bf16 && rtz_wide_ok(e->mt, e->dp);
    if (y_frags) {                                                  // Y of this round, split once into the fragments k_assign_wide3 multiplies with
        if ((rc = e->Yf.reserve(y_planes_dwords(e->K16, e->dp)))) return rc;
        launch_y_planes(e->Y.p, e->K16, e->ldy, e->dp, e->Yf.p, e->stream);
    }
    // the wide bf16-pipe assignment builds the block's table in its own prologue (one batch variable): 20 launches per sweep fewer
    const bool fuse = y_frags && e->fuse_block_table && assign_wide3_fuses_table(e->mt, e->dp, e->V);
    // ... and on a single engine the whole sweep is ONE persistent launch (k_sweep_wide3): every block has tiles, the lists carry
    // their (block, group) run offsets, a (block, group) pair is held by at most 511 chunks of sixteen tiles (the count field of the
    // hand-off words).  A launch whose waits gave up is replayed block by block (round_body), the second one retires the path.
    int max_upper = 0;
    for (int b = 0; b < e->nblk; ++b) max_upper = std::max(max_upper, tiles_upper[b]);
    if (allow_sweep && wide_sweep_planned(e, tiles_upper)) {
        HIP_TRY(hipMemcpyAsync(e->Osave.p, e->Ogrp.p, GK * sizeof(double), hipMemcpyDeviceToDevice, e->stream));   // for an exact replay
        {
            Timed t(e, F_ASSIGN_BLOCK);
            AssignArgs a = assign_args(e);
            a.cells = e->lists[e->cur].cells.p; a.tile_grp = e->lists[e->cur].tile_grp.p; a.blk_start = e->lists[e->cur].blk_start.p;
            a.run_tiles = e->lists[e->cur].run_tiles.p; a.nblk = e->nblk;
            a.S_out = e->Snew; a.S_sub = e->Sold; a.O_prev = e->Ogrp.p; a.O_out = e->Ohist.p; a.Pr_b = e->Pr_b.p; a.theta = e->theta.p;
            a.fuse_table = 1; a.Yf = e->Yf.p;
            a.spin_limit = e->spin_limit;
            a.fail = e->objacc + 2 * HMX_OBJ_SLOTS + 1;
            if (e->n_sweep_launches++ == e->test_fail_sweep) a.spin_limit = 0;
            int wgs = std::min(std::max(1, e->n_cus - 8), (max_upper + 15) / 16);   // (a few CUs stay free for the second stream's launches)
            if (e->round_wgs_cap > 0) wgs = std::min(wgs, e->round_wgs_cap);
            if ((rc = e->Opriv.reserve((size_t)2 * wgs * GK))) return rc;
            a.O_priv = e->Opriv.p;
            if (launch_sweep_wide3(a, wgs, e->stream)) return fail(HMX_ERR_ARG, "unsupported shape for k_sweep_wide3");
            e->wide_sweep_launched = true;
            e->n_sweeps_wide++;
        }
        e->n_sweeps_bf16++;
        Timed t(e, F_BLOCK_TABLE);
        TableArgs ta = table_args(e);  // close the round: O, T state and the cross-entropy term
        ta.O_prev = e->Ohist.p + GK * (e->nblk - 1);
        ta.S_add = e->Snew + GK * (e->nblk - 1);
        ta.O_out = e->Ogrp.p; ta.T_out = e->Tmass.p;
        if (flags & HMX_ROUND_OBJECTIVE) ta.obj_cross = e->objacc + 2 * HMX_OBJ_SLOTS;
        launch_block_table(ta, e->K16, e->stream);
        return 0;
    }
    for (int b = 0; b < e->nblk; ++b) {
        const double* O_prev = (b == 0) ? e->Ogrp.p : e->Ohist.p + GK * (b - 1);
        const double* S_add = (b == 0) ? nullptr : e->Snew + GK * (b - 1);
        if (!fuse || tiles_upper[b] <= 0) {   // (a block without a tile on this rank: the chain of O still has to move on)
            Timed t(e, F_BLOCK_TABLE);
            TableArgs ta = table_args(e);
            ta.O_prev = O_prev; ta.S_add = S_add;
            ta.S_sub = e->Sold + GK * b;
            ta.O_out = e->Ohist.p + GK * b;
            ta.rp = e->rp.p; ta.lrp = e->lrp.p;
            launch_block_table(ta, e->K16, e->stream);
        }
        if (tiles_upper[b] > 0) {
            Timed t(e, F_ASSIGN_BLOCK);
            AssignArgs a = assign_args(e);
            a.cells = e->lists[e->cur].cells.p; a.tile_grp = e->lists[e->cur].tile_grp.p; a.S_out = e->Snew + GK * b;
            a.blk_start = e->lists[e->cur].blk_start.p; a.blk = b;
            a.tile_begin = 0; a.tile_end = tiles_upper[b];
            if (fuse) {
                a.fuse_table = 1; a.O_prev = O_prev; a.S_add = S_add; a.S_sub = e->Sold + GK * b; a.O_out = e->Ohist.p + GK * b;
                a.Pr_b = e->Pr_b.p; a.theta = e->theta.p;
            }
            if (y_frags) a.Yf = e->Yf.p;
            const int la = launch_assign(a, true, e->max_wgs, e->stream);
            if (la < 0) return fail(HMX_ERR_ARG, "unsupported cluster count");
            if (la > 0) bf16_blocks++;
        }
        // the block's new sums (:506-507) over all ranks; the last block takes the two objective
        // sums (:399, :402) along: objacc follows Snew in xch
        const bool last = b == e->nblk - 1;
        if ((rc = sum_over_ranks(e, e->Snew + GK * b, GK + (last ? 2 * HMX_OBJ_SLOTS : 0)))) return rc;
    }
    if (bf16_blocks > 0) e->n_sweeps_bf16++;
    Timed t(e, F_BLOCK_TABLE);
    TableArgs ta = table_args(e);  // close the round: O, T state and the cross-entropy term
    ta.O_prev = e->Ohist.p + GK * (e->nblk - 1);
    ta.S_add = e->Snew + GK * (e->nblk - 1);
    ta.O_out = e->Ogrp.p; ta.T_out = e->Tmass.p;
    if (flags & HMX_ROUND_OBJECTIVE) ta.obj_cross = e->objacc + 2 * HMX_OBJ_SLOTS;
    launch_block_table(ta, e->K16, e->stream);
    return 0;
}

// A grid-wide wait of the persistent sweep gave up: EXACT replay of the round, block by block, from the round's own start
// -- O as it was (Osave), the removal sums and centroids the failed launch used (Sold, Y: untouched by it), the round's own
// lists (lists[cur]).  Rows the failed launch already replaced are computed again: a new row depends on Z_cos, Y and its
// block's table, never on the old row.  The sticky word that makes later kernels stand still is cleared here (the stream
// is drained first: nothing that tests it may still be queued), and whatever was prepared ahead on the side stream is void.
static int replay_round(hmx_engine* e, int flags, const std::vector<int>& tiles_upper, double obj_out[4]) {
    int rc;
    const size_t GK = (size_t)e->G * e->K16;
    note_sweep_timeout(e);
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (e->stream2) HIP_TRY(hipStreamSynchronize(e->stream2));
    e->pre_valid = false;
    e->pre_outstanding = false;
    e->clean_sweeps = 0;        // no read-back is deferred again before a sweep has been seen to complete
    HIP_TRY(hipMemsetAsync(e->frozen(), 0, sizeof(unsigned long long), e->stream));
    HIP_TRY(hipMemcpyAsync(e->Ogrp.p, e->Osave.p, GK * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
    HIP_TRY(hipMemsetAsync(e->Snew, 0, (size_t)((e->objacc + 2 * HMX_OBJ_SLOTS + 2) - e->Snew) * sizeof(double), e->stream));
    if ((rc = blocks_loop(e, flags, tiles_upper, false))) return rc;
    return read_objective(e, obj_out);
}

// The group-affine tile map of k_round (one batch variable, one engine, at most `cap` compute workgroups): workgroup w owns
// ONE batch group; a group gets workgroups in proportion to the largest run of tiles it is expected to have in any block --
// its share of the largest block (`max_tiles` tiles) + 5 sigma of that count, padded to whole tiles -- at HMX_ROUND_GA_TILES =
// 14 tiles per workgroup and block (the workgroup's last wave, which also runs the hand-off, then seldom carries tiles), at
// all 16 slots when the grid cannot carry that.  When even that does not fit (blocks larger than the grid), all `cap`
// workgroups are dealt out so that the largest per-workgroup share is as small as possible and the kernel's extra-tile
// loop takes the rest (ga_extra).  A run that exceeds the estimate is served by the same loop: the map decides speed,
// never results.  It depends on the group sizes, the block size and the cap only: planned once, kept on the device.
static bool plan_ga(hmx_engine* e, int max_tiles, int cap) {
    if (!e->allow_round_ga || e->V != 1 || e->G > cap || (int)e->gsize.size() != e->G || e->N <= 0) return false;
    if ((int64_t)max_tiles * HMX_TILE >= HMX_ROUND_GA_MAX_BLOCK_CELLS) return false;   // the fixed-point words of the hand-off
    if (e->ga_nwg > 0 && e->ga_key_block == max_tiles && e->ga_key_cap == cap) return true;
    const int G = e->G;
    const double block_cells = 16.0 * std::max(1, max_tiles - G);      // the engine's cells in its largest block (upper estimate)
    std::vector<double> est(G);
    std::vector<int> ng(G);
    int total = 0;
    for (int per_wg : {HMX_ROUND_GA_TILES, 16}) {
        total = 0;
        e->ga_per_wg = per_wg;
        for (int g = 0; g < G; ++g) {
            const double mean = block_cells * (double)e->gsize[g] / (double)e->N;
            est[g] = std::ceil((mean + 5.0 * std::sqrt(mean)) / 16.0) + (e->gsize[g] > 0 ? 1.0 : 0.0);
            ng[g] = std::max(1, (int)std::ceil(est[g] / per_wg));
            total += ng[g];
        }
        if (total <= cap) break;
    }
    e->ga_extra = total > cap;
    if (e->ga_extra) {                                                  // min-max: one more workgroup to the most loaded group
        std::fill(ng.begin(), ng.end(), 1);
        for (total = G; total < cap; ++total) {
            int best = 0;
            for (int g = 1; g < G; ++g)
                if (est[g] / ng[g] > est[best] / ng[best]) best = g;
            ng[best]++;
        }
    }
    std::vector<int> map;
    map.reserve(3 * (size_t)total);
    for (int g = 0; g < G; ++g)
        for (int r = 0; r < ng[g]; ++r) { map.push_back(g); map.push_back(r); map.push_back(ng[g]); }
    if (e->ga_map.reserve(map.size())) return false;
    if (hipMemcpyAsync(e->ga_map.p, map.data(), map.size() * sizeof(int), hipMemcpyHostToDevice, e->stream) != hipSuccess) return false;
    if (hipStreamSynchronize(e->stream) != hipSuccess) return false;   // (the vector goes out of scope)
    e->ga_nwg = total; e->ga_key_block = max_tiles; e->ga_key_cap = cap;
    return true;
}

// Kernel sequence of one round; the lists (cells, tile groups, block_tile_start) and, for the streaming R^T.Z pass, the
// block ids in static tile order (tile_blk) are already in device memory.  tiles_upper[b] bounds the tile count of block b
// (grid sizing only).
static int round_body(hmx_engine* e, int flags, int n_tiles_upper, const std::vector<int>& tiles_upper, double obj_out[4],
                      const std::function<int(int)>& before_sweep = nullptr, double* defer_slot = nullptr, bool no_replay = false) {
    int rc;
    const size_t GK = (size_t)e->G * e->K16;
    const bool persistent = (flags & HMX_ROUND_UPDATE_R) && e->round_mode == 1 && (!sharded(e) || e->peers_enabled);
    // the sweep's tile map: group-affine for a single engine when the shape allows it and the lists carry their run offsets
    // (cells sharded over ranks: the classic map -- its exchange through the peer boxes is the one that has run on hardware)
    int max_upper = 0;
    for (int b = 0; b < e->nblk; ++b) max_upper = std::max(max_upper, tiles_upper[b]);
    const bool multi = e->peers_enabled && e->n_ranks > 1;
    int ga_cap = e->n_cus - (multi ? 1 : 0);
    if (e->round_wgs_cap > 0) ga_cap = std::min(ga_cap, e->round_wgs_cap);
    const bool ga = persistent && !sharded(e) && e->mt <= 7 && round_row_floats(e->d) == e->dp && e->lists[e->cur].runs_ok && plan_ga(e, max_upper, ga_cap);
    const bool mega = persistent && e->mt <= 7 && round_row_floats(e->d) == e->dp &&
                      round_lds_bytes(e->K16, e->dp, e->G, e->B, e->V, false, ga, e->nblk) <= HMX_ROUND_LDS_LIMIT;
    const bool r3 = streaming_rtz(e);
    // a single engine on the persistent sweep: k_rtz3_finish normalises the centroids itself (no collective is due in
    // between) and does the sweep kernel's fills -- three launches per round
    const bool fused = r3 && mega && !sharded(e);
    // Wide shapes: the next round's list build runs beside the R^T.Z PASS, whose workgroups leave registers free on every CU (k_rtzw2b:
    // 4 waves x 408), not beside the sweep: k_sweep_wide3 fills every register of its CUs (2 x 254 per SIMD lane) -- the scan kernel of
    // the build sat out the whole sweep and pushed the next pass back by 110 us -- and the per-block launches paid 1.2 ms per iteration
    // for their neighbour (profiles/r06_c5_timeline.txt).  Marked HERE, in front of the pass; the launches are enqueued behind the
    // pass's own -- host time the main stream does not wait for.
    const bool wsweep = (flags & HMX_ROUND_UPDATE_R) && !mega && r3 && rtz_wide_ok(e->mt, e->dp) && e->lists_beside_rtz;
    if (wsweep && before_sweep && (rc = before_sweep(0))) return rc;
    if (!fused)   // Sold | Yacc64 | Snew | objacc are neighbours in xch: one fill instead of four
        HIP_TRY(hipMemsetAsync(e->Sold, 0, (size_t)((e->objacc + 2 * HMX_OBJ_SLOTS + 2) - e->Sold) * sizeof(double), e->stream));

    // ---- pass over the old R: centroid numerators (:443) and per-block removal sums (:491-492)
    if (r3) {
        const bool upd = (flags & HMX_ROUND_UPDATE_R) != 0;
        if ((rc = rtz3_pass(e, 0, upd ? e->tile_blk[e->cur].p : e->tile_blk_zero.p, upd ? e->nblk : 1,
                            fused && (flags & HMX_ROUND_CENTROIDS), fused)))
            return rc;
    } else {
        int nsub, spw;
        rtz_geometry(e->mt, e->ntd, &nsub, &spw);
        const bool rtz2 = rtz2_ok(e->mt, e->dp) && e->round_mode == 1;
        const bool rtzw = !rtz2 && rtz_wide_ok(e->mt, e->dp);
        int wgs = rtz2 ? std::min(e->rtz_wgs_per_cu * e->n_cus, std::max(1, (n_tiles_upper + 7) / 8))
                  : rtzw ? std::min(e->n_cus, std::max(1, (n_tiles_upper + 7) / 8))
                         : std::min(256, std::max(1, (n_tiles_upper + 31) / 32));
        if ((rc = e->slab.reserve(rtz2 ? (size_t)wgs * rtz2_slab_floats(e->mt, e->dp)
                                  : rtzw ? (size_t)wgs * rtz_wide_slab_floats(e->mt, e->dp) : (size_t)wgs * 4 * spw)))
            return rc;
        {
            Timed t(e, F_RTZ_ROUND);
            RtzArgs r{};
            r.R = e->R.p; r.Z = e->Zcos.p; r.cells = e->lists[e->cur].cells.p; r.tile_grp = e->lists[e->cur].tile_grp.p; r.blk_start = e->lists[e->cur].blk_start.p;
            r.S_out = e->Sold; r.slab = e->slab.p; r.n_tiles = n_tiles_upper; r.nblk = e->nblk;
            r.K = e->K; r.Kp = e->Kp; r.K16 = e->K16; r.G = e->G; r.mt = e->mt; r.dp = e->dp; r.ntd = e->ntd;
            if (rtz2) launch_rtz2(r, wgs, e->stream);
#ifdef RTZ_PROF
            if (rtz2) { static int calls = 0; if (++calls % 40 == 0) rtz_prof_dump(); }
#endif
            else if (rtzw) launch_rtz_wide(r, wgs, e->stream);
            else launch_rtz(r, wgs, e->stream);
        }
        if (flags & HMX_ROUND_CENTROIDS) {
            Timed t(e, F_RTZ_REDUCE);
            if (rtz2) launch_rtz2_reduce(e->slab.p, wgs, e->mt, e->dp, e->K16, e->ldy, e->Yacc64, nullptr, e->stream);
            else if (rtzw) launch_rtz_wide_reduce(e->slab.p, wgs, e->mt, e->dp, e->K16, e->ldy, e->Yacc64, nullptr, e->stream);
            else launch_rtz_reduce(e->slab.p, wgs * 4, e->mt, e->ntd, e->K16, e->ldy, e->Yacc64, nullptr, e->stream);
        }
    }
    if (!fused) {
        // removal sums of every block and the centroid numerators: one collective (neighbours in xch)
        if ((rc = sum_over_ranks(e, e->Sold, GK * e->nblk + ((flags & HMX_ROUND_CENTROIDS) ? (size_t)e->K16 * e->ldy : 0)))) return rc;
        if (flags & HMX_ROUND_CENTROIDS) {
            Timed t(e, F_RTZ_REDUCE);
            launch_y_normalize_d(e->Yacc64, e->Y.p, e->K, e->K16, e->d, e->ldy, e->stream);  // :444
        }
    }
    // side-stream work that should run beside the sweep, not beside the R^T.Z pass: its start is MARKED in the main stream here (phase 0),
    // its launches are enqueued behind the sweep's own (phase 1) -- four launches of host time that the sweep does not wait for
    if (before_sweep && !wsweep && (rc = before_sweep(0))) return rc;
    if (before_sweep && !mega && (rc = before_sweep(1))) return rc;
    if (mega) {
        // the whole sweep in one persistent launch (k_round); closes O, T and the objective itself
        if (!fused) {
            // the slot tables and the two sync words (carved from the same allocation): one fill (size rounded up to 1 KB
            // inside the allocation: an odd size makes the runtime launch a second fill kernel for the tail); O at the
            // start of the round is kept for an exact replay
            HIP_TRY(hipMemsetAsync(e->Sslots.p, 0, ((((GK + e->K16) * (e->nblk + 1) * HMX_ROUND_SLOTS + 2 + 160) * sizeof(double) + 1023) / 1024) * 1024, e->stream));
            HIP_TRY(hipMemcpyAsync(e->Osave.p, e->Ogrp.p, GK * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
        }
        int wgs = std::min(e->n_cus - (multi ? 1 : 0), std::max(1, (max_upper + 13) / 14));   // ~14 of a workgroup's 16 tile slots: 224 of 256 CUs at C3, the rest serve the second stream
        // (sharded: every rank sizes its grid from its own share; the grid only decides how this rank's tiles are dealt out)
        if (e->round_wgs_cap > 0) wgs = std::min(wgs, e->round_wgs_cap);
        if (ga) wgs = e->ga_nwg;
        {
            Timed t(e, F_ASSIGN_BLOCK);
            RoundArgs ra{};
            ra.Zcos = e->Zcos.p; ra.Y = e->Y.p; ra.sigma = e->sigma.p; ra.R = e->R.p;
            ra.cells = e->lists[e->cur].cells.p; ra.tile_grp = e->lists[e->cur].tile_grp.p; ra.blk_start = e->lists[e->cur].blk_start.p;
            ra.O_start = e->Ogrp.p; ra.S_old = e->Sold; ra.S_new = e->Sslots.p; ra.O_out = e->Ogrp.p; ra.T_out = e->Tmass.p;
            ra.obj = e->objacc; ra.group_cols = e->group_cols.p; ra.Pr_b = e->Pr_b.p; ra.theta = e->theta.p;
            ra.counter = e->sync_words.p; ra.error = e->sync_words.p + 1; ra.wait_stats = e->wait_stats.p;
            ra.K = e->K; ra.Kp = e->Kp; ra.K16 = e->K16; ra.dp = e->dp; ra.ldy = e->ldy; ra.G = e->G; ra.B = e->B; ra.V = e->V;
            ra.nblk = e->nblk; ra.spin_limit = e->spin_limit;
            ra.frozen = e->frozen();
            if (ga) {
                // row-request placement of the group-affine sweep (k_round, tile_step): measured per shape, HMX_ROUND_REQ overrides
                ra.req_mode = e->ga_per_wg > HMX_ROUND_GA_TILES ? 1 : e->ga_nwg > 32 ? 2 : 0;
                ra.ga_slots = e->ga_nwg > 32 ? 2 : 1;
                if (const char* rq = getenv("HMX_ROUND_REQ")) ra.req_mode = std::max(0, std::min(2, atoi(rq)));
            }
            if (ga) { ra.ga = 1; ra.run_start = e->lists[e->cur].run_tiles.p; ra.wg_map = e->ga_map.p; e->n_sweeps_ga++; }
            if (e->n_sweep_launches++ == e->test_fail_sweep) ra.spin_limit = 0;
            if (multi) {
                ra.peer_box = e->peer_dev.p; ra.my_box = e->box; ra.n_ranks = e->n_ranks; ra.rank = e->rank;
                ra.epoch = e->round_epoch;
                e->round_epoch += 256;   // (> HMX_MAX_BLOCKS: the flag values of two launches never meet)
            }
#ifdef HMX_ROUND_PROF
            static DevBuf<unsigned long long> prof;
            static int prof_rounds = 0;
            if (prof.reserve((size_t)wgs * e->nblk * 32)) return -1;
            (void)hipMemsetAsync(prof.p, 0, (size_t)wgs * e->nblk * 32 * 8, e->stream);
            ra.prof = prof.p;
#endif
            const bool extra_tiles = ga ? e->ga_extra : max_upper > 16 * wgs;   // (ROUND_TPW x ROUND_WAVES slots per workgroup; group-affine: 14 per workgroup of the run's group)
            if (launch_round(ra, e->mt, multi ? wgs + 1 : wgs, e->stream, extra_tiles, e->allow_round_bf16)) return fail(HMX_ERR_ARG, "unsupported shape for k_round");
            if (before_sweep && (rc = before_sweep(1))) return rc;
            if (round_uses_bf16_pipe(ra.K16, ra.dp, ra.G, ra.B, ra.V, extra_tiles, e->allow_round_bf16, ga, e->nblk)) e->n_sweeps_bf16++;
#ifdef HMX_ROUND_PROF
            if (++prof_rounds == 25) {   // one round in steady state: phase durations over workgroups and blocks
                std::vector<unsigned long long> h((size_t)wgs * e->nblk * 32);
                (void)hipStreamSynchronize(e->stream);
                (void)hipMemcpy(h.data(), prof.p, h.size() * 8, hipMemcpyDeviceToHost);
                const char* names[5] = {"wait", "table", "post", "flush+arrive", "pre(next)"};
                for (int ph = 0; ph < 5; ++ph) {
                    double sum = 0, mx = 0;
                    for (int w = 0; w < wgs; ++w)
                        for (int b = 0; b < e->nblk; ++b) {
                            const double dtk = (double)(h[((size_t)w * e->nblk + b) * 32 + ph + 1] - h[((size_t)w * e->nblk + b) * 32 + ph]);
                            sum += dtk; mx = std::max(mx, dtk);
                        }
                    fprintf(stderr, "[k_round prof] %-14s mean %.0f ticks  max %.0f\n", names[ph], sum / (wgs * e->nblk), mx);
                }
                double tot = 0;
                for (int w = 0; w < wgs; ++w) tot += (double)(h[((size_t)w * e->nblk + e->nblk - 1) * 32 + 5] - h[(size_t)w * e->nblk * 32]);
                {
                    double s1 = 0, s2 = 0, s3 = 0;
                    for (int w = 0; w < wgs; ++w)
                        for (int b = 0; b < e->nblk; ++b) {
                            const unsigned long long* r = &h[((size_t)w * e->nblk + b) * 32];
                            s1 += (double)(r[6] - r[1]); s2 += (double)(r[7] - r[6]); s3 += (double)(r[2] - r[7]);
                        }
                    fprintf(stderr, "[k_round prof] table split: O update %.0f, pow %.0f, rp/log %.0f\n", s1 / (wgs * e->nblk), s2 / (wgs * e->nblk), s3 / (wgs * e->nblk));
                }
                {
                    double s1 = 0, s2 = 0, s3 = 0, sp = 0;
                    for (int w = 0; w < wgs; ++w)
                        for (int b = 1; b < e->nblk; ++b) {
                            const unsigned long long* r = &h[((size_t)w * e->nblk + b) * 32];
                            s1 += (double)(r[8] - r[0]); s2 += (double)(r[9] - r[8]); s3 += (double)(r[1] - r[9]); sp += (double)r[10];
                        }
                    const double n = (double)wgs * (e->nblk - 1);
                    fprintf(stderr, "[k_round prof] wait split: gather issue %.0f, poll %.0f (%.1f spins), syncthreads %.0f\n", s1 / n, s2 / n, sp / n, s3 / n);
                }
                {   // the tile step of wave 0 (blocks 0 .. nblk-2): ids + row fragments, requests, split + start values, k-step 0, the rest
                    double s[5] = {0, 0, 0, 0, 0};
                    for (int w = 0; w < wgs; ++w)
                        for (int b = 0; b + 1 < e->nblk; ++b) {
                            const unsigned long long* r = &h[((size_t)w * e->nblk + b) * 32];
                            if (!r[11] || !r[13]) continue;
                            s[0] += (double)(r[11] - r[4]); s[1] += (double)(r[12] - r[11]); s[2] += (double)(r[13] - r[12]);
                            s[3] += (double)(r[14] - r[13]); s[4] += (double)(r[5] - r[14]);
                        }
                    const double n = (double)wgs * (e->nblk - 1);
                    fprintf(stderr, "[k_round prof] pre(next) split: fragments %.0f, requests %.0f, split %.0f, k-step 0 %.0f, rest %.0f\n", s[0] / n, s[1] / n, s[2] / n, s[3] / n, s[4] / n);
                }
                fprintf(stderr, "[k_round prof] whole sweep mean %.0f ticks over %d workgroups\n", tot / wgs, wgs);
            }
#endif
        }
        // objacc = [2 x SLOTS partial sums of (:399, :402) | cross-entropy term (:405-411) | number of workgroups whose
        // grid-wide wait gave up].  Sharded: one all-reduce of the whole block -- the cross term was formed from job-wide
        // tables and is contributed by rank 0 only, and every rank learns of a time-out on ANY rank at the same point.
        if (multi && (rc = sum_over_ranks(e, e->objacc, 2 * HMX_OBJ_SLOTS + 2))) return rc;
        if (defer_slot && fused && e->clean_sweeps > 0) {
            // nothing hangs on this round's objective yet (hmx_cluster): its block travels to the host behind the sweep
            // and is looked at with the next round that needs a decision.  Only on an engine whose sweeps have been seen
            // to complete -- the way a grid-wide wait fails on a single engine (workgroups not co-resident) shows at once.
            HIP_TRY(hipMemcpyAsync(defer_slot, e->objacc, (2 * HMX_OBJ_SLOTS + 2) * sizeof(double), hipMemcpyDeviceToHost, e->stream));
            return 1;   // deferred
        }
        if ((rc = read_objective(e, obj_out))) return rc;
        if (e->obj_host[2 * HMX_OBJ_SLOTS + 1] == 0.0) e->clean_sweeps++;
        if (e->obj_host[2 * HMX_OBJ_SLOTS + 1] != 0.0)   // (no_replay: older rounds of this cluster() call are still unread -- the caller finds the first that failed)
            return no_replay ? 2 : replay_round(e, flags, tiles_upper, obj_out);
        return 0;
    }
    if (flags & HMX_ROUND_UPDATE_R) {
        if ((rc = blocks_loop(e, flags, tiles_upper))) return rc;
        if (e->wide_sweep_launched) {                                // the persistent wide sweep: a wait that gave up shows in the objective block
            if ((rc = read_objective(e, obj_out))) return rc;
            if (e->obj_host[2 * HMX_OBJ_SLOTS + 1] != 0.0) {
                if (e->n_sweep_fallbacks >= 1) e->wide_sweep = false;   // (the second time-out: this engine stays on the per-block path)
                return replay_round(e, flags, tiles_upper, obj_out);
            }
            return 0;
        }
    }
    return read_objective(e, obj_out);
}

static int check_round_flags(hmx_engine* e, int flags, double* obj_out) {
    if (!e || !obj_out) return fail(HMX_ERR_ARG, "null argument");
    if (!e->clustered) return fail(HMX_ERR_STATE, "hmx_init_cluster (or hmx_set of R) must come first");
    if ((flags & HMX_ROUND_OBJECTIVE) && !(flags & HMX_ROUND_UPDATE_R))
        return fail(HMX_ERR_ARG, "HMX_ROUND_OBJECTIVE needs HMX_ROUND_UPDATE_R in this build");
    if (!(flags & (HMX_ROUND_CENTROIDS | HMX_ROUND_UPDATE_R))) return fail(HMX_ERR_ARG, "nothing to do");
    return 0;
}

// block ids in static tile order for the streaming R^T.Z pass, from the lists `which` (on stream s, behind their build)
static void build_tile_blocks(hmx_engine* e, int which, int64_t n_pos_upper, hipStream_t s) {
    if (!streaming_rtz(e)) return;
    launch_tile_blocks(e->lists[which].cells.p, e->lists[which].tile_grp.p, e->lists[which].blk_start.p, e->nblk, n_pos_upper,
                       e->gstart.p, e->s_tile_start.p, e->tile_blk[which].p, s);
}

int hmx_cluster_round(hmx_engine* e, int flags, const int32_t* cells, int64_t n_pos, const int32_t* tile_group,
                      int32_t n_tiles, const int32_t* block_tile_start, double obj_out[4]) {
    int rc;
    if ((rc = check_round_flags(e, flags, obj_out))) return rc;
    if (!cells || !tile_group || !block_tile_start) return fail(HMX_ERR_ARG, "null update-order list");
    if (n_pos != (int64_t)n_tiles * HMX_TILE) return fail(HMX_ERR_ARG, "n_pos must be 16*n_tiles");
    if (block_tile_start[0] != 0 || block_tile_start[e->nblk] != n_tiles) return fail(HMX_ERR_ARG, "block_tile_start must span [0, n_tiles]");
    std::vector<int> upper(e->nblk);
    for (int b = 0; b < e->nblk; ++b) {
        if (block_tile_start[b + 1] < block_tile_start[b]) return fail(HMX_ERR_ARG, "block_tile_start must be non-decreasing");
        upper[b] = block_tile_start[b + 1] - block_tile_start[b];
    }
    {
        int64_t live = 0;
        for (int64_t i = 0; i < n_pos; ++i) {
            if (cells[i] < -1 || cells[i] >= e->N) return fail(HMX_ERR_ARG, "cells[%lld] = %d out of range", (long long)i, cells[i]);
            live += cells[i] >= 0;
        }
        if (live != e->N) return fail(HMX_ERR_ARG, "the update-order list holds %lld cells, the engine %lld: every cell must appear exactly once", (long long)live, (long long)e->N);
        for (int i = 0; i < n_tiles; ++i)
            if (tile_group[i] < 0 || tile_group[i] >= e->G) return fail(HMX_ERR_ARG, "tile_group[%d] out of range", i);
    }
    if ((rc = use_device(e))) return rc;
    if (e->pre_outstanding) {   // a list build may still be running on the second stream: it shares the scratch tables
        HIP_TRY(hipStreamSynchronize(e->stream2));
        e->pre_outstanding = false;
    }
    e->pre_valid = false;   // caller-provided lists: whatever was prepared ahead is void
    if ((rc = e->lists[e->cur].cells.reserve(n_pos)) || (rc = e->lists[e->cur].tile_grp.reserve(n_tiles))) return rc;
    {   // first tile of every (block, group) run, when every block's tiles are sorted by group (what harmonypy_amd builds):
        // k_round's group-affine map takes a workgroup's tiles from its group's run
        std::vector<int> runs((size_t)e->nblk * e->G + 1, n_tiles);
        bool sorted = true;
        for (int b = 0; b < e->nblk && sorted; ++b) {
            int t = block_tile_start[b];
            for (int g = 0; g < e->G; ++g) {
                runs[(size_t)b * e->G + g] = t;
                while (t < block_tile_start[b + 1] && tile_group[t] == g) ++t;
            }
            sorted = t == block_tile_start[b + 1];
        }
        e->lists[e->cur].runs_ok = sorted;
        if (sorted) HIP_TRY(hipMemcpyAsync(e->lists[e->cur].run_tiles.p, runs.data(), runs.size() * sizeof(int), hipMemcpyHostToDevice, e->stream));
    }
    HIP_TRY(hipMemcpyAsync(e->lists[e->cur].cells.p, cells, n_pos * sizeof(int), hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipMemcpyAsync(e->lists[e->cur].tile_grp.p, tile_group, n_tiles * sizeof(int), hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipMemcpyAsync(e->lists[e->cur].blk_start.p, block_tile_start, (e->nblk + 1) * sizeof(int), hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));  // host lists are borrowed for the call only
    if (flags & HMX_ROUND_UPDATE_R) build_tile_blocks(e, e->cur, n_pos, e->stream);
    return round_body(e, flags, n_tiles, upper, obj_out);
}

// One round whose update order comes from the keyed bijection (seed, round counter): lists built on the device, the
// next round's lists prepared on the second stream beside the sweep kernel.
// Returns 0, 1 (objective read-back deferred into defer_slot) or 2 (no_replay: a wait timed out, nothing was repeated).
// replay_only: the lists of the round are built and the round is REPLAYED block by block from the state a timed-out sweep
// left (replay_round) -- no R^T.Z pass, no sweep kernel.
static int seeded_round(hmx_engine* e, int flags, uint64_t seed, int64_t cells_per_block, double obj_out[4], double* defer_slot = nullptr,
                        bool no_replay = false, bool replay_only = false) {
    int rc;
    const int nkeys = e->nblk * e->G;
    const int nchunks = order_chunks(e->N);
    const size_t pos_cap = (size_t)e->N + (size_t)nkeys * (HMX_TILE - 1) + HMX_TILE;
    for (auto& L : e->lists)
        if ((rc = L.cells.reserve(pos_cap)) || (rc = L.tile_grp.reserve(pos_cap / HMX_TILE + 1))) return rc;
    if ((rc = e->chunk_tab.reserve((size_t)nchunks * nkeys)) || (rc = e->run_count.reserve(nkeys)) || (rc = e->run_start.reserve(nkeys)))
        return rc;
    // build the lists of round `counter` into lists[which] on stream s
    auto build = [&](uint64_t counter, int which, hipStream_t s) {
        OrderArgs o{};
        o.N = e->N; o.Ng = e->Ng; o.cpb = cells_per_block; o.nblk = e->nblk; o.G = e->G;
        o.global_id = e->global_id.p;
        int bits = 1;
        while (((int64_t)1 << bits) < e->Ng) ++bits;
        o.half_bits = (bits + 1) / 2;
        // splitmix64 of (seed, round counter) -> two 32-bit round keys
        uint64_t z = seed + 0x9E3779B97F4A7C15ull * (counter + 1);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        o.key0 = (uint32_t)z; o.key1 = (uint32_t)(z >> 32) | 1u;
        o.gstart = e->gstart.p; o.chunk_tab = e->chunk_tab.p; o.run_count = e->run_count.p; o.run_start = e->run_start.p;
        o.blk_start = e->lists[which].blk_start.p; o.cells = e->lists[which].cells.p; o.tile_grp = e->lists[which].tile_grp.p;
        o.run_tiles = e->lists[which].run_tiles.p; e->lists[which].runs_ok = true;
        if (streaming_rtz(e)) { o.tile_blk = e->tile_blk[which].p; o.s_tile_start = e->s_tile_start.p; }   // block ids in static tile order, by the way
        o.frozen = e->frozen();
        launch_order(o, s);
    };
    const uint64_t counter = e->seeded_rounds++;
    if (e->pre_valid && e->pre_seed == seed && e->pre_counter == counter && e->pre_cpb == cells_per_block) {
        e->cur ^= 1;                                            // prepared while the previous round ran
        HIP_TRY(hipStreamWaitEvent(e->stream, e->pre_event, 0));
    } else {
        if (e->pre_outstanding) HIP_TRY(hipStreamSynchronize(e->stream2));   // the scratch tables are shared
        build(counter, e->cur, e->stream);
    }
    e->pre_valid = false;
    e->pre_outstanding = false;   // either the main stream now waits for it, or stream2 was drained
    // next round's lists depend on (seed, counter) only: they are built on the second stream beside the sweep
    // kernel, which leaves a few CUs idle.  The scratch tables (chunk_tab, run_*) are free by then.
    // (enqueueing the four list launches in front of the sweep's launch instead made no measurable difference: the host is ahead of the device)
    auto prefetch = [&](int phase) -> int {
        if (!e->prefetch_lists || !e->stream2 || replay_only) return 0;
        if (phase == 0) {
            HIP_TRY(hipEventRecord(e->pre_event, e->stream));
            return 0;
        }
        HIP_TRY(hipStreamWaitEvent(e->stream2, e->pre_event, 0));
        build(counter + 1, e->cur ^ 1, e->stream2);
        HIP_TRY(hipEventRecord(e->pre_event, e->stream2));
        e->pre_outstanding = true;
        e->pre_valid = true; e->pre_seed = seed; e->pre_counter = counter + 1; e->pre_cpb = cells_per_block;
        return 0;
    };
    std::vector<int> upper(e->nblk);
    int total = 0;
    if (e->Ng == e->N) {
        // every block's size is known: cells_per_block, the last one takes the remainder (:482-484)
        for (int b = 0; b < e->nblk; ++b) {
            const int64_t size_b = (b == e->nblk - 1) ? e->N - cells_per_block * (e->nblk - 1) : cells_per_block;
            upper[b] = size_b > 0 ? (int)((size_b + HMX_TILE - 1) / HMX_TILE) + e->G : 0;
            total += upper[b];
        }
    } else if (e->peers_enabled && e->round_mode == 1 && e->mt <= 7 && round_row_floats(e->d) == e->dp) {
        // a shard holds a random share of every block.  The sweep kernel and the R^T.Z pass read the
        // tile offsets on the device and cope with any grid, so a generous estimate (mean + 8 sigma
        // of a hypergeometric share) sizes the grids without a host round trip
        for (int b = 0; b < e->nblk; ++b) {
            const double size_b = (b == e->nblk - 1) ? (double)(e->Ng - cells_per_block * (e->nblk - 1)) : (double)cells_per_block;
            const double p = (double)e->N / (double)e->Ng;
            const double mean = size_b * p, sd = std::sqrt(std::max(size_b * p * (1.0 - p), 0.0));
            upper[b] = size_b > 0 ? (int)((mean + 8.0 * sd + HMX_TILE - 1) / HMX_TILE) + e->G : 0;
            total += upper[b];
        }
    } else {
        // per-block launches need the exact tile counts: read the tile offsets back (84 bytes)
        std::vector<int> bs(e->nblk + 1);
        HIP_TRY(hipMemcpyAsync(bs.data(), e->lists[e->cur].blk_start.p, bs.size() * sizeof(int), hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
        for (int b = 0; b < e->nblk; ++b) upper[b] = bs[b + 1] - bs[b];
        total = bs[e->nblk];
    }
    if (replay_only) return replay_round(e, flags, upper, obj_out);
    return round_body(e, flags, total, upper, obj_out, prefetch, defer_slot, no_replay);
}

int hmx_cluster_round_seeded(hmx_engine* e, int flags, uint64_t seed, int64_t cells_per_block, double obj_out[4]) {
    int rc;
    if ((rc = check_round_flags(e, flags, obj_out))) return rc;
    if (cells_per_block < 0 || cells_per_block * (e->nblk - 1) > e->Ng) return fail(HMX_ERR_ARG, "cells_per_block out of range");
    if ((rc = use_device(e))) return rc;
    return seeded_round(e, flags, seed, cells_per_block, obj_out);
}

// harmony.py:437-462, the rounds of ONE cluster() call without leaving the library: rounds of hmx_cluster_round_seeded
// with flags HMX_ROUND_ALL; after round i > window (0-based, :455) the windowed test of check_convergence(0)
// (:517-523) on objective = (sum of the three terms) * 2000 / N_global (:412-416) -- the same double arithmetic on the
// same fp32-rounded terms Python would do -- ends the call.  forced_rounds >= 0 runs exactly that many rounds.
int hmx_cluster(hmx_engine* e, uint64_t seed, int64_t cells_per_block, int max_rounds, int forced_rounds, int window,
                double epsilon, double* obj_out, int32_t* rounds_out) {
    int rc;
    double probe[4];
    if (!rounds_out) return fail(HMX_ERR_ARG, "null argument");
    if ((rc = check_round_flags(e, HMX_ROUND_ALL, obj_out ? obj_out : probe))) return rc;
    if (!obj_out) return fail(HMX_ERR_ARG, "null argument");
    if (cells_per_block < 0 || cells_per_block * (e->nblk - 1) > e->Ng) return fail(HMX_ERR_ARG, "cells_per_block out of range");
    if (window < 1 || max_rounds < 0) return fail(HMX_ERR_ARG, "window must be >= 1, max_rounds >= 0");
    if ((rc = use_device(e))) return rc;
    const int n = forced_rounds >= 0 ? forced_rounds : max_rounds;
    const double norm_const = 2000.0 / (double)e->Ng;               // :412
    std::vector<double> hist;
    hist.reserve(n);
    *rounds_out = 0;
    // Rounds i <= window never end in a decision (harmony.py:455): their objective blocks follow the sweep to the host and
    // are folded when round window + 1 (or the last round) is read -- the history is the same, the host round trip is paid
    // only where the algorithm needs it.  A run of forced rounds is treated like a natural one: from round window + 1 on
    // every round is read back before the next starts, as the test of harmony.py:517-523 would require.
    // A time-out in a round whose read-back was deferred is noticed late, but nothing is lost: the sweep kernel that gave up
    // set a sticky word, and every kernel queued behind it -- the R^T.Z pass, its finish kernel, the sweep and the list
    // builds of the later rounds -- found it set and returned at once.  So the device still holds exactly what the failed
    // round left (R rows are only ever outputs; O at its start, its removal sums and centroids are untouched), the unread
    // objective blocks all repeat the failed round's, and the call resumes there: the round's lists are rebuilt (they depend
    // on seed and counter only), the round is replayed block by block (replay_round), the rounds behind it run again.
    const uint64_t counter0 = e->seeded_rounds;
    int first_pending = 0, n_pending = 0;
    for (int i = 0; i < n; ++i) {
        double* o = obj_out + 4 * (size_t)i;
        const bool may_defer = !(i > window) && i + 1 < n && n_pending < HMX_DEFER_MAX;
        rc = seeded_round(e, HMX_ROUND_ALL, seed, cells_per_block, o,
                          may_defer ? e->obj_defer + (size_t)n_pending * (2 * HMX_OBJ_SLOTS + 2) : nullptr, n_pending > 0);
        if (rc < 0) { thaw(e); return rc; }
        if (rc == 1) {                                              // deferred
            if (n_pending == 0) first_pending = i;
            ++n_pending;
            continue;
        }
        // this round was read back: everything older has landed
        int failed_round = rc == 2 ? i : -1;
        for (int j = 0; j < n_pending; ++j) {
            const double* blk = e->obj_defer + (size_t)j * (2 * HMX_OBJ_SLOTS + 2);
            if (blk[2 * HMX_OBJ_SLOTS + 1] != 0.0) { failed_round = first_pending + j; break; }
            fold_objective(blk, obj_out + 4 * (size_t)(first_pending + j));
        }
        n_pending = 0;
        if (failed_round >= 0) {
            HIP_TRY(hipStreamSynchronize(e->stream));
            if (e->stream2) HIP_TRY(hipStreamSynchronize(e->stream2));
            HIP_TRY(hipMemsetAsync(e->frozen(), 0, sizeof(unsigned long long), e->stream));   // the list build below must run
            e->pre_valid = false;
            e->pre_outstanding = false;
            e->seeded_rounds = counter0 + (uint64_t)failed_round;
            i = failed_round;
            if ((rc = seeded_round(e, HMX_ROUND_ALL, seed, cells_per_block, obj_out + 4 * (size_t)i, nullptr, false, true))) {
                thaw(e);
                return rc < 0 ? rc : fail(HMX_ERR_STATE, "replay of round %d failed", i);
            }
        }
        for (size_t j = hist.size(); j <= (size_t)i; ++j) {
            const double* oj = obj_out + 4 * j;
            hist.push_back((oj[0] + oj[1] + oj[2]) * norm_const);   // :413
        }
        *rounds_out = i + 1;
        if (forced_rounds < 0 && i > window) {                      // :455-458
            double obj_old = 0.0, obj_new = 0.0;                    // :519-522, summed left to right like Python's sum()
            const size_t m = hist.size();
            for (int j = 0; j < window; ++j) obj_old += hist[m - window - 1 + j];
            for (int j = 0; j < window; ++j) obj_new += hist[m - window + j];
            if (std::fabs(obj_old - obj_new) / std::fabs(obj_old) < epsilon) break;
        }
    }
    if (n_pending > 0) { thaw(e); return fail(HMX_ERR_STATE, "internal: %d objective blocks unread at the end of hmx_cluster", n_pending); }   // (the last round is always read back)
    *rounds_out = (int)std::max<size_t>(hist.size(), (size_t)*rounds_out);
    return HMX_OK;
}

const char* hmx_build_id(void) {
#ifdef HMX_BUILD_ID
    return HMX_BUILD_ID;
#else
    return "unknown";
#endif
}

int hmx_comm_unique_id(void* out_id) {
    if (!out_id) return fail(HMX_ERR_ARG, "null argument");
    int rc;
    if ((rc = rccl_load())) return rc;
    hmx_nccl_id id;
    std::memset(&id, 0, sizeof id);
    const int r = g_rccl.GetUniqueId(&id);
    if (r != 0) return fail(HMX_ERR_COMM, "ncclGetUniqueId failed: %s", g_rccl.GetErrorString(r));
    std::memcpy(out_id, &id, sizeof id);
    return HMX_OK;
}

int hmx_comm_init(hmx_engine* e, const void* unique_id, int n_ranks, int rank) {
    if (!e || !unique_id) return fail(HMX_ERR_ARG, "null argument");
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(HMX_ERR_ARG, "rank %d of %d", rank, n_ranks);
    if (e->host_fn) return fail(HMX_ERR_STATE, "a host transport is already attached");
    int rc;
    if ((rc = use_device(e)) || (rc = rccl_load())) return rc;
    comm_release(e);
    hmx_nccl_id id;
    std::memcpy(&id, unique_id, sizeof id);
    void* comm = nullptr;
    const int r = g_rccl.CommInitRank(&comm, n_ranks, id, rank);
    if (r != 0 || !comm) return fail(HMX_ERR_COMM, "ncclCommInitRank(rank %d of %d) failed: %s", rank, n_ranks, g_rccl.GetErrorString(r));
    e->nccl_comm = comm;
    e->n_ranks = n_ranks;
    e->rank = rank;
    return HMX_OK;
}

int hmx_set_ranks(hmx_engine* e, int n_ranks, int rank) {
    if (!e) return fail(HMX_ERR_ARG, "null argument");
    if (n_ranks < 1 || n_ranks > 8 || rank < 0 || rank >= n_ranks) return fail(HMX_ERR_ARG, "rank %d of %d (at most 8 ranks)", rank, n_ranks);
    if (e->box) return fail(HMX_ERR_STATE, "peer boxes already exist");
    e->n_ranks = n_ranks;
    e->rank = rank;
    return HMX_OK;
}

int hmx_peer_export(hmx_engine* e, void* out_handle) {
    if (!e || !out_handle) return fail(HMX_ERR_ARG, "null argument");
    if (e->n_ranks < 1 || e->n_ranks > 8) return fail(HMX_ERR_ARG, "peer boxes support at most 8 ranks");
    static_assert(sizeof(hipIpcMemHandle_t) == HMX_PEER_HANDLE_BYTES, "handle size");
    int rc;
    if ((rc = use_device(e))) return rc;
    peer_release(e);
    const size_t GKs = (size_t)(e->G + 1) * e->K16;   // a rank's share: the G group rows + one spare row (the slot tables' stride)
    e->box_doubles = peer_box_doubles(e->n_ranks, GKs);
    // FINE-GRAINED device memory: the box is written by other GPUs over xGMI and polled by a kernel that is already running
    // (k_round).  HIP specifies visibility of another device's writes during a kernel only for fine-grained allocations (a
    // coarse-grained one may be cached, and is only promised coherent at kernel boundaries); HMX_PEER_BOX=coarse keeps the
    // plain hipMalloc of rounds 3-5 (A/B), and an allocation or an export of the fine-grained box that fails falls back to it.
    const char* kind = getenv("HMX_PEER_BOX");
    bool fine = !(kind && std::string(kind) == "coarse");
    hipIpcMemHandle_t h;
    for (int attempt = 0; attempt < 2; ++attempt) {
        hipError_t ae = fine ? hipExtMallocWithFlags(reinterpret_cast<void**>(&e->box), e->box_doubles * sizeof(double), hipDeviceMallocFinegrained)
                             : hipMalloc(reinterpret_cast<void**>(&e->box), e->box_doubles * sizeof(double));
        if (ae == hipSuccess) {
            HIP_TRY(hipMemset(e->box, 0, e->box_doubles * sizeof(double)));
            HIP_TRY(hipDeviceSynchronize());
            ae = hipIpcGetMemHandle(&h, e->box);
            if (ae == hipSuccess) break;
            (void)hipFree(e->box);
            e->box = nullptr;
        }
        (void)hipGetLastError();
        if (!fine) return fail(HMX_ERR_HIP, "peer box: %s", hipGetErrorString(ae));
        fine = false;                                   // second attempt: coarse-grained
    }
    e->box_fine = fine;
    std::memcpy(out_handle, &h, sizeof h);
    return HMX_OK;
}

int hmx_peer_attach(hmx_engine* e, const void* handles) {
    if (!e || !handles) return fail(HMX_ERR_ARG, "null argument");
    if (!e->box) return fail(HMX_ERR_STATE, "hmx_peer_export must come first");
    int rc;
    if ((rc = use_device(e))) return rc;
    e->peer_ptrs.assign(e->n_ranks, nullptr);
    {   // every GPU of the node must be reachable from this one (the usual layout: rank r drives device r): a missing link would
        // only show as a failed hipIpcOpenMemHandle below, or as a self-test that times out
        int n_dev = 0, mine = e->cfg.device_id;
        if (hipGetDeviceCount(&n_dev) == hipSuccess && n_dev >= e->n_ranks)
            for (int d = 0; d < n_dev; ++d) {
                int can = 1;
                if (d != mine && hipDeviceCanAccessPeer(&can, mine, d) == hipSuccess && !can)
                    return fail(HMX_ERR_COMM, "device %d cannot access device %d: no peer exchange on this node", mine, d);
            }
        (void)hipGetLastError();
    }
    for (int r = 0; r < e->n_ranks; ++r) {
        if (r == e->rank) { e->peer_ptrs[r] = e->box; continue; }
        hipIpcMemHandle_t h;
        std::memcpy(&h, (const char*)handles + (size_t)r * HMX_PEER_HANDLE_BYTES, sizeof h);
        void* p = nullptr;
        hipError_t he = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
        if (he != hipSuccess || !p) {
            (void)hipGetLastError();
            return fail(HMX_ERR_COMM, "hipIpcOpenMemHandle(rank %d) failed: %s", r, hipGetErrorString(he));
        }
        e->peer_ptrs[r] = p;
    }
    if ((rc = e->peer_dev.reserve(e->n_ranks))) return rc;
    HIP_TRY(hipMemcpy(e->peer_dev.p, e->peer_ptrs.data(), e->n_ranks * sizeof(double*), hipMemcpyHostToDevice));
    e->peers_attached = true;
    return HMX_OK;
}

int hmx_peer_selftest(hmx_engine* e) {
    if (!e) return fail(HMX_ERR_ARG, "null argument");
    if (!e->peers_attached) return fail(HMX_ERR_STATE, "hmx_peer_attach must come first");
    int rc;
    if ((rc = use_device(e))) return rc;
    HIP_TRY(hipMemsetAsync(e->sync_words.p, 0, 2 * sizeof(unsigned), e->stream));
    e->selftest_token += 1;
    int iters = 8;                                      // HMX_PEER_SELFTEST_ITERS: a soak (tests), at most 65535 cycles
    if (const char* it = getenv("HMX_PEER_SELFTEST_ITERS")) iters = std::max(1, std::min(65535, atoi(it)));
    launch_peer_selftest(e->peer_dev.p, e->box, e->n_ranks, e->rank, (size_t)(e->G + 1) * e->K16, e->selftest_token, iters, e->sync_words.p, e->stream);
    HIP_TRY(hipMemcpyAsync(e->sync_host, e->sync_words.p, 2 * sizeof(unsigned), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipGetLastError());
    return e->sync_host[0] == 1u ? 1 : 0;
}

int hmx_peer_enable(hmx_engine* e, int on) {
    if (!e) return fail(HMX_ERR_ARG, "null argument");
    if (on && !e->peers_attached) return fail(HMX_ERR_STATE, "no peer boxes attached");
    e->peers_enabled = on != 0;
    return HMX_OK;
}

int hmx_set_host_allreduce(hmx_engine* e, hmx_host_allreduce_fn fn, void* ctx) {
    if (!e) return fail(HMX_ERR_ARG, "null argument");
    if (fn) comm_release(e);   // the host transport replaces an RCCL communicator
    e->host_fn = fn;
    e->host_ctx = ctx;
    return HMX_OK;
}

int hmx_moe_correct_ridge(hmx_engine* e) {
    if (!e) return fail(HMX_ERR_ARG, "null argument");
    if (!e->clustered) return fail(HMX_ERR_STATE, "no soft assignment yet");
    int rc;
    if ((rc = use_device(e))) return rc;
    const size_t GK = (size_t)e->G * e->K16;
    int nsub, spw;
    rtz_geometry(e->mt, e->ntd, &nsub, &spw);
    const bool rtz2 = rtz2_ok(e->mt, e->dp);   // tasks were cut for it in hmx_upload
    const bool rtzw = !rtz2 && rtz_wide_ok(e->mt, e->dp);
    if ((rc = e->slab.reserve((size_t)std::max(e->ntasks, 1) * (rtz2 ? rtz2_slab_floats(e->mt, e->dp)
                                                                     : rtzw ? rtz_wide_slab_floats(e->mt, e->dp) : spw))))
        return rc;
    if (streaming_rtz(e)) {
        // the streaming pass: Sr and Oxr are written whole by k_rtz3_finish (no fills)
        if ((rc = rtz3_pass(e, 1, e->tile_blk_zero.p, 1, false, false))) return rc;
    } else {
    HIP_TRY(hipMemsetAsync(e->Sr, 0, GK * e->ldy * sizeof(double), e->stream));
    HIP_TRY(hipMemsetAsync(e->Oxr, 0, GK * sizeof(double), e->stream));
    {
        Timed t(e, F_RIDGE_STATS);
        RtzArgs r{};
        r.R = e->R.p; r.Z = e->Zorig.p; r.cells = e->s_cells.p; r.tile_grp = e->s_tile_grp.p;
        r.task_tile0 = e->task_t0.p; r.task_tile1 = e->task_t1.p; r.task_grp = e->task_grp.p;
        r.S_out = e->Oxr; r.slab = e->slab.p; r.n_tiles = e->n_s_tiles; r.ntasks = e->ntasks;
        r.K = e->K; r.Kp = e->Kp; r.K16 = e->K16; r.G = e->G; r.mt = e->mt; r.dp = e->dp; r.ntd = e->ntd;
        if (rtz2) {
            launch_rtz2(r, e->ntasks, e->stream);
            launch_rtz2_reduce(e->slab.p, e->ntasks, e->mt, e->dp, e->K16, e->ldy, e->Sr, e->task_grp.p, e->stream);
        } else if (rtzw) {
            launch_rtz_wide(r, e->ntasks, e->stream);
            launch_rtz_wide_reduce(e->slab.p, e->ntasks, e->mt, e->dp, e->K16, e->ldy, e->Sr, e->task_grp.p, e->stream);
        } else {
            launch_rtz(r, (e->ntasks + 3) / 4, e->stream);
            launch_rtz_reduce(e->slab.p, e->ntasks, e->mt, e->ntd, e->K16, e->ldy, e->Sr, e->task_grp.p, e->stream);
        }
    }
    }
    if ((rc = sum_over_ranks(e, e->Sr, GK * e->ldy + GK))) return rc;   // Sr and Oxr are neighbours
    {
        Timed t(e, F_RIDGE_SOLVE);
        RidgeSolveArgs s{};
        s.S = e->Sr; s.Ox = e->Oxr; s.T = e->Tmass.p; s.lamb = e->lamb.p; s.Pr_b = e->Pr_b.p; s.group_cols = e->group_cols.p;
        s.W = e->W.p; s.scratch = e->scratch.p; s.alpha = e->cfg.alpha; s.lambda_est = e->cfg.lambda_estimation;
        s.K = e->K; s.K16 = e->K16; s.G = e->G; s.B = e->B; s.V = e->V; s.d = e->d; s.lds = e->ldy; s.ldw = e->ldy;
        launch_ridge_solve(s, e->stream);
    }
    {
        Timed t(e, F_RIDGE_APPLY);
        ApplyArgs a{};
        a.R = e->R.p; a.Zorig = e->Zorig.p; a.W = e->W.p; a.Zcorr = e->Zcorr.p; a.Zcos = e->Zcos.p;
        a.cells = e->s_cells.p; a.tile_grp = e->s_tile_grp.p; a.n_tiles = e->n_s_tiles;
        a.Kp = e->Kp; a.K16 = e->K16; a.dp = e->dp; a.ldw = e->ldy; a.mtd = e->ntd;
        if (rtz2 || rtzw) { a.task_tile0 = e->task_t0.p; a.task_tile1 = e->task_t1.p; a.task_grp = e->task_grp.p; a.ntasks = e->ntasks; }
        e->zcf_valid = false;                                        // Z_cos is rewritten below
        if (rtzw && e->allow_round_bf16) {                           // the correction GEMM on the bf16 pipe: W split once into fragments
            if ((rc = e->Wf.reserve(w_planes_dwords(e->G, e->K16, e->dp)))) return rc;
            launch_w_planes(e->W.p, e->G, e->K16, e->ldy, e->dp, e->Wf.p, e->stream);
            a.Wf = e->Wf.p;
        }
        if (launch_ridge_apply(a, e->max_wgs, e->stream)) return fail(HMX_ERR_ARG, "unsupported n_pcs");
    }
    HIP_TRY(hipGetLastError());
    return HMX_OK;
}

static int locate(hmx_engine* e, int which, void** p, size_t* bytes, int* rows, int* cols, int* ld, int* elem) {
    const size_t N = (size_t)e->N;
    switch (which) {
        case HMX_Z_ORIG: *p = e->Zorig.p; *rows = (int)N; *cols = e->d; *ld = e->dp; *elem = 4; break;
        case HMX_Z_COS: *p = e->Zcos.p; *rows = (int)N; *cols = e->d; *ld = e->dp; *elem = 4; break;
        case HMX_Z_CORR: *p = e->Zcorr.p; *rows = (int)N; *cols = e->d; *ld = e->dp; *elem = 4; break;
        case HMX_R: *p = e->R.p; *rows = (int)N; *cols = e->K; *ld = e->Kp; *elem = 4; break;
        case HMX_Y: *p = e->Y.p; *rows = e->K; *cols = e->d; *ld = e->ldy; *elem = 4; break;
        case HMX_O_GROUP: *p = e->Ogrp.p; *rows = e->G; *cols = e->K; *ld = e->K16; *elem = 8; break;
        case HMX_T_MASS: *p = e->Tmass.p; *rows = 1; *cols = e->K; *ld = e->K16; *elem = 8; break;
        case HMX_W: *p = e->W.p; *rows = e->G * e->K16; *cols = e->d; *ld = e->ldy; *elem = 4; break;
        default: return fail(HMX_ERR_ARG, "unknown array selector %d", which);
    }
    *bytes = (size_t)(*rows) * (*cols) * (*elem);
    return 0;
}

int hmx_get(hmx_engine* e, int which, void* host_out, size_t bytes) {
    if (!e || !host_out) return fail(HMX_ERR_ARG, "null argument");
    void* p; size_t need; int rows, cols, ld, elem, rc;
    if (which == HMX_ROUND_BLOCK_START || which == HMX_ROUND_CELLS || which == HMX_ROUND_TILE_GROUP) {
        hmx_engine::Lists& L = e->lists[e->cur];
        DevBuf<int>& b = which == HMX_ROUND_BLOCK_START ? L.blk_start : which == HMX_ROUND_CELLS ? L.cells : L.tile_grp;
        if (bytes > b.n * sizeof(int)) return fail(HMX_ERR_ARG, "round list holds %zu bytes, caller asked for %zu", b.n * sizeof(int), bytes);
        if ((rc = use_device(e))) return rc;
        HIP_TRY(hipStreamSynchronize(e->stream));
        HIP_TRY(hipMemcpy(host_out, b.p, bytes, hipMemcpyDeviceToHost));
        return HMX_OK;
    }
    if ((rc = locate(e, which, &p, &need, &rows, &cols, &ld, &elem))) return rc;
    if (which == HMX_W) need = (size_t)e->G * e->K * e->d * 4;
    if (bytes != need) return fail(HMX_ERR_ARG, "array %d holds %zu bytes, caller passed %zu", which, need, bytes);
    if ((rc = use_device(e))) return rc;
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (which == HMX_W) {  // G x K16 x ld -> G x K x d
        for (int g = 0; g < e->G; ++g)
            HIP_TRY(hipMemcpy2D((char*)host_out + (size_t)g * e->K * e->d * 4, (size_t)e->d * 4,
                                (char*)p + (size_t)g * e->K16 * ld * 4, (size_t)ld * 4, (size_t)e->d * 4, e->K, hipMemcpyDeviceToHost));
        return HMX_OK;
    }
    HIP_TRY(hipMemcpy2D(host_out, (size_t)cols * elem, p, (size_t)ld * elem, (size_t)cols * elem, rows, hipMemcpyDeviceToHost));
    return HMX_OK;
}

int hmx_get_rows(hmx_engine* e, int which, const int32_t* rows, int32_t n_rows, float* host_out, size_t bytes) {
    if (!e || !rows || !host_out) return fail(HMX_ERR_ARG, "null argument");
    if (which != HMX_Z_ORIG && which != HMX_Z_COS && which != HMX_Z_CORR && which != HMX_R) return fail(HMX_ERR_ARG, "hmx_get_rows: not an N-sized float array");
    void* p; size_t need; int nrows, cols, ld, elem, rc;
    if ((rc = locate(e, which, &p, &need, &nrows, &cols, &ld, &elem))) return rc;
    if (n_rows < 0 || bytes != (size_t)n_rows * cols * sizeof(float)) return fail(HMX_ERR_ARG, "hmx_get_rows: output size mismatch");
    for (int i = 0; i < n_rows; ++i)
        if (rows[i] < 0 || rows[i] >= nrows) return fail(HMX_ERR_ARG, "hmx_get_rows: row %d out of range", rows[i]);
    if (n_rows == 0) return HMX_OK;
    if ((rc = use_device(e))) return rc;
    DevBuf<int> d_rows;
    DevBuf<float> d_out;
    if ((rc = d_rows.reserve(n_rows)) || (rc = d_out.reserve((size_t)n_rows * cols))) { d_rows.release(); d_out.release(); return rc; }
    hipError_t he = hipMemcpyAsync(d_rows.p, rows, (size_t)n_rows * sizeof(int), hipMemcpyHostToDevice, e->stream);
    if (he == hipSuccess) {
        launch_gather_rows((const float*)p, ld, cols, d_rows.p, n_rows, d_out.p, e->stream);
        he = hipMemcpyAsync(host_out, d_out.p, bytes, hipMemcpyDeviceToHost, e->stream);
    }
    if (he == hipSuccess) he = hipStreamSynchronize(e->stream);
    d_rows.release();
    d_out.release();
    if (he != hipSuccess) return fail(HMX_ERR_HIP, "hmx_get_rows: %s", hipGetErrorString(he));
    return HMX_OK;
}

int hmx_set(hmx_engine* e, int which, const void* host_in, size_t bytes) {
    if (!e || !host_in) return fail(HMX_ERR_ARG, "null argument");
    if (which == HMX_W) return fail(HMX_ERR_ARG, "W is an output");
    void* p; size_t need; int rows, cols, ld, elem, rc;
    if ((rc = locate(e, which, &p, &need, &rows, &cols, &ld, &elem))) return rc;
    if (bytes != need) return fail(HMX_ERR_ARG, "array %d holds %zu bytes, caller passed %zu", which, need, bytes);
    if (!e->uploaded) return fail(HMX_ERR_STATE, "hmx_upload must come first");
    if ((rc = use_device(e))) return rc;
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (which == HMX_Z_COS) e->zcf_valid = false;
    HIP_TRY(hipMemset(p, 0, (size_t)rows * ld * elem));
    HIP_TRY(hipMemcpy2D(p, (size_t)ld * elem, host_in, (size_t)cols * elem, (size_t)cols * elem, rows, hipMemcpyHostToDevice));
    if (which == HMX_R) {  // keep O and T consistent with the new assignment (exact sums)
        const size_t GK = (size_t)e->G * e->K16;
        HIP_TRY(hipMemsetAsync(e->Ogrp.p, 0, GK * sizeof(double), e->stream));
        launch_group_sums(e->R.p, e->Kp, e->K, e->K16, e->s_cells.p, e->s_tile_grp.p, e->n_s_tiles, e->Ogrp.p, e->stream);
        if ((rc = sum_over_ranks(e, e->Ogrp.p, GK))) return rc;
        TableArgs ta = table_args(e);
        ta.O_prev = e->Ogrp.p; ta.T_out = e->Tmass.p;
        launch_block_table(ta, e->K16, e->stream);
        HIP_TRY(hipStreamSynchronize(e->stream));
        HIP_TRY(hipGetLastError());
        e->clustered = true;
    }
    return HMX_OK;
}

int hmx_sync(hmx_engine* e) {
    if (!e) return fail(HMX_ERR_ARG, "null argument");
    int rc;
    if ((rc = use_device(e))) return rc;
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipGetLastError());
    return HMX_OK;
}

int hmx_device_ptr(hmx_engine* e, int which, void** d_ptr, size_t* bytes) {
    if (!e || !d_ptr || !bytes) return fail(HMX_ERR_ARG, "null argument");
    size_t need; int rows, cols, ld, elem, rc;
    if ((rc = locate(e, which, d_ptr, &need, &rows, &cols, &ld, &elem))) return rc;
    if (which == HMX_Z_COS) e->allow_zcf = false;   // the caller may write Z_cos behind the engine's back: no cached planes of it any more
    *bytes = (size_t)rows * ld * elem;
    return HMX_OK;
}

int hmx_counters(hmx_engine* e, int64_t out[HMX_N_COUNTERS]) {
    if (!e || !out) return fail(HMX_ERR_ARG, "null argument");
    int rc;
    if ((rc = use_device(e))) return rc;
    unsigned long long ws[4] = {0, 0, 0, 0};
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipMemcpy(ws, e->wait_stats.p, sizeof ws, hipMemcpyDeviceToHost));
    out[0] = e->n_collectives;
    out[1] = e->n_sweep_fallbacks;
    out[2] = (int64_t)e->seeded_rounds;
    out[3] = e->n_sweeps_bf16;
    out[4] = (int64_t)ws[0];
    out[5] = (int64_t)ws[1];
    out[6] = (int64_t)ws[2];
    out[7] = e->n_rtz_bf16;
    out[8] = e->n_sweeps_ga;
    out[9] = e->n_sweeps_ga > 0 ? e->ga_nwg : 0;
    out[10] = e->box ? (e->box_fine ? 2 : 1) : 0;
    out[11] = e->n_rtz_zf;
    out[12] = e->n_sweeps_wide;
    for (int i = 13; i < HMX_N_COUNTERS; ++i) out[i] = 0;
    return HMX_OK;
}

int hmx_enable_timing(hmx_engine* e, int on) {
    if (!e) return fail(HMX_ERR_ARG, "null argument");
    (void)hipSetDevice(e->cfg.device_id);
    (void)hipStreamSynchronize(e->stream);
    drain_spans(e);
    e->timing = on != 0;                            // any non-zero value: the families selected by hmx_set_timing_families (default: all)
    for (int i = 0; i < F_COUNT; ++i) { e->fam_ms[i] = 0; e->fam_n[i] = 0; e->fam_seen[i] = 0; }
    return HMX_OK;
}

int hmx_set_timing_families(hmx_engine* e, unsigned mask) {
    if (!e) return fail(HMX_ERR_ARG, "null argument");
    if (mask == 0) return fail(HMX_ERR_ARG, "empty family mask (hmx_enable_timing(e, 0) turns timing off)");
    e->timing_mask = mask;                          // bit f selects family f of hmx_kernel_times
    return HMX_OK;
}

int hmx_set_timing_stride(hmx_engine* e, int stride) {
    if (!e || stride < 1) return fail(HMX_ERR_ARG, "stride must be >= 1");
    e->timing_stride = stride;
    return HMX_OK;
}

int hmx_kernel_times(hmx_engine* e, double* ms_out, int n, const char** names_out) {
    if (!e || !ms_out) return fail(HMX_ERR_ARG, "null argument");
    (void)hipSetDevice(e->cfg.device_id);
    (void)hipStreamSynchronize(e->stream);
    drain_spans(e);
    // ms_out[2*i] = total ms of family i, ms_out[2*i+1] = number of launches
    for (int i = 0; i < F_COUNT && 2 * i + 1 < n; ++i) { ms_out[2 * i] = e->fam_ms[i]; ms_out[2 * i + 1] = (double)e->fam_n[i]; }
    if (names_out) *names_out = kFamilyNames;
    return F_COUNT;
}

}  // extern "C"
