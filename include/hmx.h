/*
 * hmx.h -- C ABI of the MI355X-native Harmony iteration engine (libhmx.so).
 *
 * The reference (slowkow/harmonypy v0.2.0) has no FFI boundary of its own: its
 * hot path is the body of `class Harmony` (harmonypy/harmony.py:218-569), whose
 * arithmetic is delegated to torch.  This header is the boundary a maintainer
 * would bind instead of torch for that path (ctypes stub in INTEGRATION.md).
 * Every entry point names the reference code it replaces.
 *
 * Conventions
 *   - plain C, no C++/torch types; all pointers are HOST pointers unless named d_*.
 *   - return 0 on success, negative hmx_status on error; text via hmx_last_error()
 *     (thread-local).  No exception crosses the boundary.
 *   - caller owns every host buffer (borrowed for the duration of the call);
 *     the engine owns all device memory until hmx_destroy().
 *   - calls on one engine are not re-entrant; different engines are independent.
 *   - "internal cell order": the caller sorts cells by batch group (the distinct
 *     multi-hot rows of Phi) and hands rows in that order; the engine never sees
 *     the original order.  A *tile* is 16 consecutive list positions that share
 *     one group; position lists are padded with -1 up to tile boundaries.
 */
#ifndef HMX_H
#define HMX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HMX_ABI_VERSION 8
#define HMX_TILE 16 /* cells per tile */
/* limits of this build, checked by hmx_create (the reference has none: harmony.py:123-124 caps only the default K) */
#define HMX_MAX_CLUSTERS 320 /* beyond 208 clusters or PCs: the generic kernels (no MFMA-tiled fast path) */
#define HMX_MAX_PCS 320
#define HMX_MAX_BLOCKS 250 /* block_size >= 0.004: a cell's block id of the round travels as one byte */
#define HMX_MAX_VARS 32

typedef enum hmx_status {
    HMX_OK = 0,
    HMX_ERR_ARG = -1,     /* bad argument / unsupported shape */
    HMX_ERR_HIP = -2,     /* a HIP runtime call failed */
    HMX_ERR_STATE = -3,   /* call order violated (e.g. round before upload) */
    HMX_ERR_COMM = -4     /* RCCL failure */
} hmx_status;

/* which = selector for hmx_get / hmx_set (sizes in elements of the listed type) */
typedef enum hmx_array {
    HMX_Z_ORIG = 0, /* float  N x d   internal order                (harmony.py:235)  */
    HMX_Z_COS = 1,  /* float  N x d                                  (harmony.py:238)  */
    HMX_Z_CORR = 2, /* float  N x d                                  (harmony.py:234)  */
    HMX_R = 3,      /* float  N x K   soft assignments               (harmony.py:363)  */
    HMX_Y = 4,      /* float  K x d   unit-length centroids as rows  (harmony.py:364)  */
    HMX_O_GROUP = 5,/* double G x K   sum of R over the cells of a group; O[k,b] of
                       harmony.py:360 is the sum over the groups containing column b   */
    HMX_T_MASS = 6, /* double K       cluster mass; E[k,b] = T[k]*Pr_b[b] (harmony.py:361) */
    HMX_W = 7,      /* float  G x K x d  per-group correction vectors of the last ridge */
    /* the update-order lists of the last round (read-only; int32) */
    HMX_ROUND_BLOCK_START = 8, /* n_blocks+1 tile offsets                                  */
    HMX_ROUND_CELLS = 9,       /* 16 * block_start[n_blocks] list positions, -1 = padding  */
    HMX_ROUND_TILE_GROUP = 10  /* block_start[n_blocks] groups                              */
} hmx_array;

typedef struct hmx_config {
    int64_t n_cells;          /* N  cells held by this engine (this rank's shard)      */
    int64_t n_cells_global;   /* cells of the whole job over all ranks (0 = n_cells);
                                 the update order (harmony.py:471-484) is a permutation of
                                 this many cells and blocks are cut from it              */
    int32_t n_pcs;            /* d                                                      */
    int32_t n_clusters;       /* K                                                      */
    int32_t n_batches;        /* B  = number of Phi rows                                */
    int32_t n_groups;         /* G  distinct multi-hot patterns (== B for one variable) */
    int32_t n_vars;           /* V  active Phi rows per cell (len(vars_use))            */
    int32_t n_blocks;         /* ceil(1/block_size)                 (harmony.py:474)    */
    int32_t device_id;        /* HIP device ordinal                                     */
    int32_t lambda_estimation;/* 0/1                                (harmony.py:541)    */
    float alpha;              /*                                    (harmony.py:590)    */
    int32_t reserved[4];
} hmx_config;

typedef struct hmx_engine hmx_engine;

const char* hmx_last_error(void);
int hmx_abi_version(void);
/* Identity of the kernel set in this library: 12 hex digits of the SHA-256 over csrc/ and this header, computed by
 * harmonypy_amd/_build.py at build time.  Profiles and counter files name the build they were collected on. */
const char* hmx_build_id(void);

/* Allocate the device state of one `Harmony` object (harmony.py:230-278, 357-364). */
int hmx_create(const hmx_config* cfg, hmx_engine** out);
void hmx_destroy(hmx_engine* e);

/* Upload inputs (harmony.py:234-271).  Z: N x d row-major; the engine derives Z_cos
 * (harmony.py:238).  source_row: for every internal (group-sorted) cell its row in Z,
 * so that Z travels in the caller's order and is regrouped on the device; NULL = Z is
 * already in internal order.  static_cells/static_tile_group: the
 * group-sorted identity list padded to tiles (n_static_pos = 16*n_static_tiles).
 * One batch variable (n_vars == 1): group g IS batch g -- n_groups == n_batches and group_cols[g] == g are
 * required (the closed-form ridge solve and the sweep kernel's tables index by it): batch levels that
 * hold no cell must be dropped by the caller (harmony.py:134 counts the levels present, too).
 * group_cols: G x V Phi-row indices of every group.  lamb: B+1 floats, ignored
 * when lambda_estimation.  Pr_b (harmony.py:170) is the batch proportion over the
 * whole job.  global_id: for every internal cell its index in [0, n_cells_global)
 * (the cell's row in the unsharded input); NULL = the internal index itself. */
int hmx_upload(hmx_engine* e, const float* Z, const int32_t* static_cells, int64_t n_static_pos,
               const int32_t* static_tile_group, int32_t n_static_tiles, const int32_t* group_cols,
               const float* Pr_b, const float* theta, const float* sigma, const float* lamb,
               const int32_t* global_id, const int32_t* source_row);

/* ---- cells sharded over several engines (one process per GPU) -------------------------
 * Every cross-cell quantity of the path is a small fp64 table -- centroid numerators K x d
 * (harmony.py:443), batch-by-cluster sums K x B of a block (harmony.py:491-492, 506-507),
 * the three objective sums (harmony.py:399-411), the ridge statistics (harmony.py:550,
 * 559-563).  With a transport attached the engine sums each of them over all ranks at the
 * point where the reference forms it, so every rank walks the same O/E/Y/W.  All ranks must
 * make the same calls in the same order.
 *
 * hmx_comm_init: RCCL communicator over xGMI bound to the engine's stream (stream-ordered
 * ncclAllReduce, no host round trip).  unique_id: NCCL_UNIQUE_ID_BYTES (128) bytes obtained
 * from hmx_comm_unique_id on one rank and broadcast by the caller.
 * hmx_set_host_allreduce: any other transport (the tests use gloo): the engine stages the
 * table to host memory, calls fn(ctx, buf, count) which must sum buf over the ranks in
 * place, and copies it back. */
#define HMX_UNIQUE_ID_BYTES 128
int hmx_comm_unique_id(void* out_id);
int hmx_comm_init(hmx_engine* e, const void* unique_id, int n_ranks, int rank);
typedef int (*hmx_host_allreduce_fn)(void* ctx, double* buf, size_t count);
int hmx_set_host_allreduce(hmx_engine* e, hmx_host_allreduce_fn fn, void* ctx);

/* Peer exchange inside the sweep kernel.  The 20 per-block sums of a round (harmony.py:506-507)
 * sit on the critical path of update_R; with peer boxes attached they are exchanged INSIDE the
 * persistent sweep kernel -- every GPU writes its block sums straight into the other GPUs' boxes
 * over xGMI -- instead of one collective launch per block.  Each rank exports the inter-process
 * handle of its box (hmx_peer_export, HMX_PEER_HANDLE_BYTES bytes), the caller gathers the handles
 * of all ranks in rank order and hands them to hmx_peer_attach.  hmx_peer_selftest runs one
 * exchange cycle with a time-out and returns 1 when every peer's token arrived, 0 otherwise (the
 * caller enables the in-kernel exchange only when all ranks report 1).  n_ranks / rank must be
 * given first (hmx_comm_init, or hmx_set_ranks with a host transport). */
#define HMX_PEER_HANDLE_BYTES 64
int hmx_set_ranks(hmx_engine* e, int n_ranks, int rank);
int hmx_peer_export(hmx_engine* e, void* out_handle);
int hmx_peer_attach(hmx_engine* e, const void* handles /* n_ranks x HMX_PEER_HANDLE_BYTES */);
int hmx_peer_selftest(hmx_engine* e);
int hmx_peer_enable(hmx_engine* e, int on);

/* harmony.py:376-392 given the k-means centres of harmony.py:370-373.
 * Y0: K x d row-major (centroids as rows, not yet normalised).
 * obj_out = {sum R*dist, sum sigma*R*log R, cross-entropy term, 0}, each rounded to
 * fp32 like the `.item()` calls of harmony.py:399-411. */
int hmx_init_cluster(hmx_engine* e, const float* Y0, double obj_out[4]);

/* LISI, the integration metric of the reference (harmonypy/lisi.py:24-133 `compute_lisi` +
 * `compute_simpson`), on the device.  X: n x d row-major float64 (host).  label_codes: n_labels x n
 * category codes (one row per label column, as pd.Categorical(...).codes).  perplexity: as the
 * reference's; the search asks for int(3*perplexity) neighbours (the cell itself included, then
 * dropped, lisi.py:53-60), at most 2040 (perplexity 680).  lisi_out: n x n_labels row-major float64 (-1 where the
 * reference returns -1).  knn_dist_out / knn_idx_out (both or neither, may be NULL): the
 * int(3*perplexity)-1 neighbours of every cell, nearest first, n x (nn-1).
 * Neighbours are exact: a float32 MFMA pass preselects 128 candidates per cell (512 above 120
 * neighbours, 2048 above 504), float64 distances from direct differences rank them.  Independent of
 * any engine handle. */
int hmx_compute_lisi(int32_t device_id, const double* X, int64_t n, int32_t d, const int32_t* label_codes,
                     int32_t n_labels, double perplexity, double* lisi_out, double* knn_dist_out,
                     int32_t* knn_idx_out);

/* k-means++ seeding on the device: K = n_clusters centres chosen among `points` (n_points x d
 * row-major host floats, rows of unit length -- a subsample of Z_cos) by the greedy k-means++ that
 * sklearn's KMeans(init='k-means++') runs for the reference (harmony.py:370; sklearn 1.7
 * `_kmeans_plusplus`: first centre uniform, then 2+int(log K) candidates per centre drawn with
 * probability proportional to the squared distance to the closest centre, the one with the smallest
 * remaining potential kept).  The draws come from a counter-based generator keyed by `seed`, not
 * from NumPy's, so the centres are statistically, not bitwise, those of sklearn; the selection is
 * integer arithmetic and reproducible (oracle/kmeans_seed.py).  centers_out: K x d; chosen_out
 * (may be NULL): the K point indices.  Does not need hmx_upload. */
int hmx_kmeans_seed(hmx_engine* e, const float* points, int64_t n_points, uint64_t seed, float* centers_out,
                    int32_t* chosen_out);

/* Lloyd iterations of the initial k-means on the device.  The reference fits sklearn's KMeans on the
 * host (harmony.py:369-373: k-means++ seeding + at most 25 Lloyd iterations; 18 s at 1M cells); for
 * jobs too large for that the caller seeds on a subsample and lets the engine run the Lloyd
 * iterations over all cells of Z_cos (Euclidean k-means, centroid = mean of its members, empty
 * clusters keep their centre; sums over all ranks when cells are sharded).  centers_in / centers_out:
 * K x d row-major; the result is what hmx_init_cluster takes as Y0.  Shapes with K > 112 or d > 64 use R as
 * scratch (hard assignment as a one-hot R, member sums as the R^T.Z statistics of it): an assignment the engine
 * held is void afterwards, hmx_init_cluster must follow. */
int hmx_kmeans_lloyd(hmx_engine* e, const float* centers_in, int n_iter, float* centers_out);
/* 1 when hmx_kmeans_lloyd serves this engine's shape and layout, 0 when it would return HMX_ERR_ARG (K > 112 or d > 64 with
 * static tiles that do not hold consecutive cells, or more batch groups than its statistics pass tabulates: 85 at 200 PCs).
 * Every rank of a sharded job asks BEFORE the call and all fall back together (the iterations contain all-reduces).  No
 * reference counterpart (harmony.py:369-373 always runs sklearn's Lloyd on the host). */
int hmx_can_lloyd(hmx_engine* e);

/* One pass of the loop body harmony.py:443-453.
 *   flags: HMX_ROUND_* bits; the reference's round is all three.
 *   cells / tile_group / block_tile_start describe this round's update order
 *   (harmony.py:471-484): positions are grouped by block, inside a block by group,
 *   each (block, group) run padded with -1 to a multiple of 16.  block b owns tiles
 *   [block_tile_start[b], block_tile_start[b+1]). */
#define HMX_ROUND_CENTROIDS 1 /* harmony.py:443-447 */
#define HMX_ROUND_UPDATE_R 2  /* harmony.py:450, 464-513 */
#define HMX_ROUND_OBJECTIVE 4 /* harmony.py:453, 394-417 */
#define HMX_ROUND_ALL 7       /* the reference's round: all three */
int hmx_cluster_round(hmx_engine* e, int flags, const int32_t* cells, int64_t n_pos,
                      const int32_t* tile_group, int32_t n_tiles, const int32_t* block_tile_start,
                      double obj_out[4]);

/* Same round, update order drawn on the device: a keyed bijection of [0, n_cells_global)
 * (round key from `seed` and the engine's round counter) stands in for torch.randperm
 * (harmony.py:471): a cell's position in the order is the inverse bijection of its global id,
 * so every rank of a sharded job finds the blocks of its own cells without communication and
 * the blocks do not depend on how the cells are sharded.  Blocks are cells_per_block positions
 * of the global order each, the last one takes the remainder (harmony.py:475-484).
 * Statistically equivalent to, not bitwise the same stream as, the reference's generator
 * (the reference itself changes stream between its 'cpu' and 'cuda' devices). */
int hmx_cluster_round_seeded(hmx_engine* e, int flags, uint64_t seed, int64_t cells_per_block,
                             double obj_out[4]);

/* harmony.py:437-462: ALL rounds of one cluster() call, update order as in hmx_cluster_round_seeded, without a trip
 * through the caller between rounds.  Round i (0-based) is followed, when i > window (harmony.py:455), by the windowed
 * test of check_convergence(0) (harmony.py:517-523) on objective_i = (sum of the three terms) * 2000 / n_cells_global
 * (harmony.py:412-413) -- the double arithmetic Python does on the same fp32-rounded terms; the call ends when the
 * relative change of the window sums is below `epsilon` or after max_rounds rounds.  forced_rounds >= 0 runs exactly
 * that many rounds instead (no test).  obj_out: 4 doubles per round run (as hmx_cluster_round: the three terms, 0),
 * room for max(max_rounds, forced_rounds) rounds; *rounds_out: rounds run (harmony.py:460 kmeans_rounds entry). */
int hmx_cluster(hmx_engine* e, uint64_t seed, int64_t cells_per_block, int max_rounds, int forced_rounds, int window,
                double epsilon, double* obj_out, int32_t* rounds_out);

/* harmony.py:535-569. */
int hmx_moe_correct_ridge(hmx_engine* e);

/* Copy a state array to / from the host (property getters harmony.py:288-351). */
int hmx_get(hmx_engine* e, int which, void* host_out, size_t bytes);
int hmx_set(hmx_engine* e, int which, const void* host_in, size_t bytes);
/* n_rows rows (internal cell ids) of a float N-sized array (HMX_Z_ORIG / Z_COS / Z_CORR / R), gathered on
 * the device: host_out receives n_rows x cols floats. */
int hmx_get_rows(hmx_engine* e, int which, const int32_t* rows, int32_t n_rows, float* host_out, size_t bytes);

/* Block until all queued work of the engine finished. */
int hmx_sync(hmx_engine* e);

/* Device pointer of a state array (for callers that keep data resident). */
int hmx_device_ptr(hmx_engine* e, int which, void** d_ptr, size_t* bytes);

/* Kernel time per kernel family, measured with HIP events on the engine's stream while timing is on.
 * hmx_enable_timing(e, on): on == 0 off, anything else on -- for every family, or for the families last selected with
 * hmx_set_timing_families(e, mask) (bit f = family f in the order of names_out; ABI 5 passed the mask as `on`, which made
 * on == 1 mean "family 0 only": since ABI 6 the switch is a switch again).  Every bracketed launch costs two event records on
 * the stream (a few microseconds of queue time each: at C3 all families together slow a Harmony iteration by 11 %), so a
 * caller that times a run brackets only what it reports.
 * hmx_kernel_times: ms_out[2 f] = total milliseconds of family f, ms_out[2 f + 1] = launches; names_out receives a static
 * NUL-separated list of the family names. */
int hmx_kernel_times(hmx_engine* e, double* ms_out, int n, const char** names_out);
int hmx_enable_timing(hmx_engine* e, int on);
int hmx_set_timing_families(hmx_engine* e, unsigned mask);
/* Bracket only every stride-th launch of a timed family (default 1: every launch): a uniform sample of the launches,
 * for callers whose timed loop must not carry the event records of every launch. */
int hmx_set_timing_stride(hmx_engine* e, int stride);

/* Event counters of the engine since hmx_create: out[0] collectives issued (sharded jobs), out[1] rounds whose
 * persistent sweep kernel gave up on a grid-wide wait and were repeated block by block (harmony.py:464-513 has no
 * counterpart: it runs the blocks one torch call at a time), out[2] rounds with a device-side update order, out[3] sweeps
 * whose distance GEMM ran on the bf16 matrix pipe (fp32 operands as three bf16 terms; the others
 * used the f32-input instances: shapes whose tables need the LDS, or HMX_ROUND_F32=1);
 * the grid-wide waits of the persistent sweep kernel (per workgroup and block: the hop every block of a sweep pays, across
 * ranks when cells are sharded): out[4] waits, out[5] polls that found the hand-off incomplete (each followed by an
 * s_sleep of ~64 shader cycles), out[6] the most such polls any single wait took, out[7] streaming R^T.Z passes
 * (centroid numerators + removal sums, ridge statistics) that ran on the bf16 matrix pipe (k_rtz3c; engines created under
 * HMX_RTZ3_BF16=0 keep the f32-input kernel k_rtz3; wide shapes: k_rtzw2b), out[8] sweeps launched with the group-affine tile
 * map (one batch variable: every workgroup of k_round owns one batch group; HMX_ROUND_GA=0 keeps the
 * classic map), out[9] workgroups of the last such sweep, out[10] the peer box of a sharded engine: 0 none, 1 coarse-grained
 * device memory (HMX_PEER_BOX=coarse, or the fall-back), 2 fine-grained; out[11] wide streaming R^T.Z passes that read Z_cos as
 * pre-split bf16 planes (k_rtzw2b<.., true>; HMX_RTZW_ZF=0: fp32 rows split in every pass); out[12] wide sweeps (clusters or PCs beyond 112 / 64) that ran as ONE persistent
 * launch (k_sweep_wide3: single engine, one batch variable, at most 32 groups; HMX_WIDE_SWEEP=0 keeps one launch per block);
 * out[13..15] reserved (0). */
#define HMX_N_COUNTERS 16
int hmx_counters(hmx_engine* e, int64_t out[HMX_N_COUNTERS]);

#ifdef __cplusplus
}
#endif
#endif /* HMX_H */
