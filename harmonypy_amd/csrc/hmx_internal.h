// hmx_internal.h -- argument blocks shared by the kernels (hmx_kernels.hip) and the C ABI
// host code (hmx_capi.cpp).  Not part of the public interface (include/hmx.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#define HMX_RTZ_MTW 7 /* output tile block of k_rtz: 7 x 4 tiles of 16 x 16 = 112 accumulators */
#define HMX_RTZ_NTW 4
#define HMX_OBJ_SLOTS 64 /* objective partials are spread over this many fp64 slot pairs */

struct AssignArgs {
    const float* Zcos;     // N x dp
    const float* Y;        // K16 x ldy, unit rows, zero padded
    const float* sigma;    // K16 (zero padded)
    const float* rp;       // G x K16 powered diversity ratio summed over variables (PENALTY only)
    const float* lrp;      // G x K16 log(rp)
    float* R;              // N x Kp
    const int* cells;      // list positions -> internal cell id, -1 = padding
    const int* tile_grp;   // group of every tile
    double* S_out;         // G x K16: sum of the new R per (group, cluster)
    double* obj;           // HMX_OBJ_SLOTS x {sum R*dist, sum sigma R log R} partial sums
    const int* blk_start;  // device block_tile_start (nblk+1) or null => tile_begin/tile_end
    int blk;
    int tile_begin, tile_end;  // host-side range (or an upper bound of its length when blk_start)
    int K, Kp, K16, mt, dp, ldy;
    int G, ldy_lds, tables_in_lds, tiles_per_wave, ablate;
    const unsigned* Yf;    // wide shapes, bf16 pipe: Y as the A fragments of k_assign_wide3 (launch_y_planes), or null (the f32-input kernel k_assign_wide)
    int bf16_pipe;         // wide shapes: the bf16-pipe instance k_assign_wide3 (0: engines created under HMX_ROUND_F32=1 keep k_assign_wide)
    // k_assign_wide3 building the block's table itself (assign_wide3_fuses_table; one batch variable): O_b = O_prev + S_add - S_sub
    int fuse_table;
    const double* O_prev;  // G x K16: O behind the previous block
    const double* S_add;   // G x K16: the previous block's new sums (null for the first block)
    const double* S_sub;   // G x K16: this block's old sums
    double* O_out;         // G x K16: O without this block (written by workgroup 0: the chain's next O_prev)
    const float* Pr_b;
    const float* theta;
    // k_sweep_wide3 (all blocks in one persistent launch): S_out / S_sub / O_out are the bases of the per-block tables, O_prev is O at the
    // start of the round; S_out of every block but the last is read as 64-bit fixed-point words (zeroed by the caller)
    int nblk;
    const int* run_tiles;  // nblk*G + 1: first tile of every (block, group) run
    unsigned spin_limit;   // polls a wait for a block's sums may take
    double* O_priv;        // 2 x grid x G x K16: every workgroup's own copies of O (each entry read and written by one thread only)
    double* fail;          // += 1 per workgroup whose wait gave up (the launch ends, the host replays the round block by block)
    const float* hn;       // k_assign_wide without penalty only: half squared norms of the centres in Y (+inf for pads) -> HARD
                           // assignment of the device k-means (a one-hot row of R per cell) instead of the softmax; null otherwise
};

#ifndef HMX_ROUND_SLOTS
#define HMX_ROUND_SLOTS 4 /* k_round: a block's new sums are spread over this many fp64 tables (2 and 8 measured in round 6: profiles/r06_ab_k_round_slots_poll.txt) */
#endif

// One whole update_R sweep (all blocks) in one persistent launch (k_round).
struct RoundArgs {
    const float* Zcos;       // N x dp
    const float* Y;          // K16 x ldy
    const float* sigma;      // K16
    float* R;                // N x Kp
    const int* cells;        // block-major padded list
    const int* tile_grp;
    const int* blk_start;    // nblk+1 tile offsets (device)
    const double* O_start;   // G x K16: O at the start of the round
    const double* S_old;     // nblk x G x K16: removal sums of every block (from the old R)
    double* S_new;           // nblk x HMX_ROUND_SLOTS x (G + 1) x K16, zeroed by the caller (row G: cluster mass, group-affine map only)
    double* O_out;           // G x K16: O after the round
    double* T_out;           // K16: cluster mass after the round
    double* obj;             // HMX_OBJ_SLOTS x 2 partial sums + cross-entropy term at [2*HMX_OBJ_SLOTS]
    const int* group_cols;   // G x V
    const float* Pr_b;
    const float* theta;
    unsigned* counter;       // arrivals (zeroed by the caller)
    unsigned* error;         // set to 1 when a wait gave up (zeroed by the caller)
    unsigned* frozen;        // sticky: set with `error`, cleared by the HOST only -- every kernel of a later round returns at once while it is set (see hmx_cluster)
    unsigned long long* wait_stats;   // {waits, polls that found the hand-off incomplete, most polls of one wait}: accumulated
    unsigned long long* prof;  // HMX_ROUND_PROF builds: wgs x nblk x 8 time stamps (or null)
    // cells sharded over ranks: the block sums travel through peer boxes (null / 1: single engine)
    double* const* peer_box;   // n_ranks box base pointers (own box included), device array
    double* my_box;
    unsigned long long epoch;  // flag value of block b in this launch = epoch + b + 1
    unsigned spin_limit;       // polls a wait may take before it gives up
    int n_ranks, rank;
    int K, Kp, K16, dp, ldy, ldy_lds, G, B, V, nblk;
    // group-affine tile map (one batch variable, one engine): every compute workgroup owns ONE batch group and takes its tiles
    // from that group's run inside each block; the hand-off then carries K16 entries of the group + K16 cluster masses instead
    // of G x K16, as self-validating words (count << 55 | sum in 2^-32 fixed point: see k_round)
    int ga;                    // 1: group-affine map, 0: classic (tile pairs dealt round-robin over all workgroups)
    const int* run_start;      // ga: nblk * G + 1 tile offsets of the (block, group) runs of the list, key = block * G + group
    const int* wg_map;         // ga: per compute workgroup {group, rank among the group's workgroups, workgroups of the group}
    int ga_slots;              // ga: slot tables a block's adds are spread over (1 on small grids, else 2)
    int req_mode;              // ga: where the waves issue the next block's row requests (0 classic stagger, 1 behind the GEMM, 2 between its k-steps)
};

struct RtzArgs {
    const float* R;        // N x Kp
    const float* Z;        // N x dp
    const int* cells;
    const int* tile_grp;
    const int* blk_start;  // nblk+1 (block-major list) or null
    const int* task_tile0; // per-wave task ranges (ridge) or null
    const int* task_tile1;
    const int* task_grp;
    double* S_out;         // [blk][G][K16] column sums of R (or null)
    float* slab;
    int n_tiles, ntasks, nblk;
    int K, Kp, K16, G, mt, dp, ntd;
    int ldr, ldz;          // k_rtz2: LDS row strides (set by the launcher)
};

// The R^T.Z pass in storage order (k_rtz3, hmx_rtz3.hip): tasks are runs of static tiles of ONE group.
struct Rtz3Args {
    const float* R;            // N x Kp (+ 16 rows of slack)
    const float* Z;            // N x dp (+ 16 rows of slack): Z_cos (k-means round) or Z_orig (ridge)
    const unsigned char* tile_blk;   // n_static_tiles x 16 block ids in static tile order (all 0: column 0 = plain column sums)
    const int* task_t0;        // static tile range of a task
    const int* task_t1;
    const int* task_stride;    // tiles between a wave's consecutive tiles (4: contiguous task; 4 x tasks of the group: interleaved)
    const int* task_c0;        // first cell of tile task_t0 (the cells of a group's tiles are consecutive)
    const int* task_cend;      // first cell behind the task's group
    float* slab;               // ntasks x MT x NT x 256 accumulators, [tile][lane][r]
    unsigned long long* prof;  // -DHMX_RTZ3_PROF builds: ntasks x waves x 8 time stamps (else null)
    const unsigned* frozen;    // non-zero: an earlier sweep of this cluster() call timed out -- do nothing (or null)
    const unsigned* Zf;        // k_rtzw2b: Z as bf16 planes in its B-fragment order per static tile (launch_zplanes), or null: fp32 rows, split per pass
    int ntasks, Kp;
    int dp, d, nt;             // k_rtzw (wide shapes; set by its launcher): row floats of Z, PCs, output column tiles
};
struct Rtz3FinishArgs {
    const float* slab;
    const int* task_grp;
    int ntasks, MT, KS, NTB, K, K16, d, ld, G, nblk;
    int wide, NT;              // k_rtzw's slabs: plain column tiles, NT of them (KS = dp / 4 there too)
    int mode;                  // 0: k-means round (Ysum, Sold, optional Yout), 1: ridge (Sr, Oxr)
    double* Ysum;              // K16 x ld
    float* Yout;               // K16 x ld unit rows, or null (a collective comes first)
    double* Sold;              // nblk x G x K16
    double* Sr;                // G x K16 x ld
    double* Oxr;               // G x K16
    double* zero_p;            // fill duties for the sweep kernel (or null): slot tables + sync words, objective accumulators
    size_t zero_n;
    double* zero2_p;
    size_t zero2_n;
    const unsigned* frozen;    // as Rtz3Args.frozen: a frozen engine keeps Y, Sold, Osave and the objective block of the failed round
    const double* copy_src;    // copy duty: O at the start of the round, kept for an exact replay (or null)
    double* copy_dst;
    int copy_n;
};
bool rtz3_ok(int mt, int dp, int nblk, int G);
int rtz3_ntb(int dp, int nblk);
int rtz3_slab_floats(int mt, int dp, int nblk);
bool rtz3b_ok(int mt, int dp, int nblk, int Kp);   // launch_rtz3 takes the bf16-pipe kernel k_rtz3c (one workgroup per CU)
int rtz3_quad(int mt, int dp, int nblk, int Kp, bool allow_bf16);   // 8: the tasks are cut for k_rtz3c's eight waves, 4: for k_rtz3c / k_rtz3
int launch_rtz3(const Rtz3Args& a, int mt, int dp, int nblk, hipStream_t s, bool allow_bf16, int quad);   // returns 1 when the bf16-pipe instance (k_rtz3c) ran, 0 for k_rtz3, -1 unsupported
void launch_rtz3_finish(const Rtz3FinishArgs& a, hipStream_t s);
bool rtzw_ok(int mt, int dp, int d, int nblk, int G);
int rtzw_nt(int dp, int d, int nblk);
int rtzw_slab_floats(int mt, int dp, int d, int nblk);
size_t rtzw_zf_tile_words(int dp);
bool rtzw2b_zf_ok(int mt, int dp);   // the pre-split Z planes fit k_rtzw2b's LDS
void launch_zplanes(const float* Z, int dp, int n_tiles, const int* tile_grp, const int* gstart, const int* s_tile_start, unsigned* Zf, hipStream_t s);
bool rtzw2b_ok(int mt, int dp, int d, int nblk);   // launch_rtzw takes the bf16-pipe kernel k_rtzw2b (one workgroup per CU)
int launch_rtzw(const Rtz3Args& a, int mt, int dp, int d, int nblk, hipStream_t s, bool allow_bf16);   // 1: k_rtzw2b ran, 0: an f32-input kernel, -1 unsupported
void launch_tile_blocks(const int* cells, const int* tile_grp, const int* blk_start, int nblk, int64_t n_pos_upper, const int* gstart,
                        const int* s_tile_start, unsigned char* tile_blk, hipStream_t s);

struct TableArgs {
    const double* O_prev;  // G x K16
    const double* S_add;   // G x K16 or null
    const double* S_sub;   // G x K16 or null
    double* O_out;         // G x K16 or null
    double* T_out;         // K16 or null
    float* rp;             // G x K16 or null
    float* lrp;
    double* obj_cross;     // scalar accumulator or null
    const int* group_cols; // G x V
    const float* Pr_b;
    const float* theta;
    const float* sigma;
    int G, B, V, K16;
};

struct RidgeSolveArgs {
    const double* S;       // G x K16 x lds
    const double* Ox;      // G x K16
    const double* T;       // K16
    const float* lamb;     // B+1
    const float* Pr_b;
    const int* group_cols;
    float* W;              // G x K16 x ldw
    double* scratch;       // general path: K16 x (B+1) x (B+1+d)
    float alpha;
    int lambda_est;
    int K, K16, G, B, V, d, lds, ldw;
};

struct ApplyArgs {
    const float* R;
    const float* Zorig;
    const float* W;        // G x K16 x ldw
    const unsigned* Wf;    // wide shapes, bf16 pipe: W as the A fragments of k_ridge_apply_wideb (launch_w_planes), or null
    float* Zcorr;
    float* Zcos;
    const int* cells;
    const int* tile_grp;
    int n_tiles;
    int Kp, K16, dp, ldw, mtd;
    const int* task_tile0; // k_ridge_apply2: one workgroup per task (null: k_ridge_apply)
    const int* task_tile1;
    const int* task_grp;
    int ntasks, ldw_lds;
};

struct OrderArgs {
    int64_t N, Ng, cpb;    // local cells, cells of the whole job, positions per block (global)
    const int* global_id;  // N (null = the internal index)
    int nblk, G, half_bits;
    uint32_t key0, key1;
    const int* gstart;     // G+1 first internal cell of every group
    int* chunk_tab;        // nchunks x (nblk*G): histogram, then exclusive offsets
    int* run_count;        // nblk*G
    int* run_start;        // nblk*G (padded positions)
    int* blk_start;        // nblk+1 (tiles)
    int* run_tiles;        // nblk*G + 1: first tile of every (block, group) run, then the tile count (or null)
    int* cells;            // padded list
    int* tile_grp;
    // for the streaming R^T.Z pass (k_rtz3): every cell's block id in STATIC tile order, written by the histogram pass
    // (position of internal cell c of group g: 16 * s_tile_start[g] + c - gstart[g]); null = not wanted
    unsigned char* tile_blk;
    const int* s_tile_start;
    const unsigned* frozen;  // non-zero: leave the lists alone (they may be the lists of a round that is about to be replayed), or null
};

size_t round_lds_bytes(int K16, int dp, int G, int B, int V, bool bf3, bool ga, int nblk);
bool round_uses_bf16_pipe(int K16, int dp, int G, int B, int V, bool extra_tiles, bool allow_bf16, bool ga, int nblk);   // which k_round instance launch_round picks
#define HMX_ROUND_GA_TILES 14  /* tiles per workgroup and block the group-affine map is planned for where the grid allows it (16 slots: the last wave, which runs the hand-off, then seldom carries tiles) */
#define HMX_ROUND_GA_MAX_BLOCK_CELLS (1 << 22)   /* a block's sums must fit 23 + 32 bits of fixed point below the count field */
// k_round has no static LDS and one workgroup per CU: everything the CU has (160 KB), less a small margin
#define HMX_ROUND_LDS_LIMIT (size_t)(159 * 1024)
int round_row_floats(int d);
int launch_round(const RoundArgs& a, int mt, int wgs, hipStream_t s, bool extra_tiles, bool allow_bf16);   // extra_tiles: a block holds more tiles than the grid's 16 slots per workgroup
size_t peer_box_doubles(int n_ranks, size_t GK);
void launch_peer_selftest(double* const* peer_box, double* my_box, int n_ranks, int rank, size_t GK, unsigned long long token, int iters,
                          unsigned* result, hipStream_t s);
// LISI (hmx_lisi.hip)
#define LISI_KNN_WAVES 4
#define LISI_MAX_NEIGHBOURS (2048 - 8)   /* 3 * perplexity of the largest candidate-list size */
struct LisiKnnArgs {
    const float* X;                 // npad x dp centred float32 rows (zero padded)
    const float* cn;                // npad squared norms, +inf for padding rows
    int64_t n, npad;
    int dp;
    int cap;                        // entries of a query's candidate list: 256, 1024 or 4096 (lisi_list_cap); the best cap / 2 survive a
                                    // compaction and are ranked exactly in float64 (8 of them are slack for rank inversions of float32)
    unsigned long long* lists;      // npad x cap (order bits of the key << 32 | candidate)
    int* counts;                    // n: entries of the final, sorted list
    unsigned long long* prof;       // cycle sums per loop segment (LISI_PROF builds), else null
};
struct LisiFinishArgs {
    const double* X;                // n x d float64 input
    int64_t n;
    int d, nn, n_labels;            // nn = neighbours asked of the search (the cell itself included)
    int cap;                        // as LisiKnnArgs.cap
    const unsigned long long* lists;
    const int* counts;
    const int* labels;              // n_labels x n category codes
    double perplexity, tol;
    double* out;                    // n x n_labels
    double* knn_dist;               // n x (nn-1) or null
    int* knn_idx;
};
void launch_lisi_prepare(const double* X, int64_t n, int64_t npad, int d, int dp, double* sums, float* X32, float* cn, hipStream_t s);
int launch_lisi_knn(const LisiKnnArgs& a, hipStream_t s);
int lisi_list_cap(int nn);          // 0: more neighbours than the largest list ranks
void launch_lisi_finish(const LisiFinishArgs& a, hipStream_t s);

// k-means++ seeding on the device (k_seed_*): n points of d floats, row-major X and its transpose Xt
struct SeedArgs {
    const float* X;               // n x d
    float* Xt;                    // d x n
    int n, d, n_trials;
    unsigned long long seed;
    float* closest;               // n: squared distance to the closest centre so far
    float* cand_min;              // 8 x n: min(closest, distance to candidate j)
    unsigned long long* chunk_sum;// ceil(n/256) fixed-point sums of closest
    unsigned long long* pots;     // K x 8 candidate potentials (zeroed by the caller)
    int* cand;                    // K x 8 candidate point indices
    int* chosen;                  // K chosen point indices
    float* centers;               // K x d
};
#define HMX_SEED_SLOTS 8
void launch_kmeans_seed(const SeedArgs& a, int K, hipStream_t s);
void launch_load_rows(const float* src, int d, const int* source_row, float* dst, int dp, int64_t N, hipStream_t s);
size_t kmeans_slab_floats(int wgs, int K16, int dp);
int launch_kmeans_step(const float* Zcos, const float* C, const float* hn, const int* cells, int n_tiles, float* slab, int K,
                       int K16, int dp, int ldy, int wgs, hipStream_t s);
void launch_kmeans_sums(const float* slab, int wgs, int K16, int dp, int d, double* sums, hipStream_t s);
void launch_kmeans_update(const double* sums, float* C, float* hn, int K, int K16, int d, int ldy, hipStream_t s);
void launch_kmeans_sums_from_stats(const double* Sr, const double* Oxr, int G, int K16, int ld, int d, double* sums, hipStream_t s);
void launch_gather_rows(const float* src, int ld, int cols, const int* rows, int n_rows, float* dst, hipStream_t s);
void launch_order(const OrderArgs& a, hipStream_t s);
int order_chunks(int64_t N);
void launch_normalize_rows(const float* Z, float* Zc, int64_t N, int dp, hipStream_t s);
void launch_y_normalize(const float* src, float* dst, int K, int K16, int d, int ldy, hipStream_t s);
size_t y_planes_dwords(int K16, int dp);
void launch_y_planes(const float* Y, int K16, int ldy, int dp, unsigned* Yf, hipStream_t s);
size_t w_planes_dwords(int G, int K16, int dp);
void launch_w_planes(const float* W, int G, int K16, int ldw, int dp, unsigned* Wf, hipStream_t s);
size_t assign_wide3_lds_bytes(int mt);
bool assign_wide3_fuses_table(int mt, int dp, int V);
bool sweep_wide3_ok(int mt, int dp, int V, int G, int nblk);
int launch_sweep_wide3(const AssignArgs& a, int wgs, hipStream_t s);   // k_sweep_wide3: all blocks of a wide sweep in one persistent launch
int launch_assign(const AssignArgs& a, bool penalty, int max_wgs, hipStream_t s);   // 1: the bf16-pipe wide instance ran, 0: another kernel, -1 unsupported
void rtz_geometry(int mt, int ntd, int* nsub, int* slab_per_wave);
bool rtz2_ok(int mt, int dp);
bool rtz_wide_ok(int mt, int dp);
int rtz_wide_slab_floats(int mt, int dp);
void launch_rtz_wide(const RtzArgs& a, int wgs, hipStream_t s);
void launch_rtz_wide_reduce(const float* slab, int nslabs, int mt, int dp, int K16, int ld, double* out, const int* task_grp, hipStream_t s);
int rtz2_slab_floats(int mt, int dp);
void launch_rtz2(const RtzArgs& a, int wgs, hipStream_t s);
void launch_rtz2_reduce(const float* slab, int nslabs, int mt, int dp, int K16, int ld, double* out, const int* task_grp, hipStream_t s);
void launch_rtz(const RtzArgs& a, int wgs, hipStream_t s);
void launch_y_normalize_d(const double* src, float* dst, int K, int K16, int d, int ldy, hipStream_t s);
void launch_rtz_reduce(const float* slab, int nwaves, int mt, int ntd, int K16, int ld, double* out,
                       const int* task_grp, hipStream_t s);
void launch_block_table(const TableArgs& a, int K16, hipStream_t s);
void launch_group_sums(const float* R, int Kp, int K, int K16, const int* cells, const int* tile_grp, int n_tiles,
                       double* Ogrp, hipStream_t s);
void launch_ridge_solve(const RidgeSolveArgs& a, hipStream_t s);
int launch_ridge_apply(const ApplyArgs& a, int max_wgs, hipStream_t s);
