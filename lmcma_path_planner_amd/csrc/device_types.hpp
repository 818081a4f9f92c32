// Internal declarations shared by the kernels and the C ABI of liblmcma_hip.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lmcma_hip.h"

namespace lmcma {

// Several independent optimisers ("queries": restarts or planning queries on one map) in one
// context.  Every buffer of a context exists once per query, at the same offset inside a
// per-query block of `stride` bytes, so query q's copy of any of them is the pointer of query 0
// plus q * stride.  A launch covers all queries: blockIdx.y is the query, and each kernel
// shifts the pointers of its argument block first.  queries = 1, stride = 0: a plain context.
struct Batch {
    int queries;
    size_t stride;
};

// ---- direct exchange between the shards of a population ------------------------------------
// Every shard owns a "mailbox" in fine-grained device memory that its peers can write (mapped
// through IPC handles, or plain pointers inside one process): room for all sorted costs and all
// partial rows, and one arrival counter per source and exchange.  A producer stores its block into
// every peer's mailbox, makes the stores visible at system scope and adds 1 to its counter there.
// On the receiving side a small launch (k_peer_gather) waits until every other source's counter
// has reached the value this exchange brings it to and copies the blocks into the context's
// ordinary buffers, which the consuming kernels then read as they read them after an all-gather
// (reads of fine-grained memory are not cached: the merge of the sorted lists took 16 us instead
// of 7 reading the mailbox itself, the update at the head of the sampling launch 30 instead of 15).
// Counters only grow (wrap-around compare).
constexpr int kMaxPeerRanks = 16;

struct PeerWait {
    const uint32_t *flags;  // [n] counters of this exchange in the own mailbox; null: nothing to wait for
    uint32_t want;          // value every other source's counter reaches with this exchange
    int n, self;            // sources; the own one is not waited for
    // word 128 of the mailbox: set to 1 + the source when a wait gives up (a peer never arrived); the
    // three words after it then hold the counter's value, the value wanted and the counter's index
    uint32_t *error;
};

struct PeerSend {
    // destination d of n: where this shard's block goes in that mailbox, and the counter to bump
    // (null: the shard's own copy, which needs no counter).  The kernel that produces the block
    // stores it there itself; every one of its workgroups that holds a part of the block adds 1 to
    // the counters when its part is on its way (`arrivals` workgroups per exchange).
    void *dst[kMaxPeerRanks];
    uint32_t *flag[kMaxPeerRanks];
    int n;
};

// Scalars the sampling kernel reads from device memory, so that a captured
// graph of one generation can be replayed without patching kernel arguments.
struct SampleScalars {
    float sigma;          // step size
    float mk;             // M^k, the factor on z itself in A z
    int32_t k;            // active direction vectors
    uint32_t generation;  // Philox stream selector
};

// fp64 optimiser state kept on the device (single copy per context).
struct OptScalars {
    double sigma;
    double s;            // success-rule accumulator
    double best_f;
    long long counteval;
    int32_t itr;
    int32_t iterator_sz;
    // constants derived once at init (L/lmcma.cpp:136-156, 266-272)
    double mueff, c1, cc, cs, val_target, K, M;
    int32_t imin_last;   // first recomputed vector of the last update (diagnostic)
    // 1 once sigma < 1e-20 (LMCMA::isBehaviorLearningDone, L/lmcma.cpp:426-429): the update
    // kernels then leave the state as it is, so itr stays at the generation that reached it
    int32_t done;
    double sigma0;       // step size lmcma_hip_opt_reset returns to
};

struct CostParams {
    const float *rows;   // [n_rows][ld]
    int ld;
    int n_rows;
    float *cost;         // [n_rows]
    float *terms;        // [n_rows][3] or nullptr
    const void *grid;
    int grid_type;       // LMCMA_GRID_BITS or LMCMA_GRID_F32
    int nx, ny, nz;
    float c_obs;
    int W;
    float start[3], goal[3];
    float c_start, c_goal;   // grid value at the fixed endpoints
    float w_len, w_grid, w_smooth;
    int pad_zero;
    int grid_words;      // uint32 words of a bit grid (0 when not staged)
};

// Lower-triangular factor of a structured sampling covariance (applyCovL, L/lmcma.cpp:857-864):
// z <- L z before the low-rank transform.  L is row-major fp32, (16 ntiles) x ld with
// ld >= 16 ntiles, zero outside the N x N block; kblocks[t] = 16-wide column blocks of rows
// 16 t .. 16 t + 15 that hold a non-zero (t + 1 for a full lower triangle).
struct CovFactor {
    const float *L;
    int ld, ntiles;
    uint8_t kblocks[32];
};

struct GenParams {
    float *rows;              // [n_local][ld], written
    const float *xmean, *V, *B, *lo, *hi;
    const SampleScalars *sc;
    int ld, n_local, m;
    uint32_t first_candidate, key0, key1;
    CostParams cost;          // cost.rows is not used: the rows come from LDS
    CovFactor cov;            // cov.L null: identity
    Batch batch;              // query q draws from the Philox key (key0 + q, key1)
    unsigned *fused_counter;  // the fused form (launch_generate's post): a zeroed word of the context
};

// Raises a kernel's dynamic LDS limit (hipFuncAttributeMaxDynamicSharedMemorySize) when `lds`
// exceeds what has been configured for it on the calling thread's current device; the sizes
// are kept per (device, kernel) under a lock (capi.hip).
hipError_t ensure_dynamic_lds(const void *kernel, size_t lds);

// ---- launchers (each in the .hip file that defines its kernels) -------------
struct UpdateArgs;
// pre: the argument block of this generation's update when its first part may ride along as one
// more workgroup of the launch (*pre_done tells whether it did); null: never
// post: the complete argument block of the PREVIOUS generation's update when its second part may
// run at the head of this launch, in every workgroup (*post_done tells whether it did); null: never
hipError_t launch_generate(hipStream_t s, const GenParams &g, int dims, bool lds_grid, const UpdateArgs *pre = nullptr,
                           bool *pre_done = nullptr, const UpdateArgs *post = nullptr, bool *post_done = nullptr);
// whether launch_generate would carry `post`
bool generate_fuses(const GenParams &g, bool lds_grid, const UpdateArgs *post);

// cov: null for the identity
hipError_t launch_sample(hipStream_t s, float *rows, const float *zbuf, const float *xmean,
                         const float *V, const float *B, const float *lo, const float *hi,
                         const SampleScalars *sc, int ld, int n_local, uint32_t first_candidate,
                         uint64_t seed, const CovFactor *cov);

// Exact Euclidean distance transform (k_edt.hip).  occ: [nz][ny][nx] bytes on the device;
// d2: [2][cells] scratch; dist, cost: [cells] floats or null; cost = 1 / max(dist, d_min).
hipError_t launch_edt(hipStream_t s, const uint8_t *occ, int nx, int ny, int nz, double d_min, uint32_t *d2,
                      float *dist, float *cost);

// lds_grid: stage a bit grid in LDS when it fits
hipError_t launch_cost(hipStream_t s, const CostParams &p, int dims, bool lds_grid);

constexpr int kSortTile = 4096;
constexpr int kRankCountMax = 16384;  // populations up to this size are ranked by counting
int select_num_tiles(int lambda);
size_t select_workspace_bytes(int lambda);
bool select_by_counting(int lambda);
// rows of block_partials the counting path can write for a local candidate range
int select_max_partial_rows(int first_candidate, int n_local);

struct RecombineArgs {
    const float *rows;
    int ld, N, first_candidate, n_local;
    const double *weights;
    int mu;
    double *block_partials;
};

// Ranking of all lambda costs (ascending, ties in candidate order) into sorted_cost / perm.
// With opt, prev_sorted and psr_accum: when opt->itr > 0 the launch also counts, for every
// candidate of this generation, the entries of prev_sorted (last generation's ascending
// costs) below its cost (population success rule).  The sorting path adds the counts into
// *psr_accum; the counting path (populations up to kRankCountMax) leaves one count per
// workgroup in the first words of keys_ws (SelectResult::psr_parts of them).
// rc (counting path only): the same launch adds the elite rows of the local candidates
// into SelectResult::partial_rows rows of rc->block_partials.
struct UpdateArgs;
struct SelectArgs {
    const float *cost_all;
    int lambda;
    uint64_t *keys_ws;             // select_workspace_bytes(lambda) bytes
    float *sorted_cost;
    int32_t *perm;
    const OptScalars *opt;
    const float *prev_sorted;
    unsigned long long *psr_accum;
    const RecombineArgs *rc;
    // the first part of the update (update_pre_body) may ride along as one more workgroup of
    // the counting launch; SelectResult::pre_done tells whether it did
    const UpdateArgs *pre;
    Batch batch;                   // counting path only
    // counting path: > 0 = the population is ranked in tiles of this many keys, each tile among
    // itself (sorted_cost and perm then hold one ascending list and one tile-local permutation per
    // tile, back to back); lambda must be a multiple of it.  launch_rank_merge combines the tiles.
    int tile_len;
};
struct SelectResult {
    int partial_rows;   // rows of block_partials written (0: recombination not done)
    int psr_parts;      // per-workgroup success-rule counts left in keys_ws (0: see psr_accum)
    int pre_done;
};
hipError_t launch_select(hipStream_t s, const SelectArgs &a, SelectResult *res);
hipError_t launch_select_sort(hipStream_t s, const SelectArgs &a);

// One shard of a population of G = lambda / n_local contexts, after the exchange of the
// contexts' ASCENDING local costs (`lists`, context g's block at g n_local): global ranks of
// the local candidates by merging, their share of the success-rule count, the elite rows
// (one row of block_partials per workgroup, *rows_out of them) and the population's lowest
// cost.  sorted_cost / perm receive the local candidates' entries only.
struct MergeArgs {
    const float *lists;
    // owners = 1: the caller owns list first_candidate / n_local (a shard of a population);
    // owners = lambda / n_local: it owns all of them (one context that ranked its population in
    // tiles): local_perm, prev_own and own_out then cover all lists, rows the whole population
    int owners;
    int lambda, n_local, first_candidate;
    const int32_t *local_perm;    // [n_local] local candidate at local position p (the local ranking's perm)
    float *sorted_cost;
    int32_t *perm;
    const OptScalars *opt;        // success-rule count when opt->itr > 0
    const uint32_t *prev_own;     // [n_local] keys of this context's list of last generation
    uint32_t *own_out;            // [n_local] keys of this context's list of this generation
    uint32_t *psr_parts;          // [n_local / 32] per-workgroup success-rule counts
    float *best_out;
    const float *rows;
    int ld, N;
    const double *weights;
    int mu;
    double *block_partials;
};
bool rank_merge_possible(int lambda, int n_local, int first_candidate);
// tile length for ranking a whole population in one context by tiles + merge, 0: not this way
int rank_tile_len(int lambda);
hipError_t launch_rank_merge(hipStream_t s, const MergeArgs &a, int *rows_out);

constexpr int kRecombineRanksPerBlock = 32;
int recombine_num_blocks(int mu);
hipError_t launch_recombine(hipStream_t s, const float *rows, int ld, int N, int first_candidate,
                            int n_local, const int32_t *perm, const double *weights, int mu,
                            double *block_partials, Batch batch = Batch{1, 0});

// partial_local[N] = sum of the first `rows` rows of block_partials, in the order in which a
// workgroup of `threads` threads adds them (the update adds its partial rows in the same
// order, so reducing first and updating from the one row gives the same bits)
// extras != nullptr: partial_local has N + 2 entries, the last two being the context's
// success-rule count (psr_parts[0 .. n_parts) + *psr_accum, or 0 when !count_psr) and *best
struct FinalExtras {
    const uint32_t *psr_parts;
    int n_parts;
    const unsigned long long *psr_accum;
    int count_psr;
    const float *best;
    // the finished row (N + 2 entries) also goes into these mailboxes; *arrivals_out = workgroups
    // that bump each destination's counter.  n = 0: nowhere.
    PeerSend send;
    // The receiving side in the same launch (k_recombine_final_wide only; flags = null: not here):
    // after its own stores every workgroup waits for all sources' row counters, the shard's own
    // included (its strips come from the launch's other workgroups), and the launch copies the
    // gathered rows, gather_quads 16-byte units, from the mailbox into the context's buffer.
    PeerWait gather;
    const uint4 *gather_src;
    uint4 *gather_dst;
    int gather_quads;
};
// whether launch_recombine_final(threads = 0) takes the many-workgroup kernel (which can also receive)
bool recombine_final_is_wide(int rows, int N);
// threads: the workgroup size whose summation order the result must have (a row that feeds the
// context's own update: the update's); 0: any fixed order will do (a row that goes into an exchange)
hipError_t launch_recombine_final(hipStream_t s, const double *block_partials, int rows, int N,
                                  double *partial_local, int threads, const FinalExtras *extras, int *arrivals_out = nullptr);
// The sum over many partial rows, ahead of the update: rows -> reduce_rows_out(rows, N) <=
// kReducedRowsMax rows of dst (one row for even N); used above kReduceRowsFrom rows.
constexpr int kReducedRowsMax = 32;
constexpr int kReduceRowsFrom = 192;
int reduce_rows_out(int rows, int N);
hipError_t launch_reduce_rows(hipStream_t s, const double *src, int rows, int N, double *dst, Batch batch);

// workgroup size the update uses for m direction vectors of N parameters
int update_threads(int N, int m);

struct UpdateArgs {
    OptScalars *opt;
    SampleScalars *sc;
    int N, ld, lambda, m;
    const double *partials;  // [n_partials][partial_stride], the first N entries of a row are summed
    int n_partials, partial_stride;
    // 1: entries [N] (success-rule count) and [N + 1] (lowest cost) of the rows carry the scalars
    // of a sharded population; 0: they come from psr_accum / psr_parts / sorted_cost[0]
    int scalars_in_rows;
    const float *sorted_cost;   // [lambda] this generation, ascending
    // success rule: sum over this generation's candidates of #{previous costs below it} =
    // *psr_accum (read and zeroed by the update) + psr_parts[0 .. n_psr_parts)
    unsigned long long *psr_accum;
    const uint32_t *psr_parts;
    int n_psr_parts;
    double *xmean, *pc;         // [N]
    double *pc_arr, *v_arr;     // [m][N]
    double *nj, *lj;            // [m]
    int32_t *t, *vec, *iterator; // [m]
    float *xmean_f;             // [ld]
    float *V_f, *B_f;           // [m][ld] in iterator order, B = M^(k-1-t) nj pc
    Batch batch;
    // a word in host memory the device can write (or null): set to 3 when an update that runs at the
    // head of a sampling launch gives up its bounded wait, so that the host sees it without a copy
    int *host_flag;
};
hipError_t launch_update(hipStream_t s, const UpdateArgs &a);
// The update in two launches (update_body.hpp): vectors that do not depend on this generation's
// costs, then the rest.  update_is_split: the sizes the two-part kernels cover.
bool update_is_split(int N, int m);
// the first part is short enough to run as an extra workgroup of the ranking launch; otherwise it
// rides in the sampling launch (launch_generate) or stays in the single update kernel
bool update_pre_rides_with_ranking(int m);
int update_pre_threads(int m);  // threads of the first part that do vector work
hipError_t launch_update_pre(hipStream_t s, const UpdateArgs &a);
hipError_t launch_update_post(hipStream_t s, const UpdateArgs &a);

}  // namespace lmcma
