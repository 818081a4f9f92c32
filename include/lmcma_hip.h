/*
 * lmcma_hip.h — C ABI of the MI355X (gfx950) LM-CMA population path.
 *
 * The reference optimiser has no FFI: callers include lmcma.hpp and link
 * lmcma.cpp (lmcma_path_planner/CMakeLists.txt:39).  The functions below are
 * what a reference-side binding calls from inside LMCMA::sample(),
 * LMCMA::update() and CMABase::setEvaluationFeedback() to run the population
 * work on the GPU.  Each entry point names the reference code it replaces as
 *   L/ = lmcma_path_planner/src/
 *   S/ = sample_based_optimisation_based_path_planner/src/
 *
 * Conventions
 *   - plain C types only; every function returns LMCMA_HIP_OK (0) or a negative
 *     error code, and lmcma_hip_last_error() returns the message of the calling
 *     thread's last failure.  No exception crosses this boundary.
 *   - one context owns one HIP stream (its own, or the caller's) and one set of
 *     device buffers.  A context is not re-entrant; distinct contexts are
 *     independent, like distinct LMCMA objects in the reference
 *     (RNG state is per instance, L/lmcma.hpp:62).
 *   - host pointers are caller-owned and are copied before the call returns
 *     unless a function says otherwise.
 *   - candidate rows hold n_params = dims * waypoints values, dimension-major
 *     ([dim0: wp0..wpW-1 | dim1: ...], the block layout of L/lmcma.cpp:786-791),
 *     and the population is row-major [lambda][n_params] like arx
 *     (L/lmcma.cpp:138,308).
 */
#ifndef LMCMA_HIP_H
#define LMCMA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LMCMA_HIP_OK 0
#define LMCMA_HIP_ERR_ARG (-1)         /* invalid argument                      */
#define LMCMA_HIP_ERR_RUNTIME (-2)     /* a HIP runtime call failed             */
#define LMCMA_HIP_ERR_STATE (-3)       /* call made in the wrong order          */
#define LMCMA_HIP_ERR_UNSUPPORTED (-4) /* size outside what the kernels handle  */

enum { LMCMA_Z_HOST = 0, LMCMA_Z_PHILOX = 1 };
enum { LMCMA_GRID_BITS = 0, LMCMA_GRID_U8 = 1, LMCMA_GRID_F32 = 2 };
enum { LMCMA_PAD_ENDPOINT = 0, LMCMA_PAD_ZERO = 1 };
enum { LMCMA_COST_TRAJECTORY = 0, LMCMA_COST_EXTERNAL = 1 };

/* limits of the kernels (checked by lmcma_hip_create) */
#define LMCMA_HIP_MAX_PARAMS 512
#define LMCMA_HIP_MAX_VECTORS 64            /* device-resident update (lmcma_hip_opt_init)   */
#define LMCMA_HIP_MAX_VECTORS_STEPWISE 4096 /* stepwise API: the caller keeps the state      */
#define LMCMA_HIP_MAX_LAMBDA (1 << 20)
#define LMCMA_HIP_MAX_QUERIES 256

typedef struct lmcma_hip_ctx lmcma_hip_ctx;

typedef struct {
    int32_t struct_size;     /* sizeof(lmcma_hip_config), for ABI checking           */
    int32_t device;          /* HIP device ordinal                                   */
    int32_t dims;            /* D in 1..3 for a trajectory cost; 0 = plain vector    */
    int32_t waypoints;       /* W                                                    */
    int32_t n_params;        /* N; must equal dims*waypoints when dims > 0           */
    int32_t lambda_total;    /* population size over all contexts of one optimiser   */
    int32_t lambda_local;    /* candidates this context samples and scores           */
    int32_t first_candidate; /* global index of this context's candidate 0           */
    int32_t m;               /* direction vectors kept (reference: = lambda,         */
                             /* L/lmcma.cpp:266)                                     */
    int32_t mu;              /* elites; reference: lambda/2 (L/lmcma.cpp:136)        */
    int32_t z_source;        /* LMCMA_Z_HOST or LMCMA_Z_PHILOX                       */
    int32_t cost_source;     /* LMCMA_COST_TRAJECTORY or LMCMA_COST_EXTERNAL         */
    int32_t pad_mode;        /* acceleration stencil at the path ends                */
    int32_t queries;         /* independent optimisers in this context (0 = 1), see  */
                             /* "several queries in one context" below               */
    uint64_t seed;           /* Philox key (query q draws from seed + q)             */
    double w_len, w_grid, w_smooth;
    double start[3], goal[3];
    void *stream;            /* hipStream_t to run on; NULL: the context makes one   */
} lmcma_hip_config;

const char *lmcma_hip_last_error(void);
int lmcma_hip_version(void);

/* Replaces the allocations of CMABase::init / LMCMA::init (L/lmcma.cpp:138-142,
 * 274-290) with device buffers. */
int lmcma_hip_create(const lmcma_hip_config *cfg, lmcma_hip_ctx **out);
int lmcma_hip_destroy(lmcma_hip_ctx *ctx);

/* The cost map the candidates are scored against: the role of EDT_Matrix
 * (S/...path_planner.cpp:37) behind ValidityChecker::clearance (:605-628).
 * cells is indexed [z][y][x] (row = y, col = x, :599-602).
 *   LMCMA_GRID_BITS  uint32 words, cell c is bit (c & 31) of word c >> 5;
 *                    an occupied cell costs c_obs, a free one 0
 *   LMCMA_GRID_U8    one byte per cell, non-zero = occupied; packed to bits
 *   LMCMA_GRID_F32   one float per cell holding the state cost itself,
 *                    e.g. 1 / max(clearance, d_min)  (:667)                     */
int lmcma_hip_set_grid(lmcma_hip_ctx *ctx, const void *cells, int nx, int ny,
                       int nz, int cell_type, double c_obs);

/* The same cost map built on the device from an occupancy grid: what the
 * reference does with dynamicEDT3D before planning (S/...path_planner.cpp:
 * 299-321: DynamicEDT3D::initializeMap + update, then getDistance of every
 * cell into EDT_Matrix) followed by the state cost 1 / clearance (:655-668).
 * occ is [nz][ny][nx] bytes, non-zero = occupied; the distance of a cell is the
 * exact Euclidean distance, in cells, to the nearest occupied one (0 on it; 1e6
 * when the grid holds none); the installed map is LMCMA_GRID_F32 with
 * 1 / max(distance, d_min) per cell.  dist_out (optional, host, cells floats)
 * receives the distances.  Extents up to 1024 per axis.                        */
int lmcma_hip_set_grid_from_occupancy(lmcma_hip_ctx *ctx, const uint8_t *occ, int nx,
                                      int ny, int nz, double d_min, float *dist_out);

/* loBounds / hiBounds of the constructor (L/lmcma.hpp:131-133), applied like
 * CMABase::applyBoundaries (L/lmcma.cpp:220-230).  Either may be NULL.         */
int lmcma_hip_set_bounds(lmcma_hip_ctx *ctx, const double *lo, const double *hi);

/* Uploads the search state LMCMA::sample reads (L/lmcma.cpp:301-311, 431-447):
 * xmean[N], sigma, and the k active direction pairs in iterator order:
 * V[k][N] = v_arr rows, PC[k][N] = pc_arr rows, nj[k] = Nj_arr entries,
 * M = sqrt(1-c1).  The library folds the scalar recurrence of computeAz into
 *   Az = M^k z + sum_t M^(k-1-t) nj[t] (V[t].z) PC[t].                         */
int lmcma_hip_set_state(lmcma_hip_ctx *ctx, const double *xmean, double sigma,
                        const double *V, const double *PC, const double *nj,
                        int k, double M);

/* A factor applied to every candidate's normals before the low-rank transform:
 * z <- Lm z, in fp32 on the matrix cores.  L[c*N + r] = Lm(r, c), i.e. Lm
 * column-major: the layout cholesky() leaves the factor of the constructor's
 * `covariance` argument in (L/lmcma.hpp:131-133, L/lmcma.cpp:165-169, 844-855),
 * which sampleStandardNormal + applyCovL apply the same way (:212-218,
 * 857-864); any dense matrix is accepted (CMAChol::sample multiplies by its
 * full factor A, L/lmcma.cpp:649-664).  NULL returns to the identity.  The
 * factor applies to device-drawn normals (z_source PHILOX) and to normals
 * handed to lmcma_hip_sample alike; a caller that has applied its factor on
 * the host simply does not set one.                                            */
int lmcma_hip_set_cov_factor(lmcma_hip_ctx *ctx, const double *L);

/* LMCMA::sample (L/lmcma.cpp:301-311) for this context's candidates.
 * z_host: [lambda_local][N] standard normals drawn by the caller (z_source
 * HOST, e.g. from random_Gauss, L/lmcma.cpp:63-82, with applyCovL already
 * applied); NULL with z_source PHILOX, where generation selects the stream.   */
int lmcma_hip_sample(lmcma_hip_ctx *ctx, uint32_t generation, const double *z_host);

/* The caller's loop body between getNextParameterVector and
 * setEvaluationFeedback (L/example_lmcma.cpp:51-54), for the trajectory cost:
 * length + grid + smoothness terms summed left to right like
 * CMABase::setEvaluationFeedback (L/lmcma.cpp:186-188).                        */
int lmcma_hip_evaluate(lmcma_hip_ctx *ctx);

/* Costs computed by the caller (cost_source EXTERNAL): arfitness of
 * L/lmcma.cpp:190 for candidates [first, first+count) of the whole population. */
int lmcma_hip_set_costs(lmcma_hip_ctx *ctx, int first, int count, const double *costs);

/* myqsort(lambda, arfitness, arindex, ...) (L/lmcma.cpp:91-104, called :315):
 * ascending, equal costs keep candidate order.  Outputs (host, may be NULL):
 * sorted_cost[lambda_total], perm[lambda_total].  Needs the costs of all
 * lambda_total candidates (own ones from evaluate, others via set_costs or the
 * device buffer LMCMA_BUF_COST_ALL).                                            */
int lmcma_hip_select(lmcma_hip_ctx *ctx, float *sorted_cost, int32_t *perm);

/* Weighted recombination of the mu best rows (L/lmcma.cpp:321-326) over the
 * candidates this context owns; with lambda_local == lambda_total this is the
 * new xmean.  weights[mu]; xmean_out[N] (host, may be NULL).                   */
int lmcma_hip_recombine(lmcma_hip_ctx *ctx, const double *weights, double *xmean_out);

/* CMABase::getNextParameterVector (L/lmcma.cpp:172-182) for a block of local
 * candidates: out[count][N].                                                   */
int lmcma_hip_get_rows(lmcma_hip_ctx *ctx, int first_local, int count, double *out);
/* local costs, and the three unweighted terms (J_len, J_grid, J_smooth)       */
int lmcma_hip_get_costs(lmcma_hip_ctx *ctx, int first_local, int count, double *out);
int lmcma_hip_get_terms(lmcma_hip_ctx *ctx, int first_local, int count, double *out3);
/* score caller-supplied rows (host, float, [count][N]) without touching the
 * population; count <= lambda_local.  cost_out[count], terms_out[count][3] or NULL */
int lmcma_hip_score_rows(lmcma_hip_ctx *ctx, const float *rows, int count,
                         float *cost_out, float *terms_out);

/* The ranking of lmcma_hip_select for callers that cannot allocate device memory (code under
 * HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).  No counterpart
 * in the reference.  All pointers are device memory of the calling thread's current device;
 * `stream` (a hipStream_t, NULL = the null stream) belongs to that device.  Ascending order,
 * equal costs in candidate order, NaN as +inf, exactly like lmcma_hip_select:
 * sorted_cost_dev[lambda], perm_dev[lambda].  The kernels are queued on the stream; nothing
 * is synchronised, and neither hipMalloc/hipFree nor any allocating routine is called.
 * The sorted key tiles live in a fixed 1 MiB array of the library's code object on each GPU
 * (held until the process ends), hence lambda <= LMCMA_HIP_RANK_NOALLOC_MAX_LAMBDA (larger:
 * LMCMA_HIP_ERR_UNSUPPORTED).  Every call on a GPU shares that GPU's array: calls that overlap
 * on one GPU (different streams, or threads) race; calls on one stream are safe back to
 * back.  The array's address is looked up on a device's first lmcma_hip_create or first call
 * here, whichever comes first: make that one outside a stream capture.               */
#define LMCMA_HIP_RANK_NOALLOC_MAX_LAMBDA 131072
int lmcma_hip_rank_noalloc(const float *cost_dev, int lambda, float *sorted_cost_dev,
                           int32_t *perm_dev, void *stream);

/* ---- device-resident optimiser: L/lmcma.cpp:313-424 without leaving the GPU -- */

/* CMABase::init + LMCMA::init (L/lmcma.cpp:130-170, 262-296): xmean0[N] (the
 * initialParams), sigma0; weights, mueff, c1, cc, K, M are derived inside.
 * Does not sample.                                                             */
int lmcma_hip_opt_init(lmcma_hip_ctx *ctx, const double *xmean0, double sigma0);

/* Puts the optimiser back into the state the last lmcma_hip_opt_init left (a new
 * LMCMA object with the same arguments, L/lmcma.cpp:233-239, 262-296), queued on
 * the stream without synchronising: for planning query after query on one map. */
int lmcma_hip_opt_reset(lmcma_hip_ctx *ctx);

/* n whole generations: sample -> evaluate -> select -> recombine -> update
 * (the body of CMABase::setEvaluationFeedback's every-lambda-th call,
 * L/lmcma.cpp:199-204, plus the caller's cost loop), queued on the stream.
 * Needs lambda_local == lambda_total, z_source PHILOX, cost_source TRAJECTORY.
 * The reference's caller stops when isBehaviorLearningDone() (sigma < 1e-20,
 * L/lmcma.cpp:426-429); here the update kernels evaluate that rule: once it has
 * fired the state is left as it is, so generations queued beyond it change
 * nothing and lmcma_hip_opt_summary.itr is the generation that reached it.
 * The first call on a context also records the HIP graphs later calls replay
 * (32, 8, 2 and 1 generations; a call is cut into these), and a call length that
 * is none of them gets a graph of the whole call, recorded behind the first call
 * of that length and replayed by later ones (up to four lengths of at most 256).
 * Inside a call the update of one generation may run at the head of the next generation's
 * sampling launch; the state in memory is complete when the call's work has finished, as with
 * the stepwise calls, and the results are the same bits.                                  */
int lmcma_hip_run(lmcma_hip_ctx *ctx, int n_generations);

/* ---- several queries in one context ------------------------------------------------
 * One optimiser's generation is a chain of two or three dependent launches that leaves most of the
 * GPU idle.  A context created with queries = Q > 1 holds Q independent optimisers on the same
 * map, start, goal, bounds and sizes (restarts of one planning problem from different seeds or
 * initial paths: the reference's caller would run its ask/tell loop Q times, L/example_lmcma.cpp:
 * 47-60) and advances all of them with every launch: lmcma_hip_run(ctx, n) = n generations of
 * each.  Query q samples from the Philox key seed + q, so it is the run a one-query context
 * with that seed makes: bit for bit with fewer than 12 queries; from 12 queries on every
 * population is ranked by one workgroup and its elite rows are added in rank order, from 16 on
 * the products of the sampling run on the matrix cores (same normals, same ranking, sums in
 * another order: equal to rounding).  Needs lambda_local == lambda_total <= 16384, z_source
 * PHILOX, cost_source TRAJECTORY, m <= 64; the stepwise calls (set_state ... recombine) are not
 * available.  lmcma_hip_opt_init starts every query from the same mean and step size;
 * lmcma_hip_opt_set_query gives one query its own (and puts that query back to generation 0).
 * The stop rule freezes each query on its own.  The readers (get_rows, get_costs, get_terms,
 * opt_get, opt_get_vectors, buffer) address the query chosen by lmcma_hip_query_select
 * (0 at creation).                                                                        */
int lmcma_hip_opt_set_query(lmcma_hip_ctx *ctx, int query, const double *xmean0, double sigma0);
int lmcma_hip_query_select(lmcma_hip_ctx *ctx, int query);
int lmcma_hip_queries(lmcma_hip_ctx *ctx);

/* The same generation in pieces, for a population sharded over contexts:
 *   phase_local   : sample + evaluate own candidates  -> LMCMA_BUF_COST_LOCAL
 *   (exchange costs into LMCMA_BUF_COST_ALL of every context)
 *   phase_rank    : ranks of the own candidates among all costs + partial
 *                   recombination of own elites      -> LMCMA_BUF_PARTIAL_LOCAL
 *   (exchange partial sums into LMCMA_BUF_PARTIAL_ALL, one row per context)
 *   phase_update  : LMCMA::update's remainder, identical on every context
 * When the shards are equal blocks (lambda_total a multiple of lambda_local,
 * first_candidate a multiple of lambda_local, lambda_local a multiple of 64 and
 * at most 32768) a context exchanges its costs in ASCENDING order: phase_local
 * also ranks the local candidates among themselves, LMCMA_BUF_COST_LOCAL holds
 * the sorted local costs, and phase_rank merges the contexts' lists, so that a
 * context's work does not grow with the square of the whole population
 * (lmcma_hip_get_costs still returns the costs in candidate order).  After
 * phase_rank LMCMA_BUF_SORTED_COST / LMCMA_BUF_PERM then hold the entries of the
 * local candidates only.  All contexts of a population take the same path.    */
int lmcma_hip_phase_local(lmcma_hip_ctx *ctx);
int lmcma_hip_phase_rank(lmcma_hip_ctx *ctx);
int lmcma_hip_phase_update(lmcma_hip_ctx *ctx, int n_partials);

/* ---- a population sharded over the GPUs of a node: the loop behind this boundary ------
 * The reference is one process with no exchange step (its caller's loop is
 * L/example_lmcma.cpp:47-60); a population of lambda_total candidates sharded over G
 * processes, one GPU each, exchanges twice per generation: the shards' sorted costs and
 * one row of N + 2 doubles per shard (weighted elite sum, success-rule count, lowest cost).
 * These entry points run that loop inside the library with RCCL all-gathers over xGMI queued
 * on the context's stream between the phases, so a generation needs no host code of the
 * caller's.  RCCL (librccl.so.1) is opened with dlopen on first use: a single-GPU caller
 * needs no RCCL at all.
 *
 * lmcma_hip_comm_unique_id   ncclGetUniqueId; one process calls it and hands the 128 bytes
 *                            to the others by its own means (MPI, a file, torch.distributed)
 * lmcma_hip_comm_init        ncclCommInitRank on the context's device; collective: every rank
 *                            of the population calls it with the same id.  The context must
 *                            be rank's shard: lambda_total = world * lambda_local,
 *                            first_candidate = rank * lambda_local, world <= 64
 * lmcma_hip_comm_attach      the same with a communicator the caller already has
 *                            (nccl_comm is an ncclComm_t; it is used, not owned)
 * lmcma_hip_comm_destroy     ncclCommDestroy of a communicator made by comm_init;
 *                            lmcma_hip_destroy does it too
 * lmcma_hip_run_sharded      n generations: phase_local, all-gather of the costs (in place:
 *                            the shard's block inside the buffer of all costs), phase_rank,
 *                            all-gather of the rows, phase_update(world).  Every rank makes
 *                            the same calls; the replicated state stays bit-identical.
 *                            A one-rank communicator gives the bits of lmcma_hip_run.      */
#define LMCMA_HIP_UNIQUE_ID_BYTES 128
int lmcma_hip_comm_unique_id(char id[LMCMA_HIP_UNIQUE_ID_BYTES]);
int lmcma_hip_comm_init(lmcma_hip_ctx *ctx, const char id[LMCMA_HIP_UNIQUE_ID_BYTES], int world, int rank);
int lmcma_hip_comm_attach(lmcma_hip_ctx *ctx, void *nccl_comm, int world, int rank);
int lmcma_hip_comm_destroy(lmcma_hip_ctx *ctx);
int lmcma_hip_run_sharded(lmcma_hip_ctx *ctx, int n_generations);

/* ---- the same loop with a direct exchange between the shards (no reference counterpart) -----
 * Two all-gathers of a few KB per generation cost their latency, not their bandwidth.  Here every
 * shard owns a "mailbox" in fine-grained device memory that the other shards map (an IPC handle
 * across processes, the plain pointer inside one process); a shard stores its sorted costs and
 * its row of N + 2 doubles straight into the peers' mailboxes over xGMI and bumps an arrival
 * counter there, and the kernels that consume them (the merge of the sorted lists, the update)
 * wait for the counters at their head.  Bit-identical to the RCCL form.
 *
 * lmcma_hip_peer_handle      allocates the context's mailbox and fills `handle`; the caller hands
 *                            the 128 bytes of every shard to every shard (as it does with the
 *                            RCCL id)
 * lmcma_hip_peer_connect     handles = world blobs in rank order (its own included); maps the
 *                            peers' mailboxes.  From then on lmcma_hip_run_sharded exchanges
 *                            directly (a communicator the context may also have is not used).
 *                            world <= 16.  Every shard must connect before any shard runs, and
 *                            all shards make the same run_sharded calls.  A shard's own handle in
 *                            another shard's place makes that shard an image of this one (its
 *                            blocks land in that place too): one shard timed with the traffic
 *                            and the waits of `world` shards; the optimisation is then not one.
 * lmcma_hip_peer_enable      on = 0: run_sharded goes back to the communicator while the mailboxes stay
 *                            mapped; on = 1: the direct exchange again.  Every shard makes the same
 *                            switch between the same two run_sharded calls.
 * lmcma_hip_peer_disconnect  unmaps; run_sharded is back to the communicator
 * lmcma_hip_peer_status      after the context's work has finished: whether the mailbox is
 *                            fine-grained memory, and whether a wait for a peer gave up (non-zero:
 *                            1 + the source that did not arrive within 4 s; the state is then
 *                            meaningless; lmcma_hip_opt_get reports it as an error too)
 * lmcma_hip_run_sharded_group  G shards held by ONE process (ctxs[r] = shard r, on any devices):
 *                            the loop with the two exchanges as device-to-device copies ordered
 *                            by events between the contexts' streams.  No communicator, no
 *                            mailbox, no kernel waits for another.                          */
#define LMCMA_HIP_PEER_HANDLE_BYTES 128
int lmcma_hip_peer_handle(lmcma_hip_ctx *ctx, char handle[LMCMA_HIP_PEER_HANDLE_BYTES]);
int lmcma_hip_peer_connect(lmcma_hip_ctx *ctx, const char *handles, int world, int rank);
int lmcma_hip_peer_enable(lmcma_hip_ctx *ctx, int on);
int lmcma_hip_peer_disconnect(lmcma_hip_ctx *ctx);
int lmcma_hip_peer_status(lmcma_hip_ctx *ctx, int *fine_grained, int *wait_error);
int lmcma_hip_run_sharded_group(lmcma_hip_ctx **ctxs, int world, int n_generations);

/* ---- the rows exchange: every rank holds the whole population --------------------------
 * The other way to spread a population over G ranks: the candidates' ROWS are all-gathered,
 * not the ranks' partial sums.  Every rank's context is created for the whole population
 * (lambda_local = lambda_total, first_candidate = 0), samples and scores only its range
 * [rank * lambda_total / G, (rank + 1) * lambda_total / G), receives the other ranges' rows
 * and costs by all-gather (in place, at their positions in the context's buffers), and then
 * ranks, recombines and updates exactly as a single context does: L/lmcma.cpp:313-329
 * (the mean from the mu best rows of arx in rank order) over the whole of arx.  A
 * candidate's row depends on its global index only (the Philox counter), so the state and
 * every row are those of lmcma_hip_run on one context, bit for bit, for any G.  The price
 * is the traffic (lambda_total * row stride * 4 bytes received per rank and generation
 * instead of G * (N + 2) * 8) and that every rank ranks the whole population; the
 * partial-sum exchange above is the fast form, this one the form whose result does not
 * depend on G.
 *
 * lmcma_hip_phase_local_range  sample + score candidates [first, first + count) of a
 *                            context that ranks its candidates itself (first and count
 *                            multiples of 4); the other rows and costs stay as they are.
 *                            Then: copy the other ranges in (LMCMA_BUF_ROWS_LOCAL,
 *                            LMCMA_BUF_COST_ALL), phase_rank, phase_update(0).
 * lmcma_hip_comm_init / comm_attach  accept such a context when world > 1, lambda_total is a
 *                            multiple of world and the range a multiple of 16 candidates;
 *                            lmcma_hip_run_sharded then runs this form (two all-gathers per
 *                            generation: rows, costs).
 * lmcma_hip_run_sharded_group  likewise when the G contexts each hold the whole population:
 *                            the ranges travel as device-to-device copies.
 * lmcma_hip_get_terms of such a context covers the own range only.                          */
int lmcma_hip_phase_local_range(lmcma_hip_ctx *ctx, int first, int count);

typedef struct {
    int32_t itr;          /* generations completed (L/lmcma.hpp:60)              */
    int32_t iterator_sz;  /* active direction vectors                            */
    int64_t counteval;    /* L/lmcma.hpp:64                                      */
    double best_f;        /* L/lmcma.hpp:65                                      */
    double sigma;
    double s;             /* success-rule accumulator                            */
    int32_t done;         /* LMCMA::isBehaviorLearningDone, L/lmcma.cpp:426-429:  */
                          /* sigma < 1e-20 was reached in generation itr          */
    int32_t reserved;
} lmcma_hip_opt_summary;

/* synchronises the stream */
int lmcma_hip_opt_get(lmcma_hip_ctx *ctx, lmcma_hip_opt_summary *summary,
                      double *xmean /* [N] or NULL */);
/* full fp64 state for parity checks; any pointer may be NULL.
 * v_arr/pc_arr: [m][N] slot-indexed like the reference; nj/lj: [m];
 * iterator: [m]; pc: [N].  Read it between generations: from phase_local to
 * phase_update of one generation part of the direction store is already that
 * of the next state (the rebuild of the older vectors runs beside the ranking). */
int lmcma_hip_opt_get_vectors(lmcma_hip_ctx *ctx, double *pc, double *v_arr,
                              double *pc_arr, double *nj, double *lj,
                              int32_t *iterator);

/* device buffers, for collectives run by the caller (RCCL through
 * torch.distributed).  Valid until lmcma_hip_destroy.                          */
enum {
    LMCMA_BUF_ROWS_LOCAL = 0,    /* float  [lambda_local][ld]                    */
    LMCMA_BUF_COST_LOCAL = 1,    /* float  [lambda_local], this context's block of COST_ALL */
    LMCMA_BUF_COST_ALL = 2,      /* float  [lambda_total]                        */
    LMCMA_BUF_PARTIAL_LOCAL = 3, /* double [N + 2]: the context's elite sum, then its
                                    success-rule count and the population's lowest cost */
    LMCMA_BUF_PARTIAL_ALL = 4,   /* double [max_partials][N + 2], one such row per context */
    LMCMA_BUF_XMEAN = 5,         /* double [N]                                   */
    LMCMA_BUF_SORTED_COST = 6,   /* float  [lambda_total]                        */
    LMCMA_BUF_PERM = 7           /* int32  [lambda_total]                        */
};
int lmcma_hip_buffer(lmcma_hip_ctx *ctx, int which, void **device_ptr, size_t *bytes);
int lmcma_hip_row_stride(lmcma_hip_ctx *ctx); /* ld: floats between rows */
/* Waits for the context's stream.  Returns LMCMA_HIP_ERR_RUNTIME when an update that ran at
 * the head of a sampling launch gave up its bounded wait for the launch's other workgroups
 * (the kernels report it through a word in pinned host memory, so no copy is needed to see
 * it): the state is that of the last good generation and stays so; lmcma_hip_run refuses
 * further generations, lmcma_hip_opt_get reports the same, lmcma_hip_opt_reset / opt_init
 * start anew.                                                                           */
int lmcma_hip_synchronize(lmcma_hip_ctx *ctx);

/* Device time of a stretch of work on the context's stream: mark(0) before it is queued,
 * mark(1) after; stream_elapsed waits for mark 1 and returns the milliseconds between.   */
int lmcma_hip_stream_mark(lmcma_hip_ctx *ctx, int which);
int lmcma_hip_stream_elapsed(lmcma_hip_ctx *ctx, double *ms);

/* kernel timing with HIP events on the context's stream: records around every
 * kernel class while enabled; totals in milliseconds and launch counts.       */
enum { LMCMA_K_SAMPLE = 0, LMCMA_K_COST = 1, LMCMA_K_SELECT = 2, LMCMA_K_RECOMBINE = 3,
       LMCMA_K_UPDATE = 4, LMCMA_K_GENERATE = 5 /* sample + cost fused */, LMCMA_K_COUNT = 6 };
int lmcma_hip_timing_enable(lmcma_hip_ctx *ctx, int on);
int lmcma_hip_timing_get(lmcma_hip_ctx *ctx, double ms[LMCMA_K_COUNT],
                         int64_t launches[LMCMA_K_COUNT]);
/* average reading of `pairs` (1..4096) empty event brackets on the context's stream:
 * the share of a bracketed launch's reading that is not the kernel itself.      */
int lmcma_hip_timing_event_overhead(lmcma_hip_ctx *ctx, int pairs, double *ms_per_pair);

#ifdef __cplusplus
}
#endif
#endif /* LMCMA_HIP_H */
