// C ABI of liblmcma_hip: context lifetime, host<->device plumbing and the
// per-generation launch sequences.  Entry points are documented in lmcma_hip.h.
#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include <rccl/rccl.h>   // types and enumerators only: the functions are bound with dlsym (rccl_api)

#include "device_funcs.hpp"

using namespace lmcma;

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string &msg)
{
    g_last_error = msg;
    return code;
}

}  // namespace

namespace lmcma {
// for entry points defined beside their kernels
int set_last_error(int code, const char *msg) { return fail(code, msg); }
hipError_t rank_scratch_current_device(uint64_t **out);  // k_select.hip

hipError_t ensure_dynamic_lds(const void *kernel, size_t lds)
{
    if (lds <= 48 * 1024) return hipSuccess;  // within every kernel's default limit
    static std::mutex mu;
    static std::map<std::pair<int, const void *>, size_t> configured;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    size_t &have = configured[{dev, kernel}];
    if (lds > have) {
        e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        have = lds;
    }
    return hipSuccess;
}
}  // namespace lmcma

namespace {

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess)                                                                \
            return fail(LMCMA_HIP_ERR_RUNTIME, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

__global__ void k_set_sample_scalars(SampleScalars *sc, float sigma, float mk, int k, int set_state,
                                     uint32_t generation, int set_generation)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        if (set_state) {
            sc->sigma = sigma;
            sc->mk = mk;
            sc->k = k;
        }
        if (set_generation) sc->generation = generation;
    }
}

// back to the state lmcma_hip_opt_init left: mean, step size, empty direction store
__global__ void k_opt_reset(OptScalars *opt, SampleScalars *sc, const double *xmean0,
                            double *xmean, float *xmean_f, double *pc, int N, int ld,
                            unsigned long long *psr_accum, size_t qstride, unsigned *fused_counter, int *host_flag)
{
    // (the counter of the fused update: a launch that gave up leaves it anywhere; not per query; and
    // the word in host memory that told of it, cleared here in stream order)
    if (fused_counter && blockIdx.y == 0 && threadIdx.x == 0) *fused_counter = 0u;
    if (host_flag && blockIdx.y == 0 && threadIdx.x == 0) __hip_atomic_store(host_flag, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const size_t off = (size_t)blockIdx.y * qstride;  // query blockIdx.y of a batch
    shift_ptr(opt, off); shift_ptr(sc, off); shift_ptr(xmean0, off); shift_ptr(xmean, off);
    shift_ptr(xmean_f, off); shift_ptr(pc, off); shift_ptr(psr_accum, off);
    const double sigma0 = opt->sigma0;
    for (int n = threadIdx.x; n < ld; n += blockDim.x) {
        const double x = n < N ? xmean0[n] : 0.0;
        if (n < N) {
            xmean[n] = x;
            pc[n] = 0.0;
        }
        xmean_f[n] = (float)x;
    }
    if (threadIdx.x == 0) {
        opt->sigma = sigma0;
        opt->s = 0.0;
        opt->best_f = 1.7976931348623157e308;
        opt->counteval = 0;
        opt->itr = 0;
        opt->iterator_sz = 0;
        opt->done = 0;
        *psr_accum = 0ull;  // (an update that the stop rule froze has not cleared it)
        sc->sigma = (float)sigma0;
        sc->mk = 1.0f;
        sc->k = 0;
        sc->generation = 0u;
    }
}

struct EventPair {
    hipEvent_t a, b;
    int cls;
};

constexpr int kMaxPartials = 64;
// generations per recorded graph: long ones for the bulk of a run, short ones for its tail.  A
// graph starts with a sampling launch of its own and ends with an update of its own, so longer
// graphs have fewer of both: cfg2 over episodes of 100 generations, 24.2 us per generation with
// 8/2, 23.6 with 32/8/2.  The two sorted-cost buffers alternate per generation, so every length
// is recorded twice, once per buffer the first generation may find current; with a graph of one
// generation among them a call of any length is replay only.
constexpr int kGraphKinds = 4;
constexpr int kCallGraphs = 4, kCallGraphMaxGens = 256;
constexpr int kGensPerGraphDefault[kGraphKinds] = {32, 8, 2, 1};  // descending, 0 = unused

}  // namespace

struct lmcma_hip_ctx {
    lmcma_hip_config cfg{};
    int N = 0, ld = 0, D = 0, W = 0, lambda = 0, nloc = 0, first = 0, m = 0, mu = 0;
    // Q independent optimisers in this context (see Batch): every arena buffer exists Q times, qstride
    // bytes apart; the pointers below are query 0's.  qsel: the query the reader functions address.
    int Q = 1, qsel = 0, pending_rows = 0;
    size_t qstride = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;

    float *rows = nullptr, *zbuf = nullptr, *cost_all = nullptr, *terms = nullptr;
    float *sorted_cost = nullptr;            // = sorted_buf[parity], this generation's ascending costs
    float *sorted_buf[2] = {nullptr, nullptr};
    int parity = 0;
    bool warmed = false;
    unsigned long long *psr_accum = nullptr;
    // A shard of a population that is exchanged as sorted lists (rank_merge_possible): the local
    // costs in candidate order, the local ranking's permutation, and the keys of this / the last
    // generation's own sorted list
    bool merge_protocol = false;
    // A whole population in one context above one counting tile: ranked in tiles of tile_len keys
    // (each among itself, ascending lists in `lists`), the tiles then merged like the lists of
    // shards; local_perm / key_buf cover all tiles.  0: not used.
    int tile_len = 0;
    float *lists = nullptr;
    float *cost_raw = nullptr;
    int32_t *local_perm = nullptr;
    uint32_t *key_buf[2] = {nullptr, nullptr};
    float *best = nullptr;                      // a shard: lowest cost of the population
    bool ranked_local_only = false;             // the last ranking covered the local candidates only
    bool pre_done = false;   // the first part of this generation's update ran with a ranking launch
    // the candidates the next sampling launch covers, relative to the local ones (gen_count < 0: all of
    // them): a context that holds a whole population of which it samples one rank's range
    int gen_first = 0, gen_count = -1;
    // the word on which workgroup 0 of a sampling launch that carries the previous generation's
    // update waits for the other workgroups (GenParams::fused_counter); zero between launches
    unsigned *fused_counter = nullptr;
    // pinned host word the update kernels can write (UpdateArgs::host_flag) and its device address;
    // null when the mapping could not be made (lmcma_hip_opt_get still reports the failure)
    int *host_flag = nullptr, *host_flag_dev = nullptr;
    int psr_parts_n = 0;   // per-workgroup success-rule counts the last ranking left at the head of `keys`
    int32_t *perm = nullptr;
    uint64_t *keys = nullptr;
    float *xmean_f = nullptr, *V_f = nullptr, *B_f = nullptr, *lo_f = nullptr, *hi_f = nullptr;
    SampleScalars *sc = nullptr;
    double *weights = nullptr, *block_partials = nullptr, *reduced_rows = nullptr, *partial_local = nullptr,
           *partial_all = nullptr, *xmean = nullptr, *pc = nullptr, *pc_arr = nullptr,
           *v_arr = nullptr, *nj = nullptr, *lj = nullptr;
    int32_t *t = nullptr, *vec = nullptr, *iterator = nullptr;
    OptScalars *opt = nullptr;
    float *scratch_rows = nullptr, *scratch_cost = nullptr, *scratch_terms = nullptr;
    double *xmean0 = nullptr;   // initial mean kept for lmcma_hip_opt_reset
    void *arena = nullptr;      // the one allocation behind every buffer above
    double sigma0 = 1.0;

    void *grid = nullptr;
    size_t grid_bytes = 0;
    int grid_type = -1, nx = 0, ny = 0, nz = 0, grid_words = 0;
    float c_obs = 1.f, c_start = 0.f, c_goal = 0.f;
    bool grid_lds = false;
    bool fused = true;   // sample + score in one kernel on the device-resident path
    // factor of a structured sampling covariance (device Philox normals only)
    float *cov_L = nullptr;
    CovFactor cov{};
    bool have_cov = false;

    bool have_state = false, have_weights = false, opt_ready = false;
    bool sampled = false, costs_ready = false, selected = false;

    // a population sharded over processes: the RCCL communicator of its shards
    ncclComm_t comm = nullptr;
    bool own_comm = false;
    int world = 0, rank = 0;

    // direct exchange between the shards (lmcma_hip_peer_*): the own mailbox (fine-grained device
    // memory the peers write into) and the peers' mailboxes as mapped here
    void *mailbox = nullptr;
    size_t mailbox_bytes = 0;
    bool mailbox_fine = false;
    bool peer_mapped = false;              // the peers' mailboxes are mapped (lmcma_hip_peer_connect)
    bool peer_on = false;                  // ... and lmcma_hip_run_sharded exchanges through them
    bool peer_image = false;               // the peers are images of this shard (its own handle in their places)
    void *peer_base[kMaxPeerRanks] = {};
    bool peer_ipc[kMaxPeerRanks] = {};     // mapped with hipIpcOpenMemHandle (to be closed)
    uint32_t peer_epoch = 0;               // exchanges of each kind made so far (every shard counts alike)
    uint32_t sent_costs = 0, sent_rows = 0; // counter increments per source so far (what the next wait wants)
    bool peer_sending = false;             // inside a generation of the direct form: the producers send
    int peer_arrivals = 0;                 // increments per destination the last producer made
    bool peer_rows_received = false;       // ... and that producer also waited for and copied the peers' rows
    PeerWait wait_costs{}, wait_rows{};
    hipEvent_t group_ev[2] = {nullptr, nullptr};   // lmcma_hip_run_sharded_group: this shard's two hand-over points
    bool group_copied = false;                     // ... the rows form has recorded group_ev[1] (its copies of the others' ranges)

    hipGraphExec_t graph_exec[2][kGraphKinds] = {};  // [buffer parity at the start][length]
    hipGraph_t graph[2][kGraphKinds] = {};
    int graph_gens[kGraphKinds] = {};
    // graphs of whole calls: a call length that is none of the above gets a graph of its own,
    // recorded behind the first call of that length, which later calls of that length replay
    struct CallGraph {
        int gens = 0, parity = 0;
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        uint64_t used = 0;   // call_tick of its last launch: the slot used longest ago makes room
    } call_graph[kCallGraphs];
    uint64_t call_tick = 0;
    // two sharded generations, collectives included (lmcma_hip_run_sharded, opt-in)
    hipGraphExec_t shard_graph_exec = nullptr;
    hipGraph_t shard_graph = nullptr;
    int shard_graph_parity = 0;
    hipEvent_t mark[2] = {nullptr, nullptr};

    bool timing = false;
    std::vector<EventPair> ev_pool;
    size_t ev_used = 0;
    double t_ms[LMCMA_K_COUNT] = {};
    int64_t t_n[LMCMA_K_COUNT] = {};
};

namespace {

template <typename T>
hipError_t dalloc(T **p, size_t count)
{
    hipError_t e = hipMalloc(reinterpret_cast<void **>(p), count * sizeof(T) + 16);
    if (e == hipSuccess) e = hipMemset(*p, 0, count * sizeof(T) + 16);
    // the fill runs on the null stream, which the context's own (non-blocking) stream does not
    // wait for: finish it before anything is queued on that stream
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    return e;
}

// All buffers of a context that exist for its whole life come from one allocation: a
// kernel that touches a dozen small buffers then needs a few address translations
// instead of one per buffer (they share pages), which shortens every kernel's start.
struct Arena {
    std::vector<std::pair<void **, size_t>> req;
    template <typename T>
    void add(T **p, size_t count)
    {
        req.push_back({reinterpret_cast<void **>(p), (count * sizeof(T) + 16 + 255) & ~(size_t)255});
    }
    // copies: the whole set of buffers that many times, *stride bytes apart (the pointers are those of copy 0)
    hipError_t commit(void **base, int copies, size_t *stride)
    {
        size_t total = 0;
        for (auto &r : req) total += r.second;
        *stride = total;
        total *= (size_t)copies;
        hipError_t e = hipMalloc(base, total);
        if (e == hipSuccess) e = hipMemset(*base, 0, total);
        if (e == hipSuccess) e = hipStreamSynchronize(nullptr);  // see dalloc
        if (e != hipSuccess) return e;
        char *p = static_cast<char *>(*base);
        for (auto &r : req) {
            *r.first = p;
            p += r.second;
        }
        return hipSuccess;
    }
};

void drop_graph(lmcma_hip_ctx *c)
{
    for (int q = 0; q < 2 * kGraphKinds; ++q) {
        hipGraphExec_t &ge = c->graph_exec[q / kGraphKinds][q % kGraphKinds];
        hipGraph_t &gr = c->graph[q / kGraphKinds][q % kGraphKinds];
        if (ge) { (void)hipGraphExecDestroy(ge); ge = nullptr; }
        if (gr) { (void)hipGraphDestroy(gr); gr = nullptr; }
    }
    for (auto &cg : c->call_graph) {
        if (cg.exec) (void)hipGraphExecDestroy(cg.exec);
        if (cg.graph) (void)hipGraphDestroy(cg.graph);
        cg = lmcma_hip_ctx::CallGraph{};
    }
    if (c->shard_graph_exec) { (void)hipGraphExecDestroy(c->shard_graph_exec); c->shard_graph_exec = nullptr; }
    if (c->shard_graph) { (void)hipGraphDestroy(c->shard_graph); c->shard_graph = nullptr; }
}

int fold_events(lmcma_hip_ctx *c)
{
    if (c->ev_used == 0) return LMCMA_HIP_OK;
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (size_t i = 0; i < c->ev_used; ++i) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, c->ev_pool[i].a, c->ev_pool[i].b));
        c->t_ms[c->ev_pool[i].cls] += ms;
        c->t_n[c->ev_pool[i].cls] += 1;
    }
    c->ev_used = 0;
    return LMCMA_HIP_OK;
}

// brackets one kernel class with a pair of events while timing is on
struct Timed {
    lmcma_hip_ctx *c;
    EventPair *ev = nullptr;
    Timed(lmcma_hip_ctx *ctx, int cls) : c(ctx)
    {
        if (!c->timing) return;
        if (c->ev_used == c->ev_pool.size()) {
            if (c->ev_pool.size() >= 4096) {
                if (fold_events(c) != LMCMA_HIP_OK) return;
            } else {
                EventPair p;
                if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) return;
                c->ev_pool.push_back(p);
            }
        }
        ev = &c->ev_pool[c->ev_used++];
        ev->cls = cls;
        (void)hipEventRecord(ev->a, c->stream);
    }
    ~Timed()
    {
        if (ev) (void)hipEventRecord(ev->b, c->stream);
    }
};

// query q's copy of a context buffer, and the copy the reader functions address
template <typename T>
T *query_ptr(const lmcma_hip_ctx *c, T *p, int q)
{
    return reinterpret_cast<T *>(reinterpret_cast<char *>(p) + (size_t)q * c->qstride);
}
template <typename T>
T *sel(const lmcma_hip_ctx *c, T *p) { return query_ptr(c, p, c->qsel); }

Batch batch_of(const lmcma_hip_ctx *c) { return Batch{c->Q, c->Q > 1 ? c->qstride : 0}; }

// where the kernels leave the costs of the local candidates, in candidate order
float *local_costs(const lmcma_hip_ctx *c) { return c->merge_protocol ? c->cost_raw : c->cost_all + c->first; }

CostParams cost_params(const lmcma_hip_ctx *c, const float *rows, int n_rows, float *cost, float *terms)
{
    CostParams p{};
    p.rows = rows;
    p.ld = c->ld;
    p.n_rows = n_rows;
    p.cost = cost;
    p.terms = terms;
    p.grid = c->grid;
    p.grid_type = c->grid_type;
    p.nx = c->nx; p.ny = c->ny; p.nz = c->nz;
    p.c_obs = c->c_obs;
    p.W = c->W;
    for (int d = 0; d < 3; ++d) {
        p.start[d] = (float)c->cfg.start[d];
        p.goal[d] = (float)c->cfg.goal[d];
    }
    p.c_start = c->c_start;
    p.c_goal = c->c_goal;
    p.w_len = (float)c->cfg.w_len;
    p.w_grid = (float)c->cfg.w_grid;
    p.w_smooth = (float)c->cfg.w_smooth;
    p.pad_zero = c->cfg.pad_mode == LMCMA_PAD_ZERO;
    p.grid_words = c->grid_words;
    return p;
}

int do_sample(lmcma_hip_ctx *c, const float *z)
{
    Timed tm(c, LMCMA_K_SAMPLE);
    // a factor set on the context applies to the normals of either source
    HIP_TRY(launch_sample(c->stream, c->rows, z, c->xmean_f, c->V_f, c->B_f, c->lo_f, c->hi_f, c->sc,
                          c->ld, c->nloc, (uint32_t)c->first, c->cfg.seed, c->have_cov ? &c->cov : nullptr));
    c->sampled = true;
    return LMCMA_HIP_OK;
}

int do_evaluate(lmcma_hip_ctx *c)
{
    Timed tm(c, LMCMA_K_COST);
    CostParams p = cost_params(c, c->rows, c->nloc, local_costs(c), c->terms);
    HIP_TRY(launch_cost(c->stream, p, c->D, c->grid_lds));
    return LMCMA_HIP_OK;
}

UpdateArgs update_args(const lmcma_hip_ctx *c, const double *partials, int n_partials, bool rows_with_scalars);
PeerSend peer_send_table_if_sending(const lmcma_hip_ctx *c, int which);  // n = 0 outside the direct exchange
// the receiving side of the rows' exchange inside the kernel that sends the row (x.send set, `arrivals`
// counter increments per source): fills x.gather*; false: that kernel cannot, a launch of its own follows
bool peer_gather_in_sender(lmcma_hip_ctx *c, FinalExtras &x, int arrivals);

// sample + score in one launch (device Philox stream only).
// post: the previous generation's update, not yet applied, is to run at the head of this launch
// (every workgroup computes it, the first stores it); *post_done = false: it cannot, nothing
// has been launched, and the caller applies the update first.
int do_generate(lmcma_hip_ctx *c, const UpdateArgs *post = nullptr, bool *post_done = nullptr)
{
    Timed tm(c, LMCMA_K_GENERATE);
    GenParams g{};
    const int gf = c->gen_count >= 0 ? c->gen_first : 0, gn = c->gen_count >= 0 ? c->gen_count : c->nloc;
    g.rows = c->rows + (size_t)gf * c->ld;
    g.xmean = c->xmean_f; g.V = c->V_f; g.B = c->B_f; g.lo = c->lo_f; g.hi = c->hi_f;
    g.sc = c->sc;
    g.ld = c->ld; g.n_local = gn; g.m = c->m;
    g.first_candidate = (uint32_t)(c->first + gf);
    g.key0 = (uint32_t)c->cfg.seed;
    g.key1 = (uint32_t)(c->cfg.seed >> 32);
    g.cost = cost_params(c, g.rows, gn, local_costs(c) + gf, c->terms ? c->terms + 3 * (size_t)gf : nullptr);
    if (c->have_cov) g.cov = c->cov;
    g.batch = batch_of(c);
    g.fused_counter = c->fused_counter;
    if (post != nullptr) {
        *post_done = false;
        if (!c->pre_done || !generate_fuses(g, c->grid_lds, post)) return LMCMA_HIP_OK;
        HIP_TRY(launch_generate(c->stream, g, c->D, c->grid_lds, nullptr, nullptr, post, post_done));
        if (!*post_done) return fail(LMCMA_HIP_ERR_RUNTIME, "the sampling launch did not take the update it was to carry");
        c->pre_done = false;  // (consumed, as launch_update_post would have)
        c->sampled = true;
        return LMCMA_HIP_OK;
    }
    // the first part of this generation's update rides along when it is too long for the ranking launch
    const UpdateArgs pre = update_args(c, nullptr, 0, false);
    bool pre_done = false;
    HIP_TRY(launch_generate(c->stream, g, c->D, c->grid_lds, (c->opt_ready && !c->pre_done) ? &pre : nullptr, &pre_done));
    if (pre_done) c->pre_done = true;
    c->sampled = true;
    return LMCMA_HIP_OK;
}

// rows_with_scalars: the partial rows are partial_local / partial_all (N + 2 entries each, the
// last two the contexts' success-rule counts and the lowest cost), else block_partials (N)
UpdateArgs update_args(const lmcma_hip_ctx *c, const double *partials, int n_partials, bool rows_with_scalars)
{
    UpdateArgs a{};
    a.partial_stride = rows_with_scalars ? c->N + 2 : c->N;
    a.scalars_in_rows = rows_with_scalars ? 1 : 0;
    a.opt = c->opt;
    a.sc = c->sc;
    a.N = c->N; a.ld = c->ld; a.lambda = c->lambda; a.m = c->m;
    a.partials = partials;
    a.n_partials = n_partials;
    a.sorted_cost = c->sorted_cost;
    a.psr_accum = c->psr_accum;
    a.psr_parts = reinterpret_cast<const uint32_t *>(c->keys);
    a.n_psr_parts = c->psr_parts_n;
    a.xmean = c->xmean; a.pc = c->pc;
    a.pc_arr = c->pc_arr; a.v_arr = c->v_arr;
    a.nj = c->nj; a.lj = c->lj;
    a.t = c->t; a.vec = c->vec; a.iterator = c->iterator;
    a.xmean_f = c->xmean_f; a.V_f = c->V_f; a.B_f = c->B_f;
    a.batch = batch_of(c);
    a.host_flag = c->host_flag_dev;
    return a;
}

// with_recombine: the ranking launch also leaves the elite partial sums of the local rows
// in block_partials (*rows_out rows); only populations ranked by counting
int do_select(lmcma_hip_ctx *c, bool with_recombine, int *rows_out, bool with_pre = false)
{
    Timed tm(c, LMCMA_K_SELECT);
    // the two sorted-cost buffers alternate: this generation's list is written beside last one's
    c->parity ^= 1;
    c->sorted_cost = c->sorted_buf[c->parity];
    if (c->tile_len > 0 && with_recombine && c->have_weights && c->opt_ready) {
        // tiles ranked among themselves (the first part of the update may ride along), then merged:
        // global ranks, the success-rule count and the elite rows, as for the lists of shards
        SelectArgs t{};
        t.cost_all = c->cost_all;
        t.lambda = c->lambda;
        t.keys_ws = c->keys;
        t.sorted_cost = c->lists;
        t.perm = c->local_perm;
        t.tile_len = c->tile_len;
        t.batch = batch_of(c);
        const UpdateArgs pre = update_args(c, nullptr, 0, false);
        t.pre = (with_pre && !c->pre_done) ? &pre : nullptr;
        SelectResult tres{};
        HIP_TRY(launch_select(c->stream, t, &tres));
        if (tres.pre_done) c->pre_done = true;
        MergeArgs a{};
        a.lists = c->lists;
        a.owners = c->lambda / c->tile_len;
        a.lambda = c->lambda; a.n_local = c->tile_len; a.first_candidate = 0;
        a.local_perm = c->local_perm;
        a.sorted_cost = c->sorted_cost;
        a.perm = c->perm;
        a.opt = c->opt;
        a.prev_own = c->key_buf[c->parity ^ 1];
        a.own_out = c->key_buf[c->parity];
        a.psr_parts = reinterpret_cast<uint32_t *>(c->keys);
        a.best_out = c->best;
        a.rows = c->rows; a.ld = c->ld; a.N = c->N;
        a.weights = c->weights; a.mu = c->mu;
        a.block_partials = c->block_partials;
        int rows = 0;
        HIP_TRY(launch_rank_merge(c->stream, a, &rows));
        c->psr_parts_n = rows;
        c->ranked_local_only = false;
        if (rows_out) *rows_out = rows;
        c->selected = true;
        return LMCMA_HIP_OK;
    }
    RecombineArgs rc{c->rows, c->ld, c->N, c->first, c->nloc, c->weights, c->mu, c->block_partials};
    const bool recombine = with_recombine && c->have_weights && select_by_counting(c->lambda);
    SelectArgs a{};
    a.cost_all = c->cost_all;
    a.lambda = c->lambda;
    a.keys_ws = c->keys;
    a.sorted_cost = c->sorted_cost;
    a.perm = c->perm;
    a.opt = c->opt_ready ? c->opt : nullptr;
    a.prev_sorted = c->sorted_buf[c->parity ^ 1];
    a.psr_accum = c->psr_accum;
    a.rc = recombine ? &rc : nullptr;
    a.batch = batch_of(c);
    const UpdateArgs pre = update_args(c, nullptr, 0, false);
    a.pre = (with_pre && c->opt_ready && !c->pre_done) ? &pre : nullptr;
    SelectResult res{};
    HIP_TRY(launch_select(c->stream, a, &res));
    if (res.pre_done) c->pre_done = true;
    c->psr_parts_n = res.psr_parts;
    c->ranked_local_only = false;
    if (rows_out) *rows_out = res.partial_rows;
    c->selected = true;
    return LMCMA_HIP_OK;
}

// rank-ordered elite sums of the local rows -> rows of block_partials
int do_recombine(lmcma_hip_ctx *c)
{
    Timed tm(c, LMCMA_K_RECOMBINE);
    HIP_TRY(launch_recombine(c->stream, c->rows, c->ld, c->N, c->first, c->nloc, c->perm, c->weights,
                             c->mu, c->block_partials, batch_of(c)));
    return LMCMA_HIP_OK;
}

// Many rows of block_partials are first summed to a few by many workgroups; *src / *rows are
// what the single workgroup after that reads.
int reduce_if_many(lmcma_hip_ctx *c, int *rows, const double **src)
{
    *src = c->block_partials;
    if (*rows < kReduceRowsFrom) return LMCMA_HIP_OK;
    Timed tm(c, LMCMA_K_RECOMBINE);
    HIP_TRY(launch_reduce_rows(c->stream, c->block_partials, *rows, c->N, c->reduced_rows, batch_of(c)));
    *src = c->reduced_rows;
    *rows = reduce_rows_out(*rows, c->N);
    return LMCMA_HIP_OK;
}

// block_partials -> partial_local (needed when the partial sum leaves the context; the
// update can add the block partials itself).  with_scalars: also the context's success-rule
// count and the lowest cost, entries N and N + 1 of the row.
int do_recombine_final(lmcma_hip_ctx *c, int rows, bool with_scalars)
{
    const double *src = nullptr;
    int rc = reduce_if_many(c, &rows, &src);
    if (rc != LMCMA_HIP_OK) return rc;
    Timed tm(c, LMCMA_K_RECOMBINE);
    FinalExtras x{};
    x.psr_parts = reinterpret_cast<const uint32_t *>(c->keys);
    x.n_parts = c->psr_parts_n;
    x.psr_accum = c->ranked_local_only ? nullptr : c->psr_accum;
    // a context that ranked the whole population holds the whole count: only the first one reports it
    x.count_psr = (c->ranked_local_only || c->first == 0) ? 1 : 0;
    x.best = c->ranked_local_only ? c->best : c->sorted_cost;
    // A context whose row goes into an exchange may add its rows with any (fixed) workgroup size;
    // one that feeds its own update must use the update's, so that reducing first and updating
    // from the one row gives the same bits as updating from the block rows.
    const int threads = c->merge_protocol ? 0 : update_threads(c->N, std::min(c->m, LMCMA_HIP_MAX_VECTORS));
    if (with_scalars) x.send = peer_send_table_if_sending(c, 1);
    c->peer_rows_received = false;
    if (x.send.n > 0 && threads == 0 && recombine_final_is_wide(rows, c->N))
        c->peer_rows_received = peer_gather_in_sender(c, x, (c->N + 15) / 16);
    int arrivals = 0;
    HIP_TRY(launch_recombine_final(c->stream, src, rows, c->N, c->partial_local, threads,
                                   with_scalars ? &x : nullptr, &arrivals));
    if (x.send.n > 0) c->peer_arrivals = arrivals;
    return LMCMA_HIP_OK;
}

int do_update(lmcma_hip_ctx *c, const double *partials, int n_partials, bool rows_with_scalars)
{
    Timed tm(c, LMCMA_K_UPDATE);
    const UpdateArgs a = update_args(c, partials, n_partials, rows_with_scalars);
    if (update_is_split(a.N, a.m)) {
        // The part that does not depend on this generation's costs has run with the sampling or the
        // ranking launch.  If it has not: a short one (up to 16 vectors) runs here, a long one does
        // not pay on its own, and the single kernel below does the whole update.
        if (!c->pre_done && update_pre_rides_with_ranking(a.m)) {
            HIP_TRY(launch_update_pre(c->stream, a));
            c->pre_done = true;
        }
        if (c->pre_done) {
            c->pre_done = false;
            HIP_TRY(launch_update_post(c->stream, a));
            return LMCMA_HIP_OK;
        }
    }
    HIP_TRY(launch_update(c->stream, a));
    return LMCMA_HIP_OK;
}

// n generations in a row.  Between two of them the update of the first runs at the head of the
// sampling launch of the second where that form exists (do_generate with post); the last
// generation's update is a launch of its own, so the state is complete when the call returns.
int generations(lmcma_hip_ctx *c, int n)
{
    int rc, rows = 0;
    const double *src = nullptr;
    for (int i = 0; i < n; ++i) {
        bool carried = false;
        if (i > 0) {
            const UpdateArgs post = update_args(c, src, rows, false);
            if (c->fused && (rc = do_generate(c, &post, &carried)) != LMCMA_HIP_OK) return rc;
            if (!carried && (rc = do_update(c, src, rows, false)) != LMCMA_HIP_OK) return rc;
        }
        if (!carried) {
            if (c->fused) {
                if ((rc = do_generate(c)) != LMCMA_HIP_OK) return rc;
            } else {
                if ((rc = do_sample(c, nullptr)) != LMCMA_HIP_OK) return rc;
                if ((rc = do_evaluate(c)) != LMCMA_HIP_OK) return rc;
            }
        }
        if ((rc = do_select(c, true, &rows, true)) != LMCMA_HIP_OK) return rc;
        if (rows == 0) {
            if ((rc = do_recombine(c)) != LMCMA_HIP_OK) return rc;
            rows = recombine_num_blocks(c->mu);
        }
        if ((rc = reduce_if_many(c, &rows, &src)) != LMCMA_HIP_OK) return rc;
    }
    return n > 0 ? do_update(c, src, rows, false) : LMCMA_HIP_OK;
}

// the word the update kernels set in host memory when their bounded wait gives up
int update_gave_up(const lmcma_hip_ctx *c)
{
    if (c->host_flag && *static_cast<volatile const int *>(c->host_flag) == 3)
        return fail(LMCMA_HIP_ERR_RUNTIME, "an update gave up waiting for the other workgroups of its launch; the state is stale "
                                           "(lmcma_hip_opt_reset or lmcma_hip_opt_init starts anew)");
    return LMCMA_HIP_OK;
}

int check_ctx(const lmcma_hip_ctx *c)
{
    if (!c) return fail(LMCMA_HIP_ERR_ARG, "null context");
    hipError_t e = hipSetDevice(c->cfg.device);
    if (e != hipSuccess) return fail(LMCMA_HIP_ERR_RUNTIME, std::string("hipSetDevice: ") + hipGetErrorString(e));
    return LMCMA_HIP_OK;
}

#define CTX_TRY(c)                                  \
    do {                                            \
        int _rc = check_ctx(c);                     \
        if (_rc != LMCMA_HIP_OK) return _rc;        \
    } while (0)

float host_grid_value(const lmcma_hip_ctx *c, const void *cells_host_bits_or_f32, const double *q)
{
    // the fixed endpoints' cell, with the same rounding as the kernel
    auto cl = [](double v, int n) {
        float f = fminf(fmaxf((float)v, 0.f), (float)(n - 1));
        return (int)nearbyintf(f);
    };
    const int ix = cl(q[0], c->nx);
    const int iy = c->D > 1 ? cl(q[1], c->ny) : 0;
    const int iz = c->D > 2 ? cl(q[2], c->nz) : 0;
    const long cell = ((long)iz * c->ny + iy) * c->nx + ix;
    if (c->grid_type == LMCMA_GRID_BITS) {
        const uint32_t *w = (const uint32_t *)cells_host_bits_or_f32;
        return ((w[cell >> 5] >> (cell & 31)) & 1u) ? c->c_obs : 0.f;
    }
    return ((const float *)cells_host_bits_or_f32)[cell];
}

}  // namespace

extern "C" {

const char *lmcma_hip_last_error(void) { return g_last_error.c_str(); }
int lmcma_hip_version(void) { return 1; }

int lmcma_hip_create(const lmcma_hip_config *cfg, lmcma_hip_ctx **out)
{
    if (!cfg || !out) return fail(LMCMA_HIP_ERR_ARG, "null argument");
    if (cfg->struct_size != (int32_t)sizeof(lmcma_hip_config))
        return fail(LMCMA_HIP_ERR_ARG, "lmcma_hip_config.struct_size does not match this library");
    if (cfg->n_params < 1 || cfg->n_params > LMCMA_HIP_MAX_PARAMS)
        return fail(LMCMA_HIP_ERR_UNSUPPORTED, "n_params must be in [1, 512]");
    if (cfg->dims < 0 || cfg->dims > 3) return fail(LMCMA_HIP_ERR_ARG, "dims must be 0..3");
    if (cfg->dims > 0 && (cfg->waypoints < 1 || cfg->dims * cfg->waypoints != cfg->n_params))
        return fail(LMCMA_HIP_ERR_ARG, "n_params must equal dims * waypoints");
    if (cfg->dims == 0 && cfg->cost_source != LMCMA_COST_EXTERNAL)
        return fail(LMCMA_HIP_ERR_ARG, "a plain vector problem (dims = 0) needs cost_source EXTERNAL");
    if (cfg->lambda_total < 2 || cfg->lambda_total > LMCMA_HIP_MAX_LAMBDA)
        return fail(LMCMA_HIP_ERR_UNSUPPORTED, "lambda_total must be in [2, 2^20]");
    if (cfg->lambda_local < 1 || cfg->first_candidate < 0 ||
        cfg->first_candidate + cfg->lambda_local > cfg->lambda_total)
        return fail(LMCMA_HIP_ERR_ARG, "local candidate range lies outside the population");
    if (cfg->m < 1 || cfg->m > LMCMA_HIP_MAX_VECTORS_STEPWISE)
        return fail(LMCMA_HIP_ERR_UNSUPPORTED, "m must be in [1, 4096]");
    if (cfg->mu < 1 || cfg->mu > cfg->lambda_total) return fail(LMCMA_HIP_ERR_ARG, "mu must be in [1, lambda]");
    if (cfg->z_source != LMCMA_Z_HOST && cfg->z_source != LMCMA_Z_PHILOX)
        return fail(LMCMA_HIP_ERR_ARG, "unknown z_source");
    const int queries = cfg->queries == 0 ? 1 : cfg->queries;
    if (queries < 1 || queries > LMCMA_HIP_MAX_QUERIES)
        return fail(LMCMA_HIP_ERR_UNSUPPORTED, "queries must be in [1, 256]");
    if (queries > 1 && (cfg->lambda_local != cfg->lambda_total || cfg->z_source != LMCMA_Z_PHILOX ||
                        cfg->cost_source != LMCMA_COST_TRAJECTORY || !select_by_counting(cfg->lambda_total) ||
                        cfg->m > LMCMA_HIP_MAX_VECTORS))
        return fail(LMCMA_HIP_ERR_UNSUPPORTED,
                    "several queries in one context: the whole population of each in the context (lambda_local = "
                    "lambda_total <= 16384), z_source PHILOX, cost_source TRAJECTORY, m <= 64");

    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (ndev < 1) return fail(LMCMA_HIP_ERR_RUNTIME, "no HIP device visible");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(LMCMA_HIP_ERR_ARG, "device ordinal out of range");
    HIP_TRY(hipSetDevice(cfg->device));

    {
        // set-up of lmcma_hip_rank_noalloc on this device, so that its first call need not be the one
        // that looks the scratch array up
        uint64_t *unused = nullptr;
        HIP_TRY(rank_scratch_current_device(&unused));
    }
    lmcma_hip_ctx *c = new lmcma_hip_ctx();
    c->cfg = *cfg;
    c->N = cfg->n_params;
    c->ld = (c->N + 3) & ~3;
    c->D = cfg->dims;
    c->W = cfg->waypoints;
    c->lambda = cfg->lambda_total;
    c->nloc = cfg->lambda_local;
    c->first = cfg->first_candidate;
    c->m = cfg->m;
    c->mu = cfg->mu;
    c->Q = queries;
    if (cfg->stream) {
        c->stream = (hipStream_t)cfg->stream;
    } else {
        hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { delete c; return fail(LMCMA_HIP_ERR_RUNTIME, "hipStreamCreate failed"); }
        c->own_stream = true;
    }
    {
        void *h = nullptr, *dptr = nullptr;
        if (hipHostMalloc(&h, 64, hipHostMallocMapped) == hipSuccess) {
            memset(h, 0, 64);
            if (hipHostGetDevicePointer(&dptr, h, 0) == hipSuccess) {
                c->host_flag = static_cast<int *>(h);
                c->host_flag_dev = static_cast<int *>(dptr);
            } else {
                (void)hipHostFree(h);
            }
        }
        (void)hipGetLastError();
    }
    const size_t ld = c->ld, N = c->N, nloc = c->nloc, lam = c->lambda, m = c->m;
    // the small, hot buffers first; the candidate rows (the only large one) last
    Arena ar;
    ar.add(&c->opt, 1);
    ar.add(&c->sc, 1);
    ar.add(&c->psr_accum, 1);
    ar.add(&c->fused_counter, 1);
    ar.add(&c->t, m);
    ar.add(&c->vec, m);
    ar.add(&c->iterator, m);
    ar.add(&c->nj, m);
    ar.add(&c->lj, m);
    ar.add(&c->xmean, N);
    ar.add(&c->xmean0, N);
    ar.add(&c->pc, N);
    ar.add(&c->xmean_f, ld);
    ar.add(&c->lo_f, ld);
    ar.add(&c->hi_f, ld);
    ar.add(&c->partial_local, N + 2);
    ar.add(&c->best, 1);
    ar.add(&c->V_f, m * ld);
    ar.add(&c->B_f, m * ld);
    ar.add(&c->pc_arr, m * N);
    ar.add(&c->v_arr, m * N);
    ar.add(&c->weights, (size_t)c->mu);
    ar.add(&c->cost_all, lam);
    ar.add(&c->sorted_buf[0], lam + 4);
    ar.add(&c->sorted_buf[1], lam + 4);
    ar.add(&c->perm, lam);
    ar.add(&c->keys, select_workspace_bytes(c->lambda) / sizeof(uint64_t) + 1);
    ar.add(&c->partial_all, (size_t)kMaxPartials * (N + 2));
    c->merge_protocol = rank_merge_possible(c->lambda, c->nloc, c->first);
    if (c->merge_protocol) {
        ar.add(&c->cost_raw, nloc);
        ar.add(&c->local_perm, nloc);
        ar.add(&c->key_buf[0], nloc);
        ar.add(&c->key_buf[1], nloc);
    }
    if (c->nloc == c->lambda && c->Q == 1) c->tile_len = rank_tile_len(c->lambda);
    if (c->tile_len > 0) {
        ar.add(&c->lists, lam + 4);
        ar.add(&c->local_perm, lam);
        ar.add(&c->key_buf[0], lam);
        ar.add(&c->key_buf[1], lam);
    }
    ar.add(&c->block_partials,
           (size_t)std::max(recombine_num_blocks(c->mu), select_max_partial_rows(c->first, c->nloc)) * N);
    ar.add(&c->reduced_rows, (size_t)kReducedRowsMax * N);
    ar.add(&c->terms, nloc * 3);
    ar.add(&c->rows, nloc * ld);
    hipError_t e = ar.commit(&c->arena, c->Q, &c->qstride);
    auto A = [&](hipError_t r) { if (e == hipSuccess) e = r; };
    c->sorted_cost = c->sorted_buf[0];
    if (e == hipSuccess) {
        // no bounds until set_bounds: -inf / +inf
        std::vector<float> lo(ld, -INFINITY), hi(ld, INFINITY);
        for (int q = 0; q < c->Q; ++q) {
            A(hipMemcpy(query_ptr(c, c->lo_f, q), lo.data(), ld * sizeof(float), hipMemcpyHostToDevice));
            A(hipMemcpy(query_ptr(c, c->hi_f, q), hi.data(), ld * sizeof(float), hipMemcpyHostToDevice));
        }
    }
    if (e != hipSuccess) {
        std::string msg = std::string("device allocation failed: ") + hipGetErrorString(e);
        lmcma_hip_destroy(c);
        return fail(LMCMA_HIP_ERR_RUNTIME, msg);
    }
    if (const char *f = getenv("LMCMA_HIP_FUSED")) c->fused = f[0] != '0' || c->Q > 1;
    *out = c;
    return LMCMA_HIP_OK;
}

int lmcma_hip_destroy(lmcma_hip_ctx *c)
{
    if (!c) return LMCMA_HIP_OK;
    (void)hipSetDevice(c->cfg.device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    (void)lmcma_hip_comm_destroy(c);
    (void)lmcma_hip_peer_disconnect(c);
    if (c->mailbox) (void)hipFree(c->mailbox);
    for (hipEvent_t ev : c->group_ev)
        if (ev) (void)hipEventDestroy(ev);
    drop_graph(c);
    for (auto &p : c->ev_pool) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    for (hipEvent_t ev : c->mark)
        if (ev) (void)hipEventDestroy(ev);
    void *bufs[] = {c->arena, c->zbuf, c->cov_L, c->scratch_rows, c->scratch_cost, c->scratch_terms, c->grid};
    for (void *b : bufs)
        if (b) (void)hipFree(b);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    if (c->host_flag) (void)hipHostFree(c->host_flag);
    delete c;
    return LMCMA_HIP_OK;
}

int lmcma_hip_set_grid(lmcma_hip_ctx *c, const void *cells, int nx, int ny, int nz, int cell_type,
                       double c_obs)
{
    CTX_TRY(c);
    if (c->D < 1) return fail(LMCMA_HIP_ERR_STATE, "this context has no trajectory cost (dims = 0)");
    if (!cells || nx < 1 || ny < 1 || nz < 1) return fail(LMCMA_HIP_ERR_ARG, "bad grid extent");
    if (c->D < 3 && nz != 1) return fail(LMCMA_HIP_ERR_ARG, "nz must be 1 for a 1-D or 2-D problem");
    if (c->D < 2 && ny != 1) return fail(LMCMA_HIP_ERR_ARG, "ny must be 1 for a 1-D problem");
    const long cells_n = (long)nx * ny * nz;
    if (cells_n > (1l << 30)) return fail(LMCMA_HIP_ERR_UNSUPPORTED, "grid larger than 2^30 cells");
    HIP_TRY(hipStreamSynchronize(c->stream));
    drop_graph(c);
    if (c->grid) { (void)hipFree(c->grid); c->grid = nullptr; }

    std::vector<uint32_t> packed;
    const void *src = cells;
    size_t bytes;
    if (cell_type == LMCMA_GRID_BITS || cell_type == LMCMA_GRID_U8) {
        // words padded to a multiple of 4 so the LDS copy can move 16 bytes per lane
        const size_t words = (((size_t)cells_n + 31) / 32 + 3) & ~(size_t)3;
        packed.assign(words, 0u);
        if (cell_type == LMCMA_GRID_U8) {
            const uint8_t *b = (const uint8_t *)cells;
            for (long i = 0; i < cells_n; ++i)
                if (b[i]) packed[i >> 5] |= 1u << (i & 31);
        } else {
            memcpy(packed.data(), cells, (((size_t)cells_n + 31) / 32) * sizeof(uint32_t));
        }
        src = packed.data();
        bytes = words * sizeof(uint32_t);
        c->grid_type = LMCMA_GRID_BITS;
        c->grid_words = (int)words;
    } else if (cell_type == LMCMA_GRID_F32) {
        bytes = (size_t)cells_n * sizeof(float);
        c->grid_type = LMCMA_GRID_F32;
        c->grid_words = 0;
    } else {
        return fail(LMCMA_HIP_ERR_ARG, "unknown cell_type");
    }
    c->nx = nx; c->ny = ny; c->nz = nz;
    c->c_obs = (float)c_obs;
    HIP_TRY(hipMalloc(&c->grid, bytes + 16));
    HIP_TRY(hipMemcpy(c->grid, src, bytes, hipMemcpyHostToDevice));
    c->grid_bytes = bytes;
    c->c_start = host_grid_value(c, src, c->cfg.start);
    c->c_goal = host_grid_value(c, src, c->cfg.goal);
    // A bit grid can be staged in LDS when it fits beside the row tile (160 KiB per
    // CU).  Each block then copies the whole grid (16 bytes per lane and load) before
    // its first lookup, so staging only pays once a block's own lookups outnumber those
    // copy loads several times over; measured at lambda = 4096 on a 512^2 grid (16
    // candidates per block), lookups straight from L2 are 4 us per generation faster.
    const size_t row_tile = (size_t)16 * c->ld * sizeof(float);
    const size_t cand_per_block = ((size_t)c->nloc + 255) / 256;
    c->grid_lds = c->grid_type == LMCMA_GRID_BITS && bytes + row_tile <= 160 * 1024 &&
                  cand_per_block * (size_t)(c->W + 2) >= 4 * (bytes / 16);
    if (const char *force = getenv("LMCMA_HIP_GRID_LDS")) {
        if (force[0] == '0') c->grid_lds = false;
        if (force[0] == '1' && c->grid_type == LMCMA_GRID_BITS && bytes + row_tile <= 160 * 1024)
            c->grid_lds = true;
    }
    return LMCMA_HIP_OK;
}

int lmcma_hip_set_grid_from_occupancy(lmcma_hip_ctx *c, const uint8_t *occ, int nx, int ny, int nz, double d_min,
                                      float *dist_out)
{
    CTX_TRY(c);
    if (c->D < 1) return fail(LMCMA_HIP_ERR_STATE, "this context has no trajectory cost (dims = 0)");
    if (!occ || nx < 1 || ny < 1 || nz < 1) return fail(LMCMA_HIP_ERR_ARG, "bad grid extent");
    if (nx > 1024 || ny > 1024 || nz > 1024) return fail(LMCMA_HIP_ERR_UNSUPPORTED, "extents up to 1024 cells per axis");
    if (c->D < 3 && nz != 1) return fail(LMCMA_HIP_ERR_ARG, "nz must be 1 for a 1-D or 2-D problem");
    if (c->D < 2 && ny != 1) return fail(LMCMA_HIP_ERR_ARG, "ny must be 1 for a 1-D problem");
    if (!(d_min > 0.0)) return fail(LMCMA_HIP_ERR_ARG, "d_min must be positive");
    const long cells_n = (long)nx * ny * nz;
    if (cells_n > (1l << 30)) return fail(LMCMA_HIP_ERR_UNSUPPORTED, "grid larger than 2^30 cells");
    HIP_TRY(hipStreamSynchronize(c->stream));
    drop_graph(c);
    if (c->grid) { (void)hipFree(c->grid); c->grid = nullptr; }
    // scratch of the transform: the occupancy bytes, squared distances, and the distances if asked for
    uint8_t *d_occ = nullptr;
    uint32_t *d_d2 = nullptr;
    float *d_dist = nullptr;
    auto release = [&]() {
        if (d_occ) (void)hipFree(d_occ);
        if (d_d2) (void)hipFree(d_d2);
        if (d_dist) (void)hipFree(d_dist);
    };
    hipError_t e = hipMalloc(&d_occ, (size_t)cells_n);
    if (e == hipSuccess) e = hipMalloc(&d_d2, 2 * (size_t)cells_n * sizeof(uint32_t));
    if (e == hipSuccess && dist_out) e = hipMalloc(&d_dist, (size_t)cells_n * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(&c->grid, (size_t)cells_n * sizeof(float) + 16);
    if (e == hipSuccess) e = hipMemcpy(d_occ, occ, (size_t)cells_n, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = launch_edt(c->stream, d_occ, nx, ny, nz, d_min, d_d2, d_dist, (float *)c->grid);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess && dist_out)
        e = hipMemcpy(dist_out, d_dist, (size_t)cells_n * sizeof(float), hipMemcpyDeviceToHost);
    c->nx = nx; c->ny = ny; c->nz = nz;
    c->grid_type = LMCMA_GRID_F32;
    c->grid_words = 0;
    c->grid_lds = false;
    c->c_obs = 1.f;
    c->grid_bytes = (size_t)cells_n * sizeof(float);
    // the fixed endpoints' cost: two cells of the finished map
    float ends[2] = {0.f, 0.f};
    if (e == hipSuccess) {
        auto cl = [](double v, int n) { return (int)nearbyintf(fminf(fmaxf((float)v, 0.f), (float)(n - 1))); };
        const double *pts[2] = {c->cfg.start, c->cfg.goal};
        for (int i = 0; i < 2 && e == hipSuccess; ++i) {
            const long cell = ((long)(c->D > 2 ? cl(pts[i][2], nz) : 0) * ny + (c->D > 1 ? cl(pts[i][1], ny) : 0)) * nx +
                              cl(pts[i][0], nx);
            e = hipMemcpy(&ends[i], (const float *)c->grid + cell, sizeof(float), hipMemcpyDeviceToHost);
        }
    }
    release();
    if (e != hipSuccess) {
        if (c->grid) { (void)hipFree(c->grid); c->grid = nullptr; }
        return fail(LMCMA_HIP_ERR_RUNTIME, std::string("set_grid_from_occupancy: ") + hipGetErrorString(e));
    }
    c->c_start = ends[0];
    c->c_goal = ends[1];
    return LMCMA_HIP_OK;
}

int lmcma_hip_set_bounds(lmcma_hip_ctx *c, const double *lo, const double *hi)
{
    CTX_TRY(c);
    std::vector<float> l(c->ld, -INFINITY), h(c->ld, INFINITY);
    // a bound that is not representable in float is moved inwards so that a
    // clamped coordinate never leaves the caller's fp64 interval
    for (int n = 0; n < c->N; ++n) {
        if (lo) {
            float f = (float)lo[n];
            if ((double)f < lo[n]) f = nextafterf(f, INFINITY);
            l[n] = f;
        }
        if (hi) {
            float f = (float)hi[n];
            if ((double)f > hi[n]) f = nextafterf(f, -INFINITY);
            h[n] = f;
        }
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int q = 0; q < c->Q; ++q) {  // the same box for every query
        HIP_TRY(hipMemcpy(query_ptr(c, c->lo_f, q), l.data(), c->ld * sizeof(float), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(query_ptr(c, c->hi_f, q), h.data(), c->ld * sizeof(float), hipMemcpyHostToDevice));
    }
    return LMCMA_HIP_OK;
}

int lmcma_hip_set_state(lmcma_hip_ctx *c, const double *xmean, double sigma, const double *V,
                        const double *PC, const double *nj, int k, double M)
{
    CTX_TRY(c);
    if (c->Q > 1) return fail(LMCMA_HIP_ERR_STATE, "set_state: the stepwise calls serve one query per context");
    if (!xmean) return fail(LMCMA_HIP_ERR_ARG, "xmean is null");
    if (k < 0 || k > c->m) return fail(LMCMA_HIP_ERR_ARG, "k must be in [0, m]");
    if (k > 0 && (!V || !PC || !nj)) return fail(LMCMA_HIP_ERR_ARG, "V, PC and nj are needed when k > 0");
    const int ld = c->ld, N = c->N;
    std::vector<float> xm(ld, 0.f), vf((size_t)std::max(k, 1) * ld, 0.f), bf((size_t)std::max(k, 1) * ld, 0.f);
    for (int n = 0; n < N; ++n) xm[n] = (float)xmean[n];
    for (int t = 0; t < k; ++t) {
        double coef = nj[t];
        for (int e = 0; e < k - 1 - t; ++e) coef *= M;
        for (int n = 0; n < N; ++n) {
            vf[(size_t)t * ld + n] = (float)V[(size_t)t * N + n];
            bf[(size_t)t * ld + n] = (float)(coef * PC[(size_t)t * N + n]);
        }
    }
    double mk = 1.0;
    for (int e = 0; e < k; ++e) mk *= M;
    // pageable sources: these copies complete for the host before returning
    HIP_TRY(hipMemcpyAsync(c->xmean_f, xm.data(), ld * sizeof(float), hipMemcpyHostToDevice, c->stream));
    if (k > 0) {
        HIP_TRY(hipMemcpyAsync(c->V_f, vf.data(), (size_t)k * ld * sizeof(float), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(c->B_f, bf.data(), (size_t)k * ld * sizeof(float), hipMemcpyHostToDevice, c->stream));
    }
    HIP_TRY(hipMemcpyAsync(c->xmean, xmean, N * sizeof(double), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_set_sample_scalars, dim3(1), dim3(64), 0, c->stream, c->sc, (float)sigma,
                       (float)mk, k, 1, 0u, 0);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->have_state = true;
    return LMCMA_HIP_OK;
}

int lmcma_hip_set_cov_factor(lmcma_hip_ctx *c, const double *L)
{
    CTX_TRY(c);
    HIP_TRY(hipStreamSynchronize(c->stream));
    drop_graph(c);  // a captured generation holds the kernels of the other sampling mode
    if (!L) {
        c->have_cov = false;
        return LMCMA_HIP_OK;
    }
    const int N = c->N, nt = (std::max(N, c->ld) + 15) / 16, np = 16 * nt;
    if (nt > 32) return fail(LMCMA_HIP_ERR_UNSUPPORTED, "set_cov_factor: N up to 512");
    std::vector<float> lf((size_t)np * np, 0.f);
    CovFactor f{};
    f.ld = np;
    f.ntiles = nt;
    for (int t = 0; t < nt; ++t) f.kblocks[t] = 1;
    for (int col = 0; col < N; ++col)
        for (int row = 0; row < N; ++row) {
            const double v = L[(size_t)col * N + row];  // column-major, as cholesky() leaves it (L/lmcma.cpp:846-853)
            if (!std::isfinite(v)) return fail(LMCMA_HIP_ERR_ARG, "set_cov_factor: the factor holds a value that is not finite");
            const float fv = (float)v;
            lf[(size_t)row * np + col] = fv;
            if (fv != 0.f && f.kblocks[row / 16] < col / 16 + 1) f.kblocks[row / 16] = (uint8_t)(col / 16 + 1);
        }
    if (!c->cov_L) HIP_TRY(dalloc(&c->cov_L, (size_t)np * np));
    HIP_TRY(hipMemcpy(c->cov_L, lf.data(), lf.size() * sizeof(float), hipMemcpyHostToDevice));
    f.L = c->cov_L;
    c->cov = f;
    c->have_cov = true;
    return LMCMA_HIP_OK;
}

int lmcma_hip_sample(lmcma_hip_ctx *c, uint32_t generation, const double *z_host)
{
    CTX_TRY(c);
    if (c->Q > 1) return fail(LMCMA_HIP_ERR_STATE, "sample: the stepwise calls serve one query per context");
    if (!c->have_state && !c->opt_ready) return fail(LMCMA_HIP_ERR_STATE, "set_state or opt_init must come first");
    if (c->cfg.z_source == LMCMA_Z_HOST && !z_host)
        return fail(LMCMA_HIP_ERR_ARG, "z_source HOST needs z_host");
    if (c->cfg.z_source == LMCMA_Z_PHILOX && z_host)
        return fail(LMCMA_HIP_ERR_ARG, "z_source PHILOX takes no z_host");
    const float *z = nullptr;
    if (z_host) {
        if (!c->zbuf) HIP_TRY(dalloc(&c->zbuf, (size_t)c->nloc * c->ld));
        std::vector<float> zf((size_t)c->nloc * c->ld, 0.f);
        for (int i = 0; i < c->nloc; ++i)
            for (int n = 0; n < c->N; ++n)
                zf[(size_t)i * c->ld + n] = (float)z_host[(size_t)i * c->N + n];
        HIP_TRY(hipMemcpyAsync(c->zbuf, zf.data(), zf.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        z = c->zbuf;
    }
    hipLaunchKernelGGL(k_set_sample_scalars, dim3(1), dim3(64), 0, c->stream, c->sc, 0.f, 0.f, 0, 0,
                       generation, 1);
    HIP_TRY(hipGetLastError());
    c->costs_ready = false;
    c->selected = false;
    return do_sample(c, z);
}

int lmcma_hip_evaluate(lmcma_hip_ctx *c)
{
    CTX_TRY(c);
    if (c->Q > 1) return fail(LMCMA_HIP_ERR_STATE, "evaluate: the stepwise calls serve one query per context");
    if (c->cfg.cost_source != LMCMA_COST_TRAJECTORY)
        return fail(LMCMA_HIP_ERR_STATE, "cost_source is EXTERNAL: use lmcma_hip_set_costs");
    if (!c->grid) return fail(LMCMA_HIP_ERR_STATE, "set_grid must come first");
    if (!c->sampled) return fail(LMCMA_HIP_ERR_STATE, "sample must come first");
    int rc = do_evaluate(c);
    if (rc == LMCMA_HIP_OK && c->merge_protocol)  // the stepwise calls rank the costs in candidate order
        HIP_TRY(hipMemcpyAsync(c->cost_all + c->first, c->cost_raw, c->nloc * sizeof(float), hipMemcpyDeviceToDevice,
                               c->stream));
    if (rc == LMCMA_HIP_OK && c->nloc == c->lambda) c->costs_ready = true;
    return rc;
}

int lmcma_hip_set_costs(lmcma_hip_ctx *c, int first, int count, const double *costs)
{
    CTX_TRY(c);
    if (c->Q > 1) return fail(LMCMA_HIP_ERR_STATE, "set_costs: the stepwise calls serve one query per context");
    if (!costs || first < 0 || count < 1 || first + count > c->lambda)
        return fail(LMCMA_HIP_ERR_ARG, "bad candidate range");
    std::vector<float> f(count);
    for (int i = 0; i < count; ++i) f[i] = (float)costs[i];
    HIP_TRY(hipMemcpyAsync(c->cost_all + first, f.data(), count * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->costs_ready = true;
    return LMCMA_HIP_OK;
}

int lmcma_hip_select(lmcma_hip_ctx *c, float *sorted_cost, int32_t *perm)
{
    CTX_TRY(c);
    // a context that holds the whole population knows whether it has this generation's costs (a
    // shard's other costs arrive through LMCMA_BUF_COST_ALL, which the library cannot see)
    if (c->Q > 1) return fail(LMCMA_HIP_ERR_STATE, "select: the stepwise calls serve one query per context");
    if (c->nloc == c->lambda && !c->costs_ready)
        return fail(LMCMA_HIP_ERR_STATE, "select: no costs yet (evaluate or set_costs must come first)");
    int rc = do_select(c, false, nullptr);
    if (rc != LMCMA_HIP_OK) return rc;
    if (sorted_cost)
        HIP_TRY(hipMemcpyAsync(sorted_cost, c->sorted_cost, c->lambda * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    if (perm)
        HIP_TRY(hipMemcpyAsync(perm, c->perm, c->lambda * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return LMCMA_HIP_OK;
}

int lmcma_hip_recombine(lmcma_hip_ctx *c, const double *weights, double *xmean_out)
{
    CTX_TRY(c);
    if (c->Q > 1) return fail(LMCMA_HIP_ERR_STATE, "recombine: the stepwise calls serve one query per context");
    if (!c->selected) return fail(LMCMA_HIP_ERR_STATE, "select must come first");
    if (weights) {
        HIP_TRY(hipMemcpyAsync(c->weights, weights, c->mu * sizeof(double), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->have_weights = true;
    }
    if (!c->have_weights) return fail(LMCMA_HIP_ERR_STATE, "no recombination weights set");
    int rc = do_recombine(c);
    if (rc == LMCMA_HIP_OK) rc = do_recombine_final(c, recombine_num_blocks(c->mu), false);
    if (rc != LMCMA_HIP_OK) return rc;
    if (xmean_out)
        HIP_TRY(hipMemcpyAsync(xmean_out, c->partial_local, c->N * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return LMCMA_HIP_OK;
}

int lmcma_hip_get_rows(lmcma_hip_ctx *c, int first_local, int count, double *out)
{
    CTX_TRY(c);
    if (!out || first_local < 0 || count < 1 || first_local + count > c->nloc)
        return fail(LMCMA_HIP_ERR_ARG, "bad candidate range");
    std::vector<float> tmp((size_t)count * c->ld);
    HIP_TRY(hipMemcpyAsync(tmp.data(), sel(c, c->rows) + (size_t)first_local * c->ld, tmp.size() * sizeof(float),
                           hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int i = 0; i < count; ++i)
        for (int n = 0; n < c->N; ++n) out[(size_t)i * c->N + n] = (double)tmp[(size_t)i * c->ld + n];
    return LMCMA_HIP_OK;
}

int lmcma_hip_get_costs(lmcma_hip_ctx *c, int first_local, int count, double *out)
{
    CTX_TRY(c);
    if (!out || first_local < 0 || count < 1 || first_local + count > c->nloc)
        return fail(LMCMA_HIP_ERR_ARG, "bad candidate range");
    std::vector<float> tmp(count);
    HIP_TRY(hipMemcpyAsync(tmp.data(), sel(c, local_costs(c)) + first_local, count * sizeof(float),
                           hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int i = 0; i < count; ++i) out[i] = (double)tmp[i];
    return LMCMA_HIP_OK;
}

int lmcma_hip_get_terms(lmcma_hip_ctx *c, int first_local, int count, double *out3)
{
    CTX_TRY(c);
    if (!out3 || first_local < 0 || count < 1 || first_local + count > c->nloc)
        return fail(LMCMA_HIP_ERR_ARG, "bad candidate range");
    std::vector<float> tmp((size_t)count * 3);
    HIP_TRY(hipMemcpyAsync(tmp.data(), sel(c, c->terms) + (size_t)first_local * 3, tmp.size() * sizeof(float),
                           hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (size_t i = 0; i < tmp.size(); ++i) out3[i] = (double)tmp[i];
    return LMCMA_HIP_OK;
}

int lmcma_hip_score_rows(lmcma_hip_ctx *c, const float *rows, int count, float *cost_out, float *terms_out)
{
    CTX_TRY(c);
    if (c->D < 1 || !c->grid) return fail(LMCMA_HIP_ERR_STATE, "set_grid must come first");
    if (!rows || !cost_out || count < 1 || count > c->nloc)
        return fail(LMCMA_HIP_ERR_ARG, "count must be in [1, lambda_local]");
    if (!c->scratch_rows) {
        HIP_TRY(dalloc(&c->scratch_rows, (size_t)c->nloc * c->ld));
        HIP_TRY(dalloc(&c->scratch_cost, (size_t)c->nloc));
        HIP_TRY(dalloc(&c->scratch_terms, (size_t)c->nloc * 3));
    }
    std::vector<float> padded((size_t)count * c->ld, 0.f);
    for (int i = 0; i < count; ++i)
        memcpy(&padded[(size_t)i * c->ld], rows + (size_t)i * c->N, c->N * sizeof(float));
    HIP_TRY(hipMemcpyAsync(c->scratch_rows, padded.data(), padded.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
    CostParams p = cost_params(c, c->scratch_rows, count, c->scratch_cost, c->scratch_terms);
    {
        Timed tm(c, LMCMA_K_COST);
        HIP_TRY(launch_cost(c->stream, p, c->D, c->grid_lds));
    }
    HIP_TRY(hipMemcpyAsync(cost_out, c->scratch_cost, count * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    if (terms_out)
        HIP_TRY(hipMemcpyAsync(terms_out, c->scratch_terms, (size_t)count * 3 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return LMCMA_HIP_OK;
}

/* ---- device-resident optimiser ------------------------------------------------ */

int lmcma_hip_opt_init(lmcma_hip_ctx *c, const double *xmean0, double sigma0)
{
    CTX_TRY(c);
    if (!xmean0 || !(sigma0 > 0.0)) return fail(LMCMA_HIP_ERR_ARG, "xmean0 and a positive sigma0 are needed");
    if (c->m > LMCMA_HIP_MAX_VECTORS)
        return fail(LMCMA_HIP_ERR_UNSUPPORTED, "the device-resident update keeps at most 64 direction vectors");
    HIP_TRY(hipStreamSynchronize(c->stream));
    const int N = c->N, mu = c->mu, m = c->m;
    // weights and mueff, lmcma.cpp:144-156
    std::vector<double> w(mu);
    double sum = 0;
    for (int i = 0; i < mu; ++i) {
        w[i] = log((double)mu + 0.5) - log((double)(1 + i));
        sum += w[i];
    }
    double mueff = 0;
    for (int i = 0; i < mu; ++i) {
        w[i] /= sum;
        mueff += w[i] * w[i];
    }
    mueff = 1.0 / mueff;
    OptScalars o{};
    o.sigma = sigma0;
    o.s = 0.0;
    o.best_f = 1.7976931348623157e308;
    o.counteval = 0;
    o.itr = 0;
    o.iterator_sz = 0;
    o.mueff = mueff;
    o.c1 = 1.0 / (10 * log((double)(N + 1)));  // lmcma.cpp:268
    o.cc = 1.0 / m;                            // lmcma.cpp:269 with nvectors = m
    o.cs = 0.3;
    o.val_target = 0.25;                       // lmcma.cpp:238
    o.K = 1 / sqrt(1 - o.c1);
    o.M = sqrt(1 - o.c1);
    o.sigma0 = sigma0;
    std::vector<float> xm(c->ld, 0.f);
    for (int n = 0; n < N; ++n) xm[n] = (float)xmean0[n];
    SampleScalars sc{(float)sigma0, 1.0f, 0, 0u};
    for (int q = 0; q < c->Q; ++q) {  // every query starts from the same mean and step size (opt_set_query changes one)
        HIP_TRY(hipMemcpy(query_ptr(c, c->opt, q), &o, sizeof(o), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(query_ptr(c, c->weights, q), w.data(), mu * sizeof(double), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(query_ptr(c, c->xmean, q), xmean0, N * sizeof(double), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(query_ptr(c, c->xmean_f, q), xm.data(), c->ld * sizeof(float), hipMemcpyHostToDevice));
        HIP_TRY(hipMemset(query_ptr(c, c->pc, q), 0, N * sizeof(double)));
        HIP_TRY(hipMemset(query_ptr(c, c->pc_arr, q), 0, (size_t)m * N * sizeof(double)));
        HIP_TRY(hipMemset(query_ptr(c, c->v_arr, q), 0, (size_t)m * N * sizeof(double)));
        HIP_TRY(hipMemset(query_ptr(c, c->nj, q), 0, m * sizeof(double)));
        HIP_TRY(hipMemset(query_ptr(c, c->lj, q), 0, m * sizeof(double)));
        HIP_TRY(hipMemset(query_ptr(c, c->t, q), 0, m * sizeof(int32_t)));
        HIP_TRY(hipMemset(query_ptr(c, c->vec, q), 0, m * sizeof(int32_t)));
        HIP_TRY(hipMemset(query_ptr(c, c->iterator, q), 0, m * sizeof(int32_t)));
        HIP_TRY(hipMemset(query_ptr(c, c->psr_accum, q), 0, sizeof(unsigned long long)));
        HIP_TRY(hipMemcpy(query_ptr(c, c->sc, q), &sc, sizeof(sc), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(query_ptr(c, c->xmean0, q), xmean0, N * sizeof(double), hipMemcpyHostToDevice));
    }
    c->have_weights = true;
    HIP_TRY(hipStreamSynchronize(nullptr));  // the fills above: done before the context's stream is used
    if (c->host_flag) *static_cast<volatile int *>(c->host_flag) = 0;  // (the stream was drained above)
    c->sigma0 = sigma0;
    c->opt_ready = true;
    c->pre_done = false;
    c->have_state = true;
    return LMCMA_HIP_OK;
}

int lmcma_hip_opt_reset(lmcma_hip_ctx *c)
{
    CTX_TRY(c);
    if (!c->opt_ready) return fail(LMCMA_HIP_ERR_STATE, "opt_init must come first");
    c->pre_done = false;
    if (c->host_flag && *static_cast<volatile int *>(c->host_flag) != 0) {
        // an update has given up: the reset kernel below clears the word in stream order, but the
        // calls that follow look at it at once, so the (failed) work is waited for and it is cleared here
        HIP_TRY(hipStreamSynchronize(c->stream));
        *static_cast<volatile int *>(c->host_flag) = 0;
    }
    hipLaunchKernelGGL(k_opt_reset, dim3(1, c->Q), dim3(256), 0, c->stream, c->opt, c->sc, c->xmean0,
                       c->xmean, c->xmean_f, c->pc, c->N, c->ld, c->psr_accum, c->qstride, c->fused_counter, c->host_flag_dev);
    HIP_TRY(hipGetLastError());
    return LMCMA_HIP_OK;
}

int lmcma_hip_opt_set_query(lmcma_hip_ctx *c, int query, const double *xmean0, double sigma0)
{
    CTX_TRY(c);
    if (!c->opt_ready) return fail(LMCMA_HIP_ERR_STATE, "opt_init must come first");
    if (query < 0 || query >= c->Q) return fail(LMCMA_HIP_ERR_ARG, "opt_set_query: no such query");
    if (!xmean0 || !(sigma0 > 0.0)) return fail(LMCMA_HIP_ERR_ARG, "xmean0 and a positive sigma0 are needed");
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(query_ptr(c, c->xmean0, query), xmean0, c->N * sizeof(double), hipMemcpyHostToDevice));
    OptScalars *opt = query_ptr(c, c->opt, query);
    HIP_TRY(hipMemcpy(&opt->sigma0, &sigma0, sizeof(double), hipMemcpyHostToDevice));
    // that query alone goes back to its (new) initial state
    hipLaunchKernelGGL(k_opt_reset, dim3(1, 1), dim3(256), 0, c->stream, opt, query_ptr(c, c->sc, query),
                       query_ptr(c, c->xmean0, query), query_ptr(c, c->xmean, query), query_ptr(c, c->xmean_f, query),
                       query_ptr(c, c->pc, query), c->N, c->ld, query_ptr(c, c->psr_accum, query), (size_t)0,
                       (unsigned *)nullptr, (int *)nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    return LMCMA_HIP_OK;
}

int lmcma_hip_query_select(lmcma_hip_ctx *c, int query)
{
    if (!c) return fail(LMCMA_HIP_ERR_ARG, "null context");
    if (query < 0 || query >= c->Q) return fail(LMCMA_HIP_ERR_ARG, "query_select: no such query");
    c->qsel = query;
    return LMCMA_HIP_OK;
}

int lmcma_hip_queries(lmcma_hip_ctx *c) { return c ? c->Q : 0; }

namespace {

// Records `gens` generations, the first of which finds sorted-cost buffer `parity` current, into a
// graph.  A capture records only: none of the recorded generations has run, and the host-side
// bookkeeping they went through (the buffer parity, pre_done) is put back to where it was.
int record_generations(lmcma_hip_ctx *c, int gens, int parity, hipGraph_t *graph, hipGraphExec_t *exec)
{
    const int parity0 = c->parity;
    const bool pre0 = c->pre_done;
    HIP_TRY(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
    c->parity = parity;
    c->sorted_cost = c->sorted_buf[parity];
    const int crc = generations(c, gens);
    hipError_t e = hipStreamEndCapture(c->stream, graph);
    c->parity = parity0;
    c->sorted_cost = c->sorted_buf[parity0];
    c->pre_done = pre0;
    if (crc != LMCMA_HIP_OK) return crc;
    if (e != hipSuccess) return fail(LMCMA_HIP_ERR_RUNTIME, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
    e = hipGraphInstantiate(exec, *graph, nullptr, nullptr, 0);
    if (e != hipSuccess) return fail(LMCMA_HIP_ERR_RUNTIME, std::string("hipGraphInstantiate: ") + hipGetErrorString(e));
    return LMCMA_HIP_OK;
}

}  // namespace

int lmcma_hip_run(lmcma_hip_ctx *c, int n_generations)
{
    CTX_TRY(c);
    if (!c->opt_ready) return fail(LMCMA_HIP_ERR_STATE, "opt_init must come first");
    if (c->nloc != c->lambda) return fail(LMCMA_HIP_ERR_STATE, "run needs the whole population in one context");
    if (c->cfg.z_source != LMCMA_Z_PHILOX || c->cfg.cost_source != LMCMA_COST_TRAJECTORY)
        return fail(LMCMA_HIP_ERR_STATE, "run needs z_source PHILOX and cost_source TRAJECTORY");
    if (!c->grid) return fail(LMCMA_HIP_ERR_STATE, "set_grid must come first");
    if (n_generations < 0) return fail(LMCMA_HIP_ERR_ARG, "negative generation count");
    {
        const int rc = update_gave_up(c);   // (as far as the device has got: nothing is waited for)
        if (rc != LMCMA_HIP_OK) return rc;
    }
    int left = n_generations;
    static const bool no_graph = getenv("LMCMA_HIP_NO_GRAPH") != nullptr;  // (read once, not per call)
    const bool use_graph = !c->timing && !no_graph;
    if (use_graph && left > 0 && !c->graph_exec[0][0]) {
        // The graphs are recorded on the first call, whatever its length, so that no later call
        // pays for it: one eager generation first (module load and LDS attribute calls stay out
        // of the capture), then launches of 32, 8, 2 and 1 generations are recorded, each for both
        // sorted-cost buffers the first of its generations may find current.  A capture records
        // only: none of the recorded generations has run, and the host-side bookkeeping they went
        // through (the buffer parity, pre_done) is put back to where it was.
        if (!c->warmed) {
            // (two when the call has them: the sampling launch that carries an update only runs
            // between two generations of one call, and a capture must not be its first launch)
            const int w = left >= 2 ? 2 : 1;
            int rc = generations(c, w);
            if (rc != LMCMA_HIP_OK) return rc;
            left -= w;
            c->warmed = w == 2;
        }
        for (int q = 0; q < kGraphKinds; ++q) c->graph_gens[q] = kGensPerGraphDefault[q];
        if (const char *e = getenv("LMCMA_HIP_GRAPH_GENS")) {
            // "16,4,1": generations per recorded graph, longest first
            int v[kGraphKinds] = {}, n = 0;
            for (const char *p = e; *p && n < kGraphKinds;) {
                char *end = nullptr;
                const long g = strtol(p, &end, 10);
                if (end == p) break;
                v[n++] = (int)g;
                p = *end == ',' ? end + 1 : end;
            }
            bool ok = n > 0;
            for (int q = 0; q < n; ++q) ok = ok && v[q] >= 1 && v[q] <= 1024 && (q == 0 || v[q] < v[q - 1]);
            if (!ok) return fail(LMCMA_HIP_ERR_ARG, "LMCMA_HIP_GRAPH_GENS: a descending list of generation counts (1..1024) is needed");
            for (int q = 0; q < kGraphKinds; ++q) c->graph_gens[q] = v[q];
        }
        for (int sp = 0; sp < 2; ++sp)
            for (int q = 0; q < kGraphKinds && c->graph_gens[q] > 0; ++q) {
                const int rc = record_generations(c, c->graph_gens[q], sp, &c->graph[sp][q], &c->graph_exec[sp][q]);
                if (rc != LMCMA_HIP_OK) { drop_graph(c); return rc; }
            }
    }
    // a call whose whole length has a graph of its own: one launch
    if (use_graph && left > 0 && left == n_generations)
        for (auto &cg : c->call_graph)
            if (cg.exec && cg.gens == left && cg.parity == c->parity) {
                cg.used = ++c->call_tick;
                HIP_TRY(hipGraphLaunch(cg.exec, c->stream));
                if (left & 1) {
                    c->parity ^= 1;
                    c->sorted_cost = c->sorted_buf[c->parity];
                }
                return LMCMA_HIP_OK;
            }
    for (int q = 0; use_graph && q < kGraphKinds && c->graph_gens[q] > 0; ++q)
        for (; left >= c->graph_gens[q]; left -= c->graph_gens[q]) {
            HIP_TRY(hipGraphLaunch(c->graph_exec[c->parity][q], c->stream));
            if (c->graph_gens[q] & 1) {  // an odd number of generations leaves the other buffer current
                c->parity ^= 1;
                c->sorted_cost = c->sorted_buf[c->parity];
            }
        }
    if (left > 0) {
        int rc = generations(c, left);
        if (rc != LMCMA_HIP_OK) return rc;
    }
    // A caller that steps n generations at a time tends to do so again.  A call cut into several
    // recorded graphs pays a start and an end per piece (and the host a launch each): behind such a
    // call, while the device is busy with it, a graph of the whole length is recorded for the
    // buffer parity the next call will find, and calls of that length replay it from then on.
    if (use_graph && n_generations > 2 && n_generations <= kCallGraphMaxGens) {
        bool piece = false, have = false;
        for (int q = 0; q < kGraphKinds; ++q) piece = piece || c->graph_gens[q] == n_generations;
        for (auto &cg : c->call_graph) have = have || (cg.exec && cg.gens == n_generations && cg.parity == c->parity);
        if (!piece && !have) {
            // a free slot, else the one whose graph was launched longest ago (one-off lengths, such
            // as the pieces an episode boundary cuts a call into, must not hold the slots for good)
            lmcma_hip_ctx::CallGraph *slot = &c->call_graph[0];
            for (auto &cg : c->call_graph) {
                if (!cg.exec) { slot = &cg; break; }
                if (cg.used < slot->used) slot = &cg;
            }
            if (slot->exec) {
                // (its last launch may still be running: a graph in flight must not be destroyed)
                HIP_TRY(hipStreamSynchronize(c->stream));
                (void)hipGraphExecDestroy(slot->exec);
                if (slot->graph) (void)hipGraphDestroy(slot->graph);
                *slot = lmcma_hip_ctx::CallGraph{};
            }
            // (the call's own work is queued already: a graph that cannot be recorded is done
            // without, the call has not failed)
            if (record_generations(c, n_generations, c->parity, &slot->graph, &slot->exec) != LMCMA_HIP_OK) {
                if (slot->graph) (void)hipGraphDestroy(slot->graph);
                *slot = lmcma_hip_ctx::CallGraph{};
            } else {
                slot->gens = n_generations;
                slot->parity = c->parity;
                slot->used = ++c->call_tick;
            }
        }
    }
    return LMCMA_HIP_OK;
}

namespace {

int phase_local_checks(lmcma_hip_ctx *c)
{
    if (!c->opt_ready) return fail(LMCMA_HIP_ERR_STATE, "opt_init must come first");
    if (!c->grid) return fail(LMCMA_HIP_ERR_STATE, "set_grid must come first");
    if (c->cfg.z_source != LMCMA_Z_PHILOX) return fail(LMCMA_HIP_ERR_STATE, "phases need z_source PHILOX");
    return LMCMA_HIP_OK;
}

// sample + score the local candidates
int local_generate(lmcma_hip_ctx *c)
{
    if (c->fused) return do_generate(c);
    int rc = do_sample(c, nullptr);
    if (rc == LMCMA_HIP_OK) rc = do_evaluate(c);
    return rc;
}

// a shard exchanges its costs in ascending order: rank the local candidates among themselves
int local_rank(lmcma_hip_ctx *c)
{
    if (!c->merge_protocol) return LMCMA_HIP_OK;
    Timed tm(c, LMCMA_K_SELECT);
    SelectArgs a{};
    a.cost_all = c->cost_raw;
    a.lambda = c->nloc;
    a.keys_ws = c->keys;
    a.sorted_cost = c->cost_all + c->first;
    a.perm = c->local_perm;
    // (the first part of the update rides along with this launch when it can)
    const UpdateArgs pre = update_args(c, nullptr, 0, false);
    a.pre = c->pre_done ? nullptr : &pre;
    SelectResult res{};
    HIP_TRY(launch_select(c->stream, a, &res));
    if (res.pre_done) c->pre_done = true;
    return LMCMA_HIP_OK;
}

}  // namespace

int lmcma_hip_phase_local(lmcma_hip_ctx *c)
{
    CTX_TRY(c);
    int rc = phase_local_checks(c);
    if (rc == LMCMA_HIP_OK) rc = local_generate(c);
    if (rc == LMCMA_HIP_OK) rc = local_rank(c);
    return rc;
}

namespace {

// sample + score candidates [first, first + count) of the local ones; the rest is left as it is
int generate_range(lmcma_hip_ctx *c, int first, int count)
{
    if (!c->fused) return fail(LMCMA_HIP_ERR_STATE, "a range of candidates needs the one-launch sampling path (LMCMA_HIP_FUSED=0 is set)");
    if (first < 0 || count < 1 || first + count > c->nloc || (first & 3) || (count & 3))
        return fail(LMCMA_HIP_ERR_ARG, "the range must lie inside the local candidates and start and end on a multiple of 4");
    c->gen_first = first;
    c->gen_count = count;
    const int rc = do_generate(c);
    c->gen_first = 0;
    c->gen_count = -1;
    return rc;
}

}  // namespace

int lmcma_hip_phase_local_range(lmcma_hip_ctx *c, int first, int count)
{
    CTX_TRY(c);
    int rc = phase_local_checks(c);
    if (rc != LMCMA_HIP_OK) return rc;
    if (c->merge_protocol || c->Q > 1)
        return fail(LMCMA_HIP_ERR_STATE, "phase_local_range: for a context that ranks its candidates itself (not a shard that "
                                         "exchanges sorted lists, not several queries)");
    return generate_range(c, first, count);
}

int lmcma_hip_phase_rank(lmcma_hip_ctx *c)
{
    CTX_TRY(c);
    if (!c->opt_ready) return fail(LMCMA_HIP_ERR_STATE, "opt_init must come first");
    int rows = 0;
    if (c->merge_protocol) {
        if (!c->have_weights) return fail(LMCMA_HIP_ERR_STATE, "no recombination weights set");
        {
            Timed tm(c, LMCMA_K_SELECT);
            c->parity ^= 1;
            c->sorted_cost = c->sorted_buf[c->parity];
            MergeArgs a{};
            a.lists = c->cost_all;
            a.lambda = c->lambda; a.n_local = c->nloc; a.first_candidate = c->first;
            a.local_perm = c->local_perm;
            a.sorted_cost = c->sorted_cost;
            a.perm = c->perm;
            a.opt = c->opt;
            a.prev_own = c->key_buf[c->parity ^ 1];
            a.own_out = c->key_buf[c->parity];
            a.psr_parts = reinterpret_cast<uint32_t *>(c->keys);
            a.best_out = c->best;
            a.rows = c->rows; a.ld = c->ld; a.N = c->N;
            a.weights = c->weights; a.mu = c->mu;
            a.block_partials = c->block_partials;
            HIP_TRY(launch_rank_merge(c->stream, a, &rows));
            c->psr_parts_n = rows;
            c->ranked_local_only = true;
            c->selected = true;
        }
        return do_recombine_final(c, rows, true);
    }
    int rc = do_select(c, true, &rows, true);
    if (rc != LMCMA_HIP_OK) return rc;
    if (c->Q > 1) {
        // several queries: nothing leaves the context, the update adds the block rows itself
        if (rows == 0) {  // many queries are ranked one workgroup each; their elite rows are added here
            if ((rc = do_recombine(c)) != LMCMA_HIP_OK) return rc;
            rows = recombine_num_blocks(c->mu);
        }
        c->pending_rows = rows;
        return LMCMA_HIP_OK;
    }
    if (rows > 0) return do_recombine_final(c, rows, true);
    rc = do_recombine(c);
    if (rc != LMCMA_HIP_OK) return rc;
    return do_recombine_final(c, recombine_num_blocks(c->mu), true);
}

int lmcma_hip_phase_update(lmcma_hip_ctx *c, int n_partials)
{
    CTX_TRY(c);
    if (!c->opt_ready) return fail(LMCMA_HIP_ERR_STATE, "opt_init must come first");
    if (n_partials < 0 || n_partials > kMaxPartials) return fail(LMCMA_HIP_ERR_ARG, "n_partials must be in [0, 64]");
    if (c->Q > 1) {
        if (n_partials != 0) return fail(LMCMA_HIP_ERR_ARG, "phase_update: a context with several queries is not a shard");
        int rows = c->pending_rows;
        const double *src = nullptr;
        int rc = reduce_if_many(c, &rows, &src);
        if (rc != LMCMA_HIP_OK) return rc;
        return do_update(c, src, rows, false);
    }
    // n_partials = 0: a single context, its own partial sum is the new mean
    if (n_partials == 0) return do_update(c, c->partial_local, 1, true);
    return do_update(c, c->partial_all, n_partials, true);
}

/* ---- sharded population: RCCL behind the boundary ------------------------------------ */

namespace {

// The few RCCL functions the sharded loop needs, bound on first use.  In a process that has
// loaded PyTorch this is PyTorch's own copy of the library (same soname), so there is one
// RCCL and one HIP runtime in the process.
struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

const RcclApi *rccl_api()
{
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {"librccl.so.1", "librccl.so"};
        for (const char *n : names) {
            api.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (api.handle) break;
        }
        if (!api.handle) {
            api.error = std::string("cannot open librccl.so.1: ") + dlerror();
            return;
        }
        auto bind = [&](const char *sym) {
            void *p = dlsym(api.handle, sym);
            if (!p && api.error.empty()) api.error = std::string("librccl has no ") + sym;
            return p;
        };
        api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(bind("ncclGetUniqueId"));
        api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(bind("ncclCommInitRank"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(bind("ncclCommDestroy"));
        api.CommCount = reinterpret_cast<decltype(api.CommCount)>(bind("ncclCommCount"));
        api.CommUserRank = reinterpret_cast<decltype(api.CommUserRank)>(bind("ncclCommUserRank"));
        api.AllGather = reinterpret_cast<decltype(api.AllGather)>(bind("ncclAllGather"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(bind("ncclGetErrorString"));
    });
    return &api;
}

#define RCCL_TRY(api, expr)                                                                        \
    do {                                                                                           \
        ncclResult_t _r = (expr);                                                                  \
        if (_r != ncclSuccess)                                                                     \
            return fail(LMCMA_HIP_ERR_RUNTIME, std::string(#expr) + ": " + (api)->GetErrorString(_r)); \
    } while (0)

// A context that holds the WHOLE population (lambda_local = lambda_total) as one of `world` > 1
// ranks: the rows exchange.  It samples and scores its rank's range of candidates only; the other
// ranges' rows and costs arrive by all-gather, and it ranks and recombines the whole population
// itself, exactly as a single context does (L/lmcma.cpp:313-329 over all of arx), so the state
// is the single context's bit for bit whatever the number of ranks.
bool whole_population_member(const lmcma_hip_ctx *c, int world)
{
    return world > 1 && c->Q == 1 && c->nloc == c->lambda && c->first == 0 && c->lambda % world == 0 &&
           (c->lambda / world) % 16 == 0;
}

int check_shard(const lmcma_hip_ctx *c, int world, int rank)
{
    if (world < 1 || world > kMaxPartials || rank < 0 || rank >= world)
        return fail(LMCMA_HIP_ERR_ARG, "comm: world must be in [1, 64] and rank in [0, world)");
    if (c->Q > 1) return fail(LMCMA_HIP_ERR_STATE, "comm: a context with several queries is not a shard");
    if (whole_population_member(c, world)) return LMCMA_HIP_OK;
    if ((long)c->nloc * world != c->lambda || c->first != rank * c->nloc)
        return fail(LMCMA_HIP_ERR_STATE, "comm: the context is neither this rank's shard (lambda_total = world * lambda_local "
                                         "and first_candidate = rank * lambda_local) nor a whole population that "
                                         "world ranks can divide into ranges of a multiple of 16 candidates");
    return LMCMA_HIP_OK;
}

}  // namespace

int lmcma_hip_comm_unique_id(char id[LMCMA_HIP_UNIQUE_ID_BYTES])
{
    static_assert(sizeof(ncclUniqueId) == LMCMA_HIP_UNIQUE_ID_BYTES, "ncclUniqueId is 128 bytes");
    if (!id) return fail(LMCMA_HIP_ERR_ARG, "null id");
    const RcclApi *api = rccl_api();
    if (!api->error.empty()) return fail(LMCMA_HIP_ERR_UNSUPPORTED, api->error);
    ncclUniqueId u;
    RCCL_TRY(api, api->GetUniqueId(&u));
    memcpy(id, &u, sizeof(u));
    return LMCMA_HIP_OK;
}

int lmcma_hip_comm_init(lmcma_hip_ctx *c, const char id[LMCMA_HIP_UNIQUE_ID_BYTES], int world, int rank)
{
    CTX_TRY(c);
    if (!id) return fail(LMCMA_HIP_ERR_ARG, "null id");
    if (c->comm) return fail(LMCMA_HIP_ERR_STATE, "comm_init: the context has a communicator already");
    int rc = check_shard(c, world, rank);
    if (rc != LMCMA_HIP_OK) return rc;
    const RcclApi *api = rccl_api();
    if (!api->error.empty()) return fail(LMCMA_HIP_ERR_UNSUPPORTED, api->error);
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    ncclComm_t comm = nullptr;
    RCCL_TRY(api, api->CommInitRank(&comm, world, u, rank));   // on the context's device (CTX_TRY made it current)
    c->comm = comm;
    c->own_comm = true;
    c->world = world;
    c->rank = rank;
    return LMCMA_HIP_OK;
}

int lmcma_hip_comm_attach(lmcma_hip_ctx *c, void *nccl_comm, int world, int rank)
{
    CTX_TRY(c);
    if (!nccl_comm) return fail(LMCMA_HIP_ERR_ARG, "null communicator");
    if (c->comm) return fail(LMCMA_HIP_ERR_STATE, "comm_attach: the context has a communicator already");
    int rc = check_shard(c, world, rank);
    if (rc != LMCMA_HIP_OK) return rc;
    const RcclApi *api = rccl_api();
    if (!api->error.empty()) return fail(LMCMA_HIP_ERR_UNSUPPORTED, api->error);
    int count = 0, urank = -1;
    RCCL_TRY(api, api->CommCount(static_cast<ncclComm_t>(nccl_comm), &count));
    RCCL_TRY(api, api->CommUserRank(static_cast<ncclComm_t>(nccl_comm), &urank));
    if (count != world || urank != rank) return fail(LMCMA_HIP_ERR_ARG, "comm_attach: world / rank differ from the communicator's");
    c->comm = static_cast<ncclComm_t>(nccl_comm);
    c->own_comm = false;
    c->world = world;
    c->rank = rank;
    return LMCMA_HIP_OK;
}

int lmcma_hip_comm_destroy(lmcma_hip_ctx *c)
{
    if (!c) return fail(LMCMA_HIP_ERR_ARG, "null context");
    if (c->comm && c->own_comm) {
        (void)hipSetDevice(c->cfg.device);
        if (c->stream) (void)hipStreamSynchronize(c->stream);
        const RcclApi *api = rccl_api();
        if (api->CommDestroy) (void)api->CommDestroy(c->comm);
    }
    if (c->comm) drop_graph(c);   // a recorded sharded generation holds this communicator's collectives
    c->comm = nullptr;
    c->own_comm = false;
    if (!c->peer_mapped) {
        c->world = 0;
        c->rank = 0;
    }
    return LMCMA_HIP_OK;
}

/* ---- sharded population: the direct exchange ------------------------------------------ */

namespace {

// The exchange of the sorted costs in one launch.  Send: the shard's list (`quads` 16-byte units at
// src) into every mailbox of `s`, one unit per thread across the grid, as write-through stores; every
// workgroup then bumps every destination's counter (gridDim.x arrivals per source and exchange).
// Receive: wait for the peers' counters, then copy their lists (all of the mailbox's cost buffer but
// the units [skip0, skip1) of the shard's own list) into the context's buffer of all costs.
__global__ __launch_bounds__(256) void k_peer_exchange(const PeerSend s, const ulonglong2 *__restrict__ src, int quads,
                                                       const PeerWait w, const uint4 *__restrict__ box, uint4 *__restrict__ all,
                                                       int all_quads, int skip0, int skip1)
{
    for (int i = blockIdx.x * 256 + threadIdx.x; i < s.n * quads; i += gridDim.x * 256) {
        const int d = i / quads, q = i - d * quads;
        const ulonglong2 v = src[q];
        unsigned long long *dst = static_cast<unsigned long long *>(s.dst[d]) + 2 * (size_t)q;
        peer_store(dst, v.x);
        peer_store(dst + 1, v.y);
    }
    peer_arrive(s);
    peer_wait(w);
    for (int q = blockIdx.x * 256 + threadIdx.x; q < all_quads; q += gridDim.x * 256)
        if (q < skip0 || q >= skip1) all[q] = box[q];
}

// The receiving side of the exchange of the rows: every workgroup waits for the peers' arrival
// counters, then the launch copies the rows from the mailbox into the context's own buffer.
__global__ __launch_bounds__(256) void k_peer_gather(const PeerWait w, const uint4 *__restrict__ src, uint4 *__restrict__ dst,
                                                     int quads)
{
    peer_wait(w);
    for (int q = blockIdx.x * 256 + threadIdx.x; q < quads; q += gridDim.x * 256) dst[q] = src[q];
}

// layout of a mailbox (the same in every shard of a population: it depends on lambda and N only)
constexpr size_t kMailboxFlagBytes = 1024;   // [0, 64) cost counters, [64, 128) row counters, [128] error word
constexpr int kMailboxRowFlags = 64, kMailboxError = 128;

size_t mailbox_cost_offset() { return kMailboxFlagBytes; }
size_t mailbox_rows_offset(int lambda) { return kMailboxFlagBytes + (((size_t)lambda * sizeof(float) + 16 + 255) & ~(size_t)255); }
size_t mailbox_total(int lambda, int N) { return mailbox_rows_offset(lambda) + (size_t)kMaxPeerRanks * (N + 2) * sizeof(double) + 256; }

struct PeerHandle {
    uint32_t magic;
    int32_t pid, device, fine;
    uint64_t ptr, bytes;
    int32_t lambda, N;
    hipIpcMemHandle_t ipc;
};
static_assert(sizeof(PeerHandle) <= LMCMA_HIP_PEER_HANDLE_BYTES, "the handle blob holds a PeerHandle");
constexpr uint32_t kPeerMagic = 0x4c4d5058u;

uint32_t *mailbox_flags(void *base) { return static_cast<uint32_t *>(base); }
float *mailbox_cost(void *base) { return reinterpret_cast<float *>(static_cast<char *>(base) + mailbox_cost_offset()); }
double *mailbox_rows(void *base, int lambda)
{
    return reinterpret_cast<double *>(static_cast<char *>(base) + mailbox_rows_offset(lambda));
}

int ensure_mailbox(lmcma_hip_ctx *c)
{
    if (c->mailbox) return LMCMA_HIP_OK;
    const size_t bytes = mailbox_total(c->lambda, c->N);
    void *p = nullptr;
    // fine-grained: stores of a peer become visible to this device's kernels while they run
    hipError_t e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained);
    c->mailbox_fine = e == hipSuccess;
    if (e != hipSuccess) {
        (void)hipGetLastError();
        HIP_TRY(hipMalloc(&p, bytes));
    }
    HIP_TRY(hipMemset(p, 0, bytes));
    HIP_TRY(hipStreamSynchronize(nullptr));
    c->mailbox = p;
    c->mailbox_bytes = bytes;
    return LMCMA_HIP_OK;
}

// where this shard's block of exchange `which` (0: sorted costs, 1: its row) goes: every peer's
// mailbox, and for the row the own mailbox too (the costs' own block never leaves the context)
PeerSend peer_send_table(const lmcma_hip_ctx *c, int which)
{
    PeerSend s{};
    const int row = c->N + 2;
    for (int r = 0; r < c->world; ++r) {
        if (which == 0 && r == c->rank) continue;
        char *base = static_cast<char *>(r == c->rank ? c->mailbox : c->peer_base[r]);
        // (an image of this shard stands in for shard r: the block lands in r's place of the own mailbox)
        const int from = (c->peer_image && r != c->rank) ? r : c->rank;
        const int d = s.n++;
        if (which == 0) {
            s.dst[d] = mailbox_cost(base) + (size_t)from * c->nloc;
            s.flag[d] = mailbox_flags(base) + from;
        } else {
            s.dst[d] = mailbox_rows(base, c->lambda) + (size_t)from * row;
            // (the own mailbox's counter for the own row moves too: the kernel that sends may be the one
            // that copies the gathered rows, and its workgroups need each other's strips)
            s.flag[d] = mailbox_flags(base) + kMailboxRowFlags + from;
        }
    }
    return s;
}

PeerSend peer_send_table_if_sending(const lmcma_hip_ctx *c, int which)
{
    return (c->peer_on && c->peer_sending) ? peer_send_table(c, which) : PeerSend{};
}

// The exchange of the sorted costs: one launch sends this shard's list and receives the peers'.
int peer_exchange_costs(lmcma_hip_ctx *c)
{
    const PeerSend s = peer_send_table(c, 0);
    const int quads = c->lambda / 4, own0 = c->first / 4;   // (lambda_local is a multiple of 64)
    const int blocks = std::min(32, (std::max(s.n * (c->nloc / 4), quads) + 255) / 256);
    PeerWait &w = c->wait_costs;
    w.want = c->sent_costs += (uint32_t)blocks;
    hipLaunchKernelGGL(k_peer_exchange, dim3(blocks), dim3(256), 0, c->stream, s,
                       reinterpret_cast<const ulonglong2 *>(c->cost_all + c->first), c->nloc / 4, w,
                       reinterpret_cast<const uint4 *>(mailbox_cost(c->mailbox)), reinterpret_cast<uint4 *>(c->cost_all),
                       quads, own0, own0 + c->nloc / 4);
    HIP_TRY(hipGetLastError());
    return LMCMA_HIP_OK;
}

// The receiving side of the exchange of the rows, queued behind the kernel that sent this shard's
// (`arrivals` counter increments per source): wait for the peers, then mailbox -> partial_all.
int peer_receive_rows(lmcma_hip_ctx *c, int arrivals)
{
    PeerWait &w = c->wait_rows;
    w.want = c->sent_rows += (uint32_t)arrivals;
    const int quads = (c->world * (c->N + 2) * (int)sizeof(double) + 15) / 16;   // (both buffers end in slack)
    hipLaunchKernelGGL(k_peer_gather, dim3(std::min(8, (quads + 255) / 256)), dim3(256), 0, c->stream, w,
                       reinterpret_cast<const uint4 *>(mailbox_rows(c->mailbox, c->lambda)),
                       reinterpret_cast<uint4 *>(c->partial_all), quads);
    HIP_TRY(hipGetLastError());
    return LMCMA_HIP_OK;
}

bool peer_gather_in_sender(lmcma_hip_ctx *c, FinalExtras &x, int arrivals)
{
    static const bool off = getenv("LMCMA_HIP_PEER_GATHER_LAUNCH") != nullptr;   // (a measurement switch)
    if (off) return false;
    PeerWait &w = c->wait_rows;
    w.want = c->sent_rows += (uint32_t)arrivals;
    x.gather = w;
    x.gather.self = -1;   // the own row's counter is waited for as well
    x.gather_src = reinterpret_cast<const uint4 *>(mailbox_rows(c->mailbox, c->lambda));
    x.gather_dst = reinterpret_cast<uint4 *>(c->partial_all);
    x.gather_quads = (c->world * (c->N + 2) * (int)sizeof(double) + 15) / 16;   // (both buffers end in slack)
    return true;
}

// One generation of a shard, in the three stretches between which the shards exchange.
// update_due: the update of the generation before has not run yet (the gathered rows are in
// partial_all): it runs at the head of this generation's sampling launch where that form exists,
// as between two generations of lmcma_hip_run, else as a launch of its own first.
int shard_stretch_local(lmcma_hip_ctx *c, bool update_due)
{
    int rc = phase_local_checks(c);
    if (rc != LMCMA_HIP_OK) return rc;
    bool carried = false;
    if (update_due) {
        const UpdateArgs post = update_args(c, c->partial_all, c->world, true);
        if (c->fused && (rc = do_generate(c, &post, &carried)) != LMCMA_HIP_OK) return rc;
        if (!carried && (rc = do_update(c, c->partial_all, c->world, true)) != LMCMA_HIP_OK) return rc;
    }
    if (!carried && (rc = local_generate(c)) != LMCMA_HIP_OK) return rc;
    return local_rank(c);
}

// leave_update: this generation's update is left to the next generation's first stretch.
int sharded_generation(lmcma_hip_ctx *c, const RcclApi *api, bool update_due = false, bool leave_update = false)
{
    const int row = c->N + 2;
    // direct exchange: the kernels that produce the shard's two blocks store them into the peers'
    // mailboxes themselves where that pays (do_recombine_final looks at peer_sending)
    c->peer_sending = c->peer_on;
    c->peer_arrivals = 0;
    int rc = shard_stretch_local(c, update_due);
    if (rc == LMCMA_HIP_OK && c->peer_on) {
        ++c->peer_epoch;
        // (the ranking scatters its list four bytes at a time: sent from there, 28 000 small stores
        // into uncached memory made that launch 24 us instead of 7)
        rc = peer_exchange_costs(c);
    } else if (rc == LMCMA_HIP_OK) {
        // in place: this shard's block lies at its offset inside the buffer of all costs
        RCCL_TRY(api, api->AllGather(c->cost_all + c->first, c->cost_all, (size_t)c->nloc, ncclFloat32, c->comm, c->stream));
    }
    c->peer_arrivals = 0;
    if (rc == LMCMA_HIP_OK) rc = lmcma_hip_phase_rank(c);
    if (rc == LMCMA_HIP_OK && c->peer_on) {
        if (!c->peer_rows_received) rc = peer_receive_rows(c, c->peer_arrivals);
    } else if (rc == LMCMA_HIP_OK) {
        RCCL_TRY(api, api->AllGather(c->partial_local, c->partial_all, (size_t)row, ncclFloat64, c->comm, c->stream));
    }
    c->peer_sending = false;
    if (rc != LMCMA_HIP_OK) return rc;
    return leave_update ? LMCMA_HIP_OK : lmcma_hip_phase_update(c, c->world);
}

// One generation of the rows exchange (whole_population_member): this rank's range sampled and
// scored, rows and costs of all ranges gathered in place, then the whole population ranked,
// recombined and the state updated as in a single context.
int gathered_generation(lmcma_hip_ctx *c, const RcclApi *api)
{
    const int cnt = c->lambda / c->world, first = c->rank * cnt;
    int rc = phase_local_checks(c);
    if (rc == LMCMA_HIP_OK) rc = generate_range(c, first, cnt);
    if (rc != LMCMA_HIP_OK) return rc;
    RCCL_TRY(api, api->AllGather(c->rows + (size_t)first * c->ld, c->rows, (size_t)cnt * c->ld, ncclFloat32, c->comm, c->stream));
    RCCL_TRY(api, api->AllGather(c->cost_all + first, c->cost_all, (size_t)cnt, ncclFloat32, c->comm, c->stream));
    rc = lmcma_hip_phase_rank(c);
    if (rc == LMCMA_HIP_OK) rc = lmcma_hip_phase_update(c, 0);
    return rc;
}

}  // namespace

int lmcma_hip_peer_handle(lmcma_hip_ctx *c, char handle[LMCMA_HIP_PEER_HANDLE_BYTES])
{
    CTX_TRY(c);
    if (!handle) return fail(LMCMA_HIP_ERR_ARG, "null handle");
    if (c->Q > 1) return fail(LMCMA_HIP_ERR_STATE, "peer_handle: a context with several queries is not a shard");
    if (!c->merge_protocol)
        return fail(LMCMA_HIP_ERR_UNSUPPORTED, "peer_handle: the direct exchange serves shards that exchange sorted lists "
                                                "(lambda_local a multiple of 64, at most 32768, lambda_total at most 2^19)");
    int rc = ensure_mailbox(c);
    if (rc != LMCMA_HIP_OK) return rc;
    PeerHandle h{};
    h.magic = kPeerMagic;
    h.pid = (int32_t)getpid();
    h.device = c->cfg.device;
    h.fine = c->mailbox_fine ? 1 : 0;
    h.ptr = (uint64_t)(uintptr_t)c->mailbox;
    h.bytes = c->mailbox_bytes;
    h.lambda = c->lambda;
    h.N = c->N;
    // (a process that only talks to itself needs no IPC handle; a failure here surfaces when another process opens it)
    if (hipIpcGetMemHandle(&h.ipc, c->mailbox) != hipSuccess) {
        (void)hipGetLastError();
        memset(&h.ipc, 0, sizeof(h.ipc));
        h.fine |= 2;  // no IPC handle
    }
    memset(handle, 0, LMCMA_HIP_PEER_HANDLE_BYTES);
    memcpy(handle, &h, sizeof(h));
    return LMCMA_HIP_OK;
}

int lmcma_hip_peer_disconnect(lmcma_hip_ctx *c)
{
    if (!c) return fail(LMCMA_HIP_ERR_ARG, "null context");
    (void)hipSetDevice(c->cfg.device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (int r = 0; r < kMaxPeerRanks; ++r) {
        if (c->peer_base[r] && c->peer_ipc[r]) (void)hipIpcCloseMemHandle(c->peer_base[r]);
        c->peer_base[r] = nullptr;
        c->peer_ipc[r] = false;
    }
    if (c->peer_mapped) {
        drop_graph(c);
        c->peer_on = false;
        c->peer_mapped = false;
        c->peer_image = false;
        c->wait_costs = PeerWait{};
        c->wait_rows = PeerWait{};
        if (!c->comm) { c->world = 0; c->rank = 0; }
    }
    return LMCMA_HIP_OK;
}

int lmcma_hip_peer_connect(lmcma_hip_ctx *c, const char *handles, int world, int rank)
{
    CTX_TRY(c);
    if (!handles) return fail(LMCMA_HIP_ERR_ARG, "null handles");
    if (c->peer_mapped) return fail(LMCMA_HIP_ERR_STATE, "peer_connect: the context is connected already");
    if (!c->mailbox) return fail(LMCMA_HIP_ERR_STATE, "peer_connect: lmcma_hip_peer_handle must come first");
    if (world > kMaxPeerRanks) return fail(LMCMA_HIP_ERR_UNSUPPORTED, "peer_connect: at most 16 shards exchange directly");
    int rc = check_shard(c, world, rank);
    if (rc != LMCMA_HIP_OK) return rc;
    if (c->comm && (c->world != world || c->rank != rank))
        return fail(LMCMA_HIP_ERR_ARG, "peer_connect: world / rank differ from the communicator's");
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int r = 0; r < world; ++r) {
        PeerHandle h;
        memcpy(&h, handles + (size_t)r * LMCMA_HIP_PEER_HANDLE_BYTES, sizeof(h));
        if (h.magic != kPeerMagic || h.lambda != c->lambda || h.N != c->N || h.bytes != c->mailbox_bytes) {
            (void)lmcma_hip_peer_disconnect(c);
            return fail(LMCMA_HIP_ERR_ARG, "peer_connect: handle " + std::to_string(r) + " is not the handle of a shard of this population");
        }
        if (r == rank) {
            if (h.ptr != (uint64_t)(uintptr_t)c->mailbox || h.pid != (int32_t)getpid()) {
                (void)lmcma_hip_peer_disconnect(c);
                return fail(LMCMA_HIP_ERR_ARG, "peer_connect: handle `rank` is not this context's");
            }
            continue;
        }
        if (h.pid == (int32_t)getpid() && h.ptr == (uint64_t)(uintptr_t)c->mailbox) {
            // this context's own handle in another shard's place: an image of itself (a measurement
            // aid: one shard timed with the exchange traffic and the waits of a population of `world`)
            c->peer_base[r] = c->mailbox;
            c->peer_image = true;
            continue;
        }
        if (h.pid == (int32_t)getpid()) {
            // a shard of this process: its pointer as it is; another device's memory needs peer access
            if (h.device != c->cfg.device) {
                const hipError_t e = hipDeviceEnablePeerAccess(h.device, 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) {
                    (void)lmcma_hip_peer_disconnect(c);
                    return fail(LMCMA_HIP_ERR_RUNTIME, std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e));
                }
                (void)hipGetLastError();
            }
            c->peer_base[r] = (void *)(uintptr_t)h.ptr;
        } else {
            if (h.fine & 2) {
                (void)lmcma_hip_peer_disconnect(c);
                return fail(LMCMA_HIP_ERR_RUNTIME, "peer_connect: shard " + std::to_string(r) + " could not export its mailbox (hipIpcGetMemHandle failed there)");
            }
            void *p = nullptr;
            const hipError_t e = hipIpcOpenMemHandle(&p, h.ipc, hipIpcMemLazyEnablePeerAccess);
            if (e != hipSuccess) {
                (void)lmcma_hip_peer_disconnect(c);
                return fail(LMCMA_HIP_ERR_RUNTIME, std::string("hipIpcOpenMemHandle: ") + hipGetErrorString(e));
            }
            c->peer_base[r] = p;
            c->peer_ipc[r] = true;
        }
    }
    drop_graph(c);
    uint32_t *flags = mailbox_flags(c->mailbox);
    c->wait_costs = PeerWait{flags, 0u, world, rank, flags + kMailboxError};
    c->wait_rows = PeerWait{flags + kMailboxRowFlags, 0u, world, rank, flags + kMailboxError};
    // (peer_epoch is not reset: the counters in the mailboxes only grow, and every shard has made
    // the same number of exchanges)
    c->world = world;
    c->rank = rank;
    c->peer_on = true;
    c->peer_mapped = true;
    return LMCMA_HIP_OK;
}

int lmcma_hip_peer_enable(lmcma_hip_ctx *c, int on)
{
    CTX_TRY(c);
    if (!c->peer_mapped) return fail(LMCMA_HIP_ERR_STATE, "peer_enable: lmcma_hip_peer_connect must come first");
    if ((on != 0) == c->peer_on) return LMCMA_HIP_OK;
    HIP_TRY(hipStreamSynchronize(c->stream));
    drop_graph(c);
    c->peer_on = on != 0;
    return LMCMA_HIP_OK;
}

int lmcma_hip_peer_status(lmcma_hip_ctx *c, int *fine_grained, int *wait_error)
{
    CTX_TRY(c);
    if (!c->mailbox) return fail(LMCMA_HIP_ERR_STATE, "peer_status: the context has no mailbox");
    HIP_TRY(hipStreamSynchronize(c->stream));
    uint32_t err[4] = {0, 0, 0, 0};
    HIP_TRY(hipMemcpy(err, mailbox_flags(c->mailbox) + kMailboxError, sizeof(err), hipMemcpyDeviceToHost));
    if (fine_grained) *fine_grained = c->mailbox_fine ? 1 : 0;
    if (wait_error) *wait_error = (int)err[0];
    if (err[0] != 0)
        g_last_error = "wait gave up: counter " + std::to_string(err[3]) + " held " + std::to_string(err[1]) + ", wanted " +
                       std::to_string(err[2]) + " (exchange " + std::to_string(c->peer_epoch) + " of this shard)";
    return LMCMA_HIP_OK;
}

int lmcma_hip_run_sharded(lmcma_hip_ctx *c, int n_generations)
{
    CTX_TRY(c);
    if (!c->comm && !c->peer_on) return fail(LMCMA_HIP_ERR_STATE, "run_sharded: comm_init, comm_attach or peer_connect must come first");
    if (!c->opt_ready) return fail(LMCMA_HIP_ERR_STATE, "opt_init must come first");
    if (n_generations < 0) return fail(LMCMA_HIP_ERR_ARG, "negative generation count");
    const RcclApi *api = c->peer_on ? nullptr : rccl_api();
    int left = n_generations;
    // (LMCMA_HIP_ROWS_EXCHANGE=1: a whole population with a one-rank communicator goes through the rows
    // exchange too, all-gathers of one rank included: the form's calls, testable on one GPU)
    static const bool rows_with_one_rank = getenv("LMCMA_HIP_ROWS_EXCHANGE") != nullptr;
    if (!c->peer_on && (whole_population_member(c, c->world) ||
                        (rows_with_one_rank && c->world == 1 && c->nloc == c->lambda && c->Q == 1 && c->lambda % 16 == 0))) {
        for (; left > 0; --left) {
            const int rc = gathered_generation(c, api);
            if (rc != LMCMA_HIP_OK) return rc;
        }
        return LMCMA_HIP_OK;
    }
    // Opt-in (LMCMA_HIP_SHARD_GRAPH=1): two generations, collectives included, recorded into one
    // HIP graph and replayed (two because the sorted-cost buffers alternate).  Queuing a generation
    // from the host takes ~17 us without it, well below a sharded generation's device time, so the
    // default is the plain loop; every rank of a population must make the same choice.
    static const bool want_graph = getenv("LMCMA_HIP_SHARD_GRAPH") != nullptr && getenv("LMCMA_HIP_NO_GRAPH") == nullptr;
    if (want_graph && !c->peer_on && !c->timing && left > 0) {
        if (!c->shard_graph_exec) {
            if (!c->warmed) {
                int rc = sharded_generation(c, api);   // module loads and LDS attribute calls stay out of the capture
                if (rc != LMCMA_HIP_OK) return rc;
                --left;
                c->warmed = true;
            }
            c->shard_graph_parity = c->parity;
            HIP_TRY(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
            int crc = LMCMA_HIP_OK;
            for (int g = 0; g < 2 && crc == LMCMA_HIP_OK; ++g) crc = sharded_generation(c, api);
            hipError_t e = hipStreamEndCapture(c->stream, &c->shard_graph);
            if (crc != LMCMA_HIP_OK) { drop_graph(c); return crc; }
            if (e != hipSuccess) { drop_graph(c); return fail(LMCMA_HIP_ERR_RUNTIME, std::string("hipStreamEndCapture: ") + hipGetErrorString(e)); }
            e = hipGraphInstantiate(&c->shard_graph_exec, c->shard_graph, nullptr, nullptr, 0);
            if (e != hipSuccess) { drop_graph(c); return fail(LMCMA_HIP_ERR_RUNTIME, std::string("hipGraphInstantiate: ") + hipGetErrorString(e)); }
        }
        if (left > 0 && c->parity != c->shard_graph_parity) {
            int rc = sharded_generation(c, api);
            if (rc != LMCMA_HIP_OK) return rc;
            --left;
        }
        for (; left >= 2; left -= 2) HIP_TRY(hipGraphLaunch(c->shard_graph_exec, c->stream));
    }
    // inside the call a generation's update rides with the next generation's sampling launch; the
    // last one's is a launch of its own, so the state is complete when the call's work has finished
    for (bool due = false; left > 0; --left) {
        int rc = sharded_generation(c, api, due, left > 1);
        if (rc != LMCMA_HIP_OK) return rc;
        due = true;
    }
    return LMCMA_HIP_OK;
}

int lmcma_hip_run_sharded_group(lmcma_hip_ctx **ctxs, int world, int n_generations)
{
    if (!ctxs || world < 1 || world > kMaxPartials) return fail(LMCMA_HIP_ERR_ARG, "run_sharded_group: 1 to 64 contexts are needed");
    if (n_generations < 0) return fail(LMCMA_HIP_ERR_ARG, "negative generation count");
    for (int r = 0; r < world; ++r) {
        lmcma_hip_ctx *c = ctxs[r];
        CTX_TRY(c);
        int rc = check_shard(c, world, r);
        if (rc != LMCMA_HIP_OK) return rc;
        if (!c->opt_ready) return fail(LMCMA_HIP_ERR_STATE, "opt_init must come first");
        if (c->lambda != ctxs[0]->lambda || c->N != ctxs[0]->N || c->nloc != ctxs[0]->nloc)
            return fail(LMCMA_HIP_ERR_ARG, "run_sharded_group: the contexts are not shards of one population");
        if (c->peer_mapped || c->comm) return fail(LMCMA_HIP_ERR_STATE, "run_sharded_group: the contexts must not be connected otherwise");
        for (int q = 0; q < r; ++q)
            if (ctxs[q] == c) return fail(LMCMA_HIP_ERR_ARG, "run_sharded_group: a context is named twice");
        for (hipEvent_t &ev : c->group_ev)
            if (!ev) HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    }
    if (whole_population_member(ctxs[0], world)) {
        // The rows exchange: every context holds the whole population and samples its range.  ev[0]:
        // its range is sampled; ev[1]: it has copied the others' ranges (which they may then overwrite).
        const size_t cnt = ctxs[0]->lambda / world, ld = ctxs[0]->ld;
        for (int g = 0; g < n_generations; ++g) {
            for (int r = 0; r < world; ++r) {
                lmcma_hip_ctx *c = ctxs[r];
                CTX_TRY(c);
                int rc = phase_local_checks(c);
                if (rc != LMCMA_HIP_OK) return rc;
                // (also across calls: the other contexts may still be copying this one's range of the
                // call before)
                for (int p = 0; p < world; ++p)
                    if (p != r && ctxs[p]->group_copied) HIP_TRY(hipStreamWaitEvent(c->stream, ctxs[p]->group_ev[1], 0));
                rc = generate_range(c, (int)(r * cnt), (int)cnt);
                if (rc != LMCMA_HIP_OK) return rc;
                HIP_TRY(hipEventRecord(c->group_ev[0], c->stream));
            }
            for (int r = 0; r < world; ++r) {
                lmcma_hip_ctx *c = ctxs[r];
                CTX_TRY(c);
                for (int p = 0; p < world; ++p) {
                    if (p == r) continue;
                    const lmcma_hip_ctx *o = ctxs[p];
                    HIP_TRY(hipStreamWaitEvent(c->stream, o->group_ev[0], 0));
                    HIP_TRY(hipMemcpyPeerAsync(c->rows + p * cnt * ld, c->cfg.device, o->rows + p * cnt * ld, o->cfg.device,
                                               cnt * ld * sizeof(float), c->stream));
                    HIP_TRY(hipMemcpyPeerAsync(c->cost_all + p * cnt, c->cfg.device, o->cost_all + p * cnt, o->cfg.device,
                                               cnt * sizeof(float), c->stream));
                }
                HIP_TRY(hipEventRecord(c->group_ev[1], c->stream));
                c->group_copied = true;
                int rc = lmcma_hip_phase_rank(c);
                if (rc == LMCMA_HIP_OK) rc = lmcma_hip_phase_update(c, 0);
                if (rc != LMCMA_HIP_OK) return rc;
            }
        }
        return LMCMA_HIP_OK;
    }
    const size_t nloc = ctxs[0]->nloc, row = ctxs[0]->N + 2;
    for (int g = 0; g < n_generations; ++g) {
        for (int r = 0; r < world; ++r) {
            lmcma_hip_ctx *c = ctxs[r];
            CTX_TRY(c);
            c->world = world;  // (the update reads `world` gathered rows)
            c->rank = r;
            int rc = shard_stretch_local(c, g > 0);
            if (rc != LMCMA_HIP_OK) return rc;
            HIP_TRY(hipEventRecord(c->group_ev[0], c->stream));
        }
        for (int r = 0; r < world; ++r) {
            lmcma_hip_ctx *c = ctxs[r];
            CTX_TRY(c);
            for (int p = 0; p < world; ++p) {
                if (p == r) continue;
                const lmcma_hip_ctx *o = ctxs[p];
                HIP_TRY(hipStreamWaitEvent(c->stream, o->group_ev[0], 0));
                HIP_TRY(hipMemcpyPeerAsync(c->cost_all + p * nloc, c->cfg.device, o->cost_all + p * nloc, o->cfg.device,
                                           nloc * sizeof(float), c->stream));
            }
            int rc = lmcma_hip_phase_rank(c);
            if (rc != LMCMA_HIP_OK) return rc;
            HIP_TRY(hipEventRecord(c->group_ev[1], c->stream));
        }
        for (int r = 0; r < world; ++r) {
            lmcma_hip_ctx *c = ctxs[r];
            CTX_TRY(c);
            for (int p = 0; p < world; ++p) {
                const lmcma_hip_ctx *o = ctxs[p];
                if (p != r) HIP_TRY(hipStreamWaitEvent(c->stream, o->group_ev[1], 0));
                HIP_TRY(hipMemcpyPeerAsync(c->partial_all + p * row, c->cfg.device, o->partial_local, o->cfg.device,
                                           row * sizeof(double), c->stream));
            }
            if (g == n_generations - 1) {
                int rc = lmcma_hip_phase_update(c, world);
                if (rc != LMCMA_HIP_OK) return rc;
            }
        }
    }
    for (int r = 0; r < world; ++r) { ctxs[r]->world = 0; ctxs[r]->rank = 0; }
    return LMCMA_HIP_OK;
}

int lmcma_hip_opt_get(lmcma_hip_ctx *c, lmcma_hip_opt_summary *summary, double *xmean)
{
    CTX_TRY(c);
    if (!c->opt_ready) return fail(LMCMA_HIP_ERR_STATE, "opt_init must come first");
    OptScalars o;
    HIP_TRY(hipMemcpyAsync(&o, sel(c, c->opt), sizeof(o), hipMemcpyDeviceToHost, c->stream));
    if (xmean) HIP_TRY(hipMemcpyAsync(xmean, sel(c, c->xmean), c->N * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (o.done == 3)
        return fail(LMCMA_HIP_ERR_RUNTIME, "an update gave up waiting for the other workgroups of its launch; the state is stale");
    if (c->mailbox && c->peer_on) {  // (a run that has gone back to the communicator is not held to an earlier failure)
        uint32_t err = 0;
        HIP_TRY(hipMemcpy(&err, static_cast<uint32_t *>(c->mailbox) + 128, sizeof(err), hipMemcpyDeviceToHost));
        if (err != 0)
            return fail(LMCMA_HIP_ERR_RUNTIME, "a kernel gave up waiting for the block of shard " + std::to_string((int)err - 1) +
                                               " (direct exchange); the state is meaningless");
    }
    if (summary) {
        summary->itr = o.itr;
        summary->iterator_sz = o.iterator_sz;
        summary->counteval = o.counteval;
        summary->best_f = o.best_f;
        summary->sigma = o.sigma;
        summary->s = o.s;
        summary->done = o.done;
        summary->reserved = 0;
    }
    return LMCMA_HIP_OK;
}

int lmcma_hip_opt_get_vectors(lmcma_hip_ctx *c, double *pc, double *v_arr, double *pc_arr, double *nj,
                              double *lj, int32_t *iterator)
{
    CTX_TRY(c);
    const size_t N = c->N, m = c->m;
    if (pc) HIP_TRY(hipMemcpyAsync(pc, sel(c, c->pc), N * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (v_arr) HIP_TRY(hipMemcpyAsync(v_arr, sel(c, c->v_arr), m * N * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (pc_arr) HIP_TRY(hipMemcpyAsync(pc_arr, sel(c, c->pc_arr), m * N * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (nj) HIP_TRY(hipMemcpyAsync(nj, sel(c, c->nj), m * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (lj) HIP_TRY(hipMemcpyAsync(lj, sel(c, c->lj), m * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (iterator) HIP_TRY(hipMemcpyAsync(iterator, sel(c, c->iterator), m * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return LMCMA_HIP_OK;
}

int lmcma_hip_buffer(lmcma_hip_ctx *c, int which, void **device_ptr, size_t *bytes)
{
    if (!c || !device_ptr) return fail(LMCMA_HIP_ERR_ARG, "null argument");
    void *p = nullptr;
    size_t b = 0;
    switch (which) {
        case LMCMA_BUF_ROWS_LOCAL: p = c->rows; b = (size_t)c->nloc * c->ld * sizeof(float); break;
        case LMCMA_BUF_COST_LOCAL: p = c->cost_all + c->first; b = (size_t)c->nloc * sizeof(float); break;
        case LMCMA_BUF_COST_ALL: p = c->cost_all; b = (size_t)c->lambda * sizeof(float); break;
        case LMCMA_BUF_PARTIAL_LOCAL: p = c->partial_local; b = (size_t)(c->N + 2) * sizeof(double); break;
        case LMCMA_BUF_PARTIAL_ALL: p = c->partial_all; b = (size_t)kMaxPartials * (c->N + 2) * sizeof(double); break;
        case LMCMA_BUF_XMEAN: p = c->xmean; b = (size_t)c->N * sizeof(double); break;
        case LMCMA_BUF_SORTED_COST: p = c->sorted_cost; b = (size_t)c->lambda * sizeof(float); break;
        case LMCMA_BUF_PERM: p = c->perm; b = (size_t)c->lambda * sizeof(int32_t); break;
        default: return fail(LMCMA_HIP_ERR_ARG, "unknown buffer id");
    }
    *device_ptr = sel(c, static_cast<char *>(p));
    if (bytes) *bytes = b;
    return LMCMA_HIP_OK;
}

int lmcma_hip_row_stride(lmcma_hip_ctx *c) { return c ? c->ld : 0; }

int lmcma_hip_synchronize(lmcma_hip_ctx *c)
{
    CTX_TRY(c);
    // A blocking wait returns some tens of microseconds after the stream has drained: as long as a
    // call of a few generations.  Poll first (about a millisecond), then block.
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t e = hipStreamQuery(c->stream);
        if (e == hipSuccess) return update_gave_up(c);
        if (e != hipErrorNotReady) HIP_TRY(e);
        if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(1)) break;
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    return update_gave_up(c);
}

int lmcma_hip_stream_mark(lmcma_hip_ctx *c, int which)
{
    CTX_TRY(c);
    if (which != 0 && which != 1) return fail(LMCMA_HIP_ERR_ARG, "stream_mark: which is 0 or 1");
    if (!c->mark[which]) HIP_TRY(hipEventCreate(&c->mark[which]));
    HIP_TRY(hipEventRecord(c->mark[which], c->stream));
    return LMCMA_HIP_OK;
}

int lmcma_hip_stream_elapsed(lmcma_hip_ctx *c, double *ms)
{
    CTX_TRY(c);
    if (!ms) return fail(LMCMA_HIP_ERR_ARG, "stream_elapsed: null argument");
    if (!c->mark[0] || !c->mark[1]) return fail(LMCMA_HIP_ERR_STATE, "stream_elapsed: both marks must have been recorded");
    HIP_TRY(hipEventSynchronize(c->mark[1]));
    float f = 0.f;
    HIP_TRY(hipEventElapsedTime(&f, c->mark[0], c->mark[1]));
    *ms = (double)f;
    return LMCMA_HIP_OK;
}

#ifdef LMCMA_DIAG
hipError_t lmcma_diag_update_stamps(unsigned long long *out16);
hipError_t lmcma_diag_rank_stamps(unsigned long long *out16);
hipError_t lmcma_diag_generate_stamps(unsigned long long *out16);
hipError_t lmcma_diag_set_skip_report(int on);
// on != 0: one workgroup of every sampling launch that carries the update withholds its report, so
// the storing workgroup's bounded wait gives up (lmcma_hip_opt_get then returns the error)
int lmcma_hip_diag_skip_report(lmcma_hip_ctx *c, int on)
{
    CTX_TRY(c);
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(lmcma_diag_set_skip_report(on));
    return LMCMA_HIP_OK;
}
// which = 0: stamps of the update kernel, 1: of the ranking kernel, 2: of the generation kernel
int lmcma_hip_diag_stamps(lmcma_hip_ctx *c, int which, unsigned long long *out16)
{
    CTX_TRY(c);
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(which == 2 ? lmcma_diag_generate_stamps(out16)
                       : which ? lmcma_diag_rank_stamps(out16) : lmcma_diag_update_stamps(out16));
    return LMCMA_HIP_OK;
}
#endif

int lmcma_hip_timing_enable(lmcma_hip_ctx *c, int on)
{
    CTX_TRY(c);
    int rc = fold_events(c);
    if (rc != LMCMA_HIP_OK) return rc;
    c->timing = on != 0;
    if (on) {
        for (int i = 0; i < LMCMA_K_COUNT; ++i) { c->t_ms[i] = 0; c->t_n[i] = 0; }
    }
    return LMCMA_HIP_OK;
}

int lmcma_hip_timing_get(lmcma_hip_ctx *c, double ms[LMCMA_K_COUNT], int64_t launches[LMCMA_K_COUNT])
{
    CTX_TRY(c);
    int rc = fold_events(c);
    if (rc != LMCMA_HIP_OK) return rc;
    for (int i = 0; i < LMCMA_K_COUNT; ++i) {
        if (ms) ms[i] = c->t_ms[i];
        if (launches) launches[i] = c->t_n[i];
    }
    return LMCMA_HIP_OK;
}

// What a pair of event records with nothing between them measures on this stream: the
// part of a bracketed kernel's reading that is not the kernel.
int lmcma_hip_timing_event_overhead(lmcma_hip_ctx *c, int pairs, double *ms_per_pair)
{
    CTX_TRY(c);
    if (pairs < 1 || pairs > 4096 || !ms_per_pair) return fail(LMCMA_HIP_ERR_ARG, "timing_event_overhead: pairs in 1..4096");
    std::vector<EventPair> ev((size_t)pairs);
    int rc = LMCMA_HIP_OK;
    size_t made = 0;
    for (; made < ev.size(); ++made) {
        if (hipEventCreate(&ev[made].a) != hipSuccess) break;
        if (hipEventCreate(&ev[made].b) != hipSuccess) { (void)hipEventDestroy(ev[made].a); break; }
    }
    if (made < ev.size()) rc = fail(LMCMA_HIP_ERR_RUNTIME, "timing_event_overhead: hipEventCreate failed");
    double total = 0.0;
    if (rc == LMCMA_HIP_OK) {
        for (auto &p : ev) {
            (void)hipEventRecord(p.a, c->stream);
            (void)hipEventRecord(p.b, c->stream);
        }
        if (hipStreamSynchronize(c->stream) != hipSuccess) rc = fail(LMCMA_HIP_ERR_RUNTIME, "timing_event_overhead: synchronize failed");
        for (auto &p : ev) {
            float ms = 0.f;
            if (rc == LMCMA_HIP_OK && hipEventElapsedTime(&ms, p.a, p.b) != hipSuccess)
                rc = fail(LMCMA_HIP_ERR_RUNTIME, "timing_event_overhead: hipEventElapsedTime failed");
            total += ms;
        }
    }
    for (size_t i = 0; i < made; ++i) { (void)hipEventDestroy(ev[i].a); (void)hipEventDestroy(ev[i].b); }
    if (rc == LMCMA_HIP_OK) *ms_per_pair = total / pairs;
    return rc;
}

}  // extern "C"
