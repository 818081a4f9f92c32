// Fused generation kernel: draw a candidate and score it in one launch.
//
// The sampling half is LMCMA::sample / computeAz / applyBoundaries
// (lmcma_path_planner/src/lmcma.cpp:301-311, 431-447, 220-230), the scoring half the
// trajectory cost described in k_cost.hip; both halves use the same device
// functions as the separate kernels, so a fused launch produces the same bits.
//
// 32 lanes own one candidate through both halves.  In the sampling half lane l owns
// the float4 chunks l, l+32, ... of the row (Philox normals, the k dot products
// reduced over the 32 lanes, the rank-one updates, clamping); the finished row goes
// to global memory (the recombination reads it) and to LDS.  In the scoring half
// lane l owns waypoints l, l+32, ... and reads its +-2 neighbours from the LDS copy,
// so the row never comes back from HBM.  The direction vectors (V, B) are staged in
// LDS once per block when they fit, and a bit-packed occupancy grid can be staged
// beside them; blocks walk several candidate tiles to amortise that staging.
#include <stdlib.h>

#include <algorithm>

#include "update_body.hpp"

namespace lmcma {

namespace {

#ifdef LMCMA_DIAG
// diagnostic build only: wall-clock stamps (100 MHz counter) of the phases of the last workgroup
static __device__ unsigned long long g_generate_stamps[16];
#define GEN_STAMP(i) st[i] = __builtin_amdgcn_s_memrealtime()
#else
#define GEN_STAMP(i)
#endif

constexpr int kLanes = 32;           // lanes per candidate
// direction vectors per pass of the sampling loop: 16 float4 loads in flight per lane
template <int J>
constexpr int kDirectionsPerPass = J == 1 ? 16 : J == 2 ? 8 : 4;
constexpr int kTile = 256 / kLanes;  // candidates per block pass of a 256-thread workgroup
constexpr int kBatchBlocks = 2048;   // workgroups of a launch that covers several queries

// floats of a row in LDS with each of its D dimensions padded by two entries on either side
__host__ __device__ constexpr int padded_row_floats(int D, int W) { return (D * (W + 4) + 3) & ~3; }

// Register tile of the update's first part when it rides in a launch whose rows have J * 128
// floats at most; 0: it does not ride.  Rows of more than 256 parameters do not: with 20
// coordinates per lane the chain of steps, slowed by the sampling workgroups around it, outlasts
// the sampling by more than the shorter second part gains (lambda = 16384, N = 288, m = 32:
// 119 against 114 us per generation with 256-thread sampling workgroups, 105 against 98 with
// 512-thread ones).
__host__ __device__ constexpr int pre_emax_for(int J, int N)
{
    return J == 1 ? 8 : J == 2 ? (N <= 160 ? 10 : N <= 192 ? 12 : 16) : 0;
}

// J = float4 chunks per lane (row stride <= 128 J floats)
// COV: the normals pass through the factor of a structured covariance first (apply_factor)
// PRE: workgroup 0 of the launch does not sample: it runs the first part of this generation's
// update (update_pre_body2: the direction vectors that need nothing of this generation's costs,
// L/lmcma.cpp:375-390 for all but the newest), which reads and writes nothing the sampling or
// ranking kernels touch and is too long to ride in the ranking launch when there are more than
// 16 vectors.  The other workgroups are numbered from 1.
// TB: threads per workgroup, 256 or 512 (TB / 32 candidates per pass).  The LDS copies of 32
// long direction vectors leave room for one or two workgroups per CU; the larger workgroup
// then has twice the waves share one copy.
// FUSED: every workgroup first runs the second part of the PREVIOUS generation's update
// (update_post_body<8, true>, the argument block in `pre`), which leaves the mean, the direction
// vectors and the sampling scalars in LDS instead of sending them through memory and another
// launch; workgroup 0 also stores the state.  512 threads, rows of up to 128 floats, one query.
template <int J, int D, bool GRID_LDS, bool VB_LDS, bool COV, bool PRE, int TB, bool FUSED = false>
__device__ __forceinline__ void generate_body(const GenParams &g, const UpdateArgs &pre)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int kTile = TB / kLanes;
    static_assert(TB == 256 || ((TB == 512) && !COV && !GRID_LDS), "the wide workgroup: plain variants only");
    static_assert(!FUSED || (TB == 512 && J == 1 && VB_LDS && !PRE && !COV && !GRID_LDS), "the fused form: one variant");
    if (PRE && blockIdx.x == 0) {
        if (threadIdx.x >= 256) return;  // the first part of the update is a 256-thread workgroup
        // This workgroup is a chain of dependent steps that shares its CU with sampling workgroups:
        // its waves get the instruction issue ahead of theirs.
        __builtin_amdgcn_s_setprio(3);
        double *s_dyn = reinterpret_cast<double *>(smem);
        const UpdateArgs a = shifted(pre, blockIdx.y);
        if (J == 1) update_pre_body2<8>(a, s_dyn);
        else if (J == 2 && a.N <= 160) update_pre_body2<10>(a, s_dyn);
        else if (J == 2 && a.N <= 192) update_pre_body2<12>(a, s_dyn);
        else if (J == 2) update_pre_body2<16>(a, s_dyn);
#ifdef LMCMA_DIAG
        if (threadIdx.x == 0) {  // this translation unit's copy of the update's stamps, beside the sampling ones
            g_generate_stamps[8] = g_update_stamps[6];
            g_generate_stamps[9] = g_update_stamps[7];
            g_generate_stamps[10] = g_update_stamps[13];
            g_generate_stamps[11] = g_update_stamps[14];
            g_generate_stamps[12] = g_update_stamps[15];
        }
#endif
        return;
    }
    const int block = (int)blockIdx.x - (PRE ? 1 : 0), nblocks = (int)gridDim.x - (PRE ? 1 : 0);
    // query blockIdx.y of a batch: its copies of the context's buffers, its own Philox key.  (Only
    // the pointers are copied: a modified copy of the whole argument block would live in scratch
    // memory, because the covariance factor's block table in it is indexed at run time.)
    const size_t qoff = (size_t)blockIdx.y * g.batch.stride;
    float *q_rows = g.rows, *q_cost = g.cost.cost, *q_terms = g.cost.terms;
    const float *q_xmean = g.xmean, *q_V = g.V, *q_B = g.B, *q_lo = g.lo, *q_hi = g.hi;
    const SampleScalars *q_sc = g.sc;
    shift_ptr(q_rows, qoff); shift_ptr(q_cost, qoff); shift_ptr(q_terms, qoff); shift_ptr(q_xmean, qoff);
    shift_ptr(q_V, qoff); shift_ptr(q_B, qoff); shift_ptr(q_lo, qoff); shift_ptr(q_hi, qoff); shift_ptr(q_sc, qoff);
    const uint32_t q_key0 = g.key0 + blockIdx.y;
    const CostParams &p = g.cost;
    const int ld = g.ld;
    const int nchunks = ld >> 2;
    // (the fused form gets these from the update it runs first, and must not wait for memory here)
    int k = 0;
    float sigma = 0.f, mk = 0.f;
    uint32_t generation = 0u;
    if (!FUSED) {
        k = q_sc->k;
        sigma = q_sc->sigma;
        mk = q_sc->mk;
        generation = q_sc->generation;
    }

#ifdef LMCMA_DIAG
    unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    GEN_STAMP(0);
    constexpr int LDP = J * 128;               // LDS stride of a direction vector: reads need no bounds
    constexpr int U = kDirectionsPerPass<J>;   // direction vectors per pass of the sampling loop
    const int mp = (g.m + U - 1) / U * U;      // rows of the LDS copies: whole passes, zero beyond k
    // The tile's rows for the scoring half, every dimension padded with the fixed end points:
    // dimension d of a row at [d (W + 4) + 2, + W), two copies of start[d] before and of goal[d]
    // after it (waypoint_prepare_padded)
    const int LP = padded_row_floats(D, p.W);
    float *srows = smem;                       // [kTile][LP]
    float *sV = srows + kTile * LP;            // [mp][LDP] when VB_LDS, zero beyond ld and beyond row k
    float *sB = sV + (VB_LDS ? mp * LDP : 0);  // [mp][LDP] when VB_LDS, likewise
    float *s_z = sB + (VB_LDS ? mp * LDP : 0);  // [16][16 ntiles + 4] when COV
    uint32_t *sbits = reinterpret_cast<uint32_t *>(s_z + (COV ? factor_lds_floats(g.cov.ntiles) : 0));
    // FUSED: [LDP] mean, [4] scalars, then the update's own LDS (doubles)
    float *s_fx = s_z;
    double *s_upd = reinterpret_cast<double *>(s_fx + LDP + 4);

    const int lane = threadIdx.x & (kLanes - 1);
    const int slot = threadIdx.x / kLanes;
    // the bounds of this lane's chunks (used after the sampling loop): requested first
    float4 xm[J], l4[J], h4[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int chunk = j * kLanes + lane;
        const int cc = chunk < nchunks ? chunk : nchunks - 1;
        l4[j] = reinterpret_cast<const float4 *>(q_lo)[cc];
        h4[j] = reinterpret_cast<const float4 *>(q_hi)[cc];
    }
    // FUSED: the normals of this workgroup's first tile, drawn while the update's partial rows arrive
    float4 z_first[J];
    bool from_update = false;
    if (FUSED) {
        for (int i = threadIdx.x; i < 2 * mp * (LDP >> 2); i += TB)
            reinterpret_cast<float4 *>(sV)[i] = make_float4(0.f, 0.f, 0.f, 0.f);  // (sB follows sV)
        FusedOut fo{sV, sB, s_fx, s_fx + LDP, LDP, blockIdx.x == 0, g.fused_counter, gridDim.x - 1};
        const int c0 = (int)blockIdx.x * kTile + slot;
        const uint32_t cand0 = g.first_candidate + (uint32_t)(c0 < g.n_local ? c0 : g.n_local - 1);
        auto draw = [&](uint32_t gen) {
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const int chunk = j * kLanes + lane;
                z_first[j] = (chunk < nchunks) ? philox_normals((uint32_t)chunk, cand0, gen, q_key0, g.key1)
                                               : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        };
        from_update = update_post_body<8, true>(pre, s_upd, 256, TB, &fo, draw);
        __syncthreads();
#ifdef LMCMA_DIAG
        if (threadIdx.x == 0 && blockIdx.x == gridDim.x - 1)
            for (int q = 0; q < 8; ++q) g_generate_stamps[8 + q] = g_update_stamps[q];
#endif
        if (from_update) {
            sigma = s_fx[LDP];
            mk = s_fx[LDP + 1];
            k = __float_as_int(s_fx[LDP + 2]);
            generation = __float_as_uint(s_fx[LDP + 3]);
        } else {  // the stop rule has fired: the state in memory stays, sample from it
            k = q_sc->k;
            sigma = q_sc->sigma;
            mk = q_sc->mk;
            generation = q_sc->generation;
        }
    }
    // The update kernel wrote the mean and V, B, so their first touch misses this XCD's
    // L2: every such load of the kernel is requested here, before anything waits.
    // mean and bounds of this lane's chunks (used after the sampling loop)
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int chunk = j * kLanes + lane;
        const int cc = chunk < nchunks ? chunk : nchunks - 1;
        xm[j] = (FUSED && from_update) ? reinterpret_cast<const float4 *>(s_fx)[cc] : reinterpret_cast<const float4 *>(q_xmean)[cc];
    }
    if (VB_LDS && !(FUSED && from_update)) {
        // All rows, whatever k is: the loads then do not wait for the scalar load of k (rows
        // from k on are stored as zeros: the sampling loop below runs in whole passes).
        const int n4 = mp * (LDP >> 2);
        const float4 *srcV = reinterpret_cast<const float4 *>(q_V);
        const float4 *srcB = reinterpret_cast<const float4 *>(q_B);
        float4 *dstV = reinterpret_cast<float4 *>(sV), *dstB = reinterpret_cast<float4 *>(sB);
        // Only the rows' own chunks are read (m * nchunks float4 per matrix), in batches of SB per
        // thread and matrix that are all requested before the first LDS store: the update wrote
        // these rows on another XCD, so every batch is a round trip to memory, and 32 vectors of up
        // to 256 floats go in one.  The padding (beyond the row, and rows from m on) is zeroed
        // without a load.
        constexpr int SB = J == 1 ? 4 : TB == 512 ? 4 : 8;
        const int nreal = g.m * nchunks;
        const bool dense = nchunks == (LDP >> 2);
        for (int i0 = threadIdx.x; i0 < nreal; i0 += SB * TB) {
            float4 tv[SB], tb[SB];
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                const int i = i0 + u * TB;
                const int at = i < nreal ? i : nreal - 1;
                tv[u] = srcV[at];
                tb[u] = srcB[at];
            }
            asm volatile("" ::: "memory");  // the loads of a batch are all issued before the first LDS store
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                const int i = i0 + u * TB;
                if (i < nreal) {
                    int row = i / (LDP >> 2), at = i;   // rows that fill their LDS stride
                    if (!dense) {
                        row = i / nchunks;
                        at = row * (LDP >> 2) + (i - row * nchunks);
                    }
                    const bool pad = row >= k;   // vectors not in use yet
                    dstV[at] = pad ? make_float4(0.f, 0.f, 0.f, 0.f) : tv[u];
                    dstB[at] = pad ? make_float4(0.f, 0.f, 0.f, 0.f) : tb[u];
                }
            }
        }
        if (!dense || g.m < mp) {
            for (int i = threadIdx.x; i < n4; i += TB) {
                if (i % (LDP >> 2) >= nchunks || i / (LDP >> 2) >= g.m) {
                    dstV[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                    dstB[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
        }
    }
    // where this lane's elements go in a padded row (-1: beyond the row), and the pads themselves
    int pos[J][4];
#pragma unroll
    for (int j = 0; j < J; ++j) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int n = 4 * (j * kLanes + lane) + e;
            const int d = (D > 1 && n >= p.W ? 1 : 0) + (D > 2 && n >= 2 * p.W ? 1 : 0);  // n / W without a division
            pos[j][e] = n < D * p.W ? n + 4 * d + 2 : -1;
        }
    }
    if (threadIdx.x < kTile * D * 4) {
        const int e = threadIdx.x & 3, d = (threadIdx.x >> 2) % D, sl = threadIdx.x / (4 * D);
        srows[sl * LP + d * (p.W + 4) + (e < 2 ? e : p.W + e)] = e < 2 ? p.start[d] : p.goal[d];
    }
    const uint32_t *bits;
    if (GRID_LDS) {
        const uint4 *src = reinterpret_cast<const uint4 *>(p.grid);
        uint4 *dst = reinterpret_cast<uint4 *>(sbits);
        for (int i = threadIdx.x; i < (p.grid_words >> 2); i += TB) dst[i] = src[i];
        bits = sbits;
    } else {
        bits = reinterpret_cast<const uint32_t *>(p.grid);
    }
    const float *Vp = VB_LDS ? sV : q_V;
    const float *Bp = VB_LDS ? sB : q_B;

    const int ntiles = (g.n_local + kTile - 1) / kTile;

    for (int tile = block; tile < ntiles; tile += nblocks) {
        const int cand = tile * kTile + slot;
        const bool live = cand < g.n_local;
        const int c = live ? cand : g.n_local - 1;  // idle groups shadow the last row (shuffles need all lanes)

        // ---- sample --------------------------------------------------------------------
        float4 z[J], acc[J];
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int chunk = j * kLanes + lane;
            if (FUSED && from_update && tile == block) {
                z[j] = z_first[j];
                continue;
            }
            z[j] = (chunk < nchunks) ? philox_normals((uint32_t)chunk, g.first_candidate + (uint32_t)c,
                                                      generation, q_key0, g.key1)
                                     : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        // (the scoring half of the previous tile does not touch the factor's LDS, and two
        // barriers separate it from this call)
        if (COV) apply_factor<J, kLanes>(z, g.cov, s_z);
#pragma unroll
        for (int j = 0; j < J; ++j) acc[j] = make_float4(mk * z[j].x, mk * z[j].y, mk * z[j].z, mk * z[j].w);
        GEN_STAMP(1);
        __syncthreads();  // staging done (first pass) / previous tile's rows consumed (later passes)
        GEN_STAMP(2);

        // U direction vectors per pass: all dot products are with z itself, so they and their
        // lane reductions are independent and the cross-lane latency of one hides behind the
        // others; the rank-one updates are still applied in the order t = 0, 1, 2, ...  With
        // the LDS copies (zero rows from k on) a pass has no branch in it.
        for (int t0 = 0; t0 < k; t0 += U) {
            float d[U];
            if (VB_LDS) {
                // all reads of the pass are requested before the first product (left to itself the
                // scheduler reuses one register quad and waits for every read in turn)
                float4 vv[U][J];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const float4 *v = reinterpret_cast<const float4 *>(Vp + (size_t)(t0 + u) * LDP);
#pragma unroll
                    for (int j = 0; j < J; ++j) vv[u][j] = v[j * kLanes + lane];
                }
                asm volatile("" ::: "memory");
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    d[u] = 0.f;
#pragma unroll
                    for (int j = 0; j < J; ++j) {
                        d[u] = fmaf(vv[u][j].x, z[j].x, d[u]);
                        d[u] = fmaf(vv[u][j].y, z[j].y, d[u]);
                        d[u] = fmaf(vv[u][j].z, z[j].z, d[u]);
                        d[u] = fmaf(vv[u][j].w, z[j].w, d[u]);
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const float4 *b = reinterpret_cast<const float4 *>(Bp + (size_t)(t0 + u) * LDP);
#pragma unroll
                    for (int j = 0; j < J; ++j) vv[u][j] = b[j * kLanes + lane];
                }
                asm volatile("" ::: "memory");
#pragma unroll
                for (int u = 0; u < U; ++u) d[u] = group_sum<kLanes>(d[u]);
#pragma unroll
                for (int u = 0; u < U; ++u) {
#pragma unroll
                    for (int j = 0; j < J; ++j) {
                        acc[j].x = fmaf(d[u], vv[u][j].x, acc[j].x);
                        acc[j].y = fmaf(d[u], vv[u][j].y, acc[j].y);
                        acc[j].z = fmaf(d[u], vv[u][j].z, acc[j].z);
                        acc[j].w = fmaf(d[u], vv[u][j].w, acc[j].w);
                    }
                }
                continue;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                d[u] = 0.f;
                if (t0 + u < k) {
                    const float4 *v = reinterpret_cast<const float4 *>(Vp + (size_t)(t0 + u) * ld);
#pragma unroll
                    for (int j = 0; j < J; ++j) {
                        const int chunk = j * kLanes + lane;
                        if (chunk < nchunks) {
                            const float4 vv = v[chunk];
                            d[u] = fmaf(vv.x, z[j].x, d[u]);
                            d[u] = fmaf(vv.y, z[j].y, d[u]);
                            d[u] = fmaf(vv.z, z[j].z, d[u]);
                            d[u] = fmaf(vv.w, z[j].w, d[u]);
                        }
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) d[u] = group_sum<kLanes>(d[u]);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (t0 + u < k) {
                    const float4 *b = reinterpret_cast<const float4 *>(Bp + (size_t)(t0 + u) * ld);
#pragma unroll
                    for (int j = 0; j < J; ++j) {
                        const int chunk = j * kLanes + lane;
                        if (chunk < nchunks) {
                            const float4 bb = b[chunk];
                            acc[j].x = fmaf(d[u], bb.x, acc[j].x);
                            acc[j].y = fmaf(d[u], bb.y, acc[j].y);
                            acc[j].z = fmaf(d[u], bb.z, acc[j].z);
                            acc[j].w = fmaf(d[u], bb.w, acc[j].w);
                        }
                    }
                }
            }
        }

        GEN_STAMP(3);
        float4 *out = reinterpret_cast<float4 *>(q_rows + (size_t)c * ld);
        float *lrow = srows + slot * LP;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int chunk = j * kLanes + lane;
            if (chunk < nchunks) {
                float4 x;
                // max with the lower bound first, then min with the upper one (lmcma.cpp:222-229)
                x.x = fminf(fmaxf(fmaf(sigma, acc[j].x, xm[j].x), l4[j].x), h4[j].x);
                x.y = fminf(fmaxf(fmaf(sigma, acc[j].y, xm[j].y), l4[j].y), h4[j].y);
                x.z = fminf(fmaxf(fmaf(sigma, acc[j].z, xm[j].z), l4[j].z), h4[j].z);
                x.w = fminf(fmaxf(fmaf(sigma, acc[j].w, xm[j].w), l4[j].w), h4[j].w);
                if (pos[j][0] >= 0) lrow[pos[j][0]] = x.x;
                if (pos[j][1] >= 0) lrow[pos[j][1]] = x.y;
                if (pos[j][2] >= 0) lrow[pos[j][2]] = x.z;
                if (pos[j][3] >= 0) lrow[pos[j][3]] = x.w;
                if (live) out[chunk] = x;
            }
        }
        __syncthreads();  // the tile's rows are complete in LDS
        GEN_STAMP(4);

        // ---- score ---------------------------------------------------------------------
        float j_len = 0.f, j_grid = 0.f, j_smooth = 0.f;
        const float *row = srows + slot * LP;
        // two waypoints per pass: both grid lookups (the only global loads of this half, at
        // data-dependent addresses) are in flight together; the terms are added in waypoint order
        for (int w0 = lane; w0 < p.W; w0 += 2 * kLanes) {
            const int w1 = w0 + kLanes;
            const bool has1 = w1 < p.W;
            const WaypointPart a = waypoint_prepare_padded<D>(p, row, w0);
            const WaypointPart b = waypoint_prepare_padded<D>(p, row, has1 ? w1 : w0);
            const float ca = grid_value_at(p, bits, a.cell);
            const float cb = grid_value_at(p, bits, b.cell);
            waypoint_finish(p, a, w0, ca, j_len, j_grid, j_smooth);
            if (has1) waypoint_finish(p, b, w1, cb, j_len, j_grid, j_smooth);
        }
        j_len = group_sum<kLanes>(j_len);
        j_grid = group_sum<kLanes>(j_grid);
        j_smooth = group_sum<kLanes>(j_smooth);
        GEN_STAMP(5);
        if (live && lane == 0) {
            q_cost[cand] = total_cost(p, j_len, j_grid, j_smooth);
            if (q_terms) {
                float *t = q_terms + 3 * (size_t)cand;
                t[0] = j_len; t[1] = j_grid; t[2] = j_smooth;
            }
        }
    }
#ifdef LMCMA_DIAG
    st[6] = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && block == nblocks - 1)
        for (int q = 0; q < 7; ++q) g_generate_stamps[q] = st[q];
#endif
}

#undef GEN_STAMP

// two entry points, so that a launch without the update's first part does not carry its arguments
template <int J, int D, bool GRID_LDS, bool VB_LDS, bool COV, int TB>
__global__ __launch_bounds__(TB) void k_generate(const GenParams g)
{
    generate_body<J, D, GRID_LDS, VB_LDS, COV, false, TB>(g, UpdateArgs{});
}

template <int J, int D, bool VB_LDS, int TB>
__global__ __launch_bounds__(TB) void k_generate_pre(const GenParams g, const UpdateArgs pre)
{
    generate_body<J, D, false, VB_LDS, false, true, TB>(g, pre);
}

template <int D>
__global__ __launch_bounds__(512) void k_generate_fused(const GenParams g, const UpdateArgs post)
{
    generate_body<1, D, false, true, false, false, 512, true>(g, post);
}

// ---- many queries per launch: the two products on the matrix cores ---------------------------
// With one query the sampling kernel waits for memory most of its time; with many queries in
// one launch it is bound by vector-instruction issue, and over half of its vector instructions
// are the two products of computeAz (L/lmcma.cpp:431-447 in closed form): D = V z (m dot
// products per candidate, reduced over lanes) and x = M^k z + sum_t d_t B_t.  Over a tile of
// candidates they are two small dense matrix products, and here they run as
// v_mfma_f32_16x16x4_f32 (fp32 in, fp32 accumulate: no precision given up, only the order of
// the sums changes):
//   a wave owns 16 candidates; lane (c = l % 16, q = l / 16) draws the chunks q, q + 4, ..., q + 28
//   of candidate c's normals (the same Philox counters as everywhere else);
//   D^T [t][c] = sum_n V[t][n] z[c][n]: A = V from LDS, B = the lane's own normals, 32 steps;
//   X^T [n][c] = M^k z[c][n] + sum_t B[t][n] D^T[t][c]: per 16 rows n, C starts as M^k z, A = B
//   from LDS, B = D^T as the first product left it in registers (K index t = 4 (l / 16) + step),
//   and the result lands chunk for chunk in the lane that drew that chunk.
// The scoring half is the one of generate_body, 32 lanes per candidate, eight candidates per pass.
// Rows of up to 128 floats, up to 16 direction vectors (one 16-row tile of D^T; with more vectors
// the first part of the update rides in the sampling launch, which this kernel does not carry).
constexpr int kMfmaLD = 132;    // LDS stride of a direction vector: 16-byte aligned, rows four banks apart
constexpr int kMfmaTile = 64;   // candidates per workgroup pass (four waves)
typedef float v4f __attribute__((ext_vector_type(4)));

size_t generate_mfma_lds_bytes(int D, int W, int m)
{
    const int mp = 16;
    (void)m;
    return ((size_t)kMfmaTile * padded_row_floats(D, W) + 2 * (size_t)mp * kMfmaLD + 3 * 128) * sizeof(float);
}

template <int D>
__global__ __launch_bounds__(256) void k_generate_mfma(const GenParams g)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int block = (int)blockIdx.x, nblocks = (int)gridDim.x;
    const size_t qoff = (size_t)blockIdx.y * g.batch.stride;
    float *q_rows = g.rows, *q_cost = g.cost.cost, *q_terms = g.cost.terms;
    const float *q_xmean = g.xmean, *q_V = g.V, *q_B = g.B, *q_lo = g.lo, *q_hi = g.hi;
    const SampleScalars *q_sc = g.sc;
    shift_ptr(q_rows, qoff); shift_ptr(q_cost, qoff); shift_ptr(q_terms, qoff); shift_ptr(q_xmean, qoff);
    shift_ptr(q_V, qoff); shift_ptr(q_B, qoff); shift_ptr(q_lo, qoff); shift_ptr(q_hi, qoff); shift_ptr(q_sc, qoff);
    const uint32_t q_key0 = g.key0 + blockIdx.y;
    const CostParams &p = g.cost;
    const int ld = g.ld, nchunks = ld >> 2;
    const int k = q_sc->k;
    const float sigma = q_sc->sigma, mk = q_sc->mk;
    const uint32_t generation = q_sc->generation;
    constexpr int mp = 16;
    const int LP = padded_row_floats(D, p.W);
    float *srows = smem;                        // [64][LP]
    float *sV = srows + kMfmaTile * LP;         // [mp][132], zero beyond the row and from vector k on
    float *sB = sV + mp * kMfmaLD;              // [mp][132]
    float *s_mean = sB + mp * kMfmaLD;          // [128] mean, [128] lower, [128] upper bounds
    const int tid = threadIdx.x, l = tid & 63, wave = tid >> 6, c16 = l & 15, q = l >> 4;

    // ---- staging (every load requested before the first LDS store)
    {
        float4 mv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tid < 96) {
            const int which = tid >> 5, chunk = tid & 31;
            const float *src = which == 0 ? q_xmean : which == 1 ? q_lo : q_hi;
            mv = reinterpret_cast<const float4 *>(src)[chunk < nchunks ? chunk : nchunks - 1];
            if (chunk >= nchunks) mv = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        float4 tv[2], tb[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = tid + 256 * u, row = i >> 5, chunk = i & 31;
            const bool real = row < g.m && chunk < nchunks;
            const int at = real ? row * nchunks + chunk : 0;
            tv[u] = reinterpret_cast<const float4 *>(q_V)[at];
            tb[u] = reinterpret_cast<const float4 *>(q_B)[at];
        }
        asm volatile("" ::: "memory");
        if (tid < 96) reinterpret_cast<float4 *>(s_mean)[tid] = mv;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = tid + 256 * u, row = i >> 5, chunk = i & 31;  // 16 rows x 32 chunks = 512
            const bool real = row < g.m && row < k && chunk < nchunks;
            const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4 *>(sV + row * kMfmaLD + 4 * chunk) = real ? tv[u] : zero;
            *reinterpret_cast<float4 *>(sB + row * kMfmaLD + 4 * chunk) = real ? tb[u] : zero;
        }
        // the fixed end points around every dimension of every row of the tile
        for (int i = tid; i < kMfmaTile * D * 4; i += 256) {
            const int e = i & 3, d = (i >> 2) % D, sl = i / (4 * D);
            srows[sl * LP + d * (p.W + 4) + (e < 2 ? e : p.W + e)] = e < 2 ? p.start[d] : p.goal[d];
        }
    }
    const uint32_t *bits = reinterpret_cast<const uint32_t *>(p.grid);
    const int ntiles = (g.n_local + kMfmaTile - 1) / kMfmaTile;

    for (int tile = block; tile < ntiles; tile += nblocks) {
        const int slot = wave * 16 + c16;
        const int cand = tile * kMfmaTile + slot;
        const bool live = cand < g.n_local;
        const int c = live ? cand : g.n_local - 1;

        // ---- the lane's normals: chunks q, q + 4, ..., q + 28 of candidate c
        float4 z[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int chunk = q + 4 * j;
            z[j] = chunk < nchunks ? philox_normals((uint32_t)chunk, g.first_candidate + (uint32_t)c, generation, q_key0, g.key1)
                                   : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __syncthreads();  // staging done (first pass) / the previous tile's rows have been scored

        // ---- D^T = V Z^T
        v4f dT0 = {0.f, 0.f, 0.f, 0.f};
        if (k > 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float4 va = *reinterpret_cast<const float4 *>(sV + c16 * kMfmaLD + 4 * (q + 4 * j));
                dT0 = __builtin_amdgcn_mfma_f32_16x16x4f32(va.x, z[j].x, dT0, 0, 0, 0);
                dT0 = __builtin_amdgcn_mfma_f32_16x16x4f32(va.y, z[j].y, dT0, 0, 0, 0);
                dT0 = __builtin_amdgcn_mfma_f32_16x16x4f32(va.z, z[j].z, dT0, 0, 0, 0);
                dT0 = __builtin_amdgcn_mfma_f32_16x16x4f32(va.w, z[j].w, dT0, 0, 0, 0);
            }
        }

        // ---- X^T = M^k Z^T + B^T D^T, sixteen rows (four chunks of the row) at a time; clamp; out
        float *lrow = srows + slot * LP;
        float4 *out = reinterpret_cast<float4 *>(q_rows + (size_t)c * ld);
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) {
            v4f x = {mk * z[nt].x, mk * z[nt].y, mk * z[nt].z, mk * z[nt].w};
            if (k > 0) {
                const float *b0 = sB + (4 * q) * kMfmaLD + 16 * nt + c16;
                x = __builtin_amdgcn_mfma_f32_16x16x4f32(b0[0], dT0[0], x, 0, 0, 0);
                x = __builtin_amdgcn_mfma_f32_16x16x4f32(b0[kMfmaLD], dT0[1], x, 0, 0, 0);
                x = __builtin_amdgcn_mfma_f32_16x16x4f32(b0[2 * kMfmaLD], dT0[2], x, 0, 0, 0);
                x = __builtin_amdgcn_mfma_f32_16x16x4f32(b0[3 * kMfmaLD], dT0[3], x, 0, 0, 0);
            }
            const int chunk = 4 * nt + q;
            if (chunk < nchunks) {
                const float4 xm = reinterpret_cast<const float4 *>(s_mean)[chunk];
                const float4 lo = reinterpret_cast<const float4 *>(s_mean + 128)[chunk];
                const float4 hi = reinterpret_cast<const float4 *>(s_mean + 256)[chunk];
                float4 r;
                // max with the lower bound first, then min with the upper one (lmcma.cpp:222-229)
                r.x = fminf(fmaxf(fmaf(sigma, x[0], xm.x), lo.x), hi.x);
                r.y = fminf(fmaxf(fmaf(sigma, x[1], xm.y), lo.y), hi.y);
                r.z = fminf(fmaxf(fmaf(sigma, x[2], xm.z), lo.z), hi.z);
                r.w = fminf(fmaxf(fmaf(sigma, x[3], xm.w), lo.w), hi.w);
                const float re[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int n = 4 * chunk + e;
                    const int d = (D > 1 && n >= p.W ? 1 : 0) + (D > 2 && n >= 2 * p.W ? 1 : 0);
                    if (n < D * p.W) lrow[n + 4 * d + 2] = re[e];
                }
                if (live) out[chunk] = r;
            }
        }
        __syncthreads();  // the tile's rows are complete in LDS

        // ---- score: 32 lanes per candidate, eight candidates per pass
        const int lane = tid & 31;
        for (int pass = 0; pass < kMfmaTile / 8; ++pass) {
            const int sl = pass * 8 + (tid >> 5), sc_cand = tile * kMfmaTile + sl;
            float j_len = 0.f, j_grid = 0.f, j_smooth = 0.f;
            const float *row = srows + sl * LP;
            for (int w0 = lane; w0 < p.W; w0 += 2 * kLanes) {
                const int w1 = w0 + kLanes;
                const bool has1 = w1 < p.W;
                const WaypointPart a = waypoint_prepare_padded<D>(p, row, w0);
                const WaypointPart b = waypoint_prepare_padded<D>(p, row, has1 ? w1 : w0);
                const float ca = grid_value_at(p, bits, a.cell);
                const float cb = grid_value_at(p, bits, b.cell);
                waypoint_finish(p, a, w0, ca, j_len, j_grid, j_smooth);
                if (has1) waypoint_finish(p, b, w1, cb, j_len, j_grid, j_smooth);
            }
            j_len = group_sum<kLanes>(j_len);
            j_grid = group_sum<kLanes>(j_grid);
            j_smooth = group_sum<kLanes>(j_smooth);
            if (sc_cand < g.n_local && lane == 0) {
                q_cost[sc_cand] = total_cost(p, j_len, j_grid, j_smooth);
                if (q_terms) {
                    float *t = q_terms + 3 * (size_t)sc_cand;
                    t[0] = j_len; t[1] = j_grid; t[2] = j_smooth;
                }
            }
        }
    }
}

template <int J, int D, bool GRID_LDS, bool VB_LDS, bool COV, bool PRE, int TB = 256>
hipError_t launch_t(hipStream_t s, const GenParams &g, size_t lds, int blocks, const UpdateArgs *pre)
{
    if constexpr (PRE) {
        lds = std::max(lds, update_pre_lds_doubles(pre->N, pre->m) * sizeof(double));
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(&k_generate_pre<J, D, VB_LDS, TB>), lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((k_generate_pre<J, D, VB_LDS, TB>), dim3(blocks + 1, g.batch.queries), dim3(TB), lds, s, g, *pre);
    } else {
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(&k_generate<J, D, GRID_LDS, VB_LDS, COV, TB>), lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((k_generate<J, D, GRID_LDS, VB_LDS, COV, TB>), dim3(blocks, g.batch.queries), dim3(TB), lds, s, g);
    }
    return hipGetLastError();
}

template <int J, int D>
hipError_t launch_jd(hipStream_t s, const GenParams &g, bool grid_lds, const UpdateArgs *pre, bool *pre_done)
{
    const size_t row_b = (size_t)padded_row_floats(D, g.cost.W) * sizeof(float);
    size_t rows_b = kTile * row_b;
    constexpr int U = kDirectionsPerPass<J>;
    const size_t vb_b = 2 * (size_t)((g.m + U - 1) / U * U) * (J * 128) * sizeof(float);  // whole passes of padded rows
    const bool cov = g.cov.L != nullptr;
    if (cov) grid_lds = false;  // with a factor the grid is read through L2: half as many variants of the kernel
    const size_t cov_b = cov ? factor_lds_floats(g.cov.ntiles) * sizeof(float) : 0;
    const size_t grid_b = grid_lds ? (size_t)g.cost.grid_words * sizeof(uint32_t) : 0;
    const size_t budget = 160 * 1024;
    const bool vb_lds = rows_b + vb_b + grid_b + cov_b <= budget && g.m <= 32;
    if (grid_lds && rows_b + (vb_lds ? vb_b : 0) + grid_b > budget) grid_lds = false;
    // Workgroups of 512 threads (16 candidates per pass) when the staged vectors leave room for
    // one 256-thread workgroup per CU only: twice the waves then share one LDS copy (lambda =
    // 16384, N = 288, m = 32 with a 64 MB clearance map: 115 -> 101 us per generation; where two
    // 256-thread workgroups fit, the wide one gains nothing: N = 144 and 256 measured).
    bool wide = J >= 2 && vb_lds && !cov && !grid_lds && 2 * kTile * row_b + vb_b <= budget &&
                2 * (rows_b + vb_b) > budget;
    // The update's first part rides as one more workgroup (below).  Where the 256-thread workgroups
    // fill the machine exactly (lambda = 8192, N = 144, m = 32: 70 KB of LDS each, two per CU, 512 of
    // them), that extra workgroup displaces one of them into a second round, and the launch takes
    // 21.8 us instead of 16.5 however short the first part is (halving its chain of steps left the
    // launch at 21.8 us).  The 512-thread form has half as many workgroups: if they and the extra one
    // are all resident at once, it is the better shape (cfg3: 47.8 -> 45.5 us per generation; at
    // N = 256 a 512-thread workgroup needs 81 KB, one per CU, and nothing is gained: 54.6 against 50.9).
    static const bool no_pre_switch = getenv("LMCMA_HIP_UPDATE_ONE_KERNEL") != nullptr;
    const bool pre_rides = pre && !no_pre_switch && !cov && !grid_lds && pre->m <= 32 && pre_emax_for(J, pre->N) > 0 &&
                           update_emax(pre->N) <= pre_emax_for(J, pre->N) && update_is_split(pre->N, pre->m) &&
                           !update_pre_rides_with_ranking(pre->m);
    if (pre_rides && !wide && J >= 2 && vb_lds && g.batch.queries == 1) {
        const size_t lds_narrow = rows_b + vb_b, lds_wide = 2 * kTile * row_b + vb_b;
        const int blocks_narrow = std::min((g.n_local + kTile - 1) / kTile, 512);
        const int blocks_wide = std::min((g.n_local + 2 * kTile - 1) / (2 * kTile), 256);
        const int slots_narrow = 256 * (int)std::min<size_t>(budget / lds_narrow, 8);   // (256 threads: 8 workgroups of waves per CU)
        const int slots_wide = lds_wide <= budget ? 256 * (int)std::min<size_t>(budget / lds_wide, 4) : 0;
        if (blocks_narrow + 1 > slots_narrow && blocks_wide + 1 <= slots_wide) wide = true;
    }
    // (measurement switches are read once per process, not per launch)
    static const int forced_threads = getenv("LMCMA_HIP_GENERATE_THREADS") ? atoi(getenv("LMCMA_HIP_GENERATE_THREADS")) : 0;
    static const int forced_batch_blocks = getenv("LMCMA_HIP_BATCH_BLOCKS") ? atoi(getenv("LMCMA_HIP_BATCH_BLOCKS")) : 0;
    if (forced_threads != 0)
        wide = J >= 2 && vb_lds && !cov && !grid_lds && 2 * kTile * row_b + vb_b <= budget && forced_threads == 512;
    const int tile = wide ? 2 * kTile : kTile;
    if (wide) rows_b = tile * row_b;
    const int ntiles = (g.n_local + tile - 1) / tile;
    const size_t lds = rows_b + (vb_lds ? vb_b : 0) + (grid_lds ? grid_b : 0) + cov_b;
    // staging is paid per block: one block per CU when something is staged and there
    // are enough tiles, otherwise one block per tile up to eight blocks per CU
    int blocks = (grid_lds || vb_lds) ? (ntiles < 512 ? ntiles : (grid_lds ? 256 : 512))
                                      : (ntiles < 2048 ? ntiles : 2048);
    if (wide) blocks = std::min(ntiles, 256);  // one per CU at a time: no second round of staging (101 -> 98 us)
    // several queries: the launch has that many times the workgroups, so each query gets fewer, and a
    // workgroup's staged vectors serve several tiles (about eight workgroups per CU over the launch)
    if (g.batch.queries > 1) {
        int per_query = kBatchBlocks / g.batch.queries;
        if (forced_batch_blocks != 0) per_query = forced_batch_blocks / g.batch.queries;
        blocks = std::min(blocks, std::max(per_query, 1));
    }
    // Many queries in one launch: the products on the matrix cores (k_generate_mfma), when nothing
    // else has to ride in this launch.  From 16 queries on (lambda = 4096: no gain with 12, a loss
    // with 8 and fewer, where the launch still waits for memory rather than for instruction issue).
    static const int mfma_from = getenv("LMCMA_HIP_MFMA_FROM") ? atoi(getenv("LMCMA_HIP_MFMA_FROM")) : 16;
    if (J == 1 && g.batch.queries >= mfma_from && !cov && g.m <= 16) {
        const size_t mlds = generate_mfma_lds_bytes(D, g.cost.W, g.m);
        if (mlds <= budget) {
            const int mtiles = (g.n_local + kMfmaTile - 1) / kMfmaTile;
            int per_query = kBatchBlocks / g.batch.queries;
            if (forced_batch_blocks != 0) per_query = forced_batch_blocks / g.batch.queries;
            const int mblocks = std::min(mtiles, std::max(per_query, 1));
            hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(&k_generate_mfma<D>), mlds);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL((k_generate_mfma<D>), dim3(mblocks, g.batch.queries), dim3(256), mlds, s, g);
            return hipGetLastError();
        }
    }
    if (cov) {
        if (vb_lds) return launch_t<J, D, false, true, true, false>(s, g, lds, blocks, nullptr);
        return launch_t<J, D, false, false, true, false>(s, g, lds, blocks, nullptr);
    }
    if (grid_lds) {
        if (vb_lds) return launch_t<J, D, true, true, false, false>(s, g, lds, blocks, nullptr);
        return launch_t<J, D, true, false, false, false>(s, g, lds, blocks, nullptr);
    }
    // the first part of the update as one more workgroup (see k_generate): stores of 17 to 32
    // vectors whose register tile exists for this row length
    if (pre_rides) {
        if (pre_done) *pre_done = true;
        if (J >= 2 && wide) return launch_t<J, D, false, true, false, true, (J >= 2 ? 512 : 256)>(s, g, lds, blocks, pre);
        if (vb_lds) return launch_t<J, D, false, true, false, true>(s, g, lds, blocks, pre);
        return launch_t<J, D, false, false, false, true>(s, g, lds, blocks, pre);
    }
    if (J >= 2 && wide) return launch_t<J, D, false, true, false, false, (J >= 2 ? 512 : 256)>(s, g, lds, blocks, nullptr);
    if (vb_lds) return launch_t<J, D, false, true, false, false>(s, g, lds, blocks, nullptr);
    return launch_t<J, D, false, false, false, false>(s, g, lds, blocks, nullptr);
}

template <int D>
hipError_t launch_d(hipStream_t s, const GenParams &g, bool grid_lds, const UpdateArgs *pre, bool *pre_done)
{
    switch ((g.ld + 127) / 128) {
        case 1: return launch_jd<1, D>(s, g, grid_lds, pre, pre_done);
        case 2: return launch_jd<2, D>(s, g, grid_lds, pre, pre_done);
        case 3: return launch_jd<3, D>(s, g, grid_lds, pre, pre_done);
        case 4: return launch_jd<4, D>(s, g, grid_lds, pre, pre_done);
        default: return hipErrorInvalidValue;
    }
}

// The fused form: see generate_body.  One query, rows of up to 128 floats, up to 16 direction
// vectors (the sizes whose first update part rides with the ranking), the grid read through L2.
template <int D>
hipError_t launch_fused(hipStream_t s, const GenParams &g, const UpdateArgs &post)
{
    constexpr int TB = 512, tile = TB / kLanes;
    const size_t lds = ((size_t)tile * padded_row_floats(D, g.cost.W) + 2 * 16 * 128 + 128 + 4) * sizeof(float) +
                       update_post_lds_doubles(post.N, post.m, TB) * sizeof(double);
    const int ntiles = (g.n_local + tile - 1) / tile;
    const int blocks = std::min(ntiles, 256);
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(&k_generate_fused<D>), lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_generate_fused<D>), dim3(blocks, 1), dim3(TB), lds, s, g, post);
    return hipGetLastError();
}

bool generate_can_fuse(const GenParams &g, bool grid_lds, const UpdateArgs *post)
{
    static const bool off = getenv("LMCMA_HIP_NO_FUSED_UPDATE") != nullptr;
    return !off && post != nullptr && g.fused_counter != nullptr && g.batch.queries == 1 && g.ld <= 128 && g.cov.L == nullptr &&
           !grid_lds && g.m <= 16 && post->N <= 128 && post->m == g.m && update_is_split(post->N, post->m) &&
           update_pre_rides_with_ranking(post->m) && post->n_partials >= 1 && g.n_local >= 16;
}

}  // namespace

#ifdef LMCMA_DIAG
extern "C" hipError_t lmcma_diag_generate_stamps(unsigned long long *out16)
{
    return hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_generate_stamps), 16 * sizeof(unsigned long long));
}
extern "C" hipError_t lmcma_diag_set_skip_report(int on)
{
    return hipMemcpyToSymbol(HIP_SYMBOL(g_diag_skip_report), &on, sizeof(int));
}
#endif

bool generate_fuses(const GenParams &g, bool grid_lds, const UpdateArgs *post) { return generate_can_fuse(g, grid_lds, post); }

hipError_t launch_generate(hipStream_t s, const GenParams &g, int dims, bool grid_lds, const UpdateArgs *pre,
                           bool *pre_done, const UpdateArgs *post, bool *post_done)
{
    if (pre_done) *pre_done = false;
    if (post_done) *post_done = false;
    if (g.n_local <= 0) return hipSuccess;
    if (g.batch.queries < 1) return hipErrorInvalidValue;
    if (g.ld <= 0 || (g.ld & 3) || g.ld > 512 || g.cost.W < 1 || dims * g.cost.W > g.ld)
        return hipErrorInvalidValue;
    if (grid_lds && (g.cost.grid_type != LMCMA_GRID_BITS || (g.cost.grid_words & 3))) return hipErrorInvalidValue;
    if (g.cov.L && (g.cov.ntiles < 1 || 16 * g.cov.ntiles < g.ld || g.cov.ntiles > 32 || g.cov.ld < 16 * g.cov.ntiles ||
                    (g.cov.ld & 3)))
        return hipErrorInvalidValue;
    if (generate_can_fuse(g, grid_lds, post)) {
        if (post_done) *post_done = true;
        switch (dims) {
            case 1: return launch_fused<1>(s, g, *post);
            case 2: return launch_fused<2>(s, g, *post);
            case 3: return launch_fused<3>(s, g, *post);
            default: return hipErrorInvalidValue;
        }
    }
    switch (dims) {
        case 1: return launch_d<1>(s, g, grid_lds, pre, pre_done);
        case 2: return launch_d<2>(s, g, grid_lds, pre, pre_done);
        case 3: return launch_d<3>(s, g, grid_lds, pre, pre_done);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace lmcma
