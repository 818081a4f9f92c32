// Device functions shared by the sampling, cost and fused generation kernels.
#pragma once

#include <type_traits>

#include "device_types.hpp"

namespace lmcma {

// query q's copy of a context buffer (see Batch); null stays null
template <typename T>
__device__ __forceinline__ void shift_ptr(T *&p, size_t bytes)
{
    // pointer arithmetic, not an integer round trip: the compiler keeps treating p as global memory
    using Byte = typename std::conditional<std::is_const<T>::value, const char, char>::type;
    if (p) p = reinterpret_cast<T *>(reinterpret_cast<Byte *>(p) + bytes);
}


// ---- direct exchange: the consumer's wait (PeerWait) ------------------------------------
// Called by every thread of a workgroup, before its first read of what the peers send.  The first
// wave watches one counter per lane; the wait ends when every other source has arrived or after
// kPeerWaitTicks of the 100 MHz clock (then *error is set and the kernel goes on with whatever the
// mailbox holds: every wave reaches its end whatever the peers do, and the host reports the error).
constexpr unsigned long long kPeerWaitTicks = 400000000ull;  // 4 s

__device__ __forceinline__ void peer_wait(const PeerWait &w)
{
    if (w.flags == nullptr) return;  // (uniform: a kernel argument)
    if (threadIdx.x < 64) {
        // (a wait that gave up before has set the error word: the run is lost, nothing waits again)
        const bool lost = __hip_atomic_load(w.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u;
        if (!lost && (int)threadIdx.x < w.n && (int)threadIdx.x != w.self) {
            const uint32_t *f = w.flags + threadIdx.x;
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            while ((int32_t)(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - w.want) < 0) {
                __builtin_amdgcn_s_sleep(2);
                if (__builtin_amdgcn_s_memrealtime() - t0 > kPeerWaitTicks) {
                    // what was waited for, for the report: the counter's value, the value wanted, the counter's index
                    w.error[1] = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    w.error[2] = w.want;
                    w.error[3] = (uint32_t)(f - (w.error - 128));
                    __hip_atomic_store(w.error, 1u + (uint32_t)threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    break;
                }
            }
        }
    }
    __syncthreads();
    // what the peers stored before they bumped their counters is visible to the loads below
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
}

// ---- direct exchange: the producer's side (PeerSend) --------------------------------------
// A store into a peer's mailbox: write-through at system scope.  Nothing of it stays in this
// device's caches, so once the stores have been acknowledged (the wait in peer_arrive) they are in
// the peer's memory, and no write-back of this device's L2 is needed before the counter moves (a
// release fence would order one: every workgroup of a producer then walks the L2, which costs tens
// of microseconds per launch).
template <typename T>
__device__ __forceinline__ void peer_store(T *p, T v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Called by every thread of a workgroup after its peer_store calls: the workgroup's part of the
// block is on its way, the destinations' arrival counters move by one.
__device__ __forceinline__ void peer_arrive(const PeerSend &s)
{
    __builtin_amdgcn_s_waitcnt(0);                          // this lane's stores have been acknowledged
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // (and the compiler moves nothing across)
    __syncthreads();
    if ((int)threadIdx.x < s.n && s.flag[threadIdx.x] != nullptr)
        __hip_atomic_fetch_add(s.flag[threadIdx.x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- counter-based normals ---------------------------------------------------------

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t (&out)[4])
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0;
        const uint32_t n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// sin(2 pi u), cos(2 pi u) for u in [0, 1): the angle in quarter turns n + r, |r| <= 1/2, and the
// two polynomials of sinpi / cospi on [-1/4, 1/4] turns (degrees 7 and 8; absolute error below
// 2e-7), exchanged and negated by quadrant.  The library's sincospif adds the handling of huge,
// infinite and NaN arguments, which cannot occur here.
__device__ __forceinline__ void sincos_2pi(float u, float &s, float &c)
{
    const float t = 4.0f * u;                 // quarter turns, [0, 4)
    const float n = rintf(t);                 // 0 .. 4
    const float r = (t - n) * 0.25f;          // turns / 2 ... in [-1/8, 1/8]: angle = 2 pi (n / 4 + r)
    const float x = 2.0f * r;                 // half turns in [-1/4, 1/4]: sinpi(x), cospi(x)
    const float x2 = x * x;
    // sinpi(x) = x (pi + x^2 (-pi^3/6 + x^2 (pi^5/120 + x^2 (-pi^7/5040))))
    float sp = fmaf(x2, -0.59926452f, 2.55016404f);
    sp = fmaf(x2, sp, -5.16771278f);
    sp = fmaf(x2, sp, 3.14159265f);
    sp *= x;
    // cospi(x) = 1 + x^2 (-pi^2/2 + x^2 (pi^4/24 + x^2 (-pi^6/720 + x^2 pi^8/40320)))
    float cp = fmaf(x2, 0.23533063f, -1.33526277f);
    cp = fmaf(x2, cp, 4.05871213f);
    cp = fmaf(x2, cp, -4.93480220f);
    cp = fmaf(x2, cp, 1.0f);
    const int q = (int)n;                     // quadrant 0..4 (4 = a full turn)
    const bool swap = (q & 1) != 0;
    const float ss = swap ? cp : sp, cc = swap ? sp : cp;
    // quadrant 1: (cos, -sin); 2: (-sin, -cos); 3: (-cos, sin)
    s = (q & 2) ? -ss : ss;
    c = ((q + 1) & 2) ? -cc : cc;
}

// Four standard normals for element quad `chunk` of (generation, candidate):
// counter (chunk, candidate, generation, 0), key = seed; two Box-Muller transforms
// on u = ((r >> 9) + 0.5) 2^-23 in (0,1), which fp32 holds exactly.
__device__ __forceinline__ float4 philox_normals(uint32_t chunk, uint32_t candidate,
                                                 uint32_t generation, uint32_t k0, uint32_t k1)
{
    uint32_t r[4];
    philox4x32_10(chunk, candidate, generation, 0u, k0, k1, r);
    const float scale = 1.0f / 8388608.0f;
    const float u1a = ((float)(r[0] >> 9) + 0.5f) * scale;
    const float u2a = ((float)(r[1] >> 9) + 0.5f) * scale;
    const float u1b = ((float)(r[2] >> 9) + 0.5f) * scale;
    const float u2b = ((float)(r[3] >> 9) + 0.5f) * scale;
    // (v_sqrt_f32: 1 ulp)
    const float ra = __builtin_amdgcn_sqrtf(-2.0f * logf(u1a));
    const float rb = __builtin_amdgcn_sqrtf(-2.0f * logf(u1b));
    float sa, ca, sb, cb;
    sincos_2pi(u2a, sa, ca);
    sincos_2pi(u2b, sb, cb);
    return make_float4(ra * ca, ra * sa, rb * cb, rb * sb);
}

// ---- cross-lane sums ----------------------------------------------------------------
// DPP moves data between lanes of a 16-lane row inside the VALU (no LDS round trip):
// row_ror:n rotates a row by n lanes, quad_perm permutes within groups of four.

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v)
{
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xF, 0xF, true);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}

// Sum over each aligned row of 16 lanes, result in all 16.  Steps: rotate by 8, by 4,
// swap pairs of pairs, swap neighbours.  At every step the lanes that end up with the
// same set of terms add the same two values (in either order), so all 16 lanes hold
// identical bits at the end.
template <typename T>
__device__ __forceinline__ T row16_sum(T v)
{
    v += dpp_mov<0x128>(v);  // row_ror:8
    v += dpp_mov<0x124>(v);  // row_ror:4
    v += dpp_mov<0x4E>(v);   // quad_perm:[2,3,0,1]
    v += dpp_mov<0xB1>(v);   // quad_perm:[1,0,3,2]
    return v;
}

// sum over groups of G consecutive lanes (G = 16, 32 or 64), identical bits in every lane
// The steps across rows use the gfx950 row-swap instructions (VALU, no LDS round trip):
// v_permlane16_swap exchanges the odd rows of its first operand with the even rows of its
// second, so with both operands = v the two results are [r0 r0 r2 r2] and [r1 r1 r3 r3],
// whose sum is what an exchange with lane ^ 16 would give; v_permlane32_swap does the same
// with the two halves of the wave.
__device__ __forceinline__ float add_lane_xor16(float v)
{
    const uint32_t b = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane16_swap(b, b, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

__device__ __forceinline__ float add_lane_xor32(float v)
{
    const uint32_t b = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane32_swap(b, b, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

__device__ __forceinline__ uint32_t add_lane_xor16(uint32_t v)
{
    const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    return r[0] + r[1];
}

__device__ __forceinline__ uint32_t add_lane_xor32(uint32_t v)
{
    const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return r[0] + r[1];
}

// the same exchanges for a double: its two words travel separately
__device__ __forceinline__ double add_lane_xor16(double v)
{
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const auto lo = __builtin_amdgcn_permlane16_swap((uint32_t)b, (uint32_t)b, false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((uint32_t)(b >> 32), (uint32_t)(b >> 32), false, false);
    return __builtin_bit_cast(double, ((unsigned long long)hi[0] << 32) | lo[0]) +
           __builtin_bit_cast(double, ((unsigned long long)hi[1] << 32) | lo[1]);
}

__device__ __forceinline__ double add_lane_xor32(double v)
{
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const auto lo = __builtin_amdgcn_permlane32_swap((uint32_t)b, (uint32_t)b, false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((uint32_t)(b >> 32), (uint32_t)(b >> 32), false, false);
    return __builtin_bit_cast(double, ((unsigned long long)hi[0] << 32) | lo[0]) +
           __builtin_bit_cast(double, ((unsigned long long)hi[1] << 32) | lo[1]);
}

// sum over the 64 lanes of a wave, identical bits in every lane
__device__ __forceinline__ double wave_sum(double v)
{
    return add_lane_xor32(add_lane_xor16(row16_sum(v)));
}

template <int G>
__device__ __forceinline__ float group_sum(float v)
{
    v = row16_sum(v);
    if (G >= 32) v = add_lane_xor16(v);
    if (G >= 64) v = add_lane_xor32(v);
    return v;
}

template <int CTRL>
__device__ __forceinline__ uint32_t dpp_mov(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}

// ---- ranking keys --------------------------------------------------------------------
// order-preserving image of a float cost in an unsigned word: a < b  <=>  key(a) < key(b)

__device__ __forceinline__ uint32_t order_bits(float f)
{
    if (f != f) f = __builtin_inff();  // NaN ranks with +inf
    f += 0.0f;                          // -0 compares equal to +0 in the reference
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ float order_bits_inv(uint32_t k)
{
    const uint32_t u = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
    return __uint_as_float(u);
}

// ---- trajectory cost ---------------------------------------------------------------

// cell of the grid a point falls into
template <int D>
__device__ __forceinline__ int grid_cell(const CostParams &p, const float *q)
{
    // clamp in float first: a NaN coordinate falls to cell 0, huge values saturate;
    // rintf rounds half to even like nearbyint
    const int ix = (int)rintf(fminf(fmaxf(q[0], 0.f), (float)(p.nx - 1)));
    const int iy = (D > 1) ? (int)rintf(fminf(fmaxf(q[1], 0.f), (float)(p.ny - 1))) : 0;
    const int iz = (D > 2) ? (int)rintf(fminf(fmaxf(q[2], 0.f), (float)(p.nz - 1))) : 0;
    return (iz * p.ny + iy) * p.nx + ix;
}

__device__ __forceinline__ float grid_value_at(const CostParams &p, const uint32_t *bits, int cell)
{
    if (p.grid_type == LMCMA_GRID_BITS)
        return ((bits[cell >> 5] >> (cell & 31)) & 1u) ? p.c_obs : 0.f;
    return reinterpret_cast<const float *>(p.grid)[cell];
}

template <int D>
__device__ __forceinline__ float grid_value(const CostParams &p, const uint32_t *bits, const float *q)
{
    return grid_value_at(p, bits, grid_cell<D>(p, q));
}

// Contribution of waypoint k of `row` (dimension-major, W per dimension) to the
// three cost terms, in two halves so that a caller can have the grid lookups of
// several waypoints in flight at once: waypoint_prepare does everything up to the
// cell index, waypoint_finish takes the grid value.  The trapezoid grid integral is
// regrouped per node: 1/2 c(node) (len_prev + len_next), with the fixed endpoints'
// shares added at the first and last waypoint.
struct WaypointPart {
    float len_prev, len_next, acc2;
    int cell;
};

template <int D>
__device__ __forceinline__ WaypointPart waypoint_prepare(const CostParams &p, const float *row, int k)
{
    constexpr float c_far = -1.0f / 12.0f, c_near = 16.0f / 12.0f;
    const int W = p.W;
    // The waypoint and its four neighbours, per dimension, read at clamped indices and
    // selected afterwards: 5 D independent loads, all requested before the first use (the
    // barrier keeps the compiler from sinking a load into the branch of its select).
    float r0[D], rm1[D], rp1[D], rm2[D], rp2[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const float *r = row + d * W;
        r0[d] = r[k];
        rm1[d] = r[k >= 1 ? k - 1 : 0];
        rp1[d] = r[k + 1 < W ? k + 1 : W - 1];
        rm2[d] = r[k >= 2 ? k - 2 : 0];
        rp2[d] = r[k + 2 < W ? k + 2 : W - 1];
    }
    asm volatile("" ::: "memory");
    float q0[D];
    float len_prev2 = 0.f, len_next2 = 0.f, acc2 = 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const float v0 = r0[d];
        const float vm1 = (k >= 1) ? rm1[d] : p.start[d];
        const float vp1 = (k + 1 < W) ? rp1[d] : p.goal[d];
        float sm1 = vm1, sp1 = vp1;
        float sm2 = (k >= 2) ? rm2[d] : p.start[d];
        float sp2 = (k + 2 < W) ? rp2[d] : p.goal[d];
        if (p.pad_zero) {
            if (k < 1) sm1 = 0.f;
            if (k < 2) sm2 = 0.f;
            if (k + 1 >= W) sp1 = 0.f;
            if (k + 2 >= W) sp2 = 0.f;
        }
        q0[d] = v0;
        const float dm = v0 - vm1, dp = vp1 - v0;
        len_prev2 = fmaf(dm, dm, len_prev2);
        len_next2 = fmaf(dp, dp, len_next2);
        // -1/12 s[k-2] + 16/12 s[k-1] - 30/12 s[k] + 16/12 s[k+1] - 1/12 s[k+2],
        // written on differences to s[k] (the five weights sum to zero)
        const float a = fmaf(c_far, (sm2 - v0) + (sp2 - v0), c_near * ((sm1 - v0) + (sp1 - v0)));
        acc2 = fmaf(a, a, acc2);
    }
    WaypointPart w;
    // v_sqrt_f32: 1 ulp, well inside the cost's tolerance; sqrtf is a dozen instructions more
    w.len_prev = __builtin_amdgcn_sqrtf(len_prev2);
    w.len_next = __builtin_amdgcn_sqrtf(len_next2);
    w.acc2 = acc2;
    w.cell = grid_cell<D>(p, q0);
    return w;
}

// The same for a row kept in LDS with its dimensions padded: dimension d occupies
// prow[d (W + 4) + 2 .. + W), preceded by two copies of start[d] and followed by two of goal[d],
// so the waypoint's four neighbours are read at fixed offsets with no index clamp and, with the
// end points as padding (LMCMA_PAD_ENDPOINT), no select.  Same arithmetic as waypoint_prepare.
template <int D>
__device__ __forceinline__ WaypointPart waypoint_prepare_padded(const CostParams &p, const float *prow, int k)
{
    constexpr float c_far = -1.0f / 12.0f, c_near = 16.0f / 12.0f;
    const int W = p.W;
    float r0[D], rm1[D], rp1[D], rm2[D], rp2[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const float *r = prow + d * (W + 4) + 2 + k;
        r0[d] = r[0];
        rm1[d] = r[-1];
        rp1[d] = r[1];
        rm2[d] = r[-2];
        rp2[d] = r[2];
    }
    asm volatile("" ::: "memory");
    float q0[D];
    float len_prev2 = 0.f, len_next2 = 0.f, acc2 = 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const float v0 = r0[d];
        const float vm1 = rm1[d], vp1 = rp1[d];
        float sm1 = vm1, sp1 = vp1, sm2 = rm2[d], sp2 = rp2[d];
        if (p.pad_zero) {
            if (k < 1) sm1 = 0.f;
            if (k < 2) sm2 = 0.f;
            if (k + 1 >= W) sp1 = 0.f;
            if (k + 2 >= W) sp2 = 0.f;
        }
        q0[d] = v0;
        const float dm = v0 - vm1, dp = vp1 - v0;
        len_prev2 = fmaf(dm, dm, len_prev2);
        len_next2 = fmaf(dp, dp, len_next2);
        const float a = fmaf(c_far, (sm2 - v0) + (sp2 - v0), c_near * ((sm1 - v0) + (sp1 - v0)));
        acc2 = fmaf(a, a, acc2);
    }
    WaypointPart w;
    w.len_prev = __builtin_amdgcn_sqrtf(len_prev2);
    w.len_next = __builtin_amdgcn_sqrtf(len_next2);
    w.acc2 = acc2;
    w.cell = grid_cell<D>(p, q0);
    return w;
}

__device__ __forceinline__ void waypoint_finish(const CostParams &p, const WaypointPart &w, int k, float cv,
                                                float &j_len, float &j_grid, float &j_smooth)
{
    float gl = cv * (0.5f * (w.len_prev + w.len_next));
    float ll = w.len_next;
    if (k == 0) {
        ll += w.len_prev;
        gl = fmaf(p.c_start, 0.5f * w.len_prev, gl);
    }
    if (k == p.W - 1) gl = fmaf(p.c_goal, 0.5f * w.len_next, gl);
    j_len += ll;
    j_grid += gl;
    j_smooth += w.acc2;
}

template <int D>
__device__ __forceinline__ void waypoint_terms(const CostParams &p, const uint32_t *bits,
                                               const float *row, int k, float &j_len, float &j_grid,
                                               float &j_smooth)
{
    const WaypointPart w = waypoint_prepare<D>(p, row, k);
    waypoint_finish(p, w, k, grid_value_at(p, bits, w.cell), j_len, j_grid, j_smooth);
}

// weighted total, the three terms added left to right; NaN ranks last
__device__ __forceinline__ float total_cost(const CostParams &p, float j_len, float j_grid, float j_smooth)
{
    float total = fmaf(p.w_smooth, j_smooth, fmaf(p.w_grid, j_grid, p.w_len * j_len));
    if (total != total) total = __builtin_inff();
    return total;
}

// ---- structured sampling: z <- L z for the candidates of a workgroup ------------------
// CMABase::sampleStandardNormal with a Cholesky factor (L/lmcma.cpp:212-218, applyCovL
// :857-864).  Over the candidates of a workgroup that is the one dense product of the
// path, Z (16 x N) = Z0 (16 x N) L^T, and it runs on the matrix cores: the workgroup leaves
// its normals in LDS, every wave takes output tiles of 16 coordinates and walks the 16-wide
// blocks of L's rows with v_mfma_f32_16x16x4_f32 (tile t of a lower triangle needs only its
// first t + 1 blocks), and the products go back through LDS into the callers' lane layout.
//
// 256 threads; LANES (16 or 32) lanes own a candidate, lane l the float4 chunks l,
// l + LANES, ... of its row (zero beyond the row); a workgroup of 8 candidates leaves half
// of the 16 matrix rows empty.  s_z: [16][16 ntiles + 4] floats.
typedef float mfma_f32x4 __attribute__((ext_vector_type(4)));

// tile i of this wave: 4 i + wave, then 4 i + 3 - wave, alternating, so that every wave gets
// the same share of a triangle
__device__ __forceinline__ int factor_tile(int i, int wave) { return 4 * i + ((i & 1) ? 3 - wave : wave); }

// A second call needs a barrier after the first one's callers have used z (the calls share s_z).
// (Keeping a wave's blocks of L in registers for the whole kernel was measured slower than
// reading them from L2 in every call.)
template <int J, int LANES>
__device__ __forceinline__ void apply_factor(float4 (&z)[J], const CovFactor &f, float *s_z)
{
    constexpr int CANDS = 256 / LANES;
    constexpr int TPW = J * LANES / 16;  // output tiles per wave: 16 ntiles <= 4 J LANES
    const int tid = threadIdx.x, lane = tid & (LANES - 1), grp = tid / LANES;
    const int wl = tid & 63, wave = tid >> 6;
    const int np = 16 * f.ntiles, ldp = np + 4;  // the pad spreads the 16 rows over the banks
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int col = 4 * (j * LANES + lane);
        if (col < np) *reinterpret_cast<float4 *>(s_z + grp * ldp + col) = z[j];
    }
    if (CANDS < 16) {
        for (int i = tid; i < (16 - CANDS) * (ldp >> 2); i += 256)
            reinterpret_cast<float4 *>(s_z + CANDS * ldp)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    const int row = wl & 15, kq = 4 * (wl >> 4);
    const float *a_src = s_z + row * ldp + kq;
    mfma_f32x4 acc[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        acc[i] = mfma_f32x4{0.f, 0.f, 0.f, 0.f};
        const int nt = factor_tile(i, wave);
        if (nt < f.ntiles) {
            const int nkb = f.kblocks[nt];
            // lane (row, q) feeds element k = 16 kb + 4 q + s of its row in step s: the four
            // steps of a block together cover its 16 k
            const float *b_src = f.L + (size_t)(16 * nt + row) * f.ld + kq;
            for (int kb0 = 0; kb0 < nkb; kb0 += 4) {
                // four blocks per step: all eight loads first, then the sixteen products
                float4 a[4], b[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int kb = kb0 + u < nkb ? kb0 + u : nkb - 1;
                    a[u] = *reinterpret_cast<const float4 *>(a_src + 16 * kb);
                    b[u] = *reinterpret_cast<const float4 *>(b_src + 16 * kb);
                    if (kb0 + u >= nkb) a[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].x, b[u].x, acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].y, b[u].y, acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].z, b[u].z, acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].w, b[u].w, acc[i], 0, 0, 0);
                }
            }
        }
    }
    __syncthreads();  // every wave has read the normals
    // element r of a lane's result is candidate 4 (lane / 16) + r, coordinate 16 nt + lane % 16
    if (kq < CANDS) {
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int nt = factor_tile(i, wave);
            if (nt < f.ntiles) {
#pragma unroll
                for (int r = 0; r < 4; ++r) s_z[(kq + r) * ldp + 16 * nt + row] = acc[i][r];
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int col = 4 * (j * LANES + lane);
        if (col < np) z[j] = *reinterpret_cast<const float4 *>(s_z + grp * ldp + col);
    }
}

// floats of LDS apply_factor needs
__host__ __device__ inline size_t factor_lds_floats(int ntiles) { return (size_t)16 * (16 * ntiles + 4); }

}  // namespace lmcma
