// Population sampling kernel: x_i = clamp(xmean + sigma * A z_i).
//
// Replaces the candidate loop of LMCMA::sample (lmcma_path_planner/src/lmcma.cpp:301-311),
// computeAz (:431-447) and CMABase::applyBoundaries (:220-230).  The scalar
// recurrence of computeAz, Az <- M Az + nj_t (v_t . z) pc_t, is unrolled to the
// closed form
//     A z = M^k z + sum_t (V_t . z) B_t,   B_t = M^(k-1-t) nj_t pc_t,
// whose coefficients are folded into B on the producer side.
//
// Work split: 16 lanes own one candidate; lane l owns the float4 chunks
// l, l+16, l+32, ... of the row, so one group reads and writes 256 contiguous
// bytes per step and the four groups of a wave64 share the same V/B cache lines.
// The k dot products are reduced over the 16 lanes with an xor butterfly, which
// leaves bit-identical sums in every lane of the group.
//
// Structured sampling (CMABase::sampleStandardNormal with a Cholesky factor,
// lmcma.cpp:212-218, applyCovL :857-864): z <- L z before A is applied.  Over the 16
// candidates of a workgroup that is the one dense product of the path, Z (16 x N) =
// Z0 (16 x N) L^T, and it runs on the matrix cores: the workgroup leaves its Philox
// normals in LDS, every wave takes output tiles of 16 coordinates and walks the 16-wide
// blocks of L's rows with v_mfma_f32_16x16x4_f32 (L is lower triangular, so tile t needs
// only its first t + 1 blocks), and the products go back through LDS into the lane layout
// of the rest of the kernel.
#include "device_funcs.hpp"

namespace lmcma {

namespace {

// J = float4 chunks per lane; the row stride ld (a multiple of 4) must be <= 64 J.
template <int J, bool COV>
__global__ __launch_bounds__(256) void k_sample(float *__restrict__ rows,
                                                const float *__restrict__ zbuf,
                                                const float *__restrict__ xmean,
                                                const float *__restrict__ V,
                                                const float *__restrict__ B,
                                                const float *__restrict__ lo,
                                                const float *__restrict__ hi,
                                                const SampleScalars *__restrict__ sc, int ld,
                                                int n_local, uint32_t first_candidate,
                                                uint32_t key0, uint32_t key1, const CovFactor cov)
{
    extern __shared__ __attribute__((aligned(16))) float s_cov[];
    const int lane = threadIdx.x & 15;
    const int cand = blockIdx.x * 16 + (threadIdx.x >> 4);
    const bool live = cand < n_local;
    const int c = live ? cand : n_local - 1;  // idle groups shadow the last row (shuffles need all lanes)
    const float sigma = sc->sigma;
    const float mk = sc->mk;
    const int k = sc->k;
    const uint32_t generation = sc->generation;
    const int nchunks = ld >> 2;

    float4 z[J], acc[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int chunk = j * 16 + lane;
        if (chunk < nchunks) {
            if (zbuf)
                z[j] = reinterpret_cast<const float4 *>(zbuf + (size_t)c * ld)[chunk];
            else
                z[j] = philox_normals((uint32_t)chunk, first_candidate + (uint32_t)c, generation,
                                      key0, key1);
        } else {
            z[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    if (COV) apply_factor<J, 16>(z, cov, s_cov);
#pragma unroll
    for (int j = 0; j < J; ++j) acc[j] = make_float4(mk * z[j].x, mk * z[j].y, mk * z[j].z, mk * z[j].w);

    // U direction vectors per pass: every dot product is with z itself, so the loads, products
    // and lane reductions of a pass are independent and overlap; the rank-one terms are still
    // added in the order t = 0, 1, 2, ...  Rows past k - 1 are read at a clamped index and dropped.
    constexpr int U = J <= 2 ? 8 : 4;
    for (int t0 = 0; t0 < k; t0 += U) {
        float4 vv[U][J];
        float d[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = t0 + u < k ? t0 + u : k - 1;
            const float4 *v = reinterpret_cast<const float4 *>(V + (size_t)t * ld);
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const int chunk = j * 16 + lane;
                vv[u][j] = v[chunk < nchunks ? chunk : nchunks - 1];
            }
        }
        asm volatile("" ::: "memory");  // all loads of the pass are requested before the first product
#pragma unroll
        for (int u = 0; u < U; ++u) {
            d[u] = 0.f;
#pragma unroll
            for (int j = 0; j < J; ++j) {
                // z is zero beyond the row, so a chunk read at a clamped index adds nothing
                d[u] = fmaf(vv[u][j].x, z[j].x, d[u]);
                d[u] = fmaf(vv[u][j].y, z[j].y, d[u]);
                d[u] = fmaf(vv[u][j].z, z[j].z, d[u]);
                d[u] = fmaf(vv[u][j].w, z[j].w, d[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = t0 + u < k ? t0 + u : k - 1;
            const float4 *b = reinterpret_cast<const float4 *>(B + (size_t)t * ld);
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const int chunk = j * 16 + lane;
                vv[u][j] = b[chunk < nchunks ? chunk : nchunks - 1];
            }
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int u = 0; u < U; ++u) d[u] = group_sum<16>(d[u]);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (t0 + u < k) {
#pragma unroll
                for (int j = 0; j < J; ++j) {
                    acc[j].x = fmaf(d[u], vv[u][j].x, acc[j].x);
                    acc[j].y = fmaf(d[u], vv[u][j].y, acc[j].y);
                    acc[j].z = fmaf(d[u], vv[u][j].z, acc[j].z);
                    acc[j].w = fmaf(d[u], vv[u][j].w, acc[j].w);
                }
            }
        }
    }

    if (!live) return;
    float4 *out = reinterpret_cast<float4 *>(rows + (size_t)cand * ld);
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int chunk = j * 16 + lane;
        if (chunk < nchunks) {
            const float4 xm = reinterpret_cast<const float4 *>(xmean)[chunk];
            const float4 l4 = reinterpret_cast<const float4 *>(lo)[chunk];
            const float4 h4 = reinterpret_cast<const float4 *>(hi)[chunk];
            float4 x;
            // max with the lower bound first, then min with the upper one (lmcma.cpp:222-229)
            x.x = fminf(fmaxf(fmaf(sigma, acc[j].x, xm.x), l4.x), h4.x);
            x.y = fminf(fmaxf(fmaf(sigma, acc[j].y, xm.y), l4.y), h4.y);
            x.z = fminf(fmaxf(fmaf(sigma, acc[j].z, xm.z), l4.z), h4.z);
            x.w = fminf(fmaxf(fmaf(sigma, acc[j].w, xm.w), l4.w), h4.w);
            out[chunk] = x;
        }
    }
}

}  // namespace

hipError_t launch_sample(hipStream_t s, float *rows, const float *zbuf, const float *xmean,
                         const float *V, const float *B, const float *lo, const float *hi,
                         const SampleScalars *sc, int ld, int n_local, uint32_t first_candidate,
                         uint64_t seed, const CovFactor *cov)
{
    if (n_local <= 0) return hipSuccess;
    if (ld <= 0 || (ld & 3) || ld > 512) return hipErrorInvalidValue;
    const dim3 grid((n_local + 15) / 16), block(256);
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    const int J = (ld + 63) / 64;
    CovFactor f{};
    size_t lds = 0;
    if (cov) {
        f = *cov;
        // the tiles cover the row, and every wave has room for its share of them
        if (!f.L || f.ntiles < 1 || 16 * f.ntiles < ld || f.ntiles > 4 * J || f.ld < 16 * f.ntiles || (f.ld & 3))
            return hipErrorInvalidValue;
        lds = (size_t)16 * (16 * f.ntiles + 4) * sizeof(float);  // <= 33 KiB
    }
#define LMCMA_SAMPLE_CASE(JJ)                                                                  \
    case JJ:                                                                                   \
        if (cov)                                                                               \
            hipLaunchKernelGGL((k_sample<JJ, true>), grid, block, lds, s, rows, zbuf, xmean, V, B, lo, hi, sc, \
                               ld, n_local, first_candidate, k0, k1, f);                       \
        else                                                                                   \
            hipLaunchKernelGGL((k_sample<JJ, false>), grid, block, 0, s, rows, zbuf, xmean, V, B, lo, hi, sc, \
                               ld, n_local, first_candidate, k0, k1, f);                       \
        break;
    switch (J) {
        LMCMA_SAMPLE_CASE(1)
        LMCMA_SAMPLE_CASE(2)
        LMCMA_SAMPLE_CASE(3)
        LMCMA_SAMPLE_CASE(4)
        LMCMA_SAMPLE_CASE(5)
        LMCMA_SAMPLE_CASE(6)
        LMCMA_SAMPLE_CASE(7)
        LMCMA_SAMPLE_CASE(8)
        default:
            return hipErrorInvalidValue;
    }
#undef LMCMA_SAMPLE_CASE
    return hipGetLastError();
}

}  // namespace lmcma
