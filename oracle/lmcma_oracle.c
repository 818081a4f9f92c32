/*
 * CPU oracle — see lmcma_oracle.h.  TEST INFRASTRUCTURE ONLY: the product
 * library never links or calls this file.  Parity status and the meaning of the
 * L/ and S/ citation prefixes are given in the header.
 *
 * The arithmetic below keeps the reference's order of floating-point operations
 * (loops, accumulation direction, where a scale is applied) so that at m = lambda
 * it is an operation-for-operation restatement.  Build without -ffast-math and
 * with -ffp-contract=off.
 */
#include "lmcma_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ======================================================================= */
/* A1  RNG: Park-Miller minimal standard (Schrage) + 32-entry shuffle table +
 *     polar Gaussian.  L/lmcma.cpp:14-82                                    */
/* ======================================================================= */

long orc_rng_start(orc_rng *t, unsigned long inseed)
{
    /* L/lmcma.cpp:14-33: 40 warm-up steps, the last 32 fill the table top-down */
    long tmp;
    int i;
    t->flgstored = 0;
    t->hold = 0.0;
    t->startseed = (long)inseed;
    if (inseed < 1)
        inseed = 1;
    t->aktseed = (long)inseed;
    for (i = 39; i >= 0; --i) {
        tmp = t->aktseed / 127773;
        t->aktseed = 16807 * (t->aktseed - tmp * 127773) - 2836 * tmp;
        if (t->aktseed < 0)
            t->aktseed += 2147483647;
        if (i < 32)
            t->rgrand[i] = t->aktseed;
    }
    t->aktrand = t->rgrand[0];
    return (long)inseed;
}

double orc_rng_uniform(orc_rng *t)
{
    /* L/lmcma.cpp:49-61 */
    long tmp = t->aktseed / 127773;
    t->aktseed = 16807 * (t->aktseed - tmp * 127773) - 2836 * tmp;
    if (t->aktseed < 0)
        t->aktseed += 2147483647;
    tmp = t->aktrand / 67108865;
    t->aktrand = t->rgrand[tmp];
    t->rgrand[tmp] = t->aktseed;
    return (double)(t->aktrand) / (2.147483647e9);
}

double orc_rng_gauss(orc_rng *t)
{
    /* L/lmcma.cpp:63-82: returns fac*x2 first, keeps fac*x1 for the next call */
    double x1, x2, rquad, fac;
    if (t->flgstored) {
        t->flgstored = 0;
        return t->hold;
    }
    do {
        x1 = 2.0 * orc_rng_uniform(t) - 1.0;
        x2 = 2.0 * orc_rng_uniform(t) - 1.0;
        rquad = x1 * x1 + x2 * x2;
    } while (rquad >= 1 || rquad <= 0);
    fac = sqrt(-2.0 * log(rquad) / rquad);
    t->flgstored = 1;
    t->hold = fac * x1;
    return fac * x2;
}

/* ======================================================================= */
/* Counter-based generator for the device fast path (no reference
 * counterpart: the reference's generator is strictly sequential).
 * Philox4x32-10, Salmon et al. 2011.                                        */
/* ======================================================================= */

void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

void orc_philox_normals(uint64_t seed, uint32_t generation, uint32_t candidate,
                        int N, double *z)
{
    /* element quad q = n/4 uses counter (q, candidate, generation, 0); the two
     * word pairs feed two Box-Muller transforms: (r0,r1)->z[4q],z[4q+1] and
     * (r2,r3)->z[4q+2],z[4q+3].  u = ((r>>9)+0.5)*2^-23 lies in (0,1) and is
     * exactly representable in fp32 as well (24 significant bits).          */
    const uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
    const double two_pi = 6.283185307179586476925286766559;
    for (int q = 0; 4 * q < N; ++q) {
        uint32_t ctr[4] = { (uint32_t)q, candidate, generation, 0u };
        uint32_t r[4];
        orc_philox4x32_10(ctr, key, r);
        for (int h = 0; h < 2; ++h) {
            double u1 = ((double)(r[2 * h] >> 9) + 0.5) * (1.0 / 8388608.0);
            double u2 = ((double)(r[2 * h + 1] >> 9) + 0.5) * (1.0 / 8388608.0);
            double rad = sqrt(-2.0 * log(u1));
            double a = two_pi * u2;
            int n = 4 * q + 2 * h;
            if (n < N) z[n] = rad * cos(a);
            if (n + 1 < N) z[n + 1] = rad * sin(a);
        }
    }
}

/* ======================================================================= */
/* A6  sort.  L/lmcma.cpp:84-104 calls libc qsort with a value-only
 * comparator; glibc's qsort is a merge sort when it can allocate, so equal
 * values keep their input order.  A merge sort is written out here so that
 * the tie order (lower id first) does not depend on the C library.          */
/* ======================================================================= */

typedef struct { double value; int id; } orc_sv;

static void merge_sort_sv(orc_sv *a, orc_sv *tmp, int n)
{
    if (n < 2) return;
    int h = n / 2;
    merge_sort_sv(a, tmp, h);
    merge_sort_sv(a + h, tmp, n - h);
    int i = 0, j = h, k = 0;
    while (i < h && j < n) {
        /* take from the right run only when strictly smaller: stable */
        if (a[j].value < a[i].value) tmp[k++] = a[j++];
        else tmp[k++] = a[i++];
    }
    while (i < h) tmp[k++] = a[i++];
    while (j < n) tmp[k++] = a[j++];
    memcpy(a, tmp, (size_t)n * sizeof(orc_sv));
}

void orc_sort(int sz, double *values, int *index)
{
    orc_sv *a = (orc_sv *)malloc((size_t)sz * sizeof(orc_sv));
    orc_sv *tmp = (orc_sv *)malloc((size_t)sz * sizeof(orc_sv));
    for (int i = 0; i < sz; ++i) {
        /* the reference's comparator is undefined for NaN (L/lmcma.cpp:84-89);
         * a NaN cost ranks last here */
        a[i].value = isnan(values[i]) ? INFINITY : values[i];
        a[i].id = i;
    }
    merge_sort_sv(a, tmp, sz);
    for (int i = 0; i < sz; ++i) {
        values[i] = a[i].value;   /* L/lmcma.cpp:101: fitness overwritten sorted */
        index[i] = a[i].id;
    }
    free(a);
    free(tmp);
}

/* ======================================================================= */
/* A12  smoothness matrices.  L/lmcma.cpp:756-864                            */
/* ======================================================================= */

static const double ORC_DIFF_RULES[4][7] = {
    /* L/lmcma.cpp:760-765 */
    { 0, 0, 0, 1, 0, 0, 0 },
    { 0, 0, -1, 1, 0, 0, 0 },
    { 0, -1 / 12.0, 16 / 12.0, -30 / 12.0, 16 / 12.0, -1 / 12.0, 0 },
    { 0, 1 / 12.0, -17 / 12.0, 46 / 12.0, -46 / 12.0, 17 / 12.0, -1 / 12.0 }
};

void orc_differentiation_matrix(int num_time_steps, int order, double dt,
                                double *diff_matrix, int row_len)
{
    /* L/lmcma.cpp:812-834: banded matrix, stencil truncated at both ends */
    if (row_len < 0)
        row_len = num_time_steps;
    for (int i = 0; i < num_time_steps; ++i)
        for (int j = 0; j < num_time_steps; ++j)
            diff_matrix[i * row_len + j] = 0.0;
    double multiplier = 1.0 / pow(dt, (int)order);
    for (int i = 0; i < num_time_steps; ++i)
        for (int j = -3; j <= 3; ++j) {
            int index = i + j;
            if (index < 0 || index >= num_time_steps)
                continue;
            diff_matrix[i * row_len + index] += multiplier * ORC_DIFF_RULES[order][j + 3];
        }
}

int orc_invert(const double *A, double *Ainv, int N)
{
    /* L/lmcma.cpp:836-842 defers to Eigen's MatrixXd::inverse() (partial-pivot
     * LU).  Eigen is a third-party dependency that is not vendored; this is
     * Gauss-Jordan with partial pivoting, equal up to rounding.  The matrix
     * this is used on is symmetric, so row/column-major does not matter.   */
    double *w = (double *)malloc((size_t)N * 2 * N * sizeof(double));
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) {
            w[i * 2 * N + j] = A[j * N + i]; /* column-major like Eigen::Map */
            w[i * 2 * N + N + j] = (i == j) ? 1.0 : 0.0;
        }
    for (int c = 0; c < N; ++c) {
        int piv = c;
        for (int r = c + 1; r < N; ++r)
            if (fabs(w[r * 2 * N + c]) > fabs(w[piv * 2 * N + c]))
                piv = r;
        if (w[piv * 2 * N + c] == 0.0) { free(w); return -1; }
        if (piv != c)
            for (int j = 0; j < 2 * N; ++j) {
                double tsw = w[c * 2 * N + j];
                w[c * 2 * N + j] = w[piv * 2 * N + j];
                w[piv * 2 * N + j] = tsw;
            }
        double d = w[c * 2 * N + c];
        for (int j = 0; j < 2 * N; ++j)
            w[c * 2 * N + j] /= d;
        for (int r = 0; r < N; ++r) {
            if (r == c) continue;
            double f = w[r * 2 * N + c];
            if (f == 0.0) continue;
            for (int j = 0; j < 2 * N; ++j)
                w[r * 2 * N + j] -= f * w[c * 2 * N + j];
        }
    }
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j)
            Ainv[j * N + i] = w[i * 2 * N + N + j]; /* column-major out */
    free(w);
    return 0;
}

int orc_cholesky(const double *C, double *L, int N)
{
    /* L/lmcma.cpp:844-855: Lm = llt(C).matrixL(); stored as L[m*N+n] = Lm(n,m),
     * i.e. column-major lower factor.                                        */
    memset(L, 0, (size_t)N * N * sizeof(double));
    for (int j = 0; j < N; ++j) {
        double d = C[j * N + j];
        for (int k = 0; k < j; ++k)
            d -= L[k * N + j] * L[k * N + j];
        if (!(d > 0.0)) return -1;
        d = sqrt(d);
        L[j * N + j] = d;
        for (int i = j + 1; i < N; ++i) {
            double s = C[j * N + i]; /* column-major C(i,j) */
            for (int k = 0; k < j; ++k)
                s -= L[k * N + i] * L[k * N + j];
            L[j * N + i] = s / d;
        }
    }
    return 0;
}

void orc_apply_cov_l(const double *L, double *z, int N)
{
    /* L/lmcma.cpp:857-864: z <- Lm * z with Lm(r,c) = L[c*N + r] */
    double *out = (double *)calloc((size_t)N, sizeof(double));
    for (int c = 0; c < N; ++c)
        for (int r = 0; r < N; ++r)
            out[r] += L[c * N + r] * z[c];
    memcpy(z, out, (size_t)N * sizeof(double));
    free(out);
}

int orc_covariance(int num_dims, int num_waypoints, double *cov)
{
    /* L/lmcma.cpp:769-810 */
    const double dt = 1.0;
    const int order = 2;
    int n = num_dims * num_waypoints;
    double *diff = (double *)calloc((size_t)n * n, sizeof(double));
    double *inv = (double *)calloc((size_t)n * n, sizeof(double));
    for (int i = 0; i < num_dims; ++i) {
        double *offset = &diff[i * n * num_waypoints + i * num_waypoints];
        orc_differentiation_matrix(num_waypoints, order, dt, offset, n);
    }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j)
            for (int k = 0; k < n; ++k)
                inv[i * n + j] += diff[i * n + k] * diff[k * n + j];
    int rc = orc_invert(inv, cov, n);
    if (rc == 0) {
        double max_var = 0.0;
        for (int i = 0; i < n; ++i)
            max_var = fmax(max_var, cov[i * n + i]);
        const double scaling = max_var * num_waypoints;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j)
                cov[i * n + j] /= scaling;
    }
    free(diff);
    free(inv);
    return rc;
}

/* ======================================================================= */
/* A2-A11  LM-CMA.  L/lmcma.cpp:107-463                                      */
/* ======================================================================= */

struct orc_lmcma {
    int N, lambda, mu, m;
    double mueff;
    double *lo, *hi;
    double *arx, *arfitness, *xmean, *xold, *weights, *L;
    int itr, sample_idx, counteval;
    double best_f;
    orc_rng rng;
    int z_source;
    uint64_t philox_seed;
    /* LMCMA */
    double sigma, c1, cc, cs, val_target, K, M, s;
    int nvectors, maxsteps, iterator_sz;
    double *v_arr, *pc_arr, *pc, *z, *Az, *Av, *prev_arfitness, *mixed, *Nj, *Lj;
    int *iterator, *arindex, *ranks, *ranks_tmp, *t, *vec;
};

static double *dup_d(const double *src, int n)
{
    if (!src) return NULL;
    double *d = (double *)malloc((size_t)n * sizeof(double));
    memcpy(d, src, (size_t)n * sizeof(double));
    return d;
}

static void apply_boundaries(orc_lmcma *o)
{
    /* L/lmcma.cpp:220-230 */
    int N = o->N;
    if (o->lo)
        for (int i = 0; i < o->lambda; ++i)
            for (int n = 0; n < N; ++n)
                o->arx[i * N + n] = fmax(o->arx[i * N + n], o->lo[n]);
    if (o->hi)
        for (int i = 0; i < o->lambda; ++i)
            for (int n = 0; n < N; ++n)
                o->arx[i * N + n] = fmin(o->arx[i * N + n], o->hi[n]);
}

static void compute_az(const orc_lmcma *o, double *Az, const double *z)
{
    /* L/lmcma.cpp:431-447: the dot product is always with the ORIGINAL z */
    int N = o->N;
    for (int k = 0; k < N; ++k)
        Az[k] = z[k];
    for (int k = 0; k < o->iterator_sz; ++k) {
        int jcur = o->iterator[k];
        const double *pc_j = &o->pc_arr[jcur * N];
        const double *v_j = &o->v_arr[jcur * N];
        double v_j_mult_z = 0;
        for (int p = 0; p < N; ++p)
            v_j_mult_z = v_j_mult_z + v_j[p] * z[p];
        v_j_mult_z = o->Nj[jcur] * v_j_mult_z;
        for (int p = 0; p < N; ++p)
            Az[p] = o->M * Az[p] + v_j_mult_z * pc_j[p];
    }
}

static void inv_az(int N, double *Av, int iterator_sz, const int *iterator,
                   const double *v_arr, const double *Lj_arr, double K)
{
    /* L/lmcma.cpp:449-463: the dot product is with the EVOLVING Av */
    for (int j = 0; j < iterator_sz; ++j) {
        int jcur = iterator[j];
        const double *v_j = &v_arr[jcur * N];
        double v_j_mult_Av = 0;
        for (int p = 0; p < N; ++p)
            v_j_mult_Av += v_j[p] * Av[p];
        v_j_mult_Av = Lj_arr[jcur] * v_j_mult_Av;
        for (int p = 0; p < N; ++p)
            Av[p] = K * Av[p] - v_j_mult_Av * v_j[p];
    }
}

/* Host threads for the loops over candidates (the reference is single-threaded; candidates
 * are independent given the state, so the per-candidate arithmetic and results do not depend
 * on the thread count).  Used by the timed CPU baseline only; 1 = the reference's serial loop. */
static int g_threads = 1;
void orc_set_threads(int n) { g_threads = n < 1 ? 1 : n; }
int orc_get_threads(void) { return g_threads; }

static void sample_one(orc_lmcma *o, int i, const double *z, double *Az)
{
    /* L/lmcma.cpp:306-308 */
    int N = o->N;
    compute_az(o, Az, z);
    for (int k = 0; k < N; ++k)
        o->arx[i * N + k] = o->xmean[k] + o->sigma * Az[k];
}

void orc_lmcma_sample(orc_lmcma *o)
{
    /* L/lmcma.cpp:301-311 with sampleStandardNormal :212-218 */
    int N = o->N;
    if (o->z_source == ORC_Z_PHILOX && g_threads > 1) {
        /* counter-based normals: candidate i's stream does not depend on the others */
#pragma omp parallel num_threads(g_threads)
        {
            double *z = (double *)malloc(2 * (size_t)N * sizeof(double)), *Az = z + N;
#pragma omp for schedule(static)
            for (int i = 0; i < o->lambda; ++i) {
                orc_philox_normals(o->philox_seed, (uint32_t)o->itr, (uint32_t)i, N, z);
                if (o->L)
                    orc_apply_cov_l(o->L, z, N);
                sample_one(o, i, z, Az);
            }
            free(z);
        }
        apply_boundaries(o);
        return;
    }
    for (int i = 0; i < o->lambda; ++i) {
        if (o->z_source == ORC_Z_PHILOX) {
            orc_philox_normals(o->philox_seed, (uint32_t)o->itr, (uint32_t)i, N, o->z);
        } else {
            for (int n = 0; n < N; ++n)
                o->z[n] = orc_rng_gauss(&o->rng);
        }
        if (o->L)
            orc_apply_cov_l(o->L, o->z, N);
        sample_one(o, i, o->z, o->Az);
    }
    apply_boundaries(o);
}

void orc_lmcma_sample_given(orc_lmcma *o, const double *Z)
{
    int N = o->N;
    for (int i = 0; i < o->lambda; ++i) {
        memcpy(o->z, Z + (size_t)i * N, (size_t)N * sizeof(double));
        if (o->L)
            orc_apply_cov_l(o->L, o->z, N);
        sample_one(o, i, o->z, o->Az);
    }
    apply_boundaries(o);
}

orc_lmcma *orc_lmcma_create(int N, int lambda, int m, const double *initial,
                            const double *lo, const double *hi, double sigma,
                            const double *cov, long seed, int z_source)
{
    orc_lmcma *o = (orc_lmcma *)calloc(1, sizeof(orc_lmcma));
    o->N = N;
    o->z_source = z_source;
    o->philox_seed = (uint64_t)seed;
    o->best_f = DBL_MAX;                       /* L/lmcma.cpp:112 */
    orc_rng_start(&o->rng, (unsigned long)seed); /* L/lmcma.cpp:114 (seed >= 1 required) */

    /* CMABase::init, L/lmcma.cpp:130-170 */
    if (lambda < 1)
        lambda = 4 + (int)(3 * log((double)N));
    o->lambda = lambda;
    o->mu = lambda / 2;
    o->arx = (double *)calloc((size_t)N * lambda, sizeof(double));
    o->arfitness = (double *)calloc((size_t)lambda, sizeof(double));
    o->xmean = (double *)calloc((size_t)N, sizeof(double));
    o->xold = (double *)calloc((size_t)N, sizeof(double));
    o->weights = (double *)calloc((size_t)(o->mu > 0 ? o->mu : 1), sizeof(double));
    double sum_weights = 0;
    for (int i = 0; i < o->mu; ++i) {
        o->weights[i] = log((double)o->mu + 0.5) - log((double)(1 + i));
        sum_weights += o->weights[i];
    }
    o->mueff = 0.0;
    for (int i = 0; i < o->mu; ++i) {
        o->weights[i] /= sum_weights;
        o->mueff += o->weights[i] * o->weights[i];
    }
    o->mueff = 1.0 / o->mueff;
    if (initial)
        for (int i = 0; i < N; ++i) o->xmean[i] = initial[i];
    else
        for (int i = 0; i < N; ++i) o->xmean[i] = orc_rng_uniform(&o->rng);
    o->lo = dup_d(lo, N);
    o->hi = dup_d(hi, N);
    if (cov) {
        o->L = (double *)calloc((size_t)N * N, sizeof(double));
        orc_cholesky(cov, o->L, N);
    }

    /* LMCMA ctor + init, L/lmcma.cpp:233-299 */
    o->sigma = sigma;
    o->cs = 0.3;
    o->val_target = 0.25;
    o->m = (m < 1) ? lambda : m;
    o->nvectors = o->m;                        /* reference: nvectors = lambda (:266) */
    o->maxsteps = o->nvectors;
    o->c1 = 1.0 / (10 * log((double)(N + 1)));
    o->cc = 1.0 / o->nvectors;
    o->K = 1 / sqrt(1 - o->c1);
    o->M = sqrt(1 - o->c1);
    int nv = o->nvectors;
    o->v_arr = (double *)calloc((size_t)N * nv, sizeof(double));
    o->pc_arr = (double *)calloc((size_t)N * nv, sizeof(double));
    o->pc = (double *)calloc((size_t)N, sizeof(double));
    o->z = (double *)calloc((size_t)N, sizeof(double));
    o->Az = (double *)calloc((size_t)N, sizeof(double));
    o->Av = (double *)calloc((size_t)N, sizeof(double));
    o->iterator = (int *)calloc((size_t)nv, sizeof(int));
    o->prev_arfitness = (double *)calloc((size_t)lambda, sizeof(double));
    o->arindex = (int *)calloc((size_t)lambda, sizeof(int));
    o->mixed = (double *)calloc((size_t)2 * lambda, sizeof(double));
    o->ranks = (int *)calloc((size_t)2 * lambda, sizeof(int));
    o->ranks_tmp = (int *)calloc((size_t)2 * lambda, sizeof(int));
    o->Nj = (double *)calloc((size_t)nv, sizeof(double));
    o->Lj = (double *)calloc((size_t)nv, sizeof(double));
    o->t = (int *)calloc((size_t)nv, sizeof(int));
    o->vec = (int *)calloc((size_t)nv, sizeof(int));
    o->s = 0.0;
    o->iterator_sz = 0;
    if (z_source != ORC_Z_GIVEN)
        orc_lmcma_sample(o);                   /* L/lmcma.cpp:298 */
    return o;
}

void orc_lmcma_destroy(orc_lmcma *o)
{
    if (!o) return;
    free(o->lo); free(o->hi); free(o->arx); free(o->arfitness); free(o->xmean);
    free(o->xold); free(o->weights); free(o->L); free(o->v_arr); free(o->pc_arr);
    free(o->pc); free(o->z); free(o->Az); free(o->Av); free(o->iterator);
    free(o->prev_arfitness); free(o->arindex); free(o->mixed); free(o->ranks);
    free(o->ranks_tmp); free(o->Nj); free(o->Lj); free(o->t); free(o->vec);
    free(o);
}

static void lmcma_update(orc_lmcma *o, const double *xmean_override)
{
    /* L/lmcma.cpp:313-424.  xmean_override (test hook, normally NULL) replaces
     * the recombination result, e.g. with a mean that was summed shard by shard. */
    const int N = o->N, lambda = o->lambda, mu = o->mu, nvectors = o->nvectors;
    int *t = o->t, *vec = o->vec;
    orc_sort(lambda, o->arfitness, o->arindex);
    for (int i = 0; i < N; ++i) {
        o->xold[i] = o->xmean[i];
        o->xmean[i] = 0;
    }
    for (int i = 0; i < mu; ++i) {
        const double *cur_x = &o->arx[o->arindex[i] * N];
        for (int j = 0; j < N; ++j)
            o->xmean[j] += o->weights[i] * cur_x[j];
    }
    if (xmean_override)
        memcpy(o->xmean, xmean_override, (size_t)N * sizeof(double));
    for (int i = 0; i < N; ++i)
        o->pc[i] = (1 - o->cc) * o->pc[i] + sqrt(o->cc * (2 - o->cc) * o->mueff) *
                   (o->xmean[i] - o->xold[i]) / o->sigma;

    /* direction-store management, L/lmcma.cpp:331-366 */
    int imin = 1;
    if (o->itr < nvectors) {
        t[o->itr] = o->itr;
    } else if (nvectors > 1) {
        int dmin = vec[t[1]] - vec[t[0]];
        for (int j = 1; j < nvectors - 1; ++j) {
            int dcur = vec[t[j + 1]] - vec[t[j]];
            if (dcur < dmin) {
                dmin = dcur;
                imin = j + 1;
            }
        }
        if (dmin >= o->maxsteps)
            imin = 0;
        if (imin != nvectors - 1) {
            int sav = t[imin];
            for (int j = imin; j < nvectors - 1; ++j)
                t[j] = t[j + 1];
            t[nvectors - 1] = sav;
        }
    } else {
        imin = 0; /* m = 1: the single slot is always replaced */
    }
    o->iterator_sz = o->itr + 1;
    if (o->iterator_sz > nvectors)
        o->iterator_sz = nvectors;
    for (int i = 0; i < o->iterator_sz; ++i)
        o->iterator[i] = t[i];
    int newidx = t[o->iterator_sz - 1];
    vec[newidx] = o->itr;
    for (int i = 0; i < N; ++i)
        o->pc_arr[newidx * N + i] = o->pc[i];

    /* v-vector recomputation, L/lmcma.cpp:367-390 */
    if (imin == 1)
        imin = 0;
    for (int i = imin; i < o->iterator_sz; ++i) {
        int indx = t[i];
        for (int j = 0; j < N; ++j)
            o->Av[j] = o->pc_arr[indx * N + j];
        inv_az(N, o->Av, i, o->iterator, o->v_arr, o->Lj, o->K);
        for (int j = 0; j < N; ++j)
            o->v_arr[indx * N + j] = o->Av[j];
        double nv = 0;
        for (int j = 0; j < N; ++j)
            nv += o->v_arr[indx * N + j] * o->v_arr[indx * N + j];
        o->Nj[indx] = (sqrt(1 - o->c1) / nv) *
                      (sqrt(1 + (o->c1 / (1 - o->c1)) * nv) - 1);
        o->Lj[indx] = (1 / (sqrt(1 - o->c1) * nv)) *
                      (1 - (1 / sqrt(1 + (o->c1 / (1 - o->c1)) * nv)));
    }

    /* population success rule, L/lmcma.cpp:393-421.  mixed[0..lambda) holds the
     * CURRENT sorted fitness, mixed[lambda..) the previous; the reference's
     * variable names are swapped relative to what they sum - keep the arithmetic. */
    if (o->itr > 0) {
        for (int i = 0; i < lambda; ++i) {
            o->mixed[i] = o->arfitness[i];
            o->mixed[lambda + i] = o->prev_arfitness[i];
        }
        orc_sort(2 * lambda, o->mixed, o->ranks);
        double meanprev = 0, meancur = 0;
        for (int i = 0; i < 2 * lambda; ++i)
            o->ranks_tmp[i] = o->ranks[i];
        for (int i = 0; i < 2 * lambda; ++i)
            o->ranks[o->ranks_tmp[i]] = i;
        for (int i = 0; i < lambda; ++i) {
            meanprev = meanprev + o->ranks[i];
            meancur = meancur + o->ranks[lambda + i];
        }
        meanprev = meanprev / lambda;
        meancur = meancur / lambda;
        double diffv = (meancur - meanprev) / lambda;
        double z1 = diffv - o->val_target;
        o->s = (1 - o->cs) * o->s + o->cs * z1;
        double d_s = 1;
        o->sigma = o->sigma * exp(o->s / d_s);
    }
    for (int i = 0; i < lambda; ++i)
        o->prev_arfitness[i] = o->arfitness[i];
    o->itr++;
}

void orc_lmcma_ask(orc_lmcma *o, double *params)
{
    /* L/lmcma.cpp:172-182 */
    for (int d = 0; d < o->N; ++d)
        params[d] = o->arx[o->sample_idx * o->N + d];
}

void orc_lmcma_tell(orc_lmcma *o, const double *feedbacks, int num_feedbacks)
{
    /* L/lmcma.cpp:184-205 */
    double feedback = 0.0;
    for (int i = 0; i < num_feedbacks; ++i)
        feedback += feedbacks[i];
    o->arfitness[o->sample_idx] = feedback;
    o->counteval++;
    if (feedback < o->best_f || o->counteval == 1)
        o->best_f = feedback;
    if (++o->sample_idx % o->lambda == 0) {
        lmcma_update(o, NULL);
        if (o->z_source != ORC_Z_GIVEN)
            orc_lmcma_sample(o);
        o->sample_idx = 0;
    }
}

void orc_lmcma_tell_all(orc_lmcma *o, const double *fitness)
{
    for (int i = 0; i < o->lambda; ++i)
        orc_lmcma_tell(o, &fitness[i], 1);
}

static void record_fitness(orc_lmcma *o, const double *fitness)
{
    for (int i = 0; i < o->lambda; ++i) {
        double f = fitness[i];
        o->arfitness[i] = f;
        o->counteval++;
        if (f < o->best_f || o->counteval == 1)
            o->best_f = f;
    }
}

void orc_lmcma_update_only(orc_lmcma *o, const double *fitness)
{
    record_fitness(o, fitness);
    lmcma_update(o, NULL);
    o->sample_idx = 0;
}

void orc_lmcma_update_with_mean(orc_lmcma *o, const double *fitness, const double *xmean_new)
{
    record_fitness(o, fitness);
    lmcma_update(o, xmean_new);
    o->sample_idx = 0;
}

void orc_lmcma_set_arx(orc_lmcma *o, const double *rows)
{
    memcpy(o->arx, rows, (size_t)o->lambda * o->N * sizeof(double));
}

int orc_lmcma_done(const orc_lmcma *o) { return o->sigma < 1e-20; } /* L/lmcma.cpp:426-429 */

int orc_lmcma_lambda(const orc_lmcma *o) { return o->lambda; }
int orc_lmcma_mu(const orc_lmcma *o) { return o->mu; }
int orc_lmcma_m(const orc_lmcma *o) { return o->m; }
int orc_lmcma_itr(const orc_lmcma *o) { return o->itr; }
int orc_lmcma_iterator_sz(const orc_lmcma *o) { return o->iterator_sz; }
int orc_lmcma_counteval(const orc_lmcma *o) { return o->counteval; }
double orc_lmcma_bestf(const orc_lmcma *o) { return o->best_f; }
double orc_lmcma_sigma(const orc_lmcma *o) { return o->sigma; }
double orc_lmcma_s(const orc_lmcma *o) { return o->s; }
double orc_lmcma_mueff(const orc_lmcma *o) { return o->mueff; }
double orc_lmcma_c1(const orc_lmcma *o) { return o->c1; }
double orc_lmcma_cc(const orc_lmcma *o) { return o->cc; }
const double *orc_lmcma_arx(const orc_lmcma *o) { return o->arx; }
const double *orc_lmcma_xmean(const orc_lmcma *o) { return o->xmean; }
const double *orc_lmcma_pc(const orc_lmcma *o) { return o->pc; }
const double *orc_lmcma_weights(const orc_lmcma *o) { return o->weights; }
const double *orc_lmcma_v_arr(const orc_lmcma *o) { return o->v_arr; }
const double *orc_lmcma_pc_arr(const orc_lmcma *o) { return o->pc_arr; }
const double *orc_lmcma_nj(const orc_lmcma *o) { return o->Nj; }
const double *orc_lmcma_lj(const orc_lmcma *o) { return o->Lj; }
const int *orc_lmcma_iterator(const orc_lmcma *o) { return o->iterator; }
const int *orc_lmcma_arindex(const orc_lmcma *o) { return o->arindex; }
const double *orc_lmcma_sorted_fitness(const orc_lmcma *o) { return o->prev_arfitness; }

/* ======================================================================= */
/* Cost contract.  The reference never wires a trajectory cost to LM-CMA; the
 * terms are assembled from its planning ingredients:
 *   cell lookup, round-half-even, (row = y, col = x):  S/...path_planner.cpp:591-603
 *   state cost from the looked-up cell:                S/...:605-628, 655-668
 *   path length objective:                             S/...:634-639
 *   weighted sum of objectives:                        S/...:677-690
 *   acceleration stencil and its end truncation:       L/lmcma.cpp:763, 823-832
 *   terms handed to the optimiser summed left to right: L/lmcma.cpp:186-188
 * The integration of a state cost along a motion happens inside OMPL
 * (StateCostIntegralObjective, third party, version not pinned by the
 * reference): the trapezoid form used here is therefore parity-unpinned.    */
/* ======================================================================= */

static double grid_value(const orc_problem *p, const double *q)
{
    int ix = (int)nearbyint(q[0]);
    int iy = (int)nearbyint(q[1]);
    int iz = (p->dims > 2) ? (int)nearbyint(q[2]) : 0;
    if (ix < 0) ix = 0;
    if (ix > p->nx - 1) ix = p->nx - 1;
    if (iy < 0) iy = 0;
    if (iy > p->ny - 1) iy = p->ny - 1;
    if (iz < 0) iz = 0;
    if (iz > p->nz - 1) iz = p->nz - 1;
    long cell = ((long)iz * p->ny + iy) * p->nx + ix;
    if (p->grid_type == ORC_GRID_BITS) {
        const uint32_t *w = (const uint32_t *)p->grid;
        return ((w[cell >> 5] >> (cell & 31)) & 1u) ? p->c_obs : 0.0;
    }
    return (double)((const float *)p->grid)[cell];
}

/* node j of the augmented path: 0 = start, 1..W = waypoints, W+1 = goal */
static void node(const orc_problem *p, const double *row, int j, double *q)
{
    int W = p->waypoints;
    for (int d = 0; d < p->dims; ++d) {
        if (j <= 0) q[d] = p->start[d];
        else if (j >= W + 1) q[d] = p->goal[d];
        else q[d] = row[d * W + (j - 1)];
    }
}

double orc_cost(const orc_problem *p, const double *row, double *terms)
{
    const int W = p->waypoints, D = p->dims;
    double j_len = 0.0, j_grid = 0.0, j_smooth = 0.0;
    double a[3], b[3];
    /* length and trapezoid grid integral over the W+1 segments */
    for (int j = 0; j <= W; ++j) {
        node(p, row, j, a);
        node(p, row, j + 1, b);
        double d2 = 0.0;
        for (int d = 0; d < D; ++d)
            d2 += (b[d] - a[d]) * (b[d] - a[d]);
        double len = sqrt(d2);
        j_len += len;
        j_grid += 0.5 * (grid_value(p, a) + grid_value(p, b)) * len;
    }
    /* squared acceleration at every waypoint */
    static const double st[5] = { -1 / 12.0, 16 / 12.0, -30 / 12.0, 16 / 12.0, -1 / 12.0 };
    for (int d = 0; d < D; ++d)
        for (int k = 0; k < W; ++k) {
            double acc = 0.0;
            for (int o = -2; o <= 2; ++o) {
                int kk = k + o;
                double v;
                if (kk < 0)
                    v = (p->pad_mode == ORC_PAD_ZERO) ? 0.0 : p->start[d];
                else if (kk >= W)
                    v = (p->pad_mode == ORC_PAD_ZERO) ? 0.0 : p->goal[d];
                else
                    v = row[d * W + kk];
                acc += st[o + 2] * v;
            }
            j_smooth += acc * acc;
        }
    if (terms) {
        terms[0] = j_len;
        terms[1] = j_grid;
        terms[2] = j_smooth;
    }
    double fb[3] = { p->w_len * j_len, p->w_grid * j_grid, p->w_smooth * j_smooth };
    double total = 0.0;
    for (int i = 0; i < 3; ++i)
        total += fb[i];
    if (isnan(total))
        total = INFINITY;
    return total;
}

/* the cost as a plain callback (problem, row, length): what a caller's loop around the
 * reference's ask/tell interface evaluates between getNextParameterVector and
 * setEvaluationFeedback (L/example_lmcma.cpp:51-54) */
double orc_cost_callback(void *problem, const double *row, int n)
{
    (void)n;
    return orc_cost((const orc_problem *)problem, row, NULL);
}

void orc_cost_rows_f32(const orc_problem *p, const float *rows, int count,
                       double *cost, double *terms)
{
    int N = p->dims * p->waypoints;
#pragma omp parallel num_threads(g_threads) if (g_threads > 1)
    {
        double *row = (double *)malloc((size_t)N * sizeof(double));
#pragma omp for schedule(static)
        for (int i = 0; i < count; ++i) {
            for (int n = 0; n < N; ++n)
                row[n] = (double)rows[(size_t)i * N + n];
            cost[i] = orc_cost(p, row, terms ? terms + 3 * (size_t)i : NULL);
        }
        free(row);
    }
}

/* The whole generation loop on the host, for the timed CPU baseline: the ask/tell loop of
 * L/example_lmcma.cpp:49-60 with the trajectory cost as the caller's function, the rows
 * rounded to float first (the precision the device stores them in).  The scoring loop runs
 * on orc_set_threads() threads, the sampling too when the normals are counter-based. */
void orc_lmcma_run_trajectory(orc_lmcma *o, const orc_problem *p, int generations)
{
    int N = o->N, lambda = o->lambda;
    double *cost = (double *)malloc((size_t)lambda * sizeof(double));
    for (int g = 0; g < generations; ++g) {
#pragma omp parallel num_threads(g_threads) if (g_threads > 1)
        {
            double *row = (double *)malloc((size_t)N * sizeof(double));
#pragma omp for schedule(static)
            for (int i = 0; i < lambda; ++i) {
                for (int n = 0; n < N; ++n)
                    row[n] = (double)(float)o->arx[(size_t)i * N + n];
                cost[i] = orc_cost(p, row, NULL);
            }
            free(row);
        }
        orc_lmcma_tell_all(o, cost);
    }
    free(cost);
}

double orc_toy_cost(double x, double y)
{
    /* L/example_lmcma.cpp:18-26 */
    return -(4 * exp(-(pow(x - 4, 2) + pow(y - 4, 2))) +
             2 * exp(-(pow(x - 2, 2) + pow(y - 2, 2))));
}

/* ======================================================================= */
/* Grid generators                                                           */
/* ======================================================================= */

void orc_two_bar_map(int size, int nz, uint8_t *occ)
{
    /* S/...path_planner.cpp:269-297 writes EDT_Matrix(a,b)=1 for a<60,b in
     * [65,75) and a in [40,99), b in [35,45); lookups use (row=y, col=x)
     * (:599-602), so a is the row (y) and b the column (x).  Scaled by
     * size/100 with integer arithmetic; in 3-D the bars are extruded over the
     * lower three quarters of z for bar 1 and the upper three quarters for bar 2. */
    memset(occ, 0, (size_t)size * size * (size_t)nz);
    int r0a = 0, r1a = 60 * size / 100, c0a = 65 * size / 100, c1a = 75 * size / 100;
    int r0b = 40 * size / 100, r1b = 99 * size / 100, c0b = 35 * size / 100, c1b = 45 * size / 100;
    for (int z = 0; z < nz; ++z) {
        int in_a = (nz == 1) || (z < (3 * nz) / 4);
        int in_b = (nz == 1) || (z >= nz / 4);
        uint8_t *layer = occ + (size_t)z * size * size;
        if (in_a)
            for (int r = r0a; r < r1a; ++r)
                for (int c = c0a; c < c1a; ++c)
                    layer[(size_t)r * size + c] = 1;
        if (in_b)
            for (int r = r0b; r < r1b; ++r)
                for (int c = c0b; c < c1b; ++c)
                    layer[(size_t)r * size + c] = 1;
    }
}

static uint64_t splitmix64(uint64_t *s)
{
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

void orc_random_boxes(int nx, int ny, int nz, uint64_t seed, double target_fill,
                      const double *start, const double *goal, uint8_t *occ)
{
    /* adds axis-aligned boxes until target_fill of the cells are occupied,
     * never touching a ball of radius 5 % of nx around start and goal */
    long cells = (long)nx * ny * nz;
    long filled = 0;
    for (long i = 0; i < cells; ++i) filled += occ[i] != 0;
    double keep = 0.05 * nx;
    uint64_t s = seed;
    int guard = 0;
    while ((double)filled < target_fill * (double)cells && guard++ < 100000) {
        int bw = 2 + (int)(splitmix64(&s) % (uint64_t)(nx / 16 + 1));
        int bh = 2 + (int)(splitmix64(&s) % (uint64_t)(ny / 16 + 1));
        int bd = (nz > 1) ? 2 + (int)(splitmix64(&s) % (uint64_t)(nz / 16 + 1)) : 1;
        int x0 = (int)(splitmix64(&s) % (uint64_t)nx);
        int y0 = (int)(splitmix64(&s) % (uint64_t)ny);
        int z0 = (nz > 1) ? (int)(splitmix64(&s) % (uint64_t)nz) : 0;
        int x1 = x0 + bw < nx ? x0 + bw : nx;
        int y1 = y0 + bh < ny ? y0 + bh : ny;
        int z1 = z0 + bd < nz ? z0 + bd : nz;
        /* reject boxes that come too close to the endpoints */
        int bad = 0;
        for (int e = 0; e < 2 && !bad; ++e) {
            const double *q = e ? goal : start;
            double cx = fmin(fmax(q[0], (double)x0), (double)(x1 - 1));
            double cy = fmin(fmax(q[1], (double)y0), (double)(y1 - 1));
            double cz = (nz > 1) ? fmin(fmax(q[2], (double)z0), (double)(z1 - 1)) : 0.0;
            double dz = (nz > 1) ? q[2] - cz : 0.0;
            double d2 = (q[0] - cx) * (q[0] - cx) + (q[1] - cy) * (q[1] - cy) + dz * dz;
            if (d2 < keep * keep) bad = 1;
        }
        if (bad) continue;
        for (int z = z0; z < z1; ++z)
            for (int y = y0; y < y1; ++y)
                for (int x = x0; x < x1; ++x) {
                    uint8_t *c = &occ[((size_t)z * ny + y) * nx + x];
                    if (!*c) { *c = 1; ++filled; }
                }
    }
}

void orc_pack_bits(const uint8_t *occ, long cells, uint32_t *words)
{
    long nwords = (cells + 31) / 32;
    memset(words, 0, (size_t)nwords * sizeof(uint32_t));
    for (long i = 0; i < cells; ++i)
        if (occ[i])
            words[i >> 5] |= 1u << (i & 31);
}

/* 8SSEDT, S/...path_planner.cpp:406-490 */
typedef struct { int dx, dy; } orc_pt;
static int pt_d2(orc_pt p) { return p.dx * p.dx + p.dy * p.dy; }

static orc_pt sdf_get(const orc_pt *g, int w, int h, int x, int y)
{
    orc_pt empty = { 9999, 9999 };
    if (x >= 0 && y >= 0 && x < w && y < h)
        return g[y * w + x];
    return empty;
}

static void sdf_compare(const orc_pt *g, int w, int h, orc_pt *p, int x, int y, int ox, int oy)
{
    orc_pt other = sdf_get(g, w, h, x + ox, y + oy);
    other.dx += ox;
    other.dy += oy;
    if (pt_d2(other) < pt_d2(*p))
        *p = other;
}

static void sdf_generate(orc_pt *g, int w, int h)
{
    for (int y = 0; y < h; ++y) {
        for (int x = 0; x < w; ++x) {
            orc_pt p = sdf_get(g, w, h, x, y);
            sdf_compare(g, w, h, &p, x, y, -1, 0);
            sdf_compare(g, w, h, &p, x, y, 0, -1);
            sdf_compare(g, w, h, &p, x, y, -1, -1);
            sdf_compare(g, w, h, &p, x, y, 1, -1);
            g[y * w + x] = p;
        }
        for (int x = w - 1; x >= 0; --x) {
            orc_pt p = sdf_get(g, w, h, x, y);
            sdf_compare(g, w, h, &p, x, y, 1, 0);
            g[y * w + x] = p;
        }
    }
    for (int y = h - 1; y >= 0; --y) {
        for (int x = w - 1; x >= 0; --x) {
            orc_pt p = sdf_get(g, w, h, x, y);
            sdf_compare(g, w, h, &p, x, y, 1, 0);
            sdf_compare(g, w, h, &p, x, y, 0, 1);
            sdf_compare(g, w, h, &p, x, y, -1, 1);
            sdf_compare(g, w, h, &p, x, y, 1, 1);
            g[y * w + x] = p;
        }
        for (int x = 0; x < w; ++x) {
            orc_pt p = sdf_get(g, w, h, x, y);
            sdf_compare(g, w, h, &p, x, y, -1, 0);
            g[y * w + x] = p;
        }
    }
}

void orc_8ssedt(const uint8_t *inside, int width, int height, int *signed_dist,
                double *dist_to_inside)
{
    /* grid1: inside cells are sources; grid2: outside cells are sources
     * (S/...path_planner.cpp:515-523); signed distance = int(d1) - int(d2) (:538-540) */
    size_t n = (size_t)width * height;
    orc_pt *g1 = (orc_pt *)malloc(n * sizeof(orc_pt));
    orc_pt *g2 = (orc_pt *)malloc(n * sizeof(orc_pt));
    const orc_pt in = { 0, 0 }, empty = { 9999, 9999 };
    for (size_t i = 0; i < n; ++i) {
        g1[i] = inside[i] ? in : empty;
        g2[i] = inside[i] ? empty : in;
    }
    sdf_generate(g1, width, height);
    sdf_generate(g2, width, height);
    for (size_t i = 0; i < n; ++i) {
        int d1 = (int)sqrt((double)pt_d2(g1[i]));
        int d2 = (int)sqrt((double)pt_d2(g2[i]));
        if (signed_dist) signed_dist[i] = d1 - d2;
        if (dist_to_inside) dist_to_inside[i] = sqrt((double)pt_d2(g1[i]));
    }
    free(g1);
    free(g2);
}

/* exact EDT by brute-force-free separable lower envelopes (Felzenszwalb &
 * Huttenlocher).  Stands in for dynamicEDT3D (S/...path_planner.cpp:304-307),
 * a third-party library that is not vendored: its values are the exact
 * Euclidean distances in cells, which is what this computes.               */
static void edt_1d(const double *f, int n, double *d, int *v, double *z)
{
    int k = 0;
    v[0] = 0;
    z[0] = -1e300;
    z[1] = 1e300;
    for (int q = 1; q < n; ++q) {
        double s;
        for (;;) {
            int p = v[k];
            s = ((f[q] + (double)q * q) - (f[p] + (double)p * p)) / (2.0 * q - 2.0 * p);
            if (s <= z[k]) --k; else break; /* z[0] = -1e300 is never reached: f is finite */
        }
        ++k;
        v[k] = q;
        z[k] = s;
        z[k + 1] = 1e300;
    }
    k = 0;
    for (int q = 0; q < n; ++q) {
        while (z[k + 1] < q) ++k;
        double dq = (double)q - v[k];
        d[q] = dq * dq + f[v[k]];
    }
}

void orc_edt_exact(const uint8_t *occ, int nx, int ny, int nz, float *dist)
{
    const double BIG = 1e12;
    size_t cells = (size_t)nx * ny * nz;
    double *g = (double *)malloc(cells * sizeof(double));
    for (size_t i = 0; i < cells; ++i) g[i] = occ[i] ? 0.0 : BIG;
    int nmax = nx > ny ? nx : ny;
    if (nz > nmax) nmax = nz;
    double *f = (double *)malloc((size_t)nmax * sizeof(double));
    double *d = (double *)malloc((size_t)nmax * sizeof(double));
    int *v = (int *)malloc((size_t)nmax * sizeof(int));
    double *zz = (double *)malloc((size_t)(nmax + 1) * sizeof(double));
    for (int z = 0; z < nz; ++z)
        for (int y = 0; y < ny; ++y) {
            double *line = g + ((size_t)z * ny + y) * nx;
            memcpy(f, line, (size_t)nx * sizeof(double));
            edt_1d(f, nx, d, v, zz);
            memcpy(line, d, (size_t)nx * sizeof(double));
        }
    for (int z = 0; z < nz; ++z)
        for (int x = 0; x < nx; ++x) {
            for (int y = 0; y < ny; ++y) f[y] = g[((size_t)z * ny + y) * nx + x];
            edt_1d(f, ny, d, v, zz);
            for (int y = 0; y < ny; ++y) g[((size_t)z * ny + y) * nx + x] = d[y];
        }
    if (nz > 1)
        for (int y = 0; y < ny; ++y)
            for (int x = 0; x < nx; ++x) {
                for (int z = 0; z < nz; ++z) f[z] = g[((size_t)z * ny + y) * nx + x];
                edt_1d(f, nz, d, v, zz);
                for (int z = 0; z < nz; ++z) g[((size_t)z * ny + y) * nx + x] = d[z];
            }
    for (size_t i = 0; i < cells; ++i) dist[i] = (float)sqrt(g[i]);
    free(g); free(f); free(d); free(v); free(zz);
}

/* ---- Separable CMA-ES: SepCMA of L/lmcma.hpp:152-190, L/lmcma.cpp:466-577 ------------ */
struct orc_sep {
    int N, lambda, mu, itr;
    double mueff, sigma, cc, c1, cmu, ccov1, ccovmu, chiN, cs, damps;
    double *weights, *xmean, *xold, *arx, *arz, *arfitness, *diagD, *diagC, *pc, *ps, *lo, *hi, *L;
    int *arindex;
    orc_rng rng;
};

static void sep_sample(orc_sep *o)
{
    /* L/lmcma.cpp:519-528 with sampleStandardNormal :212-218 and applyBoundaries :220-230 */
    int N = o->N;
    for (int i = 0; i < o->lambda; ++i) {
        double *z = &o->arz[(size_t)i * N];
        for (int n = 0; n < N; ++n) z[n] = orc_rng_gauss(&o->rng);
        if (o->L) orc_apply_cov_l(o->L, z, N);
        for (int k = 0; k < N; ++k)
            o->arx[(size_t)i * N + k] = o->xmean[k] + o->sigma * o->diagD[k] * z[k];
    }
    for (int i = 0; i < o->lambda; ++i)
        for (int k = 0; k < N; ++k) {
            double *x = &o->arx[(size_t)i * N + k];
            if (o->lo) *x = fmax(*x, o->lo[k]);
            if (o->hi) *x = fmin(*x, o->hi[k]);
        }
}

orc_sep *orc_sep_create(int N, int lambda, const double *initial, const double *lo, const double *hi,
                        double sigma, const double *cov, long seed)
{
    orc_sep *o = (orc_sep *)calloc(1, sizeof(orc_sep));
    o->N = N;
    orc_rng_start(&o->rng, (unsigned long)seed);
    if (lambda < 1) lambda = 4 + (int)(3 * log((double)N));
    o->lambda = lambda;
    o->mu = lambda / 2;
    o->weights = (double *)calloc((size_t)(o->mu > 0 ? o->mu : 1), sizeof(double));
    double sw = 0;
    for (int i = 0; i < o->mu; ++i) {
        o->weights[i] = log((double)o->mu + 0.5) - log((double)(1 + i));
        sw += o->weights[i];
    }
    for (int i = 0; i < o->mu; ++i) {
        o->weights[i] /= sw;
        o->mueff += o->weights[i] * o->weights[i];
    }
    o->mueff = 1.0 / o->mueff;
    o->xmean = (double *)calloc((size_t)N, sizeof(double));
    o->xold = (double *)calloc((size_t)N, sizeof(double));
    if (initial) for (int i = 0; i < N; ++i) o->xmean[i] = initial[i];
    else for (int i = 0; i < N; ++i) o->xmean[i] = orc_rng_uniform(&o->rng);
    o->lo = dup_d(lo, N);
    o->hi = dup_d(hi, N);
    if (cov) {
        o->L = (double *)calloc((size_t)N * N, sizeof(double));
        orc_cholesky(cov, o->L, N);
    }
    o->sigma = sigma;
    o->arx = (double *)calloc((size_t)N * lambda, sizeof(double));
    o->arz = (double *)calloc((size_t)N * lambda, sizeof(double));
    o->arfitness = (double *)calloc((size_t)lambda, sizeof(double));
    o->arindex = (int *)calloc((size_t)lambda, sizeof(int));
    o->diagD = (double *)calloc((size_t)N, sizeof(double));
    o->diagC = (double *)calloc((size_t)N, sizeof(double));
    o->pc = (double *)calloc((size_t)N, sizeof(double));
    o->ps = (double *)calloc((size_t)N, sizeof(double));
    for (int i = 0; i < N; ++i) o->diagD[i] = o->diagC[i] = 1;
    /* L/lmcma.cpp:504-514 */
    o->cc = 1.0 / lambda;
    o->c1 = 2.0 / (pow(N + 1.3, 2.0) + o->mueff);
    o->cmu = fmin(1.0 - o->c1, 2.0 * (o->mueff - 2.0 + 1.0 / o->mueff) / (pow(N + 2.0, 2.0) + o->mueff));
    o->ccov1 = fmin(1.0, o->c1 * (N + 1.5) / 3.0);
    o->ccovmu = fmin(1.0 - o->ccov1, o->cmu * (N + 1.5) / 3);
    o->chiN = sqrt((double)N) * (1.0 - 1.0 / (4.0 * N) + 1 / (21.0 * N * N));
    o->cs = (o->mueff + 2.0) / (N + o->mueff + 5.0);
    o->damps = 1.0 + 2 * fmax(0.0, sqrt((o->mueff - 1.0) / (N + 1.0)) - 1.0) + o->cs;
    sep_sample(o);
    return o;
}

void orc_sep_destroy(orc_sep *o)
{
    if (!o) return;
    free(o->weights); free(o->xmean); free(o->xold); free(o->arx); free(o->arz); free(o->arfitness);
    free(o->diagD); free(o->diagC); free(o->pc); free(o->ps); free(o->lo); free(o->hi); free(o->L);
    free(o->arindex);
    free(o);
}

/* one generation's feedback: SepCMA::update (L/lmcma.cpp:530-577), then sample */
void orc_sep_tell_all(orc_sep *o, const double *fitness)
{
    int N = o->N, mu = o->mu;
    memcpy(o->arfitness, fitness, (size_t)o->lambda * sizeof(double));
    orc_sort(o->lambda, o->arfitness, o->arindex);
    for (int i = 0; i < N; ++i) { o->xold[i] = o->xmean[i]; o->xmean[i] = 0; }
    for (int i = 0; i < mu; ++i) {
        const double *x = &o->arx[(size_t)o->arindex[i] * N];
        for (int j = 0; j < N; ++j) o->xmean[j] += o->weights[i] * x[j];
    }
    double norm_ps = 0;
    for (int i = 0; i < N; ++i) {
        o->ps[i] = (1 - o->cs) * o->ps[i] +
                   sqrt(o->cs * (2 - o->cs) * o->mueff) * (1. / o->diagD[i]) * (o->xmean[i] - o->xold[i]) / o->sigma;
        norm_ps += o->ps[i] * o->ps[i];
    }
    norm_ps = sqrt(norm_ps);
    for (int i = 0; i < N; ++i)
        o->pc[i] = (1 - o->cc) * o->pc[i] + sqrt(o->cc * (2 - o->cc) * o->mueff) * (o->xmean[i] - o->xold[i]) / o->sigma;
    for (int i = 0; i < N; ++i) {
        double val = 0;
        for (int j = 0; j < mu; ++j) {
            double z = o->arz[(size_t)o->arindex[j] * N + i];
            val += o->weights[j] * z * z;
        }
        o->diagC[i] = (1 - o->ccov1 - o->ccovmu) * o->diagC[i] + o->ccov1 * o->pc[i] * o->pc[i] +
                      o->ccovmu * (o->diagC[i] * val);
    }
    o->sigma *= exp((o->cs / o->damps) * (norm_ps / o->chiN - 1));
    for (int i = 0; i < N; ++i) o->diagD[i] = sqrt(o->diagC[i]);
    o->itr++;
    sep_sample(o);
}

void orc_sep_set_arx(orc_sep *o, const double *rows) { memcpy(o->arx, rows, (size_t)o->lambda * o->N * sizeof(double)); }
int orc_sep_lambda(const orc_sep *o) { return o->lambda; }
double orc_sep_sigma(const orc_sep *o) { return o->sigma; }
const double *orc_sep_arx(const orc_sep *o) { return o->arx; }
const double *orc_sep_xmean(const orc_sep *o) { return o->xmean; }
const double *orc_sep_diag(const orc_sep *o) { return o->diagD; }

/* ---- Cholesky-CMA-ES: CMAChol of L/lmcma.hpp:199-239, L/lmcma.cpp:579-754 ------------ */
struct orc_chol {
    int N, lambda, mu, itr;
    double mueff, sigma, cc, c1, chiN, cs, damps;
    double *weights, *xmean, *xold, *arx, *arz, *arfitness, *A, *Ainv, *pc, *ps, *zmean, *tmp1, *tmp2, *lo, *hi, *L;
    int *arindex;
    orc_rng rng;
};

static void chol_sample(orc_chol *o)
{
    /* L/lmcma.cpp:649-664 */
    int N = o->N;
    for (int i = 0; i < o->lambda; ++i) {
        double *z = &o->arz[(size_t)i * N];
        for (int n = 0; n < N; ++n) z[n] = orc_rng_gauss(&o->rng);
        if (o->L) orc_apply_cov_l(o->L, z, N);
        for (int k = 0; k < N; ++k) {
            double Az = 0;
            for (int p = 0; p < N; ++p) Az += o->A[(size_t)k * N + p] * z[p];
            o->arx[(size_t)i * N + k] = o->xmean[k] + o->sigma * Az;
        }
    }
    for (int i = 0; i < o->lambda; ++i)
        for (int k = 0; k < N; ++k) {
            double *x = &o->arx[(size_t)i * N + k];
            if (o->lo) *x = fmax(*x, o->lo[k]);
            if (o->hi) *x = fmin(*x, o->hi[k]);
        }
}

orc_chol *orc_chol_create(int N, int lambda, const double *initial, const double *lo, const double *hi,
                          double sigma, const double *cov, long seed)
{
    orc_chol *o = (orc_chol *)calloc(1, sizeof(orc_chol));
    o->N = N;
    orc_rng_start(&o->rng, (unsigned long)seed);
    if (lambda < 1) lambda = 4 + (int)(3 * log((double)N));
    o->lambda = lambda;
    o->mu = lambda / 2;
    o->weights = (double *)calloc((size_t)(o->mu > 0 ? o->mu : 1), sizeof(double));
    o->xmean = (double *)calloc((size_t)N, sizeof(double));
    o->xold = (double *)calloc((size_t)N, sizeof(double));
    if (initial) for (int i = 0; i < N; ++i) o->xmean[i] = initial[i];
    else for (int i = 0; i < N; ++i) o->xmean[i] = orc_rng_uniform(&o->rng);
    o->lo = dup_d(lo, N);
    o->hi = dup_d(hi, N);
    if (cov) {
        o->L = (double *)calloc((size_t)N * N, sizeof(double));
        orc_cholesky(cov, o->L, N);
    }
    o->sigma = sigma;
    o->arx = (double *)calloc((size_t)N * lambda, sizeof(double));
    o->arz = (double *)calloc((size_t)N * lambda, sizeof(double));
    o->arfitness = (double *)calloc((size_t)lambda, sizeof(double));
    o->arindex = (int *)calloc((size_t)lambda, sizeof(int));
    o->A = (double *)calloc((size_t)N * N, sizeof(double));
    o->Ainv = (double *)calloc((size_t)N * N, sizeof(double));
    o->pc = (double *)calloc((size_t)N, sizeof(double));
    o->ps = (double *)calloc((size_t)N, sizeof(double));
    o->zmean = (double *)calloc((size_t)N, sizeof(double));
    o->tmp1 = (double *)calloc((size_t)N, sizeof(double));
    o->tmp2 = (double *)calloc((size_t)N, sizeof(double));
    /* L/lmcma.cpp:608-645 */
    double sum_weights = 0;
    for (int i = 0; i < o->mu; ++i) sum_weights += log(i + 1.0);
    for (int i = 0; i < o->mu; ++i) {
        o->weights[i] = (log((double)(o->mu + 1.0)) - log((double)(1.0 + i))) / (o->mu * log(o->mu + 1.0) - sum_weights);
        o->mueff += o->weights[i] * o->weights[i];
    }
    o->mueff = 1.0 / o->mueff;
    for (int i = 0; i < N; ++i) o->A[(size_t)i * N + i] = o->Ainv[(size_t)i * N + i] = 1;
    o->cc = 4.0 / (N + 4.0);
    o->c1 = 2.0 / pow(N + sqrt(2.0), 2.0);
    o->chiN = sqrt((double)N) * (1.0 - (1.0 / (4.0 * (double)N)) + (1.0 / (21.0 * (double)(N * N))));
    o->cs = sqrt(o->mueff) / (sqrt((double)N) + sqrt(o->mueff));
    o->damps = 1.0 + 2.0 * fmax(0.0, sqrt((o->mueff - 1.0) / (N + 1.0)) - 1.0) + o->cs;
    chol_sample(o);
    return o;
}

void orc_chol_destroy(orc_chol *o)
{
    if (!o) return;
    free(o->weights); free(o->xmean); free(o->xold); free(o->arx); free(o->arz); free(o->arfitness);
    free(o->A); free(o->Ainv); free(o->pc); free(o->ps); free(o->zmean); free(o->tmp1); free(o->tmp2);
    free(o->lo); free(o->hi); free(o->L); free(o->arindex);
    free(o);
}

static void chol_factor_update(orc_chol *o)
{
    /* CholeskyUpdate, L/lmcma.cpp:703-754 */
    int N = o->N;
    double *v = o->tmp1, *w = o->tmp2;
    const double alpha = 1 - o->c1, sqrtAlpha = sqrt(alpha), gamma = o->c1;
    for (int i = 0; i < N; ++i) {
        v[i] = 0.0;
        for (int j = 0; j < N; ++j) v[i] += o->Ainv[(size_t)i * N + j] * o->pc[j];
    }
    double norm_v2 = 0.0;
    for (int i = 0; i < N; ++i) norm_v2 += v[i] * v[i];
    const double tmp_root = sqrt(1.0 + (gamma / alpha) * norm_v2);
    const double k1 = sqrtAlpha, k2 = sqrtAlpha / norm_v2 * (tmp_root - 1.0);
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) o->A[(size_t)i * N + j] = k1 * o->A[(size_t)i * N + j] + k2 * o->pc[i] * v[j];
    for (int i = 0; i < N; ++i) {
        w[i] = 0.0;
        for (int j = 0; j < N; ++j) w[i] += v[j] * o->Ainv[(size_t)j * N + i];
    }
    const double k3 = 1.0 / sqrtAlpha, k4 = (1.0 - 1.0 / tmp_root) / (sqrtAlpha * norm_v2);
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) o->Ainv[(size_t)i * N + j] = k3 * o->Ainv[(size_t)i * N + j] - k4 * v[i] * w[j];
}

/* one generation's feedback: CMAChol::update (L/lmcma.cpp:666-701), then sample */
void orc_chol_tell_all(orc_chol *o, const double *fitness)
{
    int N = o->N, mu = o->mu;
    memcpy(o->arfitness, fitness, (size_t)o->lambda * sizeof(double));
    orc_sort(o->lambda, o->arfitness, o->arindex);
    for (int i = 0; i < N; ++i) { o->xold[i] = o->xmean[i]; o->xmean[i] = 0; o->zmean[i] = 0; }
    for (int i = 0; i < mu; ++i) {
        const double *x = &o->arx[(size_t)o->arindex[i] * N], *z = &o->arz[(size_t)o->arindex[i] * N];
        for (int j = 0; j < N; ++j) {
            o->xmean[j] += o->weights[i] * x[j];
            o->zmean[j] += o->weights[i] * z[j];
        }
    }
    double norm_ps = 0;
    for (int i = 0; i < N; ++i) {
        o->ps[i] = (1 - o->cs) * o->ps[i] + sqrt(o->cs * (2 - o->cs) * o->mueff) * o->zmean[i];
        norm_ps += o->ps[i] * o->ps[i];
    }
    norm_ps = sqrt(norm_ps);
    for (int i = 0; i < N; ++i)
        o->pc[i] = (1 - o->cc) * o->pc[i] + sqrt(o->cc * (2 - o->cc) * o->mueff) * (o->xmean[i] - o->xold[i]) / o->sigma;
    chol_factor_update(o);
    o->sigma *= exp((o->cs / o->damps) * (norm_ps / o->chiN - 1));
    o->itr++;
    chol_sample(o);
}

void orc_chol_set_arx(orc_chol *o, const double *rows) { memcpy(o->arx, rows, (size_t)o->lambda * o->N * sizeof(double)); }
int orc_chol_lambda(const orc_chol *o) { return o->lambda; }
double orc_chol_sigma(const orc_chol *o) { return o->sigma; }
const double *orc_chol_arx(const orc_chol *o) { return o->arx; }
const double *orc_chol_xmean(const orc_chol *o) { return o->xmean; }
const double *orc_chol_factor(const orc_chol *o) { return o->A; }
