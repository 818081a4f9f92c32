/*
 * CPU oracle for the LM-CMA population-sample + trajectory-cost + elite-select path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the shipped library links, loads or calls
 * this code; it exists so that tests/, the smoke check and the benchmark's CPU
 * baseline leg have an independent fp64 restatement of the reference algorithm
 * to compare the HIP path against.
 *
 * Parity status: PINNED for the optimiser.  The reference translation unit
 * lmcma_path_planner/src/lmcma.cpp is compiled where it lies by oracle/Makefile
 * (target `ref`: all of the file but its Eigen include, line 6, and the three
 * Eigen-based functions at its end, lines 836-864, which the optimiser path
 * never reaches) into oracle/_ref/, a harness (oracle/ref_harness.cpp) drives it,
 * and tests/golden/make_reference_fixtures.py records its outputs: the uniform
 * and Gaussian streams for four seeds, the constants of LMCMA::init, myqsort,
 * differentiationMatrix, full ask/tell traces (rows, costs, mean, paths,
 * direction store, slot tables, step size, BestF) with and without bounds, the
 * stop rule, and the traces of SepCMA and CMAChol.  tests/test_reference_pins.py
 * checks this restatement against that fixture (and, where oracle/_ref exists,
 * against the compiled reference itself) bit for bit.
 * Still "parity unpinned", because no reference output can exist for them here:
 * covariance() / invert / cholesky / applyCovL (Eigen is not installed), the
 * trajectory cost integration and the distance transform (they live in OMPL and
 * dynamicEDT3D, which the reference does not vendor), and runs with fewer stored
 * direction vectors than candidates (the reference has m = lambda only; the
 * pinned path is m = lambda, m < lambda is the same code with a shorter store).
 *
 * Each function cites the reference lines it follows, as
 *   L/  = lmcma_path_planner/src/
 *   S/  = sample_based_optimisation_based_path_planner/src/
 */
#ifndef LMCMA_ORACLE_H
#define LMCMA_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- A1: random number generator (L/lmcma.hpp:10-24, L/lmcma.cpp:14-82) ---- */
typedef struct {
    long startseed;
    long aktseed;
    long aktrand;
    long rgrand[32];
    short flgstored;
    double hold;
} orc_rng;

long orc_rng_start(orc_rng *t, unsigned long inseed);
double orc_rng_uniform(orc_rng *t);
double orc_rng_gauss(orc_rng *t);

/* ---- counter-based generator shared with the device fast path ---- */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
/* N standard normals for (seed, generation, global candidate index). */
void orc_philox_normals(uint64_t seed, uint32_t generation, uint32_t candidate,
                        int N, double *z);

/* ---- A6: ascending stable sort of (value, id) (L/lmcma.cpp:84-104) ---- */
void orc_sort(int sz, double *values, int *index);

/* ---- A12: smoothness matrices (L/lmcma.cpp:756-864) ---- */
void orc_differentiation_matrix(int num_time_steps, int order, double dt,
                                double *diff_matrix, int row_len);
int orc_covariance(int num_dims, int num_waypoints, double *cov);
int orc_invert(const double *A, double *Ainv, int N);
int orc_cholesky(const double *C, double *L, int N);
void orc_apply_cov_l(const double *L, double *z, int N);

/* ---- A2-A11: the LM-CMA state machine (L/lmcma.cpp:107-463) ---- */
typedef struct orc_lmcma orc_lmcma;

enum { ORC_Z_REFERENCE = 0, ORC_Z_PHILOX = 1, ORC_Z_GIVEN = 2 };

/* lambda < 1 selects 4 + int(3 ln N); m < 1 selects m = lambda (the
 * reference's only choice, L/lmcma.cpp:266).  lo/hi/cov/initial may be NULL
 * with the reference's meaning.  The arrays are copied. */
orc_lmcma *orc_lmcma_create(int N, int lambda, int m, const double *initial,
                            const double *lo, const double *hi, double sigma,
                            const double *cov, long seed, int z_source);
void orc_lmcma_destroy(orc_lmcma *o);
/* ask / tell exactly as the reference's per-candidate API */
void orc_lmcma_ask(orc_lmcma *o, double *params);
void orc_lmcma_tell(orc_lmcma *o, const double *feedbacks, int num_feedbacks);
int orc_lmcma_done(const orc_lmcma *o);
/* batched helpers: overwrite the whole fitness array and run update()+sample() */
void orc_lmcma_tell_all(orc_lmcma *o, const double *fitness);
/* only update(), then sample from caller-supplied Z[lambda][N] (z_source GIVEN) */
void orc_lmcma_update_only(orc_lmcma *o, const double *fitness);
/* the same, with the recombined mean supplied by the caller (a mean summed shard by shard) */
void orc_lmcma_update_with_mean(orc_lmcma *o, const double *fitness, const double *xmean_new);
void orc_lmcma_sample_given(orc_lmcma *o, const double *Z);
void orc_lmcma_sample(orc_lmcma *o);
/* overwrite the population (e.g. with rows another implementation drew) before update_only */
void orc_lmcma_set_arx(orc_lmcma *o, const double *rows);

/* state readers (sizes in comments) */
int orc_lmcma_lambda(const orc_lmcma *o);
int orc_lmcma_mu(const orc_lmcma *o);
int orc_lmcma_m(const orc_lmcma *o);
int orc_lmcma_itr(const orc_lmcma *o);
int orc_lmcma_iterator_sz(const orc_lmcma *o);
int orc_lmcma_counteval(const orc_lmcma *o);
double orc_lmcma_bestf(const orc_lmcma *o);
double orc_lmcma_sigma(const orc_lmcma *o);
double orc_lmcma_s(const orc_lmcma *o);
double orc_lmcma_mueff(const orc_lmcma *o);
double orc_lmcma_c1(const orc_lmcma *o);
double orc_lmcma_cc(const orc_lmcma *o);
const double *orc_lmcma_arx(const orc_lmcma *o);       /* [lambda][N] */
const double *orc_lmcma_xmean(const orc_lmcma *o);     /* [N] */
const double *orc_lmcma_pc(const orc_lmcma *o);        /* [N] */
const double *orc_lmcma_weights(const orc_lmcma *o);   /* [mu] */
const double *orc_lmcma_v_arr(const orc_lmcma *o);     /* [m][N] */
const double *orc_lmcma_pc_arr(const orc_lmcma *o);    /* [m][N] */
const double *orc_lmcma_nj(const orc_lmcma *o);        /* [m] */
const double *orc_lmcma_lj(const orc_lmcma *o);        /* [m] */
const int *orc_lmcma_iterator(const orc_lmcma *o);     /* [m] */
const int *orc_lmcma_arindex(const orc_lmcma *o);      /* [lambda] */
const double *orc_lmcma_sorted_fitness(const orc_lmcma *o); /* [lambda] */

/* ---- cost contract (ingredients: S/...path_planner.cpp:587-706, L/lmcma.cpp:756-834) ---- */
enum { ORC_GRID_BITS = 0, ORC_GRID_F32 = 1 };
enum { ORC_PAD_ENDPOINT = 0, ORC_PAD_ZERO = 1 };

typedef struct {
    int dims;           /* 2 or 3 */
    int waypoints;      /* W; a row holds dims*W values, dimension-major */
    int nx, ny, nz;     /* grid extent; nz = 1 in 2-D */
    int grid_type;      /* ORC_GRID_BITS: uint32 words, LSB first; ORC_GRID_F32 */
    const void *grid;
    double c_obs;       /* value of an occupied cell for ORC_GRID_BITS */
    double start[3];
    double goal[3];
    double w_len, w_grid, w_smooth;
    int pad_mode;
} orc_problem;

/* terms[0..2] = J_len, J_grid, J_smooth (unweighted); returns the weighted total */
double orc_cost(const orc_problem *p, const double *row, double *terms);
double orc_cost_callback(void *problem, const double *row, int n);
void orc_lmcma_run_trajectory(orc_lmcma *o, const orc_problem *p, int generations);
/* host threads for the loops over candidates (timed CPU baseline); default 1 */
void orc_set_threads(int n);
int orc_get_threads(void);
void orc_cost_rows_f32(const orc_problem *p, const float *rows, int count,
                       double *cost, double *terms /* may be NULL, [count][3] */);
double orc_toy_cost(double x, double y);

/* ---- grid generators (S/...path_planner.cpp:269-297, 406-490, 515-540) ---- */
/* occ[ny*nx] bytes (nz layers for 3-D), row = y, col = x */
void orc_two_bar_map(int size, int nz, uint8_t *occ);
void orc_random_boxes(int nx, int ny, int nz, uint64_t seed, double target_fill,
                      const double *start, const double *goal, uint8_t *occ);
void orc_pack_bits(const uint8_t *occ, long cells, uint32_t *words);
/* 8SSEDT two-pass distance transform of a 2-D occupancy image; returns the
 * integer-truncated signed distance dist1-dist2 per cell like the reference. */
void orc_8ssedt(const uint8_t *inside, int width, int height, int *signed_dist,
                double *dist_to_inside);
/* exact Euclidean distance (in cells) to the nearest occupied cell, 2-D or 3-D */
void orc_edt_exact(const uint8_t *occ, int nx, int ny, int nz, float *dist);

/* ---- Separable CMA-ES (SepCMA, L/lmcma.hpp:152-190, L/lmcma.cpp:466-577) ------------- */
typedef struct orc_sep orc_sep;
orc_sep *orc_sep_create(int N, int lambda, const double *initial, const double *lo, const double *hi,
                        double sigma, const double *cov, long seed);
void orc_sep_destroy(orc_sep *o);
void orc_sep_tell_all(orc_sep *o, const double *fitness);
void orc_sep_set_arx(orc_sep *o, const double *rows);
int orc_sep_lambda(const orc_sep *o);
double orc_sep_sigma(const orc_sep *o);
const double *orc_sep_arx(const orc_sep *o);
const double *orc_sep_xmean(const orc_sep *o);
const double *orc_sep_diag(const orc_sep *o);

/* ---- Cholesky-CMA-ES (CMAChol, L/lmcma.hpp:199-239, L/lmcma.cpp:579-754) ------------- */
typedef struct orc_chol orc_chol;
orc_chol *orc_chol_create(int N, int lambda, const double *initial, const double *lo, const double *hi,
                          double sigma, const double *cov, long seed);
void orc_chol_destroy(orc_chol *o);
void orc_chol_tell_all(orc_chol *o, const double *fitness);
void orc_chol_set_arx(orc_chol *o, const double *rows);
int orc_chol_lambda(const orc_chol *o);
double orc_chol_sigma(const orc_chol *o);
const double *orc_chol_arx(const orc_chol *o);
const double *orc_chol_xmean(const orc_chol *o);
const double *orc_chol_factor(const orc_chol *o);

#ifdef __cplusplus
}
#endif
#endif
