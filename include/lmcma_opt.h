/*
 * lmcma_opt.h — C bindings of the host-side optimiser class (include/lmcma.hpp)
 * and of the host helpers around it, for callers that cannot include C++
 * (the Python package binds these with ctypes).
 *
 * Every function mirrors one member of the reference's class LMCMA
 * (lmcma_path_planner/src/lmcma.hpp:89-144, cited as L/lmcma.hpp) or one of its
 * free functions; the mapping is given per function.  Return value: 0 on
 * success, negative on failure with the message in lmcma_opt_last_error().
 */
#ifndef LMCMA_OPT_H
#define LMCMA_OPT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lmcma_opt lmcma_opt;

const char *lmcma_opt_last_error(void);

/* LMCMA::LMCMA (L/lmcma.hpp:131-133).  initial/lo/hi/cov may be NULL with the
 * reference's meaning; the arrays (n values, cov n*n) are copied and kept alive
 * by the handle.  lambda < 1: 4 + int(3 ln N); seed < 1: clock-derived.        */
int lmcma_opt_create(const double *initial, const double *lo, const double *hi, const double *cov,
                     int n, int lambda, double sigma, int seed, int verbose, lmcma_opt **out);
/* SepCMA::SepCMA (L/lmcma.hpp:182-184): the same arguments; the handle then drives the
 * separable CMA-ES through the functions below (set_options: only `device` applies).  */
int lmcma_opt_create_sep(const double *initial, const double *lo, const double *hi, const double *cov,
                         int n, int lambda, double sigma, int seed, int verbose, lmcma_opt **out);
/* CMAChol::CMAChol (L/lmcma.hpp:231-233), likewise */
int lmcma_opt_create_chol(const double *initial, const double *lo, const double *hi, const double *cov,
                          int n, int lambda, double sigma, int seed, int verbose, lmcma_opt **out);
int lmcma_opt_destroy(lmcma_opt *o);

/* extensions of include/lmcma.hpp; call before lmcma_opt_init.  m < 1: lambda */
int lmcma_opt_set_options(lmcma_opt *o, int m, int device, int philox, int device_update);
/* LMCMA::setPopulationShard of include/lmcma.hpp: this handle is shard `rank` of a population
 * sharded over `world` processes; unique_id: 128 bytes from lmcma_hip_comm_unique_id      */
int lmcma_opt_set_shard(lmcma_opt *o, int world, int rank, const char *unique_id);
int lmcma_opt_set_problem(lmcma_opt *o, int dims, int waypoints, int nx, int ny, int nz,
                          int cell_type, const void *cells, double c_obs, const double *start,
                          const double *goal, double w_len, double w_grid, double w_smooth,
                          int pad_mode);

/* LMCMA::init (L/lmcma.hpp:135) */
int lmcma_opt_init(lmcma_opt *o, int N);
/* CMABase::getNextParameterVector / setEvaluationFeedback (L/lmcma.hpp:72-73) */
int lmcma_opt_ask(lmcma_opt *o, double *params, int N);
int lmcma_opt_tell(lmcma_opt *o, const double *feedbacks, int num_feedbacks);
/* LMCMA::isBehaviorLearningDone (L/lmcma.hpp:138): 1 = done, 0 = not, <0 error */
int lmcma_opt_done(lmcma_opt *o);
/* batched generations on the device (LMCMA::step of include/lmcma.hpp) */
int lmcma_opt_step(lmcma_opt *o, int generations);

typedef struct {
    int32_t n, lambda, m, generation;
    int64_t counteval;      /* L/lmcma.hpp:64 */
    double best_f;          /* L/lmcma.hpp:65 */
    double sigma;
} lmcma_opt_info;
int lmcma_opt_get_info(lmcma_opt *o, lmcma_opt_info *info);
int lmcma_opt_get_mean(lmcma_opt *o, double *xmean);
int lmcma_opt_get_population(lmcma_opt *o, double *rows);
int lmcma_opt_get_last_costs(lmcma_opt *o, double *costs);
/* SepCMA handles: diagD (L/lmcma.hpp:158), the square roots of the diagonal covariance */
int lmcma_opt_get_diagonal(lmcma_opt *o, double *diag_d);
/* CMAChol handles: the factor A (L/lmcma.hpp:202), [N][N] row-major */
int lmcma_opt_get_factor(lmcma_opt *o, double *a_rowmajor);

/* ---- host helpers ---------------------------------------------------------------- */

/* n draws from random_Gauss / random_Uniform after random_init(seed)
 * (L/lmcma.hpp:32-36) */
int lmcma_host_gauss_stream(int seed, int n, double *out);
int lmcma_host_uniform_stream(int seed, int n, double *out);
/* covariance / differentiationMatrix (L/lmcma.hpp:248-251) */
int lmcma_host_covariance(int num_dims, int num_waypoints, double *cov);
int lmcma_host_differentiation_matrix(int num_time_steps, int order, double dt, double *diff_matrix,
                                      int row_len);
/* cholesky + applyCovL (L/lmcma.hpp:253-254): z <- chol(C) z */
int lmcma_host_cholesky(const double *C, double *L, int N);
int lmcma_host_apply_cov_l(const double *L, double *z, int N);
/* myqsort (L/lmcma.hpp:38): values sorted in place, index filled */
int lmcma_host_sort(int n, double *values, int32_t *index);

/* ---- synthetic grids (host) -------------------------------------------------------
 * S/ = sample_based_optimisation_based_path_planner/src/
 *      sample_based_optimisation_based_path_planner.cpp                            */

/* the two-bar occupancy map of twodMapExample (S/:269-297) scaled from 100 to
 * `size` cells per side; nz > 1 extrudes the bars in z.  occ: [nz][size][size]
 * bytes, row = y, col = x (S/:599-602).                                           */
int lmcma_grid_two_bar(int size, int nz, uint8_t *occ);
/* adds random axis-aligned boxes until `fill` of the cells are occupied, keeping
 * the neighbourhoods of start and goal free                                        */
int lmcma_grid_random_boxes(int nx, int ny, int nz, uint64_t seed, double fill, const double *start,
                            const double *goal, uint8_t *occ);
/* exact Euclidean distance (cells) to the nearest occupied cell: the quantity
 * dynamicEDT3D returns from getDistance (S/:304-307, 319)                          */
int lmcma_grid_edt(const uint8_t *occ, int nx, int ny, int nz, float *dist);
/* the two-pass 8SSEDT of GenerateSDF (S/:447-490) on a 2-D image:
 * signed_dist = int(d_inside) - int(d_outside) (S/:538-540); dist_to_inside = the
 * unsigned distance to the nearest marked cell                                     */
int lmcma_grid_8ssedt(const uint8_t *inside, int width, int height, int32_t *signed_dist,
                      double *dist_to_inside);
/* state cost 1 / max(distance, d_min) per cell (S/:667 with the distance clamped) */
int lmcma_grid_state_cost(const float *dist, long cells, double d_min, float *cost);

/* ---- map files (host) ---------------------------------------------------------------
 * A binvox voxel file as the reference's converter reads it (binvox2bt.cpp:164-255 in
 * the same directory as S/): header keywords dim / translate / scale / data, then
 * (value, count) byte pairs, voxel i at x = i / (w h), z = (i / w) % h, y = i % w.
 * dims receives the extents along x, y, z; occ (NULL to read the header only, else
 * at least dims[0] dims[1] dims[2] bytes, `capacity` of them available) receives the
 * grid as [z][y][x].  Returns the number of occupied voxels, or < 0: -2 cannot open,
 * -3 malformed header, -4 capacity too small, -5 truncated data.                     */
int lmcma_grid_read_binvox(const char *path, int dims[3], double translate[3], double *scale, uint8_t *occ,
                           long capacity);
/* An uncompressed BMP image (8-bit palette, 24- or 32-bit pixels) thresholded like
 * Signed_Distance_Fields_test (S/:507-523): a pixel is marked when green < 128.
 * marked (NULL for the size only) is [height][width], first row = top of the image.
 * Returns the number of marked pixels, or < 0 as above.                              */
int lmcma_grid_read_bmp(const char *path, int *width, int *height, uint8_t *marked, long capacity);

#ifdef __cplusplus
}
#endif
#endif /* LMCMA_OPT_H */
