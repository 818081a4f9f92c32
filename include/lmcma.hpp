// lmcma.hpp — host-side optimiser API of the MI355X LM-CMA path planner.
//
// Source-compatible with the reference's lmcma_path_planner/src/lmcma.hpp for
// callers of class LMCMA (constructor :131-133, init :135,
// getNextParameterVector :72, setEvaluationFeedback :73,
// isBehaviorLearningDone :138, public counteval / BestF :64-65) and for the
// free functions it declares (RNG :32-36, myqsort :37-38, covariance :248,
// differentiationMatrix :250-251, invert / cholesky / applyCovL :252-254).
// A caller written against the reference header compiles against this one
// unchanged and links liblmcma_hip.so instead of lmcma.cpp; so does an optimiser
// derived from CMABase (:41-81: its protected state, constructor and pure virtual
// sample() / update() are kept).
//
// What runs where: the population work (LMCMA::sample and, with a trajectory
// problem attached, cost evaluation, ranking and recombination) runs on the GPU
// through the C ABI in lmcma_hip.h; the O(m^2 N) state update runs on the host
// in fp64 in the reference's order of operations, or on the device when
// useDeviceUpdate(true) is set.
//
// SepCMA (reference lmcma.hpp:152-190) is provided with the same division of work:
// sampling (and, with a trajectory problem, scoring) on the GPU, its O(mu N) update on the
// host.  CMAChol (reference lmcma.hpp:199-239) likewise: its dense product A z runs on
// the matrix cores, its O(N^2) factor update on the host.
#ifndef LMCMA_AMD_LMCMA_HPP
#define LMCMA_AMD_LMCMA_HPP

#include <stdint.h>

// ---- value/id pair and sort (reference lmcma.hpp:4-8, 37-38) ------------------
typedef struct {
    double value;
    int id;
} sortedvals;

// ---- random number generator (reference lmcma.hpp:10-36) ----------------------
typedef struct {
    long int startseed;
    long int aktseed;
    long int aktrand;
    long int *rgrand;
    short flgstored;
    double hold;
} random_t;

void random_exit(random_t *t);
long random_Start(random_t *t, long unsigned inseed);
long random_init(random_t *t, long unsigned inseed);
double random_Uniform(random_t *t);
double random_Gauss(random_t *t);
int compare(const void *a, const void *b);
void myqsort(int sz, double *arfitness, int *arindex, sortedvals *arr);

// ---- trajectory problem scored on the device (no reference counterpart as a
//      type; the ingredients are cited in lmcma_hip.h) ---------------------------
struct LMCMATrajectoryProblem {
    int dims = 2;                  // 1..3
    int waypoints = 0;             // N = dims * waypoints, dimension-major rows
    int nx = 0, ny = 1, nz = 1;    // grid extent, cells indexed [z][y][x]
    int cell_type = 1;             // LMCMA_GRID_* of lmcma_hip.h (default: bytes)
    const void *cells = nullptr;   // copied by setTrajectoryProblem
    double c_obs = 1.0;            // cost of an occupied cell
    double start[3] = {0, 0, 0};
    double goal[3] = {0, 0, 0};
    double w_len = 1.0, w_grid = 1000.0, w_smooth = 1.0;
    int pad_mode = 0;              // LMCMA_PAD_ENDPOINT
};

// The base class of the reference (lmcma.hpp:41-81) with its protected state, its constructor
// and its pure virtual sample() / update(), so that an optimiser a caller derived from the
// reference's CMABase compiles and runs against this header: init allocates the population and
// the recombination weights (lmcma.cpp:130-170), getNextParameterVector hands out row sampleIdx
// of arx, setEvaluationFeedback sums the feedbacks, keeps counteval / BestF and calls update();
// sample() after every lambda-th call (lmcma.cpp:172-205).  Such a subclass runs on the host like
// the reference's; the classes below keep their populations on the GPU and answer the same two
// calls through the hooks at the end (also through a CMABase pointer).
class CMABase {
protected:
    int N;
    int lambda;
    int mu;
    double mueff;
    double *initialParams;
    double *loBounds;
    double *hiBounds;
    double *arx;
    double *arfitness;
    double *xmean;
    double *xold;
    double *weights;
    double *covariance;
    int inseed;
    bool verbose;
    double *L;
    int itr;
    int sampleIdx;
    random_t ttime;

public:
    int counteval;
    double BestF;

    CMABase(double *initialParams, int lambda, double *loBounds, double *hiBounds, double *covariance, int inseed,
            bool verbose = false);
    virtual ~CMABase();
    virtual void init(int N);
    void getNextParameterVector(double *params, int N);
    void setEvaluationFeedback(double *feedbacks, int numFeedbacks);
    virtual bool isBehaviorLearningDone();

protected:
    virtual void update() = 0;
    virtual void sample() = 0;
    void sampleStandardNormal(double *z);
    void applyBoundaries();
    // Not in the reference: the two non-virtual calls above go through these.  The defaults are
    // the reference's behaviour on arx / arfitness.
    virtual void askHook(double *params, int N);
    virtual void tellHook(double *feedbacks, int numFeedbacks);
};

class LMCMA : public CMABase {
public:
    // Same parameters and defaults as the reference constructor.  The bound and
    // parameter arrays are read, not owned; loBounds / hiBounds must stay valid
    // for the lifetime of the object, as in the reference (lmcma.cpp:109-110).
    LMCMA(double *initialParams, int lambda = 0, double *loBounds = 0, double *hiBounds = 0,
          double sigma = 1.0, double *covariance = 0, int inseed = 0, bool verbose = false);
    ~LMCMA();
    void init(int N);
    bool isBehaviorLearningDone();
    void sample();
    void update();

    // ---- extensions; call before init() ----------------------------------------
    void setDirectionVectors(int m);   // default: lambda, the reference's choice
    void setDevice(int ordinal);       // default 0
    void usePhiloxSampling(bool on);   // default: the reference's sequential RNG on the host
    // default: host fp64 update.  With the device update (needs usePhiloxSampling) the state lives on
    // the device: step() runs whole generations there, and ask/tell still works with the caller's own
    // cost (candidates drawn on the device, the caller's lambda costs ranked there as floats)
    void useDeviceUpdate(bool on);
    // attaches the grid cost; evaluatePopulation()/step() then score on the GPU
    void setTrajectoryProblem(const LMCMATrajectoryProblem &problem);
    // One population sharded over `world` processes, one GPU each (the reference is a single
    // process): this object is shard `rank` and owns candidates [rank lambda / world,
    // (rank + 1) lambda / world) of the constructor's lambda, which must be a multiple of world.
    // Needs usePhiloxSampling, useDeviceUpdate and a trajectory problem; step() then runs
    // lmcma_hip_run_sharded (RCCL all-gathers between the phases) and must be called with the
    // same argument by every shard; the search state is replicated bit for bit.  uniqueId: the
    // 128 bytes one process gets from populationUniqueId and hands to the others by its own
    // means.  population() then fills the rows of the shard's own candidates only.
    void setPopulationShard(int world, int rank, const char *uniqueId);
    static void populationUniqueId(char uniqueId[128]);

    // ---- batched generation step (after init()) ----------------------------------
    // scores every candidate of the current population on the device and feeds
    // the costs back, which runs update() and sample() like the lambda-th
    // setEvaluationFeedback call does.
    void evaluatePopulation();
    // `generations` of them; with the device update they run without a host round trip between them
    void step(int generations = 1);

    // ---- state readers ----------------------------------------------------------------
    int dimension() const;
    int populationSize() const;
    int directionVectors() const;
    int generation() const;
    double stepSize() const;
    void mean(double *xmean_out) const;                 // [N]
    void population(double *rows_out) const;            // [lambda][N]
    void lastCosts(double *costs_out) const;            // [lambda], candidate order
    const char *lastError() const;                      // "" when no call has failed

    struct Impl;

protected:
    void askHook(double *params, int N);       // getNextParameterVector
    void tellHook(double *feedbacks, int numFeedbacks);   // setEvaluationFeedback

private:
    LMCMA(const LMCMA &);
    LMCMA &operator=(const LMCMA &);
    Impl *impl;
};

// Separable CMA-ES (reference lmcma.hpp:152-190, lmcma.cpp:466-577): a diagonal covariance.
// Candidates x = xmean + sigma D z are drawn on the device from normals of the reference's
// sequential generator (z <- L z first when a covariance is given, lmcma.cpp:212-218); the
// update (lmcma.cpp:530-577) runs on the host in fp64 in the reference's order of operations.
class SepCMA : public CMABase {
public:
    SepCMA(double *initialParams, int lambda = 0, double *loBounds = 0, double *hiBounds = 0,
           double sigma = 1.0, double *covariance = 0, int inseed = 0, bool verbose = false);
    ~SepCMA();
    void init(int N);
    void sample();
    void update();

    // ---- extensions, as for LMCMA ---------------------------------------------------
    void setDevice(int ordinal);
    void setTrajectoryProblem(const LMCMATrajectoryProblem &problem);
    void evaluatePopulation();
    void step(int generations = 1) { for (int g = 0; g < generations; ++g) evaluatePopulation(); }
    int dimension() const;
    int populationSize() const;
    int generation() const;
    double stepSize() const;
    void mean(double *xmean_out) const;       // [N]
    void diagonal(double *diagD_out) const;   // [N] square roots of the diagonal covariance
    void population(double *rows_out) const;  // [lambda][N]

    struct Impl;

protected:
    void askHook(double *params, int N);       // getNextParameterVector
    void tellHook(double *feedbacks, int numFeedbacks);   // setEvaluationFeedback

private:
    SepCMA(const SepCMA &);
    SepCMA &operator=(const SepCMA &);
    Impl *impl;
};

// Cholesky-CMA-ES (reference lmcma.hpp:199-239, lmcma.cpp:579-754): a dense factor A of the
// covariance, updated by a rank-one step per generation.  Candidates x = xmean + sigma A z:
// the normals come from the reference's sequential generator, the product A z (lambda N^2
// multiply-adds, lmcma.cpp:649-664) runs on the device; CMAChol::update and CholeskyUpdate
// (lmcma.cpp:666-754) run on the host in fp64 in the reference's order of operations.
class CMAChol : public CMABase {
public:
    CMAChol(double *initialParams, int lambda = 0, double *loBounds = 0, double *hiBounds = 0,
            double sigma = 1.0, double *covariance = 0, int inseed = 0, bool verbose = false);
    ~CMAChol();
    void init(int N);
    void sample();
    void update();

    // ---- extensions, as for LMCMA ---------------------------------------------------
    void setDevice(int ordinal);
    void setTrajectoryProblem(const LMCMATrajectoryProblem &problem);
    void evaluatePopulation();
    void step(int generations = 1) { for (int g = 0; g < generations; ++g) evaluatePopulation(); }
    int dimension() const;
    int populationSize() const;
    int generation() const;
    double stepSize() const;
    void mean(double *xmean_out) const;       // [N]
    void factor(double *A_out) const;         // [N][N] row-major, A A^T = covariance
    void population(double *rows_out) const;  // [lambda][N]

    struct Impl;

protected:
    void askHook(double *params, int N);       // getNextParameterVector
    void tellHook(double *feedbacks, int numFeedbacks);   // setEvaluationFeedback

private:
    CMAChol(const CMAChol &);
    CMAChol &operator=(const CMAChol &);
    void CholeskyUpdate(double *pc, double *Ainv, double *A);
    Impl *impl;
};

// ---- smoothness matrices (reference lmcma.hpp:241-254) -------------------------
void covariance(int num_dims, int num_waypoints, double *covariance);
void differentiationMatrix(int num_time_steps, int order, double dt, double *diff_matrix,
                           int rowLen = -1);
void invert(double *A, double *Ainv, int N);
void cholesky(double *C, double *L, int N);
void applyCovL(double *L, double *z, int N);

#endif  // LMCMA_AMD_LMCMA_HPP
