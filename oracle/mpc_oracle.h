/*
 * mpc_oracle.h -- C interface of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  This library is a scalar, Eigen-free CPU
 * restatement of the reference planner (src/MPCPlanner.cpp,
 * include/crawl_planner/eiquadprog.hpp, planning.h, FootScheduler.h).  It is the
 * checker the GPU path is compared against and the timed CPU baseline; nothing
 * in the shipped library links or calls it.
 *
 * All matrices crossing this interface are ROW-major (numpy default); inside,
 * the restatement uses column-major storage like Eigen does.
 *
 * Parity pin status: the reference holds no asserting test and no expected
 * output file.  The oracle is pinned by (1) the ten ZMP samples quoted at
 * tests/unconstrainedQP/testPlanner.cpp:57, (2) the handful of 6-digit values
 * of the two default driver scenarios recorded by the survey, (3) KKT residuals
 * and an independent scipy solve.  Everything else is "parity unpinned".
 */
#ifndef MPC_ORACLE_H_
#define MPC_ORACLE_H_

#ifdef __cplusplus
extern "C" {
#endif

/* Leg order of iit::dog::LegID (leg_data_map.h:19). */
enum { ORC_LF = 0, ORC_RF = 1, ORC_LH = 2, ORC_RH = 3 };

/* Formulations (MPCPlanner.cpp:427, :493, :632). */
enum { ORC_COUPLED_JERK = 0, ORC_COUPLED_VEL = 1, ORC_COUPLED_REF = 2 };

/* Solver status. */
enum { ORC_OK = 0, ORC_INFEASIBLE = 1, ORC_ITER_CAP = 2 };

typedef struct {
    int    N;          /* horizon_size_                                    */
    double Ts;         /* sample time                                      */
    double g;          /* gravity                                          */
    double R, Q, Qs;   /* weight_R, weight_Q, weight_Qs (setWeights)       */
    double height;     /* actual_height argument of the solve* methods     */
    int    dense;      /* 1: multiply through structural zeros exactly as the
                          reference's dense Eigen products do (honest CPU
                          baseline); 0: skip exact-zero multiplicands (same
                          values, several times faster; used by tests)      */
    int    iter_cap;   /* cap on l1 passes and on step-direction (l2a) passes; the
                          reference has none and can cycle on infeasible problems;
                          <=0: none */
} orc_params;

typedef struct {
    double f;          /* value returned by solve_quadprog (inf if infeasible) */
    int    status;
    int    iters;      /* l1 passes (eiquadprog.hpp:240)                   */
    int    n_active;   /* iq at return                                     */
    int    n_add, n_drop, n_l2a;
    double falg;       /* algorithmic flop count of the dense dual active-set
                          method along this instance's pivot trace          */
} orc_info;

/* buildMatrix (MPCPlanner.cpp:101-156).  which: 0=Cx 1=Cv 2=Ca 3=Cz.      */
int orc_build_matrix(const orc_params *p, int which, int size,
                     double *Sx /* size x 3 */, double *Su /* size x size */);

/* computeZMPtrajectory / computeCOMtrajectory (MPCPlanner.cpp:161-220).
 * which as above (3 = ZMP).  n = jerk length.                              */
int orc_trajectory(const orc_params *p, int which, const double x0[3],
                   const double *jerk, int n, double *traj);

/* computeCOMupdate (MPCPlanner.cpp:1155-1158). */
void orc_com_update(const orc_params *p, double state[3], double jerk_sample);

/* makeGaussian (MPCPlanner.cpp:1161-1171). */
void orc_make_gaussian(int length, int mean, int stddev, double *w);

/* solveQP (MPCPlanner.cpp:223-237). */
int orc_solve_qp(const orc_params *p, const double x0[3], const double *zref,
                 double *jerk);

/* solveQPconstraint (MPCPlanner.cpp:240-300).  active: 2N ints (first
 * n_active valid), index < N = lower bound of knot, >= N = upper bound.    */
int orc_solve_box(const orc_params *p, const double x0[3], const double *zmin,
                  const double *zmax, double *jerk, orc_info *info, int *active);

/* The same solve with a record of its decisions (a diagnostic of this restatement; the
 * reference keeps nothing of the kind): one row of ORC_TRACE_COLS doubles per step-direction pass
 * (eiquadprog.hpp:306-446) -- outer iteration, pass, working-set size, candidate row ip, s(ip),
 * runner-up of step 2's choice, t1, runner-up of the ratio test, t2, z.z, z.n, min |r_k| / max |r_k|,
 * blocking row, |d(iq)| / (eps R_norm).  Returns the number of rows written (<= max_rows). */
#define ORC_TRACE_COLS 14
int orc_solve_box_trace(const orc_params *p, const double x0[3], const double *zmin,
                        const double *zmax, double *jerk, orc_info *info, int *active,
                        int max_rows, double *rec /* [max_rows][ORC_TRACE_COLS] */);

/* planning.h:143-149, :182-197, :236-260 */
double orc_point2_is_right_of_line(const double p0[2], const double p1[2],
                                   const double p2[2]);
void orc_ccw_sort(double *pts /* n x 2 */, int n);
void orc_line_coeff(const double p0[2], const double p1[2], int normalize,
                    double pqr[3]);

/* computeSteps (MPCPlanner.cpp:1188-1329) incl. buildPolygonMatrix
 * (:1040-1082).  feet_x/feet_y in LegID order.  com0 == (0,0) disables the
 * step correction like initialCoM.isZero(0) does.  hip may be NULL (zeros).
 * Outputs (any may be NULL): A m x 2N row-major (needs 4*floor(N/steps)*steps
 * rows of capacity), b, fx/fy/swing 4 x N, hold_x/hold_y 4 x steps.
 * Returns m (>=0) or a negative error.                                     */
int orc_compute_steps(int N, int steps, const double speed[2],
                      const double feet_x[4], const double feet_y[4],
                      int swing0, const double com0[2], const double *hip,
                      double *A, double *b, double *fx, double *fy,
                      double *swing, double *hold_x, double *hold_y);

/* computeCentroid (MPCPlanner.cpp:1106-1130). */
void orc_centroid(int N, const double *fx, const double *fy,
                  const double *swing, double *cx, double *cy);

/* getConstraintViolation (MPCPlanner.cpp:1132-1153). */
void orc_constraint_violation(int N, const double *swing, const double *viol_all,
                              double *viol_knot);

/* solveQPConstraintCoupled / ...Ref (MPCPlanner.cpp:427-783).  A is m x 2N
 * row-major.  zrefx/zrefy only for ORC_COUPLED_REF; speed and rw only for
 * VEL and REF (rw == 1000 selects constant velocity weights).
 * viol_all (m) = CI*sol + ci0.  active: m ints.                            */
int orc_solve_coupled(const orc_params *p, int variant, const double x0[3],
                      const double y0[3], const double *zrefx,
                      const double *zrefy, const double *A, const double *b,
                      int m, const double speed[2], int rw, double *jerk_x,
                      double *jerk_y, double *viol_all, orc_info *info,
                      int *active);

/* Generic entry to the restated solver (eiquadprog.hpp:130-447), p = 0
 * equalities.  G n x n, CI n x m (column per constraint), both row-major.  */
int orc_solve_quadprog(int n, int m, const double *G, const double *g0,
                       const double *CI, const double *ci0, double *x,
                       orc_info *info, int *active, int iter_cap);

/* Whole planner call for one batch of coupled instances, threaded with one
 * planner per thread: computeSteps -> (computeCentroid) -> solve ->
 * getConstraintViolation -> ZMP/CoM trajectories.  Per-instance inputs are
 * [B][k] row-major; outputs likewise (any output may be NULL).             */
int orc_coupled_batch(const orc_params *p, int variant, int steps, int rw,
                      int B, const double *x0, const double *y0,
                      const double *speed, const double *feet_x,
                      const double *feet_y, const int *swing0,
                      const double *com0, const double *hip /* 8 or NULL */,
                      double *jerk_x, double *jerk_y,
                      double *com_x, double *com_y, double *zmp_x,
                      double *zmp_y, double *viol_knot, double *cost,
                      int *status, int *n_active, int *active, int active_stride,
                      int *iters, double *falg, int nthreads);

/* Receding-horizon replanning of B rollouts, the loop of
 * tests/replanning/testPlannerXYconstraintMPC.cpp:108-209 with optimizeVelocity on:
 * every replanningWindow = floor(N/steps) samples the feet are taken from knot
 * sampleW of the previous plan, the schedule advances, computeSteps runs from the
 * actual CoM (step correction if use_com_correction), the velocity formulation is
 * solved with the Gaussian weights of that window, and the state is integrated with
 * computeCOMupdate; the Y jerk is disturbed by disturbance[i] from sample
 * disturbance_start on.  An infeasible stage contributes zero jerk.
 * In/out: x0, y0 [B][3], feet_x, feet_y [B][4], swing0 [B].
 * Out (may be NULL): stage_cost/stage_status/stage_nactive [n_stages][B],
 * last-stage jerk_x/jerk_y [B][N]. */
int orc_replanning_batch(const orc_params *p, int steps, int n_stages, int use_com_correction,
                         const double *hip, int B, double *x0, double *y0, const double *speed,
                         double *feet_x, double *feet_y, int *swing0, const double *disturbance,
                         int disturbance_start, double *stage_cost, int *stage_status,
                         int *stage_nactive, double *jerk_x, double *jerk_y, int nthreads);

/* Same for X/Y pairs of box-constrained QPs. limits are [B][N]. */
int orc_box_batch(const orc_params *p, int B, const double *x0, const double *y0,
                  const double *xmin, const double *xmax, const double *ymin,
                  const double *ymax, double *jerk_x, double *jerk_y,
                  double *cost /* [B][2] */, int *status /* [B][2] */,
                  int *n_active /* [B][2] */, int *active /* [B][2][stride] */,
                  int active_stride, int *iters, double *falg, int nthreads);

/* solveQPconstraintSlack (MPCPlanner.cpp:303-424) for B instances of one axis.
   sol: [B][3N] = jerk (N) then slacks (2N); active rows: 0..N-1 lower bound,
   N..2N-1 upper bound, 2N..4N-1 slack sign. */
int orc_slack_batch(const orc_params *p, int B, const double *x0, const double *zmin,
                    const double *zmax, double *sol, double *cost, int *status, int *n_active,
                    int *active, int active_stride, int *iters, double *falg, int nthreads);

#ifdef __cplusplus
}
#endif
#endif
