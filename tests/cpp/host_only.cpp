// Host-side pieces of the MPCPlanner-shaped interface that need no GPU: footstep and
// polygon assembly (computeSteps, buildPolygonMatrix, computeCentroid), the integrator,
// FootScheduler, Timer, GaitSequencer and the planner.ini reader.  Prints name=value lines
// and whole arrays for the Python tests to compare with the CPU restatement of the reference.
#include <cstdio>
#include <cstring>
#include <fstream>

#include "mpc_amd_planner.hpp"

using namespace mpc_amd;

static void dump(const char *name, const VectorXd &v) {
    std::printf("%s=", name);
    for (double x : v) std::printf("%.17g,", x);
    std::printf("\n");
}

int main(int argc, char **argv) {
    // computeSteps with the replanning driver's arguments: N, steps, speed, swing leg, CoM
    const int N = argc > 1 ? std::atoi(argv[1]) : 60, steps = argc > 2 ? std::atoi(argv[2]) : 6;
    const int swing = argc > 3 ? std::atoi(argv[3]) : 3;
    const bool with_com = argc > 4 && std::atoi(argv[4]) != 0;
    MPCPlanner pl(N, 0.1, 9.81);
    LegDataMap<Vector2d> hips;
    hips[LF] = {0.1, 1.0};
    hips[RF] = {0.2, -1.0};
    hips[LH] = {-0.4, 1.0};
    hips[RH] = {-0.3, -1.0};
    pl.setHipOffsets(hips);
    LegDataMap<double> fx = {0.11, 0.23, -0.41, -0.28}, fy = {1.02, -0.97, 1.01, -1.03};
    LegDataMap<MPCPlanner::footState> feet, holds;
    MatrixXd A;
    VectorXd b;
    Vector2d speed = {0.13, -0.02};
    if (with_com) pl.computeSteps(speed, fx, fy, steps, N, feet, holds, A, b, pl, (LegID)swing, Vector2d{0.02, -0.01});
    else pl.computeSteps(speed, fx, fy, steps, N, feet, holds, A, b, pl, (LegID)swing);
    std::printf("m=%d\n", A.rows());
    dump("b", b);
    VectorXd rowsum(A.rows(), 0.0), rowk(A.rows(), -1.0);
    for (int r = 0; r < A.rows(); r++)
        for (int k = 0; k < N; k++)
            if (A(r, k) != 0.0 || A(r, k + N) != 0.0) {
                rowsum[r] = A(r, k) + 2.0 * A(r, k + N);
                rowk[r] = k;
            }
    dump("row_p_plus_2q", rowsum);
    dump("row_knot", rowk);
    dump("lf_x", feet[LF].x);
    dump("rh_y", feet[RH].y);
    dump("rf_swing", feet[RF].swing);
    dump("hold_lh_x", holds[LH].x);
    VectorXd cx, cy;
    pl.computeCentroid(feet, cx, cy);
    dump("centroid_x", cx);
    dump("centroid_y", cy);
    // integrator: ten samples of unit jerk from (0.1, -0.1, 0), the reference's own check values
    MPCPlanner small(10, 0.1, 9.81);
    VectorXd ones(10, 1.0), zmp;
    small.computeZMPtrajectory({0.1, -0.1, 0.0}, ones, zmp);
    dump("zmp_kat", zmp);
    dump("gauss", pl.makeGaussian(8, 3, 2));
    // schedule objects
    FootScheduler fs;
    fs.setSequence(LF, RH, RF, LH);
    fs.setCurrentSwing(RF);
    const int s_cur = fs.getCurrentSwing(), s_next = fs.getNextSwing(), s_prev = fs.getPreviousSwing(),
              s_opp = fs.getOppositeSwing();
    fs.next();
    fs.next();
    std::printf("sched=%d%d%d%d%d\n", s_cur, s_next, s_prev, s_opp, (int)fs.getCurrentSwing());
    Timer t;
    t.setDuration(0.25);
    const bool before = t.isTimeElapsed(5.0);
    t.startTimer(1.0);
    std::printf("timer=%d%d%d\nmissing=%.6g\n", (int)before, (int)t.isTimeElapsed(1.2), (int)t.isTimeElapsed(1.3),
                t.getMissingTime(1.1));
    // crawl, duty 0.75, offsets a quarter cycle apart: one leg in the air at a time
    GaitSequencer gs;
    gs.setTaskServoRate(25);
    gs.setSequence(RH, RF, LH, LF);
    gs.setDutyFactor(0.75, 0.75, 0.75, 0.75);
    gs.setOffsetAboutCycleStart(0.0, 0.25, 0.5, 0.75);
    gs.setTotalCycleDuration(4.0);
    LegDataMap<bool> prep = {false, false, false, false};
    bool sw = false;
    int max_air = 0, liftoffs = 0, air_samples[4] = {0, 0, 0, 0};
    for (int i = 0; i < 400; i++) { // four cycles of 100 samples
        gs.updateGaitScheduler(prep, sw);
        if (sw) liftoffs++;
        sw = false;
        const LegDataMap<bool> st = gs.getSwingLegState();
        int air = 0;
        for (int l = 0; l < 4; l++) {
            air += st[l];
            air_samples[l] += st[l];
        }
        if (air > max_air) max_air = air;
    }
    std::printf("gait_max_air=%d\ngait_liftoffs=%d\ngait_air=%d,%d,%d,%d\nswing_duration=%.6g\n", max_air, liftoffs,
                air_samples[0], air_samples[1], air_samples[2], air_samples[3], gs.getSwingDuration(LH));
    // ini reader
    if (argc > 5) {
        const PlannerIni ini = PlannerIni::read(argv[5]);
        std::printf("ini=%d,%.6g,%.6g,%d,%d\n", (int)ini.loaded, ini.linearSpeedX, ini.linearSpeedY,
                    ini.useComStepCorrection, ini.optimizeVelocity);
    }
    return 0;
}
