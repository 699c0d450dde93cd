// Non-interactive equivalents of the reference's drivers, written against the
// MPCPlanner-shaped C++ interface (include/mpc_amd_planner.hpp):
//   coupled   tests/constrainedQP_coupled/testPlannerXYconstraintCoupled.cpp   (defaults)
//   decoupled tests/constrainedQP_decoupled/testPlannerXYconstraintDecoupled.cpp (defaults)
//   qp        tests/unconstrainedQP/testPlanner.cpp                              (defaults)
// Each writes the same "time value" text files into $HOME through saveTraj and prints a
// few values as key=value lines for the test that runs it.
#include <cmath>
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "mpc_amd_planner.hpp"

using namespace mpc_amd;

static int coupled() {
    int horizon_size = 200, number_of_steps = 20;
    double height = 0.5, Ts = 0.1, weight_R = 0.01, weight_Q = 1, weight_Qs = 100;
    VectorXd jerk_x, jerk_y, zmp_x, zmp_y, com_x, com_y, com_xd, com_yd, viol, centroidX, centroidY;
    Vector3d initial_state_x = {0.0, 0.0, 0.0}, initial_state_y = {0.0, -0.0, 0.0};
    MatrixXd A;
    VectorXd b;
    Vector2d userSpeed = {0.15, 0.0};
    MPCPlanner myPlanner(horizon_size, Ts, 9.81);
    LegDataMap<MPCPlanner::footState> feetStates, footHolds;
    LegDataMap<double> initial_feet_x, initial_feet_y;
    initial_feet_x[LF] = initial_state_x[0] + 0.1;
    initial_feet_x[RF] = initial_state_x[0] + 0.2;
    initial_feet_x[LH] = initial_feet_x[LF] - 0.5;
    initial_feet_x[RH] = initial_feet_x[RF] - 0.5;
    initial_feet_y[LF] = initial_state_y[0] + 1.0;
    initial_feet_y[RF] = initial_state_y[0] - 1.0;
    initial_feet_y[LH] = initial_state_y[0] + 1.0;
    initial_feet_y[RH] = initial_state_y[0] - 1.0;
    myPlanner.computeSteps(userSpeed, initial_feet_x, initial_feet_y, number_of_steps, horizon_size, feetStates,
                           footHolds, A, b, myPlanner, LF);
    myPlanner.computeCentroid(feetStates, centroidX, centroidY);
    myPlanner.setWeights(weight_R, weight_Q, weight_Qs);
    myPlanner.solveQPConstraintCoupledRef(height, initial_state_x, initial_state_y, centroidX, centroidY, A, b,
                                          userSpeed, jerk_x, jerk_y);
    viol = myPlanner.getConstraintViolation(feetStates);
    myPlanner.computeZMPtrajectory(initial_state_x, jerk_x, zmp_x);
    myPlanner.computeZMPtrajectory(initial_state_y, jerk_y, zmp_y);
    myPlanner.computeCOMtrajectory(initial_state_x, jerk_x, com_x);
    myPlanner.computeCOMtrajectory(initial_state_y, jerk_y, com_y);
    myPlanner.computeCOMtrajectory(initial_state_x, jerk_x, com_xd, MPCPlanner::VELOCITY);
    myPlanner.computeCOMtrajectory(initial_state_y, jerk_y, com_yd, MPCPlanner::VELOCITY);
    myPlanner.saveTraj("jerk_x.txt", jerk_x);
    myPlanner.saveTraj("jerk_y.txt", jerk_y);
    myPlanner.saveTraj("zmp_x.txt", zmp_x);
    myPlanner.saveTraj("zmp_y.txt", zmp_y);
    myPlanner.saveTraj("com_x.txt", com_x);
    myPlanner.saveTraj("com_y.txt", com_y);
    myPlanner.saveTraj("com_xd.txt", com_xd);
    myPlanner.saveTraj("com_yd.txt", com_yd);
    myPlanner.saveTraj("viol.txt", viol);
    myPlanner.saveTraj("centroidX.txt", centroidX);
    myPlanner.saveTraj("centroidY.txt", centroidY);
    myPlanner.saveTraj("footPosLF.txt", feetStates[LF].x, feetStates[LF].y);
    myPlanner.saveTraj("footPosRF.txt", feetStates[RF].x, feetStates[RF].y);
    myPlanner.saveTraj("footPosLH.txt", feetStates[LH].x, feetStates[LH].y);
    myPlanner.saveTraj("footPosRH.txt", feetStates[RH].x, feetStates[RH].y);
    myPlanner.saveTraj("footHoldsLF.txt", footHolds[LF].x, footHolds[LF].y, 2 * number_of_steps);
    myPlanner.saveTraj("footHoldsRF.txt", footHolds[RF].x, footHolds[RF].y, 2 * number_of_steps);
    myPlanner.saveTraj("footHoldsLH.txt", footHolds[LH].x, footHolds[LH].y, 2 * number_of_steps);
    myPlanner.saveTraj("footHoldsRH.txt", footHolds[RH].x, footHolds[RH].y, 2 * number_of_steps);
    myPlanner.saveTraj("swingLF.txt", feetStates[LF].swing);
    myPlanner.saveTraj("swingRF.txt", feetStates[RF].swing);
    myPlanner.saveTraj("swingLH.txt", feetStates[LH].swing);
    myPlanner.saveTraj("swingRH.txt", feetStates[RH].swing);
    double mn = viol[0];
    for (double v : viol) mn = v < mn ? v : mn;
    std::printf("jerk_x0=%.9g\njerk_x1=%.9g\njerk_x2=%.9g\ncom_x199=%.9g\nmin_margin=%.9g\nn_active=%d\ncost=%.12g\n",
                jerk_x[0], jerk_x[1], jerk_x[2], com_x[199], mn, (int)myPlanner.lastActiveSet().size(),
                myPlanner.lastCost());
    // a ZMP reference that is not constant over the footstep phases (the reference accepts any
    // vector, MPCPlanner.cpp:632-640): the centroid plus a slow lateral sway
    {
        VectorXd swayX = centroidX, swayY = centroidY, jx, jy;
        for (int k = 0; k < horizon_size; k++) swayY[k] += 0.05 * std::sin(0.07 * k);
        myPlanner.solveQPConstraintCoupledRef(height, initial_state_x, initial_state_y, swayX, swayY, A, b, userSpeed,
                                              jx, jy);
        myPlanner.saveTraj("sway_ref_y.txt", swayY, false);
        myPlanner.saveTraj("sway_jerk_x.txt", jx, false);
        myPlanner.saveTraj("sway_jerk_y.txt", jy, false);
        std::printf("sway_cost=%.12g\nsway_n_active=%d\n", myPlanner.lastCost(), (int)myPlanner.lastActiveSet().size());
    }
    // the velocity formulation on the same polygons (constraints bind)
    myPlanner.solveQPConstraintCoupled(height, initial_state_x, initial_state_y, A, b, userSpeed, jerk_x, jerk_y);
    std::printf("vel_n_active=%d\nvel_cost=%.12g\nvel_jerk_x0=%.12g\n", (int)myPlanner.lastActiveSet().size(),
                myPlanner.lastCost(), jerk_x[0]);
    return 0;
}

static int decoupled(bool useSlacks) {
    int horizon_size = 100, number_of_steps = 10;
    double distance = 3.0, height = 0.5, lateral_bound = 0.2, Ts = 0.1, weight_R = 1, weight_Q = 1;
    VectorXd jerk_x, jerk_y, zmp_x, zmp_y, com_x, com_y;
    Vector3d initial_state_x = {0.0, 0.0, 0.0}, initial_state_y = {0.0, -0.0, 0.0};
    MPCPlanner::BoxLimits zmpLimX, zmpLimY;
    double distance_per_step = distance / number_of_steps;
    int step_knots = horizon_size / number_of_steps;
    MPCPlanner myPlanner(horizon_size, Ts, 9.81);
    myPlanner.setWeights(weight_R, weight_Q);
    zmpLimX.resize(horizon_size);
    zmpLimY.resize(horizon_size);
    int start = 0, phase = step_knots / 2;
    double low = 0.0, overlap = 0.1;
    bool goleft = true;
    auto fill = [](VectorXd &v, int from, int len, double val) {
        for (int k = from; k < from + len; k++) v[k] = val;
    };
    for (int i = 0; i < number_of_steps - 1; i++) {
        fill(zmpLimX.max, start, phase, low + distance_per_step * (0.5 + overlap));
        fill(zmpLimX.min, start, phase, low - distance_per_step * overlap);
        fill(zmpLimY.max, start, phase, 1.0);
        fill(zmpLimY.min, start, phase, -1.0);
        start += phase;
        low += distance_per_step / 2;
        fill(zmpLimX.max, start, phase, low + distance_per_step / 2);
        fill(zmpLimX.min, start, phase, low);
        if (goleft) {
            fill(zmpLimY.max, start, phase, 1.0);
            fill(zmpLimY.min, start, phase, lateral_bound);
        } else {
            fill(zmpLimY.max, start, phase, -lateral_bound);
            fill(zmpLimY.min, start, phase, -1.0);
        }
        goleft = !goleft;
        start += phase;
        low += distance_per_step / 2;
    }
    int missing = horizon_size - start;
    fill(zmpLimX.max, start, missing, distance);
    fill(zmpLimX.min, start, missing, low - distance_per_step * overlap);
    fill(zmpLimY.max, start, missing, 1.0);
    fill(zmpLimY.min, start, missing, -1.0);
    int nx, ny;
    if (!useSlacks) {
        myPlanner.solveQPconstraint(height, initial_state_x, zmpLimX, jerk_x);
        nx = (int)myPlanner.lastActiveSet().size();
        myPlanner.solveQPconstraint(height, initial_state_y, zmpLimY, jerk_y);
        ny = (int)myPlanner.lastActiveSet().size();
    } else { // testPlannerXYconstraintDecoupled.cpp:130-137
        weight_R = 1e-06;
        double weight_Qs = 1;
        myPlanner.setWeights(weight_R, weight_Q, weight_Qs);
        myPlanner.solveQPconstraintSlack(height, initial_state_x, zmpLimX, jerk_x);
        nx = (int)myPlanner.lastActiveSet().size();
        myPlanner.solveQPconstraintSlack(height, initial_state_y, zmpLimY, jerk_y);
        ny = (int)myPlanner.lastActiveSet().size();
    }
    myPlanner.computeZMPtrajectory(initial_state_x, jerk_x, zmp_x);
    myPlanner.computeZMPtrajectory(initial_state_y, jerk_y, zmp_y);
    myPlanner.computeCOMtrajectory(initial_state_x, jerk_x, com_x);
    myPlanner.computeCOMtrajectory(initial_state_y, jerk_y, com_y);
    myPlanner.saveTraj("jerk_x.txt", jerk_x);
    myPlanner.saveTraj("jerk_y.txt", jerk_y);
    myPlanner.saveTraj("zmp_x.txt", zmp_x);
    myPlanner.saveTraj("zmp_y.txt", zmp_y);
    myPlanner.saveTraj("min_y.txt", zmpLimY.min);
    myPlanner.saveTraj("max_y.txt", zmpLimY.max);
    double margin_y = 1e300; // distance of the Y ZMP to its nearer bound, minimum over the horizon
    for (int k = 0; k < horizon_size; k++)
        margin_y = std::min(margin_y, std::min(zmp_y[k] - zmpLimY.min[k], zmpLimY.max[k] - zmp_y[k]));
    std::printf("jerk_y0=%.9g\njerk_y1=%.9g\njerk_y2=%.9g\nx_active=%d\ny_active=%d\nmargin_y=%.9g\n", jerk_y[0],
                jerk_y[1], jerk_y[2], nx, ny, margin_y);
    return 0;
}

static int qp() {
    int horizon_size = 100;
    double height = 0.5, Ts = 0.1, weight_R = 1e-06, weight_Q = 1;
    VectorXd jerk, zmp_x, com_x, zmpRef(horizon_size);
    Vector3d initial_state = {0.2, -0.1, 0.0};
    MPCPlanner myPlanner(horizon_size, Ts, 9.81);
    myPlanner.setWeights(weight_R, weight_Q);
    for (int i = 0; i < horizon_size; i++) zmpRef[i] = 0.2 * i / (horizon_size - 1); // setLinSpaced(N, 0, 0.2)
    myPlanner.solveQP(height, initial_state, zmpRef, jerk);
    myPlanner.computeZMPtrajectory(initial_state, jerk, zmp_x);
    myPlanner.computeCOMtrajectory(initial_state, jerk, com_x, MPCPlanner::POSITION);
    myPlanner.saveTraj("zmpRef.txt", zmpRef);
    myPlanner.saveTraj("jerk.txt", jerk);
    myPlanner.saveTraj("zmp.txt", zmp_x);
    myPlanner.saveTraj("com.txt", com_x);
    std::printf("zmp50=%.9g\nzmp99=%.9g\ncom99=%.9g\n", zmp_x[50], zmp_x[99], com_x[99]);
    return 0;
}

// tests/replanning/testPlannerXYconstraintMPC.cpp with its prompts' defaults, except where
// argv overrides them: replanning <planner.ini or -> <disturbance> <experiment_duration>.
// The ini file supplies userSpeed and the two switches.  Hip offsets are set explicitly (the
// reference's driver leaves them uninitialised).
static int replanning(const char *ini_path, double disturbance, int experiment_duration) {
    int horizon_size = 200, number_of_steps = 20;
    double height = 0.5, Ts = 0.1, weight_R = 1e-06, weight_Q = 1;
    VectorXd jerk_x, jerk_y;
    Vector3d initial_state_x = {0.0, 0.0, 0.0}, initial_state_y = {0.0, -0.0, 0.0};
    MatrixXd A;
    VectorXd b;
    FootScheduler replanning_schedule;
    replanning_schedule.setSequence(LF, RH, RF, LH);
    PlannerIni ini;
    if (ini_path && std::strcmp(ini_path, "-")) ini = PlannerIni::read(ini_path);
    const int optimizeVelocityFlag = ini.optimizeVelocity, useComStepCorrection = ini.useComStepCorrection;
    Vector2d userSpeed = {ini.linearSpeedX, ini.linearSpeedY};
    MPCPlanner myPlanner(horizon_size, Ts, 9.81);
    LegDataMap<Vector2d> hips;
    hips[LF] = {0.1, 1.0};
    hips[RF] = {0.2, -1.0};
    hips[LH] = {-0.4, 1.0};
    hips[RH] = {-0.3, -1.0};
    myPlanner.setHipOffsets(hips);
    LegDataMap<MPCPlanner::footState> feetStates, footHolds;
    LegDataMap<double> initial_feet_x, initial_feet_y;
    initial_feet_x[LF] = initial_state_x[0] + 0.1;
    initial_feet_x[RF] = initial_state_x[0] + 0.2;
    initial_feet_x[LH] = initial_feet_x[LF] - 0.5;
    initial_feet_x[RH] = initial_feet_x[RF] - 0.5;
    initial_feet_y[LF] = initial_state_y[0] + 1.0;
    initial_feet_y[RF] = initial_state_y[0] - 1.0;
    initial_feet_y[LH] = initial_state_y[0] + 1.0;
    initial_feet_y[RH] = initial_state_y[0] - 1.0;
    Vector3d actual_state_x = initial_state_x, actual_state_y = initial_state_y;
    const int replanningWindow = horizon_size / number_of_steps;
    int sampleW = 0, replanningStage = 0;
    VectorXd jerk_disturbance(experiment_duration, 0.0);
    for (int k = 4; k < experiment_duration; k++) jerk_disturbance[k] = disturbance;
    for (int sample = 0; sample < experiment_duration; sample++) {
        if ((sample % replanningWindow) == 0) {
            replanningStage++;
            if (sample > 0) {
                for (int leg = 0; leg < 4; leg++) {
                    initial_feet_x[leg] = feetStates[leg].x[sampleW];
                    initial_feet_y[leg] = feetStates[leg].y[sampleW];
                }
                replanning_schedule.next();
            }
            if (useComStepCorrection)
                myPlanner.computeSteps(userSpeed, initial_feet_x, initial_feet_y, number_of_steps, horizon_size,
                                       feetStates, footHolds, A, b, myPlanner, replanning_schedule.getCurrentSwing(),
                                       Vector2d{actual_state_x[0], actual_state_y[0]});
            else
                myPlanner.computeSteps(userSpeed, initial_feet_x, initial_feet_y, number_of_steps, horizon_size,
                                       feetStates, footHolds, A, b, myPlanner, replanning_schedule.getCurrentSwing());
            if (!optimizeVelocityFlag) {
                myPlanner.solveQPConstraintCoupled(height, actual_state_x, actual_state_y, A, b, jerk_x, jerk_y);
            } else {
                weight_R = 0.01;
                myPlanner.setWeights(weight_R, weight_Q);
                myPlanner.solveQPConstraintCoupled(height, actual_state_x, actual_state_y, A, b, userSpeed, jerk_x,
                                                   jerk_y, replanningWindow);
            }
            sampleW = 0;
            std::printf("stage%d_cost=%.12g\nstage%d_active=%d\nstage%d_status=%d\n", replanningStage,
                        myPlanner.lastCost(), replanningStage, (int)myPlanner.lastActiveSet().size(), replanningStage,
                        myPlanner.lastStatus());
        } else {
            sampleW++;
        }
        myPlanner.computeCOMupdate(actual_state_x, jerk_x[sampleW]);
        myPlanner.computeCOMupdate(actual_state_y, jerk_y[sampleW] + jerk_disturbance[sample]);
    }
    std::printf("stages=%d\nx_pos=%.12g\nx_vel=%.12g\nx_acc=%.12g\ny_pos=%.12g\ny_vel=%.12g\ny_acc=%.12g\n",
                replanningStage, actual_state_x[0], actual_state_x[1], actual_state_x[2], actual_state_y[0],
                actual_state_y[1], actual_state_y[2]);
    std::printf("speed_x=%.12g\nspeed_y=%.12g\n", userSpeed[0], userSpeed[1]);
    return 0;
}

// tests/generateGait/generateFootSteps.cpp (time-based gaits: 0 crawl, 1 pace, 2 trot, 3 bound)
// with its prompts' defaults, followed by what that driver leaves commented out: the
// support polygons of the generated stance pattern (buildPolygonMatrix per run of identical
// knots) and a coupled velocity-formulation solve over them.  The rows of A, b are written to
// $HOME/gait_rows.txt as "row knot p q r" for cross-checks.
static int gait(int gaitType, double duty_factor) {
    int horizon_size = 200;
    double Ts = 0.04, cycle_time = 4.0;
    Vector2d userSpeed = {0.05, 0.0};
    MPCPlanner myPlanner(horizon_size, Ts, 9.81);
    GaitSequencer myGaitSchedule;
    myGaitSchedule.setTaskServoRate((int)(1 / Ts));
    double offset_gait[4] = {0.0, 0.25, 0.5, 0.75};
    switch (gaitType) {
    case 0: myGaitSchedule.setSequence(RH, RF, LH, LF); break;
    case 1:
        myGaitSchedule.setSequence(RH, RF, LH, LF);
        offset_gait[1] = 0.0, offset_gait[2] = 0.5, offset_gait[3] = 0.5;
        break;
    case 2:
        myGaitSchedule.setSequence(RH, LF, LH, RF);
        offset_gait[1] = 0.0, offset_gait[2] = 0.5, offset_gait[3] = 0.5;
        break;
    default:
        myGaitSchedule.setSequence(RH, LH, RF, LF);
        offset_gait[1] = 0.0, offset_gait[2] = 0.5, offset_gait[3] = 0.5;
        break;
    }
    myGaitSchedule.setDutyFactor(duty_factor, duty_factor, duty_factor, duty_factor);
    myGaitSchedule.setOffsetAboutCycleStart(offset_gait[0], offset_gait[1], offset_gait[2], offset_gait[3]);
    myGaitSchedule.setTotalCycleDuration(cycle_time);
    LegDataMap<MPCPlanner::footState> feetStates;
    LegDataMap<double> feetValuesX = {0.3, 0.3, -0.3, -0.3}, feetValuesY = {0.3, -0.3, 0.3, -0.3};
    for (int leg = 0; leg < 4; leg++) feetStates[leg].resize(horizon_size);
    LegDataMap<bool> swingingFlag = {false, false, false, false};
    bool detected_switch = false;
    for (int i = 0; i < horizon_size; i++) {
        myGaitSchedule.updateGaitScheduler(swingingFlag, detected_switch);
        const LegDataMap<bool> swing = myGaitSchedule.getSwingLegState();
        for (int leg = 0; leg < 4; leg++) {
            feetStates[leg].x[i] = feetValuesX[leg];
            feetStates[leg].y[i] = feetValuesY[leg];
            feetStates[leg].swing[i] = swing[leg];
        }
        if (detected_switch) {
            for (int leg = 0; leg < 4; leg++)
                if (swing[leg]) {
                    feetValuesX[leg] += cycle_time * duty_factor * userSpeed[0];
                    feetValuesY[leg] += cycle_time * duty_factor * userSpeed[1];
                }
            detected_switch = false;
        }
    }
    myPlanner.saveTraj("swingLF.txt", feetStates[LF].swing, false);
    myPlanner.saveTraj("footPosLF.txt", feetStates[LF].x, feetStates[LF].y, false);
    // polygons: one buildPolygonMatrix call per run of knots with the same stance pattern and feet
    MatrixXd A;
    VectorXd b;
    A.resize(4 * horizon_size, 2 * horizon_size);
    b.assign((size_t)4 * horizon_size, 0.0);
    int nc = 0, start = 0, min_stance = 4;
    auto same = [&](int a, int c) {
        for (int leg = 0; leg < 4; leg++)
            if (feetStates[leg].swing[a] != feetStates[leg].swing[c] || feetStates[leg].x[a] != feetStates[leg].x[c] ||
                feetStates[leg].y[a] != feetStates[leg].y[c])
                return false;
        return true;
    };
    for (int k = 1; k <= horizon_size; k++)
        if (k == horizon_size || !same(start, k)) {
            int stance = 0;
            for (int leg = 0; leg < 4; leg++) stance += feetStates[leg].swing[start] ? 0 : 1;
            min_stance = std::min(min_stance, stance);
            if (stance >= 2) myPlanner.buildPolygonMatrix(feetStates, start, k - start, horizon_size, A, b, nc);
            start = k;
        }
    MatrixXd A2;
    A2.resize(nc, 2 * horizon_size);
    for (int j = 0; j < 2 * horizon_size; j++)
        for (int i = 0; i < nc; i++) A2(i, j) = A(i, j);
    b.resize(nc);
    {
        const std::string path = std::string(std::getenv("HOME") ? std::getenv("HOME") : ".") + "/gait_rows.txt";
        std::FILE *f = std::fopen(path.c_str(), "w");
        for (int r = 0; r < nc; r++)
            for (int k = 0; k < horizon_size; k++)
                if (A2(r, k) != 0.0 || A2(r, k + horizon_size) != 0.0)
                    std::fprintf(f, "%d %d %.17g %.17g %.17g\n", r, k, A2(r, k), A2(r, k + horizon_size), b[r]);
        std::fclose(f);
    }
    VectorXd jerk_x, jerk_y, com_x;
    Vector3d x0 = {0.0, 0.0, 0.0}, y0 = {0.0, 0.0, 0.0};
    myPlanner.setWeights(0.01, 1.0);
    myPlanner.solveQPConstraintCoupled(0.5, x0, y0, A2, b, userSpeed, jerk_x, jerk_y);
    myPlanner.computeCOMtrajectory(x0, jerk_x, com_x);
    std::printf("rows=%d\nmin_stance=%d\nstatus=%d\nn_active=%d\ncost=%.12g\njerk_x0=%.12g\njerk_y0=%.12g\ncom_x_last=%.12g\n", nc,
                min_stance, myPlanner.lastStatus(), (int)myPlanner.lastActiveSet().size(), myPlanner.lastCost(),
                jerk_x[0], jerk_y[0], com_x[horizon_size - 1]);
    std::printf("active_rows=");
    for (int a : myPlanner.lastActiveSet()) std::printf("%d,", a);
    std::printf("\n");
    return 0;
}

int main(int argc, char **argv) {
    try {
        if (argc > 1 && !std::strcmp(argv[1], "coupled")) return coupled();
        if (argc > 1 && !std::strcmp(argv[1], "decoupled")) return decoupled(false);
        if (argc > 1 && !std::strcmp(argv[1], "decoupled_slack")) return decoupled(true);
        if (argc > 1 && !std::strcmp(argv[1], "qp")) return qp();
        if (argc > 1 && !std::strcmp(argv[1], "gait"))
            return gait(argc > 2 ? std::atoi(argv[2]) : 0, argc > 3 ? std::atof(argv[3]) : 0.85);
        if (argc > 1 && !std::strcmp(argv[1], "replanning"))
            return replanning(argc > 2 ? argv[2] : "-", argc > 3 ? std::atof(argv[3]) : 0.0,
                              argc > 4 ? std::atoi(argv[4]) : 60);
    } catch (const std::exception &e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 2;
    }
    std::fprintf(stderr, "usage: %s coupled|decoupled|decoupled_slack|qp|replanning [ini|-] [disturbance] [duration]|gait [type] [duty]\n", argv[0]);
    return 1;
}
