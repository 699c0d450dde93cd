// Compile check of the Eigen overload set (include/mpc_amd_eigen.hpp): the reference's coupled
// driver with its Eigen variables kept.  Built against the real Eigen where one is installed,
// otherwise against tests/cpp/mock_eigen (a few declarations, not Eigen) -- never run.
#include "mpc_amd_eigen.hpp"

#ifndef MPC_AMD_HAVE_EIGEN
#error "no <Eigen/Dense> on the include path"
#endif

using namespace Eigen;
using mpc_amd::LegDataMap;
using mpc_amd::LF;
typedef mpc_amd::eigen::MPCPlanner MPCPlanner;

int main() {
    int horizon_size = 200, number_of_steps = 20;
    double height = 0.5;
    VectorXd jerk_x, jerk_y, zmp_x, com_x, centroidX, centroidY, b, viol;
    MatrixXd A;
    Vector3d initial_state_x, initial_state_y;
    Vector2d userSpeed;
    userSpeed(0) = 0.15;
    MPCPlanner myPlanner(horizon_size, 0.1, 9.81);
    LegDataMap<MPCPlanner::footState> feetStates, footHolds;
    LegDataMap<double> initial_feet_x = {0.1, 0.2, -0.4, -0.3}, initial_feet_y = {1.0, -1.0, 1.0, -1.0};
    myPlanner.computeSteps(userSpeed, initial_feet_x, initial_feet_y, number_of_steps, horizon_size, feetStates,
                           footHolds, A, b, myPlanner, LF);
    myPlanner.computeCentroid(feetStates, centroidX, centroidY);
    myPlanner.setWeights(0.01, 1.0, 100.0);
    myPlanner.solveQPConstraintCoupledRef(height, initial_state_x, initial_state_y, centroidX, centroidY, A, b,
                                          userSpeed, jerk_x, jerk_y);
    myPlanner.solveQPConstraintCoupled(height, initial_state_x, initial_state_y, A, b, userSpeed, jerk_x, jerk_y);
    myPlanner.solveQPConstraintCoupled(height, initial_state_x, initial_state_y, A, b, jerk_x, jerk_y);
    viol = myPlanner.getConstraintViolation(feetStates);
    myPlanner.computeZMPtrajectory(initial_state_x, jerk_x, zmp_x);
    myPlanner.computeCOMtrajectory(initial_state_x, jerk_x, com_x, MPCPlanner::VELOCITY);
    MPCPlanner::BoxLimits lim;
    lim.resize(horizon_size);
    myPlanner.solveQPconstraint(height, initial_state_x, lim, jerk_x);
    myPlanner.solveQPconstraintSlack(height, initial_state_x, lim, jerk_x);
    myPlanner.solveQP(height, initial_state_x, centroidX, jerk_x);
    myPlanner.saveTraj("jerk_x.txt", jerk_x);
    myPlanner.saveTraj("footPosLF.txt", feetStates[LF].x, feetStates[LF].y);
    return 0;
}
