// Eigen overload set of the MPCPlanner-shaped class: the reference's signatures
// (include/crawl_planner/MPCPlanner.h:47-143) with Eigen::Vector3d / VectorXd / MatrixXd
// arguments, for drivers that keep their Eigen variables.  Compiled only where <Eigen/Dense>
// exists; everything converts to the std::vector types of mpc_amd_planner.hpp and back (the
// copies are O(m N) doubles per call, nothing next to a solve).
#ifndef MPC_AMD_EIGEN_HPP_
#define MPC_AMD_EIGEN_HPP_

#if defined(__has_include)
#if __has_include(<Eigen/Dense>)
#define MPC_AMD_HAVE_EIGEN 1
#endif
#endif

#ifdef MPC_AMD_HAVE_EIGEN
#include <Eigen/Dense>

#include "mpc_amd_planner.hpp"

namespace mpc_amd {
namespace eigen {

inline mpc_amd::VectorXd to_std(const Eigen::VectorXd &v) { return mpc_amd::VectorXd(v.data(), v.data() + v.size()); }
inline mpc_amd::Vector3d to_std(const Eigen::Vector3d &v) { return {v(0), v(1), v(2)}; }
inline mpc_amd::Vector2d to_std(const Eigen::Vector2d &v) { return {v(0), v(1)}; }
inline mpc_amd::MatrixXd to_std(const Eigen::MatrixXd &M) { // both column-major
    mpc_amd::MatrixXd m;
    m.resize((int)M.rows(), (int)M.cols());
    std::copy(M.data(), M.data() + M.rows() * M.cols(), m.a.begin());
    return m;
}
inline void to_eigen(const mpc_amd::VectorXd &v, Eigen::VectorXd &out) {
    out.resize((int)v.size());
    for (size_t i = 0; i < v.size(); i++) out(i) = v[i];
}
inline void to_eigen(const mpc_amd::MatrixXd &m, Eigen::MatrixXd &out) {
    out.resize(m.rows(), m.cols());
    std::copy(m.a.begin(), m.a.end(), out.data());
}

class MPCPlanner : public mpc_amd::MPCPlanner {
  public:
    typedef mpc_amd::MPCPlanner Base;
    struct BoxLimits { // MPCPlanner.h:29-34
        Eigen::VectorXd max, min;
        void resize(int size) {
            max.resize(size);
            min.resize(size);
        }
    };
    struct footState { // MPCPlanner.h:35-44
        Eigen::VectorXd x, y, swing;
        void resize(int size) {
            x.resize(size);
            y.resize(size);
            swing.resize(size);
        }
    };
    MPCPlanner(const double horizon_size, const double Ts, const double gravity, int device = 0)
        : Base(horizon_size, Ts, gravity, device) {}

    using Base::computeCOMtrajectory;
    using Base::computeCentroid;
    using Base::computeSteps;
    using Base::computeZMPtrajectory;
    using Base::getConstraintViolation;
    using Base::saveTraj;
    using Base::solveQP;
    using Base::solveQPconstraint;
    using Base::solveQPconstraintSlack;
    using Base::solveQPConstraintCoupled;
    using Base::solveQPConstraintCoupledRef;

    void solveQP(const double actual_height, const Eigen::Vector3d &x0, const Eigen::VectorXd &zmp_ref,
                 Eigen::VectorXd &jerk) {
        mpc_amd::VectorXd j;
        Base::solveQP(actual_height, to_std(x0), to_std(zmp_ref), j);
        to_eigen(j, jerk);
    }
    void solveQPconstraint(const double actual_height, const Eigen::Vector3d &x0, const BoxLimits &zmpLim,
                           Eigen::VectorXd &jerk) {
        mpc_amd::VectorXd j;
        Base::solveQPconstraint(actual_height, to_std(x0), box(zmpLim), j);
        to_eigen(j, jerk);
    }
    void solveQPconstraintSlack(const double actual_height, const Eigen::Vector3d &x0, const BoxLimits &zmpLim,
                                Eigen::VectorXd &jerk) {
        mpc_amd::VectorXd j;
        Base::solveQPconstraintSlack(actual_height, to_std(x0), box(zmpLim), j);
        to_eigen(j, jerk);
    }
    void solveQPConstraintCoupled(const double actual_height, const Eigen::Vector3d &x0, const Eigen::Vector3d &y0,
                                  const Eigen::MatrixXd &A, const Eigen::VectorXd &b, Eigen::VectorXd &jerk_x,
                                  Eigen::VectorXd &jerk_y) {
        mpc_amd::VectorXd jx, jy;
        Base::solveQPConstraintCoupled(actual_height, to_std(x0), to_std(y0), to_std(A), to_std(b), jx, jy);
        to_eigen(jx, jerk_x);
        to_eigen(jy, jerk_y);
    }
    void solveQPConstraintCoupled(const double actual_height, const Eigen::Vector3d &x0, const Eigen::Vector3d &y0,
                                  const Eigen::MatrixXd &A, const Eigen::VectorXd &b, const Eigen::Vector2d &targetSpeed,
                                  Eigen::VectorXd &jerk_x, Eigen::VectorXd &jerk_y, int replanningWindow = 1000) {
        mpc_amd::VectorXd jx, jy;
        Base::solveQPConstraintCoupled(actual_height, to_std(x0), to_std(y0), to_std(A), to_std(b), to_std(targetSpeed),
                                       jx, jy, replanningWindow);
        to_eigen(jx, jerk_x);
        to_eigen(jy, jerk_y);
    }
    void solveQPConstraintCoupledRef(const double actual_height, const Eigen::Vector3d &x0, const Eigen::Vector3d &y0,
                                     const Eigen::VectorXd &zmp_ref_x, const Eigen::VectorXd &zmp_ref_y,
                                     const Eigen::MatrixXd &A, const Eigen::VectorXd &b,
                                     const Eigen::Vector2d &targetSpeed, Eigen::VectorXd &jerk_x,
                                     Eigen::VectorXd &jerk_y, int replanningWindow = 1000) {
        mpc_amd::VectorXd jx, jy;
        Base::solveQPConstraintCoupledRef(actual_height, to_std(x0), to_std(y0), to_std(zmp_ref_x), to_std(zmp_ref_y),
                                          to_std(A), to_std(b), to_std(targetSpeed), jx, jy, replanningWindow);
        to_eigen(jx, jerk_x);
        to_eigen(jy, jerk_y);
    }
    void computeSteps(const Eigen::Vector2d &userSpeed, const LegDataMap<double> &initial_feet_x,
                      const LegDataMap<double> &initial_feet_y, const int number_of_steps, const int horizon_size,
                      LegDataMap<footState> &feetStates, LegDataMap<footState> &footHolds, Eigen::MatrixXd &A,
                      Eigen::VectorXd &b, MPCPlanner &myPlanner, LegID swing_leg_index,
                      Eigen::Vector2d initialCoM = Eigen::Vector2d::Zero()) {
        LegDataMap<Base::footState> fs, fh;
        mpc_amd::MatrixXd a;
        mpc_amd::VectorXd bb;
        Base::computeSteps(to_std(userSpeed), initial_feet_x, initial_feet_y, number_of_steps, horizon_size, fs, fh, a,
                           bb, myPlanner, swing_leg_index, to_std(initialCoM));
        for (int l = 0; l < 4; l++) {
            to_eigen(fs[l].x, feetStates[l].x);
            to_eigen(fs[l].y, feetStates[l].y);
            to_eigen(fs[l].swing, feetStates[l].swing);
            to_eigen(fh[l].x, footHolds[l].x);
            to_eigen(fh[l].y, footHolds[l].y);
            to_eigen(fh[l].swing, footHolds[l].swing);
        }
        to_eigen(a, A);
        to_eigen(bb, b);
    }
    void computeCentroid(const LegDataMap<footState> &feetStates, Eigen::VectorXd &centroidX,
                         Eigen::VectorXd &centroidY) const {
        mpc_amd::VectorXd cx, cy;
        Base::computeCentroid(feet(feetStates), cx, cy);
        to_eigen(cx, centroidX);
        to_eigen(cy, centroidY);
    }
    Eigen::VectorXd getConstraintViolation(const LegDataMap<footState> &feetStates) const {
        Eigen::VectorXd v;
        to_eigen(Base::getConstraintViolation(feet(feetStates)), v);
        return v;
    }
    void computeZMPtrajectory(const Eigen::Vector3d &x0, const Eigen::VectorXd &jerk, Eigen::VectorXd &zmp) const {
        mpc_amd::VectorXd z;
        Base::computeZMPtrajectory(to_std(x0), to_std(jerk), z);
        to_eigen(z, zmp);
    }
    void computeCOMtrajectory(const Eigen::Vector3d &x0, const Eigen::VectorXd &jerk, Eigen::VectorXd &traj,
                              const state_type state = POSITION) const {
        mpc_amd::VectorXd t;
        Base::computeCOMtrajectory(to_std(x0), to_std(jerk), t, state);
        to_eigen(t, traj);
    }
    void saveTraj(const std::string &filename, const Eigen::VectorXd &var, bool verbose = true) const {
        Base::saveTraj(filename, to_std(var), verbose);
    }
    void saveTraj(const std::string &filename, const Eigen::VectorXd &var_x, const Eigen::VectorXd &var_y,
                  bool verbose = true) const {
        Base::saveTraj(filename, to_std(var_x), to_std(var_y), verbose);
    }

  private:
    static Base::BoxLimits box(const BoxLimits &l) {
        Base::BoxLimits o;
        o.max = to_std(l.max);
        o.min = to_std(l.min);
        return o;
    }
    static LegDataMap<Base::footState> feet(const LegDataMap<footState> &f) {
        LegDataMap<Base::footState> o;
        for (int l = 0; l < 4; l++) {
            o[l].x = to_std(f[l].x);
            o[l].y = to_std(f[l].y);
            o[l].swing = to_std(f[l].swing);
        }
        return o;
    }
};

} // namespace eigen
} // namespace mpc_amd
#endif // MPC_AMD_HAVE_EIGEN
#endif // MPC_AMD_EIGEN_HPP_
