// mpc_amd_planner.hpp -- header-only C++ host side with MPCPlanner's public interface
// (include/crawl_planner/MPCPlanner.h:24-143 of the reference), on top of the C ABI in
// mpc_amd.h.  Same method names, argument order and meaning; Eigen types are replaced by
// the small std::vector-based types below (the reference's Eigen objects convert by
// copying their data()).  Every solve* call runs on the GPU as a batch of one; callers
// that have many independent plans should use the batch entry points of mpc_amd.h.
//
// Differences from the reference, all deliberate:
//   * infeasible problem: the reference prints a line and leaves the jerk vectors
//     unassigned; here they are zero-filled, lastStatus() says why;
//   * hip offsets default to zero instead of uninitialised memory;
//   * the cost and the working set, which the reference computes and drops, are available
//     through lastCost() / lastActiveSet().
#ifndef MPC_AMD_PLANNER_HPP_
#define MPC_AMD_PLANNER_HPP_

#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <map>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

#include "mpc_amd.h"

namespace mpc_amd {

enum LegID { LF = 0, RF = 1, LH = 2, RH = 3 }; // iit::dog::LegID (leg_data_map.h:19)

typedef std::vector<double> VectorXd;
typedef std::array<double, 3> Vector3d;
typedef std::array<double, 2> Vector2d;
template <class T> using LegDataMap = std::array<T, 4>;

// Dense column-major matrix, the layout of Eigen::MatrixXd.
struct MatrixXd {
    int r = 0, c = 0;
    std::vector<double> a;
    void resize(int rows, int cols) {
        r = rows;
        c = cols;
        a.assign((size_t)rows * cols, 0.0);
    }
    int rows() const { return r; }
    int cols() const { return c; }
    double &operator()(int i, int j) { return a[(size_t)j * r + i]; }
    double operator()(int i, int j) const { return a[(size_t)j * r + i]; }
};

// The four-slot cyclic swing schedule of FootScheduler.h:15-99.
class FootScheduler {
  public:
    FootScheduler() { setSequence(LF, RH, RF, LH); }
    FootScheduler(LegID leg1, LegID leg2, LegID leg3, LegID leg4) { setSequence(leg1, leg2, leg3, leg4); }
    void setSequence(LegID leg1, LegID leg2, LegID leg3, LegID leg4) {
        seq_ = {leg1, leg2, leg3, leg4};
        cur_ = 0;
    }
    LegID getCurrentSwing() const { return seq_[cur_]; }
    void setCurrentSwing(LegID start) { // stays put if the leg is not in the sequence
        for (int i = 0; i < 4; i++)
            if (seq_[i] == start) {
                cur_ = i;
                break;
            }
    }
    LegID getOppositeSwing() const {
        switch (getCurrentSwing()) {
        case LF: return RH;
        case RF: return LH;
        case LH: return RF;
        default: return LF;
        }
    }
    LegID getNextSwing() const { return seq_[(cur_ + 1) % 4]; }
    LegID getPreviousSwing() const { return seq_[(cur_ + 3) % 4]; }
    void next() { cur_ = (cur_ + 1) % 4; }

  private:
    std::array<LegID, 4> seq_;
    int cur_ = 0;
};

// One-shot timer on an externally supplied clock (timer.h:11-75).
class Timer {
  public:
    void startTimer(double startTime) {
        T0_ = startTime;
        resetFlag = false;
    }
    void setDuration(double dur) { duration_ = dur; }
    double getDuration() const { return duration_; }
    double getMissingTime(double time) const {
        if (resetFlag) return duration_;
        return (time - T0_) <= duration_ ? duration_ - (time - T0_) : 0.0;
    }
    double getElapsedTime(double time) const { return time - T0_; }
    bool isTimeElapsed(double time) const { return resetFlag || (time - T0_) > duration_; }
    bool changeDuration(double time, double newDuration) {
        if (newDuration > getElapsedTime(time)) {
            duration_ = newDuration;
            return true;
        }
        return false;
    }
    void resetTimer() { resetFlag = true; }
    bool resetFlag = true;

  private:
    double T0_ = 0.0, duration_ = 0.0;
};

// Time-based gait schedule (GaitSequencer.h:18-222): a stride phase in [0, 1) advances by
// dt / cycleDuration per update; the leg in sequence slot i lifts off when the phase passes
// offset[i] and stays in swing for (1 - dutyFactor[i]) of a cycle.
class GaitSequencer {
  public:
    GaitSequencer() { speed_ = 1.0 / cycleDuration_; }
    GaitSequencer(LegID leg1, LegID leg2, LegID leg3, LegID leg4, int taskServoRate) {
        setSequence(leg1, leg2, leg3, leg4);
        dt_ = 1.0 / (double)taskServoRate;
        dutyFactor_ = {0.75, 0.75, 0.75, 0.75};
        speed_ = 1.0 / cycleDuration_;
    }
    void setTaskServoRate(int taskServoRate) { dt_ = 1.0 / (double)taskServoRate; }
    void setSequence(LegID leg1, LegID leg2, LegID leg3, LegID leg4) { seq_ = {leg1, leg2, leg3, leg4}; }
    void setOffsetAboutCycleStart(double o0, double o1, double o2, double o3) { offset_ = {o0, o1, o2, o3}; }
    void setTotalCycleDuration(double totalCycleTime) {
        cycleDuration_ = totalCycleTime;
        speed_ = std::fabs(cycleDuration_) < 0.001 ? 0.0 : 1.0 / cycleDuration_;
        for (int i = 0; i < 4; i++) timer_[i].setDuration(1.0 - dutyFactor_[i]);
        computeSwingDuration(cycleDuration_);
    }
    void setDutyFactor(double d0, double d1, double d2, double d3) {
        dutyFactor_ = {d0, d1, d2, d3};
        for (int i = 0; i < 4; i++) {
            timer_[i].setDuration(1.0 - dutyFactor_[i]);
            timer_[i].resetTimer();
        }
        computeSwingDuration(cycleDuration_);
    }
    void computeSwingDuration(double totalCycleTime) {
        for (int i = 0; i < 4; i++) swingDuration_[i] = totalCycleTime * (1.0 - dutyFactor_[i]);
    }
    // once per control step; prepareSwinging[leg] is raised on the lift-off edge of that leg
    void updateGaitScheduler(LegDataMap<bool> &prepareSwinging, bool &detected_switch) {
        stride_ += speed_ * dt_;
        strideNotReset_ += speed_ * dt_;
        updateSwingStatus(strideNotReset_, stride_);
        if (stride_ >= (1.0 + speed_ * dt_)) { // the extra dt keeps the last leg of the cycle
            stride_ = 0.0;
            waitForCycleReset_ = {false, false, false, false};
        }
        for (int i = 0; i < 4; i++) {
            const bool rising = (swing_[i] != swingOld_[i]) && swing_[i];
            if (rising) {
                prepareSwinging[seq_[i]] = true;
                detected_switch = true;
            }
        }
        swingOld_ = swing_;
    }
    LegDataMap<bool> getSwingLegState() const {
        LegDataMap<bool> status = {false, false, false, false};
        for (int i = 0; i < 4; i++) status[seq_[i]] = swing_[i];
        return status;
    }
    double getSwingDuration(LegID leg) const {
        for (int i = 0; i < 4; i++)
            if (seq_[i] == leg) return swingDuration_[i];
        return swingDuration_[0];
    }
    double getStrideParametrization() const { return stride_; }

  private:
    void updateSwingStatus(double s_notReset, double s) {
        for (int i = 0; i < 4; i++) {
            if (timer_[i].resetFlag) {
                if (s > offset_[i] && !waitForCycleReset_[i]) {
                    timer_[i].startTimer(s_notReset);
                    swing_[i] = true;
                }
            } else if (timer_[i].isTimeElapsed(s_notReset)) {
                swing_[i] = false;
                waitForCycleReset_[i] = true;
                timer_[i].resetTimer();
            }
        }
    }
    double speed_ = 0.25, stride_ = 0.0, strideNotReset_ = 0.0, cycleDuration_ = 4.0, dt_ = 1.0 / 250.0;
    std::array<double, 4> dutyFactor_ = {0.75, 0.75, 0.75, 0.75}, offset_ = {0, 0, 0, 0}, swingDuration_ = {0, 0, 0, 0};
    std::array<LegID, 4> seq_ = {LF, RF, LH, RH};
    std::array<Timer, 4> timer_;
    std::array<bool, 4> swing_ = {false, false, false, false}, swingOld_ = {false, false, false, false},
                        waitForCycleReset_ = {false, false, false, false};
};

// The planner-related keys of the reference's config/planner.ini: [Crawl] linearSpeedX,
// linearSpeedY (the per-step displacement / velocity target handed to computeSteps and the
// velocity formulation) and [Replanning] useComStepCorrection, optimizeVelocity (the two
// switches of the replanning driver).  Unknown sections and keys are ignored, '#' starts a
// comment, a missing file or key keeps the default.
struct PlannerIni {
    double linearSpeedX = 0.15, linearSpeedY = 0.0;
    int useComStepCorrection = 1, optimizeVelocity = 1;
    bool loaded = false;

    static PlannerIni read(const std::string &path) {
        PlannerIni ini;
        std::ifstream f(path.c_str());
        if (!f) return ini;
        ini.loaded = true;
        std::string line, section;
        auto trim = [](std::string t) {
            const char *ws = " \t\r\n";
            const size_t a = t.find_first_not_of(ws);
            if (a == std::string::npos) return std::string();
            return t.substr(a, t.find_last_not_of(ws) - a + 1);
        };
        while (std::getline(f, line)) {
            const size_t hash = line.find('#');
            if (hash != std::string::npos) line.erase(hash);
            line = trim(line);
            if (line.empty()) continue;
            if (line[0] == '[') {
                section = trim(line.substr(1, line.find(']') == std::string::npos ? std::string::npos : line.find(']') - 1));
                continue;
            }
            const size_t eq = line.find('=');
            if (eq == std::string::npos) continue;
            const std::string key = trim(line.substr(0, eq)), val = trim(line.substr(eq + 1));
            if (val.empty()) continue;
            if (section == "Crawl" && key == "linearSpeedX") ini.linearSpeedX = std::atof(val.c_str());
            else if (section == "Crawl" && key == "linearSpeedY") ini.linearSpeedY = std::atof(val.c_str());
            else if (section == "Replanning" && key == "useComStepCorrection") ini.useComStepCorrection = std::atoi(val.c_str());
            else if (section == "Replanning" && key == "optimizeVelocity") ini.optimizeVelocity = std::atoi(val.c_str());
        }
        return ini;
    }
};

class MPCPlanner {
  public:
    struct BoxLimits { // MPCPlanner.h:29-34
        VectorXd max, min;
        void resize(int size) {
            max.assign(size, 0.0);
            min.assign(size, 0.0);
        }
    };
    struct footState { // MPCPlanner.h:35-44 (swing is 0.0 / 1.0)
        VectorXd x, y, swing;
        void resize(int size) {
            x.assign(size, 0.0);
            y.assign(size, 0.0);
            swing.assign(size, 0.0);
        }
    };
    enum state_type { POSITION = 0, VELOCITY, ACCELERATION };

    // MPCPlanner.cpp:22-70; `device` is the HIP device ordinal the solves run on
    MPCPlanner(const double horizon_size, const double Ts, const double gravity, int device = 0)
        : horizon_size_((int)horizon_size), Ts_(Ts), gravity_(gravity), device_(device) {
        hip_offsets_.fill(0.0);
    }
    ~MPCPlanner() {
        for (auto &kv : handles_) mpcb_destroy(kv.second);
    }
    MPCPlanner(const MPCPlanner &) = delete;
    MPCPlanner &operator=(const MPCPlanner &) = delete;

    void setWeights(double weight_R, double weight_Q, double weight_Qs = 1) { // :94-98
        weight_R_ = weight_R;
        weight_Q_ = weight_Q;
        weight_Qs_ = weight_Qs;
    }
    void setHorizonSize(int horizon) { horizon_size_ = horizon; } // :78-92
    void printSwing(LegID swing) const { // :1331-1349
        static const char *names[4] = {"LF", "RF", "LH", "RH"};
        std::printf("swing %s:\n", names[(int)swing & 3]);
    }
    void setHipOffsets(const LegDataMap<Vector2d> &hip_offsets) {  // :1352-1355
        for (int l = 0; l < 4; l++) {
            hip_offsets_[2 * l] = hip_offsets[l][0];
            hip_offsets_[2 * l + 1] = hip_offsets[l][1];
        }
    }

    // ---- trajectories (MPCPlanner.cpp:161-220): the triple integrator, O(N) ----------
    void computeZMPtrajectory(const Vector3d &x0, const VectorXd &jerk, VectorXd &zmp) const {
        rollout(x0, jerk, zmp, 3);
    }
    void computeZMPtrajectory(const Vector3d &x0, const Vector3d &y0, const VectorXd &jerk_x,
                              const VectorXd &jerk_y, VectorXd &zmp_x, VectorXd &zmp_y) const {
        rollout(x0, jerk_x, zmp_x, 3);
        rollout(y0, jerk_y, zmp_y, 3);
    }
    void computeCOMtrajectory(const Vector3d &x0, const VectorXd &jerk, VectorXd &traj,
                              const state_type state = POSITION) const {
        rollout(x0, jerk, traj, (int)state);
    }
    void computeCOMtrajectory(const Vector3d &x0, const Vector3d &y0, const VectorXd &jerk_x,
                              const VectorXd &jerk_y, VectorXd &traj_x, VectorXd &traj_y,
                              const state_type state = POSITION) const {
        rollout(x0, jerk_x, traj_x, (int)state);
        rollout(y0, jerk_y, traj_y, (int)state);
    }
    Vector3d computeCOMtrajectory(const Vector3d &x0, const VectorXd &jerk) const { // last state
        Vector3d s = x0;
        for (double u : jerk) computeCOMupdate(s, u);
        return s;
    }
    void computeCOMupdate(Vector3d &s, const double u) const { // :1155-1158
        const double T = Ts_;
        const double p = (s[0] + T * s[1] + T * T / 2 * s[2]) + T * T * T / 6 * u;
        const double v = (s[1] + T * s[2]) + T * T / 2 * u;
        const double a = s[2] + T * u;
        s = {p, v, a};
    }

    // ---- solves ----------------------------------------------------------------------
    // :223-237
    void solveQP(const double actual_height, const Vector3d &x0, const VectorXd &zmp_ref, VectorXd &jerk) {
        need((int)zmp_ref.size() == horizon_size_, "zmp_ref must have horizon_size entries");
        last_height_ = actual_height;
        mpcb_handle h = handle(MPCB_UNCONSTRAINED, actual_height, 1000, 1);
        jerk.assign(horizon_size_, 0.0);
        mpcb_outputs o = {};
        o.jerk_x = jerk.data();
        o.status = &last_status_;
        check(mpcb_solve_unconstrained(h, 1, x0.data(), zmp_ref.data(), &o));
        last_cost_ = 0.0;
        last_active_.clear();
    }
    // :240-300
    void solveQPconstraint(const double actual_height, const Vector3d &x0, const BoxLimits &zmpLim,
                           VectorXd &jerk) {
        need((int)zmpLim.min.size() == horizon_size_ && (int)zmpLim.max.size() == horizon_size_,
             "box limits must have horizon_size entries");
        last_height_ = actual_height;
        mpcb_handle h = handle(MPCB_BOX_1D, actual_height, 1000, 1);
        jerk.assign(horizon_size_, 0.0);
        mpcb_outputs o = outputs(jerk.data(), nullptr);
        check(mpcb_solve_box(h, 1, x0.data(), zmpLim.min.data(), zmpLim.max.data(), &actual_height, &o));
        finish(o);
    }
    // :303-424: one slack variable per bound (weight_Qs), kept in `slacks` like the
    // reference's member of that name (MPCPlanner.h:191)
    void solveQPconstraintSlack(const double actual_height, const Vector3d &x0, const BoxLimits &zmpLim,
                                VectorXd &jerk) {
        need((int)zmpLim.min.size() == horizon_size_ && (int)zmpLim.max.size() == horizon_size_,
             "box limits must have horizon_size entries");
        last_height_ = actual_height;
        mpcb_handle h = handle(MPCB_BOX_SLACK, actual_height, 1000, 1);
        jerk.assign(horizon_size_, 0.0);
        slacks.assign(2 * horizon_size_, 0.0);
        mpcb_outputs o = outputs(jerk.data(), nullptr);
        last_active_.assign(4 * horizon_size_, -1);
        o.active_idx = last_active_.data();
        o.active_stride = (int)last_active_.size();
        o.slacks = slacks.data();
        check(mpcb_solve_box(h, 1, x0.data(), zmpLim.min.data(), zmpLim.max.data(), &actual_height, &o));
        finish(o);
    }
    VectorXd slacks;
    // :427-490
    void solveQPConstraintCoupled(const double actual_height, const Vector3d &x0, const Vector3d &y0,
                                  const MatrixXd &A, const VectorXd &b, VectorXd &jerk_x, VectorXd &jerk_y) {
        solve_coupled(MPCB_COUPLED_JERK, actual_height, x0, y0, nullptr, nullptr, A, b, {0.0, 0.0}, jerk_x,
                      jerk_y, 1000);
    }
    // :493-628
    void solveQPConstraintCoupled(const double actual_height, const Vector3d &x0, const Vector3d &y0,
                                  const MatrixXd &A, const VectorXd &b, const Vector2d &targetSpeed,
                                  VectorXd &jerk_x, VectorXd &jerk_y, int replanningWindow = 1000) {
        solve_coupled(MPCB_COUPLED_VEL, actual_height, x0, y0, nullptr, nullptr, A, b, targetSpeed, jerk_x,
                      jerk_y, replanningWindow);
    }
    // :632-783.  The reference must be constant over each phase of computeSteps (as
    // computeCentroid's is); anything else is rejected.
    void solveQPConstraintCoupledRef(const double actual_height, const Vector3d &x0, const Vector3d &y0,
                                     const VectorXd &zmp_ref_x, const VectorXd &zmp_ref_y, const MatrixXd &A,
                                     const VectorXd &b, const Vector2d &targetSpeed, VectorXd &jerk_x,
                                     VectorXd &jerk_y, int replanningWindow = 1000) {
        solve_coupled(MPCB_COUPLED_REF, actual_height, x0, y0, &zmp_ref_x, &zmp_ref_y, A, b, targetSpeed,
                      jerk_x, jerk_y, replanningWindow);
    }

    // ---- constraint assembly (host; the batch entry points do this on the GPU) ---------
    // :1040-1082
    void buildPolygonMatrix(const LegDataMap<footState> &feetStates, const int start_phase_index,
                            const int phase_duration, const int horizon_size, MatrixXd &A, VectorXd &b,
                            int &number_of_constraints) const {
        if (phase_duration == 0) return;
        double px[4], py[4];
        int E = 0;
        for (int leg = 0; leg < 4; leg++)
            if (!feetStates[leg].swing[start_phase_index]) {
                px[E] = feetStates[leg].x[start_phase_index];
                py[E] = feetStates[leg].y[start_phase_index];
                E++;
            }
        // planning::CounterClockwiseSort (planning.h:182-197), pivot = first stance foot
        for (int i = 1; i < E - 1; i++)
            for (int j = i + 1; j < E; j++)
                if ((px[j] - px[0]) * (py[i] - py[0]) - (px[i] - px[0]) * (py[j] - py[0]) > 0.0) {
                    std::swap(px[i], px[j]);
                    std::swap(py[i], py[j]);
                }
        double lp[4], lq[4], lr[4];
        for (int e = 0; e < E; e++) { // planning::LineCoeff, normalised (planning.h:236-260)
            const int nx = (e + 1) % E;
            lp[e] = py[e] - py[nx];
            lq[e] = px[nx] - px[e];
            lr[e] = -lp[e] * px[e] - lq[e] * py[e];
            const double norm = std::hypot(lp[e], lq[e]);
            lp[e] /= norm;
            lq[e] /= norm;
            lr[e] /= norm;
        }
        for (int i = 0; i < phase_duration; i++) {
            need(number_of_constraints + E <= A.rows(), "polygon rows exceed the capacity of A");
            for (int e = 0; e < E; e++) {
                A(number_of_constraints + e, start_phase_index + i) = lp[e];
                A(number_of_constraints + e, start_phase_index + i + horizon_size) = lq[e];
                b[number_of_constraints + e] = lr[e];
            }
            number_of_constraints += E;
        }
    }
    // :1188-1329
    void computeSteps(const Vector2d &userSpeed, const LegDataMap<double> &initial_feet_x,
                      const LegDataMap<double> &initial_feet_y, const int number_of_steps, const int horizon_size,
                      LegDataMap<footState> &feetStates, LegDataMap<footState> &footHolds, MatrixXd &A, VectorXd &b,
                      MPCPlanner &myPlanner, LegID swing_leg_index, Vector2d initialCoM = {0.0, 0.0}) {
        need(number_of_steps > 0 && horizon_size > 0, "computeSteps: bad sizes");
        static const int sequence[4] = {LF, RH, RF, LH}; // :1197
        int cur = 0;
        for (int i = 0; i < 4; i++)
            if (sequence[i] == swing_leg_index) {
                cur = i;
                break;
            }
        const int step_knots = horizon_size / number_of_steps, phase = step_knots / 2;
        double fx[4], fy[4];
        for (int l = 0; l < 4; l++) {
            fx[l] = initial_feet_x[l];
            fy[l] = initial_feet_y[l];
            feetStates[l].resize(horizon_size);
            footHolds[l].resize(number_of_steps);
        }
        A.resize(4 * step_knots * number_of_steps, horizon_size * 2);
        b.assign((size_t)4 * step_knots * number_of_steps, 0.0);
        bool flag[4] = {false, false, false, false};
        double corr[4][2] = {};
        if (!(initialCoM[0] == 0.0 && initialCoM[1] == 0.0)) // :1231-1238
            for (int l = 0; l < 4; l++) {
                flag[l] = true;
                corr[l][0] = ((initialCoM[0] + hip_offsets_[2 * l]) + userSpeed[0]) - fx[l];
                corr[l][1] = ((initialCoM[1] + hip_offsets_[2 * l + 1]) + userSpeed[1]) - fy[l];
            }
        int nc = 0, start = 0;
        auto hold = [&](int from, int len) {
            for (int l = 0; l < 4; l++)
                for (int k = from; k < from + len; k++) {
                    feetStates[l].x[k] = fx[l];
                    feetStates[l].y[k] = fy[l];
                }
        };
        for (int i = 0; i < number_of_steps; i++) {
            hold(start, phase);
            myPlanner.buildPolygonMatrix(feetStates, start, phase, horizon_size, A, b, nc);
            start += phase;
            const int sw = sequence[cur];
            const double dx = flag[sw] ? corr[sw][0] : userSpeed[0];
            const double dy = flag[sw] ? corr[sw][1] : userSpeed[1];
            flag[sw] = false;
            fx[sw] += dx;
            fy[sw] += dy;
            for (int l = 0; l < 4; l++) {
                footHolds[l].x[i] = fx[l];
                footHolds[l].y[i] = fy[l];
            }
            for (int k = start; k < start + phase; k++) feetStates[sw].swing[k] = 1.0;
            hold(start, phase);
            myPlanner.buildPolygonMatrix(feetStates, start, phase, horizon_size, A, b, nc);
            start += phase;
            cur = (cur + 1) % 4;
        }
        const int missing = horizon_size - start; // :1309-1321
        hold(start, missing);
        myPlanner.buildPolygonMatrix(feetStates, start, missing, horizon_size, A, b, nc);
        MatrixXd A2; // conservativeResize (:1324-1326)
        A2.resize(nc, horizon_size * 2);
        for (int j = 0; j < horizon_size * 2; j++)
            for (int i = 0; i < nc; i++) A2(i, j) = A(i, j);
        A = A2;
        b.resize(nc);
        number_of_steps_ = number_of_steps;
        myPlanner.number_of_steps_ = number_of_steps;
        last_feet_ = feetStates;
        myPlanner.last_feet_ = feetStates;
    }
    // :1172-1185
    void computeSteps(const LegDataMap<double> &initial_feet_x, const LegDataMap<double> &initial_feet_y,
                      const double distance, const int number_of_steps, const int horizon_size,
                      LegDataMap<footState> &feetStates, LegDataMap<footState> &footHolds, MatrixXd &A, VectorXd &b,
                      MPCPlanner &myPlanner) {
        computeSteps({distance / number_of_steps, 0.0}, initial_feet_x, initial_feet_y, number_of_steps,
                     horizon_size, feetStates, footHolds, A, b, myPlanner, LF);
    }
    // :1106-1130
    void computeCentroid(const LegDataMap<footState> &feetStates, VectorXd &centroidX, VectorXd &centroidY) const {
        centroidX.assign(horizon_size_, 0.0);
        centroidY.assign(horizon_size_, 0.0);
        for (int i = 0; i < horizon_size_; i++) {
            int n = 0;
            double sx = 0.0, sy = 0.0;
            for (int leg = 0; leg < 4; leg++)
                if (!feetStates[leg].swing[i]) {
                    sx += feetStates[leg].x[i];
                    sy += feetStates[leg].y[i];
                    n++;
                }
            centroidX[i] = sx / n;
            centroidY[i] = sy / n;
        }
    }
    // :1132-1153 -- per-knot minimum of CI*x + ci0 of the last coupled solve
    VectorXd getConstraintViolation(const LegDataMap<footState> &) const { return last_margin_; }
    // :1161-1171
    VectorXd makeGaussian(const int length, const int mean, const int stddev) const {
        VectorXd w(length);
        for (int i = 0; i < length; i++) {
            const double q = ((double)(i + 1) - mean) / stddev;
            w[i] = std::exp(-0.5 * q * q) / (std::sqrt(2 * M_PI) * stddev);
        }
        return w;
    }
    // :962-1006 -- "time value" rows in $HOME/<name>, default stream precision
    void saveTraj(const std::string &filename, const VectorXd &var, bool verbose = true) const {
        const std::string path = std::string(std::getenv("HOME") ? std::getenv("HOME") : ".") + "/" + filename;
        std::ofstream file(path.c_str());
        double time = 0.0;
        for (size_t i = 0; i < var.size(); i++) {
            file << time << " " << var[i] << " " << std::endl;
            time += Ts_;
        }
        if (verbose) std::printf("done saving  in %s\n", path.c_str());
    }
    void saveTraj(const std::string &filename, const VectorXd &var_x, const VectorXd &var_y, bool verbose = true) const {
        const std::string path = std::string(std::getenv("HOME") ? std::getenv("HOME") : ".") + "/" + filename;
        std::ofstream file(path.c_str());
        double time = 0.0;
        for (size_t i = 0; i < var_x.size(); i++) {
            file << time << " " << var_x[i] << " " << var_y[i] << " " << std::endl;
            time += Ts_;
        }
        if (verbose) std::printf("done saving x and y vars in %s\n", path.c_str());
    }

    // ---- what the reference computes but does not return --------------------------------
    double lastCost() const { return last_cost_; }          // value of solve_quadprog
    int lastStatus() const { return last_status_; }         // MPCB_OK / _INFEASIBLE / _ITER_CAP
    const std::vector<int> &lastActiveSet() const { return last_active_; }

  private:
    typedef std::tuple<int, double, double, double, double, int, int, int> Key;
    int horizon_size_;
    double Ts_, gravity_;
    int device_;
    double weight_R_ = 1e-06, weight_Q_ = 1.0, weight_Qs_ = 100.0; // :39-42
    std::array<double, 8> hip_offsets_;
    int number_of_steps_ = 0;
    LegDataMap<footState> last_feet_;
    std::map<Key, mpcb_handle> handles_;
    std::map<Key, unsigned long long> last_use_;
    unsigned long long use_clock_ = 0;
    double last_cost_ = 0.0;
    int last_status_ = 0, last_nactive_ = 0, last_iters_ = 0;
    std::vector<int> last_active_;
    VectorXd last_margin_;

    static void need(bool ok, const char *msg) {
        if (!ok) throw std::invalid_argument(msg);
    }
    static void check(int rc) {
        if (rc < 0) throw std::runtime_error(std::string("mpc_amd: ") + mpcb_last_error());
    }
    // One library handle per (formulation, weights, window, steps, horizon).  The height is passed
    // with every call (as the reference does) wherever the handle's tables do not depend on it;
    // only the ZMP-reference and the unconstrained form need a handle per height.  The hip
    // offsets are not part of the key: this class always hands the library finished polygon
    // rows, so the library never runs computeSteps with them.  At most kMaxHandles handles are
    // kept; the least recently used one is destroyed when another is needed.
    static constexpr size_t kMaxHandles = 64;
    static bool height_in_tables(int formulation) {
        return formulation == MPCB_COUPLED_REF || formulation == MPCB_UNCONSTRAINED;
    }
    mpcb_handle handle(int formulation, double height, int rw, int steps) {
        const Key key(formulation, height_in_tables(formulation) ? height : 0.0, weight_R_, weight_Q_, weight_Qs_,
                      rw, steps, horizon_size_);
        auto it = handles_.find(key);
        if (it != handles_.end()) {
            last_use_[key] = ++use_clock_;
            return it->second;
        }
        if (handles_.size() >= kMaxHandles) {
            auto oldest = handles_.begin();
            for (auto jt = handles_.begin(); jt != handles_.end(); ++jt)
                if (last_use_[jt->first] < last_use_[oldest->first]) oldest = jt;
            mpcb_destroy(oldest->second);
            last_use_.erase(oldest->first);
            handles_.erase(oldest);
        }
        last_use_[key] = ++use_clock_;
        mpcb_params p = {};
        p.horizon = horizon_size_;
        p.formulation = formulation;
        p.Ts = Ts_;
        p.gravity = gravity_;
        p.height = height;
        p.weight_R = weight_R_;
        p.weight_Q = weight_Q_;
        p.weight_Qs = weight_Qs_;
        p.replanning_window = rw;
        p.number_of_steps = steps;
        for (int i = 0; i < 8; i++) p.hip_offsets[i] = hip_offsets_[i];
        p.device = device_;
        mpcb_handle h = nullptr;
        check(mpcb_create(&p, &h));
        handles_[key] = h;
        return h;
    }
    mpcb_outputs outputs(double *jx, double *jy) {
        mpcb_outputs o = {};
        o.jerk_x = jx;
        o.jerk_y = jy;
        last_margin_.assign(horizon_size_, 0.0);
        o.min_margin = last_margin_.data();
        o.cost = &last_cost_;
        o.status = &last_status_;
        o.iters = &last_iters_;
        o.n_active = &last_nactive_;
        last_active_.assign(2 * horizon_size_, -1);
        o.active_idx = last_active_.data();
        o.active_stride = (int)last_active_.size();
        return o;
    }
    void finish(const mpcb_outputs &) {
        last_active_.resize(last_nactive_ < (int)last_active_.size() ? last_nactive_ : (int)last_active_.size());
        if (last_status_ == MPCB_INFEASIBLE) std::printf("couldn't find a feasible solution\n"); // :298
    }
    void solve_coupled(int formulation, double height, const Vector3d &x0, const Vector3d &y0,
                       const VectorXd *zrefx, const VectorXd *zrefy, const MatrixXd &A, const VectorXd &b,
                       const Vector2d &speed, VectorXd &jerk_x, VectorXd &jerk_y, int rw) {
        const int N = horizon_size_;
        need(A.cols() == 2 * N && A.rows() == (int)b.size(), "A must be m x 2N and b of size m");
        last_height_ = height;
        if (number_of_steps_ <= 0 || A.rows() != rows_of_schedule(N, number_of_steps_)) {
            // rows that do not come from computeSteps (e.g. polygons of a time-based gait)
            solve_coupled_free(formulation, height, x0, y0, zrefx, zrefy, A, b, speed, jerk_x, jerk_y, rw);
            return;
        }
        mpcb_handle h = handle(formulation, height, rw, number_of_steps_);
        const int m = mpcb_num_constraints(h), nph = mpcb_num_phases(h);
        need(A.rows() == m, "A does not have the rows computeSteps produces for this horizon/steps");
        // Row r of A is (p at column k, q at column k+N) for the knot k computeSteps gave it
        // (MPCPlanner.cpp:1072-1079); rebuild k from the swing flags of the last computeSteps.
        std::vector<double> rows((size_t)m * 3), zph((size_t)nph * 2, 0.0), zkn;
        bool ref_by_phase = true;
        const int pd = (N / number_of_steps_) / 2;
        int r = 0;
        for (int k = 0; k < N; k++) {
            int E = 0;
            for (int leg = 0; leg < 4; leg++) E += last_feet_[leg].swing[k] ? 0 : 1;
            for (int e = 0; e < E; e++, r++) {
                need(r < m, "row structure mismatch");
                rows[3 * r] = A(r, k);
                rows[3 * r + 1] = A(r, k + N);
                rows[3 * r + 2] = b[r];
            }
            if (zrefx) {
                const int ph = (pd > 0 && k < 2 * pd * number_of_steps_) ? k / pd : 2 * number_of_steps_;
                const int first = (ph == 2 * number_of_steps_) ? 2 * pd * number_of_steps_ : ph * pd;
                // constant over every footstep phase (computeCentroid's is): the per-phase form;
                // anything else goes to the library knot by knot
                if ((*zrefx)[k] != (*zrefx)[first] || (*zrefy)[k] != (*zrefy)[first]) ref_by_phase = false;
                zph[2 * ph] = (*zrefx)[k];
                zph[2 * ph + 1] = (*zrefy)[k];
            }
        }
        need(r == m, "row structure mismatch");
        jerk_x.assign(N, 0.0);
        jerk_y.assign(N, 0.0);
        mpcb_coupled_inputs in = {};
        in.x0 = x0.data();
        in.y0 = y0.data();
        in.speed = speed.data();
        in.rows_pqr = rows.data();
        if (zrefx && ref_by_phase) in.zref_phase = zph.data();
        if (zrefx && !ref_by_phase) {
            zkn = interleave(*zrefx, *zrefy);
            in.zref_knot = zkn.data();
        }
        in.height = height_in_tables(formulation) ? nullptr : &height;
        mpcb_outputs o = outputs(jerk_x.data(), jerk_y.data());
        check(mpcb_solve_coupled(h, 1, &in, &o));
        finish(o);
    }
    static std::vector<double> interleave(const VectorXd &a, const VectorXd &b) { // [k][x, y]
        std::vector<double> v(2 * a.size());
        for (size_t k = 0; k < a.size(); k++) {
            v[2 * k] = a[k];
            v[2 * k + 1] = b[k];
        }
        return v;
    }
    static int rows_of_schedule(int N, int steps) { // rows computeSteps leaves in A (:1324-1326)
        const int pd = (N / steps) / 2;
        return 7 * pd * steps + 4 * (N - 2 * pd * steps);
    }
    // Any A, b whose rows each constrain the ZMP at ONE knot (p at column k, q at column
    // k + N), at most four rows per knot.  Knot k's rows go to slots 4k .. 4k+3 of a
    // schedule-free handle in the order they appear in A; for the solver to break ties the way
    // the reference does (lowest row of A first) the rows of A should be ordered by knot, as
    // buildPolygonMatrix leaves them when the phases are appended in time order.
    void solve_coupled_free(int formulation, double height, const Vector3d &x0, const Vector3d &y0,
                            const VectorXd *zrefx, const VectorXd *zrefy, const MatrixXd &A, const VectorXd &b,
                            const Vector2d &speed, VectorXd &jerk_x, VectorXd &jerk_y, int rw) {
        const int N = horizon_size_, m = A.rows();
        mpcb_handle h = handle(formulation, height, rw, 0);
        std::vector<double> rows((size_t)N * 12);
        for (int i = 0; i < 4 * N; i++) {
            rows[3 * i] = 0.0;
            rows[3 * i + 1] = 0.0;
            rows[3 * i + 2] = HUGE_VAL;
        }
        std::vector<int> used(N, 0), slot_row((size_t)4 * N, -1);
        for (int r = 0; r < m; r++) {
            int k = -1;
            for (int j = 0; j < N; j++)
                if (A(r, j) != 0.0 || A(r, j + N) != 0.0) {
                    need(k < 0, "a row of A touches more than one knot");
                    k = j;
                }
            need(k >= 0, "a row of A is empty");
            need(used[k] < 4, "more than four rows on one knot");
            const int s = 4 * k + used[k]++;
            rows[3 * s] = A(r, k);
            rows[3 * s + 1] = A(r, k + N);
            rows[3 * s + 2] = b[r];
            slot_row[s] = r;
        }
        std::vector<double> zkn;
        if (zrefx) zkn = interleave(*zrefx, *zrefy); // no footstep phases here: the reference goes knot by knot
        jerk_x.assign(N, 0.0);
        jerk_y.assign(N, 0.0);
        mpcb_coupled_inputs in = {};
        in.x0 = x0.data();
        in.y0 = y0.data();
        in.speed = speed.data();
        in.rows_pqr = rows.data();
        in.zref_knot = zrefx ? zkn.data() : nullptr;
        in.height = height_in_tables(formulation) ? nullptr : &height;
        mpcb_outputs o = outputs(jerk_x.data(), jerk_y.data());
        last_active_.assign(4 * N, -1);
        o.active_idx = last_active_.data();
        o.active_stride = (int)last_active_.size();
        check(mpcb_solve_coupled(h, 1, &in, &o));
        finish(o);
        for (int &a : last_active_) a = slot_row[a]; // back to rows of A
    }
    // which: 0 position, 1 velocity, 2 acceleration, 3 ZMP (uses the height of the last solve)
    void rollout(const Vector3d &x0, const VectorXd &jerk, VectorXd &out, int which) const {
        out.assign(jerk.size(), 0.0);
        Vector3d s = x0;
        for (size_t k = 0; k < jerk.size(); k++) {
            computeCOMupdate(s, jerk[k]);
            out[k] = which == 3 ? s[0] - last_height_ / gravity_ * s[2] : s[which];
        }
    }

    // height used by computeZMPtrajectory: the reference's Cz keeps the height of the last
    // solve* call (MPCPlanner.cpp:231, 253, ...); default 0.5 (:43)
    double last_height_ = 0.5;
};

} // namespace mpc_amd
#endif
