// Host-side construction of everything that does not depend on the instance:
// prediction matrices, the condensed Hessian block, its Cholesky factor and the
// "response" matrices the kernels read.  Extended precision (long double) so that
// the tables are correctly rounded doubles even when the Hessian is ill-conditioned.
#pragma once
#include <vector>

#include "../../include/mpc_amd.h"

namespace mpcb {

struct HostSetup {
    int N = 0;            // horizon
    int m = 0;            // inequality rows per instance
    int axes = 2;         // 1 for BOX_1D
    int nvar = 0;         // variables of one QP: axes * N, or 3N with one slack per bound
    double slack_ls = 0, slack_g0 = 0; // BOX_SLACK: sqrt(Qs), 1000 * Qs
    int steps = 0, phase_knots = 0, nphase = 0;
    // N x N, row k = response to a unit (multiplier * coefficient) at knot k
    std::vector<double> S;    // ZMP           S  = Zu  G1^-1 Zu^T   (symmetric)
    std::vector<double> Wt;   // jerk          W^T, W = G1^-1 Zu^T
    std::vector<double> Sc;   // CoM position  (Xpu W)^T
    std::vector<double> Sv;   // CoM velocity  (Xvu W)^T
    // N x 4: unconstrained response to w = (pos, vel, acc, target speed)
    std::vector<double> Kx, Kz, Kc, Kv;
    double F[16];         // f_unc = w^T F w per axis
    // ZMP-reference formulation: per-phase responses [nphase][N] and cost terms
    std::vector<double> Sp, Wp, F2, F3;
    double c1 = 0, c2 = 0;   // trace(G), trace(L^-T) of the full Hessian (eiquadprog.hpp:162,176)
    double traceG1 = 0;
    // row bookkeeping of computeSteps / buildPolygonMatrix
    std::vector<int> knot_phase, knot_row0;   // [N]
    std::vector<int> row_knot, row_slot;      // [m]
    // kept for the dense fallback and the unconstrained solve
    std::vector<double> zu_col, zx;           // Toeplitz first column of Zu; Zx (N x 3)
    std::vector<double> xp_col, xpx, xv_col, xvx; // the same for CoM position and velocity
    std::vector<double> J0;                   // L1^-T (N x N, column-major), upper triangular
    std::vector<double> Hneg;                 // -G1^-1 (N x N) row-major
    // height-free tables for per-instance heights (build_height_tables); empty until asked for
    std::vector<double> Spp, Sps, Saa, Wtp, Wta;  // N x N
    std::vector<double> Kzp, Kza;                 // N x 4
};

// Returns 0 or an MPCB_E_* code; `err` receives a message.
int build_setup(const mpcb_params &p, HostSetup &out, const char **err);

// Adds the tables that let every instance of a batch have its own CoM height (box, coupled
// jerk-only and coupled velocity formulations).  Returns 0 or an MPCB_E_* code.
int build_height_tables(const mpcb_params &p, HostSetup &out, const char **err);

} // namespace mpcb
