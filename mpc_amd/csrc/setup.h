// Host side of everything that does not depend on the instance: the layout of one QP, the
// inputs of the device-side table construction, and the long-double host construction of the
// prediction matrices, the condensed Hessian block, its Cholesky factor and the "response"
// matrices the kernels read (the checker of the device-side one).
#pragma once
#include <vector>

#include "../../include/mpc_amd.h"
#include "tables.h"

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
    // height-free tables for per-instance heights (build_height_tables); empty until asked for
    std::vector<double> Spp, Sps, Saa, Wtp, Wta;  // N x N
    std::vector<double> Kzp, Kza;                 // N x 4
};

// Validates the parameters and fills sizes and row bookkeeping (cheap, no tables).
// Returns 0 or an MPCB_E_* code; `err` receives a message.
int build_layout(const mpcb_params &p, HostSetup &out, const char **err);

// What tables.hip builds the tables of this handle from (weights and ratios in extended precision).
void table_spec(const mpcb_params &p, const HostSetup &layout, TableSpec &spec);

// c1, c2 (traces of the full Hessian and of L^-T) from the traces of one block.
void finish_scalars(const mpcb_params &p, HostSetup &out, double traceG1, double traceJ1);

// Formulations whose Hessian does not contain Zu: one handle serves every CoM height.
bool height_tables_apply(int formulation);

// The tables in long double on the host (after build_layout).
int build_tables_host(const mpcb_params &p, HostSetup &out, const char **err);

// Host construction of the tables that let every instance of a batch have its own CoM height
// (box, coupled jerk-only and coupled velocity formulations).  Returns 0 or an MPCB_E_* code.
int build_height_tables(const mpcb_params &p, HostSetup &out, const char **err);

} // namespace mpcb
