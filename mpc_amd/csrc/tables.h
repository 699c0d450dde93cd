// Device-side construction of the instance-independent tables of one or many parameter groups:
// prediction matrices (MPCPlanner::buildMatrix, MPCPlanner.cpp:101-156), the condensed Hessian
// block G1 (:233, :523-559, :723-739), its Cholesky factor, J = L^-T (eiquadprog.hpp:165-176) and
// the response matrices the solve kernels read.  tables.hip holds the kernels; setup.cpp keeps a
// host long-double construction of the same tables as the checker.
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/mpc_amd.h"

namespace mpcb {

// What one parameter group is built from.  The velocity weights (makeGaussian or a constant,
// MPCPlanner.cpp:535-547, O(N) host work) and the two ratios come in as double-double pairs so
// that the device arithmetic starts from the same numbers as the long-double checker.
struct TableSpec {
    int N = 0, form = 0, nphase = 0;
    bool height_tables = false;        // also build the height-free parts (per-instance heights)
    double Ts = 0, Qs = 0;
    double diag_hi = 0, diag_lo = 0;   // diagonal term of G1: R, or R/Q for solveQP
    double c_hi = 0, c_lo = 0;         // height / gravity
    std::vector<double> w_hi, w_lo;    // [N] velocity weights (empty: no velocity term)
    std::vector<int> knot_phase;       // [N] (COUPLED_REF: phase of every knot)
};

// Finished tables of one group (device pointers into ONE allocation, `block`, that the caller
// frees) and the scalars the host needs.
struct TableSet {
    void *block = nullptr;
    double *S = nullptr, *Wt = nullptr, *Sc = nullptr, *Sv = nullptr;      // N x N
    double *Kx = nullptr, *Kz = nullptr, *Kc = nullptr, *Kv = nullptr;      // N x 4
    double *J0 = nullptr;                                                  // N x N column-major
    double *zu_col = nullptr, *xp_col = nullptr, *xv_col = nullptr;         // N
    double *zx = nullptr, *xpx = nullptr, *xvx = nullptr;                   // N x 3
    double *Sp = nullptr, *Wp = nullptr, *F2 = nullptr, *F3 = nullptr;      // COUPLED_REF
    double *Spp = nullptr, *Sps = nullptr, *Saa = nullptr, *Wtp = nullptr, *Wta = nullptr; // height parts
    double *Kzp = nullptr, *Kza = nullptr;
    double F[16] = {0};
    double traceG1 = 0, traceJ1 = 0;   // trace(G1), trace(L1^-T)
};

// Builds the tables of n groups on the current device with one chain of launches on `stream`
// (grid y = group), waits for it and fills `out`.  Returns 0 or an MPCB_E_* code with a message.
int build_tables_device(const TableSpec *specs, int n, TableSet *out, hipStream_t stream, std::string *err);

// Tables for a ZMP reference that changes from knot to knot (every knot a segment of its own):
// Sp = Qs S, Wp = Qs W^T, F2[a][k] = -Qs (Kz[k][a] - Zx[k][a]), F3 = -Qs^2 S / 2
// (g0 gains -Qs Zu' zref, MPCPlanner.cpp:731-735).  Plain double scalings of finished tables.
hipError_t launch_knot_reference_tables(int N, double Qs, const double *S, const double *Wt, const double *Kz,
                                        const double *zx, double *Sp, double *Wp, double *F2, double *F3,
                                        hipStream_t stream);

} // namespace mpcb
