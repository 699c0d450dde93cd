// Device-side argument blocks and launchers shared by the C ABI and the kernels.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/mpc_amd.h"

namespace mpcb {

// Internal per-instance status: the fast kernel could not certify the instance
// (working set larger than its LDS-resident factor, or the reduced system became
// numerically singular).  Such instances are re-solved by the dense kernel; the
// value never leaves the library.
constexpr int STATUS_UNRESOLVED = 3;

struct DevSetup {
    int N, m, axes, form;
    int steps, phase_knots, nphase;
    int iter_cap;
    const double *S, *Wt, *Sc, *Sv;   // N x N response matrices, row k = knot k
    const double *Kx, *Kz, *Kc, *Kv;  // N x 4 unconstrained responses
    const int *knot_phase, *knot_row0; // [N]
    double F[16];
    double psi_thr;                   // m * eps * c1 * c2 * 100 (eiquadprog.hpp:267)
    double traceG1;
    double hip[8];
};

struct DevIn {
    const double *x0, *y0, *speed, *feet_x, *feet_y, *com0;
    const int *swing0;
    const double *zmin, *zmax;        // BOX_1D
};

// Enqueues the structure-exploiting dual active-set kernel for B instances.
hipError_t launch_zspace(const DevSetup &sd, const DevIn &in, const mpcb_outputs &out, int B,
                         hipStream_t stream);

} // namespace mpcb
