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

// Hand-over from the wave-per-QP kernel to the dense kernel, and the dense kernel's
// tables and HBM workspace.
struct DenseWs {
    int *list;                 // instance indices the first stage could not certify
    int *count;                // how many
    double *J, *R;             // workspace: one n x n slot of each per resident workgroup
    const double *J1;          // L1^-T (N x N, column-major, upper triangular)
    const double *zu_col, *xp_col, *xv_col; // Toeplitz first columns of Zu, Xpu, Xvu
    const double *zx, *xpx, *xvx;           // N x 3 state matrices
    const int *row_knot, *row_slot;         // [m] (polygon rows only)
};

// Enqueues the structure-exploiting dual active-set kernel for B instances.  Instances it
// cannot certify are appended to ws.list (ws.count must be zero on entry).
hipError_t launch_zspace(const DevSetup &sd, const DevIn &in, const mpcb_outputs &out, int B,
                         const DenseWs &ws, hipStream_t stream);

// Fills ws.list with 0..B-1 and ws.count with B (dense-only mode).
hipError_t launch_list_all(const DenseWs &ws, int B, hipStream_t stream);

// Enqueues the dense kernel over ws.list with `blocks` workgroups (each owns one slot of
// ws.J / ws.R and loops over list entries blockIdx.x, blockIdx.x + blocks, ...).
hipError_t launch_dense(const DevSetup &sd, const DevIn &in, const mpcb_outputs &out, const DenseWs &ws,
                        int blocks, hipStream_t stream);

} // namespace mpcb
