// Receding-horizon bookkeeping between two replans, on the device: what the loop of
// tests/replanning/testPlannerXYconstraintMPC.cpp:108-209 does on the host between two
// solves, one thread per rollout.  Keeps a sweep of sequentially dependent replans free of
// host round trips: solve -> advance -> solve -> ...
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace mpcb {
namespace {

// CoM each rollout's first computeSteps starts from (initialCoM argument, :144).
__global__ void replan_init_kernel(ReplanState st, int B, int use_com_correction) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    st.com0[2 * i] = use_com_correction ? st.x0[3 * i] : 0.0;
    st.com0[2 * i + 1] = use_com_correction ? st.y0[3 * i] : 0.0;
}

__global__ void replan_advance_kernel(ReplanState st, ReplanStep sp, int B) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    // record the stage that was just solved
    if (sp.stage_cost) sp.stage_cost[(size_t)sp.stage * B + i] = sp.cost[i];
    if (sp.stage_status) sp.stage_status[(size_t)sp.stage * B + i] = sp.status[i];
    if (sp.stage_nactive && sp.n_active) sp.stage_nactive[(size_t)sp.stage * B + i] = sp.n_active[i];

    // feet at knot sampleW of the plan (:119-123): the first scheduled leg has stepped, by the
    // one-shot CoM correction or by userSpeed (MPCPlanner.cpp:1231-1238, 1266-1283)
    const int sw = st.swing0[i];
    const double c0x = st.com0[2 * i], c0y = st.com0[2 * i + 1];
    const double spx = st.speed[2 * i], spy = st.speed[2 * i + 1];
    if (sp.feet_move) {
        double fx = st.feet_x[4 * i + sw], fy = st.feet_y[4 * i + sw];
        if (!(c0x == 0.0 && c0y == 0.0)) {
            fx += ((c0x + sp.hip[2 * sw]) + spx) - fx;
            fy += ((c0y + sp.hip[2 * sw + 1]) + spy) - fy;
        } else {
            fx += spx;
            fy += spy;
        }
        st.feet_x[4 * i + sw] = fx;
        st.feet_y[4 * i + sw] = fy;
    }
    // replanning_schedule.next() (:133) over LF, RH, RF, LH
    st.swing0[i] = sw == 0 ? 3 : sw == 3 ? 1 : sw == 1 ? 2 : 0;

    // computeCOMupdate for the samples of this window (:206-207, MPCPlanner.cpp:1155-1158)
    double x[3] = {st.x0[3 * i], st.x0[3 * i + 1], st.x0[3 * i + 2]};
    double y[3] = {st.y0[3 * i], st.y0[3 * i + 1], st.y0[3 * i + 2]};
    const double T = sp.Ts;
    const double a01 = T, a02 = T * T / 2, b0 = T * T * T / 6, b1 = T * T / 2, b2 = T;
    const double dist = st.disturbance ? st.disturbance[i] : 0.0;
    for (int j = 0; j < sp.window; j++) {
        const int sample = sp.stage * sp.window + j;
        const double ux = sp.jerk_x[(size_t)i * sp.N + j];
        const double uy = sp.jerk_y[(size_t)i * sp.N + j] + (sample >= sp.disturbance_start ? dist : 0.0);
        const double nx0 = (x[0] + a01 * x[1] + a02 * x[2]) + b0 * ux;
        const double nx1 = (x[1] + a01 * x[2]) + b1 * ux;
        const double nx2 = x[2] + b2 * ux;
        const double ny0 = (y[0] + a01 * y[1] + a02 * y[2]) + b0 * uy;
        const double ny1 = (y[1] + a01 * y[2]) + b1 * uy;
        const double ny2 = y[2] + b2 * uy;
        x[0] = nx0; x[1] = nx1; x[2] = nx2;
        y[0] = ny0; y[1] = ny1; y[2] = ny2;
    }
    for (int r = 0; r < 3; r++) {
        st.x0[3 * i + r] = x[r];
        st.y0[3 * i + r] = y[r];
    }
    st.com0[2 * i] = sp.use_com_correction ? x[0] : 0.0;
    st.com0[2 * i + 1] = sp.use_com_correction ? y[0] : 0.0;
}

} // namespace

hipError_t launch_replan_init(const ReplanState &st, int B, int use_com_correction, hipStream_t stream) {
    if (B <= 0) return hipSuccess;
    hipLaunchKernelGGL(replan_init_kernel, dim3((B + 255) / 256), dim3(256), 0, stream, st, B, use_com_correction);
    return hipGetLastError();
}

hipError_t launch_replan_advance(const ReplanState &st, const ReplanStep &sp, int B, hipStream_t stream) {
    if (B <= 0) return hipSuccess;
    hipLaunchKernelGGL(replan_advance_kernel, dim3((B + 255) / 256), dim3(256), 0, stream, st, sp, B);
    return hipGetLastError();
}

} // namespace mpcb
