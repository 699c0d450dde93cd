// Unconstrained ZMP tracking, MPCPlanner::solveQP (MPCPlanner.cpp:223-237), for a batch:
//   jerk = -G^-1 Zu' (Zx x0 - zref),  G = Zu'Zu + (R/Q) I.
// G^-1 Zu' and its products with the prediction matrices do not depend on the instance
// (the reference inverts G on every call), so a solve is one row of a [B x N] x [N x N]
// product per requested trajectory.  One wavefront per instance: the reference row is
// staged in LDS, lane l accumulates outputs l, l+64, ... from coalesced table rows.
#include <hip/hip_runtime.h>

#include "device_common.h"
#include "kernels.h"

namespace mpcb {
namespace {

template <int NPL>
__global__ __launch_bounds__(256) void unconstrained_kernel(DevSetup sd, const double *x0, const double *zref,
                                                            mpcb_outputs out, int B) {
    extern __shared__ double smem[];
    const int lane = threadIdx.x & 63;
    const int wv = ufirst((int)(threadIdx.x >> 6));
    const int inst = ufirst((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    if (inst >= B) return;
    const int N = sd.N;
    double *zr = smem + (size_t)wv * N;
    for (int k = lane; k < N; k += 64) zr[k] = zref[(size_t)inst * N + k];
    wsync();
    const double w0 = x0[(size_t)inst * 3], w1 = x0[(size_t)inst * 3 + 1], w2 = x0[(size_t)inst * 3 + 2];
    int kc[NPL];
#pragma unroll
    for (int c = 0; c < NPL; c++) kc[c] = min(lane + 64 * c, N - 1);
    auto product = [&](const double *K, const double *M, double *dst) {
        if (!dst) return;
        double v[NPL];
#pragma unroll
        for (int c = 0; c < NPL; c++) {
            const double *k4 = K + (size_t)kc[c] * 4;
            v[c] = k4[0] * w0 + k4[1] * w1 + k4[2] * w2;
        }
        for (int k = 0; k < N; k++) {
            const double z = zr[k];
            const double *row = M + (size_t)k * N;
#pragma unroll
            for (int c = 0; c < NPL; c++) v[c] += z * row[kc[c]];
        }
#pragma unroll
        for (int c = 0; c < NPL; c++)
            if (lane + 64 * c < N) dst[(size_t)inst * N + lane + 64 * c] = v[c];
    };
    product(sd.Kx, sd.Wt, out.jerk_x);
    product(sd.Kz, sd.S, out.zmp_x);
    product(sd.Kc, sd.Sc, out.com_x);
    product(sd.Kv, sd.Sv, out.com_xd);
    if (lane == 0) {
        if (out.status) out.status[inst] = MPCB_OK;
        if (out.iters) out.iters[inst] = 0;
        if (out.n_active) out.n_active[inst] = 0;
    }
}

} // namespace

hipError_t launch_unconstrained(const DevSetup &sd, const double *x0, const double *zref,
                                const mpcb_outputs &out, int B, hipStream_t stream) {
    if (B <= 0) return hipSuccess;
    const int npl = (sd.N + 63) / 64;
    const size_t shmem = 4 * (size_t)sd.N * sizeof(double);
    const dim3 grid((B + 3) / 4), block(256);
    if (npl <= 2) hipLaunchKernelGGL(unconstrained_kernel<2>, grid, block, shmem, stream, sd, x0, zref, out, B);
    else if (npl <= 4) hipLaunchKernelGGL(unconstrained_kernel<4>, grid, block, shmem, stream, sd, x0, zref, out, B);
    else hipLaunchKernelGGL(unconstrained_kernel<8>, grid, block, shmem, stream, sd, x0, zref, out, B);
    return hipGetLastError();
}

} // namespace mpcb
