// Device helpers shared by the wave-per-QP kernel and the dense fallback kernel.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace mpcb {

constexpr double EPS = 2.220446049250313e-16;
constexpr double DINF = __builtin_huge_val();

// LDS traffic between lanes of one wave: the hardware executes a wave's LDS
// instructions in order, so only the compiler has to be kept from reordering.
__device__ __forceinline__ void wsync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ int ufirst(int v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ double ufirst(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readfirstlane(lo);
    hi = __builtin_amdgcn_readfirstlane(hi);
    return __hiloint2double(hi, lo);
}

// Cross-lane reductions on DPP (data-parallel primitives: lane permutes inside the VALU,
// no LDS crossbar).  All 64 lanes must be active.  Control codes (gfx9): quad_perm 0x00-0xFF,
// row_half_mirror 0x141, row_mirror 0x140, row_bcast15 0x142, row_bcast31 0x143.
// Permutes whose source lane always exists (quad_perm, mirrors over all rows): the previous
// value of the destination is irrelevant, which saves the copy in front of each move.
template <int CTRL>
__device__ __forceinline__ double dpp_perm_f64(double v) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ int dpp_perm_i32(int v) {
    return __builtin_amdgcn_mov_dpp(v, CTRL, 0xF, 0xF, true);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double old, double v) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), CTRL, ROW_MASK, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), CTRL, ROW_MASK, 0xF, false);
    return __hiloint2double(hi, lo);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_i32(int old, int v) {
    return __builtin_amdgcn_update_dpp(old, v, CTRL, ROW_MASK, 0xF, false);
}

__device__ __forceinline__ double lane63(double v) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}

// Sum over the wave; the result is wave-uniform.  Butterfly inside each row of 16 lanes,
// then row totals are chained into lane 63.
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_perm_f64<0xB1>(v);  // quad_perm [1,0,3,2]
    v += dpp_perm_f64<0x4E>(v);  // quad_perm [2,3,0,1]
    v += dpp_perm_f64<0x141>(v); // row_half_mirror
    v += dpp_perm_f64<0x140>(v); // row_mirror
    v += dpp_f64<0x142, 0xA>(0.0, v); // row_bcast15 -> rows 1, 3
    v += dpp_f64<0x143, 0xC>(0.0, v); // row_bcast31 -> rows 2, 3
    return lane63(v);
}

// Inclusive prefix sum over the lanes (lane i receives v_0 + ... + v_i): shifts inside each row
// of 16 lanes (lanes without a source add 0), then the totals of the rows before.
__device__ __forceinline__ double wave_scan(double v) {
    v += dpp_f64<0x111, 0xF>(0.0, v); // row_shr:1
    v += dpp_f64<0x112, 0xF>(0.0, v); // row_shr:2
    v += dpp_f64<0x114, 0xF>(0.0, v); // row_shr:4
    v += dpp_f64<0x118, 0xF>(0.0, v); // row_shr:8
    v += dpp_f64<0x142, 0xA>(0.0, v); // row_bcast15 -> rows 1, 3
    v += dpp_f64<0x143, 0xC>(0.0, v); // row_bcast31 -> rows 2, 3
    return v;
}

// The hardware's minimum as one instruction.  fmin() on a value that comes out of a lane permute
// is compiled to three: the compiler cannot see that the permuted bits are a canonical number
// and quiets both operands first (v_max x, x).  v_min_f64 itself already returns the other
// operand when one is a NaN, which is all the reductions below rely on.
__device__ __forceinline__ double hw_min(double a, double b) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// NaN-ignoring minimum (fmin), like the reference's `<` comparisons that never pick a NaN.
__device__ __forceinline__ double wave_min(double v) {
    v = hw_min(v, dpp_perm_f64<0xB1>(v));
    v = hw_min(v, dpp_perm_f64<0x4E>(v));
    v = hw_min(v, dpp_perm_f64<0x141>(v));
    v = hw_min(v, dpp_perm_f64<0x140>(v));
    v = hw_min(v, dpp_f64<0x142, 0xA>(DINF, v));
    v = hw_min(v, dpp_f64<0x143, 0xC>(DINF, v));
    return lane63(v);
}

__device__ __forceinline__ int wave_min(int v) {
    v = min(v, dpp_perm_i32<0xB1>(v));
    v = min(v, dpp_perm_i32<0x4E>(v));
    v = min(v, dpp_perm_i32<0x141>(v));
    v = min(v, dpp_perm_i32<0x140>(v));
    v = min(v, dpp_i32<0x142, 0xA>(0x7fffffff, v));
    v = min(v, dpp_i32<0x143, 0xC>(0x7fffffff, v));
    return __builtin_amdgcn_readlane(v, 63);
}

// Sum of a and minimum of b over the wave in one pass: the two dependent chains interleave,
// so each fills the other's issue gaps.  Results are wave-uniform.
__device__ __forceinline__ void wave_sum_min(double a, double b, double &sum, double &mn) {
    a += dpp_perm_f64<0xB1>(a);
    b = hw_min(b, dpp_perm_f64<0xB1>(b));
    a += dpp_perm_f64<0x4E>(a);
    b = hw_min(b, dpp_perm_f64<0x4E>(b));
    a += dpp_perm_f64<0x141>(a);
    b = hw_min(b, dpp_perm_f64<0x141>(b));
    a += dpp_perm_f64<0x140>(a);
    b = hw_min(b, dpp_perm_f64<0x140>(b));
    a += dpp_f64<0x142, 0xA>(0.0, a);
    b = hw_min(b, dpp_f64<0x142, 0xA>(DINF, b));
    a += dpp_f64<0x143, 0xC>(0.0, a);
    b = hw_min(b, dpp_f64<0x143, 0xC>(DINF, b));
    sum = lane63(a);
    mn = lane63(b);
}

// Minimum of (val, key) over the wave; the smaller key wins ties; aux travels with the
// winner.  Results are wave-uniform; returns the winning lane (0 if no lane has a candidate).
// A lane without a candidate passes val = +inf.
__device__ __forceinline__ int wave_argmin(double &val, int &key, int &aux) {
    const double m = wave_min(val);
    const bool cand = val == m;
    const int k = wave_min(cand ? key : 0x7fffffff);
    const unsigned long long win = __ballot(cand && key == k);
    const int src = win ? __ffsll((long long)win) - 1 : 0;
    aux = __builtin_amdgcn_readlane(aux, src);
    key = k;
    val = m;
    return src;
}

// The same reductions over the first 16 lanes only (one DPP row: four butterfly steps, no
// cross-row steps); the other lanes' values are ignored.  Results are wave-uniform.
__device__ __forceinline__ double row0_sum(double v) {
    v += dpp_perm_f64<0xB1>(v);
    v += dpp_perm_f64<0x4E>(v);
    v += dpp_perm_f64<0x141>(v);
    v += dpp_perm_f64<0x140>(v);
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 0),
                            __builtin_amdgcn_readlane(__double2loint(v), 0));
}
__device__ __forceinline__ double row0_min(double v) {
    v = hw_min(v, dpp_perm_f64<0xB1>(v));
    v = hw_min(v, dpp_perm_f64<0x4E>(v));
    v = hw_min(v, dpp_perm_f64<0x141>(v));
    v = hw_min(v, dpp_perm_f64<0x140>(v));
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 0),
                            __builtin_amdgcn_readlane(__double2loint(v), 0));
}

// s = sqrt(x) and is = 1/sqrt(x) for x > 0 from ONE dependent chain (hardware reciprocal
// square root estimate + two coupled Goldschmidt steps + one residual correction each), accurate
// to about an ulp.  A correctly rounded sqrt followed by a division costs more than twice as
// many dependent instructions, and these sit on the critical path of every pass.
__device__ __forceinline__ void sqrt_rsqrt(double x, double &s, double &is) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    const double d = fma(-g, g, x);
    g = fma(d, h, g);
    const double y2 = h + h;
    const double e = fma(-g, y2, 1.0);
    s = g;
    is = fma(y2, e, y2);
}

// eiquadprog.hpp:90-108
__device__ __forceinline__ double distance(double a, double b) {
    const double a1 = fabs(a), b1 = fabs(b);
    if (a1 > b1) {
        const double t = b1 / a1;
        return a1 * sqrt(1.0 + t * t);
    } else if (b1 > a1) {
        const double t = a1 / b1;
        return b1 * sqrt(1.0 + t * t);
    }
    return a1 * sqrt(2.0);
}


// computeSteps (MPCPlanner.cpp:1188-1329) + buildPolygonMatrix (:1040-1082) for ONE
// phase of one instance.  The calling thread replays the footstep schedule (cheap:
// number_of_steps additions per foot), keeps the feet as they stand during phase
// `myph` (even = four feet down, odd = the scheduled leg in swing, 2*steps = the
// trailing four-stance knots) and writes that phase's edges (p, q, r) to
// tab[(myph*4 + edge)*3 ..].  speed = userSpeed, used as per-step displacement.
// cref (may be NULL) receives the centroid of the phase's stance feet, the ZMP
// reference of computeCentroid (MPCPlanner.cpp:1106-1130), at cref[2*myph .. +1].
// pfx, pfy: the four initial feet; (c0x, c0y): initialCoM; swing0: swing_leg_index.
__device__ inline void polygon_phase(const DevSetup &sd, const double *pfx, const double *pfy, double c0x,
                                     double c0y, int swing0, double spx, double spy, int myph, double *tab,
                                     double *cref) {
    const bool correct = !(c0x == 0.0 && c0y == 0.0); // initialCoM.isZero(0), :1231
    // schedule LF, RH, RF, LH (:1197); setCurrentSwing (FootScheduler.h:33-43)
    int cur = 0;
    if (swing0 == 3) cur = 1;
    else if (swing0 == 1) cur = 2;
    else if (swing0 == 2) cur = 3;
    double fx[4], fy[4], sx[4], sy[4], corrx[4], corry[4];
    bool flag[4];
#pragma unroll
    for (int l = 0; l < 4; l++) {
        fx[l] = pfx[l];
        fy[l] = pfy[l];
        sx[l] = fx[l];
        sy[l] = fy[l];
        flag[l] = correct;
        corrx[l] = ((c0x + sd.hip[2 * l]) + spx) - fx[l]; // :1234-1237
        corry[l] = ((c0y + sd.hip[2 * l + 1]) + spy) - fy[l];
    }
    // Feet during phase myph: the first (myph + 1) / 2 steps of the schedule have been taken
    // (phase 2i = four feet down before step i, phase 2i + 1 = step i under way, its foot
    // already displaced but excluded from the polygon).  Leg l, at schedule position pos_l,
    // has swung once for every j < taken with (cur + j) % 4 == pos_l; its displacements are
    // added one by one in the order the reference adds them, so the sums round identically.
    const int taken = myph >= 2 * sd.steps ? sd.steps : (myph + 1) >> 1;
    int excl = -1;
    if ((myph & 1) && myph < 2 * sd.steps) {
        const int c = (cur + (myph >> 1)) & 3;
        excl = (c == 0) ? 0 : (c == 1) ? 3 : (c == 2) ? 1 : 2;
    }
    int nsw[4];
#pragma unroll
    for (int l = 0; l < 4; l++) {
        const int pos = (l == 0) ? 0 : (l == 3) ? 1 : (l == 1) ? 2 : 3; // LF, RH, RF, LH
        const int first = (pos - cur) & 3;                              // index of the leg's first swing
        nsw[l] = taken > first ? (taken - first + 3) >> 2 : 0;
    }
    // (a swing that has not happened adds an exact zero: no branches in the replay)
#pragma unroll
    for (int l = 0; l < 4; l++) { // the first swing may carry the CoM correction (:1266-1280)
        const bool on = nsw[l] > 0;
        fx[l] += on ? (flag[l] ? corrx[l] : spx) : 0.0;
        fy[l] += on ? (flag[l] ? corry[l] : spy) : 0.0;
    }
    for (int j = 1; j < (sd.steps + 3) >> 2; j++) {
#pragma unroll
        for (int l = 0; l < 4; l++) {
            const bool on = j < nsw[l];
            fx[l] += on ? spx : 0.0;
            fy[l] += on ? spy : 0.0;
        }
    }
#pragma unroll
    for (int l = 0; l < 4; l++) {
        sx[l] = fx[l];
        sy[l] = fy[l];
    }
    if (myph >= sd.nphase) return;
    const int E = excl < 0 ? 4 : 3;
    if (cref) {
        double sumx = 0.0, sumy = 0.0;
#pragma unroll
        for (int l = 0; l < 4; l++)
            if (l != excl) {
                sumx += sx[l];
                sumy += sy[l];
            }
        cref[2 * myph] = sumx / E;
        cref[2 * myph + 1] = sumy / E;
    }
    double px[4], py[4];
#pragma unroll
    for (int e = 0; e < 4; e++) { // stance feet in LegID order (:1052-1059)
        const int leg = e + ((excl >= 0 && e >= excl) ? 1 : 0);
        px[e] = leg == 0 ? sx[0] : leg == 1 ? sx[1] : leg == 2 ? sx[2] : sx[3];
        py[e] = leg == 0 ? sy[0] : leg == 1 ? sy[1] : leg == 2 ? sy[2] : sy[3];
    }
    // CounterClockwiseSort (planning.h:182-197), pivot p[0]
    auto right_of = [&](int a, int b) { // Point2isRightOfLine(p0, pa, pb), :143-149
        return (px[b] - px[0]) * (py[a] - py[0]) - (px[a] - px[0]) * (py[b] - py[0]);
    };
    auto swp = [&](int a, int b) {
        double t = px[a];
        px[a] = px[b];
        px[b] = t;
        t = py[a];
        py[a] = py[b];
        py[b] = t;
    };
    if (right_of(1, 2) > 0.0) swp(1, 2);
    if (E == 4) {
        if (right_of(1, 3) > 0.0) swp(1, 3);
        if (right_of(2, 3) > 0.0) swp(2, 3);
    }
#pragma unroll
    for (int e = 0; e < 4; e++) {
        if (e < E) {
            // LineCoeff(pt_e, pt_{(e+1)%E}, normalize) (planning.h:236-260)
            const int nx = (e + 1 == E) ? 0 : e + 1;
            const double bx = nx == 0 ? px[0] : nx == 1 ? px[1] : nx == 2 ? px[2] : px[3];
            const double by = nx == 0 ? py[0] : nx == 1 ? py[1] : nx == 2 ? py[2] : py[3];
            double p = py[e] - by, q = bx - px[e];
            double r = -p * px[e] - q * py[e];
            const double norm = hypot(p, q);
            p /= norm;
            q /= norm;
            r /= norm;
            double *t = tab + (myph * 4 + e) * 3;
            t[0] = p;
            t[1] = q;
            t[2] = r;
        }
    }
}

// The same with the instance's feet, initial CoM and swing leg read from the batch inputs.
__device__ inline void polygon_phase(const DevSetup &sd, const DevIn &in, int inst, double spx,
                                     double spy, int myph, double *tab, double *cref) {
    if (in.phase_pqr) { // the caller's polygon of this phase, as it stands (its centroid is not known here)
        if (myph < sd.nphase) {
            const double *src = in.phase_pqr + ((size_t)inst * sd.nphase + myph) * 12;
            const int E = ((myph & 1) && myph < 2 * sd.steps) ? 3 : 4;
            for (int i = 0; i < 3 * E; i++) tab[myph * 12 + i] = src[i];
        }
        return;
    }
    double c0x = 0.0, c0y = 0.0;
    if (in.com0) {
        c0x = in.com0[(size_t)inst * 2];
        c0y = in.com0[(size_t)inst * 2 + 1];
    }
    polygon_phase(sd, in.feet_x + (size_t)inst * 4, in.feet_y + (size_t)inst * 4, c0x, c0y, in.swing0[inst], spx,
                  spy, myph, tab, cref);
}

// Cost of the unconstrained minimiser, f = 1/2 g0'x (eiquadprog.hpp:191), for one axis:
// w'Fw, plus for the ZMP-reference formulation the terms in the per-phase reference c.
// Call with every index ph in [0, nref) covered once over the calling threads and
// sum the results; `base` adds w'Fw (pass true from exactly one thread).
__device__ __forceinline__ double unconstrained_cost_part(const DevSetup &sd, const double w[4],
                                                          const double *cref, int axis, int ph,
                                                          bool base) {
    double f = 0.0;
    if (base) {
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int b = 0; b < 4; b++) f += sd.F[a * 4 + b] * w[a] * w[b];
    }
    if (sd.F2 && ph < sd.nref) {
        double t = 0.0;
#pragma unroll
        for (int a = 0; a < 4; a++) t += w[a] * sd.F2[a * sd.nref + ph];
        for (int p2 = 0; p2 < sd.nref; p2++) t += sd.F3[ph * sd.nref + p2] * cref[2 * p2 + axis];
        f += t * cref[2 * ph + axis];
    }
    return f;
}

// Called by one thread of every workgroup of the dense kernel after its last instance: the last
// workgroup out (all others have read the counters by then) keeps the counts of this solve for
// the host and returns the counter block to zero for the next solve.
__device__ __forceinline__ void retire_counters(int *counters, unsigned workgroups, int *host_count) {
    __threadfence();
    if (atomicAdd(counters + CNT_DONE, 1) == (int)workgroups - 1) {
        // (the host sizes the next solve's workgroups by everything this kernel had to work through)
        if (host_count) *host_count = counters[CNT_HANDOVER] + counters[CNT_BLOCK];
        counters[CNT_LAST_HANDOVER] = counters[CNT_HANDOVER] + counters[CNT_FELL];
        counters[CNT_LAST_WIDE] = counters[CNT_WIDE];
        counters[CNT_LAST_BLOCK] = counters[CNT_BLOCK];
        counters[CNT_HANDOVER] = 0;
        counters[CNT_WIDE] = 0;
        counters[CNT_BLOCK] = 0;
        counters[CNT_FELL] = 0;
        counters[CNT_DONE] = 0;
    }
}

} // namespace mpcb
