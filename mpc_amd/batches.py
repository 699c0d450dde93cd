"""Deterministic synthetic batches of planner inputs.

Every instance draws from its own SplitMix64 stream seeded with
``seed XOR instance_index`` (seed 20171114 by default), so a batch can be
regenerated anywhere, in any order and for any slice of instance indices: rank r
of an N-rank job generates instances [r*B, (r+1)*B) and gets exactly the values a
single process would.

Shapes follow the reference drivers:
  * coupled:   tests/constrainedQP_coupled/testPlannerXYconstraintCoupled.cpp:30-53,80-88
  * decoupled: tests/constrainedQP_decoupled/testPlannerXYconstraintDecoupled.cpp:24-33,66-107
  * replanning: tests/replanning/testPlannerXYconstraintMPC.cpp:95-104
"""
import numpy as np

SEED = 20171114
_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)

# crawl order of the planner's internal schedule (MPCPlanner.cpp:1197), as LegIDs
LF, RF, LH, RH = 0, 1, 2, 3
CRAWL_SEQUENCE = (LF, RH, RF, LH)
# Hip offsets [leg][x, y] used when the CoM step correction of computeSteps is on
# (MPCPlanner.cpp:1231-1238).  The reference's replanning driver never sets them (it reads
# uninitialised memory); zeros would send all four feet to the same point, so the synthetic
# sweeps use the nominal stance of the coupled driver relative to the CoM.
NOMINAL_HIP_OFFSETS = np.array([[0.1, 1.0], [0.2, -1.0], [-0.4, 1.0], [-0.3, -1.0]])


class SplitMix:
    """Vectorised SplitMix64: one independent stream per instance index."""

    def __init__(self, index, seed=SEED):
        self.state = np.asarray(index, dtype=np.uint64) ^ np.uint64(seed)

    def next_u64(self):
        with np.errstate(over="ignore"):
            self.state = self.state + _GOLDEN
            z = self.state.copy()
            z = (z ^ (z >> np.uint64(30))) * _M1
            z = (z ^ (z >> np.uint64(27))) * _M2
            return z ^ (z >> np.uint64(31))

    def uniform(self, lo=0.0, hi=1.0):
        u = (self.next_u64() >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
        return lo + (hi - lo) * u

    def randint(self, n):
        return (self.next_u64() % np.uint64(n)).astype(np.int32)


def coupled_batch(B, first=0, seed=SEED):
    """Inputs of B coupled-XY instances (N=200, 20 steps in the reference driver).

    Returns dict of float64/int32 arrays: x0[B,3], y0[B,3], speed[B,2], feet_x[B,4],
    feet_y[B,4] (LegID order), swing0[B], com0[B,2] (zeros: no step correction).
    """
    rng = SplitMix(np.arange(first, first + B), seed)
    base_x = np.array([0.1, 0.2, -0.4, -0.3])
    base_y = np.array([1.0, -1.0, 1.0, -1.0])
    feet_x = np.stack([base_x[l] + rng.uniform(-0.05, 0.05) for l in range(4)], axis=1)
    feet_y = np.stack([base_y[l] + rng.uniform(-0.05, 0.05) for l in range(4)], axis=1)
    speed = np.stack([rng.uniform(0.05, 0.2), rng.uniform(-0.05, 0.05)], axis=1)
    x0 = np.stack([rng.uniform(-0.05, 0.05), rng.uniform(-0.05, 0.05), np.zeros(B)], axis=1)
    y0 = np.stack([rng.uniform(-0.05, 0.05), rng.uniform(-0.05, 0.05), np.zeros(B)], axis=1)
    swing0 = np.array(CRAWL_SEQUENCE, dtype=np.int32)[rng.randint(4)]
    return dict(x0=x0, y0=y0, speed=speed, feet_x=feet_x, feet_y=feet_y, swing0=swing0,
                com0=np.zeros((B, 2)))


def box_limits(N, steps, distance, lateral_bound, phase_overlap=0.1):
    """Per-knot ZMP box limits exactly as the decoupled driver builds them
    (testPlannerXYconstraintDecoupled.cpp:55-107).  Scalars or [B] arrays."""
    distance = np.atleast_1d(np.asarray(distance, dtype=np.float64))
    lateral = np.atleast_1d(np.asarray(lateral_bound, dtype=np.float64))
    B = max(distance.size, lateral.size)
    distance = np.broadcast_to(distance, (B,))
    lateral = np.broadcast_to(lateral, (B,))
    dps = distance / steps
    step_knots = N // steps
    ph = step_knots // 2
    xmax = np.zeros((B, N))
    xmin = np.zeros((B, N))
    ymax = np.zeros((B, N))
    ymin = np.zeros((B, N))
    s = 0
    low = np.zeros(B)
    goleft = True
    for _ in range(steps - 1):
        xmax[:, s:s + ph] = (low + dps * (0.5 + phase_overlap))[:, None]
        xmin[:, s:s + ph] = (low - dps * phase_overlap)[:, None]
        ymax[:, s:s + ph] = 1.0
        ymin[:, s:s + ph] = -1.0
        s += ph
        low = low + dps / 2
        xmax[:, s:s + ph] = (low + dps / 2)[:, None]
        xmin[:, s:s + ph] = low[:, None]
        if goleft:
            ymax[:, s:s + ph] = 1.0
            ymin[:, s:s + ph] = lateral[:, None]
        else:
            ymax[:, s:s + ph] = -lateral[:, None]
            ymin[:, s:s + ph] = -1.0
        goleft = not goleft
        s += ph
        low = low + dps / 2
    xmax[:, s:] = distance[:, None]
    xmin[:, s:] = (low - dps * phase_overlap)[:, None]
    ymax[:, s:] = 1.0
    ymin[:, s:] = -1.0
    return xmin, xmax, ymin, ymax


def decoupled_batch(B, N=100, steps=10, first=0, seed=SEED):
    """Inputs of B X/Y pairs of box-constrained 1-D QPs."""
    rng = SplitMix(np.arange(first, first + B), seed)
    distance = rng.uniform(2.0, 4.0)
    lateral = rng.uniform(0.1, 0.3)
    x0 = np.stack([rng.uniform(-0.02, 0.15), rng.uniform(-0.05, 0.05), rng.uniform(-0.1, 0.1)],
                  axis=1)
    y0 = np.stack([rng.uniform(-0.5, 0.5), rng.uniform(-0.1, 0.1), np.zeros(B)], axis=1)
    xmin, xmax, ymin, ymax = box_limits(N, steps, distance, lateral)
    return dict(x0=x0, y0=y0, xmin=xmin, xmax=xmax, ymin=ymin, ymax=ymax)


def replanning_disturbance(B, first=0, seed=SEED):
    """Per-rollout Y-jerk disturbance of the replanning sweep, U[-0.5, 0.5]."""
    rng = SplitMix(np.arange(first, first + B), seed ^ 0x5EED)
    return rng.uniform(-0.5, 0.5)
