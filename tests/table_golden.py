"""Readers of tests/golden/table_reference.npz (see tests/golden/make_table_reference.py)."""
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# name in the fixture -> id of mpcb_get_matrix / mpcb_debug_host_table
TABLE_KEYS = {"S": 0, "Wt": 1, "Kz": 4, "Kx": 5, "zu_col": 6, "zx": 7, "J0": 12, "F": 13}


def golden_cases(P):
    """[(name, BatchedPlanner keyword arguments, {table name: array})] of the fixture."""
    g = np.load(os.path.join(_HERE, "golden", "table_reference.npz"))
    forms = {"VEL": P.COUPLED_VEL, "REF": P.COUPLED_REF, "UNC": P.UNCONSTRAINED, "JERK": P.COUPLED_JERK}
    out = []
    for line in g["cases"]:
        name, N, form, R, Q, Qs, rw, h = str(line).split()
        kw = dict(horizon=int(N), formulation=forms[form], weight_R=float(R), weight_Q=float(Q),
                  weight_Qs=float(Qs), replanning_window=int(rw), height=float(h), number_of_steps=4)
        ref = {k[len(name) + 1:]: g[k] for k in g.files if k.startswith(name + "/")}
        out.append((name, kw, ref))
    return out
