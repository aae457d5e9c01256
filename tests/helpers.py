"""Shared helpers for the parity tests: golden-fixture access and packing of
the reference-style params dict."""
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    with np.load(os.path.join(GOLDEN_DIR, name + ".npz")) as z:
        return {k: z[k] for k in z.files}


def params_from_golden(g):
    """Rebuild the params dict that was handed to the reference (stored as
    float64 arrays under 'param_*')."""
    p = {}
    for k, v in g.items():
        if k.startswith("param_"):
            name = k[len("param_"):]
            p[name] = float(v) if v.ndim == 0 else np.asarray(v, dtype=np.float64)
    if "num_opt" in p:
        p["num_opt"] = int(p["num_opt"])
    return p


def iterations(g):
    n = int(g["num_iterations"])
    return [{k: g["it%d_%s" % (i, k)] for k in ("noise", "u_in", "costs", "weights", "u_out")}
            for i in range(n)]


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b) / np.maximum(np.abs(b), 1e-30)


def ulp_diff_f32(a, b):
    """Distance in float32 units-in-the-last-place (sign-magnitude safe)."""
    a = np.ascontiguousarray(a, dtype=np.float32).view(np.int32).astype(np.int64)
    b = np.ascontiguousarray(b, dtype=np.float32).view(np.int32).astype(np.int64)
    a = np.where(a < 0, -(a & 0x7FFFFFFF), a)
    b = np.where(b < 0, -(b & 0x7FFFFFFF), b)
    return np.abs(a - b)
