#!/usr/bin/env python3
"""In-kernel clock stamps of one C2 iteration (developer tool; GPU box only).

    make -C mppi_numba_amd/csrc stamps
    MPPI_HIP_LIB=$PWD/build/libmppi_stamps.so python tools/stamp_timeline.py [--n 8192]

Prints, relative to the rollout kernel's first stamp, when each role of workgroup 5 passed each
point (cycles of s_memtime), and the phases of the update kernel's middle workgroup."""
import argparse
import contextlib
import ctypes as C
import io
import json
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=8192)
    ap.add_argument("--workload", default="c2")
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    from mppi_numba_amd import _lib
    with contextlib.redirect_stdout(io.StringIO()):
        from bench import build_planner as build
        w, cfg, lin, ang, planner, params = build(args.workload, args.n, math=os.environ.get("MPPI_MATH", "exact"))
        planner.solve()
        planner.iterate_async(20)
        planner.synchronize()
    buf = (C.c_ulonglong * 4096)()
    _lib.call("mppi_debug_read_stamps", buf, 4096, 1)
    samples = []
    for rep in range(5):
        planner.iterate_async(1)
        planner.synchronize()
        _lib.call("mppi_debug_read_stamps", buf, 4096, 1)
        samples.append(np.array(buf[:], dtype=np.uint64).astype(np.int64))
    st = samples[-1]
    print(planner.last_rollout_kernel())
    t0 = st[0]
    rel = lambda v: int(v - t0) if v else None
    out = {"rollout": {}, "update": {}, "noise_wg": {}}
    # (the branch that read the five-stage speculative pipeline's stamps went with that kernel: round 6)
    names = sorted({int(i) // 64 for i in np.flatnonzero(st[64:512]) + 64})
    for r in names:
        base = 64 * r
        row = {"prologue": rel(st[base]), "after_copy": rel(st[base + 1]), "first_barrier": rel(st[base + 2]),
               "intervals": [rel(v) for v in st[base + 3:base + 35] if v], "loop_end": rel(st[base + 40]),
               "tail_end": rel(st[base + 41]), "end": rel(st[base + 42])}
        out["rollout"]["role%d" % (r - 1)] = row
        iv = row["intervals"]
        print("role %d: prologue %s copy %s first-barrier %s | intervals %s | loop_end %s tail %s end %s"
              % (r - 1, row["prologue"], row["after_copy"], row["first_barrier"],
                 [b - a for a, b in zip(iv[:-1], iv[1:])], row["loop_end"], row["tail_end"], row["end"]))
        print("        interval marks:", iv)
    sub = st[1024:1024 + 128].reshape(32, 4)
    if sub.any():
        print("producer sub-stamps per interval (after band_load issue, after noise issue, after produce, after barrier), "
              "relative to the previous interval's barrier release:")
        prev = None
        for k in range(32):
            if not sub[k].any():
                break
            base = prev if prev is not None else sub[k][0]
            print("   k=%2d" % k, [int(v - base) for v in sub[k]])
            prev = sub[k][3]
    out["noise_wg"] = {"first": [rel(st[16]), rel(st[17])], "last": [rel(st[18]), rel(st[19])]}
    print("noise WG first:", out["noise_wg"]["first"], " last:", out["noise_wg"]["last"])
    u = st[512:520]
    out["update"] = [int(v - u[0]) if v else None for v in u]
    print("update kernel (middle WG) phases, cycles from its entry:", out["update"],
          " entry at", rel(u[0]), "after rollout start")
    if args.json:
        with open(args.json, "w") as fh:
            json.dump(out, fh)


if __name__ == "__main__":
    main()
