#!/usr/bin/env python3
"""Distribution check of the control-noise generator on many draws (default 1e9): chi-square over
256 equiprobable bins, two-sided tail masses, moments.  GPU box only:

    python tools/rng_tail_check.py [--draws 1e9] > gpurun_out/rng_tail_check.json
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--draws", type=float, default=1e9)
    args = ap.parse_args()
    from scipy import stats
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        from bench import build_planner as build
        w, cfg, lin, ang, planner, params = build("c2", 262144)
    edges = stats.norm.ppf(np.linspace(0.0, 1.0, 257)[1:-1])
    counts = np.zeros(256, dtype=np.int64)
    levels = (3.0, 3.5, 4.0, 4.5, 5.0)
    tails = np.zeros(len(levels), dtype=np.int64)
    total, s1, s2, s4 = 0, 0.0, 0.0, 0.0
    while total < args.draws:
        planner.sample_noise()
        a = planner.noise_samples_d.copy_to_host()
        for c, std in enumerate(params["u_std"]):
            z = (a[..., c] / np.float32(std)).ravel()
            counts += np.bincount(np.searchsorted(edges, z), minlength=256)
            az = np.abs(z)
            tails += [int((az > lv).sum()) for lv in levels]
            z64 = z.astype(np.float64)
            s1 += z64.sum(); s2 += (z64 ** 2).sum(); s4 += (z64 ** 4).sum()
            total += z.size
    chi2 = float(((counts - total / 256.0) ** 2 / (total / 256.0)).sum())
    out = dict(draws=total, chi2_255=chi2, chi2_p=float(stats.chi2.sf(chi2, 255)), mean=s1 / total,
               var=s2 / total, kurtosis=s4 / total,
               tails={str(lv): dict(count=int(c), expected=float(total * 2 * stats.norm.sf(lv)),
                                    sigma=float((c - total * 2 * stats.norm.sf(lv)) / np.sqrt(total * 2 * stats.norm.sf(lv))))
                      for lv, c in zip(levels, tails)})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
