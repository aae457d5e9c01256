#!/usr/bin/env python3
"""Rebuild profiles/traffic.json from the rocprofv3 counter passes committed under profiles/ (round 6).

    python tools/r06_traffic.py

For every bench workload: the dominant kernel of the trace pass (largest total time), its mean FETCH_SIZE and
WRITE_SIZE (KiB per dispatch, one counter per pass -- tools/r06_profiles.sh) and
    HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) KiB
(FETCH_SIZE doubled as /opt/skills/guides/MI355X_MICROARCH.md, section HBM, prescribes for gfx950; the known 6.55 MB
noise stream of k_update_rows confirmed the factor in round 2).  algorithmic bytes per launch: the bench line of the
same run (roofline.algorithmic_bytes_per_launch)."""
import json
import os
import re

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
PROF = os.path.join(ROOT, "profiles")
# key in traffic.json -> (tag, label)
SOURCES = {
    "c2": ("r06p", "c2"), "c2_fast": ("r06p", "c2fast"), "c2s": ("r06p", "c2s"), "c2c": ("r06p", "c2c"),
    "c2m": ("r06p", "c2m"), "c2m1k": ("r06p", "c2m1k"),
    "c4shard": ("r06p", "c4shard"), "ns": ("r06p", "ns"), "c3": ("r06p", "c3"), "c4": ("r06p", "c4"), "c5": ("r06p", "c5"),
}
NOTES = {
    "c2": "the launch is the whole iteration: Philox in registers, the previous update combined by 100 of its workgroups, tile packets out; bound by instruction issue, the three float32-rounded walks and the in-launch hand-off of u, not by HBM",
    "c2_fast": "as c2 (tolerance mode: k_rollout_scan)",
    "c2s": "k_rollout_scan_exact direct (exact three-wave schedule inside the kernel; semantic map): one launch per iteration -- window copy 41 KB per workgroup (L2-served), tile packets in and out; bound by the state wave's ~46 instructions per step",
    "c2c": "as c2s (CVaR-bin map, traction changes from cell to cell)",
    "c2m": "round 6: the speed-map mode on the time-parallel kernel (k_rollout_scan_exact<.., SPEED>): as c2, the lookups read 32-bit cells (risk byte) and every step pays one float64 division, off every chain",
    "c2m1k": "as c2m at the reference's own N = 1024: 32 workgroups, the launch is as long as at N = 8192 (one tile per CU either way)",
    "c4shard": "k_rollout_pipe<6, cc global> (T = 200: whole-map window, control-cost products through a global scratch) + k_update_rows; the launch also writes the next iteration's noise",
    "ns": "north_star's shape on one GPU (N = 65536, T = 100, nominal map): k_rollout_fused, two passes over the noise (exact order of terminal and control costs); round 6: no scratch in the step loop, the generator beside it ordered by device flags",
    "c3": "cellsM + noise read once, + the next iteration's noise written in the launch's tail; the 4*N*M*T term of the algorithmic count is the map gather, served by L2: the kernel is VALU-bound",
    "c4": "2 x 105 MB of noise: the control-cost pass (after the terminal cost) re-reads it",
    "c5": "64 problems x 4096: two passes over 210 MB of noise, written by k_noise in front of the launch",
}


def table(path):
    rows, counters = [], {}
    for line in open(path):
        m = re.match(r"^(.+?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+\d+", line)
        if m and not line.startswith("#") and not line.startswith("kernel"):
            rows.append((m.group(1).strip(), int(m.group(2)), float(m.group(3)), float(m.group(4))))
        m = re.match(r"^(.+?)\s+(FETCH_SIZE|WRITE_SIZE)\s+([\d.]+)\s+\(n=(\d+)\)", line)
        if m:
            counters[(m.group(1).strip(), m.group(2))] = float(m.group(3))
    return rows, counters


def main():
    out = {"_note": __doc__.split("\n\n", 1)[1].replace("\n", " ")}
    for key, (tag, label) in SOURCES.items():
        base = os.path.join(PROF, "%s_%s" % (tag, label))
        rows, _ = table(base + "_trace.txt")
        # (the rollout launch: beside a slim update launch -- C5 -- the update's total can edge past it, and the counter
        #  passes, which order the streams with events, run the other form of that kernel)
        dominant = max((r for r in rows if "k_rollout" in r[0]), key=lambda r: r[2])
        _, cf = table(base + "_fetch.txt")
        _, cw = table(base + "_write.txt")
        fetch, write = cf[(dominant[0], "FETCH_SIZE")], cw[(dominant[0], "WRITE_SIZE")]
        bench = json.loads(open(os.path.join(PROF, "%s_bench_%s.json" % (tag, label))).read().strip().splitlines()[-1])
        algo = bench["roofline"]["algorithmic_bytes_per_launch"]
        hbm = int(round((2.0 * fetch + write) * 1024))
        out[key] = {"dominant_kernel": dominant[0], "avg_us_traced": dominant[3], "fetch_size_kib": fetch, "write_size_kib": write,
                    "dominant_kernel_hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": algo,
                    "ratio": round(hbm / algo, 3), "files": "profiles/%s_%s_{trace,fetch,write}.txt" % (tag, label),
                    "note": NOTES[key]}
    with open(os.path.join(PROF, "traffic.json"), "w") as fh:
        json.dump(out, fh, indent=1)
        fh.write("\n")
    for k, v in out.items():
        if k != "_note":
            print("%-8s %-60s %7.2f us  hbm %10d  algorithmic %10d  ratio %.3f" % (k, v["dominant_kernel"][:60], v["avg_us_traced"], v["dominant_kernel_hbm_bytes_per_launch"], v["algorithmic_bytes_per_launch"], v["ratio"]))


if __name__ == "__main__":
    main()
