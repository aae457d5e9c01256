"""bench.py's host-side bookkeeping (no GPU): the algorithmic byte counts of SURVEY.md section 8d that
`roofline` is priced with, and the wording of the reference-CPU-path leg when the box lacks what it needs."""
import os

import bench


def test_algorithmic_bytes_of_the_headline_configuration():
    w = bench.WORKLOADS["c2"]
    rp = cp = 264  # 256 cells + the padding ring of bench.py's Config (max_speed_padding 5 m/s * 0.1 s / 0.25 m = 2, pitch 264)
    it, roll_pipe = bench.algorithmic_bytes(w, w["n"], rp, cp, rollout_writes_noise=True)
    _, roll_read = bench.algorithmic_bytes(w, w["n"], rp, cp, rollout_writes_noise=False)
    _, roll_fused = bench.algorithmic_bytes(w, w["n"], rp, cp, fused=True)
    steps = w["n"] * w["t"]
    assert it == steps * 28 + w["n"] * 16 + 32 * w["t"] + 4 * rp * cp          # 28 B per rollout-step per iteration
    assert roll_pipe - roll_read == 8 * steps                                  # the next iteration's noise, written
    assert roll_read == steps * 12 + w["n"] * 4 + 8 * w["t"] + 4 * rp * cp     # noise read + map bytes
    # a launch that samples the noise, rolls out and reduces the update per tile does the iteration's work
    # except the final combine: priced with the iteration's 28 B per rollout-step
    assert roll_fused == steps * 28 + w["n"] * 8 + 8 * w["t"] + 4 * rp * cp
    assert roll_fused <= it


def test_algorithmic_bytes_cvar():
    w = bench.WORKLOADS["c3"]
    n, t, m = w["n"], w["t"], w["m"]
    it, roll = bench.algorithmic_bytes(w, n, 264, 264)
    assert it == 4 * n * m * t + 24 * n * t + 16 * n + 2 * m * 264 * 264
    assert roll < it


def test_reference_leg_says_what_is_missing(monkeypatch, tmp_path):
    monkeypatch.setenv("MPPI_NUMBA_REFERENCE", str(tmp_path))           # no checkout there
    monkeypatch.setenv("MPPI_NUMBA_PYTHON", str(tmp_path / "python3"))  # no interpreter either
    out = bench.reference_cpu_path()
    assert out["status"].startswith("not measured in this run: missing ")
    assert "MPPI_NUMBA_REFERENCE" in out["status"] and "MPPI_NUMBA_PYTHON" in out["status"]
    # no figure from another machine on the line: the build container's measurement is named, not quoted
    assert "value" not in out and "stale_build_container_measurement" not in out
    assert "profiles/r02_reference_cudasim.json" in out["measured_elsewhere"]


def test_traffic_entry_follows_the_effective_configuration():
    """`--workload c4 --n 8192` is configs[3]'s shard: its own counter passes, never C4's."""
    assert bench.traffic_key("c2", 8192) == "c2"
    assert bench.traffic_key("c2", 8192, "fast") == "c2_fast"
    assert bench.traffic_key("c4", 65536) == "c4"
    assert bench.traffic_key("c4", 8192) == "c4shard"
    import json
    with open(os.path.join(bench.ROOT, "profiles", "traffic.json")) as fh:
        table = json.load(fh)
    assert bench.traffic_key("c4", 8192) in table
    assert bench.traffic_key("c4", 4096) not in table and bench.traffic_key("c2", 1024) not in table


def test_algorithmic_bytes_speed_map_counts_the_risk_byte():
    w, base = bench.WORKLOADS["c2m"], bench.WORKLOADS["c2"]
    it_m, roll_m = bench.algorithmic_bytes(w, w["n"], 264, 264, fused=True)
    it_d, roll_d = bench.algorithmic_bytes(base, base["n"], 264, 264, fused=True)
    steps = w["n"] * w["t"]
    assert it_m - it_d == steps + 264 * 264 and roll_m - roll_d == steps + 264 * 264  # 29 B per rollout-step


def test_the_published_barebone_configuration():
    """bench.py --workload bb is barebone_mppi_numba.ipynb cell 5, nothing else."""
    cfg, params = bench.barebone_problem()
    assert cfg == dict(T=5.0, dt=0.1, num_control_rollouts=1000, num_vis_state_rollouts=20, seed=1)
    assert params["num_opt"] == 1 and params["obs_penalty"] == 1e6 and params["dist_weight"] == 10
    assert params["obstacle_positions"].tolist() == [[5, 4.5], [2, 1]] and params["obstacle_radius"].tolist() == [1.5, 1]
    assert bench.BB_PUBLISHED["value_ms"] == 2.74


def test_usable_cores_is_positive():
    assert bench.usable_cores() >= 1
