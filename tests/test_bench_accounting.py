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
    # the figure of the build container is labelled as such, never as a measurement of the run
    assert "value" not in out
    if os.path.exists(os.path.join(bench.ROOT, "profiles", "r02_reference_cudasim.json")):
        assert "stale_build_container_measurement" in out


def test_usable_cores_is_positive():
    assert bench.usable_cores() >= 1
