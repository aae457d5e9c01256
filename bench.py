#!/usr/bin/env python3
"""Benchmark of the MPPI hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3|c4|c5] [--graph ITERATIONS]

A "step" is one MPPI iteration: sample control noise -> roll every control
sample through the horizon with traction-grid lookups -> min-subtract +
exp-weighted control update (+ one RCCL all-gather of 2T+2 doubles when N > 1).
Inputs (maps, sampled grids, warm-started u) are resident in HBM before the
timed region; the K steps run back to back on the planner's stream (u stays on
the device between steps, exactly the data dependence of params['num_opt'] = K).

Multi-GPU (`--gpus N`, one node): one process per GPU.  Launched plainly, this script
starts its N ranks itself (mppi_numba_amd/launch.py); launched by
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` it takes
RANK / LOCAL_RANK / WORLD_SIZE from the launcher.  Either way there is no torch in the
process: the RCCL unique id, the barrier and the max-over-ranks clock go over a small
TCP hub on 127.0.0.1, the data path is RCCL on the planner's stream.
`--single-process` instead drives all N devices from ONE process (mppi_group_*: the
communicators and every iteration's all-gathers are issued inside RCCL groups).

Workloads (synthetic; BASELINE.json configs[1..4]; default c2, the
configuration BASELINE.json's metric is quoted on):
  c2  use_det_dynamics, N=8192 per GPU, T=100, 256x256 nominal traction grid
  c3  use_tdm (CVaR), N=4096 x M=128, 16-bin PMF, 256x256
  c4  use_det_dynamics, N=65536 per GPU, T=200, CVaR-bin traction
  ns  use_det_dynamics, N=65536 per GPU, T=100, nominal grid: north_star's target shape on one GPU
  c5  batched multi-query: 64 independent problems (own start / goal) x N=4096 per GPU, T=100,
      one launch over (problem, rollout); with several GPUs every rank solves its own 64
      problems (no exchange at all)
  c2m / c2m1k  use_nom_dynamics_with_speed_map (the reference's third mode), N=8192 / the reference's N=1024, T=100
  bb  the reference's ONLY published number: barebone_mppi_numba.ipynb cell 6, `%timeit mppi_planner.solve()` =
      2.74 ms at N=1000, T=50, two disc obstacles (RTX 3070) -- a step is one solve() call, timed as the notebook does
Multi-GPU is weak scaling: every rank owns `N` control samples of a global
problem of N*world samples (noise is keyed by the global sample index).

Prints ONE JSON line (rank 0).  If the RCCL communicator cannot be created the ranks
exchange their packets through the host (the TCP hub) instead and the JSON says so
(config.exchange).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md


def synthetic_world(workload, rng):
    """Maps per SURVEY.md section 8d: 256x256 cells of 0.25 m, Bernoulli(0.02)
    obstacle / unknown masks."""
    rows = cols = 256
    res = 0.25
    obstacle = (rng.random((rows, cols)) < 0.02).astype(np.int8)
    unknown = (rng.random((rows, cols)) < 0.02).astype(np.int8)
    # keep the start and goal cells free
    for m in (obstacle, unknown):
        m[12:20, 12:20] = 0
        m[236:244, 236:244] = 0
    if workload in ("c2", "ns", "c2l"):
        bins = 2
        pmf = np.zeros((bins, rows, cols), dtype=np.int8)
        pmf[-1] = 100  # nominal traction (README.md:136-151 recipe)
        bin_values = np.array([0.0, 1.0])
        alpha = 1.0
    elif workload == "c2s":
        # A semantic map as the reference builds it in deterministic-dynamics mode (terrain.py:183-342): every
        # cell's PMF is one-hot at the bin of its terrain type's (CVaR-)expected traction, i.e. the traction is
        # piecewise constant over patches of terrain.  Four terrain types in rectangles of 4-16 m (16-64 cells).
        bins = 16
        bin_values = np.linspace(0.0, 1.0, bins)
        type_bin = np.array([15, 12, 9, 7])  # traction 1.0, 0.8, 0.6, 0.47
        kinds = np.zeros((rows, cols), dtype=np.int64)
        r0 = 0
        while r0 < rows:
            h = int(rng.integers(16, 65))
            c0 = 0
            while c0 < cols:
                wd = int(rng.integers(16, 65))
                kinds[r0:r0 + h, c0:c0 + wd] = int(rng.integers(0, len(type_bin)))
                c0 += wd
            r0 += h
        pmf = np.zeros((bins, rows, cols), dtype=np.int8)
        pmf[type_bin[kinds], np.arange(rows)[:, None], np.arange(cols)[None, :]] = 100
        alpha = 1.0
    else:
        bins = 16
        raw = rng.dirichlet(np.ones(bins), size=(rows, cols))
        p = np.floor(raw * 100).astype(np.int64)
        p[..., -1] += 100 - p.sum(axis=-1)
        pmf = np.ascontiguousarray(np.moveaxis(p, -1, 0)).astype(np.int8)
        bin_values = np.linspace(0.0, 1.0, bins)
        alpha = 0.2
    tdm_dict = dict(xlimits=(0.0, cols * res), ylimits=(0.0, rows * res), res=res,
                    bin_values=bin_values, bin_values_bounds=(0.0, 1.0),
                    det_dynamics_cvar_alpha=alpha)
    return pmf, obstacle, unknown, tdm_dict


def make_params(workload):
    return dict(
        x0=np.array([4.0, 4.0, np.pi / 4]), xgoal=np.array([60.0, 60.0]), dt=0.1,
        goal_tolerance=0.5, v_post_rollout=0.01, lambda_weight=1.0,
        cvar_alpha=0.2 if workload == "c3" else 1.0, alpha_dyn=1.0, num_opt=1,
        u_std=np.array([2.0, 3.0]), vrange=np.array([0.0, 3.0]), wrange=np.array([-np.pi, np.pi]),
        dist_weight=1.0, obs_penalty=1e5, unknown_penalty=1e2)


WORKLOADS = {
    "c2": dict(n=8192, t=100, m=1, mode=dict(use_det_dynamics=True),
               label="Unicycle MPPI det-dyn, N=8192/GPU, T=100, 256x256 nominal traction grid"),
    # north_star's own target shape on ONE GPU: the denominator of "scaling 1 -> 8 GPUs at N=65536, T=100" 
    "ns": dict(n=65536, t=100, m=1, mode=dict(use_det_dynamics=True),
               label="Unicycle MPPI det-dyn, N=65536/GPU, T=100, 256x256 nominal traction grid (north_star's shape on one GPU)"),
    # C2's map at twice the horizon (no BASELINE configuration: the study of the kernel families, profiles/r06_families.md)
    "c2l": dict(n=8192, t=200, m=1, mode=dict(use_det_dynamics=True),
                label="Unicycle MPPI det-dyn, N=8192/GPU, T=200, 256x256 nominal traction grid (kernel-family study)"),
    # the C2 shape on maps the time-parallel kernel's assumption does not hold on
    "c2s": dict(n=8192, t=100, m=1, mode=dict(use_det_dynamics=True),
                label="Unicycle MPPI det-dyn, N=8192/GPU, T=100, 256x256 SEMANTIC map: 4 terrain types in patches of 4-16 m"),
    "c2c": dict(n=8192, t=100, m=1, mode=dict(use_det_dynamics=True),
                label="Unicycle MPPI det-dyn, N=8192/GPU, T=100, 256x256 CVaR-bin traction (changes from cell to cell, as C4)"),
    # the reference's third planner mode (mppi.py:1013-1111; benchmark.ipynb times it at N=1024, T=100): nominal dynamics,
    # the time of a step charged by the risk speed map -- at C2's shape and at the reference's own
    "c2m": dict(n=8192, t=100, m=1, mode=dict(use_nom_dynamics_with_speed_map=True), speed_map=True,
                label="Unicycle MPPI nominal dynamics + risk speed map (mppi.py:1013-1111), N=8192/GPU, T=100, 256x256 16-bin PMF, CVaR(0.2) speed map"),
    "c2m1k": dict(n=1024, t=100, m=1, mode=dict(use_nom_dynamics_with_speed_map=True), speed_map=True,
                  label="Unicycle MPPI nominal dynamics + risk speed map, the reference's own size N=1024, T=100 (benchmark.ipynb), 256x256 16-bin PMF"),
    "c3": dict(n=4096, t=100, m=128, mode=dict(use_tdm=True),
               label="CVaR MPPI, N=4096/GPU x M=128 traction samples, 16-bin PMF, 256x256"),
    "c4": dict(n=65536, t=200, m=1, mode=dict(use_det_dynamics=True),
               label="Unicycle MPPI det-dyn (CVaR-bin traction), N=65536/GPU, T=200, 256x256"),
    "c5": dict(n=4096, t=100, m=1, problems=64, mode=dict(use_det_dynamics=True),
               label="Batched multi-query: 64 problems/GPU x N=4096, T=100, det-dyn, 256x256 CVaR-bin traction"),
}


def build_planner(workload, n=None, rank=0, world=1, rng="philox", math="exact", seed=1):
    """(workload dict, cfg, lin_tdm, ang_tdm, planner, params): the objects of a bench workload at
    `n` control samples per GPU (tests/ and tools/ build their planners through this)."""
    from mppi_numba_amd.config import Config
    from mppi_numba_amd.mppi import MPPI_Numba
    from mppi_numba_amd.terrain import TDM_Numba
    w = dict(WORKLOADS[workload])
    if n is not None:
        w["n"] = n
    cfg = Config(T=w["t"] * 0.1, dt=0.1, num_grid_samples=w["m"], num_control_rollouts=w["n"] * world,
                 max_speed_padding=5.0, num_vis_state_rollouts=1, max_map_dim=(260, 260), seed=seed,
                 enforce_recommended_limits=False, rng=rng, math=math, **w["mode"])
    pmf, obstacle, unknown, tdm_dict = synthetic_world(workload, np.random.default_rng(0))
    lin, ang = TDM_Numba(cfg), TDM_Numba(cfg)
    lin.set_TDM_from_PMF_grid(pmf, tdm_dict, obstacle, unknown)
    ang.set_TDM_from_PMF_grid(pmf, tdm_dict, obstacle, unknown)
    planner = MPPI_Numba(cfg, rank=rank, world_size=world)
    params = make_params(workload)
    planner.setup(params, lin, ang)
    return w, cfg, lin, ang, planner, params


def batch_problems(count, rng):
    """Start states and goals spread over the 64 m x 64 m map (c5)."""
    x0s = np.stack([rng.uniform(2, 62, count), rng.uniform(2, 62, count), rng.uniform(-np.pi, np.pi, count)],
                   axis=1).astype(np.float32)
    goals = np.stack([rng.uniform(4, 60, count), rng.uniform(4, 60, count)], axis=1).astype(np.float32)
    return x0s, goals


def algorithmic_bytes(w, n, rp, cp, rollout_writes_noise=True, fused=False):
    """SURVEY.md section 8d.  Returns (bytes per iteration, bytes per rollout-kernel launch).
    Rollout launch, det mode: 8 (noise read) + 4 (map bytes) per rollout-step, + 8 (noise
    write) when it is the pipelined kernel, whose spare workgroups also produce the noise of
    the following iteration.  `fused` (k_rollout_scan with in-launch noise): the launch does the
    noise sampling, the rollout AND the update's pass over the noise -- all 28 bytes per
    rollout-step of the reference's dataflow -- without the noise ever existing in memory; it is
    priced against the same formula, as SURVEY.md 8d prescribes for a fused implementation."""
    t, m = w["t"], w["m"]
    if m == 1:
        risk = 1 if w.get("speed_map") else 0  # + the risk map byte (29 B per rollout-step)
        it = n * t * (28 + risk) + n * 16 + 32 * t + (4 + risk) * rp * cp
        roll = n * t * (8 + 4 + risk + (8 if rollout_writes_noise else 0)) + n * 4 + 8 * t + (4 + risk) * rp * cp
        if fused:
            roll = n * t * (28 + risk) + n * 8 + 8 * t + (4 + risk) * rp * cp
    else:
        it = 4 * n * m * t + 24 * n * t + 16 * n + 2 * m * rp * cp
        roll = 4 * n * m * t + 8 * n * t + 4 * n + 2 * m * rp * cp
    return it, roll


def traffic_key(workload, n_local, math="exact"):
    """Entry of profiles/traffic.json that holds the committed counter passes of THIS configuration: the workload's own
    at its own size; `c4shard` for configs[3]'s shard (`--workload c4 --n 8192`); none for any other override (the line then
    carries traffic: null instead of another configuration's bytes -- VERDICT round 5)."""
    if n_local == WORKLOADS[workload]["n"]:
        return workload + ("_fast" if math == "fast" else "")
    if workload == "c4" and n_local == 8192 and math == "exact":
        return "c4shard"
    return "(no committed counter passes for this override)"


def usable_cores():
    """Cores this process may actually use: affinity mask, capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(w, world_params, lin, ang, planner, budget_s=12.0):
    """The oracle (C restatement of the reference's CPU path) timed on this
    host: noise + rollout + update per iteration, OpenMP over rollouts."""
    from oracle import oracle as O
    n, t, m = w["n"], w["t"], w["m"]
    P = world_params
    p = O.make_params(P, lin.res, lin.padded_xlimits, lin.padded_ylimits,
                      lin.bin_values_bounds_d.copy_to_host(), ang.bin_values_bounds_d.copy_to_host())
    rp, cp = lin.pmf_grid_d.shape[1:]
    lin_g = lin.sample_grid_batch_d.copy_to_host()
    ang_g = ang.sample_grid_batch_d.copy_to_host()
    obs, unk = lin.obstacle_map_d.copy_to_host(), lin.unknown_map_d.copy_to_host()
    risk = lin.risk_traction_map_d.copy_to_host() if w.get("speed_map") else None
    u = planner.u_cur_d.copy_to_host().reshape(-1, t, 2)[0]  # (problem 0 of a batched handle)
    # bounded sample: fewer rollouts for the CVaR workload (N*M*T is 64x the work)
    n_cpu = n if m == 1 else max(64, n // 32)
    states = O.xoroshiro_init(n_cpu * t, 1)
    O.set_num_threads(usable_cores())
    threads = O.num_threads()
    iters, t0 = 0, time.perf_counter()
    while True:
        noise = O.sample_noise(states, P["u_std"], n_cpu, t)
        if m == 1:
            costs = O.rollout_det(p, lin_g, ang_g, obs, unk, noise, u, risk=risk)
        else:
            costs = O.rollout_tdm(p, lin_g, ang_g, obs, unk, noise, u)
        O.update_useq(P["lambda_weight"], costs, noise, P["vrange"], P["wrange"], u)
        iters += 1
        el = time.perf_counter() - t0
        if el > budget_s or iters >= 5000:
            break
    return dict(value=n_cpu * iters / el, unit="rollouts/s", cores=threads, kind="port",
                parity_check=parity_margin(w, p, P, (lin_g, ang_g, obs, unk), planner, risk),
                sample="%d iterations of {xoroshiro noise, rollout, update} on %d of the %d "
                       "control samples (T=%d, M=%d), oracle/liboracle.so with OpenMP over rollouts, "
                       "%.1f s" % (iters, n_cpu, n, t, m, el))


def parity_margin(w, p, P, grids, planner, risk=None):
    """One stage-level iteration of the benchmarked handle against the oracle on the same noise
    and controls (the oracle as the CHECKER): fraction of bit-identical costs and the achieved
    max |du| / control range (bound 1e-5, BASELINE.json north_star)."""
    from oracle import oracle as O
    t = w["t"]
    if planner.num_instances != 1 or planner.world_size != 1:
        return None  # checked per problem / per shard in tests/
    planner.sample_noise()
    noise = planner.noise_samples_d.copy_to_host()
    u_in = planner.u_cur_d.copy_to_host().reshape(t, 2)
    planner.rollout()
    got = planner.costs_d.copy_to_host()
    planner.update()
    u_out = planner.u_cur_d.copy_to_host().reshape(t, 2)
    want = O.rollout_det(p, *grids, noise, u_in, risk=risk) if w["m"] == 1 else O.rollout_tdm(p, *grids, noise, u_in)
    _, u_ref, _ = O.update_useq(P["lambda_weight"], want, noise, P["vrange"], P["wrange"], u_in)
    span = np.array([P["vrange"][1] - P["vrange"][0], P["wrange"][1] - P["wrange"][0]])
    rel = np.abs(got - want) / np.abs(want)
    return dict(costs_bit_identical=float((got.view(np.int32) == want.view(np.int32)).mean()),
                costs_max_rel=float(rel.max()), costs_rel_q999=float(np.quantile(rel, 0.999)),
                costs_within_1e6=float((rel <= 1e-6).mean()),
                u_max_abs_over_range=float((np.abs(u_out - u_ref) / span).max()), u_bound=1e-5)


def reference_cpu_path():
    """BASELINE configs[0] through the reference's own code under the CUDA
    simulator, when this box has both the reference checkout (MPPI_NUMBA_REFERENCE or
    /root/reference) and an interpreter with numba (MPPI_NUMBA_PYTHON or /opt/conda/bin/python3.9).
    Otherwise says which of the two is missing; the figure measured in the build container is kept
    under a key that says what it is."""
    script = os.path.join(ROOT, "oracle", "time_reference_cudasim.py")
    ref_root = os.environ.get("MPPI_NUMBA_REFERENCE") or "/root/reference"
    if os.path.basename(os.path.normpath(ref_root)) == "mppi_numba":
        ref_root = os.path.dirname(os.path.normpath(ref_root))
    py = os.environ.get("MPPI_NUMBA_PYTHON") or "/opt/conda/bin/python3.9"
    have_ref = os.path.isfile(os.path.join(ref_root, "mppi_numba", "mppi.py")) and \
        os.path.isfile(os.path.join(ref_root, "barebone_mppi_numba.ipynb"))
    have_py = os.path.exists(py)
    if have_ref and have_py:
        import subprocess
        try:
            run = subprocess.run([py, script, "--solves", "3", "--reference", ref_root], capture_output=True, text=True,
                                 timeout=300)
            return json.loads(run.stdout.strip().splitlines()[-1])
        except Exception as e:  # noqa: BLE001 -- a diagnostic leg must not take the bench down
            return dict(status="failed: %s" % e)
    missing = []
    if not have_ref:
        missing.append("reference checkout (%s has no mppi_numba/mppi.py + barebone_mppi_numba.ipynb; set MPPI_NUMBA_REFERENCE)" % ref_root)
    if not have_py:
        missing.append("interpreter with numba's CUDA simulator (%s; set MPPI_NUMBA_PYTHON)" % py)
    # (no figure from another machine on the line: the build container's measurement of this leg is on record in
    #  profiles/r02_reference_cudasim.json -- 0.59 s per barebone solve at N=64, T=30 on one core -- and is named, not quoted)
    return dict(status="not measured in this run: missing " + " and ".join(missing),
                measured_elsewhere="profiles/r02_reference_cudasim.json (build container; not a measurement of this run)",
                stands_in="cpu_baseline (the C restatement of the same CPU path, pinned bit for bit to the reference's outputs: "
                          "tests/test_oracle_golden.py) and `--workload bb` (the reference's one published timing)")


BB_PUBLISHED = dict(value_ms=2.74, std_ms=0.425, gpu="RTX 3070 (authors' machine)",
                    source="barebone_mppi_numba.ipynb cell 6 output: `%timeit -n 5 -r 5 mppi_planner.solve()` -> "
                           "2.74 ms +- 425 us per loop (mean +- std. dev. of 5 runs, 5 loops each)")


def barebone_problem():
    """The one configuration the reference publishes a timing for (barebone_mppi_numba.ipynb cell 5): N=1000, T=5.0 s at
    dt=0.1 (50 steps), two disc obstacles, start (0, 0, pi/4), goal (7, 5)."""
    cfg_kwargs = dict(T=5.0, dt=0.1, num_control_rollouts=int(1e3), num_vis_state_rollouts=20, seed=1)
    params = dict(
        dt=0.1, x0=np.array([0, 0, np.pi / 4]), xgoal=np.array([7, 5]), goal_tolerance=0.5, dist_weight=10,
        lambda_weight=1.0, num_opt=1, u_std=np.array([1.0, 1.0]), vrange=np.array([0.0, 2.0]),
        wrange=np.array([-np.pi, np.pi]), obstacle_positions=np.array([[5, 4.5], [2, 1]]),
        obstacle_radius=np.array([1.5, 1]), obs_penalty=1e6)
    return cfg_kwargs, params


def bench_barebone(args, result_fd):
    """`--workload bb`: a step is ONE solve() call of the barebone planner (num_opt = 1: noise + rollout + update + the
    controls back on the host), wall-clocked on the host exactly as the notebook's %timeit does -- parameters packed and
    handed over, kernels, 8*T bytes back.  K solves after W untimed ones; beside it the notebook's own statistic
    (5 runs of 5 loops)."""
    import contextlib
    import io
    from mppi_numba_amd import _lib
    from mppi_numba_amd.barebone import Config as BBConfig, MPPI_Numba as BBPlanner
    import ctypes as C
    cfg_kwargs, params = barebone_problem()
    with contextlib.redirect_stdout(io.StringIO()):
        cfg = BBConfig(**cfg_kwargs)
        planner = BBPlanner(cfg)
        planner.setup(params)
    n, t = cfg.num_control_rollouts, cfg.num_steps
    for _ in range(max(1, args.warmup)):
        planner.solve()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        useq = planner.solve()
    elapsed = time.perf_counter() - t0
    assert useq.shape == (t, 2) and np.isfinite(useq).all()
    runs = []
    for _ in range(5):  # %timeit -n 5 -r 5
        t1 = time.perf_counter()
        for _ in range(5):
            planner.solve()
        runs.append((time.perf_counter() - t1) / 5)
    # per-kernel durations of the three launches of a solve (begin / end timestamps of the dispatches)
    us_roll, us_upd = C.c_float(), C.c_float()
    _lib.call("mppi_planner_time_kernels", planner._handle, None, None, 200, C.byref(us_roll), C.byref(us_upd))
    kernel = C.create_string_buffer(512)
    _lib.call("mppi_planner_describe_last_rollout", planner._handle, kernel, 512)
    ms_per_step = 1e3 * elapsed / args.steps
    bytes_roll = n * t * (8 + 8) + n * 4 + 8 * t  # noise read twice (control cost after the terminal cost), costs out
    bytes_iter = n * t * 24 + n * 16 + 32 * t      # noise write + rollout read + update read (no map)
    out = {
        "metric": "rollouts/sec (MPPI iteration = noise + rollout + update)", "value": n * args.steps / elapsed,
        "unit": "rollouts/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64 intermediates / f32 state (the reference CPU path's roundings)", "data": "synthetic",
        "config": {"workload": "barebone unicycle MPPI (barebone_mppi_numba.ipynb cell 5): N=1000, T=50 steps, two disc obstacles; "
                               "a step = one solve() call wall-clocked on the host, as the notebook's %timeit",
                   "rollouts_per_gpu": n, "horizon_steps": t, "rollout_kernel": kernel.value.decode(),
                   "rng": "Philox4x32-10 counters (rocRAND-identical engine) + hardware Box-Muller", "math": "exact"},
        "solve_ms_timeit_5x5": {"mean": 1e3 * float(np.mean(runs)), "std": 1e3 * float(np.std(runs)),
                                "how": "5 runs of 5 solve() calls each, mean +- std of the per-call time over the runs (%timeit -n 5 -r 5)"},
        "reference_published": BB_PUBLISHED,
        "speedup_vs_published": BB_PUBLISHED["value_ms"] / ms_per_step,
        "speedup_note": "other hardware (the reference publishes no MI355X number): a ratio of wall times per solve() call, "
                        "dominated on both sides by launch and host overhead, not by arithmetic",
        "kernel_us_in_loop": {"rollout": us_roll.value, "update": us_upd.value},
        "roofline": {"bound": "hbm", "kernel": kernel.value.decode().split(" ")[0], "achieved": bytes_roll / (us_roll.value * 1e-6) / 1e9,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": bytes_roll / (us_roll.value * 1e-6) / 1e9 / HBM_PEAK_GBS,
                     "traffic": None, "algorithmic_bytes_per_launch": bytes_roll,
                     "note": "16 waves of one lane per rollout walking 50 dependent steps: a latency measurement, not a bandwidth one"},
        "roofline_iteration": {"bound": "hbm", "achieved": bytes_iter / (ms_per_step * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": bytes_iter / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                               "algorithmic_bytes_per_step": bytes_iter},
    }
    if not args.no_cpu_baseline:
        from oracle import oracle as O
        p = O.make_params(params, 1.0, [0, 0], [0, 0], [0.0, 1.0], [0.0, 1.0], default_obs_cost=1e3, default_dist_weight=10)
        pos, rad = params["obstacle_positions"], params["obstacle_radius"]
        # the oracle as the CHECKER: one stage-level iteration of the benchmarked handle on the same noise and controls
        planner.sample_noise()
        noise, u_in = planner.noise_samples_d.copy_to_host(), planner.u_cur_d.copy_to_host()
        planner.rollout()
        got = planner.costs_d.copy_to_host()
        want = O.rollout_barebone(p, pos, rad, noise, u_in)
        planner.update()
        _, u_ref, _ = O.update_useq(params["lambda_weight"], want, noise, params["vrange"], params["wrange"], u_in)
        span = np.array([params["vrange"][1] - params["vrange"][0], params["wrange"][1] - params["wrange"][0]])
        check = dict(costs_bit_identical=float((got.view(np.int32) == want.view(np.int32)).mean()),
                     u_max_abs_over_range=float((np.abs(planner.u_cur_d.copy_to_host() - u_ref) / span).max()), u_bound=1e-5)
        # ... and as the CPU baseline: the same solve() -- noise, rollout, update -- on the host's cores
        O.set_num_threads(usable_cores())
        states = O.xoroshiro_init(n * t, 1)
        u = np.zeros((t, 2), dtype=np.float32)
        iters, t2 = 0, time.perf_counter()
        while time.perf_counter() - t2 < 10.0 and iters < 20000:
            nz = O.sample_noise(states, params["u_std"], n, t)
            c = O.rollout_barebone(p, pos, rad, nz, u)
            _, u, _ = O.update_useq(params["lambda_weight"], c, nz, params["vrange"], params["wrange"], u)
            iters += 1
        el = time.perf_counter() - t2
        out["cpu_baseline"] = dict(value=n * iters / el, unit="rollouts/s", cores=O.num_threads(), kind="port", parity_check=check,
                                   ms_per_solve=1e3 * el / iters,
                                   sample="%d solves of the barebone problem (xoroshiro noise, rollout, update), oracle/liboracle.so "
                                          "with OpenMP over rollouts, %.1f s" % (iters, el))
        out["cpu_baseline_reference"] = reference_cpu_path()
    sys.stdout.flush()
    os.write(result_fd, (json.dumps(out) + "\n").encode())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS) + ["bb"])
    ap.add_argument("--n", type=int, default=None, help="override the rollouts per GPU (experiments)")
    ap.add_argument("--problems", type=int, default=None, help="c5: problems per GPU (default 64)")
    ap.add_argument("--math", default="exact", choices=["exact", "fast"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--regions", type=int, default=15,
                    help="repeat the K-step timed region this many more times (ms_per_step stays the FIRST region; "
                         "median / min over all regions are reported beside it)")
    ap.add_argument("--debug-flags", type=int, default=0,
                    help="developer switches (include/mppi_hip.h MPPI_DEBUG_*): which kernel variant runs (experiments)")
    ap.add_argument("--graph", type=int, default=0, metavar="ITERATIONS",
                    help="hipGraph replay of the iteration loop, that many (even) iterations per graph; "
                         "0 = direct launches (single GPU; same results)")
    ap.add_argument("--pre-warm", type=int, default=0, dest="pre_warm",
                    help="EXTRA untimed iterations before the W warmup + K timed steps (experiments: the device in its working "
                         "state).  Default 0: ms_per_step is the contract's region -- W warm-up steps, then K timed ones -- "
                         "right after the set-up; the steady state is on the line as ms_per_step_median over --regions")
    ap.add_argument("--exchange", default="auto", choices=["auto", "p2p", "rccl", "host"],
                    help="multi-GPU packet exchange.  p2p: every rank writes its numbers straight into its peers' "
                         "inboxes from inside the rollout launch (IPC-mapped fine-grained memory; no collective, an "
                         "iteration is one launch); rccl: one RCCL all-gather per iteration on the stream; auto "
                         "(default): connect both, time both, report both, the faster one is the line's value; host: "
                         "packets staged over the rendezvous hub (debugging)")
    ap.add_argument("--shard", default="controls", choices=["controls", "samples"],
                    help="multi-GPU c3: what the ranks split -- the N control samples (default; every rank holds all "
                         "M traction maps) or the M traction samples (every rank rolls all N controls over its own "
                         "M maps of a global M*gpus; one all-gather of the (N, M) per-sample costs per step)")
    ap.add_argument("--no-kernel-timing", action="store_true", dest="no_kernel_timing",
                    help="skip the event-bracketed stages and the per-launch timing that follow the timed regions: under "
                         "rocprofv3 the trace then holds the plain loop's launches only (launches that carry events, or that are "
                         "bracketed by them, run differently where a second stream is involved and would pull the averages)")
    ap.add_argument("--single-process", action="store_true",
                    help="--gpus N from ONE process driving N devices (mppi_group_*) instead of one process per GPU")
    args = ap.parse_args()

    from mppi_numba_amd import launch
    if args.gpus > 1 and not args.single_process and not launch.launched_by_a_launcher():
        # plain `python bench.py --gpus N`: start the N ranks (one per GPU) ourselves
        def crashed(per_rank):
            print(json.dumps({"metric": "rollouts/sec (MPPI iteration = noise + rollout + update)", "value": None, "unit": "rollouts/s",
                              "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": None,
                              "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                              "error": "rank(s) %s ended abnormally before the run completed" % ", ".join(sorted(per_rank)),
                              "errors_per_rank": per_rank, "config": {"workload": WORKLOADS[args.workload]["label"]}}), flush=True)
        sys.exit(launch.spawn_ranks(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:], on_failure=crashed))

    # ONE line on stdout: everything else that writes to file descriptor 1 in this process -- librccl prints a version
    # banner when a communicator is created, the mirror classes print like the reference does -- goes to stderr
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    if args.workload == "bb":
        assert args.gpus == 1, "--workload bb: the notebook's single-GPU configuration"
        return bench_barebone(args, result_fd)

    rank, local_rank, world = launch.rank_from_env()
    if args.single_process:
        rank, local_rank, world = 0, 0, 1
    elif world != args.gpus:
        args.gpus = world  # the launcher's word counts

    def give_up(message, per_rank=None):
        """A run that cannot start says so in the ONE line the driver parses (value null, an `error`), from rank 0 -- or
        from whichever rank is left to say it when rank 0 never came up -- and exits 2: never a hang, never nothing."""
        if rank == 0 or per_rank is None:
            line = {"metric": "rollouts/sec (MPPI iteration = noise + rollout + update)", "value": None, "unit": "rollouts/s",
                    "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True,
                    "scaling": "weak", "vs_baseline": None, "error": message, "rank": rank,
                    "errors_per_rank": per_rank, "config": {"workload": WORKLOADS[args.workload]["label"]}}
            os.write(result_fd, (json.dumps(line) + "\n").encode())
        print("bench.py rank %d: %s" % (rank, message), file=sys.stderr)
        sys.exit(2)

    if os.environ.get("MPPI_BENCH_DIE_RANK") == str(rank):  # (fault injection, tests/test_launch_hub.py: a rank that never reports)
        os._exit(3)
    try:
        hub = launch.Hub(rank, world, timeout=float(os.environ.get("MPPI_HUB_TIMEOUT", "120")))
    except Exception as e:  # noqa: BLE001 -- whatever kept the ranks from meeting is the run's result
        give_up("rendezvous of %d ranks failed: %s: %s" % (world, type(e).__name__, e))

    from mppi_numba_amd import _lib
    from mppi_numba_amd.config import Config
    from mppi_numba_amd.mppi import MPPI_Numba, comm_unique_id
    from mppi_numba_amd.terrain import TDM_Numba

    w = dict(WORKLOADS[args.workload])
    if args.n:
        w["n"] = args.n
    n_local, t_steps, m = w["n"], w["t"], w["m"]
    problems = (args.problems or w["problems"]) if "problems" in w else 0
    n_global = n_local if problems else n_local * world  # c5: the ranks are independent
    by_samples = args.shard == "samples" and world > 1
    if args.shard == "samples":
        assert args.workload == "c3" and not args.single_process, "--shard samples: the CVaR workload, one process per GPU"
    m_global = m * world if by_samples else m  # weak scaling in M: every rank keeps its 128 maps
    if by_samples:
        n_global = n_local  # every rank rolls all N control samples
    group_size = args.gpus if args.single_process else 1
    if group_size > 1:
        assert not problems and not args.graph, "--single-process: sharded workloads, direct launches"
        n_global = n_local * group_size

    import contextlib
    import io
    quiet = contextlib.redirect_stdout(io.StringIO())  # the mirror prints like the reference does
    setup_error = None
    try:
      with quiet:
          if os.environ.get("MPPI_BENCH_FAIL_RANK") == str(rank):  # (fault injection: a rank that cannot open its device)
              raise RuntimeError("injected start-up failure (MPPI_BENCH_FAIL_RANK)")
          device = local_rank % max(1, _lib.device_count())
          if group_size > 1:
              assert _lib.device_count() >= group_size, "%d devices for --gpus %d" % (_lib.device_count(), group_size)
          cfg = Config(T=t_steps * 0.1, dt=0.1, num_grid_samples=m_global, num_control_rollouts=n_global,
                       max_speed_padding=5.0, num_vis_state_rollouts=1, max_map_dim=(260, 260),
                       seed=1 + (rank if problems else 0),
                       enforce_recommended_limits=False, math=args.math, device=device, **w["mode"])
          assert cfg.num_steps == t_steps, cfg.num_steps
          world_rng = np.random.default_rng(0)
          pmf, obstacle, unknown, tdm_dict = synthetic_world(args.workload, world_rng)
          shard = (rank, world) if by_samples else None
          lin, ang = TDM_Numba(cfg, sample_shard=shard), TDM_Numba(cfg, sample_shard=shard)
          lin.set_TDM_from_PMF_grid(pmf, tdm_dict, obstacle, unknown)
          ang.set_TDM_from_PMF_grid(pmf, tdm_dict, obstacle, unknown)
          params = make_params(args.workload)
          if problems:
              from mppi_numba_amd.batch import MPPI_Batch
              planner = MPPI_Batch(cfg, problems)
              planner.setup(params, lin, ang, *batch_problems(problems, np.random.default_rng(100 + rank)))
          elif group_size > 1:
              import copy
              from mppi_numba_amd.mppi import MPPI_Group
              cfgs, lins, angs = [cfg], [lin], [ang]
              for g in range(1, group_size):
                  c = copy.deepcopy(cfg)
                  c.device = g
                  lg, ag = TDM_Numba(c), TDM_Numba(c)
                  lg.set_TDM_from_PMF_grid(pmf, tdm_dict, obstacle, unknown)
                  ag.set_TDM_from_PMF_grid(pmf, tdm_dict, obstacle, unknown)
                  cfgs.append(c); lins.append(lg); angs.append(ag)
              group = MPPI_Group(cfgs)
              group.setup(params, lins, angs)
              planner = group.planners[0]
              if args.exchange in ("auto", "p2p"):
                  try:  # the peer exchange between the devices of this process (peer access)
                      group.connect_peers()
                      args.exchange = "p2p"
                  except Exception as e:
                      print("bench.py: peer exchange unavailable in the device group (%s): RCCL" % e, file=sys.stderr)
                      args.exchange = "rccl"
          elif by_samples:
              planner = MPPI_Numba(cfg, sample_shard=shard)
              planner.setup(params, lin, ang)
          else:
              planner = MPPI_Numba(cfg, rank=rank, world_size=world)
              planner.setup(params, lin, ang)
    except Exception as e:  # noqa: BLE001 -- reported below, by every rank together
        setup_error = "%s: %s" % (type(e).__name__, str(e)[:300])
    # every rank says whether it is up BEFORE anything waits for a peer on the device: a rank that could not open its
    # GPU (or build its planner) makes the run a reported error on all of them, not a hang in the first collective
    try:
        setup_errors = hub.all_gather(setup_error)
    except Exception as e:  # noqa: BLE001 -- a rank that went away after the rendezvous
        give_up("a rank went away during start-up (%s: %s)%s" % (type(e).__name__, e, "; this rank: " + setup_error if setup_error else ""))
    if any(setup_errors):
        bad = {str(r): e for r, e in enumerate(setup_errors) if e}
        give_up("%d of %d ranks failed to start: %s" % (len(bad), world, "; ".join("rank %s: %s" % kv for kv in bad.items())), bad)
    rp, cp = lin.pmf_grid_d.shape[1:]

    exchange_note = None
    p2p_ok = False
    if world > 1 and args.exchange in ("auto", "p2p") and not problems and not by_samples:
        err = ""
        try:
            handle = planner.p2p_export()
        except Exception as e:
            handle, err = b"", str(e)
        handles = hub.all_gather(handle)
        if not err and all(len(h) == _lib.P2P_HANDLE_BYTES for h in handles):
            try:
                planner.p2p_connect(handles)
            except Exception as e:
                err = str(e)
        else:
            err = err or "another rank could not export its inbox"
        p2p_ok = not hub.all_max(1 if err else 0)
        if p2p_ok:
            # connected everywhere: can the ranks actually hear each other from inside a running kernel?  (bounded, no trap)
            hub.barrier()
            try:
                heard = planner.p2p_ping(0x5eed0001, timeout_ms=500)
            except Exception as e:
                heard, err = 0, str(e)
            if hub.all_max(0 if heard == world else 1):
                err = err or "ping: %d of %d ranks heard" % (heard, world)
                p2p_ok = False
                planner.p2p_enable(False)
        if p2p_ok:
            # ... and does a short loop over it come back?  A rank whose peers' numbers do not arrive gets MPPI_ERR_COMM
            # after a few seconds (bounded waits, no trap): then every rank starts over with a fresh handle and RCCL.
            try:
                planner.solve()
                planner.iterate_async(12)
                planner.synchronize()
            except Exception as e:
                err = "trial loop: " + str(e)[:200]
            if hub.all_max(1 if err else 0):
                p2p_ok = False
                err = err or "trial loop failed on another rank"
                planner = MPPI_Numba(cfg, rank=rank, world_size=world)
                planner.setup(params, lin, ang)
        if not p2p_ok:
            if not err:
                try:
                    planner.p2p_enable(False)  # (connected here, not everywhere: never use it)
                except Exception:
                    pass
            print("bench.py rank %d: peer exchange unavailable (%s)" % (rank, err or "on another rank"), file=sys.stderr)
            if args.exchange == "p2p":
                args.exchange = "rccl"
    want_rccl = args.exchange == "rccl" or (args.exchange == "auto" and not by_samples) or (args.exchange in ("auto", "p2p") and by_samples)
    rccl_ok = False
    if world > 1 and want_rccl and not problems:
        err = ""
        try:
            uid = comm_unique_id() if rank == 0 else None
        except Exception as e:  # rank 0 could not even load RCCL
            uid, err = None, str(e)
        uid = hub.broadcast(uid)
        if uid is not None:
            try:
                planner.comm_init(uid)
            except Exception as e:
                err = str(e)
        failed = hub.all_max(1 if (err or uid is None) else 0)
        rccl_ok = not failed
        if failed and p2p_ok:
            print("bench.py rank %d: RCCL communicator unavailable (%s): the peer exchange alone" % (rank, err or "on another rank"), file=sys.stderr)
            if err == "" and uid is not None:
                sys.exit("bench.py: this rank has a communicator the others lack")  # (cannot happen on one node)
        elif failed:
            # keep the run alive and say so: same kernels, packets staged through the host
            args.exchange = "host"
            exchange_note = "RCCL communicator unavailable (%s): packets exchanged through the host" % (err or "on another rank")
            print("bench.py rank %d: %s" % (rank, exchange_note), file=sys.stderr)
            if err == "" and uid is not None:
                # this rank did create a communicator the others lack: never use it
                planner = MPPI_Numba(cfg, sample_shard=shard) if by_samples else MPPI_Numba(cfg, rank=rank, world_size=world)
                planner.setup(params, lin, ang)

    if world > 1 and args.exchange == "host" and not problems:
        def iterate(k):  # one launch sequence per iteration, packets over the hub
            gathered = None  # the packets of the previous iteration, not applied yet
            for _ in range(k):
                planner.sample_noise()
                if by_samples:  # the (N, M/G) per-sample cost slabs; the update is then local
                    planner.rollout()
                    planner.sample_costs_apply(np.stack(hub.all_gather(planner.sample_costs_local())))
                    planner.update()
                    continue
                # (a time-parallel rollout launch applies the previous update itself: no k_apply launch)
                if gathered is None:
                    planner.rollout()
                else:
                    planner.update_apply_and_rollout(gathered)
                gathered = np.stack(hub.all_gather(planner.update_local()))
            if gathered is not None:
                planner.update_apply(gathered)
        def solve_staged():  # solve() = sample the traction grids once, then iterate
            lin.sample_grids(params.get("alpha_dyn", 1.0) if w["m"] > 1 else 1.0)
            ang.sample_grids(params.get("alpha_dyn", 1.0) if w["m"] > 1 else 1.0)
            iterate(1)
        planner.iterate_async = iterate
        planner.solve = solve_staged

    runner = group if group_size > 1 else planner  # what iterates: one handle or the device group
    if args.debug_flags:
        planner.set_debug_flags(args.debug_flags)

    def barrier():
        hub.barrier()

    if args.graph and world == 1 and group_size == 1:
        planner.set_graph_replay(True, args.graph)

    # warm start: one full solve (samples the grids) + 10 iterations ...
    runner.solve()
    runner.iterate_async(10)
    runner.synchronize()
    # ---- timed region ------------------------------------------------------------
    def timed_region():
        """W untimed + exactly K timed iterations, barrier + drained stream on both sides (the contract)."""
        runner.iterate_async(args.warmup)
        runner.synchronize()
        barrier()
        t0 = time.perf_counter()
        runner.iterate_async(args.steps)
        runner.synchronize()
        # every rank stops its own clock when ITS stream has drained; the slowest rank's time is the
        # job's.  (The closing barrier -- a star of TCP messages through rank 0 -- is NOT inside the
        # timed region: at 8 ranks it would be a visible share of a 0.5 ms run.  The ranks are already
        # coupled by the exchange of every iteration.)
        own_elapsed = time.perf_counter() - t0
        barrier()
        closing = time.perf_counter() - t0 - own_elapsed
        per_rank = hub.all_gather(own_elapsed) if world > 1 else [own_elapsed]
        return dict(per_rank=per_rank, elapsed=max(per_rank), closing=closing, gpu_ms=planner.last_elapsed_ms())

    # The contract's region: W untimed + K timed iterations right after the warm start.  With --pre-warm P (experiments)
    # the same region is timed AGAIN after P further untimed iterations and that one becomes ms_per_step; the first
    # stays on the line as ms_per_step_cold.  (Round 5's default of P = 2000 made the headline 4 % better than what a
    # caller sees after the warm-up the driver asks for: VERDICT round 5.)
    cold_region = timed_region()
    done = 0
    while done < args.pre_warm:
        k = min(max(1, args.steps), args.pre_warm - done)
        runner.iterate_async(k)
        runner.synchronize()
        done += k

    # several GPUs, both exchanges connected: a TRIAL region with each decides which one the line is measured with;
    # the contract's W + K region then runs ONCE, on the chosen exchange (the faster of two one-shot
    # regions was a best-of-two selection on a 0.3 ms measurement)
    modes = (["p2p"] if p2p_ok else []) + (["rccl"] if rccl_ok else [])
    exchange_us = {}
    if world > 1 and modes and not problems:
        best = None
        for mode in modes:
            if p2p_ok:
                planner.p2p_enable(mode == "p2p")
            r = timed_region()
            exchange_us[mode] = 1e6 * r["elapsed"] / args.steps
            if best is None or r["elapsed"] < best[1]:
                best = (mode, r["elapsed"])
        # (every rank must choose alike: rank 0's clock decides)
        args.exchange = hub.all_gather(best[0])[0] if world > 1 else best[0]
        if p2p_ok:
            planner.p2p_enable(args.exchange == "p2p")
    region = timed_region() if (args.pre_warm > 0 or exchange_us) else cold_region
    own_elapsed, closing_barrier_s, gpu_ms = region["per_rank"][rank] if world > 1 else region["elapsed"], region["closing"], region["gpu_ms"]
    per_rank = region["per_rank"]
    elapsed = region["elapsed"]
    ms_per_step = 1e3 * elapsed / args.steps
    # The contract's number is the region above, one shot (K x ~17 us at C2: a fraction of a millisecond).  The
    # same region again, `--regions` times, each bracketed the same way: the spread IS the measurement's noise.
    region_ms = [ms_per_step]
    for _ in range(max(0, args.regions)):
        barrier()
        t1 = time.perf_counter()
        runner.iterate_async(args.steps)
        runner.synchronize()
        own = time.perf_counter() - t1
        barrier()
        region_ms.append(1e3 * (max(hub.all_gather(own)) if world > 1 else own) / args.steps)
    rollouts_per_step = (problems * n_local * world) if problems else n_global
    if by_samples:
        # weak scaling in M: N control samples x (M * world) traction samples per step, counted in
        # control samples at the workload's own M (= what --shard controls counts for the same work)
        rollouts_per_step = n_local * world
    value = rollouts_per_step * args.steps / elapsed
    rccl_ranks = planner.comm_count()

    # ---- per-kernel durations with HIP events on the planner's stream ---------------
    planner.set_profiling(True)
    stage = dict(noise=0.0, rollout=0.0, update=0.0, collective=0.0)
    reps = 0 if ((world > 1 and args.exchange == "host" and not problems) or group_size > 1 or args.no_kernel_timing) else 50
    for _ in range(reps):
        planner.iterate_async(3)  # the middle iteration is profiled: steady state
        planner.synchronize()
        for k, v in planner.stage_times_ms().items():
            stage[k] += v / reps
    planner.set_profiling(False)
    # the dominant kernels inside the ordinary loop: every launch carries its own start / stop HIP
    # events, which the runtime fills with the dispatch's begin / end timestamps (what rocprofv3
    # --kernel-trace reports); nothing is inserted between the kernels
    # (every rank runs it: with RCCL the iterations it times contain the all-gather, a collective)
    kernel_us = None
    if group_size == 1 and not args.graph and not (world > 1 and args.exchange == "host" and not problems) and not args.no_kernel_timing:
        kernel_us = planner.time_kernels(200)
        if kernel_us[0] <= 0.0:  # (a loop on two streams whose rollout kernel does not stamp its launch: not timed in the loop)
            kernel_us = None

    if rank != 0:
        barrier()
        hub.close()
        return

    kernel_name = planner.last_rollout_kernel().split(" ")[0]
    fused = kernel_name.startswith("k_rollout_scan") and "noise=in-kernel" in planner.last_rollout_kernel()
    bytes_iter, bytes_roll = algorithmic_bytes(w, n_local * max(1, problems), rp, cp,
                                               rollout_writes_noise=(kernel_name in ("k_rollout_pipe",)),
                                               fused=fused)
    traffic = None
    try:  # HBM bytes per launch of the dominant kernel, from the committed rocprofv3 PMC passes
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as fh:
            traffic = json.load(fh).get(traffic_key(args.workload, n_local, args.math), {}).get("dominant_kernel_hbm_bytes_per_launch")
    except (OSError, ValueError):
        pass
    roll_s = kernel_us[0] * 1e-6 if kernel_us else stage["rollout"] * 1e-3
    # (host-staged exchange / device group: the loop is driven stage by stage from Python, no kernel is timed)
    achieved = bytes_roll / roll_s / 1e9 if roll_s > 0 else None
    out = {
        "metric": "rollouts/sec (MPPI iteration = noise + rollout + update)",
        "value": value, "unit": "rollouts/s", "n_gpus": world * group_size, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f64 intermediates / f32 state (the reference CPU path's roundings)"
        if args.math == "exact" else "f32 (tolerance mode: float64 only across chunk sums; costs accumulated in the reference's order)",
        "data": "synthetic",
        "config": {"workload": w["label"], "rollouts_per_gpu": n_local, "global_rollouts": n_global,
                   "horizon_steps": t_steps, "traction_samples": m_global, "traction_samples_per_gpu": m,
                   "padded_grid": [int(rp), int(cp)],
                   "rng": "Philox4x32-10 counters (rocRAND-identical engine) + hardware Box-Muller", "math": args.math,
                   "problems_per_gpu": problems or 1,
                   "graph_replay_iterations": args.graph if world == 1 else 0,
                   "rollout_kernel": planner.last_rollout_kernel(),
                   "sharding": "independent problems over ranks, no exchange" if problems else
                               ("traction samples over ranks: every rank rolls all N controls over its M maps, 1 all-gather "
                                "of the (N, M) f32 per-sample costs per step, CVaR + update replicated; value counts "
                                "N x (M_global / M) control samples" if by_samples else
                                "control samples over ranks, 1 all-gather of (2T+2) f64 per step"),
                   "exchange": "none" if ((world == 1 and group_size == 1) or problems) else
                               ("peer exchange: every rank writes its numbers for step t into its peers' inboxes from inside "
                                "the rollout launch (IPC-mapped fine-grained memory), no collective, one launch per step"
                                if args.exchange == "p2p" else
                                "RCCL all-gather on the planner's stream" if args.exchange == "rccl" else
                                (exchange_note or "host-staged through the rendezvous hub (--exchange host)")),
                   "exchange_us_per_step": exchange_us or None,
                   # one entry per transport that connected (a trial region each; the line's own region ran on `exchange`):
                   # the first 8-GPU run reports both legs whichever wins
                   "per_exchange": {mode: {"us_per_step": us, "rollouts_per_s": rollouts_per_step / (us * 1e-6),
                                           **({"n_ranks_seen_by_rccl": rccl_ranks} if mode == "rccl" else
                                              {"inbox_memory": planner.p2p_stats()["inbox"] if hasattr(planner, "p2p_stats") else None})}
                                    for mode, us in exchange_us.items()} or None,
                   "pre_warm_iterations": args.pre_warm,
                   "n_ranks_seen_by_rccl": rccl_ranks,
                   "launcher": ("one process, %d devices (mppi_group_*)" % group_size) if group_size > 1 else
                               ("one process per GPU, %s" % ("external launcher (RANK/WORLD_SIZE)"
                                                             if "MPPI_RDZV_FILE" not in os.environ else
                                                             "started by bench.py itself")) if world > 1 else "single process"},
        "ms_per_step_cold": 1e3 * cold_region["elapsed"] / args.steps,
        "ms_per_step_cold_note": "the W + K region timed right after the warm start (one solve + 10 iterations); it IS ms_per_step "
                                 "unless --pre-warm iterations (or a multi-GPU exchange trial) were run before a second region",
        "ms_per_step_median": float(np.median(region_ms)), "ms_per_step_min": float(np.min(region_ms)),
        "ms_per_step_regions": region_ms,
        "gpu_ms_per_step_events": gpu_ms / args.steps,
        "ms_per_step_per_rank": [1e3 * e / args.steps for e in per_rank],
        "closing_barrier_ms": 1e3 * closing_barrier_s,
        "timing_note": "ms_per_step = the slowest rank's wall time of its own K steps (stream drained), over K; the closing "
                       "barrier is outside the timed region and reported beside it",
        "event_bracketed_ms": stage,
        "event_bracketed_ms_note": "each stage of ONE iteration bracketed by its own HIP events on the planner's stream; every "
                          "bracket adds ~3 us of event overhead, so the stages sum to more than ms_per_step (which has "
                          "no events inside the loop); kernel_us_in_loop has no such overhead",
        "kernel_us_in_loop": None if not kernel_us else
            {"rollout": kernel_us[0], "update": kernel_us[1],
             "how": "200 ordinary iterations of one call; every rollout / update launch carries its own start / stop HIP "
                    "events (hipExtLaunchKernelGGL: the dispatch's begin / end timestamps, as rocprofv3 --kernel-trace "
                    "reports them; averages agree with profiles/).  A loop that runs the next iteration's noise on a second "
                    "stream (N*T >= 4M: ns, c4) is timed WITHOUT events -- they shift such a loop (mppi_api.hip, "
                    "mppi_planner_time_kernels) -- by its kernels themselves: first wave in to last wave out on the device's "
                    "100 MHz clock.  `update` is the update launches' time per ITERATION: "
                    "when the next rollout launch applies the update itself (rollout_kernel says applies_update=1) only "
                    "the last iteration of a call has an update launch and its time is spread over all of them"},
        # the line must follow from its parts: the dominant launches of an iteration cannot last longer than the iteration
        "accounting": None if not kernel_us else
            {"rollout_plus_update_us": kernel_us[0] + kernel_us[1], "step_us": 1e3 * ms_per_step,
             "ok": bool(kernel_us[0] + kernel_us[1] <= 1.05 * 1e3 * max(ms_per_step, float(np.median(region_ms))))},
        "roofline": {"bound": "hbm",
                     "kernel": kernel_name + (" (rollout + next iteration's noise)" if kernel_name in ("k_rollout_pipe",) else
                                              " (noise + rollout + per-tile update sums)" if fused else ""),
                     "fused": fused,
                     "fused_note": ("the launch samples the noise, rolls out and reduces the update per tile; the noise never "
                                    "exists in memory, so `traffic` is a small fraction of `algorithmic_bytes_per_launch` and "
                                    "the kernel is bound by instruction issue and by the walks of the float32-rounded running "
                                    "sums, not by HBM (priced against the reference's dataflow as SURVEY.md 8d prescribes)") if fused else None,
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": None if achieved is None else achieved / HBM_PEAK_GBS, "traffic": traffic,
                     # (for a fused launch `achieved` is a speed-of-job index, not bus utilisation)
                     "algorithmic_equivalent_GBps": achieved,
                     "hbm_GBps_measured": None if (traffic is None or roll_s <= 0) else traffic / roll_s / 1e9,
                     "traffic_source": "profiles/traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                                       "kernel on this workload (committed; not re-measured in this run)",
                     "algorithmic_bytes_per_launch": bytes_roll,
                     "launch_ms": roll_s * 1e3 if roll_s > 0 else None,
                     "duration_source": "kernel_us_in_loop.rollout" if kernel_us else
                                        ("event_bracketed_ms.rollout" if roll_s > 0 else
                                         "not measured in this mode (stage-level loop driven from Python: --exchange host / --single-process)")},
        "roofline_iteration": {"bound": "hbm", "achieved": bytes_iter / (ms_per_step * 1e-3) / 1e9,
                               "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": bytes_iter / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                               "algorithmic_bytes_per_step": bytes_iter},
    }
    if not args.no_cpu_baseline and world * group_size == 1:
        # (the CPU legs are timed at N = 1 only: with several ranks the others would sit at the closing barrier for it)
        with quiet:
            out["cpu_baseline"] = cpu_baseline(w, params, lin, ang, planner)
        out["cpu_baseline_reference"] = reference_cpu_path()
    sys.stdout.flush()
    os.write(result_fd, (json.dumps(out) + "\n").encode())
    barrier()
    hub.close()


if __name__ == "__main__":
    main()
