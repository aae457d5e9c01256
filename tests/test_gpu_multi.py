"""Multi-GPU plumbing on a ONE-GPU box: everything except more than one physical device.
  * bench.py --gpus 2 launched plainly starts its two ranks itself, they rendezvous over the hub
    and (sharing the only device, where RCCL refuses two ranks) exchange their packets through
    the host: the launcher, the sharding and the JSON contract run end to end;
  * the one-process group API with a group of one device (RCCL groups, deferred exchange);
  * hipGraph replay of a handle that HAS a communicator: the all-gather is captured."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from test_gpu_scale import build

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args, env_extra=None, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MPPI_RDZV_FILE")}
    env.update(env_extra or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), capture_output=True,
                         text=True, timeout=timeout, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0]), out.stderr


def test_bench_self_launches_two_ranks_on_one_device():
    res, err = run_bench("--gpus", "2", "--steps", "10", "--warmup", "2", "--n", "2048", "--no-cpu-baseline",
                         "--exchange", "host")
    assert res["n_gpus"] == 2 and res["config"]["global_rollouts"] == 4096 and res["config"]["rollouts_per_gpu"] == 2048
    assert "started by bench.py itself" in res["config"]["launcher"]
    assert "host" in res["config"]["exchange"] and res["config"]["n_ranks_seen_by_rccl"] == 0
    assert res["value"] > 0 and res["scaling"] == "weak"


def test_bench_eight_ranks_on_one_device_times_its_own_steps():
    """The driver's 8-GPU line is `--gpus 8 --steps 20`: ~0.5 ms of GPU work.  ms_per_step must be
    the slowest rank's own K steps (stream drained) -- the closing barrier, 2 x 7 TCP hops through
    rank 0, is reported beside it, not inside it."""
    res, err = run_bench("--gpus", "8", "--steps", "20", "--warmup", "5", "--n", "1024", "--no-cpu-baseline",
                         "--exchange", "host", timeout=900)
    assert res["n_gpus"] == 8 and res["config"]["global_rollouts"] == 8192
    per_rank = res["ms_per_step_per_rank"]
    assert len(per_rank) == 8 and min(per_rank) > 0
    assert abs(res["ms_per_step"] - max(per_rank)) <= 1e-9 * max(per_rank)
    assert res["closing_barrier_ms"] >= 0.0
    # value = the whole job's rollouts over the slowest rank's time
    assert abs(res["value"] - 8192 * 20 / (res["ms_per_step"] * 1e-3 * 20)) <= 1e-6 * res["value"]
    # the ranks share one GPU and one hub: their own times agree within a quarter of the slowest
    assert (max(per_rank) - min(per_rank)) <= 0.25 * max(per_rank), per_rank


def test_bench_two_ranks_sharding_the_traction_samples():
    """north_star's other split: the M traction-map samples of the CVaR workload over the ranks
    (two processes on this box's one GPU, slabs exchanged through the host hub)."""
    res, err = run_bench("--gpus", "2", "--workload", "c3", "--shard", "samples", "--steps", "4", "--warmup", "1",
                         "--n", "256", "--no-cpu-baseline", "--exchange", "host")
    cfg = res["config"]
    assert res["n_gpus"] == 2 and cfg["traction_samples"] == 256 and cfg["traction_samples_per_gpu"] == 128
    assert cfg["global_rollouts"] == 256 and "traction samples over ranks" in cfg["sharding"]
    assert "k_rollout_tdm" in cfg["rollout_kernel"] and res["value"] > 0


def test_bench_under_an_external_launcher_environment():
    """The contract's launch line exports RANK / LOCAL_RANK / WORLD_SIZE: with world 1 that must
    simply be the single-GPU run, with the JSON's rccl rank count reported."""
    res, _ = run_bench("--gpus", "1", "--steps", "10", "--warmup", "2", "--n", "1024", "--no-cpu-baseline",
                       env_extra=dict(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29517"))
    assert res["n_gpus"] == 1 and res["config"]["exchange"] == "none" and res["config"]["n_ranks_seen_by_rccl"] == 0


def test_group_of_one_device_equals_the_plain_handle():
    from mppi_numba_amd.mppi import MPPI_Group
    _, cfg, lin, ang, plain, params = build("c2", 2048)
    group = MPPI_Group([cfg])
    group.setup(params, [lin], [ang])
    assert group.planners[0].comm_count() == 1
    want = plain.solve()
    got = group.solve()
    assert np.array_equal(got, want)
    plain.iterate_async(6)
    plain.synchronize()
    group.iterate_async(6)
    group.synchronize()
    assert np.array_equal(group.planners[0].u_cur_d.copy_to_host(), plain.u_cur_d.copy_to_host())
    assert np.array_equal(group.planners[0].costs_d.copy_to_host(), plain.costs_d.copy_to_host())


@pytest.mark.parametrize("workload,n", [("c2", 2048), ("c4", 65536)])
def test_graph_replay_captures_the_rccl_all_gather(workload, n):
    from mppi_numba_amd.mppi import comm_unique_id
    _, _, _, _, direct, _ = build(workload, n)
    _, _, _, _, graph, _ = build(workload, n)
    for planner in (direct, graph):
        planner.comm_init(comm_unique_id())
        assert planner.comm_count() == 1
    graph.set_graph_replay(True, 2)
    for planner in (direct, graph):
        planner.solve()
        planner.iterate_async(9)
        planner.synchronize()
    stats = graph.graph_stats()
    assert stats["captures"] >= 1 and stats["replays"] >= 3, stats
    assert np.array_equal(direct.u_cur_d.copy_to_host(), graph.u_cur_d.copy_to_host())
    assert np.array_equal(direct.costs_d.copy_to_host(), graph.costs_d.copy_to_host())


def _cvar_build(n, m, shard=None, seed=1):
    """C3-shaped CVaR problem (bench.py's world) with the traction samples optionally sharded."""
    import bench
    from mppi_numba_amd.config import Config
    from mppi_numba_amd.mppi import MPPI_Numba
    from mppi_numba_amd.terrain import TDM_Numba
    w = dict(bench.WORKLOADS["c3"])
    cfg = Config(T=w["t"] * 0.1, dt=0.1, num_grid_samples=m, num_control_rollouts=n, max_speed_padding=5.0,
                 num_vis_state_rollouts=1, max_map_dim=(260, 260), seed=seed, enforce_recommended_limits=False,
                 **w["mode"])
    pmf, obstacle, unknown, tdm_dict = bench.synthetic_world("c3", np.random.default_rng(0))
    lin, ang = TDM_Numba(cfg, sample_shard=shard), TDM_Numba(cfg, sample_shard=shard)
    lin.set_TDM_from_PMF_grid(pmf, tdm_dict, obstacle, unknown)
    ang.set_TDM_from_PMF_grid(pmf, tdm_dict, obstacle, unknown)
    planner = MPPI_Numba(cfg, sample_shard=shard)
    params = bench.make_params("c3")
    planner.setup(params, lin, ang)
    return lin, ang, planner, params


@pytest.mark.parametrize("n,m,shards", [(256, 128, 2), (256, 128, 4), (128, 1024, 8), (64, 24, 2), (64, 4096, 2)])
def test_sample_sharded_cvar_has_the_bits_of_the_unsharded_launch(n, m, shards):
    """SURVEY.md 8e / north_star: the M traction samples sharded over G ranks (here G handles on one
    GPU, the all-gather done by hand): the shards' grids are the unsharded draws, the gathered
    per-sample costs are the unsharded ones, and every rank's CVaR costs and control update have
    the BITS of the single-GPU launch -- over several iterations."""
    lin, ang, full, params = _cvar_build(n, m)
    parts = [_cvar_build(n, m, shard=(r, shards)) for r in range(shards)]
    u0 = np.zeros((full.num_steps, 2), dtype=np.float32)
    full.record_sample_costs()
    for _, _, p, _ in parts:
        p.record_sample_costs()
    for step in range(3):
        handles = [(lin, ang, full)] + [(a, b, c) for a, b, c, _ in parts]
        for tl, ta, p in handles:
            if step == 0:
                p.set_u(u0)
            tl.sample_grids(params["alpha_dyn"])  # (solve() does this once per call)
            ta.sample_grids(params["alpha_dyn"])
            p.sample_noise()
            p.rollout()
        ml = m // shards
        want_grid = lin.sample_grid_batch_d.copy_to_host()
        for r, (tl, ta, p, _) in enumerate(parts):
            np.testing.assert_array_equal(tl.sample_grid_batch_d.copy_to_host(), want_grid[r * ml:(r + 1) * ml])
            np.testing.assert_array_equal(p.noise_samples_d.copy_to_host(), full.noise_samples_d.copy_to_host())
        slabs = np.stack([p.sample_costs_local() for _, _, p, _ in parts])
        assert slabs.shape == (shards, n, ml)
        want_sc = full.sample_costs()
        np.testing.assert_array_equal(np.concatenate(list(slabs), axis=1), want_sc)
        full.update()
        for _, _, p, _ in parts:
            p.sample_costs_apply(slabs)
            np.testing.assert_array_equal(p.costs_d.copy_to_host(), full.costs_d.copy_to_host())
            np.testing.assert_array_equal(p.sample_costs(), want_sc)
            p.update()
            np.testing.assert_array_equal(p.u_cur_d.copy_to_host(), full.u_cur_d.copy_to_host())
    assert np.abs(full.u_cur_d.copy_to_host()).max() > 0


def test_sample_sharded_loop_needs_a_communicator_and_runs_with_one():
    """solve() of a sample-sharded handle refuses to run without RCCL; with a communicator of its
    own size (1 rank here is all one GPU offers: a degenerate shard layout cannot be declared, so
    the error path is what this box can show) the message names the way out."""
    from mppi_numba_amd._lib import MppiError
    lin, ang, p, params = _cvar_build(64, 8, shard=(0, 2))
    with pytest.raises(MppiError, match="no communicator"):
        p.solve()


def test_stage_level_update_of_a_sample_shard_needs_the_exchange_first():
    """rollout() of a sample-sharded handle leaves the CVaR over ITS samples in costs_d: an update()
    from that would give every rank another u without any error.  It is refused
    until the slabs of all shards have been applied."""
    from mppi_numba_amd._lib import MppiError
    parts = [_cvar_build(64, 8, shard=(r, 2)) for r in range(2)]
    for tl, ta, p, params in parts:
        tl.sample_grids(params["alpha_dyn"])
        ta.sample_grids(params["alpha_dyn"])
        p.sample_noise()
        p.rollout()
    with pytest.raises(MppiError, match="exchange"):
        parts[0][2].update()
    with pytest.raises(MppiError, match="exchange"):
        parts[0][2].update_local()
    slabs = np.stack([p.sample_costs_local() for _, _, p, _ in parts])
    for _, _, p, _ in parts:
        p.sample_costs_apply(slabs)
        p.update()
    np.testing.assert_array_equal(parts[0][2].u_cur_d.copy_to_host(), parts[1][2].u_cur_d.copy_to_host())


def test_sample_shards_need_the_counter_based_generator_and_even_offsets():
    """The shards' draws are the unsharded ones because a draw's Philox counter is its global sample
    index; the numba-compatible xoroshiro streams cannot be split that way, and a Philox block
    serves a pair of samples."""
    import bench
    from mppi_numba_amd import _lib
    from mppi_numba_amd._lib import MppiError
    from mppi_numba_amd.config import Config
    from mppi_numba_amd.terrain import TDM_Numba
    cfg = Config(T=1.0, dt=0.1, num_grid_samples=8, num_control_rollouts=64, max_speed_padding=5.0,
                 num_vis_state_rollouts=1, max_map_dim=(260, 260), seed=1, enforce_recommended_limits=False,
                 rng="xoroshiro", use_tdm=True)
    with pytest.raises(MppiError, match="counter-based"):
        TDM_Numba(cfg, sample_shard=(1, 2))
    cfg.rng = "philox"
    tdm = TDM_Numba(cfg, sample_shard=(1, 2))
    with pytest.raises(MppiError, match="even"):
        _lib.call("mppi_tdm_set_sample_shard", tdm._handle, 3)
    with pytest.raises(AssertionError):
        TDM_Numba(cfg, sample_shard=(0, 3))  # 8 samples do not split into 3 shards of whole pairs


# ---- the update of a sharded iteration applied by the NEXT rollout launch (update_kernels.h, PendingApply) ----

@pytest.mark.parametrize("math", ["exact", "fast"])
@pytest.mark.parametrize("world", [2, 8])
def test_rollout_launch_applies_the_gathered_update_same_bits(world, math):
    """Control samples over 2 / 8 ranks on the one GPU, packets gathered by hand.  One set of shards
    runs rollout -> rank packet -> k_apply; the other hands the gathered packets to its next rollout
    launch (update_apply_and_rollout): every wave forms the 8 controls it owns with k_apply's
    expressions.  Same costs every iteration, same u on every rank, bit for bit."""
    import bench

    def shards():
        out = [bench.build_planner("c2", 8192 // world, rank=r, world=world, math=math)[4] for r in range(world)]
        for s in out:
            s.lin_tdm.sample_grids()  # (solve() does this; the stage-level calls do not)
            s.ang_tdm.sample_grids()
        return out

    classic, folded = shards(), shards()
    packets_folded = None
    for it in range(4):
        packets = []
        for s in classic:
            s.sample_noise()
            s.rollout()
            packets.append(s.update_local())
        for s in classic:
            s.update_apply(np.stack(packets))
        gathered = []
        for s in folded:
            s.sample_noise()
            if packets_folded is None:
                s.rollout()
            else:
                s.update_apply_and_rollout(packets_folded)
                assert "applies_update=1" in s.last_rollout_kernel(), s.last_rollout_kernel()
            gathered.append(s.update_local())
        packets_folded = np.stack(gathered)
        assert np.array_equal(packets_folded, np.stack(packets)), "iteration %d: rank packets differ" % it
        for a, b in zip(classic, folded):
            assert np.array_equal(a.costs_d.copy_to_host(), b.costs_d.copy_to_host())
    for s in folded:
        s.update_apply(packets_folded)
    for a, b in zip(classic, folded):
        assert np.array_equal(a.u_cur_d.copy_to_host(), b.u_cur_d.copy_to_host())
        assert np.array_equal(a.u_prev_d.copy_to_host(), b.u_cur_d.copy_to_host())
    assert np.array_equal(folded[0].u_cur_d.copy_to_host(), folded[-1].u_cur_d.copy_to_host())
    w_a, w_b = classic[0].weights_d.copy_to_host(), folded[0].weights_d.copy_to_host()
    assert np.array_equal(w_a, w_b)


def test_a_rollout_that_cannot_apply_the_update_gets_it_applied_first():
    """update_apply_and_rollout on a map the time-parallel kernels do not take (traction changes from
    cell to cell, many tiles per CU): plain k_apply, then the rollout."""
    import bench
    world = 2
    a = [bench.build_planner("c4", 32768, rank=r, world=world)[4] for r in range(world)]
    b = [bench.build_planner("c4", 32768, rank=r, world=world)[4] for r in range(world)]
    for s in a + b:
        s.lin_tdm.sample_grids()
        s.ang_tdm.sample_grids()
        s.sample_noise()
        s.rollout()
    pa, pb = np.stack([s.update_local() for s in a]), np.stack([s.update_local() for s in b])
    assert np.array_equal(pa, pb)
    for s in a:
        s.update_apply(pa)
        s.sample_noise()
        s.rollout()
    for s in b:
        s.sample_noise()
        s.update_apply_and_rollout(pb)
        assert "applies_update" not in s.last_rollout_kernel()
    for x, y in zip(a, b):
        assert np.array_equal(x.u_cur_d.copy_to_host(), y.u_cur_d.copy_to_host())
        assert np.array_equal(x.costs_d.copy_to_host(), y.costs_d.copy_to_host())


@pytest.mark.parametrize("math", ["exact", "fast"])
def test_loop_with_a_communicator_leaves_the_update_to_the_next_rollout(math):
    """iterate_async() of a handle with a communicator (one rank is all this box can give RCCL): between
    iterations no k_apply is launched -- rollout, rank packet, all-gather -- and u has the bits of the
    loop that launches k_apply every iteration, ordinary and graph-replayed."""
    import time
    from mppi_numba_amd import _lib
    from mppi_numba_amd.mppi import comm_unique_id
    handles = {}
    for name, flags, graph in [("local", 0, False), ("folded", 0, False), ("k_apply", _lib.DEBUG_NO_FOLDED_APPLY, False),
                               ("folded graph", 0, True)]:
        _, _, _, _, pl, _ = build("c2", None, math=math)
        if name != "local":
            pl.comm_init(comm_unique_id())
        pl.set_debug_flags(flags)
        if graph:
            pl.set_graph_replay(True, 4)
        pl.solve()
        pl.iterate_async(13)
        pl.synchronize()
        handles[name] = pl
    assert "applies_update=1" in handles["folded"].last_rollout_kernel(), handles["folded"].last_rollout_kernel()
    assert "applies_update" not in handles["k_apply"].last_rollout_kernel()
    assert handles["folded graph"].graph_stats()["replays"] >= 2
    u = {k: v.u_cur_d.copy_to_host() for k, v in handles.items()}
    for k in ("folded", "k_apply", "folded graph"):
        assert np.array_equal(u[k], u["local"]), k
        assert np.array_equal(handles[k].costs_d.copy_to_host(), handles["local"].costs_d.copy_to_host()), k
    # and what it buys: microseconds per iteration of the three loops
    times = {}
    for k in ("local", "folded", "k_apply"):
        pl = handles[k]
        pl.iterate_async(50)
        pl.synchronize()
        t0 = time.perf_counter()
        pl.iterate_async(400)
        pl.synchronize()
        times[k] = (time.perf_counter() - t0) / 400 * 1e6
    print("\n%s: us per iteration -- unsharded %.2f, 1-rank communicator with the update left to the rollout %.2f, "
          "with k_apply %.2f" % (math, times["local"], times["folded"], times["k_apply"]))
    assert times["folded"] <= times["k_apply"] + 0.5


def test_graphs_of_a_sharded_handle_survive_odd_call_lengths():
    """Every sharded iteration flips the handle's two control buffers; a call with an odd number of
    iterations leaves the other one current.  One cached graph per parity (of the noise double buffer
    and of the control buffers): calls of odd length alternate between cached graphs, they do not
    re-capture every time."""
    from mppi_numba_amd.mppi import comm_unique_id
    _, _, _, _, direct, _ = build("c2", 2048)
    _, _, _, _, graph, _ = build("c2", 2048)
    for planner in (direct, graph):
        planner.comm_init(comm_unique_id())
    graph.set_graph_replay(True, 2)
    for planner in (direct, graph):
        planner.solve()
    for _ in range(8):
        for planner in (direct, graph):
            planner.iterate_async(3)
            planner.synchronize()
        assert np.array_equal(direct.u_cur_d.copy_to_host(), graph.u_cur_d.copy_to_host())
    stats = graph.graph_stats()
    assert stats["replays"] >= 7 and stats["captures"] <= 4, stats
