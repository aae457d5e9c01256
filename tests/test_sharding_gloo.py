"""world_size = 2 over gloo on CPU: the ALGEBRA of the sharded control update (not the kernels).

Each rank owns half of the control samples, reduces them to one packet
{beta_g, den_g, num_g[T][2]} (what k_weights / k_wsum / k_finish produce on the
GPU), the packets are all-gathered (RCCL on the GPU, gloo here) and every rank
applies the combine of k_apply.  The result must equal the reference's
single-block update (oracle) of the unsharded problem, on every rank, bit for
bit equal between ranks.  This is the algorithm of update_kernels.h restated in
numpy on the test side; the GPU test test_sharded_update_equals_single_gpu runs
the kernels themselves.
"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def local_packet(costs, noise, lam):
    """packet of one shard, float64 as on the device"""
    beta = np.float32(costs.min())
    w = np.exp(-1.0 / np.float64(np.float32(lam)) * (costs - beta).astype(np.float32).astype(np.float64))
    w = w.astype(np.float32).astype(np.float64)                     # weights are stored float32
    num = np.einsum("n,ntc->tc", w, noise.astype(np.float64))
    return np.concatenate([[np.float64(beta), w.sum()], num.ravel()])


def apply_packets(packets, lam, u, vrange, wrange):
    t = (packets.shape[1] - 2) // 2
    beta = packets[:, 0].min()
    scale = np.exp(-1.0 / np.float64(np.float32(lam)) * (packets[:, 0] - beta))
    den = (scale * packets[:, 1]).sum()
    num = (scale[:, None] * packets[:, 2:]).sum(axis=0).reshape(t, 2)
    out = u.astype(np.float32) + (num / den).astype(np.float32)
    out[:, 0] = np.clip(out[:, 0], np.float32(vrange[0]), np.float32(vrange[1]))
    out[:, 1] = np.clip(out[:, 1], np.float32(wrange[0]), np.float32(wrange[1]))
    return out.astype(np.float32)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from helpers import golden, iterations, params_from_golden
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        worst = 0.0
        for name in ("speedmap_cvar", "det_cvar", "tdm_cvar_odd"):
            g = golden(name)
            P = params_from_golden(g)
            for it in iterations(g):
                n = it["costs"].shape[0]
                lo, hi = rank * n // world, (rank + 1) * n // world
                packet = local_packet(it["costs"][lo:hi], it["noise"][lo:hi], P["lambda_weight"])
                gathered = [torch.zeros(packet.size, dtype=torch.float64) for _ in range(world)]
                dist.all_gather(gathered, torch.from_numpy(packet))
                packets = np.stack([t.numpy() for t in gathered])
                u = apply_packets(packets, P["lambda_weight"], it["u_in"], P["vrange"], P["wrange"])
                _, want, _ = O.update_useq(P["lambda_weight"], it["costs"], it["noise"], P["vrange"],
                                           P["wrange"], it["u_in"])
                worst = max(worst, float(np.abs(u - want).max()))
                # every rank must hold the same bits
                mine = torch.from_numpy(u.copy())
                others = [torch.zeros_like(mine) for _ in range(world)]
                dist.all_gather(others, mine)
                assert all(torch.equal(o, mine) for o in others)
        q.put((rank, worst))
    finally:
        dist.destroy_process_group()


def test_packet_algebra_of_the_sharded_update_over_gloo_world2():
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, worst in results:
        assert worst <= 1e-5 * np.pi, (rank, worst)


def cvar_of_all_samples(per_sample, alpha):
    """mppi.py:716-755 on the gathered (N, M) per-sample costs: sort descending when alpha < 1,
    float32 strided tree over the first ceil(M*alpha), float64 division (what k_cvar_reduce does)."""
    n, m = per_sample.shape
    numel = min(m, max(1, int(np.ceil(m * float(np.float32(alpha))))))
    sc = per_sample.astype(np.float32).copy()
    if np.float32(alpha) < 1.0:
        sc = -np.sort(-sc, axis=1)
    s = 1
    while s < numel:
        for i in range(0, m, 2 * s):
            if i + s < numel:
                sc[:, i] = sc[:, i] + sc[:, i + s]
        s *= 2
    return (sc[:, 0].astype(np.float64) / numel).astype(np.float32)


def _sample_worker(rank, world, port, q):
    """The M traction samples sharded over the ranks: every rank rolls all N controls
    over ITS grids, one all-gather of the (N, M/G) cost slabs, every rank reduces all M."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from helpers import golden, iterations, params_from_golden
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        checked = 0
        for name in ("tdm_cvar", "tdm_cvar_odd", "tdm_mean_alpha_dyn", "tdm_odd_units"):
            g = golden(name)
            P = params_from_golden(g)
            p = O.make_params(dict(P, x0=g["solve0_x0"]), g["lin_res"], g["lin_padded_xlimits"], g["lin_padded_ylimits"],
                              g["lin_bin_values_bounds"], g["ang_bin_values_bounds"])
            it = iterations(g)[0]
            lin, ang = g["solve0_lin_sample_grid"], g["solve0_ang_sample_grid"]
            m = lin.shape[0]
            if m % world:
                continue
            lo, hi = rank * m // world, (rank + 1) * m // world
            _, slab = O.rollout_tdm(p, lin[lo:hi], ang[lo:hi], g["lin_obstacle_map_padded"], g["lin_unknown_map_padded"],
                                    it["noise"], it["u_in"], want_per_sample=True)
            gathered = [torch.zeros(slab.shape, dtype=torch.float32) for _ in range(world)]
            dist.all_gather(gathered, torch.from_numpy(np.ascontiguousarray(slab)))
            per_sample = np.concatenate([t.numpy() for t in gathered], axis=1)
            costs = cvar_of_all_samples(per_sample, P["cvar_alpha"])
            np.testing.assert_array_equal(costs, it["costs"])  # the REFERENCE's costs, bit for bit
            _, u, _ = O.update_useq(P["lambda_weight"], costs, it["noise"], P["vrange"], P["wrange"], it["u_in"])
            _, want, _ = O.update_useq(P["lambda_weight"], it["costs"], it["noise"], P["vrange"], P["wrange"], it["u_in"])
            np.testing.assert_array_equal(u, want)
            checked += 1
        q.put((rank, checked))
    finally:
        dist.destroy_process_group()


def test_sample_sharded_cvar_over_gloo_world2():
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    q = ctx.Queue()
    procs = [ctx.Process(target=_sample_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(checked >= 2 for _, checked in results), results


def test_shard_ranges_cover_all_rollouts():
    for n, world in ((8192, 1), (8192, 8), (65536, 4)):
        seen = np.zeros(n, dtype=int)
        for r in range(world):
            seen[r * (n // world):(r + 1) * (n // world)] += 1
        assert (seen == 1).all()


# ---- the peer exchange (update_kernels.h: PeerExchange / exchange_step), restated ------------------------------
NOT_ARRIVED = np.uint64(0xffffffffffffffff)  # update_kernels.h: kNotArrived


def words_of(values):
    """Four doubles -> four 8-byte words, the doubles themselves: a word is its own flag (anything but all ones has
    arrived; a number that IS that pattern is sent with its lowest bit flipped)."""
    bits = np.asarray(values, dtype=np.float64).view(np.uint64).copy()
    bits[bits == NOT_ARRIVED] ^= np.uint64(1)
    return bits


def doubles_of(words):
    assert (words != NOT_ARRIVED).all(), "a word that has not arrived"
    return words.view(np.float64)


def _peer_worker(rank, world, port, q):
    """Two ranks, every step of every iteration exchanged separately through per-rank inboxes [2 sets][world][T][4]
    (gloo send / recv stand in for the stores into the peer's memory): the reader clears the slots it has read (the set
    is written again two exchanges later), the numbers are combined with k_apply's expressions.  Must equal the all-gather of
    whole packets + apply_packets, bit for bit, over several iterations (both sets in use)."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from helpers import golden, iterations, params_from_golden
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        checked = 0
        g = golden("det_cvar")
        P = params_from_golden(g)
        its = iterations(g)
        t_steps = its[0]["u_in"].shape[0]
        inbox = np.full((2, world, t_steps, 4), NOT_ARRIVED, dtype=np.uint64)
        exchange = 0
        for rep in range(3):
            for it in its:
                n = it["costs"].shape[0]
                lo, hi = rank * n // world, (rank + 1) * n // world
                packet = local_packet(it["costs"][lo:hi], it["noise"][lo:hi], P["lambda_weight"])
                s = exchange & 1
                exchange += 1
                u = np.empty_like(it["u_in"], dtype=np.float32)
                for t in range(t_steps):
                    mine = words_of([packet[0], packet[1], packet[2 + 2 * t], packet[3 + 2 * t]])
                    inbox[s, rank, t] = mine
                    for peer in range(world):
                        if peer == rank:
                            continue
                        got = torch.zeros(4, dtype=torch.int64)
                        reqs = [dist.isend(torch.from_numpy(mine.view(np.int64).copy()), peer), dist.irecv(got, peer)]
                        for r in reqs:
                            r.wait()
                        assert (inbox[s, peer, t] == NOT_ARRIVED).all(), "slot not cleared"
                        inbox[s, peer, t] = got.numpy().view(np.uint64)
                    rows = np.stack([doubles_of(inbox[s, g_, t]) for g_ in range(world)])  # beta, den, nx, ny per rank
                    step_packets = np.concatenate([rows[:, :2], rows[:, 2:]], axis=1)
                    u[t] = apply_packets(step_packets, P["lambda_weight"], it["u_in"][t:t + 1], P["vrange"], P["wrange"])[0]
                    inbox[s, :, t, :] = NOT_ARRIVED  # (read: cleared for the exchange after next, update_kernels.h exchange_step)
                gathered = [torch.zeros(packet.size, dtype=torch.float64) for _ in range(world)]
                dist.all_gather(gathered, torch.from_numpy(packet))
                want = apply_packets(np.stack([x.numpy() for x in gathered]), P["lambda_weight"], it["u_in"], P["vrange"], P["wrange"])
                np.testing.assert_array_equal(u, want)
                checked += 1
        q.put((rank, checked))
    finally:
        dist.destroy_process_group()


def test_peer_exchange_words_and_inbox_sets_over_gloo_world2():
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    q = ctx.Queue()
    procs = [ctx.Process(target=_peer_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(checked >= 3 for _, checked in results), results


def test_words_round_trip_every_bit_pattern_class():
    vals = np.array([0.0, -0.0, 1.0, -1.5e-300, 3.141592653589793e200, np.float64(np.float32(13000.25)), 5e-324, -7.25])
    for i in range(0, len(vals), 4):
        four = vals[i:i + 4]
        np.testing.assert_array_equal(doubles_of(words_of(four)).view(np.uint64), four.view(np.uint64))
