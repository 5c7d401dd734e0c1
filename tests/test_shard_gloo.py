"""world_size-2 rehearsal of the row-sharded solve on CPU (gloo).

What runs on the GPUs is HIP + RCCL; what is checked here is the host logic
that makes it correct by construction: the halo plan (which rows travel to
whom, derived independently on every rank from the symmetric pattern) and the
exchange / all-reduce choreography of the distributed CG.  The arithmetic of
the emulated ranks is done with the oracle's matrix (the checker), never with
product code."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, quadratic, result_dir):
    for p in (os.path.join(ROOT, "fea-large_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    import scipy.sparse as sp
    import feahip
    import mesh
    from oracle_binding import OracleSolver

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    deck = mesh.bar_deck(dims=(3, 40, 3) if not quadratic else (2, 12, 2), quadratic=quadratic, recipe="clamped")
    plan = feahip.shard_plan(deck, rank, world)
    r0, r1 = plan["row0"], plan["row1"]
    N = len(deck.nodes)

    # 1. ownership is a partition of the rows
    rows = [None] * world
    dist.all_gather_object(rows, (r0, r1))
    assert rows[0][0] == 0 and rows[-1][1] == N and all(rows[k][1] == rows[k + 1][0] for k in range(world - 1))
    assert r1 > r0

    # 2. what I send to a peer is exactly what it expects to receive from me (same ascending order)
    for k, peer in enumerate(plan["peers"]):
        mine = torch.from_numpy(plan["send"][k].astype(np.int64))
        n_theirs = torch.zeros(1, dtype=torch.int64)
        n_mine = torch.tensor([len(mine)], dtype=torch.int64)
        if rank < peer:
            dist.send(n_mine, peer); dist.recv(n_theirs, peer)
        else:
            dist.recv(n_theirs, peer); dist.send(n_mine, peer)
        assert int(n_theirs) == len(plan["recv"][k])
        theirs = torch.zeros(int(n_theirs), dtype=torch.int64)
        if rank < peer:
            dist.send(mine, peer); dist.recv(theirs, peer)
        else:
            dist.recv(theirs, peer); dist.send(mine, peer)
        assert np.array_equal(theirs.numpy(), plan["recv"][k])
        assert np.all((plan["send"][k] >= r0) & (plan["send"][k] < r1))
        assert np.all((plan["recv"][k] < r0) | (plan["recv"][k] >= r1))

    # the checker's matrix (pre-BC stiffness of a deformed state), identical on both ranks
    o = OracleSolver(deck)
    o.set_nodes(mesh.deformed_state(deck.nodes, k1=1.05))
    o.update_state(); o.create_stiffness(); o.create_residual_forces(); o.apply_prescribed_bc(0.0)
    K = sp.csr_matrix((o.values().copy(), o.indexes().copy(), o.offsets().copy()), shape=(3 * N, 3 * N))
    Kown = K[3 * r0:3 * r1]
    dof = lambda nodes: (3 * np.asarray(nodes)[:, None] + np.arange(3)).ravel()

    def exchange(v):
        """halo rows of v from their owners (the choreography of Transport::exchange)"""
        for k, peer in enumerate(plan["peers"]):
            out = torch.from_numpy(v[dof(plan["send"][k])].copy())
            inc = torch.zeros(3 * len(plan["recv"][k]), dtype=torch.float64)
            if rank < peer:
                dist.send(out, peer); dist.recv(inc, peer)
            else:
                dist.recv(inc, peer); dist.send(out, peer)
            v[dof(plan["recv"][k])] = inc.numpy()

    def allsum(x):
        t = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(t)
        return float(t)

    # 3. sharded SpMV: only owned entries and exchanged halo rows are ever read
    rng = np.random.default_rng(11)
    pfull = rng.normal(size=3 * N)
    p = np.full(3 * N, np.nan)
    p[3 * r0:3 * r1] = pfull[3 * r0:3 * r1]
    exchange(p)
    used = np.unique(Kown.indices)
    assert not np.isnan(p[used]).any()
    y = Kown @ np.nan_to_num(p)
    assert np.abs(y - (K @ pfull)[3 * r0:3 * r1]).max() < 1e-12 * np.abs(K @ pfull).max()

    # 4. the distributed CG of kernels_solve.hip (x0 = b, Jacobi) equals the single-rank one
    b = o.forces().copy()
    own = slice(3 * r0, 3 * r1)
    dinv = 1.0 / K.diagonal()
    x = np.zeros(3 * N); x[own] = b[own]
    exchange(x)
    r = np.zeros(3 * N); r[own] = b[own] - Kown @ x
    pv = np.zeros(3 * N); pv[own] = dinv[own] * r[own]
    rz = allsum(r[own] @ pv[own])
    for _ in range(60):
        exchange(pv)
        q = Kown @ pv
        alpha = rz / allsum(pv[own] @ q)
        x[own] += alpha * pv[own]
        r[own] -= alpha * q
        rz_new = allsum(r[own] @ (dinv[own] * r[own]))
        pv[own] = dinv[own] * r[own] + (rz_new / rz) * pv[own]
        rz = rz_new
    # reference: the same recurrences in one piece
    xs = b.copy(); rs = b - K @ xs; ps = dinv * rs; rzs = rs @ ps
    for _ in range(60):
        qs = K @ ps
        a = rzs / (ps @ qs)
        xs += a * ps; rs -= a * qs
        rzn = rs @ (dinv * rs)
        ps = dinv * rs + (rzn / rzs) * ps
        rzs = rzn
    assert np.abs(x[own] - xs[own]).max() < 1e-9 * np.abs(xs).max()
    dist.barrier()
    open(os.path.join(result_dir, f"ok{rank}"), "w").write("ok")
    dist.destroy_process_group()


@pytest.mark.parametrize("quadratic", [False, True])
def test_two_rank_halo_plan_and_cg(tmp_path, quadratic):
    world = 2
    # the port is free when it is chosen, not necessarily when rank 0 binds it: a failed RENDEZVOUS gets one more port
    for attempt in range(2):
        try:
            mp.spawn(_worker, args=(world, _free_port(), quadratic, str(tmp_path)), nprocs=world, join=True)
            break
        except Exception as exc:
            rendezvous = any(w in str(exc) for w in ("EADDRINUSE", "address already in use", "Address already in use",
                                                     "DistNetworkError", "DistStoreError", "Connection reset"))
            if attempt == 1 or not rendezvous:
                raise
    assert all(os.path.exists(tmp_path / f"ok{r}") for r in range(world))


def test_plan_covers_every_referenced_column():
    sys.path.insert(0, os.path.join(ROOT, "fea-large_amd"))
    import feahip
    import mesh
    deck = mesh.bar_deck(dims=(2, 60, 2))
    for world in (3, 4):
        plans = [feahip.shard_plan(deck, r, world) for r in range(world)]
        assert plans[0]["row0"] == 0 and plans[-1]["row1"] == len(deck.nodes)
        for r, pl in enumerate(plans):
            for k, peer in enumerate(pl["peers"]):
                kk = plans[peer]["peers"].index(r)
                assert np.array_equal(pl["send"][k], plans[peer]["recv"][kk])
            # slabs across the long axis: only the neighbouring slabs are peers
            assert set(pl["peers"]) <= {r - 1, r + 1}


def test_bench_starts_its_own_ranks_and_reports_their_failure():
    """`python3 bench.py --gpus 2` without torch.distributed.run around it becomes a launcher BEFORE it touches torch or the
    library: two child ranks with RANK / WORLD_SIZE / MASTER_* set, rank 0's output passed through, the first non-zero
    return code propagated.  On a host without a GPU every rank leaves with code 2 ("needs an MI355X") -- which is what
    this checks: the launcher ran, both ranks ran as ranks 0 and 1 of 2, and the failure came back."""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("on a GPU box the ranks would run the benchmark itself (profiles/r04_b_bench_gloo2_selflaunch.*)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--cpu-sample", "0"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 2
    assert r.stderr.count("bench.py needs an MI355X") == 2
    assert "rank 0: 2, rank 1: 2" in r.stderr
