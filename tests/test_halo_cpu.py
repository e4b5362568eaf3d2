"""The N > 1 path on CPU: world_size 2 and 3 over gloo.  Every rank runs the real halo.py
(SlabSimulation + HaloExchange) with the oracle-backed stand-in engine; rank 0 reassembles the
global array and compares it BIT FOR BIT with the single-domain oracle.  Also an in-process
SlabGroup run (no torch.distributed) and the plain exchange layer."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest

from conftest import PKG_NAME, ROOT, small_scene, to_oracle_params


def _scene(pkg, oracle):
    rec, sp = small_scene(pkg, n=2400, grid=14, seed=41)
    op = to_oracle_params(oracle, sp)
    P = oracle.substep(rec, op, steps=2)
    rng = np.random.default_rng(3)
    P["vel"][:, 2] += rng.normal(0, 60, len(P)).astype(np.float32)      # plenty of z-migration
    P["vel"][:, 0] += rng.normal(0, 10, len(P)).astype(np.float32)
    P["isGhost"][300:340] = 1                 # ghosts of every kind: active (the substep zeroes vel / acc), inactive, and "3" (an ordinary particle)
    P["isActive"][300:320] = 1
    P["isGhost"][340:360] = 3
    P["vel"][300:360, 3] = 7.0
    return P, sp, op


def _cell_z(oracle, op, P):
    g = oracle.grid_extents(op)
    q = ((P["pos"][:, 2] - np.float32(g.gridMin[2])) / np.float32(g.cellSize)).astype(np.float32)
    return np.clip(np.floor(q), 0, g.dims[2] - 1).astype(np.int64), tuple(g.dims)


def _reference(oracle, op, P, steps):
    want = P
    for s in range(steps):
        if s % 3 == 0:
            want = oracle.wave_impulse(want, 1.5, 3.0, 0.1 * s, (0.2, 1.0, 0.4), -2.0, 2.0)
        want = oracle.substep(want, op)
    return want


def _worker(rank, world, port, steps, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from oracle import oracle
    from slab_standin import OracleSlabEngine
    pkg = importlib.import_module(PKG_NAME)
    halo = importlib.import_module(PKG_NAME + ".halo")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        P, sp, op = _scene(pkg, oracle)
        cz, dims = _cell_z(oracle, op, P)
        z0, z1 = halo.slab_range(dims[2], rank, world)
        m = (cz >= z0) & (cz < z1)
        ids = np.arange(len(P), dtype=np.uint32)
        eng = OracleSlabEngine(oracle, op, P[m], ids[m], z0, z1, rank > 0, rank < world - 1)
        ex = halo.HaloExchange(rank, world, transport="direct", device="cpu")
        sim = halo.SlabSimulation(eng, ex, rank, world, (z0, z1), dims, 4096,
                                  lambda n: torch.zeros((n, halo.REC_WORDS), dtype=torch.float32))
        moved = 0
        for s in range(steps):
            if s % 3 == 0:
                sim.ApplyWaveImpulse(1.5, 3.0, 0.1 * s, (0.2, 1.0, 0.4), -2.0, 2.0)
            sim.DispatchCompute()
            moved += sim.last_counts[0] + sim.last_counts[1]
        owned = sim.download_owned()
        gathered = [None] * world if rank == 0 else None
        dist.gather_object((owned, moved), gathered, dst=0)
        if rank == 0:
            allp = np.concatenate([g[0] for g in gathered])
            got = halo.merge_into_records(P, allp)
            want = _reference(oracle, op, P, steps)
            ok = got.tobytes() == want.tobytes()
            np.save(out_path, np.array([int(ok), len(allp), len(P), sum(g[1] for g in gathered)]))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 3])
def test_gloo_slab_decomposition_bit_exact(tmp_path, world):
    import torch.multiprocessing as mp
    out = str(tmp_path / "res.npy")
    mp.spawn(_worker, args=(world, _free_port(), 7, out), nprocs=world, join=True)
    ok, n_got, n_all, moved = np.load(out)
    assert n_got == n_all, "every particle is owned by exactly one rank"
    assert moved > 0
    assert ok == 1, "decomposed run differs from the single-domain oracle"


@pytest.mark.parametrize("world", [1, 2, 4, 7])
def test_inprocess_slab_group_matches_oracle(pkg, oracle, world):
    """Same protocol without torch.distributed (SlabGroup hands buffers over directly)."""
    import torch
    from slab_standin import OracleSlabEngine
    halo = importlib.import_module(PKG_NAME + ".halo")
    P, sp, op = _scene(pkg, oracle)
    cz, dims = _cell_z(oracle, op, P)
    ids = np.arange(len(P), dtype=np.uint32)
    grp = halo.SlabGroup.from_particles(
        P, ids, op, dims, world,
        lambda p, i, prm, z0, z1, lo, hi: OracleSlabEngine(oracle, prm, p, i, z0, z1, lo, hi),
        lambda n: torch.zeros((n, halo.REC_WORDS), dtype=torch.float32), 4096, cz)
    steps = 6
    for s in range(steps):
        if s % 3 == 0:
            grp.ApplyWaveImpulse(1.5, 3.0, 0.1 * s, (0.2, 1.0, 0.4), -2.0, 2.0)
        grp.DispatchCompute()
    got = halo.merge_into_records(P, grp.download())
    assert got.tobytes() == _reference(oracle, op, P, steps).tobytes()
    if world > 1:
        assert sum(s.last_counts[0] + s.last_counts[1] for s in grp.sims) > 0


def test_slab_range_partition():
    halo = importlib.import_module(PKG_NAME + ".halo")
    for gz in (1, 7, 64, 128, 1024):
        for world in (1, 2, 3, 8):
            if world > 1 and 2 * world > gz:          # a z-slab needs at least 2 cell layers
                with pytest.raises(ValueError):
                    halo.slab_range(gz, 0, world)
                continue
            edges = [halo.slab_range(gz, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == gz
            assert all(a[1] == b[0] for a, b in zip(edges, edges[1:]))
            assert all(z1 - z0 >= (2 if world > 1 else 1) for z0, z1 in edges)
