"""z-slab decomposition on the GPU: HIP slab engines (pack / unpack / ghost handling kernels)
must reproduce the single-engine result BIT FOR BIT, for any number of slabs."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest

from conftest import PKG_NAME, ROOT, assert_records_equal, small_scene, to_oracle_params

pytestmark = pytest.mark.gpu


def _scene(pkg, oracle, n=6000, grid=20, seed=51):
    rec, sp = small_scene(pkg, n=n, grid=grid, seed=seed)
    op = to_oracle_params(oracle, sp)
    P = oracle.substep(rec, op, steps=2)
    rng = np.random.default_rng(5)
    P["vel"][:, 2] += rng.normal(0, 60, len(P)).astype(np.float32)
    P["vel"][:, 0] += rng.normal(0, 10, len(P)).astype(np.float32)
    return P, sp, op


def _cell_z(pkg, sp, P):
    g = pkg.compute_grid_extents(sp)
    q = ((P["pos"][:, 2] - np.float32(g.gridMin[2])) / np.float32(g.cellSize)).astype(np.float32)
    return np.clip(np.floor(q), 0, g.dims[2] - 1).astype(np.int64), tuple(g.dims)


def _group(pkg, halo, P, sp, world, neighbor=3):
    import torch
    cz, dims = _cell_z(pkg, sp, P)
    ids = np.arange(len(P), dtype=np.uint32)

    def make_engine(p, i, prm, z0, z1, lo, hi):
        e = halo.HipSlabEngine(p, i, prm, z0, z1, lo, hi, capacity=int(len(p) * 1.5) + 8192)
        e.set_option(pkg.SPH_OPT_NEIGHBOR_KERNEL, neighbor)
        return e

    return halo.SlabGroup.from_particles(P, ids, sp, dims, world, make_engine,
                                         lambda n: torch.zeros((n, halo.REC_WORDS), dtype=torch.float32, device="cuda"), 8192, cz)


@pytest.mark.parametrize("neighbor", [1, 2, 3])
@pytest.mark.parametrize("world", [1, 2, 3, 5])
def test_slabs_match_single_engine_and_oracle(pkg, oracle, world, neighbor):
    halo = importlib.import_module(PKG_NAME + ".halo")
    P, sp, op = _scene(pkg, oracle)
    grp = _group(pkg, halo, P, sp, world, neighbor)
    single = pkg.SPHFluidGPU.from_particles(P, sp)
    want = P
    steps = 8
    for s in range(steps):
        if s % 3 == 0:
            args = (1.5, 3.0, 0.1 * s, (0.2, 1.0, 0.4), -2.0, 2.0)
            grp.ApplyWaveImpulse(*args)
            single.ApplyWaveImpulse(*args)
            want = oracle.wave_impulse(want, *args)
        grp.DispatchCompute()
        single.DispatchCompute()
        want = oracle.substep(want, op)
    got = halo.merge_into_records(P, grp.download())
    assert_records_equal(got, single.download(), f"{world} slabs vs one engine")
    assert_records_equal(got, want, f"{world} slabs vs oracle")
    if world > 1:
        assert sum(s.last_counts[0] + s.last_counts[1] for s in grp.sims) > 0
    single.close()


@pytest.mark.parametrize("world", [2, 3])
def test_async_exchange_matches_single_engine(pkg, oracle, world):
    """The exchange without host round trips (device-side counts, engine-owned face buffers: the path
    sph_slab_exchange wraps around ncclSend/ncclRecv) gives the single-engine bits; status reports no overflow."""
    halo = importlib.import_module(PKG_NAME + ".halo")
    P, sp, op = _scene(pkg, oracle)
    grp = _group(pkg, halo, P, sp, world)
    grp.enable_async(8192)
    single = pkg.SPHFluidGPU.from_particles(P, sp)
    want = P
    for s in range(8):
        if s % 3 == 0:
            args = (1.5, 3.0, 0.1 * s, (0.2, 1.0, 0.4), -2.0, 2.0)
            grp.ApplyWaveImpulse(*args)
            single.ApplyWaveImpulse(*args)
            want = oracle.wave_impulse(want, *args)
        import torch
        torch.cuda.synchronize()                  # the engines of the group run on their own streams
        grp.DispatchCompute()
        single.DispatchCompute()
        want = oracle.substep(want, op)
    got = halo.merge_into_records(P, grp.download())
    assert_records_equal(got, single.download(), f"{world} slabs (async exchange) vs one engine")
    assert_records_equal(got, want, f"{world} slabs (async exchange) vs oracle")
    st = [s.engine.status() for s in grp.sims]
    assert all(x[4] == 0 for x in st) and sum(x[0] + x[1] for x in st) > 0
    single.close()


@pytest.mark.parametrize("world", [2, 3])
def test_boundary_first_steps_match_single_engine(pkg, oracle, world):
    """sph_slab_step_begin / sph_slab_step_finish_local: the SPH pass runs the slots next to the faces first, the exchange
    of the NEXT substep (pack -> device-to-device copy of the neighbours' send faces -> unpack) runs on each engine's second
    stream beside the interior of the pass.  Same bits as the single engine and the oracle, impulses between the steps
    included (they reach the halo copies that are already in place); no overflow, and records did cross the faces."""
    import torch
    halo = importlib.import_module(PKG_NAME + ".halo")
    P, sp, op = _scene(pkg, oracle)
    grp = _group(pkg, halo, P, sp, world)
    grp.enable_overlap(8192)
    single = pkg.SPHFluidGPU.from_particles(P, sp)
    want = P
    for s in range(12):
        if s % 3 == 0:
            args = (1.5, 3.0, 0.1 * s, (0.2, 1.0, 0.4), -2.0, 2.0)
            grp.ApplyWaveImpulse(*args)
            single.ApplyWaveImpulse(*args)
            want = oracle.wave_impulse(want, *args)
        grp.DispatchCompute()
        single.DispatchCompute()
        want = oracle.substep(want, op)
    torch.cuda.synchronize()
    got = halo.merge_into_records(P, grp.download())
    assert_records_equal(got, single.download(), f"{world} slabs (boundary-first steps) vs one engine")
    assert_records_equal(got, want, f"{world} slabs (boundary-first steps) vs oracle")
    st = [s.engine.status() for s in grp.sims]
    assert all(x[4] == 0 for x in st) and sum(x[0] + x[1] for x in st) > 0
    single.close()


def test_boundary_first_steps_with_a_container_change(pkg, oracle):
    """As test_container_change_under_the_reduced_face_scan, on the boundary-first schedule: after the container changed under
    the fluid the SPH pass is not split and the pack scans every slot, one substep long."""
    halo = importlib.import_module(PKG_NAME + ".halo")
    P, sp, op = _scene(pkg, oracle)
    grp = _group(pkg, halo, P, sp, 3)
    grp.enable_overlap(8192)
    single = pkg.SPHFluidGPU.from_particles(P, sp)
    want = P
    for s in range(9):
        if s == 4:
            sp.param_shapeType = 1                        # sphere of the same extent: same grid, corner particles jump inwards
            single.param_shapeType = 1                    # (the slab engines share `sp`; the single engine has its own copy)
            op.shapeType = 1
        grp.DispatchCompute()
        single.DispatchCompute()
        want = oracle.substep(want, op)
    got = halo.merge_into_records(P, grp.download())
    assert_records_equal(got, single.download(), "3 slabs (boundary-first steps, container change) vs one engine")
    assert_records_equal(got, want, "3 slabs (boundary-first steps, container change) vs oracle")
    single.close()


def test_container_change_under_the_reduced_face_scan(pkg, oracle):
    """k_slab_pack looks only at the two ends of the (z-major sorted) slot range while nothing can have moved a particle
    by more than one layer.  A container that changes under the fluid can: box -> sphere of the same extent (same grid)
    projects the corner particles many cells inwards, across slab boundaries.  The substep after such a change must
    scan every slot again; the result stays the single engine's, bit for bit."""
    import torch
    halo = importlib.import_module(PKG_NAME + ".halo")
    P, sp, op = _scene(pkg, oracle)
    grp = _group(pkg, halo, P, sp, 3)
    grp.enable_async(8192)
    single = pkg.SPHFluidGPU.from_particles(P, sp)
    want = P
    for s in range(9):
        if s == 4:
            sp.param_shapeType = 1                    # the slab engines share `sp`; the single engine has its own copy
            single.param_shapeType = 1
            op.shapeType = 1
        torch.cuda.synchronize()
        grp.DispatchCompute()
        single.DispatchCompute()
        want = oracle.substep(want, op)
    assert pkg.compute_grid_extents(sp).numCells == pkg.compute_grid_extents(single.params).numCells
    got = halo.merge_into_records(P, grp.download())
    moved = np.linalg.norm(got["pos"][:, :3] - P["pos"][:, :3], axis=1)
    assert moved.max() > 2.0 * sp.param_h             # some particles did jump several cells
    assert_records_equal(got, single.download(), "3 slabs vs one engine across a container change")
    assert_records_equal(got, want, "3 slabs vs oracle across a container change")
    st = [x.engine.status() for x in grp.sims]
    assert all(x[4] == 0 for x in st)
    single.close()


def test_async_exchange_reports_overflow(pkg, oracle):
    halo = importlib.import_module(PKG_NAME + ".halo")
    P, sp, op = _scene(pkg, oracle)
    grp = _group(pkg, halo, P, sp, 2)
    grp.enable_async(16)                          # far too small a face buffer
    import torch
    torch.cuda.synchronize()
    grp.DispatchCompute()
    with pytest.raises(pkg.SphError, match="overflow"):
        grp.sims[0].engine.status()


def test_rccl_comm_of_one_rank(pkg, oracle):
    """RCCL itself on the one GPU of this box: a communicator of world size 1 and sph_slab_exchange on a slab without
    neighbours (no send / recv is issued; two ranks cannot share a device).  The slab then equals the single engine."""
    halo = importlib.import_module(PKG_NAME + ".halo")
    P, sp, op = _scene(pkg, oracle, n=2400, grid=14, seed=53)
    cz, dims = _cell_z(pkg, sp, P)
    eng = halo.HipSlabEngine(P, np.arange(len(P), dtype=np.uint32), sp, 0, dims[2], False, False, capacity=len(P) + 4096)
    eng.alloc_faces(1024)
    comm = halo.RcclComm(0, 1, lambda b: b)
    for _ in range(3):
        eng.exchange(comm)
        eng.dispatch()
    got = halo.merge_into_records(P, eng.download_owned())
    assert_records_equal(got, oracle.substep(P, op, steps=3), "one-rank RCCL slab")
    comm.close()
    eng.close()


def test_slabs_config2_262k(pkg):
    """BASELINE.json configs[1] size (262 144 particles, 64^3): 4 slabs == 1 engine after 5 substeps."""
    halo = importlib.import_module(PKG_NAME + ".halo")
    syn = pkg.synthetic
    cfg = syn.CONFIGS[2]
    rec, _ = syn.make_particles(cfg)
    sp = pkg.default_params(**syn.params_fields(cfg))
    grp = _group(pkg, halo, rec, sp, 4)
    single = pkg.SPHFluidGPU.from_particles(rec, sp)
    for _ in range(5):
        grp.DispatchCompute()
        single.DispatchCompute()
    assert_records_equal(halo.merge_into_records(rec, grp.download()), single.download(), "4 slabs at 262k")
    # the slab generator of synthetic.py gives each rank exactly its particles
    owned0 = grp.sims[1].download_owned()
    z0, z1 = halo.slab_range(cfg.grid[2], 1, 4)
    part, gid = syn.make_particles(cfg, z_cells=(z0, z1))
    assert len(part) > 0 and set(gid.tolist()) <= set(range(cfg.n)) and len(owned0) > 0
    single.close()


def _gloo_worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from oracle import oracle
    pkg = importlib.import_module(PKG_NAME)
    halo = importlib.import_module(PKG_NAME + ".halo")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        P, sp, op = _scene(pkg, oracle, n=4000, grid=16, seed=52)
        cz, dims = _cell_z(pkg, sp, P)
        z0, z1 = halo.slab_range(dims[2], rank, world)
        m = (cz >= z0) & (cz < z1)
        ids = np.arange(len(P), dtype=np.uint32)
        eng = halo.HipSlabEngine(P[m], ids[m], sp, z0, z1, rank > 0, rank < world - 1, capacity=int(m.sum() * 1.5) + 8192)
        ex = halo.HaloExchange(rank, world, transport="host", device="cuda")
        sim = halo.SlabSimulation(eng, ex, rank, world, (z0, z1), dims, 8192,
                                  lambda n: torch.zeros((n, halo.REC_WORDS), dtype=torch.float32, device="cuda"))
        for _ in range(6):
            sim.DispatchCompute()
        owned = sim.download_owned()
        gathered = [None] * world if rank == 0 else None
        dist.gather_object(owned, gathered, dst=0)
        if rank == 0:
            got = halo.merge_into_records(P, np.concatenate(gathered))
            want = oracle.substep(P, op, steps=6)
            np.save(out_path, np.array([int(got.tobytes() == want.tobytes())]))
    finally:
        dist.destroy_process_group()


def test_two_processes_gloo_host_staging(tmp_path):
    """Two real ranks (two processes sharing the one GPU), torch.distributed gloo with host staging:
    the same SlabSimulation / HaloExchange code bench.py runs with nccl."""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "res.npy")
    mp.spawn(_gloo_worker, args=(2, port, out), nprocs=2, join=True)
    assert np.load(out)[0] == 1


def test_bench_two_ranks_rehearsal_over_gloo(tmp_path):
    """bench.py's N > 1 path end to end (what the driver launches with torch.distributed.run): two ranks on this one
    GPU, SPH_BENCH_BACKEND=gloo (host-staged halos; RCCL refuses two ranks on one device), BASELINE configs[1] size so
    that it takes seconds.  Checks the line's contract fields and that no rank dropped halo records."""
    import json
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, SPH_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "config2", "--steps", "6",
           "--warmup", "2", "--no-cpu-baseline"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                    # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["particles"] == 2 * 262144 and "REHEARSAL" in d["config"]["workload"]
    assert d["slab_status"]["overflow_on_any_rank"] is False
    lo_hi = d["slab_status"]["records_lo_hi_live_per_rank"]
    assert lo_hi[0][0] == 0 and lo_hi[0][1] > 0 and lo_hi[1][0] > 0 and lo_hi[1][1] == 0     # each rank has one neighbour


def test_bench_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with WORLD_SIZE unset (the form the driver uses at N = 1, extended to N > 1): bench.py
    starts torch.distributed.run itself as a child process before touching the GPU and relays the one JSON line."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(SPH_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "config2", "--steps", "4", "--warmup", "1", "--no-cpu-baseline"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["value"] > 0 and d["slab_status"]["overflow_on_any_rank"] is False


_SELFTEST_CHILD = r"""
import importlib, os, sys
sys.path.insert(0, sys.argv[1])
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
# as bench.py does at N > 1: torch's own RCCL process group first, then the engine's communicator beside it
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
t = torch.ones(4, device="cuda"); dist.all_reduce(t); torch.cuda.synchronize()
halo = importlib.import_module(sys.argv[2] + ".halo")
obj = [None]
def bcast(b):
    obj[0] = b
    dist.broadcast_object_list(obj, src=0)
    return obj[0]
comm = halo.RcclComm(0, 1, bcast)
for nbytes in (64, 1 << 20, (209716 + 1) * 64):          # the last: one face message of the weak-scaling slab (1.6 layers of 256 x 256 cells)
    comm.selftest(nbytes)
# the engine's own exchange pattern (round 5): the 64-byte plans with the handshake's polled wait, then the routine sph_slab_exchange posts its faces
# with -- per neighbour two send / receive pairs of UNEQUAL sizes in one group -- with the rank as both of its neighbours; every byte compared
for cap, counts in ((1024, (0, 0, 0, 0)), (1024, (1, 1024, 1024, 0)), (8192, (3000, 2900, 17, 411)), (217908, (163840, 171000, 5200, 900))):
    ms = comm.selftest_faces(cap, counts)
    print("faces", cap, counts, round(ms, 3), "ms")
comm.close()
dist.destroy_process_group()
print("SELFTEST OK")
"""


def test_rccl_send_recv_to_self_in_a_child_process(tmp_path):
    """ncclSend / ncclRecv through the engine's dlopen'd RCCL, executed on this one GPU: a communicator of one rank sends to
    itself (sph_comm_selftest: grouped send + recv of uint8 counts on a non-blocking stream, compared on the host), beside a
    torch.distributed RCCL process group as in bench.py's N > 1 path.  In a child process with a timeout, so that a transport
    that hangs fails this test instead of stopping the run."""
    import subprocess
    script = tmp_path / "selftest_child.py"
    script.write_text(_SELFTEST_CHILD)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, str(script), ROOT, PKG_NAME], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert res.returncode == 0 and "SELFTEST OK" in res.stdout, res.stdout[-2000:] + res.stderr[-3000:]


def test_bench_slab_path_with_rccl_and_one_rank():
    """`python bench.py --slab-path`: the code path the driver's N > 1 runs take -- torch.distributed with backend nccl (= RCCL),
    halo.SlabSimulation.from_config(transport="rccl") with the engine's own communicator, boundary-first substeps
    (sph_slab_step_begin / sph_slab_step_finish), WaveImpulse every 16th substep, the status all_gather -- with the ONE rank this
    box can give it (RCCL refuses two ranks on a device).  In a child process with a timeout."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "SPH_BENCH_BACKEND")}
    env.update(MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--slab-path", "--workload", "weak5", "--steps", "20", "--warmup", "3", "--no-cpu-baseline"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["value"] > 1e9 and "REHEARSAL" not in d["config"]["workload"]
    st = d["slab_status"]
    assert st["overflow_on_any_rank"] is False and st["multi_layer_move_on_any_rank"] is False
    assert st["records_lo_hi_live_per_rank"][0][:2] == [0, 0] and st["records_lo_hi_live_per_rank"][0][2] == 8388608


# ---- round 4: the exchange's protocol ----------------------------------------------------
def test_paused_steps_leave_the_halo_records_alone(pkg, oracle):
    """A paused DispatchCompute is a no-op (SPHFluid3D.cpp:432) and so is its exchange: 40 paused boundary-first steps on a 2-slab group
    neither fill the slots with stale halo copies nor change a bit; after the pause the run goes on bit for bit."""
    halo = importlib.import_module(PKG_NAME + ".halo")
    P, sp, op = _scene(pkg, oracle)
    grp = _group(pkg, halo, P, sp, 2)
    grp.enable_overlap(8192)
    want = P
    for _ in range(3):
        grp.DispatchCompute(); want = oracle.substep(want, op)
    slots_before = [s.engine.status()[2] for s in grp.sims]
    sp.param_pause = 1
    for _ in range(40):
        grp.DispatchCompute()
    assert [s.engine.status()[2] for s in grp.sims] == slots_before      # nothing was appended
    sp.param_pause = 0
    for _ in range(3):
        grp.DispatchCompute(); want = oracle.substep(want, op)
    st = [s.engine.status() for s in grp.sims]
    assert all(x[4] == 0 for x in st), st
    assert_records_equal(halo.merge_into_records(P, grp.download()), want, "2 slabs across 40 paused steps")


def test_step_finish_before_the_neighbour_has_begun_is_refused(pkg, oracle):
    halo = importlib.import_module(PKG_NAME + ".halo")
    P, sp, op = _scene(pkg, oracle)
    grp = _group(pkg, halo, P, sp, 2)
    grp.enable_overlap(8192)
    grp.DispatchCompute()                                        # primes and runs one step on both
    a, b = grp.sims[0].engine, grp.sims[1].engine
    a.step_begin()
    with pytest.raises(pkg.SphError, match="has not begun THIS step"):
        a.step_finish_local(None, b)                             # b still holds the events of the step before
    b.step_begin()
    a.step_finish_local(None, b); b.step_finish_local(a, None)
    want = oracle.substep(P, op, steps=2)
    assert_records_equal(halo.merge_into_records(P, grp.download()), want, "2 slabs after a refused finish")


def test_grid_change_between_two_steps_is_refused_by_the_abi_and_followed_by_the_driver(pkg, oracle):
    """The halo records in place were cut for one grid: sph_slab_step_begin refuses a grid that moved since; the Python
    driver primes again and stays bit-exact."""
    halo = importlib.import_module(PKG_NAME + ".halo")
    P, sp, op = _scene(pkg, oracle)
    grp = _group(pkg, halo, P, sp, 2)
    grp.enable_overlap(8192)
    want = P
    for _ in range(2):
        grp.DispatchCompute(); want = oracle.substep(want, op)
    sp.param_boxCenter[2] = float(sp.param_boxCenter[2]) + float(sp.param_h)       # the grid moves by one cell in z
    assert tuple(pkg.compute_grid_extents(sp).dims) == grp.sims[0].grid_dims
    with pytest.raises(pkg.SphError, match="grid changed"):
        grp.sims[0].engine.step_begin()
    op = to_oracle_params(oracle, sp)
    for _ in range(3):
        grp.DispatchCompute(); want = oracle.substep(want, op)
    st = [s.engine.status() for s in grp.sims]
    assert all(x[4] & ~16 == 0 for x in st), st
    if all(x[4] == 0 for x in st):                               # (the move itself may carry particles across more layers than the exchange follows)
        assert_records_equal(halo.merge_into_records(P, grp.download()), want, "2 slabs across a grid that moved by a cell")


def test_messages_shrink_to_the_records_in_use_and_halo_copies_are_40_bytes(pkg, oracle):
    """Once a face has been calm for two exchanges (its record counts within 3 %) the messages carry the records in use (+ 25 % + 1024)
    instead of whole faces, halo copies as 40-byte records; same bits.  (A face that is not calm keeps sending whole faces.)"""
    halo = importlib.import_module(PKG_NAME + ".halo")
    P, sp = small_scene(pkg, n=15000, grid=24, seed=51)          # the lattice scene at rest: calm faces
    op = to_oracle_params(oracle, sp)
    grp = _group(pkg, halo, P, sp, 3)
    cap = 20000
    grp.enable_overlap(cap)
    want = P
    for _ in range(8):
        grp.DispatchCompute(); want = oracle.substep(want, op)
    assert_records_equal(halo.merge_into_records(P, grp.download()), want, "3 slabs, count-sized messages")
    mid = grp.sims[1].engine
    sent_lo, sent_hi, face_lo, face_hi = mid.message_bytes()
    st = mid.status()
    assert st[4] == 0 and face_lo == face_hi == 64 + cap * (64 + 40)
    for sent, recs in ((sent_lo, st[0]), (sent_hi, st[1])):
        assert 64 + recs * 40 <= sent <= 64 + (recs * 1.3 + 2 * 1024 + 64) * 64 and sent < face_lo / 4, (sent, recs, face_lo)


def test_a_jump_across_a_whole_slab_is_a_notice_not_an_error(pkg, oracle):
    """Flag 16 (a particle crossed more layers than the exchange follows) loses no record: sph_slab_download delivers, the status call
    succeeds and carries the bit, sph_slab_clear_flags acknowledges it."""
    halo = importlib.import_module(PKG_NAME + ".halo")
    P, sp, op = _scene(pkg, oracle)
    cz, dims = _cell_z(pkg, sp, P)
    g = pkg.compute_grid_extents(sp)
    i = int(np.flatnonzero(cz == dims[2] // 3 - 1)[0])           # a particle in the top layer of the lowest of 3 slabs ...
    P = P.copy()
    P["vel"][i] = (0.0, 0.0, (dims[2] // 3 + 2) * float(g.cellSize) / float(sp.param_timeStep), 0.0)   # ... thrown across the whole middle slab
    grp = _group(pkg, halo, P, sp, 3)
    grp.enable_overlap(8192)
    for _ in range(3):
        grp.DispatchCompute()
    st = [s.engine.status() for s in grp.sims]                   # does not raise
    assert any(x[4] & 16 for x in st) and all(x[4] & ~16 == 0 for x in st), st
    owned = grp.download()                                       # does not raise, nobody is lost
    assert sorted(owned["id"].tolist()) == list(range(len(P)))
    for s in grp.sims:
        s.engine.clear_flags(16)
    assert all(s.engine.status()[4] == 0 for s in grp.sims)


def test_step_times_say_whether_the_exchange_was_hidden(pkg, oracle):
    halo = importlib.import_module(PKG_NAME + ".halo")
    P, sp, op = _scene(pkg, oracle, n=15000, grid=24)
    grp = _group(pkg, halo, P, sp, 2)
    grp.enable_overlap(20000)
    for s in grp.sims:
        s.engine.set_option(pkg.SPH_OPT_TIMING, 1)
    for _ in range(4):
        grp.DispatchCompute()
    for s in grp.sims:
        pack, transfer, unpack, exchange_end, pass_end = s.engine.step_times()
        assert pack > 0 and transfer >= 0 and unpack > 0 and exchange_end > 0 and pass_end > 0


def test_a_message_that_turns_out_too_small_is_reported(pkg, oracle):
    """Messages are sized from the counts of two exchanges ago + 25 % + 1024 records.  With the margin and the calm / hold rules taken away
    (sph_slab_debug_tight_messages on EVERY engine: the hook is part of the plan, so both ends of a link still agree) any growth of a face's
    record count cuts records off -- which must be LOUD (error flag 8: sph_slab_status / sph_slab_download fail), never silent."""
    halo = importlib.import_module(PKG_NAME + ".halo")
    rec, sp = small_scene(pkg, n=6000, grid=20, seed=51)
    grp = _group(pkg, halo, rec, sp, 2)
    grp.enable_overlap(8192)
    for s in grp.sims:
        s.engine.debug_tight_messages(True)
    for step in range(8):
        if step == 4:
            grp.ApplyWaveImpulse(40.0, 50.0, 1.0, (0.0, 0.0, 1.0), -1e9, 1e9)      # a kick along z: the faces' record counts change
        grp.DispatchCompute()
    with pytest.raises(pkg.SphError, match="more halo records than its message"):
        for x in grp.sims:
            x.engine.status()


def _primed_calm_group(pkg, oracle, halo, world=3, steps=6):
    """A group whose faces are calm: its exchanges are SIZED (not whole faces), so the two ends of a link really depend on each other's state."""
    P, sp = small_scene(pkg, n=15000, grid=24, seed=51)
    op = to_oracle_params(oracle, sp)
    grp = _group(pkg, halo, P, sp, world)
    grp.enable_overlap(20000)
    want = P
    for _ in range(steps):
        grp.DispatchCompute(); want = oracle.substep(want, op)
    plans = [s.engine.plan()[0] for s in grp.sims]
    assert all(p.sendHalo[1] < 20000 for p in plans[:-1]) and all(p.flags == 0 for p in plans), [(list(p.sendHalo), p.flags) for p in plans]
    return P, sp, op, grp, want


def test_the_plans_of_neighbouring_engines_agree_on_every_step(pkg, oracle):
    """What sph_slab_step_finish_local asserts before it copies anything: the neighbour's send sizes, computed from the
    NEIGHBOUR's state, equal this engine's receive sizes, computed from the headers it received -- on whole-face exchanges and on sized ones."""
    halo = importlib.import_module(PKG_NAME + ".halo")
    P, sp, op, grp, want = _primed_calm_group(pkg, oracle, halo)
    L = pkg.load_library()
    import ctypes as C
    for _ in range(3):
        grp.DispatchCompute(); want = oracle.substep(want, op)
        plans = [s.engine.plan()[0] for s in grp.sims]
        for r in range(len(plans) - 1):
            why = C.create_string_buffer(320)
            assert L.sph_slab_plans_agree(C.byref(plans[r]), C.byref(plans[r + 1]), 1, why, 320) == 1, why.value
            assert L.sph_slab_plans_agree(C.byref(plans[r + 1]), C.byref(plans[r]), 0, why, 320) == 1, why.value
            assert plans[r].sendHalo[1] == plans[r + 1].recvHalo[0] and plans[r].recvMig[1] == plans[r + 1].sendMig[0]
            assert 0 < plans[r].sendHalo[1] < 20000                                  # sized, not whole faces
    assert_records_equal(halo.merge_into_records(P, grp.download()), want, "3 slabs, sized exchanges, plans compared on every step")


@pytest.mark.parametrize("what", ["kick", "member", "hook"])
def test_a_call_on_one_engine_only_is_refused_by_name_before_any_record_moves(pkg, oracle, what):
    """An impulse, a member edit or a test hook issued on ONE engine of a group makes that engine plan other message sizes (or other physics) than
    its neighbours expect.  Over RCCL that is a hang or a truncated receive; here -- and in the RCCL handshake, which runs the same comparison on
    the plans it receives -- it is SPH_ERR_STATE with the difference by name, raised by the finish of the very next step, on both ends of the link."""
    halo = importlib.import_module(PKG_NAME + ".halo")
    P, sp, op, grp, want = _primed_calm_group(pkg, oracle, halo)
    a, b, c = (s.engine for s in grp.sims)
    if what == "kick":
        b.apply_wave_impulse(40.0, 50.0, 1.0, (0.0, 0.0, 1.0), -1e9, 1e9)          # the middle engine only
        word = "one rank only"
    elif what == "member":
        q = type(sp).from_buffer_copy(sp)
        q.param_viscosity = 4.0
        b._p = q                                                                     # (step_begin hands the engine its own members)
        word = "members"
    else:
        b.debug_tight_messages(True)
        word = "hook"
    for e in (a, b, c):
        e.step_begin()
    with pytest.raises(pkg.SphError, match=word):
        a.step_finish_local(None, b)                                                 # the lower end of the link refuses ...
    with pytest.raises(pkg.SphError, match=word):
        b.step_finish_local(a, c)                                                    # ... and so does the engine that was kicked
    with pytest.raises(pkg.SphError, match=word):
        c.step_finish_local(b, None)
    for e in (a, b, c):
        assert e.status()[4] == 0                                                    # nothing was cut off, nothing overflowed: nothing moved


def test_a_sized_exchange_after_the_box_moved_holds_whole_faces(pkg, oracle):
    """sph_slab_exchange sized its messages before anything had noticed a grid / container edit.  The plan notices it now: the
    exchange right after the box moved by a cell is planned with whole faces (flags bit 2), and the event is counted."""
    halo = importlib.import_module(PKG_NAME + ".halo")
    P, sp, op = _scene(pkg, oracle, n=2400, grid=14, seed=53)
    cz, dims = _cell_z(pkg, sp, P)
    eng = halo.HipSlabEngine(P, np.arange(len(P), dtype=np.uint32), sp, 0, dims[2], False, False, capacity=len(P) + 4096)
    eng.alloc_faces(1024)
    comm = halo.RcclComm(0, 1, lambda b: b)
    for _ in range(6):
        eng.exchange(comm); eng.dispatch()
    p0, _ = eng.plan()
    assert p0.flags == 0 and p0.exchangeNo == 5
    sp.param_boxCenter[0] = float(sp.param_boxCenter[0]) + float(sp.param_h)        # the grid moves by one cell
    eng.exchange(comm)
    p1, _ = eng.plan()
    assert p1.flags & 4 and p1.holdEvents > p0.holdEvents and p1.exchangeNo == 6
    eng.dispatch()
    for _ in range(4):
        eng.exchange(comm); eng.dispatch()
    p2, _ = eng.plan()
    assert p2.flags == 0                                                             # calm again
    comm.close(); eng.close()


def test_sync_with_a_deadline(pkg, oracle):
    halo = importlib.import_module(PKG_NAME + ".halo")
    P, sp, op, grp, want = _primed_calm_group(pkg, oracle, halo, world=2, steps=4)
    for s in grp.sims:
        s.engine.sync(deadline=30.0)
    assert_records_equal(halo.merge_into_records(P, grp.download()), want, "2 slabs")
