"""The engine's RCCL code path between REAL RANKS on one GPU.

RCCL refuses two ranks on one device, so on a one-GPU box `sph_slab_exchange` / `sph_slab_step_finish` had only ever run with a communicator of one rank
(no neighbour, nothing sent).  Here two and three processes each own a z-slab of one scene and drive the engine exactly as `bench.py --gpus N` does --
priming exchange, then boundary-first steps with `sph_slab_step_finish(engine, comm)` -- while the `nccl*` entry points the engine dlopens come from
tests/fake_rccl/fake_rccl.c (environment variable SPH_RCCL_LIBRARY): a stand-in that moves the messages through shared memory, keeps NCCL's matching rule
(sends / receives to one peer match in issue order; a group is issued at ncclGroupEnd) and is STRICTER than RCCL where it helps: a receive whose size differs
from its send's fails with both sizes (RCCL: a hang or a cut-off message), a receive nobody sends to fails after a few seconds.

What this covers that nothing else can on this box: the 64-byte plans really cross a link and are compared by the other process; the sizes of the four face
messages per link really come out equal on both ends, exchange after exchange, whole faces and sized; a call issued on ONE rank is refused by BOTH with the
difference by name; a rank that stops calling is an error on its neighbour, not a hang.  What it does not cover: xGMI, asynchrony, speed."""
import importlib
import os
import subprocess
import sys
import time

import numpy as np
import pytest

from conftest import PKG_NAME, ROOT, assert_records_equal, small_scene, to_oracle_params

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
FAKE_SRC = os.path.join(HERE, "fake_rccl", "fake_rccl.c")
FAKE_LIB = os.path.join(HERE, "fake_rccl", "libfake_rccl.so")

_CHILD = r'''
import importlib, os, sys, time
import numpy as np
root, pkgname, rank, world, idfile, scenario, steps, outfile = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], sys.argv[6], int(sys.argv[7]), sys.argv[8]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from conftest import small_scene
pkg = importlib.import_module(pkgname)
halo = importlib.import_module(pkgname + ".halo")
rec, sp = small_scene(pkg, n=15000, grid=24, seed=51)
g = pkg.compute_grid_extents(sp)
dims = tuple(int(v) for v in g.dims)
cz = np.clip(np.floor(((rec["pos"][:, 2] - np.float32(g.gridMin[2])) / np.float32(g.cellSize)).astype(np.float32)), 0, dims[2] - 1).astype(np.int64)
z0, z1 = halo.slab_range(dims[2], rank, world)
m = (cz >= z0) & (cz < z1)
ids = np.arange(len(rec), dtype=np.uint32)
eng = halo.HipSlabEngine(rec[m], ids[m], sp, z0, z1, rank > 0, rank < world - 1, capacity=int(m.sum() * 1.5) + 16384)
eng.alloc_faces(8192)
eng.set_deadline(15.0)

def bcast(data):
    if rank == 0:
        with open(idfile + ".tmp", "wb") as fh: fh.write(data)
        os.replace(idfile + ".tmp", idfile)
        return data
    t0 = time.time()
    while not os.path.exists(idfile):
        if time.time() - t0 > 30: raise RuntimeError("no id from rank 0")
        time.sleep(0.01)
    return open(idfile, "rb").read()

comm = halo.RcclComm(rank, world, bcast)
plans = []
try:
    eng.exchange(comm)                                       # the halo records of the first substep (sph_slab_exchange)
    for s in range(steps):
        if s % 4 == 1:                                       # a gentle wave on every rank (no hold) ...
            eng.apply_wave_impulse(0.8, 3.0, 0.3 * s, (0.0, 1.0, 0.0), -1e9, 1e9)
        if s == 6:                                           # ... and one strong kick on every rank (whole faces for three exchanges, then sized again)
            eng.apply_wave_impulse(30.0, 4.0, 1.0, (0.0, 0.2, 1.0), -1e9, 1e9)
        if scenario == "one_sided" and s == 9 and rank == world - 1:
            eng.apply_wave_impulse(30.0, 4.0, 2.0, (0.0, 0.2, 1.0), -1e9, 1e9)      # THIS rank only
        if scenario == "gone" and s == 5 and rank == world - 1:
            print("LEAVING", flush=True)
            os._exit(0)                                      # this rank stops making calls (device work of the others still queued behind it)
        eng.step_begin()
        eng.step_finish(comm)
        p, hs = eng.plan()
        plans.append([int(p.exchangeNo), int(p.holdEvents), int(p.flags)] + [int(x) for x in list(p.sendHalo) + list(p.sendMig) + list(p.recvHalo) + list(p.recvMig)])
    eng.sync(deadline=20.0)
    st = eng.status()
    owned = eng.download_owned()
    np.savez(outfile, owned=owned, status=np.array(st), plans=np.array(plans), sent=np.array(eng.message_bytes()))
    print("DONE", st, flush=True)
except pkg.SphError as ex:
    print("SPHERROR", str(ex), flush=True)
    os._exit(7)                                              # (never a normal interpreter exit with device work queued behind a dead link)
'''


@pytest.fixture(scope="module")
def fake_lib():
    if not os.path.exists(FAKE_LIB) or os.path.getmtime(FAKE_LIB) < os.path.getmtime(FAKE_SRC):
        subprocess.run(["gcc", "-O2", "-Wall", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-o", FAKE_LIB, FAKE_SRC,
                        "-L/opt/rocm/lib", "-lamdhip64", "-lrt"], check=True)
    return FAKE_LIB


def _run(tmp_path, fake_lib, world, scenario, steps):
    script = tmp_path / "rank.py"
    script.write_text(_CHILD)
    idfile = str(tmp_path / f"id_{scenario}_{world}")
    env = dict(os.environ, SPH_RCCL_LIBRARY=fake_lib, FAKE_RCCL_TIMEOUT_S="6", HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = []
    for r in range(world):
        out = str(tmp_path / f"out_{scenario}_{world}_{r}.npz")
        procs.append((out, subprocess.Popen([sys.executable, str(script), ROOT, PKG_NAME, str(r), str(world), idfile, scenario, str(steps), out],
                                            stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd=ROOT)))
    res = []
    for out, p in procs:
        try:
            so, se = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for _, q in procs:
                q.kill()                                     # exactly the processes this test started
            raise AssertionError(f"a rank of the {scenario} run with {world} ranks HUNG (what the engine must never do)")
        res.append((p.returncode, so, se, out))
    try:                                                     # a rank that left without destroying its communicator leaves the segment behind
        name = open(idfile, "rb").read().split(b"\0")[0].decode()
        if name.startswith("/sph_fake_rccl_") and os.path.exists("/dev/shm" + name):
            os.unlink("/dev/shm" + name)
    except OSError:
        pass
    return res


@pytest.mark.parametrize("world", [2, 3])
def test_ranks_in_separate_processes_match_the_oracle(pkg, oracle, tmp_path, fake_lib, world):
    """The whole N > 1 path of the engine between real processes: bit for bit the oracle's single-domain result, the plans of both ends of every link equal
    in every exchange (the transport itself refuses a receive whose size differs from its send), sized messages after the faces have been calm."""
    halo = importlib.import_module(PKG_NAME + ".halo")
    steps = 16
    res = _run(tmp_path, fake_lib, world, "plain", steps)
    for rc, so, se, _ in res:
        assert rc == 0 and "DONE" in so, so[-1500:] + se[-3000:]
    rec, sp = small_scene(pkg, n=15000, grid=24, seed=51)
    op = to_oracle_params(oracle, sp)
    want = rec
    for s in range(steps):
        if s % 4 == 1:
            want = oracle.wave_impulse(want, 0.8, 3.0, 0.3 * s, (0.0, 1.0, 0.0), -1e9, 1e9)
        if s == 6:
            want = oracle.wave_impulse(want, 30.0, 4.0, 1.0, (0.0, 0.2, 1.0), -1e9, 1e9)
        want = oracle.substep(want, op)
    outs = [np.load(out) for _, _, _, out in res]
    owned = np.concatenate([o["owned"] for o in outs])
    assert all(int(o["status"][4]) == 0 for o in outs), [o["status"] for o in outs]
    got = halo.merge_into_records(rec, owned[np.argsort(owned["id"], kind="stable")])
    assert_records_equal(got, want, f"{world} ranks in separate processes over the stand-in transport")
    # the plans: rank r's sends up == rank r + 1's receives from below, exchange by exchange; whole faces early and after the kick, sized in between and at the end
    plans = [o["plans"] for o in outs]                       # columns: exchangeNo, holdEvents, flags, sendHalo lo hi, sendMig lo hi, recvHalo lo hi, recvMig lo hi
    for r in range(world - 1):
        a, b = plans[r], plans[r + 1]
        assert (a[:, 0] == b[:, 0]).all() and (a[:, 1] == b[:, 1]).all() and (a[:, 2] == b[:, 2]).all()
        assert (a[:, 4] == b[:, 7]).all() and (a[:, 6] == b[:, 9]).all()      # r sends up (halo, migrants) == r + 1 expects from below
        assert (a[:, 8] == b[:, 3]).all() and (a[:, 10] == b[:, 5]).all()     # r expects from above == r + 1 sends down
    up = plans[0][:, 4]
    assert up[0] == 8192 and (up < 8192).any(), up                            # whole faces first, sized once the faces have been calm
    held = plans[0][:, 2] & 4
    assert held[6:9].all() and not held[-1], held                              # the strong kick of step 6 holds whole faces for three exchanges on every rank


def test_a_call_on_one_rank_only_is_refused_by_both_processes(pkg, tmp_path, fake_lib):
    """An impulse on ONE rank: that rank plans whole faces, its neighbour sized ones.  Over RCCL: two sized ncclSend / ncclRecv that do not match.  Here both
    processes end the very next step with SPH_ERR_STATE and the difference by name -- from the plans that crossed the link, before a face message was posted
    (the stand-in transport would have failed the run with SIZE MISMATCH otherwise)."""
    res = _run(tmp_path, fake_lib, 2, "one_sided", 14)
    for rc, so, se, _ in res:
        assert rc == 7 and "SPHERROR" in so and "one rank only" in so and "refused before any record moved" in so, so[-1500:] + se[-2000:]
        assert "SIZE MISMATCH" not in se


def test_a_rank_that_stops_calling_is_an_error_on_its_neighbour_not_a_hang(pkg, tmp_path, fake_lib):
    t0 = time.time()
    res = _run(tmp_path, fake_lib, 2, "gone", 14)
    (rc0, so0, se0, _), (rc1, so1, _, _) = res
    assert rc1 == 0 and "LEAVING" in so1
    assert rc0 == 7 and "SPHERROR" in so0, so0[-1500:] + se0[-2000:]
    assert time.time() - t0 < 120


def test_bench_two_ranks_engine_path_over_the_stand_in_transport(fake_lib):
    """`bench.py --gpus 2` as the driver launches it (torch.distributed.run, one rank per process), on this one GPU: torch's process group over gloo, the
    ENGINE's transport = its RCCL code path over the stand-in library (SPH_RCCL_LIBRARY).  The loop that is timed on a node -- SlabSimulation.from_config
    (transport "rccl"), priming exchange, boundary-first steps with the plans' handshake, the self-tests, the exchange diagnostics, the status gather --
    runs between two real ranks and prints its one line."""
    import json
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, SPH_BENCH_BACKEND="gloo", SPH_RCCL_LIBRARY=fake_lib, FAKE_RCCL_TIMEOUT_S="30", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "config2", "--steps", "8",
           "--warmup", "3", "--no-cpu-baseline", "--deadline", "60"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 8 and d["scaling"] == "weak" and d["value"] > 0
    assert "stand-in transport" in d["config"]["workload"]
    assert d["slab_status"]["overflow_on_any_rank"] is False
    ex = d["exchange"]
    assert ex is not None and len(ex["per_rank"]) == 2 and ex["rccl_selftest_faces_ms"] is not None
    sent = [r["sent_bytes_lo_hi"] for r in ex["per_rank"]]
    assert sent[0][0] == 0 and sent[0][1] > 0 and sent[1][0] > 0 and sent[1][1] == 0          # each rank has one neighbour, and bytes went out
