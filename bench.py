#!/usr/bin/env python3
"""bench.py -- particle-substeps/s of the SPH substep hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" = one DispatchCompute (ClearGrid -> BuildGrid -> SPHFluid -> OBBConstraints, the
call of SPHFluid3D.cpp:431-522) over the whole particle set, inputs resident in HBM.
N = 1 runs BASELINE.json configs[2] (4 194 304 particles, 128^3 grid, fp32), the
configuration the metric is quoted on.  N > 1 runs BASELINE.json configs[4]: every rank owns a
256 x 256 x 64-cell slab with 8 388 608 particles (weak scaling along z), OBBConstraints on,
ApplyWaveImpulse every 16th substep, one-deep halo exchange per substep -- see DESIGN.md
"Multi-GPU".  Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = "componentframeworks-smoothed-particle-hydrodynamics_amd"

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md "HBM3E peak BW"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--slab-path", action="store_true", help="rehearsal: run the N > 1 code path (z-slab engine, RCCL communicator, boundary-first substeps, status gather) with however many ranks there are, also one")
    # defaults keep the timed window on the SPECIFIED workload (the seeded lattice state): the 35-unit
    # column of configs[2] is not hydrostatically stable at k = 2000, g = -980 and collapses within a
    # few hundred substeps, which turns the run into a different, denser workload
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="auto", help="auto (config3 at N = 1, weak5 at N > 1) | config2 | config3 | weak5 (BASELINE configs[4] slab)")
    ap.add_argument("--settled-after", type=int, default=300, help="N = 1: after the timed run, continue the SAME run to this substep and record the settled regime (0 = skip)")
    ap.add_argument("--neighbor", type=int, default=3, choices=(1, 2, 3), help="SPH pass: 3 = k_sph_walk (engine default), 2 = k_sph_list (round 2), 1 = k_sph_slow (plain per-target sweeps)")
    ap.add_argument("--aos", default="lazy", choices=["eager", "lazy"], help="lazy (engine default): the 80-byte records are materialised once per frame; eager: by every substep")
    ap.add_argument("--deadline", type=float, default=120.0, help="N > 1: seconds a rank waits for its neighbours (handshake of an exchange, drain of its streams) before it prints where it stands and exits 3")
    ap.add_argument("--frame-substeps", type=int, default=16, help="lazy, N = 1: materialise the 80-byte record array (sph_device_particles, what a renderer binds) after every this many "
                    "substeps INSIDE the timed region (Scene0p.h:48 maxSubstepsPerFrame = 16) and once more at its end")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=0, help="substeps of the CPU sample (0 = auto, about 10-30 s)")
    ap.add_argument("--grid-build", default="sort", choices=["sort", "ll"], help="counting sort (default) or the reference's linked lists (A/B)")
    ap.add_argument("--no-breakdown", action="store_true", help="skip the extra untimed pass with per-kernel hipEvents")
    return ap.parse_args()


def cpu_baseline(pkg, rec, sp, cpu_steps):
    """The oracle (OpenMP port of the same shader math) timed on this host's cores, on the
    same particle set; the sample is a reduced number of substeps of the same workload."""
    # the CPU baseline is the oracle built -O3 -march=native ON THE HOST THAT RUNS IT (the in-tree
    # liboracle.so is built in the build container for a generic AVX2 host, so that it runs anywhere).  Same source, same
    # -ffp-contract=off: same bits, only the instruction selection differs.
    import subprocess
    import tempfile
    native = os.path.join(tempfile.gettempdir(), f"liboracle_native_{os.getuid()}.so")
    flags = "-O3 -march=native -std=c11 -fPIC -ffp-contract=off -fno-fast-math -fopenmp -shared"
    built = subprocess.run(["gcc", *flags.split(), "-o", native, os.path.join(ROOT, "oracle", "sph_oracle.c"), "-lm"], capture_output=True, text=True)
    if built.returncode == 0:
        os.environ["SPH_ORACLE_LIB"] = native
    from oracle import oracle as o
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import to_oracle_params
    op = to_oracle_params(o, sp)
    threads = o.max_threads()
    import numpy as np
    import ctypes as C
    cur = rec.copy()
    scratch = np.zeros_like(cur)
    L = o.lib()

    def run(k):
        t0 = time.perf_counter()
        for _ in range(k):
            L.sph_oracle_substep(cur.ctypes.data_as(C.c_void_p), scratch.ctypes.data_as(C.c_void_p), len(cur), C.byref(op), -1.0)
        return time.perf_counter() - t0

    t1 = run(1)                                   # also first-touch / grid warm-up
    k = cpu_steps if cpu_steps > 0 else max(1, min(50, int(15.0 / max(t1, 1e-3))))
    t = run(k)
    return {
        "value": len(rec) * k / t, "unit": "particle-substeps/s", "cores": threads, "kind": "port",
        "sample": f"{k} substeps of the same {len(rec)}-particle workload (after 1 warm-up substep), oracle/sph_oracle.c with OpenMP, "
                  + ("gcc -O3 -march=native on this host" if built.returncode == 0 else "in-tree -O3 -mavx2 build (native build failed)") + f", {t:.2f} s",
    }


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if "WORLD_SIZE" not in os.environ and args.gpus > 1:
            # `python bench.py --gpus N` on its own: start the N ranks as a CHILD process (one per GPU, the driver's own
            # launch line) before anything here has touched the GPU, relay its output and leave with its exit code.
            # Never an exec: a process that has initialised HIP must not replace itself.
            import socket
            import subprocess
            with socket.socket() as sck:
                sck.bind(("127.0.0.1", 0))
                port = sck.getsockname()[1]
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
                   "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
            env = dict(os.environ)
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            raise SystemExit(subprocess.run(cmd, env=env).returncode)
        args.gpus = world
    multi = args.gpus > 1 or args.slab_path              # the z-slab code path (--slab-path: also with a single rank, as a rehearsal)

    import numpy as np
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    # SPH_BENCH_BACKEND=gloo rehearses the N > 1 path on a box with fewer GPUs than ranks (ranks share
    # devices, halo records are staged through the host); the measured configuration is nccl = RCCL.
    backend = os.environ.get("SPH_BENCH_BACKEND", "nccl")
    # The ENGINE's transport (sph_slab_exchange / sph_slab_step_finish behind the C-ABI) is RCCL whenever torch's is; with SPH_RCCL_LIBRARY naming a stand-in
    # for librccl (tests/fake_rccl/: messages between processes on ONE GPU through shared memory) the engine's RCCL code path also runs in a gloo rehearsal --
    # the driver's exact N > 1 loop, minus xGMI (tests/test_gpu_fake_rccl.py::test_bench_two_ranks_engine_path_over_the_stand_in_transport).
    engine_rccl = backend == "nccl" or bool(os.environ.get("SPH_RCCL_LIBRARY"))
    if backend != "nccl":
        local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    pkg = importlib.import_module(PKG)
    syn = pkg.synthetic

    if multi:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "WORLD_SIZE" not in os.environ:                   # --slab-path without a launcher: a group of one
            import socket
            with socket.socket() as sck:
                sck.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(sck.getsockname()[1]))
            os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
        from importlib import import_module
        halo = import_module(PKG + ".halo")
    else:
        dist = None

    wl = args.workload
    if wl == "auto":
        wl = "config3" if not multi else "weak5"
    if wl == "config2":
        base = syn.CONFIGS[2]
    elif wl == "config3":
        base = syn.CONFIGS[3]
    elif wl == "weak5":
        base = syn.weak_config(1)
    else:
        raise SystemExit(f"unknown workload {wl}")
    gx, gy, gz = base.grid
    cfg = syn.BenchConfig(base.index, base.name, base.n * args.gpus, (gx, gy, gz * args.gpus), base.spacing_factor, args.gpus)
    sp = pkg.default_params(**syn.params_fields(cfg))
    stream = torch.cuda.current_stream().cuda_stream

    if not multi:
        rec, _ = syn.make_particles(cfg)
        sim = pkg.SPHFluidGPU.from_particles(rec, sp, stream=stream)
        n_local, n_total = len(rec), len(rec)
    else:
        sim = halo.SlabSimulation.from_config(cfg, sp, rank, world, stream=stream, transport="rccl" if engine_rccl else "host")
        if engine_rccl:
            sim.engine.set_deadline(args.deadline)     # every wait for a neighbour is bounded: a rank that is stuck says so and exits non-zero
        rec = None
        n_local, n_total = sim.num_owned(), cfg.n
    sim.set_option(pkg.SPH_OPT_NEIGHBOR_KERNEL, args.neighbor)
    sim.set_option(pkg.SPH_OPT_AOS_MODE, 0 if args.aos == "eager" else 1)
    if args.grid_build == "ll":
        sim.set_option(pkg.SPH_OPT_GRID_BUILD, 1)

    def die(ex, where):
        """A rank that cannot go on (a neighbour that planned another exchange, or never came: SPH_ERR_STATE / SPH_ERR_TIMEOUT of the handshake or of the
        deadline-bounded sync) says where it stands and ends -- the process, not just the call: device work is still queued behind a transfer that
        will never complete, so a normal interpreter exit could wait for ever.  The other ranks run into their own deadline."""
        plan = None
        try:
            p, hs = sim.engine.plan()
            plan = {"exchange": int(p.exchangeNo), "step": int(p.stepNo), "hold_events": int(p.holdEvents), "send_halo_mig_lo_hi": [int(p.sendHalo[0]), int(p.sendMig[0]), int(p.sendHalo[1]), int(p.sendMig[1])],
                    "recv_halo_mig_lo_hi": [int(p.recvHalo[0]), int(p.recvMig[0]), int(p.recvHalo[1]), int(p.recvMig[1])], "handshake_wait_ms": round(hs, 3)}
        except Exception:                                    # noqa: BLE001
            pass
        print(json.dumps({"bench_failed": True, "rank": rank, "world": world, "where": where, "substeps_issued": wave["n"], "error": str(ex), "plan": plan}), file=sys.stderr, flush=True)
        os._exit(3)

    slab_rccl = multi and engine_rccl

    def barrier():
        if slab_rccl:                              # never a blind wait: the engine's streams are polled against the deadline first
            try:
                sim.engine.sync(deadline=args.deadline)
            except pkg.SphError as ex:
                die(ex, "sync")
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    dt = -1.0
    # BASELINE.json configs[4] ("OBBConstraints + WaveImpulse enabled"): the scene's wave forcing, once per 16
    # substeps with the values of Scene0p.h:144-147; other workloads run without it
    wave_every = 16 if wl == "weak5" else 0
    wave = {"n": 0, "phase": 0.0}

    # lazy records, one GPU: the renderer's view (the 80-byte array in original order) is brought up to date once per
    # frame, as Scene0p binds it once per frame after up to 16 substeps (Scene0p.cpp:1482-1494, :1625); that work is part
    # of the timed region.  z-slab runs never held a fused record array (owned records are gathered by download_owned()).
    frame = args.frame_substeps if (not multi and args.aos == "lazy") else 0
    materialised = {"n": 0}

    def present():
        sim.device_particles()
        materialised["n"] += 1

    def step():
        if wave_every and wave["n"] % wave_every == 0:
            sim.ApplyWaveImpulse(1.5, 3.0, wave["phase"], (0.0, 1.0, 0.0))
            wave["phase"] += 4.0 * 16 * 1e-3
        wave["n"] += 1
        try:
            sim.DispatchCompute(dt)
        except pkg.SphError as ex:
            if slab_rccl:
                die(ex, "substep")
            raise
        if frame and wave["n"] % frame == 0:
            present()

    # N > 1: the transport's health before anything is timed (one grouped ncclSend + ncclRecv of a face-sized message from the rank
    # to itself: the only send / receive a one-GPU box can execute; on a node it still says that RCCL moves bytes on this rank)
    selftest_gbs = None
    selftest_faces_ms = None
    if multi and engine_rccl and isinstance(sim.exchange, halo.RcclComm):
        try:
            sim.exchange.selftest_gbs(1 << 20)
            selftest_gbs = round(sim.exchange.selftest_gbs(int(sim.engine.message_bytes()[2] or sim.engine.message_bytes()[3] or (1 << 24)) & ~3), 1)
            # ... and the engine's own exchange pattern with the rank as both of its neighbours: the 64-byte plans, then two send / receive pairs of
            # unequal sizes per neighbour in one group, every byte compared (sph_comm_selftest_faces)
            fc = int(getattr(sim.engine, "face_cap", 0) or 8192)
            selftest_faces_ms = round(sim.exchange.selftest_faces(fc, (fc * 3 // 5, fc // 2, fc // 50, fc // 40)), 3)
        except Exception as ex:                              # noqa: BLE001
            print(f"[rank {rank}] RCCL self-test failed: {ex}", file=sys.stderr, flush=True)
    for _ in range(args.warmup):
        step()
    sim.set_option(pkg.SPH_OPT_TIMING, 2)          # hipEvents around the dominant kernel only
    sim.kernel_times(reset=True)
    barrier()
    materialised["n"] = 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if not multi and args.aos == "lazy":
        present()                                  # the timed region ends with a current record array
    barrier()
    elapsed = time.perf_counter() - t0
    records_materialised = materialised["n"]
    kt = sim.kernel_times(reset=True)
    sim.set_option(pkg.SPH_OPT_TIMING, 0)
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    breakdown = None
    if not args.no_breakdown:
        sim.set_option(pkg.SPH_OPT_TIMING, 1)
        sim.kernel_times(reset=True)
        nb = min(args.steps, 20)
        for _ in range(nb):
            step()
        bt = sim.kernel_times(reset=True)
        sim.set_option(pkg.SPH_OPT_TIMING, 0)
        breakdown = {k: round(ms / nb * 1e3, 2) for k, (ms, cnt) in bt.items() if cnt}
        if not multi and args.grid_build == "sort":
            # untimed, for the record: the two other (bit-identical) SPH passes at the state the run has reached
            alt = {}
            for name, kind in (("k_sph_walk", 3), ("k_sph_list", 2), ("k_sph_slow", 1)):
                sim.set_option(pkg.SPH_OPT_NEIGHBOR_KERNEL, kind)
                sim.set_option(pkg.SPH_OPT_TIMING, 2)
                sim.kernel_times(reset=True)
                for _ in range(3):
                    step()
                ms, cnt = sim.kernel_times(reset=True)["sph"]
                alt[name] = round(ms / max(cnt, 1) * 1e3, 1)
            sim.set_option(pkg.SPH_OPT_TIMING, 0)
            sim.set_option(pkg.SPH_OPT_NEIGHBOR_KERNEL, args.neighbor)
            breakdown["sph_pass_variants_after_run"] = alt

    # The benchmark workload is not stationary (the column compresses under its own weight):
    # the headline window above is the specified lattice state; this block records, untimed for `value`, what the
    # SAME run costs once the fluid has settled into the compressed regime every long-running scene lives in.
    settled = None
    if not multi and args.settled_after > 0 and not args.no_breakdown:
        done = wave["n"]
        while wave["n"] < args.settled_after:
            step()
        ns = 20
        sim.set_option(pkg.SPH_OPT_TIMING, 2)
        sim.kernel_times(reset=True)
        barrier()
        ts = time.perf_counter()
        for _ in range(ns):
            step()
        barrier()
        tse = time.perf_counter() - ts
        sms, scnt = sim.kernel_times(reset=True)["sph"]
        sim.set_option(pkg.SPH_OPT_TIMING, 0)
        settled = {"after_substeps": max(args.settled_after, done), "substeps_timed": ns, "sph_pass_us": round(sms / max(scnt, 1) * 1e3, 1),
                   "ms_per_step": round(tse / ns * 1e3, 3), "particle_substeps_per_s": n_total * ns / tse}

    # For the record (untimed for `value`): the same workload and window on a fresh engine with the records updated
    # by every substep (SPH_OPT_AOS_MODE 0), i.e. what the fused scattered record update costs.
    aos_eager = None
    if not multi and args.aos == "lazy" and not args.no_breakdown and args.grid_build == "sort":
        sim2 = pkg.SPHFluidGPU.from_particles(rec, sp, stream=stream)
        sim2.set_option(pkg.SPH_OPT_NEIGHBOR_KERNEL, args.neighbor)
        sim2.set_option(pkg.SPH_OPT_AOS_MODE, 0)
        for _ in range(args.warmup):
            sim2.DispatchCompute(dt)
        sim2.set_option(pkg.SPH_OPT_TIMING, 2)
        sim2.kernel_times(reset=True)
        barrier()
        te = time.perf_counter()
        for _ in range(args.steps):
            sim2.DispatchCompute(dt)
        barrier()
        te = time.perf_counter() - te
        ems, ecnt = sim2.kernel_times(reset=True)["sph"]
        sim2.close()
        aos_eager = {"value": n_total * args.steps / te, "ms_per_step": round(te / args.steps * 1e3, 4), "sph_pass_us": round(ems / max(ecnt, 1) * 1e3, 1),
                     "note": "same workload, warm-up and window with SPH_OPT_AOS_MODE 0 (the SPH pass also updates every 80-byte record), fresh engine, not the headline"}

    # z-slab runs, untimed for `value`: where the exchange of a boundary-first step sat relative to the SPH pass, per rank (hipEvents on the
    # exchange stream and on the engine's stream): pack -> transfer -> unpack, the end of the exchange and the end of the pass measured from
    # the step's start.  The transfer was hidden behind the interior iff exchange_end <= pass_end; sent bytes / transfer ms = the link's rate.
    exchange_diag = None
    if multi and engine_rccl and isinstance(sim.exchange, halo.RcclComm) and getattr(sim, "overlap", False):
        sim.set_option(pkg.SPH_OPT_TIMING, 1)
        acc = [0.0] * 5
        nd = 5
        for _ in range(nd):
            step()
            for i, v in enumerate(sim.engine.step_times()):
                acc[i] += v / nd
        sim.kernel_times(reset=True)
        sim.set_option(pkg.SPH_OPT_TIMING, 0)
        mb = sim.engine.message_bytes()
        try:
            hs_ms = sim.engine.plan()[1]
        except pkg.SphError:
            hs_ms = 0.0
        mine = acc + [float(v) for v in mb] + [float(hs_ms)]
        tst = torch.tensor(mine, dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        gathered = [torch.zeros_like(tst) for _ in range(world)]
        dist.all_gather(gathered, tst)
        exchange_diag = {"per_rank": [{"pack_ms": round(g[0].item(), 4), "transfer_ms": round(g[1].item(), 4), "unpack_ms": round(g[2].item(), 4),
                                       "exchange_end_ms": round(g[3].item(), 4), "pass_end_ms": round(g[4].item(), 4),
                                       "hidden_behind_the_pass": bool(g[3].item() <= g[4].item()),
                                       "sent_bytes_lo_hi": [int(g[5].item()), int(g[6].item())], "face_bytes": int(max(g[7].item(), g[8].item())),
                                       "handshake_host_wait_ms": round(g[9].item(), 4)} for g in gathered],
                         "substeps_averaged": nd, "rccl_selftest_gbs": selftest_gbs, "rccl_selftest_faces_ms": selftest_faces_ms,
                         "verify": "the plans of both ends of every link are compared before a sized message is posted (sph_slab_set_verify 1); handshake_host_wait_ms = host time the last handshake waited for its neighbours",
                         "note": "hipEvents of the last boundary-first steps, untimed for value; halo copies travel as 40-byte records, messages are sized from the counts of two exchanges ago"}

    # z-slab runs: overflow of a face buffer or of the slot capacity is flagged on the device, never fatal in the
    # substep loop; a run that dropped records is not a measurement, so every rank's flags go into the line.
    slab_status = None
    if multi:
        try:
            st = sim.engine.status()
            mine = [int(st[0]), int(st[1]), int(st[2]), 2 if int(st[4]) & 16 else 0]
        except Exception as ex:                          # noqa: BLE001  (SphError: the status call reports an overflow as an error)
            mine = [0, 0, 0, 1]
            print(f"[rank {rank}] slab status: {ex}", file=sys.stderr, flush=True)
        tst = torch.tensor(mine, dtype=torch.int64, device="cuda" if backend == "nccl" else "cpu")
        gathered = [torch.zeros_like(tst) for _ in range(world)]
        dist.all_gather(gathered, tst)
        slab_status = {"records_lo_hi_live_per_rank": [[int(v) for v in g[:3].tolist()] for g in gathered],
                       "overflow_on_any_rank": any(int(g[3].item()) == 1 for g in gathered),
                       # a particle crossed more than one cell layer in a substep (error bit 16): the run went on, but it no longer
                       # equals the single-domain run bit for bit
                       "multi_layer_move_on_any_rank": any(int(g[3].item()) == 2 for g in gathered),
                       "face_capacity_records": int(sim.engine.face_cap) if hasattr(sim.engine, "face_cap") else None}

    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    g = sim.ComputeGridExtents() if not multi else sim.local_grid()
    C_local = int(g.numCells) if not multi else int(g["numCells"])
    sph_ms, sph_launches = kt["sph"]
    sph_avg_s = (sph_ms / max(sph_launches, 1)) * 1e-3
    alg_bytes = 164 * n_local + 4 * C_local          # SPHFluid pass, per launch
    achieved = alg_bytes / sph_avg_s / 1e9 if sph_avg_s > 0 else 0.0
    # HBM traffic, VALU instruction counts and L1 cache-line accesses are NOT measured in this run: they come from separate
    # rocprofv3 --pmc passes over this same command (tools/profile_bench.sh), summarised for exactly the timed launches in
    # profiles/r05_bench_counters.json.  That file carries the hash of the engine sources it was measured on; a file from
    # other sources is refused (traffic stays null and traffic_source says why).
    traffic = traffic_source = None
    valu = None
    cpath = os.path.join(ROOT, "profiles", "r05_bench_counters.json")
    kname = "k_sph_ll" if args.grid_build == "ll" else (None, "k_sph_slow", "k_sph_list", "k_sph_walk")[args.neighbor]
    if os.path.exists(cpath) and not multi:
        try:
            cj = json.load(open(cpath))
            here = pkg.build.csrc_hash()
            if cj.get("csrc_hash") != here:
                traffic_source = f"profiles/r05_bench_counters.json was measured on other engine sources ({cj.get('csrc_hash')} != {here}): not used"
            elif cj.get("workload") == wl and cj.get("kernel") == kname:
                traffic = cj.get("hbm_bytes_per_launch")
                traffic_source = cj.get("source")
                if cj.get("valu_wave_insts_per_launch") and sph_avg_s > 0:
                    # issue cost measured on gfx950 (profiles/r03_valu_rate2.txt): about 1.05 ns per wave-instruction and SIMD for the
                    # add / mul / fma / integer add-and-or class, 1.75 ns for v_pk_*, v_cmp, v_cndmask, v_max/min and the VOP3-only forms
                    valu = {"wave_insts_per_launch": cj["valu_wave_insts_per_launch"],
                            "wave_insts_per_s_per_simd": cj["valu_wave_insts_per_launch"] / sph_avg_s / 1024.0,
                            "tcp_line_accesses_per_launch": cj.get("tcp_line_accesses_per_launch"),
                            "tcp_line_accesses_per_cycle_per_cu_at_2GHz": (cj["tcp_line_accesses_per_launch"] / 256.0 / (sph_avg_s * 2.0e9)) if cj.get("tcp_line_accesses_per_launch") else None,
                            "source": cj.get("source")}
        except Exception as ex:                              # noqa: BLE001
            traffic, traffic_source = None, f"profiles/r05_bench_counters.json unreadable: {ex}"

    out = {
        "metric": "particle-substeps/sec", "value": n_total * args.steps / elapsed, "unit": "particle-substeps/s",
        "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": f"{wl}: {base.n} particles and a {gx}x{gy}x{gz}-cell grid per GPU (BASELINE.json configs[{base.index - 1}]"
                        + (f", weak-scaled along z to {args.gpus} slabs" if multi else "") + ")"
                        + ("" if backend == "nccl" else (f" [REHEARSAL over {backend}, the engine's RCCL path over a stand-in transport: not a measurement]" if engine_rccl else f" [REHEARSAL over {backend}, host-staged halos: not a measurement]")),
                        # (--slab-path with one rank: the same code path, a communicator of one, no neighbour to exchange with)
            "particles": n_total, "grid": list(cfg.grid), "h": 0.28, "dt": 1e-3, "spacing_over_h": base.spacing_factor,
            "neighbor_kernel": (None, "k_sph_slow", "k_sph_list", "k_sph_walk")[args.neighbor],
            "aos": (args.aos if multi or args.aos == "eager" else
                    f"lazy: 80-byte records materialised (sph_device_particles) every {frame} substeps and at the end, inside the timed region: {records_materialised} times in {args.steps} substeps"),
            "pipeline": (("bin+scan+scatter+rank -> sph(27-cell, OBB fused" if args.grid_build == "sort" else "ll clear+build -> sph(list walk, OBB fused")
                         + (", AoS update fused)" if (args.aos == "eager" and not multi) else ") -> record write-back once per frame" if not multi else ")"))
                        + (" + halo exchange" if multi else ""),
        },
        "roofline": {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_source": traffic_source, "kernel": kname,
            "window": {"kernel": kname, "first_launch": args.warmup, "launches": int(sph_launches),
                       "note": "0-based index among this kernel's launches of the run; profiles/r05_bench_kernel_window.json holds the rocprofv3 --kernel-trace average of exactly these launches"},
            "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_us": sph_avg_s * 1e6, "launches_timed": int(sph_launches),
            # the same window with the 80-byte records updated by EVERY substep (what the 164 N figure prices; aos_eager below)
            "frac_records_updated_every_substep": (alg_bytes / (aos_eager["sph_pass_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS) if aos_eager and aos_eager.get("sph_pass_us") else None,
            "whole_substep_algorithmic_GBs": (260 * n_local + 8 * C_local) / (elapsed / args.steps) / 1e9,
        },
        "kernels_us_per_substep": breakdown,
        "slab_status": slab_status,
        "exchange": exchange_diag,
        "aos_eager": aos_eager,
        "settled": settled,
        "valu": valu,
    }
    if not args.no_cpu_baseline and not multi:
        out["cpu_baseline"] = cpu_baseline(pkg, rec, sp, args.cpu_steps)
    elif not args.no_cpu_baseline:
        out["cpu_baseline"] = None
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
