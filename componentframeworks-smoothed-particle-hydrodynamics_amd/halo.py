"""z-slab domain decomposition with a one-deep halo exchange.

The reference is single-GPU; this module is new functionality.  Rank r owns the global cell
layers [z0_r, z1_r) of the grid ComputeGridExtents defines (z is the slowest index of the
reference's cell = (cz*gy + cy)*gx + cx, so a slab is a contiguous cell range).  The
interaction radius equals the cell size, and the SPH pass reads only the dispatch-entry
snapshot of its neighbours, so ONE exchange per substep suffices:

    pack      every rank classifies its particles by current position and emits, per z-neighbour,
              one stream of 64-byte records: migrants (now owned by the neighbour) and copies of
              its boundary-layer particles (ghosts for the neighbour)
    exchange  with at most two neighbours.  Measured path: sph_slab_exchange of the C-ABI (grouped
              ncclSend/ncclRecv = RCCL over xGMI on the engine's stream, record counts in a header record,
              no host round trip).  Rehearsal / test paths: torch.distributed send/recv of counts then
              payload ("gloo" + host staging on CPU boxes), or direct hand-off inside one process
    unpack    received records are appended to the local state
    dispatch  the ordinary substep; ghosts are neighbour candidates, never targets

Summation order is (cell index, GLOBAL particle id), so every rank computes bit for bit what a
single-domain run computes: tests/test_halo_cpu.py (gloo, oracle stand-in engine) and
tests/test_gpu_slab.py (HIP engines) check exactly that.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import engine as _eng
from . import synthetic as _syn

REC_WORDS = 16          # 64-byte exchange record: pos3 vel3 rho prs foam id flags pad acc3 pad
OUT_DTYPE = np.dtype([("pos", "<f4", (3,)), ("vel", "<f4", (3,)), ("acc", "<f4", (3,)), ("density", "<f4"),
                      ("pressure", "<f4"), ("padA", "<f4"), ("id", "<u4"), ("flags", "<u4"), ("pad", "<u4", (2,))])
assert OUT_DTYPE.itemsize == 64


def _grid_key(p):
    """The grid ComputeGridExtents makes of the members (SPHFluid3D.cpp:354-376): the halo records in place were cut for one grid."""
    g = _eng.compute_grid_extents(p)
    return (tuple(float(v) for v in g.gridMin), float(g.cellSize), tuple(int(v) for v in g.dims))


def _paused(engine) -> bool:
    return bool(getattr(getattr(engine, "_p", None), "param_pause", 0))


def slab_range(gz: int, rank: int, world: int):
    """Even split of gz cell layers.  Every rank needs at least TWO layers: with a single layer a migrant that
    arrives from below would have to be forwarded upwards as a halo copy in the same substep (it is only known after
    this rank's own pack has run), and the upper neighbour would integrate with a neighbour missing."""
    if world > 1 and 2 * world > gz:
        raise ValueError(f"{world} ranks for {gz} cell layers: a z-slab needs at least 2 layers")
    return (gz * rank) // world, (gz * (rank + 1)) // world


class HipSlabEngine:
    """One rank's engine in slab mode (sph_create_slab / sph_slab_* of include/sph_abi.h)."""

    def __init__(self, particles, ids, params, z0, z1, has_lo, has_hi, capacity, stream=None):
        self._L = _eng.load_library()
        self._h = C.c_void_p()
        self._p = params
        rec = np.ascontiguousarray(particles, dtype=_eng.PARTICLE_DTYPE)
        idv = np.ascontiguousarray(ids, dtype=np.uint32)
        assert len(rec) == len(idv)
        _eng._check(self._L.sph_create_slab(C.byref(self._h), rec.ctypes.data_as(C.c_void_p), idv.ctypes.data_as(C.c_void_p), len(rec),
                                            C.byref(params), z0, z1, int(has_lo), int(has_hi), int(capacity), stream))
        self.capacity = int(capacity)

    device = "cuda"

    def pack(self, send_lo, send_hi):
        counts = (C.c_uint32 * 2)()
        plo = send_lo.data_ptr() if send_lo is not None else None
        phi = send_hi.data_ptr() if send_hi is not None else None
        clo = send_lo.shape[0] if send_lo is not None else 0
        chi = send_hi.shape[0] if send_hi is not None else 0
        _eng._check(self._L.sph_slab_pack(self._h, plo, phi, clo, chi, counts))
        return int(counts[0]), int(counts[1])

    def unpack(self, recv_lo, n_lo, recv_hi, n_hi):
        plo = recv_lo.data_ptr() if (recv_lo is not None and n_lo) else None
        phi = recv_hi.data_ptr() if (recv_hi is not None and n_hi) else None
        _eng._check(self._L.sph_slab_unpack(self._h, plo, n_lo, phi, n_hi))

    # -- exchange without host round trips (device-side counts; include/sph_abi.h) --------------------------
    def alloc_faces(self, face_cap: int):
        _eng._check(self._L.sph_slab_alloc_faces(self._h, int(face_cap)))
        self.face_cap = int(face_cap)

    def face_ptr(self, which: int) -> int:
        p = C.c_void_p()
        _eng._check(self._L.sph_slab_face_buffer(self._h, which, C.byref(p)))
        return p.value

    def pack_async(self):
        _eng._check(self._L.sph_slab_pack_async(self._h))

    def unpack_async(self, recv_lo_ptr, recv_hi_ptr, recv_cap):
        _eng._check(self._L.sph_slab_unpack_async(self._h, recv_lo_ptr, recv_hi_ptr, int(recv_cap)))

    def exchange(self, comm: "RcclComm"):
        """pack -> grouped ncclSend/ncclRecv with the z-neighbours -> unpack, on the engine's stream (sph_slab_exchange).  The engine gets the
        members first, as dispatch() and step_begin() do: the exchange cuts its records for the grid the NEXT dispatch will use."""
        _eng._check(self._L.sph_set_params(self._h, C.byref(self._p)))
        _eng._check(self._L.sph_slab_exchange(self._h, comm._h))

    # -- boundary-first substep: the exchange of the next substep beside the interior of the SPH pass ----------
    def step_begin(self, dt=-1.0):
        _eng._check(self._L.sph_set_params(self._h, C.byref(self._p)))
        _eng._check(self._L.sph_slab_step_begin(self._h, dt))

    def step_finish(self, comm: "RcclComm"):
        _eng._check(self._L.sph_slab_step_finish(self._h, comm._h))

    def step_finish_local(self, lo: "HipSlabEngine | None", hi: "HipSlabEngine | None"):
        _eng._check(self._L.sph_slab_step_finish_local(self._h, lo._h if lo is not None else None, hi._h if hi is not None else None))

    def sync(self, deadline=None):
        """deadline (seconds): poll instead of block; a transfer whose neighbour never came raises (SPH_ERR_TIMEOUT) instead of hanging."""
        if deadline is None:
            _eng._check(self._L.sph_sync(self._h))
        else:
            _eng._check(self._L.sph_sync_deadline(self._h, float(deadline)))

    # -- agreement of the two ends of a link (round 5) ----------------------------------------------------------
    def set_verify(self, mode: int):
        _eng._check(self._L.sph_slab_set_verify(self._h, int(mode)))

    def set_deadline(self, seconds: float):
        _eng._check(self._L.sph_slab_set_deadline(self._h, float(seconds)))

    def plan(self):
        """(SphSlabIntent of the last sized exchange, ms its handshake waited on the host)"""
        out = _eng.SphSlabIntent()
        ms = C.c_float()
        _eng._check(self._L.sph_slab_plan(self._h, C.byref(out), C.byref(ms)))
        return out, float(ms.value)

    def debug_tight_messages(self, on: bool):
        _eng._check(self._L.sph_slab_debug_tight_messages(self._h, 1 if on else 0))

    def status(self):
        """[records packed for lo, for hi, slots in use, -, flags]; raises for the flags that lost records (1, 2, 4, 8).  Flag 16 (a particle
        crossed more cell layers in one substep than the exchange follows) is a notice: it comes back in [4]."""
        out = (C.c_uint32 * 5)()
        _eng._check(self._L.sph_slab_status(self._h, out))
        return [int(x) for x in out]

    def clear_flags(self, mask: int = 16):
        _eng._check(self._L.sph_slab_clear_flags(self._h, int(mask)))

    def message_bytes(self):
        """{sent to lo, sent to hi (last exchange), bytes of a whole face lo, hi}"""
        out = (C.c_uint64 * 4)()
        _eng._check(self._L.sph_slab_message_bytes(self._h, out))
        return [int(x) for x in out]

    def step_times(self):
        """ms of the last boundary-first step (SPH_OPT_TIMING on): pack, transfer, unpack, end of the exchange, end of the SPH pass (the last two from the step's start)"""
        out = (C.c_float * 5)()
        _eng._check(self._L.sph_slab_step_times(self._h, out))
        return [float(x) for x in out]

    def dispatch(self, dt=-1.0):
        _eng._check(self._L.sph_set_params(self._h, C.byref(self._p)))
        _eng._check(self._L.sph_dispatch(self._h, dt))

    def apply_wave_impulse(self, amplitude, wavelength, phase, direction, y_min, y_max):
        _eng._check(self._L.sph_apply_wave_impulse(self._h, amplitude, wavelength, phase, _eng._f3(direction), y_min, y_max))

    def set_option(self, option, value):
        _eng._check(self._L.sph_set_option(self._h, option, value))

    def kernel_times(self, reset=False):
        ms = (C.c_double * len(_eng.KERNEL_CLASSES))()
        cnt = (C.c_int64 * len(_eng.KERNEL_CLASSES))()
        _eng._check(self._L.sph_kernel_times(self._h, ms, cnt, 1 if reset else 0))
        return {k: (ms[i], cnt[i]) for i, k in enumerate(_eng.KERNEL_CLASSES)}

    def download_owned(self) -> np.ndarray:
        out = np.zeros(self.capacity, OUT_DTYPE)
        n = C.c_size_t()
        _eng._check(self._L.sph_slab_download(self._h, out.ctypes.data_as(C.c_void_p), len(out), C.byref(n)))
        return out[: n.value].copy()

    def close(self):
        if self._h:
            self._L.sph_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class RcclComm:
    """ncclCommInitRank behind the C-ABI (sph_comm_*): one rank per process on the current HIP device.  `bcast` hands
    rank 0's 128-byte unique id to the other ranks (a callable bytes -> bytes, e.g. a torch.distributed broadcast)."""

    def __init__(self, rank: int, world: int, bcast):
        self._L = _eng.load_library()
        self._h = C.c_void_p()
        buf = (C.c_ubyte * 128)()
        if rank == 0:
            _eng._check(self._L.sph_comm_unique_id(buf))
        data = bcast(bytes(buf))
        idb = (C.c_ubyte * 128).from_buffer_copy(data)
        _eng._check(self._L.sph_comm_create(C.byref(self._h), idb, rank, world))
        self.rank, self.world = rank, world

    def selftest(self, nbytes: int = 1 << 20):
        """One grouped ncclSend + ncclRecv of `nbytes` from this rank to itself, checked on the host (sph_comm_selftest)."""
        _eng._check(self._L.sph_comm_selftest(self._h, int(nbytes)))

    def selftest_gbs(self, nbytes: int = 1 << 24) -> float:
        """GB/s of one grouped ncclSend + ncclRecv of `nbytes` from this rank to itself (hipEvents around the group; checked on the host)."""
        ms = C.c_float()
        _eng._check(self._L.sph_comm_selftest_timed(self._h, int(nbytes), C.byref(ms)))
        return nbytes / max(ms.value, 1e-6) / 1e6

    def selftest_faces(self, face_cap: int, counts) -> float:
        """The engine's exchange pattern with the rank as its own two neighbours (sph_comm_selftest_faces): plans first, then per neighbour
        two send / receive pairs of unequal sizes in one group; every byte compared.  Returns the ms of the faces' group."""
        ms = C.c_float()
        arr = (C.c_uint32 * 4)(*[int(x) for x in counts])
        _eng._check(self._L.sph_comm_selftest_faces(self._h, int(face_cap), arr, C.byref(ms)))
        return float(ms.value)

    def close(self):
        if self._h:
            self._L.sph_comm_destroy(self._h)
            self._h = C.c_void_p()


class HaloExchange:
    """Counts + payload exchange with the (at most two) z-neighbours over torch.distributed.

    transport "direct": tensors are sent as they are (device tensors with backend nccl = RCCL).
    transport "host":   device tensors are staged through host memory (backend gloo)."""

    def __init__(self, rank, world, group=None, transport="direct", device="cpu"):
        import torch
        self.torch = torch
        self.rank, self.world, self.group = rank, world, group
        self.transport = transport
        self.device = device
        self.lo = rank - 1 if rank > 0 else None
        self.hi = rank + 1 if rank < world - 1 else None
        cdev = device if transport == "direct" else "cpu"
        self._cnt_send = torch.zeros(2, dtype=torch.int64, device=cdev)
        self._cnt_recv = torch.zeros(2, dtype=torch.int64, device=cdev)

    def _batch(self, ops):
        dist = self.torch.distributed
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
            if self.transport == "direct" and str(self.device) != "cpu":
                # req.wait() only orders torch's current stream; the engine packs / unpacks on its own stream, so the
                # receive must have LANDED (and the sends must have left) before the engine touches the buffers again
                self.torch.cuda.synchronize()

    def exchange(self, send_lo, n_lo, send_hi, n_hi, recv_lo, recv_hi):
        """Returns (m_lo, m_hi): records received from the lower / upper neighbour, written to the
        head of recv_lo / recv_hi."""
        torch, dist = self.torch, self.torch.distributed
        self._cnt_send[0] = n_lo
        self._cnt_send[1] = n_hi
        self._cnt_recv.zero_()
        ops = []
        if self.lo is not None:
            ops.append(dist.P2POp(dist.isend, self._cnt_send[0:1], self.lo, self.group))
            ops.append(dist.P2POp(dist.irecv, self._cnt_recv[0:1], self.lo, self.group))
        if self.hi is not None:
            ops.append(dist.P2POp(dist.isend, self._cnt_send[1:2], self.hi, self.group))
            ops.append(dist.P2POp(dist.irecv, self._cnt_recv[1:2], self.hi, self.group))
        self._batch(ops)
        m_lo, m_hi = (int(x) for x in self._cnt_recv.tolist())
        if (recv_lo is not None and m_lo > recv_lo.shape[0]) or (recv_hi is not None and m_hi > recv_hi.shape[0]):
            raise _eng.SphError(f"halo receive buffer too small ({m_lo}, {m_hi})")
        host = self.transport == "host"
        stage = []
        ops = []

        def snd(buf, n, peer):
            if peer is not None and n:
                t = buf[:n].cpu() if host else buf[:n]
                ops.append(dist.P2POp(dist.isend, t, peer, self.group))

        def rcv(buf, n, peer):
            if peer is not None and n:
                t = torch.empty((n, buf.shape[1]), dtype=buf.dtype) if host else buf[:n]
                if host:
                    stage.append((buf, n, t))
                ops.append(dist.P2POp(dist.irecv, t, peer, self.group))

        snd(send_lo, n_lo, self.lo)
        rcv(recv_lo, m_lo, self.lo)
        snd(send_hi, n_hi, self.hi)
        rcv(recv_hi, m_hi, self.hi)
        self._batch(ops)
        for buf, n, t in stage:
            buf[:n].copy_(t)
        return m_lo, m_hi


class SlabSimulation:
    """One rank of the decomposed simulation, with the reference's method names where they apply."""

    def __init__(self, engine, exchange, rank, world, z_range, grid_dims, face_capacity, make_buffer):
        self.engine, self.exchange = engine, exchange
        self.rank, self.world = rank, world
        self.z0, self.z1 = z_range
        self.grid_dims = tuple(grid_dims)
        self.has_lo, self.has_hi = rank > 0, rank < world - 1
        self.send_lo = make_buffer(face_capacity) if self.has_lo else None
        self.recv_lo = make_buffer(face_capacity) if self.has_lo else None
        self.send_hi = make_buffer(face_capacity) if self.has_hi else None
        self.recv_hi = make_buffer(face_capacity) if self.has_hi else None
        self.last_counts = (0, 0, 0, 0)
        self._n_owned0 = None

    # -- construction from a synthetic config (bench.py) ---------------------------------------
    @classmethod
    def from_config(cls, cfg, params, rank, world, stream=None, group=None, transport="direct", face_factor=1.6, slot_factor=1.5):
        import torch
        gx, gy, gz = cfg.grid
        z0, z1 = slab_range(gz, rank, world)
        rec, gid = _syn.make_particles(cfg, z_cells=(z0, z1))
        # a face carries one boundary layer of copies plus the migrants (face_factor cell layers' worth of the average
        # layer); slots: slot_factor x the owned particles (a collapsing column moves particles into the lower slabs)
        # + two faces.  Spare slots cost almost nothing: blocks past the live count leave at once.
        # The face capacity is the MESSAGE SIZE of sph_slab_exchange (fixed-size buffers, the count travels in record 0):
        # it must be the same number on both ends of a link, so it is derived from the global configuration, never from
        # this rank's own particle count (ranks hold 82 or 83 lattice planes of the same column), and agreed on by an
        # all-reduce when a process group exists.
        per_layer = max(1, -(-cfg.n // max(1, gz)))
        face_cap = int(per_layer * face_factor + 8192)
        if world > 1:
            import torch.distributed as dist
            if dist.is_initialized():
                t = torch.tensor([face_cap], dtype=torch.int64, device=torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else "cpu")
                dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
                face_cap = int(t.item())
        cap = int(len(rec) * slot_factor + 2 * face_cap)
        eng = HipSlabEngine(rec, gid.astype(np.uint32), params, z0, z1, rank > 0, rank < world - 1, cap, stream=stream)
        dev = torch.device("cuda", torch.cuda.current_device())
        if transport == "rccl":
            import torch.distributed as dist

            def bcast(data: bytes) -> bytes:
                if world == 1 or not dist.is_initialized():
                    return data
                t = torch.tensor(list(data), dtype=torch.uint8, device=dev if dist.get_backend(group) == "nccl" else "cpu")
                dist.broadcast(t, src=0, group=group)
                return bytes(t.cpu().tolist())

            eng.alloc_faces(face_cap)
            sim = cls(eng, RcclComm(rank, world, bcast), rank, world, (z0, z1), cfg.grid, 0, lambda n: None)
            sim.has_lo, sim.has_hi = rank > 0, rank < world - 1
            sim._n_owned0 = len(rec)
            return sim
        ex = HaloExchange(rank, world, group=group, transport=transport, device=dev)
        sim = cls(eng, ex, rank, world, (z0, z1), cfg.grid, face_cap,
                  lambda n: torch.zeros((n, REC_WORDS), dtype=torch.float32, device=dev))
        sim._n_owned0 = len(rec)
        return sim

    # -- the substep -------------------------------------------------------------------------------
    overlap = True       # RCCL path: boundary-first substeps (sph_slab_step_*), the exchange hidden behind the interior of the SPH pass

    def DispatchCompute(self, overrideDt: float = -1.0):
        if isinstance(self.exchange, RcclComm):          # the measured path: everything behind the C-ABI, no host round trip
            if self.overlap:
                # the first substep's halos; every later exchange rides inside a step.  A member that moves the grid (box centre /
                # half / angles, h, grid_cap) invalidates the halo records in place: prime again (sph_slab_step_begin would refuse)
                key = _grid_key(self.engine._p)
                if not getattr(self, "_primed", False) or key != getattr(self, "_grid_key", None):
                    if not self.engine._p.param_pause:
                        _eng._check(self.engine._L.sph_set_params(self.engine._h, C.byref(self.engine._p)))
                        self.engine.exchange(self.exchange)
                        self._primed = True
                        self._grid_key = key
                self.engine.step_begin(overrideDt)
                self.engine.step_finish(self.exchange)
                return
            self.engine.exchange(self.exchange)          # (a no-op while param_pause is set, like the dispatch behind it)
            self.engine.dispatch(overrideDt)
            return
        if _paused(self.engine):                         # a paused DispatchCompute is a no-op (SPHFluid3D.cpp:432): no exchange either, or every
            return                                       # paused substep would append one more set of halo copies behind the live slots
        n_lo, n_hi = self.engine.pack(self.send_lo, self.send_hi)
        m_lo, m_hi = self.exchange.exchange(self.send_lo, n_lo, self.send_hi, n_hi, self.recv_lo, self.recv_hi)
        self.engine.unpack(self.recv_lo, m_lo, self.recv_hi, m_hi)
        self.engine.dispatch(overrideDt)
        self.last_counts = (n_lo, n_hi, m_lo, m_hi)

    SimulateSubstep = DispatchCompute

    def ApplyWaveImpulse(self, amplitude, wavelength, phase, dir, yMin=-_eng.FLT_MAX, yMax=_eng.FLT_MAX):
        self.engine.apply_wave_impulse(amplitude, wavelength, phase, dir, yMin, yMax)

    # -- helpers used by bench.py / tests --------------------------------------------------------
    def set_option(self, option, value):
        self.engine.set_option(option, value)

    def set_deadline(self, seconds: float):
        """Limit of every wait for a neighbour rank (the plans' handshake of an exchange, sync(deadline=...)): SphError (SPH_ERR_TIMEOUT) instead of a hang."""
        self.engine.set_deadline(seconds)
        self._deadline = float(seconds)

    def sync(self, deadline=None):
        """Drain this rank's streams; with a deadline (default: the one set by set_deadline) the streams are polled and a transfer whose neighbour never
        posted its half raises instead of blocking for ever -- print the error and end the process."""
        d = deadline if deadline is not None else getattr(self, "_deadline", None)
        self.engine.sync(deadline=d)

    def kernel_times(self, reset=False):
        return self.engine.kernel_times(reset)

    def num_owned(self) -> int:
        return self._n_owned0 if self._n_owned0 is not None else len(self.engine.download_owned())

    def local_grid(self) -> dict:
        gx, gy, _ = self.grid_dims
        return {"dims": (gx, gy, self.z1 - self.z0 + 2), "numCells": gx * gy * (self.z1 - self.z0 + 2)}

    def download_owned(self) -> np.ndarray:
        return self.engine.download_owned()


class SlabGroup:
    """W slab ranks inside one process, exchanging through direct buffer hand-off (no
    torch.distributed).  Used to validate the HIP pack/unpack/ghost path on a single GPU and as
    the reference driver for the CPU stand-in engine."""

    def __init__(self, sims):
        self.sims = sims

    @classmethod
    def from_particles(cls, particles, ids, params, grid_dims, world, make_engine, make_buffer, face_capacity, cell_z):
        """cell_z: global z cell of every particle (decides the initial owner)."""
        sims = []
        gz = grid_dims[2]
        for r in range(world):
            z0, z1 = slab_range(gz, r, world)
            m = (cell_z >= z0) & (cell_z < z1)
            eng = make_engine(particles[m], ids[m], params, z0, z1, r > 0, r < world - 1)
            sims.append(SlabSimulation(eng, None, r, world, (z0, z1), grid_dims, face_capacity, make_buffer))
        return cls(sims)

    def enable_async(self, face_capacity: int):
        """Exchange through the engines' own face buffers with device-side counts (sph_slab_pack_async /
        sph_slab_unpack_async): the path sph_slab_exchange runs around its ncclSend/ncclRecv, minus the transport."""
        for s in self.sims:
            s.engine.alloc_faces(face_capacity)
        self._async_cap = int(face_capacity)

    def enable_overlap(self, face_capacity: int):
        """Boundary-first substeps (sph_slab_step_begin / sph_slab_step_finish_local): every engine packs on its second stream as
        soon as the slots next to its faces are computed, copies its neighbours' send faces device to device and unpacks, all
        beside the interior of its SPH pass -- the schedule sph_slab_step_finish runs with RCCL in place of the copies."""
        self.enable_async(face_capacity)
        self._overlap = True
        self._primed = False

    def _exchange_async(self):
        for s in self.sims:
            s.engine.pack_async()
        for s in self.sims:                       # the engines run on their own streams: a pack must have finished
            s.engine.sync()                       # before the neighbour reads it (RCCL gives that order on real ranks)
        for r, s in enumerate(self.sims):
            lo = self.sims[r - 1].engine.face_ptr(1) if r > 0 else None                 # lower neighbour's "send hi"
            hi = self.sims[r + 1].engine.face_ptr(0) if r < len(self.sims) - 1 else None  # upper neighbour's "send lo"
            s.engine.unpack_async(lo, hi, self._async_cap)
        for s in self.sims:
            s.engine.sync()

    def DispatchCompute(self, overrideDt: float = -1.0):
        if getattr(self, "_overlap", False):
            key = _grid_key(self.sims[0].engine._p)
            paused = bool(self.sims[0].engine._p.param_pause)
            if (not self._primed or key != getattr(self, "_grid_key", None)) and not paused:
                for s in self.sims:               # (a grid that moved between two steps: the halo records in place were cut for the old one)
                    _eng._check(s.engine._L.sph_set_params(s.engine._h, C.byref(s.engine._p)))
                self._exchange_async()
                self._primed = True
                self._grid_key = key
            for s in self.sims:                   # every begin before any finish: a finish waits for the neighbours' packs
                s.engine.step_begin(overrideDt)
            for r, s in enumerate(self.sims):
                s.engine.step_finish_local(self.sims[r - 1].engine if r > 0 else None,
                                           self.sims[r + 1].engine if r < len(self.sims) - 1 else None)
            return
        if _paused(self.sims[0].engine):                  # (the overlap path above leaves the no-op to the engines: sph_slab_step_begin / _finish_local)
            return
        if getattr(self, "_async_cap", 0):
            for s in self.sims:
                s.engine.pack_async()
            for s in self.sims:                       # the engines run on their own streams: a pack must have finished
                s.engine.sync()                       # before the neighbour reads it (RCCL gives that order on real ranks)
            for r, s in enumerate(self.sims):
                lo = self.sims[r - 1].engine.face_ptr(1) if r > 0 else None                 # lower neighbour's "send hi"
                hi = self.sims[r + 1].engine.face_ptr(0) if r < len(self.sims) - 1 else None  # upper neighbour's "send lo"
                s.engine.unpack_async(lo, hi, self._async_cap)
            for s in self.sims:
                s.engine.sync()
            for s in self.sims:
                s.engine.dispatch(overrideDt)
            return
        counts = [s.engine.pack(s.send_lo, s.send_hi) for s in self.sims]
        for r, s in enumerate(self.sims):
            lo_buf, m_lo, hi_buf, m_hi = None, 0, None, 0
            if r > 0:
                lo_buf, m_lo = self.sims[r - 1].send_hi, counts[r - 1][1]
            if r < len(self.sims) - 1:
                hi_buf, m_hi = self.sims[r + 1].send_lo, counts[r + 1][0]
            s.engine.unpack(lo_buf, m_lo, hi_buf, m_hi)
            s.last_counts = (counts[r][0], counts[r][1], m_lo, m_hi)
        for s in self.sims:
            s.engine.dispatch(overrideDt)

    def ApplyWaveImpulse(self, *a, **kw):
        for s in self.sims:
            s.ApplyWaveImpulse(*a, **kw)

    def download(self) -> np.ndarray:
        """All owned records of all ranks, sorted by global id."""
        parts = [s.download_owned() for s in self.sims]
        out = np.concatenate(parts) if parts else np.zeros(0, OUT_DTYPE)
        return out[np.argsort(out["id"], kind="stable")]


def merge_into_records(initial: np.ndarray, owned: np.ndarray, stepped: bool = True) -> np.ndarray:
    """Global 80-byte array in ORIGINAL order from the ranks' owned records: dynamic fields come
    from the owned records (by global id = original index), constant fields from `initial`.
    Ghosts (isGhost == 1) keep their record, except that an ACTIVE ghost that has been through a substep
    (`stepped`) carries what SPHFluid.comp:72-83 writes: vel = acc = vec4(0), density = rho0, pressure = 0."""
    out = initial.copy()
    idx = owned["id"].astype(np.int64)
    fluid = (initial["isGhost"][idx] != 1)
    i_f = idx[fluid]
    out["pos"][i_f, :3] = owned["pos"][fluid]
    out["vel"][i_f, :3] = owned["vel"][fluid]
    out["acc"][i_f, :3] = owned["acc"][fluid]
    out["acc"][i_f, 3] = 0.0
    out["density"][i_f] = owned["density"][fluid]
    out["pressure"][i_f] = owned["pressure"][fluid]
    out["padA"][i_f] = owned["padA"][fluid]
    if stepped:
        ghost = (~fluid) & (initial["isActive"][idx] != 0)
        i_g = idx[ghost]
        out["vel"][i_g, :] = 0.0
        out["acc"][i_g, :] = 0.0
        out["density"][i_g] = owned["density"][ghost]      # rho0 of the parameters the substep ran with
        out["pressure"][i_g] = 0.0
    return out
