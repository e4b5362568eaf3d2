"""CPU stand-in for one slab rank's engine (TEST INFRASTRUCTURE): numpy pack / unpack with the
semantics of k_slab_pack / k_slab_unpack, the oracle as the substep.  It lets the gloo tests run
the real decomposition + exchange code of halo.py without a GPU and compare bit for bit with the
single-domain oracle."""
import numpy as np

F_GHOST1, F_GHOSTNZ, F_INACTIVE, F_HALO, F_DEAD = 1, 2, 4, 8, 16


class OracleSlabEngine:
    device = "cpu"

    def __init__(self, oracle, oparams, particles, ids, z0, z1, has_lo, has_hi):
        self.o, self.op = oracle, oparams
        self.z0, self.z1, self.has_lo, self.has_hi = z0, z1, has_lo, has_hi
        n = len(particles)
        self.pos = particles["pos"][:, :3].astype(np.float32).copy()
        self.vel = particles["vel"][:, :3].astype(np.float32).copy()
        self.acc = np.zeros((n, 3), np.float32)
        self.rho = particles["density"].copy()
        self.prs = particles["pressure"].copy()
        self.foam = particles["padA"].copy()
        self.id = np.asarray(ids, np.uint32).copy()
        g, a = particles["isGhost"], particles["isActive"]
        self.flags = ((g == 1) * F_GHOST1 | (g != 0) * F_GHOSTNZ | (a == 0) * F_INACTIVE).astype(np.uint32)

    def _cell_z(self, pz):
        g = self.o.grid_extents(self.op)
        q = ((pz - np.float32(g.gridMin[2])) / np.float32(g.cellSize)).astype(np.float32)
        return np.clip(np.floor(q), 0, g.dims[2] - 1).astype(np.int64)

    def _records(self, idx, flags):
        r = np.zeros((len(idx), 16), np.float32)
        r[:, 0:3] = self.pos[idx]
        r[:, 3:6] = self.vel[idx]
        r[:, 6] = self.rho[idx]
        r[:, 7] = self.prs[idx]
        r[:, 8] = self.foam[idx]
        r[:, 9] = self.id[idx].view(np.float32)
        r[:, 10] = flags.astype(np.uint32).view(np.float32)
        r[:, 12:15] = self.acc[idx]
        return r

    def pack(self, send_lo, send_hi):
        fl = self.flags
        live = (fl & F_DEAD) == 0
        stale = live & ((fl & F_HALO) != 0)
        own = live & ~stale
        cz = self._cell_z(self.pos[:, 2])
        go_lo, go_hi = own & (cz < self.z0), own & (cz >= self.z1)
        n_lo = n_hi = 0
        if self.has_lo:
            idx = np.nonzero(go_lo | (own & (cz == self.z0)))[0]
            f = np.where(go_lo[idx], fl[idx], fl[idx] | F_HALO)
            send_lo[: len(idx)] = self._t(self._records(idx, f))
            n_lo = len(idx)
        if self.has_hi:
            idx = np.nonzero(go_hi | (own & (cz == self.z1 - 1)))[0]
            f = np.where(go_hi[idx], fl[idx], fl[idx] | F_HALO)
            send_hi[: len(idx)] = self._t(self._records(idx, f))
            n_hi = len(idx)
        new = fl.copy()
        new[stale] |= F_DEAD
        new[go_lo] = np.where(cz[go_lo] == self.z0 - 1, fl[go_lo] | F_HALO, fl[go_lo] | F_DEAD)
        new[go_hi] = np.where(cz[go_hi] == self.z1, fl[go_hi] | F_HALO, fl[go_hi] | F_DEAD)
        self.flags = new
        return n_lo, n_hi

    @staticmethod
    def _t(a):
        import torch
        return torch.from_numpy(a)

    def unpack(self, recv_lo, n_lo, recv_hi, n_hi):
        parts = []
        if n_lo:
            parts.append(recv_lo[:n_lo].numpy())
        if n_hi:
            parts.append(recv_hi[:n_hi].numpy())
        if not parts:
            return
        r = np.concatenate(parts).astype(np.float32)
        self.pos = np.concatenate([self.pos, r[:, 0:3]])
        self.vel = np.concatenate([self.vel, r[:, 3:6]])
        self.acc = np.concatenate([self.acc, r[:, 12:15]])
        self.rho = np.concatenate([self.rho, r[:, 6]])
        self.prs = np.concatenate([self.prs, r[:, 7]])
        self.foam = np.concatenate([self.foam, r[:, 8]])
        self.id = np.concatenate([self.id, np.ascontiguousarray(r[:, 9]).view(np.uint32)])
        self.flags = np.concatenate([self.flags, np.ascontiguousarray(r[:, 10]).view(np.uint32)])

    def _live_sorted(self):
        live = np.nonzero((self.flags & F_DEAD) == 0)[0]
        return live[np.argsort(self.id[live], kind="stable")]

    def _compact(self, keep):
        for name in ("pos", "vel", "acc", "rho", "prs", "foam", "id", "flags"):
            setattr(self, name, getattr(self, name)[keep])

    def dispatch(self, dt=-1.0):
        keep = self._live_sorted()          # ascending global id => the oracle's canonical in-cell order
        self._compact(keep)
        n = len(self.id)
        P = np.zeros(n, self.o.PARTICLE_DTYPE)
        P["pos"][:, :3] = self.pos
        P["vel"][:, :3] = self.vel
        P["density"], P["pressure"], P["padA"] = self.rho, self.prs, self.foam
        P["isGhost"] = np.where(self.flags & F_GHOST1, 1, np.where(self.flags & F_GHOSTNZ, 2, 0))
        P["isActive"] = np.where(self.flags & F_INACTIVE, 0, 1)
        out = self.o.substep(P, self.op, dt=dt)
        own = (self.flags & F_HALO) == 0
        self.pos[own] = out["pos"][own, :3]
        self.vel[own] = out["vel"][own, :3]
        self.acc[own] = out["acc"][own, :3]
        self.rho[own], self.prs[own], self.foam[own] = out["density"][own], out["pressure"][own], out["padA"][own]

    def apply_wave_impulse(self, amplitude, wavelength, phase, direction, y_min, y_max):
        n = len(self.id)
        P = np.zeros(n, self.o.PARTICLE_DTYPE)
        P["pos"][:, :3] = self.pos
        P["vel"][:, :3] = self.vel
        P["isGhost"] = np.where(self.flags & F_GHOSTNZ, 1, 0)
        out = self.o.wave_impulse(P, amplitude, wavelength, phase, direction, y_min, y_max)
        self.vel = out["vel"][:, :3].copy()

    def set_option(self, *_):
        pass

    def download_owned(self):
        from conftest import PKG_NAME
        import importlib
        halo = importlib.import_module(PKG_NAME + ".halo")
        m = np.nonzero((self.flags & (F_DEAD | F_HALO)) == 0)[0]
        out = np.zeros(len(m), halo.OUT_DTYPE)
        out["pos"], out["vel"], out["acc"] = self.pos[m], self.vel[m], self.acc[m]
        out["density"], out["pressure"], out["padA"] = self.rho[m], self.prs[m], self.foam[m]
        out["id"], out["flags"] = self.id[m], self.flags[m]
        return out
