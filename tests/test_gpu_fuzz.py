"""Seeded random scenes through k_sph_walk (and now and then the other passes) against the oracle, bit for bit.

What the fixed scenes of test_gpu_parity.py do not vary: the kernel length h (h^2 > 1 selects the kernel's other template
instance: no clamp modifier), non-cubic grids down to three cells, Poisson-like (not lattice) particle placement at 0.3 .. 20
particles per cell, speeds up to the velocity cap, clumps, ghosts, rotated containers of every shape, the time step."""
import numpy as np
import pytest

from conftest import assert_records_equal, to_oracle_params

pytestmark = pytest.mark.gpu


def _scene(pkg, seed):
    rng = np.random.default_rng(1000 + seed)
    h = float(rng.choice([0.1, 0.28, 0.28, 0.6, 1.25, 2.0]))
    dims = [int(rng.integers(3, 22)) for _ in range(3)]
    half = [(g / 2.0 - 1.0) * h - 0.01 * h for g in dims]
    lam = float(rng.choice([0.3, 1.0, 2.0, 2.0, 6.0, 20.0]))
    fill = float(rng.uniform(0.3, 1.0))                       # fraction of the box height that holds particles
    cells = max(1, (dims[0] - 2) * (dims[2] - 2) * max(1, int((dims[1] - 2) * fill)))
    n = int(min(14000, max(1, lam * cells)))
    dt = float(rng.choice([1e-3, 5e-4, 2e-3]))
    s = 0.8 * h
    sp = pkg.default_params(
        param_h=h, param_mass=float(np.float32(1000.0 * s ** 3)), param_restDensity=1000.0, param_gasConstant=2000.0,
        param_viscosity=3.5, param_gravityX=0.0, param_gravityY=-980.0 * h / 0.28, param_gravityZ=0.0, param_surfaceTension=0.0728,
        param_timeStep=dt, param_foamGen=1.0, param_foamVelRef=8.0, param_boxCenter=(0.0, 0.0, 0.0), param_boxHalf=tuple(half),
        param_boxEulerDeg=(0.0, 0.0, 0.0), param_shapeType=0, param_wallRestitution=0.15, param_wallFriction=0.02, grid_cap=160)
    if rng.random() < 0.5:
        sp.param_shapeType = int(rng.integers(0, 15))
        for a in range(3):
            sp.param_shapeAux[a] = float(rng.uniform(0.2, 1.0))
    if rng.random() < 0.4:
        for a in range(3):
            sp.param_boxEulerDeg[a] = float(rng.uniform(-40, 40))
    rec = np.zeros(n, pkg.PARTICLE_DTYPE)
    lo = np.array([-half[0], -half[1], -half[2]], np.float32)
    ext = np.array([2 * half[0], 2 * half[1] * fill, 2 * half[2]], np.float32)
    rec["pos"][:, :3] = lo + rng.random((n, 3)).astype(np.float32) * ext
    rec["pos"][:, 3] = 1.0
    vcap = 0.4 * h / dt
    sigma = float(rng.choice([0.0, 0.02, 0.1, 0.5])) * vcap
    rec["vel"][:, :3] = rng.normal(0, 1, (n, 3)).astype(np.float32) * np.float32(sigma)
    rec["isActive"][:] = 1
    if n > 200 and rng.random() < 0.4:                         # a clump: hundreds of particles inside one cell
        k = int(rng.integers(50, min(n, 1500)))
        rec["pos"][:k, :3] = rec["pos"][0, :3] + rng.normal(0, 0.2 * h, (k, 3)).astype(np.float32)
    if n > 100 and rng.random() < 0.4:                         # ghosts of every kind, some inactive particles
        g = rng.choice(n, size=n // 20, replace=False)
        rec["isGhost"][g] = rng.choice([1, 1, 3], size=len(g))
        rec["isActive"][g[: len(g) // 2]] = 0
    if n > 100 and rng.random() < 0.3:                         # particles outside the grid
        rec["pos"][rng.choice(n, size=n // 50 + 1, replace=False), 0] += np.float32(4.0 * half[0] + 3 * h)
    steps = int(rng.integers(1, 4))
    return rec, sp, steps, dict(h=h, dims=dims, n=n, per_cell=lam, sigma_over_cap=sigma / vcap, dt=dt, steps=steps, shape=int(sp.param_shapeType))


@pytest.mark.parametrize("seed", list(range(28)))
def test_random_scene_against_the_oracle(pkg, oracle, seed):
    rec, sp, steps, what = _scene(pkg, seed)
    op = to_oracle_params(oracle, sp)
    want = oracle.substep(rec, op, steps=steps)
    for neighbor in ((3, 2, 1) if seed % 7 == 0 else (3,)):
        f = pkg.SPHFluidGPU.from_particles(rec, sp)
        f.set_option(pkg.SPH_OPT_NEIGHBOR_KERNEL, neighbor)
        f.DispatchN(steps)
        got = f.download()
        f.close()
        assert_records_equal(got, want, f"seed {seed}, pass {neighbor}: {what}")
    assert np.isfinite(want["pos"]).all()


def test_the_fuzz_reaches_both_template_instances_and_the_fallbacks(pkg):
    """The scene generator is only worth something if it reaches what it is meant to reach."""
    hs, dense, fast = set(), 0, 0
    for seed in range(28):
        _, _, _, what = _scene(pkg, seed)
        hs.add(what["h"] * what["h"] > 1.0)
        dense += what["per_cell"] >= 6.0
        fast += what["sigma_over_cap"] >= 0.1
    assert hs == {True, False} and dense >= 3 and fast >= 3


# The decomposition's one assumption: no particle crosses more than one cell layer in z per substep.  SPHFluid.comp moves a particle
# with the UNCAPPED velocity (v + a dt) dt and OBBConstraints.comp projects it, so violent scenes (20 particles per cell, particles
# placed outside a rotated container) break it.  Where such a jump makes the slabs differ from the single domain the engine must SAY
# so (error bit 16 of sph_slab_status: raised by the SPH pass of the owning rank when the pack's reduced scan would miss the
# particle, or by the rank that receives a migrant it cannot place); a run that stays silent must be bit-exact, and a report is
# only legitimate where the oracle sees a jump of more than one layer.
def _layer_jumps(oracle, op, rec, g, dims, steps):
    gm, cs = np.float32(g.gridMin[2]), np.float32(g.cellSize)
    layer = lambda P: np.clip(np.floor(((P["pos"][:, 2] - gm) / cs).astype(np.float32)), 0, dims[2] - 1).astype(np.int64)
    cur, worst = rec, 0
    for _ in range(steps):
        nxt = oracle.substep(cur, op)
        moving = rec["isGhost"] != 1
        worst = max(worst, int(np.abs(layer(nxt) - layer(cur))[moving].max(initial=0)))
        cur = nxt
    return worst, cur


@pytest.mark.parametrize("seed", list(range(28)))
def test_random_scene_as_z_slabs_with_boundary_first_steps(pkg, oracle, seed):
    """The same random scenes cut into 2 or 3 z-slabs (several engines in one process, the exchange of the next substep beside
    the interior of the SPH pass): the merged records equal the oracle's, no face or slot overflow -- or the exchange reports
    that a particle jumped further than it can follow."""
    import importlib
    import torch
    from conftest import PKG_NAME
    halo = importlib.import_module(PKG_NAME + ".halo")
    rec, sp, steps, what = _scene(pkg, seed)
    g = pkg.compute_grid_extents(sp)
    dims = tuple(int(v) for v in g.dims)
    world = 3 if dims[2] >= 9 else 2
    if dims[2] < 2 * world:
        pytest.skip(f"grid {dims} too thin for {world} slabs")
    steps += 2
    op = to_oracle_params(oracle, sp)
    worst_jump, want = _layer_jumps(oracle, op, rec, g, dims, steps)
    q = ((rec["pos"][:, 2] - np.float32(g.gridMin[2])) / np.float32(g.cellSize)).astype(np.float32)
    cz = np.clip(np.floor(q), 0, dims[2] - 1).astype(np.int64)
    ids = np.arange(len(rec), dtype=np.uint32)

    def make_engine(p, i, prm, z0, z1, lo, hi):
        return halo.HipSlabEngine(p, i, prm, z0, z1, lo, hi, capacity=int(len(rec) * 1.2) + 8192)   # any slab may end up holding everything

    face = len(rec) + 1024
    grp = halo.SlabGroup.from_particles(rec, ids, sp, dims, world, make_engine,
                                        lambda n: torch.zeros((n, halo.REC_WORDS), dtype=torch.float32, device="cuda"), face, cz)
    grp.enable_overlap(face)
    for _ in range(steps):
        grp.DispatchCompute()
    reported = []
    for s in grp.sims:
        st = s.engine.status()                     # (raises for the flags that lose records)
        assert st[4] & ~16 == 0, st
        if st[4] & 16:
            reported.append(st)
    if reported:                                   # allowed only where a particle crossed more layers than the exchange follows (3, or the slab's thickness)
        thin = min(s.z1 - s.z0 for s in grp.sims)
        assert worst_jump > 3 or (worst_jump > 1 and world > 2 and thin <= worst_jump), f"seed {seed}: largest layer jump {worst_jump} (thinnest slab {thin}), yet flag 16: {what}"
        for s in grp.sims:
            s.engine.close()
        return
    # silence means: the decomposition followed every particle
    got = halo.merge_into_records(rec, grp.download())
    assert_records_equal(got, want, f"seed {seed} as {world} slabs: {what}")


# ---- random call sequences on one engine -------------------------------------------------------------------------------------
def _mirror(pkg, oracle, seed):
    """A random scene, then 14-22 random calls of the SPHFluidGPU surface, each mirrored on the oracle."""
    rng = np.random.default_rng(7000 + seed)
    rec, sp, _, what = _scene(pkg, seed + 100)
    if len(rec) > 6000:                                   # keep the oracle quick: the interest here is the engine's state handling
        rec = rec[:6000].copy()
    f = pkg.SPHFluidGPU.from_particles(rec, sp)
    op = to_oracle_params(oracle, f.params)
    want = rec.copy()
    log = []
    h = float(sp.param_h)
    targets = rng.uniform(-2 * h, 2 * h, (64, 4)).astype(np.float32)
    f.SetStencilTargets(targets)
    of = oracle.default_fountain(mode=0, seed=int(f.fountainSeed))        # FountainRecycle.comp, step 6 of DispatchCompute, while fountainMode is on
    fountain = lambda: of if of.mode else None
    for _ in range(int(rng.integers(14, 23))):
        opn = rng.choice(["dispatch", "dispatch", "dispatch", "dispatch_n", "wave", "vortex", "attractor", "curl", "stencil", "param", "option",
                          "download", "upload", "device", "container", "fountain"])
        if opn == "dispatch":
            dt = float(rng.choice([-1.0, -1.0, 5e-4, 1.5e-3]))
            f.DispatchCompute(dt); want = oracle.substep(want, op, dt=dt, fountain=fountain())
        elif opn == "dispatch_n":
            k = int(rng.integers(2, 5))
            f.DispatchN(k); want = oracle.substep(want, op, steps=k, fountain=fountain())
        elif opn == "fountain":
            if of.mode and rng.random() < 0.4:
                f.fountainMode = 0; of.mode = 0
            else:
                off = (float(rng.uniform(-h, h)), float(rng.uniform(-6 * h, 0.0)), float(rng.uniform(-h, h)))
                rate, level = float(rng.uniform(50, 600)), float(rng.uniform(0.5 * h, 6 * h))
                f.fountainMode = 1; f.fountainOffset = off; f.fountainDrainPerSec = rate; f.fountainDrainLevel = level
                of.mode = 1; of.drainPerSec = rate; of.drainLevel = level
                for a_ in range(3):
                    of.offset[a_] = off[a_]
        elif opn == "wave":
            a = (float(rng.uniform(0.2, 2.0)), float(rng.uniform(1.0, 4.0)), float(rng.uniform(0, 6.0)), (0.3, 1.0, -0.2), -2.0 * h * 5, 3.0 * h * 5)
            f.ApplyWaveImpulse(*a); want = oracle.wave_impulse(want, *a)
        elif opn == "vortex":
            a = (float(rng.uniform(-1, 1)), float(rng.uniform(-0.3, 0.3)))
            f.ApplyVortexImpulse(*a); want = oracle.vortex_impulse(want, op, *a)
        elif opn == "attractor":
            a = ((float(rng.uniform(-h, h)), float(rng.uniform(-h, h)), 0.1), float(rng.uniform(0.1, 0.8)), float(rng.uniform(2 * h, 8 * h)))
            f.ApplyAttractorImpulse(*a); want = oracle.attractor_impulse(want, *a)
        elif opn == "curl":
            a = (float(rng.uniform(0.1, 0.6)), float(rng.uniform(0.3, 1.5)), float(rng.uniform(0, 3)))
            f.ApplyCurlFlow(*a); want = oracle.curl_flow(want, *a)
        elif opn == "stencil":
            a = (float(rng.uniform(0.01, 0.1)), float(rng.uniform(0.0, 0.05)))
            f.ApplyStencilAttract(*a); want = oracle.stencil_attract(want, targets, *a)
        elif opn == "param":
            which = rng.choice(["viscosity", "gasConstant", "gravityY", "timeStep", "surfaceTension", "wallRestitution", "pause"])
            if which == "viscosity": f.param_viscosity = float(rng.uniform(1, 8))
            elif which == "gasConstant": f.param_gasConstant = float(rng.uniform(500, 4000))
            elif which == "gravityY": f.param_gravityY = float(rng.uniform(-1500, 200))
            elif which == "timeStep": f.param_timeStep = float(rng.choice([5e-4, 1e-3, 2e-3]))
            elif which == "surfaceTension": f.param_surfaceTension = float(rng.uniform(0, 0.2))
            elif which == "wallRestitution": f.param_wallRestitution = float(rng.uniform(0, 0.9))
            else: f.param_pause = int(1 - f.param_pause)
            op = to_oracle_params(oracle, f.params)
            opn = f"param {which}"
        elif opn == "container":                             # the walls (and with them the grid) change under the fluid
            which = rng.choice(["half", "euler", "shape"])
            if which == "half":
                f.param_boxHalf = tuple(float(x) * float(rng.uniform(0.8, 1.15)) for x in f.param_boxHalf)
            elif which == "euler":
                f.param_boxEulerDeg = tuple(float(rng.uniform(-30, 30)) for _ in range(3))
            else:
                f.param_shapeType = int(rng.integers(0, 15))
            op = to_oracle_params(oracle, f.params)
            opn = f"container {which}"
        elif opn == "option":
            which = rng.choice(["neighbor", "aos", "graph"])
            if which == "neighbor": f.set_option(pkg.SPH_OPT_NEIGHBOR_KERNEL, int(rng.choice([1, 2, 3, 3, 3])))
            elif which == "aos": f.set_option(pkg.SPH_OPT_AOS_MODE, int(rng.integers(0, 2)))
            else: f.set_option(pkg.SPH_OPT_GRAPH, int(rng.integers(0, 2)))
            opn = f"option {which}"
        elif opn == "download":
            assert_records_equal(f.download(), want, f"seed {seed} after {log}")
        elif opn == "upload":                                # the host edits records (as Scene0p does for its dye patterns) and uploads them
            cur = f.download()
            assert_records_equal(cur, want, f"seed {seed} before upload after {log}")
            pick = rng.choice(len(cur), size=max(1, len(cur) // 10), replace=False)
            cur["vel"][pick, :3] += rng.normal(0, 5, (len(pick), 3)).astype(np.float32)
            cur["isGhost"][pick[: len(pick) // 4]] = 1
            cur["isActive"][pick[: len(pick) // 8]] = 0
            f.upload(cur); want = cur.copy()
        elif opn == "device":
            assert f.device_particles() != 0
        log.append(opn)
    assert int(f.fountainSeed) == int(of.seed), (int(f.fountainSeed), int(of.seed), log)
    return f, want, log, what


@pytest.mark.parametrize("seed", list(range(16)))
def test_random_call_sequences_against_the_oracle(pkg, oracle, seed):
    """Every public call in random order -- substeps, the five impulses, live parameter and container edits, option switches
    (pass, record mode, graph replay), uploads of edited records, downloads -- leaves the engine where the oracle is."""
    f, want, log, what = _mirror(pkg, oracle, seed)
    try:
        assert_records_equal(f.download(), want, f"seed {seed}: {what}: {log}")
    finally:
        f.close()


@pytest.mark.parametrize("seed", list(range(12)) + [5123, 5219])     # (5123, 5219: the two seeds of round 4's last sweep that raised a notice after a box move)
def test_random_call_sequences_on_z_slabs(pkg, oracle, seed):
    """What a multi-GPU caller does between the substeps of a slab group -- wave impulses, live parameter edits, a container that
    changes shape under the fluid, time-step overrides -- in random order, with boundary-first steps: the merged records equal the
    oracle's unless the exchange reports a particle it could not follow (and then the oracle must show such a jump)."""
    import importlib
    import torch
    from conftest import PKG_NAME
    halo = importlib.import_module(PKG_NAME + ".halo")
    rng = np.random.default_rng(9000 + seed)
    rec, sp, _, what = _scene(pkg, seed + 300)
    if len(rec) > 6000:
        rec = rec[:6000].copy()
    g = pkg.compute_grid_extents(sp)
    dims = tuple(int(v) for v in g.dims)
    world = 3 if dims[2] >= 9 else 2
    if dims[2] < 2 * world:
        pytest.skip(f"grid {dims} too thin for {world} slabs")
    op = to_oracle_params(oracle, sp)
    gm, cs = np.float32(g.gridMin[2]), np.float32(g.cellSize)
    layer = lambda P: np.clip(np.floor(((P["pos"][:, 2] - gm) / cs).astype(np.float32)), 0, dims[2] - 1).astype(np.int64)
    cz = layer(rec)
    ids = np.arange(len(rec), dtype=np.uint32)

    def make_engine(p, i, prm, z0, z1, lo, hi):
        return halo.HipSlabEngine(p, i, prm, z0, z1, lo, hi, capacity=int(len(rec) * 1.2) + 8192)

    face = len(rec) + 1024
    grp = halo.SlabGroup.from_particles(rec, ids, sp, dims, world, make_engine,
                                        lambda n: torch.zeros((n, halo.REC_WORDS), dtype=torch.float32, device="cuda"), face, cz)
    grp.enable_overlap(face)
    want, worst, log = rec.copy(), 0, []
    stepped = False                                          # (a sequence may draw no unpaused substep at all: the ghosts' records then stay as uploaded)
    shifts = 0                                               # grid moves by a cell ("move") SINCE THE LAST UNPAUSED SUBSTEP: they add to what that substep's exchange has to follow at once
    moving = rec["isGhost"] != 1
    for _ in range(int(rng.integers(10, 18))):
        opn = rng.choice(["dispatch", "dispatch", "dispatch", "wave", "param", "shape", "pause", "move"])
        if opn == "pause":                                   # a few paused DispatchCompute calls: no-ops, exchange included (SPHFluid3D.cpp:432)
            sp.param_pause = 1
            for _ in range(int(rng.integers(1, 4))):
                grp.DispatchCompute()
            sp.param_pause = 0
        elif opn == "move":                                  # the container (and with it the grid) moves by about a cell between two steps
            old_c = tuple(sp.param_boxCenter)
            sp.param_boxCenter[2] = old_c[2] + float(rng.choice([-1.0, 1.0])) * float(sp.param_h)
            if tuple(int(v) for v in pkg.compute_grid_extents(sp).dims) != dims:
                sp.param_boxCenter[2] = old_c[2]
                continue
            op = to_oracle_params(oracle, sp)
            g = pkg.compute_grid_extents(sp)
            gm, cs = np.float32(g.gridMin[2]), np.float32(g.cellSize)
            shifts += 1                                      # every particle's layer index moved by one with the grid: on top of what a substep adds
        elif opn == "dispatch":
            dt = float(rng.choice([-1.0, -1.0, 5e-4]))
            grp.DispatchCompute(dt)
            stepped = True
            nxt = oracle.substep(want, op, dt=dt)
            # what THIS exchange had to follow: the substep's own jump + the cells the grid moved under the particles since the substep before
            # (round 4 added the moves of the whole call sequence, which excused any notice after three moves: VERDICT r04)
            worst = max(worst, int(np.abs(layer(nxt) - layer(want))[moving].max(initial=0)) + shifts)
            shifts = 0
            want = nxt
        elif opn == "wave":
            a = (float(rng.uniform(0.2, 2.0)), float(rng.uniform(1.0, 4.0)), float(rng.uniform(0, 6.0)), (0.3, 1.0, -0.2), -1e9, 1e9)
            grp.ApplyWaveImpulse(*a); want = oracle.wave_impulse(want, *a)
        elif opn == "param":
            which = rng.choice(["viscosity", "gasConstant", "gravityY", "timeStep"])
            if which == "viscosity": sp.param_viscosity = float(rng.uniform(1, 8))
            elif which == "gasConstant": sp.param_gasConstant = float(rng.uniform(500, 4000))
            elif which == "gravityY": sp.param_gravityY = float(rng.uniform(-1500, 200))
            else: sp.param_timeStep = float(rng.choice([5e-4, 1e-3]))
            op = to_oracle_params(oracle, sp)
            opn = f"param {which}"
        else:                                                # box <-> sphere of the same extent: the grid stays, the walls move by cells
            old = int(sp.param_shapeType)
            sp.param_shapeType = 1 if old == 0 else 0
            if tuple(int(v) for v in pkg.compute_grid_extents(sp).dims) != dims:
                sp.param_shapeType = old
                continue
            op = to_oracle_params(oracle, sp)
        log.append(opn)
    reported = []
    for s in grp.sims:
        st = s.engine.status()
        assert st[4] & ~16 == 0, st
        if st[4] & 16:
            reported.append(st)
    if reported:
        thin = min(s.z1 - s.z0 for s in grp.sims)
        eff = worst
        assert eff > 3 or (eff > 1 and world > 2 and thin <= eff), f"seed {seed}: largest layer jump of one exchange (substep + grid moves before it) {worst} (thinnest slab {thin}), yet flag 16: {what}: {log}"
    else:
        got = halo.merge_into_records(rec, grp.download(), stepped=stepped)
        assert_records_equal(got, want, f"seed {seed} as {world} slabs: {what}: {log}")
    # Whatever state the sequence left the group in (whole faces or sized messages, any exchange number): a call on ONE engine only must end the
    # next step with SPH_ERR_STATE on both ends of that engine's links, before a record moves (the plans' comparison of sph_slab_step_finish_local).
    if getattr(grp, "_primed", False) and not reported:
        sp.param_pause = 0
        grp.DispatchCompute()                                # (a grid that moved last: primed again here)
        k = int(rng.integers(0, world))
        kind = rng.choice(["kick", "member"])
        if kind == "kick":
            grp.sims[k].engine.apply_wave_impulse(400.0 * float(sp.param_h), 2.0, 0.5, (0.0, 0.3, 1.0), -1e9, 1e9)   # (strong against h: a kick that holds whole faces)
        else:
            q = type(sp).from_buffer_copy(sp)
            q.param_gasConstant = float(sp.param_gasConstant) * 1.5
            grp.sims[k].engine._p = q
        for s in grp.sims:
            s.engine.step_begin()
        for r, s in enumerate(grp.sims):
            lo = grp.sims[r - 1].engine if r > 0 else None
            hi = grp.sims[r + 1].engine if r < world - 1 else None
            touches = abs(r - k) <= 1
            if touches:
                with pytest.raises(pkg.SphError, match="one rank only|members"):
                    s.engine.step_finish_local(lo, hi)
            else:
                s.engine.step_finish_local(lo, hi)
    for s in grp.sims:
        s.engine.close()
