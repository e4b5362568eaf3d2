"""River / stream mode on the GPU: DispatchCompute with step 5 (k_river = TerrainConstraints + ChannelConstraint +
StreamEmit) against the oracle, bit for bit, through the C-ABI."""
import ctypes as C
import importlib

import numpy as np
import pytest

from conftest import PKG_NAME, assert_records_equal, to_oracle_params

pytestmark = pytest.mark.gpu


def _scene(pkg, oracle, n=6000, seed=5):
    sp = pkg.default_params()
    river, heights = pkg.generate_river_terrain(sp, seed)                    # also sets gravity (0, -120, 0)
    river.riverMode = 1
    P, mass = pkg.spawn_river_particles(sp, river, heights, n, 11)
    sp.param_mass = mass
    rng = np.random.default_rng(3)
    k = len(P) // 20
    idx = rng.permutation(len(P))
    P["pos"][idx[:k], 1] -= rng.uniform(0.3, 1.5, k).astype(np.float32)     # under the terrain
    P["vel"][idx[:k], :3] = rng.normal(0, 3, (k, 3)).astype(np.float32)
    P["pos"][idx[k:2 * k], 0] += rng.choice([-1.0, 1.0], k).astype(np.float32) * (river.riverChannelWidth + 0.5)   # outside the channel
    P["vel"][idx[k:2 * k], 0] = rng.normal(0, 2, k).astype(np.float32)
    P["pos"][idx[2 * k:3 * k], 2] = river.riverSinkZMax - rng.uniform(0.0, 0.02, k).astype(np.float32)   # about to leave downstream
    P["vel"][idx[2 * k:3 * k], 2] = 6.0
    P["pos"][idx[3 * k:3 * k + 10], 1] = river.riverSinkY - 0.5              # below the sink ...
    P["pos"][idx[3 * k:3 * k + 10], 0] = 7.5                                 # ... outside the terrain's footprint
    P["isGhost"][idx[3 * k + 10:3 * k + 20]] = 1                             # ghosts: skipped by all three passes
    P["isActive"][idx[3 * k + 10:3 * k + 15]] = 1
    P["vel"][:, 3] = rng.normal(0, 1, len(P)).astype(np.float32)             # the record's vel.w: kept, except by the emit
    op = to_oracle_params(oracle, sp)
    orv = oracle.ORiver.from_buffer_copy(bytes(river))
    return P, sp, op, river, orv, heights


@pytest.mark.parametrize("aos", [0, 1])
@pytest.mark.parametrize("neighbor", [3, 2, 1])
def test_river_dispatch_matches_oracle(pkg, oracle, neighbor, aos):
    P, sp, op, river, orv, heights = _scene(pkg, oracle)
    f = pkg.SPHFluidGPU.from_particles(P, sp)
    f.set_option(pkg.SPH_OPT_NEIGHBOR_KERNEL, neighbor)
    f.set_option(pkg.SPH_OPT_AOS_MODE, aos)
    f.set_river(river, heights)
    want = P
    for steps in (1, 9, 50):
        for _ in range(steps):
            f.DispatchCompute()
        want = oracle.substep_river(want, op, orv, heights, steps=steps)
        assert_records_equal(f.download(), want, f"river mode after +{steps} substeps (pass {neighbor}, aos mode {aos})")
        if steps == 1:   # StreamEmit ran in the very first substep (particles placed at the sink); it zeroes the record's vel.w
            recycled = (want["density"] == np.float32(sp.param_restDensity)) & (want["pressure"] == 0.0) & (want["isGhost"] == 0)
            assert recycled.sum() >= 10 and np.all(want["vel"][recycled, 3] == 0.0) and np.any(want["vel"][~recycled, 3] != 0.0)
    f.close()


def test_river_off_without_heightfield_and_fountain_skipped(pkg, oracle):
    P, sp, op, river, orv, heights = _scene(pkg, oracle, n=3000)
    f = pkg.SPHFluidGPU.from_particles(P, sp)
    f.riverMode = 1                                                          # no terrainHeights yet: step 5 does not run (:512) ...
    f.fountainMode = 1                                                       # ... and neither does the fountain (:519 `fountainMode && !riverMode`)
    f.DispatchCompute()
    assert_records_equal(f.download(), oracle.substep(P, op), "riverMode without a heightfield")
    assert f.fountainSeed == 0
    f.set_river(river, heights)
    f.DispatchN(3)
    want = oracle.substep_river(oracle.substep(P, op), op, orv, heights, steps=3)
    assert_records_equal(f.download(), want, "river + fountain flags: river only")
    f.riverMode = 0                                                          # back to the fountain
    f.DispatchCompute()
    fo = oracle.default_fountain(mode=1)
    assert_records_equal(f.download(), oracle.substep(want, op, fountain=fo), "fountain after riverMode = false")
    assert f.fountainSeed == 1
    f.close()


def test_reset_in_river_mode_spawns_along_the_channel(pkg, oracle):
    sp = pkg.default_params(param_boxHalf=(6.0, 5.0, 8.0))
    f = pkg.SPHFluidGPU(4000, params=sp, seed=2)
    f.GenerateRiverTerrain(21)
    assert (f.param_gravityY, f.param_gravityZ) == (-120.0, 0.0)
    f.riverMode = 1
    f.ResetSimulation(seed=8)
    op = to_oracle_params(oracle, f.params)
    orv, oh = oracle.river_terrain(op, 21)
    orv.riverMode = 1
    want, mass = oracle.river_spawn(op, orv, oh, 4000, 8)
    assert f.GetNumFluids() == 4000 and np.float32(f.param_mass) == np.float32(mass)
    assert_records_equal(f.download(), want, "ResetSimulation in river mode")
    assert f.terrainHeights.tobytes() == oh.tobytes()
    op.mass = mass
    f.DispatchN(5)
    assert_records_equal(f.download(), oracle.substep_river(want, op, orv, oh, steps=5), "river mode after reset + 5 substeps")
    f.close()


def test_river_mode_refused_on_a_slab_engine(pkg, oracle):
    halo = importlib.import_module(PKG_NAME + ".halo")
    P, sp, op, river, orv, heights = _scene(pkg, oracle, n=2000)
    g = pkg.compute_grid_extents(sp)
    eng = halo.HipSlabEngine(P, np.arange(len(P), dtype=np.uint32), sp, 0, g.dims[2], False, False, capacity=len(P) + 1024)
    L = pkg.load_library()
    assert L.sph_set_river(eng._h, C.byref(river), heights.ctypes.data_as(C.c_void_p)) == 0
    with pytest.raises(pkg.SphError, match="riverMode on a z-slab"):
        eng.dispatch()
    eng.close()
