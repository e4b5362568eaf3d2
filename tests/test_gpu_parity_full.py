"""Parity of the HIP path with the CPU oracle where round 2 only compared the HIP path with itself or with its own
contract ("parity hardening"):

  (a) BASELINE.json configs[2] at FULL size (4 194 304 particles, 128^3) and one rank's share of configs[3] / configs[4]
      against the ORACLE, bit for bit, for a few substeps (the oracle takes about half a second per substep on the GPU
      box's host cores);
  (b) the HIP path against the LITERAL restatement of SPHFluid.comp (oracle contract 0: IEEE sqrt / division exactly where
      the shader has them) on BASELINE config 1 at 1 / 10 / 25 / 50 substeps, with the fp32 tolerances written here;
  (c) BASELINE.json's tolerance in its honest form: a scene that does NOT collapse (the reference's default 50 000-particle
      scene, settled into its pool), 100 substeps, per-particle density and pressure of the HIP path against the literal
      restatement.

What "parity" is measured against: bit equality holds against the engine's OWN arithmetic contract (oracle contract 1,
DESIGN.md section 3); against the literal shader arithmetic the HIP path differs by fp32 rounding (specified rsqrt,
factored sums), which (b) and (c) bound.  The reference itself holds no vectors for this path: parity is unpinned.
"""
import os

import numpy as np
import pytest

from conftest import assert_records_equal, to_oracle_params

pytestmark = pytest.mark.gpu

KERNELS = [("walk", 3), ("list", 2)]


def _engine(pkg, rec, sp, neighbor):
    f = pkg.SPHFluidGPU.from_particles(rec, sp)
    f.set_option(pkg.SPH_OPT_NEIGHBOR_KERNEL, neighbor)
    return f


def _rel(a, b, floor=1e-30):
    a = a.astype(np.float64)
    return np.abs(a - b.astype(np.float64)) / np.maximum(np.abs(a), floor)


@pytest.mark.parametrize("name,neighbor", KERNELS)
def test_config3_full_size_against_the_oracle(pkg, oracle, name, neighbor):
    """BASELINE.json configs[2], the configuration the metric is quoted on: all 4 194 304 records, 3 substeps, every byte."""
    syn = pkg.synthetic
    cfg = syn.CONFIGS[3]
    rec, _ = syn.make_particles(cfg)
    sp = pkg.default_params(**syn.params_fields(cfg))
    f = _engine(pkg, rec, sp, neighbor)
    f.DispatchN(3)
    got = f.download()
    f.close()
    want = oracle.substep(rec, to_oracle_params(oracle, sp), steps=3)
    assert_records_equal(got, want, f"{name}: config 3 (4M / 128^3) after 3 substeps")
    assert want["pressure"].max() > 0


@pytest.mark.parametrize("which", ["configs[3] share", "configs[4] share"])
def test_one_ranks_share_of_the_slab_configs_against_the_oracle(pkg, oracle, which):
    """One rank's share of BASELINE.json configs[3] (4 194 304 particles of the 16 M run, spacing 0.85 h) and configs[4]
    (8 388 608 particles, spacing 0.775 h), both in a 256 x 256 x 64-cell slab, as a single-domain run: 2 substeps, every byte."""
    syn = pkg.synthetic
    if which.startswith("configs[3]"):
        cfg = syn.BenchConfig(4, "16M/256^3 share", 4194304, (256, 256, 64), 0.85, 1)
    else:
        cfg = syn.weak_config(1)
    rec, _ = syn.make_particles(cfg)
    sp = pkg.default_params(**syn.params_fields(cfg))
    g = pkg.compute_grid_extents(sp)
    assert tuple(g.dims) == (256, 256, 64) and len(rec) == cfg.n
    f = _engine(pkg, rec, sp, 3)
    f.DispatchN(2)
    got = f.download()
    f.close()
    want = oracle.substep(rec, to_oracle_params(oracle, sp), steps=2)
    assert_records_equal(got, want, f"{which} after 2 substeps")


def test_hip_against_the_literal_shader_arithmetic(pkg, oracle):
    """HIP (engine contract) vs oracle contract 0 (literal SPHFluid.comp arithmetic, canonical order) on BASELINE config 1.
    Tolerances (max relative density difference): 1e-6 after 1 substep, 5e-6 after 10, 2e-5 after 25, 1e-3 after 50 (measured
    round 3: 3.6e-7 / 1.0e-6 / 2.7e-6 / 6.7e-5; two legal summation orders of the literal arithmetic differ by as much,
    tests/test_oracle_contract.py).  By 100 substeps this COLLAPSING scene has amplified any rounding difference to O(0.1):
    the 100-substep tolerance of BASELINE.json is asserted on a scene that does not collapse, below."""
    syn = pkg.synthetic
    cfg = syn.CONFIGS[1]
    rec, _ = syn.make_particles(cfg)
    sp = pkg.default_params(**syn.params_fields(cfg))
    op = to_oracle_params(oracle, sp)
    f = _engine(pkg, rec, sp, 3)
    lit = rec.copy()
    done = 0
    rows = []
    try:
        oracle.set_contract(0)
        for upto, tol in ((1, 1e-6), (10, 5e-6), (25, 2e-5), (50, 1e-3)):
            f.DispatchN(upto - done)
            lit = oracle.substep(lit, op, steps=upto - done)
            done = upto
            got = f.download()
            rd = float(_rel(lit["density"], got["density"]).max())
            rp = float(np.abs(lit["pos"] - got["pos"]).max())
            rows.append((upto, rd, rp, tol))
    finally:
        oracle.set_contract(1)
        f.close()
    for r in rows:
        print("substeps %3d: HIP vs literal restatement: max rel density diff %.2e, max |pos| diff %.1e (tolerance %.0e)" % r)
    for upto, rd, _, tol in rows:
        assert rd <= tol, (upto, rd, tol)


def test_settled_pool_100_substeps_within_1e4_of_the_literal_arithmetic(pkg, oracle):
    """north_star's tolerance on a scene that stays put: the reference's default scene (50 000 requested particles, box half 7,
    h = 0.28: SPHFluid3D.h:94-113), run 2 000 substeps on the HIP path until the column has settled into its pool -- which must be, bit for
    bit, the state the CPU oracle reaches in 2 000 substeps (tests/golden/settled_pool.npz) -- then 100 substeps on the HIP path against the
    LITERAL restatement (oracle contract 0) from the same state.
    Asserted: per-particle density within 1e-4 relative after 100 substeps (measured 2.9e-5); pressure within 1e-4 relative after 25 substeps
    (5.9e-5) and within 3e-4 after 100 (1.75e-4).  Whose 1.75e-4 that is: tests/test_oracle_contract.py::test_settled_pool_whose_1_7e_4_it_is
    measures, on this very state, 1.73e-4 between the literal restatement in canonical order and in the shader's own traversal order -- two legal
    orders of the REFERENCE (its list order is arbitrary).  1e-4 in pressure is not attainable by the reference against itself here; the engine's
    contract is at 1.01 x the reference's own spread, and that is asserted too (against the fixture's second-order end state)."""
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "settled_pool.npz"))
    sp = pkg.default_params()
    rec, mass = pkg.spawn_particles(sp, 50000, seed=5)
    sp.param_mass = mass
    assert np.float32(mass) == fx["mass"]
    op = to_oracle_params(oracle, sp)
    f = _engine(pkg, rec, sp, 3)
    f.DispatchN(2000)
    settled = f.download()
    assert_records_equal(settled, fx["settled"], "2 000 substeps of the HIP path against 2 000 substeps of the oracle (fixture)")
    assert float(np.abs(settled["vel"]).max()) < 60.0 and settled["density"].max() > 2000.0      # a pool, not a falling block
    lit = settled.copy()
    done = 0
    rows = []
    try:
        oracle.set_contract(0)
        for upto in (25, 100):
            f.DispatchN(upto - done)
            lit = oracle.substep(lit, op, steps=upto - done)
            done = upto
            got = f.download()
            rows.append((upto, float(_rel(lit["density"], got["density"]).max()),
                         float(_rel(lit["pressure"], got["pressure"])[lit["pressure"] > 0].max()),
                         float(np.abs(lit["pos"] - got["pos"]).max())))
    finally:
        oracle.set_contract(1)
    got = f.download()
    f.close()
    assert got[::16].tobytes() == fx["after100_contract1_every16"].tobytes()                    # the engine's own contract: bit for bit
    assert lit[::16].tobytes() == fx["after100_contract0_every16"].tobytes()
    for r in rows:
        print("settled pool +%3d substeps: HIP vs literal restatement: density %.2e  pressure %.2e  |pos| %.1e" % r)
    (_, d25, p25, x25), (_, d100, p100, x100) = rows
    assert d25 <= 1e-4 and p25 <= 1e-4
    assert d100 <= 1e-4 and p100 <= 3e-4 and x100 <= 2e-5
    # the reference against itself on the sampled records: literal arithmetic, canonical order vs the shader's own traversal order
    l0, l2 = fx["after100_contract0_every16"], fx["after100_contract2_every16"]
    pm = l0["pressure"] > 0
    own = float(_rel(l0["pressure"], l2["pressure"])[pm].max())
    hip = float(_rel(l0["pressure"], got[::16]["pressure"])[pm].max())
    print("  on every 16th record: the reference's two legal orders differ by %.2e in pressure, the HIP path from the literal one by %.2e" % (own, hip))
    assert hip <= 1.5 * own + 1e-5


def test_config2_counting_sort_and_linked_list_against_the_oracle(pkg, oracle):
    """BASELINE.json configs[1] at its own size (262 144 particles, 64^3 cells): the A/B of the two grid builds.  Counting sort
    (this engine's build, every SPH pass) against the oracle bit for bit; the linked-list build (BuildGrid.comp:21-37 as it stands:
    atomic arrival order, so sums are not reproducible) against the oracle within the tolerances of
    tests/test_gpu_parity.py::test_linked_list_variant_within_tolerance."""
    syn = pkg.synthetic
    cfg = syn.CONFIGS[2]
    rec, _ = syn.make_particles(cfg)
    assert len(rec) == 262144
    sp = pkg.default_params(**syn.params_fields(cfg))
    assert tuple(pkg.compute_grid_extents(sp).dims) == (64, 64, 64)
    op = to_oracle_params(oracle, sp)
    want1 = oracle.substep(rec, op)
    want3 = oracle.substep(want1, op, steps=2)
    for name, neighbor in KERNELS + [("slow", 1)]:
        f = _engine(pkg, rec, sp, neighbor)
        f.DispatchN(3)
        assert_records_equal(f.download(), want3, f"configs[1], counting sort + {name}, 3 substeps")
        f.close()
    f = _engine(pkg, want1, sp, 3)                           # (from a state with densities: the first substep after a reset has no pair terms)
    f.set_option(pkg.SPH_OPT_GRID_BUILD, 1)
    f.DispatchCompute()
    got, want = f.download(), oracle.substep(want1, op)
    err = _rel(want["density"], got["density"])
    assert err.max() < 1e-5, err.max()
    assert np.abs(got["pressure"] - want["pressure"]).max() <= sp.param_gasConstant * 1e-5 * want["density"].max()
    assert np.abs(got["pos"] - want["pos"]).max() < 1e-5 and np.abs(got["vel"] - want["vel"]).max() < 1e-2
    f.DispatchN(9)
    got, want = f.download(), oracle.substep(want, op, steps=9)
    err = _rel(want["density"], got["density"])
    print("configs[1], linked list, 10 substeps: max rel density err", err.max())
    assert err.max() < 1e-3
    f.close()
