"""The engine's arithmetic contract (oracle contract 1: specified rsqrt, factored sums, re-associated pair terms --
DESIGN.md section 3.5) measured against the LITERAL restatement of SPHFluid.comp (contract 0: IEEE sqrt / division where
the shader has them) and against the reference's OWN irreproducibility (contract 2: the literal arithmetic in the shader's
traversal order dx -> dy -> dz with descending in-cell lists, one legal atomicExchange arrival order).

BASELINE.json configs[0] inputs (32 768 particles, 32^3 grid).  The deviation of contract 1 from the literal restatement
must be of the size of the deviation between two legal orders of the literal restatement; both grow chaotically with the
substep count (the collapse of this workload), which is why the HIP parity tests compare bits under ONE contract."""
import numpy as np
import pytest

from conftest import to_oracle_params


def _rel(a, b):
    d = np.abs(a.astype(np.float64) - b.astype(np.float64))
    return float((d / np.maximum(np.abs(a.astype(np.float64)), 1e-30)).max())


def test_rsqrt_accuracy(oracle):
    """sph_oracle_rsqrt: <= 2 ulp (GLSL's bound for inversesqrt) over two binades and a spread of magnitudes."""
    xs = np.concatenate([np.arange(0x3F800000, 0x40800000, 97, dtype=np.uint32).view(np.float32),
                         np.float32([1e-12, 3e-7, 0.0784, 12345.0, 7e11])])
    got = np.array([oracle.rsqrt(float(x)) for x in xs[::37]], np.float64)
    ref = 1.0 / np.sqrt(xs[::37].astype(np.float64))
    ulp = np.abs(got - ref) / np.spacing(ref.astype(np.float32)).astype(np.float64)
    assert ulp.max() <= 2.0, ulp.max()


def test_contract_stays_inside_the_references_own_spread(pkg, oracle):
    syn = pkg.synthetic
    cfg = syn.CONFIGS[1]
    rec, _ = syn.make_particles(cfg)
    op = to_oracle_params(oracle, pkg.default_params(**syn.params_fields(cfg)))
    st = {c: rec.copy() for c in (0, 1, 2)}
    done, rows = 0, []
    try:
        for upto in (1, 10, 25, 50):
            for c in (0, 1, 2):
                oracle.set_contract(c)
                st[c] = oracle.substep(st[c], op, steps=upto - done)
            done = upto
            rows.append((upto, _rel(st[0]["density"], st[1]["density"]), _rel(st[0]["density"], st[2]["density"]),
                         float(np.abs(st[0]["pos"] - st[1]["pos"]).max()), float(np.abs(st[0]["pos"] - st[2]["pos"]).max())))
    finally:
        oracle.set_contract(1)
    for r in rows:
        print("substeps %3d: max rel density diff  literal vs contract %.2e   literal vs shader-order literal %.2e   |pos| %.1e / %.1e" % r)
    assert rows[0][1] < 1e-6 and rows[1][1] < 5e-6 and rows[2][1] < 2e-5          # 1, 10, 25 substeps
    assert rows[3][1] < 1e-3 and rows[3][2] < 1e-3                                # 50 substeps: both pairs, same order of magnitude
    for _, d01, d02, _, _ in rows:
        assert d01 <= 10.0 * d02 + 1e-7                                            # never far outside the order-induced spread


def test_settled_pool_whose_1_7e_4_it_is(pkg, oracle):
    """north_star: "per-particle density / pressure within 1e-4 rel of the CPU reference after 100 substeps".  On the one scene where that is a
    meaningful question -- the reference's default 50 000-particle scene settled into its pool (tests/golden/settled_pool.npz, written by
    tests/golden/make_settled_pool.py) -- 100 substeps under the three oracle contracts from the same state:
        density   literal vs engine contract 2.9e-5    literal vs the SHADER'S OWN second legal order 2.2e-5      (both inside 1e-4)
        pressure  literal vs engine contract 1.75e-4   literal vs the shader's own second legal order 1.73e-4
    The reference does not hold 1e-4 in pressure against ITSELF on this scene (its atomicExchange list order is arbitrary, SURVEY 8a "semantics" 2):
    P = k (rho - rho0) magnifies a relative density difference by rho / (rho - rho0).  The engine's contract sits at 1.01 x the reference's own
    order-induced spread; that, not 1e-4, is what a HIP-vs-literal comparison can be held to (tests/test_gpu_parity_full.py asserts 3e-4)."""
    import os
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "settled_pool.npz"))
    settled = fx["settled"]
    sp = pkg.default_params()
    sp.param_mass = float(fx["mass"])
    op = to_oracle_params(oracle, sp)
    end = {}
    try:
        for c in (0, 1, 2):
            oracle.set_contract(c)
            end[c] = oracle.substep(settled, op, steps=100)
    finally:
        oracle.set_contract(1)
    for c in (0, 1, 2):                                      # the fixture's samples: bit for bit (regression pin of all three contracts)
        assert end[c][::16].tobytes() == fx[f"after100_contract{c}_every16"].tobytes(), c
    pm = end[0]["pressure"] > 0

    def rel(a, b, m=None):
        d = np.abs(a.astype(np.float64) - b.astype(np.float64)) / np.maximum(np.abs(a.astype(np.float64)), 1e-30)
        return float(d[m].max() if m is not None else d.max())
    d01, d02 = rel(end[0]["density"], end[1]["density"]), rel(end[0]["density"], end[2]["density"])
    p01, p02 = rel(end[0]["pressure"], end[1]["pressure"], pm), rel(end[0]["pressure"], end[2]["pressure"], pm)
    print("settled pool + 100 substeps: density  literal vs engine %.2e, literal vs shader-order literal %.2e; pressure %.2e, %.2e" % (d01, d02, p01, p02))
    assert d01 <= 1e-4 and d02 <= 1e-4                       # density: north_star's tolerance holds, for the engine's contract and for the reference against itself
    assert p02 >= 1.5e-4                                      # pressure: the reference's OWN two legal orders are 1.7e-4 apart ...
    assert p01 <= 1.1 * p02                                   # ... and the engine's contract is no further from the literal arithmetic than that
    assert np.allclose(fx["spread"][1, 1:5], [d01, d02, p01, p02], rtol=1e-12)
