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
