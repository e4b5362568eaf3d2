"""The C oracle against the independent brute-force numpy restatement (no cell lists).
Catches grid / stencil / ordering bugs a single implementation could hide.  The numpy
version emulates fmaf through float64 (one extra rounding in ~2^-29 of the ops), hence a
few-ulp tolerance instead of bit equality."""
import numpy as np
import pytest

from conftest import small_scene, to_oracle_params


def _check(a, b, tol=2e-6):
    for f in ("pos", "vel", "acc"):
        x, y = a[f][:, :3].astype(np.float64), b[f][:, :3].astype(np.float64)
        scale = np.maximum(np.abs(y).max(axis=0), 1e-3)
        assert np.max(np.abs(x - y) / scale) < tol, f
    for f in ("density", "pressure", "padA"):
        x, y = a[f].astype(np.float64), b[f].astype(np.float64)
        assert np.max(np.abs(x - y) / np.maximum(np.abs(y).max(), 1e-3)) < tol, f
    for f in ("padB", "isGhost", "isActive", "padC", "pad0"):
        assert np.array_equal(a[f], b[f])


@pytest.mark.parametrize("steps_before", [0, 1, 12])
def test_sph_pass_matches_bruteforce(pkg, oracle, steps_before):
    rec, sp = small_scene(pkg, n=1500, grid=14, seed=21)
    op = to_oracle_params(oracle, sp)
    P = oracle.substep(rec, op, steps=steps_before) if steps_before else rec
    if steps_before:
        assert P["density"].min() > 0 and P["pressure"].max() > 0
    _check(oracle.sph_pass(P, op), oracle.brute_force_sph_pass(P, op))


def test_sph_pass_matches_bruteforce_moving(pkg, oracle):
    """Fast particles (large sweep-3 displacement), a clump, ghosts and particles outside the grid."""
    rec, sp = small_scene(pkg, n=1200, grid=12, seed=22)
    op = to_oracle_params(oracle, sp)
    P = oracle.substep(rec, op, steps=3)
    rng = np.random.default_rng(4)
    P["vel"][:, :3] += rng.normal(0, 40, (len(P), 3)).astype(np.float32)
    P["pos"][:40, :3] = P["pos"][0, :3] + rng.normal(0, 0.03, (40, 3)).astype(np.float32)   # clump in ~1 cell
    P["pos"][40:60, 0] += 5.0                                                               # beyond the grid: clamped cells
    P["isGhost"][100:110] = 1
    P["isActive"][100:105] = 1
    P["isGhost"][110:115] = 2
    _check(oracle.sph_pass(P, op), oracle.brute_force_sph_pass(P, op), tol=5e-6)


def test_substep_is_pass_plus_obb(pkg, oracle):
    rec, sp = small_scene(pkg, n=600, grid=10, seed=23)
    op = to_oracle_params(oracle, sp)
    P = oracle.substep(rec, op, steps=2)
    a = oracle.substep(P, op)
    b = oracle.obb(oracle.sph_pass(P, op), op)
    assert a.tobytes() == b.tobytes()
