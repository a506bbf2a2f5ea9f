"""The committed oracle fixtures (tests/golden/oracle_fixtures.npz) still are what the
oracle computes: every array re-derived on CPU, bit for bit.  A change of the oracle's arithmetic
shows up here before it can silently move the bar of the GPU parity tests."""
import importlib.util
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def load_generator():
    spec = importlib.util.spec_from_file_location("make_oracle_fixtures", os.path.join(HERE, "golden", "make_oracle_fixtures.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_fixtures_are_reproduced_bit_for_bit(oracle):
    gen = load_generator()
    fresh = gen.build()
    stored = np.load(os.path.join(HERE, "golden", "oracle_fixtures.npz"))
    assert sorted(stored.files) == sorted(fresh)
    for key in stored.files:
        assert stored[key].shape == fresh[key].shape, key
        assert np.array_equal(stored[key], fresh[key], equal_nan=True), key
    assert len(stored.files) >= 60 and all(stored[k].nbytes <= 256 * 1024 for k in stored.files)


def test_fixture_sanity():
    """the fixtures are not degenerate: unit-norm orthogonal store, evolved states differ from the start"""
    f = np.load(os.path.join(HERE, "golden", "oracle_fixtures.npz"))
    for name in ("harmonic_3pt", "coulomb_5pt", "cornell_7pt", "fullcornell_3pt"):
        l0, l1 = f[f"{name}/lower0"], f[f"{name}/lower1"]
        assert abs(np.sum(l0 * l0) - 1) < 1e-13 and abs(np.sum(l1 * l1) - 1) < 1e-13 and abs(np.sum(l0 * l1)) < 1e-13
        assert not np.array_equal(f[f"{name}/ground_5steps"], f[f"{name}/phi0"])
        ex = f[f"{name}/excited_wnum2_3steps"]
        assert abs(np.sum(ex * l0)) < 1e-12 and abs(np.sum(ex * l1)) < 1e-12     # projected out after every step
