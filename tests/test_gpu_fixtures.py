"""The HIP path against the COMMITTED oracle fixtures (tests/golden/oracle_fixtures.npz) -- no
oracle library involved at run time: potentials, a/b, initial conditions, observables, ground and
excited-state evolve, normalise + Gram-Schmidt, for all three stencil orders."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
CASES = {"harmonic_3pt": "Harmonic", "coulomb_5pt": "Coulomb", "cornell_7pt": "SimpleCornell", "fullcornell_3pt": "FullCornell"}


@pytest.fixture(scope="module")
def fx():
    return np.load(os.path.join(HERE, "golden", "oracle_fixtures.npz"))


@pytest.fixture(scope="module")
def wa():
    import wafer_amd
    wafer_amd.load_library()
    return wafer_amd


def params(wa, fx, name, **kw):
    nx, ny, nz, ext, dn, dt, mass, sig = fx[f"{name}/params"]
    return wa.Params(int(nx), int(ny), int(nz), dn=dn, dt=dt, mass=mass, sig=sig, central_difference=int(ext), max_states=2, **kw)


def potsub_of(fx, name):
    kind, scalar = fx[f"{name}/potsub_kind_scalar"]
    return int(kind), float(scalar), (fx[f"{name}/potsub"] if f"{name}/potsub" in fx.files else None)


@pytest.mark.parametrize("name", list(CASES))
def test_builtin_potential_and_ics(wa, fx, name):
    with wa.Context(params(wa, fx, name)) as ctx:
        ctx.set_potential(CASES[name])
        v, a, b = (ctx.download_array(k) for k in ("v", "a", "b"))
        if name == "fullcornell_3pt":    # device sin / exp / log against glibc's
            assert np.allclose(v, fx[f"{name}/v"], rtol=1e-12, atol=0) and np.allclose(a, fx[f"{name}/a"], rtol=1e-12)
        else:
            assert np.array_equal(v, fx[f"{name}/v"]) and np.array_equal(a, fx[f"{name}/a"]) and np.array_equal(b, fx[f"{name}/b"])
        kind, scalar, arr = potsub_of(fx, name)
        assert ctx.potsub()[0] == kind and ctx.potsub()[1] == pytest.approx(scalar, rel=1e-15)
        if arr is not None:
            assert np.allclose(ctx.download_array("potsub"), arr, rtol=1e-12, equal_nan=True)
        for ic in ("Boolean", "Constant"):
            ctx.set_initial_condition(ic)
            assert np.array_equal(ctx.download_phi(), fx[f"{name}/ic_{ic}"])


@pytest.mark.parametrize("variant", [-1, 0, 1])
@pytest.mark.parametrize("name", list(CASES))
def test_evolve_and_sums(wa, fx, name, variant):
    kind, scalar, arr = potsub_of(fx, name)
    with wa.Context(params(wa, fx, name)) as ctx:
        ctx.set_stencil_variant(variant)
        ctx.set_potential_host(fx[f"{name}/v"], kind, scalar, arr)      # identical inputs: V from the fixture
        assert np.array_equal(ctx.download_array("a"), fx[f"{name}/a"]) and np.array_equal(ctx.download_array("b"), fx[f"{name}/b"])
        ctx.upload_phi(fx[f"{name}/phi0"])
        assert ctx.norm2() == pytest.approx(float(fx[f"{name}/norm2_0"][0]), rel=1e-12)
        obs = ctx.observables()
        for key, want in zip(("energy", "norm2", "v_infinity", "r2"), fx[f"{name}/observables0"]):
            if np.isnan(want):
                assert np.isnan(obs[key])
            else:
                assert obs[key] == pytest.approx(want, rel=1e-12, abs=1e-300), key
        ctx.evolve(0, 5)
        assert np.array_equal(ctx.download_phi(), fx[f"{name}/ground_5steps"])          # bit for bit
        # normalise + modified Gram-Schmidt against the stored pair
        for j in range(2):
            ctx.upload_phi(fx[f"{name}/lower{j}"])
            ctx.push_state()
        ctx.upload_phi(fx[f"{name}/phi0"])
        ctx.normalise(ctx.norm2())
        ctx.orthogonalise(2)
        assert np.allclose(ctx.download_phi(), fx[f"{name}/normalised_orthogonalised"], rtol=0, atol=1e-14)
        ctx.upload_phi(fx[f"{name}/phi0"])
        ctx.evolve(2, 3)                                                                # renormalise + project every step
        want = fx[f"{name}/excited_wnum2_3steps"]
        assert np.allclose(ctx.download_phi(), want, rtol=0, atol=1e-13 * max(1.0, float(np.max(np.abs(want)))))
