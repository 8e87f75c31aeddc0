"""Committed golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py with the
oracle on the reference's fixture inputs, incl. the reference's own plate_int_data.npz)."""
import os

import numpy as np
import pytest

from tests.golden import make_golden as mg

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KEYS = ["K", "C0", "C1", "C2", "H"]


def _rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


@pytest.mark.parametrize("name", list(mg.CASES))
def test_oracle_reproduces_golden(oracle_lib, name):
    from oracle.oracle_py import Oracle
    g = np.load(os.path.join(HERE, name + ".npz"))
    A, h, u = mg.state(mg.CASES[name]())
    O = Oracle(A, thickness=h, u=u)
    vals = O.assemble()
    assert _rel(O.residual(), g["R"]) < 1e-12
    for w, k in enumerate(KEYS):
        assert _rel(vals[w], g[k]) < 1e-12
    assert abs(O.functionals()["Wint"] - g["Wint"]) < 1e-12 * abs(g["Wint"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(mg.CASES))
def test_hip_matches_golden(name):
    from goldfish_amd import _lib
    g = np.load(os.path.join(HERE, name + ".npz"))
    A, h, u = mg.state(mg.CASES[name]())
    D = _lib.DeviceModel(A)
    D.set_thickness(h)
    D.set_u(u)
    D.assemble()
    assert _rel(D.residual(), g["R"]) < 1e-10
    for w, k in enumerate(KEYS):
        assert _rel(D.values(w), g[k]) < 1e-10
    F = D.functionals()
    assert abs(F["Wint"] - g["Wint"]) < 1e-11 * abs(g["Wint"]) and _rel(F["dWdu"], g["dWdu"]) < 1e-10
    D.close()


def test_iges_reader_on_the_reference_plate_geometry():
    """N2: the minimal IGES-128 reader on the reference's own CAD file
    (demos_csdl_alpha/thickness_opt/geometry/plate_geometry.igs, committed as data): six cubic B-spline
    surfaces, identical (to the file's 9 digits) to the six-patch plate the parity fixtures are built on."""
    import os
    from goldfish_amd import geometry as G
    from goldfish_amd.utils.iges import read_iges_surfaces
    S = read_iges_surfaces(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_plate_geometry.igs"))
    spec = G.plate_6patch()
    assert len(S) == 6 == len(spec.patches)
    for s, p in zip(S, spec.patches):
        assert (s.p, s.q) == (p.p, p.q) == (3, 3) and s.control.shape == p.control.shape
        assert np.abs(s.control - p.control).max() < 1e-8
        for d in range(2):
            assert np.abs(s.knots[d] - p.knots[d]).max() < 1e-8
        assert np.abs(s.control[..., 3] - 1.0).max() == 0.0              # polynomial surfaces
