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
