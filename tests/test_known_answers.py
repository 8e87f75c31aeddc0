"""Known answers and invariants that pin the formulation of the oracle physically:
Scordelis-Lo roof = 0.3006 (GOLDFISH/tests/test_slr.py:42-50, `QoI_ref`), rigid-body
invariance of the geometrically exact KL strains and of the penalty energy, symmetry."""
import numpy as np
import pytest
import scipy.sparse.linalg as spl

from goldfish_amd import geometry as G
from goldfish_amd.model import arrays_from_spec
from oracle.oracle_py import Oracle

QOI_REF = 0.3006


def _slr_disp(spec, patch, xi):
    A = arrays_from_spec(spec)
    O = Oracle(A, thickness=np.full(A.total_cp, spec.h_th))
    R = O.residual()
    K = O.csr(0, O.assemble(dRdCP=(), dRdh=False)[0]).tocsc()
    O.set_u(spl.spsolve(K, -R))
    return O.eval_point(patch, xi)[1]


def test_scordelis_lo_single_patch(oracle_lib):
    """0.3006 is quoted to four digits (+-1.7e-4 relative); the 16 x 16 bicubic patch gives 0.300584."""
    U = _slr_disp(G.scordelis_lo_single(16), 0, (0.0, 0.5))
    assert abs(abs(U[1]) - QOI_REF) / QOI_REF < 3e-4


def test_scordelis_lo_nine_nonmatching_patches(oracle_lib):
    """The reference's nine non-matching NURBS patches with penalty coupling (1e3): 0.300531 at 5 .. 8 spans per side."""
    U = _slr_disp(G.scordelis_lo_9patch(6), 3, (0.0, 0.5))
    assert abs(abs(U[1]) - QOI_REF) / QOI_REF < 5e-4


@pytest.mark.parametrize("penalty", [1.0e3, 1.0e4])
def test_nine_patch_roof_converges_monotonically_to_the_reference_value(oracle_lib, penalty):
    """Mesh refinement of the 9-patch roof for two penalty coefficients: |d - 0.3006| falls monotonically and ends within the
    rounding of the quoted value.  This constrains the penalty conventions PENGoLINS decides and this oracle had to fix (side-A
    tangent, trapezoid vertex weights, minimum rule, mean edge length): a wrong scaling of alpha_d / alpha_r with h_e or a wrong
    quadrature weight shows up as a penalty-dependent limit.  (Observed: 1e3 -> 0.29853, 0.30036, 0.30053, 0.30060;
    1e4 -> 0.29660, 0.30014, 0.30029, 0.30050, 0.30059.)"""
    errs = []
    for nel in (3, 4, 6, 8) + ((12,) if penalty > 1.0e3 else ()):
        spec = G.scordelis_lo_9patch(nel)
        spec.penalty_coefficient = penalty
        errs.append(abs(abs(_slr_disp(spec, 3, (0.0, 0.5))[1]) - QOI_REF) / QOI_REF)
    assert all(b < a for a, b in zip(errs, errs[1:])), errs
    assert errs[-1] < 2e-4, errs


def test_rigid_body_motion_gives_zero_internal_force(oracle_lib):
    spec = G.scordelis_lo_9patch(3, nels=[2, 1, 2, 3, 2, 3, 2, 1, 2])
    spec.body_force = [[0, 0, 0]] * 9
    for p in spec.patches:
        p.zero_dofs.clear()
    A = arrays_from_spec(spec)
    O = Oracle(A, thickness=np.full(A.total_cp, spec.h_th))
    c = np.stack(A.cp_hom, 1)
    th = 0.7
    Q = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1.0]]) @ \
        np.array([[1, 0, 0], [0, np.cos(0.3), -np.sin(0.3)], [0, np.sin(0.3), np.cos(0.3)]])
    tr = np.array([1.0, -2.0, 0.5])
    U = c @ Q.T - c + A.weights[:, None] * tr          # homogeneous dofs: U_a = w_a * u_a
    O.set_u(U.ravel())
    F = O.functionals()
    scale = np.abs(O.csr(0, O.assemble(dRdCP=(), dRdh=False)[0]).diagonal()).max() * 25.0
    assert F["Wint"] < 1e-18 * scale and F["Wpen"] < 1e-18 * scale
    assert np.abs(O.residual()).max() < 1e-12 * scale


def test_tangent_symmetric_and_positive_at_rest(oracle_lib):
    spec = G.tbeam_2patch(4)
    A = arrays_from_spec(spec)
    O = Oracle(A, thickness=np.full(A.total_cp, spec.h_th))
    K = O.csr(0, O.assemble(dRdCP=(), dRdh=False)[0])
    assert abs(K - K.T).max() < 1e-12 * abs(K).max()
    assert np.linalg.eigvalsh(K.toarray()).min() > 0
