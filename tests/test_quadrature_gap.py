"""How far is the tensor-Gauss assembly (this oracle, the HIP kernels) from what the reference integrates?

The reference assembles through FEniCS on tIGAr's extraction mesh: two triangles per knot span with a quadrature rule of declared degree
quad_deg = 3p (GOLDFISH/tests/test_tbeam.py:31), 2p (tests/test_slr.py:37) or 4p (demos_om/thickness_opt/plate/plate_const_th_opt_wint.py:139-144)
-- SURVEY.md section 7 "hard parts" / App. A.1 [ext-recall].  The oracle can integrate the shell terms with exactly that rule
(oracle_py.two_triangle_rule -> gfo_set_quadrature); the numbers below are the honest error bar on "matches FEniCS" that can be had
without FEniCS: same formulation, the reference's quadrature against ours, on the reference's own fixtures.  `python tests/test_quadrature_gap.py`
prints the table of DESIGN.md section 2."""
import os
import sys

import numpy as np
import pytest
import scipy.sparse.linalg as spl

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from goldfish_amd import geometry as G                     # noqa: E402
from goldfish_amd.model import arrays_from_spec            # noqa: E402

FIXTURES = {
    # name: (spec builder, quad_deg of the reference for this fixture)
    "T-beam, 2 patches, p = 3 (tests/test_tbeam.py, quad_deg 3p = 9)": (lambda: G.tbeam_2patch(10), 9),
    "Scordelis-Lo roof, 9 NURBS patches, p = 3 (tests/test_slr.py, quad_deg 2p = 6)": (lambda: G.scordelis_lo_9patch(6), 6),
    "plate, 6 patches of plate_geometry.igs, p = 3 (plate_const_th_opt_wint.py, quad_deg 4p = 12)": (lambda: G.plate_6patch(), 12),
}
NAMES = ["R", "K", "dR/dCP_0", "dR/dCP_1", "dR/dCP_2", "dR/dh"]


def gaps(spec, quad_deg, scale_u=1.0):
    """max-norm relative differences tensor Gauss vs two-triangle rule of R, K, dR/dCP_f, dR/dh at the linear solution (scaled), and of
    that solution itself."""
    from oracle.oracle_py import Oracle, two_triangle_rule
    A = arrays_from_spec(spec)
    h = np.full(A.total_cp, spec.h_th)
    out = {}
    sols = []
    for rule in (None, two_triangle_rule(quad_deg)):
        O = Oracle(A, thickness=h)
        O.set_quadrature(rule)
        K0 = O.csr(0, O.assemble(dRdCP=(), dRdh=False)[0]).tocsc()
        sols.append(spl.spsolve(K0, -O.residual()))
    u = scale_u * sols[0]
    vals = []
    for rule in (None, two_triangle_rule(quad_deg)):
        O = Oracle(A, thickness=h, u=u)
        O.set_quadrature(rule)
        m = O.assemble()
        # internal force only: the residual of an equilibrium state is a difference of large terms
        vals.append([O.functionals(apply_bcs=True)["dWdu"]] + [m[w] for w in range(5)])
    # scale: the shell terms alone (the penalty blocks are identical under both rules and would dominate max |K|)
    import copy
    bare = copy.copy(spec)
    bare.interfaces = []
    Ab = arrays_from_spec(bare)
    Ob = Oracle(Ab, thickness=h, u=u)
    mb = Ob.assemble()
    scale = [np.abs(vals[0][0]).max()] + [np.abs(mb[w]).max() for w in range(5)]
    for name, a, b, sc in zip(NAMES, vals[0], vals[1], scale):
        out[name] = float(np.abs(a - b).max() / sc)
    out["u (linear solve)"] = float(np.abs(sols[0] - sols[1]).max() / np.abs(sols[0]).max())
    return out


@pytest.mark.parametrize("name", list(FIXTURES))
def test_tensor_gauss_vs_reference_triangle_rule(oracle_lib, name):
    build, deg = FIXTURES[name]
    g = gaps(build(), deg)
    # Observed (table in DESIGN.md section 2): entries of the matrices differ by 2e-8 .. 3e-4 on the polynomial geometries -- there the
    # tensor rule is the exact one for the u = 0 forms (bi-degree 2p) and the degree-3p / 4p triangle rule is not -- and by up to 5e-3 on
    # the rational roof with its degree-2p rule; the solutions differ by 1e-8 resp. 2e-6.  Bounds = observed x ~5: a formulation change
    # in either rule's code path moves these by orders of magnitude.
    roof = "Scordelis" in name
    for k, v in g.items():
        bound = (1e-5 if roof else 5e-8) if k.startswith("u") else (2e-2 if roof else 2e-3)
        assert v < bound, (name, k, v)
    assert g["u (linear solve)"] > 0.0                       # the two rules are really different rules


def test_roof_known_answer_under_the_reference_rule(oracle_lib):
    """0.3006 (GOLDFISH/tests/test_slr.py:50) with the shell terms integrated by the reference's rule (degree 2p on two triangles)."""
    from oracle.oracle_py import Oracle, two_triangle_rule
    spec = G.scordelis_lo_9patch(6)
    A = arrays_from_spec(spec)
    O = Oracle(A, thickness=np.full(A.total_cp, spec.h_th))
    O.set_quadrature(two_triangle_rule(6))
    K = O.csr(0, O.assemble(dRdCP=(), dRdh=False)[0]).tocsc()
    O.set_u(spl.spsolve(K, -O.residual()))
    assert abs(abs(O.eval_point(3, (0.0, 0.5))[1][1]) - 0.3006) < 1e-3 * 0.3006


if __name__ == "__main__":
    print("| fixture | " + " | ".join(NAMES + ["u (linear solve)"]) + " |")
    print("|---|" + "---|" * (len(NAMES) + 1))
    for name, (build, deg) in FIXTURES.items():
        g = gaps(build(), deg)
        print("| %s | " % name + " | ".join("%.1e" % g[k] for k in NAMES + ["u (linear solve)"]) + " |")
