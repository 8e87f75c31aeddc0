"""Shape optimisation of an arch under a snow load -- the problem of the reference's
demos_om/shape_opt/arch/arch_shape_opt_wint.py (four non-matching patches, span 10, pinned ends, load per unit
projected area ``-load * cos(beta) e_z`` :294-301, design = z-coordinates of a quadratic FFD block with 4 x 1 x 1
elements :237-238,318-327, aligned along y, bottom layer pinned, minimise the internal energy; the demo prints
"Maximum F2 (reference: 5.4779)" :419-420).  5.4779 = 0.54779 L is the ANALYTICAL optimum: the funicular shape of a load
per unit projected length is the parabola, and among parabolas int N^2/(2EA) ds is smallest at rise/span = 0.547789.

Driven through the operations of goldfish_amd (reduced space):

    state      R(u; CP) = 0                                    DispImOpeartion.solve_nonlinear
    objective  W(u, CP)                                        IntEnergyExOperation
    adjoint    K^T lam = dW/du                                 DispImOpeartion.solve_linear_rev
    gradient   D^T A^T [dW/dCP_2 - (dR/dCP_2)^T lam]           D = shopt_dcpsurf_fedcpffd, A = shopt_dcpaligndcpffd[0]

The reference's IGES geometry is not redistributable; the start is a parabolic arch of rise 3 (exact in the cubic patches).
Usage: python examples/arch_shape_opt.py
"""
import os
import sys

import numpy as np
from scipy.optimize import minimize

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from goldfish_amd.model import Interface                                  # noqa: E402
from goldfish_amd.nonmatching_opt import NonMatchingOptFFD, SVKResidual   # noqa: E402
from goldfish_amd.operations.disp_imop import DispImOpeartion            # noqa: E402
from goldfish_amd.operations.int_energy_exop import IntEnergyExOperation  # noqa: E402
from goldfish_amd.splines import NURBSPatch, open_uniform_knots           # noqa: E402
from goldfish_amd.utils.ffd_utils import create_3D_block                  # noqa: E402

SPAN, WIDTH, ANALYTIC_RISE = 10.0, 1.0, 5.477893528918274


def _poly2_coeffs(knots, p, a, b, c):
    """B-spline coefficients (degree p >= 2) of a + b u + c u^2: blossom at p consecutive knots."""
    n = len(knots) - p - 1
    out = np.zeros(n)
    for i in range(n):
        t = knots[i + 1:i + p + 1]
        pairs = (t.sum() ** 2 - (t * t).sum()) / 2.0
        out[i] = a + b * t.mean() + c * pairs / (p * (p - 1) / 2.0)
    return out


def parabola_patch(x0, x1, rise, nel_u, nel_v, p=3):
    """Exact piece x in [x0, x1] of the arch z = 4 rise x (SPAN - x) / SPAN^2, y in [0, WIDTH]."""
    ku, kv = open_uniform_knots(nel_u, p), open_uniform_knots(nel_v, p)
    dx, k = x1 - x0, 4.0 * rise / SPAN ** 2
    xs = _poly2_coeffs(ku, p, x0, dx, 0.0)
    zs = _poly2_coeffs(ku, p, k * (x0 * SPAN - x0 * x0), k * (SPAN * dx - 2 * x0 * dx), -k * dx * dx)
    ys = _poly2_coeffs(kv, p, 0.0, WIDTH, 0.0)
    ctrl = np.ones((xs.size, ys.size, 4))
    ctrl[:, :, 0], ctrl[:, :, 1], ctrl[:, :, 2] = xs[:, None], ys[None, :], zs[:, None]
    return NURBSPatch((p, p), (ku, kv), ctrl)


def build(rise0=3.0, nel_u=(5, 6, 5, 6), nel_v=(2, 3, 2, 3), p=3, mortar_nel=8, device=0):
    xs = np.linspace(0.0, SPAN, 5)
    patches = [parabola_patch(xs[k], xs[k + 1], rise0, nel_u[k], nel_v[k], p) for k in range(4)]
    for f in range(3):                                                    # pinned supports (one layer of control points)
        patches[0].add_zero_dofs(f, patches[0].get_side_dofs(0, 0, 1))
        patches[3].add_zero_dofs(f, patches[3].get_side_dofs(0, 1, 1))
    nm = NonMatchingOptFFD(patches, 1.0e12, 0.01, 0.0, device=device)
    nm.create_mortar_meshes([mortar_nel] * 3)
    ends_a, ends_b = [[1.0, 0.0], [1.0, 1.0]], [[0.0, 0.0], [0.0, 1.0]]
    nm.mortar_meshes_setup([[k, k + 1] for k in range(3)], [[ends_a, ends_b]] * 3, 1.0e3)
    nm.set_residuals([SVKResidual(body_force=(0.0, 0.0, -1.0), projected=(0.0, 0.0, 1.0))] * 4)
    nm.set_shopt_surf_inds_FFD([2], [0, 1, 2, 3])
    lims = [list(l) for l in nm.cpsurf_des_lims]
    lims[2][1] += 0.2 * (lims[2][1] - lims[2][0])
    blk = create_3D_block([4, 1, 1], 2, lims)
    nm.set_shopt_FFD(blk.knots, blk.control)
    nm.set_shopt_align_CPFFD(align_dir=[[1]])
    nm.set_shopt_pin_CPFFD(pin_dir0=[2], pin_side0=[[0]], pin_dir1=[1], pin_side1=[[0]])
    nm.set_shopt_regu_CPFFD()
    return nm


class ReducedShapeProblem:

    def __init__(self, nm, newton_rtol=1e-3):
        # rtol = the reference's default (DispStatesComp.init_parameters, nonlinear_solver_rtol=1e-3).  With E h / |f| = 1e10 the response is
        # linear to 1e-9 and the residual has an evaluation floor of ~4e-5 |R_0| (strain = difference of two metrics of size 6, times E h):
        # the first Newton step from u = 0 -- where the strain is exactly zero -- IS the accurate state, further steps only add the
        # floor's noise to it (profiles/r03_newton_history.txt).
        self.nm = nm
        self.disp, self.wint = DispImOpeartion(nm), IntEnergyExOperation(nm)
        self.D = nm.shopt_dcpsurf_fedcpffd.tocsr()                       # surface control points <- FFD control points
        self.A = nm.shopt_dcpaligndcpffd[0].tocsr()                      # FFD control points <- design dofs
        self.d0 = np.asarray(nm.shopt_init_cpffd_design[0], float).copy()
        self.rtol, self._d, self.n_state_solves = newton_rtol, None, 0

    def _solve(self, d):
        d = np.asarray(d, float)
        if self._d is None or not np.array_equal(d, self._d):
            self.nm.update_CPIGA(self.D @ (self.A @ d), 2)
            self.nm.update_uIGA(self.disp.solve_nonlinear(max_it=30, rtol=self.rtol))
            self._d = d.copy()
            self.n_state_solves += 1

    def objective(self, d):
        self._solve(d)
        return self.wint.Wint()

    def gradient(self, d):
        self._solve(d)
        self.disp.linearize()
        lam = self.disp.solve_linear_rev(self.wint.dWintduIGA(apply_bcs=True), np.zeros(self.nm.vec_iga_dof))
        g = [np.zeros(self.D.shape[0])]
        self.disp.apply_linear_rev(g, None, lam)                          # g[0] = (dR/dCP_2)^T lam
        return self.A.T @ (self.D.T @ (self.wint.dWintdCPIGA(2) - g[0]))

    def crown_height(self):
        P = self.nm.splines[1]
        return float(type(P)((P.p, P.q), P.knots, self._control(1)).eval((1.0, 0.5))[2])

    def _control(self, s):
        P = self.nm.splines[s]
        ctrl = P.control.copy()
        sl = slice(int(self.nm.cp_off[s]), int(self.nm.cp_off[s + 1]))
        for f in range(3):
            ctrl[:, :, f] = self.nm.cp_iga[f][sl].reshape(P.n_v, P.n_u).T
        return ctrl


def run(maxiter=200, verbose=True, **kw):
    nm = build(**kw)
    prob = ReducedShapeProblem(nm)
    d0 = prob.d0
    pin = nm.shopt_dcppindcpffd[0].tocsr()
    regu = nm.shopt_dcpregudcpffd[0].tocsr()
    w0 = prob.objective(d0)
    h0 = prob.crown_height()
    s = 1.0 / w0
    cons = [dict(type="eq", fun=lambda d: pin @ d - nm.shopt_pin_vals[0], jac=lambda d: pin.toarray()),
            dict(type="ineq", fun=lambda d: regu @ d - 1.0e-1, jac=lambda d: regu.toarray())]      # arch demo :136-138
    res = minimize(lambda d: s * prob.objective(d), d0, jac=lambda d: s * prob.gradient(d), method="SLSQP",
                   bounds=[(-1.0e-3, 12.0)] * d0.size, constraints=cons, options=dict(maxiter=maxiter, ftol=1e-14, disp=False))
    w1 = prob.objective(res.x)
    h1 = prob.crown_height()
    if verbose:
        print("internal energy %.6e -> %.6e, %d iterations, %d state solves" % (w0, w1, res.nit, prob.n_state_solves))
        print("crown height %.4f -> %.4f   (analytical optimum %.4f; the reference's demo prints 5.4779)" % (h0, h1, ANALYTIC_RISE))
    return dict(d=res.x, w0=w0, w1=w1, h0=h0, h1=h1, problem=prob, result=res)


if __name__ == "__main__":
    run()
