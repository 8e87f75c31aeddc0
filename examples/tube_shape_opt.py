"""Shape optimisation of a tube under internal pressure -- the load case of the reference's
demos_om/shape_opt/tube/tube_shape_opt_wint.py (E = 1e12, nu = 0, h = 0.01, follower pressure 1 along
sqrt(det a / det A) a2 :258-262, 303-324; design = x and y coordinates (opt_field [0, 1]) of a cubic FFD block aligned along the tube
axis :237-241, 330-341; objective = internal energy).  The reference's IGES start geometry is not redistributable; here the
start is a closed ring whose radius varies as 1 + amp (1 - cos 4 theta) / 2 (four non-matching NURBS patches, penalty coupling).
A non-circular ring carries the pressure in bending -- orders of magnitude more strain energy than the hoop membrane state of the
circle -- so the optimiser rounds the cross-section; the FFD control points on the faces of the block are pinned (the size of the
tube is fixed: W ~ r^3 would otherwise shrink it).

The follower pressure is shape dependent: it enters dR/dCP, and its load stiffness makes K non-symmetric (the adjoint solve is a
true K^T solve).  Driven through the operations of goldfish_amd (reduced space):

    state      R(u; CP) = 0                                    DispImOpeartion.solve_nonlinear
    objective  W(u, CP)                                        IntEnergyExOperation
    adjoint    K^T lam = dW/du                                 DispImOpeartion.solve_linear_rev
    gradient   sum_f A_f^T D^T [dW/dCP_f - (dR/dCP_f)^T lam]   D = shopt_dcpsurf_fedcpffd, A_f = shopt_dcpaligndcpffd[f]

Usage: python examples/tube_shape_opt.py
"""
import os
import sys

import numpy as np
from scipy.optimize import minimize

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from goldfish_amd import geometry as G                                    # noqa: E402
from goldfish_amd.nonmatching_opt import NonMatchingOptFFD                # noqa: E402
from goldfish_amd.operations.disp_imop import DispImOpeartion            # noqa: E402
from goldfish_amd.operations.int_energy_exop import IntEnergyExOperation  # noqa: E402
from goldfish_amd.utils.ffd_utils import create_3D_block                  # noqa: E402


def build(amp=0.06, p=3, nels=((5, 2), (6, 3), (4, 2), (7, 3)), E=1.0e12, pressure=1.0, device=0):
    spec = G.pressurised_tube(nels=nels, p=p, E=E, pressure=pressure, shape=lambda th: 1.0 + 0.5 * amp * (1.0 - np.cos(4.0 * th)))
    nm = NonMatchingOptFFD.from_spec(spec, device=device, klass=NonMatchingOptFFD)
    nm.set_shopt_surf_inds_FFD([0, 1], [0, 1, 2, 3])
    blk = create_3D_block([2, 2, 1], 3, [list(l) for l in nm.cpsurf_des_lims])
    nm.set_shopt_FFD(blk.knots, blk.control)
    nm.set_shopt_align_CPFFD(align_dir=[[2], [2]])                       # the cross-section does not vary along the axis
    return nm


class ReducedShapeProblem:

    def __init__(self, nm, newton_rtol=1e-3):
        self.nm = nm
        self.disp, self.wint = DispImOpeartion(nm), IntEnergyExOperation(nm)
        self.D = nm.shopt_dcpsurf_fedcpffd.tocsr()
        self.A = [a.tocsr() for a in nm.shopt_dcpaligndcpffd]
        self.d0 = [np.asarray(v, float).copy() for v in nm.shopt_init_cpffd_design]
        self.n = [v.size for v in self.d0]
        # free design dofs: the FFD control points inside the block (those on its faces are pinned: the tube keeps its size)
        l, m, _ = nm.shopt_cpffd_shape
        lat = nm._lattice(nm.shopt_cpffd_shape)
        inner = [k for k, dof in enumerate(nm.shopt_cpffd_design_dof[0]) if 0 < lat[dof][0] < l - 1 and 0 < lat[dof][1] < m - 1]
        self.free = np.array(inner)
        self.rtol, self._x, self.n_state_solves = newton_rtol, None, 0

    def full(self, x):
        d = [v.copy() for v in self.d0]
        nf = self.free.size
        d[0][self.free] = x[:nf]
        d[1][self.free] = x[nf:]
        return d

    @property
    def x0(self):
        return np.concatenate([self.d0[0][self.free], self.d0[1][self.free]])

    def _solve(self, x):
        x = np.asarray(x, float)
        if self._x is None or not np.array_equal(x, self._x):
            d = self.full(x)
            for f in (0, 1):
                self.nm.update_CPIGA(self.D @ (self.A[f] @ d[f]), f)
            self.nm.update_uIGA(self.disp.solve_nonlinear(max_it=30, rtol=self.rtol))
            self._x = x.copy()
            self.n_state_solves += 1

    def objective(self, x):
        self._solve(x)
        return self.wint.Wint()

    def gradient(self, x):
        self._solve(x)
        self.disp.linearize()
        lam = self.disp.solve_linear_rev(self.wint.dWintduIGA(apply_bcs=True), np.zeros(self.nm.vec_iga_dof))
        g = [np.zeros(self.D.shape[0]), np.zeros(self.D.shape[0])]
        self.disp.apply_linear_rev(g, None, lam)                          # g[f] = (dR/dCP_f)^T lam
        out = [self.A[f].T @ (self.D.T @ (self.wint.dWintdCPIGA(f) - g[f])) for f in (0, 1)]
        return np.concatenate([out[0][self.free], out[1][self.free]])

    def roundness(self):
        """Standard deviation of the radius of the current control net's cross-section curve, evaluated on the patches."""
        r = []
        for s, P in enumerate(self.nm.splines):
            ctrl = P.control.copy()
            sl = slice(int(self.nm.cp_off[s]), int(self.nm.cp_off[s + 1]))
            for f in range(3):
                ctrl[:, :, f] = self.nm.cp_iga[f][sl].reshape(P.n_v, P.n_u).T
            Q = type(P)((P.p, P.q), P.knots, ctrl)
            for t in np.linspace(0.0, 1.0, 17):
                X = Q.eval((t, 0.5))
                r.append(np.hypot(X[0], X[1]))
        r = np.array(r)
        return float(r.std()), float(r.mean())


def run(maxiter=60, verbose=True, **kw):
    nm = build(**kw)
    prob = ReducedShapeProblem(nm)
    x0 = prob.x0
    w0 = prob.objective(x0)
    r0 = prob.roundness()
    s = 1.0 / w0
    res = minimize(lambda x: s * prob.objective(x), x0, jac=lambda x: s * prob.gradient(x), method="SLSQP",
                   bounds=[(-2.0, 2.0)] * x0.size, options=dict(maxiter=maxiter, ftol=1e-10, disp=False))
    w1 = prob.objective(res.x)
    r1 = prob.roundness()
    if verbose:
        print("internal energy %.6e -> %.6e (x %.1f smaller), %d iterations, %d state solves" % (w0, w1, w0 / w1, res.nit, prob.n_state_solves))
        print("cross-section radius: mean %.4f -> %.4f, standard deviation %.4f -> %.4f (circle: 0)" % (r0[1], r1[1], r0[0], r1[0]))
    return dict(x=res.x, w0=w0, w1=w1, r0=r0, r1=r1, problem=prob, result=res)


if __name__ == "__main__":
    run()
