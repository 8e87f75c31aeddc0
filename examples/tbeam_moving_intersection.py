"""Shape optimisation with a MOVING intersection -- the set-up of the reference's
demos_om/shape_opt_mint/T-beam/T_beam_2patch_shopt_mi.py (design = x position of the web under the flange, bounds -1..1,
minimise the internal energy): when the web slides, the patch intersection slides with it, so the parametric coordinates
of the mortar vertices are states xi(CP) (CPIGA2Xi) and the total derivative carries dR/dxi:

    xi(CP)      Rxi(xi; CP) = 0                         cpiga2xi.CPIGA2Xi.solve_xi            (host, 4 unknowns per vertex)
    u(CP, xi)   R(u; CP, xi) = 0                        DispMintImOpeartion.solve_nonlinear   (device assembly)
    dW/ds = dW/dCP . c' - lam^T [ dR/dCP . c' + dR/dxi . xi' ],   xi' = -(dRxi/dxi)^-1 dRxi/dCP . c',   K^T lam = dW/du

With a load that is symmetric about the flange's centre line the optimum is the centred web, s = 0 (known by symmetry);
the start is s = 0.4.  Usage: python examples/tbeam_moving_intersection.py
"""
import os
import sys

import numpy as np
from scipy.optimize import minimize

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from goldfish_amd import geometry as G                                    # noqa: E402
from goldfish_amd.nonmatching_opt import NonMatchingOptFFD                # noqa: E402
from goldfish_amd.operations.disp_mi_imop import DispMintImOpeartion      # noqa: E402
from goldfish_amd.operations.int_energy_exop import IntEnergyExOperation  # noqa: E402


class SlidingWebProblem:

    def __init__(self, num_el=4, newton_rtol=1e-8):
        spec = G.tbeam_2patch(num_el, load=(0.0, 0.0, -1.0), tip_load=0.0)
        spec.body_force = [[0.0, 0.0, -1.0], [0.0, 0.0, 0.0]]               # the flange carries the load: symmetric about x = 0
        self.nm = nm = NonMatchingOptFFD.from_spec(spec)
        nm.set_shopt_surf_inds_FFD([0], [0, 1])
        nm.create_diff_intersections()
        self.c2x = nm.cpiga2xi
        self.disp, self.wint = DispMintImOpeartion(nm), IntEnergyExOperation(nm)
        n0 = spec.patches[0].ncp
        self.cp0 = nm.get_init_CPIGA()[0].copy()
        self.dcp = np.zeros(self.cp0.size)
        self.dcp[n0:] = spec.patches[1].cp_hom_flat()[:, 3]               # d(homogeneous x of the web)/ds
        self.rtol, self._s, self.n_state_solves = newton_rtol, None, 0

    def _solve(self, s):
        s = float(np.ravel(s)[0])
        if self._s is None or s != self._s:
            cp = self.cp0 + s * self.dcp
            self.nm.update_CPIGA(cp, 0)
            self.c2x.update_CPs(cp, 0)
            self.xi = self.c2x.solve_xi(self.c2x.xi_flat_global)
            self.nm.update_xi(self.xi)
            self.nm.update_transfer_matrices()
            self.nm.update_uIGA(self.disp.solve_nonlinear(max_it=30, rtol=self.rtol))
            self._s = s
            self.n_state_solves += 1

    def objective(self, s):
        self._solve(s)
        return self.wint.Wint()

    def gradient(self, s):
        self._solve(s)
        nm, c2x = self.nm, self.c2x
        self.disp.linearize()
        lam = self.disp.solve_linear_rev(self.wint.dWintduIGA(apply_bcs=True), np.zeros(nm.vec_iga_dof))
        dxi = -np.linalg.solve(c2x.dRdxi(self.xi), c2x.dRdCP(self.xi, 0, coo=False) @ self.dcp)
        back = [np.zeros(self.cp0.size), np.zeros(nm.xi_size)]
        self.disp.apply_linear_rev(back, None, lam)                       # (dR/dCP)^T lam, (dR/dxi)^T lam
        return np.array([self.wint.dWintdCPIGA(0) @ self.dcp - back[0] @ self.dcp - back[1] @ dxi])


def run(s0=0.4, verbose=True, **kw):
    prob = SlidingWebProblem(**kw)
    w0 = prob.objective(s0)
    res = minimize(lambda s: prob.objective(s) / w0, [s0], jac=lambda s: prob.gradient(s) / w0, method="SLSQP",
                   bounds=[(-0.8, 0.8)], options=dict(maxiter=50, ftol=1e-14))
    s1 = float(res.x[0])
    w1 = prob.objective(s1)
    if verbose:
        print("web position %.4f -> %.6f (symmetric optimum: 0), internal energy %.6e -> %.6e, %d state solves"
              % (s0, s1, w0, w1, prob.n_state_solves))
    return dict(s0=s0, s1=s1, w0=w0, w1=w1, problem=prob, result=res)


if __name__ == "__main__":
    run()
