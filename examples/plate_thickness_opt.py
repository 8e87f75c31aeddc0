#!/usr/bin/env python3
"""Thickness optimisation of the six-patch plate (config C1) on the MI355X path.

Mirrors the reference demo demos_om/thickness_opt/plate/plate_const_th_opt_wint.py (ThicknessOptGroup,
:12-124): design variables = one thickness per patch, objective = internal energy W_int, constraint =
constant material volume, state = displacements solved by Newton; total derivatives by the adjoint
(DispImOpeartion.linearize / solve_linear_rev / apply_linear_rev + IntEnergyExOperation partials).
OpenMDAO is optional: without it the same operations are driven by scipy.optimize (SLSQP)."""
import os
import sys

import numpy as np
from scipy.optimize import minimize

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from goldfish_amd import geometry as G                                   # noqa: E402
from goldfish_amd.nonmatching_opt import NonMatchingOpt                   # noqa: E402
from goldfish_amd.operations.disp_imop import DispImOpeartion             # noqa: E402
from goldfish_amd.operations.int_energy_exop import IntEnergyExOperation  # noqa: E402
from goldfish_amd.operations.volume_exop import VolumeExOperation         # noqa: E402


def build_problem():
    spec = G.plate_6patch()
    nm = NonMatchingOpt.from_spec(spec)
    nm.set_thickness_opt(var_thickness=False)
    return nm


def run(max_iter=15, verbose=True):
    nm = build_problem()
    disp, wint, vol = DispImOpeartion(nm), IntEnergyExOperation(nm), VolumeExOperation(nm)
    h0 = nm.init_h_th.copy()
    scale = 1.0 / h0.mean()

    def state(h):
        nm.update_h_th(h)
        nm.solve_nonlinear_nonmatching_problem(rtol=1e-10, max_it=30)

    def objective(x):
        state(x / scale)
        return wint.Wint()

    def gradient(x):
        state(x / scale)
        disp.linearize()
        lam = disp.solve_linear_rev(wint.dWintduIGA(apply_bcs=True).copy(), np.zeros(nm.vec_iga_dof))
        g = np.zeros(h0.size)
        disp.apply_linear_rev([g], None, lam)
        return (wint.dWintdh_th() - g) / scale

    state(h0)
    V0, W0 = vol.volume(), wint.Wint()

    def vol_con(x):
        nm.update_h_th(x / scale)
        return (vol.volume() - V0) / V0

    def vol_jac(x):
        nm.update_h_th(x / scale)
        return vol.dvoldh_th() / V0 / scale

    res = minimize(objective, h0 * scale, jac=gradient, method="SLSQP",
                   bounds=[(0.2, 5.0)] * h0.size, constraints=[{"type": "eq", "fun": vol_con, "jac": vol_jac}],
                   options={"maxiter": max_iter, "ftol": 1e-12})
    h = res.x / scale
    state(h)
    out = dict(W0=W0, W=wint.Wint(), V0=V0, V=vol.volume(), h0=h0, h=h, nit=res.nit)
    if verbose:
        print("W_int: %.6e -> %.6e  (%.1f %% lower), volume %.6e -> %.6e, %d iterations" %
              (out["W0"], out["W"], 100 * (1 - out["W"] / out["W0"]), out["V0"], out["V"], out["nit"]))
        print("thickness per patch:", np.array2string(h, precision=5))
    return out


if __name__ == "__main__":
    run()
