"""Residual histories of the Newton solves that VERDICT r02 names (the sliding-web T-beam of examples/tbeam_moving_intersection.py
and the arch of examples/arch_shape_opt.py, p = 2, 3) with the device L D L^T and with the host SuperLU: per iteration the relative
residual, the relative Newton correction and the step length -- whether the iteration ends at rtol, at a negligible correction, at the
evaluation floor of the residual, or not at all.  Usage (GPU box): python tools/newton_history.py > profiles/r03_newton_history.txt"""
import importlib.util
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from goldfish_amd.nonmatching_opt import NonMatchingOpt  # noqa: E402


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "examples", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def report(tag, nm, rtol):
    for solver in ("device", "host"):
        NonMatchingOpt.linear_solver = solver
        nm._dsolver = nm._hlu = None
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            nm.solve_nonlinear_nonmatching_problem(rtol=rtol, max_it=30)
        print("%s  [%s solver]  rtol %.0e: converged=%s by_step=%s stagnated=%s iterations=%d  |u|=%.3e" % (
            tag, solver, rtol, nm.newton_converged, nm.newton_converged_by_step, nm.newton_stagnated, nm.newton_iterations, np.linalg.norm(nm.u_iga)))
        for k, (r, s, lam) in enumerate(nm.newton_history):
            print("    it %2d  |R|/|R0| %.3e   |du|/|u| %.3e   step %.4f" % (k + 1, r, s, lam))
        for x in w:
            print("    warning:", str(x.message)[:200])
    NonMatchingOpt.linear_solver = "device"


def main():
    mint = _load("tbeam_moving_intersection")
    prob = mint.SlidingWebProblem()
    cp = prob.cp0 + 0.4 * prob.dcp
    prob.nm.update_CPIGA(cp, 0)
    prob.c2x.update_CPs(cp, 0)
    xi = prob.c2x.solve_xi(prob.c2x.xi_flat_global)
    prob.nm.update_xi(xi)
    prob.nm.update_transfer_matrices()
    report("sliding-web T-beam, s = 0.4", prob.nm, 1e-11)
    arch = _load("arch_shape_opt")
    for p in (2, 3, 4):
        nm = arch.build(p=p)
        report("arch, rise 3, p = %d" % p, nm, 1e-10)
        pr = arch.ReducedShapeProblem(nm)
        nm.update_CPIGA(pr.D @ (pr.A @ (pr.d0 * 1.6)), 2)
        report("arch, FFD design x 1.6, p = %d" % p, nm, 1e-10)


if __name__ == "__main__":
    main()
