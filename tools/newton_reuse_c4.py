"""Measurement: a Newton solve of C4 (smooth start, the bench model's loads) with a factorisation per step (the reference's iteration) and with chord steps."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from goldfish_amd import geometry as G
from goldfish_amd.nonmatching_opt import NonMatchingOpt
n = int(os.environ.get("GF_NR_PATCHES", "16"))
spec0 = G.synthetic_shell(n, n, nel=48, p=3, jitter=2)
# the bench model's load (1e3 N/m^2 on a 16 m cantilever of 1 cm) is far beyond what Newton reaches from zero without load steps; here: loads that bend it by a few
# thicknesses
import dataclasses
for q in [float(x) for x in os.environ.get("GF_NR_LOADS", "2e-3,2e-2").split(",")]:
    spec = dataclasses.replace(spec0, body_force=[[0.0, 0.0, -q]] * len(spec0.patches))
    amp = 0.0
    for reuse in (False, True):
        nm = NonMatchingOpt.from_spec(spec)
        nm.newton_reuse_factors = reuse
        nm.newton_reuse_min_dofs = 0
        nm.update_uIGA(G.smooth_displacement(spec, amp * spec.h_th))
        nm._assemble(3); nm.solve_K(np.ones(nm.vec_iga_dof))                      # the one-off cost (ordering, handles, graphs) outside the timing
        nm.update_uIGA(G.smooth_displacement(spec, amp * spec.h_th))
        t = time.perf_counter(); _, u = nm.solve_nonlinear_nonmatching_problem(rtol=1e-8, max_it=40, zero_mortar_funcs=False); dt = time.perf_counter() - t
        print("load %.0e, reuse %s: %.2f s, %d iterations (%d chord), converged %s (by step %s), |R|/|R0| %.2e; history %s"
              % (q, reuse, dt, nm.newton_iterations, nm.newton_chord_steps, nm.newton_converged, nm.newton_converged_by_step, nm.newton_relative_residual,
                 " ".join("%.1e" % h[0] for h in nm.newton_history)), flush=True)
        if reuse: assert np.abs(u - u_ref).max() < 1e-6 * np.abs(u_ref).max()
        else: u_ref = u
        nm._drop_device() if hasattr(nm, "_drop_device") else None
        del nm
