"""Thickness optimisation of the six-patch plate -- the problem of the reference's
demos_om/thickness_opt/plate/plate_const_th_opt_wint.py (ThicknessOptGroup :13-117: minimise the internal energy
subject to constant volume, one thickness per patch, bounds 4e-3 .. 5e-2, SLSQP) -- driven directly through the
operations of goldfish_amd instead of an OpenMDAO group:

    state      R(u; h) = 0                       DispImOpeartion.solve_nonlinear     (Newton, K on the device)
    objective  W(u, h)                           IntEnergyExOperation
    adjoint    K^T lam = dW/du                   DispImOpeartion.solve_linear_rev
    gradient   dW/dh - (dR/dh)^T lam             DispImOpeartion.apply_linear_rev    (dR/dh on the device)
    constraint V(h) = V(h_0)                     VolumeExOperation

With openmdao installed the same problem is the reference's group wired from goldfish_amd.om_comps (HthMapComp,
DispStatesComp, IntEnergyComp, VolumeComp) unchanged.  Usage: python examples/plate_thickness_opt.py

Several GPUs (the reference runs its demos under mpirun with ``comm`` passed to NonMatchingOpt):
    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 examples/plate_thickness_opt.py
every rank runs this script on replicated design and state vectors; the patches, the assembly, the products and the functionals are sharded over the ranks
(one GPU each over RCCL; with fewer GPUs than ranks -- a rehearsal -- all ranks share GPU 0 over gloo), the direct solves run on the device of every rank
(GF_SHARDED_SOLVER=distributed: the distributed factorisation also for a model this small).
"""
import os
import sys

import numpy as np
from scipy.optimize import minimize

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from goldfish_amd import geometry as G                                   # noqa: E402
from goldfish_amd.nonmatching_opt import NonMatchingOptFFD               # noqa: E402
from goldfish_amd.operations.disp_imop import DispImOpeartion            # noqa: E402
from goldfish_amd.operations.int_energy_exop import IntEnergyExOperation  # noqa: E402
from goldfish_amd.operations.volume_exop import VolumeExOperation        # noqa: E402


class ReducedThicknessProblem:
    """Objective / constraint of the reduced (state-eliminated) problem in the per-patch thicknesses."""

    def __init__(self, nm, newton_rtol=1e-5):      # the residual of this plate's equilibrium states has a floor of ~7e-7 |R_0| (cond(K) eps: penalty coefficient 1e3)
        self.nm = nm
        nm.set_thickness_opt(var_thickness=False)
        self.disp, self.wint, self.vol = DispImOpeartion(nm), IntEnergyExOperation(nm), VolumeExOperation(nm)
        self.rtol = newton_rtol
        self._h = None
        self.n_state_solves = 0

    def _solve(self, h):
        h = np.asarray(h, float)
        if self._h is None or not np.array_equal(h, self._h):
            self.nm.update_h_th(h)
            self.nm.update_uIGA(self.disp.solve_nonlinear(max_it=30, rtol=self.rtol))
            self._h = h.copy()
            self.n_state_solves += 1

    def objective(self, h):
        self._solve(h)
        return self.wint.Wint()

    def gradient(self, h):
        self._solve(h)
        self.disp.linearize()                                           # K, dR/dh at the converged state
        lam = self.disp.solve_linear_rev(self.wint.dWintduIGA(apply_bcs=True), np.zeros(self.nm.vec_iga_dof))
        g = [np.zeros(self.nm.h_th_dof)]
        self.disp.apply_linear_rev(g, None, lam)                        # g[0] = (dR/dh)^T lam   (opt_shape is off: one input)
        return self.wint.dWintdh_th() - g[0]

    def volume(self, h):
        self.nm.update_h_th(np.asarray(h, float))
        self._h = None
        return self.vol.volume()

    def volume_gradient(self, h):
        self.nm.update_h_th(np.asarray(h, float))
        self._h = None
        return self.vol.dvoldh_th()


def problem_from_reference_files():
    """The same problem assembled the way the demo does it: the six surfaces from the reference's IGES file
    (demos_csdl_alpha/thickness_opt/geometry/plate_geometry.igs) and the interfaces from its intersection cache
    (plate_int_data.npz), both committed as data under tests/golden/."""
    from goldfish_amd.cpiga2xi import IntersectionData
    from goldfish_amd.nonmatching_opt import PointSource, SVKResidual
    from goldfish_amd.utils.iges import read_iges_surfaces
    gold = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
    patches = read_iges_surfaces(os.path.join(gold, "ref_plate_geometry.igs"))
    s0 = patches[0]                                                        # clampedBC of the demo (:119-133)
    s0.add_zero_dofs(0, s0.get_side_dofs(0, 0, 1))
    for f in (1, 2):
        s0.add_zero_dofs(f, s0.get_side_dofs(0, 0, 2))
    data = IntersectionData.load_intersections_data(os.path.join(gold, "ref_plate_int_data.npz"), patches)
    nm = NonMatchingOptFFD(patches, 68e9, 1.0e-2, 0.35)
    nm.create_mortar_meshes(data.mortar_nels)
    nm.mortar_meshes_setup(data.mapping_list, data.intersections_para_coords, 1.0e3)
    nm.set_residuals([SVKResidual()] * len(patches))
    loads = G.edge_traction_point_loads(patches, 5, 0, 1, (0.0, 0.0, -100.0))
    nm.set_point_sources([PointSource(xi, f, v) for (_, xi, f, v) in loads], [s for (s, _, _, _) in loads])
    return nm, 1.0e-2


def run(p=3, maxiter=60, lower=4e-3, upper=5e-2, verbose=True, from_files=False, comm=None, device=0):
    if from_files:
        nm, h_init = problem_from_reference_files()
        spec = type("S", (), {"h_th": h_init})()
    else:
        spec = G.plate_6patch(p)
        nm = NonMatchingOptFFD.from_spec(spec, comm=comm, device=device)
    prob = ReducedThicknessProblem(nm)
    h0 = np.full(nm.num_splines, spec.h_th)
    v0, w0 = prob.volume(h0), prob.objective(h0)
    s = 1.0 / w0                                                        # the demo scales the objective (scaler=1e3)
    res = minimize(lambda h: s * prob.objective(h), h0, jac=lambda h: s * prob.gradient(h), method="SLSQP",
                   bounds=[(lower, upper)] * h0.size,
                   constraints=[dict(type="eq", fun=lambda h: (prob.volume(h) - v0) / v0, jac=lambda h: prob.volume_gradient(h) / v0)],
                   options=dict(maxiter=maxiter, ftol=1e-10, disp=False))
    w1 = prob.objective(res.x)
    if verbose:
        print("internal energy %.6e -> %.6e  (%.1f %% of the uniform plate), volume drift %.2e, %d state solves"
              % (w0, w1, 100 * w1 / w0, abs(prob.volume(res.x) - v0) / v0, prob.n_state_solves))
        for k, t in enumerate(res.x):
            print("Thickness for patch %2d: %10.6f" % (k, t))
    return dict(h=res.x, w0=w0, w1=w1, v0=v0, v1=prob.volume(res.x), problem=prob, result=res)


def main():
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1:
        run()
        return
    import torch
    import torch.distributed as dist
    rank, local = int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if torch.cuda.device_count() >= world:                                # one GPU per rank: RCCL
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    else:                                                                 # rehearsal: every rank on GPU 0, exchanges through the host
        local = 0
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        run(comm=dist, device=local, verbose=rank == 0)
        dist.barrier()                      # only on the way out of a run that finished: a barrier in ``finally`` turns one rank's exception into a hang of the others
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
