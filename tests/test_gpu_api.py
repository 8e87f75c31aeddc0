"""GPU tests of the drop-in surface: NonMatchingOpt, the operations and the OpenMDAO
components (reference: GOLDFISH/nonmatching_opt.py, operations/, om_comps/), through the C ABI."""
import os
import socket

import numpy as np
import pytest

from goldfish_amd import geometry as G
from goldfish_amd.model import arrays_from_spec

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def _problem(var_thickness=True, seed=0):
    from goldfish_amd.nonmatching_opt import NonMatchingOptFFD
    spec = G.tbeam_2patch(4)
    rng = np.random.default_rng(seed)
    th = [spec.h_th * rng.uniform(0.8, 1.2, p.ncp) for p in spec.patches] if var_thickness else None
    nm = NonMatchingOptFFD.from_spec(spec, thickness=th)
    nm.set_shopt_surf_inds_FFD([0, 1, 2], [[0, 1]] * 3)
    nm.set_thickness_opt(var_thickness=var_thickness)
    return spec, th, nm


def test_functionals_parity(oracle_lib):
    from goldfish_amd import _lib
    from oracle.oracle_py import Oracle
    spec = G.scordelis_lo_9patch(3, nels=[2, 1, 2, 3, 2, 3, 2, 1, 2])
    rng = np.random.default_rng(1)
    th = [spec.h_th * rng.uniform(0.8, 1.2, p.ncp) for p in spec.patches]
    A = arrays_from_spec(spec, th)
    h, u = np.concatenate(th), 2e-2 * rng.standard_normal(A.ndof)
    O = Oracle(A, thickness=h, u=u)
    D = _lib.DeviceModel(A)
    D.set_thickness(h)
    D.set_u(u)
    for bcs in (True, False):
        F, Fo = D.functionals(apply_bcs=bcs), O.functionals(apply_bcs=bcs)
        for k in ("Wint", "volume", "Wpen"):
            assert abs(F[k] - Fo[k]) < 1e-11 * abs(Fo[k]), k
        assert _rel(F["dWdu"], Fo["dWdu"]) < 1e-10
        assert _rel(F["dWdh"], Fo["dWdh"]) < 1e-10 and _rel(F["dVdh"], Fo["dVdh"]) < 1e-10
        for f in range(3):
            assert _rel(F["dWdcp"][f], Fo["dWdcp"][f]) < 1e-10 and _rel(F["dVdcp"][f], Fo["dVdcp"][f]) < 1e-10
    D.close()


def test_compliance_parity_and_comp(oracle_lib):
    from goldfish_amd import _lib
    from goldfish_amd.om_comps import ComplianceComp, om
    from oracle.oracle_py import Oracle
    spec = G.scordelis_lo_9patch(3, nels=[2, 1, 2, 3, 2, 3, 2, 1, 2])
    rng = np.random.default_rng(2)
    A = arrays_from_spec(spec)
    h, u = np.full(A.total_cp, spec.h_th), 2e-2 * rng.standard_normal(A.ndof)
    forces = rng.standard_normal((9, 3))
    O, D = Oracle(A, thickness=h, u=u), _lib.DeviceModel(A)
    D.set_thickness(h)
    D.set_u(u)
    for bcs in (True, False):
        Cd, Co = D.compliance(forces, apply_bcs=bcs), O.compliance(forces, apply_bcs=bcs)
        assert abs(Cd["C"] - Co["C"]) < 1e-11 * abs(Co["C"])
        assert _rel(Cd["dCdu"], Co["dCdu"]) < 1e-10
        for f in range(3):
            assert _rel(Cd["dCdcp"][f], Co["dCdcp"][f]) < 1e-10
    with pytest.raises(ValueError):
        D.compliance(np.zeros(5))
    D.close()
    spec, th, nm = _problem()
    comp = ComplianceComp(nonmatching_opt=nm, forces=[[0.0, 0.0, 1.0]] * 2)
    comp.init_parameters()
    prob = om.Problem(model=comp)
    prob.setup()
    prob.run_model()
    free = np.ones(nm.vec_iga_dof, bool)
    free[np.asarray(nm.dev and nm.zero_dofs)] = False             # dC/du with its Dirichlet rows zeroed (compliance_comp.py:130 of the reference)
    assert max(prob.check_partials(compact_print=False, free_mask=free).values()) < 1e-6


def test_nonmatching_opt_surface(oracle_lib):
    from oracle.oracle_py import Oracle
    spec, th, nm = _problem()
    A = arrays_from_spec(spec, th)
    rng = np.random.default_rng(2)
    u = 1e-2 * rng.standard_normal(A.ndof)
    nm.update_uIGA(u)
    O = Oracle(A, thickness=np.concatenate(th), u=u)
    assert nm.vec_iga_dof == A.ndof and nm.vec_scalar_iga_dof == A.total_cp
    assert _rel(nm.RIGA(), O.residual()) < 1e-10
    vals = O.assemble()
    assert abs(nm.dRIGAduIGA() - O.csr(0, vals[0])).max() < 1e-10 * abs(vals[0]).max()
    assert abs(nm.dRIGAdCPIGA(1) - O.csr(2, vals[2])).max() < 1e-10 * abs(vals[2]).max()
    assert abs(nm.dRIGAdh_th() - O.csr(4, vals[4])).max() < 1e-10 * abs(vals[4]).max()
    with pytest.raises(ValueError):
        nm.update_uIGA(np.zeros(7))
    # reference-style FD check of the shape Jacobian on a few columns (nonmatching_opt.py:975-990)
    cp = nm.get_init_CPIGA()[2].copy()
    J = nm.dRIGAdCPIGA(2).toarray()
    dc = rng.standard_normal(cp.size)
    nm.update_CPIGA(cp + 1e-6 * dc, 2)
    Rp = nm.RIGA()
    nm.update_CPIGA(cp - 1e-6 * dc, 2)
    Rm = nm.RIGA()
    nm.update_CPIGA(cp, 2)
    assert _rel((Rp - Rm) / 2e-6, J @ dc) < 1e-6


def test_scordelis_lo_on_gpu():
    from goldfish_amd.nonmatching_opt import NonMatchingOpt
    from goldfish_amd.splines import basis_ders, find_span
    spec = G.scordelis_lo_9patch(6)
    nm = NonMatchingOpt.from_spec(spec)
    u = nm.solve_linear_nonmatching_problem()
    P = spec.patches[3]                      # free edge u=0, mid length v=0.5
    su, sv = find_span(P.n_u, P.p, P.knots[0], 0.0), find_span(P.n_v, P.q, P.knots[1], 0.5)
    Nu, Nv = basis_ders(su, 0.0, P.p, P.knots[0], 0)[0], basis_ders(sv, 0.5, P.q, P.knots[1], 0)[0]
    off, num, W = int(nm.cp_off[3]), 0.0, 0.0
    for jv in range(4):
        for ju in range(4):
            a = P.flat(su - 3 + ju, sv - 3 + jv)
            num += Nu[ju] * Nv[jv] * u[3 * (off + a) + 1]
            W += Nu[ju] * Nv[jv] * P.cp_hom_flat()[a, 3]
    assert abs(abs(num / W) - 0.3006) / 0.3006 < 5e-4


def test_newton_solve_converges():
    from goldfish_amd.nonmatching_opt import NonMatchingOpt
    spec = G.tbeam_2patch(6)
    nm = NonMatchingOpt.from_spec(spec)
    _, u = nm.solve_nonlinear_nonmatching_problem(rtol=1e-9, max_it=30)
    assert np.linalg.norm(nm.RIGA()) < 1e-6 * 10.0 and np.abs(u).max() > 1e-3


@pytest.mark.parametrize("var_thickness", [True, False])
def test_disp_states_comp_partials(var_thickness):
    from goldfish_amd.om_comps import DispStatesComp, om
    spec, th, nm = _problem(var_thickness)
    comp = DispStatesComp(nonmatching_opt=nm)
    comp.init_parameters(nonlinear_solver_rtol=1e-8)
    prob = om.Problem(model=comp)
    prob.setup()
    prob.run_model()
    assert np.abs(prob["displacements"]).max() > 1e-4
    free = np.ones(nm.vec_iga_dof, bool)
    free[nm.zero_dofs] = False
    errs = prob.check_partials(compact_print=False, free_mask=free)
    assert max(errs.values()) < 1e-5, errs
    # reverse mode == transpose of forward mode; solve_linear inverts K
    rng = np.random.default_rng(3)
    n = nm.vec_iga_dof
    op = comp.disp_state_imop
    du, lam = rng.standard_normal(n), rng.standard_normal(n)
    fwd = op.apply_linear_fwd(None, du, np.zeros(n))
    _, rev = op.apply_linear_rev(None, np.zeros(n), lam)
    assert abs(lam @ fwd - du @ rev) < 1e-9 * abs(lam @ fwd)
    x = op.solve_linear_fwd(np.zeros(n), fwd.copy())
    assert _rel(x[free], du[free]) < 1e-8
    lt = op.solve_linear_rev(rev.copy(), np.zeros(n))
    assert _rel(lt[free], lam[free]) < 1e-8


@pytest.mark.parametrize("var_thickness", [True, False])
def test_energy_and_volume_comp_partials(var_thickness):
    from goldfish_amd.om_comps import IntEnergyComp, VolumeComp, om
    spec, th, nm = _problem(var_thickness)
    nm.solve_linear_nonmatching_problem()
    for Comp in (IntEnergyComp, VolumeComp):
        comp = Comp(nonmatching_opt=nm)
        comp.init_parameters()
        prob = om.Problem(model=comp)
        prob.setup()
        prob.run_model()
        free = np.ones(nm.vec_iga_dof, bool)
        free[np.asarray(nm.dev and nm.zero_dofs)] = False         # d w_int / d displacements has its Dirichlet rows zeroed (apply_bcs=True, as in the reference)
        errs = prob.check_partials(compact_print=False, step=1e-6, free_mask=free)
        assert max(errs.values()) < 1e-6, (Comp.__name__, errs)


def _rank_worker(rank, world, port, q, backend="gloo"):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dev = rank if backend == "nccl" else 0               # nccl (= RCCL): one GPU per rank; gloo: every rank on GPU 0
    if backend == "nccl":
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from goldfish_amd import _lib, sharding
    spec = G.synthetic_shell(3, 2, nel=4, p=3, jitter=1)
    th, u = G.random_thickness(spec), G.smooth_displacement(spec, 0.5 * spec.h_th)
    S = sharding.ShardedDeviceModel(spec, dist, rank, world, device=dev, thickness_global=th)
    S.set_thickness(np.concatenate(th))
    S.set_u(u)
    S.assemble()
    if backend == "nccl":                                 # bench.py's exchange: device-resident all-gather of the owned residual rows
        Rg = sharding.allgather_owned_rows(S.shard, torch.from_numpy(S.D.residual()).cuda(), dist, 3)
        assert np.array_equal(Rg.cpu().numpy(), S.residual())
    rng = np.random.default_rng(11)                       # same seed on every rank: replicated inputs
    xu, xc, lam = rng.standard_normal(S.ndof), rng.standard_normal(S.total_cp), rng.standard_normal(S.ndof)
    res = dict(R=S.residual(), Ku=S.apply(_lib.MAT_K, xu), KTl=S.apply(_lib.MAT_K, lam, transpose=True),
               Cc=S.apply(_lib.MAT_DRDCP1, xc), CTl=S.apply(_lib.MAT_DRDCP1, lam, transpose=True),
               HTl=S.apply(_lib.MAT_DRDH, lam, transpose=True), F=S.functionals(apply_bcs=False), xu=xu, xc=xc, lam=lam,
               S=S.stress_forms(1, 3.0, 1e6 * (1.0 + np.arange(len(spec.patches))), -1, 0, apply_bcs=False),
               G=S.shape_regu(2, 0.98 * np.concatenate([p.cp_hom_flat()[:, 2] for p in spec.patches]), 1.0 + np.arange(len(spec.patches))))
    if rank == 0:
        q.put(res)
    dist.barrier()
    S.close()
    dist.destroy_process_group()


def _run_ranks(world, backend):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")                         # fresh interpreters: the children initialise their own GPUs
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_worker, args=(r, world, port, q, backend)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=600)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    return res


def test_rccl_sharded_assembly_when_several_gpus_are_visible(oracle_lib):
    """The N > 1 path over RCCL (backend "nccl"): min(device_count, 4) ranks, one GPU each -- residual, products, functionals and
    the device-resident all-gather of bench.py against the unsharded oracle.  Skips on a one-GPU box (the gloo test below covers
    the same code with both ranks on GPU 0); on a multi-GPU lease it is the first thing that exercises RCCL."""
    import torch
    n = torch.cuda.device_count()                         # does not initialise the GPU in this process
    if n < 2:
        pytest.skip("one GPU visible: RCCL needs at least two")
    _check_sharded_results(_run_ranks(min(n, 4), "nccl"))


def test_two_rank_sharded_assembly_on_gpu(oracle_lib):
    """Two processes (gloo; both on the single GPU of the box) own half of the patches each."""
    _check_sharded_results(_run_ranks(2, "gloo"))


def _check_sharded_results(res):
    from oracle.oracle_py import Oracle
    spec = G.synthetic_shell(3, 2, nel=4, p=3, jitter=1)
    th = G.random_thickness(spec)
    O = Oracle(arrays_from_spec(spec, th), thickness=np.concatenate(th), u=G.smooth_displacement(spec, 0.5 * spec.h_th))
    assert _rel(res["R"], O.residual()) < 1e-10
    vals = O.assemble()
    K, C1, H = O.csr(0, vals[0]), O.csr(2, vals[2]), O.csr(4, vals[4])
    assert _rel(res["Ku"], K @ res["xu"]) < 1e-10 and _rel(res["KTl"], K.T @ res["lam"]) < 1e-10
    assert _rel(res["Cc"], C1 @ res["xc"]) < 1e-10 and _rel(res["CTl"], C1.T @ res["lam"]) < 1e-10
    assert _rel(res["HTl"], H.T @ res["lam"]) < 1e-10
    Fo, F = O.functionals(apply_bcs=False), res["F"]
    for k in ("Wint", "volume", "Wpen"):
        assert abs(F[k] - Fo[k]) < 1e-11 * abs(Fo[k]), k
    assert _rel(F["dWdu"], Fo["dWdu"]) < 1e-10 and _rel(F["dWdh"], Fo["dWdh"]) < 1e-10
    assert _rel(F["dWdcp"][2], Fo["dWdcp"][2]) < 1e-10 and _rel(F["dVdcp"][0], Fo["dVdcp"][0]) < 1e-10
    So, Sd = O.stress_forms(1, 3.0, 1e6 * (1.0 + np.arange(len(spec.patches))), -1.0, 0, apply_bcs=False), res["S"]
    assert _rel(Sd["I"], So["I"]) < 1e-11 and _rel(Sd["vmax"], So["vmax"]) < 1e-11
    assert _rel(Sd["dIdu"], So["dIdu"]) < 1e-10 and _rel(Sd["dIdh"], So["dIdh"]) < 1e-10
    assert _rel(Sd["dIdcp"][1], So["dIdcp"][1]) < 1e-10
    Go = O.shape_regu(2, 0.98 * np.concatenate([p.cp_hom_flat()[:, 2] for p in spec.patches]), 1.0 + np.arange(len(spec.patches)))
    assert abs(res["G"]["value"] - Go["value"]) < 1e-11 * abs(Go["value"]) and _rel(res["G"]["dcp"][0], Go["dcp"][0]) < 1e-10


def test_ffd_chain_rule_through_the_gpu_path():
    """N2: dR/d(FFD control points) = dRIGAdCPIGA(field) @ shopt_dcpsurf_fedcpffd, checked by finite
    differences of RIGA through CPFFD2SurfComp -> update_CPIGA."""
    from goldfish_amd.om_comps import om
    from goldfish_amd.om_comps.ffd_comps import CPFE2IGAComp, CPFFD2SurfComp
    from goldfish_amd.utils.ffd_utils import create_3D_block
    spec, th, nm = _problem()
    blk = create_3D_block([2, 3, 1], 3, nm.cpsurf_lims)
    D = nm.set_shopt_FFD(blk.knots, blk.control)
    comp = CPFFD2SurfComp(nonmatching_opt_ffd=nm)
    comp.init_parameters()
    prob = om.Problem(model=comp)
    prob.setup()
    prob.run_model()
    for i, f in enumerate(nm.opt_field):
        assert _rel(prob["CP_FE%d" % f], nm.get_init_CPIGA()[i]) < 1e-12       # identity at the start
    assert max(prob.check_partials(compact_print=False).values()) < 1e-8
    c2 = CPFE2IGAComp(nonmatching_opt=nm)
    c2.init_parameters()
    p2 = om.Problem(model=c2)
    p2.setup()
    p2.run_model()
    assert max(p2.check_partials(compact_print=False).values()) < 1e-8
    nm.update_uIGA(1e-2 * np.random.default_rng(3).standard_normal(nm.vec_iga_dof))
    field = 2
    J = nm.dRIGAdCPIGA(field) @ D.tocsr()
    q0 = nm.shopt_cpffd_flat[:, field].copy()
    dq = np.random.default_rng(4).standard_normal(q0.size)
    R = []
    for sgn in (1, -1):
        nm.update_CPIGA(D @ (q0 + sgn * 1e-6 * dq), field)
        R.append(nm.RIGA())
    nm.update_CPIGA(D @ q0, field)
    assert _rel((R[0] - R[1]) / 2e-6, J @ dq) < 1e-6
    # design dofs after alignment (CPFFDesign2FullComp): dR/d(design) = dR/dCP @ FFD map @ align map
    from goldfish_amd.om_comps.ffd_comps import CPFFDesign2FullComp
    fi = nm.opt_field.index(field)
    al = [None] * len(nm.opt_field)
    al[fi] = [1]
    A = nm.set_shopt_align_CPFFD(al)[fi].tocsr()
    c3 = CPFFDesign2FullComp(nonmatching_opt_ffd=nm)
    c3.init_parameters()
    p3 = om.Problem(model=c3)
    p3.setup()
    p3.run_model()
    assert max(p3.check_partials(compact_print=False).values()) < 1e-8
    d0 = nm.shopt_init_cpffd_design[fi].copy()
    dd = np.random.default_rng(5).standard_normal(d0.size)
    R = []
    for sgn in (1, -1):
        nm.update_CPIGA(D @ (A @ (d0 + sgn * 1e-6 * dd)), field)
        R.append(nm.RIGA())
    nm.update_CPIGA(D @ (A @ d0), field)
    assert _rel((R[0] - R[1]) / 2e-6, J @ (A @ dd)) < 1e-6


def test_adjoint_total_derivatives_vs_finite_differences():
    """End-to-end use of the path: d W_int / d(design) by the adjoint
    K^T lambda = dW/du, dW/dx - lambda^T dR/dx, against central differences of the reduced objective
    (state re-solved by Newton for every perturbed design)."""
    from goldfish_amd.operations.disp_imop import DispImOpeartion
    from goldfish_amd.operations.int_energy_exop import IntEnergyExOperation
    spec, th, nm = _problem(var_thickness=False)
    disp, wint = DispImOpeartion(nm), IntEnergyExOperation(nm)
    h0 = nm.init_h_th.copy()
    cp0 = nm.get_init_CPIGA()[2].copy()

    def reduced(h, cp):
        nm.update_h_th(h)
        nm.update_CPIGA(cp, 2)
        nm.solve_nonlinear_nonmatching_problem(rtol=1e-8, max_it=30)
        return wint.Wint()

    W0 = reduced(h0, cp0)
    disp.linearize()
    n = nm.vec_iga_dof
    lam = disp.solve_linear_rev(wint.dWintduIGA(apply_bcs=True).copy(), np.zeros(n))
    g_h, g_cp = np.zeros(h0.size), np.zeros(cp0.size)
    d_in = [np.zeros(nm._shopt_cols[0].size), np.zeros(nm._shopt_cols[1].size), g_cp, g_h]
    disp.apply_linear_rev(d_in, None, lam)
    tot_h = wint.dWintdh_th() - g_h
    tot_cp = wint.dWintdCPIGA(2) - g_cp
    rng = np.random.default_rng(8)
    dh, dcp = rng.standard_normal(h0.size), rng.standard_normal(cp0.size) * (np.abs(cp0) > -1)
    eh, ec = 1e-5 * h0.mean(), 1e-5
    fd_h = (reduced(h0 + eh * dh, cp0) - reduced(h0 - eh * dh, cp0)) / (2 * eh)
    fd_cp = (reduced(h0, cp0 + ec * dcp) - reduced(h0, cp0 - ec * dcp)) / (2 * ec)
    assert W0 > 0
    assert abs(fd_h - tot_h @ dh) < 1e-5 * abs(fd_h)
    assert abs(fd_cp - tot_cp @ dcp) < 1e-5 * max(abs(fd_cp), 1e-12)


def _solver_case(spec, uamp, method="skyline", leaf=256):
    """K of a deformed state, a right-hand side, the device solution and the host (SuperLU) solution refined with the same K."""
    import scipy.sparse.linalg as spla
    from goldfish_amd import _lib, _solver
    A = arrays_from_spec(spec)
    D = _lib.DeviceModel(A)
    D.set_thickness(np.full(A.total_cp, spec.h_th))
    D.set_u(G.smooth_displacement(spec, uamp * spec.h_th))
    D.assemble(_lib.ASM_R | _lib.ASM_K)
    K = D.csr(_lib.MAT_K).tocsc()
    b = -D.residual()
    S = _solver.DeviceSolver(D, coords=np.stack([A.cp_hom[f] / A.weights for f in range(3)], 1), method=method, leaf=leaf)
    assert S.method == method
    x = S.solve(b)
    lu = spla.splu(K)
    xh = lu.solve(b)
    for _ in range(3):                                            # the same refinement on the host side
        xh = xh + lu.solve(b - K @ xh)
    info, rr = S.info(), S.rel_residual
    res_host = np.linalg.norm(b - K @ xh) / np.linalg.norm(b)
    # how far two backward-stable solutions of THIS system lie apart: SuperLU with another column ordering, refined the same way
    lu2 = spla.splu(K, permc_spec="NATURAL" if A.ndof < 20000 else "MMD_ATA")
    xh2 = lu2.solve(b)
    for _ in range(3):
        xh2 = xh2 + lu2.solve(b - K @ xh2)
    self_err = _rel(xh2, xh)
    # a second right-hand side and a re-factorisation after a state change reuse the handle
    D.set_u(G.smooth_displacement(spec, 0.5 * uamp * spec.h_th))
    D.assemble(_lib.ASM_R | _lib.ASM_K)
    S.refactor()
    K2, b2 = D.csr(_lib.MAT_K).tocsc(), -D.residual()
    x2 = S.solve(b2)
    r2 = np.linalg.norm(b2 - K2 @ x2) / np.linalg.norm(b2)
    S.close()
    D.close()
    return _rel(x, xh), np.linalg.norm(b - K @ x) / np.linalg.norm(b), rr, r2, info, A.ndof, res_host, self_err


def test_device_linear_solver_against_superlu():
    """N1: block-banded L D L^T factorisation, substitutions and iterative refinement on the device
    (csrc/gf_solver.hip, no library dependency) against scipy SuperLU: a 2-patch T-beam, C2 (4-patch T-beam, 10.5 k dofs) and
    a 6 x 6-patch non-matching shell with 79 k dofs at a deformed state; residuals at round-off."""
    for spec, tol in ((G.tbeam_2patch(6), 1e-9), (G.tbeam_4patch(), 1e-9), (G.synthetic_shell(6, 6, nel=24, p=3, jitter=2), 1e-9)):
        err, res, rr, r2, info, ndof, res_host, self_err = _solver_case(spec, 0.5)
        print("device solver: %d dofs, half bandwidth %d, %.2f GB, err vs SuperLU %.2e (SuperLU vs SuperLU with another ordering %.2e), residual %.2e "
              "(reported %.2e; SuperLU + refinement %.2e), after refactor %.2e" % (ndof, info["half_bandwidth"], info["device_bytes"] / 1e9, err, self_err, res, rr, res_host, r2))
        # residuals at the round-off floor of this K (what refined SuperLU reaches); solutions as close to SuperLU's as SuperLU's
        # own solutions are to each other when only the elimination order changes (the conditioning of K sets that distance)
        assert res < 10 * res_host + 1e-13 and r2 < 1e-9 and rr < 10 * res + 1e-13
        assert err < max(tol, 20 * self_err), (ndof, err, self_err)


def test_device_linear_solver_nested_dissection_against_superlu():
    """The nested-dissection multifrontal mode (goldfish_amd/_nd.py: recursive coordinate bisection, vertex separators, fronts;
    gfs_create_nd: dense fronts on the skyline solver's tile kernels, extend-add, post-order substitutions) on the same systems as the
    skyline test, with leaves small enough for trees of depth 3 .. 7: same acceptance as the skyline -- residuals at the round-off floor of K,
    solutions as close to SuperLU's as two SuperLU orderings are to each other; re-factorisation after a state change."""
    for spec, leaf in ((G.tbeam_2patch(6), 24), (G.tbeam_4patch(), 96), (G.synthetic_shell(6, 6, nel=24, p=3, jitter=2), 256),
                       (G.wing_16patch_from_interface_data(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_wing_int_data.npz"), allow_pickle=True)), 64)):
        err, res, rr, r2, info, ndof, res_host, self_err = _solver_case(spec, 0.5, method="nd", leaf=leaf)
        print("nested dissection: %d dofs, %.3f GB, err vs SuperLU %.2e (SuperLU vs SuperLU %.2e), residual %.2e (reported %.2e; SuperLU + refinement %.2e), after refactor %.2e"
              % (ndof, info["device_bytes"] / 1e9, err, self_err, res, rr, res_host, r2))
        assert res < 10 * res_host + 1e-13 and r2 < 1e-9 and rr < 10 * res + 1e-13
        assert err < max(1e-9, 20 * self_err), (ndof, err, self_err)


def test_device_solver_general_mode_in_both_factorisations():
    """gfs_set_general (a K that is not symmetric: the load stiffness of a follower pressure) in the skyline AND in the nested-dissection mode:
    factors of the symmetric part, refinement against K (gfs_solve) or K^T (gfs_solve_transposed) through the device-built reverse index --
    both systems of the pressurised ring at hoop strain 0.3 against the assembled matrix, and the two must differ."""
    from goldfish_amd import _solver
    from goldfish_amd.nonmatching_opt import NonMatchingOpt
    nm = NonMatchingOpt.from_spec(G.pressurised_tube(pressure=3.0e6, E=1.0e9))
    nm.solve_nonlinear_nonmatching_problem(rtol=1e-9, max_it=30)
    nm._assemble(3)
    K = nm.dRIGAduIGA()
    assert abs(K - K.T).max() > 1e-8 * abs(K).max()
    b = np.random.default_rng(5).standard_normal(nm.vec_iga_dof)
    w = np.concatenate([sp_.cp_hom_flat()[:, 3] for sp_ in nm.splines])
    X = np.stack([nm.cp_iga[f] / w for f in range(3)], 1)
    for method, kw in (("skyline", {}), ("nd", dict(leaf=48))):
        S = _solver.DeviceSolver(nm.dev, coords=X, method=method, general=True, **kw)
        x = S.solve(b)
        assert _rel(K @ x, b) < 1e-8 and S.backward_error < 1e-12, (method, S.rel_residual, S.backward_error)
        xt = S.solve(b, transpose=True)
        assert _rel(K.T @ xt, b) < 1e-8 and _rel(K @ xt, b) > 1e-6, method
        S.close()
    # a symmetric-mode handle on the same matrix factors the lower triangle only: its refinement still runs against K itself, the transposed solve is the plain one
    S = _solver.DeviceSolver(nm.dev, coords=X, method="skyline")
    assert np.array_equal(S.solve(b), S.solve(b, transpose=True))
    S.close()


def test_device_solver_chain_variants_and_prepared_storage():
    """Round 5: the factorisation's chain in its forms -- diagonal tiles factored by the workgroup that completes them or in launches of their own (GF_SOLVER_FUSE_DIAG),
    panel groups with and without sub-groups (GF_SOLVER_SUBGROUP), sub-groups in two launches or per block column (GF_SOLVER_BLOCKCHAIN), Schur blocks first written by the wide update or cleared and accumulated (GF_SOLVER_LAZY_S), wide updates in 64 x 64 tiles or 128 x 128 macro tiles (GF_SOLVER_MACRO_ROWS) -- gives the same solution to round-off in the skyline, the level-batched and the large-front paths; and
    gfs_prepare_refactor: the factors are gone at once (a solve refuses), the next factorisation finds its storage cleared and returns the same bits."""
    from goldfish_amd import _solver
    from goldfish_amd.nonmatching_opt import NonMatchingOpt
    nm = NonMatchingOpt.from_spec(G.tbeam_4patch())
    nm.update_uIGA(G.smooth_displacement(G.tbeam_4patch(), 0.5 * G.tbeam_4patch().h_th))
    nm._assemble(3)
    K = nm.dRIGAduIGA()
    b = np.random.default_rng(3).standard_normal(nm.vec_iga_dof)
    w = np.concatenate([sp_.cp_hom_flat()[:, 3] for sp_ in nm.splines])
    X = np.stack([nm.cp_iga[f] / w for f in range(3)], 1)
    for method, kw in (("skyline", {}), ("nd", dict(leaf=96)), ("nd", dict(leaf=400)), ("nd-large", dict(leaf=96)), ("nd-large", dict(leaf=400))):
        ref = None
        for env in ({}, {"GF_SOLVER_LAZY_S": "0"}, {"GF_SOLVER_LAZY_S": "0", "GF_SOLVER_BLOCKCHAIN": "0"}, {"GF_SOLVER_BLOCKCHAIN": "3"}, {"GF_SOLVER_BLOCKCHAIN": "3", "GF_SOLVER_SUBGROUP": "3", "GF_SOLVER_PANEL_W": "7"}, {"GF_SOLVER_BLOCKCHAIN": "0"},
                    {"GF_SOLVER_BLOCKCHAIN": "0", "GF_SOLVER_FUSE_DIAG": "0"}, {"GF_SOLVER_BLOCKCHAIN": "0", "GF_SOLVER_SUBGROUP": "2"}, {"GF_SOLVER_SUBGROUP": "0"}, {"GF_SOLVER_SUBGROUP": "2"}, {"GF_SOLVER_SUBGROUP": "3", "GF_SOLVER_PANEL_W": "7"},
                    {"GF_SOLVER_MACRO_ROWS": "2"}, {"GF_SOLVER_MACRO_ROWS": "3", "GF_SOLVER_PANEL_W": "5"}):      # the large fronts' wide updates in 128 x 128 macro tiles (odd and even numbers of block rows)
            if method == "nd-large":                         # every front through the large fronts' kernels (factorisation: per front on streams; substitutions: per block-column group)
                os.environ["GF_SOLVER_FUSE_MAX_BLK"] = "2"
                os.environ["GF_SOLVER_BATCH_BLK"] = "2"
            os.environ.update(env)
            try:
                S = _solver.DeviceSolver(nm.dev, coords=X, method=method.split("-")[0], **kw)
            finally:
                for k_ in list(env) + ["GF_SOLVER_FUSE_MAX_BLK", "GF_SOLVER_BATCH_BLK"]:
                    os.environ.pop(k_, None)
            x = S.solve(b)
            assert S.backward_error < 1e-12 and _rel(K @ x, b) < 1e-7, (method, kw, env)
            if ref is None:
                ref = x
                S.prepare()
                with pytest.raises(RuntimeError, match="no factorisation"):
                    S.solve(b)
                S.refactor()
                assert np.array_equal(S.solve(b), ref), (method, kw)
                S.refactor()                                 # and without the preparation again
                assert np.array_equal(S.solve(b), ref), (method, kw)
            else:
                assert _rel(x, ref) < 1e-9, (method, kw, env)
            S.close()


def test_device_solver_several_right_hand_sides_in_one_call():
    """gfs_solve_multi (round-3 verdict, next 5): the adjoints of several functionals share K^T -- three right-hand sides in one call equal the three single solves
    bit for bit in both factorisation modes (nested dissection: groups of three right-hand sides share one pass over the factors, every tile entry multiplied into
    three sums in the order of the single solve; the groups on their own streams), through NonMatchingOpt.solve_K with a (k, ndof) array, and more right-hand sides
    than workspaces (10 > 8) in two rounds."""
    from goldfish_amd import _solver
    from goldfish_amd.nonmatching_opt import NonMatchingOpt
    nm = NonMatchingOpt.from_spec(G.tbeam_4patch())
    nm.update_uIGA(G.smooth_displacement(G.tbeam_4patch(), 0.5 * G.tbeam_4patch().h_th))
    nm._assemble(3)
    K = nm.dRIGAduIGA()
    B = np.random.default_rng(9).standard_normal((10, nm.vec_iga_dof))
    w = np.concatenate([sp_.cp_hom_flat()[:, 3] for sp_ in nm.splines])
    X = np.stack([nm.cp_iga[f] / w for f in range(3)], 1)
    # "nd-large": the same fronts through the kernels of the LARGE fronts (per block-column group; GF_SOLVER_FUSE_MAX_BLK: test switch of the library)
    for method, kw in (("skyline", {}), ("nd", dict(leaf=96)), ("nd-large", dict(leaf=96))):
        if method == "nd-large":
            os.environ["GF_SOLVER_FUSE_MAX_BLK"] = "2"
        try:
            S = _solver.DeviceSolver(nm.dev, coords=X, method=method.split("-")[0], **kw)
        finally:
            os.environ.pop("GF_SOLVER_FUSE_MAX_BLK", None)
        single = np.stack([S.solve(b) for b in B[:3]])
        multi = S.solve_multi(B[:3])
        assert np.array_equal(single, multi), method
        assert S.rel_residuals.shape == (3,) and S.backward_error < 1e-12
        allx = S.solve_multi(B)
        assert np.array_equal(allx[:3], multi) and max(_rel(K @ x, b) for x, b in zip(allx, B)) < 1e-7      # floor: eps cond(K) for random right-hand sides
        S.close()
    xk = nm.solve_K(B[:2])
    assert xk.shape == (2, nm.vec_iga_dof) and _rel(K @ xk[1], B[1]) < 1e-7
    nm.linear_solver = "host"
    assert _rel(nm.solve_K(B[:2]), xk) < 1e-5


def test_device_solver_is_the_default_newton_and_adjoint_path():
    """solve_nonlinear / solve_linear run on the device solver by default (GOLDFISH/operations/disp_imop.py:38-44, 130-142);
    ``linear_solver = "host"`` (SuperLU on a copy of K) gives the same Newton solution and adjoint."""
    from goldfish_amd.nonmatching_opt import NonMatchingOpt
    spec = G.tbeam_2patch(6)
    nm_d, nm_h = NonMatchingOpt.from_spec(spec), NonMatchingOpt.from_spec(spec)
    assert nm_d.linear_solver == "device"
    nm_h.linear_solver = "host"
    _, ud = nm_d.solve_nonlinear_nonmatching_problem(rtol=1e-8, max_it=30)
    _, uh = nm_h.solve_nonlinear_nonmatching_problem(rtol=1e-8, max_it=30)
    assert nm_d.newton_relative_residual < 1e-8 and nm_h.newton_relative_residual < 1e-8
    assert _rel(ud, uh) < 1e-7
    lam = np.random.default_rng(0).standard_normal(nm_d.vec_iga_dof)
    nm_d._assemble(3)
    nm_h._assemble(3)
    assert _rel(nm_d.solve_K(lam), nm_h.solve_K(lam)) < 1e-7
    assert nm_d._dsolver.rel_residual < 1e-8


@pytest.mark.parametrize("p", [2, 3, 4])
def test_stress_forms_parity(oracle_lib, p):
    """gf_stress_forms (kl_stress_kernel) against the oracle's complex-step forms: KS and power integrands, Cauchy and
    2nd Piola-Kirchhoff measure, top / bottom / middle surface, with and without Dirichlet zeroing."""
    from goldfish_amd import _lib
    from oracle.oracle_py import Oracle
    spec = G.scordelis_lo_9patch(p, nels=[2, 1, 2, 3, 2, 3, 2, 1, 2])
    rng = np.random.default_rng(3 + p)
    th = [spec.h_th * rng.uniform(0.8, 1.2, q.ncp) for q in spec.patches]
    A = arrays_from_spec(spec, th)
    h, u = np.concatenate(th), 2e-2 * rng.standard_normal(A.ndof)
    O, D = Oracle(A, thickness=h, u=u), _lib.DeviceModel(A)
    D.set_thickness(h)
    D.set_u(u)
    n = len(spec.patches)
    vmax = D.stress_forms(1, 1.0, np.ones(n), 1, 0, gradients=False)["vmax"]
    assert _rel(vmax, O.stress_forms(1, 1.0, np.ones(n), 1.0, 0)["vmax"]) < 1e-11
    for mode, rho, surf, measure, bcs in ((0, 4.0 / vmax.max(), 1, 0, True), (1, 6.0, -1, 0, False), (1, 3.0, 0, 0, True), (1, 5.0, 1, 1, True)):
        m_list = vmax * (1.0 + 0.05 * np.arange(n))
        Fd, Fo = D.stress_forms(mode, rho, m_list, surf, measure, apply_bcs=bcs), O.stress_forms(mode, rho, m_list, float(surf), measure, apply_bcs=bcs)
        assert _rel(Fd["I"], Fo["I"]) < 1e-11, (mode, surf, measure)
        assert _rel(Fd["dIdu"], Fo["dIdu"]) < 1e-10, (mode, surf, measure)
        assert _rel(Fd["dIdh"], Fo["dIdh"]) < 1e-10 * max(1.0, float(surf != 0)) + (1e-300 if surf else 0.0), (mode, surf, measure)
        for f in range(3):
            assert _rel(Fd["dIdcp"][f], Fo["dIdcp"][f]) < 1e-10, (mode, surf, measure, f)
        if surf == 0:
            assert np.abs(Fd["dIdh"]).max() == 0.0          # mid-surface stress does not depend on the thickness
    with pytest.raises(ValueError):
        D.stress_forms(1, 2.0, np.ones(n - 1))
    with pytest.raises(ValueError):
        D.stress_forms(2, 2.0, np.ones(n))
    D.close()


@pytest.mark.parametrize("method", ["KS", "pnorm", "induced power"])
def test_max_vm_stress_operation_and_comp(method):
    """MaxvMStressExOperation: global aggregate close to (and for these rho above a fraction of) the true maximum, and
    its gradients wrt u, CP and h against central differences of the operation itself; MaxvMStressComp partials."""
    from goldfish_amd.operations.max_vmstress_exop import MaxvMStressExOperation
    from goldfish_amd.om_comps import MaxvMStressComp, om
    spec, th, nm = _problem()
    rng = np.random.default_rng(5)
    nm.update_uIGA(1e-3 * rng.standard_normal(nm.vec_iga_dof))
    probe = MaxvMStressExOperation(nm, rho=1.0, method="pnorm")
    _, true_max = probe.compute_max_vM()
    rho = 8.0 / true_max if method == "KS" else 8.0
    op = MaxvMStressExOperation(nm, rho=rho, m=true_max, surf="top", method=method)
    val = op.max_vM_stress_global()
    assert 0.2 * true_max < val < 5.0 * true_max
    u0 = nm.u_iga.copy()

    def fd(setter, x0, g, eps):
        d = rng.standard_normal(x0.size)
        setter(x0 + eps * d); vp = op.max_vM_stress_global()
        setter(x0 - eps * d); vm = op.max_vM_stress_global()
        setter(x0)
        return (vp - vm) / (2 * eps), float(g @ d)

    num, ana = fd(nm.update_uIGA, u0, op.dmax_vMduIGA_global(apply_bcs=False), 1e-7)
    assert abs(num - ana) < 1e-5 * abs(ana), ("u", num, ana)
    for i, field in enumerate(nm.opt_field):
        cp0 = nm.get_init_CPIGA()[i].copy()
        num, ana = fd(lambda v, f=field: nm.update_CPIGA(v, f), cp0, op.dmax_vMdCPIGA_global(field), 1e-5)
        assert abs(num - ana) < 1e-5 * abs(ana), ("cp", field, num, ana)
    h0 = np.concatenate(nm.h_th)
    num, ana = fd(nm.update_h_th_IGA, h0, op.dmax_vMdh_th_global(), 1e-5 * spec.h_th)
    assert abs(num - ana) < 1e-5 * abs(ana), ("h", num, ana)
    comp = MaxvMStressComp(nonmatching_opt=nm, rho=rho, m=true_max, method=method)
    comp.init_parameters()
    prob = om.Problem(model=comp)
    prob.setup()
    prob.run_model()
    free = np.ones(nm.vec_iga_dof, bool)
    free[nm.zero_dofs] = False
    errs = prob.check_partials(compact_print=False, free_mask=free, step=1e-6)
    assert max(errs.values()) < 1e-4, errs       # the CP_IGA1 directional derivative is a small difference of large terms


def test_moving_intersection_residual_derivative(oracle_lib):
    """N3: NonMatchingOpt.dRIGAdxi (gf_penalty_dxi -> pen_dxi_kernel, dual-number pass per mortar vertex) against the analytic
    mixed derivative of the interface energy (autograd through the rational basis at the vertices and the tangent stencil,
    tests/torch_model.py: penalty_residual_dxi) to 1e-9, against central differences of the ORACLE's residual with the interface
    rebuilt at perturbed parametric coordinates, and against the reference-style forward-difference check dRIGAdxi_FD."""
    from goldfish_amd.model import Interface
    from oracle.oracle_py import Oracle
    spec, th, nm = _problem()
    rng = np.random.default_rng(9)
    # a curved, non-uniformly spaced intersection strictly inside both patches exercises every term
    n = 6
    t = np.linspace(0, 1, n)
    xa = np.stack([0.45 + 0.1 * t + 0.03 * np.sin(3 * t), 0.1 + 0.8 * t ** 1.3], 1)
    xb = np.stack([0.3 + 0.2 * t ** 2, 0.15 + 0.7 * t], 1)
    nm.mortar_nels = [n - 1]
    nm.mortar_meshes_setup(nm.mapping_list, [[xa, xb]], nm.penalty_coefficient)
    nm.update_uIGA(2e-2 * rng.standard_normal(nm.vec_iga_dof))
    nm.create_diff_intersections()
    assert nm.xi_size == 4 * n
    xi0 = nm.cpiga2xi.xi_flat_global.copy()
    J = nm.dRIGAdxi().toarray()
    assert J.shape == (nm.vec_iga_dof, 4 * n) and np.abs(J[nm.zero_dofs]).max() == 0.0
    # analytic: d/dxi of dE/dU of the interface energy
    from tests.torch_model import penalty_residual_dxi
    A0 = nm._arrays()
    Jt = penalty_residual_dxi(nm.splines, nm.cp_off, nm.mapping_list[0][0], nm.mapping_list[0][1], A0.weights, np.stack(nm.cp_iga, 1), nm.u_iga,
                              xa, xb, (A0.if_alpha[0], A0.if_alpha[1]), A0.if_wt, nm.zero_dofs)
    assert np.abs(Jt - J).max() < 1e-9 * np.abs(J).max(), np.abs(Jt - J).max() / np.abs(J).max()

    def oracle_residual(xi):
        itf = Interface(nm.mapping_list[0][0], nm.mapping_list[0][1], xi[:2 * n].reshape(-1, 2), xi[2 * n:].reshape(-1, 2))
        A = nm._arrays()
        A2 = type(A)(nm.splines, nm.E, nm.nu, [list(r.body_force) for r in nm.residuals], [itf],
                     [(A.if_alpha[0], A.if_alpha[1])], [])
        O = Oracle(A2, thickness=np.concatenate(nm.h_th), u=nm.u_iga)
        for f in range(3):
            O.set_cp(f, nm.cp_iga[f])
        return O.residual()

    eps = 1e-6
    for k in rng.choice(4 * n, 10, replace=False):
        e = np.zeros(4 * n)
        e[k] = eps
        fd = (oracle_residual(xi0 + e) - oracle_residual(xi0 - e)) / (2 * eps)
        assert np.abs(fd - J[:, k]).max() < 2e-6 * np.abs(J).max(), k
    Jfd = nm.dRIGAdxi_FD(xi0, h=1e-7)
    assert np.abs(Jfd - J).max() < 1e-4 * np.abs(J).max()


def test_disp_states_with_moving_intersections_component():
    """DispMintStatesComp (om_comps/disp_states_mi_comp.py:6-117): partials of R(u; CP_IGA, int_para) through the
    OpenMDAO protocol -- K du, dR/dCP dcp and dR/dxi dxi against central differences of apply_nonlinear (the device model
    is re-created for every perturbed set of parametric coordinates) -- and the reverse products against the forward ones."""
    from goldfish_amd.om_comps import DispMintStatesComp, om
    spec, th, nm = _problem()
    rng = np.random.default_rng(12)
    n = 5
    t = np.linspace(0, 1, n)
    xa = np.stack([0.45 + 0.1 * t + 0.03 * np.sin(3 * t), 0.1 + 0.8 * t ** 1.3], 1)
    xb = np.stack([0.3 + 0.2 * t ** 2, 0.15 + 0.7 * t], 1)
    nm.mortar_nels = [n - 1]
    nm.mortar_meshes_setup(nm.mapping_list, [[xa, xb]], nm.penalty_coefficient)
    nm.create_diff_intersections()
    comp = DispMintStatesComp(nonmatching_opt=nm)
    comp.init_parameters()
    prob = om.Problem(model=comp)
    prob.setup()
    prob["displacements"] = 1e-2 * rng.standard_normal(nm.vec_iga_dof)
    free = np.ones(nm.vec_iga_dof, bool)
    free[np.asarray(nm.dev and nm.zero_dofs)] = False
    errs = prob.check_partials(compact_print=False, free_mask=free, step=1e-6)
    assert max(errs.values()) < 2e-5, errs
    assert ("displacements", "int_para") in errs
    # reverse mode: <lam, J dx> == <J^T lam, dx> for the xi block
    op = comp.disp_mint_state_imop
    op.linearize()                                                   # the products below are those of the current state
    lam, dxi = rng.standard_normal(nm.vec_iga_dof), rng.standard_normal(nm.xi_size)
    dres = np.zeros(nm.vec_iga_dof)
    din = [np.zeros(s) for s in comp.input_cp_shapes] + [dxi]
    op.apply_linear_fwd(din, None, dres)
    back = [np.zeros(s) for s in comp.input_cp_shapes] + [np.zeros(nm.xi_size)]
    op.apply_linear_rev(back, None, lam)
    assert abs(lam @ dres - back[-1] @ dxi) < 1e-10 * abs(lam @ dres)


def test_plate_thickness_optimisation_end_to_end():
    """examples/plate_thickness_opt.py: the reference's plate thickness-optimisation problem driven through the
    operations (Newton states, adjoint with K^T, dR/dh, functionals on the device).  The reduced gradient matches central
    differences; SLSQP lowers the internal energy at constant volume and tapers the plate from the clamped edge to the tip."""
    import importlib.util
    spec_ = importlib.util.spec_from_file_location("plate_thickness_opt", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "plate_thickness_opt.py"))
    mod = importlib.util.module_from_spec(spec_)
    spec_.loader.exec_module(mod)
    out = mod.run(maxiter=40, verbose=False)
    prob = out["problem"]
    rng = np.random.default_rng(0)
    h = 1e-2 * rng.uniform(0.8, 1.2, 6)
    g, d = prob.gradient(h), rng.standard_normal(6)
    eps = 1e-6
    fd = (prob.objective(h + eps * d) - prob.objective(h - eps * d)) / (2 * eps)
    assert abs(fd - g @ d) < 1e-5 * abs(fd), (fd, g @ d)
    assert out["w1"] < 0.8 * out["w0"]
    assert abs(out["v1"] - out["v0"]) < 1e-8 * out["v0"]
    assert np.all(np.diff(out["h"]) <= 1e-9) and out["h"][0] > 1.5 * out["h"][-1]       # thick at the clamp, thin at the loaded edge
    # the same problem read from the reference's own files (IGES surfaces + .npz intersection cache): same optimum
    out2 = mod.run(maxiter=40, verbose=False, from_files=True)
    assert np.abs(out2["h"] - out["h"]).max() < 2e-5 and abs(out2["w1"] - out["w1"]) < 1e-4 * out["w1"]


def test_incremental_assembly_and_functional_cache():
    """NonMatchingOpt assembles an output once per state and only what is missing (RIGA, then K, then dR/dCP, then dR/dh
    give the same matrices as one fused pass), evaluates a functional once per state, and re-assembles after any update."""
    from goldfish_amd import _lib
    spec, th, nm = _problem()
    rng = np.random.default_rng(4)
    nm.update_uIGA(1e-2 * rng.standard_normal(nm.vec_iga_dof))
    calls = []
    dev = nm.dev
    orig_asm, orig_fun = dev.assemble, dev.functionals
    dev.assemble = lambda flags, **kw: (calls.append(("asm", flags)), orig_asm(flags, **kw))[1]
    dev.functionals = lambda **kw: (calls.append(("fun",)), orig_fun(**kw))[1]
    R = nm.RIGA().copy()
    K = nm.dRIGAduIGA().copy()
    C1 = nm.dRIGAdCPIGA(1).copy()
    H = nm.dRIGAdh_th().copy()
    assert np.array_equal(nm.RIGA(), R)                                   # no new assembly
    assert [c[1] for c in calls if c[0] == "asm"] == [_lib.ASM_R, _lib.ASM_K, _lib.ASM_DRDCP, _lib.ASM_DRDH]
    from goldfish_amd.operations.int_energy_exop import IntEnergyExOperation
    from goldfish_amd.operations.volume_exop import VolumeExOperation
    w, v = IntEnergyExOperation(nm), VolumeExOperation(nm)
    w.Wint(); w.dWintduIGA(); w.dWintdCPIGA(0); w.dWintdh_th(); v.volume(); v.dvoldh_th()
    assert sum(1 for c in calls if c[0] == "fun") == 1
    spec2, th2, ref = _problem()
    ref.update_uIGA(nm.u_iga)
    ref.dev.assemble(_lib.ASM_ALL)
    # the separate passes give the fused pass's matrices to round-off: a pass without dR/dCP / dR/dh runs leaner kernel instances (p = 4: another path altogether,
    # row records instead of element blocks -- each pass kind on its faster path), whose compiler-scheduled FMA contractions / summation order differ in the last bits
    assert np.abs(ref.dev.residual() - R).max() <= 1e-13 * np.abs(R).max()
    assert abs(ref.dev.csr(_lib.MAT_K) - K).max() <= 1e-13 * abs(K).max()
    assert abs(ref.dRIGAdCPIGA(1) - C1).max() == 0.0 and abs(ref.dRIGAdh_th() - H).max() == 0.0
    n0 = len(calls)
    nm.update_uIGA(nm.u_iga * 1.01)                                       # new state: everything is stale again
    assert not np.array_equal(nm.RIGA(), R) and len(calls) == n0 + 1
    w.Wint()
    assert sum(1 for c in calls if c[0] == "fun") == 2


@pytest.mark.parametrize("p", [3, 2, 4])
def test_arch_shape_optimisation_known_answer(p):
    """examples/arch_shape_opt.py: the reference's arch demo (demos_om/shape_opt/arch/arch_shape_opt_wint.py prints
    "Maximum F2 ... (reference: 5.4779)") through the device path -- load per unit projected area, dR/dCP, FFD maps,
    adjoint.  5.4779 is the analytical optimum rise of a parabolic arch of span 10 (0.547789 L); start: rise 3."""
    import importlib.util
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec_ = importlib.util.spec_from_file_location("arch_shape_opt", os.path.join(here, "examples", "arch_shape_opt.py"))
    mod = importlib.util.module_from_spec(spec_)
    spec_.loader.exec_module(mod)
    out = mod.run(verbose=False, p=p)                       # finer p = 3 mesh: 5.47787
    assert abs(out["h0"] - 3.0) < 1e-12
    # every product on the path is fixed-order (no atomics), so the optimisation is reproducible on one path.  Round 2 iterated the
    # state to rtol 1e-10, below the residual's evaluation floor (4e-5 |R_0|: each extra Newton step added that noise to u), and the
    # p = 2 optimum then depended on the assembly path (5.4673 vs 5.4784); with the reference's rtol 1e-3 the state is the clean
    # first Newton step and every degree lands inside the 5e-3 window.
    assert abs(out["h1"] - 5.4779) < 5e-3, out["h1"]
    assert out["problem"].nm.newton_converged
    assert out["w1"] < 0.8 * out["w0"]
    if p != 3:
        return
    prob = out["problem"]
    rng = np.random.default_rng(1)
    d = prob.d0 * (1 + 0.05 * rng.standard_normal(prob.d0.size))
    g, v = prob.gradient(d), rng.standard_normal(prob.d0.size)
    eps = 1e-5
    fd = (prob.objective(d + eps * v) - prob.objective(d - eps * v)) / (2 * eps)
    assert abs(fd - g @ v) < 1e-4 * abs(fd), (fd, g @ v)          # W ~ 1e-8: the difference quotient carries ~1e-5 of noise


def test_total_derivative_through_a_moving_intersection():
    """N3 end to end (the chain of demos_om/shape_opt_mint): the web of the T-beam slides in x by s, the intersection follows
    through CPIGA2Xi (xi(CP) implicit), the coupling moves with it, the state is re-solved.  The total derivative of the
    internal energy  dW/ds = dW/dCP . c' - lam^T [ dR/dCP . c' + dR/dxi . xi' ],  xi' = -(dRxi/dxi)^-1 dRxi/dCP . c',
    K^T lam = dW/du  (device: dR/dCP, dR/dxi, K; host: CPIGA2Xi) against central differences of the whole pipeline."""
    from goldfish_amd.nonmatching_opt import NonMatchingOptFFD
    from goldfish_amd.operations.disp_mi_imop import DispMintImOpeartion
    from goldfish_amd.operations.int_energy_exop import IntEnergyExOperation
    spec = G.tbeam_2patch(4)
    nm = NonMatchingOptFFD.from_spec(spec)
    nm.set_shopt_surf_inds_FFD([0], [0, 1])
    nm.create_diff_intersections()
    c2x = nm.cpiga2xi
    disp, wint = DispMintImOpeartion(nm), IntEnergyExOperation(nm)
    n0 = spec.patches[0].ncp
    cp0 = nm.get_init_CPIGA()[0].copy()
    dcp = np.zeros(cp0.size)
    dcp[n0:] = spec.patches[1].cp_hom_flat()[:, 3]                       # d(homogeneous x of the web)/ds

    def pipeline(s):
        cp = cp0 + s * dcp
        nm.update_CPIGA(cp, 0)
        c2x.update_CPs(cp, 0)
        xi = c2x.solve_xi(c2x.xi_flat_global)
        nm.update_xi(xi)
        nm.update_transfer_matrices()
        nm.update_uIGA(disp.solve_nonlinear(max_it=30, rtol=1e-8))
        return xi

    xi = pipeline(0.0)
    n = c2x.diff_int_num_pts[0]
    assert np.abs(xi[:2 * n].reshape(-1, 2)[:, 0] - 0.5).max() < 1e-10
    disp.linearize()
    lam = disp.solve_linear_rev(wint.dWintduIGA(apply_bcs=True), np.zeros(nm.vec_iga_dof))
    dxi = -np.linalg.solve(c2x.dRdxi(xi), c2x.dRdCP(xi, 0, coo=False) @ dcp)
    assert np.abs(dxi[:2 * n].reshape(-1, 2)[:, 0] - 0.5).max() < 1e-8       # flange x in [-1, 1]: d(xi_u)/ds = 1/2
    back = [np.zeros(cp0.size), np.zeros(nm.xi_size)]
    disp.apply_linear_rev(back, None, lam)                               # (dR/dCP)^T lam, (dR/dxi)^T lam
    total = wint.dWintdCPIGA(0) @ dcp - back[0] @ dcp - back[1] @ dxi
    eps = 1e-4
    pipeline(eps)
    wp = wint.Wint()
    pipeline(-eps)
    wm = wint.Wint()
    fd = (wp - wm) / (2 * eps)
    assert abs(back[1] @ dxi) > 1e-3 * abs(total)                        # the moving-intersection term matters here
    assert abs(fd - total) < 1e-5 * abs(fd), (fd, total, back[1] @ dxi)


@pytest.mark.parametrize("p", [2, 3, 4])
def test_shape_regularisation_parity(oracle_lib, p):
    """gf_shape_regu (kl_pointfun_kernel<P, 2>) against the oracle: value and the gradient wrt all three coordinate fields."""
    from goldfish_amd import _lib
    from oracle.oracle_py import Oracle
    spec = G.scordelis_lo_9patch(p, nels=[2, 1, 2, 3, 2, 3, 2, 1, 2])
    rng = np.random.default_rng(20 + p)
    A = arrays_from_spec(spec)
    O, D = Oracle(A, thickness=np.full(A.total_cp, spec.h_th), u=np.zeros(A.ndof)), _lib.DeviceModel(A)
    cp = np.stack(A.cp_hom, 1)
    for field in (2, 1):
        cp0 = cp[:, field] + 0.05 * rng.standard_normal(A.total_cp)
        coef = rng.uniform(0.5, 2.0, len(spec.patches))
        Fd, Fo = D.shape_regu(field, cp0, coef), O.shape_regu(field, cp0, coef)
        assert abs(Fd["value"] - Fo["value"]) < 1e-11 * abs(Fo["value"])
        for f in range(3):
            assert _rel(Fd["dcp"][f], Fo["dcp"][f]) < 1e-10, (field, f)
    assert D.shape_regu(2, cp[:, 2], np.ones(len(spec.patches)))["value"] == 0.0        # no change of shape, no penalty
    with pytest.raises(ValueError):
        D.shape_regu(3, cp[:, 2], np.ones(len(spec.patches)))
    D.close()


def test_regularised_energy_operation_and_comp():
    """IntEnergyReguExOperation / IntEnergyReguComp (demos_om/shape_opt/eVTOL): W_int + regularisation, dW/dCP against central
    differences, dW/du and dW/dh unchanged."""
    from goldfish_amd.operations.int_energy_exop import IntEnergyExOperation
    from goldfish_amd.operations.int_energy_regu_exop import IntEnergyReguExOperation
    from goldfish_amd.om_comps import IntEnergyReguComp, om
    spec, th, nm = _problem()
    rng = np.random.default_rng(8)
    nm.update_uIGA(1e-3 * rng.standard_normal(nm.vec_iga_dof))
    op, base = IntEnergyReguExOperation(nm, regu_para=1.0e6), IntEnergyExOperation(nm)
    assert op.Wint() == base.Wint()                                   # initial shape: the term vanishes
    cp2 = nm.get_init_CPIGA()[2].copy()
    nm.update_CPIGA(cp2 + 0.02 * rng.standard_normal(cp2.size), 2)
    assert op.Wint() > base.Wint() * (1 + 1e-6)
    assert np.array_equal(op.dWintduIGA(), base.dWintduIGA()) and np.array_equal(op.dWintdh_th(), base.dWintdh_th())
    for i, field in enumerate(nm.opt_field):
        x0 = nm.cp_iga[field][nm._shopt_cols[i]].copy()
        g, d = op.dWintdCPIGA(field), rng.standard_normal(x0.size)
        eps = 1e-6
        nm.update_CPIGA(x0 + eps * d, field); wp = op.Wint()
        nm.update_CPIGA(x0 - eps * d, field); wm = op.Wint()
        nm.update_CPIGA(x0, field)
        assert abs((wp - wm) / (2 * eps) - g @ d) < 1e-6 * abs(g @ d), field
    comp = IntEnergyReguComp(nonmatching_opt=nm, regu_para=1.0e6)
    comp.init_parameters()
    prob = om.Problem(model=comp)
    prob.setup()
    prob.run_model()
    free = np.ones(nm.vec_iga_dof, bool)
    free[np.asarray(nm.dev and nm.zero_dofs)] = False             # d w_int / d displacements with its Dirichlet rows zeroed (int_energy_regu_comp.py:90 of the reference)
    assert max(prob.check_partials(compact_print=False, free_mask=free).values()) < 1e-5


def test_volume_of_a_patch_subset_and_wint_regu_terms(oracle_lib):
    """VolumeExOperation(vol_surf_inds = subset) (volume_exop.py:9-27: the other patches contribute Constant(0) dx) and
    IntEnergyExOperation(wint_regu = [...]) (int_energy_exop.py:15-32) -- both raised NotImplementedError until round 4.  The subset volume
    against the oracle's per-patch model, its partials against central differences; the regularised energy against IntEnergyReguExOperation."""
    from goldfish_amd.operations.volume_exop import VolumeExOperation
    from goldfish_amd.operations.int_energy_exop import IntEnergyExOperation, ShapeRegu
    from goldfish_amd.operations.int_energy_regu_exop import IntEnergyReguExOperation
    from goldfish_amd.om_comps import VolumeComp, IntEnergyComp, om
    from oracle.oracle_py import Oracle
    spec, th, nm = _problem()
    rng = np.random.default_rng(11)
    full, sub = VolumeExOperation(nm), VolumeExOperation(nm, vol_surf_inds=[1])
    f = nm.functionals()
    assert abs(f["volume_patch"].sum() - f["volume"]) < 1e-13 * f["volume"] and abs(f["Wint_patch"].sum() - f["Wint"]) <= 1e-13 * abs(f["Wint"]) + 1e-300
    one = G.ProblemSpec(patches=[spec.patches[1]], E=spec.E, nu=spec.nu, h_th=spec.h_th, interfaces=[], body_force=[spec.body_force[1]], point_loads=[]) \
        if hasattr(G, "ProblemSpec") else None
    if one is not None:                                               # patch 1 alone in the oracle: its volume is the subset's
        Ao = arrays_from_spec(one, [th[1]])
        Vo = Oracle(Ao, thickness=th[1], u=np.zeros(Ao.ndof)).functionals()["volume"]
        assert abs(sub.volume() - Vo) < 1e-11 * Vo
    assert 0.0 < sub.volume() < full.volume() and abs(sub.volume() + VolumeExOperation(nm, [0]).volume() - full.volume()) < 1e-12 * full.volume()
    gh = sub.dvoldh_th()
    assert np.all(gh[:nm.cp_off[1]] == 0.0) and np.array_equal(gh[nm.cp_off[1]:], full.dvoldh_th()[nm.cp_off[1]:])
    for i, field in enumerate(nm.opt_field):
        x0 = nm.cp_iga[field][nm._shopt_cols[i]].copy()
        g, d = sub.dvoldCPIGA(field), rng.standard_normal(x0.size)
        eps = 1e-6
        nm.update_CPIGA(x0 + eps * d, field); vp = sub.volume()
        nm.update_CPIGA(x0 - eps * d, field); vm = sub.volume()
        nm.update_CPIGA(x0, field)
        assert abs((vp - vm) / (2 * eps) - g @ d) < 1e-6 * max(abs(g @ d), 1e-12), field
    comp = VolumeComp(nonmatching_opt=nm, vol_surf_inds=[1])
    comp.init_parameters()
    prob = om.Problem(model=comp)
    prob.setup(); prob.run_model()
    assert max(prob.check_partials(compact_print=False).values()) < 1e-5
    # ---- wint_regu: the eVTOL demo's term on every patch through the base operation = IntEnergyReguExOperation
    nm.update_uIGA(1e-3 * rng.standard_normal(nm.vec_iga_dof))
    ref = IntEnergyReguExOperation(nm, regu_para=1.0e6)
    op = IntEnergyExOperation(nm, wint_regu=[ShapeRegu(c, field=2) for c in ref.regu_para_full])
    only1 = IntEnergyExOperation(nm, wint_regu=[None, ShapeRegu(ref.regu_para_full[1], field=2)])
    cp2 = nm.get_init_CPIGA()[2].copy()
    nm.update_CPIGA(cp2 + 0.02 * rng.standard_normal(cp2.size), 2)
    base = IntEnergyExOperation(nm)
    assert abs(op.Wint() - ref.Wint()) < 1e-12 * abs(ref.Wint()) and base.Wint() < only1.Wint() < op.Wint()
    for field in nm.opt_field:
        assert _rel(op.dWintdCPIGA(field), ref.dWintdCPIGA(field)) < 1e-12
    g1 = only1.dWintdCPIGA(2) - base.dWintdCPIGA(2)
    assert np.all(g1[:nm.cp_off[1]] == 0.0) and np.abs(g1[nm.cp_off[1]:]).max() > 0.0       # patch 0 carries no term
    with pytest.raises(TypeError):
        IntEnergyExOperation(nm, wint_regu=[object(), None])
    # ---- c_regu of ComplianceExOperation (compliance_exop.py:9-28: raised NotImplementedError until round 5): the same terms added to the compliance form
    from goldfish_amd.operations.compliance_exop import ComplianceExOperation
    forces = rng.standard_normal((nm.num_splines, 3))
    c0 = ComplianceExOperation(nm, forces)
    c1 = ComplianceExOperation(nm, forces, c_regu=[None, ShapeRegu(ref.regu_para_full[1], field=2, cp0=cp2[nm.cp_off[1]:nm.cp_off[2]])])     # P^0 = the control net before the perturbation (as only1's)
    assert abs((c1.cpl() - c0.cpl()) - (only1.Wint() - base.Wint())) < 1e-10 * abs(only1.Wint())
    assert _rel(c1.dcpldCPIGA(2) - c0.dcpldCPIGA(2), g1) < 1e-10 and np.array_equal(c1.dcplduIGA(), c0.dcplduIGA())
    nm.update_CPIGA(cp2, 2)
    comp = IntEnergyComp(nonmatching_opt=nm)
    comp.init_parameters(wint_regu=[None, ShapeRegu(ref.regu_para_full[1], field=2)])
    prob = om.Problem(model=comp)
    prob.setup(); prob.run_model()
    free = np.ones(nm.vec_iga_dof, bool)
    free[np.asarray(nm.dev and nm.zero_dofs)] = False
    assert max(prob.check_partials(compact_print=False, free_mask=free).values()) < 5e-5      # central differences of the component's run_model


def _dxi_rev_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    q_ = _dxi_rev_case(comm=dist)
    if rank == 0:
        q.put(q_)
    dist.barrier()
    dist.destroy_process_group()


def _dxi_rev_case(comm=None):
    from goldfish_amd.nonmatching_opt import NonMatchingOpt
    spec = G.tbeam_2patch(4)
    nm = NonMatchingOpt.from_spec(spec, comm=comm)
    nm.set_shopt_surf_inds([0], [[0, 1]])
    nm.create_diff_intersections()
    rng = np.random.default_rng(21)
    nm.update_uIGA(1e-2 * rng.standard_normal(nm.vec_iga_dof))
    lam = rng.standard_normal(nm.vec_iga_dof)
    out = dict(rev=nm.dRIGAdxi_rev(lam), lam=lam)
    out["J"] = nm.dRIGAdxi()              # sharded: the owner of side A evaluates the blocks, every rank assembles the same matrix (round 5)
    return out


def test_reverse_product_with_dRdxi_on_the_device_and_on_shards():
    """(dR/dxi)^T lam formed on the device (gf_penalty_dxi_rev: the per-vertex blocks never reach the host) equals the product with the assembled dR/dxi, Dirichlet
    rows zeroed; the same on a problem sharded over two ranks (each owns one patch of the T-beam, the other is its ghost: every rank contracts its own rows, one
    all-reduce of 6 doubles per mortar vertex) -- round-3 verdict, missing 4 / next 7."""
    import torch.multiprocessing as mp
    one = _dxi_rev_case()
    ref = one["J"].T @ one["lam"]
    assert np.abs(ref).max() > 0 and _rel(one["rev"], ref) < 1e-12
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dxi_rev_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    two = q.get(timeout=600)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    assert _rel(two["rev"], ref) < 1e-12
    assert abs(two["J"] - one["J"]).max() <= 1e-13 * abs(one["J"]).max()        # dR/dxi as a matrix on the sharded problem (raised NotImplementedError until round 5)


def test_update_transfer_matrices_patches_the_moved_interface_in_place():
    """update_transfer_matrices (nonmatching_opt.py:567-600) while every mortar vertex stays in its knot spans: the SAME device model (same handle, the direct
    solver keeps its symbolic phase) with re-evaluated vertex tables gives the residual and tangent of a freshly created model; a vertex that crosses a knot line
    forces a new model (round-3 verdict, missing 4: every call re-created the model, 2.7 s at C4 size)."""
    from goldfish_amd.nonmatching_opt import NonMatchingOpt
    spec = G.tbeam_2patch(4)
    rng = np.random.default_rng(31)

    def make():
        nm = NonMatchingOpt.from_spec(spec)
        nm.set_shopt_surf_inds([0], [[0, 1]])
        nm.create_diff_intersections()
        nm.update_uIGA(u)
        return nm
    u = 1e-2 * rng.standard_normal(3 * sum(p.ncp for p in spec.patches))
    nm = make()
    dev0 = nm.dev
    nm._assemble(3)
    nm.solve_K(-nm.dev.residual())                                       # builds the device solver on this pattern
    ds0 = nm._dsolver
    xi0 = nm.xi_flat.copy()
    xi = xi0.copy()
    n = nm.cpiga2xi.diff_int_num_pts[0]
    xi[0:2 * n:2] += 0.01                                                # side A slides by 0.01 in u: inside its span (the flange has two spans in u, [0, 0.5) and [0.5, 1]; the vertices sit at 0.5)
    nm.update_xi(xi)
    nm.update_transfer_matrices()
    assert nm.dev is dev0                                                 # patched in place
    R1, K1 = nm.RIGA(), nm.dRIGAduIGA()
    x1 = nm.solve_K(-R1)
    assert nm._dsolver is ds0                                             # the solver was re-factored, not rebuilt
    ref = make()
    ref.update_xi(xi)
    ref._drop_device()
    from goldfish_amd.model import Interface
    a, b = ref.mapping_list[0]
    ref.interfaces[0] = Interface(a, b, xi[:2 * n].reshape(-1, 2), xi[2 * n:].reshape(-1, 2))
    R2, K2 = ref.RIGA(), ref.dRIGAduIGA()
    assert _rel(R1, R2) < 1e-13 and abs(K1 - K2).max() < 1e-13 * abs(K2).max()
    assert _rel(K2 @ x1, -R2) < 1e-8
    xi = xi0.copy()
    xi[0:2 * n:2] -= 0.3                                                  # across the knot line at 0.5: new support windows
    nm.update_xi(xi)
    nm.update_transfer_matrices()
    assert nm.dev is not dev0
    assert np.isfinite(nm.RIGA()).all()


def test_shape_opt_mint_group_wired_like_the_reference_demo():
    """The reference's moving-intersection ShapeOptGroup (demos_om/shape_opt_mint/T-beam/T_beam_2patch_shopt_mi.py:18-305): IndepVarComp -> CPSurfAlignComp ->
    CPSurfOrderElevationComp -> CPSurfKnotRefinementComp -> CPIGA2XiComp -> DispMintStatesComp -> IntEnergyComp, connected by absolute names, with the demo's
    design (a bilinear net for the x coordinate of the web, aligned in both directions: ONE design variable, the position of the web under the flange) -- through
    om.Problem (the protocol stand-in where OpenMDAO is not installed).  The total derivative of the internal energy (states xi(CP) and u(CP, xi), dR/dxi in the
    adjoint) against central differences of run_model; the optimum of the symmetric load case is the centred web (round-3 verdict, missing 4)."""
    from goldfish_amd.nonmatching_opt import NonMatchingOpt
    from goldfish_amd.om_comps import CPIGA2XiComp, DispMintStatesComp, IntEnergyComp, om
    from goldfish_amd.om_comps.surf_comps import CPSurfAlignComp, CPSurfOrderElevationComp, CPSurfKnotRefinementComp
    from goldfish_amd.utils.bsp_utils import CPSurfDesign2Analysis

    class ShapeOptGroup(om.Group):

        def initialize(self):
            self.options.declare('nonmatching_opt')
            self.options.declare('cpdesign2analysis')
            for name, default in (('cp_design_name_pre', 'CP_design'), ('cp_coarse_name_pre', 'CP_coarse'), ('cp_order_ele_name_pre', 'CP_order_ele'),
                                  ('cp_analysis_name_pre', 'CP_analysis'), ('int_name', 'int_para'), ('disp_name', 'displacements'), ('int_energy_name', 'int_E')):
                self.options.declare(name, default=default)

        def init_parameters(self):
            for k in ('nonmatching_opt', 'cpdesign2analysis', 'cp_design_name_pre', 'cp_coarse_name_pre', 'cp_order_ele_name_pre', 'cp_analysis_name_pre',
                      'int_name', 'disp_name', 'int_energy_name'):
                setattr(self, k, self.options[k])
            self.opt_field = self.nonmatching_opt.opt_field
            self.init_cp_design = self.cpdesign2analysis.init_cp_design
            self.names = {pre: [getattr(self, pre) + str(f) for f in self.opt_field]
                          for pre in ('cp_design_name_pre', 'cp_coarse_name_pre', 'cp_order_ele_name_pre', 'cp_analysis_name_pre')}

        def setup(self):
            nm, d2a = self.nonmatching_opt, self.cpdesign2analysis
            inputs_comp = om.IndepVarComp()
            for i in range(len(self.opt_field)):
                inputs_comp.add_output(self.names['cp_design_name_pre'][i], shape=len(self.init_cp_design[i]), val=self.init_cp_design[i])
            self.add_subsystem('inputs_comp', inputs_comp)
            comps = [('CP_design_align_comp', CPSurfAlignComp(cpdesign2analysis=d2a, input_cp_design_name_pre=self.cp_design_name_pre, output_cp_coarse_name_pre=self.cp_coarse_name_pre)),
                     ('CP_order_ele_comp', CPSurfOrderElevationComp(cpdesign2analysis=d2a, input_cp_coarse_name_pre=self.cp_coarse_name_pre, output_cp_order_ele_name_pre=self.cp_order_ele_name_pre)),
                     ('CP_knot_refine_comp', CPSurfKnotRefinementComp(cpdesign2analysis=d2a, input_cp_order_ele_name_pre=self.cp_order_ele_name_pre, output_cp_fine_name_pre=self.cp_analysis_name_pre)),
                     ('CPIGA2Xi_comp', CPIGA2XiComp(nonmatching_opt=nm, input_cp_iga_name_pre=self.cp_analysis_name_pre, output_xi_name=self.int_name))]
            for name, c in comps:
                c.init_parameters()
                self.add_subsystem(name, c)
            disp = DispMintStatesComp(nonmatching_opt=nm, input_cp_iga_name_pre=self.cp_analysis_name_pre, input_xi_name=self.int_name, output_u_name=self.disp_name)
            disp.init_parameters(save_files=False, nonlinear_solver_rtol=1e-9, nonlinear_solver_max_it=20)
            self.add_subsystem('disp_states_comp', disp)
            wint = IntEnergyComp(nonmatching_opt=nm, input_cp_iga_name_pre=self.cp_analysis_name_pre, input_u_name=self.disp_name, output_wint_name=self.int_energy_name)
            wint.init_parameters()
            self.add_subsystem('internal_energy_comp', wint)
            for i in range(len(self.opt_field)):
                d, c, o, a = (self.names[k][i] for k in ('cp_design_name_pre', 'cp_coarse_name_pre', 'cp_order_ele_name_pre', 'cp_analysis_name_pre'))
                self.connect('inputs_comp.' + d, 'CP_design_align_comp.' + d)
                self.connect('CP_design_align_comp.' + c, 'CP_order_ele_comp.' + c)
                self.connect('CP_order_ele_comp.' + o, 'CP_knot_refine_comp.' + o)
                for tgt in ('CPIGA2Xi_comp', 'disp_states_comp', 'internal_energy_comp'):
                    self.connect('CP_knot_refine_comp.' + a, tgt + '.' + a)
            self.connect('CPIGA2Xi_comp.' + self.int_name, 'disp_states_comp.' + self.int_name)
            self.connect('disp_states_comp.' + self.disp_name, 'internal_energy_comp.' + self.disp_name)
            self.add_design_var('inputs_comp.' + self.names['cp_design_name_pre'][0], lower=-1.0, upper=1.0)
            self.add_objective('internal_energy_comp.' + self.int_energy_name, scaler=1e2)

    spec = G.tbeam_2patch(4, load=(0.0, 0.0, -1.0), tip_load=0.0)
    spec.body_force = [[0.0, 0.0, -1.0], [0.0, 0.0, 0.0]]               # the flange carries the load: symmetric about x = 0
    nm = NonMatchingOpt.from_spec(spec)
    nm.set_shopt_surf_inds(opt_field=[0], shopt_surf_inds=[[1]])
    nm.create_diff_intersections()
    d2a = CPSurfDesign2Analysis(nm.preprocessor, opt_field=[0], shopt_surf_inds=[[1]])
    lin = [[0.0, 0.0, 1.0, 1.0], [0.0, 0.0, 1.0, 1.0]]
    d2a.set_init_knots_by_field([[[1, 1]]], [[lin]])
    web = spec.patches[1]
    d2a.set_order_elevation_by_field([[[web.p, web.q]]], [[[[0.0] * (web.p + 1) + [1.0] * (web.p + 1), [0.0] * (web.q + 1) + [1.0] * (web.q + 1)]]])
    d2a.set_knot_refinement()
    d2a.get_init_cp_coarse()
    d2a.set_cp_align(field=0, align_dir_list=[[0, 1]])
    assert d2a.init_cp_design[0].shape == (1,)                              # one design variable: the web's x
    A = d2a.knot_refine_operator_list[0].tocsr() @ d2a.order_ele_operator_list[0].tocsr() @ d2a.cp_coarse_align_deriv_list[0].tocsr()
    assert A.shape == (web.ncp, 1) and np.abs(A.toarray() - 1.0).max() < 1e-12        # the whole chain moves every analysis control point of the web by the design value
    model = ShapeOptGroup(nonmatching_opt=nm, cpdesign2analysis=d2a)
    model.init_parameters()
    prob = om.Problem(model=model)
    prob.setup()
    x0 = float(d2a.init_cp_design[0][0])
    of, wrt = 'internal_energy_comp.int_E', 'inputs_comp.CP_design0'
    vals = {}
    for s in (0.3, 0.3 + 1e-5, 0.3 - 1e-5, -0.3, 0.0):
        prob.set_val(wrt, [x0 + s])
        prob.run_model()
        vals[s] = float(np.ravel(prob.get_val(of))[0])
        if s == 0.3:
            tot = float(np.ravel(prob.compute_totals(of=[of], wrt=[wrt])[(of, wrt)])[0])
    fd = (vals[0.3 + 1e-5] - vals[0.3 - 1e-5]) / 2e-5
    assert abs(tot - fd) < 2e-5 * abs(fd), (tot, fd)
    assert abs(vals[0.3] - vals[-0.3]) < 1e-6 * vals[0.3] and vals[0.0] < vals[0.3]      # symmetric load case: the centred web is the optimum


def test_moving_intersection_optimisation_finds_the_symmetric_optimum():
    """examples/tbeam_moving_intersection.py (set-up of demos_om/shape_opt_mint/T-beam): the web starts 0.4 off centre under a
    load that is symmetric about the flange's centre line; with the intersection moving along (xi(CP), dR/dxi in the total
    derivative) SLSQP brings it back to the centre, the optimum known by symmetry."""
    import importlib.util
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec_ = importlib.util.spec_from_file_location("tbeam_mint", os.path.join(here, "examples", "tbeam_moving_intersection.py"))
    mod = importlib.util.module_from_spec(spec_)
    spec_.loader.exec_module(mod)
    out = mod.run(verbose=False)
    assert abs(out["s1"]) < 2e-3, out["s1"]
    assert out["w1"] < out["w0"]


def test_plain_c_consumer_of_the_c_abi(tmp_path):
    """tests/c_abi/c_consumer.c: a C program (no Python, no torch) creates a model through gf_model_desc, assembles, checks the
    load resultant and the symmetry of K through gf_apply, and the error convention."""
    import subprocess
    from tests.test_host_logic import _build_c_consumer
    out = subprocess.run([_build_c_consumer(tmp_path)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr + out.stdout
    assert "c consumer ok" in out.stdout


def _radial(nm, u, patch, xi):
    """Radial displacement and radius at a parametric point of a patch (the homogeneous dofs rationalised)."""
    from goldfish_amd.splines import basis_ders, find_span
    P = nm.splines[patch]
    su, sv = find_span(P.n_u, P.p, P.knots[0], xi[0]), find_span(P.n_v, P.q, P.knots[1], xi[1])
    Nu, Nv = basis_ders(su, xi[0], P.p, P.knots[0], 0)[0], basis_ders(sv, xi[1], P.q, P.knots[1], 0)[0]
    off, H = int(nm.cp_off[patch]), P.cp_hom_flat()
    num, X, W = np.zeros(3), np.zeros(3), 0.0
    for jv in range(P.q + 1):
        for ju in range(P.p + 1):
            a = P.flat(su - P.p + ju, sv - P.q + jv)
            N = Nu[ju] * Nv[jv]
            num += N * u[3 * (off + a):3 * (off + a) + 3]
            X += N * H[a, :3]
            W += N * H[a, 3]
    X, U = X / W, num / W
    er = np.array([X[0], X[1], 0.0]) / np.hypot(X[0], X[1])
    return float(U @ er), float(np.hypot(X[0], X[1]))


def test_tube_under_internal_pressure_known_answers():
    """Third reference-derived known answer: the load case of demos_om/shape_opt/tube/tube_shape_opt_wint.py:258-262, 303-324 (follower
    pressure p sqrt(det a / det A) a2, E = 1e12, nu = 0, h = 0.01) on the closed circular ring of four non-matching patches: uniform
    expansion, linearised u_r = p r^2 / (E h (1 + h^2 / 12 r^2) - p r), geometrically exact r (sqrt(1 + 2 p r / (E h)) - 1) (membrane part).
    The load stiffness makes K non-symmetric: K^T products and solves are checked on the way."""
    from goldfish_amd.nonmatching_opt import NonMatchingOpt
    pts = [(0, (0.0, 0.5)), (0, (0.5, 0.5)), (1, (0.3, 0.2)), (2, (0.9, 0.9)), (3, (0.5, 0.1))]
    h = 0.01
    # the demo's numbers: strain 1e-10, the linear regime
    nm = NonMatchingOpt.from_spec(G.pressurised_tube(pressure=1.0, E=1.0e12))
    u = nm.solve_linear_nonmatching_problem()
    lin = 1.0 / (1.0e12 * h * (1 + h * h / 12) - 1.0)
    for patch, xi in pts:
        assert abs(_radial(nm, u, patch, xi)[0] / lin - 1.0) < 5e-4, (patch, xi)
    assert not nm.symmetric_K
    # finite deformation: hoop strain 0.1 and 0.3
    for press in (1.0e6, 3.0e6):
        nm = NonMatchingOpt.from_spec(G.pressurised_tube(pressure=press, E=1.0e9))
        _, u = nm.solve_nonlinear_nonmatching_problem(rtol=1e-9, max_it=30)
        assert nm.newton_converged
        exact = np.sqrt(1 + 2 * press / (1.0e9 * h)) - 1
        for patch, xi in pts:
            assert abs(_radial(nm, u, patch, xi)[0] / exact - 1.0) < 5e-4, (press, patch, xi)
    # non-symmetric tangent: transposed products and solves are those of K^T
    K = nm.dRIGAduIGA()
    asym = abs(K - K.T).max() / abs(K).max()
    assert asym > 1e-8
    rng = np.random.default_rng(3)
    x = rng.standard_normal(nm.vec_iga_dof)
    y = np.zeros(nm.vec_iga_dof)
    from goldfish_amd import _lib
    nm.dev.apply(_lib.MAT_K, x, y, transpose=True)
    assert _rel(y, K.T @ x) < 1e-12
    b = rng.standard_normal(nm.vec_iga_dof)
    assert _rel(K.T @ nm.solve_K(b, transpose=True), b) < 1e-8 and _rel(K @ nm.solve_K(b), b) < 1e-8
    # ... and they ran on the device: factors of the symmetric part, refinement against K / K^T (gfs_set_general), accepted by the backward error
    assert nm._dsolver is not None and nm._dsolver.general and getattr(nm, "_dsolver_failed_version", None) != nm._k_version
    assert nm.linear_solve_backward_error <= nm.linear_solve_rtol
    xt = nm._dsolver.solve(b, transpose=True)
    assert _rel(K.T @ xt, b) < 1e-8 and _rel(K @ xt, b) > 1e-6          # the transposed system, not the plain one


def test_tube_shape_optimisation_rounds_the_cross_section():
    """examples/tube_shape_opt.py (set-up of demos_om/shape_opt/tube/tube_shape_opt_wint.py): follower pressure in dR/dCP and in a
    non-symmetric K^T adjoint solve; the adjoint gradient against central differences of the whole pipeline, then the optimisation:
    the bending-dominated start (radius varying by 6 %) moves towards the membrane state of the circle."""
    import importlib.util
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec_ = importlib.util.spec_from_file_location("tube_shape_opt", os.path.join(here, "examples", "tube_shape_opt.py"))
    mod = importlib.util.module_from_spec(spec_)
    spec_.loader.exec_module(mod)
    # gradient check on a softer tube (E = 1e7: strains of 1e-5, state iterated to 1e-10): with the demo's E = 1e12 the strains are 1e-10 and
    # the energy -- a sum of squares of differences of metrics -- carries 1e-3 of round-off in a difference quotient
    prob = mod.ReducedShapeProblem(mod.build(E=1.0e7), newton_rtol=1e-8)
    rng = np.random.default_rng(2)
    x = prob.x0 * (1 + 0.02 * rng.standard_normal(prob.x0.size))
    g, v = prob.gradient(x), rng.standard_normal(prob.x0.size)
    eps = 1e-6
    fd = (prob.objective(x + eps * v) - prob.objective(x - eps * v)) / (2 * eps)
    assert abs(fd - g @ v) < 1e-5 * abs(fd), (fd, g @ v)
    out = mod.run(verbose=False)
    assert out["w1"] < 0.2 * out["w0"], (out["w0"], out["w1"])
    assert out["r1"][0] < 0.4 * out["r0"][0], (out["r0"], out["r1"])
    assert abs(out["r1"][1] / out["r0"][1] - 1.0) < 0.05


def test_shape_opt_group_wired_like_the_reference_demo():
    """The reference's ShapeOptGroup (demos_om/shape_opt/T-beam/T_beam_shape_opt_wint.py:12-218; the tube demo wires the same components):
    IndepVarComp -> CPFFDesign2FullComp -> CPFFD2SurfComp -> CPFE2IGAComp -> DispStatesComp -> IntEnergyComp / VolumeComp, the design dofs also into
    CPFFDPinComp / CPFFDReguComp, connected by absolute names per optimised field, design variables / constraints / objective as in the demo, through
    ``om.Problem`` (openmdao.api when installed, else the protocol stand-in).  Model: the pressurised ring with two optimised fields -- the follower
    pressure makes the adjoint of DispStatesComp a K^T solve.  Totals of the objective and the volume wrt both design fields against central
    differences of run_model; the linear constraint components against their matrices."""
    import importlib.util
    from goldfish_amd.om_comps import DispStatesComp, IntEnergyComp, VolumeComp, om
    from goldfish_amd.om_comps.ffd_comps import CPFE2IGAComp, CPFFD2SurfComp, CPFFDesign2FullComp, CPFFDPinComp, CPFFDReguComp
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec_ = importlib.util.spec_from_file_location("tube_shape_opt", os.path.join(here, "examples", "tube_shape_opt.py"))
    mod = importlib.util.module_from_spec(spec_)
    spec_.loader.exec_module(mod)
    nm = mod.build(E=1.0e7)                                    # strains of 1e-5: difference quotients of the energy are meaningful (see the tube test)
    nm.set_shopt_pin_CPFFD(pin_dir0=[0, 1], pin_side0=[[0], [0]])      # per optimised field: one face of the FFD block keeps its place
    nm.set_shopt_regu_CPFFD()

    class ShapeOptGroup(om.Group):

        def initialize(self):
            self.options.declare('nonmatching_opt_ffd')
            for name, default in (('cpffd_design_name_pre', 'CP_design_FFD'), ('cpffd_full_name_pre', 'CP_FFD'), ('cpsurf_fe_name_pre', 'CPS_FE'),
                                  ('cpsurf_iga_name_pre', 'CPS_IGA'), ('disp_name', 'displacements'), ('int_energy_name', 'int_E'),
                                  ('cpffd_pin_name_pre', 'CP_FFD_pin'), ('cpffd_regu_name_pre', 'CP_FFD_regu'), ('volume_name', 'volume')):
                self.options.declare(name, default=default)

        def init_parameters(self):
            o = self.options
            self.nm = o['nonmatching_opt_ffd']
            self.opt_field = self.nm.opt_field
            self.init_cpffd = self.nm.shopt_init_cpffd_design
            self.names = {k: [o[k] + str(f) for f in self.opt_field] for k in ('cpffd_design_name_pre', 'cpffd_full_name_pre', 'cpsurf_fe_name_pre',
                                                                              'cpsurf_iga_name_pre', 'cpffd_pin_name_pre', 'cpffd_regu_name_pre')}

        def setup(self):
            o, nm = self.options, self.nm
            inputs_comp = om.IndepVarComp()
            for i in range(len(self.opt_field)):
                inputs_comp.add_output(self.names['cpffd_design_name_pre'][i], shape=self.init_cpffd[i].size, val=self.init_cpffd[i])
            self.add_subsystem('inputs_comp', inputs_comp)
            comps = {
                'CPFFDDesign2Full_comp': CPFFDesign2FullComp(nonmatching_opt_ffd=nm, input_cpffd_design_name_pre=o['cpffd_design_name_pre'], output_cpffd_full_name_pre=o['cpffd_full_name_pre']),
                'CPFFD2FE_comp': CPFFD2SurfComp(nonmatching_opt_ffd=nm, input_cpffd_name_pre=o['cpffd_full_name_pre'], output_cpsurf_name_pre=o['cpsurf_fe_name_pre']),
                'CPFE2IGA_comp': CPFE2IGAComp(nonmatching_opt=nm, input_cp_fe_name_pre=o['cpsurf_fe_name_pre'], output_cp_iga_name_pre=o['cpsurf_iga_name_pre']),
                'disp_states_comp': DispStatesComp(nonmatching_opt=nm, input_cp_iga_name_pre=o['cpsurf_iga_name_pre'], output_u_name=o['disp_name']),
                'internal_energy_comp': IntEnergyComp(nonmatching_opt=nm, input_cp_iga_name_pre=o['cpsurf_iga_name_pre'], input_u_name=o['disp_name'], output_wint_name=o['int_energy_name']),
                'CPFFD_pin_comp': CPFFDPinComp(nonmatching_opt_ffd=nm, input_cpffd_design_name_pre=o['cpffd_design_name_pre'], output_cppin_name_pre=o['cpffd_pin_name_pre']),
                'CPFFD_regu_comp': CPFFDReguComp(nonmatching_opt_ffd=nm, input_cpffd_design_name_pre=o['cpffd_design_name_pre'], output_cpregu_name_pre=o['cpffd_regu_name_pre']),
                'volume_comp': VolumeComp(nonmatching_opt=nm, input_cp_iga_name_pre=o['cpsurf_iga_name_pre'], output_vol_name=o['volume_name']),
            }
            for name, comp in comps.items():
                if name == 'disp_states_comp':
                    comp.init_parameters(save_files=False, nonlinear_solver_rtol=1e-8)
                else:
                    comp.init_parameters()
                self.add_subsystem(name, comp)
            self.comps = comps
            N = self.names
            for i in range(len(self.opt_field)):
                d, full, fe, iga = N['cpffd_design_name_pre'][i], N['cpffd_full_name_pre'][i], N['cpsurf_fe_name_pre'][i], N['cpsurf_iga_name_pre'][i]
                self.connect('inputs_comp.' + d, 'CPFFDDesign2Full_comp.' + d)
                self.connect('CPFFDDesign2Full_comp.' + full, 'CPFFD2FE_comp.' + full)
                self.connect('CPFFD2FE_comp.' + fe, 'CPFE2IGA_comp.' + fe)
                for user in ('disp_states_comp', 'internal_energy_comp', 'volume_comp'):
                    self.connect('CPFE2IGA_comp.' + iga, user + '.' + iga)
                self.connect('inputs_comp.' + d, 'CPFFD_pin_comp.' + d)
                self.connect('inputs_comp.' + d, 'CPFFD_regu_comp.' + d)
            self.connect('disp_states_comp.' + o['disp_name'], 'internal_energy_comp.' + o['disp_name'])
            for i in range(len(self.opt_field)):
                self.add_design_var('inputs_comp.' + N['cpffd_design_name_pre'][i], lower=-1.2, upper=1.2)
                self.add_constraint('CPFFD_pin_comp.' + N['cpffd_pin_name_pre'][i], equals=nm.shopt_pin_vals[i])
                self.add_constraint('CPFFD_regu_comp.' + N['cpffd_regu_name_pre'][i], lower=1.0e-1)
            self.add_constraint('volume_comp.' + o['volume_name'], equals=1.0)
            self.add_objective('internal_energy_comp.' + o['int_energy_name'], scaler=1e6)

    model = ShapeOptGroup(nonmatching_opt_ffd=nm)
    model.init_parameters()
    prob = om.Problem(model=model)
    prob.setup()
    rng = np.random.default_rng(11)
    wrt = ['inputs_comp.' + n for n in model.names['cpffd_design_name_pre']]
    d0 = [np.asarray(v, float) * (1 + 0.01 * rng.standard_normal(np.size(v))) for v in model.init_cpffd]
    for n, v in zip(wrt, d0):
        prob.set_val(n, v)
    prob.run_model()
    assert not nm.symmetric_K
    of = ['internal_energy_comp.int_E', 'volume_comp.volume']
    tot = prob.compute_totals(of=of, wrt=wrt)
    for k, n in enumerate(wrt):
        v = rng.standard_normal(d0[k].size)
        f = []
        for sgn in (1, -1):
            prob.set_val(n, d0[k] + sgn * 1e-6 * v)
            prob.run_model()
            f.append([float(np.ravel(prob.get_val(o))[0]) for o in of])
        prob.set_val(n, d0[k])
        fd = (np.array(f[0]) - np.array(f[1])) / 2e-6
        for r, o in enumerate(of):
            t = float(np.asarray(tot[(o, n)]).reshape(-1) @ v)
            assert abs(t - fd[r]) < 2e-5 * max(abs(fd[r]), 1e-12), (o, n, t, fd[r])
    # the linear constraint components: values = their matrices times the design dofs
    prob.run_model()
    for i in range(len(nm.opt_field)):
        pin = np.ravel(prob.get_val('CPFFD_pin_comp.' + model.names['cpffd_pin_name_pre'][i]))
        assert _rel(pin, nm.shopt_dcppindcpffd[i] @ d0[i]) < 1e-12 if pin.size else True
        regu = np.ravel(prob.get_val('CPFFD_regu_comp.' + model.names['cpffd_regu_name_pre'][i]))
        assert _rel(regu, nm.shopt_dcpregudcpffd[i] @ d0[i]) < 1e-12


def _thickness_opt_group_totals(comm=None, device=0):
    """Totals of the reference's ThicknessOptGroup wired through om.Problem and their central differences; ``comm``: torch.distributed -> patch-sharded problem.
    The reference's ThicknessOptGroup (demos_om/thickness_opt/plate/plate_const_th_opt_wint.py:12-124): IndepVarComp -> HthMapComp ->
    DispStatesComp -> IntEnergyComp / VolumeComp connected by absolute names, design variable / constraint / objective as in the demo,
    run through ``om.Problem`` -- the real openmdao.api when it is installed, the protocol stand-in (goldfish_amd/om_shim.py: Group,
    connect, reverse-mode compute_totals, OpenMDAO's size / declaration checks) in the build image, where no OpenMDAO wheel exists.
    Total derivatives of the objective and of the constraint wrt the per-patch thicknesses against central differences of run_model."""
    from goldfish_amd.nonmatching_opt import NonMatchingOpt
    from goldfish_amd.om_comps import DispStatesComp, IntEnergyComp, VolumeComp, om
    from goldfish_amd.om_comps.ffd_comps.hth_map_comp import HthMapComp

    class ThicknessOptGroup(om.Group):

        def initialize(self):
            self.options.declare('nonmatching_opt')
            self.options.declare('h_th_name_design', default='thickness')
            self.options.declare('h_th_name_full', default='thickness_full')
            self.options.declare('disp_name', default='displacements')
            self.options.declare('int_energy_name', default='w_int')
            self.options.declare('volume_name', default='volume')

        def init_parameters(self):
            self.nonmatching_opt = self.options['nonmatching_opt']
            self.h_th_name_design, self.h_th_name_full = self.options['h_th_name_design'], self.options['h_th_name_full']
            self.disp_name, self.volume_name, self.int_energy_name = self.options['disp_name'], self.options['volume_name'], self.options['int_energy_name']
            self.num_splines = self.nonmatching_opt.num_splines
            self.init_h_th = [np.average(h) for h in self.nonmatching_opt.init_h_th_list]

        def setup(self):
            nm = self.nonmatching_opt
            inputs_comp = om.IndepVarComp()
            inputs_comp.add_output(self.h_th_name_design, shape=self.num_splines, val=self.init_h_th)
            self.add_subsystem('inputs_comp', inputs_comp)
            self.h_th_map_comp = HthMapComp(nonmatching_opt=nm, input_h_th_name_design=self.h_th_name_design, output_h_th_name_full=self.h_th_name_full)
            self.h_th_map_comp.init_parameters()
            self.add_subsystem('h_th_map_comp', self.h_th_map_comp)
            self.disp_states_comp = DispStatesComp(nonmatching_opt=nm, input_h_th_name=self.h_th_name_full, output_u_name=self.disp_name)
            self.disp_states_comp.init_parameters(save_files=False, nonlinear_solver_rtol=2e-6)
            self.add_subsystem('disp_states_comp', self.disp_states_comp)
            self.int_energy_comp = IntEnergyComp(nonmatching_opt=nm, input_h_th_name=self.h_th_name_full, input_u_name=self.disp_name, output_wint_name=self.int_energy_name)
            self.int_energy_comp.init_parameters()
            self.add_subsystem('int_energy_comp', self.int_energy_comp)
            self.volume_comp = VolumeComp(nonmatching_opt=nm, input_h_th_name=self.h_th_name_full, output_vol_name=self.volume_name)
            self.volume_comp.init_parameters()
            self.add_subsystem('volume_comp', self.volume_comp)
            self.connect('inputs_comp.' + self.h_th_name_design, 'h_th_map_comp.' + self.h_th_name_design)
            self.connect('h_th_map_comp.' + self.h_th_name_full, 'disp_states_comp.' + self.h_th_name_full)
            self.connect('h_th_map_comp.' + self.h_th_name_full, 'volume_comp.' + self.h_th_name_full)
            self.connect('h_th_map_comp.' + self.h_th_name_full, 'int_energy_comp.' + self.h_th_name_full)
            self.connect('disp_states_comp.' + self.disp_name, 'int_energy_comp.' + self.disp_name)
            self.add_design_var('inputs_comp.' + self.h_th_name_design, lower=4e-3, upper=5e-2, scaler=1e2)
            self.add_constraint('volume_comp.' + self.volume_name, equals=1.0e-2)
            self.add_objective('int_energy_comp.' + self.int_energy_name, scaler=1e3)

    nm = NonMatchingOpt.from_spec(G.plate_6patch(), comm=comm, device=device)
    nm.set_thickness_opt(var_thickness=False)
    model = ThicknessOptGroup(nonmatching_opt=nm)
    model.init_parameters()
    prob = om.Problem(model=model)
    prob.setup()
    h0 = np.array([1.2e-2, 0.9e-2, 1.0e-2, 1.1e-2, 0.8e-2, 1.0e-2])
    prob.set_val('inputs_comp.thickness', h0)
    prob.run_model()
    of, wrt = ['int_energy_comp.w_int', 'volume_comp.volume'], ['inputs_comp.thickness']
    tot = prob.compute_totals(of=of, wrt=wrt)
    J = np.zeros((2, 6))
    for k in range(6):
        f = []
        for sgn in (1, -1):
            h = h0.copy()
            h[k] += sgn * 1e-6
            prob.set_val('inputs_comp.thickness', h)
            prob.run_model()
            f.append([float(np.ravel(prob.get_val(o))[0]) for o in of])
        J[:, k] = (np.array(f[0]) - np.array(f[1])) / 2e-6
    prob.set_val('inputs_comp.thickness', h0)
    prob.run_model()
    T = np.stack([np.asarray(tot[(o, wrt[0])]).reshape(-1) for o in of])
    return dict(T=T, J=J, volume=float(np.ravel(prob.get_val('volume_comp.volume'))[0]), w_int=float(np.ravel(prob.get_val('int_energy_comp.w_int'))[0]),
                u=np.ravel(prob.get_val('disp_states_comp.displacements')).copy(), solver=nm.linear_solver, sharded=nm.sharded,
                device_solver_used=getattr(nm, "_dsolver", None) is not None and getattr(nm, "_dsolver_failed_version", None) is None)


def test_thickness_opt_group_wired_like_the_reference_demo():
    r = _thickness_opt_group_totals()
    for row in range(2):
        assert _rel(r["T"][row], r["J"][row]) < 1e-5, (row, r["T"][row], r["J"][row])
    assert abs(r["volume"] - 1.0e-2) < 2e-3     # unit plate, thickness ~1e-2


def _group_rank_worker(rank, world, port, q, backend):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dev = rank if backend == "nccl" else 0               # nccl (= RCCL): one GPU per rank; gloo: every rank on GPU 0
    if backend == "nccl":
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    r = _thickness_opt_group_totals(comm=dist, device=dev)
    if rank == 0:
        q.put(r)
    dist.barrier()
    dist.destroy_process_group()


def _run_group_ranks(world, backend):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_group_rank_worker, args=(r, world, port, q, backend)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=900)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    return res


def _check_sharded_group(r):
    ref = _thickness_opt_group_totals()                   # the same problem in this process, unsharded
    assert r["sharded"] and r["solver"] == "device" and r["device_solver_used"]       # Newton steps and adjoints on the device: every rank factors the gathered K
    for row in range(2):
        assert _rel(r["T"][row], r["J"][row]) < 1e-5, (row, r["T"][row], r["J"][row])
        assert _rel(r["T"][row], ref["T"][row]) < 1e-7
    assert abs(r["volume"] - ref["volume"]) < 1e-12 * ref["volume"] and abs(r["w_int"] - ref["w_int"]) < 1e-8 * abs(ref["w_int"])
    assert _rel(r["u"], ref["u"]) < 1e-7


def test_thickness_opt_group_on_a_sharded_problem_two_ranks_one_gpu():
    """NonMatchingOpt(comm = torch.distributed): the reference's ThicknessOptGroup with the patches sharded over two ranks (gloo; both on the one GPU of the
    box) -- states, functionals, adjoint solves and totals equal the unsharded run's (round-3 verdict, next 3)."""
    _check_sharded_group(_run_group_ranks(2, "gloo"))


def test_thickness_opt_group_on_a_sharded_problem_over_rccl():
    import torch
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip("one GPU visible: RCCL needs at least two")
    _check_sharded_group(_run_group_ranks(min(n, 3), "nccl"))


def _dist_solver_worker(rank, world, port, q, backend):
    import torch
    import torch.distributed as dist
    from goldfish_amd.nonmatching_opt import NonMatchingOpt
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dev = rank if backend == "nccl" else 0
    if backend == "nccl":
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    spec = G.synthetic_shell(4, 4, nel=10, p=3, jitter=2)
    nm = NonMatchingOpt.from_spec(spec, comm=dist, device=dev)
    nm.sharded_solver = "distributed"
    nm.update_uIGA(G.smooth_displacement(spec, 0.5 * spec.h_th))
    nm._assemble(3)
    B = np.random.default_rng(4).standard_normal((2, nm.vec_iga_dof))
    x0 = nm.solve_K(B[0])
    ds = nm._dsolver
    out = dict(x0=x0, rr0=nm.linear_solve_relative_residual, be0=nm.linear_solve_backward_error, method=ds.method, owner=ds.owner.copy(), nroots=len(ds.roots),
               has_subtrees=ds.A is not None, failed=getattr(nm, "_dsolver_failed_version", None) is not None)
    # a new tangent: refactor (collective), two right-hand sides in one call, a Newton solve of the sharded problem
    nm.update_uIGA(G.smooth_displacement(spec, 0.8 * spec.h_th))
    nm._assemble(3)
    out["X"] = nm.solve_K(B)
    out["K"] = nm.dRIGAduIGA()
    _, u = nm.solve_nonlinear_nonmatching_problem(rtol=1e-8, max_it=30)
    out["u"], out["newton_rr"] = u, nm.newton_relative_residual
    res = [None] * world
    dist.all_gather_object(res, (rank, out["has_subtrees"], out["method"]))
    out["ranks"] = res
    if rank == 0:
        q.put(out)
    dist.barrier()
    dist.destroy_process_group()


def _dist_failure_worker(rank, world, port, q):
    import warnings
    import torch.distributed as dist
    from goldfish_amd import _dsolver
    from goldfish_amd.nonmatching_opt import NonMatchingOpt
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if rank == 1:                                     # this rank's subtree handle "does not fit": the others' do
        orig = _dsolver._Part.__init__

        def failing(self, *a, **k):
            if k.get("row_ok") is not None:
                raise RuntimeError("gfs_create_nd_partial: hipMalloc failed: out of memory (forced by the test)")
            return orig(self, *a, **k)
        _dsolver._Part.__init__ = failing
    spec = G.synthetic_shell(4, 4, nel=10, p=3, jitter=2)
    nm = NonMatchingOpt.from_spec(spec, comm=dist, device=0)
    nm.sharded_solver = "distributed"
    nm.update_uIGA(G.smooth_displacement(spec, 0.5 * spec.h_th))
    nm._assemble(3)
    b = np.random.default_rng(4).standard_normal(nm.vec_iga_dof)
    with warnings.catch_warnings(record=True) as wl:
        warnings.simplefilter("always")
        x = nm.solve_K(b)                              # both ranks must come back: the failing one and the one whose own handle was fine
    out = (rank, [str(w.message)[:120] for w in wl], nm._dsolver is None, getattr(nm, "_dsolver_permanent_failure", None), float(np.abs(nm.dRIGAduIGA() @ x - b).max() / np.abs(b).max()))
    res = [None] * world
    dist.all_gather_object(res, out)
    if rank == 0:
        q.put(res)
    dist.barrier()
    dist.destroy_process_group()


def test_a_failure_of_the_distributed_solver_on_one_rank_is_every_ranks_failure():
    """a rank whose partial handle cannot be created (its GPU is out of memory) used to fall back to the host solve ALONE and met the other ranks
    inside their all-gather.  Two ranks on the one GPU, rank 1's handle creation forced to fail: the outcome is agreed on before the first collective
    (DistributedSolver._raise_together), both ranks latch the permanent failure and solve through the (collective) host path, with the right answer."""
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dist_failure_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=600)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    for rank, msgs, dropped, latched, err in res:
        assert dropped and latched is not None and "out of memory" in latched, (rank, latched)
        assert any("cannot be used for this model" in m for m in msgs), (rank, msgs)
        assert err < 1e-7


def test_distributed_factorisation_over_rccl():
    """The same on one GPU per rank over RCCL (Schur complements and boundary contributions by all_gather_into_tensor, x by all_reduce on device tensors): runs as
    soon as two GPUs are visible."""
    import torch
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip("one GPU visible: RCCL needs at least two")
    test_distributed_factorisation_on_a_sharded_problem(min(n, 3), backend="nccl")


@pytest.mark.parametrize("world", [2, 3])
def test_distributed_factorisation_on_a_sharded_problem(world, backend="gloo"):
    """goldfish_amd/_dsolver.py (SURVEY 8(e) + N1, stage 2): the subtrees of the nested-dissection tree are factored by the ranks that own them (partial handles of
    libgoldfish_solver), their Schur complements all-gathered into stub fronts below the replicated top; forward / backward sweeps in halves with the boundary
    contributions exchanged.  World 2 and 3 over gloo on the one GPU: the solutions solve K x = b of the unsharded matrix, a refactorisation after a new assembly,
    several right-hand sides, and the Newton loop of the sharded problem on top of it."""
    import torch.multiprocessing as mp
    from goldfish_amd.nonmatching_opt import NonMatchingOpt
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dist_solver_worker, args=(r, world, port, q, backend)) for r in range(world)]
    for p in procs:
        p.start()
    r = q.get(timeout=900)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    assert r["method"] == "nd-distributed" and not r["failed"] and all(m == "nd-distributed" for _, _, m in r["ranks"])
    assert sum(1 for _, has, _ in r["ranks"] if has) == world                 # every rank factors subtrees of its own
    assert r["nroots"] >= world and (r["owner"] == -1).sum() >= 1 and set(np.unique(r["owner"])) == set(range(-1, world))
    # the unsharded problem in this process
    spec = G.synthetic_shell(4, 4, nel=10, p=3, jitter=2)
    nm = NonMatchingOpt.from_spec(spec)
    B = np.random.default_rng(4).standard_normal((2, nm.vec_iga_dof))
    nm.update_uIGA(G.smooth_displacement(spec, 0.5 * spec.h_th)); nm._assemble(3)
    K0 = nm.dRIGAduIGA()
    assert _rel(K0 @ r["x0"], B[0]) < 1e-7 and r["be0"] < 1e-12            # floor of the residual: eps cond(K) for a random right-hand side
    assert _rel(r["x0"], nm.solve_K(B[0])) < 1e-6
    nm.update_uIGA(G.smooth_displacement(spec, 0.8 * spec.h_th)); nm._assemble(3)
    K1 = nm.dRIGAduIGA()
    assert abs(K1 - r["K"]).max() < 1e-9 * abs(K1).max()
    for k in range(2):
        assert _rel(K1 @ r["X"][k], B[k]) < 1e-7
    _, u = nm.solve_nonlinear_nonmatching_problem(rtol=3e-5, max_it=30)
    # the Newton loop on top of it: the same path as the unsharded run, down to the same floor (this start is far from equilibrium: 26 iterations)
    assert r["newton_rr"] < 2.0 * nm.newton_relative_residual + 1e-12 and _rel(r["u"], u) < 1e-6


def test_plate_thickness_example_under_torchrun_on_two_ranks():
    """examples/plate_thickness_opt.py launched as the reference launches its demos under MPI: ``python -m torch.distributed.run --nproc-per-node 2`` (one-GPU box: both ranks
    on GPU 0 over gloo), the distributed factorisation forced for this small model -- the optimum equals the one-process run's."""
    import importlib.util, subprocess, sys, re
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, GF_SHARDED_SOLVER="distributed", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                          os.path.join(here, "examples", "plate_thickness_opt.py")], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    h = np.array([float(x) for x in re.findall(r"Thickness for patch\s+\d+:\s+([0-9.eE+-]+)", out.stdout)])
    assert h.size == 6, out.stdout[-2000:]
    spec_ = importlib.util.spec_from_file_location("plate_thickness_opt", os.path.join(here, "examples", "plate_thickness_opt.py"))
    mod = importlib.util.module_from_spec(spec_)
    spec_.loader.exec_module(mod)
    ref = mod.run(verbose=False)
    assert np.abs(h - ref["h"]).max() < 2e-6, (h, ref["h"])


def test_newton_chord_steps_on_the_device_solver():
    """NonMatchingOpt.newton_reuse_factors: chord steps with the factors at hand (solve_K(..., stale_factors=True): substitutions only, no refactorisation, no acceptance
    test of the linear solve) reach the state of the reference's iteration -- here on the 4 x 4-patch shell under a load that bends it by several thicknesses, with both
    factorisation modes of the device solver."""
    import dataclasses
    from goldfish_amd.nonmatching_opt import NonMatchingOpt
    spec0 = G.synthetic_shell(4, 4, nel=10, p=3, jitter=2)
    spec = dataclasses.replace(spec0, body_force=[[0.0, 0.0, -100.0]] * len(spec0.patches))
    out = {}
    for reuse in (False, True):
        nm = NonMatchingOpt.from_spec(spec)
        nm.newton_reuse_factors = reuse
        _, u = nm.solve_nonlinear_nonmatching_problem(rtol=3e-5, max_it=40)
        assert nm.newton_converged and nm.linear_solver == "device" and getattr(nm, "_dsolver_failed_version", None) is None
        out[reuse] = (u, nm.newton_iterations, nm.newton_chord_steps, nm.newton_relative_residual)
    assert out[False][2] == 0 and out[True][2] >= 1                       # chord steps were taken, and only with the switch on
    assert _rel(out[True][0], out[False][0]) < 1e-6
    assert out[True][1] - out[True][2] < out[False][1]                     # fewer factorisations than the plain iteration


@pytest.mark.gpu
def test_device_pcg_is_the_reference_ksp_helper_on_the_device():
    """GOLDFISH/utils/opt_utils.py:104-131 PETSc_ksp_solve (cg + jacobi, max_it, rtol) with K left on the device (goldfish_amd/_krylov.py: products through
    gf_apply_dev, vectors as torch tensors ordered against the library's stream by events): a thick single-patch roof (cond(K) ~ 1e7) converges to the direct
    solution with every preconditioner; the thin nine-patch roof with penalty coupling (cond(K) ~ 1e13) does not within 2000 iterations -- and says so."""
    import warnings
    import dataclasses
    import scipy.sparse.linalg as spla
    from goldfish_amd import _lib
    from goldfish_amd.utils.opt_utils import PETSc_ksp_solve
    from goldfish_amd._krylov import DevicePCG
    spec = G.scordelis_lo_single(8, 3)
    spec = dataclasses.replace(spec, h_th=2.5)                       # R / h = 10
    A = arrays_from_spec(spec, None)
    D = _lib.DeviceModel(A)
    D.set_thickness(np.full(A.total_cp, 2.5))
    D.set_u(np.zeros(A.ndof))
    D.assemble(_lib.ASM_R | _lib.ASM_K)
    b = -D.residual()
    x_ref = spla.spsolve(D.csr(_lib.MAT_K).tocsc(), b)
    its = {}
    for pc in ("jacobi", "bjacobi", "none"):
        x = np.zeros(A.ndof)
        with warnings.catch_warnings():
            warnings.simplefilter("error")
            PETSc_ksp_solve(D, x, b, pc_type=pc, rtol=1e-11, max_it=20000)
        assert _rel(x, x_ref) < 1e-7, pc
        S = DevicePCG(D, pc_type=pc)
        S.solve(b, rtol=1e-11, max_it=20000, check_every=10)
        assert S.converged and S.rel_residual < 1e-9
        its[pc] = S.iterations
    assert its["bjacobi"] <= its["none"]
    D.close()
    # the thin, penalty-coupled roof: honest failure
    spec = G.scordelis_lo_9patch(3, nels=[2, 1, 2, 3, 2, 3, 2, 1, 2])
    A = arrays_from_spec(spec, None)
    D = _lib.DeviceModel(A)
    D.set_thickness(np.full(A.total_cp, spec.h_th))
    D.set_u(np.zeros(A.ndof))
    D.assemble(_lib.ASM_R | _lib.ASM_K)
    x = np.zeros(A.ndof)
    with pytest.warns(RuntimeWarning, match="ended after"):
        PETSc_ksp_solve(D, x, -D.residual(), pc_type="jacobi", rtol=1e-12, max_it=300)
    D.close()
