"""Pins oracle/kl_oracle.c: every gradient / Hessian it assembles is compared with
torch.autograd of the bare energies (oracle/kl_energy_torch.py) on small multi-patch
models with true NURBS weights, variable thickness and a finite displacement state.
(The reference's own verification is analytic-vs-FD `check_partials`,
GOLDFISH/om_comps/disp_states_comp.py:146-159; autograd is the exact version of it.)"""
import numpy as np
import pytest
import torch

from goldfish_amd import geometry as G
from goldfish_amd.model import arrays_from_spec
from oracle import oracle_py
from oracle.oracle_py import Oracle
from tests.torch_model import TorchModel

RTOL = 1e-10


def _setup(spec, seed=0, uamp=2e-2):
    rng = np.random.default_rng(seed)
    th = [spec.h_th * rng.uniform(0.8, 1.2, p.ncp) for p in spec.patches]
    A = arrays_from_spec(spec, th)
    h = np.concatenate(th)
    u = uamp * rng.standard_normal(A.ndof)
    O = Oracle(A, thickness=h, u=u)
    T = TorchModel(spec, A)
    c = torch.tensor(np.stack(A.cp_hom, 1), requires_grad=True)
    U = torch.tensor(u.reshape(-1, 3), requires_grad=True)
    ht = torch.tensor(h, requires_grad=True)
    return A, O, T, c, U, ht


def _relerr(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


@pytest.mark.parametrize("name", ["slr9", "tbeam", "slr9_projected_load", "slr9_pressure_edge_traction", "tbeam_pressure_edge_traction"])
def test_oracle_vs_autograd(oracle_lib, name):
    spec = G.scordelis_lo_9patch(3, nels=[2, 1, 2, 3, 2, 3, 2, 1, 2]) if name.startswith("slr9") else G.tbeam_2patch(4)
    if name == "slr9_projected_load":                 # load per unit projected area along a general direction (arch demo: e_z)
        spec.load_proj = [[0.3, -0.2, 1.0]] * len(spec.patches)
    if name.endswith("pressure_edge_traction"):       # follower pressure (tube demo) on every second patch + dead edge tractions (plate demo)
        n = len(spec.patches)
        spec.pressure = [(-1.0e4 if s % 2 == 0 else 0.0) * (1 + s) for s in range(n)]
        spec.edge_traction = [(n - 1, 0, 1, (30.0, -20.0, -100.0)), (0, 1, 0, (0.0, 50.0, 10.0)), (n - 1, 1, 1, (5.0, 0.0, 7.0))]
    A, O, T, c, U, ht = _setup(spec)
    free = np.ones(A.ndof, bool)
    free[A.zero_dofs] = False

    def grad_total(U_, c_, h_):
        g = torch.autograd.grad(T.total(c_, U_, h_), U_, create_graph=True)[0]
        V = torch.zeros_like(U_, requires_grad=True)          # loads without an energy: the residual is the coefficient of the virtual displacement
        ge = torch.autograd.grad(T.extra_virtual_work(c_, U_, V), V, create_graph=True, allow_unused=True)[0]
        return (g - ge).reshape(-1) if ge is not None else g.reshape(-1)

    def grad_shell(U_, h_):
        return torch.autograd.grad(T.shell_energy(c, U_, h_), U_, create_graph=True)[0].reshape(-1)

    R_ref = grad_total(U, c, ht).detach().numpy().copy()
    for d, v in zip(A.pl_dof, A.pl_val):
        R_ref[d] -= v
    R_ref[~free] = 0
    R = O.residual()
    assert _relerr(R, R_ref) < RTOL

    vals = O.assemble()
    K = O.csr(oracle_py.MAT_K, vals[0]).toarray()
    C = [O.csr(1 + f, vals[1 + f]).toarray() for f in range(3)]
    H = O.csr(oracle_py.MAT_DRDH, vals[4]).toarray()
    n = A.ndof
    JU, Jc, _ = torch.autograd.functional.jacobian(grad_total, (U, c, ht), vectorize=True)
    K_ref = JU.reshape(n, n).numpy()
    C_ref = Jc.numpy().transpose(2, 0, 1)
    # Dirichlet conventions (nonmatching_opt.py:660-724, 992-1015)
    Kb = K_ref.copy()
    Kb[~free, :] = 0
    Kb[:, ~free] = 0
    Kb[~free, ~free] = 1
    assert _relerr(K, Kb) < RTOL
    for f in range(3):
        Cb = C_ref[f].copy()
        Cb[~free, :] = 0
        assert _relerr(C[f], Cb) < RTOL
    # dR/dh: shell terms only, no BC treatment -> compare with the shell-only energy
    _, Jh = torch.autograd.functional.jacobian(grad_shell, (U, ht), vectorize=True)
    H_ref = Jh.numpy()
    assert _relerr(H, H_ref) < RTOL


def test_functionals_vs_autograd(oracle_lib):
    spec = G.scordelis_lo_9patch(3, nels=[2, 1, 2, 3, 2, 3, 2, 1, 2])
    A, O, T, c, U, ht = _setup(spec, seed=3)
    F = O.functionals(apply_bcs=False)
    W = T.shell_energy(c, U, ht, with_load=False)
    gU, gc, gh = torch.autograd.grad(W, (U, c, ht))
    assert abs(F["Wint"] - W.item()) < 1e-12 * abs(W.item())
    assert _relerr(F["dWdu"], gU.numpy().ravel()) < RTOL
    for f in range(3):
        assert _relerr(F["dWdcp"][f], gc.numpy()[:, f]) < RTOL
    assert _relerr(F["dWdh"], gh.numpy()) < RTOL
    V = T.volume(c, ht)
    gc, gh = torch.autograd.grad(V, (c, ht))
    assert abs(F["volume"] - V.item()) < 1e-13 * abs(V.item())
    for f in range(3):
        assert _relerr(F["dVdcp"][f], gc.numpy()[:, f]) < RTOL
    assert _relerr(F["dVdh"], gh.numpy()) < RTOL
    Wp = T.penalty_energy(c, U)
    assert abs(F["Wpen"] - Wp.item()) < 1e-11 * abs(Wp.item())
    forces = np.random.default_rng(7).standard_normal((len(spec.patches), 3))
    Cg = O.compliance(forces, apply_bcs=False)
    Ct = T.compliance(c, U, forces)
    gU, gc = torch.autograd.grad(Ct, (U, c))
    assert abs(Cg["C"] - Ct.item()) < 1e-12 * abs(Ct.item())
    assert _relerr(Cg["dCdu"], gU.numpy().ravel()) < RTOL
    for f in range(3):
        assert _relerr(Cg["dCdcp"][f], gc.numpy()[:, f]) < RTOL


def test_stress_forms_vs_autograd(oracle_lib):
    """C oracle's stress aggregation forms (complex-step gradients) vs torch.autograd of the same statement:
    KS and p-norm integrands, Cauchy / 2nd PK measure, top and bottom surface."""
    spec = G.scordelis_lo_9patch(3, nels=[2, 1, 2, 1, 2, 1, 2, 1, 2])
    A, O, T, c, U, ht = _setup(spec, seed=4)
    npatch = len(spec.patches)
    base = O.stress_forms(1, 1.0, np.ones(npatch))["vmax"]
    for mode, rho, sgn, measure in ((0, 3.0, 1.0, 0), (1, 4.0, -1.0, 0), (1, 3.0, 1.0, 1)):
        m_list = base * (1.0 + 0.1 * np.arange(npatch))
        if mode == 0:
            rho = rho / base.max()
        F = O.stress_forms(mode, rho, m_list, sgn, measure, apply_bcs=False)
        I = T.stress_forms(c, U, ht, mode, rho, m_list, sgn, "cauchy" if measure == 0 else "pk2")
        assert _relerr(F["I"], I.detach().numpy()) < 1e-12
        wts = torch.tensor(np.random.default_rng(1).uniform(0.5, 1.5, npatch))
        gU, gc, gh = torch.autograd.grad((wts * I).sum(), (U, c, ht))
        # the oracle returns the un-weighted per-patch gradients in one field (every control point belongs to one patch)
        wcp = np.repeat(wts.numpy(), [P.ncp for P in spec.patches])
        assert _relerr(F["dIdu"] * np.repeat(wcp, 3), gU.numpy().ravel()) < RTOL
        for f in range(3):
            assert _relerr(F["dIdcp"][f] * wcp, gc.numpy()[:, f]) < RTOL
        assert _relerr(F["dIdh"] * wcp, gh.numpy()) < RTOL


def test_shape_regularisation_vs_autograd(oracle_lib):
    """Oracle's shape regularisation term (explicit contravariant metric, complex-step gradient) vs torch.autograd of the
    pseudo-inverse statement of tIGAr's manifold gradient."""
    spec = G.scordelis_lo_9patch(3, nels=[2, 1, 2, 1, 2, 1, 2, 1, 2])
    A, O, T, c, U, ht = _setup(spec, seed=6)
    rng = np.random.default_rng(3)
    cp_now = c.detach().numpy()
    for field in (2, 0):
        cp0 = cp_now[:, field] + 0.05 * rng.standard_normal(A.total_cp)
        coef = rng.uniform(0.5, 2.0, len(spec.patches))
        F = O.shape_regu(field, cp0, coef)
        V = T.shape_regu(c, field, cp0, coef)
        gc, = torch.autograd.grad(V, c)
        assert abs(F["value"] - V.item()) < 1e-12 * abs(V.item())
        for f in range(3):
            assert _relerr(F["dcp"][f], gc.numpy()[:, f]) < RTOL, (field, f)


def test_penalty_point_hessians(oracle_lib):
    """complex-step Hessians of one mortar vertex vs torch autograd."""
    from oracle import kl_energy_torch as ke
    rng = np.random.default_rng(5)
    Y = rng.standard_normal(12) + np.array([1, 0, 0, 0, 1, 0, 0, 1, 0, 0, 0, 1.0]) * 3
    y = np.zeros(18)
    y[0:3], y[9:12] = 0.1 * rng.standard_normal(3), 0.1 * rng.standard_normal(3)
    y[3:9] = Y[0:6] + 0.05 * rng.standard_normal(6)
    y[12:18] = Y[6:12] + 0.05 * rng.standard_normal(6)
    tau = np.array([0.3, 0.9])
    en, g, Hyy, HyY = oracle_py.penalty_point(y, Y, tau, 7.0, 3.0, 0.25)
    yt, Yt = torch.tensor(y, requires_grad=True), torch.tensor(Y, requires_grad=True)

    def fn(yt, Yt):
        return ke.penalty_energy_point(yt[0:3], yt[3:9].reshape(2, 3), yt[9:12], yt[12:18].reshape(2, 3),
                                       Yt[0:6].reshape(2, 3), Yt[6:12].reshape(2, 3), torch.tensor(tau), 7.0, 3.0, 0.25)
    e = fn(yt, Yt)
    ge, = torch.autograd.grad(e, yt, create_graph=True)
    assert abs(en - e.item()) < 1e-13 * abs(en)
    assert _relerr(g, ge.detach().numpy()) < 1e-12
    for r in range(18):
        a, b = torch.autograd.grad(ge[r], (yt, Yt), retain_graph=True)
        assert np.abs(Hyy[r] - a.numpy()).max() < 1e-10 * np.abs(Hyy).max()
        assert np.abs(HyY[r] - b.numpy()).max() < 1e-10 * np.abs(HyY).max()
