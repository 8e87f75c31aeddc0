"""Closed-form pointwise gradients/Hessians (the formulation the HIP kernels use,
documented in oracle/kl_point_numpy.py) vs torch.autograd of the bare energies."""
import numpy as np
import torch

from oracle import kl_energy_torch as ke
from oracle import kl_point_numpy as kp
from oracle import oracle_py


def _rand_state(rng):
    Z = np.zeros((5, 3))
    Z[0] = [1.0, 0.1, 0.2]
    Z[1] = [-0.2, 1.3, 0.1]
    Z += 0.3 * rng.standard_normal((5, 3))
    z = Z + 0.1 * rng.standard_normal((5, 3))
    return z, Z


def test_shell_point_vs_autograd():
    rng = np.random.default_rng(1)
    for trial in range(3):
        z, Z = _rand_state(rng)
        t, E, nu = 0.37, 3.0, 0.3
        out = kp.shell_point(z.ravel(), Z.ravel(), t, E, nu)
        zt = torch.tensor(z, requires_grad=True)
        Zt = torch.tensor(Z, requires_grad=True)
        tt = torch.tensor(t, requires_grad=True)
        Psi = ke.shell_energy_density(zt, Zt, tt, E, nu)
        assert abs(out["Psi"] - Psi.item()) < 1e-13 * abs(Psi.item())
        gz, = torch.autograd.grad(Psi, zt, create_graph=True)
        gflat = gz.reshape(-1)
        assert np.abs(out["Pz"] - gflat.detach().numpy()).max() < 1e-12 * np.abs(out["Pz"]).max()
        Pzz, PzZ, Pzt = np.zeros((15, 15)), np.zeros((15, 15)), np.zeros(15)
        for r in range(15):
            a, b, c = torch.autograd.grad(gflat[r], (zt, Zt, tt), retain_graph=True)
            Pzz[r], PzZ[r], Pzt[r] = a.numpy().ravel(), b.numpy().ravel(), c.item()
        for name, ref in (("Pzz", Pzz), ("PzZ", PzZ), ("Pzt", Pzt)):
            assert np.abs(out[name] - ref).max() < 1e-11 * np.abs(ref).max(), name
        J = ke.area_jacobian(Zt)
        gJ, = torch.autograd.grad(J, Zt)
        assert np.abs(out["JZ"] - gJ.numpy().ravel()).max() < 1e-13


def test_penalty_point_closed_form(oracle_lib):
    rng = np.random.default_rng(2)
    for trial in range(3):
        Y = rng.standard_normal(12) + np.array([1, 0, 0, 0, 1, 0, 0, 1, 0, 0, 0, 1.0]) * 3
        y = np.zeros(18)
        y[0:3], y[9:12] = 0.1 * rng.standard_normal(3), 0.1 * rng.standard_normal(3)
        y[3:9] = Y[0:6] + 0.05 * rng.standard_normal(6)
        y[12:18] = Y[6:12] + 0.05 * rng.standard_normal(6)
        tau = rng.standard_normal(2)
        en, g, Hyy, HyY = kp.penalty_point(y, Y, tau, 7.0, 3.0, 0.25)
        en2, g2, Hyy2, HyY2 = oracle_py.penalty_point(y, Y, tau, 7.0, 3.0, 0.25)   # complex-step oracle
        assert abs(en - en2) < 1e-13 * abs(en2)
        assert np.abs(g - g2).max() < 1e-12 * np.abs(g2).max()
        assert np.abs(Hyy - Hyy2).max() < 1e-11 * np.abs(Hyy2).max()
        assert np.abs(HyY - HyY2).max() < 1e-11 * np.abs(HyY2).max()
