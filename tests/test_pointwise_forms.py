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


def _host_kernel_lib():
    """kl_point.hpp (the header the HIP kernels include) compiled for the host: goldfish_amd/csrc/point_host_test.cpp."""
    import ctypes as C
    import os
    from goldfish_amd import build
    build.build()
    return C.CDLL(os.path.join(os.path.dirname(build.__file__), "csrc", "libgf_point_host_test.so"))


def test_kernel_header_pointwise_forms_on_host():
    """The kernels' own closed forms (kl_point.hpp: shell_point + the entry expansions) against the numpy statement
    of the formulation, and the column-split variant used by the MFMA element kernel against shell_point."""
    import ctypes as C
    L = _host_kernel_lib()
    dp = C.POINTER(C.c_double)
    L.gfh_shell_point.argtypes = [dp, dp, C.c_double, C.c_double, C.c_double, dp, dp, dp, dp, dp]
    L.gfh_shell_point_cols.argtypes = [dp, dp, C.c_double, C.c_double, C.c_double, dp]
    L.gfh_sizes.restype = C.c_int
    n = L.gfh_sizes(0)
    P = lambda a: a.ctypes.data_as(dp)
    rng = np.random.default_rng(5)
    for trial in range(4):
        z, Z = _rand_state(rng)
        z, Z = np.ascontiguousarray(z.ravel()), np.ascontiguousarray(Z.ravel())
        t, E, nu = 0.21 + 0.1 * trial, 2.5, 0.3
        im, im2 = np.zeros(n), np.zeros(n)
        Pzz, PzZ, Pz, Pzt = np.zeros((15, 15)), np.zeros((15, 15)), np.zeros(15), np.zeros(15)
        L.gfh_shell_point(P(z), P(Z), t, E, nu, P(im), P(Pzz), P(PzZ), P(Pz), P(Pzt))
        ref = kp.shell_point(z, Z, t, E, nu)
        for name, got in (("Pz", Pz), ("Pzz", Pzz), ("PzZ", PzZ), ("Pzt", Pzt)):
            assert np.abs(got - ref[name]).max() < 1e-12 * np.abs(ref[name]).max(), name
        L.gfh_shell_point_cols(P(z), P(Z), t, E, nu, P(im2))
        scale = np.abs(im).max()
        assert np.abs(im2 - im).max() < 1e-13 * scale, np.argmax(np.abs(im2 - im))


def test_kernel_header_stress_point_vs_autograd():
    """kl_point.hpp shell_stress_point (invariant form, hand-written adjoints) against torch.autograd of the
    local-Cartesian statement of the von Mises stress (oracle/kl_energy_torch.py) -- value and all 31 derivatives,
    Cauchy and 2nd Piola-Kirchhoff measures, top / bottom / middle surface."""
    import ctypes as C
    L = _host_kernel_lib()
    dp = C.POINTER(C.c_double)
    L.gfh_shell_stress_point.argtypes = [dp, dp, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, dp]
    P = lambda a: a.ctypes.data_as(dp)
    rng = np.random.default_rng(11)
    for trial in range(6):
        z, Z = _rand_state(rng)
        t, E, nu = 0.15 + 0.05 * trial, 4.0, 0.3
        sgn = (1.0, -1.0, 0.0)[trial % 3]
        for mi, measure in enumerate(("cauchy", "pk2")):
            out = np.zeros(39)
            L.gfh_shell_stress_point(P(np.ascontiguousarray(z.ravel())), P(np.ascontiguousarray(Z.ravel())), t, E, nu, sgn, mi, P(out))
            zt, Zt, tt = torch.tensor(z, requires_grad=True), torch.tensor(Z, requires_grad=True), torch.tensor(t, requires_grad=True)
            s = ke.von_mises_stress(zt, Zt, tt, E, nu, sgn, measure)
            gz, gZ, gt = torch.autograd.grad(s, (zt, Zt, tt), allow_unused=True)
            assert abs(out[0] - s.item()) < 1e-12 * abs(s.item()), (measure, sgn)
            assert np.abs(out[3:18] - gz.numpy().ravel()).max() < 1e-11 * np.abs(gz.numpy()).max(), (measure, sgn)
            assert np.abs(out[18:33] - gZ.numpy().ravel()).max() < 1e-11 * np.abs(gZ.numpy()).max(), (measure, sgn)
            assert abs(out[2] - (0.0 if gt is None else gt.item())) < 1e-11 * max(abs(out[2]), 1e-30) + 1e-14, (measure, sgn)


def test_kernel_header_penalty_gradient_in_dual_numbers(oracle_lib):
    """kl_point.hpp penalty_grad_t<Dual> (moving intersections, N3): value = the oracle's vertex gradient, directional
    derivative = Hyy dy + HyY dY (complex-step oracle) for y / Y seeds and a central difference for the tangent seed."""
    import ctypes as C
    L = _host_kernel_lib()
    dp = C.POINTER(C.c_double)
    L.gfh_penalty_grad_dual.argtypes = [dp] * 6 + [C.c_double] * 3 + [dp, dp]
    P = lambda a: a.ctypes.data_as(dp)
    rng = np.random.default_rng(21)
    for trial in range(3):
        Y = rng.standard_normal(12) + np.array([1, 0, 0, 0, 1, 0, 0, 1, 0, 0, 0, 1.0]) * 3
        y = np.zeros(18)
        y[0:3], y[9:12] = 0.1 * rng.standard_normal(3), 0.1 * rng.standard_normal(3)
        y[3:9] = Y[0:6] + 0.05 * rng.standard_normal(6)
        y[12:18] = Y[6:12] + 0.05 * rng.standard_normal(6)
        tau = rng.standard_normal(2)
        dy, dY, dtau = rng.standard_normal(18), rng.standard_normal(12), rng.standard_normal(2)
        en, g, Hyy, HyY = oracle_py.penalty_point(y, Y, tau, 7.0, 3.0, 0.25)
        gr, dgr, z18, z12, z2 = np.zeros(18), np.zeros(18), np.zeros(18), np.zeros(12), np.zeros(2)
        L.gfh_penalty_grad_dual(P(y), P(Y), P(tau), P(dy), P(dY), P(z2), 7.0, 3.0, 0.25, P(gr), P(dgr))
        assert np.abs(gr - g).max() < 1e-12 * np.abs(g).max()
        ref = Hyy @ dy + HyY @ dY
        assert np.abs(dgr - ref).max() < 1e-10 * np.abs(ref).max()
        L.gfh_penalty_grad_dual(P(y), P(Y), P(tau), P(z18), P(z12), P(dtau), 7.0, 3.0, 0.25, P(gr), P(dgr))
        eps = 1e-6
        gp = oracle_py.penalty_point(y, Y, tau + eps * dtau, 7.0, 3.0, 0.25)[1]
        gm = oracle_py.penalty_point(y, Y, tau - eps * dtau, 7.0, 3.0, 0.25)[1]
        fd = (gp - gm) / (2 * eps)
        assert np.abs(dgr - fd).max() < 1e-7 * np.abs(fd).max()
