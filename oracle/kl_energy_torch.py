"""TEST INFRASTRUCTURE ONLY -- ground-truth *energies* for the KL-shell hot path.

PARITY UNPINNED vs FEniCS: the reference's arithmetic lives in un-vendored
third-party packages (ShNAPr / PENGoLINS / tIGAr); nothing
in /root/reference pins a number except the Scordelis-Lo value 0.3006
(GOLDFISH/tests/test_slr.py:50), which tests/test_known_answers.py checks.

This file states only the *scalar energies* of the published formulation and
lets torch.autograd (float64) produce every derivative, so that no
hand-derived gradient/Hessian in oracle/kl_oracle.c or in the HIP kernels is
trusted without an independent check.  Nothing under goldfish_amd/ may import it.

Follows (read as text):
  * GOLDFISH/operations/int_energy_exop.py:22-30  (X = F, x = X + u/W,
    wint = surfaceEnergyDensitySVK(spline, X, x, E, nu, h) * dx)
  * SURVEY.md Appendix A.2/A.3 (Kiendl 2009 KL kinematics, SVK energy with the
    through-thickness integration done analytically: h*D membrane, h^3/12*D bending)
  * SURVEY.md Appendix A.4 (Herrema 2019 penalty energy, vertex quadrature)

Pointwise variables (all per quadrature point):
  z = (g1, g2, h11, h22, h12)  deformed   tangents / second derivatives, shape (5,3)
  Z = (G1, G2, H11, H22, H12)  reference  tangents / second derivatives, shape (5,3)
  t = thickness
Both z and Z are *linear* in the homogeneous control coefficients because the
geometry is X = sum_a (N_a/W) c_a and the deformed surface is
x = sum_a (N_a/W) (c_a + U_a)   (c_a = w_a P_a, U_a = IGA displacement dof).
"""
import torch

torch.set_default_dtype(torch.float64)


def material_voigt(G1, G2, E, nu):
    """Isotropic plane-stress tensor in contravariant curvilinear components,
    Voigt order (11, 22, 12) for strains [e11, e22, 2 e12].  Identical (by
    tensor invariance) to ShNAPr's local-Cartesian-basis D = E/(1-nu^2) [[1,nu,0],
    [nu,1,0],[0,0,(1-nu)/2]] route."""
    A11 = (G1 * G1).sum(-1)
    A22 = (G2 * G2).sum(-1)
    A12 = (G1 * G2).sum(-1)
    det = A11 * A22 - A12 * A12
    c11 = A22 / det
    c22 = A11 / det
    c12 = -A12 / det
    Eb = E / (1.0 - nu * nu)
    C = torch.stack([
        torch.stack([c11 * c11, nu * c11 * c22 + (1 - nu) * c12 * c12, c11 * c12], -1),
        torch.stack([nu * c11 * c22 + (1 - nu) * c12 * c12, c22 * c22, c22 * c12], -1),
        torch.stack([c11 * c12, c22 * c12, 0.5 * ((1 - nu) * c11 * c22 + (1 + nu) * c12 * c12)], -1),
    ], -2)
    Eb = torch.as_tensor(Eb)
    C = C * Eb.reshape(Eb.shape + (1, 1)) if Eb.dim() > 0 else C * Eb
    return C, torch.sqrt(det)


def unit_normal(g1, g2):
    n = torch.linalg.cross(g1, g2)
    return n / torch.linalg.norm(n, dim=-1, keepdim=True)


def metric_voigt(z):
    g1, g2 = z[..., 0, :], z[..., 1, :]
    return torch.stack([(g1 * g1).sum(-1), (g2 * g2).sum(-1), 2 * (g1 * g2).sum(-1)], -1)


def curvature_voigt(z):
    n = unit_normal(z[..., 0, :], z[..., 1, :])
    return torch.stack([(z[..., 2, :] * n).sum(-1), (z[..., 3, :] * n).sum(-1),
                        2 * (z[..., 4, :] * n).sum(-1)], -1)


def shell_energy_density(z, Z, t, E, nu):
    """Psi = psi_SVK * |G1 x G2|  (energy per unit *parametric* area)."""
    C, J = material_voigt(Z[..., 0, :], Z[..., 1, :], E, nu)
    eps = 0.5 * (metric_voigt(z) - metric_voigt(Z))
    kap = curvature_voigt(Z) - curvature_voigt(z)
    Ce = (C @ eps.unsqueeze(-1)).squeeze(-1)
    Ck = (C @ kap.unsqueeze(-1)).squeeze(-1)
    psi = 0.5 * t * (eps * Ce).sum(-1) + (t ** 3 / 24.0) * (kap * Ck).sum(-1)
    return psi * J


def von_mises_stress(z, Z, t, E, nu, sgn=1.0, measure="cauchy"):
    """von Mises stress of the SVK Kirchhoff-Love shell at the through-thickness station xi3 = sgn*t/2
    (GOLDFISH/operations/max_vmstress_exop.py:17-47 evaluates PENGoLINS' ``ShellStressSVK(...).vonMisesStress(xi2)``
    at +h/2, -h/2 or 0; PENGoLINS is not vendored, so this restates the published route in the classical
    local-Cartesian form -- deliberately not the invariant form of goldfish_amd/csrc/kl_point.hpp):

      E_ab = eps_ab + xi3 kappa_ab                       (covariant Green-Lagrange strain, Kiendl 2009)
      e1 = A1/|A1|, e2 = unit(A2 - (A2.e1) e1)           (ShNAPr orthonormalize2D)
      Ebar_ij = E_ab (A^a.e_i)(A^b.e_j)                  (covariantRank2TensorToCartesian2D)
      Sbar = E/(1-nu^2) [[1,nu,0],[nu,1,0],[0,0,(1-nu)/2]] voigt(Ebar)      (2nd Piola-Kirchhoff)
      measure "pk2":    vM of Sbar
      measure "cauchy": sigma = F S F^T / Jr,  F = a_a (x) A^a,  Jr = |a1 x a2| / |A1 x A2|,
                        vM^2 = 3/2 tr(sigma^2) - 1/2 tr(sigma)^2   (plane stress in the deformed tangent plane)
    """
    A1, A2 = Z[0], Z[1]
    a1, a2 = z[0], z[1]
    eps = 0.5 * (metric_voigt(z) - metric_voigt(Z))          # (e11, e22, 2 e12)
    kap = curvature_voigt(Z) - curvature_voigt(z)
    Ev = eps + 0.5 * sgn * t * kap
    Ecov = torch.stack([torch.stack([Ev[0], 0.5 * Ev[2]]), torch.stack([0.5 * Ev[2], Ev[1]])])
    Am = torch.stack([torch.stack([A1 @ A1, A1 @ A2]), torch.stack([A1 @ A2, A2 @ A2])])
    Ai = torch.linalg.inv(Am)
    Ac = [Ai[0, 0] * A1 + Ai[0, 1] * A2, Ai[1, 0] * A1 + Ai[1, 1] * A2]     # contravariant basis
    e1 = A1 / torch.linalg.norm(A1)
    e2 = A2 - (A2 @ e1) * e1
    e2 = e2 / torch.linalg.norm(e2)
    T = torch.stack([torch.stack([Ac[0] @ e1, Ac[0] @ e2]), torch.stack([Ac[1] @ e1, Ac[1] @ e2])])   # T[a, i] = A^a . e_i
    Eb = T.T @ Ecov @ T
    D = E / (1 - nu * nu)
    S11 = D * (Eb[0, 0] + nu * Eb[1, 1])
    S22 = D * (Eb[1, 1] + nu * Eb[0, 0])
    S12 = D * (1 - nu) * Eb[0, 1]
    if measure == "pk2":
        return torch.sqrt(S11 * S11 - S11 * S22 + S22 * S22 + 3 * S12 * S12)
    Fe1 = a1 * T[0, 0] + a2 * T[1, 0]                         # F e_i = a_a (A^a . e_i)
    Fe2 = a1 * T[0, 1] + a2 * T[1, 1]
    Jr = torch.linalg.norm(torch.linalg.cross(a1, a2)) / torch.linalg.norm(torch.linalg.cross(A1, A2))
    sig = (S11 * torch.outer(Fe1, Fe1) + S22 * torch.outer(Fe2, Fe2) + S12 * (torch.outer(Fe1, Fe2) + torch.outer(Fe2, Fe1))) / Jr
    return torch.sqrt(1.5 * (sig * sig).sum() - 0.5 * torch.trace(sig) ** 2)


def area_jacobian(Z):
    return torch.linalg.norm(torch.linalg.cross(Z[..., 0, :], Z[..., 1, :]), dim=-1)


def penalty_energy_point(uA, gA, uB, gB, GA, GB, tau, alpha_d, alpha_r, dt):
    """Herrema-2019 penalty energy of ONE mortar vertex.

    uA,uB: physical displacements (3,); gA,gB: deformed tangents (2,3);
    GA,GB: reference tangents (2,3); tau = d(xi_A)/d(mortar parameter) (2,),
    dt = vertex-quadrature weight in the mortar parameter.
    Line Jacobian and interface tangent are taken from side A (pointwise)."""
    tref = tau[0] * GA[0] + tau[1] * GA[1]
    L = torch.linalg.norm(tref)
    At = tref / L
    tdef = tau[0] * gA[0] + tau[1] * gA[1]
    at = tdef / torch.linalg.norm(tdef)
    nA, nB = unit_normal(gA[0], gA[1]), unit_normal(gB[0], gB[1])
    NA, NB = unit_normal(GA[0], GA[1]), unit_normal(GB[0], GB[1])
    an = torch.linalg.cross(at, nA)
    An = torch.linalg.cross(At, NA)
    e1 = (nA * nB).sum() - (NA * NB).sum()
    e2 = (an * nB).sum() - (An * NB).sum()
    d = uA - uB
    return dt * L * (0.5 * alpha_d * (d * d).sum() + 0.5 * alpha_r * (e1 * e1 + e2 * e2))
