"""Whole-model energies in torch (float64) built from oracle/kl_energy_torch.py;
autograd of these scalars is the derivative ground truth for the oracle tests."""
import numpy as np
import torch

from goldfish_amd.splines import basis_ders, find_span
from oracle import kl_energy_torch as ke


def _gauss(n):
    return np.polynomial.legendre.leggauss(n)


def _rational(Nb, wl):
    """Nb (6, nb) -> Rb (6, nb) for R = N/W (quotient rule)."""
    W = Nb @ wl
    R = Nb[0] / W[0]
    R1 = (Nb[1] - R * W[1]) / W[0]
    R2 = (Nb[2] - R * W[2]) / W[0]
    R11 = (Nb[3] - 2 * R1 * W[1] - R * W[3]) / W[0]
    R22 = (Nb[4] - 2 * R2 * W[2] - R * W[4]) / W[0]
    R12 = (Nb[5] - R1 * W[2] - R2 * W[1] - R * W[5]) / W[0]
    return np.stack([R, R1, R2, R11, R22, R12])


def _point_basis(P, wflat, xi):
    su = find_span(P.n_u, P.p, P.knots[0], xi[0])
    sv = find_span(P.n_v, P.q, P.knots[1], xi[1])
    du = basis_ders(su, xi[0], P.p, P.knots[0], 2)
    dv = basis_ders(sv, xi[1], P.q, P.knots[1], 2)
    ids, Nb = [], []
    for jv in range(P.q + 1):
        for ju in range(P.p + 1):
            ids.append(P.flat(su - P.p + ju, sv - P.q + jv))
            Nb.append([du[0, ju] * dv[0, jv], du[1, ju] * dv[0, jv], du[0, ju] * dv[1, jv],
                       du[2, ju] * dv[0, jv], du[0, ju] * dv[2, jv], du[1, ju] * dv[1, jv]])
    ids = np.array(ids)
    Nb = np.array(Nb).T
    return ids, Nb, _rational(Nb, wflat[ids])


class TorchModel:
    def __init__(self, spec, arrays):
        self.spec, self.A = spec, arrays
        ids, Rb, N0, wq, pid, Nb12 = [], [], [], [], [], []
        for s, P in enumerate(spec.patches):
            w = P.cp_hom_flat()[:, 3]
            off = int(arrays.cp_off[s])
            for d, (kn, deg) in enumerate(((P.knots[0], P.p), (P.knots[1], P.q))):
                pass
            ku, kv = np.unique(P.knots[0]), np.unique(P.knots[1])
            gxu, gwu = _gauss(P.p + 1)
            gxv, gwv = _gauss(P.q + 1)
            for b0, b1 in zip(kv[:-1], kv[1:]):
                for a0, a1 in zip(ku[:-1], ku[1:]):
                    for gv, wv in zip(gxv, gwv):
                        for gu, wu in zip(gxu, gwu):
                            xi = (0.5 * (a0 + a1) + 0.5 * (a1 - a0) * gu, 0.5 * (b0 + b1) + 0.5 * (b1 - b0) * gv)
                            i, Nb, R = _point_basis(P, w, xi)
                            ids.append(i + off)
                            Rb.append(R)
                            N0.append(Nb[0])
                            Nb12.append(Nb[1:3])
                            wq.append(0.25 * (a1 - a0) * (b1 - b0) * wu * wv)
                            pid.append(s)
        nbmax = max(len(i) for i in ids)
        assert all(len(i) == nbmax for i in ids), "mixed degrees not supported by the test helper"
        self.ids = torch.tensor(np.array(ids))
        self.Rb = torch.tensor(np.array(Rb))
        self.N0 = torch.tensor(np.array(N0))
        self.Nb12 = torch.tensor(np.array(Nb12))
        self.wq = torch.tensor(np.array(wq))
        self.pid = torch.tensor(np.array(pid))
        self.E = torch.tensor(arrays.young)[self.pid]
        self.nu = torch.tensor(arrays.poisson)[self.pid]
        self.f = torch.tensor(arrays.body_force.reshape(-1, 3))[self.pid]
        self.pd = torch.tensor(arrays.load_proj.reshape(-1, 3))[self.pid]
        self.proj = (self.pd.abs().sum(-1) > 0).to(torch.float64)
        self.press = torch.tensor(arrays.pressure)[self.pid]
        # Gauss points of the loaded patch edges (gf_model_desc.edge_traction): ids, rational basis, weight, force, direction along the edge
        self.edges = []
        et = arrays.edge_traction.reshape(-1, 4, 3)
        for s, P in enumerate(spec.patches):
            w = P.cp_hom_flat()[:, 3]
            off = int(arrays.cp_off[s])
            for e in range(4):
                if not np.any(et[s, e]):
                    continue
                d, side, td = e // 2, e % 2, 1 - e // 2
                deg = (P.p, P.q)[td]
                kn = np.unique(P.knots[td])
                gx, gw = _gauss(deg + 1)
                fixed = P.knots[d][0] if side == 0 else P.knots[d][-1]
                for a0, a1 in zip(kn[:-1], kn[1:]):
                    for x, wg in zip(gx, gw):
                        xi = [0.0, 0.0]
                        xi[d], xi[td] = fixed, 0.5 * (a0 + a1) + 0.5 * (a1 - a0) * x
                        i, Nb, R = _point_basis(P, w, xi)
                        self.edges.append((torch.tensor(i + off), torch.tensor(R[0]), torch.tensor(R[1 + td]), 0.5 * (a1 - a0) * wg, torch.tensor(et[s, e])))
        # mortar points
        self.mp = []
        for k, itf in enumerate(spec.interfaces):
            PA, PB = spec.patches[itf.a], spec.patches[itf.b]
            wA, wB = PA.cp_hom_flat()[:, 3], PB.cp_hom_flat()[:, 3]
            for v in range(itf.npts):
                ia, _, RA = _point_basis(PA, wA, itf.xi_a[v])
                ib, _, RB = _point_basis(PB, wB, itf.xi_b[v])
                self.mp.append((torch.tensor(ia + int(arrays.cp_off[itf.a])), torch.tensor(RA[:3]),
                                torch.tensor(ib + int(arrays.cp_off[itf.b])), torch.tensor(RB[:3]),
                                torch.tensor(itf.tau[v]), float(arrays.if_alpha[2 * k]), float(arrays.if_alpha[2 * k + 1]),
                                float(itf.wt[v])))

    def shell_energy(self, c, U, h, with_load=True):
        """c, U: (total_cp, 3); h: (total_cp,).  Returns W_int - W_ext."""
        cl, dl = c[self.ids], (c + U)[self.ids]
        Z = torch.einsum("gma,gak->gmk", self.Rb[:, 1:], cl)
        z = torch.einsum("gma,gak->gmk", self.Rb[:, 1:], dl)
        t = (self.N0 * h[self.ids]).sum(-1)
        Psi = ke.shell_energy_density(z, Z, t, self.E, self.nu)
        W = (self.wq * Psi).sum()
        if with_load:
            uphys = torch.einsum("ga,gak->gk", self.Rb[:, 0], U[self.ids])
            # distributed load per unit area (|G1 x G2|) or per unit projected area (d . (G1 x G2), gf_model_desc.load_proj)
            Nt = torch.linalg.cross(Z[:, 0, :], Z[:, 1, :])
            s = self.proj * (self.pd * Nt).sum(-1) + (1 - self.proj) * ke.area_jacobian(Z)
            W = W - (self.wq * s * (self.f * uphys).sum(-1)).sum()
        return W

    def extra_virtual_work(self, c, U, V):
        """Virtual work of the loads that are not the gradient of a stored energy, linear in the virtual displacement coefficients V:
        follower pressure p (x_,1 x x_,2) . z dxi on the deformed configuration (tube demo) + dead edge tractions f . z |X_,t| dt."""
        dl = (c + U)[self.ids]
        z = torch.einsum("gma,gak->gmk", self.Rb[:, 1:3], dl)
        zt = torch.einsum("ga,gak->gk", self.Rb[:, 0], V[self.ids])
        vw = (self.wq * self.press * (torch.linalg.cross(z[:, 0, :], z[:, 1, :]) * zt).sum(-1)).sum()
        for ids, R0, Rt, w, f in self.edges:
            Xt = (Rt[:, None] * c[ids]).sum(0)
            vw = vw + w * (f * (R0[:, None] * V[ids]).sum(0)).sum() * torch.sqrt((Xt * Xt).sum())
        return vw

    def volume(self, c, h):
        Z = torch.einsum("gma,gak->gmk", self.Rb[:, 1:], c[self.ids])
        t = (self.N0 * h[self.ids]).sum(-1)
        return (self.wq * ke.area_jacobian(Z) * t).sum()

    def compliance(self, c, U, forces):
        """C = sum int forces . u_hom dA with the non-rationalised displacement function."""
        Z = torch.einsum("gma,gak->gmk", self.Rb[:, 1:], c[self.ids])
        uh = torch.einsum("ga,gak->gk", self.N0, U[self.ids])
        f = torch.tensor(np.asarray(forces, float).reshape(-1, 3))[self.pid]
        return (self.wq * ke.area_jacobian(Z) * (f * uh).sum(-1)).sum()

    def stress_forms(self, c, U, h, mode, rho, m_list, sgn, measure):
        """Per-patch forms int g(sigma_vM) dA (max_vmstress_exop.py:167-175) from the pointwise torch statement."""
        Z = torch.einsum("gma,gak->gmk", self.Rb[:, 1:], c[self.ids])
        z = torch.einsum("gma,gak->gmk", self.Rb[:, 1:], (c + U)[self.ids])
        t = (self.N0 * h[self.ids]).sum(-1)
        J = ke.area_jacobian(Z)
        out = [torch.zeros((), dtype=torch.float64) for _ in m_list]
        for g in range(z.shape[0]):
            s = int(self.pid[g])
            sig = ke.von_mises_stress(z[g], Z[g], t[g], self.E[g], self.nu[g], sgn, measure)
            val = torch.exp(rho * (sig - m_list[s])) if mode == 0 else (sig / m_list[s]) ** rho
            out[s] = out[s] + self.wq[g] * J[g] * val
        return torch.stack(out)

    def shape_regu(self, c, field, cp0, coef):
        """sum_s coef_s int |grad_s(P_f - P_f^0)|^2 dA with the surface gradient through the pseudo-inverse of DF (tIGAr
        spline.grad on a manifold), demos_om/shape_opt/eVTOL/int_energy_regu_exop.py:30-38."""
        Z = torch.einsum("gma,gak->gmk", self.Rb[:, 1:3], c[self.ids])             # (g, 2, 3): G1, G2
        dc = (c[:, field] - torch.as_tensor(cp0))[self.ids]
        D = torch.einsum("gma,ga->gm", self.Nb12, dc)                             # non-rational derivatives
        DF = Z.transpose(1, 2)                                                    # (g, 3, 2)
        grad = torch.einsum("gm,gmk->gk", D, torch.linalg.pinv(DF))               # D_,a (DF^+)_a,k
        J = ke.area_jacobian(Z)
        return (self.wq * torch.as_tensor(coef)[self.pid] * (grad * grad).sum(-1) * J).sum()

    def penalty_energy(self, c, U):
        W = torch.zeros((), dtype=torch.float64)
        for ia, RA, ib, RB, tau, ad, ar, wt in self.mp:
            uA = RA[0] @ U[ia]
            uB = RB[0] @ U[ib]
            gA = RA[1:] @ (c + U)[ia]
            gB = RB[1:] @ (c + U)[ib]
            GA = RA[1:] @ c[ia]
            GB = RB[1:] @ c[ib]
            W = W + ke.penalty_energy_point(uA, gA, uB, gB, GA, GB, tau, ad, ar, wt)
        return W

    def total(self, c, U, h):
        W = self.shell_energy(c, U, h)
        if self.mp:
            W = W + self.penalty_energy(c, U)
        return W


# ---- analytic d(penalty residual)/d(xi) of one interface by autograd (moving intersections) ----------------------
def _bspline_torch(span, xi, p, U):
    """Values and first derivatives (torch, differentiable in xi) of the p + 1 B-spline functions that are non-zero on the span:
    Cox-de Boor triangle for degrees 0 .. p, derivative from degree p - 1 (The NURBS Book, eq. 2.7)."""
    N = [[None] * (p + 1) for _ in range(p + 1)]          # N[d][j]: function span - d + j of degree d
    N[0][0] = torch.ones((), dtype=torch.float64)
    for d in range(1, p + 1):
        for j in range(d + 1):
            i = span - d + j
            v = torch.zeros((), dtype=torch.float64)
            if j > 0 and U[i + d] > U[i]:
                v = v + (xi - U[i]) / (U[i + d] - U[i]) * N[d - 1][j - 1]
            if j < d and U[i + d + 1] > U[i + 1]:
                v = v + (U[i + d + 1] - xi) / (U[i + d + 1] - U[i + 1]) * N[d - 1][j]
            N[d][j] = v
    val = torch.stack(N[p])
    der = []
    for j in range(p + 1):
        i = span - p + j
        v = torch.zeros((), dtype=torch.float64)
        if j > 0 and U[i + p] > U[i]:
            v = v + p / (U[i + p] - U[i]) * N[p - 1][j - 1]
        if j < p and U[i + p + 1] > U[i + 1]:
            v = v - p / (U[i + p + 1] - U[i + 1]) * N[p - 1][j]
        der.append(v)
    return val, torch.stack(der)


def penalty_residual_dxi(patches, cp_off, pa, pb, weights, c, U, xi_a, xi_b, alpha, wt, zero_dofs):
    """J[:, k] = d(penalty residual)/d(xi_flat[k]) for ONE interface between patches pa, pb, xi_flat = [xi_A (n x 2) | xi_B (n x 2)]:
    the interface energy as a torch function of (U, xi) -- rational basis values and first derivatives at the vertices, the curve
    tangent tau = G xi_A (second-order differences, model.Interface), Herrema's vertex energy (oracle/kl_energy_torch.py) with
    frozen parameters alpha and vertex weights wt -- differentiated once in U and once in xi by autograd.  Dirichlet rows zeroed."""
    n = xi_a.shape[0]
    xi = torch.tensor(np.concatenate([xi_a.ravel(), xi_b.ravel()]), dtype=torch.float64, requires_grad=True)
    Ut = torch.tensor(U.reshape(-1, 3), dtype=torch.float64, requires_grad=True)
    ct = torch.tensor(c, dtype=torch.float64)
    G = torch.tensor(np.gradient(np.eye(n), 1.0 / (n - 1), axis=0, edge_order=2 if n > 2 else 1), dtype=torch.float64)
    XA, XB = xi[:2 * n].reshape(n, 2), xi[2 * n:].reshape(n, 2)
    tau = G @ XA
    E = torch.zeros((), dtype=torch.float64)
    for v in range(n):
        side = []
        for s, X in ((pa, XA), (pb, XB)):
            P = patches[s]
            su = find_span(P.n_u, P.p, P.knots[0], float(X[v, 0]))
            sv = find_span(P.n_v, P.q, P.knots[1], float(X[v, 1]))
            nu, du = _bspline_torch(su, X[v, 0], P.p, P.knots[0])
            nv, dv = _bspline_torch(sv, X[v, 1], P.q, P.knots[1])
            ids = np.array([cp_off[s] + P.flat(su - P.p + ju, sv - P.q + jv) for jv in range(P.q + 1) for ju in range(P.p + 1)])
            w = torch.tensor(weights[ids], dtype=torch.float64)
            N0 = (nv[:, None] * nu[None, :]).reshape(-1)
            N1 = (nv[:, None] * du[None, :]).reshape(-1)
            N2 = (dv[:, None] * nu[None, :]).reshape(-1)
            W0, W1, W2 = (N0 * w).sum(), (N1 * w).sum(), (N2 * w).sum()
            R = N0 / W0
            R1, R2 = (N1 - R * W1) / W0, (N2 - R * W2) / W0
            Rg = torch.stack([R1, R2])
            side.append((R @ Ut[ids], Rg @ (ct[ids] + Ut[ids]), Rg @ ct[ids]))
        (uA, gA, GA), (uB, gB, GB) = side
        E = E + ke.penalty_energy_point(uA, gA, uB, gB, GA, GB, tau[v], alpha[0], alpha[1], wt[v])
    gxi = torch.autograd.grad(E, xi, create_graph=True)[0]
    J = np.zeros((Ut.numel(), 4 * n))
    for k in range(4 * n):
        J[:, k] = torch.autograd.grad(gxi[k], Ut, retain_graph=True)[0].reshape(-1).numpy()
    J[np.asarray(zero_dofs, dtype=np.int64)] = 0.0
    return J
