"""TEST INFRASTRUCTURE ONLY -- readable NumPy twin of the *pointwise* closed forms the
HIP kernels implement (goldfish_amd/csrc/kl_point.hpp), checked against torch.autograd
in tests/test_pointwise_forms.py.  It documents the derivation; nothing under
goldfish_amd/ imports it.

Shell: Psi(z, Z, t) = psi_SVK * |G1 x G2| with z = (g1,g2,h11,h22,h12), Z likewise
(5x3 each, flattened index 3*m+i).  Needed per Gauss point:
    Pz  = dPsi/dz           (15)      -> internal force
    Pzz = d2Psi/dz dz       (15x15)   -> tangent K            (phi_a^T Pzz[i,j] phi_b)
    PzZ = d2Psi/dz dZ       (15x15)   -> reference part of dR/dCP
    Pzt = d2Psi/dz dt       (15)      -> dR/dh
Penalty: pi(y, Y) with y = (uA, gA1, gA2, uB, gB1, gB2) (18), Y = (GA1, GA2, GB1, GB2) (12).
"""
import numpy as np

F3 = np.array([1.0, 1.0, 2.0])


def skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0.0]])


def normal_and_derivs(g1, g2):
    """n, j=|g1xg2|, Dn (3x6) = dn_i/d(g1,g2)_j."""
    nt = np.cross(g1, g2)
    j = np.linalg.norm(nt)
    n = nt / j
    P = (np.eye(3) - np.outer(n, n)) / j
    B = np.hstack([-skew(g2), skew(g1)])          # d(g1 x g2)/d(g1,g2), 3x6
    return n, j, P @ B, B


def hess_M_dot_n(g1, g2, n, j, B, M):
    """6x6 Hessian of M . n(g1,g2) for a fixed vector M."""
    Mn = M @ n
    Q = -(np.outer(M, n) + np.outer(n, M) + Mn * (np.eye(3) - 3 * np.outer(n, n))) / (j * j)
    v = (M - Mn * n) / j
    H = B.T @ Q @ B
    H[0:3, 3:6] += -skew(v)
    H[3:6, 0:3] += skew(v)
    return H


def material(G1, G2, E, nu):
    A11, A22, A12 = G1 @ G1, G2 @ G2, G1 @ G2
    det = A11 * A22 - A12 * A12
    c11, c22, c12 = A22 / det, A11 / det, -A12 / det
    Eb = E / (1 - nu * nu)

    def Cof(c11, c22, c12):
        return Eb * np.array([[c11 * c11, nu * c11 * c22 + (1 - nu) * c12 * c12, c11 * c12],
                              [nu * c11 * c22 + (1 - nu) * c12 * c12, c22 * c22, c22 * c12],
                              [c11 * c12, c22 * c12, 0.5 * ((1 - nu) * c11 * c22 + (1 + nu) * c12 * c12)]])
    C = Cof(c11, c22, c12)
    # dC/dA_q, q = (A11, A22, A12) via d(c11,c22,c12)/dA_q
    dc = np.array([[-c11 * c11, -c12 * c12, -2 * c11 * c12],
                   [-c12 * c12, -c22 * c22, -2 * c12 * c22],
                   [-c11 * c12, -c12 * c22, -(c11 * c22 + c12 * c12)]])     # rows c11,c22,c12; cols q
    dC_dc = [Eb * np.array([[2 * c11, nu * c22, c12], [nu * c22, 0, 0], [c12, 0, 0.5 * (1 - nu) * c22]]),
             Eb * np.array([[0, nu * c11, 0], [nu * c11, 2 * c22, c12], [0, c12, 0.5 * (1 - nu) * c11]]),
             Eb * np.array([[0, 2 * (1 - nu) * c12, c11], [2 * (1 - nu) * c12, 0, c22], [c11, c22, (1 + nu) * c12]])]
    dC = [sum(dC_dc[p] * dc[p, q] for p in range(3)) for q in range(3)]
    return C, dC, np.sqrt(det)


def metric_grad(g1, g2):
    """d m_v / d(g1,g2): 3 x 6 with m_v = [g1.g1, g2.g2, 2 g1.g2]."""
    mz = np.zeros((3, 6))
    mz[0, 0:3] = 2 * g1
    mz[1, 3:6] = 2 * g2
    mz[2, 0:3] = 2 * g2
    mz[2, 3:6] = 2 * g1
    return mz


def curvature_grad(zz, n, Dn):
    """d beta_v / dz : 3 x 15, beta_k = f_k h_k . n."""
    bz = np.zeros((3, 15))
    for k in range(3):
        bz[k, 0:6] = F3[k] * (zz[2 + k] @ Dn)
        bz[k, 6 + 3 * k:9 + 3 * k] = F3[k] * n
    return bz


def shell_point(z, Z, t, E, nu):
    z, Z = z.reshape(5, 3), Z.reshape(5, 3)
    n, j, Dn, B = normal_and_derivs(z[0], z[1])
    N, Jn, DN, BN = normal_and_derivs(Z[0], Z[1])
    C, dC, J = material(Z[0], Z[1], E, nu)
    mv = lambda w: np.array([w[0] @ w[0], w[1] @ w[1], 2 * w[0] @ w[1]])
    bv = lambda w, nn: F3 * np.array([w[2] @ nn, w[3] @ nn, w[4] @ nn])
    eps = 0.5 * (mv(z) - mv(Z))
    kap = bv(Z, N) - bv(z, n)
    t3 = t ** 3 / 12
    Ce, Ck = C @ eps, C @ kap
    nv, mo = t * Ce, t3 * Ck
    psi = 0.5 * t * eps @ Ce + 0.5 * t3 * kap @ Ck
    ez = np.zeros((3, 15))
    ez[:, 0:6] = 0.5 * metric_grad(z[0], z[1])          # d eps/dz
    bz = curvature_grad(z, n, Dn)                        # d beta/dz  (kappa_z = -bz)
    eZ = np.zeros((3, 15))
    eZ[:, 0:6] = -0.5 * metric_grad(Z[0], Z[1])          # d eps/dZ
    bZ = curvature_grad(Z, N, DN)                        # d kappa/dZ = +bZ
    Pz = J * (nv @ ez - mo @ bz)
    Pzt = J * (Ce @ ez - 0.25 * t * t * (Ck @ bz))
    # ---- Pzz
    Pzz = J * (t * ez.T @ C @ ez + t3 * bz.T @ C @ bz)
    Ngeo = np.array([[nv[0], nv[2]], [nv[2], nv[1]]])
    for a in range(2):
        for b in range(2):
            Pzz[3 * a:3 * a + 3, 3 * b:3 * b + 3] += J * Ngeo[a, b] * np.eye(3)
    M = sum(mo[k] * F3[k] * z[2 + k] for k in range(3))
    Pzz[0:6, 0:6] -= J * hess_M_dot_n(z[0], z[1], n, j, B, M)
    for k in range(3):
        blk = J * mo[k] * F3[k] * Dn                     # 3 x 6: d2 beta_k / dh_k dg
        Pzz[6 + 3 * k:9 + 3 * k, 0:6] -= blk
        Pzz[0:6, 6 + 3 * k:9 + 3 * k] -= blk.T
    # ---- PzZ
    JZ = np.zeros(15)
    JZ[0:3] = np.cross(Z[1], N)
    JZ[3:6] = np.cross(N, Z[0])
    AqZ = np.zeros((3, 15))                              # d(A11, A22, A12)/dZ
    AqZ[0, 0:3] = 2 * Z[0]
    AqZ[1, 3:6] = 2 * Z[1]
    AqZ[2, 0:3] = Z[1]
    AqZ[2, 3:6] = Z[0]
    dnv = t * (sum(np.outer(dC[q] @ eps, AqZ[q]) for q in range(3)) + C @ eZ)     # 3 x 15
    dmo = t3 * (sum(np.outer(dC[q] @ kap, AqZ[q]) for q in range(3)) + C @ bZ)
    PzZ = np.outer(Pz, JZ) / J + J * (ez.T @ dnv - bz.T @ dmo)
    return dict(Psi=psi * J, J=J, JZ=JZ, Pz=Pz, Pzz=Pzz, PzZ=PzZ, Pzt=Pzt)


# ------------------------------------------------------------------ penalty
def tangent_and_derivs(g1, g2, tau):
    tt = tau[0] * g1 + tau[1] * g2
    L = np.linalg.norm(tt)
    at = tt / L
    Pt = (np.eye(3) - np.outer(at, at)) / L
    Bt = np.hstack([tau[0] * np.eye(3), tau[1] * np.eye(3)])
    return at, L, Pt @ Bt, Bt


def hess_M_dot_t(at, L, Bt, M):
    Ma = M @ at
    Q = -(np.outer(M, at) + np.outer(at, M) + Ma * (np.eye(3) - 3 * np.outer(at, at))) / (L * L)
    return Bt.T @ Q @ Bt


def _s_terms(gA, gB, tau, second):
    """s1 = nA.nB, s2 = at.(nA x nB): values, gradients (12) and Hessians (12x12) wrt (gA1,gA2,gB1,gB2)."""
    nA, jA, DnA, BA = normal_and_derivs(gA[0], gA[1])
    nB, jB, DnB, BB = normal_and_derivs(gB[0], gB[1])
    at, L, Dt, Bt = tangent_and_derivs(gA[0], gA[1], tau)
    s1 = nA @ nB
    s2 = at @ np.cross(nA, nB)
    g1 = np.concatenate([DnA.T @ nB, DnB.T @ nA])
    g2 = np.concatenate([Dt.T @ np.cross(nA, nB) + DnA.T @ np.cross(nB, at), DnB.T @ np.cross(at, nA)])
    if not second:
        return s1, s2, g1, g2, None, None, L, at
    H1 = np.zeros((12, 12))
    H1[0:6, 0:6] = hess_M_dot_n(gA[0], gA[1], nA, jA, BA, nB)
    H1[6:12, 6:12] = hess_M_dot_n(gB[0], gB[1], nB, jB, BB, nA)
    H1[0:6, 6:12] = DnA.T @ DnB
    H1[6:12, 0:6] = H1[0:6, 6:12].T
    H2 = np.zeros((12, 12))
    X = -Dt.T @ skew(nB) @ DnA
    H2[0:6, 0:6] = (hess_M_dot_t(at, L, Bt, np.cross(nA, nB)) + hess_M_dot_n(gA[0], gA[1], nA, jA, BA, np.cross(nB, at))
                    + X + X.T)
    H2[0:6, 6:12] = (Dt.T @ skew(nA) - DnA.T @ skew(at)) @ DnB
    H2[6:12, 0:6] = H2[0:6, 6:12].T
    H2[6:12, 6:12] = hess_M_dot_n(gB[0], gB[1], nB, jB, BB, np.cross(at, nA))
    return s1, s2, g1, g2, H1, H2, L, at


def penalty_point(y, Y, tau, ad, ar, dt):
    uA, gA, uB, gB = y[0:3], y[3:9].reshape(2, 3), y[9:12], y[12:18].reshape(2, 3)
    GA, GB = Y[0:6].reshape(2, 3), Y[6:12].reshape(2, 3)
    s1, s2, g1, g2, H1, H2, _, _ = _s_terms(gA, gB, tau, True)
    S1, S2, G1, G2, _, _, L, At = _s_terms(GA, GB, tau, False)
    e1, e2 = s1 - S1, s2 - S2
    c0 = dt * L
    d = uA - uB
    en = c0 * (0.5 * ad * d @ d + 0.5 * ar * (e1 * e1 + e2 * e2))
    tan = [3, 4, 5, 6, 7, 8, 12, 13, 14, 15, 16, 17]          # tangent slots of y
    grad = np.zeros(18)
    grad[0:3], grad[9:12] = c0 * ad * d, -c0 * ad * d
    grad[tan] = c0 * ar * (e1 * g1 + e2 * g2)
    Hyy = np.zeros((18, 18))
    I3 = np.eye(3)
    Hyy[0:3, 0:3] = Hyy[9:12, 9:12] = c0 * ad * I3
    Hyy[0:3, 9:12] = Hyy[9:12, 0:3] = -c0 * ad * I3
    Hyy[np.ix_(tan, tan)] = c0 * ar * (np.outer(g1, g1) + e1 * H1 + np.outer(g2, g2) + e2 * H2)
    c0Y = np.zeros(12)
    c0Y[0:3], c0Y[3:6] = dt * tau[0] * At, dt * tau[1] * At
    HyY = np.outer(grad, c0Y) / c0
    HyY[tan, :] -= c0 * ar * (np.outer(g1, G1) + np.outer(g2, G2))
    return en, grad, Hyy, HyY
