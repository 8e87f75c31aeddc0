"""CPU tests of the host-side logic and of the C-ABI surface (no compute calls)."""
import ctypes
import os
import re

import numpy as np
import pytest

from goldfish_amd import geometry as G
from goldfish_amd import sharding
from goldfish_amd.model import Interface, arrays_from_spec
from goldfish_amd.splines import NURBSPatch, open_uniform_knots

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cylinder_sector_is_exact_and_refinement_preserves_geometry():
    s = NURBSPatch.cylinder_sector(25.0, np.radians(50), np.radians(130), 0.0, 50.0, 1, 1, 3)
    f = NURBSPatch.cylinder_sector(25.0, np.radians(50), np.radians(130), 0.0, 50.0, 7, 5, 3)
    rng = np.random.default_rng(0)
    for xi in rng.uniform(0, 1, (20, 2)):
        X, Y = s.eval(xi), f.eval(xi)
        assert abs(np.hypot(X[0], X[1]) - 25.0) < 1e-12
        assert np.abs(X - Y).max() < 1e-12
    assert f.n_u == 7 + 3 and f.n_v == 5 + 3 and f.nel == (7, 5)
    assert np.ptp(f.control[:, :, 3]) > 1e-3          # truly rational


def test_bilinear_patch_and_inversion():
    s = NURBSPatch.bilinear([[0, 0, 0], [2, 0, 0], [0, 3, 1], [2, 3, 1]], 4, 5, 3)
    xi = np.array([0.3, 0.7])
    X = s.eval(xi)
    assert np.allclose(X, [0.6, 2.1, 0.7], atol=1e-13)
    assert np.allclose(s.invert(X, (0.5, 0.5)), xi, atol=1e-10)
    assert s.get_side_dofs(1, 0, 1) == list(range(s.n_u))


def test_interface_weights_and_tangent():
    itf = Interface.from_endpoints(0, 1, [[1.0, 0.0], [1.0, 1.0]], [[0.0, 0.0], [0.0, 1.0]], 10)
    assert itf.npts == 11 and abs(itf.wt.sum() - 1.0) < 1e-14
    assert np.allclose(itf.tau, [[0.0, 1.0]] * 11)


def test_model_arrays_layout():
    spec = G.tbeam_2patch(4)
    A = arrays_from_spec(spec)
    assert A.total_cp == sum(p.ncp for p in spec.patches) and A.ndof == 3 * A.total_cp
    assert A.n_gauss_points == sum(p.nel[0] * p.nel[1] * 16 for p in spec.patches)
    p0 = spec.patches[0]
    assert np.allclose(A.cp_hom[0][:p0.ncp].reshape(p0.n_v, p0.n_u).T, p0.control[:, :, 0])   # u-index fastest
    assert len(A.zero_dofs) == 3 * (p0.n_u + spec.patches[1].n_u)
    assert A.if_alpha[0] > A.if_alpha[1] > 0


def test_library_exports_every_declared_symbol():
    """The C-ABI library loads and exports exactly what include/goldfish_hip.h declares."""
    from goldfish_amd import _lib, build
    build.build()
    hdr = open(os.path.join(ROOT, "include", "goldfish_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(gf_[a-z_]+)\s*\(", hdr)))
    assert declared == sorted(_lib.EXPORTS)
    L = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(L, name), name
    _lib.lib()


def test_partition_and_shards():
    spec = G.synthetic_shell(4, 3, nel=4, p=3, jitter=1)
    for world in (1, 2, 3, 4):
        parts = sharding.partition_patches(spec, world)
        assert parts[0][0] == 0 and parts[-1][1] == 12 and all(a < b for a, b in parts)
        assert all(parts[r][1] == parts[r + 1][0] for r in range(world - 1))
    sh = sharding.shard_spec(spec, 1, 2)
    own = set(sh.order[:sh.n_owned])
    for itf_g in spec.interfaces:
        if (itf_g.a in own) != (itf_g.b in own):
            assert {itf_g.a, itf_g.b} <= set(sh.order)
    g0, g1 = sh.owned_global_range(3)
    assert g1 - g0 == sh.owned_local_size(3)
    v = np.arange(3 * sh.total_cp_global, dtype=float)
    assert np.array_equal(sh.to_local(v, 3)[:g1 - g0], v[g0:g1])


def test_no_gpu_means_loud_failure():
    """The product path must fail loudly (never fall back) when no GPU / extension is present."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from goldfish_amd import _lib
    spec = G.tbeam_2patch(4)
    with pytest.raises(RuntimeError):
        _lib.DeviceModel(arrays_from_spec(spec))


def test_ffd_block_reproduces_surface_control_points():
    """N2: the FFD operator of an undeformed block is the identity on the surface control points
    (reference assumption, GOLDFISH/utils/ffd_utils.py:37-39) and its columns sum to one."""
    from goldfish_amd.utils.ffd_utils import CP_FFD_matrix, create_3D_block
    rng = np.random.default_rng(1)
    X = rng.uniform([-1, 0, -2], [1, 20, 0], (50, 3))
    blk = create_3D_block([3, 2, 1], 3, [[-1, 1], [0, 20], [-2, 0]])
    D = CP_FFD_matrix(X, blk.degree, blk.knots).tocsr()
    q = blk.control[..., 0:3].transpose(2, 1, 0, 3).reshape(-1, 3)       # i + j*l + k*l*m ordering
    assert D.shape == (50, q.shape[0])
    assert np.abs(D @ q - X).max() < 1e-12
    assert np.abs(np.asarray(D.sum(1)).ravel() - 1.0).max() < 1e-12
    flat = create_3D_block([2, 2, 1], 2, [[0, 1], [0, 1], [0.5, 0.5]])    # degenerate direction gets thickened
    assert flat.knots[2][-1] > flat.knots[2][0]
