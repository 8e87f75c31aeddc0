"""GPU parity tests proper: libgoldfish_hip.so (through the C ABI) vs the CPU oracle on the
same seeded inputs.  FP64 tolerance 1e-10 relative (BASELINE.json north_star); only
summation order differs, observed ~1e-14."""
import os

import numpy as np
import pytest

from goldfish_amd import geometry as G
from goldfish_amd.model import arrays_from_spec

pytestmark = pytest.mark.gpu
RTOL = 1e-10


def _state(spec, seed=0, uamp=2e-2):
    rng = np.random.default_rng(seed)
    th = [spec.h_th * rng.uniform(0.8, 1.2, p.ncp) for p in spec.patches]
    A = arrays_from_spec(spec, th)
    return A, np.concatenate(th), uamp * rng.standard_normal(A.ndof)


def _rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


CASES = {
    "tbeam2_p3": lambda: G.tbeam_2patch(4),
    "slr9_nurbs_p3": lambda: G.scordelis_lo_9patch(3, nels=[2, 1, 2, 3, 2, 3, 2, 1, 2]),
    "slr9_p3": lambda: G.scordelis_lo_9patch(6),
    "plate6_p3": lambda: G.plate_6patch(),
    "tbeam2_p2": lambda: G.tbeam_2patch(4, p=2),
    "shell3x2_p3": lambda: G.synthetic_shell(3, 2, nel=5, p=3, jitter=1),
    "shell2x2_p4": lambda: G.synthetic_shell(2, 2, nel=4, p=4, jitter=1),
    "C2_tbeam4_10kdof": lambda: G.tbeam_4patch(),
    "C3_wing16_refdata": lambda: G.wing_16patch_from_interface_data(
        np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_wing_int_data.npz"), allow_pickle=True)),
    # C3 at the size BASELINE.json names: the reference's 16-patch / 62-interface wing topology with 42 .. 46 spans per patch side (~101 k dofs, ~0.47 M Gauss points)
    "C3_wing16_100kdof": lambda: G.wing_16patch_from_interface_data(
        np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_wing_int_data.npz"), allow_pickle=True), nel=44),
    "C5_fuselage3x2_p4": lambda: G.synthetic_fuselage(3, 2, nel=6, p=4, jitter=1),
    "shell2x2_p4_nel11": lambda: G.synthetic_shell(2, 2, nel=11, p=4, jitter=1),      # strips long enough for two segments of >= 5 elements (row records, p = 4)
    # load per unit projected area along a general direction (gf_model_desc.load_proj; the arch demo's source term)
    "slr9_nurbs_p3_projected_load": lambda: _with_projected_load(G.scordelis_lo_9patch(3, nels=[2, 1, 2, 3, 2, 3, 2, 1, 2])),
    "shell2x2_p3_double_knots": lambda: G.with_double_knots(G.synthetic_shell(2, 2, nel=6, p=3, jitter=1)),
    "slr9_p2_projected_load": lambda: _with_projected_load(G.scordelis_lo_9patch(4, p=2)),
    "shell2x2_p4_projected_load": lambda: _with_projected_load(G.synthetic_shell(2, 2, nel=4, p=4, jitter=1)),
    # follower pressure (tube demo: R, non-symmetric load stiffness, dR/dCP) + dead edge tractions (plate demo: R, dR/dCP): gf_extra_loads.hpp
    "slr9_nurbs_p3_pressure_edge": lambda: _with_pressure_edge(G.scordelis_lo_9patch(3, nels=[2, 1, 2, 3, 2, 3, 2, 1, 2])),
    "tbeam2_p2_pressure_edge": lambda: _with_pressure_edge(G.tbeam_2patch(4, p=2)),
    "shell2x2_p4_pressure_edge": lambda: _with_pressure_edge(G.synthetic_shell(2, 2, nel=4, p=4, jitter=1)),
    "shell3x2_p3_double_knots_pressure_edge": lambda: _with_pressure_edge(G.with_double_knots(G.synthetic_shell(3, 2, nel=6, p=3, jitter=1))),
}


def _with_pressure_edge(spec):
    from tests.golden.make_golden import with_pressure_and_edge_tractions
    return with_pressure_and_edge_tractions(spec)


def _with_projected_load(spec):
    spec.load_proj = [[0.3, -0.2, 1.0]] * len(spec.patches)
    return spec


@pytest.mark.parametrize("case", list(CASES))
def test_assembly_parity(oracle_lib, case):
    from goldfish_amd import _lib
    from oracle.oracle_py import Oracle
    spec = CASES[case]()
    A, h, u = _state(spec)
    O = Oracle(A, thickness=h, u=u)
    D = _lib.DeviceModel(A)
    # p = 2, 3: walking MFMA kernel + row records + record gather; p = 4: hybrid (Newton passes through the row records of gf_element_rec4.hpp, passes with
    # dR/dCP / dR/dh through element blocks: each pass kind on its faster path)
    assert D.assembly_path == (6 if int(A.degree[0]) == 4 else 4)
    D.set_thickness(h)
    D.set_u(u)
    D.assemble(_lib.ASM_ALL)
    # static patterns identical to the oracle's
    for which in range(5):
        rp, col = D.pattern(which)
        rpo, colo = O.pattern(which)
        assert np.array_equal(rp, rpo) and np.array_equal(col, colo), "pattern %d differs" % which
    assert _rel(D.residual(), O.residual()) < RTOL
    vals = O.assemble()
    for which in range(5):
        assert _rel(D.values(which), vals[which]) < RTOL, "matrix %d" % which
    # residual-only and matrix-only passes give the same answers
    D.assemble(_lib.ASM_R)
    assert _rel(D.residual(), O.residual()) < RTOL
    D.assemble(_lib.ASM_K)
    assert _rel(D.values(0), vals[0]) < RTOL
    D.assemble(_lib.ASM_DRDCP)
    for which in (1, 2, 3):
        assert _rel(D.values(which), vals[which]) < RTOL, "dR/dCP-only pass, matrix %d" % which
    D.assemble(_lib.ASM_DRDH)
    assert _rel(D.values(4), vals[4]) < RTOL
    D.close()


@pytest.mark.parametrize("seg", ["4", "7", "1000"])
def test_row_record_path_segment_lengths_and_block_path(oracle_lib, monkeypatch, seg):
    """Default path for p = 2, 3: the walking element kernel that stores row records (gf_element_rec.hpp: a control-point pair is stored once per
    strip and segment, when its lower row leaves the window) + the record gather, with work items of 4 / 7 elements / whole strips
    (a pair then lies in two / one segment), against the oracle and the element-block + gather path, for every flag subset;
    bitwise reproducible run to run (fixed strip and segment order per entry)."""
    from goldfish_amd import _lib
    from oracle.oracle_py import Oracle
    for case in ("tbeam2_p2", "shell3x2_p3", "C3_wing16_refdata", "slr9_nurbs_p3_projected_load", "shell2x2_p3_double_knots", "shell2x2_p4_nel11", "shell2x2_p4_projected_load"):
        A, h, u = _state(CASES[case](), seed=11)
        deg = int(A.degree[0])
        O = Oracle(A, thickness=h, u=u)
        vals, Ro = O.assemble(), O.residual()
        out = {}
        for walk in ("2", "0"):
            if walk == "2": monkeypatch.setenv("GF_ASSEMBLY", "rec")              # row records for every pass (the default for p = 2, 3; p = 4: the low-memory path)
            else: monkeypatch.setenv("GF_ASSEMBLY", "block")                    # one block per element + row gather: the cross-check path
            monkeypatch.setenv("GF_REC_SEG", seg)
            D = _lib.DeviceModel(A)
            assert D.assembly_path == ((5 if deg == 4 else 4) if walk == "2" else 0)
            D.set_thickness(h)
            D.set_u(u)
            D.assemble(_lib.ASM_ALL)
            out[walk] = [D.residual().copy()] + [D.values(w).copy() for w in range(5)]
            assert _rel(out[walk][0], Ro) < RTOL
            for w in range(5):
                assert _rel(out[walk][1 + w], vals[w]) < RTOL, (case, walk, w)
            D.assemble(_lib.ASM_ALL)
            assert np.array_equal(out[walk][0], D.residual())
            for w in range(5):
                assert np.array_equal(out[walk][1 + w], D.values(w)), (case, walk, w)
            for flags, which in ((_lib.ASM_R | _lib.ASM_K, (0,)), (_lib.ASM_DRDCP | _lib.ASM_DRDH, (1, 2, 3, 4)), (_lib.ASM_K | _lib.ASM_DRDH, (0, 4))):
                D.assemble(flags)
                for w in which:
                    assert _rel(D.values(w), vals[w]) < RTOL, (case, walk, flags, w)
            D.close()
        for x, y in zip(out["2"], out["0"]):
            assert _rel(x, y) < 1e-12


def test_penalty_kernel_variants(oracle_lib, monkeypatch):
    """Two penalty kernel pairs: pen_point16_kernel + pen_row16_kernel (default for p = 2, 3: 16 lanes per mortar vertex / one 16-lane row per
    visit, LDS accumulators) and pen_point_kernel + pen_owner_kernel (p = 4; GF_PENALTY=owner for p = 2, 3: the cross-check).  Both against
    the oracle; each is bitwise reproducible run to run."""
    from goldfish_amd import _lib
    from oracle.oracle_py import Oracle
    for case in ("tbeam2_p2", "C3_wing16_refdata", "slr9_nurbs_p3_projected_load"):
        A, h, u = _state(CASES[case](), seed=5)
        O = Oracle(A, thickness=h, u=u)
        vals, Ro = O.assemble(), O.residual()
        for row16, point16 in (("0", "0"), ("1", "1")):
            if row16 == "0": monkeypatch.setenv("GF_PENALTY", "owner")
            else: monkeypatch.delenv("GF_PENALTY", raising=False)
            D = _lib.DeviceModel(A)
            D.set_thickness(h)
            D.set_u(u)
            D.assemble(_lib.ASM_ALL)
            first = [D.residual().copy()] + [D.values(w).copy() for w in range(5)]
            assert _rel(first[0], Ro) < RTOL
            for w in range(5):
                assert _rel(first[1 + w], vals[w]) < RTOL, (case, row16, point16, w)
            D.assemble(_lib.ASM_ALL)
            assert np.array_equal(first[0], D.residual())
            for w in range(5):
                assert np.array_equal(first[1 + w], D.values(w)), (case, row16, point16, w)
            for flags, which in ((_lib.ASM_R | _lib.ASM_K, (0,)), (_lib.ASM_DRDCP | _lib.ASM_DRDH, (1, 2, 3, 4)), (_lib.ASM_R, ())):
                D.assemble(flags)
                assert _rel(D.residual(), Ro) < RTOL or not (flags & _lib.ASM_R)
                for w in which:
                    assert _rel(D.values(w), vals[w]) < RTOL, (case, row16, point16, flags, w)
            D.close()


def test_p3_contraction_variants(oracle_lib, monkeypatch):
    """The p = 3 walking kernel contracts either with the 16 x 16 x 4 products of rounds 2 - 4 (GF_SUMFACT=0) or with the row-side sum factorisation
    (v_mfma_f64_4x4x4 as inline assembly with hand-placed interlocks, accumulators parked in AGPRs): on polynomial patches only (=1) or on rational
    patches too (=2, default).  Every variant against the oracle, for every pass kind, on models with polynomial patches, rational patches, both,
    repeated knots, a projected load, follower pressure -- and bitwise reproducible run to run."""
    from goldfish_amd import _lib
    from oracle.oracle_py import Oracle
    for case in ("shell3x2_p3", "C3_wing16_refdata", "slr9_nurbs_p3_projected_load", "shell3x2_p3_double_knots_pressure_edge"):
        A, h, u = _state(CASES[case](), seed=11)
        O = Oracle(A, thickness=h, u=u)
        vals, Ro = O.assemble(), O.residual()
        for variant in ("0", "1", "2"):
            monkeypatch.setenv("GF_SUMFACT", variant)
            D = _lib.DeviceModel(A)
            D.set_thickness(h)
            D.set_u(u)
            D.assemble(_lib.ASM_ALL)
            first = [D.residual().copy()] + [D.values(w).copy() for w in range(5)]
            assert _rel(first[0], Ro) < RTOL, (case, variant)
            for w in range(5):
                assert _rel(first[1 + w], vals[w]) < RTOL, (case, variant, w)
            D.assemble(_lib.ASM_ALL)
            for w in range(5):
                assert np.array_equal(first[1 + w], D.values(w)), (case, variant, w)
            for flags, which in ((_lib.ASM_R | _lib.ASM_K, (0,)), (_lib.ASM_DRDCP | _lib.ASM_DRDH, (1, 2, 3, 4)), (_lib.ASM_K | _lib.ASM_DRDCP, (0, 1, 2, 3))):
                D.assemble(flags)
                for w in which:
                    assert _rel(D.values(w), vals[w]) < RTOL, (case, variant, flags, w)
            D.close()
    monkeypatch.delenv("GF_SUMFACT", raising=False)


def test_valu_element_kernel_still_matches(oracle_lib, monkeypatch):
    """The FP64-VALU element kernel (GF_ELEMENT=valu) stays a supported path for every degree."""
    from goldfish_amd import _lib
    from oracle.oracle_py import Oracle
    monkeypatch.setenv("GF_ELEMENT", "valu")
    for case in ("tbeam2_p2", "shell3x2_p3", "shell2x2_p4", "slr9_nurbs_p3_projected_load"):
        A, h, u = _state(CASES[case]())
        O = Oracle(A, thickness=h, u=u)
        D = _lib.DeviceModel(A)
        D.set_thickness(h)
        D.set_u(u)
        D.assemble(_lib.ASM_ALL)
        assert _rel(D.residual(), O.residual()) < RTOL
        vals = O.assemble()
        for which in range(5):
            assert _rel(D.values(which), vals[which]) < RTOL, (case, which)
        D.close()


@pytest.mark.parametrize("p", [2, 3, 4])
def test_large_deformation_and_thickness_contrast(oracle_lib, p):
    """Parity away from the small-strain regime: displacements of 30x the thickness scale, thickness varying 4x."""
    from goldfish_amd import _lib
    from oracle.oracle_py import Oracle
    spec = G.synthetic_shell(3, 2, nel=4, p=p, jitter=1)
    rng = np.random.default_rng(7)
    th = [spec.h_th * rng.uniform(0.5, 2.0, q.ncp) for q in spec.patches]
    A = arrays_from_spec(spec, th)
    h, u = np.concatenate(th), 0.3 * rng.standard_normal(A.ndof)
    O, D = Oracle(A, thickness=h, u=u), _lib.DeviceModel(A)
    D.set_thickness(h)
    D.set_u(u)
    D.assemble(_lib.ASM_ALL)
    assert _rel(D.residual(), O.residual()) < RTOL
    vals = O.assemble()
    for which in range(5):
        assert _rel(D.values(which), vals[which]) < RTOL, which
    D.close()


def test_zero_state_and_reproducible(oracle_lib):
    """u = 0 (reference == deformed) and bitwise run-to-run reproducibility of the assembly
    (atomic-free owner gathers)."""
    from goldfish_amd import _lib
    from oracle.oracle_py import Oracle
    spec = G.scordelis_lo_9patch(4)
    A = arrays_from_spec(spec)
    h = np.full(A.total_cp, spec.h_th)
    D = _lib.DeviceModel(A)
    D.set_thickness(h)
    D.assemble()
    O = Oracle(A, thickness=h)
    assert _rel(D.residual(), O.residual()) < RTOL
    v0 = [D.values(w).copy() for w in range(5)]
    R0 = D.residual()
    D.assemble()
    assert np.array_equal(R0, D.residual())
    for w in range(5):
        assert np.array_equal(v0[w], D.values(w))
    D.close()


@pytest.mark.parametrize("case", ["tbeam2_p3", "C3_wing16_refdata", "shell2x2_p4", "tbeam2_p2_pressure_edge"])
def test_apply_linear(oracle_lib, case):
    """DispImOpeartion.apply_linear_fwd / rev (disp_imop.py:58-128): y += J x and z += J^T w for the five Jacobians against the ORACLE's matrices
    (not only the device's own CSR), in place; transposed products bitwise reproducible.  The pressure case has a non-symmetric K."""
    from goldfish_amd import _lib
    from oracle.oracle_py import Oracle
    spec = CASES[case]() if case in CASES else G.tbeam_2patch(6)
    A, h, u = _state(spec, seed=4)
    D = _lib.DeviceModel(A)
    D.set_thickness(h)
    D.set_u(u)
    D.assemble()
    O = Oracle(A, thickness=h, u=u)
    ovals = O.assemble()
    rng = np.random.default_rng(9)
    for which in range(5):
        Mx, Mo = D.csr(which), O.csr(which, ovals[which])
        x, y0 = rng.standard_normal(Mx.shape[1]), rng.standard_normal(Mx.shape[0])
        y = y0.copy()
        D.apply(which, x, y)
        assert _rel(y, y0 + Mx @ x) < 1e-12
        assert _rel(y - y0, Mo @ x) < RTOL, (case, which)
        xt, z0 = rng.standard_normal(Mx.shape[0]), rng.standard_normal(Mx.shape[1])
        z = z0.copy()
        D.apply(which, xt, z, transpose=True)
        assert _rel(z, z0 + Mx.T @ xt) < 1e-12
        assert _rel(z - z0, Mo.T @ xt) < RTOL, (case, which)
        z2 = z0.copy()
        D.apply(which, xt, z2, transpose=True)
        assert np.array_equal(z, z2)                        # fixed-order transposed products: bitwise reproducible
    with pytest.raises(ValueError):
        D.apply(0, np.zeros(3), np.zeros(A.ndof))
    D.close()


def test_error_behaviour():
    from goldfish_amd import _lib
    spec = G.tbeam_2patch(4)
    A, h, u = _state(spec)
    D = _lib.DeviceModel(A)
    with pytest.raises(ValueError):
        D.set_u(np.zeros(5))
    with pytest.raises(RuntimeError):
        D.values(0)          # not assembled yet
    D.close()


def test_chunked_scratch_gives_identical_results(oracle_lib, monkeypatch):
    """Element-block path (p = 4; GF_ASSEMBLY=block for p = 3): GF_SCRATCH_GB small enough to force one chunk per patch gives the same bits
    as the single-chunk run."""
    from goldfish_amd import _lib
    monkeypatch.setenv("GF_ASSEMBLY", "block")
    for p in (3, 4):
        spec = G.synthetic_shell(3, 2, nel=5, p=p, jitter=1)
        A, h, u = _state(spec, seed=5)
        out = []
        for gb in ("40", "0.0015" if p == 3 else "0.004"):
            monkeypatch.setenv("GF_SCRATCH_GB", gb)
            D = _lib.DeviceModel(A)
            assert D.assembly_path == 0
            D.set_thickness(h)
            D.set_u(u)
            D.assemble()
            out.append([D.residual()] + [D.values(w) for w in range(5)])
            D.close()
        for x, y in zip(*out):
            assert np.array_equal(x, y)


def test_p4_record_chunks_give_identical_results(monkeypatch):
    """p = 4 row-record path: GF_SCRATCH_GB small enough to force one chunk of work items per patch (the walks and the gather of a chunk run before the next
    chunk reuses the record buffer; the gather's record rows are relative to the chunk's first item) gives the same bits as the single-chunk run, for the full
    pass and the Newton pass."""
    from goldfish_amd import _lib
    spec = G.synthetic_shell(3, 2, nel=6, p=4, jitter=1)
    A, h, u = _state(spec, seed=5)
    out = []
    monkeypatch.setenv("GF_ASSEMBLY", "rec")
    for gb in ("16", "0.0012"):
        monkeypatch.setenv("GF_SCRATCH_GB", gb)
        D = _lib.DeviceModel(A)
        assert D.assembly_path == 5
        D.set_thickness(h)
        D.set_u(u)
        D.assemble()
        res = [D.residual()] + [D.values(w) for w in range(5)]
        D.assemble(_lib.ASM_R | _lib.ASM_K)
        res += [D.residual(), D.values(0)]
        out.append(res)
        D.close()
    for x, y in zip(*out):
        assert np.array_equal(x, y)
    assert np.array_equal(out[0][0], out[0][6]) and np.array_equal(out[0][1], out[0][7])       # Newton pass (9 values per pair) = full pass


def test_p4_hybrid_path_chunks_and_pass_kinds(oracle_lib, monkeypatch):
    """p = 4 default (hybrid): a Newton pass (R, K) runs through the row records, a pass with dR/dCP / dR/dh through element blocks -- each on its own chunks of
    patches.  Both against the oracle, K of the two pass kinds equal to round-off, and chunked scratch (one chunk per patch for either path) bitwise equal to one chunk."""
    from goldfish_amd import _lib
    from oracle.oracle_py import Oracle
    spec = G.synthetic_shell(3, 2, nel=6, p=4, jitter=1)
    A, h, u = _state(spec, seed=5)
    O = Oracle(A, thickness=h, u=u)
    vals, Ro = O.assemble(), O.residual()
    out = []
    for gb in (None, "0.004", "0.0012"):
        if gb is None: monkeypatch.delenv("GF_SCRATCH_GB", raising=False)
        else: monkeypatch.setenv("GF_SCRATCH_GB", gb)
        D = _lib.DeviceModel(A)
        assert D.assembly_path == 6
        D.set_thickness(h)
        D.set_u(u)
        D.assemble(_lib.ASM_R | _lib.ASM_K)                       # records
        res = [D.residual(), D.values(0)]
        D.assemble()                                              # element blocks
        res += [D.residual()] + [D.values(w) for w in range(5)]
        D.assemble(_lib.ASM_K)                                    # records again: the full pass left nothing behind that they depend on
        res += [D.values(0)]
        out.append(res)
        D.close()
    for other in out[1:]:
        for x, y in zip(out[0], other):
            assert np.array_equal(x, y)
    r = out[0]
    assert _rel(r[0], Ro) < RTOL and _rel(r[2], Ro) < RTOL and _rel(r[1], vals[0]) < RTOL
    for w in range(5):
        assert _rel(r[3 + w], vals[w]) < RTOL
    assert np.array_equal(r[1], r[8]) and _rel(r[1], r[3]) < 1e-13


def test_small_strain_residual_matches_the_displacement_based_oracle(oracle_lib):
    """Round 5: the kernels evaluate the strains and the penalty's rotation measures from the displacement derivatives (kl_point.hpp: kl_strains, pen_rot_measures).
    At displacement coefficients of 1e-11 and no external load the residual IS the internal force: the GPU agrees with the oracle's displacement-based mode
    to 1e-10, while the oracle's default mode (differences of metrics: the reference's arithmetic) is off by the cancellation it carries -- 1e-9 and more -- at the same
    state; at the strains of the other parity tests (1e-2) the two modes are indistinguishable (tests/test_strain_evaluation.py)."""
    import dataclasses
    from goldfish_amd import _lib
    from oracle import oracle_py
    from oracle.oracle_py import Oracle
    for spec in (G.scordelis_lo_9patch(3, nels=[2, 1, 2, 3, 2, 3, 2, 1, 2]), G.synthetic_shell(2, 2, nel=4, p=4, jitter=1)):
        spec = dataclasses.replace(spec, body_force=[[0.0, 0.0, 0.0]] * len(spec.patches), point_loads=[])
        A, h, _ = _state(spec)
        u = 1e-11 * np.random.default_rng(8).standard_normal(A.ndof)
        D = _lib.DeviceModel(A)
        D.set_thickness(h)
        D.set_u(u)
        D.assemble(_lib.ASM_R | _lib.ASM_K)
        Rg, Kg = D.residual(), D.values(0)
        D.close()
        with oracle_py.strain_mode(1):
            O = Oracle(A, thickness=h, u=u)
            R1, K1 = O.residual(), O.assemble(dRdCP=(), dRdh=False)[0]
        with oracle_py.strain_mode(0):
            R0 = Oracle(A, thickness=h, u=u).residual()
        assert _rel(Rg, R1) < 1e-10 and _rel(Kg, K1) < RTOL
        assert _rel(R0, R1) > max(1e-9, 10.0 * _rel(Rg, R1))      # what the difference form loses at these strains (the GPU path would fail the 1e-10 bar against it)


def test_single_patch_without_interfaces(oracle_lib):
    from goldfish_amd import _lib
    from oracle.oracle_py import Oracle
    spec = G.scordelis_lo_single(5)
    A, h, u = _state(spec, seed=6, uamp=5e-2)
    O, D = Oracle(A, thickness=h, u=u), _lib.DeviceModel(A)
    D.set_thickness(h)
    D.set_u(u)
    D.assemble()
    vals = O.assemble()
    assert _rel(D.residual(), O.residual()) < RTOL
    for w in range(5):
        assert _rel(D.values(w), vals[w]) < RTOL
    D.close()


def test_create_rejects_bad_models():
    from goldfish_amd import _lib
    from goldfish_amd.splines import NURBSPatch
    from goldfish_amd.geometry import ProblemSpec
    p1 = NURBSPatch.bilinear([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0]], 2, 2, 3)
    p2 = NURBSPatch.bilinear([[1, 0, 0], [2, 0, 0], [1, 1, 0], [2, 1, 0]], 2, 2, 2)
    spec = ProblemSpec([p1, p2], [], 1.0, 0.3, 0.1, [[0, 0, 0]] * 2)
    with pytest.raises(RuntimeError, match="degree"):
        _lib.DeviceModel(arrays_from_spec(spec))
