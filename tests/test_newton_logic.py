"""Host logic of NonMatchingOpt.solve_nonlinear_nonmatching_problem and solve_K on stand-in residuals (no GPU): plain Newton through a
non-monotone start (1, 50, 5, 0.6 must not be mistaken for a floor), backtracking, the evaluation floor of the residual,
honest warnings, and the fallback of a rejected device solve.  Reference behaviour: GOLDFISH/operations/disp_imop.py:38-44
(max_it 30, rtol 1e-3 relative to the first residual)."""
import warnings

import numpy as np
import pytest

from goldfish_amd import _lib
from goldfish_amd.nonmatching_opt import NonMatchingOpt


class _FakeDev:
    def __init__(self, nm):
        self.nm = nm

    def residual(self):
        return self.nm.fun(self.nm.u_iga)


class FakeNM(NonMatchingOpt):
    """R(u) and its Jacobian given as Python callables; everything device-side replaced."""

    def __init__(self, fun, jac, n, u0=None):
        self.fun, self.jac, self.vec_iga_dof = fun, jac, n
        self.u_iga = np.zeros(n) if u0 is None else np.asarray(u0, float)
        self._dev = _FakeDev(self)
        self.n_assemblies = 0

    dev = property(lambda self: self._dev)

    def update_uIGA(self, u):
        self.u_iga = np.asarray(u, float).copy()

    def _assemble(self, flags):
        self.n_assemblies += 1

    def solve_K(self, rhs, transpose=False, refine=None):
        return np.linalg.solve(self.jac(self.u_iga), rhs)


def test_non_monotone_start_is_not_a_floor():
    """A stiffening bar: the linear first step overshoots (residual rises well above |R_0|), plain Newton then converges
    quadratically -- the round-2 exit (no halving over three residuals that include |R_0|) stopped this at iteration 4."""
    k, f = 1.0e4, 1.0
    nm = FakeNM(lambda u: u + k * u ** 3 - f, lambda u: np.diag(1.0 + 3.0 * k * u ** 2), 1)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        _, u = nm.solve_nonlinear_nonmatching_problem(rtol=1e-12, max_it=30)
    hist = [h[0] for h in nm.newton_history]
    assert hist[0] > 100.0 and nm.newton_converged and not nm.newton_stagnated
    assert abs(u[0] + k * u[0] ** 3 - f) < 1e-11
    assert all(h[2] == 1.0 for h in nm.newton_history)           # monotone after the first step: never backtracked
    assert u is not nm.u_iga                                     # a copy: editing it cannot alias the cached state


def test_backtracking_rescues_a_diverging_newton():
    nm = FakeNM(lambda u: np.arctan(u), lambda u: np.diag(1.0 / (1.0 + u ** 2)), 1, u0=[3.0])
    _, u = nm.solve_nonlinear_nonmatching_problem(rtol=1e-12, max_it=30, zero_mortar_funcs=False)
    assert nm.newton_converged and abs(u[0]) < 1e-10
    assert min(h[2] for h in nm.newton_history) < 1.0              # some step was shortened (plain Newton diverges from 3)


def test_evaluation_floor_warns_every_time_and_can_raise():
    rng = np.random.default_rng(0)
    A = np.diag([1.0, 1.0e3, 1.0e6])

    def fun(u):                                                   # linear residual + evaluation noise of relative size 1e-7
        return A @ u - np.ones(3) + 1e-7 * rng.standard_normal(3)
    nm = FakeNM(fun, lambda u: A, 3)
    for _ in range(2):                                            # every unconverged solve warns, not only the first
        with pytest.warns(RuntimeWarning, match="stagnates"):
            nm.solve_nonlinear_nonmatching_problem(rtol=1e-14, max_it=30)
        assert not nm.newton_converged and nm.newton_stagnated and nm.newton_iterations < 10
        assert nm.newton_relative_residual < 1e-5
    nm.newton_raise_unconverged = True
    with pytest.raises(RuntimeError, match="not converged"):
        nm.solve_nonlinear_nonmatching_problem(rtol=1e-14, max_it=30)
    # the same floor with an achievable tolerance: converged, silent
    nm.newton_raise_unconverged = False
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        nm.solve_nonlinear_nonmatching_problem(rtol=1e-5, max_it=30)
    assert nm.newton_converged


def test_negligible_correction_ends_the_iteration_but_is_not_convergence():
    """Residual noise that a tight rtol cannot pass and a Newton correction far below newton_step_rtol |u|: the iteration ends ("by step"), but
    ``newton_converged`` follows the reference's criterion alone -- |R| / ref < rtol (disp_imop.py:38-44) -- and the solve warns."""
    rng = np.random.default_rng(1)
    nm = FakeNM(lambda u: u - 1.0 + 1e-12 * rng.standard_normal(2), lambda u: np.eye(2), 2)
    with pytest.warns(RuntimeWarning, match="Newton correction is negligible"):
        nm.solve_nonlinear_nonmatching_problem(rtol=1e-15, max_it=30)
    assert not nm.newton_converged and nm.newton_converged_by_step and nm.newton_iterations < 5
    # the same noise with an achievable tolerance converges silently
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        nm.solve_nonlinear_nonmatching_problem(rtol=1e-9, max_it=30)
    assert nm.newton_converged and not nm.newton_converged_by_step


def test_the_c4_history_of_round_4_is_reported_as_unconverged():
    """gpurun_out/newton_reuse_c4.txt (round 4, C4, load 2e-2): |R| / |R_0| = 1.3e5, 0.93, 0.95, 0.93, 0.89, 0.87 with corrections that fall below 1e-9 |u| --
    the evaluation floor of the residual sits at 0.87 |R_0|.  Round 4 returned ``newton_converged = True`` for it, silently."""
    seq = iter([1.0, 1.3e5, 0.93, 0.95, 0.93, 0.89, 0.87, 0.87, 0.87, 0.87])
    steps = iter([1.0, 1e-3, 1e-6, 1e-8, 1e-10, 1e-11, 1e-11, 1e-11, 1e-11])

    class Scripted(FakeNM):
        def __init__(self):
            super().__init__(None, None, 1)
            self.cur = next(seq)
            self.u_iga = np.zeros(1)

        def _assemble(self, flags):
            pass

        def update_uIGA(self, u):
            self.u_iga = np.asarray(u, float).copy()
            self.cur = next(seq)

        def solve_K(self, rhs, transpose=False, refine=None):
            return np.array([next(steps)])
    nm = Scripted()
    nm._dev = type("D", (), {"residual": lambda self_: np.array([nm.cur])})()
    with pytest.warns(RuntimeWarning, match="not converged"):
        nm.solve_nonlinear_nonmatching_problem(rtol=1e-3, max_it=30, zero_mortar_funcs=False)
    assert not nm.newton_converged and (nm.newton_converged_by_step or nm.newton_stagnated)
    assert abs(nm.newton_relative_residual - 0.87) < 0.1


def test_a_diverging_step_after_an_overshooting_first_step_is_shortened():
    """the non-monotone window held the unconditionally accepted first step (the T-beam's jumps to 208 |R_0|), so the third step could rise to that
    overshoot at full length.  Scripted residuals: 1, 200 (first step), 0.5 (second), then a full third step to 150 -- below the overshoot, far above everything
    else in the window: it must be cut."""
    calls = []

    class Scripted(FakeNM):
        def __init__(self):
            super().__init__(None, None, 1)
            self.r = 1.0

        def update_uIGA(self, u):
            self.u_iga = np.asarray(u, float).copy()
            u = float(self.u_iga[0])
            calls.append(u)
            # |R|: 1 at u = 0; 200 at the first full step (u = 1); 0.5 at the second (u = 2); a full third step (u = 3) gives 150, half of it (u = 2.5) gives 1e-5
            self.r = {0.0: 1.0, 1.0: 200.0, 2.0: 0.5, 3.0: 150.0}.get(u, 1e-5)

        def solve_K(self, rhs, transpose=False, refine=None):
            return np.array([1.0])
    nm = Scripted()
    nm._dev = type("D", (), {"residual": lambda self_: np.array([nm.r])})()
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        nm.solve_nonlinear_nonmatching_problem(rtol=1e-3, max_it=10, zero_mortar_funcs=False)
    assert nm.newton_converged and [h[2] for h in nm.newton_history] == [1.0, 1.0, 0.5]      # first step full (overshoot allowed), second full, third halved
    assert 3.0 in calls and 2.5 in calls


def test_load_steps_follow_the_path_and_measure_against_the_full_load():
    """``load_steps = n``: R_s(u) = R(u) - (1 - s) R(0), one Newton solve per increment from the previous state, tolerance against the full load's |R(0)|.  A softening
    spring whose full load plain Newton (first step = the linear solution, far past the limit of the stiffening branch) reaches only with cut steps is walked in
    increments; the end state is the full-load root either way."""
    f = 2.0
    fun, jac = (lambda u: np.tanh(u) + 0.05 * u - f * np.ones(1)), (lambda u: np.diag(1.0 / np.cosh(u) ** 2 + 0.05))
    one = FakeNM(fun, jac, 1)
    _, u1 = one.solve_nonlinear_nonmatching_problem(rtol=1e-10, max_it=60)
    nm = FakeNM(fun, jac, 1)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        _, u = nm.solve_nonlinear_nonmatching_problem(rtol=1e-10, max_it=60, load_steps=8)
    assert nm.newton_converged and nm.newton_load_steps_done == 8
    assert abs(fun(u)[0]) < 1e-9 * f and abs(u[0] - u1[0]) < 1e-7 * abs(u1[0])
    assert all(h[2] == 1.0 for h in nm.newton_history)                                                   # the increments never needed a cut step
    assert max(h[0] for h in nm.newton_history) < max(h[0] for h in one.newton_history)                   # and never left the path as far as the single solve's first step did
    assert nm.newton_relative_residual < 1e-10                                                            # relative to the FULL load
    assert nm._newton_load_offset is None                                                                  # nothing of the increment scheme is left behind
    # the class-level switch does the same
    nm2 = FakeNM(fun, jac, 1); nm2.newton_load_steps = 4
    _, u2 = nm2.solve_nonlinear_nonmatching_problem(rtol=1e-10, max_it=60)
    assert nm2.newton_load_steps_done == 4 and abs(u2[0] - u1[0]) < 1e-7 * abs(u1[0])
    # a follower pressure depends on u: R(0) is not its load vector
    from goldfish_amd.nonmatching_opt import SVKResidual
    nm3 = FakeNM(fun, jac, 1); nm3.residuals = [SVKResidual(pressure=1.0)]
    with pytest.raises(NotImplementedError, match="follower pressure"):
        nm3.solve_nonlinear_nonmatching_problem(load_steps=2)


def test_max_it_without_convergence_warns():
    nm = FakeNM(lambda u: u ** 2 + 1.0, lambda u: np.diag(2.0 * u + 1e-3), 1, u0=[1.0])       # no root
    with pytest.warns(RuntimeWarning, match="not converged after 5 iterations"):
        nm.solve_nonlinear_nonmatching_problem(rtol=1e-8, max_it=5, zero_mortar_funcs=False)
    assert not nm.newton_converged


class _FakeSolver:
    def __init__(self, D, x, rr):
        self.D, self.x, self.rel_residual, self.backward_error, self.closed = D, x, rr, rr, False

    def solve(self, b, transpose=False, max_refine=None):
        return self.x(b)

    def refactor(self):
        pass

    def close(self):
        self.closed = True


def test_rejected_device_solve_falls_back_to_the_host(monkeypatch):
    """solve_K returned the device solution unchecked.  A solve whose refined residual stays above
    linear_solve_rtol (indefinite tangent under unpivoted L D L^T) or that is not finite goes to the host path, once per K."""
    import scipy.sparse as sp
    K = sp.csr_matrix(np.array([[2.0, 1.0], [1.0, -3.0]]))

    class Dev:
        def csr(self, which):
            assert which == _lib.MAT_K
            return K
    nm = NonMatchingOpt.__new__(NonMatchingOpt)
    nm._dev, nm._k_version = Dev(), 7
    monkeypatch.setattr(NonMatchingOpt, "dev", property(lambda self: self._dev))
    monkeypatch.setattr(NonMatchingOpt, "linear_solver", "device")
    b = np.array([1.0, 2.0])
    nm._dsolver, nm._dsolver_version = _FakeSolver(nm._dev, lambda r: np.array([np.nan, 0.0]), 0.0), 7
    with pytest.warns(RuntimeWarning, match="falling back"):
        x = nm.solve_K(b)
    assert np.allclose(K @ x, b)
    with warnings.catch_warnings():
        warnings.simplefilter("error")                            # same K: straight to the host factors, no second warning
        assert np.allclose(K @ nm.solve_K(2 * b), 2 * b)
    nm._k_version = 8                                             # new tangent: the device gets its chance again
    nm._dsolver = _FakeSolver(nm._dev, lambda r: np.linalg.solve(K.toarray(), r), 1e-13)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        assert np.allclose(K @ nm.solve_K(b), b)
    nm._dsolver = _FakeSolver(nm._dev, lambda r: np.zeros(2), 0.5)
    nm._k_version = 9
    with pytest.warns(RuntimeWarning, match="backward error"):
        assert np.allclose(K @ nm.solve_K(b), b)


def test_dropping_the_device_model_drops_its_factorisations():
    """mortar_meshes_setup / set_residuals / set_point_sources left a DeviceSolver bound to the old model."""
    nm = NonMatchingOpt.__new__(NonMatchingOpt)
    closed = []

    class Dev:
        def close(self):
            closed.append("dev")
    ds = _FakeSolver(None, None, 0.0)
    nm._dev, nm._dsolver, nm._hlu = Dev(), ds, object()
    nm.num_splines = 1
    nm.set_residuals([object()])
    assert nm._dev is None and nm._dsolver is None and nm._hlu is None and ds.closed and closed == ["dev"]


def test_lazy_fields_is_a_complete_mapping():
    """dict(g), iteration and len dropped the gradient fields not yet fetched; a failed fetch lost its key."""
    fetched = []

    class L:
        @staticmethod
        def gf_get_functional_gradient(h, field, ptr, n):
            fetched.append(field)
            return 0 if field != 9 else 1

        @staticmethod
        def gf_last_error():
            return b"boom"

    class Dev:
        h = None
    g = _lib._LazyFields(Dev(), {"a": (0, (2,)), "b": (1, (3,)), "bad": (9, (1,))}, dict(W=1.0))
    old = _lib.lib
    _lib.lib = lambda: L
    try:
        assert len(g) == 4 and set(g) == {"W", "a", "b", "bad"} and "a" in g and g.get("zz") is None
        assert g["a"].shape == (2,) and fetched == [0]
        with pytest.raises(RuntimeError):
            g["bad"]
        assert "bad" in g                                         # still pending after the failed fetch
        del g._pending["bad"]
        d = dict(g)
        assert set(d) == {"W", "a", "b"} and d["b"].shape == (3,) and fetched == [0, 9, 1]
        assert set(g.copy()) == {"W", "a", "b"}
    finally:
        _lib.lib = old


def test_overshooting_first_step_is_not_a_contraction():
    """hist = 1, 50, 0.9, 0.8, 0.7, ... -- the overshoot of the linear first step made min(hist) < 0.1 max(hist) true although nothing had
    contracted below the first residual; the loop then declared a floor at 0.7.  A slowly but steadily descending iteration runs on."""
    seq = [1.0, 50.0, 0.9, 0.8, 0.7, 0.6, 0.5, 0.4, 0.3, 0.2, 0.05, 1e-4]
    calls = {"n": 0}

    class NM(FakeNM):
        def update_uIGA(self, u):
            super().update_uIGA(u)
            calls["n"] = int(round(self.u_iga[0]))

    nm = NM(lambda u: np.array([seq[min(int(round(u[0])), len(seq) - 1)]]), lambda u: np.array([[-seq[min(int(round(u[0])), len(seq) - 1)]]]), 1)
    # K = -R  =>  du = solve(K, -R) = +1: every Newton step moves to the next entry of the sequence
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        nm.solve_nonlinear_nonmatching_problem(rtol=1e-3, max_it=30)
    assert nm.newton_converged and not nm.newton_stagnated and nm.newton_iterations == len(seq) - 1


def test_backtracking_trials_assemble_the_residual_only():
    """every rejected trial paid a tangent assembly.  The full step asks for R + K, shortened trials for R, the accepted state for K."""
    asked = []

    class NM(FakeNM):
        def _assemble(self, flags):
            asked.append(flags)
    nm = NM(lambda u: np.arctan(u), lambda u: np.diag(1.0 / (1.0 + u ** 2)), 1, u0=[3.0])
    nm.solve_nonlinear_nonmatching_problem(rtol=1e-12, max_it=30, zero_mortar_funcs=False)
    assert nm.newton_converged and _lib.ASM_R in asked and _lib.ASM_K in asked
    n_trials = sum(1 for f in asked if f == _lib.ASM_R)
    assert n_trials >= 1 and asked.count(_lib.ASM_R | _lib.ASM_K) == nm.newton_iterations + 1


def test_stalled_general_mode_solve_and_small_pivot_fall_back(monkeypatch):
    """with linear_solve_rtol = 1e-10 on the Frobenius-norm backward error a general-mode refinement that stalled (follower pressure: the
    stationary iteration on the symmetric part's factors) could pass.  The bar is 1e-12 now, 1e-14 when the factorisation reported a small pivot; a
    permanent failure (the factors do not fit) is latched instead of rebuilding the solver for every tangent."""
    import scipy.sparse as sp
    from goldfish_amd import _solver
    K = sp.csr_matrix(np.array([[2.0, 1.0], [0.5, -3.0]]))

    class Dev:
        def csr(self, which):
            return K
    nm = NonMatchingOpt.__new__(NonMatchingOpt)
    nm._dev, nm._k_version = Dev(), 1
    monkeypatch.setattr(NonMatchingOpt, "dev", property(lambda self: self._dev))
    monkeypatch.setattr(NonMatchingOpt, "linear_solver", "device")
    monkeypatch.setattr(NonMatchingOpt, "symmetric_K", property(lambda self: False))
    b = np.array([1.0, 2.0])
    exact = lambda r: np.linalg.solve(K.toarray(), r)
    nm._dsolver, nm._dsolver_version = _FakeSolver(nm._dev, lambda r: 0.9 * exact(r), 3e-11), 1       # stalled at 3e-11: passed in round 3
    with pytest.warns(RuntimeWarning, match="backward error"):
        assert np.allclose(K @ nm.solve_K(b), b)
    nm._k_version = 2
    ok = _FakeSolver(nm._dev, exact, 1e-13)
    ok.small_pivot = True                                                                              # 1e-13 passes 1e-12 but not the small-pivot bar
    nm._dsolver, nm._dsolver_version = ok, 2
    with pytest.warns(RuntimeWarning, match="small pivot"):
        assert np.allclose(K.T @ nm.solve_K(b, transpose=True), b)
    # permanent failure: latched, one warning, no second construction
    built = []

    class Boom:
        def __init__(self, *a, **k):
            built.append(1)
            raise RuntimeError("gfs_create_nd: the fronts need 400 GB, more than the free device memory")
    monkeypatch.setattr(_solver, "DeviceSolver", Boom)
    nm.splines, nm.cp_iga = [type("S", (), {"cp_hom_flat": staticmethod(lambda: np.ones((2, 4)))})()], [np.zeros(2)] * 3
    nm._k_version, nm._dsolver = 3, None
    with pytest.warns(RuntimeWarning, match="cannot be used for this model"):
        assert np.allclose(K @ nm.solve_K(b), b)
    nm._k_version = 4
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        assert np.allclose(K @ nm.solve_K(b), b)
    assert built == [1]
    # a model too large for a host LU (C5: 10 M dofs) gets an error that names the path that works, not a silent SuperLU of 10 M dofs (VERDICT r04 missing 1)
    nm._dsolver_permanent_failure, nm._k_version, nm.vec_iga_dof = None, 5, 10 ** 7
    with pytest.raises(RuntimeError, match="distributed factorisation"):
        nm.solve_K(b)


def test_a_device_solve_that_fails_on_one_rank_sends_every_rank_the_same_way(monkeypatch):
    """under a sharded problem a failure only ONE rank sees (its GPU ran out of memory) must not send that rank alone into the host solve -- the host
    solve gathers K with a collective the others never join.  solve_K agrees on the outcome (one all-reduce of two flags) before anybody leaves: the rank whose
    solve succeeded drops its solver too and both fall back; a permanent failure is latched on both."""
    import scipy.sparse as sp
    K = sp.csr_matrix(np.array([[2.0, 1.0], [1.0, 3.0]]))
    b = np.array([1.0, 2.0])

    def make(rank, fails):
        class Dev:
            def csr(self, which):
                return K

            def _allreduce(self, arr):           # stand-in for the collective: rank 1's flags are (failed, permanent), rank 0's are zeros
                return np.asarray(arr, float) + (np.array([1.0, 1.0]) if not fails else np.zeros(2))
        nm = NonMatchingOpt.__new__(NonMatchingOpt)
        nm._dev, nm._k_version, nm._dist, nm.vec_iga_dof = Dev(), 1, object(), 2
        closed = []
        if fails:
            class Bad(_FakeSolver):
                def solve(self, b, transpose=False, max_refine=None):
                    raise RuntimeError("gfs_refactor: hipMalloc failed: out of memory")
            nm._dsolver = Bad(nm._dev, None, 0.0)
        else:
            nm._dsolver = _FakeSolver(nm._dev, lambda r: np.linalg.solve(K.toarray(), r), 1e-16)
        nm._dsolver.close = lambda: closed.append(1)
        nm._dsolver_version = 1
        return nm, closed
    monkeypatch.setattr(NonMatchingOpt, "dev", property(lambda self: self._dev))
    monkeypatch.setattr(NonMatchingOpt, "linear_solver", "device")
    monkeypatch.setattr(NonMatchingOpt, "symmetric_K", property(lambda self: True))
    for fails in (True, False):                 # the failing rank and the rank whose own solve was fine
        nm, closed = make(0, fails)
        with pytest.warns(RuntimeWarning, match="cannot be used for this model"):
            x = nm.solve_K(b)
        assert np.allclose(K @ x, b) and closed == [1] and nm._dsolver is None and nm._dsolver_permanent_failure is not None


def test_non_monotone_shell_history_is_not_cut_short():
    """The sliding-web T-beam's residual history under plain Newton (the reference's iteration): 1, 208, 0.056, 0.5, 1e-4, 1e-9.  A monotone backtracking
    rule shortens the step to 0.5 sixteen-fold and creeps at 0.05 (round 4, found by the shape_opt_mint group); the non-monotone rule lets it through."""
    seq = [1.0, 208.0, 0.056, 0.5, 1e-4, 1e-9, 1e-14]
    nm = FakeNM(lambda u: np.array([seq[min(int(round(u[0])), len(seq) - 1)]]), lambda u: np.array([[-seq[min(int(round(u[0])), len(seq) - 1)]]]), 1)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        nm.solve_nonlinear_nonmatching_problem(rtol=1e-8, max_it=30)
    assert nm.newton_converged and nm.newton_iterations == 5 and all(h[2] == 1.0 for h in nm.newton_history)


class ChordNM(FakeNM):
    """FakeNM whose solve_K keeps 'factors': the Jacobian of the state it last factored (what the device solver does when solve_K is asked to reuse them)."""
    newton_reuse_factors = True

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.J, self.n_factor, self.n_stale = None, 0, 0

    def solve_K(self, rhs, transpose=False, refine=None, stale_factors=False):
        if stale_factors and self.J is not None:
            self.n_stale += 1
        else:
            self.J = self.jac(self.u_iga)
            self.n_factor += 1
        return np.linalg.solve(self.J, rhs)


def test_chord_steps_reuse_the_factors_while_they_contract():
    """Newton loop on large models: after a step that contracted the residual five-fold the next correction reuses the factorisation (chord step); the converged state is
    the full Newton iteration's, with fewer factorisations; a chord step that does not halve the residual is thrown away and repeated with fresh factors."""
    k, f = 1.0e4, 1.0
    fun, jac = (lambda u: u + k * u ** 3 - f), (lambda u: np.diag(1.0 + 3.0 * k * u ** 2))
    full = FakeNM(fun, jac, 1)
    _, u_full = full.solve_nonlinear_nonmatching_problem(rtol=1e-12, max_it=40)
    nm = ChordNM(fun, jac, 1)
    _, u = nm.solve_nonlinear_nonmatching_problem(rtol=1e-12, max_it=40)
    assert (nm.newton_converged or nm.newton_converged_by_step) and abs(fun(u)[0]) < 1e-9 and abs(u[0] - u_full[0]) < 1e-9 * abs(u_full[0])      # both end at the same root (by residual or by a negligible correction)
    assert nm.newton_chord_steps >= 1 and nm.n_stale >= nm.newton_chord_steps
    assert nm.n_factor < full.newton_iterations                      # fewer factorisations than plain Newton needs
    # a tangent that changes fast: chord steps get rejected (they do not halve the residual) and are redone as Newton steps -- same solution, never more than one
    # wasted substitution per Newton step
    fun2, jac2 = (lambda u: np.exp(3.0 * u) - 5.0), (lambda u: np.diag(3.0 * np.exp(3.0 * u)))
    full2 = FakeNM(fun2, jac2, 1, u0=[2.0]); full2.solve_nonlinear_nonmatching_problem(rtol=1e-13, max_it=60, zero_mortar_funcs=False)
    nm2 = ChordNM(fun2, jac2, 1, u0=[2.0])
    _, u2 = nm2.solve_nonlinear_nonmatching_problem(rtol=1e-13, max_it=60, zero_mortar_funcs=False)
    assert nm2.newton_converged and abs(u2[0] - np.log(5.0) / 3.0) < 1e-11
    assert nm2.n_stale - nm2.newton_chord_steps <= nm2.n_factor       # rejected chord steps are bounded by the Newton steps
    # the switch: off = the reference's iteration, "auto" = only large models on the device solver
    off = ChordNM(fun, jac, 1); off.newton_reuse_factors = False
    off.solve_nonlinear_nonmatching_problem(rtol=1e-12, max_it=40)
    assert off.n_stale == 0 and off.newton_chord_steps == 0 and off.newton_iterations == full.newton_iterations
    auto = ChordNM(fun, jac, 1); auto.newton_reuse_factors = "auto"; auto.linear_solver = "device"
    assert not auto._newton_reuses_factors()
    auto.vec_iga_dof = 10 ** 6
    assert auto._newton_reuses_factors()
