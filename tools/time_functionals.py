import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from goldfish_amd import _lib, geometry as G
from goldfish_amd.model import arrays_from_spec
spec = G.synthetic_shell(16, 16, nel=48, p=3, jitter=2)
th = G.random_thickness(spec)
D = _lib.DeviceModel(arrays_from_spec(spec, th))
A_ = arrays_from_spec(spec, th)
cp2_0 = np.asarray(A_.cp_hom[2]) * 1.001
D.set_thickness(np.concatenate(th)); D.set_u(G.smooth_displacement(spec, 0.5 * spec.h_th))
for name, fn in (("functionals", lambda: D.functionals()), ("compliance", lambda: D.compliance(np.ones((256, 3)))),
                 ("stress forms (KS, top, Cauchy)", lambda: D.stress_forms(0, 1e-8, np.full(256, 1e8))),
                 ("stress forms, values only", lambda: D.stress_forms(1, 8.0, np.full(256, 1e8), gradients=False)),
                 ("shape regularisation", lambda: D.shape_regu(2, cp2_0, np.ones(256)))):
    fn(); D.sync()
    t0 = time.perf_counter()
    for _ in range(3): fn()
    D.sync()
    print(name, "%.2f ms per call" % ((time.perf_counter() - t0) / 3 * 1e3), flush=True)
t0 = time.perf_counter()
for _ in range(3): F = None; F = D.functionals(); F["dWdu"]
print("functionals + dW/du fetched: %.2f ms per call" % ((time.perf_counter() - t0) / 3 * 1e3))
t0 = time.perf_counter()
for _ in range(3): F = None; F = D.functionals(); F.materialize()
print("functionals + all five gradient fields fetched: %.2f ms per call" % ((time.perf_counter() - t0) / 3 * 1e3))
