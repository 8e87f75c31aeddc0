"""Measurement for BASELINE.json configs[4] ("mixed FP64/FP32 basis eval", SURVEY.md 8(d): "error vs FP64 reported"): one GPU's share
of C5 (128 patches of the 32 x 32-patch p = 4 fuselage, 53 spans a side) assembled with FP64 1-D basis tables and with the
tables rounded to FP32 (GF_BASIS_FP32=1: what an FP32 basis evaluation delivers; all accumulation in FP64 in both runs):
relative error of R, K, dR/dCP, dR/dh and the time per pass of both.  usage: mixed_precision.py [patches_x patches_y nel p]"""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from goldfish_amd import _lib, geometry as G
from goldfish_amd.model import arrays_from_spec

px, py, nel, p = (int(a) for a in sys.argv[1:5]) if len(sys.argv) >= 5 else (16, 8, 53, 4)
spec = G.synthetic_fuselage(px, py, nel=nel, p=p, jitter=2)
th = G.random_thickness(spec)
A = arrays_from_spec(spec, th)
u = G.smooth_displacement(spec, 0.5 * spec.h_th)
out = {}
for mode in ("fp64", "fp32_basis"):
    if mode == "fp32_basis":
        os.environ["GF_BASIS_FP32"] = "1"
    D = _lib.DeviceModel(A)
    D.set_thickness(np.concatenate(th)); D.set_u(u)
    D.assemble(); D.sync()
    t0 = time.perf_counter()
    for _ in range(3): D.assemble(sync=False)
    D.sync()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    out[mode] = dict(ms=ms, R=D.residual().copy(), vals=[D.values(w).copy() for w in range(5)])
    D.close()
    os.environ.pop("GF_BASIS_FP32", None)
names = ["K", "dR/dCP_0", "dR/dCP_1", "dR/dCP_2", "dR/dh"]
rel = lambda a, b: float(np.abs(a - b).max() / np.abs(b).max())
res = {"workload": "%s: %d patches, p = %d, %d spans a side, %d dofs, %d Gauss points" % (spec.name, len(spec.patches), p, nel, A.ndof, A.n_gauss_points),
       "ms_per_pass": {m: out[m]["ms"] for m in out},
       "max_abs_error_over_max_abs_value": dict([("R", rel(out["fp32_basis"]["R"], out["fp64"]["R"]))] + [(n, rel(out["fp32_basis"]["vals"][w], out["fp64"]["vals"][w])) for w, n in enumerate(names)]),
       "note": "GF_BASIS_FP32=1 rounds the 1-D basis values / derivatives to FP32 at setup (the tables are precomputed on the host: their evaluation is "
               "not on the device's critical path, so no time is saved); the parity bar of north_star is 1e-10"}
print(json.dumps(res, indent=1))
