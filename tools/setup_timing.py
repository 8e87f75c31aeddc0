"""Set-up cost of a model by phase: Python geometry, flattening into gf_model_desc, gf_create (its own phases
with GF_SETUP_TIMING=1: printed to stderr).  Usage: python tools/setup_timing.py [bench.py's workload arguments]"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["GF_SETUP_TIMING"] = "1"
import bench                                           # noqa: E402  (make_spec)
from goldfish_amd import _lib, geometry as G, sharding  # noqa: E402
ap = argparse.ArgumentParser()
ap.add_argument("--geometry", choices=["shell", "fuselage"], default="shell")
ap.add_argument("--patches", type=int, nargs=2, default=[16, 16])
ap.add_argument("--nel", type=int, default=48)
ap.add_argument("--degree", type=int, default=3)
args = ap.parse_args()
t = [time.perf_counter()]
spec = bench.make_spec(args, *args.patches); t.append(time.perf_counter())
th = G.random_thickness(spec); u = G.smooth_displacement(spec, 0.5 * spec.h_th); t.append(time.perf_counter())
part = sharding.partition_patches(spec, 1); shard = sharding.shard_spec(spec, 0, 1, part); t.append(time.perf_counter())
A = sharding.shard_arrays(shard, th); t.append(time.perf_counter())
D = _lib.DeviceModel(A); t.append(time.perf_counter())
D.set_thickness(np.concatenate(th)); D.set_u(u); D.sync(); t.append(time.perf_counter())
D.assemble(_lib.ASM_ALL); D.sync(); t.append(time.perf_counter())
names = ["Python geometry (patches, interfaces)", "thickness + displacement fields", "partition + shard", "flatten into gf_model_desc (penalty parameters)",
         "gf_create", "state upload", "first assembly (incl. code-object load)"]
for n, a, b in zip(names, t, t[1:]):
    print("%8.2f s  %s" % (b - a, n))
print("%8.2f s  total; %d dofs, %d Gauss points, device memory %.1f GB" % (t[-1] - t[0], A.ndof, A.n_gauss_points, D.device_bytes / 1e9))
