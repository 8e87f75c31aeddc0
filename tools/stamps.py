"""Diagnostic (not product): where do the element kernel's wave cycles go?  Needs the -DGF_STAMPS build."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["GF_LIB"] = os.environ.get("GF_STAMPS_LIB") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "goldfish_amd", "libgoldfish_hip_stamps.so")
from goldfish_amd import _lib, geometry as G
from goldfish_amd.model import arrays_from_spec
P = int(os.environ.get("GF_P", "3"))
spec = G.synthetic_shell(8, 8, nel=48, p=3, jitter=2) if P != 4 else G.synthetic_fuselage(8, 4, nel=53, p=4, jitter=2)
th = G.random_thickness(spec)
A = arrays_from_spec(spec, th)
D = _lib.DeviceModel(A)
D.set_thickness(np.concatenate(th)); D.set_u(G.smooth_displacement(spec, 0.5 * spec.h_th))
L = _lib.lib(); out = (C.c_ulonglong * 8)()
FLAGS = int(os.environ.get("GF_FLAGS", "15"))      # 15: full pass; 3: R + K (p = 4 records: PASS 0 only); 12: dR/dCP + dR/dh (PASS 1 + 2)
D.assemble(FLAGS); L.gf_debug_stamps(out)
D.assemble(FLAGS); L.gf_debug_stamps(out)
mfma = os.environ.get("GF_ELEMENT", "mfma") != "valu"
rec = D.assembly_path == 4
nw = (D.n_elements / 8.0) if rec else (1 if mfma else 2) * ((D.n_elements + 31) // 32)      # 1 in 32 elements sampled (row-record kernel: 1 in 8 items; per element)
if D.assembly_path == 5:        # p = 4 row records: per element, summed over the walks the flags launch (1 in 8 items sampled)
    nw = D.n_elements / 8.0
    names = ["phase1 pointwise (one lane per Gauss point)", "basis of both tiles at the Gauss point", "row expansion", "rz / rh prefactors + dR/dh MFMAs", "T formation + MFMAs",
             "residual reduction, fetch issue", "flush (record stores)", "park next inputs"]
elif P == 4 and mfma:
    names = ["phase0 load", "phase1 pointwise (one lane per Gauss point)", "basis of both tiles at the Gauss point", "row expansion (per pass)", "rz / rh + dR/dh MFMAs", "K: T formation + MFMAs", "dR/dCP: T formation + MFMAs", "element-block stores"]
elif rec and os.environ.get("GF_STAMPS_FINE"):       # library built with -DGF_STAMPS -DGF_STAMPS_FINE: the sections of the shared group step
    names = ["ring -> staging", "phase1 pointwise", "group: loads + basis function", "group: row expansion", "group: residual / dR/dh prefactors", "group: K (T formation + MFMAs)", "group: dR/dCP (T formation + MFMAs)", "fetch, park, residual, record stores"]
elif rec:
    names = ["ring -> staging", "phase1 pointwise", "-", "fetch issue (next element)", "group loop (expansion, T, MFMA)", "park + residual", "record stores", "-"]
elif mfma:
    names = ["phase0 load", "phase1 pointwise", "-", "-", "group loop (expansion, T, MFMA)", "-", "-", "-"]
else:
    names = ["phase0 load", "phase1 pointwise(+barrier wait)", "descriptors", "S1 expansion", "barrier1", "S2 T-formation", "barrier2", "S3 contraction"]
tot = sum(out)
for n, v in zip(names, out):
    print("%-34s %10.0f cycles/wave  %5.1f%%" % (n, v / nw, 100.0 * v / tot))
print("total %.0f cycles/wave" % (tot / nw))
