"""Diagnostic (not product): run-to-run reproducibility of the assembly paths on the parity cases -- models of both paths (row records, element blocks) are
created, assembled and closed in turn (as tests/test_gpu_parity.py does) and every output is compared bitwise with the first run of
its path."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from goldfish_amd import _lib
import test_gpu_parity as T
bad = 0
for case in ("tbeam2_p2", "shell3x2_p3", "C3_wing16_refdata", "slr9_nurbs_p3_projected_load"):
    A, h, u = T._state(T.CASES[case](), seed=11)
    ref = {}
    for rep in range(int(os.environ.get("REPS", "15"))):
        for walk in os.environ.get("PATHS", "rec,block").split(","):
            os.environ["GF_ASSEMBLY"] = walk
            os.environ["GF_REC_SEG"] = ("4", "7", "1000")[rep % 3]
            D = _lib.DeviceModel(A)
            D.set_thickness(h); D.set_u(u)
            for it in range(2):
                D.assemble(_lib.ASM_ALL)
                cur = [D.residual().copy()] + [D.values(w).copy() for w in range(5)]
                key = walk if walk == "block" else walk + os.environ["GF_REC_SEG"]
                if key not in ref: ref[key] = cur
                for w, (x, y) in enumerate(zip(cur, ref[key])):
                    if not np.array_equal(x, y):
                        bad += 1
                        d = np.abs(x - y); k = int(d.argmax())
                        print("%s path %s rep %d it %d: output %d differs, max %.3e at %d (of %d), n diff %d" % (case, key, rep, it, w, d.max(), k, x.size, int((d > 0).sum())), flush=True)
            D.close()
print("mismatches:", bad)
