"""Builds libgoldfish_hip.so (gfx950) in-tree with hipcc.  The .so is git-ignored but
travels to the GPU box with the gpurun snapshot."""
import os
import sys
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libgoldfish_hip.so")
SOURCES = ["gf_lib.hip", "gf_kernels.hpp", "gf_gauss_loop.hpp", "gf_element_mfma.hpp", "gf_element_mfma4.hpp", "gf_element_rec.hpp", "gf_element_rec4.hpp", "gf_extra_loads.hpp", "gf_penalty_row16.hpp", "gf_penalty_point16.hpp", "gf_setup.hpp", "kl_point.hpp",
           os.path.join("..", "..", "include", "goldfish_hip.h"), os.path.join("..", "..", "include", "goldfish_model.h")]


def _hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def needs_build(lib=LIB, sources=SOURCES):
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.getmtime(os.path.join(CSRC, s)) > t for s in sources)


BUILD_INFO = os.path.join(HERE, "_build_info.json")      # git-ignored like the libraries, travels with them to the GPU box (no git there): bench.py's "head"


def _write_build_info():
    """Where git is available (the build container), record the commit the libraries were built from; elsewhere (the GPU box) keep the file that came along."""
    import json
    try:
        root = os.path.dirname(HERE)
        head = subprocess.check_output(["git", "-C", root, "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
        dirty = bool(subprocess.check_output(["git", "-C", root, "status", "--porcelain", "--untracked-files=no"], stderr=subprocess.DEVNULL).decode().strip())
        with open(BUILD_INFO, "w") as f:
            json.dump({"head": head, "dirty": dirty, "libraries": dict(REPORT)}, f)
    except Exception:
        pass


def build_info():
    import json
    try:
        return json.load(open(BUILD_INFO))
    except Exception:
        return {}


REPORT = {}      # library -> "compiled" | "reused" of the last build() call (the driver's log shows whether the box compiled anything)


def build(force=False, verbose=False):
    REPORT.clear()
    REPORT[os.path.basename(LIB)] = "compiled" if (force or needs_build()) else "reused"
    if force or needs_build():
        cmd = [_hipcc(), "-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC"] + os.environ.get("GF_LIB_CXXFLAGS", "").split() + [
               os.path.join(CSRC, "gf_lib.hip"), "-o", LIB]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    host = os.path.join(CSRC, "libgf_point_host_test.so")
    if force or needs_build(host, ["point_host_test.cpp", "kl_point.hpp"]):
        subprocess.check_call([_hipcc(), "-O2", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC", "-x", "hip",
                               os.path.join(CSRC, "point_host_test.cpp"), "-o", host])
    solver = os.path.join(HERE, "libgoldfish_solver.so")      # block-banded L D L^T factorisation + solves on the device, include/goldfish_solver.h
    REPORT[os.path.basename(solver)] = "compiled" if (force or needs_build(solver, ["gf_solver.hip", "gf_nd_symbolic.hpp", os.path.join("..", "..", "include", "goldfish_solver.h")])) else "reused"
    if force or needs_build(solver, ["gf_solver.hip", "gf_nd_symbolic.hpp", os.path.join("..", "..", "include", "goldfish_solver.h")]):
        # -amdgpu-mfma-vgpr-form: v_mfma_f64_16x16x4 with its accumulators in arch VGPRs issues every 64 cycles, with AGPR accumulators (the
        # compiler's default for these kernels) every 131 (tools/ubench_acc.hip, profiles/r03_ubench_fp64_mfma.txt); the tile kernels have registers to spare
        # Measured without effect (round 4): -Xclang -target-feature -Xclang -load-store-opt, which keeps the MFMA operand reads as ds_read_b64 (the pass fuses two
        # k-steps into ds_read2_b64: half the LDS bandwidth, two-way conflicts on the 66-double row stride; SQ_LDS_BANK_CONFLICT 2.5 x SQ_ACTIVE_INST_LDS in
        # profiles/r04_solver_pmc_c4.txt) -- C4 factorisation 0.261 vs 0.257 s: the update kernels do not wait for the LDS.
        extra = os.environ.get("GF_SOLVER_CXXFLAGS", "").split()
        subprocess.check_call([_hipcc(), "-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC", "-pthread", "-mllvm", "-amdgpu-mfma-vgpr-form"] + extra +
                              [os.path.join(CSRC, "gf_solver.hip"), "-o", solver])
    _write_build_info()
    if verbose or os.environ.get("GF_BUILD_REPORT", "1") == "1":
        print("goldfish_amd.build: " + ", ".join("%s %s" % kv for kv in REPORT.items()) + " (hipcc --offload-arch=gfx950; a library is reused when no source is newer than it)", file=sys.stderr, flush=True)
    return LIB


if __name__ == "__main__":
    build(force=True, verbose=True)
