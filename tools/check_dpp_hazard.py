"""Static check of the built device code: the v_fmac_f64_dpp of the element / penalty kernels are inline assembly, so LLVM's
hazard recogniser does not see them.  gfx90a+ needs two wait states between a VALU write of a VGPR and a DPP read of it; the kernels put
an `s_nop 1` tied to the source registers in front of the first DPP read (dpp_source_fence), but a copy or a spill reload placed by the
register allocator behind that fence would read stale lanes without any diagnostic.  This script disassembles the code object and checks
every DPP instruction: no VALU instruction (v_*, MFMA included) may write its DPP source (src0) within the two preceding wait states
(an instruction counts one, `s_nop N` counts N + 1).  The same window is checked between a v_fmac_f64_dpp result and an MFMA reading it.
usage: check_dpp_hazard.py [lib.so]   -> prints a summary, exit code 1 on a violation"""
import os, re, shutil, subprocess, sys, tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def _regs(op):
    op = op.strip()
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", op)
    if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", op)
    return {int(m.group(1))} if m else set()


def disassemble(lib):
    tmp = tempfile.mkdtemp()
    try:
        so = os.path.join(tmp, "lib.so")
        shutil.copy(lib, so)
        subprocess.run([OBJDUMP, "--offloading", so], check=True, capture_output=True)
        co = [f for f in os.listdir(tmp) if "amdgcn" in f]
        if not co: raise RuntimeError("no device code object in " + lib)
        dis = subprocess.run([OBJDUMP, "-d", os.path.join(tmp, co[0])], check=True, capture_output=True, text=True).stdout
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return dis


def scan(dis):
    """dis: llvm-objdump -d text.  Returns (number of DPP instructions, kernels that hold them, violations)."""
    kernel, hist, ndpp, nk, bad = None, [], 0, set(), []
    for line in dis.split("\n"):
        m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
        if m: kernel, hist = m.group(1), []; continue
        code = line.split("//")[0].strip()
        if not code or kernel is None: continue
        parts = code.split(None, 1)
        mn, ops = parts[0], (parts[1] if len(parts) > 1 else "")
        opl = [o.strip() for o in re.split(r",\s*(?![^\[]*\])", ops)]
        if mn.endswith("_dpp") or "_dpp" in mn:
            ndpp += 1; nk.add(kernel)
            src = _regs(opl[1].split()[0]) if len(opl) > 1 else set()
            ws = 0
            for pm, pd, pw in reversed(hist):
                if ws >= 2: break
                if pm.startswith("v_") and (pd & src): bad.append((kernel, "DPP source %s written by %s %d wait state(s) earlier" % (opl[1], pm, ws)))
                ws += pw
        if mn.startswith("v_mfma"):
            srcs = set()
            for o in opl[1:3]: srcs |= _regs(o)
            ws = 0
            for pm, pd, pw in reversed(hist):
                if ws >= 2: break
                if "_dpp" in pm and (pd & srcs): bad.append((kernel, "MFMA operand written by %s %d wait state(s) earlier" % (pm, ws)))
                ws += pw
        dst = _regs(opl[0]) if (mn.startswith("v_") and opl) else set()
        wait = 1
        if mn == "s_nop":
            try: wait = int(ops.strip(), 0) + 1
            except ValueError: wait = 1
        hist.append((mn, dst, wait))
        if len(hist) > 8: hist.pop(0)
    return ndpp, sorted(nk), bad


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "goldfish_amd", "libgoldfish_hip.so")
    n, kernels, bad = scan(disassemble(lib))
    print("%d DPP instructions in %d kernels checked, %d violations" % (n, len(kernels), len(bad)))
    for k, msg in bad[:20]: print("  ", k[:80], msg)
    sys.exit(1 if bad else 0)
