"""Builds profiles/traffic.json from the PMC passes of tools/profile_round.sh (FETCH_SIZE, WRITE_SIZE; KB units;
gfx950 correction: FETCH_SIZE counts 64 B per 128-B request -> bytes = 2*FETCH + WRITE, MI355X_MICROARCH.md HBM section).

usage: traffic_from_pmc.py <tag> <Gauss points of the profiled workload> [output] [chunks]
``chunks``: launches of the element kernel per pass (full C5 on one GPU runs its element blocks in chunks of patches: 5 launches per pass); the per-"launch" figures
of the element kernel and the per-step sum are then per PASS (all chunks), which is what bench.py compares them with (its kernel time is the sum over a step's launches).

Which launches belong to the full pass (R + K + dR/dCP + dR/dh):  bench.py first runs its warm-up + timed FULL passes, then (unless
--full-pass-only) a few Newton passes (R + K).  A kernel whose template arguments carry the pass (WITHC / WITHK instances) is told apart
by name; a kernel WITHOUT such an argument (kl_gather_kernel<4>, pen_point*, zero_rows ...) is launched by both kinds of pass under one
name, so its dispatches are split by launch order: the first n_full dispatches (n_full = the launches of the full-pass element kernel) are
the full-pass ones.  Per-launch averages never mix the two (round-3 verdict: the p = 4 gather's 30.6 GB was such a mix)."""
import collections
import csv
import json
import os
import sys


def short_name(kernel_name):
    name = kernel_name.split("(")[0].replace("void ", "").replace("gf::", "")
    if not name.startswith(("kl_", "pen_")):
        name = name.split("<")[0]       # the gf kernels keep their template arguments: the full-pass and Newton-pass instances move different bytes
    return name


def targs(name):
    return name.replace(" ", "").split("<")[1].rstrip(">").split(",") if "<" in name else []


def pass_tag(name):
    """'full' / 'other' when the template arguments say which pass the instance belongs to, None when the name does not tell."""
    a = targs(name)
    if name.startswith(("kl_element_rec4_kernel", "kl_gather_rec4_kernel")):      # <PASS, NC> / <NC>: NC = 21 is the full record layout (full pass), 9 the Newton pass
        return "full" if a and a[-1] == "21" else "other"
    if name.startswith(("kl_element_rec_kernel", "kl_gather_rec_kernel", "kl_element_mfma_kernel", "kl_element_mfma4_kernel", "kl_gather1_kernel")):
        if name.startswith("kl_element_rec_kernel"):
            # <P, WITHC, ALLF, SF>: the full pass runs the ALLF instances (round 5: one per kind of patch, SF = 1 polynomial / 2 rational; their bytes add up);
            # <P, true, false, .> is a partial pass with dR/dCP (linearize after a Newton solve), <P, false, ...> the Newton pass
            return "full" if a[1:3] == ["true", "true"] else "other"
        return "full" if a and a[-1] == "true" else ("other" if a else None)
    if name.startswith("pen_owner_kernel"):
        return "full" if a[-2:] == ["true", "true"] else "other"
    if name.startswith("pen_row16_kernel"):
        return "full" if a[-2:] == ["true", "true"] else "other"
    return None


GF_PASS_KERNELS = ("kl_element", "kl_gather", "kl_rgather", "kl_extra_loads", "pen_point", "pen_owner", "pen_row16", "zero_rows", "residual_finish")


def read_counter(path, counter):
    """{kernel: [(dispatch id, bytes)] in launch order}"""
    rows = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        rows[short_name(r["Kernel_Name"])].append((int(r["Dispatch_Id"]), float(r["Counter_Value"]) * 1024.0))
    return {k: sorted(v) for k, v in rows.items()}


def full_pass_launches(rows):
    """Number of full passes in the run = launches of the full-pass element kernel (per chunk of the model: all chunks count as one pass each
    only if the element kernel is launched once per pass, which holds for every profiled workload; stated in the output)."""
    el = [k for k in rows if k.startswith("kl_element") and pass_tag(k) == "full"]
    if not el:
        el = [k for k in rows if k.startswith("kl_element")]
    return max((len(rows[k]) for k in el), default=0), el


def split(rows):
    """Per kernel: (average bytes per FULL-pass launch or None, average per other launch or None, launches full, launches other)."""
    n_full, _ = full_pass_launches(rows)
    out = {}
    for k, v in rows.items():
        tag = pass_tag(k)
        vals = [b for _, b in v]
        if tag == "full":
            full, other = vals, []
        elif tag == "other":
            full, other = [], vals
        elif k.startswith(GF_PASS_KERNELS):
            full, other = vals[:n_full], vals[n_full:]      # launch order: the full passes come first (bench.py)
        else:
            full, other = [], vals
        out[k] = (sum(full) / len(full) if full else None, sum(other) / len(other) if other else None, len(full), len(other))
    return out


def build(tag, gps, prof_dir="profiles", chunks=1):
    f = split(read_counter(os.path.join(prof_dir, "%s_pmc_fetch.csv" % tag), "FETCH_SIZE"))
    w = split(read_counter(os.path.join(prof_dir, "%s_pmc_write.csv" % tag), "WRITE_SIZE"))
    kern = {}
    for k in sorted(set(f) | set(w)):
        ff, fo, nf, no = f.get(k, (None, None, 0, 0))
        wf, wo, nwf, nwo = w.get(k, (None, None, 0, 0))
        e = {"launches_full_pass": max(nf, nwf), "launches_other": max(no, nwo)}
        if max(nf, nwf):
            e["FETCH_SIZE_bytes_raw_per_launch"] = ff or 0.0
            e["WRITE_SIZE_bytes_raw_per_launch"] = wf or 0.0
            e["hbm_side_bytes_corrected_per_launch"] = 2 * (ff or 0.0) + (wf or 0.0)
        if max(no, nwo):
            e["other_pass_hbm_side_bytes_corrected_per_launch"] = 2 * (fo or 0.0) + (wo or 0.0)
        if not max(nf, nwf):           # a kernel only other passes launch: keep the old key so that its bytes are still listed
            e["hbm_side_bytes_corrected_per_launch"] = e.get("other_pass_hbm_side_bytes_corrected_per_launch", 0.0)
        kern[k] = e
    full = sorted(k for k in kern if kern[k]["launches_full_pass"] and k.startswith(GF_PASS_KERNELS))
    el = sorted((k for k in full if k.startswith("kl_element")), key=lambda k: -kern[k]["hbm_side_bytes_corrected_per_launch"])
    n_full = max((kern[k]["launches_full_pass"] for k in el), default=0)
    if chunks > 1:
        n_full = max(1, n_full // chunks)              # passes, not launches
    out = {"launches_of_the_element_kernel_per_pass": chunks, "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (profiles/%s_pmc_*.csv), bench.py (workload of %d Gauss points); averages per launch. "
                   "Counters are KB at the L2<->fabric interface (Infinity-Cache hits included); corrected = 2*FETCH + WRITE "
                   "(gfx950: FETCH_SIZE counts 64 B per 128-B request).  Kernels launched under one name by full and Newton passes are split by launch order "
                   "(tools/traffic_from_pmc.py)." % (tag, gps),
           "workload_gps": gps, "source": "profiles/%s_pmc_{fetch,write,fp64}.csv" % tag, "full_passes_in_run": n_full, "kernels": kern,
           "element_kernel": el[0] if el else None,
           "element_kernel_bytes_per_launch": sum(kern[k]["hbm_side_bytes_corrected_per_launch"] * kern[k]["launches_full_pass"] for k in el) / n_full if el else None,
           "element_kernels": el}
    out["full_pass_kernels"] = full
    # bytes of one full pass: every full-pass launch of every kernel of the pass, divided by the number of full passes in the run
    out["full_pass_bytes_per_step"] = (sum(kern[k]["hbm_side_bytes_corrected_per_launch"] * kern[k]["launches_full_pass"] for k in full) / n_full) if n_full else None
    # FP64 work per launch as the SQ counters see it (tools/profile_round.sh, fourth pass): flop = 512 * MFMA_MOPS_F64 + 64 * (2 FMA + ADD + MUL + TRANS)
    # wave-level VALU instructions (all 64 lanes counted, active or not: issued work, what occupies the FP64 pipe)
    fp = os.path.join(prof_dir, "%s_pmc_fp64.csv" % tag)
    if os.path.exists(fp):
        acc, disp = collections.defaultdict(lambda: collections.defaultdict(float)), collections.defaultdict(set)
        for r in csv.DictReader(open(fp)):
            name = short_name(r["Kernel_Name"])
            acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[name].add(r["Dispatch_Id"])
        for k, c in acc.items():
            n = max(1, len(disp[k]))
            flop = (512.0 * c.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0.0) + 64.0 * (2.0 * c.get("SQ_INSTS_VALU_FMA_F64", 0.0) + c.get("SQ_INSTS_VALU_ADD_F64", 0.0)
                    + c.get("SQ_INSTS_VALU_MUL_F64", 0.0) + c.get("SQ_INSTS_VALU_TRANS_F64", 0.0))) / n
            kern.setdefault(k, {})["fp64_flop_issued_per_launch"] = flop
            kern[k]["fp64_mfma_flop_per_launch"] = 512.0 * c.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0.0) / n
            if c.get("SQ_BUSY_CYCLES", 0.0) > 0:
                kern[k]["mfma_busy_over_busy_cycles"] = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / c["SQ_BUSY_CYCLES"]
            if c.get("SQ_WAVE_CYCLES", 0.0) > 0:
                kern[k]["mfma_busy_cycles_over_wave_cycles_x4"] = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4.0 * c["SQ_WAVE_CYCLES"])
        out["fp64_note"] = "SQ counters (profiles/%s_pmc_fp64.csv): issued FP64 flop per launch = 512 * MFMA_MOPS_F64 + 64 * (2 FMA_F64 + ADD_F64 + MUL_F64 + TRANS_F64); SQ_WAVE_CYCLES counts quad-cycles" % tag
        if el:
            out["element_kernel_fp64_flop_issued_per_launch"] = sum(kern[k].get("fp64_flop_issued_per_launch", 0.0) * kern[k]["launches_full_pass"] for k in el) / n_full
    return out


if __name__ == "__main__":
    tag, gps = sys.argv[1], int(sys.argv[2])
    outfile = sys.argv[3] if len(sys.argv) > 3 else "profiles/traffic.json"
    out = build(tag, gps, chunks=int(sys.argv[4]) if len(sys.argv) > 4 else 1)
    try:
        import subprocess
        out["generated_at_commit"] = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
    except Exception:
        out["generated_at_commit"] = None
    json.dump(out, open(outfile, "w"), indent=1)
    print(json.dumps({k: round(v["hbm_side_bytes_corrected_per_launch"] / 1e9, 2) for k, v in out["kernels"].items() if "hbm_side_bytes_corrected_per_launch" in v}))
    print("full pass:", out["full_pass_kernels"], "%.2f GB per step" % (out["full_pass_bytes_per_step"] / 1e9))
