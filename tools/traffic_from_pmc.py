"""Builds profiles/traffic.json from the two PMC passes of tools/profile_round.sh (FETCH_SIZE, WRITE_SIZE; KB units;
gfx950 correction: FETCH_SIZE counts 64 B per 128-B request -> bytes = 2*FETCH + WRITE, MI355X_MICROARCH.md HBM section)."""
import csv, json, sys, collections
tag, gps = sys.argv[1], int(sys.argv[2])
outfile = sys.argv[3] if len(sys.argv) > 3 else "profiles/traffic.json"       # usage: traffic_from_pmc.py <tag> <Gauss points of the profiled workload> [output]
def per_kernel(path, counter):
    tot, n = collections.defaultdict(float), collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter: continue
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("gf::", "")
        if not name.startswith(("kl_", "pen_")): name = name.split("<")[0]   # the gf kernels keep their template arguments: the full-pass and Newton-pass instances move different bytes
        tot[name] += float(r["Counter_Value"]) * 1024.0
        n[name].add(r["Dispatch_Id"])
    return {k: tot[k] / max(1, len(n[k])) for k in tot}, {k: len(n[k]) for k in tot}
f, nf = per_kernel("profiles/%s_pmc_fetch.csv" % tag, "FETCH_SIZE")
w, nw = per_kernel("profiles/%s_pmc_write.csv" % tag, "WRITE_SIZE")
kern = {}
for k in sorted(set(f) | set(w)):
    kern[k] = {"FETCH_SIZE_bytes_raw_per_launch": f.get(k, 0.0), "WRITE_SIZE_bytes_raw_per_launch": w.get(k, 0.0),
               "hbm_side_bytes_corrected_per_launch": 2 * f.get(k, 0.0) + w.get(k, 0.0), "launches": nf.get(k, nw.get(k, 0))}
el = sorted((k for k in kern if k.startswith("kl_element")), key=lambda k: -kern[k]["hbm_side_bytes_corrected_per_launch"])   # the full-pass instance first
out = {"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (profiles/%s_pmc_*.csv), bench.py (workload of %d Gauss points); averages per launch. "
               "Counters are KB at the L2<->fabric interface (Infinity-Cache hits included); corrected = 2*FETCH + WRITE "
               "(gfx950: FETCH_SIZE counts 64 B per 128-B request)." % (tag, gps),
       "workload_gps": gps, "kernels": kern,
       "element_kernel": el[0] if el else None,
       "element_kernel_bytes_per_launch": kern[el[0]]["hbm_side_bytes_corrected_per_launch"] if el else None}
# all kernels of one full pass (R + K + dR/dCP + dR/dh): the instances the full pass launches (element / gather kernels with the dR/dCP
# blocks, penalty kernels of the full pass, residual gather and clean-up), one launch each per step
def in_full_pass(k):
    if k.startswith(("kl_element", "kl_gather")): return "true" in k or k.startswith(("kl_element_mfma2", "kl_element_kernel")) or "<" not in k
    if k.startswith("pen_owner"): return k.replace(" ", "").split("<")[1].startswith(("3,2,true,true", "2,2,true,true", "4,"))
    if k.startswith("pen_row16"): return k.replace(" ", "").endswith("true,true>")
    return k.startswith(("pen_point", "kl_rgather", "zero_rows", "residual_finish"))      # pen_point_kernel / pen_point16_kernel: one launch per pass
out["full_pass_kernels"] = sorted(k for k in kern if in_full_pass(k))
out["full_pass_bytes_per_step"] = sum(kern[k]["hbm_side_bytes_corrected_per_launch"] for k in kern if in_full_pass(k))
# FP64 work per launch as the SQ counters see it (tools/profile_round.sh, fourth pass): flop = 512 * MFMA_MOPS_F64 + 64 * (2 FMA + ADD + MUL + TRANS)
# wave-level VALU instructions (all 64 lanes counted, active or not: issued work, what occupies the FP64 pipe)
import os
fp = "profiles/%s_pmc_fp64.csv" % tag
if os.path.exists(fp):
    acc, disp = collections.defaultdict(lambda: collections.defaultdict(float)), collections.defaultdict(set)
    for r in csv.DictReader(open(fp)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("gf::", "")
        if not name.startswith(("kl_", "pen_")): name = name.split("<")[0]
        acc[name][r["Counter_Name"]] += float(r["Counter_Value"]); disp[name].add(r["Dispatch_Id"])
    for k, c in acc.items():
        n = max(1, len(disp[k]))
        flop = (512.0 * c.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0.0) + 64.0 * (2.0 * c.get("SQ_INSTS_VALU_FMA_F64", 0.0) + c.get("SQ_INSTS_VALU_ADD_F64", 0.0)
                + c.get("SQ_INSTS_VALU_MUL_F64", 0.0) + c.get("SQ_INSTS_VALU_TRANS_F64", 0.0))) / n
        kern.setdefault(k, {})["fp64_flop_issued_per_launch"] = flop
        kern[k]["fp64_mfma_flop_per_launch"] = 512.0 * c.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0.0) / n
        if c.get("SQ_BUSY_CYCLES", 0.0) > 0: kern[k]["mfma_busy_over_busy_cycles"] = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / c["SQ_BUSY_CYCLES"]
        if c.get("SQ_WAVE_CYCLES", 0.0) > 0: kern[k]["mfma_busy_cycles_over_wave_cycles_x4"] = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4.0 * c["SQ_WAVE_CYCLES"])
    out["fp64_note"] = "SQ counters (profiles/%s_pmc_fp64.csv): issued FP64 flop per launch = 512 * MFMA_MOPS_F64 + 64 * (2 FMA_F64 + ADD_F64 + MUL_F64 + TRANS_F64); SQ_WAVE_CYCLES counts quad-cycles" % tag
    if el: out["element_kernel_fp64_flop_issued_per_launch"] = kern[el[0]].get("fp64_flop_issued_per_launch")
json.dump(out, open(outfile, "w"), indent=1)
print(json.dumps({k: round(v["hbm_side_bytes_corrected_per_launch"] / 1e9, 2) for k, v in kern.items()}))
