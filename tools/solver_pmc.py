"""Diagnostic: sums the SQ counters of a rocprofv3 --pmc run of tools/solver_bench.py per kernel kind."""
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if (k, r["Dispatch_Id"]) not in seen: seen.add((k, r["Dispatch_Id"])); n[k] += 1
for k, c in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0)):
    if c.get("SQ_BUSY_CYCLES", 0) <= 0: continue
    line = "%-34s launches %6d" % (k[:34], n[k])
    for name in sorted(c): line += "  %s %.3g" % (name.replace("SQ_", ""), c[name])
    if c.get("SQ_WAVE_CYCLES"): line += "  | MFMA busy / (4 x wave cycles) %.3f" % (c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (4 * c["SQ_WAVE_CYCLES"]))
    # (no "MFMA busy / busy cycles": SQ_VALU_MFMA_BUSY_CYCLES is summed per SIMD, SQ_BUSY_CYCLES per shader engine -- their quotient (~20 in profiles/r04_solver_pmc_c4.txt)
    #  is not a utilisation)
    print(line)
