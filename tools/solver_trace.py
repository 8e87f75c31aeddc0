"""Diagnostic: reads a rocprofv3 --kernel-trace CSV of tools/solver_bench.py (GF_SOLVER_GRAPH=0 GF_SOLVER_C4=1) and prints, for the LAST re-factorisation in the
trace, the wall time, the union of kernel-busy time, the time during which only a given kernel kind runs, and the time by number of kernels in flight."""
import csv, sys, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0],
                 int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1) // max(1, int(r["Workgroup_Size_X"]))))
rows.sort()
# factorisations start with nd_scatter_kernel; take the last one and everything up to the first substitution kernel behind it
starts = [i for i, r in enumerate(rows) if r[2].startswith("nd_scatter_kernel")]
i0 = starts[-1]
i1 = next((i for i in range(i0, len(rows)) if rows[i][2].startswith(("nd_fwd_front", "nd_gather_rhs", "residual"))), len(rows))
F = rows[i0:i1]
t0, t1 = F[0][0], max(r[1] for r in F)
print("factorisation: %d launches, wall %.1f ms" % (len(F), (t1 - t0) / 1e6))
pre = [r for r in rows[max(0, i0 - 6):i0] if "fillBuffer" in r[2] or "sumsq" in r[2]]
if pre: print("in front of it: " + ", ".join("%s %.2f ms" % (r[2][:40], (r[1] - r[0]) / 1e6) for r in pre) + "; from the start of the first of them to the scatter kernel %.2f ms" % ((t0 - pre[0][0]) / 1e6))
ev = []
for s, e, n, _ in F:
    ev.append((s, 1, n)); ev.append((e, -1, n))
ev.sort()
live = collections.Counter(); last = t0; by_n = collections.Counter(); only = collections.Counter(); idle = 0
for t, d, n in ev:
    dt = t - last
    k = sum(live.values())
    if k == 0: idle += dt
    else:
        by_n[min(k, 8)] += dt
        kinds = [x for x, c in live.items() if c > 0]
        if len(kinds) == 1: only[kinds[0]] += dt
    live[n] += d; last = t
print("idle (no kernel running) %.1f ms" % (idle / 1e6))
print("time by kernels in flight:", {k: round(v / 1e6, 1) for k, v in sorted(by_n.items())})
print("time with only one KIND of kernel running (ms):", {k: round(v / 1e6, 1) for k, v in only.most_common(8)})
tot = collections.Counter(); cnt = collections.Counter()
for s, e, n, _ in F: tot[n] += e - s; cnt[n] += 1
print("sum of durations (ms):", {k: (round(v / 1e6, 1), cnt[k]) for k, v in tot.most_common(10)})
# update_wide_kernel: one workgroup per 64 x 64 output tile, w = 4 panel columns per launch (the last group of a front may have fewer): 4 x 2 x 64^3 flop per workgroup
bins = collections.defaultdict(lambda: [0, 0.0, 0.0])
for s_, e, n, wg in F:
    if n.startswith("update_wide_kernel"):
        b = 1 << max(0, (wg - 1).bit_length())
        bins[b][0] += 1; bins[b][1] += (e - s_) / 1e6; bins[b][2] += wg * 4 * 2 * 64.0 ** 3
print("update_wide_kernel by launch size (workgroups <= bin: launches, ms, Tflop/s if w = 4):", {b: (v[0], round(v[1], 1), round(v[2] / v[1] / 1e9, 1)) for b, v in sorted(bins.items())})
# per tree height: a level of the factorisation starts with its extend-add rounds (nd_extend_add_batch_kernel; the leaves' level has none) -- wall time of the level,
# launches of the large fronts' chain kernels in it, summed durations of its wide updates, and how long NO wide update (batched or not) was running
lv = []
cur = None
for s, e, n, wg in F[1:]:
    if n.startswith("nd_extend_add_batch_kernel") and (cur is None or cur["seen_other"]):
        cur = {"t0": s, "t1": e, "rows": [], "seen_other": False}; lv.append(cur)
    if cur is None:
        cur = {"t0": s, "t1": e, "rows": [], "seen_other": False}; lv.append(cur)
    if not n.startswith("nd_extend_add_batch_kernel"): cur["seen_other"] = True
    cur["rows"].append((s, e, n, wg)); cur["t1"] = max(cur["t1"], e)
print("per level (height ascending): wall ms | diag_kernel launches (mean us) | panel us | narrow us | wide: launches, summed ms | batched kernels summed ms | time without any wide update running ms")
for i, L in enumerate(lv):
    R = L["rows"]; d = [r for r in R if r[2] == "diag_kernel"]; w = [r for r in R if r[2].startswith("update_wide_kernel")]
    pk = [r for r in R if r[2] == "panel_kernel"]; nk = [r for r in R if r[2].startswith("update_narrow_kernel")]
    b = [r for r in R if r[2].startswith("nd_")]
    ev2 = sorted([(r[0], 1) for r in R if "update_wide" in r[2]] + [(r[1], -1) for r in R if "update_wide" in r[2]])
    live2 = 0; last2 = L["t0"]; nowide = 0
    for t, dd in ev2:
        if live2 == 0: nowide += t - last2
        live2 += dd; last2 = t
    nowide += L["t1"] - last2 if live2 == 0 else 0
    mean = lambda rows: sum(r[1] - r[0] for r in rows) / max(1, len(rows)) / 1e3
    print("  level %2d: %7.2f | %4d (%5.1f) | %5.1f | %5.1f | %3d, %6.2f | %6.2f | %6.2f" % (i, (L["t1"] - L["t0"]) / 1e6, len(d), mean(d), mean(pk), mean(nk), len(w),
          sum(r[1] - r[0] for r in w) / 1e6, sum(r[1] - r[0] for r in b) / 1e6, nowide / 1e6))
# substitution sweeps behind the last factorisation: a sweep ends with nd_out_kernel; the last sweep with ONE right-hand side (kernel names "<1>") and the last with three
rest = rows[i1:]
sweeps, cur = [], []
for r in rest:
    cur.append(r)
    if r[2].startswith("nd_out_kernel"):
        sweeps.append(cur); cur = []
for tag in ("<1>", "<3>"):
    sel = [sw for sw in sweeps if any(tag in r[2] for r in sw) and any("nd_fwd_front" in r[2] for r in sw)]
    if not sel: continue
    sw = [r for r in sel[-1] if "residual" not in r[2] and "sumsq" not in r[2]]
    a, b = min(r[0] for r in sw), max(r[1] for r in sw)
    tot = collections.Counter(); cnt = collections.Counter()
    for s_, e, n, _ in sw: tot[n.split("(")[0]] += e - s_; cnt[n.split("(")[0]] += 1
    ev3 = sorted([(r[0], 1) for r in sw] + [(r[1], -1) for r in sw]); live3 = 0; last3 = a; idle3 = 0
    for t, d in ev3:
        if live3 == 0: idle3 += t - last3
        live3 += d; last3 = t
    print("last substitution sweep %s: %d launches, wall %.2f ms, no kernel running %.2f ms; sum of durations (ms, launches): %s" % (
        tag, len(sw), (b - a) / 1e6, idle3 / 1e6, {k: (round(v / 1e6, 2), cnt[k]) for k, v in tot.most_common(12)}))
# the whole-front substitution kernels of the last one-right-hand-side sweep, launch by launch (tree height ascending for the forward kernel, descending for the backward one)
sel = [sw for sw in sweeps if any("<1>" in r[2] for r in sw) and any("nd_fwd_front" in r[2] for r in sw)]
if sel:
    for nm in ("nd_fwd_front_kernel<1>", "nd_bwd_front_kernel<1>"):
        print(nm + " per launch (ms, workgroups):", [(round((r[1] - r[0]) / 1e6, 2), r[3]) for r in sel[-1] if r[2].startswith(nm)])
