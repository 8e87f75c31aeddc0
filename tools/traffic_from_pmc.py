"""Builds profiles/traffic.json from the two PMC passes of tools/profile_round.sh (FETCH_SIZE, WRITE_SIZE; KB units;
gfx950 correction: FETCH_SIZE counts 64 B per 128-B request -> bytes = 2*FETCH + WRITE, MI355X_MICROARCH.md HBM section)."""
import csv, json, sys, collections
tag, gps = sys.argv[1], int(sys.argv[2])
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
out = {"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (profiles/%s_pmc_*.csv), bench.py C4; averages per launch. "
               "Counters are KB at the L2<->fabric interface (Infinity-Cache hits included); corrected = 2*FETCH + WRITE "
               "(gfx950: FETCH_SIZE counts 64 B per 128-B request)." % tag,
       "workload_gps": gps, "kernels": kern,
       "element_kernel": el[0] if el else None,
       "element_kernel_bytes_per_launch": kern[el[0]]["hbm_side_bytes_corrected_per_launch"] if el else None}
json.dump(out, open("profiles/traffic.json", "w"), indent=1)
print(json.dumps({k: round(v["hbm_side_bytes_corrected_per_launch"] / 1e9, 2) for k, v in kern.items()}))
