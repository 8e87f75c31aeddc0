#!/bin/bash
# Diagnostic: kernel start / end times of one bench step (do the penalty kernels overlap the gather?)
out=$GRAFT_REPO_ROOT/gpurun_out; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out/tl -o tl -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/tl.log 2>&1
python3 - <<'PY'
import csv, glob, os
f = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/tl/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last full pass: find the last kl_element_rec_kernel<3, true>
idx = [i for i, r in enumerate(rows) if "kl_element_rec_kernel<3, true>" in r["Kernel_Name"]]
i0 = idx[2] if len(idx) > 2 else idx[-1]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i0 + 9]:
    print("%-50s start %9.1f us  end %9.1f us  queue %s" % (r["Kernel_Name"].split("(")[0].replace("void gf::", "")[:50], (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, r.get("Queue_Id", "?")))
PY
rm -rf $out/tl
