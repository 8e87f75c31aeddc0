#!/bin/bash
# Diagnostic: PMC counters (separate passes) of tools/variant_time.py for the row-record path (GF_WALK=2) and the block path, per kernel
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmcr
mkdir -p $O
for w in 2 0; do
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
  tag=$(echo $set | cut -d' ' -f1)
  GF_WALK=$w GF_WALK_SEG=${SEG:-24} rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/w${w}_$tag -o p -- python3 $R/tools/variant_time.py > $O/w${w}_$tag.log 2>&1 || echo "fail $w $tag"
done; done
python3 - <<'PY'
import csv, glob, os, collections
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/pmcr"
for f in sorted(glob.glob(O+"/*/**/*counter_collection.csv", recursive=True)):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0].replace("void gf::","")[:36]
        if k.startswith(("kl_", "pen_")):
            acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    print(f.split("/pmcr/")[1].split("/")[0])
    for k in acc:
        print("   %-38s" % k, {c: "%.4g"%(v/len(n[k])) for c,v in acc[k].items()})
PY
rm -rf $O/w*_*/
