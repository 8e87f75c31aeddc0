#!/bin/bash
# Diagnostic: LDS / wait counters of the penalty kernels (tools/variant_time.py, 8 x 8-patch slice)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmcp
mkdir -p $O
for set in "SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  tag=$(echo $set | cut -d' ' -f2)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/$tag -o p -- python3 $R/tools/variant_time.py > $O/$tag.log 2>&1 || echo "fail $tag"
done
python3 - <<'PY'
import csv, glob, os, collections
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/pmcp"
for f in sorted(glob.glob(O+"/*/**/*counter_collection.csv", recursive=True)):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0].replace("void gf::","")[:36]
        if k.startswith("pen_"):
            acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    for k in acc:
        print("   %-38s" % k, {c: "%.4g"%(v/len(n[k])) for c,v in acc[k].items()})
PY
rm -rf $O/SQ_*/
