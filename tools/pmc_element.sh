#!/bin/bash
# Run ON THE GPU BOX: SQ counters of the element kernel variants (one wave / two waves per element) on the 8x8-patch slice of C4.
# usage: tools/pmc_element.sh  -> gpurun_out/pmc_element.txt
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmce; mkdir -p $O
for tw in 0 1; do
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_LDS"; do
  tag=$(echo $set | cut -d' ' -f2)
  GF_TWOWAVE=$tw rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/tw${tw}_$tag -o p -- python3 $R/tools/variant_time.py > $O/tw${tw}_$tag.log 2>&1 || echo "fail $tw $tag"
done; done
python3 - <<'PY'
import csv, glob, os, collections
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/pmce"
for f in sorted(glob.glob(O+"/*/**/*counter_collection.csv", recursive=True)):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "kl_element_mfma" in k and ("true" in k or "mfma2" in k):
            k=k.split("(")[0][-40:]; acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[(k,r["Counter_Name"])]+=1
    print(f.split("/pmce/")[1].split("/")[0])
    for k in acc: print("   ", k, {c: "%.4g"%(v/n[(k,c)]) for c,v in acc[k].items()})
PY
