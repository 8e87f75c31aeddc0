set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmcw
mkdir -p $O
rocprofv3 -L > $O/counters.txt 2>&1 || true
for lib in libgf_NOLOAD libgoldfish_hip; do
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_WRITE_sum" "TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum"; do
  tag=$(echo $set | cut -d' ' -f1)
  GF_LIB=$R/goldfish_amd/$lib.so rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/${lib}_$tag -o p -- python3 $R/tools/variant_time.py > $O/${lib}_$tag.log 2>&1 || echo "fail $lib $tag"
done; done
python3 - <<'PY'
import csv, glob, os, collections
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/pmcw"
for f in sorted(glob.glob(O+"/*/**/*counter_collection.csv", recursive=True)):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"][:40]
        if "walk" in k or "element_mfma" in k or "gather1" in k:
            acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[(k,r["Counter_Name"])]+=1
    print(f.split("/pmcw/")[1].split("/")[0])
    for k in acc:
        print("   ", k, {c: "%.4g"%(v/n[(k,c)]) for c,v in acc[k].items()})
PY
