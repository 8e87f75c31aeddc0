#!/bin/bash
# Run ON THE GPU BOX (via gpurun): kernel-trace stats + the two PMC passes the microarch guide prescribes for HBM bytes.
# usage: tools/profile_round.sh <tag> [bench.py arguments]   -> gpurun_out/<tag>_{kernel_stats.csv,pmc_fetch.csv,pmc_write.csv,pmc_fp64.csv,bench_under_rocprof.json}
set -e
tag=$1; shift; extra="$@"; root=$GRAFT_REPO_ROOT; out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$tag -o st -- python3 $root/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary $extra > $out/${tag}_bench_under_rocprof.json 2> $out/${tag}_prof.log
cp $(find $out/prof_$tag -name "*kernel_stats.csv" | head -1) $out/${tag}_kernel_stats.csv
echo "stats done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/prof_${tag}_f -o f -- python3 $root/bench.py --steps 1 --warmup 1 --no-cpu-baseline --full-pass-only $extra > /dev/null 2>> $out/${tag}_prof.log
cp $(find $out/prof_${tag}_f -name "*counter_collection.csv" | head -1) $out/${tag}_pmc_fetch.csv
echo "fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/prof_${tag}_w -o w -- python3 $root/bench.py --steps 1 --warmup 1 --no-cpu-baseline --full-pass-only $extra > /dev/null 2>> $out/${tag}_prof.log
cp $(find $out/prof_${tag}_w -name "*counter_collection.csv" | head -1) $out/${tag}_pmc_write.csv
echo "write done"
# FP64 work as the hardware counts it (SQ counters, one more pass): MFMA ops (512 flop each), FMA / ADD / MUL / TRANS wave instructions, matrix-pipe busy cycles
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $out/prof_${tag}_q -o q -- python3 $root/bench.py --steps 1 --warmup 1 --no-cpu-baseline --full-pass-only $extra > /dev/null 2>> $out/${tag}_prof.log
cp $(find $out/prof_${tag}_q -name "*counter_collection.csv" | head -1) $out/${tag}_pmc_fp64.csv
rm -rf $out/prof_$tag $out/prof_${tag}_f $out/prof_${tag}_w $out/prof_${tag}_q
echo "fp64 done"
