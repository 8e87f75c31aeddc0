#!/bin/bash
# Run ON THE GPU BOX (via gpurun): kernel-trace stats + the two PMC passes the microarch guide prescribes for HBM bytes.
# usage: tools/profile_round.sh <tag>      -> gpurun_out/<tag>_{kernel_stats.csv,pmc_fetch.csv,pmc_write.csv,bench_under_rocprof.json}
set -e
tag=$1; root=$GRAFT_REPO_ROOT; out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$tag -o st -- python3 $root/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $out/${tag}_bench_under_rocprof.json 2> $out/${tag}_prof.log
cp $(find $out/prof_$tag -name "*kernel_stats.csv" | head -1) $out/${tag}_kernel_stats.csv
echo "stats done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/prof_${tag}_f -o f -- python3 $root/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>> $out/${tag}_prof.log
cp $(find $out/prof_${tag}_f -name "*counter_collection.csv" | head -1) $out/${tag}_pmc_fetch.csv
echo "fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/prof_${tag}_w -o w -- python3 $root/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>> $out/${tag}_prof.log
cp $(find $out/prof_${tag}_w -name "*counter_collection.csv" | head -1) $out/${tag}_pmc_write.csv
rm -rf $out/prof_$tag $out/prof_${tag}_f $out/prof_${tag}_w
echo "write done"
