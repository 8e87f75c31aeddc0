#!/bin/bash
# Run ON THE GPU BOX: kernel trace of tools/solver_bench.py at C4 with direct launches (GF_SOLVER_GRAPH=0) -> gpurun_out/<tag>_solver_timeline.txt (tools/solver_trace.py)
tag=${1:-solver}; root=$GRAFT_REPO_ROOT; out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
GF_SOLVER_GRAPH=0 GF_SOLVER_C4=1 GF_SOLVER_HOST=0 rocprofv3 --kernel-trace --output-format csv -d $out/prof_$tag -o tr -- python3 $root/tools/solver_bench.py > $out/${tag}_solver_bench_under_rocprof.txt 2> $out/${tag}_solver_prof.log || exit 1
python3 $root/tools/solver_trace.py $(find $out/prof_$tag -name "*kernel_trace.csv" | head -1) > $out/${tag}_solver_timeline.txt || exit 1
rm -rf $out/prof_$tag
cat $out/${tag}_solver_timeline.txt | cut -c1-400
