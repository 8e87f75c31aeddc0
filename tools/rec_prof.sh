#!/bin/bash
# Diagnostic: rocprofv3 kernel stats of tools/variant_time.py for the row-record path (GF_WALK=2) and the block path
out=$GRAFT_REPO_ROOT/gpurun_out; cd /tmp && export TMPDIR=/tmp
for w in 2 0; do
  export GF_WALK=$w GF_WALK_SEG=${SEG:-24}
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_rec$w -o st -- python3 $GRAFT_REPO_ROOT/tools/variant_time.py > $out/rec_prof$w.log 2>&1 || exit 1
  cp $(find $out/prof_rec$w -name "*kernel_stats.csv" | head -1) $out/rec_kernel_stats_walk$w.csv
  rm -rf $out/prof_rec$w
done
