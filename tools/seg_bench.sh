#!/bin/bash
# Diagnostic: bench.py at C4 for several work-item lengths of the row-record path
cd $GRAFT_REPO_ROOT
for seg in "$@"; do
  GF_WALK_SEG=$seg timeout -k 10 300 python3 bench.py --no-cpu-baseline 2>/dev/null > gpurun_out/seg_$seg.json || exit 1
  SEG=$seg python3 -c "
import json,os
b=json.loads(open('gpurun_out/seg_%s.json'%os.environ['SEG']).read().strip().split(chr(10))[-1])
print('seg', os.environ['SEG'], 'ms/step %.3f'%b['ms_per_step'], 'GP/s %.4g'%b['value'], 'newton %.3f'%b['newton_pass']['ms'], 'element %.3f'%b['roofline']['avg_launch_ms'], 'bytes %.4g'%b['device_bytes'])"
done
