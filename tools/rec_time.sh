#!/bin/bash
# Diagnostic (not product): element-kernel and step time of the assembly paths on the 8x8-patch slice of C4 (tools/variant_time.py)
cd $GRAFT_REPO_ROOT
for w in 0 2; do for seg in 12 24 48; do
  if [ $w = 0 ] && [ $seg != 12 ]; then continue; fi
  GF_WALK=$w GF_WALK_SEG=$seg GF_TAG="walk=$w seg=$seg" GF_TWOWAVE=${TW:-1} timeout -k 10 300 python3 tools/variant_time.py || exit 1
done; done
GF_TWOWAVE=0 GF_TAG="one wave" timeout -k 10 300 python3 tools/variant_time.py
