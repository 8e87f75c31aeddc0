# measurement: C4 re-factorisation time against the solver's tuning switches (run on the GPU box)
run() { echo "$1"; env $1 GF_SOLVER_C4=1 GF_SOLVER_HOST=0 timeout -k 10 300 python tools/solver_bench.py 2>&1 | tail -1 | grep -o "re-factorisation [0-9.]* s ([0-9.]* TFLOP/s), Newton solve [0-9.]* s"; }
run "GF_SOLVER_BATCH_PANEL_W=8"
run "GF_SOLVER_BATCH_PANEL_W=6"
run "GF_SOLVER_BATCH_PANEL_W=4"
run "GF_SOLVER_BATCH_PANEL_W=3"
run "GF_SOLVER_BATCH_PANEL_W=4 GF_SOLVER_BATCH_BLK=64"
run "GF_SOLVER_BATCH_PANEL_W=4 GF_SOLVER_BATCH_BLK=128"
