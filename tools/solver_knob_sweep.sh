# measurement: C4 re-factorisation / solve time against the solver's tuning switches (run on the GPU box); round 5: leaf size, batch threshold, panel group, sub-group
run() { echo -n "$1: "; env $1 GF_SOLVER_C4=1 GF_SOLVER_HOST=0 GF_SOLVER_REFACTOR_SAMPLES=5 timeout -k 10 300 python tools/solver_bench.py 2>&1 | tail -1 | grep -o "factor storage [0-9.]* GB\|re-factorisation [0-9.]* s\|[0-9.]* TFLOP/s\|Newton solve [0-9.]* s" | tr "\n" ";"; echo; }
run "GF_SOLVER_LEAF=128"
run "GF_SOLVER_LEAF=96"
run "GF_SOLVER_LEAF=192"
run "GF_SOLVER_LEAF=256"
run "GF_SOLVER_LEAF=384"
run "GF_SOLVER_BATCH_BLK=64"
run "GF_SOLVER_BATCH_BLK=128"
run "GF_SOLVER_BATCH_BLK=160"
run "GF_SOLVER_PANEL_W=4 GF_SOLVER_SUBGROUP=0"
run "GF_SOLVER_PANEL_W=6 GF_SOLVER_SUBGROUP=3"
