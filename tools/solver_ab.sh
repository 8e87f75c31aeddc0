for lib in std vf std vf; do
  if [ $lib = vf ]; then cp goldfish_amd/libgoldfish_solver_vf.so /tmp/libs.so; else cp goldfish_amd/libgoldfish_solver.so /tmp/libs.so; fi
  GF_SOLVER_LIB=/tmp/libs.so GF_SOLVER_C4=1 timeout -k 10 300 python tools/solver_bench.py 2>&1 | cut -c60-330 || exit 1
done
