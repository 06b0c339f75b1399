# Run bench.py once per environment setting given as arguments ("VAR=value [VAR2=value]"), on the GPU box.
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/sweep
i=0
for s in "$@"; do
  i=$((i+1))
  env $s timeout -k 10 120 python3 bench.py --no-cpu-baseline > gpurun_out/sweep/e_$i.json 2> gpurun_out/sweep/e_$i.err || exit 1
  echo "$s: $(python3 tools/bench_summary.py gpurun_out/sweep/e_$i.json | head -1)"
done
