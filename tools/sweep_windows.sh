# Window-width sweep for the key tables (run on the GPU box): prints one bench line per setting.
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/sweep
run() { tag=$1; shift; env "$@" timeout -k 10 120 python3 bench.py --no-cpu-baseline > gpurun_out/sweep/b_$tag.json 2> gpurun_out/sweep/b_$tag.err || exit 1; echo "$tag: $(python3 tools/bench_summary.py gpurun_out/sweep/b_$tag.json | head -1)"; }
for s in "$@"; do
  tag=$(echo "$s" | tr ' =' '__' | sed 's/ZG16_MSM_TABLE_WINDOW//g')
  run "$tag" $s
done
