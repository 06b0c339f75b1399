for lr in 8 10 12 14 16 18; do timeout -k 10 120 python bench.py --log-rows $lr --steps 20 --warmup 3 > gpurun_out/small_$lr.log 2>gpurun_out/small_$lr.err || exit 1; done
