R=r06s
mkdir -p gpurun_out/$R
python bench.py --steps 50 --warmup 5 > gpurun_out/$R/bench.json 2> gpurun_out/$R/bench.err; cut -c1-300 gpurun_out/$R/bench.json; tail -3 gpurun_out/$R/bench.err
for c in 6 7 8; do python bench.py --config $c --steps 50 --warmup 5 --no-cpu > gpurun_out/$R/bench_config$c.json 2>gpurun_out/$R/bench_config$c.err; cut -c1-200 gpurun_out/$R/bench_config$c.json; tail -2 gpurun_out/$R/bench_config$c.err; done
python bench.py --config 1 --steps 50 --warmup 5 --new-frames 0 > gpurun_out/$R/bench_config1.json 2>/dev/null; cut -c1-200 gpurun_out/$R/bench_config1.json
python bench.py --config 2 --steps 50 --warmup 5 --no-cpu > gpurun_out/$R/bench_config2.json 2>/dev/null; cut -c1-200 gpurun_out/$R/bench_config2.json
