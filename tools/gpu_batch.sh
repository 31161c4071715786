mkdir -p gpurun_out/r06l
timeout 900 python -m pytest tests/test_gpu_sharded.py -x -q -m gpu > gpurun_out/r06l/tests.log 2>&1; tail -15 gpurun_out/r06l/tests.log
TCGNN_BENCH_FORCE_SHARDED=1 timeout 600 python bench.py --steps 5 --warmup 2 > gpurun_out/r06l/bench_sharded.log 2> gpurun_out/r06l/bench_sharded.err; tail -c 1500 gpurun_out/r06l/bench_sharded.log; tail -5 gpurun_out/r06l/bench_sharded.err
