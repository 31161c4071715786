mkdir -p gpurun_out/r06p
timeout 1500 python bench.py > gpurun_out/r06p/bench_default.log 2> gpurun_out/r06p/bench_default.err; tail -c 4200 gpurun_out/r06p/bench_default.log; tail -3 gpurun_out/r06p/bench_default.err
