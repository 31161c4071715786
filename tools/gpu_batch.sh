mkdir -p gpurun_out/r06i
timeout 1700 python -m pytest tests -q -m gpu -rs > gpurun_out/r06i/gpu_tests.log 2>&1; tail -15 gpurun_out/r06i/gpu_tests.log
