mkdir -p gpurun_out/r06m
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "slice_synchronised" > gpurun_out/r06m/tests.log 2>&1; tail -3 gpurun_out/r06m/tests.log
DIMS=128 KBS=3072 timeout 600 python tools/exp_r06b.py > gpurun_out/r06m/exp_sbm.log 2>&1; grep -v "^\[tcgnn\]" gpurun_out/r06m/exp_sbm.log | tail -4
