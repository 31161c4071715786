mkdir -p gpurun_out/r06o
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "slice_synchronised" > gpurun_out/r06o/tests.log 2>&1; tail -3 gpurun_out/r06o/tests.log
DIMS=128 KBS=3072 timeout 600 python tools/exp_r06b.py > gpurun_out/r06o/exp_sbm.log 2>&1; grep -v "^\[tcgnn\]" gpurun_out/r06o/exp_sbm.log | tail -3
export TCGNN_PROFILE_SHAPE=ogbn-products TCGNN_PROFILE_GEN=sbm; timeout 700 tools/pmc_calls.sh r06_sync_spmm_s512 spmm 128 0 > /dev/null 2>&1; grep -A14 "spmm_sync" gpurun_out/pmc_r06_sync_spmm_s512/summary.txt | tail -4
