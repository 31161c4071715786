mkdir -p gpurun_out/r06g
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_binding.py -x -q -m gpu -k "device_sgt or slice_synchronised or binding or sddmm_range_major or preprocess" > gpurun_out/r06g/tests.log 2>&1; tail -6 gpurun_out/r06g/tests.log
DIMS=128 KBS=3072 timeout 600 python tools/exp_r06b.py > gpurun_out/r06g/exp_sbm.log 2>&1; grep -v "^\[tcgnn\]" gpurun_out/r06g/exp_sbm.log | tail -4
GEN=uniform DIMS=128 KBS=3072 timeout 600 python tools/exp_r06b.py > gpurun_out/r06g/exp_uniform.log 2>&1; grep -v "^\[tcgnn\]" gpurun_out/r06g/exp_uniform.log | tail -4
TCGNN_VERBOSE=2 timeout 600 python bench.py --no-extra --no-cpu --steps 5 --warmup 2 > gpurun_out/r06g/bench_noextra.log 2> gpurun_out/r06g/bench_noextra.err; grep "plan_create:\|sync walk" gpurun_out/r06g/bench_noextra.err | tail -20; python - <<'P'
import json
l=[x for x in open('gpurun_out/r06g/bench_noextra.log') if x.startswith('{')]
d=json.loads(l[-1]); print({k: d.get('extra',{}).get(k) for k in ('host_sgt_ms','device_sgt_ms','device_sgt_ms_runs','plan_create_ms')}, d.get('summary',{}).get('device_sgt_ms'))
P
