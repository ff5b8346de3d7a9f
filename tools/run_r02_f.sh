#!/bin/bash
mkdir -p gpurun_out/r02f
F=$PWD/fermat_amd
( timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "cli_batch" ) > gpurun_out/r02f/tests_cli.log 2>&1; tail -3 gpurun_out/r02f/tests_cli.log
for v in w8g w8ng; do
  FPT_LIB_PATH=$F/libfermat_pt_hip_$v.so python tools/trace_bench.py --bounces 1,3 > gpurun_out/r02f/tb_$v.json 2> gpurun_out/r02f/tb_$v.err
  python -c "
import json
j=json.loads([l for l in open('gpurun_out/r02f/tb_$v.json') if l.startswith('{')][-1])
print('$v', {b:(round(x['closest']['ms'],3), round(x['any']['ms'],3)) for b,x in j['bounces'].items()})
" || tail -3 gpurun_out/r02f/tb_$v.err
FPT_LIB_PATH=$F/libfermat_pt_hip_$v.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02f/bench_${v}_driver.json 2> gpurun_out/r02f/bench_${v}_driver.err
FPT_LIB_PATH=$F/libfermat_pt_hip_$v.so python bench.py --no-cpu-baseline > gpurun_out/r02f/bench_${v}_default.json 2> gpurun_out/r02f/bench_${v}_default.err
for f in bench_${v}_driver bench_${v}_default; do python -c "
import json
j=json.loads([l for l in open('gpurun_out/r02f/$f.json') if l.startswith('{')][-1])
print('$f', round(j['value'],1), j['kernel_ms_per_step'])
"; done; done
