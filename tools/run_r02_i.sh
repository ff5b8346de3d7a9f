#!/bin/bash
mkdir -p gpurun_out/r02i
F=$PWD/fermat_amd
( FPT_LIB_PATH=$F/libfermat_pt_hip_one.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_edge_cases.py -q -m gpu -x -k "not native_library and not cli" ) > gpurun_out/r02i/tests_one.log 2>&1
tail -3 gpurun_out/r02i/tests_one.log
for v in all one onexcd xcd; do
  FPT_LIB_PATH=$F/libfermat_pt_hip_$v.so python tools/trace_bench.py --bounces 0,1,3 > gpurun_out/r02i/tb_$v.json 2> gpurun_out/r02i/tb_$v.err
  python -c "
import json
j=json.loads([l for l in open('gpurun_out/r02i/tb_$v.json') if l.startswith('{')][-1])
print('$v', {b:(round(x['closest']['ms'],3), round(x['any']['ms'],3), round(x['closest']['nodes_per_ray'],2), round(x['closest']['tris_per_ray'],2)) for b,x in j['bounces'].items()})
" || tail -3 gpurun_out/r02i/tb_$v.err
FPT_LIB_PATH=$F/libfermat_pt_hip_$v.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02i/bench_${v}_driver.json 2> gpurun_out/r02i/bench_${v}_driver.err
FPT_LIB_PATH=$F/libfermat_pt_hip_$v.so python bench.py --no-cpu-baseline > gpurun_out/r02i/bench_${v}_default.json 2> gpurun_out/r02i/bench_${v}_default.err
for f in bench_${v}_driver bench_${v}_default; do python -c "
import json
j=json.loads([l for l in open('gpurun_out/r02i/$f.json') if l.startswith('{')][-1])
print('$f', round(j['value'],1), j['kernel_ms_per_step'], round(j['roofline']['nodes_per_ray'],2), round(j['roofline']['tris_per_ray'],2))
"; done; done
