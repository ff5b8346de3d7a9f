#!/bin/bash
mkdir -p gpurun_out/r02d
F=$PWD/fermat_amd
python tools/debug_batch_cli.py > gpurun_out/r02d/debug_batch.txt 2>&1; cat gpurun_out/r02d/debug_batch.txt | tail -8
( FPT_LIB_PATH=$F/libfermat_pt_hip_w8v2.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_edge_cases.py tests/test_bpt.py tests/test_psfpt.py tests/test_multi_gpu.py -q -m gpu -k "not native_library and not cli_batch" ) > gpurun_out/r02d/tests_w8v2.log 2>&1
tail -6 gpurun_out/r02d/tests_w8v2.log
for v in w8v2 w8v2np w8v2p3 w8v2p10 w8v2r8 w8v2r32 w8v2o8 w8v2o6; do
  FPT_LIB_PATH=$F/libfermat_pt_hip_$v.so python tools/trace_bench.py --bounces 0,1,3 > gpurun_out/r02d/tb_$v.json 2> gpurun_out/r02d/tb_$v.err
  python -c "
import json
j=json.loads([l for l in open('gpurun_out/r02d/tb_$v.json') if l.startswith('{')][-1])
print('$v', {b:(round(x['closest']['ms'],3), round(x['any']['ms'],3), round(x['closest']['nodes_per_ray'],2), round(x['closest']['tris_per_ray'],2)) for b,x in j['bounces'].items()})
" || tail -3 gpurun_out/r02d/tb_$v.err
done
FPT_LIB_PATH=$F/libfermat_pt_hip_w8v2.so python tools/trace_bench.py --workload testball-room --bounces 0,1 > gpurun_out/r02d/tb_w8v2_testball.json 2>/dev/null; tail -c 900 gpurun_out/r02d/tb_w8v2_testball.json
for v in w8v2 w8v2np; do
FPT_LIB_PATH=$F/libfermat_pt_hip_$v.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02d/bench_${v}_driver.json 2> gpurun_out/r02d/bench_${v}_driver.err
FPT_LIB_PATH=$F/libfermat_pt_hip_$v.so python bench.py --no-cpu-baseline > gpurun_out/r02d/bench_${v}_default.json 2> gpurun_out/r02d/bench_${v}_default.err
for f in bench_${v}_driver bench_${v}_default; do python -c "
import json
j=json.loads([l for l in open('gpurun_out/r02d/$f.json') if l.startswith('{')][-1])
print('$f', round(j['value'],1), j['kernel_ms_per_step'], round(j['roofline']['nodes_per_ray'],2), round(j['roofline']['tris_per_ray'],2))
"; done; done
