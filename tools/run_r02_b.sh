#!/bin/bash
# round-2 GPU call B: the 8-wide BVH variant: parity subset, then kernel-level and bench-level A/B against the BVH2 library
mkdir -p gpurun_out/r02b
W8=$PWD/fermat_amd/libfermat_pt_hip_w8.so
( FPT_LIB_PATH=$W8 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_edge_cases.py -x -q ) > gpurun_out/r02b/tests_w8.log 2>&1
tail -5 gpurun_out/r02b/tests_w8.log
for wl in standin testball-room; do
  python tools/trace_bench.py --workload $wl > gpurun_out/r02b/tb_bvh2_$wl.json 2> gpurun_out/r02b/tb_bvh2_$wl.err
  FPT_LIB_PATH=$W8 python tools/trace_bench.py --workload $wl > gpurun_out/r02b/tb_w8_$wl.json 2> gpurun_out/r02b/tb_w8_$wl.err
  tail -1 gpurun_out/r02b/tb_bvh2_$wl.json; tail -1 gpurun_out/r02b/tb_w8_$wl.json; tail -2 gpurun_out/r02b/tb_w8_$wl.err
done
FPT_LIB_PATH=$W8 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02b/bench_w8_driver.json 2> gpurun_out/r02b/bench_w8_driver.err
FPT_LIB_PATH=$W8 python bench.py --no-cpu-baseline > gpurun_out/r02b/bench_w8_default.json 2> gpurun_out/r02b/bench_w8_default.err
FPT_LIB_PATH=$W8 python bench.py --workload testball-room --no-cpu-baseline > gpurun_out/r02b/bench_w8_testball.json 2> gpurun_out/r02b/bench_w8_testball.err
for f in bench_w8_driver bench_w8_default bench_w8_testball; do echo == $f; python -c "
import json
j=json.loads([l for l in open('gpurun_out/r02b/$f.json') if l.startswith('{')][-1])
print(round(j['value'],1), j['kernel_ms_per_step'], j['roofline']['nodes_per_ray'], j['roofline']['tris_per_ray'])
"; tail -2 gpurun_out/r02b/$f.err; done
