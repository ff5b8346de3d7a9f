#!/bin/bash
# round-2 GPU call G: the adopted 8-wide kernel as the product library: whole -m gpu suite, bench lines, shade-fission experiment, rocprofv3 stats + PMC
mkdir -p gpurun_out/r02g
( time timeout 1200 python -m pytest tests -m gpu -q -s ) > gpurun_out/r02g/tests.log 2>&1
tail -8 gpurun_out/r02g/tests.log | cut -c1-300
python bench.py --steps 20 --warmup 5 > gpurun_out/r02g/bench_driver.json 2> gpurun_out/r02g/bench_driver.err
python bench.py --no-cpu-baseline > gpurun_out/r02g/bench_default.json 2> gpurun_out/r02g/bench_default.err
FPT_SHADE_SPLIT=1 python bench.py --no-cpu-baseline > gpurun_out/r02g/bench_split.json 2> gpurun_out/r02g/bench_split.err
python bench.py --workload testball-room --no-cpu-baseline > gpurun_out/r02g/bench_testball.json 2> gpurun_out/r02g/bench_testball.err
python bench.py --detail 4 --steps 32 --warmup 32 --no-cpu-baseline > gpurun_out/r02g/bench_detail4.json 2> gpurun_out/r02g/bench_detail4.err
for f in bench_driver bench_default bench_split bench_testball bench_detail4; do python -c "
import json
j=json.loads([l for l in open('gpurun_out/r02g/$f.json') if l.startswith('{')][-1])
r=j['roofline']
print('$f', round(j['value'],1), j['kernel_ms_per_step'], 'frac', round(r['frac'],3), 'nodes', round(r['nodes_per_ray'],2), 'tris', round(r['tris_per_ray'],2), 'bvh MB', r['bvh_bytes']/1e6, 'tri', j['config']['triangles'])
" || tail -3 gpurun_out/r02g/$f.err; done
