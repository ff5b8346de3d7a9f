#!/bin/bash
mkdir -p gpurun_out/r02k
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_edge_cases.py tests/test_gpu_full_size.py -q -m gpu -x -k "not cli and not config5 and not config2" ) > gpurun_out/r02k/tests.log 2>&1
tail -3 gpurun_out/r02k/tests.log
for cfg in "off:0:1e-5" "a5:1:1e-5" "a4:1:1e-4" "a3:1:1e-3"; do
n=${cfg%%:*}; r=${cfg#*:}; sp=${r%%:*}; al=${r#*:}
export FPT_BVH_SPATIAL_SPLITS=$sp FPT_BVH_SPATIAL_ALPHA=$al
python bench.py --no-cpu-baseline > gpurun_out/r02k/bench_${n}_default.json 2> gpurun_out/r02k/bench_${n}_default.err
python bench.py --workload testball-room --no-cpu-baseline > gpurun_out/r02k/bench_${n}_testball.json 2> gpurun_out/r02k/bench_${n}_testball.err
for f in bench_${n}_default bench_${n}_testball; do python -c "
import json
j=json.loads([l for l in open('gpurun_out/r02k/$f.json') if l.startswith('{')][-1])
print('$f', round(j['value'],1), {k:(round(v,4) if isinstance(v,float) else v) for k,v in j['kernel_ms_per_step'].items() if 'trace_p' in k or k=='shade'}, round(j['roofline']['nodes_per_ray'],2), round(j['roofline']['tris_per_ray'],2), 'bvh MB', j['roofline']['bvh_bytes']/1e6)
" || tail -3 gpurun_out/r02k/$f.err; done; done
