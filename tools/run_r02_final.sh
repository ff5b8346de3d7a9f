#!/bin/bash
# round-2 final GPU call: the whole -m gpu suite, smoke(), the bench lines that go to profiles/, rocprofv3 stats and PMC collections
mkdir -p gpurun_out/r02z gpurun_out/profiles_new
( time timeout 1500 python -m pytest tests -m gpu -q -s ) > gpurun_out/r02z/tests.log 2>&1
tail -6 gpurun_out/r02z/tests.log | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r02z/smoke.log 2>&1; tail -1 gpurun_out/r02z/smoke.log
python bench.py --steps 20 --warmup 5 > gpurun_out/profiles_new/r02_bench_line_driver_form.json 2> gpurun_out/r02z/b1.err
python bench.py > gpurun_out/profiles_new/r02_bench_line.json 2> gpurun_out/r02z/b2.err
python bench.py --workload testball-room --no-cpu-baseline > gpurun_out/profiles_new/r02_bench_line_testball_room.json 2> gpurun_out/r02z/b3.err
python bench.py --detail 4 --steps 32 --warmup 32 --no-cpu-baseline > gpurun_out/profiles_new/r02_bench_line_detail4.json 2> gpurun_out/r02z/b4.err
python bench.py --batch 1 --steps 64 --warmup 8 --no-cpu-baseline > gpurun_out/profiles_new/r02_bench_line_sequential.json 2> gpurun_out/r02z/b6c.err
FPT_BENCH_FORCE_DEVICE=0 FPT_BENCH_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/r02z/bench_n2_gloo.json 2> gpurun_out/r02z/b7.err
for f in gpurun_out/profiles_new/r02_bench_line*.json gpurun_out/r02z/bench_n2_gloo.json; do python -c "
import json,sys
j=json.loads([l for l in open('$f') if l.startswith('{')][-1])
r=j['roofline']
print('$f'.split('/')[-1], round(j['value'],1), 'ms/step', round(j['ms_per_step'],4), {k:(round(v,4) if isinstance(v,float) else v) for k,v in j['kernel_ms_per_step'].items()}, 'frac', round(r['frac'],3), 'traffic', r.get('traffic'), 'counter_frac', r.get('counter_frac'))
" || echo "FAILED $f"; done
bash tools/run_r02_h.sh > gpurun_out/r02z/collect.log 2>&1
tail -3 gpurun_out/r02z/collect.log
bash tools/run_r02_r.sh > gpurun_out/r02z/widened.log 2>&1; grep -E "^(bpt|psfpt)" gpurun_out/r02z/widened.log | cut -c1-260
ls gpurun_out/profiles_new
