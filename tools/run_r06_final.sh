#!/bin/bash
# round-6 GPU collection: the whole -m gpu suite, smoke(), the bench lines that go to profiles/, rocprofv3 --stats and PMC collections (run on the GPU box from the repo root)
cd "$(dirname "$0")/.." || exit 1
ulimit -c 0
O=gpurun_out/r06z; N=gpurun_out/profiles_new; rm -rf $N; mkdir -p $O $N
R=$PWD; export TMPDIR=/tmp
if [ -z "$SKIP_TESTS" ]; then
( time timeout 1700 python -m pytest tests -m gpu -q ) > $O/tests.log 2>&1
grep -E "passed|failed" $O/tests.log | tail -2 | cut -c1-200
fi
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
# PMC collections first, one per configuration a bench line is printed for: the lines below then carry roofline.traffic / bound / lane_utilisation of THIS build (source hash)
bash tools/collect_pmc.sh r06_pmc_bathroom2_b20 --steps 20 --warmup 5 > $O/pmc_b20.txt 2>&1
bash tools/collect_pmc.sh r06_pmc_bathroom2_b64 > $O/pmc_b64.txt 2>&1
bash tools/collect_pmc.sh r06_pmc_bpt_water_caustic_b32 --renderer bpt --steps 32 --warmup 32 > $O/pmc_bpt.txt 2>&1
bash tools/collect_pmc.sh r06_pmc_psfpt_b32 --renderer psfpt --steps 32 --warmup 32 > $O/pmc_psf.txt 2>&1
cp $N/r06_pmc_*.json profiles/ 2>/dev/null
tail -14 $O/pmc_b20.txt | cut -c1-200
python bench.py --steps 20 --warmup 5 > $N/r06_bench_line_driver_form.json 2> $O/b1.err
python bench.py > $N/r06_bench_line.json 2> $O/b2.err
python bench.py --batch 1 --steps 64 --warmup 8 --no-cpu-baseline --no-extra > $N/r06_bench_line_sequential.json 2> $O/b4.err
python bench.py --steps 20 --warmup 5 --api render --no-cpu-baseline --no-extra > $N/r06_bench_line_api_render.json 2> $O/b5.err
python bench.py --config c4 --batch 32 --steps 32 --warmup 32 --no-cpu-baseline --no-extra > $N/r06_bench_line_c4_one_gpu_32_in_flight.json 2> $O/b6.err
python bench.py --workload standin --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $N/r06_bench_line_standin_r1_r3_driver_form.json 2> $O/b7.err
python bench.py --workload standin --no-cpu-baseline --no-extra > $N/r06_bench_line_standin_r1_r3.json 2> $O/b7b.err
python bench.py --renderer bpt > $N/r06_bench_line_bpt_water_caustic.json 2> $O/w1.err
python bench.py --renderer bpt --sc 0 --no-cpu-baseline > $N/r06_bench_line_bpt_water_caustic_sc0.json 2> $O/w1b.err
python bench.py --renderer bpt --workload bathroom2 --no-cpu-baseline > $N/r06_bench_line_bpt_bathroom2.json 2> $O/w1c.err
python bench.py --renderer psfpt --no-cpu-baseline > $N/r06_bench_line_psfpt.json 2> $O/w2.err
FPT_BENCH_FORCE_DEVICE=0 FPT_BENCH_BACKEND=gloo timeout 400 python bench.py --gpus 2 --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_n2_gloo.json 2> $O/b8.err
for f in $N/r06_bench_line*.json $O/bench_n2_gloo.json; do python -c "
import json,sys
j=json.loads([l for l in open('$f') if l.startswith('{')][-1])
r=j['roofline']
print('$f'.split('/')[-1], round(j['value'],1), 'ms/step', round(j['ms_per_step'],4), 'P', j['config']['passes_in_flight'], {k:(round(v,4) if isinstance(v,float) else v) for k,v in j['kernel_ms_per_step'].items() if 'busy' not in k}, 'frac', round(r['frac'],3), r['bound'], 'counter', r.get('counter_frac'), r.get('counter_gbs_profiled'), 'lanes', r.get('lane_utilisation'), j.get('value_weak'), {k: round(v['value'],1) for k,v in j.get('extra',{}).items()})
" || echo "FAILED $f"; done
# rocprofv3 --stats of the bench command lines
for cfg in "default:" "driver:--steps 20 --warmup 5" "bpt:--renderer bpt --steps 32 --warmup 32"; do
  n=${cfg%%:*}; a=${cfg#*:}
  rm -rf $R/$O/stats_$n; cd /tmp
  rocprofv3 --kernel-trace --stats -d $R/$O/stats_$n -o s -- python $R/bench.py $a --no-cpu-baseline --no-extra > $R/$O/stats_$n.log 2>&1
  cd $R
  python tools/summarize_stats.py $O/stats_$n r06_kernel_stats_$n "python bench.py $a --no-cpu-baseline --no-extra" > $O/stats_$n.txt 2>&1
  cp profiles/r06_kernel_stats_$n.md $N/
  rm -rf $O/stats_$n
done
rm -rf $R/gpurun_out/pmc
# the host builder on the box's host threads (profiles/r06_build_time.txt) and BASELINE configs[4] at its full 4096 passes (profiles/r06_config5_4096spp.*)
{ FPT_BVH_TIMERS=1 python tools/time_build.py 2>&1 | grep -v "unable to find texture" | tail -9; python tools/time_refit.py 2>/dev/null | tail -1; } > $N/r06_build_time_raw.txt
# the acceleration structure on the device: refit, fast build (and the traversal cost of its tree), the fixed part of a traversal launch
timeout 300 python tools/time_device_refit.py 2>&1 | grep -v "amdgpu.ids\|unable to find" | tail -2 > $N/r06_device_refit_raw.txt
FPT_BVH_TIMERS=1 timeout 300 python tools/time_device_build.py 2>&1 | grep -E "create_geometry|build_acceleration_device" | cut -c1-330 > $N/r06_device_build_raw.txt
FPT_BVH_BUILD=fast timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $N/r06_bench_line_driver_form_device_built_tree.json 2> $O/b9.err
timeout 300 python tools/diag_tail.py 2>&1 | grep -v "amdgpu.ids\|unable to find" > $N/r06_launch_tail.txt
FPT_BVH_TIMERS=1 timeout 300 python tools/time_update_model.py 2>&1 | grep -E "fpt_rt_|build_emitter|lights_init|update_model|context created" | tail -12 > $N/r06_update_model_time_raw.txt
if [ -n "$WITH_CONFIG5" ]; then timeout 600 python tools/run_config5_full.py 4096 32 > $O/config5.log 2>&1; cp gpurun_out/r06_config5_4096spp.* $N/ 2>/dev/null; fi
ls $N
