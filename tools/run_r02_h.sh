#!/bin/bash
# round-2 GPU call H: the committed profiles: rocprofv3 --stats of the bench command lines + PMC collections per configuration
mkdir -p gpurun_out/r02h gpurun_out/profiles_new
R=$PWD; export TMPDIR=/tmp
for cfg in "default:" "driver:--steps 20 --warmup 5"; do
  n=${cfg%%:*}; a=${cfg#*:}
  rm -rf $R/gpurun_out/r02h/stats_$n; cd /tmp
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r02h/stats_$n -o s -- python $R/bench.py $a --no-cpu-baseline > $R/gpurun_out/r02h/stats_$n.log 2>&1
  cd $R
  python tools/summarize_stats.py gpurun_out/r02h/stats_$n r02_kernel_stats_$n "python bench.py $a --no-cpu-baseline" > gpurun_out/r02h/stats_$n.txt 2>&1
  cp profiles/r02_kernel_stats_$n.md gpurun_out/profiles_new/
  find gpurun_out/r02h/stats_$n -name "*.db" -size +20M -delete
done
bash tools/collect_pmc.sh r02_pmc_standin_b20 --steps 20 --warmup 5 > gpurun_out/r02h/pmc_b20.txt 2>&1
bash tools/collect_pmc.sh r02_pmc_standin_b64 --steps 64 --warmup 0 > gpurun_out/r02h/pmc_b64.txt 2>&1
bash tools/collect_pmc.sh r02_pmc_standin_detail4_b32 --detail 4 --steps 32 --warmup 0 > gpurun_out/r02h/pmc_d4.txt 2>&1
bash tools/collect_pmc.sh r02_pmc_testball_b64 --workload testball-room --steps 64 --warmup 0 > gpurun_out/r02h/pmc_tb.txt 2>&1
tail -25 gpurun_out/r02h/pmc_b20.txt gpurun_out/r02h/pmc_b64.txt gpurun_out/r02h/pmc_d4.txt gpurun_out/r02h/pmc_tb.txt | cut -c1-220
head -30 gpurun_out/r02h/stats_default.txt | cut -c1-200
