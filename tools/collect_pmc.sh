#!/bin/bash
# usage: tools/collect_pmc.sh <tag> <bench.py args...>   (run on the GPU box, from the repo root)
# three separate rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE / VALU counters) + one plain run of the same command line, summarised
# into profiles/<tag>.json (tools/summarize_pmc.py); copy gpurun_out/profiles_new/* into profiles/ afterwards
TAG=$1; shift
R=$PWD; export TMPDIR=/tmp
D=$R/gpurun_out/pmc/$TAG; rm -rf $D; mkdir -p $D
python bench.py "$@" --no-cpu-baseline --no-extra > $D/line.json 2> $D/line.err
cd /tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $D/fetch -o p -- python $R/bench.py "$@" --no-cpu-baseline --no-extra > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $D/write -o p -- python $R/bench.py "$@" --no-cpu-baseline --no-extra > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $D/valu -o p -- python $R/bench.py "$@" --no-cpu-baseline --no-extra > /dev/null 2>&1
cd $R
python tools/summarize_pmc.py $D $D/line.json $TAG
mkdir -p gpurun_out/profiles_new; cp profiles/$TAG.json gpurun_out/profiles_new/
find $D -name "*.db" -size +20M -delete      # keep the merged gpurun_out under its size cap
