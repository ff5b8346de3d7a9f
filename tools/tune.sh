#!/bin/bash
# usage: tools/tune.sh "<extra CXXFLAGS>" <blocks_per_cu> : rebuild the trace kernel with flags and run a short bench
cd $GRAFT_REPO_ROOT
rm -f fermat_amd/csrc/fpt_trace.o fermat_amd/csrc/fpt_pt.o fermat_amd/csrc/fpt_api.o
make -s -C fermat_amd/csrc CXXFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -w $1" >/dev/null 2>&1
FPT_TRACE_BLOCKS_PER_CU=$2 python bench.py --steps ${STEPS:-16} --warmup ${WARM:-2} --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('flags=[$1] bpc=$2 -> %.1f Msample/s  %.3f ms/step  kernels %s  roofline %.3f' % (d['value'], d['ms_per_step'], {k: round(v,3) for k,v in d['kernel_ms_per_step'].items()}, d['roofline']['frac']))"
