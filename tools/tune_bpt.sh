#!/bin/bash
# usage: tools/tune_bpt.sh "<extra CXXFLAGS>" : rebuild the BPT kernels with flags and run the BPT bench line
cd $GRAFT_REPO_ROOT
rm -f fermat_amd/csrc/fpt_bpt.o
make -s -C fermat_amd/csrc CXXFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -w $1" >/dev/null 2>&1
python bench.py --renderer bpt --steps 64 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('flags=[$1] -> %.1f Msample/s  %.3f ms/step  kernels %s' % (d['value'], d['ms_per_step'], {k: round(v,3) for k,v in d['kernel_ms_per_step'].items()}))"
