#!/bin/bash
mkdir -p gpurun_out/r02c
F=$PWD/fermat_amd
( FPT_LIB_PATH=$F/libfermat_pt_hip_w8.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_edge_cases.py tests/test_bpt.py tests/test_psfpt.py -x -q -m gpu -k "not native_library" ) > gpurun_out/r02c/tests_w8.log 2>&1
tail -4 gpurun_out/r02c/tests_w8.log
for v in w8pad w8o6 w8o4 w8r16 w8r48; do
  FPT_LIB_PATH=$F/libfermat_pt_hip_$v.so python tools/trace_bench.py --bounces 0,1 > gpurun_out/r02c/tb_$v.json 2> gpurun_out/r02c/tb_$v.err
  python -c "
import json
j=json.loads([l for l in open('gpurun_out/r02c/tb_$v.json') if l.startswith('{')][-1])
print('$v', {b:(round(x['closest']['ms'],3), round(x['any']['ms'],3)) for b,x in j['bounces'].items()})
"
done
bash tools/pmc_trace.sh bvh2 "" > gpurun_out/r02c/pmc_bvh2.txt 2>&1
bash tools/pmc_trace.sh w8 $F/libfermat_pt_hip_w8.so > gpurun_out/r02c/pmc_w8.txt 2>&1
grep -E "^(bvh2|w8) " gpurun_out/r02c/pmc_bvh2.txt gpurun_out/r02c/pmc_w8.txt | sed 's/gpurun_out.r02c.pmc_[a-z0-9]*.txt://'
