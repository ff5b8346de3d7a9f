#!/bin/bash
cd /root/repo
O=gpurun_out/r03x; mkdir -p $O
B="--no-extra --no-cpu-baseline"
for rep in 1 2; do
for v in $VARIANTS; do
  if [ $v = base ]; then L=""; else L="FPT_LIB_PATH=$PWD/fermat_amd/libfermat_pt_hip_$v.so"; fi
  env $L timeout 600 python bench.py --steps 20 --warmup 5 $B > $O/d_${v}_$rep.json 2> $O/d_${v}_$rep.err
  env $L timeout 600 python bench.py $B > $O/f_${v}_$rep.json 2> $O/f_${v}_$rep.err
done; done
for f in $O/*.json; do python - "$f" <<'PY'
import json, sys
try:
    j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
    print(sys.argv[1].split("/")[-1], "value %.1f" % j["value"], "trace %.3f shade %.3f" % (j["kernel_ms_per_step"]["trace_primary+mixed"], j["kernel_ms_per_step"]["shade"]))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
