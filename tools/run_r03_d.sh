#!/bin/bash
# round-3 GPU session D: hit compaction as its own kernel
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r03d; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_psfpt.py -m gpu -x -q ) > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
B="--no-extra --no-cpu-baseline"
timeout 600 python bench.py --steps 20 --warmup 5 $B > $O/driver_form.json 2> $O/driver_form.err
timeout 600 python bench.py $B > $O/default.json 2> $O/default.err
timeout 600 python bench.py $B --workload testball-room > $O/testball.json 2> $O/testball.err
tail -4 $O/tests.log
for f in $O/*.json; do python - "$f" <<'PY'
import json, sys
try:
    j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
    print(sys.argv[1].split("/")[-1], "value %.1f" % j["value"], "ms/step %.3f" % j["ms_per_step"], "P", j["config"]["passes_in_flight"],
          "trace %.3f shade %.3f" % (j["kernel_ms_per_step"]["trace_primary+mixed"], j["kernel_ms_per_step"]["shade"]))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
