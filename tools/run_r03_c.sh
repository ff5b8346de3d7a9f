#!/bin/bash
# round-3 GPU session C: shade-kernel tile compaction, deferred render(), cleanup: parity + numbers
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r03c; mkdir -p $O
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
B="--no-extra --no-cpu-baseline"
timeout 600 python bench.py --steps 20 --warmup 5 $B > $O/driver_form.json 2> $O/driver_form.err
timeout 600 python bench.py --steps 20 --warmup 5 $B --api render > $O/driver_form_api_render.json 2> $O/driver_form_api_render.err
timeout 600 python bench.py --steps 20 --warmup 5 $B --lanes 2 > $O/driver_form_lanes2.json 2> $O/driver_form_lanes2.err
timeout 600 python bench.py $B > $O/default.json 2> $O/default.err
timeout 600 python bench.py $B --batch 1 --steps 64 --warmup 8 --lanes 1 > $O/seq.json 2> $O/seq.err
timeout 600 python bench.py $B --workload testball-room > $O/testball.json 2> $O/testball.err
tail -4 $O/tests.log
for f in $O/*.json; do python - "$f" <<'PY'
import json, sys
try:
    j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
    print(sys.argv[1].split("/")[-1], "value %.1f" % j["value"], "ms/step %.3f" % j["ms_per_step"], "lanes", j["config"]["render_lanes"], "P", j["config"]["passes_in_flight"],
          "trace %.3f shade %.3f busy %.3f/%.3f" % (j["kernel_ms_per_step"]["trace_primary+mixed"], j["kernel_ms_per_step"]["shade"], j["kernel_ms_per_step"]["trace_busy"], j["kernel_ms_per_step"]["shade_busy"]),
          "nodes/ray %.2f tris/ray %.2f" % (j["roofline"]["nodes_per_ray"], j["roofline"]["tris_per_ray"]), "frac %.3f" % j["roofline"]["frac"], j["config"].get("api"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
