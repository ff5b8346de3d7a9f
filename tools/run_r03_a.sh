#!/bin/bash
# round-3 GPU session A: parity of the new builder + lanes, first numbers (driver form, default form, the reference's one-pass mode with lanes)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r03a; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -x -q ) > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
B="--no-extra --no-cpu-baseline"
timeout 600 python bench.py --steps 20 --warmup 5 $B > $O/driver_form.json 2> $O/driver_form.err
for L in 2 4; do timeout 600 python bench.py --steps 20 --warmup 5 $B --lanes $L > $O/driver_form_lanes$L.json 2> $O/driver_form_lanes$L.err; done
for L in 1 2 4 8; do timeout 600 python bench.py --batch 1 --steps 64 --warmup 8 $B --lanes $L > $O/seq_lanes$L.json 2> $O/seq_lanes$L.err; done
timeout 900 python bench.py > $O/default.json 2> $O/default.err
tail -3 $O/tests.log
for f in $O/*.json; do python - "$f" <<'PY'
import json, sys
try:
    j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
    print(sys.argv[1].split("/")[-1], "value %.1f" % j["value"], "ms/step %.3f" % j["ms_per_step"], "lanes", j["config"]["render_lanes"], "P", j["config"]["passes_in_flight"],
          "trace %.3f shade %.3f busy %.3f/%.3f" % (j["kernel_ms_per_step"]["trace_primary+mixed"], j["kernel_ms_per_step"]["shade"], j["kernel_ms_per_step"]["trace_busy"], j["kernel_ms_per_step"]["shade_busy"]),
          "nodes/ray %.2f tris/ray %.2f" % (j["roofline"]["nodes_per_ray"], j["roofline"]["tris_per_ray"]), "frac %.3f" % j["roofline"]["frac"],
          ("extra %.1f" % j["extra"]["testball_room"]["value"]) if "extra" in j else "")
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
