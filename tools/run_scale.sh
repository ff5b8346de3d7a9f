#!/bin/bash
# tools/run_scale.sh [out_dir]   -- the 1 / 2 / 4 / 8-GPU curve in one command, for the day an 8-GPU MI355X node is at hand (VERDICT r4 task 9; north_star: Msample/s
# at 1, 2, 4 and 8 GPUs, >= 6x tile-parallel speed-up at 8).  Run from the repo root on the node.  Per N it runs bench.py exactly as the driver does -- one rank per
# GPU under torch.distributed.run, RCCL over xGMI -- for
#   (a) the driver's job           : --steps 20 --warmup 5          (strong scaling: 20 passes of the 1600x900 frame whatever N)
#   (b) configs[2]'s own job       : --steps 256 --warmup 64        (256 spp)
#   (c) configs[3]                 : --config c4 --steps 64 --warmup 16   (3840x2160; the 8-GPU tile job)
# writes one JSON line per (job, N) to out_dir/scale_<job>_n<N>.json, asserts that RCCL's communicator really spans N ranks (config.rccl_ranks, read from
# ncclCommCount inside the library) and prints the speed-ups against the N = 1 line of the same session.
set -e
OUT=${1:-gpurun_out/scale}; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
PORT=29511
run() {          # job-name N args...
	local job=$1 n=$2; shift 2
	local f=$OUT/scale_${job}_n${n}.json
	if [ $n -eq 1 ]; then python bench.py --gpus 1 --no-extra --no-cpu-baseline "$@" > $f 2> ${f%.json}.err
	else python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $PORT bench.py --gpus $n --no-extra --no-cpu-baseline "$@" > $f 2> ${f%.json}.err; fi
	PORT=$((PORT + 1))
	python - "$f" "$n" << 'PY'
import json, sys
line = [l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")]
assert line, "no JSON line in " + sys.argv[1]
j = json.loads(line[-1]); n = int(sys.argv[2])
assert j["n_gpus"] == n, (j["n_gpus"], n)
if n > 1:
    assert j["config"].get("rccl_ranks") == n, "RCCL's communicator spans %r ranks, not %d" % (j["config"].get("rccl_ranks"), n)
    assert "fpt_gather_framebuffer" in j["config"]["gather"], j["config"]["gather"]
print("%-40s N=%d  %9.1f %s  (%.3f ms/step, scaling %s%s)" % (sys.argv[1].split("/")[-1], n, j["value"], j["unit"], j["ms_per_step"], j["scaling"],
      ", weak %.1f" % j["value_weak"] if "value_weak" in j else ""))
PY
}
NG=$(python -c "import torch; print(torch.cuda.device_count())")
for n in 1 2 4 8; do
	[ $n -le $NG ] || { echo "only $NG GPUs visible: stopping before N=$n"; break; }
	run driver20 $n --steps 20 --warmup 5
	run spp256 $n --steps 256 --warmup 64
	run c4 $n --config c4 --steps 64 --warmup 16
done
python - $OUT << 'PY'
import json, os, sys
d = sys.argv[1]
for job in ("driver20", "spp256", "c4"):
    v = {}
    for n in (1, 2, 4, 8):
        f = os.path.join(d, "scale_%s_n%d.json" % (job, n))
        if os.path.exists(f):
            line = [l for l in open(f).read().splitlines() if l.startswith("{")]
            if line: v[n] = json.loads(line[-1])["value"]
    if 1 in v:
        print(job, " ".join("N=%d %.1f Msample/s (%.2fx)" % (n, x, x / v[1]) for n, x in sorted(v.items())))
PY
