#!/usr/bin/env python3
"""One GPU's view of an N-way split (run on the GPU box from the repo root): rank 0's share of the bench frame for N = 1, 2, 4, 8 and the two jobs
DESIGN.md 8 tabulates (the driver's 20 passes; configs[2]'s 256 passes), through bench.py's FPT_BENCH_EMULATE_WORLD switch.  Writes
gpurun_out/profiles_new/r05_emulated_shares.json (copy it into profiles/)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = []
for steps, warm in ((20, 5), (256, 8)):
    base = None
    for world in (1, 2, 4, 8):
        env = dict(os.environ, FPT_BENCH_EMULATE_WORLD=str(world))
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(steps), "--warmup", str(warm), "--no-extra", "--no-cpu-baseline"],
                             env=env, capture_output=True, text=True, cwd=ROOT).stdout
        j = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
        ms = j["ms_per_step"] * j["steps"]
        base = ms if world == 1 else base
        rows.append({"steps": steps, "emulated_world": world, "passes_in_flight": j["config"]["passes_in_flight"], "job_ms_on_rank0": ms,
                     "speedup_if_all_ranks_equal": base / ms, "msample_per_s_per_rank": j["value"], "kernel_ms_per_step": j["kernel_ms_per_step"]})
        print(rows[-1], flush=True)
doc = {"what": "rank 0's share of an N-way scanline split of the bench frame, rendered on ONE MI355X (FPT_BENCH_EMULATE_WORLD=N python bench.py --steps K "
               "--warmup W --no-extra --no-cpu-baseline): compute only, no gather; the speed-up a job would show if every rank took as long as rank 0 "
               "(ranks' shares differ by +-2 %, DESIGN 8)", "rows": rows}
os.makedirs(os.path.join(ROOT, "gpurun_out", "profiles_new"), exist_ok=True)
json.dump(doc, open(os.path.join(ROOT, "gpurun_out", "profiles_new", "r05_emulated_shares.json"), "w"), indent=1)
