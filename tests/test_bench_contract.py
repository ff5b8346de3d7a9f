"""GPU test (-m gpu): bench.py prints ONE JSON line that carries the driver's contract (metric / value / unit / n_gpus / steps / warmup /
ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data / config.workload) plus the `roofline` and `cpu_baseline` objects, for the
headline path and for the widened rows."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def check_contract(j, steps, warmup):
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in j, k
    assert j["unit"] == "Msample/s" and j["n_gpus"] == 1 and j["steps"] == steps and j["higher_is_better"] is True and j["vs_baseline"] is None
    assert j["dtype"] == "f32" and j["data"] == "synthetic" and j["scaling"] in ("weak", "strong") and "workload" in j["config"] and "model" not in j["config"]
    assert j["value"] > 0 and abs(j["value"] - 1600 * 900 * steps / (j["ms_per_step"] * steps * 1e-3) / 1e6) < 1e-6 * j["value"]
    r = j["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    # "valu": the record says so itself when the matching PMC collection shows little HBM traffic under a busy VALU (the HBM pricing stays beside it)
    assert r["bound"] in ("hbm", "mfma", "valu") and "lane_utilisation" in r and r["unit"] == "GB/s" and r["peak"] == 8000.0
    # frac is a fraction: the contract's ratio capped at 1 (algorithmic bytes served from cache can exceed the HBM roof); the uncapped ratio travels as frac_algorithmic
    assert 0.0 < r["frac"] <= 1.0 and abs(r["frac"] - min(1.0, r["achieved"] / r["peak"])) < 1e-9 and abs(r["frac_algorithmic"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["traffic"] is None or r["traffic"] > 0          # PMC bytes only beside a collection of this exact configuration
    assert "traffic_source" in r


def test_headline_line_with_cpu_baseline():
    j = run_bench("--steps", "4", "--warmup", "2", "--no-extra")        # (the default run adds extra.{standin_r1_r3, testball_room, bpt, psfpt})
    assert "bathroom2-standin-r4" in j["config"]["workload"] and j["config"]["triangles"] > 1500000
    check_contract(j, 4, 2)
    assert "PT" in j["metric"] and j["config"]["passes_in_flight"] == 4 and j["config"]["max_path_length"] == 9
    assert j["scaling"] == "strong" and j["config"]["baseline_config"] == "configs[2]" and j["config"]["resolution"] == [1600, 900]
    b = j["config"]["bvh"]                                   # fpt_rt_bvh_stats: the SAH-optimal collapse fills the 8-wide nodes
    assert b["avg_used_slots"] >= 6.0 and sum(b["slot_hist"]) == b["nodes"] and b["stack_need"] <= 48
    c = j["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "Msample/s"
    assert j["value"] > 20 * c["value"]                      # sanity only: the ratio says nothing about kernel quality


@pytest.mark.parametrize("renderer", ["bpt", "psfpt"])
def test_widened_lines(renderer):
    j = run_bench("--renderer", renderer, "--steps", "4", "--warmup", "2", "--no-cpu-baseline")
    check_contract(j, 4, 4 if renderer in ("bpt", "psfpt") else 2)
    assert renderer.upper() in j["metric"] and "cpu_baseline" not in j
    # configs[4] in kind: the bidirectional tracer's line runs on the water_caustic stand-in (the reference's own .mtl and camera), the PSFPT's on the headline scene
    assert ("water_caustic-standin" if renderer == "bpt" else "bathroom2-standin-r4") in j["config"]["workload"]
