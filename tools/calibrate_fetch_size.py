#!/usr/bin/env python3
"""Calibrates rocprofv3's FETCH_SIZE on the traversal kernel's access pattern (VERDICT r4 task 2b; MI355X_MICROARCH.md, HBM: "calibrate on a known byte count in
your own access pattern").  Run on the GPU box from the repo root:

    python tools/calibrate_fetch_size.py            # -> gpurun_out/profiles_new/r05_fetch_size_calibration.json (copy into profiles/)

tools/micro/gather_rate calib: a pointer chase over 1 GB of records (the L2s hold 0.4 % of it, so every record fetch goes to the fabric), every lane of a wave at a
different record -- 16 / 64 / 80 / 128 bytes read of a 128-byte-aligned record, and 80 bytes of packed 80-byte records (the node step: half of them straddle two lines).
The same binary runs once plain (timings) and once under `rocprofv3 --kernel-trace --pmc FETCH_SIZE`; counted bytes per record fetch = FETCH_SIZE x 1024 / fetches."""
import json, os, re, sqlite3, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
exe = os.path.join(ROOT, "tools", "micro", "gather_rate")
out_dir = os.path.join(ROOT, "gpurun_out", "pmc", "fetch_calib")
subprocess.run(["rm", "-rf", out_dir]); os.makedirs(out_dir, exist_ok=True)
plain = subprocess.run([exe, "calib"], capture_output=True, text=True).stdout
env = dict(os.environ, TMPDIR="/tmp")
subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", "FETCH_SIZE", "-d", out_dir, "-o", "p", "--", exe, "calib"], cwd="/tmp", env=env, capture_output=True, text=True)
dbs = [os.path.join(r, x) for r, _, f in os.walk(out_dir) for x in f if x.endswith(".db")]
rows = {}
if dbs:
    cur = sqlite3.connect(dbs[0]).cursor()
    for kn, v, du, dispatch in cur.execute("select kernel_name, value, duration, dispatch_id from counters_collection where counter_name = 'FETCH_SIZE' order by dispatch_id"):
        m = re.search(r"gather<(\d+),\s*(\d+)>", kn)
        if m:
            rows.setdefault("gather<%s,%s>" % m.groups(), []).append((v, du))
res = {"what": "FETCH_SIZE against known record fetches, every lane at a different record of a 1 GB array (tools/micro/gather_rate calib)", "kernels": {}}
for ln in plain.splitlines():
    m = re.match(r"CALIB kernel=(\S+) record_stride=(\d+) bytes_read_per_record=(\d+) record_fetches=(\d+) ms=([\d.]+)", ln)
    if not m:
        continue
    k, stride, rd, fetches, ms = m.group(1), int(m.group(2)), int(m.group(3)), float(m.group(4)), float(m.group(5))
    e = {"record_stride_bytes": stride, "bytes_read_per_record": rd, "record_fetches": fetches, "ms_unprofiled": ms, "g_records_per_s": fetches / (ms * 1e-3) / 1e9}
    if k in rows:
        v, du = rows[k][-1]          # the timed launch (the first launch of each kernel runs zero iterations)
        e["FETCH_SIZE_KiB"] = v; e["counted_bytes_per_record_fetch"] = v * 1024.0 / fetches; e["ms_profiled"] = du / 1e6
    res["kernels"][k] = e
k = res["kernels"]
if "gather<8,8>" in k and "counted_bytes_per_record_fetch" in k["gather<8,8>"]:
    full = k["gather<8,8>"]["counted_bytes_per_record_fetch"]
    res["factor_whole_line"] = 128.0 / full
    if "gather<5,5>" in k:
        # a packed 80-byte record at a 16-byte-aligned offset lies in ONE 128-byte line for offsets 0..48 of 128 (4 of 8 positions) and in two otherwise: 1.5 lines = 192 bytes
        res["factor_node_pattern"] = 192.0 / k["gather<5,5>"]["counted_bytes_per_record_fetch"]
    res["reading"] = ("counted bytes per fetch of an aligned record whose 16 / 64 / 80 / 128 bytes are read: %s; the unprofiled record rates say what really moves (equal rates = whole lines)"
                      % ", ".join("%.1f" % k[n]["counted_bytes_per_record_fetch"] for n in ("gather<1,8>", "gather<4,8>", "gather<5,8>", "gather<8,8>") if n in k))
os.makedirs(os.path.join(ROOT, "gpurun_out", "profiles_new"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "profiles_new", "r05_fetch_size_calibration.json"), "w"), indent=1)
print(plain); print(json.dumps(res, indent=1))
subprocess.run(["find", out_dir, "-name", "*.db", "-size", "+20M", "-delete"])
