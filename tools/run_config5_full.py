#!/usr/bin/env python3
"""BASELINE configs[4] AS SPECIFIED, once: the bidirectional path tracer (`-bpt`, the reference's default `-sc 1`), 1600x900, 8 bounces, **4096 passes**, on the
water_caustic stand-in (fermat_amd/scene.py water_caustic_standin: the reference's own water_caustic.mtl and camera on procedural geometry; the .obj is absent from
the checkout).  The parity tests run this configuration's size and kind for 2 passes against the oracle (tests/test_water_caustic.py) -- the oracle needs ten hours
for 4096; this run is the configuration's own length on the GPU: wall time, rate, and what the frame looks like on the way (finite, converging as 1 / sqrt(N)).

    python tools/run_config5_full.py [passes] [in_flight]      -> gpurun_out/r05_config5_4096spp.json + .png (the name is round 5s; copy into profiles/ under the current round) (tone-mapped thumbnail)"""
import json
import os
import struct
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fermat_amd as fa                      # noqa: E402
from fermat_amd import scene                 # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
P = int(sys.argv[2]) if len(sys.argv) > 2 else 32
W, H, L = 1600, 900, 9
COMPOSITED = 5


def write_png(path, rgb):
    h, w, _ = rgb.shape
    raw = b"".join(b"\x00" + rgb[y].tobytes() for y in range(h))
    def chunk(tag, data):
        c = struct.pack(">I", len(data)) + tag + data
        return c + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    open(path, "wb").write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, 9)) + chunk(b"IEND", b""))


table = np.fromfile(os.path.join(scene.DATA_DIR, "glossy_reflectance.dat"), np.float32)
s = scene.water_caustic_standin()
r = fa.Renderer(s, W, H, fa.default_options(L), table=table, gbuffer=False, bpt_options=fa.default_bpt_options(L, single_connection=1))
r.bpt_set_batch(P)
r.bpt_render_batch(0, P, sync=True)          # warm-up: allocations, first-launch costs; the frame is cleared again below
r.clear_framebuffer()
snaps = {}
marks = sorted(set(m for m in (N // 16, N // 4, N // 2, N) if m >= P and m % P == 0))
t0 = time.time(); t_render = 0.0
done = 0
for first in range(0, N, P):
    t1 = time.time()
    r.bpt_render_batch(first, min(P, N - first))
    done = first + min(P, N - first)
    if done in marks:
        r.synchronize(); t_render += time.time() - t1
        snaps[done] = r.framebuffer()[COMPOSITED][:, :3].astype(np.float64).copy()
    else:
        t_render += time.time() - t1
r.synchronize()
wall = time.time() - t0
fb = r.framebuffer()
final = snaps[N]
lum = lambda a: 0.2126 * a[:, 0] + 0.7152 * a[:, 1] + 0.0722 * a[:, 2]
out = {"config": "BASELINE configs[4]: -bpt -sc 1, %dx%d, max path length %d, %d passes, %d in flight; water_caustic stand-in (%d triangles)" % (W, H, L, N, P, s.num_triangles),
       "passes": N, "wall_s": wall, "render_s_excluding_snapshots": t_render, "msample_per_s": W * H * N / t_render / 1e6,
       "finite": bool(np.isfinite(fb).all()), "mean_rgb": final.mean(0).tolist(), "max_rgb": final.max(0).tolist(),
       "pixels_above_1000": int((final.max(1) > 1000.0).sum()),
       "relative_rmse_of_luminance_vs_final": {str(m): float(np.sqrt(np.mean((lum(snaps[m]) - lum(final)) ** 2)) / max(lum(final).mean(), 1e-30)) for m in marks if m != N}}
# the error of an N/2-pass average against the N-pass one falls as 1/sqrt(N): half-run vs final over sixteenth-run vs final should be near sqrt((1/(N/2) - 1/N) / (1/(N/16) - 1/N)) = sqrt(1/15)
rr = out["relative_rmse_of_luminance_vs_final"]
if str(N // 2) in rr and str(N // 16) in rr and rr[str(N // 16)] > 0:
    out["rmse_ratio_half_over_sixteenth"] = rr[str(N // 2)] / rr[str(N // 16)]; out["rmse_ratio_expected_for_monte_carlo"] = float(np.sqrt(1.0 / 15.0))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r05_config5_%dspp.json" % N), "w"), indent=1)
rgba = r.to_rgba()          # rows bottom-up, like the frame buffer
small = rgba[::-1][: H // 4 * 4, : W // 4 * 4, :3].reshape(H // 4, 4, W // 4, 4, 3).astype(np.float32).mean((1, 3)).astype(np.uint8)
write_png(os.path.join(ROOT, "gpurun_out", "r05_config5_%dspp.png" % N), np.ascontiguousarray(small))
print(json.dumps(out, indent=1))
r.close()
