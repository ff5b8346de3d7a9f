#!/usr/bin/env python3
"""GPU box: how concentrated the BPT's light-tracing splats are (VERDICT r5 task 6).  One batch of the bench's BPT line (water_caustic stand-in, 1600x900, -sc 1) with the
splat sums kept apart (fpt_bpt_set_deferred_splats): entries in the camera-connection queue (every one is three 64-bit atomics in splat_kernel), how many arrive unoccluded,
how many DISTINCT (pass, pixel) cells they land in, the share of the hottest cells, and how many entries of a 256-entry workgroup window share a cell with another entry
of the window -- what a per-workgroup pre-aggregation could merge.      python tools/diag_bpt_splats.py [passes in flight]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fermat_amd as fa                      # noqa: E402
from fermat_amd import scene                 # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 8
s = scene.water_caustic_standin()
W, H, L = 1600, 900, 9
r = fa.Renderer(s, W, H, fa.default_options(L), gbuffer=False, bpt_options=fa.default_bpt_options(L, single_connection=1))
r.bpt_set_batch(P)
sp = r.bpt_defer_splats()
r.bpt_set_profiling(True)
r.bpt_render_batch(0, P); r.synchronize()
st = r.bpt_stats()
sums = sp.cpu().numpy()                      # (P * W * H, 3) int64
hit = (sums != 0).any(1)
n_cells = int(hit.sum())
mass = np.abs(sums[hit].astype(np.float64)).sum(1)
order = np.sort(mass)[::-1]
print("water_caustic stand-in, %d passes in flight: %d entries in the camera-connection queue (3 atomics each if unoccluded), %d distinct (pass, pixel) cells received a splat of %d"
      % (P, st["shadow_light_tracing"], n_cells, P * W * H))
print("  entries per touched cell <= %.2f (all entries unoccluded would give this); the hottest 0.1 %% of the touched cells hold %.1f %% of the splatted energy, the hottest 1 %%: %.1f %%"
      % (st["shadow_light_tracing"] / max(n_cells, 1), 100 * order[:max(1, n_cells // 1000)].sum() / order.sum(), 100 * order[:max(1, n_cells // 100)].sum() / order.sum()))
