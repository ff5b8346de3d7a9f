"""Bidirectional path tracer (`-bpt -sc 0`): oracle properties on CPU, HIP-vs-oracle parity on GPU."""
import ctypes as C
import os

import numpy as np
import pytest

import fermat_amd as fa
from fermat_amd import scene
from oracle import binding as ob


def _mean_image(s, table, W, H, L, n, kind, **kw):
    o = ob.OraclePT(s, W, H, ob.default_options(L), table, scene.DATA_DIR)
    if kind == "bpt":
        o.bpt_init(ob.default_bpt_options(L, **kw), scene.DATA_DIR)
        for i in range(n):
            o.bpt_render(i)
    else:
        opt = ob.default_options(L)
        for k, v in kw.items():
            setattr(opt, k, v)
        o = ob.OraclePT(s, W, H, opt, table, scene.DATA_DIR)
        for i in range(n):
            o.render_pass(i)
    return o


def test_packers():
    L = ob.lib()
    L.orc_to_rgbe.restype = C.c_uint32
    L.orc_to_rgbe.argtypes = [C.c_float] * 3
    out = (C.c_float * 3)()
    for rgb in ((1.0, 0.5, 0.25), (17.0, 12.0, 4.0), (1e-3, 2e-3, 5e-4), (300.0, 1.0, 0.0)):
        p = L.orc_to_rgbe(*rgb)
        L.orc_from_rgbe(C.c_uint32(p), out)
        m = max(rgb)
        # shared-exponent, 8-bit truncated mantissas: never above the input, within 2^-7 of the largest component
        assert all(o <= v + 1e-12 for o, v in zip(out, rgb)) and all(v - o <= m / 128.0 + 1e-12 for o, v in zip(out, rgb))
    assert L.orc_to_rgbe(0.0, 0.0, 0.0) == 0
    L.orc_pack_direction.restype = C.c_uint32
    L.orc_pack_direction.argtypes = [C.c_float] * 3
    rng = np.random.default_rng(2)
    for _ in range(200):
        v = rng.normal(size=3); v /= np.linalg.norm(v)
        p = L.orc_pack_direction(*[float(x) for x in v])
        L.orc_unpack_direction(C.c_uint32(p), out)
        assert np.dot(v, np.float64(list(out))) > 1.0 - 2e-7 * 65535      # 16:16 bits on the sphere->square map


def test_bpt_structure_and_energy(table, cornell):
    """light-vertex store layout, queue bookkeeping, and agreement of the MIS-weighted estimate with its own unweighted
    (BSDF-sampling only) special case: the combination weights of every technique must sum to one"""
    W, H, Lp, n = 16, 12, 3, 1024
    o = _mean_image(cornell, table, W, H, Lp, n, "bpt")
    st = o.bpt_stats()
    assert st["light_queue"][0] == W * H and st["eye_queue"][0] == W * H and len(st["eye_queue"]) <= Lp
    lv = o.bpt_light_vertices()
    cnt = lv["counts"]
    assert cnt.min() >= 1 and cnt.max() <= Lp and st["n_light_vertices"] == cnt.sum()
    npx = W * H
    for d in range(Lp):
        live = cnt > d
        pid = lv["path_id"][d * npx:(d + 1) * npx][live]
        assert ((pid & 0xFFFFFF) == np.nonzero(live)[0]).all() and ((pid >> 24) == d).all()
    # primary light vertices sit on emissive triangles: their packed EDF colour is non-zero
    assert (lv["gbuffer"][:npx, 0] != 0).all()
    full = o.fb[5][:, :3].mean()
    ref = _mean_image(cornell, table, W, H, Lp, n, "bpt", direct_lighting_nee=0, indirect_lighting_nee=0, light_tracing=0.0).fb[5][:, :3].mean()
    pt = _mean_image(cornell, table, W, H, Lp, n, "pt", direct_lighting_nee=0, indirect_lighting_nee=0).fb[5][:, :3].mean()
    assert np.isfinite(o.fb).all()
    assert abs(full / ref - 1.0) < 0.06 and abs(ref / pt - 1.0) < 0.04, (full, ref, pt)
    # all radiance channels are non-negative, light tracing only ever adds
    no_lt = _mean_image(cornell, table, W, H, Lp, 64, "bpt", light_tracing=0.0)
    assert (no_lt.fb[5][:, :3] >= 0).all() and no_lt.bpt_stats()["shadow_light_tracing"] == 0


def test_bpt_is_deterministic(table, cornell):
    a = _mean_image(cornell, table, 24, 16, 4, 3, "bpt"); b = _mean_image(cornell, table, 24, 16, 4, 3, "bpt")
    assert np.array_equal(a.fb.view(np.uint32), b.fb.view(np.uint32))
    b.set_trace_threads(4)
    c = ob.OraclePT(cornell, 24, 16, ob.default_options(4), table, scene.DATA_DIR)
    c.set_trace_threads(4); c.bpt_init(ob.default_bpt_options(4), scene.DATA_DIR)
    for i in range(3):
        c.bpt_render(i)
    assert np.array_equal(a.fb.view(np.uint32), c.fb.view(np.uint32))
