"""Bidirectional path tracer (`-bpt`, both connection modes `-sc 1` (the reference's default) and `-sc 0`):
oracle properties on CPU, HIP-vs-oracle parity on GPU."""
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


def test_bpt_tile_sharding_equals_full_frame(table, cornell):
    """N>1 path of the BPT: each rank traces the light AND eye sub-paths of its tiles, the light-tracing splat sums
    (order-independent integers) are added over the ranks, and every rank's own pixels equal the single-process frame bit for bit"""
    from fermat_amd.api import tile_pixel_lists
    W, H, L = 32, 24, 4
    full = ob.OraclePT(cornell, W, H, ob.default_options(L), table, scene.DATA_DIR)
    full.bpt_init(ob.default_bpt_options(L), scene.DATA_DIR)
    lists = tile_pixel_lists(W, H, 3, tile=8)
    parts = []
    for px in lists:
        o = ob.OraclePT(cornell, W, H, ob.default_options(L), table, scene.DATA_DIR)
        o.bpt_init(ob.default_bpt_options(L), scene.DATA_DIR)
        parts.append((o, px, o.bpt_defer_splats()))
    for i in range(3):
        full.bpt_render(i)
        for o, px, sp in parts:
            o.bpt_render(i, px)
        total = sum(sp.copy() for _, _, sp in parts)          # the integer all-reduce
        assert total.any()
        for o, px, sp in parts:
            sp[...] = total
            o.bpt_resolve_splats()
    merged = np.zeros_like(full.fb)
    for o, px, sp in parts:
        merged[:, px, :] = o.fb[:, px, :]
    for c in range(6):
        assert np.array_equal(merged[c].view(np.uint32), full.fb[c].view(np.uint32)), c


# ---------------------------------------------------------------------------------------------------------------- GPU parity
def _bpt_pair(s, table, W, H, L, **kw):
    r = fa.Renderer(s, W, H, fa.default_options(L), table=table, bpt_options=fa.default_bpt_options(L, **kw))
    o = ob.OraclePT(s, W, H, ob.default_options(L), table, scene.DATA_DIR)
    o.bpt_init(ob.default_bpt_options(L, **kw), scene.DATA_DIR)
    return r, o


@pytest.mark.gpu
@pytest.mark.parametrize("sc", [0, 1])
@pytest.mark.parametrize("scene_name,L", [("CornellBox-JP", 4), ("CornellBox-Glossy", 5)])
def test_gpu_bpt_parity(table, scene_name, L, sc):
    s = scene.cornell_box(scene_name)
    r, o = _bpt_pair(s, table, 64, 48, L, single_connection=sc)
    r.bpt_set_profiling(True)
    r.clear_gbuffer(); o.clear_gbuffer()
    for i in range(3):
        r.bpt_render(i, sync=True); o.bpt_render(i)
        sg, so = r.bpt_stats(), o.bpt_stats()
        assert sg["light_queue"].tolist() == so["light_queue"].tolist() and sg["eye_queue"].tolist() == so["eye_queue"].tolist()
        assert sg["n_light_vertices"] == so["n_light_vertices"]
        # light-vertex store: integer records bit-exact
        lg, lo = r.bpt_light_vertices(), o.bpt_light_vertices()
        assert np.array_equal(lg["counts"], lo["counts"])
        n = 64 * 48
        for d in range(L):
            live = lo["counts"] > d
            sl = slice(d * n, (d + 1) * n)
            for k in ("path_id", "input", "gbuffer"):
                assert np.array_equal(lg[k][sl][live], lo[k][sl][live]), (k, d)
            for k in ("pos", "weights"):
                assert np.array_equal(lg[k][sl][live].view(np.uint32), lo[k][sl][live].view(np.uint32)), (k, d)
    fb = r.framebuffer()
    for c in (0, 1, 2, 3, 4, 5):
        assert np.array_equal(fb[c].view(np.uint32), o.fb[c].view(np.uint32)), "channel %d" % c
    assert np.array_equal(r.gb_geo.cpu().numpy().view(np.uint32), o.gb_geo.view(np.uint32))
    r.close()


@pytest.mark.gpu
def test_cli_bpt_matches_oracle_image(tmp_path, table):
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "fermat_amd", "bin", "fermat_hip")
    d = os.path.join(scene.DATA_DIR, "scenes", "CornellBox")
    out = str(tmp_path / "bpt")
    r = subprocess.run([exe, "-i", os.path.join(d, "CornellBox-JP.obj"), "-c", os.path.join(d, "camera-frontal.txt"), "-r", "48", "36", "-bpt", "-sc", "0",
                        "-pl", "4", "-passes", "2", "-o", out], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    s = scene.cornell_box("CornellBox-JP")
    o = ob.OraclePT(s, 48, 36, ob.default_options(4), table, scene.DATA_DIR)
    o.bpt_init(ob.default_bpt_options(4), scene.DATA_DIR)
    for i in range(3):
        o.bpt_render(i)
    got = (scene.load_tga(out + ".tga")[..., :3] * 255.0 + 0.5).astype(np.uint8)
    assert np.array_equal(got, o.to_rgba().reshape(36, 48, 4)[..., :3])
    # no -sc on the command line = the reference's default single-connection mode (src/renderers/bpt.h:62)
    r = subprocess.run([exe, "-i", os.path.join(d, "CornellBox-JP.obj"), "-c", os.path.join(d, "camera-frontal.txt"), "-r", "48", "36", "-bpt",
                        "-pl", "4", "-passes", "2", "-o", out + "_sc1"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    o1 = ob.OraclePT(s, 48, 36, ob.default_options(4), table, scene.DATA_DIR)
    o1.bpt_init(ob.default_bpt_options(4, single_connection=1), scene.DATA_DIR)
    for i in range(3):
        o1.bpt_render(i)
    got1 = (scene.load_tga(out + "_sc1.tga")[..., :3] * 255.0 + 0.5).astype(np.uint8)
    assert np.array_equal(got1, o1.to_rgba().reshape(36, 48, 4)[..., :3]) and not np.array_equal(got1, got)
    # the runs above kept the three passes in flight behind render() (the default); 2 + 1 and one pass per call write the same file
    for batch in ("2", "1"):
        r = subprocess.run([exe, "-i", os.path.join(d, "CornellBox-JP.obj"), "-c", os.path.join(d, "camera-frontal.txt"), "-r", "48", "36", "-bpt",
                            "-pl", "4", "-passes", "2", "-batch", batch, "-o", out + "_b" + batch], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        assert open(out + "_b" + batch + ".tga", "rb").read() == open(out + "_sc1.tga", "rb").read(), batch


@pytest.mark.gpu
def test_gpu_bpt_option_variants(table, cornell):
    for kw in (dict(light_tracing=0.0), dict(rr=0), dict(direct_lighting_nee=0), dict(indirect_lighting_nee=0, light_tracing=0.0),
               dict(visible_lights=0, direct_lighting_bsdf=0), dict(use_vpls=1), dict(max_path_length=2), dict(max_path_length=1),
               dict(single_connection=1, light_tracing=0.0), dict(single_connection=1, direct_lighting_nee=0), dict(single_connection=1, use_vpls=1, rr=0),
               dict(single_connection=1, max_path_length=2), dict(single_connection=1, max_path_length=1)):
        L = kw.pop("max_path_length", 3)
        r, o = _bpt_pair(cornell, table, 40, 30, L, **kw)
        for i in range(2):
            r.bpt_render(i); o.bpt_render(i)
        fb = r.framebuffer()
        assert np.array_equal(fb[5].view(np.uint32), o.fb[5].view(np.uint32)) and np.array_equal(fb[4].view(np.uint32), o.fb[4].view(np.uint32)), kw
        r.close()


@pytest.mark.gpu
def test_gpu_bpt_tile_sharding(table, cornell):
    """two contexts on one GPU stand in for two ranks: disjoint tiles, deferred splats summed as the RCCL integer all-reduce would"""
    W, H, L = 96, 64, 4
    full = fa.Renderer(cornell, W, H, fa.default_options(L), table=table, bpt_options=fa.default_bpt_options(L))
    lists = fa.tile_pixel_lists(W, H, 2, tile=32)
    parts = [fa.Renderer(cornell, W, H, fa.default_options(L), table=table, pixels=px, bpt_options=fa.default_bpt_options(L)) for px in lists]
    sps = [p.bpt_defer_splats() for p in parts]
    for i in range(2):
        full.bpt_render(i)
        for p in parts:
            p.bpt_render(i, sync=True)
        total = sps[0] + sps[1]
        for p, sp in zip(parts, sps):
            sp.copy_(total); p.torch.cuda.synchronize(p.dev)
            p.bpt_resolve_splats()
    ref = full.framebuffer()
    merged = np.zeros_like(ref)
    for p, px in zip(parts, lists):
        merged[:, px, :] = p.framebuffer()[:, px, :]
    for c in range(6):
        assert np.array_equal(merged[c].view(np.uint32), ref[c].view(np.uint32)), c
    for p in parts + [full]:
        p.close()


@pytest.mark.gpu
def test_gpu_bpt_config5_size_properties(table):
    """BASELINE config 5 size (1600x900, -bpt, 8 bounces) on the bathroom stand-in (water_caustic's OBJ is absent from the reference
    checkout): the pass is deterministic bit for bit (light-tracing splats are fixed-point integer atomics), finite and non-negative,
    and one rank's interleaved-scanline share -- with the splat buffers summed as the integer all-reduce would -- reproduces the
    full-frame pixels exactly."""
    W, H, L = 1600, 900, 9
    s = scene.bathroom_standin(0.25)
    opts = lambda: dict(table=table, gbuffer=False, bpt_options=fa.default_bpt_options(L))
    full = fa.Renderer(s, W, H, fa.default_options(L), **opts())
    full.bpt_render(0, sync=True)
    ref = full.framebuffer()[5].copy()
    full.clear_framebuffer()
    full.bpt_render(0, sync=True)
    assert np.array_equal(full.framebuffer()[5].view(np.uint32), ref.view(np.uint32))
    assert np.isfinite(ref).all() and ref[:, :3].min() >= 0 and ref[:, :3].mean() > 1e-3
    full.close()
    lists = fa.tile_pixel_lists(W, H, 2, tile=(W, 1))
    parts = [fa.Renderer(s, W, H, fa.default_options(L), pixels=px, **opts()) for px in lists]
    sps = [p.bpt_defer_splats() for p in parts]
    for p in parts:
        p.bpt_render(0, sync=True)
    total = sps[0] + sps[1]
    for p, sp, px in zip(parts, sps, lists):
        sp.copy_(total); p.torch.cuda.synchronize(p.dev)
        p.bpt_resolve_splats()
        assert np.array_equal(p.framebuffer()[5][px].view(np.uint32), ref[px].view(np.uint32))
        p.close()


@pytest.mark.gpu
@pytest.mark.parametrize("sc", [0, 1])
def test_gpu_bpt_batched_passes_match_sequential(table, sc):
    """fpt_bpt_render_batch ("passes in flight"): the same light / eye sub-paths and contributions as n fpt_bpt_render calls (per-bounce
    queue sizes are the sums of the sequential ones); every term an eye path hands the frame is kept in its own cell of the batch's log and the
    merge applies them in the order of the sequential launches, so every channel is BIT-IDENTICAL to the sequential frame and to the oracle, whatever
    the grouping of the passes into batches."""
    s = scene.cornell_box("CornellBox-Glossy")
    W, H, L, n = 80, 60, 5, 6
    mk = lambda: fa.Renderer(s, W, H, fa.default_options(L), table=table, bpt_options=fa.default_bpt_options(L, single_connection=sc))
    seq = mk(); seq.bpt_set_profiling(True)
    o = ob.OraclePT(s, W, H, ob.default_options(L), table, scene.DATA_DIR)
    o.bpt_init(ob.default_bpt_options(L, single_connection=sc), scene.DATA_DIR)
    tot_l = np.zeros(L, np.int64); tot_e = np.zeros(L, np.int64); tot_s = np.zeros(L, np.int64)
    for i in range(n):
        seq.bpt_render(i, sync=True); o.bpt_render(i)
        st = seq.bpt_stats()
        tot_l[:len(st["light_queue"])] += st["light_queue"]; tot_e[:len(st["eye_queue"])] += st["eye_queue"]; tot_s[:len(st["shadow_eye"])] += st["shadow_eye"]
    ref = seq.framebuffer().astype(np.float64)
    seq.close()
    frames = {}
    for group in (n, 3, 2):
        r = mk(); r.bpt_set_batch(group)
        if group == n:
            r.bpt_set_profiling(True)
        for first in range(0, n, group):
            r.bpt_render_batch(first, group, sync=True)
        if group == n:
            st = r.bpt_stats()
            assert np.array_equal(st["light_queue"], tot_l[:len(st["light_queue"])]) and np.array_equal(st["eye_queue"], tot_e[:len(st["eye_queue"])])
            assert np.array_equal(st["shadow_eye"], tot_s[:len(st["shadow_eye"])])
        frames[group] = r.framebuffer()
        r.close()
    # deferred render(): the library collects the calls (4 at a time, then the 2 that are left when the frame is read)
    r = mk(); r.bpt_set_deferred(4)
    for i in range(n):
        r.bpt_render(i)
    frames["deferred"] = r.framebuffer()
    r.close()
    for c in range(6):
        assert np.array_equal(frames[n][c].view(np.uint32), frames[3][c].view(np.uint32)), c
        assert np.array_equal(frames[n][c].view(np.uint32), frames[2][c].view(np.uint32)), c
        assert np.array_equal(frames[n][c].view(np.uint32), frames["deferred"][c].view(np.uint32)), c
        for other in (ref[c], o.fb[c].astype(np.float64)):
            d = frames[n][c].astype(np.float64) - other
            assert float(np.sqrt((d * d).sum(1).mean())) < 1e-5, c
        assert np.array_equal(frames[n][c].view(np.uint32), o.fb[c].view(np.uint32)), "channel %d of the batch differs from the oracle" % c
    assert frames[n][5][:, :3].mean() > 1e-2


@pytest.mark.gpu
def test_gpu_bpt_batched_tile_sharding(table, cornell):
    """batched + sharded: two ranks' deferred splat sums (3 x int64 per pixel PER PASS IN FLIGHT) are added as one integer all-reduce
    per batch would, then each rank folds them in and merges its planes: bit-identical to the full-frame batched render"""
    W, H, L, n = 96, 64, 4, 3
    full = fa.Renderer(cornell, W, H, fa.default_options(L), table=table, bpt_options=fa.default_bpt_options(L)); full.bpt_set_batch(n)
    full.bpt_render_batch(0, n, sync=True)
    ref = full.framebuffer()
    lists = fa.tile_pixel_lists(W, H, 2, tile=(W, 1))
    parts = [fa.Renderer(cornell, W, H, fa.default_options(L), table=table, pixels=px, bpt_options=fa.default_bpt_options(L)) for px in lists]
    sps = []
    for p in parts:
        p.bpt_set_batch(n); sps.append(p.bpt_defer_splats())
    assert tuple(sps[0].shape) == (W * H * n, 3)
    for p in parts:
        p.bpt_render_batch(0, n, sync=True)
    total = sps[0] + sps[1]
    merged = np.zeros_like(ref)
    for p, sp, px in zip(parts, sps, lists):
        sp.copy_(total); p.torch.cuda.synchronize(p.dev)
        p.bpt_resolve_splats()
        merged[:, px, :] = p.framebuffer()[:, px, :]
    for c in range(6):
        assert np.array_equal(merged[c].view(np.uint32), ref[c].view(np.uint32)), c
    for p in parts + [full]:
        p.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n_ranks,batch", [(2, 1), (3, 1), (2, 3)])
def test_gpu_bpt_sc1_tile_sharding_with_shared_light_vertices(table, cornell, n_ranks, batch):
    """-sc 1 (the reference's default: ONE connection per eye vertex into the list of ALL light vertices) under tile sharding.  With shared light
    vertices every rank stops after its light sub-paths, the ranks hand each other their stored vertices (export / import here, the records stay on
    the one GPU; fpt_bpt_exchange_light_vertices over RCCL between GPUs), and fpt_bpt_finish draws the connections from the same list a single GPU
    builds: the assembled frame is BIT-IDENTICAL to the full-frame render for any number of ranks (without the exchange it is only unbiased)."""
    W, H, L = 96, 64, 4
    bo = lambda: fa.default_bpt_options(L, single_connection=1)      # noqa: E731
    full = fa.Renderer(cornell, W, H, fa.default_options(L), table=table, bpt_options=bo())
    lists = fa.tile_pixel_lists(W, H, n_ranks, tile=(W, 1))
    parts = [fa.Renderer(cornell, W, H, fa.default_options(L), table=table, pixels=px, bpt_options=bo()) for px in lists]
    if batch > 1:
        full.bpt_set_batch(batch)
        for p in parts:
            p.bpt_set_batch(batch)
    sps = [p.bpt_defer_splats() for p in parts]
    for p in parts:
        p.bpt_set_shared_light_vertices(True)
    for first in range(0, 2 * batch, batch):
        if batch > 1:
            full.bpt_render_batch(first, batch, sync=True)
        else:
            full.bpt_render(first, sync=True)
        for p in parts:                                   # light sub-paths only
            (p.bpt_render_batch(first, batch, sync=True) if batch > 1 else p.bpt_render(first, sync=True))
        exported = [p.bpt_export_light_vertices() for p in parts]
        assert all(n > 0 for _, n in exported)
        for i, p in enumerate(parts):
            for j, (ptr, n) in enumerate(exported):
                if i != j:
                    p.bpt_import_light_vertices(ptr, n)
        for p in parts:
            p.bpt_finish(sync=True)
        total = sum(sps[1:], sps[0].clone())
        for p, sp in zip(parts, sps):
            sp.copy_(total); p.torch.cuda.synchronize(p.dev)
            p.bpt_resolve_splats()
    ref = full.framebuffer()
    merged = np.zeros_like(ref)
    for p, px in zip(parts, lists):
        merged[:, px, :] = p.framebuffer()[:, px, :]
    for c in range(6):
        assert np.array_equal(merged[c].view(np.uint32), ref[c].view(np.uint32)), c
    # the calls are refused out of order
    Lb = fa.lib()
    assert Lb.fpt_bpt_finish(parts[0].ctx, C.byref(parts[0].view)) != 0 and b"waiting" in Lb.fpt_last_error(parts[0].ctx)
    for p in parts + [full]:
        p.close()
