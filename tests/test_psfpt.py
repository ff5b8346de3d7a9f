"""Path-space-filtering path tracer (`-psfpt`): oracle properties on CPU, HIP-vs-oracle parity on GPU."""
import numpy as np
import pytest

import fermat_amd as fa
from fermat_amd import scene
from oracle import binding as ob


def _oracle(s, table, W, H, L, n, psf=True, **kw):
    o = ob.OraclePT(s, W, H, ob.default_options(L), table, scene.DATA_DIR)
    if psf:
        o.psf_enable(ob.default_psf_options(**kw))
    for i in range(n):
        o.render_pass(i)
    return o


def test_psf_cache_properties(table, cornell):
    W, H, L = 40, 30, 4
    o = _oracle(cornell, table, W, H, L, 8)
    cells = o.psf_cells()
    assert len(cells["keys"]) > 50 and len(np.unique(cells["keys"])) == len(cells["keys"])
    assert (cells["counts"] >= 1).all() and (cells["sums"] >= 0).all()
    # a path opens a cache vertex at most once per bounce (again after a glossy bounce, which keeps the cache id invalid)
    assert 0 < o.psf_ref_count() <= W * H * L
    # key layout: 3 x 17-bit cell coordinates, 5-bit level, 4-bit normal (src/spatial_hash.h:160-166)
    level = (cells["keys"] >> np.uint64(51)) & np.uint64(31)
    assert level.max() <= 17 and (cells["keys"] >> np.uint64(60) == 0).all()
    coords = [(cells["keys"] >> np.uint64(sh)) & np.uint64((1 << 17) - 1) for sh in (0, 17, 34)]
    assert all((c <= (np.uint64(1) << level) + np.uint64(3)).all() for c in coords)      # the jitter disk may step just outside the box
    # the cache is kept across passes and cleared every psf_temporal_reuse passes
    o2 = _oracle(cornell, table, W, H, L, 5, psf_temporal_reuse=4)      # passes 0..3 fill, pass 4 clears and refills
    o1 = _oracle(cornell, table, W, H, L, 4, psf_temporal_reuse=4)
    assert o2.psf_cells()["counts"].sum() < o1.psf_cells()["counts"].sum()
    # clamp_frame(100): no radiance above the clamp; filtering lowers the pixel-to-pixel noise of the indirect term
    pt = _oracle(cornell, table, W, H, L, 8, psf=False)
    assert o.fb[5][:, :3].max() <= 100.0 and np.isfinite(o.fb).all()
    rough = lambda a: float(np.abs(np.diff(a.reshape(H, W, 3), axis=1)).mean())   # noqa: E731
    ind_pt = pt.fb[0][:, :3] + pt.fb[2][:, :3]; ind_psf = o.fb[0][:, :3] + o.fb[2][:, :3]
    assert rough(ind_psf) < 0.9 * rough(ind_pt)
    # direct lighting seen from the eye is not cached (psf_depth = 1): DIRECT_C equals the path tracer's
    assert np.array_equal(o.fb[4].view(np.uint32), pt.fb[4].view(np.uint32))


def test_psf_depth_beyond_path_length_is_the_clamped_path_tracer(table, cornell):
    """with no vertex eligible for caching the PSFPT vertex processor only differs from the PT one by its NEE weights
    (no doubled indirect term) and the firefly clamp: the BSDF-sampling-only estimates coincide"""
    W, H, L = 24, 18, 3
    opt = ob.default_options(L); opt.direct_lighting_nee = 0; opt.indirect_lighting_nee = 0
    a = ob.OraclePT(cornell, W, H, opt, table, scene.DATA_DIR)
    b = ob.OraclePT(cornell, W, H, opt, table, scene.DATA_DIR)
    b.psf_enable(ob.default_psf_options(psf_depth=99, firefly_filter=1e30))
    for i in range(3):
        a.render_pass(i); b.render_pass(i)
    assert b.psf_ref_count() == 0 and len(b.psf_cells()["keys"]) == 0
    assert np.array_equal(a.fb[5][:, :3].view(np.uint32), b.fb[5][:, :3].view(np.uint32))


@pytest.mark.gpu
def test_cli_psfpt_matches_oracle_image(tmp_path, table):
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "fermat_amd", "bin", "fermat_hip")
    d = os.path.join(scene.DATA_DIR, "scenes", "CornellBox")
    out = str(tmp_path / "psf")
    r = subprocess.run([exe, "-i", os.path.join(d, "CornellBox-Glossy.obj"), "-c", os.path.join(d, "camera-frontal.txt"), "-r", "48", "36", "-psfpt",
                        "-pl", "4", "-filter-width", "2.5", "-passes", "2", "-o", out], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    s = scene.cornell_box("CornellBox-Glossy")
    o = ob.OraclePT(s, 48, 36, ob.default_options(4), table, scene.DATA_DIR)
    o.psf_enable(ob.default_psf_options(psf_width=2.5))
    for i in range(3):
        o.render_pass(i)
    got = (scene.load_tga(out + ".tga")[..., :3] * 255.0 + 0.5).astype(np.uint8)
    assert np.array_equal(got, o.to_rgba().reshape(36, 48, 4)[..., :3])
    # -batch: 5 passes as 3 + 2 in flight (the first run above used the default, 3 in flight): the image is the pass-by-pass one, bit for bit;
    # -batch 1 (one pass per call) gives it too
    for i in range(3, 5):
        o.render_pass(i)
    want = o.to_rgba().reshape(36, 48, 4)[..., :3].astype(np.int32)
    r = subprocess.run([exe, "-i", os.path.join(d, "CornellBox-Glossy.obj"), "-c", os.path.join(d, "camera-frontal.txt"), "-r", "48", "36", "-psfpt",
                        "-pl", "4", "-filter-width", "2.5", "-passes", "4", "-batch", "3", "-o", out + "_b"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    got = (scene.load_tga(out + "_b.tga")[..., :3] * 255.0 + 0.5).astype(np.int32)
    assert np.array_equal(got, want)
    r = subprocess.run([exe, "-i", os.path.join(d, "CornellBox-Glossy.obj"), "-c", os.path.join(d, "camera-frontal.txt"), "-r", "48", "36", "-psfpt",
                        "-pl", "4", "-filter-width", "2.5", "-passes", "4", "-batch", "1", "-o", out + "_s"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert np.array_equal((scene.load_tga(out + "_s.tga")[..., :3] * 255.0 + 0.5).astype(np.int32), want)


@pytest.mark.gpu
@pytest.mark.parametrize("scene_name,L,kw", [("CornellBox-JP", 4, {}), ("CornellBox-Glossy", 5, dict(psf_width=2.0, psf_max_prob=8.0)),
                                               ("CornellBox-JP", 4, dict(psf_temporal_reuse=2, firefly_filter=5.0))])
def test_gpu_psfpt_parity(table, scene_name, L, kw):
    s = scene.cornell_box(scene_name)
    W, H = 64, 48
    r = fa.Renderer(s, W, H, fa.default_options(L), table=table, psf_options=fa.default_psf_options(**kw))
    o = ob.OraclePT(s, W, H, ob.default_options(L), table, scene.DATA_DIR)
    o.psf_enable(ob.default_psf_options(**kw))
    for i in range(4):
        r.psf_render(i, sync=True); o.render_pass(i)
        cg, co = r.psf_cells(), o.psf_cells()
        assert np.array_equal(cg["keys"], co["keys"]) and np.array_equal(cg["counts"], co["counts"]) and np.array_equal(cg["sums"], co["sums"]), i
    fb = r.framebuffer()
    for c in range(8):
        assert np.array_equal(fb[c].view(np.uint32), o.fb[c].view(np.uint32)), "channel %d" % c
    r.close()


@pytest.mark.gpu
def test_gpu_psfpt_full_size_is_deterministic(table):
    """1600x900, 8 bounces, 2^24-cell cache on the bathroom stand-in: cache sums are fixed-point integer atomics and cells are addressed
    by key, so two runs give the same cells (as multisets) and the same frame bit for bit; the frame is finite and clamped to 100."""
    W, H, L = 1600, 900, 9
    s = scene.bathroom_standin(0.25)
    frames, cells = [], []
    for _ in range(2):
        r = fa.Renderer(s, W, H, fa.default_options(L), table=table, gbuffer=False, psf_options=fa.default_psf_options())
        for i in range(2):
            r.psf_render(i, sync=True)
        frames.append(r.framebuffer()[5].copy())
        c = r.psf_cells()
        order = np.argsort(c["keys"], kind="stable")
        cells.append((c["keys"][order], c["counts"][order], c["sums"][order]))
        r.close()
    assert np.array_equal(frames[0].view(np.uint32), frames[1].view(np.uint32))
    for a, b in zip(cells[0], cells[1]):
        assert np.array_equal(a, b)
    assert np.isfinite(frames[0]).all() and frames[0][:, :3].min() >= 0 and frames[0][:, :3].max() <= 100.0 and len(cells[0][0]) > 1000


@pytest.mark.gpu
@pytest.mark.parametrize("scene_name,W,H,L,n_ranks", [("CornellBox-Glossy", 64, 48, 5, 2), ("CornellBox-JP", 96, 64, 6, 3)])
def test_gpu_psfpt_tile_sharded_equals_full_frame(table, scene_name, W, H, L, n_ranks):
    """PSFPT under tile sharding (fpt_psfpt_set_sharded): n rank contexts on ONE GPU, each rendering its scanlines; after every pass the
    ranks' cell records are merged by key into every rank's global table (export / import, the hand-driven twin of the RCCL exchange) and
    fpt_psfpt_finish blends.  The tables of all ranks equal the full-frame renderer's cell for cell, and the assembled frame equals the
    full-frame one bit for bit on every channel -- the cache sums are integers, so the sharding changes nothing."""
    import torch
    from fermat_amd.distributed import device_bytes, PSF_RECORD_BYTES
    s = scene.cornell_box(scene_name)
    psf = dict(psf_temporal_reuse=3)                      # the reset of the cache happens inside the test's 5 passes
    full = fa.Renderer(s, W, H, fa.default_options(L), table=table, psf_options=fa.default_psf_options(**psf))
    lists = fa.tile_pixel_lists(W, H, n_ranks, tile=(W, 1))
    ranks = [fa.Renderer(s, W, H, fa.default_options(L), table=table, pixels=lists[k], psf_options=fa.default_psf_options(**psf)) for k in range(n_ranks)]
    for r in ranks:
        r.psf_set_sharded(True)
    total_records = 0
    for i in range(5):
        full.psf_render(i, sync=True)
        for r in ranks:
            r.psf_render(i)
        exported = []
        for r in ranks:
            ptr, n = r.psf_export_cells()
            exported.append(device_bytes(ptr, n * PSF_RECORD_BYTES, r.dev).clone() if n else None)      # what a host would put on the wire
            total_records += n
        for r in ranks:
            for t in exported:
                if t is not None:
                    r.psf_import_cells(t.data_ptr(), t.numel() // PSF_RECORD_BYTES)
            r.psf_finish(sync=True)
        want = full.psf_cells()
        for r in ranks:
            got = r.psf_cells()
            assert np.array_equal(got["keys"], want["keys"]) and np.array_equal(got["counts"], want["counts"]) and np.array_equal(got["sums"], want["sums"]), i
    assert total_records > 0
    ref = full.framebuffer()
    for k, r in enumerate(ranks):
        fb = r.framebuffer()
        px = lists[k]
        for c in range(8):
            assert np.array_equal(fb[c][px].view(np.uint32), ref[c][px].view(np.uint32)), "rank %d channel %d" % (k, c)
    # errors: a second render before the pending pass is finished is refused
    ranks[0].psf_render(5)
    with pytest.raises(fa.FptError):
        ranks[0].psf_render(6)
    for r in ranks + [full]:
        r.close()


@pytest.mark.gpu
@pytest.mark.parametrize("scene_name,W,H,L,groups,reuse", [("CornellBox-Glossy", 64, 48, 5, (3, 3), 64), ("CornellBox-JP", 96, 64, 6, (4, 2, 1), 3)])
def test_gpu_psfpt_passes_in_flight(table, scene_name, W, H, L, groups, reuse):
    """fpt_psfpt_render_batch: passes in flight, each into its own pass table, folded into the cache in pass order (a reset of the reuse
    window may fall inside a batch).  After every batch the cache equals the sequential renderer's (= the oracle's) cell for cell, bit for
    bit; and since round 3 so does the FRAME, every channel, .w included: a path's frame contributions -- emission, the frame share of its light samples,
    the blends of its cache references -- are kept in the cells of a contribution log and applied by the merge in the sequential order."""
    s = scene.cornell_box(scene_name)
    seq = fa.Renderer(s, W, H, fa.default_options(L), table=table, psf_options=fa.default_psf_options(psf_temporal_reuse=reuse))
    bat = fa.Renderer(s, W, H, fa.default_options(L), table=table, psf_options=fa.default_psf_options(psf_temporal_reuse=reuse))
    bat.psf_set_batch(max(groups))
    first = 0
    for g in groups:
        for i in range(first, first + g):
            seq.psf_render(i, sync=True)
        if g > 1:
            bat.psf_render_batch(first, g, sync=True)
        else:
            bat.psf_render(first, sync=True)
        first += g
        a, b = seq.psf_cells(), bat.psf_cells()
        assert np.array_equal(a["keys"], b["keys"]) and np.array_equal(a["counts"], b["counts"]) and np.array_equal(a["sums"], b["sums"]), first
    want, got = seq.framebuffer(), bat.framebuffer()
    assert np.isfinite(got).all()
    for c in (5, 0, 1, 2, 3, 4, 7):
        d = got[c][:, :3].astype(np.float64) - want[c][:, :3].astype(np.float64)
        assert np.array_equal(got[c].view(np.uint32), want[c].view(np.uint32)), (c, float(np.sqrt((d * d).sum(1).mean())))
    assert want[5][:, :3].mean() > 1e-3
    seq.close(); bat.close()
