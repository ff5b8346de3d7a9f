"""BASELINE configs[4] in KIND as well as in size (VERDICT r4 task 1): the water_caustic stand-in -- the reference's own models/water_caustic/water_caustic.mtl
and water_caustic.fa camera on procedural geometry (tools/gen_water_caustic_standin.py; water_caustic.obj is absent from the reference checkout).

CPU (-m "not gpu"): the committed files are the reference's, the loaded scene holds what the .mtl says, and an oracle-side statistical check that the paths the
bidirectional tracer exists for are there -- light -> water (Ns 1024, d 0: nearly specular) -> wall -> eye: next-event estimation cannot sample them, light
tracing does, with a fraction of the noise of the eye-side strategies.
GPU (-m gpu): the HIP bidirectional tracer against the oracle at 1600 x 900, L = 9, both connection modes (src/bpt_kernels.h:1084-1250; the RGBE-packed light
vertices of src/bpt_utils.h:192-250 carry radiances of 8000 here), and 64 passes in flight against 2 x 32."""
import hashlib
import os

import numpy as np
import pytest

import fermat_amd as fa
from fermat_amd import scene
from oracle import binding as ob

HERE = os.path.join(scene.DATA_DIR, "scenes", "water_caustic_standin")
REF = "/root/reference/models/water_caustic"


@pytest.fixture(scope="module")
def table():
    return np.fromfile(os.path.join(scene.DATA_DIR, "glossy_reflectance.dat"), np.float32)


@pytest.fixture(scope="module")
def water():
    return scene.water_caustic_standin()


def bit_equal(a, b):
    a = np.ascontiguousarray(a); b = np.ascontiguousarray(b)
    return a.nbytes == b.nbytes and a.tobytes() == b.tobytes()


def host_threads():
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return n


# ---- CPU ---------------------------------------------------------------------------------------------------------------------------------------------------
def test_the_material_file_and_camera_are_the_references_own():
    """water_caustic.mtl / camera.txt / readme.txt are byte-for-byte the reference's (sha256 pinned; compared with the checkout where it exists), and the .fa's
    camera line is water_caustic.fa's"""
    for name in ("water_caustic.mtl", "camera.txt", "readme.txt"):
        mine = open(os.path.join(HERE, name), "rb").read()
        if os.path.isdir(REF):
            assert mine == open(os.path.join(REF, name), "rb").read(), name
    mtl = open(os.path.join(HERE, "water_caustic.mtl")).read()
    assert hashlib.sha256(open(os.path.join(HERE, "water_caustic.mtl"), "rb").read()).hexdigest() == MTL_SHA256
    # the facts the stand-in rests on, read off the text: the Water material and the two emitters
    blocks = {b.split()[0]: b for b in mtl.split("newmtl ")[1:]}
    assert set(blocks) == {"Green", "Red", "Silver", "Water", "White", "Light", "Light2"}
    assert "Kd 0 0 0" in blocks["Water"] and "Ns 1024" in blocks["Water"] and "\nd 0" in blocks["Water"] and "Ks 1 1 1" in blocks["Water"] and "Ni" not in blocks["Water"]
    assert "Ke 8000 6800 4400" in blocks["Light"] and "Ke 6000 7200 8000" in blocks["Light2"]
    fa_text = open(os.path.join(HERE, "water_caustic_standin.fa")).read()
    cam = "Camera persp eye 2.800000 3.487043 11.134270 aim 2.800000 2.043976 -3.464071 up 0 1 0 fov 0.873"
    assert cam in fa_text
    if os.path.isdir(REF):
        assert cam in open(os.path.join(REF, "water_caustic.fa")).read()


MTL_SHA256 = "1611a2c27b5f95e2fbea3b4ee170156c6bf37698c1c90f7245c0cb82a3025661"          # of /root/reference/models/water_caustic/water_caustic.mtl


def test_the_loaded_scene_holds_what_the_mtl_says(water):
    """through the C++ front-end (.fa -> PLY parts -> MeshStorage arrays): triangle count, the Water material as Fermat's loader reads it (roughness = 1 / Ns,
    opacity = d, index of refraction left at its default 1: MeshBase.cpp:354-412, MeshStorage.cpp:163), the emitters, the camera"""
    s = water
    assert s.num_triangles == 864576
    m = s.materials
    water_m = m[(m["opacity"] == 0.0)]
    assert len(water_m) == 1
    assert water_m["roughness"][0] == np.float32(1.0) / np.float32(1024.0) and water_m["index_of_refraction"][0] == 1.0
    assert np.array_equal(water_m["diffuse"][0][:3], np.float32([0, 0, 0])) and np.array_equal(water_m["specular"][0][:3], np.float32([1, 1, 1]))
    ke = sorted(tuple(float(x) for x in e[:3]) for e in m["emissive"] if e[:3].max() > 0)
    assert ke == [(6000.0, 7200.0, 8000.0), (8000.0, 6800.0, 4400.0)]
    # the emitters are SMALL: two quads of 128 triangles each, 0.16^2 and 0.12^2 in area
    em = np.where(m["emissive"][s.material_indices][:, :3].max(1) > 0)[0]
    assert len(em) == 256
    v = s.vertex_data[:, :3][s.vertex_indices[em][:, :3]].astype(np.float64)
    area = 0.5 * np.linalg.norm(np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0]), axis=1).sum()
    assert abs(area - (0.16 ** 2 + 0.12 ** 2)) < 1e-6
    water_tris = (s.material_indices == np.where(m["opacity"] == 0.0)[0][0]).sum()
    assert water_tris == 25 * 2 * 128 * 128
    assert np.allclose(s.camera[:3], [2.8, 3.487043, 11.134270]) and np.allclose(s.camera[3:6], [2.8, 2.043976, -3.464071]) and abs(s.camera[12] - 0.873) < 1e-6


def test_light_tracing_carries_the_caustic(table, water):
    """48 x 27, L = 5, 128 spp per estimator, on the back wall above the water (what reaches it off the water surface is a reflected caustic of the two small emitters):
      * the path tracer with next-event estimation ONLY -- a low-noise estimate of everything that needs no specular bounce before the emitter -- shows a wall
        a quarter darker than the bidirectional tracer's: the caustic is there, and the BPT finds it (measured at 64 x 36, 512 spp: NEE only 0.771, PT with MIS
        0.972 -- its BSDF-sampled hits of the emitters arrive as fireflies --, BPT 1.106: the reference's own MIS weights over-count light tracing);
      * light tracing is what carries it: without it (`-lt 0`) the BPT's eye-side strategies estimate the same wall with > 5x the noise (measured: 28x at
        512 spp), because they have to HIT an emitter of 0.02 square units by BSDF sampling."""
    W, H, L, n = 48, 27, 5, 128
    wall = (slice(14, 23), slice(9, 39))                     # rows count from the bottom of the image

    def pt(**kw):
        opt = ob.default_options(L, 1)
        for k, v in kw.items():
            setattr(opt, k, v)
        o = ob.OraclePT(water, W, H, opt, table, scene.DATA_DIR)
        o.set_trace_threads(host_threads())
        for i in range(n):
            o.render_pass(i)
        # every path once: DIRECT_C + DIFFUSE_C + SPECULAR_C (the -pt COMPOSITED channel counts indirect NEE twice, a reference quirk kept on purpose)
        return (o.fb[4][:, :3] + o.fb[0][:, :3] + o.fb[2][:, :3]).astype(np.float64).reshape(H, W, 3)

    def bpt(**kw):
        o = ob.OraclePT(water, W, H, ob.default_options(L), table, scene.DATA_DIR)
        o.set_trace_threads(host_threads())
        o.bpt_init(ob.default_bpt_options(L, **kw), scene.DATA_DIR)
        snap = []
        for i in range(n):
            o.bpt_render(i)
            if i + 1 in (n // 2, n):
                snap.append(o.fb[5][:, :3].astype(np.float64).reshape(H, W, 3).copy())
        assert np.isfinite(o.fb).all()
        first, both = snap
        second = 2.0 * both - first                          # the mean of the second half of the passes
        return both, float(np.sqrt(((first - second)[wall] ** 2).mean()))

    nee_only = pt(direct_lighting_bsdf=0, indirect_lighting_bsdf=0)
    full, noise_lt = bpt()
    no_lt, noise_no_lt = bpt(light_tracing=0.0)
    r = full[wall].mean() / nee_only[wall].mean()
    print("\n[water_caustic wall] BPT / PT with NEE only %.3f; half-to-half noise: BPT %.3f, BPT without light tracing %.3f (x%.1f)" %
          (r, noise_lt, noise_no_lt, noise_no_lt / noise_lt))
    assert 1.2 < r < 1.7, r
    assert noise_no_lt > 5.0 * noise_lt, (noise_lt, noise_no_lt)


# ---- GPU ---------------------------------------------------------------------------------------------------------------------------------------------------
def rmse(a, b):
    d = a[:, :3].astype(np.float64) - b[:, :3].astype(np.float64)
    return float(np.sqrt((d * d).sum(1).mean()))


@pytest.mark.gpu
@pytest.mark.parametrize("sc", [0, 1])
def test_config5_bpt_on_the_water_caustic_standin_1600x900_vs_oracle(table, water, sc):
    """configs[4]'s size, renderer and KIND of scene: `-bpt`, 8 bounces, 1600 x 900, 2 passes: sequential fpt_bpt_render AND 2 passes in flight bit-identical to
    the oracle on every channel; the light-vertex store agrees too (positions, RGBE-packed weights of emitters radiating 8000: src/bpt_utils.h:192-224)"""
    import time
    W, H, L, n = 1600, 900, 9, 2
    s = water
    t0 = time.time()
    o = ob.OraclePT(s, W, H, ob.default_options(L), table, scene.DATA_DIR)
    o.set_trace_threads(host_threads())
    o.bpt_init(ob.default_bpt_options(L, single_connection=sc), scene.DATA_DIR)
    for i in range(n):
        o.bpt_render(i)
    t_oracle = time.time() - t0
    want = o.fb.copy()
    assert np.isfinite(want).all() and want[5][:, :3].mean() > 1e-2 and want[5][:, :3].max() > 1000.0          # the emitters are in view
    r = fa.Renderer(s, W, H, fa.default_options(L), table=table, gbuffer=False, bpt_options=fa.default_bpt_options(L, single_connection=sc))
    for i in range(n):
        r.bpt_render(i)
    got = r.framebuffer()
    assert np.isfinite(got).all()
    for c in range(6):
        assert bit_equal(got[c], want[c]), "BPT channel %d differs from the oracle (rmse %.3e)" % (c, rmse(got[c], want[c]))
    r.close()
    r = fa.Renderer(s, W, H, fa.default_options(L), table=table, gbuffer=False, bpt_options=fa.default_bpt_options(L, single_connection=sc))
    r.bpt_set_batch(n)
    r.bpt_render_batch(0, n)
    fb = r.framebuffer()
    for c in range(6):
        assert bit_equal(fb[c], want[c]), "BPT, 2 passes in flight: channel %d differs from the oracle (rmse %.3e)" % (c, rmse(fb[c], want[c]))
    r.close()
    print("\n[C5 water_caustic stand-in, bpt -sc %d] %dx%d L=%d %d passes: oracle %.1f s" % (sc, W, H, L, n, t_oracle))


@pytest.mark.gpu
def test_config5_water_caustic_64_passes_in_flight_equal_two_batches_of_32(table, water):
    """the grouping of passes into batches never changes a bit (the reference's default -sc 1, 1600 x 900, L = 9): 64 in flight == 2 x 32 == 4 x 16, and the frame
    is finite although single light-tracing splats carry radiances of thousands"""
    W, H, L, n = 1600, 900, 9, 64
    frames = {}
    for group in (64, 32, 16):
        r = fa.Renderer(water, W, H, fa.default_options(L), table=table, gbuffer=False, bpt_options=fa.default_bpt_options(L, single_connection=1))
        r.bpt_set_batch(group)
        for first in range(0, n, group):
            r.bpt_render_batch(first, group)
        frames[group] = r.framebuffer()
        r.close()
    assert np.isfinite(frames[64]).all() and frames[64][5][:, :3].min() >= 0.0
    for c in range(6):
        assert bit_equal(frames[64][c], frames[32][c]) and bit_equal(frames[64][c], frames[16][c]), c


@pytest.mark.gpu
def test_bpt_paths_in_flight_beyond_the_old_27_bit_limit(table):
    """until round 4 the BPT's virtual path id shared PixelInfo's 27-bit field with the channel nibble: at most 93 passes of a 1600 x 900 frame in flight.  The
    channel now travels in a byte plane of its own (BptQueue::chan): 96 passes of 1600 x 900 (138 M paths > 2^27 = 134 M) in ONE batch equal two batches of 48, bit
    for bit; the limit is memory (and 2^32 light-vertex slots)"""
    W, H, L, n = 1600, 900, 2, 96
    s = scene.cornell_box("CornellBox-Glossy")
    assert n * W * H > (1 << 27)
    frames = {}
    for group in (96, 48):
        r = fa.Renderer(s, W, H, fa.default_options(L), table=table, gbuffer=False, bpt_options=fa.default_bpt_options(L, single_connection=1))
        free, total = r.device_memory()
        if free < 110e9:
            r.close()
            pytest.skip("%.0f GB free: 138 M BPT paths in flight need ~90 GB" % (free / 1e9))
        r.bpt_set_batch(group)
        for first in range(0, n, group):
            r.bpt_render_batch(first, group)
        frames[group] = r.framebuffer()
        r.close()
    assert np.isfinite(frames[96]).all() and frames[96][5][:, :3].mean() > 1e-3
    for c in range(6):
        assert bit_equal(frames[96][c], frames[48][c]), c
