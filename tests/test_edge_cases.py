"""Edge cases across the renderers (GPU vs oracle): scenes without emitters, a single triangle, path length 1, tiny frames whose size is not
a multiple of any block / tile size."""
import copy

import numpy as np
import pytest

import fermat_amd as fa
from fermat_amd import scene
from oracle import binding as ob

pytestmark = pytest.mark.gpu


def _dark(s):
    d = copy.copy(s)
    d.materials = s.materials.copy()
    d.materials["emissive"][:] = 0.0
    return d


def _single_triangle():
    raw = scene.RawMesh()
    raw.positions = np.float32([[-1, 0, -3], [1, 0, -3], [0, 1.5, -3]])
    raw.v_idx = np.int32([[0, 1, 2]]); raw.n_idx = np.int32([[-1, -1, -1]]); raw.t_idx = np.int32([[-1, -1, -1]])
    raw.mat_idx = np.int32([0])
    m = scene.default_material_params(); m.update(emissive=[2.0, 1.5, 1.0], diffuse=[0.5, 0.5, 0.5])
    raw.materials = [m]
    return scene.Scene(raw, scene.make_camera([0, 0.5, 0], [0, 0.5, -1], [0, 1, 0], 0.9))


@pytest.mark.parametrize("kind", ["pt", "bpt", "psfpt"])
def test_scene_without_emitters_renders_black(table, cornell, kind):
    s = _dark(cornell)
    W, H, L = 33, 21, 3
    kw = dict(bpt_options=fa.default_bpt_options(L)) if kind == "bpt" else dict(psf_options=fa.default_psf_options()) if kind == "psfpt" else {}
    r = fa.Renderer(s, W, H, fa.default_options(L), table=table, **kw)
    o = ob.OraclePT(s, W, H, ob.default_options(L), table, scene.DATA_DIR)
    if kind == "bpt":
        o.bpt_init(ob.default_bpt_options(L), scene.DATA_DIR)
    if kind == "psfpt":
        o.psf_enable(ob.default_psf_options())
    for i in range(2):
        if kind == "bpt":
            r.bpt_render(i); o.bpt_render(i)
        elif kind == "psfpt":
            r.psf_render(i); o.render_pass(i)
        else:
            r.render_pass(i); o.render_pass(i)
    fb = r.framebuffer()
    assert not fb[5][:, :3].any() and np.isfinite(fb).all()
    for c in (0, 1, 2, 3, 4, 5):
        assert np.array_equal(fb[c].view(np.uint32), o.fb[c].view(np.uint32)), (kind, c)
    r.close()


@pytest.mark.parametrize("kind,L", [("pt", 1), ("pt", 4), ("bpt", 1), ("bpt", 3), ("psfpt", 3)])
def test_single_emissive_triangle(table, kind, L):
    s = _single_triangle()
    W, H = 17, 13
    kw = dict(bpt_options=fa.default_bpt_options(L)) if kind == "bpt" else dict(psf_options=fa.default_psf_options()) if kind == "psfpt" else {}
    r = fa.Renderer(s, W, H, fa.default_options(L), table=table, **kw)
    o = ob.OraclePT(s, W, H, ob.default_options(L), table, scene.DATA_DIR)
    if kind == "bpt":
        o.bpt_init(ob.default_bpt_options(L), scene.DATA_DIR)
    if kind == "psfpt":
        o.psf_enable(ob.default_psf_options())
    for i in range(2):
        if kind == "bpt":
            r.bpt_render(i); o.bpt_render(i)
        elif kind == "psfpt":
            r.psf_render(i); o.render_pass(i)
        else:
            r.render_pass(i); o.render_pass(i)
    fb = r.framebuffer()
    assert fb[5][:, :3].max() > 0.5            # the triangle is visible and emits
    for c in (0, 1, 2, 3, 4, 5):
        assert np.array_equal(fb[c].view(np.uint32), o.fb[c].view(np.uint32)), (kind, L, c)
    r.close()


def test_new_api_error_paths(table, cornell):
    r = fa.Renderer(cornell, 16, 16, fa.default_options(3), table=table)
    L = r.L
    import ctypes as C
    # renderers that were not initialised refuse to render, with a message
    assert L.fpt_bpt_render(r.ctx, C.c_uint32(0), C.byref(r.view)) != 0 and b"fpt_bpt_init" in L.fpt_last_error(r.ctx)
    assert L.fpt_psfpt_render(r.ctx, C.c_uint32(0), C.byref(r.view)) != 0 and b"fpt_psfpt_init" in L.fpt_last_error(r.ctx)
    bad = fa.default_bpt_options(40)
    assert L.fpt_bpt_init(r.ctx, C.byref(bad), C.byref(r.view), scene.DATA_DIR.encode(), None, C.c_uint32(0)) != 0
    assert b"max_path_length" in L.fpt_last_error(r.ctx)
    # the filter needs a gbuffer
    r2 = fa.Renderer(cornell, 16, 16, fa.default_options(3), table=table, gbuffer=False)
    assert L.fpt_filter(r2.ctx, C.byref(r2.view), C.c_uint32(0)) != 0 and b"gbuffer" in L.fpt_last_error(r2.ctx)
    r.close(); r2.close()
