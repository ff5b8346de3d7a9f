"""Post-process path: variance box filter, edge-avoiding a-trous wavelet steps, RenderingContextImpl::filter and
the per-ShadingMode to_rgba.  CPU tests pin the oracle (oracle/o_filter.h) through properties of the reference's definition;
GPU tests compare the HIP kernels with it bit for bit (integer output) / bit-exact floats (shared deterministic math)."""
import os

import numpy as np
import pytest

import fermat_amd as fa
from fermat_amd import api, scene
from oracle import binding as ob


def _pack_geo(pos, nrm, miss=None):
    """GBufferView::pack_geometry (src/framebuffer.h:84-90) in numpy: sphere -> square, 15:15 bits, miss flag in bit 31"""
    phi = np.arctan2(nrm[..., 1], nrm[..., 0]); phi = np.where(phi < 0, phi + 2 * np.pi, phi)
    phi = np.where(np.abs(nrm[..., 2]) >= 1 - 1e-5, 0.0, phi)
    sx = phi / (2 * np.pi); sy = (nrm[..., 2] + 1) * 0.5
    q = lambda v: np.minimum((np.clip(v, 0, 1) * 32767).astype(np.uint32), 32766)   # noqa: E731  quantize(x, n) = min(uint(x n), n-1)
    w = q(sx) | (q(sy) << 15)
    if miss is not None:
        w = w | (miss.astype(np.uint32) << 31)
    out = np.zeros(pos.shape[:-1] + (4,), np.float32)
    out[..., :3] = pos; out[..., 3] = w.view(np.float32)
    return out


def _synthetic(h=40, w=56, seed=0):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    pos = np.stack([xx / w * 2 - 1, yy / h * 2 - 1, -3.0 - 0.5 * (xx > w / 2)], -1).astype(np.float32)      # a depth step in the middle
    nrm = np.zeros((h, w, 3), np.float32); nrm[..., 2] = 1.0
    nrm[:, w // 2:, 0] = 0.6; nrm[:, w // 2:, 2] = 0.8
    miss = np.zeros((h, w), bool); miss[:3, :5] = True
    geo = _pack_geo(pos, nrm, miss)
    img = rng.random((h, w, 4)).astype(np.float32)
    wimg = (rng.random((h, w, 4)) * 0.9 + 0.05).astype(np.float32)
    var = rng.random((h, w)).astype(np.float32)
    # phi_normal, phi_position, phi_color, E, U, V, W
    params = np.float32([2.0, 1.0, 0.01, 0, 0, 0, 1.2, 0, 0, 0, 0.9, 0, 0, 0, -1.5])
    return img, wimg, geo, var, params, miss


def test_filter_variance_is_a_clamped_box_mean():
    rng = np.random.default_rng(1)
    img = rng.random((9, 13, 4)).astype(np.float32)
    v = ob.filter_variance(img, 2)
    for (y, x) in ((0, 0), (4, 6), (8, 12), (1, 11)):
        ys, ye = max(0, y - 2), min(8, y + 2); xs, xe = max(0, x - 2), min(12, x + 2)
        ref = np.float32(0)
        for yy in range(ys, ye + 1):
            for xx in range(xs, xe + 1):
                ref = np.float32(ref + img[yy, xx, 3])
        ref = np.float32(ref / np.float32((ye - ys + 1) * (xe - xs + 1)))
        assert v[y, x] == ref


def test_eaw_properties():
    img, wimg, geo, var, params, miss = _synthetic()
    h, w = img.shape[:2]
    # a constant image is a fixed point of the plain step (weights are normalised), whatever the geometry
    const = np.full_like(img, 0.25)
    out = ob.eaw_step(np.zeros_like(img), -1, None, 0.0, const, geo, var, params, 2)
    assert np.allclose(out[~miss][:, :3], 0.25, rtol=2e-6) and np.array_equal(out[..., 3], const[..., 3])
    # miss pixels pass through untouched and never contribute
    out = ob.eaw_step(np.zeros_like(img), -1, None, 0.0, img, geo, var, params, 1)
    assert np.array_equal(out[miss], img[miss])
    # edges stop the filter: with a huge normal weight the two halves do not mix
    p2 = params.copy(); p2[0] = 1e6; p2[2] = 0.0
    left = img.copy(); left[:, w // 2:, :3] = 0.0; left[:, :w // 2, :3] = 1.0
    out = ob.eaw_step(np.zeros_like(img), -1, None, 0.0, left, geo, None, p2, 1)
    assert np.allclose(out[5:, :w // 2, :3], 1.0, atol=1e-5) and np.allclose(out[5:, w // 2:, :3], 0.0, atol=1e-5)
    # demodulate-in/replace followed by modulate-out/add of the SAME data with all edge weights off is a weighted round trip:
    # dst + w * (img / w) smoothed ~ dst + smoothed img; check the add mode really adds
    base = np.full_like(img, 0.5)
    o1 = ob.eaw_step(base, api.FILTER_OP_MODULATE_OUTPUT | api.FILTER_OP_ADD_MODE, wimg, 1e-4, img, geo, var, params, 1)
    o0 = ob.eaw_step(np.zeros_like(img), api.FILTER_OP_MODULATE_OUTPUT | api.FILTER_OP_ADD_MODE, wimg, 1e-4, img, geo, var, params, 1)
    assert np.allclose(o1 - o0, 0.5, atol=1e-6)
    # the a-trous stride: a single bright pixel spreads to +-2*step only
    spike = np.zeros_like(img); spike[20, 28, :3] = 100.0
    p3 = params.copy(); p3[0] = 0; p3[1] = 0; p3[2] = 0
    flat_geo = _pack_geo(np.zeros((h, w, 3), np.float32) + np.float32([0, 0, -3]), np.tile(np.float32([0, 0, 1]), (h, w, 1)))
    out = ob.eaw_step(np.zeros_like(img), -1, None, 0.0, spike, flat_geo, None, p3, 4)
    nz = np.argwhere(out[..., 0] > 0)
    assert set(np.unique(nz[:, 0] - 20)) == {-8, -4, 0, 4, 8} and set(np.unique(nz[:, 1] - 28)) == {-8, -4, 0, 4, 8}
    # kernel weights 1, 2/3, 1/6 per axis: the central value of the smoothed spike is 100 * 1 / (1 + 2*2/3 + 2/6)^2
    assert out[20, 28, 0] == pytest.approx(100.0 / (1 + 4 / 3 + 1 / 3) ** 2, rel=1e-5)


def test_filter_and_shading_modes_on_a_render(table, cornell):
    o = ob.OraclePT(cornell, 48, 36, ob.default_options(4), table, scene.DATA_DIR)
    o.clear_gbuffer()
    for i in range(3):
        o.render_pass(i)
    o.filter(2)
    f = o.fb[6]; direct = o.fb[4]
    assert np.isfinite(f).all() and (f[:, :3] >= 0).all()
    # FILTERED_C = DIRECT_C + filtered indirect: never below the direct term, equal in total energy to within the filter's smoothing
    assert (f[:, :3] >= direct[:, :3] - 1e-6).all()
    comp = o.fb[5][:, :3].sum(); assert abs(f[:, :3].sum() - comp) / comp < 0.25
    # the filter reduces pixel noise of the indirect term
    ind = (o.fb[5][:, :3] - direct[:, :3]).reshape(36, 48, 3); fi = (f[:, :3] - direct[:, :3]).reshape(36, 48, 3)
    rough = lambda a: float(np.abs(np.diff(a, axis=1)).mean())   # noqa: E731
    assert rough(fi) < 0.7 * rough(ind)
    # shading modes: kShaded through the generic kernel equals the dedicated one; the others are plain channel views
    assert np.array_equal(o.to_rgba(api.SHADING_SHADED), o.to_rgba())
    alb = o.to_rgba(api.SHADING_DIFFUSE_ALBEDO).reshape(-1, 4)
    ref = np.minimum(o.fb[1] * 256.0, 255.0).astype(np.uint8)
    assert np.array_equal(alb, ref)
    nrm = o.to_rgba(api.SHADING_NORMAL).reshape(-1, 4)
    assert (nrm[:, 3] == 0).all() and nrm[:, :3].max() > 200


# ---------------------------------------------------------------------------------------------------------------- GPU parity
@pytest.mark.gpu
def test_gpu_eaw_steps_bit_exact(table, cornell):
    r = fa.Renderer(cornell, 16, 16, fa.default_options(2), table=table)
    img, wimg, geo, var, params, miss = _synthetic(seed=4)
    assert np.array_equal(r.filter_variance(img, 2).view(np.uint32), ob.filter_variance(img, 2).view(np.uint32))
    assert np.array_equal(r.filter_variance(img, 1).view(np.uint32), ob.filter_variance(img, 1).view(np.uint32))
    base = np.random.default_rng(9).random(img.shape).astype(np.float32)
    for op, step, v in ((-1, 1, var), (-1, 8, None), (api.FILTER_OP_DEMODULATE_INPUT | api.FILTER_OP_REPLACE_MODE, 1, var),
                        (api.FILTER_OP_MODULATE_OUTPUT | api.FILTER_OP_ADD_MODE, 64, var), (api.FILTER_OP_MODULATE_INPUT | api.FILTER_OP_DEMODULATE_OUTPUT, 2, var), (0, 4, None)):
        g = r.eaw(base, op, wimg, 1e-4, img, geo, v, params, step)
        o = ob.eaw_step(base, op, wimg, 1e-4, img, geo, v, params, step)
        assert np.array_equal(g.view(np.uint32), o.view(np.uint32)), (op, step)
    r.close()


@pytest.mark.gpu
@pytest.mark.parametrize("scene_name", ["CornellBox-JP", "CornellBox-Glossy"])
def test_gpu_filter_and_shading_modes_parity(table, scene_name):
    s = scene.cornell_box(scene_name)
    r = fa.Renderer(s, 80, 60, fa.default_options(5), table=table)
    o = ob.OraclePT(s, 80, 60, ob.default_options(5), table, scene.DATA_DIR)
    r.clear_gbuffer(); o.clear_gbuffer()
    for i in range(3):
        r.render_pass(i); o.render_pass(i)
    r.filter(2); o.filter(2)
    fb = r.framebuffer()
    assert np.array_equal(r.gb_geo.cpu().numpy().view(np.uint32), o.gb_geo.view(np.uint32))
    assert np.array_equal(fb[6].view(np.uint32), o.fb[6].view(np.uint32))
    for mode in (api.SHADING_SHADED, api.SHADING_FILTERED, api.SHADING_ALBEDO, api.SHADING_DIFFUSE_ALBEDO, api.SHADING_SPECULAR_ALBEDO, api.SHADING_DIFFUSE_COLOR,
                 api.SHADING_SPECULAR_COLOR, api.SHADING_DIRECT_LIGHTING, api.SHADING_VARIANCE, api.SHADING_UV, api.SHADING_NORMAL):
        assert np.array_equal(r.to_rgba(mode), o.to_rgba(mode)), mode
    r.close()
