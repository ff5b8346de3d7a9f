"""Independent statistical pins of the ORACLE (CPU; `-m "not gpu"`), VERDICT r1 "next round" item 1(b).

The reference ships one property test for its BSDFs (contrib/cugar/bsdf/bsdf_test.h:49-149: sampled g, p against f_and_p within 3 %) and
no golden images, so an error shared by the oracle's restatement and the kernels would pass every HIP-vs-oracle comparison.  These
tests do not compare two pieces of code that were written from the same reading; they check the restated model against what the
reference's model MUST satisfy whatever its implementation:

  1. `Bsdf::sample` (src/bsdf.h:921-1199) is an importance sampler of the lobes `Bsdf::f_and_p` (src/bsdf.h:366-412) evaluates: for
     every lobe c,  E_z[ g * 1(comp == c) ]  =  integral of f_c(w_i, w_o) d(projected solid angle)  -- the left side uses only sample(),
     the right side only f_and_p() on a lat-long quadrature that knows nothing about the sampler (bsdf_test.h's recipe lifted from one
     GGX lobe to the composite layered BSDF incl. clearcoat, transmission and the Kelemen diffuse weight);
  2. the layered model never creates energy (white furnace): the directional albedo  sum_c E[g 1_c] + coat  <= 1  for white materials;
  3. the emitter tables sample points with the density `MeshLight::map` reports (src/lights.h:299-431; src/mesh_lights.cu:164-424):
     triangle frequencies of the VPL set against  max(Ke) * area / norm  computed here from the scene arrays (chi-square);
  4. three estimators of the same integral agree within Monte-Carlo error: the path tracer with BSDF sampling only, with next-event
     estimation only, and with both (MIS), summed over the channels that count every path once (DIRECT_C + DIFFUSE_C + SPECULAR_C);
     and the bidirectional tracer (`-bpt`) agrees with them -- this ties emitter pdfs, BSDF f and p, geometric terms and MIS weights
     together without looking at any of them;
  5. a closed furnace with a known answer: inside a closed box whose walls all emit L_e = 1 and reflect rho (Lambert), a path of at
     most L vertices gathers exactly  sum_{k<L} rho^k  -- checks cosine sampling, throughput, emission accounting, Russian roulette
     and the path-length semantics of `-pl` in one number.
"""
import ctypes as C
import os

import numpy as np
import pytest

from fermat_amd import scene
from oracle import binding as ob

K_DIFF_R, K_DIFF_T, K_GLOSS_R, K_GLOSS_T = 0, 1, 2, 3                 # lobe slots of f_and_p (src/bsdf.h:133-141)
COMP_BITS = {1: K_DIFF_R, 2: K_DIFF_T, 4: K_GLOSS_R, 8: K_GLOSS_T}     # Bsdf::ComponentType bits of sample()'s out_comp
COAT = 0x10


def material(**kw):
    m = np.zeros(1, scene.MATERIAL_DTYPE)
    p = scene.default_material_params()
    p.update(kw)
    for k in ("diffuse", "diffuse_trans", "ambient", "specular", "emissive", "reflectivity"):
        m[0][k][:3] = np.float32(p[k])
    pe = np.float32(p["phong_exponent"])
    m[0]["roughness"] = np.float32(1.0) / pe if pe != 0 else np.float32(1.0)
    m[0]["index_of_refraction"] = p["index_of_refraction"]; m[0]["opacity"] = p["opacity"]
    for k in ("ambient_map", "diffuse_map", "diffuse_trans_map", "specular_map", "emissive_map", "bump_map"):
        m[0][k]["texture"] = scene.INVALID_TEXTURE; m[0][k]["scaling"] = (1.0, 1.0)
    return m


MATERIALS = {
    "matte":        dict(diffuse=[0.7, 0.6, 0.5]),
    "plastic":      dict(diffuse=[0.5, 0.5, 0.5], specular=[0.9, 0.9, 0.9], phong_exponent=4.0, index_of_refraction=1.5),
    "rough_metal":  dict(diffuse=[0.05, 0.05, 0.05], specular=[3.0, 2.5, 1.5], phong_exponent=2.5, index_of_refraction=1.5),
    "coated":       dict(diffuse=[0.6, 0.2, 0.2], specular=[0.6, 0.6, 0.6], phong_exponent=3.0, index_of_refraction=1.5, reflectivity=[0.08, 0.08, 0.08]),
    "frosted":      dict(diffuse=[0.3, 0.3, 0.3], diffuse_trans=[0.5, 0.5, 0.4], specular=[0.8, 0.8, 0.8], phong_exponent=3.0, index_of_refraction=1.3, opacity=0.5),
    "glass":        dict(diffuse=[0.1, 0.1, 0.1], specular=[0.9, 0.9, 0.9], phong_exponent=5.0, index_of_refraction=1.5, opacity=0.2),
}


def _sphere_quadrature(n_theta=384, n_phi=768):
    """midpoint lat-long rule over the whole sphere: directions and solid-angle weights"""
    ct = (np.arange(n_theta) + 0.5) / n_theta * 2.0 - 1.0                  # uniform in cos(theta): equal-area bands
    ph = (np.arange(n_phi) + 0.5) / n_phi * 2.0 * np.pi
    CT, PH = np.meshgrid(ct, ph, indexing="ij")
    st = np.sqrt(1.0 - CT * CT)
    d = np.stack([st * np.cos(PH), st * np.sin(PH), CT], -1).reshape(-1, 3).astype(np.float32)
    return d, 4.0 * np.pi / len(d)


def _lobe_integrals(olib, m, table, w_i, visible_microfacets_only=False):
    d, dw = _sphere_quadrature()
    out = np.zeros((len(d), 16), np.float32)
    olib.orc_bsdf_f_and_p_n(C.c_void_p(m.ctypes.data), C.c_void_p(table.ctypes.data), C.c_uint32(len(d)), C.c_void_p(w_i.ctypes.data),
                            C.c_void_p(d.ctypes.data), C.c_void_p(out.ctypes.data))
    f = out[:, :12].reshape(-1, 4, 3).astype(np.float64); p = out[:, 12:].astype(np.float64)
    cos = np.abs(d[:, 2].astype(np.float64))
    f = np.where(np.isfinite(f), f, 0.0); p = np.where(np.isfinite(p), p, 0.0)
    if visible_microfacets_only:
        # the refraction half-vector of (w_i, w_o), oriented towards w_i (vndf_microfacet, contrib/cugar/bsdf/ggx_common.h:68-84); the
        # VNDF sampler only ever produces microfacet normals on w_i's side of the surface (vndf_ggx_smith_sample clamps N.z >= 0,
        # ggx_common.h:287), while f_and_p evaluates D(|N.H|) and a G1 without the visibility step function for any half-vector
        V = w_i.astype(np.float64); ior = float(m[0]["index_of_refraction"])
        inv_eta = ior if V[2] >= 0 else 1.0 / ior
        Hh = V[None, :] + inv_eta * d.astype(np.float64)
        Hh *= np.where((Hh * V[None, :]).sum(1) < 0, -1.0, 1.0)[:, None]
        reach = (Hh[:, 2] * np.sign(V[2]) > 0) & (d[:, 2] * V[2] < 0)
        f[:, K_GLOSS_T] *= reach[:, None]
    return (f * cos[:, None, None]).sum(0) * dw, (p * cos[:, None]).sum(0) * dw          # integral of f_c and of p_c over projected solid angle


def _sampled_moments(olib, m, table, w_i, n=400000, seed=5):
    z = np.random.default_rng(seed).random((n, 3), dtype=np.float32)
    out = np.zeros((n, 9), np.float32)
    olib.orc_bsdf_sample_n(C.c_void_p(m.ctypes.data), C.c_void_p(table.ctypes.data), C.c_uint32(n), C.c_void_p(z.ctypes.data), C.c_void_p(w_i.ctypes.data),
                           C.c_void_p(out.ctypes.data))
    comp = out[:, 0].astype(np.uint32); g = out[:, 6:9].astype(np.float64)
    assert np.isfinite(g).all()
    mean = np.zeros((5, 3)); err = np.zeros((5, 3)); freq = np.zeros(6)
    for bit, slot in list(COMP_BITS.items()) + [(COAT, 4)]:
        sel = comp == bit
        x = g * sel[:, None]
        mean[slot] = x.mean(0); err[slot] = x.std(0) / np.sqrt(n); freq[slot] = sel.mean()
    freq[5] = (comp == 0).mean()
    return mean, err, freq, out


@pytest.mark.parametrize("name", list(MATERIALS))
@pytest.mark.parametrize("cos_i", [0.95, 0.5, 0.2])
def test_composite_bsdf_sampler_integrates_the_lobes_it_evaluates(olib, table, name, cos_i):
    """(1): per lobe, the sampler's estimator g = f / p integrates to the quadrature of f_and_p's f"""
    m = material(**MATERIALS[name])
    w_i = np.float32([np.sqrt(1.0 - cos_i * cos_i), 0.0, cos_i])
    quad_all, quad_p = _lobe_integrals(olib, m, table, w_i)
    # A property of the reference's rough-dielectric TRANSMISSION lobe this test brought out: f_and_p is positive for outgoing directions
    # whose refraction half-vector faces away from w_i's side of the surface (it evaluates D(|N.H|) and a Smith G1 with no visibility
    # step, contrib/cugar/bsdf/ggx_smith.h:430-476), directions the VNDF sampler can never produce.  So next-event estimation sees energy
    # BSDF sampling does not -- up to 30 % of the lobe at grazing incidence -- and the sampler integrates exactly the rest:
    quad_f, _ = _lobe_integrals(olib, m, table, w_i, visible_microfacets_only=True)
    mean, err, freq, _ = _sampled_moments(olib, m, table, w_i)
    for slot in range(4):
        # MC error + quadrature error of the narrower lobes (the reference's own point-wise test grants 3 %, bsdf_test.h:121-141)
        rel = 0.04 if slot == K_GLOSS_T else 0.015          # the transmission lobe keeps a residual of up to 3 % once the unreachable part is masked
        tol = 5.0 * err[slot] + rel * np.abs(quad_f[slot]) + 2e-4
        assert (np.abs(mean[slot] - quad_f[slot]) <= tol).all(), (name, cos_i, slot, mean[slot], quad_f[slot], err[slot])
        assert (quad_all[slot] >= quad_f[slot] - 1e-9).all()
    # the lobes' a-priori pdfs (what NEE's MIS weight uses, src/pathtracer_core.h:1055-1059) are sub-normalised densities
    # (the transmission lobe's density also covers its unreachable directions, so it may exceed one by the same margin)
    assert (quad_p >= -1e-6).all() and quad_p.sum() <= (1.0 + 5e-3 if m[0]["opacity"] == 1.0 else 1.10)
    # a lobe that can never be sampled must carry no energy
    for slot in range(4):
        if freq[slot] == 0.0:
            assert np.abs(quad_f[slot]).max() < 1e-4, (name, slot, quad_f[slot])


@pytest.mark.parametrize("cos_i", [1.0, 0.7, 0.3, 0.1])
@pytest.mark.parametrize("rough", [0.1, 0.4, 1.0])
def test_white_furnace_directional_albedo_is_at_most_one(olib, table, rough, cos_i):
    """(2): a white diffuse base under a white glossy layer (and under a clearcoat) reflects no more than it receives"""
    for kw in (dict(diffuse=[1, 1, 1], specular=[np.pi, np.pi, np.pi], phong_exponent=1.0 / rough, index_of_refraction=1.5),
               dict(diffuse=[1, 1, 1], specular=[np.pi, np.pi, np.pi], phong_exponent=1.0 / rough, index_of_refraction=1.5, reflectivity=[0.3, 0.3, 0.3]),
               dict(diffuse=[1, 1, 1], diffuse_trans=[1, 1, 1], specular=[1, 1, 1], phong_exponent=1.0 / rough, index_of_refraction=1.5, opacity=0.5)):
        m = material(**kw)
        w_i = np.float32([np.sqrt(max(0.0, 1.0 - cos_i * cos_i)), 0.0, cos_i])
        mean, err, freq, _ = _sampled_moments(olib, m, table, w_i, n=200000)
        albedo = mean.sum(0)
        # transmission through an interface scales radiance by the squared relative index (compression_factor, src/bsdf.h:1237-1251);
        # undo it for the energy balance: transmitted lobes carry eta^2 x the flux fraction
        flux = mean[[K_DIFF_R, K_GLOSS_R, 4]].sum(0) + mean[[K_DIFF_T, K_GLOSS_T]].sum(0) / (1.5 ** 2 if kw.get("opacity", 1.0) < 1.0 else 1.0)
        assert (flux <= 1.0 + 4.0 * err.sum(0) + 0.02).all(), (kw, cos_i, flux, albedo)
        assert abs(freq.sum() - 1.0) < 1e-6 and (albedo >= 0).all()


def _areas(s):
    v = s.vertex_data[:, :3].astype(np.float64); t = s.vertex_indices[:, :3]
    return 0.5 * np.linalg.norm(np.cross(v[t[:, 1]] - v[t[:, 0]], v[t[:, 2]] - v[t[:, 0]]), axis=1)


def test_emitter_tables_sample_with_the_density_they_report(table):
    """(3): triangle frequencies of the mesh CDF and of the VPL set = max(Ke) * area / norm, computed here from the scene arrays"""
    s = scene.cornell_box("CornellBox-Glossy")
    o = ob.OraclePT(s, 256, 256, ob.default_options(4), table, scene.DATA_DIR)
    lt = o.lights()
    area = _areas(s)
    ke = np.abs(s.materials["emissive"][s.material_indices][:, :3]).max(1).astype(np.float64)
    want = ke * area
    assert want.sum() > 0
    # mesh mode: cdf differences are the emission-weighted areas; pdf x area sums to one
    cdf = lt["mesh_cdf"].astype(np.float64); p_tri = np.diff(np.concatenate([[0.0], cdf]))
    assert np.allclose(p_tri, want / want.sum(), atol=2e-6) and abs(cdf[-1] - 1.0) < 1e-6
    pdf_area = p_tri * lt["mesh_inv_area"].astype(np.float64)
    assert np.allclose((pdf_area * area)[want > 0].sum(), 1.0, atol=1e-5)
    assert np.allclose(lt["mesh_inv_area"][want > 0], 1.0 / area[want > 0], rtol=1e-5)
    # VPL mode: the reported density is max(Ke) / norm per unit area whatever VPL was drawn (src/lights.h:59-76,415);
    # it integrates to one over the emitters, and the VPL set's triangle histogram follows it (chi-square, 5 sigma)
    assert abs((want / lt["norm"]).sum() - 1.0) < 2e-3      # norm = fp32 mean of 65 536 per-VPL estimates
    prim = lt["vpls"]["prim_id"]; n = len(prim)
    counts = np.bincount(prim, minlength=s.num_triangles).astype(np.float64)
    assert counts[want == 0].sum() == 0
    e = n * want / want.sum(); em = e > 0
    chi2 = (((counts - e) ** 2)[em] / e[em]).sum(); dof = em.sum() - 1
    assert chi2 < dof + 5.0 * np.sqrt(2.0 * dof) + 10.0, (chi2, dof)
    # VPL barycentrics are uniform on the triangle: mean (1/3, 1/3), inside the triangle
    uv = lt["vpls"]["uv"].astype(np.float64)
    assert (uv >= 0).all() and (uv.sum(1) <= 1.0 + 1e-6).all() and np.abs(uv.mean(0) - 1.0 / 3.0).max() < 5e-3


def _proper(o):
    """every path counted once: DIRECT_C (bounce 0 emission) + DIFFUSE_C + SPECULAR_C (the -pt COMPOSITED channel counts indirect NEE
    twice, a reference quirk kept on purpose: src/pathtracer_vertex_processor.h:103-104,224)"""
    return (o.fb[4][:, :3] + o.fb[0][:, :3] + o.fb[2][:, :3]).astype(np.float64)


def _pt(s, table, W, H, L, n, nee_type=1, **kw):
    opt = ob.default_options(L, nee_type)
    for k, v in kw.items():
        setattr(opt, k, v)
    o = ob.OraclePT(s, W, H, opt, table, scene.DATA_DIR)
    o.set_trace_threads(os.cpu_count() or 1)
    for i in range(n):
        o.render_pass(i)
    return o


@pytest.mark.parametrize("name", ["CornellBox-JP", "CornellBox-Glossy"])
def test_three_estimators_of_one_integral_agree(table, name):
    """(4): BSDF sampling only == NEE only == MIS of both (VPL and mesh emitter sampling) == BPT, on image means and on 4x4 blocks"""
    s = scene.cornell_box(name)
    W, H, L, n = 32, 32, 4, 384
    est = {
        "bsdf": _pt(s, table, W, H, L, n, direct_lighting_nee=0, indirect_lighting_nee=0),
        "nee_vpl": _pt(s, table, W, H, L, n, 1, direct_lighting_bsdf=0, indirect_lighting_bsdf=0),
        "nee_mesh": _pt(s, table, W, H, L, n, 0, direct_lighting_bsdf=0, indirect_lighting_bsdf=0),
        "mis_vpl": _pt(s, table, W, H, L, n, 1),
        "mis_mesh": _pt(s, table, W, H, L, n, 0),
    }
    img = {k: _proper(o).reshape(H, W, 3) for k, o in est.items()}
    for key, kw in (("bpt_no_light_tracing", dict(light_tracing=0.0)), ("bpt_sc1_no_light_tracing", dict(light_tracing=0.0, single_connection=1)), ("bpt", dict()),
                    ("bpt_whatif_true_distance", dict()), ("bpt_whatif_consistent", dict())):
        o = ob.OraclePT(s, W, H, ob.default_options(L), table, scene.DATA_DIR)
        o.bpt_init(ob.default_bpt_options(L, **kw), scene.DATA_DIR)
        o.bpt_set_whatif({"bpt_whatif_true_distance": 1, "bpt_whatif_consistent": 3}.get(key, 0))
        for i in range(n):
            o.bpt_render(i)
        img[key] = o.fb[5][:, :3].astype(np.float64).reshape(H, W, 3)        # the BPT's COMPOSITED channel has no double counting
    ref = img["mis_vpl"]
    blocks = lambda a: a.reshape(H // 4, 4, W // 4, 4, 3).mean((1, 3))
    for k, a in img.items():
        assert np.isfinite(a).all()
        # With light tracing on (`-lt 1`, the default) the restated BPT is 4-9 % brighter than the five path-tracing estimators and than itself
        # without light tracing (which agree to <1 %).  ORIGIN: two places where the reference's MIS weights are not the same function
        # on the light-tracing and on the eye side -- (1) EyeVertex::setup divides G' by hit.t^2 with hit.t in units of the UN-NORMALISED primary ray
        # (src/bpt_utils.h:636, src/bpt_kernels.h:569-572): the eye strategies under-price the lens connection by cos^2 of the camera angle (brighter);
        # (2) connect_to_camera prices its neighbour with max_comp(f_L) instead of the reverse pdf (src/bpt_kernels.h:1009; darker).  The oracle's
        # test-only what-if switches put the consistent quantities there: with both, light tracing on agrees with everything else to <1 %.
        # (`-sc 1` draws ONE light vertex of any depth per eye vertex: unbiased, noisier, and it also forms paths beyond max_path_length)
        tol_mean, tol_blocks = (0.11, 0.2) if k == "bpt" else ((0.04, 0.2) if "sc1" in k else ((0.05, 0.12) if k == "bpt_whatif_true_distance" else (0.03, 0.12)))
        assert abs(a.mean() / ref.mean() - 1.0) < tol_mean, (k, a.mean(), ref.mean())
        d = np.abs(blocks(a) - blocks(ref)).mean() / blocks(ref).mean()
        assert d < tol_blocks, (k, d)
    r = {k: img[k].mean() / ref.mean() for k in ("bpt", "bpt_whatif_true_distance", "bpt_whatif_consistent", "bpt_no_light_tracing")}
    assert r["bpt"] > 1.03                                                       # the reference's excess ...
    assert r["bpt_whatif_true_distance"] < r["bpt_no_light_tracing"] < r["bpt"]  # ... is (1); what is left below is (2) ...
    assert abs(r["bpt_whatif_consistent"] - r["bpt_no_light_tracing"]) < 0.01    # ... and with both, light tracing changes nothing
    print("\n[%s] image mean / PT(MIS): %s" % (name, ", ".join("%s %.4f" % kv for kv in r.items())))


def _closed_furnace(tmp_path, rho, ke=1.0, inward=True):
    """a closed axis-aligned box [-1,1]^3 whose six walls emit `ke` and reflect `rho` (Lambert); the EDF emits on the side the
    shading normal points to (src/edf.h:49-65), so every face is wound to face the interior"""
    d = str(tmp_path)
    with open(os.path.join(d, "furnace.mtl"), "w") as f:
        f.write("newmtl wall\nKd %g %g %g\nKs 0 0 0\nKe %g %g %g\n" % (rho, rho, rho, ke, ke, ke))
    corners = np.array([[x, y, z] for x in (-1, 1) for y in (-1, 1) for z in (-1, 1)], np.float64)       # index = 4*(x>0) + 2*(y>0) + (z>0)
    quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    with open(os.path.join(d, "furnace.obj"), "w") as f:
        f.write("mtllib furnace.mtl\n")
        for c in corners:
            f.write("v %g %g %g\n" % tuple(c))
        f.write("usemtl wall\n")
        for q in quads:
            v = corners[list(q)]
            n = np.cross(v[1] - v[0], v[2] - v[0])
            facing_in = np.dot(n, -v.mean(0)) > 0
            if facing_in != inward:
                q = q[::-1]
            f.write("f %d %d %d %d\n" % tuple(i + 1 for i in q))
    s = scene.load_scene(os.path.join(d, "furnace.obj"))
    s.camera = scene.make_camera([0.1, -0.05, 0.2], [0.3, 0.2, -1.0], [0, 1, 0], 1.2)
    return s


@pytest.mark.parametrize("rho,L", [(0.5, 6), (0.8, 4), (0.3, 9)])
def test_closed_furnace_has_the_known_answer(tmp_path, table, rho, L):
    """(5): inside a closed emitting Lambertian box every path of at most L vertices gathers sum_{k<L} rho^k"""
    s = _closed_furnace(tmp_path, rho)
    want = sum(rho ** k for k in range(L))
    W, H, n = 16, 16, 256
    a = _pt(s, table, W, H, L, n, direct_lighting_nee=0, indirect_lighting_nee=0)
    got = a.fb[5][:, :3].astype(np.float64)                                    # no NEE -> COMPOSITED counts every path once
    assert abs(got.mean() / want - 1.0) < 0.01, (got.mean(), want)
    assert np.abs(got.reshape(-1, 3).mean(0) / want - 1.0).max() < 0.01
    # with next-event estimation + MIS (both emitter samplers) the properly counted channels give the same number
    for nee in (1, 0):
        b = _pt(s, table, W, H, L, n, nee)
        assert abs(_proper(b).mean() / want - 1.0) < 0.015, (nee, _proper(b).mean(), want)
    # and the bidirectional tracer: the same number (measured +0.3 ... +1.3 % with every technique on)
    o = ob.OraclePT(s, W, H, ob.default_options(L), table, scene.DATA_DIR)
    o.bpt_init(ob.default_bpt_options(L), scene.DATA_DIR)
    for i in range(n):
        o.bpt_render(i)
    assert abs(o.fb[5][:, :3].astype(np.float64).mean() / want - 1.0) < 0.02


@pytest.mark.parametrize("name", ["CornellBox-JP", "CornellBox-Glossy"])
def test_psfpt_estimator_against_the_path_tracer(table, name):
    """the PSFPT vertex processor (src/psfpt_vertex_processor.h) against the path tracer's channels that count every path once, on the pixels
    that do not see the emitter (its radiance, up to 200, is clamped by firefly_filter = 100 and clamp_frame(100): 39 % of the Glossy box's
    image energy), firefly clamp off.
      * no cache vertex (psf_depth beyond the path length): the same estimator, means agree to 1 %;
      * cells of one path each (psf_width 0.001: no filtering): the restated PSFPT is 3.6 % / 4.3 % darker.  ORIGIN: shade_vertex hands the
        shadow sample `vertex_info` where compute_nee_weights computed `out_vertex_info` (src/pathtracer_core.h:984,1102 vs
        src/psfpt_vertex_processor.h:231-236), so at a NEW cache vertex accumulate_nee sees comp 0 instead of DIFFUSE_COMP and folds the
        un-demodulated glossy NEE term into the cell, where the blend multiplies it by w * diffuse again.  The oracle's test-only what-if
        switch passes out_vertex_info: the loss is gone (1.0000 / 0.9999);
      * default cell size: 4.8 % / 7.9 % darker as the reference is; 0.6 % / 4.2 % with the what-if -- that part is the bias of filtering."""
    s = scene.cornell_box(name)
    W, H, L, n = 48, 48, 5, 128
    ref = _proper(_pt(s, table, W, H, L, n))
    m = ref.max(1) < 20.0
    assert m.sum() > 0.9 * W * H
    got = {}
    for key, kw, whatif in (("no_cache", dict(psf_depth=1000), 0), ("one_path_cells", dict(psf_width=0.001), 0), ("one_path_cells_whatif", dict(psf_width=0.001), 1),
                            ("default", dict(), 0), ("default_whatif", dict(), 1)):
        o = ob.OraclePT(s, W, H, ob.default_options(L), table, scene.DATA_DIR)
        o.set_trace_threads(os.cpu_count() or 1)
        o.psf_enable(ob.default_psf_options(firefly_filter=1e8, **kw))
        o.psf_set_whatif(whatif)
        for i in range(n):
            o.render_pass(i)
        got[key] = o.fb[5][:, :3].astype(np.float64)[m].mean() / ref[m].mean()
        assert (len(o.psf_cells()["keys"]) == 0) == (key == "no_cache")
    print("\n[%s] PSFPT / PT on non-emitter pixels: %s" % (name, ", ".join("%s %.4f" % kv for kv in got.items())))
    assert abs(got["no_cache"] - 1.0) < 0.01, got
    assert 0.93 < got["one_path_cells"] < 0.985, got                  # the reference's loss ...
    assert abs(got["one_path_cells_whatif"] - 1.0) < 0.01, got        # ... is the vertex_info / out_vertex_info mix-up
    assert 0.88 < got["default"] < got["default_whatif"] < 1.0, got   # what is left with the what-if is the bias of filtering
