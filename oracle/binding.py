"""ORACLE — TEST INFRASTRUCTURE ONLY.

ctypes binding of oracle/liboracle.so.  Importable only from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; the product package (fermat_amd/) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        try:
            build()                      # a no-op when liboracle.so is newer than its sources; never run a stale checker
        except (OSError, subprocess.CalledProcessError):
            if not os.path.exists(so):
                raise
        L = C.CDLL(so)
        for name, res in (("orc_randfloat", C.c_float), ("orc_hash", C.c_uint32), ("orc_permute", C.c_uint32), ("orc_f2h", C.c_uint16),
                          ("orc_h2f", C.c_float), ("orc_pack_normal", C.c_uint32), ("orc_det_atan2", C.c_float), ("orc_det_pow", C.c_float),
                          ("orc_f2u", C.c_uint32), ("orc_quantize", C.c_uint32), ("orc_morton60", C.c_uint64), ("orc_pt_create", C.c_void_p),
                          ("orc_pt_stats", C.c_uint32), ("orc_pt_get_captured", C.c_uint32), ("orc_pt_n_dims", C.c_uint32),
                          ("orc_pt_sample_2d", C.c_float), ("orc_pt_get_lights", C.c_uint32), ("orc_pt_bvh_info", C.c_uint32)):
            getattr(L, name).restype = res
        _LIB = L
    return _LIB


class Texture(C.Structure):
    _fields_ = [("texels", C.c_void_p), ("res_x", C.c_uint32), ("res_y", C.c_uint32)]


class SceneDesc(C.Structure):
    _fields_ = [("num_triangles", C.c_int32), ("num_vertices", C.c_int32), ("num_materials", C.c_int32), ("num_textures", C.c_int32),
                ("vertex_indices", C.c_void_p), ("vertex_data", C.c_void_p), ("texture_indices_comp", C.c_void_p),
                ("material_indices", C.c_void_p), ("materials", C.c_void_p), ("textures", C.c_void_p), ("dir_lights", C.c_void_p),
                ("glossy_reflectance", C.c_void_p), ("texture_data", C.c_void_p), ("tex_bias", C.c_float * 2), ("tex_scale", C.c_float * 2), ("camera", C.c_float * 13),
                ("dir_lights_count", C.c_int32), ("res_x", C.c_uint32), ("res_y", C.c_uint32),
                ("aspect", C.c_float), ("exposure", C.c_float), ("gamma", C.c_float)]


class PTOptions(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("max_path_length", "direct_lighting", "direct_lighting_nee", "direct_lighting_bsdf",
                                          "indirect_lighting_nee", "indirect_lighting_bsdf", "visible_lights", "diffuse_scattering",
                                          "glossy_scattering", "indirect_glossy", "rr", "nee_type")]


class PSFOptions(C.Structure):
    """PSFPTOptions beyond PTOptions (src/renderers/psfpt.h:39-78)"""
    _fields_ = [("psf_depth", C.c_uint32), ("psf_width", C.c_float), ("psf_min_dist", C.c_float), ("psf_max_prob", C.c_float),
                ("psf_temporal_reuse", C.c_uint32), ("firefly_filter", C.c_float)]


def default_psf_options(**kw):
    o = PSFOptions(1, 3.0, 0.1, 32.0, 64, 100.0)
    for k, v in kw.items():
        setattr(o, k, v)
    return o


class BPTOptions(C.Structure):
    """BPTOptionsBase + BPTOptions::rr / single_connection (src/bpt_options.h:42-66, src/renderers/bpt.h:47-72)"""
    _fields_ = [("max_path_length", C.c_uint32), ("direct_lighting_nee", C.c_uint32), ("direct_lighting_bsdf", C.c_uint32),
                ("indirect_lighting_nee", C.c_uint32), ("indirect_lighting_bsdf", C.c_uint32), ("visible_lights", C.c_uint32),
                ("use_vpls", C.c_uint32), ("rr", C.c_uint32), ("light_tracing", C.c_float), ("single_connection", C.c_uint32)]


def default_bpt_options(max_path_length=6, **kw):
    """the tests' default is the all-connections mode (single_connection=0); the reference's own default (`-sc 1`) is single_connection=1"""
    o = BPTOptions(max_path_length, 1, 1, 1, 1, 1, 0, 1, 1.0, 0)
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def default_options(max_path_length=6, nee_type=1):
    """PTOptions defaults (src/renderers/pathtracer.h:186-199)."""
    return PTOptions(max_path_length, 1, 1, 1, 1, 1, 1, 1, 1, 0, 1, nee_type)


RAY_DTYPE = np.dtype([("origin", "<f4", (3,)), ("mask", "<u4"), ("dir", "<f4", (3,)), ("tmax", "<f4")])
HIT_DTYPE = np.dtype([("t", "<f4"), ("triId", "<i4"), ("u", "<f4"), ("v", "<f4")])
PATH_ENTRY_DTYPE = np.dtype([("ray", RAY_DTYPE), ("hit", HIT_DTYPE), ("weight", "<f4", (4,)), ("pixel_info", "<u4"), ("cone", "<f4", (2,)), ("vertex_info", "<u4")])
VPL_DTYPE = np.dtype([("uv", "<f4", (2,)), ("prim_id", "<u4"), ("E", "<f4")])
STATS_DTYPE = np.dtype([("in_size", "<u4"), ("shadow_dir_size", "<u4"), ("shadow_size", "<u4"), ("scatter_size", "<u4")])
assert RAY_DTYPE.itemsize == 32 and HIT_DTYPE.itemsize == 16 and PATH_ENTRY_DTYPE.itemsize == 80


class OraclePT:
    """One oracle path tracer bound to a pre-processed fermat_amd.scene.Scene (arrays are kept alive here)."""

    def __init__(self, scene, res_x, res_y, options=None, table=None, samples_dir=None, n_vpls=None, exposure=1.0, gamma=2.2):
        L = lib()
        self.scene = scene
        self.res = (res_x, res_y)
        self.options = options or default_options()
        self._keep = []
        d = SceneDesc()
        d.num_triangles = scene.num_triangles; d.num_vertices = scene.num_vertices
        d.num_materials = len(scene.materials); d.num_textures = len(scene.textures)
        self._arr = dict(vi=scene.vertex_indices, vd=scene.vertex_data, mi=scene.material_indices, mats=scene.materials,
                         table=np.ascontiguousarray(table, np.float32), dl=np.ascontiguousarray(scene.dir_lights, np.float32))
        d.vertex_indices = self._arr["vi"].ctypes.data; d.vertex_data = self._arr["vd"].ctypes.data
        d.material_indices = self._arr["mi"].ctypes.data; d.materials = self._arr["mats"].ctypes.data
        if scene.texture_indices_comp is not None:
            d.texture_indices_comp = scene.texture_indices_comp.ctypes.data
        if getattr(scene, "texture_data", None) is not None:
            d.texture_data = scene.texture_data.ctypes.data
        tex = (Texture * max(1, len(scene.textures)))()
        for i, t in enumerate(scene.textures):
            if t is not None:
                t = np.ascontiguousarray(t, np.float32); self._keep.append(t)
                tex[i].texels = t.ctypes.data; tex[i].res_x = t.shape[1]; tex[i].res_y = t.shape[0]
        self._tex = tex
        d.textures = C.addressof(tex)
        d.dir_lights = self._arr["dl"].ctypes.data; d.dir_lights_count = len(scene.dir_lights)
        d.glossy_reflectance = self._arr["table"].ctypes.data
        d.tex_bias = (C.c_float * 2)(*scene.tex_bias); d.tex_scale = (C.c_float * 2)(*scene.tex_scale)
        d.camera = (C.c_float * 13)(*scene.camera)
        d.res_x = res_x; d.res_y = res_y; d.aspect = np.float32(res_x) / np.float32(res_y); d.exposure = exposure; d.gamma = gamma
        self._desc = d
        sd = samples_dir.encode() if samples_dir else None
        self.h = C.c_void_p(L.orc_pt_create(C.byref(d), C.byref(self.options), sd, C.c_uint32(res_x * res_y if n_vpls is None else n_vpls)))
        n = res_x * res_y
        self.fb = np.zeros((8, n, 4), np.float32)
        self.gb_geo = np.zeros((n, 4), np.float32); self.gb_uv = np.zeros((n, 4), np.float32)
        self.gb_tri = np.full(n, 0xFFFFFFFF, np.uint32); self.gb_depth = np.zeros(n, np.float32)
        ch = (C.c_void_p * 8)(*[self.fb[c].ctypes.data for c in range(8)])
        L.orc_pt_set_framebuffer(self.h, ch, C.c_void_p(self.gb_geo.ctypes.data), C.c_void_p(self.gb_uv.ctypes.data),
                                 C.c_void_p(self.gb_tri.ctypes.data), C.c_void_p(self.gb_depth.ctypes.data))

    def __del__(self):
        try:
            lib().orc_pt_destroy(self.h)
        except Exception:
            pass

    def render_pass(self, instance, pixels=None):
        L = lib()
        if pixels is None:
            L.orc_pt_render_pass(self.h, C.c_uint32(instance), None, C.c_uint32(self.res[0] * self.res[1]))
        else:
            pixels = np.ascontiguousarray(pixels, np.uint32)
            L.orc_pt_render_pass(self.h, C.c_uint32(instance), C.c_void_p(pixels.ctypes.data), C.c_uint32(len(pixels)))

    def stats(self):
        out = np.zeros(64, STATS_DTYPE)
        n = lib().orc_pt_stats(self.h, C.c_void_p(out.ctypes.data), C.c_uint32(64))
        return out[:n]

    def set_capture(self, bounce):
        lib().orc_pt_set_capture(self.h, C.c_int32(bounce))

    def captured(self):
        n = lib().orc_pt_get_captured(self.h, None, C.c_uint32(0))
        out = np.zeros(n, PATH_ENTRY_DTYPE)
        if n:
            lib().orc_pt_get_captured(self.h, C.c_void_p(out.ctypes.data), C.c_uint32(n))
        return out

    def trace(self, rays, shadow=False, n_threads=1):
        rays = np.ascontiguousarray(rays, RAY_DTYPE)
        hits = np.zeros(len(rays), HIT_DTYPE)
        fn = lib().orc_pt_trace_shadow if shadow else lib().orc_pt_trace
        fn(self.h, C.c_uint32(len(rays)), C.c_void_p(rays.ctypes.data), C.c_void_p(hits.ctypes.data), C.c_int32(n_threads))
        return hits

    def counters(self):
        out = (C.c_uint64 * 4)()
        lib().orc_pt_counters(self.h, out)
        return list(out)

    def sequence(self, instance=None):
        L = lib()
        if instance is not None:
            L.orc_pt_set_instance(self.h, C.c_uint32(instance))
        nd = L.orc_pt_n_dims(self.h)
        shifts = np.zeros((nd, 65536), np.float32); samples = np.zeros((nd, 65536), np.float32)
        L.orc_pt_get_sequence(self.h, C.c_void_p(shifts.ctypes.data), C.c_void_p(samples.ctypes.data))
        return shifts, samples

    def lights(self):
        L = lib()
        nt = self.scene.num_triangles
        n = L.orc_pt_get_lights(self.h, None, None, None, None, None)
        vpls = np.zeros(n, VPL_DTYPE); cdf = np.zeros(n, np.float32)
        mcdf = np.zeros(nt, np.float32); minv = np.zeros(nt, np.float32); norm = C.c_float()
        L.orc_pt_get_lights(self.h, C.c_void_p(vpls.ctypes.data) if n else None, C.c_void_p(cdf.ctypes.data) if n else None,
                            C.c_void_p(mcdf.ctypes.data), C.c_void_p(minv.ctypes.data), C.byref(norm))
        return dict(vpls=vpls, vpl_cdf=cdf, mesh_cdf=mcdf, mesh_inv_area=minv, norm=norm.value)

    def to_rgba(self, mode=None):
        """to_rgba_kernel (src/renderer.cu:83-282); mode = ShadingMode (0 kShaded ... 10 kFiltered, 11 kVariance, 12 kNormal)"""
        out = np.zeros((self.res[1], self.res[0], 4), np.uint8)
        if mode is None:
            lib().orc_pt_to_rgba(self.h, C.c_void_p(out.ctypes.data))
        else:
            lib().orc_pt_to_rgba_mode(self.h, C.c_uint32(mode), C.c_void_p(out.ctypes.data))
        return out

    # -- path-space filtering (oracle/o_psfpt.h): render_pass then runs the PSFPT vertex processor
    def psf_enable(self, options):
        self.psf_options = options
        lib().orc_psf_enable(self.h, C.byref(options))

    def psf_set_whatif(self, bits):
        """statistical tests only: 1 = shadow samples carry compute_nee_weights' out_vertex_info (0 = the reference: vertex_info)"""
        lib().orc_psf_set_whatif(self.h, C.c_uint32(bits))

    def psf_cells(self):
        n = lib().orc_psf_get_cells(self.h, None, None, None, C.c_uint32(0))
        keys = np.zeros(n, np.uint64); counts = np.zeros(n, np.uint64); sums = np.zeros((n, 3), np.int64)
        if n:
            lib().orc_psf_get_cells(self.h, C.c_void_p(keys.ctypes.data), C.c_void_p(counts.ctypes.data), C.c_void_p(sums.ctypes.data), C.c_uint32(n))
        return dict(keys=keys, counts=counts, sums=sums)

    def psf_ref_count(self):
        return int(lib().orc_psf_ref_count(self.h))

    # -- bidirectional path tracer (oracle/o_bpt.h), sharing this context's scene / BVH / lights / frame buffer
    def bpt_init(self, options, samples_dir):
        self.bpt_options = options
        lib().orc_bpt_init(self.h, C.byref(options), samples_dir.encode())

    def bpt_set_whatif(self, bits):
        """statistical tests only: 1 = true distance in the first eye vertex's G', 2 = reverse pdf in connect_to_camera (0 = the reference)"""
        lib().orc_bpt_set_whatif(self.h, C.c_uint32(bits))

    def bpt_render(self, instance, pixels=None):
        if pixels is None:
            lib().orc_bpt_render(self.h, C.c_uint32(instance))
        else:
            px = np.ascontiguousarray(pixels, np.uint32); self._bpt_px = px
            lib().orc_bpt_render_pixels(self.h, C.c_uint32(instance), C.c_void_p(px.ctypes.data), C.c_uint32(len(px)))

    def bpt_defer_splats(self):
        """returns a writable (n_pixels, 6) int64 view of the light-tracing splat sums; call bpt_resolve_splats after summing over ranks"""
        lib().orc_bpt_set_deferred_splats(self.h, C.c_int32(1))
        f = lib().orc_bpt_splats; f.restype = C.c_void_p
        n = self.res[0] * self.res[1]
        return np.frombuffer((C.c_longlong * (n * 6)).from_address(f(self.h)), dtype=np.int64).reshape(n, 6)

    def bpt_resolve_splats(self):
        lib().orc_bpt_resolve_splats(self.h)

    def bpt_stats(self):
        a = np.zeros(100, np.uint32)
        lib().orc_bpt_get_stats(self.h, C.c_void_p(a.ctypes.data))
        nl, ne = int(a[98]), int(a[99])
        return dict(light_queue=a[:nl].copy(), eye_queue=a[32:32 + ne].copy(), shadow_eye=a[64:64 + ne].copy(), n_light_vertices=int(a[96]), shadow_light_tracing=int(a[97]))

    def bpt_light_vertices(self):
        n = self.res[0] * self.res[1]; L = self.bpt_options.max_path_length; nv = n * L
        pos = np.zeros((nv, 4), np.float32); inp = np.zeros((nv, 2), np.uint32); gb = np.zeros((nv, 4), np.uint32)
        w = np.zeros((nv, 2), np.float32); pid = np.zeros(nv, np.uint32); cnt = np.zeros(n, np.uint32)
        lib().orc_bpt_get_light_vertices(self.h, *[C.c_void_p(x.ctypes.data) for x in (pos, inp, gb, w, pid, cnt)])
        return dict(pos=pos, input=inp, gbuffer=gb, weights=w, path_id=pid, counts=cnt)

    def set_trace_threads(self, n):
        """host threads for the BVH traces inside render_pass (results do not depend on it)"""
        lib().orc_pt_set_trace_threads(self.h, C.c_int32(int(n)))

    def trace_seconds(self):
        f = lib().orc_pt_trace_seconds; f.restype = C.c_double
        return float(f(self.h))

    def shade_seconds(self):
        f = lib().orc_pt_shade_seconds; f.restype = C.c_double
        return float(f(self.h))

    def clear_gbuffer(self):
        """GBufferStorage::clear (src/framebuffer.h:178-185): 0xFF fill"""
        for a in (self.gb_geo, self.gb_uv, self.gb_tri, self.gb_depth):
            a.view(np.uint8)[...] = 0xFF

    def filter(self, instance):
        """RenderingContextImpl::filter (src/renderer.cu:1099-1151): FILTERED_C = DIRECT_C + EAW(diffuse) + EAW(specular)"""
        lib().orc_pt_filter(self.h, C.c_uint32(instance))


def filter_variance(img, fw):
    """filter_variance_kernel (src/renderer.cu:366-399) on an (H, W, 4) float32 image -> (H, W) variance"""
    img = np.ascontiguousarray(img, np.float32); h, w = img.shape[:2]
    var = np.zeros((h, w), np.float32)
    lib().orc_filter_variance(C.c_uint32(w), C.c_uint32(h), C.c_void_p(img.ctypes.data), C.c_void_p(var.ctypes.data), C.c_uint32(fw))
    return var


def eaw_step(dst, op, w_img, w_min, img, gb_geo, var, params, step):
    """one EAW step (src/eaw.cu:45-252); op < 0 = EAW_kernel, else EAW_mad_kernel with FilterOp bits; returns the new dst"""
    img = np.ascontiguousarray(img, np.float32); h, w = img.shape[:2]
    dst = np.ascontiguousarray(dst, np.float32).copy()
    w_img = np.ascontiguousarray(w_img if w_img is not None else np.ones_like(img), np.float32)
    gb_geo = np.ascontiguousarray(gb_geo, np.float32)
    p = np.ascontiguousarray(params, np.float32)
    v = np.ascontiguousarray(var, np.float32) if var is not None else None
    lib().orc_eaw_step(C.c_uint32(w), C.c_uint32(h), C.c_void_p(dst.ctypes.data), C.c_int(op), C.c_void_p(w_img.ctypes.data), C.c_float(w_min),
                       C.c_void_p(img.ctypes.data), C.c_void_p(gb_geo.ctypes.data), C.c_void_p(v.ctypes.data) if v is not None else None,
                       C.c_void_p(p.ctypes.data), C.c_uint32(step))
    return dst
