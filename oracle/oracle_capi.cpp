// ORACLE — TEST INFRASTRUCTURE ONLY (see o_math.h header).
//
// C entry points for ctypes (tests/, __graft_entry__.smoke(), bench.py cpu_baseline leg).
// The product (fermat_amd/, include/) never includes, links or loads this file.
#include "o_pt.h"
#include "o_lights.h"
#include "o_filter.h"
#include "o_bpt.h"
#include <cstdlib>
#include <string>
#ifdef _OPENMP
#include <omp.h>
#endif

using namespace orc;

extern "C" {

struct orc_texture { const float* texels; u32 res_x, res_y; };

struct orc_scene_desc
{
	i32 num_triangles, num_vertices, num_materials, num_textures;
	const i32*   vertex_indices;
	const float* vertex_data;
	const i32*   texture_indices_comp;
	const i32*   material_indices;
	const Material* materials;
	const orc_texture* textures;
	const float* dir_lights;          // 6 floats per light: dir.xyz, color.xyz
	const float* glossy_reflectance;  // 32^4 floats
	const float* texture_data;        // float2 per vertex (MeshView::texture_data after unify) or NULL
	float tex_bias[2], tex_scale[2];
	float camera[13];                 // eye, aim, up, dx, fov  (src/camera.h:46-52)
	i32 dir_lights_count;
	u32 res_x, res_y;
	float aspect, exposure, gamma;
};

struct orc_pt
{
	PathTracer pt;
	BPT bpt;
	PsfState psf;
	std::vector<Texture> textures;
	std::vector<DirectionalLight> dir_lights;
	MeshLightsStorage lights;
};

// ---- math-layer probes (known-answer / property tests) --------------------------------------------------------------
float    orc_randfloat(u32 i, u32 p) { return randfloat(i, p); }
u32      orc_hash(u32 a) { return hash(a); }
u32      orc_permute(u32 i, u32 l, u32 p) { return permute(i, l, p); }
uint16_t orc_f2h(float f) { return f2h(f); }
float    orc_h2f(uint16_t h) { return h2f(h); }
u32      orc_pack_normal(float x, float y, float z) { return pack_normal(V3(x, y, z)); }
void     orc_unpack_normal(u32 p, float* o) { const V3 n = unpack_normal(p); o[0] = n.x; o[1] = n.y; o[2] = n.z; }
void     orc_det_sincos(float x, float* s, float* c) { det_sincos(x, s, c); }
float    orc_det_atan2(float y, float x) { return det_atan2(y, x); }
float    orc_det_pow(float x, float y) { return det_pow(x, y); }
u32      orc_f2u(float x) { return f2u(x); }
u32      orc_quantize(float x, u32 n) { return quantize(x, n); }
u64      orc_morton60(u32 x, u32 y, u32 z) { return morton60(x, y, z); }
void     orc_orthogonal(const float* v, float* o) { const V3 r = orthogonal(V3(v[0], v[1], v[2])); o[0] = r.x; o[1] = r.y; o[2] = r.z; }
void     orc_square_to_cosine_hemisphere(float u, float v, float* o) { const V3 r = square_to_cosine_hemisphere(u, v); o[0] = r.x; o[1] = r.y; o[2] = r.z; }
void     orc_msvc_rand(u32 n, i32* out) { MsvcRand r; for (u32 i = 0; i < n; ++i) out[i] = r.next(); }
void     orc_lfsr_stream(u32 n, u32 instance, float* out)
{
	LFSRMatrix g(32, true); LFSRStream s(&g, 1u, hash(1351u + instance));
	for (u32 i = 0; i < n; ++i) out[i] = s.next();
}
// GGXSmithBsdf(roughness[,transmission,int_ior,ext_ior]).sample(u, canonical frame, V) -> L(3), g, p, p_proj
void orc_ggx_sample(float roughness, i32 transmission, float int_ior, float ext_ior, const float* u, const float* V, float* out)
{
	Frame g; g.tangent = V3(1, 0, 0); g.binormal = V3(0, 1, 0); g.normal_s = g.normal_g = V3(0, 0, 1);
	GGXSmith b(roughness, transmission != 0, int_ior, ext_ior);
	V3 L(0.0f), gg(0.0f); float p = 0, pp = 0;
	b.sample(u[0], u[1], g, V3(V[0], V[1], V[2]), L, gg, p, pp);
	out[0] = L.x; out[1] = L.y; out[2] = L.z; out[3] = gg.x; out[4] = p; out[5] = pp;
}
// GGXSmithBsdf::invert on the canonical frame -> z0, z1, 1/p, 1/p_proj
void orc_ggx_invert(float roughness, i32 transmission, float int_ior, float ext_ior, const float* V, const float* L, float* out)
{
	Frame g; g.tangent = V3(1, 0, 0); g.binormal = V3(0, 1, 0); g.normal_s = g.normal_g = V3(0, 0, 1);
	GGXSmith b(roughness, transmission != 0, int_ior, ext_ior);
	float z0 = 0, z1 = 0, p = 0, pp = 0;
	b.invert(g, V3(V[0], V[1], V[2]), V3(L[0], L[1], L[2]), z0, z1, p, pp);
	out[0] = z0; out[1] = z1; out[2] = p; out[3] = pp;
}
void orc_ggx_f_and_p(float roughness, i32 transmission, float int_ior, float ext_ior, const float* V, const float* L, float* out)
{
	Frame g; g.tangent = V3(1, 0, 0); g.binormal = V3(0, 1, 0); g.normal_s = g.normal_g = V3(0, 0, 1);
	GGXSmith b(roughness, transmission != 0, int_ior, ext_ior);
	V3 f; float p;
	b.f_and_p(g, V3(V[0], V[1], V[2]), V3(L[0], L[1], L[2]), f, p);
	out[0] = f.x; out[1] = p;
}
// the pieces tests/golden/cugar_kat.npz holds known answers for (tests/test_oracle.py::test_cugar_known_answers)
void orc_correlated_multijitter(u32 s, u32 m, u32 n, u32 p, float* out) { correlated_multijitter(s, m, n, p, out[0], out[1]); }
void orc_fresnel_schlick(float cos_theta_i, float eta, const float* base, float* out) { const V3 f = fresnel_schlick(cos_theta_i, eta, V3(base[0], base[1], base[2])); out[0] = f.x; out[1] = f.y; out[2] = f.z; }
float orc_fresnel_dielectric(float ci, float ct, float eta) { return fresnel_dielectric(ci, ct, eta); }
i32 orc_refract(const float* w_i, const float* N, float cos_theta_i, float eta, float* out)
{
	V3 o(0.0f); float F = 0.0f;
	const bool ok = refract(V3(w_i[0], w_i[1], w_i[2]), V3(N[0], N[1], N[2]), cos_theta_i, eta, &o, &F);
	out[0] = o.x; out[1] = o.y; out[2] = o.z; out[3] = F;
	return ok ? 1 : 0;
}
// LambertBsdf / LambertTransBsdf on the canonical frame: f_and_p -> f(3), p (projected solid angle); sample -> L(3), g(3), p, p_proj
void orc_lambert_f_and_p(i32 trans, const float* color, const float* V, const float* L, float* out)
{
	Frame g; g.tangent = V3(1, 0, 0); g.binormal = V3(0, 1, 0); g.normal_s = g.normal_g = V3(0, 0, 1);
	Lambert b; b.color = V3(color[0], color[1], color[2]); b.trans = trans != 0;
	V3 f; float p;
	b.f_and_p(g, V3(V[0], V[1], V[2]), V3(L[0], L[1], L[2]), f, p);
	out[0] = f.x; out[1] = f.y; out[2] = f.z; out[3] = p;
}
void orc_lambert_sample(i32 trans, const float* color, const float* u, const float* V, float* out)
{
	Frame g; g.tangent = V3(1, 0, 0); g.binormal = V3(0, 1, 0); g.normal_s = g.normal_g = V3(0, 0, 1);
	Lambert b; b.color = V3(color[0], color[1], color[2]); b.trans = trans != 0;
	V3 L(0.0f), gg(0.0f); float p = 0, pp = 0;
	b.sample(u[0], u[1], g, V3(V[0], V[1], V[2]), L, gg, p, pp);
	out[0] = L.x; out[1] = L.y; out[2] = L.z; out[3] = gg.x; out[4] = gg.y; out[5] = gg.z; out[6] = p; out[7] = pp;
}
// composite Bsdf probes on the canonical frame: f_and_p -> f[4][3], p[4]; sample -> comp, out(3), p, p_proj, g(3)
void orc_bsdf_f_and_p(const Material* m, const float* table, const float* w_i, const float* w_o, float* out)
{
	Frame g; g.tangent = V3(1, 0, 0); g.binormal = V3(0, 1, 0); g.normal_s = g.normal_g = V3(0, 0, 1);
	Bsdf b; b.setup(*m, table);
	V3 f[4]; float p[4];
	b.f_and_p(g, V3(w_i[0], w_i[1], w_i[2]), V3(w_o[0], w_o[1], w_o[2]), f, p);
	for (int i = 0; i < 4; ++i) { out[3 * i] = f[i].x; out[3 * i + 1] = f[i].y; out[3 * i + 2] = f[i].z; out[12 + i] = p[i]; }
}
void orc_bsdf_sample(const Material* m, const float* table, const float* z, const float* w_i, float* out)
{
	Frame g; g.tangent = V3(1, 0, 0); g.binormal = V3(0, 1, 0); g.normal_s = g.normal_g = V3(0, 0, 1);
	Bsdf b; b.setup(*m, table);
	u32 comp; V3 o, gg; float p, pp;
	b.sample(g, z, V3(w_i[0], w_i[1], w_i[2]), comp, o, p, pp, gg);
	out[0] = float(comp); out[1] = o.x; out[2] = o.y; out[3] = o.z; out[4] = p; out[5] = pp; out[6] = gg.x; out[7] = gg.y; out[8] = gg.z;
}
// batched forms of the two probes (statistical tests draw 10^5..10^6 samples): z / w_o are n x 3, outputs n x 9 / n x 16
void orc_bsdf_sample_n(const Material* m, const float* table, u32 n, const float* z, const float* w_i, float* out)
{
	#pragma omp parallel for schedule(static)
	for (i32 i = 0; i < i32(n); ++i) orc_bsdf_sample(m, table, z + 3 * size_t(i), w_i, out + 9 * size_t(i));
}
void orc_bsdf_f_and_p_n(const Material* m, const float* table, u32 n, const float* w_i, const float* w_o, float* out)
{
	#pragma omp parallel for schedule(static)
	for (i32 i = 0; i < i32(n); ++i) orc_bsdf_f_and_p(m, table, w_i, w_o + 3 * size_t(i), out + 16 * size_t(i));
}
// glossy reflectance table cells [begin, end) : src/bsdf.cu:36-102
void orc_glossy_reflectance_cells(u32 begin, u32 end, float* out)
{
	#pragma omp parallel for schedule(dynamic, 64)
	for (i32 c = i32(begin); c < i32(end); ++c) out[c - begin] = glossy_reflectance_cell(u32(c));
}
// tiled sequence shifts as a renderer sees them (consume_context_setup: replay RenderingContextImpl::init's setup(72,256) first)
void orc_sequence_shifts(u32 n_dims, u32 tile, const char* samples_dir, i32 consume_context_setup, float* out_shifts)
{
	MsvcRand rng;
	if (consume_context_setup) { TiledSequence ctx; ctx.setup(72, 256, samples_dir, rng); }
	TiledSequence s; s.setup(n_dims, tile, samples_dir, rng);
	std::memcpy(out_shifts, s.shifts.data(), s.shifts.size() * sizeof(float));
}

// known-answer probes of the Fermat layer (tests/golden/fermat_kat.npz): the shift layers of src/tiled_sampling.h:287-308 on a caller-held rand() state, src/mis_utils.h:43-52
u32 orc_build_tiled_samples_3d(u32 X, u32 Y, u32 Z, u32 rand_state, float* out) { MsvcRand r; r.state = rand_state; build_tiled_samples_3d(X, Y, Z, out, r); return r.state; }
u32 orc_msvc_rand_from(u32 rand_state, u32 n, i32* out) { MsvcRand r; r.state = rand_state; for (u32 i = 0; i < n; ++i) out[i] = r.next(); return r.state; }
float orc_power_heuristic(float p1, float p2) { return power_heuristic(p1, p2); }

// ---- path tracer -----------------------------------------------------------------------------------------------------
orc_pt* orc_pt_create(const orc_scene_desc* d, const PTOptions* opts, const char* samples_dir, u32 n_vpls)
{
	orc_pt* h = new orc_pt();
	PathTracer& pt = h->pt;
	pt.options = *opts;
	SceneView& s = pt.scene;
	s.camera.eye = V3(d->camera[0], d->camera[1], d->camera[2]);
	s.camera.aim = V3(d->camera[3], d->camera[4], d->camera[5]);
	s.camera.up  = V3(d->camera[6], d->camera[7], d->camera[8]);
	s.camera.dx  = V3(d->camera[9], d->camera[10], d->camera[11]);
	s.camera.fov = d->camera[12];
	h->dir_lights.resize(d->dir_lights_count);
	for (i32 i = 0; i < d->dir_lights_count; ++i)
	{
		h->dir_lights[i].dir = V3(d->dir_lights[6 * i], d->dir_lights[6 * i + 1], d->dir_lights[6 * i + 2]);
		h->dir_lights[i].color = V3(d->dir_lights[6 * i + 3], d->dir_lights[6 * i + 4], d->dir_lights[6 * i + 5]);
	}
	s.dir_lights_count = u32(d->dir_lights_count); s.dir_lights = h->dir_lights.data();
	Mesh& m = s.mesh;
	m.num_triangles = d->num_triangles; m.num_vertices = d->num_vertices; m.num_materials = d->num_materials;
	m.vertex_indices = d->vertex_indices; m.vertex_data = d->vertex_data; m.texture_indices_comp = d->texture_indices_comp; m.texture_data = d->texture_data;
	m.material_indices = d->material_indices; m.materials = d->materials;
	m.tex_bias[0] = d->tex_bias[0]; m.tex_bias[1] = d->tex_bias[1]; m.tex_scale[0] = d->tex_scale[0]; m.tex_scale[1] = d->tex_scale[1];
	h->textures.resize(d->num_textures > 0 ? d->num_textures : 1);
	for (i32 i = 0; i < d->num_textures; ++i) { h->textures[i].texels = d->textures[i].texels; h->textures[i].res_x = d->textures[i].res_x; h->textures[i].res_y = d->textures[i].res_y; }
	s.textures = h->textures.data();
	s.glossy_reflectance = d->glossy_reflectance;
	s.res_x = d->res_x; s.res_y = d->res_y; s.aspect = d->aspect; s.exposure = d->exposure; s.gamma = d->gamma;

	// init order of the reference: context sequence (72 dims) -> renderer sequence -> mesh lights (src/renderer.cu:949-953, pathtracer_impl.h:148-157)
	MsvcRand rng;
	{ TiledSequence ctx; ctx.setup(72, 256, samples_dir, rng); }
	pt.sequence.setup(6 * (opts->max_path_length + 1), 256, samples_dir, rng);
	h->lights.init(n_vpls, m, s.textures, 0);
	MeshLight ml;
	ml.n_prims = u32(m.num_triangles); ml.prims_cdf = h->lights.mesh_cdf.data(); ml.prims_inv_area = h->lights.mesh_inv_area.data();
	ml.mesh = &s.mesh; ml.textures = s.textures; ml.n_vpls = 0; ml.vpls = 0; ml.norm = h->lights.normalization_coeff;
	s.mesh_light = ml;
	ml.n_vpls = u32(h->lights.vpls.size()); ml.vpls = h->lights.vpls.data();
	s.mesh_vpls = ml;
	if (ml.n_vpls == 0) pt.options.nee_type = 0;        // pathtracer_impl.h:165-166
	pt.caster.build(s.mesh);
	for (int c = 0; c < FB_NUM_CHANNELS; ++c) pt.fb.channels[c] = 0;
	pt.fb.gb_geo = pt.fb.gb_uv = 0; pt.fb.gb_tri = 0; pt.fb.gb_depth = 0;
	pt.fb.res_x = d->res_x; pt.fb.res_y = d->res_y;
	return h;
}
void orc_pt_destroy(orc_pt* h) { delete h; }

void orc_pt_set_framebuffer(orc_pt* h, float* const* channels, float* gb_geo, float* gb_uv, u32* gb_tri, float* gb_depth)
{
	for (int c = 0; c < FB_NUM_CHANNELS; ++c) h->pt.fb.channels[c] = channels[c];
	h->pt.fb.gb_geo = gb_geo; h->pt.fb.gb_uv = gb_uv; h->pt.fb.gb_tri = gb_tri; h->pt.fb.gb_depth = gb_depth;
}
void orc_pt_render_pass(orc_pt* h, u32 instance, const u32* pixels, u32 n_pixels) { h->pt.render_pass(instance, pixels, n_pixels); }
u32  orc_pt_stats(orc_pt* h, BounceStats* out, u32 max_n)
{
	const u32 n = u32(h->pt.stats.size()) < max_n ? u32(h->pt.stats.size()) : max_n;
	for (u32 i = 0; i < n; ++i) out[i] = h->pt.stats[i];
	return u32(h->pt.stats.size());
}
void orc_pt_set_capture(orc_pt* h, i32 bounce) { h->pt.capture_bounce = bounce; }
u32  orc_pt_get_captured(orc_pt* h, PathEntry* out, u32 max_n)
{
	const u32 n = u32(h->pt.captured.size()) < max_n ? u32(h->pt.captured.size()) : max_n;
	if (out) for (u32 i = 0; i < n; ++i) out[i] = h->pt.captured[i];
	return u32(h->pt.captured.size());
}
void orc_pt_to_rgba(orc_pt* h, uint8_t* rgba) { h->pt.to_rgba(rgba); }
// post-process (o_filter.h): RenderingContextImpl::filter and the per-ShadingMode to_rgba
void orc_pt_filter(orc_pt* h, u32 instance) { filter_frame(h->pt.fb, h->pt.scene, instance); }
void orc_pt_to_rgba_mode(orc_pt* h, u32 mode, uint8_t* rgba) { to_rgba_mode(h->pt.fb, h->pt.scene, mode, rgba); }
void orc_filter_variance(u32 res_x, u32 res_y, float* img, float* var, u32 FW) { Image i = { img, res_x, res_y }; filter_variance(i, var, FW); }
// one EAW step on caller-provided buffers (op < 0: EAW_kernel; else EAW_mad_kernel with FilterOp bits); params = phi_normal, phi_position, phi_color, E, U, V, W
void orc_eaw_step(u32 res_x, u32 res_y, float* dst, int op, float* w_img, float w_min, float* img, const float* gb_geo, const float* var, const float* params, u32 step_size)
{
	EAWParams p; p.phi_normal = params[0]; p.phi_position = params[1]; p.phi_color = params[2];
	p.E = V3(params[3], params[4], params[5]); p.U = V3(params[6], params[7], params[8]); p.V = V3(params[9], params[10], params[11]); p.W = V3(params[12], params[13], params[14]);
	Image d = { dst, res_x, res_y }, w = { w_img, res_x, res_y }, i = { img, res_x, res_y };
	eaw_step(d, op, w, w_min, i, gb_geo, var, p, step_size);
}
// ---- path-space filtering (o_psfpt.h): switch the context's path tracer to the PSFPT vertex processor -------------------------------
// what-if switch for tests/test_oracle_statistics.py: 1 = the shadow samples of a vertex carry out_vertex_info (what compute_nee_weights computed) instead of
// vertex_info (what the reference passes, src/pathtracer_core.h:984,1102).  0 = the reference's behaviour.  Call after orc_psf_enable.
void orc_psf_set_whatif(orc_pt* h, u32 bits) { h->psf.whatif_nee_vertex_info = (bits & 1u) != 0; }
void orc_psf_enable(orc_pt* h, const PSFOptions* opts)
{
	h->psf.options = *opts;
	// m_bbox = renderer.compute_bbox() (src/renderer.cu:1086-1097): the bounding box of the mesh vertices
	const Mesh& m = h->pt.scene.mesh;
	V3 lo(1.0e30f), hi(-1.0e30f);
	for (i32 i = 0; i < m.num_vertices; ++i)
	{
		const V3 p = load_vertex(m, i);
		lo = V3(minf(lo.x, p.x), minf(lo.y, p.y), minf(lo.z, p.z)); hi = V3(maxf(hi.x, p.x), maxf(hi.y, p.y), maxf(hi.z, p.z));
	}
	h->psf.bbox_lo = lo; h->psf.bbox_hi = hi;
	h->psf.clear();
	h->pt.psf = &h->psf;
}
// cache cells of the current frame set, sorted by key: key, count, and the three fixed-point sums per cell
u32 orc_psf_get_cells(orc_pt* h, u64* keys, u64* counts, long long* sums, u32 max_n)
{
	const PsfState& s = h->psf;
	const u32 n = u32(s.cells.size());
	if (!keys) return n;
	std::vector<u32> order(n);
	for (u32 i = 0; i < n; ++i) order[i] = i;
	std::sort(order.begin(), order.end(), [&](u32 a, u32 b) { return s.keys[a] < s.keys[b]; });
	for (u32 i = 0; i < n && i < max_n; ++i)
	{
		const u32 k = order[i];
		keys[i] = s.keys[k]; counts[i] = s.cells[k].count; sums[3 * i] = s.cells[k].x; sums[3 * i + 1] = s.cells[k].y; sums[3 * i + 2] = s.cells[k].z;
	}
	return n;
}
u32 orc_psf_ref_count(orc_pt* h) { return u32(h->psf.refs.size()); }

// ---- bidirectional path tracer (o_bpt.h) on the same context: scene, BVH, mesh lights and frame buffer are shared -------------------
void orc_bpt_init(orc_pt* h, const BPTOptions* opts, const char* samples_dir) { h->bpt.init(&h->pt, *opts, samples_dir); }
void orc_bpt_render(orc_pt* h, u32 instance) { h->bpt.render(instance); }
// what-if switches for tests/test_oracle_statistics.py (bit 0: true distance in the first eye vertex's G'; bit 1: reverse pdf in connect_to_camera): the two
// places where the reference's MIS weights are not consistent between light tracing and the eye strategies.  0 = the reference's behaviour.
void orc_bpt_set_whatif(orc_pt* h, u32 bits) { h->bpt.whatif_consistent_mis = bits; }
void orc_bpt_render_pixels(orc_pt* h, u32 instance, const u32* pixels, u32 n) { h->bpt.render(instance, pixels, n); }
void orc_bpt_set_deferred_splats(orc_pt* h, i32 on) { h->bpt.deferred_splats = on != 0; }
long long* orc_bpt_splats(orc_pt* h) { return h->bpt.splat.data(); }      // 6 per pixel: COMPOSITED xyz, DIRECT xyz
void orc_bpt_resolve_splats(orc_pt* h) { h->bpt.resolve_splats(); }
void orc_bpt_get_stats(orc_pt* h, u32* out /* 32 light queue, 32 eye queue, 32 eye shadow, n_light_vertices, shadow_lt, n_bounces_light, n_bounces_eye */)
{
	const BPT::Stats& s = h->bpt.stats;
	for (int i = 0; i < 32; ++i) { out[i] = s.light_queue[i]; out[32 + i] = s.eye_queue[i]; out[64 + i] = s.shadow_eye[i]; }
	out[96] = s.n_light_vertices; out[97] = s.shadow_light_tracing; out[98] = s.n_bounces_light; out[99] = s.n_bounces_eye;
}
// light-vertex store of the last pass: pos float4, input uint2, gbuffer uint4, weights float2, path_id u32 per slot; counts per path
void orc_bpt_get_light_vertices(orc_pt* h, float* pos, u32* input, u32* gbuffer, float* weights, u32* path_id, u32* counts)
{
	const BPT& b = h->bpt;
	std::memcpy(pos, b.v_pos.data(), b.v_pos.size() * 4); std::memcpy(input, b.v_input.data(), b.v_input.size() * 4);
	std::memcpy(gbuffer, b.v_gbuffer.data(), b.v_gbuffer.size() * sizeof(PackedBsdf)); std::memcpy(weights, b.v_weights.data(), b.v_weights.size() * 4);
	std::memcpy(path_id, b.v_path_id.data(), b.v_path_id.size() * 4); std::memcpy(counts, b.v_counts.data(), b.v_counts.size() * 4);
}
// probes for the packers
u32 orc_to_rgbe(float r, float g, float b) { return to_rgbe(V3(r, g, b)); }
void orc_from_rgbe(u32 p, float* o) { const V3 v = from_rgbe(p); o[0] = v.x; o[1] = v.y; o[2] = v.z; }
u32 orc_pack_direction(float x, float y, float z) { return pack_direction(V3(x, y, z)); }
void orc_unpack_direction(u32 p, float* o) { const V3 v = unpack_direction(p); o[0] = v.x; o[1] = v.y; o[2] = v.z; }

// host threads used for the queue traces inside render_pass, and the wall time spent in them so far
void orc_debug_set_box_clause(i32 on) { box_clause_enabled() = on != 0; }
void orc_pt_log_rays(orc_pt* h, i32 on) { h->pt.log_rays = on != 0; if (on) { h->pt.logged_rays.clear(); h->pt.logged_hits.clear(); h->pt.logged_kind.clear(); } }
u32  orc_pt_get_logged_rays(orc_pt* h, Ray* rays, Hit* hits, u32* kind, u32 max_n)
{
	const u32 n = u32(h->pt.logged_rays.size());
	for (u32 i = 0; i < n && i < max_n; ++i) { rays[i] = h->pt.logged_rays[i]; hits[i] = h->pt.logged_hits[i]; kind[i] = h->pt.logged_kind[i]; }
	return n;
}
void orc_pt_set_trace_threads(orc_pt* h, i32 n) { h->pt.trace_threads = n > 1 ? n : 1; }
double orc_pt_trace_seconds(orc_pt* h) { return h->pt.trace_seconds; }
double orc_pt_shade_seconds(orc_pt* h) { return h->pt.shade_seconds; }
void orc_pt_rescale_frame(orc_pt* h, u32 instance) { h->pt.rescale_frame(instance); }
void orc_pt_update_variances(orc_pt* h, u32 instance) { h->pt.update_variances(instance); }

void orc_pt_trace(orc_pt* h, u32 n, const Ray* rays, Hit* hits, i32 n_threads)
{
	(void)n_threads;
#ifdef _OPENMP
	if (n_threads > 1)
	{
		#pragma omp parallel num_threads(n_threads)
		{
			RayCaster local = h->pt.caster;     // private counters
			#pragma omp for schedule(static)
			for (i32 i = 0; i < i32(n); ++i) hits[i] = local.trace(rays[i]);
		}
		return;
	}
#endif
	for (u32 i = 0; i < n; ++i) hits[i] = h->pt.caster.trace(rays[i]);
}
void orc_pt_trace_shadow(orc_pt* h, u32 n, const Ray* rays, Hit* hits, i32 n_threads)
{
	(void)n_threads;
#ifdef _OPENMP
	if (n_threads > 1)
	{
		#pragma omp parallel num_threads(n_threads)
		{
			RayCaster local = h->pt.caster;
			#pragma omp for schedule(static)
			for (i32 i = 0; i < i32(n); ++i) hits[i] = local.trace_shadow(rays[i]);
		}
		return;
	}
#endif
	for (u32 i = 0; i < n; ++i) hits[i] = h->pt.caster.trace_shadow(rays[i]);
}
// counters: [0] closest-hit rays, [1] shadow rays, [2] bvh nodes visited, [3] triangles tested
void orc_pt_counters(orc_pt* h, u64* out)
{
	out[0] = h->pt.rays_traced; out[1] = h->pt.shadow_rays_traced; out[2] = h->pt.caster.nodes_visited; out[3] = h->pt.caster.tris_tested;
}
u32  orc_pt_n_dims(orc_pt* h) { return h->pt.sequence.n_dimensions; }
void orc_pt_get_sequence(orc_pt* h, float* shifts, float* samples)
{
	if (shifts)  std::memcpy(shifts, h->pt.sequence.shifts.data(), h->pt.sequence.shifts.size() * sizeof(float));
	if (samples) std::memcpy(samples, h->pt.sequence.samples.data(), h->pt.sequence.samples.size() * sizeof(float));
}
void orc_pt_set_instance(orc_pt* h, u32 instance) { h->pt.sequence.set_instance(instance); }
float orc_pt_sample_2d(orc_pt* h, u32 px, u32 py, u32 dim) { return h->pt.sequence.sample_2d(px, py, dim); }
u32  orc_pt_get_lights(orc_pt* h, VPL* vpls, float* vpl_cdf, float* mesh_cdf, float* mesh_inv_area, float* norm)
{
	const MeshLightsStorage& L = h->lights;
	if (vpls && !L.vpls.empty()) std::memcpy(vpls, L.vpls.data(), L.vpls.size() * sizeof(VPL));
	if (vpl_cdf && !L.vpl_cdf.empty()) std::memcpy(vpl_cdf, L.vpl_cdf.data(), L.vpl_cdf.size() * sizeof(float));
	if (mesh_cdf) std::memcpy(mesh_cdf, L.mesh_cdf.data(), L.mesh_cdf.size() * sizeof(float));
	if (mesh_inv_area) std::memcpy(mesh_inv_area, L.mesh_inv_area.data(), L.mesh_inv_area.size() * sizeof(float));
	if (norm) *norm = L.normalization_coeff;
	return u32(L.vpls.size());
}
u32 orc_pt_bvh_info(orc_pt* h, u32* n_nodes) { *n_nodes = u32(h->pt.caster.bvh.nodes.size()); return u32(h->pt.caster.bvh.index.size()); }

} // extern "C"
