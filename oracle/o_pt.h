// ORACLE — TEST INFRASTRUCTURE ONLY (see o_math.h header).
//
// o_pt.h : CPU restatement of the -pt wavefront path tracer.
//   per-bounce options / ray gen / shade_vertex / solve_occlusion : src/pathtracer_core.h:594-656,705-749,771-1254
//   vertex processor and frame-buffer accumulation                 : src/pathtracer_vertex_processor.h:46-241, src/framebuffer.h:425-444
//   loop                                                           : src/pathtracer_kernels.h:133-181,189-280,309-391
//   frame-buffer utility kernels                                   : src/renderer.cu:83-106,292-312,333-362,403-437
//   renderer glue                                                  : src/renderers/pathtracer.h:161-199, pathtracer_impl.h:197-324
// Queues are plain vectors processed in slot order; entries carry the same fields as PTRayQueue (src/pathtracer_queues.h:44-93).
// Directional-light shadow samples are kept in their own queue and resolved BEFORE the mesh-light samples of the same
// bounce: the reference resolves both in one racy launch (SURVEY §5 "known benign race"); this order is the
// program order of shade_vertex and is the build's specification.
#pragma once
#include <chrono>
#include "o_psfpt.h"
#include "o_scene.h"
#include "o_sequence.h"
#include "o_bvh.h"

namespace orc {

// src/renderers/pathtracer.h:170-199
struct PTOptions
{
	u32 max_path_length;
	u32 direct_lighting, direct_lighting_nee, direct_lighting_bsdf, indirect_lighting_nee, indirect_lighting_bsdf;
	u32 visible_lights, diffuse_scattering, glossy_scattering, indirect_glossy, rr, nee_type;   // nee_type: 0 mesh, 1 vpl
};

// src/renderer_view.h:133-145
enum { FB_DIFFUSE_C = 0, FB_DIFFUSE_A = 1, FB_SPECULAR_C = 2, FB_SPECULAR_A = 3, FB_DIRECT_C = 4, FB_COMPOSITED_C = 5, FB_FILTERED_C = 6, FB_LUMINANCE = 7, FB_NUM_CHANNELS = 8 };

struct FrameBuffer
{
	u32 res_x, res_y;
	float* channels[FB_NUM_CHANNELS];   // float4 per pixel each
	float* gb_geo; float* gb_uv; u32* gb_tri; float* gb_depth;   // gbuffer (src/framebuffer.h:49-143); may be NULL
	V4 get(u32 c, u32 p) const { const float* f = channels[c] + 4 * size_t(p); return V4(f[0], f[1], f[2], f[3]); }
	void set(u32 c, u32 p, V4 v) { float* f = channels[c] + 4 * size_t(p); f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w; }
};

// src/framebuffer.h:425-444
inline void add_in(FrameBuffer& fb, u32 channel, u32 pixel, V3 f, float inv_n, bool alpha_as_variance)
{
	V4 mean = fb.get(channel, pixel);
	const V3 delta = f - mean.xyz();
	mean.x += f.x * inv_n;
	mean.y += f.y * inv_n;
	mean.z += f.z * inv_n;
	if (alpha_as_variance)
	{
		const float lum_delta = max_comp(delta);
		mean.w += lum_delta * lum_delta * inv_n;
	}
	fb.set(channel, pixel, mean);
}

// src/mis_utils.h:43-52
inline float power_heuristic(float p1, float p2)
{
	const bool p1_inf = !finite_f(p1);
	const bool p2_inf = !finite_f(p2);
	return p1_inf ? 1.0f : p2_inf ? 0.0f : (p1 * p1) / (p1 * p1 + p2 * p2);
}
// src/bpt_utils.h:84-90
inline float pdf_product(float p1, float p2) { return finite_f(p1) && finite_f(p2) ? p1 * p2 : finf(); }

// contrib/cugar/spherical/mappings_inline.h:174-185 + contrib/cugar/linalg/vector_inl.h:456-462 ; src/framebuffer.h:97-104
inline float pack_geometry_normal(V3 N)
{
	float phi;
	if (fabsf(N.z) >= 1.0f - 1.0e-5f) phi = 0.0f;
	else { phi = det_atan2(N.y, N.x); phi = phi < 0.0f ? phi + 2.0f * PI_F : phi; }
	const float sx = phi / (2.0f * PI_F), sy = (N.z + 1.0f) * 0.5f;
	const u32 MAXV = (1u << 15) - 1u;
	const u32 n_i = quantize(sx, MAXV) | (quantize(sy, MAXV) << 15);
	return bits2f((0u << 31) | n_i);
}

struct PathEntry      // in_queue / scatter_queue entry : 88 B in the reference SoA
{
	Ray ray; Hit hit; V4 weight; u32 pixel_info; float cone_x, cone_y;
	u32 vertex_info;    // pixels.y of the reference's uint4: what the vertex processor returned at the previous vertex (0xFFFFFFFF for the plain PT)
};
struct ShadowEntry    // shadow_queue entry : 112 B
{
	Ray ray; Hit hit; V3 w, w_d, w_g; u32 pixel_info;
	u32 vertex_info;
};

// PixelInfo : src/pathtracer_core.h:527-542 — pixel:27 | comp:4 | diffuse:1
inline u32 pixel_info_pack(u32 pixel, u32 comp, u32 diffuse) { return (pixel & 0x7FFFFFFu) | ((comp & 0xFu) << 27) | ((diffuse & 1u) << 31); }
inline u32 pi_pixel(u32 p) { return p & 0x7FFFFFFu; }
inline u32 pi_comp(u32 p) { return (p >> 27) & 0xFu; }
inline u32 pi_diffuse(u32 p) { return p >> 31; }

struct BounceStats { u32 in_size, shadow_dir_size, shadow_size, scatter_size; };

struct PathTracer
{
	PTOptions options;
	SceneView scene;
	TiledSequence sequence;
	RayCaster caster;
	FrameBuffer fb;
	// per-bounce state (PTContextBase, src/pathtracer_core.h:569-584)
	u32 in_bounce; bool do_nee, do_accumulate_emissive, do_scatter; float frame_weight;
	std::vector<PathEntry> in_queue, scatter_queue;
	std::vector<ShadowEntry> shadow_dir_queue, shadow_queue;
	std::vector<BounceStats> stats;
	// optional capture of the in-queue (after tracing) of one bounce, for stage-level parity tests
	int capture_bounce; std::vector<PathEntry> captured;
	u64 rays_traced, shadow_rays_traced;

	PsfState* psf = nullptr;        // non-null: the PSFPT vertex processor (o_psfpt.h) replaces the PT one
	PathTracer() : capture_bounce(-1), rays_traced(0), shadow_rays_traced(0) {}

	const MeshLight& light() const { return options.nee_type == 1 ? scene.mesh_vpls : scene.mesh_light; }   // pathtracer_impl.h:272

	// src/pathtracer_core.h:594-620
	void compute_per_bounce_options()
	{
		do_nee = scene.mesh_vpls.n_vpls &&
			((in_bounce + 2 <= options.max_path_length) &&
			 ((in_bounce == 0 && options.direct_lighting_nee && options.direct_lighting) ||
			  (in_bounce >  0 && options.indirect_lighting_nee)));
		do_accumulate_emissive =
			((in_bounce == 0 && options.visible_lights) ||
			 (in_bounce == 1 && options.direct_lighting_bsdf && options.direct_lighting) ||
			 (in_bounce >  1 && options.indirect_lighting_bsdf));
		const u32 max_path_vertices = options.max_path_length +
			((options.max_path_length == 2 && options.direct_lighting_bsdf) ||
			 (options.max_path_length >  2 && options.indirect_lighting_bsdf) ? 1 : 0);
		do_scatter = (in_bounce + 2 < max_path_vertices);
	}

	// src/renderer.cu:292-312,403-416
	// (pixels != nullptr: only the listed pixels, as a tile-sharded rank does -- the other pixels of this frame are then not maintained)
	void rescale_frame(u32 instance, const u32* pixels = nullptr, u32 n_pixels = 0)
	{
		const float scale = float(instance) / float(instance + 1);
		const u32 n = pixels ? n_pixels : fb.res_x * fb.res_y;
		for (u32 i = 0; i < n; ++i)
		{
			const u32 p = pixels ? pixels[i] : i;
			fb.set(FB_LUMINANCE, p, V4(max_comp(fb.get(FB_DIRECT_C, p).xyz()), max_comp(fb.get(FB_DIFFUSE_C, p).xyz()),
			                           max_comp(fb.get(FB_SPECULAR_C, p).xyz()), max_comp(fb.get(FB_COMPOSITED_C, p).xyz())));
			const u32 ch[6] = { FB_DIFFUSE_C, FB_DIFFUSE_A, FB_SPECULAR_C, FB_SPECULAR_A, FB_DIRECT_C, FB_COMPOSITED_C };
			for (int c = 0; c < 6; ++c) fb.set(ch[c], p, fb.get(ch[c], p) * scale);
		}
	}
	// src/renderer.cu:333-362,431-437
	void update_variances(u32 instance, const u32* pixels = nullptr, u32 n_pixels = 0)
	{
		const u32 n = instance + 1;
		const u32 np = pixels ? n_pixels : fb.res_x * fb.res_y;
		for (u32 k = 0; k < np; ++k)
		{
			const u32 p = pixels ? pixels[k] : k;
			const V4 old_lum = fb.get(FB_LUMINANCE, p);
			const V4 new_lum(max_comp(fb.get(FB_DIRECT_C, p).xyz()), max_comp(fb.get(FB_DIFFUSE_C, p).xyz()),
			                 max_comp(fb.get(FB_SPECULAR_C, p).xyz()), max_comp(fb.get(FB_COMPOSITED_C, p).xyz()));
			const float fn = float(n), fn1 = float(n - 1), fnn = float(n * n);
			const float d[4] = { new_lum.x - old_lum.x, new_lum.y - old_lum.y, new_lum.z - old_lum.z, new_lum.w - old_lum.w };
			float dv[4];
			for (int i = 0; i < 4; ++i) dv[i] = ((fn * d[i]) * (fn1 * d[i])) / fnn;
			const u32 ch[4] = { FB_DIRECT_C, FB_DIFFUSE_C, FB_SPECULAR_C, FB_COMPOSITED_C };
			for (int i = 0; i < 4; ++i) { V4 v = fb.get(ch[i], p); v.w += dv[i]; fb.set(ch[i], p, v); }
		}
	}
	// src/renderer.cu:83-106 (kShaded) ; powf restated with det_pow (detmath v1)
	void to_rgba(uint8_t* rgba) const
	{
		const u32 np = fb.res_x * fb.res_y;
		for (u32 p = 0; p < np; ++p)
		{
			V4 c = fb.get(FB_COMPOSITED_C, p) * scene.exposure;
			float v[4] = { c.x / (c.x + 1.0f), c.y / (c.y + 1.0f), c.z / (c.z + 1.0f), c.w / (c.w + 1.0f) };
			for (int i = 0; i < 4; ++i)
			{
				const float g = det_pow(v[i], 1.0f / scene.gamma);
				rgba[4 * size_t(p) + i] = uint8_t(f2u(fmin_ieee(g * 256.0f, 255.0f)));
			}
		}
	}

	// src/pathtracer_kernels.h:133-181 + src/pathtracer_core.h:633-656.  `pixels` = absolute pixel indices to render
	// (the whole frame in the reference; a tile subset under multi-GPU sharding)
	void generate_primary_rays(const u32* pixels, u32 n)
	{
		V3 U, V, W;
		camera_frame(scene.camera, scene.aspect, U, V, W);
		const float W_len = length(W);
		const float sq_focal = square_pixel_focal_length(scene.camera, scene.res_x, scene.res_y);
		in_queue.resize(n);
		for (u32 i = 0; i < n; ++i)
		{
			const u32 idx = pixels ? pixels[i] : i;
			const u32 px = idx % scene.res_x, py = idx / scene.res_x;
			const float ux = sequence.sample_2d(px, py, 0), uy = sequence.sample_2d(px, py, 1);
			const float dx = ((float(px) + ux) / float(scene.res_x)) * 2.f - 1.f;
			const float dy = ((float(py) + uy) / float(scene.res_y)) * 2.f - 1.f;
			const V3 dir = dx * U + dy * V + W;
			PathEntry& e = in_queue[i];
			e.ray.ox = scene.camera.eye.x; e.ray.oy = scene.camera.eye.y; e.ray.oz = scene.camera.eye.z; e.ray.mask_or_tmin = 0u;
			e.ray.dx = dir.x; e.ray.dy = dir.y; e.ray.dz = dir.z; e.ray.tmax = 1e34f;
			e.weight = V4(1.0f, 1.0f, 1.0f, 1.0f);
			e.pixel_info = idx;     // make_uint4(idx, -1, -1, -1): comp = 0, diffuse = 0
			e.vertex_info = 0xFFFFFFFFu;
			e.cone_x = 0.0f; e.cone_y = camera_direction_pdf(U, V, W, W_len, sq_focal, dir);
		}
	}

	// src/pathtracer_vertex_processor.h:151-183
	void accumulate_emissive(u32 pixel_info, V3 w)
	{
		const u32 pixel = pi_pixel(pixel_info), comp = pi_comp(pixel_info);
		add_in(fb, FB_COMPOSITED_C, pixel, w, frame_weight, false);
		if (in_bounce == 0) add_in(fb, FB_DIRECT_C, pixel, w, frame_weight, false);
		else
		{
			if (comp & kDiffuseMask) add_in(fb, FB_DIFFUSE_C, pixel, w, frame_weight, true);
			if (comp & kGlossyMask)  add_in(fb, FB_SPECULAR_C, pixel, w, frame_weight, true);
		}
	}
	// src/pathtracer_vertex_processor.h:202-239
	void accumulate_nee(u32 pixel_info, bool shadow_hit, V3 w_d, V3 w_g)
	{
		if (shadow_hit) return;
		const u32 pixel = pi_pixel(pixel_info), comp = pi_comp(pixel_info);
		add_in(fb, FB_COMPOSITED_C, pixel, w_d + w_g, frame_weight, false);
		if (in_bounce == 0)
		{
			add_in(fb, FB_DIFFUSE_C, pixel, w_d, frame_weight, true);
			add_in(fb, FB_SPECULAR_C, pixel, w_g, frame_weight, true);
		}
		else
		{
			if (comp & kDiffuseMask) add_in(fb, FB_DIFFUSE_C, pixel, w_d, frame_weight, true);
			if (comp & kGlossyMask)  add_in(fb, FB_SPECULAR_C, pixel, w_g, frame_weight, true);
		}
	}

	// NEE / directional shared tail : src/pathtracer_core.h:1013-1106 (mesh) and :895-988 (directional)
	void nee_sample(const EyeVertex& ev, const PathEntry& e, const VertexGeometry& lg, float light_pdf, const Edf& edf,
	                bool use_mis, float origin_eps, u32 mask, std::vector<ShadowEntry>& queue, u32 vertex_info = 0xFFFFFFFFu)
	{
		const V3 w = e.weight.xyz();
		V3 out = lg.position - ev.geom.position;
		const float d2 = fmax_ieee(1.0e-8f, dot(out, out));
		out = out * (1.0f / sqrtf(d2));                       // rsqrtf restated as 1/sqrtf (detmath v1)
		V3 f_s[4]; float p_s[4];
		ev.bsdf.f_and_p(ev.geom, ev.in, out, f_s, p_s);
		const bool eval_diffuse = options.diffuse_scattering, eval_glossy = options.glossy_scattering;
		float p_sum = 0.0f;
		if (eval_diffuse) p_sum += p_s[kDiffR] + p_s[kDiffT];
		if (eval_glossy)  p_sum += p_s[kGlossR] + p_s[kGlossT];
		const V3 f_L = edf.f(lg, -out) / light_pdf;
		const float G = fabsf(dot(out, ev.geom.normal_s) * dot(out, lg.normal_s)) / d2;
		float mis_w = 1.0f;
		if (use_mis)
		{
			const float p1 = light_pdf, p2 = p_sum * G;
			mis_w = ((in_bounce == 0 && options.direct_lighting_bsdf) || (in_bounce > 0 && options.indirect_lighting_bsdf)) ? power_heuristic(p1, p2) : 1.0f;
		}
		// compute_nee_weights : src/pathtracer_vertex_processor.h:83-105
		const V3 f_d = eval_diffuse ? f_s[kDiffR] + f_s[kDiffT] : V3(0.0f);
		const V3 f_g = eval_glossy ? f_s[kGlossR] + f_s[kGlossT] : V3(0.0f);
		const V3 fl = f_L * G * mis_w;
		V3 out_w_d = (in_bounce == 0 ? f_d : f_d + f_g) * w * fl;
		V3 out_w_g = (in_bounce == 0 ? f_g : f_d + f_g) * w * fl;
		if (psf) psf_nee_weights(ev, vertex_info, f_d, f_g, w, fl, out_w_d, out_w_g);
		const V3 out_w = out_w_d + out_w_g;
		if (max_comp(out_w) > 0.0f && finite3(out_w))
		{
			const V3 rd(e.ray.dx, e.ray.dy, e.ray.dz);
			const V3 org = ev.geom.position - rd * origin_eps;
			const V3 dir = lg.position - org;
			ShadowEntry s;
			s.ray.ox = org.x; s.ray.oy = org.y; s.ray.oz = org.z; s.ray.mask_or_tmin = mask;
			s.ray.dx = dir.x; s.ray.dy = dir.y; s.ray.dz = dir.z; s.ray.tmax = 0.9999f;
			s.w = out_w; s.w_d = out_w_d; s.w_g = out_w_g; s.pixel_info = e.pixel_info;
			s.vertex_info = vertex_info;          // trace_shadow_ray receives vertex_info, not out_vertex_info (src/pathtracer_core.h:984,1102)
			// ... so at a NEW cache vertex accumulate_nee sees comp 0, not DIFFUSE_COMP, and folds the un-demodulated glossy term into the cell too,
			// where the blend multiplies it by w * diffuse once more: the PSFPT's NEE energy loss.  Test-only what-if: what
			// compute_nee_weights computed (src/psfpt_vertex_processor.h:231-236)
			if (psf && psf->whatif_nee_vertex_info)
				s.vertex_info = (in_bounce < psf->options.psf_depth) ? 0xFFFFFFFFu : cache_info(ci_slot(vertex_info), ci_new(vertex_info) ? 1u : 3u, 0);
			queue.push_back(s);
		}
	}

	// src/pathtracer_core.h:771-1254
	// the three output queues are parameters so that shade_queue() can give every thread its own
	void shade_vertex(const PathEntry& e, std::vector<ShadowEntry>& out_shadow_dir, std::vector<ShadowEntry>& out_shadow, std::vector<PathEntry>& out_scatter)
	{
		const Hit& hit = e.hit;
		const float p_prev = e.weight.w;
		const u32 pixel_index = pi_pixel(e.pixel_info);
		const u32 px = pixel_index % scene.res_x, py = pixel_index / scene.res_x;
		if (!(hit.t > 0.0f && hit.triId >= 0)) return;

		EyeVertex ev;
		ev.setup(e.ray, hit, scene);
		const V3 w = e.weight.xyz();
		const V3 rd(e.ray.dx, e.ray.dy, e.ray.dz);

		if (in_bounce == 0)
		{
			if (fb.gb_geo)
			{
				float* g = fb.gb_geo + 4 * size_t(pixel_index);
				g[0] = ev.geom.position.x; g[1] = ev.geom.position.y; g[2] = ev.geom.position.z; g[3] = pack_geometry_normal(ev.geom.normal_s);
				float* uv = fb.gb_uv + 4 * size_t(pixel_index);
				uv[0] = hit.u; uv[1] = hit.v; uv[2] = ev.geom.texture_coords.x; uv[3] = ev.geom.texture_coords.y;
				fb.gb_tri[pixel_index] = u32(hit.triId);
				fb.gb_depth[pixel_index] = hit.t;
			}
			fb.set(FB_DIFFUSE_A, pixel_index, fb.get(FB_DIFFUSE_A, pixel_index) + ev.material.diffuse * frame_weight);
			fb.set(FB_SPECULAR_A, pixel_index, fb.get(FB_SPECULAR_A, pixel_index) + (ev.material.specular + V4(1, 1, 1, 1)) * 0.5f * frame_weight);
		}

		const float area_prob = 1.0f / sqrtf(e.cone_y * ev.prev_G_prime);    // cugar::rsqrtf == 1/sqrtf (numbers.h:1000-1008)
		const float cone_radius = e.cone_x + area_prob;

		const u32 prev_vertex_info = e.vertex_info;
		const u32 vertex_info = psf ? psf_preprocess_vertex(e.pixel_info, ev, cone_radius, prev_vertex_info, w, p_prev) : 0xFFFFFFFFu;

		float samples[6];
		for (u32 i = 0; i < 6; ++i) samples[i] = sequence.sample_2d(px, py, (in_bounce + 1) * 6 + i);

		// directional lights : :870-988
		if ((in_bounce + 2 <= options.max_path_length) && (in_bounce > 0 || options.direct_lighting) && scene.dir_lights_count)
		{
			const u32 li = quantize(samples[2], scene.dir_lights_count);
			const DirectionalLight& L = scene.dir_lights[li];
			VertexGeometry lg;
			const float FAR = 1.0e8f;                                   // src/lights.h:276-294
			lg.position = ev.geom.position - L.dir * FAR;
			lg.normal_s = lg.normal_g = L.dir;
			lg.tangent = orthogonal(L.dir);
			lg.binormal = cross(L.dir, lg.tangent);
			float light_pdf = 1.0f;
			Edf edf; edf.color = FAR * FAR * L.color;
			light_pdf /= float(scene.dir_lights_count);
			nee_sample(ev, e, lg, light_pdf, edf, false, 1.0e-3f, 0x1u, out_shadow_dir, vertex_info);
		}
		// mesh / VPL next-event estimation : :991-1106
		if (do_nee)
		{
			u32 prim; float lu, lv; VertexGeometry lg; float light_pdf; Edf edf;
			light().sample(samples, &prim, &lu, &lv, &lg, &light_pdf, &edf);
			nee_sample(ev, e, lg, light_pdf, edf, true, 1.0e-4f, 0x2u, out_shadow, vertex_info);
		}
		// emissive hit : :1109-1154
		if (do_accumulate_emissive)
		{
			float light_pdf; Edf edf;
			light().map_geom(u32(hit.triId), ev.geom, &light_pdf, &edf);
			const V3 f_L = edf.f(ev.geom, ev.in);
			const float d2 = fmax_ieee(1.0e-10f, hit.t * hit.t);
			const float G_partial = fabsf(dot(ev.in, ev.geom.normal_s)) / d2;
			const float p1 = pdf_product(G_partial, p_prev);
			const float p2 = light_pdf;
			const float mis_w = ((in_bounce == 1 && options.direct_lighting_nee) || (in_bounce > 1 && options.indirect_lighting_nee)) ? power_heuristic(p1, p2) : 1.0f;
			const V3 out_w = w * f_L * mis_w;
			if (max_comp(out_w) > 0.0f && finite3(out_w))
			{
				if (psf) psf_accumulate_emissive(e.pixel_info, prev_vertex_info, out_w);
				else accumulate_emissive(e.pixel_info, out_w);
			}
		}
		// scattering : :1157-1247
		if (do_scatter)
		{
			const float z[3] = { samples[3], samples[4], samples[5] };
			V3 out(0.0f), g(0.0f); float p = 0.0f, p_proj = 0.0f; u32 out_comp = kAbsorption;
			ev.bsdf.sample(ev.geom, z, ev.in, out_comp, out, p, p_proj, g);
			V3 out_w = g * w;                                              // compute_scattering_weights
			u32 out_vertex_info = 0xFFFFFFFFu;
			if (psf) psf_scattering_weights(ev, prev_vertex_info, vertex_info, out_comp, g, w, out_w, out_vertex_info);
			if (out_comp != kAbsorption && p != 0.0f && max_comp(out_w) > 0.0f && finite3(out_w))
			{
				PathEntry s;
				s.ray.ox = ev.geom.position.x; s.ray.oy = ev.geom.position.y; s.ray.oz = ev.geom.position.z;
				s.ray.mask_or_tmin = f2bits(1.0e-3f);
				s.ray.dx = out.x; s.ray.dy = out.y; s.ray.dz = out.z; s.ray.tmax = 1.0e8f;
				const float min_p = 32.0f;
				s.cone_x = cone_radius; s.cone_y = maxf(p, min_p);
				const u32 is_diffuse = (pi_diffuse(e.pixel_info) || (out_comp & kDiffuseMask)) ? 1u : 0u;
				s.pixel_info = pixel_info_pack(pixel_index, out_comp, is_diffuse);
				s.weight = V4(out_w.x, out_w.y, out_w.z, p);
				s.hit.t = -1.0f; s.hit.triId = -1; s.hit.u = s.hit.v = 0.0f;
				s.vertex_info = out_vertex_info;
				out_scatter.push_back(s);
			}
		}
	}

	// ---- PSFPT vertex processor (src/psfpt_vertex_processor.h) -------------------------------------------------------------------
	u32 psf_instance = 0;
	static V3 demodulate(V3 f, V3 c) { return V3(f.x / maxf(c.x, 1.0e-4f), f.y / maxf(c.y, 1.0e-4f), f.z / maxf(c.z, 1.0e-4f)); }     // src/filters.h:63-67
	// preprocess_vertex : :76-187
	u32 psf_preprocess_vertex(u32 pixel_info, const EyeVertex& ev, float cone_radius, u32 prev_vertex_info, V3 w, float p_prev)
	{
		u32 new_cache_slot = ci_slot(prev_vertex_info);
		bool new_cache_entry = false;
		if (!ci_valid(prev_vertex_info) && in_bounce >= psf->options.psf_depth && p_prev < psf->options.psf_max_prob)
		{
			const u32 pixel_hash = pi_pixel(pixel_info) + psf_instance * scene.res_x * scene.res_y;
			float jitter[6];
			for (u32 k = 0; k < 6; ++k) jitter[k] = randfloat(k, pixel_hash);
			const float cone_scale = psf->options.psf_width;
			const float filter_scale = (in_bounce == 0 ? 2.0f : 1.0f);
			const V3 N = dot(ev.in, ev.geom.normal_s) > 0.0f ? ev.geom.normal_s : -ev.geom.normal_s;
			const u64 key = spatial_hash(ev.geom.position, N, ev.geom.tangent, ev.geom.binormal, psf->bbox_lo, psf->bbox_hi, jitter, cone_radius * cone_scale, filter_scale);
			new_cache_slot = psf->insert(key);
			psf->cells[new_cache_slot].count += 1;
			const V4 md = ev.material.diffuse;
			const V4 w_mod(w.x * maxf(md.x, 1.0e-4f), w.y * maxf(md.y, 1.0e-4f), w.z * maxf(md.z, 1.0e-4f), 0.0f * maxf(md.w, 1.0e-4f));     // modulate(Vector4f(w,0), diffuse)
			const u32 comp = pi_comp(pixel_info);
			PsfState::Ref r;
			r.pixel_info = pixel_info; r.cache = cache_info(new_cache_slot, 3u, 0);
			r.w_d = (comp & kDiffuseMask) ? w_mod : V4(0, 0, 0, 0);
			r.w_g = ((comp & kGlossyMask) && in_bounce) ? w_mod : V4(0, 0, 0, 0);
			psf->refs.push_back(r);
			new_cache_entry = true;
		}
		return cache_info(new_cache_slot, 0, new_cache_entry ? 1u : 0u);
	}
	// compute_nee_weights : :189-248 (f_L already carries G and the MIS weight)
	void psf_nee_weights(const EyeVertex& ev, u32 vertex_info, V3 f_d, V3 f_g, V3 w, V3 f_L, V3& out_w_d, V3& out_w_g) const
	{
		const bool new_cache_entry = ci_new(vertex_info) != 0;
		const bool out_valid = !(in_bounce < psf->options.psf_depth) && ci_valid(cache_info(ci_slot(vertex_info), 0, 0));
		if (new_cache_entry && out_valid)
		{
			out_w_d = demodulate(f_d, ev.material.diffuse.xyz()) * f_L;
			out_w_g = f_g * w * f_L;
		}
		else
		{
			out_w_d = f_d * w * f_L;
			out_w_g = f_g * w * f_L;
		}
	}
	// compute_scattering_weights : :250-286
	void psf_scattering_weights(const EyeVertex& ev, u32 prev_vertex_info, u32 vertex_info, u32 out_comp, V3 g, V3 w, V3& out_w, u32& out_vertex_info) const
	{
		const bool new_cache_entry = ci_new(vertex_info) != 0;
		out_vertex_info = (!ci_valid(prev_vertex_info) && (out_comp & kGlossyMask)) ? prev_vertex_info : cache_info(ci_slot(vertex_info), 3u, 0);
		if (new_cache_entry && (out_comp & kDiffuseMask)) out_w = demodulate(g, ev.material.diffuse.xyz());
		else out_w = g * w;
	}
	// accumulate_emissive : :288-343
	void psf_accumulate_emissive(u32 pixel_info, u32 prev_vertex_info, V3 out_w)
	{
		const V3 c = psf->clamp_sample(out_w);
		const u32 pixel = pi_pixel(pixel_info), comp = pi_comp(pixel_info);
		if (!ci_valid(prev_vertex_info))
		{
			add_in(fb, FB_COMPOSITED_C, pixel, c, frame_weight, false);
			if (in_bounce == 0) add_in(fb, FB_DIRECT_C, pixel, c, frame_weight, false);
			else
			{
				if (comp & kDiffuseMask) add_in(fb, FB_DIFFUSE_C, pixel, c, frame_weight, true);
				if (comp & kGlossyMask)  add_in(fb, FB_SPECULAR_C, pixel, c, frame_weight, true);
			}
		}
		else psf->add(ci_slot(prev_vertex_info), c);
	}
	// accumulate_nee : :345-441
	void psf_accumulate_nee(u32 pixel_info, u32 vertex_info, bool shadow_hit, V3 w_d, V3 w_g)
	{
		if (shadow_hit) return;
		const u32 pixel = pi_pixel(pixel_info), comp = pi_comp(pixel_info);
		if (ci_valid(vertex_info))
		{
			const V3 w = (ci_comp(vertex_info) == 1u) ? w_d : w_d + w_g;
			psf->add(ci_slot(vertex_info), w);
			if (ci_comp(vertex_info) == 1u)
			{
				add_in(fb, FB_COMPOSITED_C, pixel, psf->clamp_sample(w_g), frame_weight, false);
				add_in(fb, (in_bounce == 0 || (comp & kGlossyMask)) ? FB_SPECULAR_C : FB_DIFFUSE_C, pixel, psf->clamp_sample(w_g), frame_weight, true);
			}
		}
		else
		{
			add_in(fb, FB_COMPOSITED_C, pixel, psf->clamp_sample(w_d + w_g), frame_weight, false);
			if (in_bounce == 0)
			{
				add_in(fb, FB_DIFFUSE_C, pixel, psf->clamp_sample(w_d), frame_weight, true);
				add_in(fb, FB_SPECULAR_C, pixel, psf->clamp_sample(w_g), frame_weight, true);
			}
			else
			{
				if (comp & kDiffuseMask) add_in(fb, FB_DIFFUSE_C, pixel, psf->clamp_sample(w_d + w_g), frame_weight, true);
				if (comp & kGlossyMask)  add_in(fb, FB_SPECULAR_C, pixel, psf->clamp_sample(w_d + w_g), frame_weight, true);
			}
		}
	}
	// psf_blending_kernel : src/renderers/psfpt_impl.h:86-125
	void psf_blending()
	{
		for (size_t i = 0; i < psf->refs.size(); ++i)
		{
			const PsfState::Ref& r = psf->refs[i];
			if (!ci_valid(r.cache)) continue;
			const PsfState::Cell& cell = psf->cells[ci_slot(r.cache)];
			const float cw = float(cell.count);
			const V3 cv(float(double(cell.x) * (1.0 / 4294967296.0)) / cw, float(double(cell.y) * (1.0 / 4294967296.0)) / cw, float(double(cell.z) * (1.0 / 4294967296.0)) / cw);
			const u32 pixel = pi_pixel(r.pixel_info), comp = pi_comp(r.pixel_info);
			const V3 w = ((comp & kDiffuseMask) ? r.w_d.xyz() : V3(0.0f)) + ((comp & kGlossyMask) ? r.w_g.xyz() : V3(0.0f));
			const V3 cvw = cv * w;
			const float ff = psf->options.firefly_filter;
			add_in(fb, FB_COMPOSITED_C, pixel, V3(minf(cvw.x, ff), minf(cvw.y, ff), minf(cvw.z, ff)), frame_weight, false);
			if (comp & kDiffuseMask) add_in(fb, FB_DIFFUSE_C, pixel, cv * r.w_d.xyz(), frame_weight, true);
			if (comp & kGlossyMask)  add_in(fb, FB_SPECULAR_C, pixel, cv * r.w_g.xyz(), frame_weight, true);
		}
	}
	// clamp_frame : src/renderer.cu:314-331 (all four components)
	void clamp_frame(float max_value)
	{
		const u32 np = fb.res_x * fb.res_y;
		const u32 ch[4] = { FB_DIFFUSE_C, FB_SPECULAR_C, FB_DIRECT_C, FB_COMPOSITED_C };
		for (u32 p = 0; p < np; ++p)
			for (int c = 0; c < 4; ++c)
			{
				const V4 v = fb.get(ch[c], p);
				fb.set(ch[c], p, V4(minf(v.x, max_value), minf(v.y, max_value), minf(v.z, max_value), minf(v.w, max_value)));
			}
	}

	// RTContext::trace / trace_shadow over a whole queue.  Rays are independent, so the loop may run on several host threads
	// (trace_threads; results and visit counters are identical for any thread count); the time spent here is the
	// "CUGAR host-BVH CPU trace of the same rays" figure bench.py reports as cpu_baseline.trace_mray_per_s.
	int trace_threads = 1;
	double trace_seconds = 0.0;
	// diagnostic tap (tools/diag_bpt_rays.py): every ray traced while `log_rays` is set, with its result, in trace order
	bool log_rays = false;
	std::vector<Ray> logged_rays; std::vector<Hit> logged_hits; std::vector<u32> logged_kind;
	template <typename Queue>
	void trace_queue(Queue& q, bool shadow)
	{
		struct LogAfter { PathTracer* self; Queue& q; bool shadow; ~LogAfter() { if (self->log_rays) for (size_t i = 0; i < q.size(); ++i) { self->logged_rays.push_back(q[i].ray); self->logged_hits.push_back(q[i].hit); self->logged_kind.push_back(shadow ? 1u : 0u); } } } log_after{ this, q, shadow };
		const auto t0 = std::chrono::steady_clock::now();
		const long long n = (long long)q.size();
	#ifdef _OPENMP
		if (trace_threads > 1 && n >= 4096)
		{
			u64 nv_total = 0, tt_total = 0;
			#pragma omp parallel num_threads(trace_threads) reduction(+ : nv_total, tt_total)
			{
				u64 nv = 0, tt = 0;
				const RayCaster& rc = caster;
				#pragma omp for schedule(dynamic, 1024)
				for (long long i = 0; i < n; ++i) q[size_t(i)].hit = shadow ? rc.trace_shadow(q[size_t(i)].ray, nv, tt) : rc.trace(q[size_t(i)].ray, nv, tt);
				nv_total += nv; tt_total += tt;
			}
			caster.nodes_visited += nv_total; caster.tris_tested += tt_total;
		}
		else
	#endif
		for (size_t i = 0; i < q.size(); ++i) q[i].hit = shadow ? caster.trace_shadow(q[i].ray) : caster.trace(q[i].ray);
		trace_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	}

	// shade_hits_kernel over the whole in-queue (src/pathtracer_kernels.h:189-241).  With trace_threads > 1 (the timed CPU baseline)
	// the queue is cut into one contiguous slice per thread, every thread appends to queues of its own and the slices are
	// concatenated in order: queue contents, their order and the frame buffer are identical to the sequential loop (a pass holds one
	// path per pixel, so two entries never touch the same pixel).  The path-space-filtering processor stays sequential: its cache
	// inserts and reference list are order dependent.
	double shade_seconds = 0.0;
	void shade_queue()
	{
		const auto t0 = std::chrono::steady_clock::now();
		const size_t n = in_queue.size();
		if (psf || trace_threads <= 1 || n < 4096)
			for (size_t i = 0; i < n; ++i) shade_vertex(in_queue[i], shadow_dir_queue, shadow_queue, scatter_queue);
		else
		{
			const int T = trace_threads;
			std::vector<std::vector<ShadowEntry>> ldir(T), lnee(T);
			std::vector<std::vector<PathEntry>> lsc(T);
			#pragma omp parallel num_threads(T)
			{
				#pragma omp for schedule(static, 1)
				for (int t = 0; t < T; ++t)
				{
					const size_t b = n * size_t(t) / size_t(T), e = n * size_t(t + 1) / size_t(T);
					for (size_t i = b; i < e; ++i) shade_vertex(in_queue[i], ldir[t], lnee[t], lsc[t]);
				}
			}
			for (int t = 0; t < T; ++t)
			{
				shadow_dir_queue.insert(shadow_dir_queue.end(), ldir[t].begin(), ldir[t].end());
				shadow_queue.insert(shadow_queue.end(), lnee[t].begin(), lnee[t].end());
				scatter_queue.insert(scatter_queue.end(), lsc[t].begin(), lsc[t].end());
			}
		}
		shade_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	}

	// src/pathtracer_kernels.h:309-391 + src/renderers/pathtracer_impl.h:197-324 for one pass
	void render_pass(u32 instance, const u32* pixels, u32 n_pixels)
	{
		rescale_frame(instance, pixels, n_pixels);
		sequence.set_instance(instance);
		frame_weight = 1.0f / float(instance + 1);
		stats.clear(); captured.clear();
		if (psf)
		{
			if ((instance % psf->options.psf_temporal_reuse) == 0) psf->clear();        // src/renderers/psfpt_impl.h:385-387
			psf->refs.clear();
			psf_instance = instance;
		}
		generate_primary_rays(pixels, n_pixels);
		for (in_bounce = 0; in_bounce < options.max_path_length; ++in_bounce)
		{
			if (in_queue.empty()) break;
			compute_per_bounce_options();
			trace_queue(in_queue, false);
			rays_traced += in_queue.size();
			if (int(in_bounce) == capture_bounce) captured = in_queue;
			shadow_dir_queue.clear(); shadow_queue.clear(); scatter_queue.clear();
			shade_queue();
			trace_queue(shadow_dir_queue, true);
			trace_queue(shadow_queue, true);
			shadow_rays_traced += shadow_dir_queue.size() + shadow_queue.size();
			// solve_occlusion : src/pathtracer_core.h:705-738
			for (std::vector<ShadowEntry>* q : { &shadow_dir_queue, &shadow_queue })
				for (size_t i = 0; i < q->size(); ++i)
				{
					const ShadowEntry& s = (*q)[i];
					if (psf) psf_accumulate_nee(s.pixel_info, s.vertex_info, s.hit.t > 0.0f, s.w_d, s.w_g);
					else accumulate_nee(s.pixel_info, s.hit.t > 0.0f, s.w_d, s.w_g);
				}
			BounceStats bs; bs.in_size = u32(in_queue.size()); bs.shadow_dir_size = u32(shadow_dir_queue.size()); bs.shadow_size = u32(shadow_queue.size()); bs.scatter_size = u32(scatter_queue.size());
			stats.push_back(bs);
			in_queue.swap(scatter_queue);
		}
		if (psf) psf_blending();
		update_variances(instance, pixels, n_pixels);
		if (psf) clamp_frame(100.0f);                                                    // PSFPT::render, src/renderers/psfpt_impl.h:283
	}
};

} // namespace orc
