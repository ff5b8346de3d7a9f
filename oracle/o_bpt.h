// TEST INFRASTRUCTURE ONLY: CPU restatement of Fermat's bidirectional path tracer (`-bpt`): the default
// single-connection mode (`-sc 1`: VertexOrdering::kRandomOrdering, src/renderers/bpt.h:57-62, bpt_impl.h:92) and the all-connections
// mode (`-sc 0`: VertexOrdering::kPathOrdering), both with VertexSampling::kAll.  Parity unpinned at image level.
//
//   BPT::init / BPT::render, BPTConfig, ConnectionsSink             src/renderers/bpt.cu:33-111 ; src/renderers/bpt_impl.h:55-260
//   sample_light_subpaths / sample_eye_subpaths / light_tracing     src/bpt_control.h:290-600
//   generate_primary_light_vertex .. solve_occlusion                src/bpt_kernels.h:275-1000
//   PathWeights, TempPathWeights, pack/unpack, Light/EyeVertex,
//   eval_connection, eval_incoming_emission, scatter, bpt_mis       src/bpt_utils.h:57-1100
//   primary sample coordinates                                      src/bpt_samplers.h:43-88 ; src/tiled_sequence.h:53-107
//   pack_direction / unpack_direction                               src/vertex.h:123-140
//   to_rgbe / from_rgbe                                             contrib/cugar/color/rgbe.h:35-75
//   camera_direction_pdf (3 variants), square_screen_focal_length   src/camera.h:130-252
//
// Where the reference is undefined this file says what it does instead (each marked "DEFINED HERE"):
//   * `-sc 1` stores light vertices at slots handed out by an atomic counter (src/bpt_kernels.h:493-501): the order inside one depth is
//     whatever the GPU's scheduling made it, and the eye vertices pick "vertex number quantize(z, n)" from that list (:714-733), so the
//     reference's image depends on scheduling.  The host dispatchers record the counter after every depth (:1181-1183, :1209-1211), so
//     the list is depth-major by construction; DEFINED HERE: inside a depth, vertices are ordered by light-path id.  (The vertices
//     themselves are kept in the path-ordered store; `flat` lists their slots in that order.)  Light tracing reads
//     vertex_counts[max_path_length - 1] as the list length (:1086-1100), which is stale when the light loop ends early
//     (src/bpt_control.h:343-345); DEFINED HERE: every stored vertex is connected to the lens, as in `-sc 0`.
//   * Bsdf's second constructor (stored light vertices) reads an uninitialised m_reflectivity (src/bsdf.h:248-273) -> zero.
//   * ConnectionsSink<false> adds with plain read-modify-writes from concurrent threads (src/renderers/bpt_impl.h:157-163) ->
//     here every pixel's contributions of a bounce are added in a fixed order: albedo, emission, then connections by light depth.
//   * light-tracing splats are float atomics in the reference (order-dependent rounding, :141-155) -> here each splat is rounded to
//     2^-32 fixed point and summed in 64-bit integers (order-independent), added to the frame once per pass.
#pragma once
#include "o_pt.h"

namespace orc {

struct BPTOptions
{
	u32 max_path_length;
	u32 direct_lighting_nee, direct_lighting_bsdf, indirect_lighting_nee, indirect_lighting_bsdf, visible_lights, use_vpls, rr;
	float light_tracing;
	u32 single_connection;     // BPTOptions::single_connection (src/renderers/bpt.h:57-70), `-sc`; the reference's default is 1
};

static const float SHADOW_BIAS = 1.0e-4f;      // src/renderer_view.h:44-45
static const float SHADOW_TMIN = 0.0f;
static const float MIN_G_DENOM = 1.0e-8f;      // src/bpt_utils.h:49

// ---- src/bpt_utils.h:57-99 --------------------------------------------------------------------------------------------------
inline float bpt_mis(float pGp, float prev_pGp, float next_pGp, float pGp_sum)
{ return pGp && prev_pGp && next_pGp ? (1 / pGp) / ((1 / pGp) + (1 / prev_pGp) + (1 / next_pGp) + pGp_sum) : 0.0f; }
inline float bpt_mis(float pGp, float other_pGp, float pGp_sum)
{ return pGp && other_pGp ? (1 / pGp) / ((1 / pGp) + (1 / other_pGp) + pGp_sum) : 0.0f; }
inline float pdf_product3(float p1, float p2, float p3) { return finite_f(p1) && finite_f(p2) && finite_f(p3) ? p1 * p2 * p3 : finf(); }

struct PathWeights { float pGp_sum, pG; };
struct TempPathWeights { float pGp_sum, pG, out_p, out_cos_theta; };

// ---- contrib/cugar/color/rgbe.h -----------------------------------------------------------------------------------------------
inline u32 to_rgbe(V3 c)
{
	float v = 0;
	if (c.x > v) v = c.x;
	if (c.y > v) v = c.y;
	if (c.z > v) v = c.z;
	u32 x = f2bits(v);
	const int exponent = int(((x >> 23u) & 0xFF) - 126u);
	int rgbe = int((u32(exponent) + 128u) & 0xFF);
	if (rgbe < 10) return 0;
	x = ((((u32(rgbe) & 0xFF) - (128u + 8u)) + 127u) << 23u) & 0x7F800000u;
	const float f = 1.0f / bits2f(x);
	u32 r = u32(rgbe);
	r |= (f2u(c.x * f) << 24);
	r |= (f2u(c.y * f) << 16);
	r |= (f2u(c.z * f) << 8);
	return r;
}
inline V3 from_rgbe(u32 rgbe)
{
	const u32 x = (((rgbe & 0xFF) - 9u) << 23u) & 0x7F800000u;
	const float f = bits2f(x);
	return V3(f * float(rgbe >> 24), f * float((rgbe >> 16) & 0xFF), f * float((rgbe >> 8) & 0xFF));
}
// ---- src/vertex.h:123-140 -------------------------------------------------------------------------------------------------------
inline V2 uniform_sphere_to_square(V3 v)
{
	float phi;
	if (fabsf(v.z) >= 1.0f - 1.0e-5f) phi = 0.0f;
	else { phi = det_atan2(v.y, v.x); phi = phi < 0.0f ? phi + 2.0f * PI_F : phi; }
	V2 r; r.x = phi / (2.0f * PI_F); r.y = (v.z + 1.0f) * 0.5f;
	return r;
}
inline V3 uniform_square_to_sphere(float ux, float uy)
{
	const float cosTheta = uy * 2.0f - 1.0f;
	const float sinTheta = sqrtf(fmax_ieee(1.0f - cosTheta * cosTheta, 0.0f));
	float s, c; det_sincos(ux * (2.0f * PI_F), &s, &c);
	return V3(c * sinTheta, s * sinTheta, cosTheta);
}
inline u32 pack_direction(V3 dir) { const V2 s = uniform_sphere_to_square(dir); return quantize(s.x, 0xFFFFu) + (quantize(s.y, 0xFFFFu) << 16); }
inline V3 unpack_direction(u32 p) { return uniform_square_to_sphere(float(p & 0xFFFFu) / float(0xFFFFu), float((p >> 16) & 0xFFFFu) / float(0xFFFFu)); }

// ---- src/bpt_utils.h:185-260 : the 16-byte material record of a stored light vertex -----------------------------------------------
struct PackedBsdf { u32 x, y, z, w; };
inline PackedBsdf pack_bsdf(const Material& m)
{
	const u32 roughness_i = u32(uint16_t(quantize(m.roughness, 65535u)));
	const u32 opacity_i   = u32(uint16_t(quantize(m.opacity, 255u)));
	const u32 ior_i       = u32(uint16_t(quantize(m.index_of_refraction / 3.0f, 255u)));
	PackedBsdf r;
	r.x = to_rgbe(m.diffuse.xyz()); r.y = to_rgbe(m.specular.xyz()); r.z = roughness_i | (opacity_i << 16) | (ior_i << 24); r.w = to_rgbe(m.diffuse_trans.xyz());
	return r;
}
inline void unpack_bsdf(const PackedBsdf& p, const float* table, Bsdf& b)
{
	const float roughness = float(p.z & 65535u) / 65535.0f;
	const float opacity = float((p.z >> 16) & 255u) / 255.0f;
	const float ior = maxf(3.0f * (float(p.z >> 24) / 255.0f), 0.00001f);
	b.setup_unpacked(from_rgbe(p.x), from_rgbe(p.y), roughness, from_rgbe(p.w), opacity, ior, table, true);      // kParticleTransport default, :244
}

// ---- src/camera.h ---------------------------------------------------------------------------------------------------------------
inline float square_screen_focal_length(const Camera& c) { const float t = tanf(c.fov / 2); return (1.0f / 4.0f) / (t * t); }      // :132-136, host libm
// :206-227 (pixel coordinates out, projected-solid-angle value) and :232-252 (projected flag)
inline float camera_direction_pdf_xy(V3 U, V3 V, V3 W, float W_len, float sq_focal, V3 out, float* out_x, float* out_y, bool projected = true)
{
	const float t = dot(out, W) / (W_len * W_len);
	if (t < 0.0f) return 0.0f;
	const V3 I = out / t - W;
	const float Ix = dot(I, U) / dot(U, U);
	const float Iy = dot(I, V) / dot(V, V);
	if (Ix >= -1.0f && Ix <= 1.0f && Iy >= -1.0f && Iy <= 1.0f)
	{
		if (out_x) *out_x = Ix;
		if (out_y) *out_y = Iy;
		const float cos_theta = dot(out, W) / W_len;
		return projected ? sq_focal / (cos_theta * cos_theta * cos_theta * cos_theta) : sq_focal / (cos_theta * cos_theta * cos_theta);
	}
	return 0.0f;
}

// Edf sampling : contrib/cugar/bsdf/lambert_edf.h:82-99
inline void edf_sample(const Edf& e, float u0, float u1, const Frame& g, V3& out, V3& gg, float& p, float& p_proj)
{
	const V3 l = square_to_cosine_hemisphere(u0, u1);
	out = l.x * g.tangent + l.y * g.binormal + l.z * g.normal_s;
	gg = e.color * PI_F;
	p = l.z / PI_F;
	p_proj = 1.0f / PI_F;
}

// ---- vertices : src/bpt_utils.h:311-700 ------------------------------------------------------------------------------------------
struct BptVertex            // the fields LightVertex and EyeVertex share
{
	u32 prim_id; float uv_u, uv_v;
	VertexGeometry geom;
	V3 in, alpha;
	u32 depth;
	Material material;
	Edf edf;
	Bsdf bsdf;
	float prev_G_prime, prev_pG, pGp_sum;
	PathWeights weights;       // light vertices
	TempPathWeights tweights;  // eye vertices

	void shade(const Ray& ray, const Hit& hit, const SceneView& r, bool particle)
	{
		prim_id = u32(hit.triId); uv_u = hit.u; uv_v = hit.v;
		setup_differential_geometry(r.mesh, u32(hit.triId), hit.u, hit.v, &geom);
		geom.position = V3(ray.ox, ray.oy, ray.oz) + hit.t * V3(ray.dx, ray.dy, ray.dz);
		material = r.mesh.materials[r.mesh.material_indices[hit.triId]];
		const V4 one(1, 1, 1, 1);
		material.diffuse       = material.diffuse       * bilinear_texture_lookup(geom.texture_coords, material.diffuse_map, r.textures, one);
		material.specular      = material.specular      * bilinear_texture_lookup(geom.texture_coords, material.specular_map, r.textures, one);
		material.emissive      = material.emissive      * bilinear_texture_lookup(geom.texture_coords, material.emissive_map, r.textures, one);
		material.diffuse_trans = material.diffuse_trans * bilinear_texture_lookup(geom.texture_coords, material.diffuse_trans_map, r.textures, one);
		in = -normalize(V3(ray.dx, ray.dy, ray.dz));
		bsdf.setup(material, r.glossy_reflectance);
		bsdf.particle_transport = particle;
	}
	// LightVertex::setup(ray, hit, alpha, TempPathWeights, depth) : :340-361
	void setup_light(const Ray& ray, const Hit& hit, V3 _alpha, const TempPathWeights& w, u32 _depth, const SceneView& r)
	{
		shade(ray, hit, r, true);
		alpha = _alpha; depth = _depth; weights.pGp_sum = w.pGp_sum; weights.pG = w.pG;
		prev_G_prime = fabsf(dot(in, geom.normal_s)) / fmax_ieee(hit.t * hit.t, MIN_G_DENOM);
		prev_pG = pdf_product(w.out_p, w.out_cos_theta * prev_G_prime);
		pGp_sum = w.pGp_sum + (1 / pdf_product(w.pG, w.out_p));
	}
	// LightVertex::setup(pos, packed info...) : :313-337
	void setup_stored(const float* pos, u32 packed_in, u32 packed_alpha, const PackedBsdf& gb, PathWeights w, u32 _depth, const SceneView& r)
	{
		in = unpack_direction(packed_in);
		alpha = from_rgbe(packed_alpha);
		weights = w; depth = _depth;
		geom.position = V3(pos[0], pos[1], pos[2]);
		geom.normal_s = unpack_direction(f2bits(pos[3]));
		geom.normal_g = geom.normal_s;
		geom.tangent = orthogonal(geom.normal_s);
		geom.binormal = cross(geom.normal_s, geom.tangent);
		if (depth == 0) edf.color = from_rgbe(gb.x);
		else unpack_bsdf(gb, r.glossy_reflectance, bsdf);
	}
	// EyeVertex::setup(ray, hit, alpha, TempPathWeights, depth) : :585-642
	// NB hit.t is in units of |ray.dir| and the primary eye rays are not normalised (src/bpt_kernels.h:569-572), so at the first eye vertex the
	// reference's G' is 1/cos^2 of the camera angle too large: one of the two origins of the BPT's energy excess with light tracing on
	// (DESIGN.md 3).  `true_distance` is the what-if switch of tests/test_oracle_statistics.py, never set by a renderer.
	void setup_eye(const Ray& ray, const Hit& hit, V3 _alpha, const TempPathWeights& w, u32 _depth, const SceneView& r, bool true_distance = false)
	{
		shade(ray, hit, r, false);
		alpha = _alpha; depth = _depth; tweights = w;
		const float len2 = true_distance ? (ray.dx * ray.dx + ray.dy * ray.dy) + ray.dz * ray.dz : 1.0f;
		prev_G_prime = fabsf(dot(in, geom.normal_s)) / (hit.t * hit.t * len2);
		prev_pG = pdf_product(w.out_p, w.out_cos_theta * prev_G_prime);
		pGp_sum = w.pGp_sum + (1 / pdf_product(w.pG, w.out_p));
	}
};

// scatter : src/bpt_utils.h:1068-1100 (output_alpha = true, evaluate_full_bsdf = BPT_FULL_BSDF_EVALUATION = 1)
inline bool bpt_scatter(const BptVertex& v, const float z[3], u32& comp, V3& out, float& p, float& p_proj, V3& out_w, bool RR)
{
	const bool s = v.bsdf.sample_ex(v.geom, z, v.in, comp, out, p, p_proj, out_w, RR, true);
	out_w = out_w * v.alpha;
	return s;
}

// eval_connection : src/bpt_utils.h:911-980 (mis_selector is the identity: DEBUG_S = DEBUG_T = -1)
inline void eval_connection(const BptVertex& ev, const BptVertex& lv, V3& out, V3& out_w, float& d, bool RR, bool direct_lighting_nee, bool direct_lighting_bsdf)
{
	const V3 delta = lv.geom.position - ev.geom.position;
	const float d2 = fmax_ieee(MIN_G_DENOM, dot(delta, delta));
	d = sqrtf(d2);
	out = delta / d;
	const float G = fabsf(dot(out, ev.geom.normal_s) * dot(out, lv.geom.normal_s)) / d2;
	V3 f_s; float p_s;
	ev.bsdf.f_and_p_sum(ev.geom, ev.in, out, f_s, p_s, RR);
	const float prev_pGp = pdf_product(ev.prev_pG, p_s);
	if (lv.depth == 0)
	{
		if (direct_lighting_nee == false) { out_w = V3(0.0f); return; }
		const V3 f_L = lv.edf.f(lv.geom, -out);
		const float p_L = 1.0f / PI_F;                                           // LambertEdf::p, projected solid angle
		const float pGp = pdf_product3(p_s, G, p_L);
		const float next_pGp = pdf_product(p_L, lv.weights.pG);
		const float mis_w = (ev.depth == 0 && direct_lighting_bsdf == false) ? 1.0f : bpt_mis(pGp, prev_pGp, next_pGp, ev.pGp_sum + lv.weights.pGp_sum);
		out_w = ev.alpha * lv.alpha * f_L * f_s * G * mis_w;
	}
	else
	{
		V3 f_L; float p_L;
		lv.bsdf.f_and_p_sum(lv.geom, lv.in, -out, f_L, p_L, RR);
		const float pGp = pdf_product3(p_s, G, p_L);
		const float next_pGp = pdf_product(p_L, lv.weights.pG);
		const float mis_w = bpt_mis(pGp, prev_pGp, next_pGp, ev.pGp_sum + lv.weights.pGp_sum);
		out_w = ev.alpha * lv.alpha * f_L * f_s * G * mis_w;
	}
}

struct BPT
{
	BPTOptions options;
	PathTracer* host;               // scene, BVH and frame buffer are the rendering context's
	TiledSequence sequence;
	u32 n_light_paths, n_eye_paths;
	float light_tracing;            // options.light_tracing * n_light_paths / n_pixels (src/renderers/bpt.cu:75)
	u32 whatif_consistent_mis = 0;  // TEST-ONLY what-if switches (orc_bpt_set_whatif): 1 = true distance at the first eye vertex, 2 = reverse pdf in connect_to_camera
	V3 U, V, W; float W_len, sq_focal;
	float frame_weight;
	// VertexStorage, path ordering: slot = path + depth * n_light_paths (src/vertex_storage.h, src/bpt_kernels.h:496-500)
	std::vector<float> v_pos, v_weights;
	std::vector<u32> v_input, v_path_id, v_counts;
	std::vector<PackedBsdf> v_gbuffer;
	// queues
	struct Entry { Ray ray; Hit hit; V4 w; float prob; u32 pixel; TempPathWeights pw; };
	struct Shadow { Ray ray; Hit hit; V4 w; u32 pixel; u32 light_path_id; };
	std::vector<Entry> in_queue, scatter_queue;
	std::vector<Shadow> shadow_queue;
	std::vector<long long> splat;   // 6 per pixel: COMPOSITED xyz, DIRECT xyz in 2^-32 fixed point
	std::vector<u32> flat;          // -sc 1: slots of the stored light vertices, depth-major, light-path id minor
	u32 n_flat_primary = 0;         // -sc 1: vertex_counts[0], the number of primary light vertices
	struct Stats { u32 light_queue[32], eye_queue[32], shadow_eye[32], n_light_vertices, shadow_light_tracing, n_bounces_light, n_bounces_eye; } stats;

	SceneView& scene() const { return host->scene; }
	FrameBuffer& fb() const { return host->fb; }

	void init(PathTracer* h, const BPTOptions& o, const char* samples_dir)
	{
		host = h; options = o;
		const u32 n_pixels = scene().res_x * scene().res_y;
		n_light_paths = n_eye_paths = n_pixels;
		light_tracing = options.light_tracing * (float(n_light_paths) / float(n_pixels));
		// the context's 72-dimensional sequence first, then this renderer's (src/renderer.cu:949-953, src/renderers/bpt.cu:83-87)
		MsvcRand rng;
		{ TiledSequence ctx; ctx.setup(72, 256, samples_dir, rng); }
		sequence.setup((options.max_path_length + 1) * 2 * 6, 256, samples_dir, rng);
		const size_t nv = size_t(n_light_paths) * options.max_path_length;
		v_pos.assign(nv * 4, 0.0f); v_weights.assign(nv * 2, 0.0f); v_input.assign(nv * 2, 0u); v_path_id.assign(nv, 0xFFFFFFFFu);
		v_gbuffer.assign(nv, PackedBsdf{ 0, 0, 0, 0 }); v_counts.assign(n_light_paths, 0u);
		camera_frame(scene().camera, scene().aspect, U, V, W);
		W_len = length(W);
		sq_focal = square_screen_focal_length(scene().camera);
		splat.assign(size_t(n_pixels) * 6, 0);
	}

	// primary coordinates : src/bpt_samplers.h:43-88
	float light_sample(u32 idx, u32 vertex, u32 dim) const
	{
		const size_t T = size_t(sequence.tile_size) * sequence.tile_size;
		return sequence.samples[(vertex * 3 + dim) * T + (idx & (T - 1))];
	}
	float eye_sample(u32 idx, u32 vertex, u32 dim) const
	{
		const u32 px = idx % scene().res_x, py = idx / scene().res_x;
		if (vertex == 1 && dim < 2)
			return dim == 0 ? (float(px) + sequence.sample_2d(px, py, dim)) / float(scene().res_x) : (float(py) + sequence.sample_2d(px, py, dim)) / float(scene().res_y);
		return sequence.sample_2d(px, py, (vertex - 1) * 6 + dim);
	}
	bool terminate(u32 s) const { return s >= options.max_path_length + 1; }

	// ConnectionsSink<false>::sink (src/renderers/bpt_impl.h:131-163)
	void sink(u32 channel, V4 value, u32 pixel)
	{
		FrameBuffer& f = fb();
		f.set(FB_COMPOSITED_C, pixel, f.get(FB_COMPOSITED_C, pixel) + value * frame_weight);
		if (channel != FB_COMPOSITED_C) f.set(channel, pixel, f.get(channel, pixel) + value * frame_weight);
	}

	static Ray make_ray(V3 o, float tmin, V3 d, float tmax)
	{ Ray r; r.ox = o.x; r.oy = o.y; r.oz = o.z; r.mask_or_tmin = f2bits(tmin); r.dx = d.x; r.dy = d.y; r.dz = d.z; r.tmax = tmax; return r; }

	// generate_primary_light_vertex : src/bpt_kernels.h:275-392
	void generate_primary_light_vertex(u32 id)
	{
		v_counts[id] = 0; v_path_id[id] = 0xFFFFFFFFu;
		if (terminate(0)) return;
		float samples[3];
		for (u32 i = 0; i < 3; ++i) samples[i] = light_sample(id, 0, i);
		u32 prim; float u, v, pdf; VertexGeometry geom; Edf edf;
		if (options.use_vpls)
		{
			const MeshLight& L = scene().mesh_vpls;           // BPT::init copies the VPL array into the vertex store (:101-102)
			prim = L.vpls[id].prim_id; u = L.vpls[id].u; v = L.vpls[id].v;
			L.map(prim, u, v, &geom, &pdf, &edf);
		}
		else scene().mesh_light.sample(samples, &prim, &u, &v, &geom, &pdf, &edf);
		const bool term = terminate(1);
		{
			const u32 slot = id;
			const u32 packed_normal = pack_direction(geom.normal_s);
			v_gbuffer[slot] = PackedBsdf{ to_rgbe(edf.color), 0, 0, 0 };
			v_pos[4 * slot] = geom.position.x; v_pos[4 * slot + 1] = geom.position.y; v_pos[4 * slot + 2] = geom.position.z; v_pos[4 * slot + 3] = bits2f(packed_normal);
			v_input[2 * slot] = 0; v_input[2 * slot + 1] = to_rgbe(V3(1.0f) / pdf);
			v_weights[2 * slot] = 0.0f; v_weights[2 * slot + 1] = 1.0f * pdf;
			v_path_id[slot] = id;
			v_counts[id] = 1;
		}
		if (!term)
		{
			for (u32 i = 0; i < 3; ++i) samples[i] = light_sample(id, 1, i);
			V3 out, g; float p, p_proj;
			edf_sample(edf, samples[0], samples[1], geom, out, g, p, p_proj);
			g = g / pdf;
			Entry e;
			e.ray = make_ray(geom.position, 1.0e-4f, out, 1.0e8f);
			e.w = V4(g.x, g.y, g.z, 0.0f); e.prob = p; e.pixel = pixel_info_pack(id, FB_DIFFUSE_C, 0);
			e.pw.pGp_sum = 0.0f; e.pw.pG = 1.0f * pdf; e.pw.out_p = p_proj; e.pw.out_cos_theta = fabsf(dot(geom.normal_s, out));
			scatter_queue.push_back(e);
		}
	}

	// process_secondary_light_vertex : src/bpt_kernels.h:394-521
	void process_secondary_light_vertex(const Entry& q, u32 in_bounce)
	{
		const u32 id = pi_pixel(q.pixel);
		if (!(q.hit.t > 0.0f && q.hit.triId >= 0)) return;
		BptVertex lv;
		lv.setup_light(q.ray, q.hit, q.w.xyz(), q.pw, in_bounce + 1, scene());
		if (!terminate(in_bounce + 2))
		{
			float z[3];
			for (u32 i = 0; i < 3; ++i) z[i] = light_sample(id, in_bounce + 2, i);
			V3 out(0.0f), out_w(0.0f); float p = 0, p_proj = 0; u32 comp = kAbsorption;
			bpt_scatter(lv, z, comp, out, p, p_proj, out_w, options.rr != 0);
			if (max_comp(out_w) > 0.0f)
			{
				Entry e;
				e.ray = make_ray(lv.geom.position, 1.0e-4f, out, 1.0e8f);
				e.pixel = q.pixel; e.w = V4(out_w.x, out_w.y, out_w.z, q.w.w); e.prob = 0.0f;
				e.pw.pGp_sum = lv.pGp_sum; e.pw.pG = lv.prev_pG; e.pw.out_p = p_proj; e.pw.out_cos_theta = fabsf(dot(lv.geom.normal_s, out));
				scatter_queue.push_back(e);
			}
		}
		{
			const u32 slot = id + v_counts[id] * n_light_paths;
			v_gbuffer[slot] = pack_bsdf(lv.material);
			v_pos[4 * slot] = lv.geom.position.x; v_pos[4 * slot + 1] = lv.geom.position.y; v_pos[4 * slot + 2] = lv.geom.position.z;
			v_pos[4 * slot + 3] = bits2f(pack_direction(lv.geom.normal_s));
			v_input[2 * slot] = pack_direction(lv.in); v_input[2 * slot + 1] = to_rgbe(q.w.xyz());
			v_weights[2 * slot] = lv.pGp_sum; v_weights[2 * slot + 1] = lv.prev_pG;
			v_path_id[slot] = id | ((in_bounce + 1) << 24);
			v_counts[id]++;
		}
	}

	void trace_entries(std::vector<Entry>& q) { host->trace_queue(q, false); }

	// sample_light_subpaths : src/bpt_control.h:290-350
	// tile sharding: the sub-paths (light AND eye) of a subset of the pixels; NULL = all
	const u32* shard_pixels = nullptr; u32 shard_count = 0;
	bool deferred_splats = false;
	u32 n_shard() const { return shard_pixels ? shard_count : n_light_paths; }
	u32 shard_id(u32 i) const { return shard_pixels ? shard_pixels[i] : i; }

	void sample_light_subpaths()
	{
		scatter_queue.clear();
		for (u32 i = 0; i < n_shard(); ++i) generate_primary_light_vertex(shard_id(i));
		in_queue.swap(scatter_queue);
		stats.n_bounces_light = 0;
		for (u32 in_bounce = 0; in_bounce + 1 < options.max_path_length; ++in_bounce)
		{
			if (in_queue.empty()) break;
			stats.light_queue[stats.n_bounces_light++] = u32(in_queue.size());
			scatter_queue.clear();
			trace_entries(in_queue);
			for (size_t i = 0; i < in_queue.size(); ++i) process_secondary_light_vertex(in_queue[i], in_bounce);
			in_queue.swap(scatter_queue);
		}
		stats.n_light_vertices = 0;
		for (u32 i = 0; i < n_shard(); ++i) stats.n_light_vertices += v_counts[shard_id(i)];
	}

	// generate_primary_eye_vertex : src/bpt_kernels.h:523-571
	void generate_primary_eye_vertex(u32 idx)
	{
		const float ux = eye_sample(idx, 1, 0), uy = eye_sample(idx, 1, 1);
		const float dx = ux * 2.f - 1.f, dy = uy * 2.f - 1.f;
		const V3 dir = dx * U + dy * V + W;
		Entry e;
		e.ray = make_ray(scene().camera.eye, 0.0f, dir, 1e34f);
		e.pixel = idx; e.w = V4(1, 1, 1, 1); e.prob = 0.0f;
		const float p_e = camera_direction_pdf_xy(U, V, W, W_len, sq_focal, normalize(dir), 0, 0, true);
		const float cos_theta = dot(normalize(dir), W) / W_len;
		e.pw.pGp_sum = 0.0f; e.pw.pG = 1.0e8f;
		e.pw.out_p = light_tracing ? p_e / light_tracing : 1.0f;
		e.pw.out_cos_theta = light_tracing ? cos_theta : 1.0e8f;
		in_queue.push_back(e);
	}

	// eval_incoming_emission : src/bpt_utils.h:1031-1066
	V3 eval_incoming_emission(const BptVertex& ev)
	{
		const MeshLight& L = options.use_vpls ? scene().mesh_vpls : scene().mesh_light;
		float light_pdf; Edf light_edf;
		L.map_geom(ev.prim_id, ev.geom, &light_pdf, &light_edf);
		const V3 f_L = light_edf.f(ev.geom, ev.in);
		const float p_L = 1.0f / PI_F;
		const float pGp = pdf_product(p_L, light_pdf);
		const float prev_pGp = pdf_product(ev.prev_pG, p_L);
		const float mis_w = (ev.depth == 0 || pGp == 0.0f || (ev.depth == 1 && !options.direct_lighting_nee) || (ev.depth > 1 && !options.indirect_lighting_nee)) ? 1.0f
			: bpt_mis(pGp, prev_pGp, ev.pGp_sum);
		return ev.alpha * f_L * mis_w;
	}

	// process_secondary_eye_vertex : src/bpt_kernels.h:573-897 (path ordering, all connections)
	void process_secondary_eye_vertex(const Entry& q, u32 in_bounce)
	{
		const u32 pixel = pi_pixel(q.pixel);
		if (!(q.hit.t > 0.0f && q.hit.triId >= 0)) return;
		BptVertex ev;
		ev.setup_eye(q.ray, q.hit, q.w.xyz(), q.pw, in_bounce, scene(), (whatif_consistent_mis & 1u) != 0);
		// BPTConfig::visit_eye_vertex (src/renderers/bpt_impl.h:96-113)
		if (in_bounce + 1 == 1 && fb().gb_geo)
		{
			float* g = fb().gb_geo + 4 * size_t(pixel);
			g[0] = ev.geom.position.x; g[1] = ev.geom.position.y; g[2] = ev.geom.position.z; g[3] = pack_geometry_normal(ev.geom.normal_s);
			float* uv = fb().gb_uv + 4 * size_t(pixel);
			uv[0] = q.hit.u; uv[1] = q.hit.v; uv[2] = ev.geom.texture_coords.x; uv[3] = ev.geom.texture_coords.y;
			fb().gb_tri[pixel] = u32(q.hit.triId);
		}
		if (!terminate(in_bounce + 2))
		{
			float z[3];
			for (u32 i = 0; i < 3; ++i) z[i] = eye_sample(pixel, in_bounce + 2, i);
			V3 out(0.0f), out_w(0.0f); float p = 0, p_proj = 0; u32 comp = kAbsorption;
			bpt_scatter(ev, z, comp, out, p, p_proj, out_w, options.rr != 0);
			if (max_comp(out_w) > 0.0f)
			{
				// sink_eye_scattering_event : albedo of the visible surface (src/renderers/bpt_impl.h:167-186)
				if (in_bounce + 2 == 2)
				{
					const V4 value(out_w.x, out_w.y, out_w.z, q.w.w);
					if (comp == kDiffuseReflection) fb().set(FB_DIFFUSE_A, pixel, fb().get(FB_DIFFUSE_A, pixel) + value * frame_weight);
					else if (comp == kGlossyReflection) fb().set(FB_SPECULAR_A, pixel, fb().get(FB_SPECULAR_A, pixel) + value * frame_weight);
				}
				Entry e;
				e.ray = make_ray(ev.geom.position, 1.0e-4f, out, 1.0e8f);
				e.pixel = in_bounce ? q.pixel : pixel_info_pack(pixel, (comp & kDiffuseMask) ? FB_DIFFUSE_C : FB_SPECULAR_C, 0);
				e.w = V4(out_w.x, out_w.y, out_w.z, q.w.w); e.prob = p;
				e.pw.pGp_sum = ev.pGp_sum; e.pw.pG = ev.prev_pG; e.pw.out_p = p_proj; e.pw.out_cos_theta = fabsf(dot(ev.geom.normal_s, out));
				scatter_queue.push_back(e);
			}
		}
		// emission along the incoming direction is sunk at once ...
		const u32 t = in_bounce + 2;
		const bool emissive = (t == 2 && options.visible_lights) || (t == 3 && options.direct_lighting_bsdf) || (t > 3 && options.indirect_lighting_bsdf);
		// ... connections are queued and resolved after the whole bounce has been processed; per pixel the order is emission, connections
		V3 emission(0.0f);
		if (emissive) emission = eval_incoming_emission(ev);
		if (emissive && max_comp(emission) > 0.0f && finite_f(emission.x) && finite_f(emission.y) && finite_f(emission.z))
			sink(pi_comp(q.pixel), V4(emission.x, emission.y, emission.z, q.w.w), pixel);
		const i32 max_light_depth = i32(options.max_path_length + 1) - i32(ev.depth) - 2 - 1;
		const bool connect = (t == 1 && options.direct_lighting_nee) || (t > 1 && options.indirect_lighting_nee);
		if (max_light_depth >= 0 && connect && options.single_connection)
		{
			// one connection per eye vertex: a light vertex picked uniformly from ALL stored vertices, weighted by
			// #vertices / #light paths (src/bpt_kernels.h:714-760); z[0], z[1] of the sample triple are fetched and unused there
			const u32 n_light_vertices = u32(flat.size());
			if (n_light_vertices)
			{
				const float z2 = eye_sample(pixel, in_bounce + 2, 5);
				const u32 li = flat[quantize(z2, n_light_vertices)];
				const float light_weight = float(n_light_vertices) / float(n_flat_primary);
				const u32 light_vertex_id = v_path_id[li];
				if (light_vertex_id != 0xFFFFFFFFu)
				{
					const u32 light_depth = light_vertex_id >> 24;
					BptVertex lv;
					PathWeights lw; lw.pGp_sum = v_weights[2 * li]; lw.pG = v_weights[2 * li + 1];
					lv.setup_stored(&v_pos[4 * li], v_input[2 * li], v_input[2 * li + 1], v_gbuffer[li], lw, light_depth, scene());
					V3 out, out_w; float d;
					eval_connection(ev, lv, out, out_w, d, options.rr != 0, options.direct_lighting_nee != 0, options.direct_lighting_bsdf != 0);
					out_w = out_w * light_weight;
					if (max_comp(out_w) > 0.0f && finite_f(out_w.x) && finite_f(out_w.y) && finite_f(out_w.z))
					{
						Shadow s;
						const V3 origin = ev.geom.position + ev.in * SHADOW_BIAS;
						s.ray = make_ray(origin, SHADOW_TMIN, lv.geom.position - origin, 0.9999f);
						s.pixel = in_bounce ? q.pixel : pixel_info_pack(pixel, FB_DIRECT_C, 0);
						s.w = V4(out_w.x, out_w.y, out_w.z, q.w.w);
						s.light_path_id = (light_vertex_id & 0xFFFFFFu) | ((light_depth + 1) << 24) | ((ev.depth + 2) << 28);
						shadow_queue.push_back(s);
					}
				}
			}
		}
		else if (max_light_depth >= 0 && connect)
		{
			const u32 light_path_id = pixel;        // n_light_paths == n_eye_paths
			const i32 n_light_vertices = i32(v_counts[light_path_id]);
			for (u32 light_depth = options.direct_lighting_nee ? 0 : 1; i32(light_depth) < (n_light_vertices < max_light_depth + 1 ? n_light_vertices : max_light_depth + 1); ++light_depth)
			{
				const u32 li = light_path_id + light_depth * n_light_paths;
				BptVertex lv;
				PathWeights lw; lw.pGp_sum = v_weights[2 * li]; lw.pG = v_weights[2 * li + 1];
				lv.setup_stored(&v_pos[4 * li], v_input[2 * li], v_input[2 * li + 1], v_gbuffer[li], lw, light_depth, scene());
				V3 out, out_w; float d;
				eval_connection(ev, lv, out, out_w, d, options.rr != 0, options.direct_lighting_nee != 0, options.direct_lighting_bsdf != 0);
				if (max_comp(out_w) > 0.0f && finite_f(out_w.x) && finite_f(out_w.y) && finite_f(out_w.z))
				{
					Shadow s;
					const V3 origin = ev.geom.position + ev.in * SHADOW_BIAS;
					s.ray = make_ray(origin, SHADOW_TMIN, lv.geom.position - origin, 0.9999f);
					s.pixel = in_bounce ? q.pixel : pixel_info_pack(pixel, FB_DIRECT_C, 0);
					s.w = V4(out_w.x, out_w.y, out_w.z, q.w.w);
					s.light_path_id = light_path_id | ((light_depth + 1) << 24) | ((ev.depth + 2) << 28);
					shadow_queue.push_back(s);
				}
			}
		}
	}

	// solve_shadows for the eye-subpath connections : src/bpt_control.h:352-382 ; solve_occlusion src/bpt_kernels.h:899-916
	void solve_eye_shadows()
	{
		host->trace_queue(shadow_queue, false);                    // RTContext::trace: closest hit, tmin = SHADOW_TMIN
		for (size_t i = 0; i < shadow_queue.size(); ++i)
		{
			const Shadow& s = shadow_queue[i];
			const float vis = (s.hit.t < 0.0f) ? 1.0f : 0.0f;
			sink(pi_comp(s.pixel), s.w * vis, pi_pixel(s.pixel));
		}
		shadow_queue.clear();
	}

	// sample_eye_subpaths : src/bpt_control.h:384-470
	void sample_eye_subpaths()
	{
		shadow_queue.clear(); scatter_queue.clear(); in_queue.clear();
		for (u32 i = 0; i < n_shard(); ++i) generate_primary_eye_vertex(shard_id(i));
		stats.n_bounces_eye = 0;
		for (u32 in_bounce = 0; in_bounce < options.max_path_length; ++in_bounce)
		{
			if (in_queue.empty()) break;
			scatter_queue.clear();
			trace_entries(in_queue);
			for (size_t i = 0; i < in_queue.size(); ++i) process_secondary_eye_vertex(in_queue[i], in_bounce);
			stats.eye_queue[stats.n_bounces_eye] = u32(in_queue.size());
			stats.shadow_eye[stats.n_bounces_eye++] = u32(shadow_queue.size());
			solve_eye_shadows();
			in_queue.swap(scatter_queue);
		}
	}

	// connect_to_camera : src/bpt_kernels.h:919-1032 ; light_tracing src/bpt_control.h:572-600
	void light_tracing_pass()
	{
		stats.shadow_light_tracing = 0;
		if (!light_tracing) return;
		shadow_queue.clear();
		const V3 eye = scene().camera.eye;
		for (u32 ii = 0; ii < n_shard(); ++ii)
			for (u32 k = 0; k < v_counts[shard_id(ii)]; ++k)
			{
				const u32 id = shard_id(ii);
				const u32 li = id + k * n_light_paths;
				const u32 light_depth = v_path_id[li] >> 24;
				const float light_weight = 1.0f / float(n_light_paths);
				if (light_depth == 0) continue;                     // primary light vertices: "visible lights (a very silly strategy)" is compiled out
				VertexGeometry geom;
				geom.position = V3(v_pos[4 * li], v_pos[4 * li + 1], v_pos[4 * li + 2]);
				geom.normal_s = unpack_direction(f2bits(v_pos[4 * li + 3]));
				geom.normal_g = geom.normal_s;
				geom.tangent = orthogonal(geom.normal_s);
				geom.binormal = cross(geom.normal_s, geom.tangent);
				const V3 in_dir = unpack_direction(v_input[2 * li]);
				const V3 in_alpha = from_rgbe(v_input[2 * li + 1]);
				const V3 delta = geom.position - eye;
				const float d2 = fmax_ieee(1.0e-8f, dot(delta, delta));
				const float d = sqrtf(d2);
				const V3 out = delta / d;
				const float cos_theta = dot(out, W) / W_len;
				const float G = fabsf(cos_theta * dot(out, geom.normal_s)) / d2;
				float out_x = 0, out_y = 0;
				const float p_s = camera_direction_pdf_xy(U, V, W, W_len, sq_focal, out, &out_x, &out_y, true);
				const float f_s = p_s * float(scene().res_x * scene().res_y);
				if (!f_s) continue;
				Bsdf light_bsdf;
				unpack_bsdf(v_gbuffer[li], scene().glossy_reflectance, light_bsdf);
				const V3 f_L = light_bsdf.f_sum(geom, in_dir, -out);
				const float p_L = light_bsdf.p_sum(geom, in_dir, -out, true);
				const float pGp = pdf_product3(p_s, G, p_L);
				// the reference prices the neighbouring strategy with max_comp(f_L) where its own scheme wants the reverse pdf (src/bpt_kernels.h:1009):
				// the second origin of the light-tracing energy mismatch (it darkens; what-if bit 1 puts the pdf there)
				const float next_pGp = pdf_product((whatif_consistent_mis & 2u) ? light_bsdf.p_sum(geom, -out, in_dir, true) : max_comp(f_L), v_weights[2 * li + 1]);
				const float mis_w =
					(light_depth == 1 && !options.direct_lighting_nee && !options.direct_lighting_bsdf) ? 1.0f :
					(light_depth > 1 && !options.indirect_lighting_nee && !options.indirect_lighting_bsdf) ? 1.0f :
					bpt_mis(pGp / light_tracing, next_pGp, v_weights[2 * li]);
				const V3 c = in_alpha * f_L * f_s * G * mis_w;
				const V4 out_w = V4(c.x, c.y, c.z, 1.0f) * light_weight;
				if (max_comp(out_w.xyz()) > 0.0f && finite_f(out_w.x) && finite_f(out_w.y) && finite_f(out_w.z))
				{
					Shadow s;
					const V3 origin = geom.position + in_dir * SHADOW_BIAS;
					s.ray = make_ray(origin, SHADOW_TMIN, eye - origin, 0.9999f);
					s.pixel = pixel_info_pack(quantize(out_x * 0.5f + 0.5f, scene().res_x) + quantize(out_y * 0.5f + 0.5f, scene().res_y) * scene().res_x, FB_DIRECT_C, 0);
					s.w = out_w; s.light_path_id = 0;
					shadow_queue.push_back(s);
				}
			}
		stats.shadow_light_tracing = u32(shadow_queue.size());
		host->trace_queue(shadow_queue, false);
		// ConnectionsSink<true>: xyz of COMPOSITED_C and of the entry's channel (DIRECT_C) — order-independent fixed-point sums (DEFINED HERE)
		for (size_t i = 0; i < shadow_queue.size(); ++i)
		{
			const Shadow& s = shadow_queue[i];
			if (!(s.hit.t < 0.0f)) continue;
			const u32 pixel = pi_pixel(s.pixel);
			const float v[3] = { s.w.x * frame_weight, s.w.y * frame_weight, s.w.z * frame_weight };
			for (int c = 0; c < 3; ++c)
			{
				const long long q = (long long)rint(double(v[c]) * 4294967296.0);
				splat[size_t(pixel) * 6 + c] += q;
				splat[size_t(pixel) * 6 + 3 + c] += q;
			}
		}
		shadow_queue.clear();
		if (!deferred_splats) resolve_splats();
	}
	void resolve_splats()
	{
		FrameBuffer& f = fb();
		const u32 n = scene().res_x * scene().res_y;
		for (u32 p = 0; p < n; ++p)
		{
			long long* q = &splat[size_t(p) * 6];
			if (!(q[0] | q[1] | q[2])) continue;
			V4 c = f.get(FB_COMPOSITED_C, p), dch = f.get(FB_DIRECT_C, p);
			c.x += float(double(q[0]) * (1.0 / 4294967296.0)); c.y += float(double(q[1]) * (1.0 / 4294967296.0)); c.z += float(double(q[2]) * (1.0 / 4294967296.0));
			dch.x += float(double(q[3]) * (1.0 / 4294967296.0)); dch.y += float(double(q[4]) * (1.0 / 4294967296.0)); dch.z += float(double(q[5]) * (1.0 / 4294967296.0));
			f.set(FB_COMPOSITED_C, p, c); f.set(FB_DIRECT_C, p, dch);
			for (int k = 0; k < 6; ++k) q[k] = 0;
		}
	}

	// BPT::render : src/renderers/bpt_impl.h:198-258
	void render(u32 instance, const u32* pixels = nullptr, u32 n_pixels = 0)
	{
		shard_pixels = pixels; shard_count = n_pixels;
		host->rescale_frame(instance);              // renderer.multiply_frame(instance / (instance + 1))
		sequence.set_instance(instance);
		frame_weight = 1.0f / float(instance + 1);
		std::memset(&stats, 0, sizeof(stats));
		sample_light_subpaths();
		if (options.single_connection)
		{
			// the flat vertex list of VertexOrdering::kRandomOrdering in the order defined in this file's header
			flat.clear();
			for (u32 d = 0; d < options.max_path_length; ++d)
			{
				for (u32 id = 0; id < n_light_paths; ++id)          // light paths outside this rank's shard keep a count of zero
					if (v_counts[id] > d) flat.push_back(id + d * n_light_paths);
				if (d == 0) n_flat_primary = u32(flat.size());
			}
		}
		sample_eye_subpaths();
		light_tracing_pass();
	}
};

} // namespace orc
