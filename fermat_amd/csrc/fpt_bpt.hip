// fpt_bpt.hip — wavefront kernels of the bidirectional path tracer (`-bpt`, all-connections mode `-sc 0`).
//
//   light_primary_kernel      generate_primary_light_vertex      src/bpt_kernels.h:275-392
//   light_vertices_kernel     process_secondary_light_vertex     src/bpt_kernels.h:394-521
//   eye_primary_kernel        generate_primary_eye_vertex        src/bpt_kernels.h:523-571
//   eye_vertices_kernel       process_secondary_eye_vertex       src/bpt_kernels.h:573-897   (+ eval_connection, eval_incoming_emission, scatter: src/bpt_utils.h:911-1100)
//   eye_resolve_kernel        solve_occlusion + ConnectionsSink<false>   src/bpt_kernels.h:899-916 ; src/renderers/bpt_impl.h:131-163
//   connect_camera_kernel     light_tracing_kernel / connect_to_camera   src/bpt_kernels.h:919-1070
//   splat_kernel, splat_resolve_kernel    ConnectionsSink<true>  src/renderers/bpt_impl.h:141-155
//
// MI355X design:
//  * the light-vertex store is SoA indexed `path + depth * n_paths` (the reference's kPathOrdering), so a bounce of light
//    vertices is one coalesced stream and an eye vertex reads "its" light sub-path with stride n_paths;
//  * an eye vertex can queue up to L connection rays.  Each thread takes a CONTIGUOUS range of the shadow queue (one
//    block-wide scan + one atomic per workgroup); after the any-hit launch the resolve kernel runs one thread per eye vertex and
//    adds its connections in light-depth order — no atomics, and the frame buffer is bit-reproducible (the reference's sink is a
//    racy read-modify-write);
//  * light-tracing splats land on arbitrary pixels: they are accumulated as 2^-32 fixed-point 64-bit integer atomics
//    (order-independent) and added to the frame once per pass.
#include "fpt_bpt.h"

namespace fpt {

namespace {

constexpr int BPT_BLOCK = 256;
constexpr float kShadowBias = 1.0e-4f;      // SHADOW_BIAS / SHADOW_TMIN, src/renderer_view.h:44-45
constexpr float kMinGDenom = 1.0e-8f;       // MIN_G_DENOM, src/bpt_utils.h:49

// ---- small helpers ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float pdf2(float a, float b) { return is_finite(a) && is_finite(b) ? a * b : inf_f(); }
__device__ __forceinline__ float pdf3(float a, float b, float c) { return is_finite(a) && is_finite(b) && is_finite(c) ? a * b * c : inf_f(); }
__device__ __forceinline__ float mis4(float pGp, float prev, float next, float sum) { return pGp && prev && next ? (1 / pGp) / ((1 / pGp) + (1 / prev) + (1 / next) + sum) : 0.0f; }
__device__ __forceinline__ float mis3(float pGp, float other, float sum) { return pGp && other ? (1 / pGp) / ((1 / pGp) + (1 / other) + sum) : 0.0f; }
__device__ __forceinline__ bool finite3(f3 v) { return is_finite(v.x) && is_finite(v.y) && is_finite(v.z); }

// shared-exponent colour (contrib/cugar/color/rgbe.h:35-75)
__device__ __forceinline__ uint32_t to_rgbe(f3 c)
{
	float v = 0;
	if (c.x > v) v = c.x;
	if (c.y > v) v = c.y;
	if (c.z > v) v = c.z;
	uint32_t x = as_u32(v);
	const int exponent = int(((x >> 23u) & 0xFF) - 126u);
	const int e = int((uint32_t(exponent) + 128u) & 0xFF);
	if (e < 10) return 0;
	x = ((((uint32_t(e) & 0xFF) - (128u + 8u)) + 127u) << 23u) & 0x7F800000u;
	const float f = 1.0f / as_f32(x);
	return uint32_t(e) | (to_u32_sat(c.x * f) << 24) | (to_u32_sat(c.y * f) << 16) | (to_u32_sat(c.z * f) << 8);
}
__device__ __forceinline__ f3 from_rgbe(uint32_t p)
{
	const float f = as_f32((((p & 0xFF) - 9u) << 23u) & 0x7F800000u);
	return mk3(f * float(p >> 24), f * float((p >> 16) & 0xFF), f * float((p >> 8) & 0xFF));
}
// 16:16-bit direction (src/vertex.h:123-140 over contrib/cugar/spherical/mappings_inline.h:162-185)
__device__ __forceinline__ void sphere_to_square(f3 v, float& sx, float& sy)
{
	float phi;
	if (fabsf(v.z) >= 1.0f - 1.0e-5f) phi = 0.0f;
	else { phi = det_atan2(v.y, v.x); phi = phi < 0.0f ? phi + 2.0f * kPi : phi; }
	sx = phi / (2.0f * kPi); sy = (v.z + 1.0f) * 0.5f;
}
__device__ __forceinline__ uint32_t pack_direction(f3 d) { float sx, sy; sphere_to_square(d, sx, sy); return quantize(sx, 0xFFFFu) + (quantize(sy, 0xFFFFu) << 16); }
__device__ __forceinline__ f3 unpack_direction(uint32_t p)
{
	const float ux = float(p & 0xFFFFu) / float(0xFFFFu), uy = float((p >> 16) & 0xFFFFu) / float(0xFFFFu);
	const float ct = uy * 2.0f - 1.0f;
	const float st = sqrtf(ieee_max(1.0f - ct * ct, 0.0f));
	float s, c; det_sincos(ux * (2.0f * kPi), s, c);
	return mk3(c * st, s * st, ct);
}
__device__ __forceinline__ float pack_gbuffer_normal(f3 N)          // GBufferView::pack_geometry, src/framebuffer.h:84-90
{
	float sx, sy; sphere_to_square(N, sx, sy);
	const uint32_t M = (1u << 15) - 1u;
	return as_f32(quantize(sx, M) | (quantize(sy, M) << 15));
}
// stored-vertex material (src/bpt_utils.h:203-260)
__device__ __forceinline__ uint4 pack_material(f3 diffuse, f3 specular, f3 diffuse_trans, float roughness, float opacity, float ior)
{
	const uint32_t r = quantize(roughness, 65535u) & 0xFFFFu, o = quantize(opacity, 255u) & 0xFFFFu, i = quantize(ior / 3.0f, 255u) & 0xFFFFu;
	return make_uint4(to_rgbe(diffuse), to_rgbe(specular), r | (o << 16) | (i << 24), to_rgbe(diffuse_trans));
}
__device__ __forceinline__ SurfaceModel unpack_material(uint4 p, const float* table)
{
	const float roughness = float(p.z & 65535u) / 65535.0f;
	const float opacity = float((p.z >> 16) & 255u) / 255.0f;
	const float ior = sel_max(3.0f * (float(p.z >> 24) / 255.0f), 0.00001f);
	return make_surface_model_unpacked(from_rgbe(p.x), from_rgbe(p.y), roughness, from_rgbe(p.w), opacity, ior, table);
}
// camera_direction_pdf, projected-solid-angle form, optionally returning the screen position (src/camera.h:206-252)
__device__ __forceinline__ float camera_pdf(const BptParams& P, f3 out, float* ox, float* oy)
{
	const float t = dot(out, P.W) / (P.W_len * P.W_len);
	if (t < 0.0f) return 0.0f;
	const f3 I = out / t - P.W;
	const float Ix = dot(I, P.U) / dot(P.U, P.U);
	const float Iy = dot(I, P.V) / dot(P.V, P.V);
	if (Ix >= -1.0f && Ix <= 1.0f && Iy >= -1.0f && Iy <= 1.0f)
	{
		if (ox) *ox = Ix;
		if (oy) *oy = Iy;
		const float ct = dot(out, P.W) / P.W_len;
		return P.sq_focal / (ct * ct * ct * ct);
	}
	return 0.0f;
}
// virtual path id -> (pass offset, pixel / light-path index); see BptParams
struct PathRef { uint32_t k, id; };
__device__ __forceinline__ PathRef path_ref(const BptParams& P, uint32_t vid)
{
	PathRef r;
	r.k = P.n_passes == 1 ? 0u : vid / P.n_paths;
	r.id = vid - r.k * P.n_paths;
	return r;
}
__device__ __forceinline__ float frame_weight(const BptParams& P, uint32_t k) { return 1.0f / float(P.instance + k + 1); }
__device__ __forceinline__ float4* fb_cell(const BptParams& P, uint32_t channel, PathRef r) { return P.fb.ch[channel] + size_t(r.k) * P.plane_stride + r.id; }

// primary sample coordinates (src/bpt_samplers.h:43-88) with the per-frame table folded in
__device__ __forceinline__ float light_coord(const BptParams& P, PathRef r, uint32_t vertex, uint32_t dim)
{
	const uint32_t T2 = P.seq.tile_size * P.seq.tile_size, d = vertex * 3 + dim;
	return frac_pos(randfloat(d, P.instance + r.k + 1) + P.seq.shifts[size_t(d) * T2 + (r.id & (T2 - 1u))]);
}
__device__ __forceinline__ float tiled_coord(const BptParams& P, uint32_t k, uint32_t px, uint32_t py, uint32_t dim)
{
	const uint32_t T = P.seq.tile_size;
	const uint32_t shift = (px & (T - 1)) + (py & (T - 1)) * T;
	const uint32_t tile  = ((px / T) & (T - 1)) + ((py / T) & (T - 1)) * T;
	const size_t base = size_t(dim) * T * T;
	const float sample = frac_pos(randfloat(dim, P.instance + k + 1) + P.seq.shifts[base + shift]);
	return frac_pos(sample + P.seq.shifts[base + tile]);
}
__device__ __forceinline__ float eye_coord(const BptParams& P, uint32_t k, uint32_t px, uint32_t py, uint32_t vertex, uint32_t dim)
{
	if (vertex == 1 && dim < 2)
		return dim == 0 ? (float(px) + tiled_coord(P, k, px, py, dim)) / float(P.res_x) : (float(py) + tiled_coord(P, k, px, py, dim)) / float(P.res_y);
	return tiled_coord(P, k, px, py, (vertex - 1) * 6 + dim);
}

// ---- queue-slot allocation -----------------------------------------------------------------------------------------------------
// every thread asks for `n` CONTIGUOUS slots; one atomic per workgroup
struct RangeScratch { uint32_t wave_total[BPT_BLOCK / 64]; uint32_t base; };
// exclusive scan of n over the block's threads (in thread order) + the block's total; every thread of the block must call it
__device__ __forceinline__ uint32_t block_range_scan(uint32_t n, RangeScratch& sc, uint32_t& block_total)
{
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	uint32_t incl = n;
	#pragma unroll
	for (int d = 1; d < 64; d <<= 1) { const uint32_t v = __shfl_up(incl, d); if (int(lane) >= d) incl += v; }
	__syncthreads();
	if (lane == 63) sc.wave_total[wave] = incl;
	__syncthreads();
	uint32_t before = 0, total = 0;
	#pragma unroll
	for (int w = 0; w < BPT_BLOCK / 64; ++w) { const uint32_t c = sc.wave_total[w]; if (w < int(wave)) before += c; total += c; }
	block_total = total;
	return before + incl - n;
}
__device__ __forceinline__ uint32_t block_range_alloc(uint32_t* counter, uint32_t n, RangeScratch& sc)
{
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	uint32_t incl = n;
	#pragma unroll
	for (int d = 1; d < 64; d <<= 1) { const uint32_t v = __shfl_up(incl, d); if (int(lane) >= d) incl += v; }
	if (lane == 63) sc.wave_total[wave] = incl;
	__syncthreads();
	if (threadIdx.x == 0)
	{
		uint32_t total = 0;
		for (int w = 0; w < BPT_BLOCK / 64; ++w) { const uint32_t c = sc.wave_total[w]; sc.wave_total[w] = total; total += c; }
		sc.base = total ? atomicAdd(counter, total) : 0u;
	}
	__syncthreads();
	const uint32_t r = sc.base + sc.wave_total[wave] + (incl - n);
	__syncthreads();
	return r;
}

__device__ __forceinline__ void write_ray(float4* rays, uint32_t slot, f3 o, float tmin, f3 d, float tmax)
{
	rays[2 * size_t(slot)]     = make_float4(o.x, o.y, o.z, tmin);
	rays[2 * size_t(slot) + 1] = make_float4(d.x, d.y, d.z, tmax);
}

// ---- a shaded path vertex: EyeVertex / LightVertex::setup(ray, hit, ...) (src/bpt_utils.h:340-361, 585-642) -------------------------
struct Vertex
{
	SurfacePoint sp;
	f3 in, alpha;
	SurfaceModel bsdf;
	f3 diffuse, specular, diffuse_trans;      // texture-modulated, for pack_material
	float roughness, opacity, ior;
	float prev_pG, pGp_sum;
};
__device__ __forceinline__ void shade_vertex(const BptParams& P, f3 ro, f3 rd, float t, uint32_t tri, float u, float v, f3 alpha, float4 pw, bool light, Vertex& x)
{
	uint32_t material_index;
	if (P.shade_records)
	{
		const ShadeRecord rec = P.shade_records[tri];          // the vertex's fifteen words in one 64-byte fetch (fpt_shading.h)
		surface_point(rec, P.mesh, u, v, x.sp);
		material_index = as_u32(rec.d.w);
	}
	else
	{
		surface_point(P.mesh, tri, u, v, x.sp);
		material_index = uint32_t(P.mesh.material_indices[tri]);
	}
	x.sp.position = ro + t * rd;
	const fpt_material* mat = P.mesh.materials + material_index;
	const f4 one4 = mk4(1, 1, 1, 1);
	const f4 m_diffuse  = load4(mat->diffuse)       * sample_texture(P.textures, mat->diffuse_map, x.sp.s, x.sp.t, one4);
	const f4 m_specular = load4(mat->specular)      * sample_texture(P.textures, mat->specular_map, x.sp.s, x.sp.t, one4);
	const f4 m_dtrans   = load4(mat->diffuse_trans) * sample_texture(P.textures, mat->diffuse_trans_map, x.sp.s, x.sp.t, one4);
	x.in = -normalize(rd);
	x.alpha = alpha;
	x.diffuse = xyz(m_diffuse); x.specular = xyz(m_specular); x.diffuse_trans = xyz(m_dtrans);
	x.roughness = mat->roughness; x.opacity = mat->opacity; x.ior = mat->index_of_refraction;
	x.bsdf = make_surface_model(x.diffuse, x.diffuse_trans, x.specular, xyz(load4(mat->reflectivity)), mat->roughness, mat->index_of_refraction, mat->opacity, P.table);
	// MIS bookkeeping for the next vertex: pw = (pGp_sum, pG, out_p, out_cos_theta) of the edge that led here
	const float G_prime = light ? fabsf(dot(x.in, x.sp.frame.n)) / ieee_max(t * t, kMinGDenom) : fabsf(dot(x.in, x.sp.frame.n)) / (t * t);
	x.prev_pG = pdf2(pw.z, pw.w * G_prime);
	x.pGp_sum = pw.x + (1 / pdf2(pw.y, pw.z));
}

// a stored light vertex: LightVertex::setup(pos, packed...) (src/bpt_utils.h:313-337)
struct StoredVertex { ShadingFrame fr; f3 position, in, alpha; SurfaceModel bsdf; f3 edf; float pGp_sum, pG; uint32_t depth; };
__device__ __forceinline__ void store_vertex(const BptParams& P, uint32_t slot, float4 pos, uint4 gb, uint2 inp, float2 w, uint32_t path_id)
{
	float4* rec = reinterpret_cast<float4*>(P.store.rec + slot);
	rec[0] = pos;
	rec[1] = make_float4(as_f32(gb.x), as_f32(gb.y), as_f32(gb.z), as_f32(gb.w));
	rec[2] = make_float4(as_f32(inp.x), as_f32(inp.y), w.x, w.y);
	rec[3] = make_float4(as_f32(path_id), 0.0f, 0.0f, 0.0f);
	P.store.pos[slot] = pos;
}
__device__ __forceinline__ void load_stored(const BptParams& P, uint32_t slot, uint32_t depth, StoredVertex& s)
{
	const float4* rec = reinterpret_cast<const float4*>(P.store.rec + slot);          // four 16-byte loads from one 64-byte line
	const float4 pos = rec[0], r1 = rec[1], r2 = rec[2];
	const uint4 gb = make_uint4(as_u32(r1.x), as_u32(r1.y), as_u32(r1.z), as_u32(r1.w));
	const uint2 inp = make_uint2(as_u32(r2.x), as_u32(r2.y));
	const float2 w = make_float2(r2.z, r2.w);
	s.in = unpack_direction(inp.x);
	s.alpha = from_rgbe(inp.y);
	s.pGp_sum = w.x; s.pG = w.y; s.depth = depth;
	s.position = mk3(pos.x, pos.y, pos.z);
	s.fr.n = unpack_direction(as_u32(pos.w));
	s.fr.ng = s.fr.n;
	s.fr.t = orthogonal(s.fr.n);
	s.fr.b = cross(s.fr.n, s.fr.t);
	s.edf = splat3(0.0f);
	if (depth == 0) s.edf = from_rgbe(gb.x);
	else s.bsdf = unpack_material(gb, P.table);
}

// eval_connection (src/bpt_utils.h:911-980); returns the connection weight (0 when nothing is to be traced)
__device__ __forceinline__ f3 connect(const BptParams& P, const Vertex& ev, uint32_t ev_depth, const StoredVertex& lv)
{
	const bool RR = P.opt.rr != 0;
	const f3 delta = lv.position - ev.sp.position;
	const float d2 = ieee_max(kMinGDenom, dot(delta, delta));
	const float d = sqrtf(d2);
	const f3 out = delta / d;
	const float G = fabsf(dot(out, ev.sp.frame.n) * dot(out, lv.fr.n)) / d2;
	f3 f_s; float p_s;
	surface_f_and_p_sum(ev.bsdf, ev.sp.frame, ev.in, out, RR, false, f_s, p_s);
	const float prev_pGp = pdf2(ev.prev_pG, p_s);
	if (lv.depth == 0)
	{
		if (!P.opt.direct_lighting_nee) return splat3(0.0f);
		const f3 f_L = dot(lv.fr.n, -out) > 0.0f ? lv.edf : splat3(0.0f);
		const float p_L = 1.0f / kPi;
		const float pGp = pdf3(p_s, G, p_L);
		const float next_pGp = pdf2(p_L, lv.pG);
		const float mis_w = (ev_depth == 0 && !P.opt.direct_lighting_bsdf) ? 1.0f : mis4(pGp, prev_pGp, next_pGp, ev.pGp_sum + lv.pGp_sum);
		return ev.alpha * lv.alpha * f_L * f_s * G * mis_w;
	}
	f3 f_L; float p_L;
	surface_f_and_p_sum(lv.bsdf, lv.fr, lv.in, -out, RR, true, f_L, p_L);
	const float pGp = pdf3(p_s, G, p_L);
	const float next_pGp = pdf2(p_L, lv.pG);
	const float mis_w = mis4(pGp, prev_pGp, next_pGp, ev.pGp_sum + lv.pGp_sum);
	return ev.alpha * lv.alpha * f_L * f_s * G * mis_w;
}

// ConnectionsSink<false>::sink (src/renderers/bpt_impl.h:131-163): all four components, COMPOSITED_C and the path's channel.  One pass per render():
// straight into the frame; passes in flight: into the eye path's cell `cell` of the batch's log (BptLog), applied in order by merge_exact_kernel
__device__ __forceinline__ void sink(const BptParams& P, uint32_t channel, f3 v, float w, uint32_t vid, uint32_t cell)
{
	if (P.n_passes > 1)
	{
		P.log.val[size_t(cell) * P.log.cap + vid] = make_float4(v.x, v.y, v.z, w);
		P.log.chan[size_t(cell) * P.log.cap + vid] = channel;
		uint32_t* m = P.log.mask + size_t(vid) * P.log.mask_words + (cell >> 5);      // the word belongs to this path alone; one writer per launch
		*m |= 1u << (cell & 31u);
		return;
	}
	const PathRef r = path_ref(P, vid);
	const float fw = frame_weight(P, r.k);
	float4* c = fb_cell(P, FPT_FB_COMPOSITED_C, r);
	float4 a = *c;
	a.x += v.x * fw; a.y += v.y * fw; a.z += v.z * fw; a.w += w * fw;
	*c = a;
	if (channel != FPT_FB_COMPOSITED_C)
	{
		float4* k = fb_cell(P, channel, r);
		float4 b = *k;
		b.x += v.x * fw; b.y += v.y * fw; b.z += v.z * fw; b.w += w * fw;
		*k = b;
	}
}

// ---- kernels -------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(BPT_BLOCK) light_primary_kernel(const BptParams P)
{
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	if (i >= P.n_local * P.n_passes) return;
	PathRef r; r.k = i / P.n_local;
	const uint32_t li = i - r.k * P.n_local;
	r.id = P.pixels ? P.pixels[li] : li;
	const uint32_t id = r.id, vid = r.k * P.n_paths + r.id;
	P.store.counts[vid] = 0; P.store.rec[vid].path_id = 0xFFFFFFFFu;
	const uint32_t L = P.opt.max_path_length;
	SurfacePoint lp; f3 radiance; float pdf;
	if (P.opt.use_vpls)
	{
		const fpt_vpl vp = P.emitters.vpls[id];             // BPT::init copies the VPL set into the vertex store (src/renderers/bpt.cu:101-102)
		surface_point(P.mesh, vp.prim_id, vp.uv[0], vp.uv[1], lp);
		emitter_at(P.emitters, P.mesh, P.textures, vp.prim_id, lp.s, lp.t, radiance, pdf);
	}
	else
		emitter_sample(P.emitters, P.mesh, P.textures, light_coord(P, r, 0, 0), light_coord(P, r, 0, 1), light_coord(P, r, 0, 2), lp, radiance, pdf);
	store_vertex(P, vid, make_float4(lp.position.x, lp.position.y, lp.position.z, as_f32(pack_direction(lp.frame.n))), make_uint4(to_rgbe(radiance), 0, 0, 0),
	             make_uint2(0u, to_rgbe(splat3(1.0f) / pdf)), make_float2(0.0f, 1.0f * pdf), id);
	P.store.counts[vid] = 1;
	if (1 >= L + 1) return;
	// Edf::sample: cosine-distributed emission (contrib/cugar/bsdf/lambert_edf.h:82-99)
	const f3 l = cosine_hemisphere(light_coord(P, r, 1, 0), light_coord(P, r, 1, 1));
	const f3 out = l.x * lp.frame.t + l.y * lp.frame.b + l.z * lp.frame.n;
	const f3 g = (radiance * kPi) / pdf;
	const float p_proj = 1.0f / kPi;
	write_ray(P.out.rays, i, lp.position, 1.0e-4f, out, 1.0e8f);
	P.out.weights[i] = make_float4(g.x, g.y, g.z, 0.0f);
	P.out.pixels[i] = vid;                                  // PixelInfo(light path, DIFFUSE_C = 0)
	P.out.path_weights[i] = make_float4(0.0f, 1.0f * pdf, p_proj, fabsf(dot(lp.frame.n, out)));
	if (i == 0) *P.out.size = P.n_local * P.n_passes;
}

__global__ void __launch_bounds__(BPT_BLOCK) light_vertices_kernel(const BptParams P)
{
	__shared__ RangeScratch sc;
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	const uint32_t n = *P.in.size;
	bool active = i < n, want = false;
	f3 o = splat3(0.0f), dir = splat3(0.0f); float4 w_out = make_float4(0, 0, 0, 0), pw_out = make_float4(0, 0, 0, 0); uint32_t pixel_info = 0;
	if (active)
	{
		const float4 hit4 = P.in.hits[i];
		const int32_t tri = int32_t(as_u32(hit4.y));
		if (hit4.x > 0.0f && tri >= 0)
		{
			const float4 ro = P.in.rays[2 * size_t(i)], rd4 = P.in.rays[2 * size_t(i) + 1];
			const float4 w4 = P.in.weights[i];
			pixel_info = P.in.pixels[i];                    // a light path's queue word is its virtual id (PixelInfo(light path, DIFFUSE_C = 0))
			const uint32_t vid = pixel_info;
			const PathRef r = path_ref(P, vid);
			Vertex lv;
			shade_vertex(P, mk3(ro.x, ro.y, ro.z), mk3(rd4.x, rd4.y, rd4.z), hit4.x, uint32_t(tri), hit4.z, hit4.w, mk3(w4.x, w4.y, w4.z), P.in.path_weights[i], true, lv);
			if (P.bounce + 2 < P.opt.max_path_length + 1)
			{
				f3 out, g; float p, p_proj;
				const uint32_t comp = surface_sample_ex(lv.bsdf, lv.sp.frame, light_coord(P, r, P.bounce + 2, 0), light_coord(P, r, P.bounce + 2, 1), light_coord(P, r, P.bounce + 2, 2),
				                                        lv.in, P.opt.rr != 0, true, true, out, p, p_proj, g);
				(void)comp;
				const f3 out_w = g * lv.alpha;
				if (max_comp(out_w) > 0.0f)
				{
					want = true; o = lv.sp.position; dir = out;
					w_out = make_float4(out_w.x, out_w.y, out_w.z, w4.w);
					pw_out = make_float4(lv.pGp_sum, lv.prev_pG, p_proj, fabsf(dot(lv.sp.frame.n, out)));
				}
			}
			const uint32_t slot = vid + P.store.counts[vid] * P.n_store;
			store_vertex(P, slot, make_float4(lv.sp.position.x, lv.sp.position.y, lv.sp.position.z, as_f32(pack_direction(lv.sp.frame.n))),
			             pack_material(lv.diffuse, lv.specular, lv.diffuse_trans, lv.roughness, lv.opacity, lv.ior),
			             make_uint2(pack_direction(lv.in), to_rgbe(mk3(w4.x, w4.y, w4.z))), make_float2(lv.pGp_sum, lv.prev_pG), r.id | ((P.bounce + 1) << 24));
			P.store.counts[vid] += 1;
		}
	}
	const uint32_t slot = block_range_alloc(P.out.size, want ? 1u : 0u, sc);
	if (want)
	{
		write_ray(P.out.rays, slot, o, 1.0e-4f, dir, 1.0e8f);
		P.out.weights[slot] = w_out; P.out.pixels[slot] = pixel_info; P.out.path_weights[slot] = pw_out;
	}
}

__global__ void __launch_bounds__(BPT_BLOCK) eye_primary_kernel(const BptParams P)
{
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	if (i >= P.n_local * P.n_passes) return;
	const uint32_t k = i / P.n_local, li = i - k * P.n_local;
	const uint32_t idx = P.pixels ? P.pixels[li] : li;
	const uint32_t px = idx % P.res_x, py = idx / P.res_x;
	const float ux = eye_coord(P, k, px, py, 1, 0), uy = eye_coord(P, k, px, py, 1, 1);
	const float dx = ux * 2.f - 1.f, dy = uy * 2.f - 1.f;
	const f3 dir = dx * P.U + dy * P.V + P.W;
	write_ray(P.out.rays, i, P.eye, 0.0f, dir, 1e34f);
	P.out.weights[i] = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
	P.out.pixels[i] = k * P.n_paths + idx; P.out.chan[i] = 0;
	const float p_e = camera_pdf(P, normalize(dir), nullptr, nullptr);
	const float cos_theta = dot(normalize(dir), P.W) / P.W_len;
	P.out.path_weights[i] = make_float4(0.0f, 1.0e8f, P.light_tracing ? p_e / P.light_tracing : 1.0f, P.light_tracing ? cos_theta : 1.0e8f);
	if (i == 0) *P.out.size = P.n_local * P.n_passes;
}

#ifndef FPT_BPT_EYE_WAVES
#define FPT_BPT_EYE_WAVES 3      // 168 VGPRs instead of 176: 3 waves per SIMD, measured +5 % on the BPT pass (4 waves spill too much: -11 %)
#endif
__global__ void __launch_bounds__(BPT_BLOCK, FPT_BPT_EYE_WAVES) eye_vertices_kernel(const BptParams P)
{
	__shared__ RangeScratch sc, sc2;
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	const uint32_t n = *P.in.size;
	const uint32_t L = P.opt.max_path_length;
	bool active = false, want = false;
	Vertex ev;
	uint32_t chan = 0, pixel = 0, vid = 0, n_conn = 0, first_depth = 0;
	PathRef pr; pr.k = 0; pr.id = 0;
	float w_alpha = 0.0f;
	f3 o = splat3(0.0f), dir = splat3(0.0f); float4 w_out = make_float4(0, 0, 0, 0), pw_out = make_float4(0, 0, 0, 0); uint32_t out_chan = 0;
	if (i < n)
	{
		const float4 hit4 = P.in.hits[i];
		const int32_t tri = int32_t(as_u32(hit4.y));
		if (hit4.x > 0.0f && tri >= 0)
		{
			active = true;
			const float4 ro = P.in.rays[2 * size_t(i)], rd4 = P.in.rays[2 * size_t(i) + 1];
			const float4 w4 = P.in.weights[i];
			vid = P.in.pixels[i]; chan = P.in.chan[i];
			pr = path_ref(P, vid);
			pixel = pr.id;
			w_alpha = w4.w;
			const uint32_t px = pixel % P.res_x, py = pixel / P.res_x;
			shade_vertex(P, mk3(ro.x, ro.y, ro.z), mk3(rd4.x, rd4.y, rd4.z), hit4.x, uint32_t(tri), hit4.z, hit4.w, mk3(w4.x, w4.y, w4.z), P.in.path_weights[i], false, ev);
			// BPTConfig::visit_eye_vertex (src/renderers/bpt_impl.h:96-113)
			if (P.bounce == 0 && P.fb.gb_geo && pr.k + 1 == P.n_passes)          // the frame's gbuffer = the last pass of the batch
			{
				P.fb.gb_geo[pixel] = make_float4(ev.sp.position.x, ev.sp.position.y, ev.sp.position.z, pack_gbuffer_normal(ev.sp.frame.n));
				P.fb.gb_uv[pixel] = make_float4(hit4.z, hit4.w, ev.sp.s, ev.sp.t);
				P.fb.gb_tri[pixel] = uint32_t(tri);
			}
			if (P.bounce + 2 < L + 1)
			{
				f3 out, g; float p, p_proj;
				const uint32_t comp = surface_sample_ex(ev.bsdf, ev.sp.frame, eye_coord(P, pr.k, px, py, P.bounce + 2, 0), eye_coord(P, pr.k, px, py, P.bounce + 2, 1), eye_coord(P, pr.k, px, py, P.bounce + 2, 2),
				                                        ev.in, P.opt.rr != 0, true, false, out, p, p_proj, g);
				const f3 out_w = g * ev.alpha;
				if (max_comp(out_w) > 0.0f)
				{
					if (P.bounce + 2 == 2)          // sink_eye_scattering_event: albedo of the visible surface (src/renderers/bpt_impl.h:167-186)
					{
						const int ch = comp == COMP_DIFF_R ? FPT_FB_DIFFUSE_A : (comp == COMP_GLOSSY_R ? FPT_FB_SPECULAR_A : -1);
						if (ch >= 0)
						{
							float4* cell = fb_cell(P, uint32_t(ch), pr);
							const float fw = frame_weight(P, pr.k);
							float4 a = *cell;
							a.x += out_w.x * fw; a.y += out_w.y * fw; a.z += out_w.z * fw; a.w += w_alpha * fw;
							*cell = a;
						}
					}
					want = true; o = ev.sp.position; dir = out;
					out_chan = P.bounce ? chan : uint32_t((comp & COMP_DIFFUSE_MASK) ? FPT_FB_DIFFUSE_C : FPT_FB_SPECULAR_C);
					w_out = make_float4(out_w.x, out_w.y, out_w.z, w_alpha);
					pw_out = make_float4(ev.pGp_sum, ev.prev_pG, p_proj, fabsf(dot(ev.sp.frame.n, out)));
					(void)p;
				}
			}
			// emission seen along the incoming direction (eval_incoming_emission, src/bpt_utils.h:1031-1066)
			const uint32_t t = P.bounce + 2;
			if ((t == 2 && P.opt.visible_lights) || (t == 3 && P.opt.direct_lighting_bsdf) || (t > 3 && P.opt.indirect_lighting_bsdf))
			{
				f3 radiance; float light_pdf;
				emitter_at(P.emitters, P.mesh, P.textures, uint32_t(tri), ev.sp.s, ev.sp.t, radiance, light_pdf);
				const f3 f_L = dot(ev.sp.frame.n, ev.in) > 0.0f ? radiance : splat3(0.0f);
				const float p_L = 1.0f / kPi;
				const float pGp = pdf2(p_L, light_pdf);
				const float prev_pGp = pdf2(ev.prev_pG, p_L);
				const float mis_w = (P.bounce == 0 || pGp == 0.0f || (P.bounce == 1 && !P.opt.direct_lighting_nee) || (P.bounce > 1 && !P.opt.indirect_lighting_nee)) ? 1.0f : mis3(pGp, prev_pGp, ev.pGp_sum);
				const f3 e = ev.alpha * f_L * mis_w;
				if (max_comp(e) > 0.0f && finite3(e)) sink(P, chan & 0xFu, e, w_alpha, vid, P.bounce * (1u + P.log.conn_cells));
			}
			// how many light vertices this eye vertex may connect to
			const int32_t max_light_depth = int32_t(L + 1) - int32_t(P.bounce) - 2 - 1;
			const bool do_connect = (t == 1 && P.opt.direct_lighting_nee) || (t > 1 && P.opt.indirect_lighting_nee);
			if (max_light_depth >= 0 && do_connect && P.opt.single_connection)
				n_conn = (P.flat_meta[2 * (pr.k + 1)] > P.flat_meta[2 * pr.k]) ? 1u : 0u;      // one connection into the pass's flat vertex list
			else if (max_light_depth >= 0 && do_connect)
			{
				const int32_t nlv = int32_t(P.store.counts[vid]);
				const int32_t end = nlv < max_light_depth + 1 ? nlv : max_light_depth + 1;
				first_depth = P.opt.direct_lighting_nee ? 0u : 1u;
				n_conn = end > int32_t(first_depth) ? uint32_t(end) - first_depth : 0u;
			}
		}
	}
	const uint32_t slot = block_range_alloc(P.out.size, want ? 1u : 0u, sc);
	if (want)
	{
		write_ray(P.out.rays, slot, o, 1.0e-4f, dir, 1.0e8f);
		P.out.weights[slot] = w_out; P.out.pixels[slot] = vid; P.out.chan[slot] = uint8_t(out_chan); P.out.path_weights[slot] = pw_out;
	}
	// connections: a contiguous range of the shadow queue per eye vertex, filled in light-depth order (unused tail = null rays)
	const uint32_t base = block_range_alloc(P.shadow.size, n_conn, sc2);
	if (i < n) P.conn[i] = make_uint2(base, 0u);
	if (active && n_conn)
	{
		uint32_t k = 0;
		const f3 origin = ev.sp.position + ev.in * kShadowBias;
		const uint32_t sh_chan = P.bounce ? chan : uint32_t(FPT_FB_DIRECT_C);
		for (uint32_t d = 0; d < n_conn; ++d)
		{
			uint32_t light_depth = first_depth + d, light_slot = vid + light_depth * P.n_store;
			float light_weight = 1.0f;
			if (P.opt.single_connection)
			{
				// a light vertex drawn uniformly from ALL stored vertices of this pass, weighted #vertices / #light paths (src/bpt_kernels.h:714-760)
				const uint32_t first = P.flat_meta[2 * pr.k], n_vertices = P.flat_meta[2 * (pr.k + 1)] - first, n_primary = P.flat_meta[2 * pr.k + 1] - first;
				const uint32_t px = pixel % P.res_x, py = pixel / P.res_x;
				light_slot = P.flat[first + quantize(eye_coord(P, pr.k, px, py, P.bounce + 2, 5), n_vertices)];
				light_depth = P.store.rec[light_slot].path_id >> 24;
				light_weight = float(n_vertices) / float(n_primary);
			}
			StoredVertex lv;
			load_stored(P, light_slot, light_depth, lv);
			const f3 w = connect(P, ev, P.bounce, lv) * light_weight;
			if (max_comp(w) > 0.0f && finite3(w))
			{
				write_ray(P.shadow.rays, base + k, origin, 0.0f, lv.position - origin, 0.9999f);
				P.shadow.weights[base + k] = make_float4(w.x, w.y, w.z, w_alpha);
				P.shadow.pixels[base + k] = vid; P.shadow.chan[base + k] = uint8_t(sh_chan);
				++k;
			}
		}
		P.conn[i] = make_uint2(base, k);
		for (uint32_t d = k; d < n_conn; ++d)
		{
			write_ray(P.shadow.rays, base + d, splat3(0.0f), 0.0f, splat3(0.0f), -1.0f);
			P.shadow.weights[base + d] = make_float4(0, 0, 0, 0);
			P.shadow.pixels[base + d] = 0; P.shadow.chan[base + d] = 0;
		}
	}
}

// one thread per eye vertex: its visible connections are added in light-depth order
__global__ void __launch_bounds__(BPT_BLOCK) eye_resolve_kernel(const BptParams P)
{
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	if (i >= *P.in.size) return;
	const uint2 c = P.conn[i];
	for (uint32_t k = 0; k < c.y; ++k)
	{
		const uint32_t s = c.x + k;
		const float4 w = P.shadow.weights[s];
		const uint32_t pi = P.shadow.pixels[s], ch = P.shadow.chan[s];
		const float vis = (P.shadow.hits[s].x < 0.0f) ? 1.0f : 0.0f;
		// an occluded connection adds zeros: no cell.  (One pass per render() adds w * 0 instead, which differs only for a non-finite weight -- NaN there, nothing
		// here: the bit-identity of batched and sequential passes is for finite samples, as include/fermat_pt_hip.h says)
		if (P.n_passes > 1 && vis == 0.0f) continue;
		sink(P, ch & 0xFu, mk3(w.x * vis, w.y * vis, w.z * vis), w.w * vis, pi, P.bounce * (1u + P.log.conn_cells) + 1u + k);
	}
}

// ---- -sc 1: the flat light-vertex list ---------------------------------------------------------------------------------------------
// Element e = (k * L + d) * n_paths + id  <->  "light path id of pass k stored a vertex with store index d"; the list holds the slots of
// the set elements in element order (pass-major, depth-major, light-path id minor: the order DEFINED in include/fermat_pt_hip.h).
// Three launches: per-block counts, one block scanning the counts, per-block fill.
static constexpr uint32_t FLAT_ITEMS = 16;          // elements per thread
__device__ __forceinline__ bool flat_flag(const BptParams& P, uint64_t e, uint32_t& slot)
{
	const uint32_t L = P.opt.max_path_length;
	const uint64_t total = uint64_t(P.n_passes) * L * P.n_paths;
	if (e >= total) return false;
	const uint32_t id = uint32_t(e % P.n_paths), d = uint32_t((e / P.n_paths) % L), k = uint32_t(e / (uint64_t(P.n_paths) * L));
	const uint32_t vid = k * P.n_paths + id;
	slot = vid + d * P.n_store;
	return P.store.counts[vid] > d;
}
__global__ void __launch_bounds__(BPT_BLOCK) flat_count_kernel(const BptParams P)
{
	__shared__ RangeScratch sc;
	const uint64_t e0 = (uint64_t(blockIdx.x) * BPT_BLOCK + threadIdx.x) * FLAT_ITEMS;
	uint32_t n = 0, slot;
	for (uint32_t j = 0; j < FLAT_ITEMS; ++j) n += flat_flag(P, e0 + j, slot) ? 1u : 0u;
	uint32_t block_total = 0;
	(void)block_range_scan(n, sc, block_total);
	if (threadIdx.x == 0) P.flat_block_sums[blockIdx.x] = block_total;
}
__global__ void __launch_bounds__(BPT_BLOCK) flat_scan_kernel(uint32_t* sums, uint32_t n_blocks, uint32_t* total_out)
{
	__shared__ RangeScratch sc;
	uint32_t carry = 0;
	for (uint32_t base = 0; base < n_blocks; base += BPT_BLOCK)
	{
		const uint32_t i = base + threadIdx.x;
		const uint32_t v = i < n_blocks ? sums[i] : 0u;
		uint32_t chunk_total = 0;
		const uint32_t excl = block_range_scan(v, sc, chunk_total);
		if (i < n_blocks) sums[i] = carry + excl;
		carry += chunk_total;
		__syncthreads();
	}
	if (threadIdx.x == 0) *total_out = carry;
}
__global__ void __launch_bounds__(BPT_BLOCK) flat_fill_kernel(const BptParams P)
{
	__shared__ RangeScratch sc;
	const uint64_t e0 = (uint64_t(blockIdx.x) * BPT_BLOCK + threadIdx.x) * FLAT_ITEMS;
	uint32_t n = 0, slot;
	for (uint32_t j = 0; j < FLAT_ITEMS; ++j) n += flat_flag(P, e0 + j, slot) ? 1u : 0u;
	uint32_t block_total = 0;
	uint32_t pos = P.flat_block_sums[blockIdx.x] + block_range_scan(n, sc, block_total);
	const uint32_t L = P.opt.max_path_length;
	for (uint32_t j = 0; j < FLAT_ITEMS; ++j)
	{
		const uint64_t e = e0 + j;
		// the list position at which a pass begins (d == 0) and at which its depth-1 vertices begin (d == 1; with L == 1 the pass's end)
		if (e % P.n_paths == 0 && e < uint64_t(P.n_passes) * L * P.n_paths)
		{
			const uint32_t d = uint32_t((e / P.n_paths) % L), k = uint32_t(e / (uint64_t(P.n_paths) * L));
			if (d == 0) { P.flat_meta[2 * k] = pos; if (L == 1 && k > 0) P.flat_meta[2 * k - 1] = pos; }
			if (d == 1) P.flat_meta[2 * k + 1] = pos;
		}
		if (flat_flag(P, e, slot)) P.flat[pos++] = slot;
	}
}

// ---- shared light vertices (-sc 1 under tile sharding, the same image for any number of ranks) ------------------------------------------
// pack: one thread per own light path; a block reserves the range of its paths' vertices with one atomic
__global__ void __launch_bounds__(BPT_BLOCK) pack_light_vertices_kernel(const LightVertexRecord* __restrict__ rec, const uint32_t* __restrict__ counts, const uint32_t* __restrict__ pixels,
                                                                        uint32_t n_local, uint32_t n_paths, uint32_t n_passes, LightVertexWire* __restrict__ out, uint32_t* out_count)
{
	__shared__ RangeScratch sc;
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	uint32_t vid = 0, cnt = 0;
	if (i < n_local * n_passes)
	{
		const uint32_t k = i / n_local, lj = i - k * n_local;
		vid = k * n_paths + (pixels ? pixels[lj] : lj);
		cnt = counts[vid];
	}
	const uint32_t base = block_range_alloc(out_count, cnt, sc);
	const uint32_t n_store = n_paths * n_passes;
	for (uint32_t d = 0; d < cnt; ++d)
	{
		LightVertexWire w; w.slot = vid + d * n_store; w.pad[0] = w.pad[1] = w.pad[2] = 0u; w.rec = rec[w.slot];
		out[base + d] = w;
	}
}
__global__ void __launch_bounds__(BPT_BLOCK) unpack_light_vertices_kernel(const LightVertexWire* __restrict__ in, uint32_t n, LightVertexRecord* __restrict__ rec, float4* __restrict__ pos,
                                                                          uint32_t* counts, uint32_t n_store)
{
	const uint32_t j = threadIdx.x + blockIdx.x * blockDim.x;
	if (j >= n) return;
	const LightVertexWire w = in[j];
	rec[w.slot] = w.rec; pos[w.slot] = w.rec.pos;
	atomicMax(counts + w.slot % n_store, w.slot / n_store + 1u);
}

// pure light tracing: every stored vertex of depth >= 1 is connected to the lens (connect_to_camera).  A workgroup takes 256 light paths in
// two stages: (A) one thread per path walks its vertices and lists in LDS those inside the view frustum (a position load and the camera pdf:
// cheap, divergent); (B) the threads share the list, one vertex each per round, for the expensive part (unpack the vertex, both BSDF
// evaluations, the MIS weight) and write the samples to the shadow queue compacted, one queue atomic per round.  The samples' sums are
// order-independent 2^-32 fixed-point integers, so the queue order is free.  (A thread per path doing everything ran at 23 % lane utilisation
// (PMC): path lengths differ and most vertices lie outside the frustum; a thread per vertex needs 8x the workgroups, and their queue atomics
// -- ~90 per microsecond on one counter -- then cost more than the divergence did.)
__global__ void __launch_bounds__(BPT_BLOCK) connect_camera_kernel(const BptParams P)
{
	__shared__ RangeScratch sc;
	__shared__ uint32_t list_n;
	__shared__ uint2 list[BPT_BLOCK * 14];         // (store slot, pass offset << 8 | depth): max_path_length <= 15 (fpt_bpt_init), so a path stores depths 0..14
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	if (threadIdx.x == 0) list_n = 0;
	__syncthreads();
	if (i < P.n_local * P.n_passes)
	{
		const uint32_t pass_k = i / P.n_local, lj = i - pass_k * P.n_local;
		const uint32_t vid = pass_k * P.n_paths + (P.pixels ? P.pixels[lj] : lj);
		const uint32_t cnt = P.store.counts[vid];
		for (uint32_t depth = 1; depth < cnt && depth <= 14; ++depth)          // depth 0 never splats ("visible lights (a very silly strategy)" is compiled out)
		{
			const uint32_t li = vid + depth * P.n_store;
			const float4 pos4 = P.store.pos[li];
			const f3 delta = mk3(pos4.x, pos4.y, pos4.z) - P.eye;
			const float d2 = ieee_max(1.0e-8f, dot(delta, delta));
			const float d = sqrtf(d2);
			const f3 out = delta / d;
			float ox = 0.0f, oy = 0.0f;
			const float p_s = camera_pdf(P, out, &ox, &oy);
			const float f_s = p_s * float(P.res_x * P.res_y);
			if (f_s) list[atomicAdd(&list_n, 1u)] = make_uint2(li, (pass_k << 8) | depth);
		}
	}
	__syncthreads();
	const uint32_t n_list = list_n;
	for (uint32_t e0 = 0; e0 < n_list; e0 += BPT_BLOCK)          // block-uniform trip count: block_range_alloc has barriers inside
	{
		const uint32_t e = e0 + threadIdx.x;
		bool want = false;
		f3 origin = splat3(0.0f), w = splat3(0.0f); float light_weight = 0.0f; uint32_t out_pixel = 0;
		if (e < n_list)
		{
			const uint32_t li = list[e].x, depth = list[e].y & 0xFFu, pass_k = list[e].y >> 8;
			StoredVertex lv;
			load_stored(P, li, depth, lv);
			const f3 delta = lv.position - P.eye;
			const float d2 = ieee_max(1.0e-8f, dot(delta, delta));
			const float d = sqrtf(d2);
			const f3 out = delta / d;
			const float cos_theta = dot(out, P.W) / P.W_len;
			const float G = fabsf(cos_theta * dot(out, lv.fr.n)) / d2;
			float ox = 0.0f, oy = 0.0f;
			const float p_s = camera_pdf(P, out, &ox, &oy);
			const float f_s = p_s * float(P.res_x * P.res_y);
			const f3 f_L = surface_f_sum(lv.bsdf, lv.fr, lv.in, -out, true);
			const float p_L = surface_p_sum(lv.bsdf, lv.fr, lv.in, -out, true);
			const float pGp = pdf3(p_s, G, p_L);
			const float next_pGp = pdf2(max_comp(f_L), lv.pG);
			const float mis_w =
				(depth == 1 && !P.opt.direct_lighting_nee && !P.opt.direct_lighting_bsdf) ? 1.0f :
				(depth > 1 && !P.opt.indirect_lighting_nee && !P.opt.indirect_lighting_bsdf) ? 1.0f :
				mis3(pGp / P.light_tracing, next_pGp, lv.pGp_sum);
			const f3 c = lv.alpha * f_L * f_s * G * mis_w;
			light_weight = 1.0f / float(P.n_paths);
			w = mk3(c.x * light_weight, c.y * light_weight, c.z * light_weight);
			if (max_comp(w) > 0.0f && finite3(w))
			{
				want = true;
				origin = lv.position + lv.in * kShadowBias;
				out_pixel = pass_k * P.n_paths + quantize(ox * 0.5f + 0.5f, P.res_x) + quantize(oy * 0.5f + 0.5f, P.res_y) * P.res_x;
			}
		}
		const uint32_t slot = block_range_alloc(P.shadow.size, want ? 1u : 0u, sc);
		if (want)
		{
			write_ray(P.shadow.rays, slot, origin, 0.0f, P.eye - origin, 0.9999f);
			P.shadow.weights[slot] = make_float4(w.x, w.y, w.z, 1.0f * light_weight);
			P.shadow.pixels[slot] = out_pixel;
		}
		__syncthreads();          // the scratch of block_range_alloc is reused by the next round
	}
}

// ConnectionsSink<true>: xyz of COMPOSITED_C and DIRECT_C, as order-independent 2^-32 fixed-point sums
__global__ void __launch_bounds__(BPT_BLOCK) splat_kernel(const BptParams P)
{
	const uint32_t s = threadIdx.x + blockIdx.x * blockDim.x;
	if (s >= *P.shadow.size) return;
	const float4 w = P.shadow.weights[s];
	if (!(w.x > 0.0f || w.y > 0.0f || w.z > 0.0f)) return;
	if (!(P.shadow.hits[s].x < 0.0f)) return;
	const uint32_t pixel = P.shadow.pixels[s];                 // virtual: pass offset * n_paths + pixel
	const float fw = frame_weight(P, path_ref(P, pixel).k);
	const float v[3] = { w.x * fw, w.y * fw, w.z * fw };
	#pragma unroll
	for (int c = 0; c < 3; ++c)
	{
		const long long q = __double2ll_rn(double(v[c]) * 4294967296.0);
		if (q) atomicAdd(reinterpret_cast<unsigned long long*>(P.splat + size_t(pixel) * 3 + c), (unsigned long long)q);
	}
}
__global__ void __launch_bounds__(BPT_BLOCK) splat_resolve_kernel(const BptParams P)
{
	const uint32_t p = threadIdx.x + blockIdx.x * blockDim.x;
	if (p >= P.n_paths * P.n_passes) return;
	long long* q = P.splat + size_t(p) * 3;
	const long long a = q[0], b = q[1], c = q[2];
	if (!(a | b | c)) return;
	const PathRef r = path_ref(P, p);
	const float fx = float(double(a) * (1.0 / 4294967296.0)), fy = float(double(b) * (1.0 / 4294967296.0)), fz = float(double(c) * (1.0 / 4294967296.0));
	float4* cc = fb_cell(P, FPT_FB_COMPOSITED_C, r); float4 v = *cc; v.x += fx; v.y += fy; v.z += fz; *cc = v;
	float4* dc = fb_cell(P, FPT_FB_DIRECT_C, r); float4 dch = *dc; dch.x += fx; dch.y += fy; dch.z += fz; *dc = dch;
	q[0] = q[1] = q[2] = 0;
}

// passes in flight: what n sequential BPT::render calls do to a pixel, pass by pass and in registers -- multiply_frame(i / (i + 1)) on the six
// channels; the albedo of the visible surface (one term per pass, summed into a zeroed plane); the eye path's cells in the order of the sequential
// launches (per bounce: emission, then the connections in light-depth order), each through the sink's arithmetic; the light-tracing splat sums of the
// pass (order-independent 2^-32 fixed point) -- and clear planes and fill bits (the caller clears the sums: a tile-sharded rank holds the sums of EVERY
// pixel but merges only its own).  Bit-identical to the sequential frame.
__global__ void __launch_bounds__(BPT_BLOCK) merge_exact_kernel(FrameBufferDev fb, float4* __restrict__ albedo_d, float4* __restrict__ albedo_s, BptLog log, long long* __restrict__ splat,
                                                                const uint32_t* __restrict__ pixels, uint32_t n_local, uint32_t n_paths, uint32_t base_instance, uint32_t n_passes)
{
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	if (i >= n_local) return;
	const uint32_t p = pixels ? pixels[i] : i;
	float4 c[6];
	#pragma unroll
	for (int ch = 0; ch < 6; ++ch) c[ch] = fb.ch[ch][p];
	for (uint32_t k = 0; k < n_passes; ++k)
	{
		const uint32_t inst = base_instance + k;
		const float scale = float(inst) / float(inst + 1), fw = 1.0f / float(inst + 1);
		const size_t vid = size_t(k) * n_paths + p;
		#pragma unroll
		for (int ch = 0; ch < 6; ++ch) c[ch] = make_float4(c[ch].x * scale, c[ch].y * scale, c[ch].z * scale, c[ch].w * scale);
		{
			const float4 a = albedo_d[vid], b = albedo_s[vid];
			albedo_d[vid] = make_float4(0.0f, 0.0f, 0.0f, 0.0f); albedo_s[vid] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
			c[FPT_FB_DIFFUSE_A].x += a.x; c[FPT_FB_DIFFUSE_A].y += a.y; c[FPT_FB_DIFFUSE_A].z += a.z; c[FPT_FB_DIFFUSE_A].w += a.w;
			c[FPT_FB_SPECULAR_A].x += b.x; c[FPT_FB_SPECULAR_A].y += b.y; c[FPT_FB_SPECULAR_A].z += b.z; c[FPT_FB_SPECULAR_A].w += b.w;
		}
		for (uint32_t word = 0; word < log.mask_words; ++word)
		{
			uint32_t* mp = log.mask + vid * log.mask_words + word;
			uint32_t m = *mp;
			if (!m) continue;
			*mp = 0u;
			while (m)
			{
				const uint32_t bit = uint32_t(__builtin_ctz(m)); m &= m - 1u;
				const size_t cell = size_t(word * 32u + bit) * log.cap + vid;
				const float4 v = log.val[cell];
				const uint32_t channel = log.chan[cell];
				float4& a = c[FPT_FB_COMPOSITED_C];
				a.x += v.x * fw; a.y += v.y * fw; a.z += v.z * fw; a.w += v.w * fw;
				if (channel != FPT_FB_COMPOSITED_C)
				{
					// (a switch keeps c[] in registers: a dynamically indexed array would go to scratch)
					#pragma unroll
					for (int ch = 0; ch < 6; ++ch)
						if (uint32_t(ch) == channel && ch != FPT_FB_COMPOSITED_C) { c[ch].x += v.x * fw; c[ch].y += v.y * fw; c[ch].z += v.z * fw; c[ch].w += v.w * fw; }
				}
			}
		}
		{
			long long* q = splat + vid * 3;
			const long long sa = q[0], sb = q[1], sc = q[2];
			if (sa | sb | sc)
			{
				const float fx = float(double(sa) * (1.0 / 4294967296.0)), fy = float(double(sb) * (1.0 / 4294967296.0)), fz = float(double(sc) * (1.0 / 4294967296.0));
				c[FPT_FB_COMPOSITED_C].x += fx; c[FPT_FB_COMPOSITED_C].y += fy; c[FPT_FB_COMPOSITED_C].z += fz;
				c[FPT_FB_DIRECT_C].x += fx; c[FPT_FB_DIRECT_C].y += fy; c[FPT_FB_DIRECT_C].z += fz;
			}
		}
	}
	#pragma unroll
	for (int ch = 0; ch < 6; ++ch) fb.ch[ch][p] = c[ch];
}

inline dim3 grid_for(uint32_t n) { return dim3((n + BPT_BLOCK - 1) / BPT_BLOCK); }

} // namespace

void launch_bpt_light_primary(const BptParams& p, hipStream_t s) { hipLaunchKernelGGL(light_primary_kernel, grid_for(p.n_local * p.n_passes), dim3(BPT_BLOCK), 0, s, p); }
void launch_bpt_light_vertices(const BptParams& p, uint32_t max_entries, hipStream_t s) { hipLaunchKernelGGL(light_vertices_kernel, grid_for(max_entries), dim3(BPT_BLOCK), 0, s, p); }
void launch_bpt_eye_primary(const BptParams& p, hipStream_t s) { hipLaunchKernelGGL(eye_primary_kernel, grid_for(p.n_local * p.n_passes), dim3(BPT_BLOCK), 0, s, p); }
void launch_bpt_eye_vertices(const BptParams& p, uint32_t max_entries, hipStream_t s) { hipLaunchKernelGGL(eye_vertices_kernel, grid_for(max_entries), dim3(BPT_BLOCK), 0, s, p); }
void launch_bpt_eye_resolve(const BptParams& p, uint32_t max_entries, hipStream_t s) { hipLaunchKernelGGL(eye_resolve_kernel, grid_for(max_entries), dim3(BPT_BLOCK), 0, s, p); }
void launch_bpt_pack_light_vertices(const LightVertexRecord* rec, const uint32_t* counts, const uint32_t* pixels, uint32_t n_local, uint32_t n_paths, uint32_t n_passes,
                                    LightVertexWire* out, uint32_t* out_count, hipStream_t s)
{ hipLaunchKernelGGL(pack_light_vertices_kernel, grid_for(n_local * n_passes), dim3(BPT_BLOCK), 0, s, rec, counts, pixels, n_local, n_paths, n_passes, out, out_count); }
void launch_bpt_unpack_light_vertices(const LightVertexWire* in, uint32_t n, LightVertexRecord* rec, float4* pos, uint32_t* counts, uint32_t n_store, hipStream_t s)
{ hipLaunchKernelGGL(unpack_light_vertices_kernel, grid_for(n), dim3(BPT_BLOCK), 0, s, in, n, rec, pos, counts, n_store); }
void launch_bpt_build_flat_list(const BptParams& p, hipStream_t s)
{
	const uint64_t total = uint64_t(p.n_passes) * p.opt.max_path_length * p.n_paths;
	const uint32_t n_blocks = uint32_t((total + uint64_t(BPT_BLOCK) * FLAT_ITEMS - 1) / (uint64_t(BPT_BLOCK) * FLAT_ITEMS));
	hipLaunchKernelGGL(flat_count_kernel, dim3(n_blocks), dim3(BPT_BLOCK), 0, s, p);
	hipLaunchKernelGGL(flat_scan_kernel, dim3(1), dim3(BPT_BLOCK), 0, s, p.flat_block_sums, n_blocks, p.flat_meta + 2 * p.n_passes);
	hipLaunchKernelGGL(flat_fill_kernel, dim3(n_blocks), dim3(BPT_BLOCK), 0, s, p);
}
void launch_bpt_connect_camera(const BptParams& p, hipStream_t s)
{ hipLaunchKernelGGL(connect_camera_kernel, grid_for(p.n_local * p.n_passes), dim3(BPT_BLOCK), 0, s, p); }
void launch_bpt_splat(const BptParams& p, uint32_t max_entries, hipStream_t s) { hipLaunchKernelGGL(splat_kernel, grid_for(max_entries), dim3(BPT_BLOCK), 0, s, p); }
void launch_bpt_splat_resolve(const BptParams& p, hipStream_t s) { hipLaunchKernelGGL(splat_resolve_kernel, grid_for(p.n_paths * p.n_passes), dim3(BPT_BLOCK), 0, s, p); }
void launch_bpt_merge_exact(const FrameBufferDev& fb, float4* albedo_d, float4* albedo_s, const BptLog& log, long long* splat, const uint32_t* pixels, uint32_t n_local,
                            uint32_t n_paths, uint32_t base_instance, uint32_t n_passes, uint32_t, hipStream_t s)
{ hipLaunchKernelGGL(merge_exact_kernel, grid_for(n_local), dim3(BPT_BLOCK), 0, s, fb, albedo_d, albedo_s, log, splat, pixels, n_local, n_paths, base_instance, n_passes); }

} // namespace fpt
