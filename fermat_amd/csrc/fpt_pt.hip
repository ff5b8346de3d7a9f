// fpt_pt.hip — the wavefront path-tracing kernels for gfx950 (everything of Fermat's -pt loop that is not traversal).
//
//   sequence_kernel        setup_samples_kernel                 src/tiled_sequence.cu:37-52,100-110
//   primary_rays_kernel    generate_primary_rays_kernel         src/pathtracer_kernels.h:133-181, pathtracer_core.h:633-656
//   shade_kernel           shade_hits_kernel -> shade_vertex    src/pathtracer_kernels.h:189-241, pathtracer_core.h:771-1254
//   (solve_occlusion_kernel, src/pathtracer_kernels.h:248-280, is fused into the any-hit traversal kernel, fpt_trace.hip)
//   rescale/variance/rgba  multiply_frame / update_variances /  src/renderer.cu:83-106,292-312,333-362
//                          to_rgba (kShaded)
//   merge_passes_exact_kernel  (no counterpart) ordered, bit-exact application of the passes in flight (PT and PSFPT), see fpt_pt_render_batch
// CDNA4 notes: wave64; queue appends are aggregated per WORKGROUP (ballot + popcount per wave, LDS prefix, one atomic per block —
// the gfx950 form of cugar::cuda::warp_increment, contrib/cugar/basic/cuda/warp_atomics.h:55-91; one atomic per wave saturates the
// counter at ~90 atomics/us); queue sizes stay in device memory and every kernel bounds itself by them, so a pass needs no host
// round trip; all queue traffic is 16-byte vector loads/stores; the per-frame QMC table is evaluated on the fly.
#include "fpt_device.h"
#include "fpt_kernels.h"
#include <cstdlib>
#include "fpt_psf.h"

namespace fpt {

// Queue-slot allocation aggregated over the whole workgroup: ONE device-scope atomic per block per queue.
// (A single counter sustains only ~90 atomics/us on MI355X; one atomic per wave made the append, not the shading, the
//  bottleneck of this kernel: 45k atomics = 0.5 ms per 1.44M-vertex launch.)  Every thread of the block must call this.
struct AppendScratch { uint32_t wave_count[SHADE_BLOCK / 64]; uint32_t base; };
__device__ __forceinline__ uint32_t block_append_slot(uint32_t* counter, bool want, AppendScratch& sc)
{
	const unsigned long long mask = __ballot(want);
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	if (lane == 0) sc.wave_count[wave] = uint32_t(__popcll(mask));
	__syncthreads();
	if (threadIdx.x == 0)
	{
		uint32_t total = 0;
		for (int w = 0; w < SHADE_BLOCK / 64; ++w) { const uint32_t c = sc.wave_count[w]; sc.wave_count[w] = total; total += c; }
		sc.base = total ? atomicAdd(counter, total) : 0u;
	}
	__syncthreads();
	return sc.base + sc.wave_count[wave] + uint32_t(__popcll(mask & ((1ull << lane) - 1ull)));
}

// TiledSequenceView::sample_2d (src/tiled_sequence.h:86-105) with the per-frame table folded in (see SequenceView)
__device__ __forceinline__ float sequence_sample(const SequenceView& s, uint32_t px, uint32_t py, uint32_t dim, uint32_t instance)
{
	const uint32_t T = s.tile_size;
	const uint32_t shift = (px & (T - 1)) + (py & (T - 1)) * T;
	const uint32_t tile  = ((px / T) & (T - 1)) + ((py / T) & (T - 1)) * T;
	const size_t base = size_t(dim) * T * T;
	const float sample = frac_pos(randfloat(dim, instance + 1) + s.shifts[base + shift]);      // = samples[dim][shift]
	return frac_pos(sample + s.shifts[base + tile]);
}

__global__ void sequence_kernel(uint32_t n_dims, uint32_t tile2, uint32_t instance, const float* __restrict__ shifts, float* __restrict__ samples)
{
	const uint32_t p = threadIdx.x + blockIdx.x * blockDim.x;
	if (p >= tile2) return;
	for (uint32_t d = 0; d < n_dims; ++d)
	{
		const float s = randfloat(d, instance + 1);                // TiledSequence::set_instance (src/tiled_sequence.cu:100-110)
		samples[p + size_t(d) * tile2] = frac_pos(s + shifts[p + size_t(d) * tile2]);
	}
}

__global__ void primary_rays_kernel(const PrimaryParams P)
{
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	const uint32_t n_paths = P.n_pixels * P.pass.n_passes;
	if (i >= n_paths) return;
	const uint32_t k = i / P.n_pixels, li = i - k * P.n_pixels;         // pass offset, local pixel slot
	const uint32_t idx = P.pixels ? P.pixels[li] : li;
	const uint32_t instance = P.pass.base_instance + k;
	const uint32_t px = idx % P.res_x, py = idx / P.res_x;
	const float ux = sequence_sample(P.seq, px, py, 0, instance), uy = sequence_sample(P.seq, px, py, 1, instance);
	const float dx = ((float(px) + ux) / float(P.res_x)) * 2.f - 1.f;
	const float dy = ((float(py) + uy) / float(P.res_y)) * 2.f - 1.f;
	const f3 dir = dx * P.U + dy * P.V + P.W;
	// the ray's interval is the queue's (tmin 0 = mask 0, tmax 1e34: fpt_device.h QUEUE_PRIMARY_*); its .w words carry PixelInfo -- comp 0, diffuse 0; the pixel field: the
	// absolute pixel, or (passes in flight) the slot of the rank's pixel list -- and the pass offset
	P.out.rays[2 * size_t(i)]     = make_float4(P.eye.x, P.eye.y, P.eye.z, as_f32(P.pass.n_passes == 1 ? idx : li));
	P.out.rays[2 * size_t(i) + 1] = make_float4(dir.x, dir.y, dir.z, as_f32(P.pass.n_passes > 1 ? k : 0u));
	P.out.weights[i] = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
	if (P.out.vinfo) P.out.vinfo[i] = 0xFFFFFFFFu;                      // make_uint4(idx, -1, -1, -1): no cache cell yet
	// camera_direction_pdf (src/camera.h:231-252, solid-angle form)
	float pdf = 0.0f;
	const float t = dot(dir, P.W) / (P.W_len * P.W_len);
	if (!(t < 0.0f))
	{
		const f3 I = dir / t - P.W;
		const float Ix = dot(I, P.U) / dot(P.U, P.U);
		const float Iy = dot(I, P.V) / dot(P.V, P.V);
		if (Ix >= -1.0f && Ix <= 1.0f && Iy >= -1.0f && Iy <= 1.0f)
		{
			const float ct = dot(dir, P.W) / P.W_len;
			pdf = P.sq_focal / (ct * ct * ct);
		}
	}
	if (P.out.cones) P.out.cones[i] = make_float2(0.0f, pdf);
	if (i == 0) *P.out.size = n_paths;
}

// power heuristic with the reference's non-finite handling (src/mis_utils.h:43-52)
__device__ __forceinline__ float mis_power(float p1, float p2)
{
	return !is_finite(p1) ? 1.0f : (!is_finite(p2) ? 0.0f : (p1 * p1) / (p1 * p1 + p2 * p2));
}

// GBufferView::pack_geometry's normal word (src/framebuffer.h:97-104): sphere -> unit square -> 15:15 bits
__device__ __forceinline__ float pack_gbuffer_normal(f3 N)
{
	float phi;
	if (fabsf(N.z) >= 1.0f - 1.0e-5f) phi = 0.0f;
	else { phi = det_atan2(N.y, N.x); phi = phi < 0.0f ? phi + 2.0f * kPi : phi; }
	const float sx = phi / (2.0f * kPi), sy = (N.z + 1.0f) * 0.5f;
	const uint32_t M = (1u << 15) - 1u;
	return as_f32(quantize(sx, M) | (quantize(sy, M) << 15));
}

// one light sample -> at most one shadow-queue entry.  Shared by the directional-light and mesh-light branches
// (src/pathtracer_core.h:895-988 and :1013-1106; weights per PTVertexProcessor::compute_nee_weights,
//  src/pathtracer_vertex_processor.h:83-105).  Returns whether a shadow ray is wanted and fills its payload.
struct ShadowPayload { f3 org, dir, w_d, w_g; };
// psf_mode: 0 = PTVertexProcessor weights; 1 = PSFPTVertexProcessor, plain; 2 = PSFPTVertexProcessor at a new, valid cache vertex (the diffuse
// weight is demodulated by the surface albedo `demod`) — compute_nee_weights, src/psfpt_vertex_processor.h:189-248
__device__ __forceinline__ f3 demodulate(f3 f, f3 c) { return mk3(f.x / sel_max(c.x, 1.0e-4f), f.y / sel_max(c.y, 1.0e-4f), f.z / sel_max(c.z, 1.0e-4f)); }      // src/filters.h:63-67
__device__ __forceinline__ bool light_sample(const ShadeParams& P, const SurfaceModel& bsdf, const ViewTerms& vt, const SurfacePoint& sp, f3 in, f3 ray_dir, f3 w,
                                             f3 light_pos, f3 light_n, f3 light_radiance, float light_pdf, bool use_mis, float origin_eps, ShadowPayload& out,
                                             int psf_mode = 0, f3 demod = f3{ 1.0f, 1.0f, 1.0f })
{
	f3 dir_out = light_pos - sp.position;
	const float d2 = ieee_max(1.0e-8f, dot(dir_out, dir_out));
	dir_out = dir_out * (1.0f / sqrtf(d2));
	f3 f_s[4]; float p_s[4];
	surface_f_and_p(bsdf, sp.frame, vt, in, dir_out, f_s, p_s);
	const bool ev_d = P.opt.diffuse_scattering != 0, ev_g = P.opt.glossy_scattering != 0;
	float p_sum = 0.0f;
	if (ev_d) p_sum += p_s[LOBE_DIFF_R] + p_s[LOBE_DIFF_T];
	if (ev_g) p_sum += p_s[LOBE_GLOSSY_R] + p_s[LOBE_GLOSSY_T];
	const f3 f_L = (dot(light_n, -dir_out) > 0.0f ? light_radiance : splat3(0.0f)) / light_pdf;
	const float G = fabsf(dot(dir_out, sp.frame.n) * dot(dir_out, light_n)) / d2;
	float mis_w = 1.0f;
	if (use_mis)
		mis_w = ((P.bounce == 0 && P.opt.direct_lighting_bsdf) || (P.bounce > 0 && P.opt.indirect_lighting_bsdf)) ? mis_power(light_pdf, p_sum * G) : 1.0f;
	const f3 f_d = ev_d ? f_s[LOBE_DIFF_R] + f_s[LOBE_DIFF_T] : splat3(0.0f);
	const f3 f_g = ev_g ? f_s[LOBE_GLOSSY_R] + f_s[LOBE_GLOSSY_T] : splat3(0.0f);
	const f3 fl = f_L * G * mis_w;
	if (psf_mode == 0)
	{
		out.w_d = (P.bounce == 0 ? f_d : f_d + f_g) * w * fl;
		out.w_g = (P.bounce == 0 ? f_g : f_d + f_g) * w * fl;
	}
	else
	{
		out.w_d = psf_mode == 2 ? demodulate(f_d, demod) * fl : f_d * w * fl;
		out.w_g = f_g * w * fl;
	}
	const f3 w_sum = out.w_d + out.w_g;
	if (!(max_comp(w_sum) > 0.0f && all_finite(w_sum))) return false;
	out.org = sp.position - ray_dir * origin_eps;
	out.dir = light_pos - out.org;
	return true;
}
__device__ __forceinline__ void write_shadow_entry(const ShadowQueue& q, uint32_t slot, const ShadowPayload& pl, uint32_t mask, uint32_t pixel_info, bool batched, uint32_t pass_k)
{
	// ShadowQueue: origin | mask, dir | PixelInfo (the ray's tmax is the queue's 0.9999), w_d | pass offset, w_g
	q.rays[2 * size_t(slot)]     = make_float4(pl.org.x, pl.org.y, pl.org.z, as_f32(mask));
	q.rays[2 * size_t(slot) + 1] = make_float4(pl.dir.x, pl.dir.y, pl.dir.z, as_f32(pixel_info));
	q.w_d[slot] = make_float4(pl.w_d.x, pl.w_d.y, pl.w_d.z, as_f32(batched ? pass_k : 0u));
	q.w_g[slot] = make_float4(pl.w_g.x, pl.w_g.y, pl.w_g.z, 0.0f);
}

// ---- path-space filtering helpers (src/psfpt_vertex_processor.h, src/spatial_hash.h) ------------------------------------------------
__device__ __forceinline__ float round_half_down(float x) { const int y = x > 0.0f ? to_i32_sat(x) : to_i32_sat(x) - 1; return (x - float(y) > 0.5f) ? float(y) + 1.0f : float(y); }    // cugar::round
// spatial_hash, 10-argument overload (src/spatial_hash.h:86-167)
__device__ __forceinline__ unsigned long long spatial_hash(f3 Ppos, f3 N, f3 T, f3 B, f3 lo, f3 hi, const float s[6], float cone_radius, float filter_radius)
{
	const uint32_t normal_bits = 4;
	const float world_extent = max_comp(hi - lo);
	const float float_grid = sel_max(world_extent / (2.0f * cone_radius), 1.0f);
	const float flog = det_log2(float_grid);
	const uint32_t ilog = to_u32_sat(flog);
	const float rlog = flog - float(ilog);
	const uint32_t log_i = ilog + (s[5] < rlog ? 1u : 0u);
	const uint32_t grid = 1u << log_i;
	// concentric disk sample (the same map as cosine_hemisphere's first two components)
	const f3 dsk = cosine_hemisphere(s[0], s[1]);
	const float rs = filter_radius * cone_radius;
	const float rx = rs * dsk.x, ry = rs * dsk.y;
	const f3 q = ((Ppos + T * rx) + B * ry) - lo;
	const f3 loc = (float(grid) * q) / world_extent;
	const uint32_t lx = to_u32_sat(sel_max(round_half_down(loc.x), 0.0f)), ly = to_u32_sat(sel_max(round_half_down(loc.y), 0.0f)), lz = to_u32_sat(sel_max(round_half_down(loc.z), 0.0f));
	const float nj = float(1u << (normal_bits / 2));
	float phi;
	if (fabsf(N.z) >= 1.0f - 1.0e-5f) phi = 0.0f;
	else { phi = det_atan2(N.y, N.x); phi = phi < 0.0f ? phi + 2.0f * kPi : phi; }
	float nu = phi / (2.0f * kPi), nv = (N.z + 1.0f) * 0.5f;
	nu = mod1(nu + s[3] / nj);
	nv = sel_min(nv + s[4] / nj, 1.0f);
	const uint32_t M = (1u << (normal_bits / 2)) - 1u;
	const uint32_t normal_i = quantize(nu, M) | (quantize(nv, M) << (normal_bits / 2));
	const uint32_t cm = (1u << 17) - 1u;
	return ((unsigned long long)(lx & cm)) | ((unsigned long long)(ly & cm) << 17) | ((unsigned long long)(lz & cm) << 34) | ((unsigned long long)log_i << 51) | ((unsigned long long)normal_i << 56);
}
// open addressing, linear probing; a cell is addressed by its key, so the slot it lands in does not matter
__device__ __forceinline__ uint32_t psf_insert(const PsfDev& psf, unsigned long long key)
{
	const uint32_t mask = (1u << psf.log2_size) - 1u;
	uint32_t h = uint32_t((key * 0x9E3779B97F4A7C15ull) >> (64 - psf.log2_size)) & mask;
	for (uint32_t probe = 0; probe <= mask; ++probe)
	{
		const unsigned long long prev = atomicCAS(psf.keys + h, ~0ull, key);
		if (prev == ~0ull && psf.touched) psf.touched[atomicAdd(psf.touched_n, 1u)] = h;      // sharded: the creator lists the slot (a few thousand per pass; the list is as long as the table)
		if (prev == ~0ull || prev == key) return h;
		h = (h + 1u) & mask;
	}
	return 0x1FFFFFFFu;       // table full: the vertex stays uncached
}
#ifndef FPT_SHADE_MIN_WAVES
#define FPT_SHADE_MIN_WAVES 4
#endif
// FPT_SHADE_SKIP: a bit mask that compiles sections of shade_kernel OUT -- 1 directional lights, 2 mesh-light NEE, 4 emissive hit, 8 scattering, 16 the albedo /
// gbuffer writes of bounce 0, 32 the six QMC samples.  Never set in the product build: tools/shade_sections.py counts the instructions of each section by difference
#ifndef FPT_SHADE_SKIP
#define FPT_SHADE_SKIP 0
#endif
// One kernel per vertex: the two-way fission of this kernel (vertex set-up + NEE + emissive | vertex set-up + scatter) was measured and rejected
// (shading 0.459 vs 0.370 ms per step: the second set-up costs more than the smaller half's occupancy returns).
template <bool PSF>
__global__ __launch_bounds__(SHADE_BLOCK, FPT_SHADE_MIN_WAVES)
void shade_kernel(const ShadeParams P)
{
	__shared__ AppendScratch sc_dir, sc_nee, sc_scatter, sc_ref;
	uint32_t prev_vinfo = 0xFFFFFFFFu, vinfo = 0xFFFFFFFFu;
	int psf_mode = 0; f3 mat_diffuse = splat3(1.0f);
	// One thread per queue entry.  A miss ends the path (no sky lighting, src/pathtracer_core.h:1249-1252) and its lane idles through the kernel -- a third
	// of the bounce-1 entries of the bench frame.  Shading only the hits was measured twice and dropped: listing the hits of a 2048-entry
	// tile in LDS at the head of every block made the kernel 30 % slower (0.504 vs 0.389 ms per step: sixteen loads and two barriers in front of a
	// 13-us block); a separate compaction kernel + an index list broke even (0.380 vs 0.389, 0.365 vs 0.365: what the fuller waves save, the extra pass
	// over the hit records and the gathered loads cost).
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	const uint32_t n_in = *P.in.size;
	if (blockIdx.x * blockDim.x >= n_in) return;                       // whole block beyond the queue: uniform exit

	float4 hit4 = make_float4(-1.0f, as_f32(0xFFFFFFFFu), 0.0f, 0.0f);
	if (i < n_in) hit4 = P.in.hits[i];
	const float hit_t = hit4.x;
	const int32_t tri = int32_t(as_u32(hit4.y));
	// inactive threads still take part in the block-wide queue appends below
	const bool active = (i < n_in) && (hit_t > 0.0f && tri >= 0);

	uint32_t pixel_info = 0, pixel = 0;
	PathSlot slot; slot.pixel = 0; slot.k = 0; slot.weight = 0.0f; slot.slot = 0;
	f3 ray_dir = splat3(0.0f), w = splat3(0.0f), in = splat3(0.0f);
	float p_prev = 0.0f, cone_radius = 0.0f;
	SurfacePoint sp;
	SurfaceModel bsdf;
	ViewTerms vt;
	f4 m_emissive = mk4(0, 0, 0, 0);
	float z[6] = { 0, 0, 0, 0, 0, 0 };

	if (active)
	{
		const float4 ro = P.in.rays[2 * size_t(i)], rd4 = P.in.rays[2 * size_t(i) + 1];
		const float4 w4 = P.in.weights[i];
		pixel_info = as_u32(ro.w);                                                        // PathQueue: origin | PixelInfo, dir | pass offset
		const float2 cone = P.in.cones ? P.in.cones[i] : make_float2(0.0f, 1.0f);      // (no cone plane: the plain path tracer, whose vertices never read it)
		slot = decode_slot(P.pass, pixel_info, P.pass.n_passes > 1 ? as_u32(rd4.w) : 0u);
		pixel = slot.pixel;
		const uint32_t instance = P.pass.base_instance + slot.k;
		const uint32_t px = pixel % P.res_x, py = pixel / P.res_x;
		ray_dir = mk3(rd4.x, rd4.y, rd4.z);
		w = mk3(w4.x, w4.y, w4.z);
		p_prev = w4.w;

		// ---- EyeVertex::setup (src/bpt_utils.h:585-642) ----
		uint32_t material_index;
		if (P.shade_records)
		{
			const ShadeRecord rec = P.shade_records[tri];
			surface_point(rec, P.mesh, hit4.z, hit4.w, sp);
			material_index = as_u32(rec.d.w);
		}
		else
		{
			surface_point(P.mesh, uint32_t(tri), hit4.z, hit4.w, sp);
			material_index = uint32_t(P.mesh.material_indices[tri]);
		}
		sp.position = mk3(ro.x, ro.y, ro.z) + hit_t * ray_dir;
		const fpt_material* mat = P.mesh.materials + material_index;
		const f4 one4 = mk4(1, 1, 1, 1);
		const fpt_texture* textures = (FPT_SHADE_SKIP & 64) ? nullptr : P.textures;
		const f4 m_diffuse  = load4(mat->diffuse)       * ((FPT_SHADE_SKIP & 64) ? one4 : sample_texture(textures, mat->diffuse_map, sp.s, sp.t, one4));
		const f4 m_specular = load4(mat->specular)      * ((FPT_SHADE_SKIP & 64) ? one4 : sample_texture(textures, mat->specular_map, sp.s, sp.t, one4));
		m_emissive          = load4(mat->emissive)      * ((FPT_SHADE_SKIP & 64) ? one4 : sample_texture(textures, mat->emissive_map, sp.s, sp.t, one4));
		const f4 m_dtrans   = load4(mat->diffuse_trans) * ((FPT_SHADE_SKIP & 64) ? one4 : sample_texture(textures, mat->diffuse_trans_map, sp.s, sp.t, one4));
		in = -normalize(ray_dir);
		bsdf = make_surface_model(xyz(m_diffuse), xyz(m_dtrans), xyz(m_specular), xyz(load4(mat->reflectivity)),
		                          mat->roughness, mat->index_of_refraction, mat->opacity, P.table);
		vt = view_terms(bsdf, sp.frame, in);
		const float prev_G_prime = fabsf(dot(in, sp.frame.n)) / (hit_t * hit_t);

		if (!(FPT_SHADE_SKIP & 16) && P.bounce == 0)
		{
			// gbuffer of the frame = the last pass of the batch (the reference clears and rewrites it every pass, src/renderer.cu:1039)
			if (P.gbuffer.gb_geo && slot.k + 1 == P.pass.n_passes)
			{
				P.gbuffer.gb_geo[pixel] = make_float4(sp.position.x, sp.position.y, sp.position.z, pack_gbuffer_normal(sp.frame.n));
				P.gbuffer.gb_uv[pixel] = make_float4(hit4.z, hit4.w, sp.s, sp.t);
				P.gbuffer.gb_tri[pixel] = uint32_t(tri);
				P.gbuffer.gb_depth[pixel] = hit_t;
			}
			// surface albedos (src/pathtracer_core.h:809-811): fb += albedo * frame_weight, all four components
			float4* ca = P.fb.ch[FPT_FB_DIFFUSE_A] + size_t(slot.k) * (P.pass.n_passes == 1 ? 0u : P.pass.acc_stride) + slot.slot;
			float4* cs = P.fb.ch[FPT_FB_SPECULAR_A] + size_t(slot.k) * (P.pass.n_passes == 1 ? 0u : P.pass.acc_stride) + slot.slot;
			// (passes in flight: the cell is the path's own cell of a per-pass plane, which the merge leaves zeroed and nobody else writes -- 0 + a is stored without
			//  reading the zero; a zero of either sign adds the same to the frame)
			const bool own_cell = P.pass.n_passes > 1;
			const f4 a = (own_cell ? mk4(0.0f, 0.0f, 0.0f, 0.0f) : load4(reinterpret_cast<const float*>(ca))) + m_diffuse * slot.weight;
			store4(reinterpret_cast<float*>(ca), a);
			const f4 sa = (own_cell ? mk4(0.0f, 0.0f, 0.0f, 0.0f) : load4(reinterpret_cast<const float*>(cs))) + (m_specular + one4) * 0.5f * slot.weight;
			store4(reinterpret_cast<float*>(cs), sa);
		}
		if (P.in.cones) cone_radius = cone.x + 1.0f / sqrtf(cone.y * prev_G_prime);      // Bekaert footprint (:816-819)
		if (PSF) { prev_vinfo = P.in.vinfo[i]; mat_diffuse = xyz(m_diffuse); }
		#pragma unroll
		for (uint32_t k = 0; k < 6; ++k) z[k] = (FPT_SHADE_SKIP & 32) ? float(px + k) * 0.01f : sequence_sample(P.seq, px, py, (P.bounce + 1) * 6 + k, instance);
	}

	// ---- PSFPTVertexProcessor::preprocess_vertex (src/psfpt_vertex_processor.h:76-187) ----
	if (PSF)
	{
		bool want_ref = false; uint32_t ref_cache = 0; f4 ref_wd = mk4(0, 0, 0, 0), ref_wg = mk4(0, 0, 0, 0);
		if (active)
		{
			uint32_t new_slot = prev_vinfo & 0x1FFFFFFFu; bool new_entry = false;
			if (!ci_valid(prev_vinfo) && P.bounce >= P.psf.depth && p_prev < P.psf.max_prob)
			{
				const uint32_t pixel_hash = pixel + (P.psf.instance + slot.k) * P.res_x * P.res_y;
				float jitter[6];
				#pragma unroll
				for (uint32_t k = 0; k < 6; ++k) jitter[k] = randfloat(k, pixel_hash);
				const f3 Nf = dot(in, sp.frame.n) > 0.0f ? sp.frame.n : -sp.frame.n;
				const unsigned long long key = spatial_hash(sp.position, Nf, sp.frame.t, sp.frame.b, P.psf.bbox_lo, P.psf.bbox_hi, jitter, cone_radius * P.psf.width, P.bounce == 0 ? 2.0f : 1.0f);
				const PsfDev table = psf_pass_view(P.psf, slot.k);
				new_slot = psf_insert(table, key);
				if (new_slot != 0x1FFFFFFFu)
				{
					atomicAdd(reinterpret_cast<unsigned long long*>(table.cells + 4 * size_t(new_slot) + 3), 1ull);
					const f4 w_mod = mk4(w.x * sel_max(mat_diffuse.x, 1.0e-4f), w.y * sel_max(mat_diffuse.y, 1.0e-4f), w.z * sel_max(mat_diffuse.z, 1.0e-4f), 0.0f);
					const uint32_t comp = (pixel_info >> 27) & 0xFu;
					want_ref = true; ref_cache = ci_pack(new_slot, 3u, 0u);
					ref_wd = (comp & COMP_DIFFUSE_MASK) ? w_mod : mk4(0, 0, 0, 0);
					ref_wg = ((comp & COMP_GLOSSY_MASK) && P.bounce) ? w_mod : mk4(0, 0, 0, 0);
					new_entry = true;
				}
			}
			vinfo = ci_pack(new_slot, 0u, new_entry ? 1u : 0u);
			psf_mode = (new_entry && !(P.bounce < P.psf.depth) && ci_valid(vinfo)) ? 2 : 1;
		}
		const uint32_t rslot = block_append_slot(P.psf.ref_size, want_ref, sc_ref);
		if (want_ref)
		{
			P.psf.ref_pixels[rslot] = pixel_info; P.psf.ref_cache[rslot] = ref_cache; if (P.pass.n_passes > 1) P.psf.ref_k[rslot] = slot.k;
			P.psf.ref_wd[rslot] = make_float4(ref_wd.x, ref_wd.y, ref_wd.z, ref_wd.w); P.psf.ref_wg[rslot] = make_float4(ref_wg.x, ref_wg.y, ref_wg.z, ref_wg.w);
		}
	}
	// ---- directional lights (:870-988) ----
	if (!(FPT_SHADE_SKIP & 1) && (P.bounce + 2 <= P.opt.max_path_length) && (P.bounce > 0 || P.opt.direct_lighting) && P.n_dir_lights)
	{
		ShadowPayload pl; bool want = false;
		if (active)
		{
			const fpt_dir_light L = P.dir_lights[quantize(z[2], P.n_dir_lights)];
			const f3 ldir = mk3(L.dir[0], L.dir[1], L.dir[2]);
			const float FAR = 1.0e8f;
			const f3 lpos = sp.position - ldir * FAR;
			const f3 lrad = FAR * FAR * mk3(L.color[0], L.color[1], L.color[2]);
			const float lpdf = 1.0f / float(P.n_dir_lights);
			want = light_sample(P, bsdf, vt, sp, in, ray_dir, w, lpos, ldir, lrad, lpdf, false, 1.0e-3f, pl, psf_mode, mat_diffuse);
		}
		const uint32_t qslot = block_append_slot(P.shadow_dir.size, want, sc_dir);
		if (want) { write_shadow_entry(P.shadow_dir, qslot, pl, 0x1u, pixel_info, P.pass.n_passes > 1, slot.k); if (PSF) P.shadow_dir.vinfo[qslot] = vinfo; }
	}
	// ---- next-event estimation on the mesh emitters (:991-1106) ----
	if (!(FPT_SHADE_SKIP & 2) && P.do_nee)
	{
		ShadowPayload pl; bool want = false;
		if (active)
		{
			const LightPoint lp = emitter_light_point(P.emitters, P.mesh, P.textures, z[0], z[1], z[2]);
			want = light_sample(P, bsdf, vt, sp, in, ray_dir, w, lp.position, lp.normal, lp.radiance, lp.pdf, true, 1.0e-4f, pl, psf_mode, mat_diffuse);
		}
		const uint32_t qslot = block_append_slot(P.shadow.size, want, sc_nee);
		if (want) { write_shadow_entry(P.shadow, qslot, pl, 0x2u, pixel_info, P.pass.n_passes > 1, slot.k); if (PSF) P.shadow.vinfo[qslot] = vinfo; }
	}
	// ---- emissive surface hit, MIS against NEE at the previous vertex (:1109-1154) ----
	// (A surface that emits nothing -- nearly every hit -- has nothing to add: with m_emissive = 0 the sample e below is w * 0 * mis_w, i.e. 0 or NaN, and neither passes
	//  the `max_comp(e) > 0 && all_finite(e)` test that guards every use of it.  Testing the emission first skips the pdf look-ups and the MIS weight for whole waves.)
	if (!(FPT_SHADE_SKIP & 4) && P.do_emissive && active && (m_emissive.x != 0.0f || m_emissive.y != 0.0f || m_emissive.z != 0.0f))
	{
		f3 lrad; float lpdf;
		if (P.emitters.n_vpls || P.emitters.n_prims)
		{
			if (P.emitters.n_vpls) lpdf = emission_pdf_measure(m_emissive) / P.emitters.norm;
			else                   lpdf = (P.emitters.prims_cdf[tri] - (tri ? P.emitters.prims_cdf[tri - 1] : 0)) * P.emitters.prims_inv_area[tri];
			lrad = xyz(m_emissive);
		}
		else { lpdf = 1.0f; lrad = splat3(0.0f); }
		const f3 f_L = dot(sp.frame.n, in) > 0.0f ? lrad : splat3(0.0f);
		const float d2 = ieee_max(1.0e-10f, hit_t * hit_t);
		const float G_partial = fabsf(dot(in, sp.frame.n)) / d2;
		const float p1 = (is_finite(G_partial) && is_finite(p_prev)) ? G_partial * p_prev : inf_f();
		const float mis_w = ((P.bounce == 1 && P.opt.direct_lighting_nee) || (P.bounce > 1 && P.opt.indirect_lighting_nee)) ? mis_power(p1, lpdf) : 1.0f;
		const f3 e = w * f_L * mis_w;
		if (PSF && max_comp(e) > 0.0f && all_finite(e))
		{
			// PSFPTVertexProcessor::accumulate_emissive (src/psfpt_vertex_processor.h:288-343): to the image until a cache vertex exists, to its cell afterwards
			const f3 c = psf_clamp(P.psf, e);
			if (!ci_valid(prev_vinfo)) accumulate_emissive(P.fb, P.pass, P.log, slot, pixel_info, P.bounce, c);
			else psf_add(psf_pass_view(P.psf, slot.k), prev_vinfo & 0x1FFFFFFFu, c);
		}
		else if (max_comp(e) > 0.0f && all_finite(e))
		{
			// PTVertexProcessor::accumulate_emissive (src/pathtracer_vertex_processor.h:151-183): to the frame, or to the path's cell of the batch's log
			accumulate_emissive(P.fb, P.pass, P.log, slot, pixel_info, P.bounce, e);
		}
	}
	// ---- scattering (:1157-1247) ----
	if (!(FPT_SHADE_SKIP & 8) && P.do_scatter)
	{
		f3 out = splat3(0.0f), out_w = splat3(0.0f); float p = 0.0f; uint32_t comp = COMP_ABSORB; bool want = false;
		if (active)
		{
			f3 g; float p_proj;
			comp = surface_sample(bsdf, sp.frame, vt, z[3], z[4], z[5], in, out, p, p_proj, g);
			out_w = g * w;
			// PSFPTVertexProcessor::compute_scattering_weights (src/psfpt_vertex_processor.h:250-286)
			if (PSF && (vinfo >> 31) && (comp & COMP_DIFFUSE_MASK)) out_w = demodulate(g, mat_diffuse);
			want = comp != COMP_ABSORB && p != 0.0f && max_comp(out_w) > 0.0f && all_finite(out_w);
		}
		const uint32_t qslot = block_append_slot(P.scatter.size, want, sc_scatter);
		if (want)
		{
			// the scattered ray's interval is the queue's (1e-3, 1e8: fpt_device.h QUEUE_SCATTER_*); its .w words carry PixelInfo and the pass offset
			const uint32_t diffuse_bit = ((pixel_info >> 31) || (comp & COMP_DIFFUSE_MASK)) ? 1u : 0u;
			const uint32_t out_info = (pixel_info & 0x7FFFFFFu) | ((comp & 0xFu) << 27) | (diffuse_bit << 31);
			P.scatter.rays[2 * size_t(qslot)]     = make_float4(sp.position.x, sp.position.y, sp.position.z, as_f32(out_info));
			P.scatter.rays[2 * size_t(qslot) + 1] = make_float4(out.x, out.y, out.z, as_f32(P.pass.n_passes > 1 ? slot.k : 0u));
			P.scatter.weights[qslot] = make_float4(out_w.x, out_w.y, out_w.z, p);
			if (P.scatter.cones) P.scatter.cones[qslot] = make_float2(cone_radius, sel_max(p, 32.0f));
			if (PSF) P.scatter.vinfo[qslot] = (!ci_valid(prev_vinfo) && (comp & COMP_GLOSSY_MASK)) ? prev_vinfo : ci_pack(vinfo & 0x1FFFFFFFu, 3u, 0u);
		}
	}
}

// PSFPTVertexProcessor::accumulate_nee over a traced shadow queue (src/psfpt_vertex_processor.h:345-441)
__global__ void psf_resolve_kernel(const ResolveParams P)
{
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	if (i >= *P.q.size) return;
	if (P.hits[i].x > 0.0f) return;
	psf_resolve_sample(P, P.pass.base_instance, i);
}

// psf_blending_kernel (src/renderers/psfpt_impl.h:86-125), launched once per bounce over that bounce's references: a path owns at most
// one reference per bounce, so the frame-buffer updates need no atomics and a pixel's references are blended in creation order.
__global__ void psf_blend_kernel(PsfDev psf, FrameBufferDev fb, float frame_weight)
{
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	if (i >= *psf.ref_size) return;
	const uint32_t cache = psf.ref_cache[i];
	if (!ci_valid(cache)) return;
	const long long* cell = psf.cells + 4 * size_t(cache & 0x1FFFFFFFu);
	if (psf.g_keys)
	{
		// sharded: the reference names a slot of the pass table; its key finds the cell of the global table (merged from every rank's records, so it exists)
		const unsigned long long key = psf.keys[cache & 0x1FFFFFFFu];
		const uint32_t mask = (1u << psf.g_log2_size) - 1u;
		uint32_t h = uint32_t((key * 0x9E3779B97F4A7C15ull) >> (64 - psf.g_log2_size)) & mask;
		// the cell exists unless the global table was full when the merge tried to insert it (psf_merge_kernel drops the record then): stop at the
		// first empty slot of the probe sequence and fall back to the rank's own pass-table cell instead of walking the table
		bool found = false;
		for (uint32_t probe = 0; probe <= mask; ++probe)
		{
			const unsigned long long k = psf.g_keys[h];
			if (k == key) { found = true; break; }
			if (k == ~0ull) break;
			h = (h + 1u) & mask;
		}
		if (found) cell = psf.g_cells + 4 * size_t(h);
	}
	if (cell[3] == 0) return;                      // an empty cell holds no estimate (0 / 0)
	const float cw = float((unsigned long long)cell[3]);
	const f3 cv = mk3(float(double(cell[0]) * (1.0 / 4294967296.0)) / cw, float(double(cell[1]) * (1.0 / 4294967296.0)) / cw, float(double(cell[2]) * (1.0 / 4294967296.0)) / cw);
	const uint32_t pixel_info = psf.ref_pixels[i];
	const uint32_t pixel = pixel_info & 0x7FFFFFFu, comp = (pixel_info >> 27) & 0xFu;
	const float4 wd4 = psf.ref_wd[i], wg4 = psf.ref_wg[i];
	const f3 w_d = mk3(wd4.x, wd4.y, wd4.z), w_g = mk3(wg4.x, wg4.y, wg4.z);
	const f3 w = ((comp & COMP_DIFFUSE_MASK) ? w_d : splat3(0.0f)) + ((comp & COMP_GLOSSY_MASK) ? w_g : splat3(0.0f));
	const f3 cvw = cv * w;
	fb_add<false>(fb.ch[FPT_FB_COMPOSITED_C], pixel, mk3(sel_min(cvw.x, psf.firefly), sel_min(cvw.y, psf.firefly), sel_min(cvw.z, psf.firefly)), frame_weight);
	if (comp & COMP_DIFFUSE_MASK) fb_add<true>(fb.ch[FPT_FB_DIFFUSE_C], pixel, cv * w_d, frame_weight);
	if (comp & COMP_GLOSSY_MASK)  fb_add<true>(fb.ch[FPT_FB_SPECULAR_C], pixel, cv * w_g, frame_weight);
}

// the blend of a batch (fpt_psfpt_render_batch): a reference names a slot of ITS pass's table, which psf_prefix_kernel has turned into the state
// of the cache after that pass (what the sequential blend would read); the sample goes to the pass's accumulation plane
__global__ void psf_blend_batch_kernel(PsfDev psf, ContribLog log, uint32_t bounce, PassInfo pass)
{
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	if (i >= *psf.ref_size) return;
	const uint32_t cache = psf.ref_cache[i];
	if (!ci_valid(cache)) return;
	const uint32_t pixel_info = psf.ref_pixels[i];
	const PathSlot sl = decode_slot(pass, pixel_info, psf.ref_k[i]);
	const uint32_t comp = (pixel_info >> 27) & 0xFu;
	const long long* cell = psf_pass_view(psf, sl.k).cells + 4 * size_t(cache & 0x1FFFFFFFu);
	if (cell[3] == 0) return;                      // an empty cell holds no estimate (0 / 0): no blend cell, as psf_blend_kernel adds nothing
	const float cw = float((unsigned long long)cell[3]);
	const f3 cv = mk3(float(double(cell[0]) * (1.0 / 4294967296.0)) / cw, float(double(cell[1]) * (1.0 / 4294967296.0)) / cw, float(double(cell[2]) * (1.0 / 4294967296.0)) / cw);
	const float4 wd4 = psf.ref_wd[i], wg4 = psf.ref_wg[i];
	const f3 w_d = mk3(wd4.x, wd4.y, wd4.z), w_g = mk3(wg4.x, wg4.y, wg4.z);
	const f3 w = ((comp & COMP_DIFFUSE_MASK) ? w_d : splat3(0.0f)) + ((comp & COMP_GLOSSY_MASK) ? w_g : splat3(0.0f));
	const f3 cvw = cv * w;
	// the three terms psf_blend_kernel adds to the frame, kept in the path's blend cell of this bounce: the merge applies them after the pass's samples
	const uint32_t pidx = sl.k * pass.acc_stride + sl.slot;
	float4* out = log.blend + (size_t(bounce) * log.cap + pidx) * 3;
	const f3 d = cv * w_d, g = cv * w_g;
	out[0] = make_float4(sel_min(cvw.x, psf.firefly), sel_min(cvw.y, psf.firefly), sel_min(cvw.z, psf.firefly), as_f32(comp));
	out[1] = make_float4(d.x, d.y, d.z, 0.0f);
	out[2] = make_float4(g.x, g.y, g.z, 0.0f);
	log_mark(log, pidx, 3u * log.n_bounces + bounce);
}
// passes in flight: fold pass k into the global table (find-or-insert by key; a pass lists each of its cells once, so one thread owns a cell) and
// write the global values back into the pass table -- the cache as it stands after pass k, which is what that pass's blend reads.  Launched
// for k = 0, 1, ... in order on one stream.
__global__ void psf_prefix_kernel(PsfDev psf, uint32_t k)
{
	const PsfDev t = psf_pass_view(psf, k);
	PsfDev g = psf; g.keys = psf.g_keys; g.cells = psf.g_cells; g.log2_size = psf.g_log2_size; g.touched = nullptr;
	const uint32_t n = *t.touched_n;
	for (uint32_t i = threadIdx.x + blockIdx.x * blockDim.x; i < n; i += blockDim.x * gridDim.x)
	{
		const uint32_t slot = t.touched[i];
		const uint32_t gslot = psf_insert(g, t.keys[slot]);
		if (gslot == 0x1FFFFFFFu) continue;          // global table full: the pass keeps its own sums (as an unmerged cell would)
		#pragma unroll
		for (int c = 0; c < 4; ++c)
		{
			const long long v = g.cells[4 * size_t(gslot) + c] + t.cells[4 * size_t(slot) + c];
			g.cells[4 * size_t(gslot) + c] = v;
			t.cells[4 * size_t(slot) + c] = v;
		}
	}
}

// ---- tile-sharded cache: pass table -> records -> global table (fpt_psfpt_set_sharded) ----
__global__ void psf_collect_kernel(PsfDev psf, PsfRecord* __restrict__ out)
{
	const uint32_t n = *psf.touched_n;
	for (uint32_t i = threadIdx.x + blockIdx.x * blockDim.x; i < n; i += blockDim.x * gridDim.x)
	{
		const uint32_t slot = psf.touched[i];
		PsfRecord r; r.key = psf.keys[slot];
		#pragma unroll
		for (int k = 0; k < 4; ++k) r.v[k] = psf.cells[4 * size_t(slot) + k];
		out[i] = r;
	}
}
__global__ void psf_merge_kernel(PsfDev psf, const PsfRecord* __restrict__ records, const uint32_t* __restrict__ d_count, uint32_t count)
{
	const uint32_t n = d_count ? *d_count : count;
	PsfDev g = psf; g.keys = psf.g_keys; g.cells = psf.g_cells; g.log2_size = psf.g_log2_size; g.touched = nullptr;
	for (uint32_t i = threadIdx.x + blockIdx.x * blockDim.x; i < n; i += blockDim.x * gridDim.x)
	{
		const PsfRecord r = records[i];
		const uint32_t slot = psf_insert(g, r.key);
		if (slot == 0x1FFFFFFFu) continue;          // global table full: the cell is dropped, as an uncached vertex would be
		#pragma unroll
		for (int k = 0; k < 4; ++k) if (r.v[k]) atomicAdd(reinterpret_cast<unsigned long long*>(g.cells + 4 * size_t(slot) + k), (unsigned long long)r.v[k]);
	}
}
__global__ void psf_clear_pass_kernel(PsfDev psf)
{
	const uint32_t n = *psf.touched_n;
	for (uint32_t i = threadIdx.x + blockIdx.x * blockDim.x; i < n; i += blockDim.x * gridDim.x)
	{
		const uint32_t slot = psf.touched[i];
		psf.keys[slot] = ~0ull;
		#pragma unroll
		for (int k = 0; k < 4; ++k) psf.cells[4 * size_t(slot) + k] = 0;
	}
}

// clamp_frame_kernel (src/renderer.cu:314-331)
__global__ void clamp_frame_kernel(FrameBufferDev fb, const uint32_t* __restrict__ pixels, uint32_t n, float max_value)
{
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	if (i >= n) return;
	const uint32_t p = pixels ? pixels[i] : i;
	const int ch[4] = { FPT_FB_DIFFUSE_C, FPT_FB_SPECULAR_C, FPT_FB_DIRECT_C, FPT_FB_COMPOSITED_C };
	#pragma unroll
	for (int c = 0; c < 4; ++c)
	{
		const float4 v = fb.ch[ch[c]][p];
		fb.ch[ch[c]][p] = make_float4(sel_min(v.x, max_value), sel_min(v.y, max_value), sel_min(v.z, max_value), sel_min(v.w, max_value));
	}
}

// ShadeRecord: one thread per triangle gathers what a shaded vertex needs of it
__global__ void shade_records_kernel(fpt_mesh_view mesh, ShadeRecord* __restrict__ out)
{
	const uint32_t t = threadIdx.x + blockIdx.x * blockDim.x;
	if (t >= mesh.num_triangles) return;
	const int4 idx = *reinterpret_cast<const int4*>(mesh.vertex_indices + 4 * size_t(t));
	ShadeRecord r;
	r.a = *reinterpret_cast<const float4*>(mesh.vertex_data + 4 * size_t(idx.x));
	r.b = *reinterpret_cast<const float4*>(mesh.vertex_data + 4 * size_t(idx.y));
	r.c = *reinterpret_cast<const float4*>(mesh.vertex_data + 4 * size_t(idx.z));
	int4 tc = make_int4(-1, -1, -1, -1);
	if (mesh.texture_indices_comp) tc = *reinterpret_cast<const int4*>(mesh.texture_indices_comp + 4 * size_t(t));
	r.d = make_float4(as_f32(uint32_t(tc.x)), as_f32(uint32_t(tc.y)), as_f32(uint32_t(tc.z)), as_f32(uint32_t(mesh.material_indices[t])));
	out[t] = r;
}

// EmitterView::vpl_points: one thread per VPL runs emitter_sample's VPL branch and stores the light point
__global__ void vpl_points_kernel(EmitterView em, fpt_mesh_view mesh, const fpt_texture* textures, float4* __restrict__ out)
{
	const uint32_t l = threadIdx.x + blockIdx.x * blockDim.x;
	if (l >= em.n_vpls) return;
	const fpt_vpl vp = em.vpls[l];
	SurfacePoint lp; f3 radiance; float pdf;
	surface_point(mesh, vp.prim_id, vp.uv[0], vp.uv[1], lp);
	emitter_at(em, mesh, textures, vp.prim_id, lp.s, lp.t, radiance, pdf);
	float4* rec = out + VPL_POINT_STRIDE * size_t(l);
	rec[0] = make_float4(lp.position.x, lp.position.y, lp.position.z, pdf);
	rec[1] = make_float4(lp.frame.n.x, lp.frame.n.y, lp.frame.n.z, 0.0f);
	rec[2] = make_float4(radiance.x, radiance.y, radiance.z, 0.0f);
}

__device__ __forceinline__ float max3_xyz(float4 v) { return sel_max(v.x, sel_max(v.y, v.z)); }

__global__ void rescale_kernel(FrameBufferDev fb, const uint32_t* __restrict__ pixels, uint32_t n, float scale)
{
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	if (i >= n) return;
	const uint32_t p = pixels ? pixels[i] : i;
	const float4 dc = fb.ch[FPT_FB_DIRECT_C][p], fc = fb.ch[FPT_FB_DIFFUSE_C][p], sc = fb.ch[FPT_FB_SPECULAR_C][p], cc = fb.ch[FPT_FB_COMPOSITED_C][p];
	fb.ch[FPT_FB_LUMINANCE][p] = make_float4(max3_xyz(dc), max3_xyz(fc), max3_xyz(sc), max3_xyz(cc));
	const float4 fa = fb.ch[FPT_FB_DIFFUSE_A][p], sa = fb.ch[FPT_FB_SPECULAR_A][p];
	fb.ch[FPT_FB_DIFFUSE_C][p]    = make_float4(fc.x * scale, fc.y * scale, fc.z * scale, fc.w * scale);
	fb.ch[FPT_FB_DIFFUSE_A][p]    = make_float4(fa.x * scale, fa.y * scale, fa.z * scale, fa.w * scale);
	fb.ch[FPT_FB_SPECULAR_C][p]   = make_float4(sc.x * scale, sc.y * scale, sc.z * scale, sc.w * scale);
	fb.ch[FPT_FB_SPECULAR_A][p]   = make_float4(sa.x * scale, sa.y * scale, sa.z * scale, sa.w * scale);
	fb.ch[FPT_FB_DIRECT_C][p]     = make_float4(dc.x * scale, dc.y * scale, dc.z * scale, dc.w * scale);
	fb.ch[FPT_FB_COMPOSITED_C][p] = make_float4(cc.x * scale, cc.y * scale, cc.z * scale, cc.w * scale);
}

__global__ void variance_kernel(FrameBufferDev fb, const uint32_t* __restrict__ pixels, uint32_t n_pixels, uint32_t n)
{
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	if (i >= n_pixels) return;
	const uint32_t p = pixels ? pixels[i] : i;
	const float4 old_lum = fb.ch[FPT_FB_LUMINANCE][p];
	float4 dc = fb.ch[FPT_FB_DIRECT_C][p], fc = fb.ch[FPT_FB_DIFFUSE_C][p], sc = fb.ch[FPT_FB_SPECULAR_C][p], cc = fb.ch[FPT_FB_COMPOSITED_C][p];
	const float fn = float(n), fn1 = float(n - 1), fnn = float(n * n);
	const float d0 = max3_xyz(dc) - old_lum.x, d1 = max3_xyz(fc) - old_lum.y, d2 = max3_xyz(sc) - old_lum.z, d3 = max3_xyz(cc) - old_lum.w;
	dc.w += ((fn * d0) * (fn1 * d0)) / fnn;
	fc.w += ((fn * d1) * (fn1 * d1)) / fnn;
	sc.w += ((fn * d2) * (fn1 * d2)) / fnn;
	cc.w += ((fn * d3) * (fn1 * d3)) / fnn;
	fb.ch[FPT_FB_DIRECT_C][p] = dc; fb.ch[FPT_FB_DIFFUSE_C][p] = fc; fb.ch[FPT_FB_SPECULAR_C][p] = sc; fb.ch[FPT_FB_COMPOSITED_C][p] = cc;
}

// The path tracer's passes in flight: apply the passes base..base+n-1 to the frame buffer IN ORDER from the batch's contribution log (fpt_device.h
// ContribLog) and the two albedo planes, doing per pass exactly what rescale_kernel -> add_in per sample -> variance_kernel do on the frame
// (src/renderer.cu:292-312,333-362, src/framebuffer.h:425-444): the result equals n sequential render() calls bit for bit, .w terms included.
struct RegisterAdd
{
	float4* c; float w;
	__device__ __forceinline__ void operator()(int ch, bool variance, f3 f) const { if (variance) mean_add<true>(c[ch], f, w); else mean_add<false>(c[ch], f, w); }
};
template <bool PSF>
__global__ void merge_passes_exact_kernel(FrameBufferDev fb, float4* __restrict__ albedo_d, float4* __restrict__ albedo_s, ContribLog log, const uint32_t* __restrict__ pixels,
                                          uint32_t n_pixels, PassInfo ps, float firefly, float clamp_max)
{
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	if (i >= n_pixels) return;
	const uint32_t p = pixels ? pixels[i] : i;
	float4 c[6];
	#pragma unroll
	for (int ch = 0; ch < 6; ++ch) c[ch] = fb.ch[ch][p];
	float4 lum = fb.ch[FPT_FB_LUMINANCE][p];
	for (uint32_t k = 0; k < ps.n_passes; ++k)
	{
		const uint32_t inst = ps.base_instance + k;
		const float scale = float(inst) / float(inst + 1);
		const float w = 1.0f / float(inst + 1);
		const uint32_t pidx = k * ps.acc_stride + i;
		// rescale_kernel
		lum = make_float4(max3_xyz(c[FPT_FB_DIRECT_C]), max3_xyz(c[FPT_FB_DIFFUSE_C]), max3_xyz(c[FPT_FB_SPECULAR_C]), max3_xyz(c[FPT_FB_COMPOSITED_C]));
		#pragma unroll
		for (int ch = 0; ch < 6; ++ch) c[ch] = make_float4(c[ch].x * scale, c[ch].y * scale, c[ch].z * scale, c[ch].w * scale);
		// the surface albedos of the primary vertex: one term per pass, summed into a zeroed plane (0 + a = a)
		{
			const float4 a = albedo_d[pidx], b = albedo_s[pidx];
			albedo_d[pidx] = make_float4(0.0f, 0.0f, 0.0f, 0.0f); albedo_s[pidx] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
			c[FPT_FB_DIFFUSE_A].x += a.x; c[FPT_FB_DIFFUSE_A].y += a.y; c[FPT_FB_DIFFUSE_A].z += a.z; c[FPT_FB_DIFFUSE_A].w += a.w;
			c[FPT_FB_SPECULAR_A].x += b.x; c[FPT_FB_SPECULAR_A].y += b.y; c[FPT_FB_SPECULAR_A].z += b.z; c[FPT_FB_SPECULAR_A].w += b.w;
		}
		// the path's samples in the order the launches of a pass deliver them: bounce by bounce, emission, directional light, mesh light
		const RegisterAdd add{ c, w };
		for (uint32_t word = 0; word < log.mask_words; ++word)
		{
			uint32_t* mp = log.mask + size_t(pidx) * log.mask_words + word;
			uint32_t m = *mp;
			if (!m) continue;
			*mp = 0u;
			while (m)
			{
				const uint32_t bit = uint32_t(__builtin_ctz(m)); m &= m - 1u;
				const uint32_t j = word * 32u + bit;
				if (PSF && j >= 3u * log.n_bounces)
				{
					// a blend of the PSFPT: after every bounce's samples, bounce by bounce
					const float4* cell = log.blend + (size_t(j - 3u * log.n_bounces) * log.cap + pidx) * 3;
					const float4 c0 = cell[0], c1 = cell[1], c2 = cell[2];
					apply_psf_blend(add, as_u32(c0.w), mk3(c0.x, c0.y, c0.z), mk3(c1.x, c1.y, c1.z), mk3(c2.x, c2.y, c2.z));
					continue;
				}
				const uint32_t bounce = j / 3u, kind = j - 3u * bounce;
				if (kind == 0u)
				{
					const float4 e = log.emissive[size_t(bounce) * log.cap + pidx];
					apply_emissive(add, bounce, as_u32(e.w), mk3(e.x, e.y, e.z));
				}
				else
				{
					const float4* cell = log.nee[kind - 1u] + (size_t(bounce) * log.cap + pidx) * 2;
					const float4 wd = cell[0], wg = cell[1];
					const uint32_t tag = as_u32(wd.w);
					if (PSF) apply_psf_nee(add, bounce, tag & 0xFu, (tag & 0x100u) != 0u, (tag & 0x200u) != 0u, mk3(wd.x, wd.y, wd.z), mk3(wg.x, wg.y, wg.z), firefly);
					else     apply_nee(add, bounce, tag, mk3(wd.x, wd.y, wd.z), mk3(wg.x, wg.y, wg.z));
				}
			}
		}
		// variance_kernel
		const uint32_t n = inst + 1;
		const float fn = float(n), fn1 = float(n - 1), fnn = float(n * n);
		const float d0 = max3_xyz(c[FPT_FB_DIRECT_C]) - lum.x, d1 = max3_xyz(c[FPT_FB_DIFFUSE_C]) - lum.y, d2 = max3_xyz(c[FPT_FB_SPECULAR_C]) - lum.z, d3 = max3_xyz(c[FPT_FB_COMPOSITED_C]) - lum.w;
		c[FPT_FB_DIRECT_C].w     += ((fn * d0) * (fn1 * d0)) / fnn;
		c[FPT_FB_DIFFUSE_C].w    += ((fn * d1) * (fn1 * d1)) / fnn;
		c[FPT_FB_SPECULAR_C].w   += ((fn * d2) * (fn1 * d2)) / fnn;
		c[FPT_FB_COMPOSITED_C].w += ((fn * d3) * (fn1 * d3)) / fnn;
		if (PSF && clamp_max > 0.0f)
		{
			// clamp_frame after every pass (PSFPT::render, src/renderers/psfpt_impl.h:275-284; clamp_frame_kernel): all four components
			const int cl[4] = { FPT_FB_DIFFUSE_C, FPT_FB_SPECULAR_C, FPT_FB_DIRECT_C, FPT_FB_COMPOSITED_C };
			#pragma unroll
			for (int q = 0; q < 4; ++q)
			{
				float4& v = c[cl[q]];
				v = make_float4(sel_min(v.x, clamp_max), sel_min(v.y, clamp_max), sel_min(v.z, clamp_max), sel_min(v.w, clamp_max));
			}
		}
	}
	#pragma unroll
	for (int ch = 0; ch < 6; ++ch) fb.ch[ch][p] = c[ch];
	fb.ch[FPT_FB_LUMINANCE][p] = lum;
}

// tile-owned pixels <-> one contiguous message (fpt_gather_framebuffer): 16-byte accesses, the contiguous side coalesced
__global__ void pack_pixels_kernel(const float4* __restrict__ channel, const uint32_t* __restrict__ pixels, uint32_t n, float4* __restrict__ dst)
{
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	if (i < n) dst[i] = channel[pixels[i]];
}
__global__ void unpack_pixels_kernel(const float4* __restrict__ src, const uint32_t* __restrict__ pixels, uint32_t n, float4* __restrict__ channel)
{
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	if (i < n) channel[pixels[i]] = src[i];
}

__global__ void rgba_kernel(const float4* __restrict__ composited, uint32_t n, float exposure, float inv_gamma, uint32_t* __restrict__ rgba)
{
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	if (i >= n) return;
	const float4 c = composited[i];
	const float v[4] = { c.x * exposure, c.y * exposure, c.z * exposure, c.w * exposure };
	uint32_t packed = 0;
	#pragma unroll
	for (int k = 0; k < 4; ++k)
	{
		const float m = v[k] / (v[k] + 1.0f);
		const float g = det_pow(m, inv_gamma);
		packed |= (to_u32_sat(ieee_min(g * 256.0f, 255.0f)) & 0xffu) << (8 * k);
	}
	rgba[i] = packed;
}

__global__ void debug_math_kernel(int op, uint32_t n, const float* in0, const float* in1, float* out0, float* out1)
{
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	if (i >= n) return;
	if (op == 0) { float s, c; det_sincos(in0[i], s, c); out0[i] = s; out1[i] = c; }
	else if (op == 1) out0[i] = det_atan2(in0[i], in1[i]);
	else if (op == 2) out0[i] = det_pow(in0[i], in1[i]);
	else if (op == 3) out0[i] = round_through_half(in0[i]);
	else if (op == 4) { const f3 h = cosine_hemisphere(in0[i], in1[i]); out0[i] = h.x; out1[i] = h.z; }
}

// ---- launchers ----------------------------------------------------------------------------------------------------------
static inline uint32_t blocks_for(uint32_t n, uint32_t b) { return n ? (n + b - 1) / b : 1; }

void launch_sequence(uint32_t n_dims, uint32_t tile2, uint32_t instance, const float* shifts, float* samples, hipStream_t s)
{ hipLaunchKernelGGL(sequence_kernel, dim3(blocks_for(tile2, 256)), dim3(256), 0, s, n_dims, tile2, instance, shifts, samples); }
void launch_shade_records(const fpt_mesh_view& mesh, ShadeRecord* out, hipStream_t s)
{ hipLaunchKernelGGL(shade_records_kernel, dim3(blocks_for(mesh.num_triangles, 256)), dim3(256), 0, s, mesh, out); }
void launch_vpl_points(const EmitterView& em, const fpt_mesh_view& mesh, const fpt_texture* textures, float4* out, hipStream_t s)
{ hipLaunchKernelGGL(vpl_points_kernel, dim3(blocks_for(em.n_vpls, 256)), dim3(256), 0, s, em, mesh, textures, out); }
void launch_primary_rays(const PrimaryParams& p, hipStream_t s)
{ hipLaunchKernelGGL(primary_rays_kernel, dim3(blocks_for(p.n_pixels * p.pass.n_passes, 256)), dim3(256), 0, s, p); }
void launch_shade(const ShadeParams& p, uint32_t max_entries, hipStream_t s)
{ hipLaunchKernelGGL((shade_kernel<false>), dim3(blocks_for(max_entries, SHADE_BLOCK)), dim3(SHADE_BLOCK), 0, s, p); }
void launch_shade_psf(const ShadeParams& p, uint32_t max_entries, hipStream_t s)
{ hipLaunchKernelGGL((shade_kernel<true>), dim3(blocks_for(max_entries, SHADE_BLOCK)), dim3(SHADE_BLOCK), 0, s, p); }
void launch_psf_resolve(const ResolveParams& p, uint32_t max_entries, hipStream_t s)
{ hipLaunchKernelGGL(psf_resolve_kernel, dim3(blocks_for(max_entries, 256)), dim3(256), 0, s, p); }
void launch_psf_blend(const PsfDev& psf, const FrameBufferDev& fb, float frame_weight, uint32_t max_refs, hipStream_t s)
{ hipLaunchKernelGGL(psf_blend_kernel, dim3(blocks_for(max_refs, 256)), dim3(256), 0, s, psf, fb, frame_weight); }
void launch_psf_blend_batch(const PsfDev& psf, const ContribLog& log, uint32_t bounce, const PassInfo& pass, uint32_t max_refs, hipStream_t s)
{ hipLaunchKernelGGL(psf_blend_batch_kernel, dim3(blocks_for(max_refs, 256)), dim3(256), 0, s, psf, log, bounce, pass); }
void launch_psf_prefix(const PsfDev& psf, uint32_t k, hipStream_t s) { hipLaunchKernelGGL(psf_prefix_kernel, dim3(256), dim3(256), 0, s, psf, k); }
void launch_psf_collect(const PsfDev& psf, PsfRecord* out, hipStream_t s) { hipLaunchKernelGGL(psf_collect_kernel, dim3(256), dim3(256), 0, s, psf, out); }
void launch_psf_merge(const PsfDev& psf, const PsfRecord* records, const uint32_t* d_count, uint32_t count, hipStream_t s)
{ hipLaunchKernelGGL(psf_merge_kernel, dim3(256), dim3(256), 0, s, psf, records, d_count, count); }
void launch_psf_clear_pass(const PsfDev& psf, hipStream_t s) { hipLaunchKernelGGL(psf_clear_pass_kernel, dim3(256), dim3(256), 0, s, psf); }
void launch_clamp_frame(const FrameBufferDev& fb, const uint32_t* pixels, uint32_t n, float max_value, hipStream_t s)
{ hipLaunchKernelGGL(clamp_frame_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s, fb, pixels, n, max_value); }
void launch_rescale(const FrameBufferDev& fb, const uint32_t* pixels, uint32_t n, float scale, hipStream_t s)
{ hipLaunchKernelGGL(rescale_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s, fb, pixels, n, scale); }
void launch_variance(const FrameBufferDev& fb, const uint32_t* pixels, uint32_t n_pixels, uint32_t n, hipStream_t s)
{ hipLaunchKernelGGL(variance_kernel, dim3(blocks_for(n_pixels, 256)), dim3(256), 0, s, fb, pixels, n_pixels, n); }
void launch_pack_pixels(const float4* channel, const uint32_t* pixels, uint32_t n, float4* dst, hipStream_t s)
{ hipLaunchKernelGGL(pack_pixels_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s, channel, pixels, n, dst); }
void launch_unpack_pixels(const float4* src, const uint32_t* pixels, uint32_t n, float4* channel, hipStream_t s)
{ hipLaunchKernelGGL(unpack_pixels_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s, src, pixels, n, channel); }
void launch_merge_passes_exact(const FrameBufferDev& fb, float4* albedo_d, float4* albedo_s, const ContribLog& log, const uint32_t* pixels, uint32_t n_pixels, PassInfo pass, hipStream_t s,
                               bool psf, float firefly, float clamp_max)
{
	if (psf) hipLaunchKernelGGL((merge_passes_exact_kernel<true>), dim3(blocks_for(n_pixels, 256)), dim3(256), 0, s, fb, albedo_d, albedo_s, log, pixels, n_pixels, pass, firefly, clamp_max);
	else     hipLaunchKernelGGL((merge_passes_exact_kernel<false>), dim3(blocks_for(n_pixels, 256)), dim3(256), 0, s, fb, albedo_d, albedo_s, log, pixels, n_pixels, pass, firefly, clamp_max);
}
void launch_rgba(const float4* composited, uint32_t n, float exposure, float inv_gamma, uint32_t* rgba, hipStream_t s)
{ hipLaunchKernelGGL(rgba_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s, composited, n, exposure, inv_gamma, rgba); }
void launch_debug_math(int op, uint32_t n, const float* a, const float* b, float* o0, float* o1, hipStream_t s)
{ hipLaunchKernelGGL(debug_math_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s, op, n, a, b, o0, o1); }

} // namespace fpt
