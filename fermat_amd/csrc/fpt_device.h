// fpt_device.h — device-side views and queue layouts shared by the kernel translation units.
//
// Queues are structure-of-arrays like PTRayQueue (src/pathtracer_queues.h:44-93) but carry only the fields the plain
// PT vertex processor consumes (its vertex_info / nee_slot / nee_cluster words are always 0xFFFFFFFF,
// src/pathtracer_vertex_processor.h:72,104,137): 76 B per path entry instead of 88 B, 84 B per shadow entry
// instead of 112 B.  Counters live in device memory and are never read back by the host inside a pass.
#pragma once
#include "fpt_shading.h"

namespace fpt {

// Round 5: the words a queue ray does not need carry the entry's bookkeeping, so that an entry is 16-byte vectors only (VERDICT r4 task 5).  A path-queue ray's tmin /
// tmax are the same for every entry of a queue -- (0, 1e34) for the primary rays, (1e-3, 1e8) for scattered ones (src/pathtracer_kernels.h:173-176,
// src/pathtracer_core.h:1236-1241) -- and a shadow-queue ray's tmax is 0.9999 (:1069): the traversal kernel takes them from its mode (fpt_trace.hip MODE_*_Q,
// MODE_MIXED*, MODE_ANY_FUSED) instead of from the ray, and the two .w words of a path entry hold PixelInfo and the pass offset, those of a shadow entry the shadow
// mask and PixelInfo, with the pass offset in w_d.w: 64 B per path entry (+ 8 with the cone plane), 64 B per shadow entry, against 80 / 72 until round 4.
struct PathQueue
{
	float4*   rays;        // 2 x float4 per entry : origin | PixelInfo (pixel:27 | comp:4 | diffuse:1, src/pathtracer_core.h:527-542) , dir | pass offset k (passes in flight; 0 otherwise)
	float4*   hits;        // t, triId bits, u, v
	float4*   weights;     // path throughput rgb, .w = solid-angle pdf of the last scattering event
	float2*   cones;       // ray-cone radius, pdf
	uint32_t* size;
	uint32_t* vinfo;       // what the vertex processor returned at the previous vertex (PSFPT only; NULL for the plain PT)
};
struct ShadowQueue
{
	float4*   rays;        // origin | shadow mask , dir | PixelInfo
	float4*   w_d;         // diffuse-channel weight, .w = pass offset k
	float4*   w_g;         // glossy-channel weight
	uint32_t* size;
	uint32_t* vinfo;       // PSFPT only
};
static constexpr float QUEUE_PRIMARY_TMIN = 0.0f, QUEUE_PRIMARY_TMAX = 1.0e34f, QUEUE_SCATTER_TMIN = 1.0e-3f, QUEUE_SCATTER_TMAX = 1.0e8f, QUEUE_SHADOW_TMAX = 0.9999f;

struct FrameBufferDev
{
	float4*   ch[FPT_FB_NUM_CHANNELS];
	float4*   gb_geo; float4* gb_uv; uint32_t* gb_tri; float* gb_depth;
};

struct BvhDev { const uint4* nodes; const float4* tris; };     // 80-byte 8-wide compressed nodes (fpt_bvh.h BvhNode8), 48-byte triangle records

// Which progressive passes a launch covers.  n_passes == 1 is the reference's one-pass-per-render() behaviour: samples are
// accumulated straight into the frame buffer with Fermat's own arithmetic.  n_passes > 1 is the batched ("passes in flight")
// mode: a path's 27-bit PixelInfo.pixel field carries its `slot` (the path's index in this rank's pixel list, = the pixel when the whole frame is
// rendered here) and the queue entry's pass_k word the pass offset k (round 4: until then the field held k * n_slot + slot, which capped a 4K frame at
// 16 passes in flight); samples go to the path's cells  k * acc_stride + slot  of the contribution log / the per-pass albedo planes and a merge kernel
// applies the passes in order.  The limit is now 2^32 paths in flight -- memory, not the word.
struct PassInfo { uint32_t base_instance, n_passes, n_slot, acc_stride; const uint32_t* pixels; };
struct PathSlot { uint32_t pixel, k; float weight; uint32_t slot; };
__device__ __forceinline__ PathSlot decode_slot(const PassInfo& ps, uint32_t pixel_info, uint32_t pass_k)
{
	PathSlot r;
	const uint32_t v = pixel_info & 0x7FFFFFFu;
	if (ps.n_passes == 1) { r.k = 0; r.pixel = v; r.slot = v; }
	else { r.k = pass_k; r.slot = v; r.pixel = ps.pixels ? ps.pixels[v] : v; }
	r.weight = 1.0f / float(ps.base_instance + r.k + 1);          // frame_weight (src/renderers/pathtracer_impl.h:281)
	return r;
}

// progressive-mean accumulation with optional Welford-style luminance variance in .w (src/framebuffer.h:425-444), on a value in registers
template <bool VARIANCE>
__device__ __forceinline__ void mean_add(float4& mean, f3 f, float inv_n)
{
	const f3 delta = f - mk3(mean.x, mean.y, mean.z);
	mean.x += f.x * inv_n;
	mean.y += f.y * inv_n;
	mean.z += f.z * inv_n;
	if (VARIANCE)
	{
		const float ld = max_comp(delta);
		mean.w += ld * ld * inv_n;
	}
}
template <bool VARIANCE>
__device__ __forceinline__ void fb_add(float4* channel, uint32_t pixel, f3 f, float inv_n)
{
	float4 mean = channel[pixel];
	mean_add<VARIANCE>(mean, f, inv_n);
	channel[pixel] = mean;
}
// The path tracer's passes in flight (fpt_pt_render_batch) keep every frame-buffer contribution of a path APART: a path gives the frame at most one
// emission sample, one directional-light sample and one mesh-light sample per bounce, so each (pass, pixel slot, bounce, kind) owns a fixed cell, a
// bit per cell says which are filled, and merge_passes_exact_kernel applies them pass by pass in the order n sequential render() calls would have --
// rescale, samples by (bounce, kind), variances -- with Fermat's own add_in arithmetic.  The frame is bit-identical to the sequential one, .w included.
//   path index   pidx = k * acc_stride + slot       (k = pass offset in the batch; the planes' indexing)
//   emissive     [bounce * cap + pidx]               xyz = the sample, w = the PixelInfo comp bits
//   nee[kind]    [(bounce * cap + pidx) * 2 + {0,1}] w_d (w = comp bits), w_g;  kind 0 = directional light, 1 = mesh light / VPL
//   mask         [pidx * mask_words + (bit >> 5)]    bit = 3 * bounce + {0 emissive, 1 directional, 2 mesh}
// The PSFPT's passes in flight use the same log with one more kind: the blend of a pixel's cache references, which a pass applies bounce by bounce AFTER
// all of its bounces' samples:
//   blend        [(bounce * cap + pidx) * 3 + {0,1,2}] the clamped COMPOSITED term, cell x w_d, cell x w_g (w of the first = comp bits);  bit = 3 * n_bounces + bounce
struct ContribLog { float4* emissive; float4* nee[2]; float4* blend; uint32_t* mask; uint32_t cap, mask_words, n_bounces; };
__device__ __forceinline__ void log_mark(const ContribLog& g, uint32_t pidx, uint32_t bit)
{
	uint32_t* m = g.mask + size_t(pidx) * g.mask_words + (bit >> 5);      // the word belongs to this path alone, and a path has one writer per launch
	*m |= 1u << (bit & 31u);
}

// PTVertexProcessor::accumulate_emissive (src/pathtracer_vertex_processor.h:151-183) on registers / on the frame
template <typename ADD>
__device__ __forceinline__ void apply_emissive(ADD&& add, uint32_t bounce, uint32_t comp, f3 e)
{
	add(FPT_FB_COMPOSITED_C, false, e);
	if (bounce == 0) add(FPT_FB_DIRECT_C, false, e);
	else
	{
		if (comp & COMP_DIFFUSE_MASK) add(FPT_FB_DIFFUSE_C, true, e);
		if (comp & COMP_GLOSSY_MASK)  add(FPT_FB_SPECULAR_C, true, e);
	}
}
// PTVertexProcessor::accumulate_nee (src/pathtracer_vertex_processor.h:202-239) for an UNOCCLUDED sample
template <typename ADD>
__device__ __forceinline__ void apply_nee(ADD&& add, uint32_t bounce, uint32_t comp, f3 w_d, f3 w_g)
{
	add(FPT_FB_COMPOSITED_C, false, w_d + w_g);
	if (bounce == 0)
	{
		add(FPT_FB_DIFFUSE_C, true, w_d);
		add(FPT_FB_SPECULAR_C, true, w_g);
	}
	else
	{
		if (comp & COMP_DIFFUSE_MASK) add(FPT_FB_DIFFUSE_C, true, w_d);
		if (comp & COMP_GLOSSY_MASK)  add(FPT_FB_SPECULAR_C, true, w_g);
	}
}
// PSFPTVertexProcessor::accumulate_nee, the part that goes to the FRAME (src/psfpt_vertex_processor.h:345-441): `cached` = the sample belongs to a valid cache
// cell (its diffuse part -- or all of it -- went to the cell), `diffuse_only` = only the diffuse part did; every term through the firefly clamp
__device__ __forceinline__ f3 firefly_clamp(float firefly, f3 v) { return all_finite(v) ? mk3(sel_min(v.x, firefly), sel_min(v.y, firefly), sel_min(v.z, firefly)) : splat3(0.0f); }
template <typename ADD>
__device__ __forceinline__ void apply_psf_nee(ADD&& add, uint32_t bounce, uint32_t comp, bool cached, bool diffuse_only, f3 w_d, f3 w_g, float firefly)
{
	if (cached)
	{
		if (diffuse_only)
		{
			add(FPT_FB_COMPOSITED_C, false, firefly_clamp(firefly, w_g));
			add((bounce == 0 || (comp & COMP_GLOSSY_MASK)) ? FPT_FB_SPECULAR_C : FPT_FB_DIFFUSE_C, true, firefly_clamp(firefly, w_g));
		}
		return;
	}
	add(FPT_FB_COMPOSITED_C, false, firefly_clamp(firefly, w_d + w_g));
	if (bounce == 0)
	{
		add(FPT_FB_DIFFUSE_C, true, firefly_clamp(firefly, w_d));
		add(FPT_FB_SPECULAR_C, true, firefly_clamp(firefly, w_g));
	}
	else
	{
		if (comp & COMP_DIFFUSE_MASK) add(FPT_FB_DIFFUSE_C, true, firefly_clamp(firefly, w_d + w_g));
		if (comp & COMP_GLOSSY_MASK)  add(FPT_FB_SPECULAR_C, true, firefly_clamp(firefly, w_d + w_g));
	}
}
// psf_blending_kernel's three terms (src/renderers/psfpt_impl.h)
template <typename ADD>
__device__ __forceinline__ void apply_psf_blend(ADD&& add, uint32_t comp, f3 composited, f3 diffuse, f3 glossy)
{
	add(FPT_FB_COMPOSITED_C, false, composited);
	if (comp & COMP_DIFFUSE_MASK) add(FPT_FB_DIFFUSE_C, true, diffuse);
	if (comp & COMP_GLOSSY_MASK)  add(FPT_FB_SPECULAR_C, true, glossy);
}
struct FrameAdd      // add_in on the frame buffer (one pass per render())
{
	const FrameBufferDev& fb; uint32_t pixel; float w;
	__device__ __forceinline__ void operator()(int c, bool variance, f3 f) const { if (variance) fb_add<true>(fb.ch[c], pixel, f, w); else fb_add<false>(fb.ch[c], pixel, f, w); }
};
__device__ __forceinline__ void accumulate_emissive(const FrameBufferDev& fb, const PassInfo& ps, const ContribLog& log, const PathSlot& sl, uint32_t pixel_info, uint32_t bounce, f3 e)
{
	const uint32_t comp = (pixel_info >> 27) & 0xFu;
	if (ps.n_passes == 1) { apply_emissive(FrameAdd{ fb, sl.pixel, sl.weight }, bounce, comp, e); return; }
	const uint32_t pidx = sl.k * ps.acc_stride + sl.slot;
	log.emissive[size_t(bounce) * log.cap + pidx] = make_float4(e.x, e.y, e.z, as_f32(comp));
	log_mark(log, pidx, 3u * bounce);
}
// kind: 0 = directional light, 1 = mesh light / VPL
__device__ __forceinline__ void accumulate_nee(const FrameBufferDev& fb, const PassInfo& ps, const ContribLog& log, uint32_t kind, uint32_t pixel_info, uint32_t pass_k, uint32_t bounce, f3 w_d, f3 w_g)
{
	const PathSlot sl = decode_slot(ps, pixel_info, pass_k);
	const uint32_t comp = (pixel_info >> 27) & 0xFu;
	if (ps.n_passes == 1) { apply_nee(FrameAdd{ fb, sl.pixel, sl.weight }, bounce, comp, w_d, w_g); return; }
	const uint32_t pidx = sl.k * ps.acc_stride + sl.slot;
	float4* cell = log.nee[kind] + (size_t(bounce) * log.cap + pidx) * 2;
	cell[0] = make_float4(w_d.x, w_d.y, w_d.z, as_f32(comp));
	cell[1] = make_float4(w_g.x, w_g.y, w_g.z, 0.0f);
	log_mark(log, pidx, 3u * bounce + 1u + kind);
}

// ---- launch parameter blocks --------------------------------------------------------------------------------------------
struct TraceParams
{
	BvhDev          bvh;
	uint32_t        n_nodes;
	const float4*   rays;
	float4*         hits;          // closest / any-hit result (may be NULL for the fused shadow pass)
	uint32_t*       bits;          // 1 bit per ray (trace_shadow_bits) or NULL
	const uint32_t* count_ptr;     // device-resident queue size, or NULL to use `count`
	uint32_t        count;
	uint32_t*       work_counter;  // persistent-wave ticket dispenser (zeroed before the launch)
	unsigned long long* stats;     // instrumented variant: closest rays -> [0] nodes popped [1] triangles tested [2] rays; any-hit rays -> [4] [5] [6]
	// fused solve_occlusion (src/pathtracer_kernels.h:248-280): accumulate the NEE sample when unoccluded.  Only the ray array and the
	// queue size travel as kernel arguments; what the (rare) retirement of an unoccluded sample needs sits behind one pointer, so that it
	// does not occupy ~40 SGPRs for the whole traversal (measured: 20 SGPR + 12 VGPR spills in the MIXED kernel otherwise)
	const float4*   shadow_rays;
	const uint32_t* shadow_size;
	const struct FusedResolve* fused;
	uint32_t        base_instance; // first pass of the call: overrides fused->pass.base_instance, so that the blocks behind `fused` do not change from call to call
};
struct FusedResolve { const float4* w_d; const float4* w_g; FrameBufferDev fb; PassInfo pass; uint32_t bounce; ContribLog log; uint32_t kind; };      // (PixelInfo and the pass offset ride in the shadow entry: ShadowQueue)

uint32_t trace_blocks_per_cu();
uint32_t trace_stack_entries();      // capacity of the traversal stack (LDS + scratch levels); fpt_rt_create_geometry checks the tree's bound against it
void launch_trace_closest(const TraceParams& p, bool counted, uint32_t n_blocks, hipStream_t stream);                         // the RT boundary's rays: tmin / tmax in the .w words
void launch_trace_shadow(const TraceParams& p, bool fused_resolve, bool counted, uint32_t n_blocks, hipStream_t stream);      // fused: a renderer's shadow queue, resolved as the rays retire
void launch_trace_closest_queue(const TraceParams& p, bool primary, bool counted, uint32_t n_blocks, hipStream_t stream);     // a renderer's path queue (PathQueue): primary or scattered rays
void launch_trace_shadow_queue(const TraceParams& p, bool counted, uint32_t n_blocks, hipStream_t stream);                    // a renderer's shadow queue (ShadowQueue), results written to p.hits
void launch_trace_mixed(const TraceParams& p, bool counted, uint32_t n_blocks, hipStream_t stream);
void launch_trace_mixed_psf(const TraceParams& p, bool counted, uint32_t n_blocks, hipStream_t stream);   // p.fused = a ResolveParams block (fpt_kernels.h)
void launch_trace_mixed_hits(const TraceParams& p, float4* shadow_hits, bool counted, uint32_t n_blocks, hipStream_t stream);   // closest-hit rays -> p.hits, the any-hit rays of p.shadow_rays -> shadow_hits (written, not resolved)
// fpt_build.hip: the device-side refit of the 8-wide tree (fpt_rt_refit_geometry); d_scan = {bits of |scene|max, error bits}, boxes = 6 floats each
struct BvhNode8; struct BvhTriangle;
void launch_refit_scan(uint32_t n_tris, const int32_t* d_idx, uint32_t n_verts, const float* d_vtx, uint32_t n_records, const BvhTriangle* d_records, uint32_t* d_scan, hipStream_t s);
void launch_refit_records(uint32_t n_records, BvhTriangle* d_records, const int32_t* d_idx, const float* d_vtx, const uint32_t* d_scan, void* d_tri_box, hipStream_t s);
void launch_refit_level(BvhNode8* d_nodes, void* d_node_box, const void* d_tri_box, uint32_t begin, uint32_t count, uint32_t* d_scan, hipStream_t s);

} // namespace fpt
