// fpt_device.h — device-side views and queue layouts shared by the kernel translation units.
//
// Queues are structure-of-arrays like PTRayQueue (src/pathtracer_queues.h:44-93) but carry only the fields the plain
// PT vertex processor consumes (its vertex_info / nee_slot / nee_cluster words are always 0xFFFFFFFF,
// src/pathtracer_vertex_processor.h:72,104,137): 76 B per path entry instead of 88 B, 84 B per shadow entry
// instead of 112 B.  Counters live in device memory and are never read back by the host inside a pass.
#pragma once
#include "fpt_shading.h"

namespace fpt {

struct PathQueue
{
	float4*   rays;        // 2 x float4 per entry : origin|mask/tmin , dir|tmax
	float4*   hits;        // t, triId bits, u, v
	float4*   weights;     // path throughput rgb, .w = solid-angle pdf of the last scattering event
	uint32_t* pixels;      // PixelInfo : pixel:27 | comp:4 | diffuse:1   (src/pathtracer_core.h:527-542)
	float2*   cones;       // ray-cone radius, pdf
	uint32_t* size;
	uint32_t* vinfo;       // what the vertex processor returned at the previous vertex (PSFPT only; NULL for the plain PT)
};
struct ShadowQueue
{
	float4*   rays;
	float4*   w_d;         // diffuse-channel weight
	float4*   w_g;         // glossy-channel weight
	uint32_t* pixels;
	uint32_t* size;
	uint32_t* vinfo;       // PSFPT only
};

struct FrameBufferDev
{
	float4*   ch[FPT_FB_NUM_CHANNELS];
	float4*   gb_geo; float4* gb_uv; uint32_t* gb_tri; float* gb_depth;
};

struct BvhDev { const uint4* nodes; const float4* tris; };     // 80-byte 8-wide compressed nodes (fpt_bvh.h BvhNode8), 48-byte triangle records

// Which progressive passes a launch covers.  n_passes == 1 is the reference's one-pass-per-render() behaviour: samples are
// accumulated straight into the frame buffer with Fermat's own arithmetic.  n_passes > 1 is the batched ("passes in flight")
// mode: a path's 27-bit PixelInfo.pixel field carries  k * n_slot + slot  (k = pass offset, slot = the path's index in this rank's
// pixel list, = the pixel when the whole frame is rendered here), samples are summed into per-pass accumulation planes
// acc[channel][k * acc_stride + slot]  and a merge kernel applies the passes in order.  Counting in slots rather than absolute
// pixels lets a rank that owns 1/N of the frame keep N times as many passes in flight within the 27 bits, with planes N times smaller.
struct PassInfo { uint32_t base_instance, n_passes, n_slot, acc_stride; const uint32_t* pixels; };
struct PathSlot { uint32_t pixel, k; float weight; uint32_t slot; };
__device__ __forceinline__ PathSlot decode_slot(const PassInfo& ps, uint32_t pixel_info)
{
	PathSlot r;
	const uint32_t v = pixel_info & 0x7FFFFFFu;
	if (ps.n_passes == 1) { r.k = 0; r.pixel = v; r.slot = v; }
	else { r.k = v / ps.n_slot; r.slot = v - r.k * ps.n_slot; r.pixel = ps.pixels ? ps.pixels[r.slot] : r.slot; }
	r.weight = 1.0f / float(ps.base_instance + r.k + 1);          // frame_weight (src/renderers/pathtracer_impl.h:281)
	return r;
}

// progressive-mean accumulation with optional Welford-style luminance variance in .w (src/framebuffer.h:425-444)
template <bool VARIANCE>
__device__ __forceinline__ void fb_add(float4* channel, uint32_t pixel, f3 f, float inv_n)
{
	float4 mean = channel[pixel];
	const f3 delta = f - mk3(mean.x, mean.y, mean.z);
	mean.x += f.x * inv_n;
	mean.y += f.y * inv_n;
	mean.z += f.z * inv_n;
	if (VARIANCE)
	{
		const float ld = max_comp(delta);
		mean.w += ld * ld * inv_n;
	}
	channel[pixel] = mean;
}
// one sample into channel `c`: exact mode = Fermat's add_in on the frame buffer; batched mode = plain weighted sum into the pass plane
template <bool VARIANCE>
__device__ __forceinline__ void splat(const FrameBufferDev& fb, const PassInfo& ps, const PathSlot& sl, int c, f3 f)
{
	if (ps.n_passes == 1) fb_add<VARIANCE>(fb.ch[c], sl.pixel, f, sl.weight);
	else
	{
		float4* cell = fb.ch[c] + size_t(sl.k) * ps.acc_stride + sl.slot;
		float4 a = *cell;
		a.x += f.x * sl.weight; a.y += f.y * sl.weight; a.z += f.z * sl.weight;
		*cell = a;
	}
}

// PTVertexProcessor::accumulate_nee (src/pathtracer_vertex_processor.h:202-239) for an UNOCCLUDED sample
__device__ __forceinline__ void accumulate_nee(const FrameBufferDev& fb, const PassInfo& ps, uint32_t pixel_info, uint32_t bounce, f3 w_d, f3 w_g)
{
	const PathSlot sl = decode_slot(ps, pixel_info);
	const uint32_t comp = (pixel_info >> 27) & 0xFu;
	splat<false>(fb, ps, sl, FPT_FB_COMPOSITED_C, w_d + w_g);
	if (bounce == 0)
	{
		splat<true>(fb, ps, sl, FPT_FB_DIFFUSE_C, w_d);
		splat<true>(fb, ps, sl, FPT_FB_SPECULAR_C, w_g);
	}
	else
	{
		if (comp & COMP_DIFFUSE_MASK) splat<true>(fb, ps, sl, FPT_FB_DIFFUSE_C, w_d);
		if (comp & COMP_GLOSSY_MASK)  splat<true>(fb, ps, sl, FPT_FB_SPECULAR_C, w_g);
	}
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
struct FusedResolve { const float4* w_d; const float4* w_g; const uint32_t* pixels; FrameBufferDev fb; PassInfo pass; uint32_t bounce; };

uint32_t trace_blocks_per_cu();
uint32_t trace_stack_entries();      // capacity of the traversal stack (LDS + scratch levels); fpt_rt_create_geometry checks the tree's bound against it
void launch_trace_closest(const TraceParams& p, bool counted, uint32_t n_blocks, hipStream_t stream);
void launch_trace_shadow(const TraceParams& p, bool fused_resolve, bool counted, uint32_t n_blocks, hipStream_t stream);
void launch_trace_mixed(const TraceParams& p, bool counted, uint32_t n_blocks, hipStream_t stream);
void launch_trace_mixed_psf(const TraceParams& p, bool counted, uint32_t n_blocks, hipStream_t stream);   // p.fused = a ResolveParams block (fpt_kernels.h)
void launch_trace_mixed_hits(const TraceParams& p, float4* shadow_hits, bool counted, uint32_t n_blocks, hipStream_t stream);   // closest-hit rays -> p.hits, the any-hit rays of p.shadow_rays -> shadow_hits (written, not resolved)

} // namespace fpt
