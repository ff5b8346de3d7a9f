// fpt_psf.h — device helpers of the path-space-filtering vertex processor shared by the shading kernels (fpt_pt.hip) and the
// traversal kernel's fused occlusion resolve (fpt_trace.hip): cache-info words, cell accumulation, PSFPTVertexProcessor::accumulate_nee.
#pragma once
#include "fpt_kernels.h"

namespace fpt {

__device__ __forceinline__ bool ci_valid(uint32_t c) { return (c & 0x1FFFFFFFu) != 0x1FFFFFFFu; }
__device__ __forceinline__ uint32_t ci_pack(uint32_t slot, uint32_t comp, uint32_t new_entry) { return (slot & 0x1FFFFFFFu) | ((comp & 3u) << 29) | ((new_entry & 1u) << 31); }
// the table pass k of a batch accumulates into (the one table when pass_stride == 0)
__device__ __forceinline__ PsfDev psf_pass_view(const PsfDev& p, uint32_t k)
{
	PsfDev v = p;
	if (p.pass_stride)
	{
		v.keys += size_t(k) * p.pass_stride; v.cells += 4 * size_t(k) * p.pass_stride;
		if (p.touched) { v.touched += size_t(k) * p.pass_stride; v.touched_n += k; }
	}
	return v;
}
__device__ __forceinline__ void psf_add(const PsfDev& psf, uint32_t slot, f3 v)
{
	const float c[3] = { v.x, v.y, v.z };
	#pragma unroll
	for (int k = 0; k < 3; ++k)
	{
		const long long q = __double2ll_rn(double(c[k]) * 4294967296.0);
		if (q) atomicAdd(reinterpret_cast<unsigned long long*>(psf.cells + 4 * size_t(slot) + k), (unsigned long long)q);
	}
}
__device__ __forceinline__ f3 psf_clamp(const PsfDev& psf, f3 v) { return all_finite(v) ? mk3(sel_min(v.x, psf.firefly), sel_min(v.y, psf.firefly), sel_min(v.z, psf.firefly)) : splat3(0.0f); }


// PSFPTVertexProcessor::accumulate_nee for ONE unoccluded light sample (src/psfpt_vertex_processor.h:345-441): to the sample's cache
// cell, to the frame, or to both
// `base_instance` = the first pass of the launch (the blocks behind a fused launch do not change from call to call, so it travels separately)
__device__ __forceinline__ void psf_resolve_sample(const ResolveParams& P, uint32_t base_instance, uint32_t i)
{
	const float4 wd4 = P.q.w_d[i], wg4 = P.q.w_g[i];
	const f3 w_d = mk3(wd4.x, wd4.y, wd4.z), w_g = mk3(wg4.x, wg4.y, wg4.z);
	const uint32_t pixel_info = as_u32(P.q.rays[2 * size_t(i) + 1].w), vinfo = P.q.vinfo[i];          // ShadowQueue: dir | PixelInfo, w_d.w = pass offset
	const uint32_t comp = (pixel_info >> 27) & 0xFu;
	PassInfo ps = P.pass; ps.base_instance = base_instance;
	const PathSlot sl = decode_slot(ps, pixel_info, ps.n_passes > 1 ? as_u32(wd4.w) : 0u);          // one pass: the pixel and 1 / (instance + 1); a batch: the path's pass plane
	const bool cached = ci_valid(vinfo), diffuse_only = ((vinfo >> 29) & 3u) == 1u;
	// the cell's share: integer sums, order-independent
	if (cached) psf_add(psf_pass_view(P.psf, sl.k), vinfo & 0x1FFFFFFFu, diffuse_only ? w_d : w_d + w_g);
	if (cached && !diffuse_only) return;
	// the frame's share: straight to the frame (one pass per render()), or to the path's cell of the batch's log, applied in order by the merge
	if (ps.n_passes == 1) { apply_psf_nee(FrameAdd{ P.fb, sl.pixel, sl.weight }, P.bounce, comp, cached, diffuse_only, w_d, w_g, P.psf.firefly); return; }
	const uint32_t pidx = sl.k * ps.acc_stride + sl.slot;
	float4* cell = P.log.nee[P.kind] + (size_t(P.bounce) * P.log.cap + pidx) * 2;
	cell[0] = make_float4(w_d.x, w_d.y, w_d.z, as_f32(comp | (cached ? 0x100u : 0u) | (diffuse_only ? 0x200u : 0u)));
	cell[1] = make_float4(w_g.x, w_g.y, w_g.z, 0.0f);
	log_mark(P.log, pidx, 3u * P.bounce + 1u + P.kind);
}

} // namespace fpt
