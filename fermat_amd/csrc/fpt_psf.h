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
	const uint32_t pixel_info = P.q.pixels[i], vinfo = P.q.vinfo[i];
	const uint32_t comp = (pixel_info >> 27) & 0xFu;
	PassInfo ps = P.pass; ps.base_instance = base_instance;
	const PathSlot sl = decode_slot(ps, pixel_info);          // one pass: the pixel and 1 / (instance + 1); a batch: the path's pass plane
	if (ci_valid(vinfo))
	{
		const bool diffuse_only = ((vinfo >> 29) & 3u) == 1u;
		psf_add(psf_pass_view(P.psf, sl.k), vinfo & 0x1FFFFFFFu, diffuse_only ? w_d : w_d + w_g);
		if (diffuse_only)
		{
			splat<false>(P.fb, ps, sl, FPT_FB_COMPOSITED_C, psf_clamp(P.psf, w_g));
			splat<true>(P.fb, ps, sl, (P.bounce == 0 || (comp & COMP_GLOSSY_MASK)) ? FPT_FB_SPECULAR_C : FPT_FB_DIFFUSE_C, psf_clamp(P.psf, w_g));
		}
	}
	else
	{
		splat<false>(P.fb, ps, sl, FPT_FB_COMPOSITED_C, psf_clamp(P.psf, w_d + w_g));
		if (P.bounce == 0)
		{
			splat<true>(P.fb, ps, sl, FPT_FB_DIFFUSE_C, psf_clamp(P.psf, w_d));
			splat<true>(P.fb, ps, sl, FPT_FB_SPECULAR_C, psf_clamp(P.psf, w_g));
		}
		else
		{
			if (comp & COMP_DIFFUSE_MASK) splat<true>(P.fb, ps, sl, FPT_FB_DIFFUSE_C, psf_clamp(P.psf, w_d + w_g));
			if (comp & COMP_GLOSSY_MASK)  splat<true>(P.fb, ps, sl, FPT_FB_SPECULAR_C, psf_clamp(P.psf, w_d + w_g));
		}
	}
}

} // namespace fpt
