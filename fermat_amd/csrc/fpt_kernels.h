// fpt_kernels.h — launch-parameter blocks of the wavefront kernels (passed by value: each is a few hundred bytes, well
// under the 4 KB kernel-argument limit; the reference passes the ~1 KB RenderingContextView the same way,
// src/pathtracer_kernels.h:134,192,250).
#pragma once
#include "fpt_device.h"

namespace fpt {

#ifndef FPT_SHADE_BLOCK
#define FPT_SHADE_BLOCK 512
#endif
static constexpr int SHADE_BLOCK = FPT_SHADE_BLOCK;

struct SequenceView { const float* samples; const float* shifts; uint32_t n_dims; uint32_t tile_size; };   // TiledSequenceView, src/tiled_sequence.h:53-107

struct PrimaryParams
{
	PathQueue out;
	SequenceView seq;
	const uint32_t* pixels;      // absolute pixel index per local path, or NULL for the identity map
	uint32_t n_pixels;
	uint32_t res_x, res_y;
	f3 eye, U, V, W;
	float W_len, sq_focal;
};

struct ShadeParams
{
	PathQueue in, scatter;
	ShadowQueue shadow_dir, shadow;
	SequenceView seq;
	fpt_mesh_view mesh;
	const fpt_texture* textures;
	const float* table;
	const fpt_dir_light* dir_lights;
	uint32_t n_dir_lights;
	EmitterView emitters;        // the NEE instantiation selected by nee_type (src/renderers/pathtracer_impl.h:272)
	FrameBufferDev fb;
	fpt_pt_options opt;
	uint32_t res_x, res_y;
	uint32_t bounce;
	uint32_t do_nee, do_emissive, do_scatter;     // compute_per_bounce_options, src/pathtracer_core.h:594-620
	float frame_weight;
};

struct ResolveParams
{
	ShadowQueue q;
	const float4* hits;
	FrameBufferDev fb;
	uint32_t bounce;
	float frame_weight;
};

void launch_sequence(uint32_t n_dims, uint32_t tile2, uint32_t instance, const float* shifts, float* samples, hipStream_t s);
void launch_primary_rays(const PrimaryParams& p, hipStream_t s);
void launch_shade(const ShadeParams& p, uint32_t max_entries, hipStream_t s);
void launch_resolve(const ResolveParams& p, uint32_t max_entries, hipStream_t s);
void launch_rescale(const FrameBufferDev& fb, const uint32_t* pixels, uint32_t n, float scale, hipStream_t s);
void launch_variance(const FrameBufferDev& fb, const uint32_t* pixels, uint32_t n_pixels, uint32_t n, hipStream_t s);
void launch_rgba(const float4* composited, uint32_t n, float exposure, float inv_gamma, uint32_t* rgba, hipStream_t s);
void launch_debug_math(int op, uint32_t n, const float* a, const float* b, float* o0, float* o1, hipStream_t s);

} // namespace fpt
