// fpt_kernels.h — launch-parameter blocks of the wavefront kernels (passed by value: each is a few hundred bytes, well
// under the 4 KB kernel-argument limit; the reference passes the ~1 KB RenderingContextView the same way,
// src/pathtracer_kernels.h:134,192,250).
#pragma once
#include "fpt_device.h"

namespace fpt {

#ifndef FPT_SHADE_BLOCK
#define FPT_SHADE_BLOCK 256      // one queue-append atomic per block; 256 measured best (128: +4 %, 512: +4 %, 1024: +18 % shading time)
#endif
static constexpr int SHADE_BLOCK = FPT_SHADE_BLOCK;

// TiledSequenceView (src/tiled_sequence.h:53-107).  The per-frame table samples[d][p] = fmodf(randfloat(d,instance+1) + shifts[d][p], 1)
// (src/tiled_sequence.cu:37-52,100-110) is evaluated on the fly inside the kernels from the shift table and the integer hash: same
// arithmetic, no 15 MB table rewrite per pass, and paths of different passes can coexist in one launch.
struct SequenceView { const float* shifts; uint32_t n_dims; uint32_t tile_size; };

struct PrimaryParams
{
	PathQueue out;
	SequenceView seq;
	const uint32_t* pixels;      // absolute pixel index per local path, or NULL for the identity map
	uint32_t n_pixels;           // local pixels per pass; the launch covers n_pixels * pass.n_passes paths
	uint32_t res_x, res_y;
	PassInfo pass;
	f3 eye, U, V, W;
	float W_len, sq_focal;
};

// path-space filtering state (PSFPT, src/renderers/psfpt.h, src/psfpt_vertex_processor.h): an open-addressing table of 64-bit
// spatial-hash keys; a cell = 3 fixed-point (2^-32) 64-bit sums + a sample count, so its value is independent of addition order
struct PsfDev
{
	unsigned long long* keys;    // ~0ull = empty
	long long* cells;            // 4 per slot: x, y, z, count
	uint32_t log2_size;
	uint32_t* ref_pixels; uint32_t* ref_cache; float4* ref_wd; float4* ref_wg; uint32_t* ref_size;      // PSFRefQueue
	uint32_t* ref_k;             // passes in flight: the pass offset of the referencing path (PathQueue::pass_k)
	f3 bbox_lo, bbox_hi;
	uint32_t depth; float width, max_prob, firefly;                                                     // PSFPTOptions
	uint32_t instance;
	// tile sharding (fpt_psfpt_set_sharded): keys / cells above are then the PASS table -- this rank's contributions of the pass in flight -- whose
	// freshly created slots are listed in `touched`; the blend reads the GLOBAL table g_keys / g_cells (every rank's cells of the reuse window, merged
	// by key after the exchange).  All NULL when one GPU renders the whole frame: keys / cells are then the one table.
	unsigned long long* g_keys; long long* g_cells; uint32_t* touched; uint32_t* touched_n;
	// passes in flight (fpt_psfpt_render_batch): pass k of the batch accumulates into ITS pass table -- keys + k * pass_stride, cells + 4 * k * pass_stride,
	// touched + k * pass_stride, touched_n + k -- of 2^log2_size slots each; g_log2_size is the global table's size.  pass_stride == 0: one table.
	uint32_t pass_stride, g_log2_size;
};
struct PsfRecord { unsigned long long key; long long v[4]; };     // one cell of a pass on the wire: key, three 2^-32 fixed-point sums, count (40 B)

struct ShadeParams
{
	PathQueue in, scatter;
	ShadowQueue shadow_dir, shadow;
	SequenceView seq;
	fpt_mesh_view mesh;
	const fpt_texture* textures;
	const ShadeRecord* shade_records;      // one 64-byte record per triangle (fpt_shading.h), or NULL: the vertex is set up from the mesh view's arrays
	const float* table;
	const fpt_dir_light* dir_lights;
	uint32_t n_dir_lights;
	EmitterView emitters;        // the NEE instantiation selected by nee_type (src/renderers/pathtracer_impl.h:272)
	FrameBufferDev fb;           // exact mode: the frame buffer; batched mode: the accumulation planes
	FrameBufferDev gbuffer;      // gbuffer pointers always refer to the real frame buffer
	fpt_pt_options opt;
	uint32_t res_x, res_y;
	uint32_t bounce;
	uint32_t do_nee, do_emissive, do_scatter;     // compute_per_bounce_options, src/pathtracer_core.h:594-620
	PassInfo pass;
	ContribLog log;              // plain PT, passes in flight: where the paths' frame-buffer contributions go (fpt_device.h)
	uint32_t write_gbuffer;
	PsfDev psf;                  // used by the PSF instantiation only
};

struct ResolveParams
{
	ShadowQueue q;
	const float4* hits;
	FrameBufferDev fb;
	uint32_t bounce;
	PassInfo pass;
	PsfDev psf;
	float frame_weight;
	ContribLog log; uint32_t kind;       // passes in flight: where the sample's frame part goes (kind 0 = directional light, 1 = mesh light)
};

void launch_sequence(uint32_t n_dims, uint32_t tile2, uint32_t instance, const float* shifts, float* samples, hipStream_t s);
void launch_primary_rays(const PrimaryParams& p, hipStream_t s);
void launch_shade_records(const fpt_mesh_view& mesh, ShadeRecord* out, hipStream_t s);          // ShadeRecord (fpt_shading.h): one thread per triangle
void launch_vpl_points(const EmitterView& em, const fpt_mesh_view& mesh, const fpt_texture* textures, float4* out, hipStream_t s);      // EmitterView::vpl_points (fpt_shading.h)
void launch_shade(const ShadeParams& p, uint32_t max_entries, hipStream_t s);
void launch_shade_psf(const ShadeParams& p, uint32_t max_entries, hipStream_t s);             // PSFPT vertex processor
void launch_psf_resolve(const ResolveParams& p, uint32_t max_entries, hipStream_t s);          // PSFPTVertexProcessor::accumulate_nee over a traced shadow queue
void launch_psf_collect(const PsfDev& psf, PsfRecord* out, hipStream_t s);                       // touched slots of the pass table -> records
void launch_psf_merge(const PsfDev& psf, const PsfRecord* records, const uint32_t* d_count, uint32_t count, hipStream_t s);      // records -> global table (insert by key, integer adds); d_count (device) overrides count when not NULL
void launch_psf_clear_pass(const PsfDev& psf, hipStream_t s);                                   // empty the touched slots of the pass table, reset the list
void launch_psf_blend(const PsfDev& psf, const FrameBufferDev& fb, float frame_weight, uint32_t max_refs, hipStream_t s);      // psf_blending_kernel
void launch_psf_blend_batch(const PsfDev& psf, const ContribLog& log, uint32_t bounce, const PassInfo& pass, uint32_t max_refs, hipStream_t s);   // the same for a batch: each reference reads its pass's table and leaves its three terms in the path's blend cell of the log
void launch_psf_prefix(const PsfDev& psf, uint32_t k, hipStream_t s);      // global table += pass table k; pass table k := the global values (what pass k's blend sees)
void launch_clamp_frame(const FrameBufferDev& fb, const uint32_t* pixels, uint32_t n, float max_value, hipStream_t s);         // clamp_frame_kernel
void launch_rescale(const FrameBufferDev& fb, const uint32_t* pixels, uint32_t n, float scale, hipStream_t s);
void launch_variance(const FrameBufferDev& fb, const uint32_t* pixels, uint32_t n_pixels, uint32_t n, hipStream_t s);
// the path tracer's passes in flight: replays the contribution log pass by pass (bit-identical to sequential render() calls); clears the albedo planes and the log's mask
void launch_merge_passes_exact(const FrameBufferDev& fb, float4* albedo_d, float4* albedo_s, const ContribLog& log, const uint32_t* pixels, uint32_t n_pixels, PassInfo pass, hipStream_t s,
                               bool psf = false, float firefly = 0.0f, float clamp_max = 0.0f);      // psf: the PSFPT's cells (cache-aware NEE terms, blends) and clamp_frame(clamp_max) after every pass
// frame-buffer gather (fpt_gather_framebuffer): dst[i] = channel[pixels[i]] and its inverse
void launch_pack_pixels(const float4* channel, const uint32_t* pixels, uint32_t n, float4* dst, hipStream_t s);
void launch_unpack_pixels(const float4* src, const uint32_t* pixels, uint32_t n, float4* channel, hipStream_t s);
void launch_rgba(const float4* composited, uint32_t n, float exposure, float inv_gamma, uint32_t* rgba, hipStream_t s);
// EAWParams (src/eaw.h): edge-stopping strengths + the camera frame used to size the positional kernel
struct EawParams { float phi_normal, phi_position, phi_color; f3 E, U, V, W; };
// one à-trous step; op < 0 = EAW_kernel, otherwise EAW_mad_kernel with the FilterOp bits (src/filters.h:44-57)
// nrm (optional): normals unpacked beforehand by launch_unpack_normals, float4 per pixel
void launch_eaw(float4* dst, int op, const float4* w_img, float w_min, const float4* img, const float4* geo, const float4* nrm, const float* var, const EawParams& prm, uint32_t step,
                uint32_t res_x, uint32_t res_y, hipStream_t s);
void launch_unpack_normals(const float4* geo, float4* nrm, uint32_t n, hipStream_t s);
void launch_filter_variance(const float4* img, float* var, uint32_t FW, uint32_t res_x, uint32_t res_y, hipStream_t s);
void launch_rgba_mode(const FrameBufferDev& fb, uint32_t mode, uint32_t n, float exposure, float inv_gamma, uint32_t* rgba, hipStream_t s);
void launch_debug_math(int op, uint32_t n, const float* a, const float* b, float* o0, float* o1, hipStream_t s);

} // namespace fpt
