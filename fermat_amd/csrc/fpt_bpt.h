// fpt_bpt.h — device views and launch parameters of the bidirectional path tracer kernels (fpt_bpt.hip).
#pragma once
#include "fpt_kernels.h"

namespace fpt {

// RayQueue of the BPT (src/bpt_queues.h): rays, hits, path weight, MIS bookkeeping of the edge, PixelInfo.  The reference's PixelInfo is one word: pixel 27 bits,
// channel 4, diffuse 1 (src/pathtracer_core.h:527-542).  Here `pixels` holds the path's whole VIRTUAL id (pass offset x pixels + pixel, see BptParams) and the
// channel nibble travels in a byte plane of its own, `chan` (eye and connection queues only; round 5) -- until round 4 both shared the word, which capped the
// paths in flight at 2^27 (93 passes of a 1600 x 900 frame, 16 of a 4K frame); the cap is now 2^32 virtual ids and light-vertex slots, i.e. memory.
struct BptQueue { float4* rays; float4* hits; float4* weights; float4* path_weights; uint32_t* pixels; uint8_t* chan; uint32_t* size; };
struct BptShadowQueue { float4* rays; float4* hits; float4* weights; uint32_t* pixels; uint8_t* chan; uint32_t* size; };
// VertexStorageView (src/vertex_storage.h:46-66), path ordering: slot = path + depth * n_paths
// a stored light vertex is ONE 64-byte record: the eye vertices fetch vertices at random (-sc 1 draws from the list of all of them), and five
// parallel arrays cost five 64-byte sectors per fetch where the record costs one (the eye-vertex kernel runs at the box's copy bandwidth).
// `pos` repeats the record's first 16 bytes as a dense array for the scans that read nothing else (the frustum test of connect_camera).
struct LightVertexRecord { float4 pos; uint4 gbuffer; uint2 input; float2 weights; uint32_t path_id; uint32_t pad[3]; };
struct LightVertexStore { LightVertexRecord* rec; float4* pos; uint32_t* counts; };
// a stored vertex on the wire between ranks (shared light vertices, -sc 1 under tile sharding): its store slot + the record
struct LightVertexWire { uint32_t slot, pad[3]; LightVertexRecord rec; };
static_assert(sizeof(LightVertexWire) == 80, "light-vertex wire record must be 80 bytes");

// Passes in flight keep every frame contribution of an eye path apart (as the path tracer's ContribLog does, fpt_device.h): per bounce one emission cell
// and `conn_cells` connection cells (1 for -sc 1, max_path_length for -sc 0), each a float4 (the term, all four components) + the frame channel it goes
// to besides COMPOSITED_C; one fill bit per cell; the merge applies them pass by pass in the order of the sequential launches.
//   cell = bounce * (1 + conn_cells) + {0 emission, 1 + k: the k-th connection of the eye vertex (light-depth order)};  index [cell * cap + virtual path id]
struct BptLog { float4* val; uint32_t* chan; uint32_t* mask; uint32_t cap, mask_words, conn_cells; };

struct BptParams
{
	BptQueue in, out;
	BptShadowQueue shadow;
	uint2* conn;                 // per eye-queue entry: first shadow slot, number of connections queued
	// -sc 1 (single connection): slots of all stored light vertices, pass-major, then depth-major, light-path id minor; flat_meta[2k] = first
	// entry of pass k, flat_meta[2k + 1] = first entry of its depth-1 vertices (so [2k+1] - [2k] = #primary vertices), flat_meta[2 n_passes] = total
	uint32_t* flat; uint32_t* flat_meta; uint32_t* flat_block_sums;
	LightVertexStore store;
	long long* splat;            // 3 per pixel: light-tracing sums in 2^-32 fixed point
	SequenceView seq;
	fpt_mesh_view mesh;
	const fpt_texture* textures;
	const ShadeRecord* shade_records;      // one 64-byte record per triangle (fpt_shading.h), or NULL
	const float* table;
	EmitterView emitters;
	FrameBufferDev fb;
	fpt_bpt_options opt;
	const uint32_t* pixels;      // absolute pixel / light-path index per local path, or NULL
	uint32_t n_local, n_paths;   // paths handled here per pass; n_paths = res_x * res_y (light paths == eye paths == pixels)
	uint32_t res_x, res_y;
	uint32_t bounce, instance;   // instance = the first pass of the batch
	// passes in flight (fpt_bpt_render_batch): a path is addressed by its VIRTUAL id  k * n_paths + id  (k = pass offset, id = pixel /
	// light-path index): queues, the light-vertex store (slot = virtual id + depth * n_store), the vertex counts and the splat sums are
	// indexed by it; samples use `id` and `instance + k`; frame-buffer cells are  ch[c][k * plane_stride + id]  (per-pass
	// accumulation planes, merged in pass order).  n_passes = 1, plane_stride = 0 is the reference's one pass per render().
	uint32_t n_passes, n_store, plane_stride;
	float light_tracing;
	BptLog log;                  // passes in flight only
	f3 eye, U, V, W;
	float W_len, sq_focal;
};

void launch_bpt_light_primary(const BptParams& p, hipStream_t s);
void launch_bpt_light_vertices(const BptParams& p, uint32_t max_entries, hipStream_t s);
void launch_bpt_eye_primary(const BptParams& p, hipStream_t s);
void launch_bpt_eye_vertices(const BptParams& p, uint32_t max_entries, hipStream_t s);
void launch_bpt_eye_resolve(const BptParams& p, uint32_t max_entries, hipStream_t s);
void launch_bpt_connect_camera(const BptParams& p, hipStream_t s);
void launch_bpt_build_flat_list(const BptParams& p, hipStream_t s);     // -sc 1: count, scan, fill
// shared light vertices: this rank's stored vertices (paths `pixels`, n_passes passes) -> wire records, *out_count += their number; and the inverse
void launch_bpt_pack_light_vertices(const LightVertexRecord* rec, const uint32_t* counts, const uint32_t* pixels, uint32_t n_local, uint32_t n_paths, uint32_t n_passes,
                                    LightVertexWire* out, uint32_t* out_count, hipStream_t s);
void launch_bpt_unpack_light_vertices(const LightVertexWire* in, uint32_t n, LightVertexRecord* rec, float4* pos, uint32_t* counts, uint32_t n_store, hipStream_t s);
void launch_bpt_splat(const BptParams& p, uint32_t max_entries, hipStream_t s);
void launch_bpt_splat_resolve(const BptParams& p, hipStream_t s);
// passes in flight: per pass, in order: multiply_frame, the two albedo planes, the eye path's cells in the sequential order, the light-tracing splat sums -- the
// frame n sequential BPT::render calls leave, bit for bit; clears planes, fill bits and splat sums
void launch_bpt_merge_exact(const FrameBufferDev& fb, float4* albedo_d, float4* albedo_s, const BptLog& log, long long* splat, const uint32_t* pixels, uint32_t n_local,
                            uint32_t n_paths, uint32_t base_instance, uint32_t n_passes, uint32_t max_path_length, hipStream_t s);

} // namespace fpt
