// fpt_host.h — host-side state behind the C-ABI (include/fermat_pt_hip.h).
#pragma once
#include "fpt_kernels.h"
#include "fpt_bvh.h"
#include "fpt_bpt.h"
#include <memory>
#include <string>
#include <vector>
#include <stdexcept>
#include <cstring>
#include <cmath>
#include <chrono>

namespace fpt {

#define FPT_HIP_CHECK(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) throw std::runtime_error(std::string(#expr) + ": " + hipGetErrorString(_e)); } while (0)

template <typename T>
struct DeviceArray
{
	T* ptr = nullptr; size_t count = 0;
	~DeviceArray() { release(); }
	void release() { if (ptr) (void)hipFree(ptr); ptr = nullptr; count = 0; }
	// a failed hipMalloc leaves hipErrorOutOfMemory latched as the runtime's "last error"; it is read off here, so that a caller that recovers (the halving retry
	// of choose_passes_in_flight, host/fpt_renderer.cpp) does not meet it again in the first FPT_HIP_CHECK(hipGetLastError()) after a launch
	void alloc(size_t n)
	{
		if (n == count && ptr) return;
		release();
		if (n)
		{
			const hipError_t e = hipMalloc(&ptr, n * sizeof(T));
			if (e != hipSuccess) { ptr = nullptr; (void)hipGetLastError(); FPT_HIP_CHECK(e); }
		}
		count = n;
	}
	void upload(const T* h, size_t n, hipStream_t s) { alloc(n); if (n) { FPT_HIP_CHECK(hipMemcpyAsync(ptr, h, n * sizeof(T), hipMemcpyHostToDevice, s)); FPT_HIP_CHECK(hipStreamSynchronize(s)); } }
	void download(T* h, size_t n, hipStream_t s) const { if (n) { FPT_HIP_CHECK(hipMemcpyAsync(h, ptr, n * sizeof(T), hipMemcpyDeviceToHost, s)); FPT_HIP_CHECK(hipStreamSynchronize(s)); } }
	DeviceArray() = default; DeviceArray(const DeviceArray&) = delete; DeviceArray& operator=(const DeviceArray&) = delete;
};

// Microsoft CRT rand(): the generator behind Fermat's shift layers >= 7 (src/tiled_sampling.h:44-55; SURVEY Fact 10, A.16)
struct CrtRand
{
	uint32_t state = 1u;
	int next() { state = state * 214013u + 2531011u; return int((state >> 16) & 0x7fffu); }
};

// host builder of the Cranley-Patterson shift table (src/tiled_sampling.h:92-160,287-337)
void build_shift_table(uint32_t tile, uint32_t n_dims, const char* samples_dir, CrtRand& rng, std::vector<float>& shifts);

// host builder of the mesh-emitter tables (src/mesh_lights.cu:164-424)
struct EmitterTables
{
	std::vector<float> mesh_cdf, mesh_inv_area, vpl_cdf;
	std::vector<fpt_vpl> vpls;
	float norm = 0.0f;
};
void build_emitter_tables(uint32_t n_vpls, const fpt_mesh_view& h_mesh, const fpt_texture* h_textures, uint32_t instance, EmitterTables& out);
uint64_t emitter_fingerprint(const fpt_mesh_view& h_mesh, const fpt_texture* h_textures);          // of the emitting triangles' positions

struct QueueStorage
{
	DeviceArray<float4> rays, hits, weights; DeviceArray<uint32_t> vinfo; DeviceArray<float2> cones;
	size_t entries = 0;
	PathQueue view(uint32_t* size) { PathQueue q; q.rays = rays.ptr; q.hits = hits.ptr; q.weights = weights.ptr; q.cones = cones.ptr; q.size = size; q.vinfo = vinfo.count ? vinfo.ptr : nullptr; return q; }
	void alloc(size_t n) { rays.alloc(2 * n); hits.alloc(n); weights.alloc(n); cones.alloc(n); entries = n; }
};
struct ShadowStorage
{
	DeviceArray<float4> rays, w_d, w_g, hits; DeviceArray<uint32_t> vinfo;
	size_t entries = 0;
	ShadowQueue view(uint32_t* size) { ShadowQueue q; q.rays = rays.ptr; q.w_d = w_d.ptr; q.w_g = w_g.ptr; q.size = size; q.vinfo = vinfo.count ? vinfo.ptr : nullptr; return q; }
	void alloc(size_t n) { rays.alloc(2 * n); w_d.alloc(n); w_g.alloc(n); entries = n; }
};

} // namespace fpt

struct fpt_context
{
	int device = 0;
	hipStream_t stream = nullptr;
	uint32_t n_cus = 256;
	std::string error;

	// RT sub-boundary
	fpt::HostBvh2 host_bvh;
	fpt::DeviceArray<fpt::BvhNode8> d_nodes;            // the 8-wide compressed BVH (fpt_bvh.h)
	fpt::DeviceArray<fpt::BvhTriangle> d_tris;
	fpt::DeviceArray<uint32_t> d_counters;              // ticket dispensers + queue sizes, zeroed per pass
	fpt::DeviceArray<unsigned long long> d_trace_stats;
	bool has_geometry = false;
	uint32_t build_mode = 0;                             // fpt_rt_set_build_mode: 0 = quality (host: binned SAH + re-insertion + collapse), 1 = fast (device: Morton radix tree + collapse)
	// device-side refit (fpt_build.hip): per-record and per-node fp32 boxes, {|scene|max bits, error bits}
	fpt::DeviceArray<float> d_refit_tri_box, d_refit_node_box;
	fpt::DeviceArray<uint32_t> d_refit_scan;
	fpt::DeviceArray<uint8_t> d_build_scratch;           // the device builder's working set (fpt_build_lbvh.hip), kept between builds

	// sequence
	fpt::CrtRand crt_rand;
	uint32_t seq_dims = 0, seq_tile = 0;
	std::vector<float> h_shifts;
	fpt::DeviceArray<float> d_shifts, d_samples;

	// emitters
	fpt::EmitterTables emitters;
	fpt::DeviceArray<float> d_mesh_cdf, d_mesh_inv_area, d_vpl_cdf;
	fpt::DeviceArray<fpt_vpl> d_vpls;
	bool has_emitters = false;
	uint64_t emitters_fingerprint = 0; uint32_t emitters_n_vpls = 0, emitters_instance = 0; const void* emitters_mesh_identity[4] = { nullptr, nullptr, nullptr, nullptr };
	// the VPLs' tabulated light points (EmitterView::vpl_points): built from the view's mesh / materials / textures, so rebuilt when fpt_mesh_lights_init or
	// fpt_rt_create_geometry ran (emitter_generation) or a view names other buffers
	fpt::DeviceArray<fpt::ShadeRecord> d_shade_records; uint64_t shade_records_generation = 0; fpt_mesh_view shade_records_mesh{};      // ShadeRecord (fpt_shading.h): one per triangle of the view's mesh
	fpt::DeviceArray<float4> d_vpl_points; uint64_t emitter_generation = 1, vpl_points_generation = 0; fpt_mesh_view vpl_points_mesh{}; const fpt_texture* vpl_points_textures = nullptr;

	// renderer
	fpt_pt_options opt{};
	bool pt_ready = false;
	uint32_t n_local = 0;
	const uint32_t* d_pixels = nullptr;
	fpt::DeviceArray<fpt::FusedResolve> d_fused;        // per-(bounce, light kind) blocks read by the fused any-hit launches
	std::vector<fpt::FusedResolve> h_fused;             // what d_fused holds (re-uploaded only when it changes)
	fpt::QueueStorage q_a, q_b;
	fpt::ShadowStorage q_shadow_dir, q_shadow;
	uint32_t max_batch = 1;                              // passes in flight per fpt_pt_render_batch call
	// deferred fpt_pt_render (fpt_pt_set_deferred): consecutive render(instance) calls are collected and rendered as one batch -- bit-identical to
	// rendering them one by one -- when defer_max of them are pending or when anything is about to look at the frame (fpt_pt_flush, fpt_synchronize, ...)
	uint32_t defer_max = 1, defer_first = 0, defer_n = 0;
	uint32_t defer_kind = 0;                             // whose render() is deferred: 0 the PT's (fpt_pt_set_deferred), 1 the PSFPT's (fpt_psfpt_set_deferred), 2 the BPT's (fpt_bpt_set_deferred)
	fpt_rendering_context_view defer_view{};
	uint32_t defer_clear_at = 0;                         // fpt_clear_gbuffer while passes are pending: the number of pending passes the clear FOLLOWS (0 = none recorded)
	fpt_framebuffer_view defer_clear_fb{}; uint32_t defer_clear_pixels = 0;
	// Render lanes (fpt_pt_set_lanes): the rank's pixel list is cut into n_lanes contiguous ranges and every range is rendered by its own chain of
	// launches on its own HIP stream, so that the drain of one lane's traversal launch (it cannot end before its longest ray) overlaps the other
	// lanes' kernels.  A pixel belongs to one lane, and everything that touches a pixel stays in that lane's stream order: frames are bit-identical
	// for any number of lanes.  Lane 0 is the context's own stream and counters; the lanes share the queue arrays (disjoint ranges).
	struct PtLane
	{
		hipStream_t stream = nullptr; hipEvent_t done = nullptr;
		fpt::DeviceArray<uint32_t> counters;
		fpt::DeviceArray<fpt::FusedResolve> d_fused; std::vector<fpt::FusedResolve> h_fused;
	};
	std::vector<std::unique_ptr<PtLane>> extra_lanes;
	fpt::DeviceArray<uint32_t> d_identity;               // 0 .. n-1: the pixel list the lanes index when the caller passed none
	hipEvent_t lane_start = nullptr;
	fpt::DeviceArray<float4> filter_tmp[2], filter_nrm; fpt::DeviceArray<float> filter_var;     // fpt_filter scratch (ping-pong images, variance)
	fpt::DeviceArray<float> d_acc[6];                    // passes in flight: per-pass accumulation planes, float4 x n_local x max_batch per channel (the two albedo channels)
	// the path tracer's contribution log (fpt_device.h ContribLog): one cell per (pass in flight, pixel slot, bounce, kind) + the fill bits
	fpt::DeviceArray<float4> log_emissive, log_nee[2], log_blend; fpt::DeviceArray<uint32_t> log_mask;      // log_blend: the PSFPT's fourth kind
	uint32_t log_mask_words = 1;
	// path-space filtering (PSFPT): hash table of cache cells + reference queue
	struct PsfState
	{
		bool ready = false;
		fpt_psf_options opt{};
		uint32_t log2_size = 0;
		fpt::DeviceArray<unsigned long long> keys; fpt::DeviceArray<long long> cells;
		fpt::DeviceArray<uint32_t> ref_pixels, ref_cache, ref_size, ref_k; fpt::DeviceArray<float4> ref_wd, ref_wg;
		float bbox[6] = { 0, 0, 0, 0, 0, 0 };
		fpt::DeviceArray<fpt::ResolveParams> d_resolve;     // per-bounce blocks read by the MIXED launches with the fused cache-aware resolve
		std::vector<fpt::ResolveParams> h_resolve;
		// tile sharding (fpt_psfpt_set_sharded): the pass table, the list of its live slots, this rank's records of the pass in flight and the
		// receive buffer of the exchange; `pending` = a pass has been rendered and waits for fpt_psfpt_finish
		bool sharded = false, pending = false;
		uint32_t pending_instance = 0, pending_bounces = 0;
		fpt::DeviceArray<unsigned long long> p_keys; fpt::DeviceArray<long long> p_cells;
		fpt::DeviceArray<uint32_t> touched, touched_n;
		fpt::DeviceArray<fpt::PsfRecord> records, recv;
		fpt::DeviceArray<uint32_t> ex_counts;
		// passes in flight (fpt_psfpt_set_batch): one pass table of 2^b_log2 slots per pass of the batch and the lists of their live slots
		uint32_t max_batch = 1, b_log2 = 0;
		fpt::DeviceArray<unsigned long long> b_keys; fpt::DeviceArray<long long> b_cells;
		fpt::DeviceArray<uint32_t> b_touched, b_touched_n;
	} psf;
	// bidirectional path tracer
	struct BptState
	{
		bool ready = false, profiling = false, deferred_splats = false;
		fpt_bpt_options opt{};
		uint32_t n_local = 0, n_paths = 0;
		const uint32_t* d_pixels = nullptr;
		float light_tracing = 0.0f;
		fpt::DeviceArray<float> d_shifts; uint32_t seq_dims = 0;
		fpt::DeviceArray<float4> q_rays[2], q_hits[2], q_weights[2], q_pw[2]; fpt::DeviceArray<uint32_t> q_pixels[2]; fpt::DeviceArray<uint8_t> q_chan[2];
		fpt::DeviceArray<float4> s_rays, s_hits, s_weights; fpt::DeviceArray<uint32_t> s_pixels; fpt::DeviceArray<uint8_t> s_chan; fpt::DeviceArray<uint2> conn;
		fpt::DeviceArray<float4> v_pos; fpt::DeviceArray<fpt::LightVertexRecord> v_rec;
		fpt::DeviceArray<uint32_t> v_counts;
		fpt::DeviceArray<uint32_t> flat, flat_meta, flat_block_sums;      // -sc 1: the flat light-vertex list (fpt_bpt.h)
		fpt::DeviceArray<long long> splat; long long* splat_external = nullptr;
		// passes in flight (fpt_bpt_set_batch / fpt_bpt_render_batch): everything above is sized for max_batch passes; acc = the per-pass
		// accumulation planes; pending_* = a batch whose light-tracing splats still wait for fpt_bpt_resolve_splats (deferred mode)
		uint32_t max_batch = 1;
		fpt::DeviceArray<float4> acc[6];                     // the albedo channels' per-pass planes
		fpt::DeviceArray<float4> log_val; fpt::DeviceArray<uint32_t> log_chan, log_mask;      // the eye paths' contribution log (fpt_bpt.h BptLog)
		uint32_t pending_first = 0, pending_n = 0;
		long long* splat_ptr() { return splat_external ? splat_external : splat.ptr; }
		fpt::DeviceArray<uint32_t> counters;
		fpt_bpt_stats stats{};
		uint32_t ticket = 0;                                 // next ticket-dispenser group of the call in progress
		// shared light vertices (fpt_bpt_set_shared_light_vertices): the call stops after the light sub-paths until fpt_bpt_finish
		bool shared_lv = false, light_pending = false;
		uint32_t light_instance = 0, light_passes = 0;
		fpt::DeviceArray<fpt::LightVertexWire> lv_send, lv_recv; fpt::DeviceArray<uint32_t> lv_count;
	} bpt;
	bool profiling = false;
	int capture_bounce = -1;
	uint32_t captured_count = 0;
	std::vector<fpt_ray> cap_rays; std::vector<fpt_hit> cap_hits; std::vector<float> cap_weights; std::vector<uint32_t> cap_pixels; std::vector<float> cap_cones;
	fpt_pt_stats stats{};
	hipEvent_t ev[2] = { nullptr, nullptr };
	// asynchronous per-launch timing (profiling level 2): events are recorded on the stream and read back once, after the
	// timed region, so measuring costs no host synchronisation
	struct TimedLaunch { int bucket; uint32_t e0, e1; };
	hipEvent_t ev_ref = nullptr;                         // recorded when level-2 profiling is switched on: the common time base of the lanes' events
	float last_union_ms[5] = { 0, 0, 0, 0, 0 };          // per bucket: time during which at least one launch of the bucket was running (last collect)
	std::vector<hipEvent_t> ev_pool;
	std::vector<TimedLaunch> timed_launches;
	uint32_t ev_cursor = 0;
	int profiling_level = 0;
	bool counting = false;

	// multi-GPU (fpt_comm.cpp): the RCCL communicator of this rank (ncclComm_t), device copies of the ranks' pixel lists, message staging
	void* comm = nullptr; int comm_rank = 0, comm_world = 1; bool comm_owned = false;
	// the tile tables (fpt_set_tile_lists): device copies of the ranks' pixel lists (this rank's; on the root everybody's), their lengths, and four sampled
	// entries per list by which fpt_gather_framebuffer recognises the registered tables when a caller passes them again
	std::vector<std::unique_ptr<fpt::DeviceArray<uint32_t>>> comm_lists;
	std::vector<uint32_t> tile_counts, tile_samples; int tile_world = 0, tile_rank = 0, tile_root = 0;
	fpt::DeviceArray<float4> comm_staging, comm_recv;        // the packed message of this rank / the root's receive buffer

	uint32_t blocks_per_cu = 8;
	uint32_t trace_blocks() const { return n_cus * blocks_per_cu; }   // persistent grid: blocks_per_cu x 256-thread blocks per CU
};

// renders the passes fpt_pt_render has deferred (no-op when none are pending); throws on error.  Called by every entry point that reads or writes the
// frame, changes the renderer's set-up or synchronises
namespace fpt {
enum : uint32_t { DEFER_PT = 0, DEFER_PSFPT = 1, DEFER_BPT = 2 };
void flush_deferred(fpt_context* ctx);
// render(instance) of renderer `kind` in deferred mode: joins the pending run of consecutive instances of the same view, or starts a new one
void defer_pass(fpt_context* ctx, uint32_t kind, uint32_t instance, const fpt_rendering_context_view* view);
void psf_render_passes(fpt_context* ctx, uint32_t first, uint32_t n, const fpt_rendering_context_view* view);
void bpt_render_passes(fpt_context* ctx, uint32_t first, uint32_t n, const fpt_rendering_context_view* view);
}
// the VPLs' light-point table for this view (built or re-used; NULL when the VPL set is empty); fpt_api.cpp
namespace fpt { const float4* ensure_vpl_points(fpt_context* ctx, const fpt_rendering_context_view* view, hipStream_t s); }
// the triangles' shading records for this view's mesh (built or re-used)
namespace fpt { const ShadeRecord* ensure_shade_records(fpt_context* ctx, const fpt_rendering_context_view* view, hipStream_t s); }
// the view of the contribution log for the pixel range that starts at `first` of the rank's pixel list (fpt_api.cpp)
namespace fpt { ContribLog lane_log(fpt_context* ctx, uint32_t first); }
// BPT, shared light vertices (fpt_bpt_api.cpp): this rank's vertices of the batch in flight -> ctx->bpt.lv_send (returns their number); wire records -> the store
namespace fpt { uint32_t bpt_pack_own_vertices(fpt_context* ctx); void bpt_import_vertices(fpt_context* ctx, const LightVertexWire* d_records, uint32_t count); }

// shared by fpt_pt_set_batch and fpt_psfpt_set_batch (fpt_api.cpp): not part of the public boundary
extern "C" int fpt_internal_set_batch(fpt_context* ctx, uint32_t max_passes, const fpt_rendering_context_view* view, bool for_psfpt);

// ---- helpers shared by the C-ABI translation units (fpt_api.cpp, fpt_bpt_api.cpp) -------------------------------------------------
// counters block layout (uint32): trace ticket dispensers (8 shards, 128 B apart, x <= 3 launches per bounce x L <= 31), then queue sizes
// queue-size counters are per bounce (a fresh, pre-zeroed word for every queue of every bounce), so ONE memset per pass replaces
// the reference's two cudaMemsets per bounce (src/pathtracer_kernels.h:348-350)
enum { TICKET_STRIDE = 8 * 32, CNT_MAX_LAUNCHES = 96, CNT_TICKETS = 0, CNT_QUEUES = TICKET_STRIDE * CNT_MAX_LAUNCHES, CNT_PER_BOUNCE = 96,
       CNT_PATH = 0, CNT_SHADOW_DIR = 32, CNT_SHADOW = 64,            // offsets inside a bounce's group: each on its own 128-byte line
       CNT_TOTAL = CNT_QUEUES + CNT_PER_BOUNCE * 34 };

template <typename F>
inline int guarded(fpt_context* ctx, F&& f)
{
	if (!ctx) return -1;
	try { FPT_HIP_CHECK(hipSetDevice(ctx->device)); f(); return 0; }
	catch (const std::exception& e) { ctx->error = e.what(); (void)hipGetLastError(); return 1; }      // the error is reported through the return value: do not leave it latched
	catch (...) { ctx->error = "unknown error"; (void)hipGetLastError(); return 1; }
}

inline fpt::FrameBufferDev fb_dev(const fpt_framebuffer_view& v)
{
	fpt::FrameBufferDev f;
	for (int c = 0; c < FPT_FB_NUM_CHANNELS; ++c) f.ch[c] = reinterpret_cast<float4*>(v.channels[c]);
	f.gb_geo = reinterpret_cast<float4*>(v.gbuffer_geo); f.gb_uv = reinterpret_cast<float4*>(v.gbuffer_uv);
	f.gb_tri = v.gbuffer_tri; f.gb_depth = v.gbuffer_depth;
	return f;
}

inline void require(bool cond, const char* msg) { if (!cond) throw std::runtime_error(msg); }
inline double wall_seconds() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
// fpt_build_lbvh.hip: the device-side fast build (Morton radix tree -> SAH-optimal 8-wide collapse); false = the tree needs more stack than the kernel has: use the host builder
namespace fpt { bool build_acceleration_device(fpt_context* ctx, uint32_t tri_count, const int32_t* d_idx, uint32_t vertex_count, const float* d_vtx, uint32_t stack_limit); }

// asynchronous launch timing (fpt_pt_set_profiling level 2) for the renderers that have no synchronous profiling mode of their own
// (BPT, PSFPT): bucket 0 = closest-hit traversal, 2 = any-hit traversal, 3 = shading-side kernels; read with fpt_pt_collect_timings
template <typename F>
inline void timed_launch(fpt_context* ctx, int bucket, hipStream_t s, F&& launch)
{
	if (ctx->profiling_level == 2 && ctx->ev_cursor + 2 <= ctx->ev_pool.size())
	{
		const uint32_t e0 = ctx->ev_cursor, e1 = ctx->ev_cursor + 1; ctx->ev_cursor += 2;
		FPT_HIP_CHECK(hipEventRecord(ctx->ev_pool[e0], s));
		launch();
		FPT_HIP_CHECK(hipEventRecord(ctx->ev_pool[e1], s));
		ctx->timed_launches.push_back(fpt_context::TimedLaunch{ bucket, e0, e1 });
	}
	else launch();
}

inline fpt::TraceParams base_trace_params(fpt_context* ctx)
{
	fpt::TraceParams p; std::memset(&p, 0, sizeof(p));
	p.bvh.nodes = reinterpret_cast<const uint4*>(ctx->d_nodes.ptr);
	p.bvh.tris = reinterpret_cast<const float4*>(ctx->d_tris.ptr);
	p.n_nodes = uint32_t(ctx->host_bvh.nodes8.size());
	return p;
}

// camera_frame (src/camera.h:141-171) — host code, libm tanf as in the reference's host path
inline void camera_frame(const fpt_camera& c, float aspect, fpt::f3& U, fpt::f3& V, fpt::f3& W)
{
	using namespace fpt;
	W = mk3(c.aim[0] - c.eye[0], c.aim[1] - c.eye[1], c.aim[2] - c.eye[2]);
	const float wlen = sqrtf(dot(W, W));
	U = normalize(cross(W, mk3(c.up[0], c.up[1], c.up[2])));
	V = normalize(cross(U, W));
	const float ulen = wlen * tanf(c.fov / 2.0f);
	U = mk3(U.x * ulen, U.y * ulen, U.z * ulen);
	const float vlen = ulen / aspect;
	V = mk3(V.x * vlen, V.y * vlen, V.z * vlen);
}


