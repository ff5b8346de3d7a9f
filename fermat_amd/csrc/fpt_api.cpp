// fpt_api.cpp — implementation of the C-ABI (include/fermat_pt_hip.h): context, RT sub-boundary, sequence, emitters and the
// PathTracer::render pass.  Host-side control flow mirrors path_trace_loop (src/pathtracer_kernels.h:309-391) but, MI355X
// first, never reads a queue size back inside a pass: every kernel bounds itself by the device-resident counters, the
// traversal kernels are persistent, and shadow tracing is fused with solve_occlusion.  One pass = 3 + 3*L launches
// (+2*L with directional lights) on one stream, no host synchronisation unless profiling/capture is on.
#include "fpt_host.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <chrono>

using namespace fpt;

namespace {

thread_local std::string g_create_error;

} // namespace

extern "C" {

int fpt_create(int device_id, fpt_context** out_ctx)
{
	if (!out_ctx) return -1;
	*out_ctx = nullptr;
	try
	{
		int n = 0;
		FPT_HIP_CHECK(hipGetDeviceCount(&n));
		if (device_id < 0 || device_id >= n) throw std::runtime_error("fpt_create: no such HIP device (this library has no CPU fallback)");
		FPT_HIP_CHECK(hipSetDevice(device_id));
		fpt_context* c = new fpt_context();
		c->device = device_id;
		hipDeviceProp_t prop;
		FPT_HIP_CHECK(hipGetDeviceProperties(&prop, device_id));
		c->n_cus = uint32_t(prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256);
		c->blocks_per_cu = trace_blocks_per_cu();
		FPT_HIP_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
		c->d_counters.alloc(CNT_TOTAL);
		c->d_trace_stats.alloc(8);
		FPT_HIP_CHECK(hipMemsetAsync(c->d_trace_stats.ptr, 0, 8 * sizeof(unsigned long long), c->stream));
		FPT_HIP_CHECK(hipMemsetAsync(c->d_counters.ptr, 0, CNT_TOTAL * sizeof(uint32_t), c->stream));
		FPT_HIP_CHECK(hipEventCreate(&c->ev[0])); FPT_HIP_CHECK(hipEventCreate(&c->ev[1]));
		FPT_HIP_CHECK(hipStreamSynchronize(c->stream));
		*out_ctx = c;
		return 0;
	}
	catch (const std::exception& e) { g_create_error = e.what(); return 1; }
}

void fpt_destroy(fpt_context* ctx)
{
	if (!ctx) return;
	(void)hipSetDevice(ctx->device);
	(void)guarded(ctx, [&] { flush_deferred(ctx); });          // render(instance) calls the library still holds back: the caller believes them rendered
	if (ctx->stream) { (void)hipStreamSynchronize(ctx->stream); }
	if (ctx->comm) (void)fpt_comm_destroy(ctx);
	for (auto& X : ctx->extra_lanes) { if (X->stream) { (void)hipStreamSynchronize(X->stream); (void)hipStreamDestroy(X->stream); } if (X->done) (void)hipEventDestroy(X->done); }
	if (ctx->lane_start) (void)hipEventDestroy(ctx->lane_start);
	if (ctx->ev_ref) (void)hipEventDestroy(ctx->ev_ref);
	for (int i = 0; i < 2; ++i) if (ctx->ev[i]) (void)hipEventDestroy(ctx->ev[i]);
	for (hipEvent_t e : ctx->ev_pool) (void)hipEventDestroy(e);
	hipStream_t s = ctx->stream;
	delete ctx;
	if (s) (void)hipStreamDestroy(s);
}

const char* fpt_last_error(const fpt_context* ctx) { return ctx ? ctx->error.c_str() : g_create_error.c_str(); }
void* fpt_stream(fpt_context* ctx) { return ctx ? (void*)ctx->stream : nullptr; }
int fpt_synchronize(fpt_context* ctx) { return guarded(ctx, [&] { flush_deferred(ctx); FPT_HIP_CHECK(hipStreamSynchronize(ctx->stream)); }); }

// ---- RT sub-boundary ------------------------------------------------------------------------------------------------------
int fpt_rt_create_geometry(fpt_context* ctx, uint32_t tri_count, const int32_t* d_idx, uint32_t vertex_count, const float* d_vtx)
{
	return guarded(ctx, [&] { flush_deferred(ctx);
		const double t0 = wall_seconds();
		// fast mode (fpt_rt_set_build_mode(ctx, 1) or FPT_BVH_BUILD=fast): the whole build on the device, the mesh stays where it is (fpt_build_lbvh.hip); a tree whose
		// traversal-stack bound exceeds the kernel's stack -- degenerate inputs -- falls through to the host builder and its ladder of shallower trees
		const char* env = std::getenv("FPT_BVH_BUILD");
		const bool fast = env ? std::strcmp(env, "fast") == 0 : ctx->build_mode == 1;
		if (fast && tri_count >= 2)
		{
			require(d_idx && d_vtx, "fpt_rt_create_geometry: null mesh");
			FPT_HIP_CHECK(hipStreamSynchronize(ctx->stream));          // launches still reading the old tree
			if (build_acceleration_device(ctx, tri_count, d_idx, vertex_count, d_vtx, trace_stack_entries()))
			{
				ctx->has_geometry = true; ctx->emitter_generation++;
				if (std::getenv("FPT_BVH_TIMERS")) std::fprintf(stderr, "fpt_rt_create_geometry: built on the device in %.3f ms\n", (wall_seconds() - t0) * 1e3);
				return;
			}
		}
		NoInitVector<int32_t> idx(size_t(tri_count) * 4); NoInitVector<float> vtx(size_t(vertex_count) * 4);
		if (tri_count) FPT_HIP_CHECK(hipMemcpy(idx.data(), d_idx, idx.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
		if (vertex_count) FPT_HIP_CHECK(hipMemcpy(vtx.data(), d_vtx, vtx.size() * sizeof(float), hipMemcpyDeviceToHost));
		const double t1 = wall_seconds();
		// binned-SAH BVH2, optimised by re-insertion, collapsed into the 8-wide compressed tree the kernels walk; shallower trees for degenerate inputs
		build_acceleration(tri_count, idx.data(), vertex_count, vtx.data(), ctx->host_bvh, trace_stack_entries());
		require(ctx->host_bvh.stack_need <= trace_stack_entries(), "fpt_rt_create_geometry: the BVH needs more traversal-stack entries than the kernel has");
		const double t2 = wall_seconds();
		ctx->d_nodes.upload(ctx->host_bvh.nodes8.data(), ctx->host_bvh.nodes8.size(), ctx->stream);
		ctx->d_tris.upload(ctx->host_bvh.tris8.data(), ctx->host_bvh.tris8.size(), ctx->stream);
		ctx->host_bvh.device_nodes = uint32_t(ctx->host_bvh.nodes8.size()); ctx->host_bvh.device_records = uint32_t(ctx->host_bvh.tris8.size()); ctx->host_bvh.built_on_device = false;
		ctx->has_geometry = true; ctx->emitter_generation++;          // new geometry: the VPLs' tabulated light points are stale
		if (std::getenv("FPT_BVH_TIMERS")) std::fprintf(stderr, "fpt_rt_create_geometry: mesh to the host %.3f s, build %.3f, tree to the device %.3f\n", t1 - t0, t2 - t1, wall_seconds() - t2);
	});
}
// 0 = quality (default): the host builder; 1 = fast: the device builder -- what a host that rebuilds every frame (RenderingContext::update_model without refit) wants
int fpt_rt_set_build_mode(fpt_context* ctx, uint32_t mode)
{ return guarded(ctx, [&] { require(mode <= 1, "fpt_rt_set_build_mode: 0 = quality (host), 1 = fast (device)"); ctx->build_mode = mode; }); }

int fpt_rt_refit_geometry(fpt_context* ctx, uint32_t tri_count, const int32_t* d_idx, uint32_t vertex_count, const float* d_vtx)
{
	// Device-side (round 6, fpt_build.hip): the mesh stays where it is, the records and every node's boxes are recomputed on the context's stream -- stream-ordered
	// behind the launches that still read the old tree -- and the host reads back 8 bytes (|scene|max and the error bits).  Byte for byte the tree refit_wide8 gives.
	return guarded(ctx, [&] { flush_deferred(ctx);
		require(ctx->has_geometry, "fpt_rt_refit_geometry: fpt_rt_create_geometry has not been called");
		HostBvh2& B = ctx->host_bvh;
		require(tri_count == B.device_records || (tri_count == 0 && B.device_records <= 1), "fpt_rt_refit_geometry: the triangle count differs from the tree's");
		require(tri_count == 0 || (d_idx && d_vtx), "fpt_rt_refit_geometry: null mesh");
		const double t0 = wall_seconds();
		const uint32_t n_records = uint32_t(ctx->d_tris.count), n_nodes = uint32_t(ctx->d_nodes.count);
		ctx->d_refit_scan.alloc(2); ctx->d_refit_tri_box.alloc(size_t(n_records) * 6); ctx->d_refit_node_box.alloc(size_t(n_nodes) * 6);
		FPT_HIP_CHECK(hipMemsetAsync(ctx->d_refit_scan.ptr, 0, 2 * sizeof(uint32_t), ctx->stream));
		launch_refit_scan(tri_count, d_idx, vertex_count, d_vtx, tri_count ? n_records : 0u, ctx->d_tris.ptr, ctx->d_refit_scan.ptr, ctx->stream);
		uint32_t scan[2] = { 0, 0 };
		ctx->d_refit_scan.download(scan, 2, ctx->stream);
		require(!(scan[1] & 1u), "fpt: refit found a triangle record outside the mesh");
		require(!(scan[1] & 2u), "fpt: vertex index out of range in refit");
		if (tri_count) launch_refit_records(n_records, ctx->d_tris.ptr, d_idx, d_vtx, ctx->d_refit_scan.ptr, ctx->d_refit_tri_box.ptr, ctx->stream);
		for (size_t L = B.level_begin.size() > 0 ? B.level_begin.size() - 1 : 0; L-- > 0;)
			launch_refit_level(ctx->d_nodes.ptr, ctx->d_refit_node_box.ptr, ctx->d_refit_tri_box.ptr, B.level_begin[L], B.level_begin[L + 1] - B.level_begin[L], ctx->d_refit_scan.ptr, ctx->stream);
		FPT_HIP_CHECK(hipGetLastError());
		std::memcpy(&B.scene_mag, &scan[0], 4);
		ctx->emitter_generation++;          // shading records and light points were tabulated from the old vertices
		// a non-finite vertex makes a box that cannot be quantised (the host refit throws there); the level kernels flag it.  Reading the flag waits for the refit (a
		// millisecond): the call returns with the tree in place or with the error, like the host refit did
		ctx->d_refit_scan.download(scan, 2, ctx->stream);
		if (scan[1] & 4u) { ctx->has_geometry = false; require(false, "fpt: internal wide-BVH quantisation error (refit): non-finite vertices? the geometry is invalid until fpt_rt_create_geometry runs again"); }
		B.seconds_refit = float(wall_seconds() - t0);
		if (std::getenv("FPT_BVH_TIMERS")) std::fprintf(stderr, "fpt_rt_refit_geometry: on the device, %.3f ms to completion (%u records, %u nodes, %zu levels)\n", B.seconds_refit * 1e3, n_records, n_nodes, B.level_begin.size() - 1);
	});
}

// test / diagnostic: the DEVICE tree as it stands (after a build or a device-side refit) copied to HOST arrays of n_nodes x 20 words and n_records x 12 words
int fpt_rt_download_bvh(fpt_context* ctx, uint32_t* h_nodes, float* h_records)
{
	return guarded(ctx, [&] { flush_deferred(ctx);
		require(ctx->has_geometry, "fpt_rt_download_bvh: no geometry");
		if (h_nodes) ctx->d_nodes.download(reinterpret_cast<BvhNode8*>(h_nodes), ctx->d_nodes.count, ctx->stream);
		if (h_records) ctx->d_tris.download(reinterpret_cast<BvhTriangle*>(h_records), ctx->d_tris.count, ctx->stream);
	});
}

// the buffers behind a view's mesh / textures were edited IN PLACE (a material colour, a texture coordinate, texels): the derived device tables -- shading
// records, the VPLs' light points -- are rebuilt at the next render call.  (The VPL distribution itself follows only fpt_mesh_lights_init, as in the reference.)
int fpt_mesh_invalidate(fpt_context* ctx)
{ return guarded(ctx, [&] { flush_deferred(ctx); ctx->emitter_generation++; }); }

static void rt_launch(fpt_context* ctx, uint32_t count, const fpt_ray* d_rays, fpt_hit* d_hits, uint32_t* d_bits, bool shadow, bool counted)
{
	require(ctx->has_geometry, "fpt_rt_trace*: create_geometry has not been called");
	if (count == 0) return;
	TraceParams p = base_trace_params(ctx);
	p.rays = reinterpret_cast<const float4*>(d_rays);
	p.hits = reinterpret_cast<float4*>(d_hits);
	p.bits = d_bits;
	p.count = count;
	p.work_counter = ctx->d_counters.ptr + CNT_TICKETS;
	p.stats = ctx->d_trace_stats.ptr;
	FPT_HIP_CHECK(hipMemsetAsync(p.work_counter, 0, TICKET_STRIDE * sizeof(uint32_t), ctx->stream));
	if (counted) FPT_HIP_CHECK(hipMemsetAsync(p.stats, 0, 8 * sizeof(unsigned long long), ctx->stream));
	if (d_bits) FPT_HIP_CHECK(hipMemsetAsync(d_bits, 0, size_t((count + 31) / 32) * sizeof(uint32_t), ctx->stream));
	const uint32_t blocks = std::min(ctx->trace_blocks(), (count + 255u) / 256u);
	if (shadow) launch_trace_shadow(p, false, counted, blocks, ctx->stream);
	else        launch_trace_closest(p, counted, blocks, ctx->stream);
	FPT_HIP_CHECK(hipGetLastError());
}

int fpt_rt_trace(fpt_context* ctx, uint32_t count, const fpt_ray* d_rays, fpt_hit* d_hits)
{ return guarded(ctx, [&] { rt_launch(ctx, count, d_rays, d_hits, nullptr, false, false); }); }
int fpt_rt_trace_shadow(fpt_context* ctx, uint32_t count, const fpt_ray* d_rays, fpt_hit* d_hits)
{ return guarded(ctx, [&] { rt_launch(ctx, count, d_rays, d_hits, nullptr, true, false); }); }
int fpt_rt_trace_shadow_bits(fpt_context* ctx, uint32_t count, const fpt_ray* d_rays, uint32_t* d_bits)
{ return guarded(ctx, [&] { rt_launch(ctx, count, d_rays, nullptr, d_bits, true, false); }); }
int fpt_rt_trace_counted(fpt_context* ctx, uint32_t count, const fpt_ray* d_rays, fpt_hit* d_hits, int shadow, fpt_trace_counters* h_out)
{
	return guarded(ctx, [&] {
		rt_launch(ctx, count, d_rays, d_hits, nullptr, shadow != 0, true);
		unsigned long long s[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
		if (count) ctx->d_trace_stats.download(s, 8, ctx->stream);
		if (h_out) { h_out->rays = count; h_out->nodes_visited = shadow ? s[4] : s[0]; h_out->tris_tested = shadow ? s[5] : s[1]; }
	});
}
int fpt_rt_bvh_info(fpt_context* ctx, uint32_t* n_nodes, uint32_t* n_leaf_tris, uint32_t* max_depth)
{
	return guarded(ctx, [&] {
		require(ctx->has_geometry, "fpt_rt_bvh_info: create_geometry has not been called");
		if (n_nodes) *n_nodes = ctx->host_bvh.device_nodes;
		if (n_leaf_tris) *n_leaf_tris = ctx->host_bvh.device_records;
		if (max_depth) *max_depth = ctx->host_bvh.wide_depth;
	});
}
static void fill_bvh_stats(const HostBvh2& b, fpt_bvh_stats* s)
{
	std::memset(s, 0, sizeof(*s));
	s->n_nodes = b.built_on_device ? b.device_nodes : uint32_t(b.nodes8.size()); s->n_records = b.built_on_device ? b.device_records : uint32_t(b.tris8.size());
	s->depth = b.wide_depth; s->stack_need = b.stack_need;
	uint64_t used = 0;
	for (int k = 0; k < 9; ++k) { s->slot_hist[k] = b.slot_hist[k]; used += uint64_t(k) * b.slot_hist[k]; }
	s->n_inner_children = b.n_inner_children; s->n_leaf_children = b.n_leaf_children; s->build_threads = b.threads;
	const uint32_t n_wide = b.built_on_device ? b.device_nodes : uint32_t(b.nodes8.size());
	s->avg_used_slots = n_wide ? float(double(used) / double(n_wide)) : 0.0f;
	s->sah_cost_binary = b.sah_cost; s->sah_cost_wide = b.wide_cost; s->seconds_binary = b.seconds_bvh2; s->seconds_wide = b.seconds_wide;
	s->seconds_refit = b.seconds_refit;
	s->seconds_optimise = b.seconds_opt; s->optimise_iterations = b.opt_iterations; s->inner_area_before = b.opt_cost_before; s->inner_area_after = b.opt_cost_after; s->depth_binary = b.max_depth;
}
int fpt_rt_bvh_stats(fpt_context* ctx, fpt_bvh_stats* out)
{
	return guarded(ctx, [&] {
		require(ctx->has_geometry, "fpt_rt_bvh_stats: create_geometry has not been called");
		require(out != nullptr, "fpt_rt_bvh_stats: null output");
		fill_bvh_stats(ctx->host_bvh, out);
	});
}

// ---- sequence ---------------------------------------------------------------------------------------------------------------
int fpt_sequence_setup(fpt_context* ctx, uint32_t n_dimensions, uint32_t tile_size, const char* h_samples_dir)
{
	return guarded(ctx, [&] { flush_deferred(ctx);          // pending render(instance) calls sample the table as it stood when they were made
		require(tile_size && (tile_size & (tile_size - 1)) == 0, "fpt_sequence_setup: tile_size must be a power of two");
		require(n_dimensions % 3 == 0 && n_dimensions > 0, "fpt_sequence_setup: n_dimensions must be a positive multiple of 3");
		build_shift_table(tile_size, n_dimensions, h_samples_dir, ctx->crt_rand, ctx->h_shifts);
		ctx->seq_dims = n_dimensions; ctx->seq_tile = tile_size;
		ctx->d_shifts.upload(ctx->h_shifts.data(), ctx->h_shifts.size(), ctx->stream);
		ctx->d_samples.alloc(ctx->h_shifts.size());
	});
}
int fpt_sequence_set_instance(fpt_context* ctx, uint32_t instance)
{
	return guarded(ctx, [&] { flush_deferred(ctx);
		require(ctx->seq_dims != 0, "fpt_sequence_set_instance: sequence not set up");
		launch_sequence(ctx->seq_dims, ctx->seq_tile * ctx->seq_tile, instance, ctx->d_shifts.ptr, ctx->d_samples.ptr, ctx->stream);
		FPT_HIP_CHECK(hipGetLastError());
	});
}
int fpt_sequence_download(fpt_context* ctx, float* h_shifts, float* h_samples)
{
	return guarded(ctx, [&] {
		const size_t n = size_t(ctx->seq_dims) * ctx->seq_tile * ctx->seq_tile;
		if (h_shifts) ctx->d_shifts.download(h_shifts, n, ctx->stream);
		if (h_samples) ctx->d_samples.download(h_samples, n, ctx->stream);
	});
}

// ---- emitters ---------------------------------------------------------------------------------------------------------------
int fpt_mesh_lights_init(fpt_context* ctx, uint32_t n_vpls, const fpt_mesh_view* h_mesh, const fpt_texture* h_textures, uint32_t instance)
{
	return guarded(ctx, [&] { flush_deferred(ctx);
		require(h_mesh != nullptr, "fpt_mesh_lights_init: null mesh");
		const double t0 = wall_seconds();
		build_emitter_tables(n_vpls, *h_mesh, h_textures, instance, ctx->emitters);
		const double t1 = wall_seconds();
		const EmitterTables& e = ctx->emitters;
		ctx->d_mesh_cdf.upload(e.mesh_cdf.data(), e.mesh_cdf.size(), ctx->stream);
		ctx->d_mesh_inv_area.upload(e.mesh_inv_area.data(), e.mesh_inv_area.size(), ctx->stream);
		ctx->d_vpl_cdf.upload(e.vpl_cdf.data(), e.vpl_cdf.size(), ctx->stream);
		ctx->d_vpls.upload(e.vpls.data(), e.vpls.size(), ctx->stream);
		ctx->has_emitters = true; ctx->emitter_generation++;
		ctx->emitters_fingerprint = emitter_fingerprint(*h_mesh, h_textures); ctx->emitters_n_vpls = n_vpls; ctx->emitters_instance = instance;
		ctx->emitters_mesh_identity[0] = h_mesh->vertex_indices; ctx->emitters_mesh_identity[1] = h_mesh->materials; ctx->emitters_mesh_identity[2] = h_mesh->material_indices; ctx->emitters_mesh_identity[3] = h_textures;
		if (std::getenv("FPT_BVH_TIMERS")) std::fprintf(stderr, "fpt_mesh_lights_init: tables %.3f s, to the device %.3f\n", t1 - t0, wall_seconds() - t1);
	});
}
// update_scene's form of fpt_mesh_lights_init: VERTICES of the same mesh moved (same index / material arrays, same textures, same n_vpls and instance).  The tables are a
// function of the emitting triangles' positions only, so they are rebuilt when one of those moved (*rebuilt = 1) and left alone otherwise (*rebuilt = 0; the derived
// device tables -- shading records, light points -- still follow the geometry through fpt_rt_refit_geometry / fpt_rt_create_geometry).
int fpt_mesh_lights_update(fpt_context* ctx, uint32_t n_vpls, const fpt_mesh_view* h_mesh, const fpt_texture* h_textures, uint32_t instance, int* rebuilt)
{
	if (rebuilt) *rebuilt = 1;
	if (ctx && h_mesh && ctx->has_emitters && n_vpls == ctx->emitters_n_vpls && instance == ctx->emitters_instance && ctx->emitters_mesh_identity[0] == h_mesh->vertex_indices &&
	    ctx->emitters_mesh_identity[1] == h_mesh->materials && ctx->emitters_mesh_identity[2] == h_mesh->material_indices && ctx->emitters_mesh_identity[3] == h_textures)
	{
		bool same = false;
		const int st = guarded(ctx, [&] { same = emitter_fingerprint(*h_mesh, h_textures) == ctx->emitters_fingerprint; });
		if (st != 0) return st;
		if (same) { if (rebuilt) *rebuilt = 0; return 0; }
	}
	return fpt_mesh_lights_init(ctx, n_vpls, h_mesh, h_textures, instance);
}
int fpt_mesh_lights_download(fpt_context* ctx, uint32_t* n_vpls, fpt_vpl* h_vpls, float* h_vpl_cdf, float* h_mesh_cdf, float* h_mesh_inv_area, float* norm)
{
	return guarded(ctx, [&] {
		require(ctx->has_emitters, "fpt_mesh_lights_download: mesh lights not initialised");
		const EmitterTables& e = ctx->emitters;
		if (n_vpls) *n_vpls = uint32_t(e.vpls.size());
		if (h_vpls) ctx->d_vpls.download(h_vpls, e.vpls.size(), ctx->stream);
		if (h_vpl_cdf) ctx->d_vpl_cdf.download(h_vpl_cdf, e.vpl_cdf.size(), ctx->stream);
		if (h_mesh_cdf) ctx->d_mesh_cdf.download(h_mesh_cdf, e.mesh_cdf.size(), ctx->stream);
		if (h_mesh_inv_area) ctx->d_mesh_inv_area.download(h_mesh_inv_area, e.mesh_inv_area.size(), ctx->stream);
		if (norm) *norm = e.norm;
	});
}

// ---- renderer ---------------------------------------------------------------------------------------------------------------
int fpt_pt_init(fpt_context* ctx, const fpt_pt_options* opts, const fpt_rendering_context_view* view, const char* h_samples_dir,
                const uint32_t* d_pixels, uint32_t n_local_pixels)
{
	return guarded(ctx, [&] { flush_deferred(ctx);
		require(opts && view, "fpt_pt_init: null argument");
		require(opts->max_path_length >= 1 && opts->max_path_length <= 31, "fpt_pt_init: max_path_length out of range [1,31]");
		require(opts->nee_type <= 1, "fpt_pt_init: only the mesh and vpl NEE algorithms are implemented");
		require(uint64_t(view->res_x) * view->res_y < (1ull << 27), "fpt_pt_init: PixelInfo holds 27-bit pixel indices");
		ctx->opt = *opts;
		ctx->n_local = d_pixels ? n_local_pixels : view->res_x * view->res_y;
		ctx->d_pixels = d_pixels;
		require(ctx->n_local > 0, "fpt_pt_init: empty pixel set");
		// queue arena (alloc_queues, src/pathtracer_kernels.h:90-126): two path queues, one shadow queue per light kind
		ctx->q_a.alloc(ctx->n_local); ctx->q_b.alloc(ctx->n_local);
		ctx->q_shadow.alloc(ctx->n_local);
		ctx->q_shadow_dir.alloc(view->dir_lights_count ? ctx->n_local : 1);
		// sampler (PathTracer::init, src/renderers/pathtracer_impl.h:148-150)
		build_shift_table(256, 6 * (opts->max_path_length + 1), h_samples_dir, ctx->crt_rand, ctx->h_shifts);
		ctx->seq_dims = 6 * (opts->max_path_length + 1); ctx->seq_tile = 256;
		ctx->d_shifts.upload(ctx->h_shifts.data(), ctx->h_shifts.size(), ctx->stream);
		ctx->d_samples.alloc(ctx->h_shifts.size());
		if (ctx->has_emitters && ctx->emitters.vpls.empty()) ctx->opt.nee_type = 0;     // :165-166
		ctx->max_batch = 1; ctx->defer_max = 1;
		ctx->pt_ready = true;
	});
}

int fpt_rescale_frame(fpt_context* ctx, const fpt_rendering_context_view* view, uint32_t instance)
{
	return guarded(ctx, [&] { flush_deferred(ctx);
		const uint32_t n = ctx->pt_ready ? ctx->n_local : view->res_x * view->res_y;
		launch_rescale(fb_dev(view->fb), ctx->pt_ready ? ctx->d_pixels : nullptr, n, float(instance) / float(instance + 1), ctx->stream);
		FPT_HIP_CHECK(hipGetLastError());
	});
}
int fpt_update_variances(fpt_context* ctx, const fpt_rendering_context_view* view, uint32_t instance)
{
	return guarded(ctx, [&] { flush_deferred(ctx);
		const uint32_t n = ctx->pt_ready ? ctx->n_local : view->res_x * view->res_y;
		launch_variance(fb_dev(view->fb), ctx->pt_ready ? ctx->d_pixels : nullptr, n, instance + 1, ctx->stream);
		FPT_HIP_CHECK(hipGetLastError());
	});
}
int fpt_multiply_frame(fpt_context* ctx, const fpt_rendering_context_view* view, float scale)
{
	return guarded(ctx, [&] { flush_deferred(ctx);
		launch_rescale(fb_dev(view->fb), nullptr, view->res_x * view->res_y, scale, ctx->stream);
		FPT_HIP_CHECK(hipGetLastError());
	});
}
int fpt_clamp_frame(fpt_context* ctx, const fpt_rendering_context_view* view, float max_value)
{
	return guarded(ctx, [&] { flush_deferred(ctx);
		launch_clamp_frame(fb_dev(view->fb), nullptr, view->res_x * view->res_y, max_value, ctx->stream);
		FPT_HIP_CHECK(hipGetLastError());
	});
}
int fpt_sequence_device_view(fpt_context* ctx, const float** d_shifts, uint32_t* n_dimensions, uint32_t* tile_size)
{
	return guarded(ctx, [&] {
		require(ctx->d_shifts.ptr != nullptr, "fpt_sequence_device_view: no sequence has been set up");
		if (d_shifts) *d_shifts = ctx->d_shifts.ptr;
		if (n_dimensions) *n_dimensions = ctx->seq_dims;
		if (tile_size) *tile_size = ctx->seq_tile;
	});
}
int fpt_mesh_lights_device_view(fpt_context* ctx, fpt_mesh_lights_view* out)
{
	return guarded(ctx, [&] {
		require(ctx->has_emitters && out, "fpt_mesh_lights_device_view: fpt_mesh_lights_init has not been called");
		out->d_mesh_cdf = ctx->d_mesh_cdf.ptr; out->d_mesh_inv_area = ctx->d_mesh_inv_area.ptr; out->n_prims = uint32_t(ctx->emitters.mesh_cdf.size());
		out->d_vpls = ctx->d_vpls.ptr; out->d_vpl_cdf = ctx->d_vpl_cdf.ptr; out->n_vpls = uint32_t(ctx->emitters.vpls.size()); out->norm = ctx->emitters.norm;
	});
}
int fpt_to_rgba(fpt_context* ctx, const fpt_rendering_context_view* view, uint8_t* d_rgba)
{
	return guarded(ctx, [&] { flush_deferred(ctx);
		launch_rgba(reinterpret_cast<const float4*>(view->fb.channels[FPT_FB_COMPOSITED_C]), view->res_x * view->res_y, view->exposure, 1.0f / view->gamma,
		            reinterpret_cast<uint32_t*>(d_rgba), ctx->stream);
		FPT_HIP_CHECK(hipGetLastError());
	});
}

int fpt_to_rgba_mode(fpt_context* ctx, const fpt_rendering_context_view* view, uint32_t shading_mode, uint8_t* d_rgba)
{
	return guarded(ctx, [&] { flush_deferred(ctx);
		require(view->fb.gbuffer_geo || (shading_mode != FPT_SHADING_NORMAL && shading_mode != FPT_SHADING_UV), "fpt_to_rgba_mode: this shading mode needs the gbuffer");
		launch_rgba_mode(fb_dev(view->fb), shading_mode, view->res_x * view->res_y, view->exposure, 1.0f / view->gamma, reinterpret_cast<uint32_t*>(d_rgba), ctx->stream);
		FPT_HIP_CHECK(hipGetLastError());
	});
}
int fpt_filter_variance(fpt_context* ctx, uint32_t res_x, uint32_t res_y, const float* d_img, float* d_var, uint32_t FW)
{
	return guarded(ctx, [&] { flush_deferred(ctx);
		launch_filter_variance(reinterpret_cast<const float4*>(d_img), d_var, FW, res_x, res_y, ctx->stream);
		FPT_HIP_CHECK(hipGetLastError());
	});
}
static EawParams eaw_params(const fpt_eaw_params& p)
{
	EawParams r; r.phi_normal = p.phi_normal; r.phi_position = p.phi_position; r.phi_color = p.phi_color;
	r.E = mk3(p.E[0], p.E[1], p.E[2]); r.U = mk3(p.U[0], p.U[1], p.U[2]); r.V = mk3(p.V[0], p.V[1], p.V[2]); r.W = mk3(p.W[0], p.W[1], p.W[2]);
	return r;
}
int fpt_eaw(fpt_context* ctx, uint32_t res_x, uint32_t res_y, float* d_dst, int op, const float* d_w_img, float w_min, const float* d_img,
            const float* d_gbuffer_geo, const float* d_var, const fpt_eaw_params* params, uint32_t step_size)
{
	return guarded(ctx, [&] { flush_deferred(ctx);
		require(d_dst && d_img && d_gbuffer_geo && params, "fpt_eaw: null buffer");
		require(op < 0 || d_w_img, "fpt_eaw: the weighted step needs a weight image");
		require(d_dst != d_img, "fpt_eaw: dst must not alias img");
		launch_eaw(reinterpret_cast<float4*>(d_dst), op, reinterpret_cast<const float4*>(d_w_img), w_min, reinterpret_cast<const float4*>(d_img),
		           reinterpret_cast<const float4*>(d_gbuffer_geo), nullptr, d_var, eaw_params(*params), step_size, res_x, res_y, ctx->stream);
		FPT_HIP_CHECK(hipGetLastError());
	});
}
int fpt_filter(fpt_context* ctx, const fpt_rendering_context_view* view, uint32_t instance)
{
	return guarded(ctx, [&] { flush_deferred(ctx);
		require(view->fb.gbuffer_geo != nullptr, "fpt_filter: the view has no gbuffer");
		const uint32_t W = view->res_x, H = view->res_y;
		const size_t n = size_t(W) * H;
		hipStream_t s = ctx->stream;
		ctx->filter_tmp[0].alloc(n); ctx->filter_tmp[1].alloc(n); ctx->filter_var.alloc(n); ctx->filter_nrm.alloc(n);
		float4* output = reinterpret_cast<float4*>(view->fb.channels[FPT_FB_FILTERED_C]);
		FPT_HIP_CHECK(hipMemcpyAsync(output, view->fb.channels[FPT_FB_DIRECT_C], n * sizeof(float4), hipMemcpyDeviceToDevice, s));
		EawParams p;
		p.phi_normal = 2.0f; p.phi_position = 1.0f; p.phi_color = float(instance * instance + 1u) / 10000.0f;
		p.E = mk3(view->camera.eye[0], view->camera.eye[1], view->camera.eye[2]);
		camera_frame(view->camera, view->aspect, p.U, p.V, p.W);
		const float4* geo = reinterpret_cast<const float4*>(view->fb.gbuffer_geo);
		launch_unpack_normals(geo, ctx->filter_nrm.ptr, uint32_t(n), s);
		const float4* nrm = ctx->filter_nrm.ptr;
		const int pairs[2][2] = { { FPT_FB_DIFFUSE_C, FPT_FB_DIFFUSE_A }, { FPT_FB_SPECULAR_C, FPT_FB_SPECULAR_A } };
		const uint32_t n_iterations = 7;
		for (int k = 0; k < 2; ++k)
		{
			const float4* input = reinterpret_cast<const float4*>(view->fb.channels[pairs[k][0]]);
			const float4* weight = reinterpret_cast<const float4*>(view->fb.channels[pairs[k][1]]);
			launch_filter_variance(input, ctx->filter_var.ptr, 2, W, H, s);
			// dst += w * eaw^n(img / w)  (src/eaw.cu:320-368): demodulate on the way in, plain steps in between, modulate + add on the way out
			uint32_t in_buffer = 0;
			for (uint32_t i = 0; i < n_iterations; ++i)
			{
				const uint32_t out_buffer = in_buffer ? 0 : 1;
				const float4* src = i == 0 ? input : ctx->filter_tmp[in_buffer].ptr;
				if (i == n_iterations - 1) launch_eaw(output, FPT_FILTER_OP_MODULATE_OUTPUT | FPT_FILTER_OP_ADD_MODE, weight, 1.0e-4f, src, geo, nrm, ctx->filter_var.ptr, p, 1u << i, W, H, s);
				else if (i == 0)           launch_eaw(ctx->filter_tmp[out_buffer].ptr, FPT_FILTER_OP_DEMODULATE_INPUT | FPT_FILTER_OP_REPLACE_MODE, weight, 1.0e-4f, src, geo, nrm, ctx->filter_var.ptr, p, 1u << i, W, H, s);
				else                       launch_eaw(ctx->filter_tmp[out_buffer].ptr, -1, nullptr, 0.0f, src, geo, nrm, ctx->filter_var.ptr, p, 1u << i, W, H, s);
				in_buffer = out_buffer;
			}
		}
		FPT_HIP_CHECK(hipGetLastError());
	});
}

// what one render lane works with: its stream, counters and resolve blocks (lane 0 = the context's), and its range [first, first + n) of the rank's
// pixel list (`pixels` = that range of the list, NULL = the identity when one lane renders everything)
struct LaneRefs
{
	hipStream_t s; uint32_t* cnt; DeviceArray<FusedResolve>* d_fused; std::vector<FusedResolve>* h_fused;
	uint32_t first, n; const uint32_t* pixels;
};
static PathQueue offset_queue(PathQueue q, size_t o) { q.rays += 2 * o; q.hits += o; q.weights += o; if (q.cones) q.cones += o; if (q.vinfo) q.vinfo += o; return q; }
static ShadowQueue offset_queue(ShadowQueue q, size_t o) { q.rays += 2 * o; q.w_d += o; q.w_g += o; if (q.vinfo) q.vinfo += o; return q; }

} // extern "C"
namespace fpt {
// the fields of a mesh view, one by one (a memcmp of the struct also compares its padding bytes, which a caller need not zero: ADVICE r4)
static bool same_mesh(const fpt_mesh_view& a, const fpt_mesh_view& b)
{
	return a.num_triangles == b.num_triangles && a.num_vertices == b.num_vertices && a.num_materials == b.num_materials && a.vertex_indices == b.vertex_indices &&
	       a.vertex_data == b.vertex_data && a.texture_indices_comp == b.texture_indices_comp && a.material_indices == b.material_indices && a.materials == b.materials &&
	       a.tex_bias[0] == b.tex_bias[0] && a.tex_bias[1] == b.tex_bias[1] && a.tex_scale[0] == b.tex_scale[0] && a.tex_scale[1] == b.tex_scale[1];
}
const ShadeRecord* ensure_shade_records(fpt_context* ctx, const fpt_rendering_context_view* view, hipStream_t s)
{
	const uint32_t n = view->mesh.num_triangles;
	if (n == 0) return nullptr;
	const bool fresh = ctx->shade_records_generation == ctx->emitter_generation && ctx->d_shade_records.count == size_t(n) &&
	                   same_mesh(ctx->shade_records_mesh, view->mesh);
	if (!fresh)
	{
		ctx->d_shade_records.alloc(n);
		launch_shade_records(view->mesh, ctx->d_shade_records.ptr, s);
		FPT_HIP_CHECK(hipGetLastError());
		ctx->shade_records_generation = ctx->emitter_generation; ctx->shade_records_mesh = view->mesh;
	}
	return ctx->d_shade_records.ptr;
}
const float4* ensure_vpl_points(fpt_context* ctx, const fpt_rendering_context_view* view, hipStream_t s)
{
	const uint32_t n = uint32_t(ctx->emitters.vpls.size());
	if (n == 0) return nullptr;
	const bool fresh = ctx->vpl_points_generation == ctx->emitter_generation && ctx->d_vpl_points.count == VPL_POINT_STRIDE * size_t(n) &&
	                   same_mesh(ctx->vpl_points_mesh, view->mesh) && ctx->vpl_points_textures == view->d_textures;
	if (!fresh)
	{
		ctx->d_vpl_points.alloc(VPL_POINT_STRIDE * size_t(n));
		EmitterView em; std::memset(&em, 0, sizeof(em));
		em.n_prims = uint32_t(ctx->emitters.mesh_cdf.size()); em.prims_cdf = ctx->d_mesh_cdf.ptr; em.prims_inv_area = ctx->d_mesh_inv_area.ptr;
		em.n_vpls = n; em.vpls = ctx->d_vpls.ptr; em.norm = ctx->emitters.norm;
		launch_vpl_points(em, view->mesh, view->d_textures, ctx->d_vpl_points.ptr, s);
		FPT_HIP_CHECK(hipGetLastError());
		ctx->vpl_points_generation = ctx->emitter_generation; ctx->vpl_points_mesh = view->mesh; ctx->vpl_points_textures = view->d_textures;
	}
	return ctx->d_vpl_points.ptr;
}
// the lane's view of the contribution log: every index is linear in the path index, so a lane's range is a pointer offset
ContribLog lane_log(fpt_context* ctx, uint32_t first)
{
	ContribLog g;
	g.cap = uint32_t(size_t(ctx->n_local) * ctx->max_batch); g.mask_words = ctx->log_mask_words; g.n_bounces = ctx->opt.max_path_length;
	g.emissive = ctx->log_emissive.ptr + first;
	g.nee[0] = ctx->log_nee[0].ptr ? ctx->log_nee[0].ptr + 2 * size_t(first) : nullptr;
	g.nee[1] = ctx->log_nee[1].ptr + 2 * size_t(first);
	g.blend = ctx->log_blend.ptr ? ctx->log_blend.ptr + 3 * size_t(first) : nullptr;
	g.mask = ctx->log_mask.ptr + size_t(first) * g.mask_words;
	return g;
}
} // namespace fpt
extern "C" {

// one wavefront of `n_passes` passes (instances instance .. instance + n_passes - 1) over one lane's pixels.  `batched`: samples go to the per-pass
// accumulation planes (plane k = pass instance + k; a lane owns columns first .. first + n - 1 of every plane); the caller merges.
static void render_lane(fpt_context* ctx, const LaneRefs& L, uint32_t instance, uint32_t n_passes, bool batched, const fpt_rendering_context_view* view)
{
	{
		hipStream_t s = L.s;
		const fpt_pt_options& opt = ctx->opt;
		const FrameBufferDev real_fb = fb_dev(view->fb);
		PassInfo pass; pass.base_instance = instance; pass.n_passes = n_passes; pass.n_slot = L.n; pass.acc_stride = ctx->n_local; pass.pixels = L.pixels;
		FrameBufferDev fb = real_fb;
		ContribLog log; std::memset(&log, 0, sizeof(log));
		if (batched)
		{
			// the two albedo channels keep a plane per pass (one term per pass and pixel); every other sample goes to the path's cell of the log
			for (int c = 0; c < 6; ++c) fb.ch[c] = nullptr;
			fb.ch[FPT_FB_DIFFUSE_A] = reinterpret_cast<float4*>(ctx->d_acc[FPT_FB_DIFFUSE_A].ptr) + L.first;
			fb.ch[FPT_FB_SPECULAR_A] = reinterpret_cast<float4*>(ctx->d_acc[FPT_FB_SPECULAR_A].ptr) + L.first;
			log = lane_log(ctx, L.first);
		}
		const uint32_t n_paths = L.n * n_passes;
		const size_t q_off = size_t(L.first) * ctx->max_batch;          // the lane's share of the queue arrays
		// persistent traversal grid: no more blocks than the lane's queues can feed (closest-hit + shadow rays <= 2 per path)
		const uint32_t trace_grid = std::min(ctx->trace_blocks(), std::max(1u, uint32_t((2ull * n_paths + 255ull) / 256ull)));
		uint32_t* cnt = L.cnt;
		const bool sync_mode = ctx->profiling || ctx->capture_bounce >= 0;
		float t_ms[5] = { 0, 0, 0, 0, 0 };
		auto timed = [&](int bucket, auto&& launch) {
			if (ctx->profiling_level == 2 && ctx->ev_cursor + 2 <= ctx->ev_pool.size())
			{
				const uint32_t e0 = ctx->ev_cursor, e1 = ctx->ev_cursor + 1; ctx->ev_cursor += 2;
				FPT_HIP_CHECK(hipEventRecord(ctx->ev_pool[e0], s));
				launch();
				FPT_HIP_CHECK(hipEventRecord(ctx->ev_pool[e1], s));
				ctx->timed_launches.push_back(fpt_context::TimedLaunch{ bucket, e0, e1 });
				return;
			}
			if (ctx->profiling) FPT_HIP_CHECK(hipEventRecord(ctx->ev[0], s));
			launch();
			if (ctx->profiling) { FPT_HIP_CHECK(hipEventRecord(ctx->ev[1], s)); FPT_HIP_CHECK(hipEventSynchronize(ctx->ev[1])); float ms = 0; FPT_HIP_CHECK(hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1])); t_ms[bucket] += ms; }
		};

		// PathTracer::render (src/renderers/pathtracer_impl.h:197-324); rescale_frame / update_variances (or the merge of the planes) are the caller's
		FPT_HIP_CHECK(hipMemsetAsync(cnt, 0, CNT_TOTAL * sizeof(uint32_t), s));

		SequenceView seq; seq.shifts = ctx->d_shifts.ptr; seq.n_dims = ctx->seq_dims; seq.tile_size = ctx->seq_tile;
		// path queue of bounce b counts in group b; shade_b fills the path queue of group b+1 and the shadow queues of group b
		auto counter = [&](uint32_t bounce, uint32_t which) { return cnt + CNT_QUEUES + CNT_PER_BOUNCE * bounce + which; };
		PathQueue qin = offset_queue(ctx->q_a.view(counter(0, CNT_PATH)), q_off), qout = offset_queue(ctx->q_b.view(counter(1, CNT_PATH)), q_off);
		ShadowQueue qsd = offset_queue(ctx->q_shadow_dir.view(counter(0, CNT_SHADOW_DIR)), ctx->q_shadow_dir.entries > 1 ? q_off : 0), qs = offset_queue(ctx->q_shadow.view(counter(0, CNT_SHADOW)), q_off);
		// The ray-cone plane (PTRayQueue's cone radius + pdf, src/pathtracer_queues.h) is carried for whoever reads it: the path-space filter's hash (fpt_psf_api.cpp)
		// and fpt_pt_set_capture.  The plain path tracer's vertices do not, and 8 B read + 8 B written per vertex are 4 % of a bandwidth-bound kernel's traffic.
#ifndef FPT_KEEP_CONES
		if (ctx->capture_bounce < 0) qin.cones = qout.cones = nullptr;
#endif

		// generate_primary_rays (src/pathtracer_kernels.h:166-181)
		{
			PrimaryParams pp;
			pp.out = qin; pp.seq = seq; pp.pixels = L.pixels; pp.n_pixels = L.n; pp.res_x = view->res_x; pp.res_y = view->res_y; pp.pass = pass;
			pp.eye = mk3(view->camera.eye[0], view->camera.eye[1], view->camera.eye[2]);
			camera_frame(view->camera, view->aspect, pp.U, pp.V, pp.W);
			pp.W_len = length(pp.W);
			const float tn = tanf(view->camera.fov / 2);
			pp.sq_focal = (float(view->res_x * view->res_y) / 4.0f) / (tn * tn);        // Camera::square_pixel_focal_length, src/camera.h:120-128
			launch_primary_rays(pp, s);
		}

		ShadeParams sh; std::memset(&sh, 0, sizeof(sh));
		sh.shadow_dir = qsd; sh.shadow = qs; sh.seq = seq;
		sh.mesh = view->mesh; sh.textures = view->d_textures; sh.table = view->d_glossy_reflectance; sh.shade_records = ensure_shade_records(ctx, view, s);
		sh.dir_lights = view->d_dir_lights; sh.n_dir_lights = view->dir_lights_count;
		EmitterView em;
		em.n_prims = uint32_t(ctx->emitters.mesh_cdf.size()); em.prims_cdf = ctx->d_mesh_cdf.ptr; em.prims_inv_area = ctx->d_mesh_inv_area.ptr;
		em.n_vpls = opt.nee_type == 1 ? uint32_t(ctx->emitters.vpls.size()) : 0u; em.vpls = opt.nee_type == 1 ? ctx->d_vpls.ptr : nullptr; em.norm = ctx->emitters.norm;
		em.vpl_points = opt.nee_type == 1 ? ensure_vpl_points(ctx, view, s) : nullptr;
		sh.emitters = em;
		sh.fb = fb; sh.log = log; sh.gbuffer = real_fb; sh.opt = opt; sh.res_x = view->res_x; sh.res_y = view->res_y;
		sh.pass = pass;
		const uint32_t total_vpls = uint32_t(ctx->emitters.vpls.size());

		// what a fused any-hit launch needs to retire an unoccluded sample, one block per (bounce, light kind).  The blocks hold nothing that
		// changes from pass to pass (the first instance travels as a kernel argument), so they are uploaded -- which synchronises the
		// stream -- only when a pointer or the batch shape changed, and consecutive render calls stay asynchronous
		{
			std::vector<FusedResolve> blocks(2 * size_t(opt.max_path_length));
			std::memset(blocks.data(), 0, blocks.size() * sizeof(FusedResolve));
			PassInfo block_pass = pass; block_pass.base_instance = 0;
			for (uint32_t b = 0; b < opt.max_path_length; ++b)
				for (int kind = 0; kind < 2; ++kind)
				{
					const ShadowQueue& q = kind ? qs : qsd;
					FusedResolve& f = blocks[2 * size_t(b) + kind];
					f.w_d = q.w_d; f.w_g = q.w_g; f.fb = fb; f.pass = block_pass; f.bounce = b; f.log = log; f.kind = uint32_t(kind);
				}
			if (L.h_fused->size() != blocks.size() || std::memcmp(L.h_fused->data(), blocks.data(), blocks.size() * sizeof(FusedResolve)) != 0)
			{
				L.d_fused->upload(blocks.data(), blocks.size(), s);
				*L.h_fused = blocks;
			}
		}
		auto fused_block = [&](const ShadowQueue& q, uint32_t bounce) { return L.d_fused->ptr + 2 * size_t(bounce) + (q.w_d == qs.w_d ? 1 : 0); };

		fpt_pt_stats& st = ctx->stats;
		if (sync_mode) { std::memset(&st, 0, sizeof(st)); }
		ctx->captured_count = 0;
		uint32_t ticket = 0;

		// closest-hit trace of the primary rays (RTContext::trace); later bounces are traced by the MIXED launch at the end of
		// the previous iteration, together with that bounce's shadow rays
		{
			TraceParams tp = base_trace_params(ctx);
			tp.rays = qin.rays; tp.hits = qin.hits; tp.count_ptr = qin.size; tp.work_counter = cnt + CNT_TICKETS + TICKET_STRIDE * (ticket++);
			tp.stats = ctx->d_trace_stats.ptr;
			timed(0, [&] { launch_trace_closest_queue(tp, true, ctx->counting, trace_grid, s); });
		}
		for (uint32_t bounce = 0; bounce < opt.max_path_length; ++bounce)
		{
			// compute_per_bounce_options (src/pathtracer_core.h:594-620)
			sh.bounce = bounce;
			sh.do_nee = total_vpls && ((bounce + 2 <= opt.max_path_length) &&
				((bounce == 0 && opt.direct_lighting_nee && opt.direct_lighting) || (bounce > 0 && opt.indirect_lighting_nee)));
			sh.do_emissive = ((bounce == 0 && opt.visible_lights) || (bounce == 1 && opt.direct_lighting_bsdf && opt.direct_lighting) || (bounce > 1 && opt.indirect_lighting_bsdf));
			const uint32_t max_vertices = opt.max_path_length + (((opt.max_path_length == 2 && opt.direct_lighting_bsdf) || (opt.max_path_length > 2 && opt.indirect_lighting_bsdf)) ? 1 : 0);
			sh.do_scatter = (bounce + 2 < max_vertices);

			if (sync_mode)
			{
				uint32_t in_size = 0;
				FPT_HIP_CHECK(hipMemcpyAsync(&in_size, qin.size, sizeof(uint32_t), hipMemcpyDeviceToHost, s)); FPT_HIP_CHECK(hipStreamSynchronize(s));
				st.in_size[bounce] = in_size; st.n_bounces = bounce + 1; st.shade_events += in_size; st.rays_traced += in_size;
				if (in_size == 0) { st.n_bounces = bounce; break; }
			}
			if (ctx->capture_bounce == int(bounce))
			{
				uint32_t n = 0;
				FPT_HIP_CHECK(hipMemcpyAsync(&n, qin.size, sizeof(uint32_t), hipMemcpyDeviceToHost, s)); FPT_HIP_CHECK(hipStreamSynchronize(s));
				ctx->captured_count = n;
				ctx->cap_rays.resize(n); ctx->cap_hits.resize(n); ctx->cap_weights.resize(size_t(n) * 4); ctx->cap_pixels.resize(n); ctx->cap_cones.resize(size_t(n) * 2);
				if (n)
				{
					FPT_HIP_CHECK(hipMemcpy(ctx->cap_rays.data(), qin.rays, size_t(n) * 32, hipMemcpyDeviceToHost));
					FPT_HIP_CHECK(hipMemcpy(ctx->cap_hits.data(), qin.hits, size_t(n) * 16, hipMemcpyDeviceToHost));
					FPT_HIP_CHECK(hipMemcpy(ctx->cap_weights.data(), qin.weights, size_t(n) * 16, hipMemcpyDeviceToHost));
					// the queue's rays carry PixelInfo / the pass offset where a Ray has tmin / tmax (fpt_device.h PathQueue): hand out the reference's PTRayQueue entry
					for (uint32_t e = 0; e < n; ++e)
					{
						ctx->cap_pixels[e] = ctx->cap_rays[e].mask_or_tmin;
						const float tmin = bounce ? QUEUE_SCATTER_TMIN : QUEUE_PRIMARY_TMIN;
						std::memcpy(&ctx->cap_rays[e].mask_or_tmin, &tmin, 4);
						ctx->cap_rays[e].tmax = bounce ? QUEUE_SCATTER_TMAX : QUEUE_PRIMARY_TMAX;
					}
					FPT_HIP_CHECK(hipMemcpy(ctx->cap_cones.data(), qin.cones, size_t(n) * 8, hipMemcpyDeviceToHost));
				}
			}
			// this bounce's output counters are fresh words zeroed by the per-pass memset
			qout.size = counter(bounce + 1, CNT_PATH); qsd.size = counter(bounce, CNT_SHADOW_DIR); qs.size = counter(bounce, CNT_SHADOW);
			sh.in = qin; sh.scatter = qout; sh.shadow_dir = qsd; sh.shadow = qs;
			timed(3, [&] { launch_shade(sh, n_paths, s); });

			// directional-light samples are resolved first (their own queue), then the mesh-light samples of the same bounce
			if (view->dir_lights_count)
			{
				TraceParams sp = base_trace_params(ctx);
				sp.work_counter = cnt + CNT_TICKETS + TICKET_STRIDE * (ticket++);
				sp.shadow_rays = qsd.rays; sp.shadow_size = qsd.size; sp.fused = fused_block(qsd, bounce); sp.base_instance = instance; sp.stats = ctx->d_trace_stats.ptr;
				timed(2, [&] { launch_trace_shadow(sp, true, ctx->counting, trace_grid, s); });
			}
			if (bounce + 1 < opt.max_path_length)
			{
				// ONE launch: closest-hit trace of the scattered rays (= bounce+1's RTContext::trace) + any-hit trace of this bounce's
				// shadow rays fused with solve_occlusion (RTContext::trace_shadow + solve_occlusion)
				TraceParams mp = base_trace_params(ctx);
				mp.rays = qout.rays; mp.hits = qout.hits; mp.count_ptr = qout.size; mp.work_counter = cnt + CNT_TICKETS + TICKET_STRIDE * (ticket++);
				mp.shadow_rays = qs.rays; mp.shadow_size = qs.size; mp.fused = fused_block(qs, bounce); mp.base_instance = instance; mp.stats = ctx->d_trace_stats.ptr;
				timed(1, [&] { launch_trace_mixed(mp, ctx->counting, trace_grid, s); });
			}
			else if (sh.do_nee)
			{
				TraceParams sp = base_trace_params(ctx);
				sp.work_counter = cnt + CNT_TICKETS + TICKET_STRIDE * (ticket++);
				sp.shadow_rays = qs.rays; sp.shadow_size = qs.size; sp.fused = fused_block(qs, bounce); sp.base_instance = instance; sp.stats = ctx->d_trace_stats.ptr;
				timed(2, [&] { launch_trace_shadow(sp, true, ctx->counting, trace_grid, s); });
			}
			if (sync_mode)
			{
				uint32_t sz[2] = { 0, 0 };
				FPT_HIP_CHECK(hipMemcpyAsync(sz, qsd.size, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
				FPT_HIP_CHECK(hipMemcpyAsync(sz + 1, qs.size, sizeof(uint32_t), hipMemcpyDeviceToHost, s)); FPT_HIP_CHECK(hipStreamSynchronize(s));
				st.shadow_dir_size[bounce] = sz[0]; st.shadow_size[bounce] = sz[1]; st.shadow_rays_traced += sz[0] + sz[1];
			}
			std::swap(qin, qout);
		}
		FPT_HIP_CHECK(hipGetLastError());
		if (ctx->profiling)
		{
			st.primary_rt_ms = t_ms[0]; st.path_rt_ms = t_ms[1]; st.shadow_rt_ms = t_ms[2]; st.path_shade_ms = t_ms[3]; st.shadow_shade_ms = 0.0f;
		}
	}
}

static void render_passes_impl(fpt_context* ctx, uint32_t instance, uint32_t n_passes, const fpt_rendering_context_view* view)
{
	{
		require(ctx->pt_ready, "fpt_pt_render: fpt_pt_init has not been called");
		require(n_passes >= 1 && n_passes <= ctx->max_batch, "fpt_pt_render_batch: n_passes exceeds the batch capacity set by fpt_pt_set_batch");
		require(ctx->has_geometry, "fpt_pt_render: create_geometry has not been called");
		require(ctx->has_emitters, "fpt_pt_render: fpt_mesh_lights_init has not been called");
		hipStream_t s = ctx->stream;
		const FrameBufferDev real_fb = fb_dev(view->fb);
		const bool batched = n_passes > 1;       // batched mode accumulates into per-pass planes and merges them in order at the end
		const bool sync_mode = ctx->profiling || ctx->capture_bounce >= 0;
		// the lanes: contiguous ranges of the rank's pixel list, each with its own launch chain on its own stream
		uint32_t n_lanes = sync_mode ? 1u : 1u + uint32_t(ctx->extra_lanes.size());
		while (n_lanes > 1 && ctx->n_local / n_lanes < 4096u) --n_lanes;
		if (!ctx->d_pixels && ctx->d_identity.count != ctx->n_local) n_lanes = 1;
		const uint32_t* list = ctx->d_pixels ? ctx->d_pixels : (n_lanes > 1 ? ctx->d_identity.ptr : nullptr);
		if (ctx->opt.nee_type == 1) (void)ensure_vpl_points(ctx, view, s);          // on the context's stream, before the lanes branch off it
		(void)ensure_shade_records(ctx, view, s);
		if (n_lanes > 1) FPT_HIP_CHECK(hipEventRecord(ctx->lane_start, s));
		for (uint32_t j = 0; j < n_lanes; ++j)
		{
			const uint32_t p0 = uint32_t(uint64_t(ctx->n_local) * j / n_lanes), p1 = uint32_t(uint64_t(ctx->n_local) * (j + 1) / n_lanes);
			LaneRefs L;
			if (j == 0) L = LaneRefs{ s, ctx->d_counters.ptr, &ctx->d_fused, &ctx->h_fused, p0, p1 - p0, list };
			else
			{
				fpt_context::PtLane& X = *ctx->extra_lanes[j - 1];
				FPT_HIP_CHECK(hipStreamWaitEvent(X.stream, ctx->lane_start, 0));      // after everything queued on the context's stream so far
				L = LaneRefs{ X.stream, X.counters.ptr, &X.d_fused, &X.h_fused, p0, p1 - p0, list + p0 };
			}
			PassInfo pass; pass.base_instance = instance; pass.n_passes = n_passes; pass.n_slot = L.n; pass.acc_stride = ctx->n_local; pass.pixels = L.pixels;
			if (!batched)
			{
				// RenderingContextImpl::render's bracket around the renderer (src/renderer.cu:1036-1066), per lane: every pixel sees rescale -> samples -> variances
				launch_rescale(real_fb, L.pixels, L.n, float(instance) / float(instance + 1), L.s);
				render_lane(ctx, L, instance, 1, false, view);
				launch_variance(real_fb, L.pixels, L.n, instance + 1, L.s);
			}
			else
			{
				render_lane(ctx, L, instance, n_passes, true, view);
				launch_merge_passes_exact(real_fb, reinterpret_cast<float4*>(ctx->d_acc[FPT_FB_DIFFUSE_A].ptr) + p0, reinterpret_cast<float4*>(ctx->d_acc[FPT_FB_SPECULAR_A].ptr) + p0,
				                          lane_log(ctx, p0), L.pixels, L.n, pass, L.s);
			}
			if (j > 0) FPT_HIP_CHECK(hipEventRecord(ctx->extra_lanes[j - 1]->done, L.s));
		}
		for (uint32_t j = 1; j < n_lanes; ++j) FPT_HIP_CHECK(hipStreamWaitEvent(s, ctx->extra_lanes[j - 1]->done, 0));
		FPT_HIP_CHECK(hipGetLastError());
	}
}

} // extern "C"
namespace fpt {
static void gbuffer_fill(fpt_context* ctx, const fpt_framebuffer_view& fb, size_t n)
{
	if (!fb.gbuffer_geo) return;
	FPT_HIP_CHECK(hipMemsetAsync(fb.gbuffer_geo, 0xFF, n * 16, ctx->stream)); FPT_HIP_CHECK(hipMemsetAsync(fb.gbuffer_uv, 0xFF, n * 16, ctx->stream));
	FPT_HIP_CHECK(hipMemsetAsync(fb.gbuffer_tri, 0xFF, n * 4, ctx->stream)); FPT_HIP_CHECK(hipMemsetAsync(fb.gbuffer_depth, 0xFF, n * 4, ctx->stream));
}
void clear_gbuffer(fpt_context* ctx, const fpt_rendering_context_view* view)
{
	// pending passes write the gbuffer when they are rendered (the last one of a batch does): a clear that follows k of them is carried out where it falls in that sequence
	if (ctx->defer_n == 0) { gbuffer_fill(ctx, view->fb, size_t(view->res_x) * view->res_y); return; }
	ctx->defer_clear_at = ctx->defer_n; ctx->defer_clear_fb = view->fb; ctx->defer_clear_pixels = view->res_x * view->res_y;
}
void flush_deferred(fpt_context* ctx)
{
	if (ctx->defer_n == 0) return;
	const uint32_t first = ctx->defer_first, n = ctx->defer_n;
	ctx->defer_n = 0;
	const uint32_t clear_at = ctx->defer_clear_at; ctx->defer_clear_at = 0;
	// a recorded clear: before the batch when a later pass of the batch rewrites the gbuffer (only the last pass's hits and the last clear before it survive n sequential
	// {clear, render} calls), after the batch when the clear was the last thing the caller did
	if (clear_at && clear_at < n) gbuffer_fill(ctx, ctx->defer_clear_fb, ctx->defer_clear_pixels);
	if (ctx->defer_kind == DEFER_PSFPT)    psf_render_passes(ctx, first, n, &ctx->defer_view);
	else if (ctx->defer_kind == DEFER_BPT) bpt_render_passes(ctx, first, n, &ctx->defer_view);
	else                                   render_passes_impl(ctx, first, n, &ctx->defer_view);
	if (clear_at == n) gbuffer_fill(ctx, ctx->defer_clear_fb, ctx->defer_clear_pixels);
}
void defer_pass(fpt_context* ctx, uint32_t kind, uint32_t instance, const fpt_rendering_context_view* view)
{
	// anything but the next instance of the same renderer and view renders what is pending first
	if (ctx->defer_n && (kind != ctx->defer_kind || instance != ctx->defer_first + ctx->defer_n || std::memcmp(view, &ctx->defer_view, sizeof(*view)) != 0)) flush_deferred(ctx);
	if (ctx->defer_n == 0) { ctx->defer_first = instance; ctx->defer_view = *view; }
	ctx->defer_n++;
	if (ctx->defer_n >= ctx->defer_max) flush_deferred(ctx);
}
} // namespace fpt
extern "C" {

int fpt_pt_render(fpt_context* ctx, uint32_t instance, const fpt_rendering_context_view* view)
{
	return guarded(ctx, [&] {
		require(view != nullptr, "fpt_pt_render: null view");
		if (ctx->defer_kind != DEFER_PT || ctx->defer_max <= 1 || ctx->profiling || ctx->capture_bounce >= 0) { flush_deferred(ctx); render_passes_impl(ctx, instance, 1, view); return; }
		defer_pass(ctx, DEFER_PT, instance, view);
	});
}
int fpt_pt_render_batch(fpt_context* ctx, uint32_t first_instance, uint32_t n_passes, const fpt_rendering_context_view* view)
{ return guarded(ctx, [&] { flush_deferred(ctx); render_passes_impl(ctx, first_instance, n_passes, view); }); }
int fpt_pt_set_deferred(fpt_context* ctx, uint32_t max_passes, const fpt_rendering_context_view* view)
{
	return guarded(ctx, [&] {
		flush_deferred(ctx);
		require(max_passes >= 1, "fpt_pt_set_deferred: max_passes must be >= 1");
		if (max_passes > ctx->max_batch) require(fpt_internal_set_batch(ctx, max_passes, view, false) == 0, ctx->error.c_str());
		ctx->defer_max = max_passes; ctx->defer_kind = DEFER_PT;
	});
}
int fpt_pt_flush(fpt_context* ctx) { return guarded(ctx, [&] { flush_deferred(ctx); }); }
int fpt_clear_gbuffer(fpt_context* ctx, const fpt_rendering_context_view* view)
{ return guarded(ctx, [&] { require(view != nullptr, "fpt_clear_gbuffer: null view"); clear_gbuffer(ctx, view); }); }

// sizes the queues, the two albedo planes and the contribution log for max_passes passes in flight (the PSFPT's log has a fourth kind of cell: its blends)
int fpt_internal_set_batch(fpt_context* ctx, uint32_t max_passes, const fpt_rendering_context_view* view, bool for_psfpt)
{
	return guarded(ctx, [&] { flush_deferred(ctx);
		require(ctx->pt_ready, "fpt_pt_set_batch: fpt_pt_init has not been called");
		require(max_passes >= 1, "fpt_pt_set_batch: max_passes must be >= 1");
		require(uint64_t(ctx->n_local) * max_passes < (1ull << 32), "fpt_pt_set_batch: passes x (pixels rendered here) must stay below 2^32 paths in flight");
		FPT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
		for (auto& X : ctx->extra_lanes) FPT_HIP_CHECK(hipStreamSynchronize(X->stream));
		const size_t n = size_t(ctx->n_local) * max_passes;
		ctx->q_a.alloc(n); ctx->q_b.alloc(n); ctx->q_shadow.alloc(n);
		ctx->q_shadow_dir.alloc(view->dir_lights_count ? n : 1);
		const bool planes = max_passes > 1;
		for (int c = 0; c < 6; ++c)
		{
			const bool want = planes && (c == FPT_FB_DIFFUSE_A || c == FPT_FB_SPECULAR_A);
			ctx->d_acc[c].alloc(want ? n * 4 : 0);
			if (ctx->d_acc[c].ptr) FPT_HIP_CHECK(hipMemsetAsync(ctx->d_acc[c].ptr, 0, ctx->d_acc[c].count * sizeof(float), ctx->stream));
		}
		// the contribution log: emission / directional / mesh cells per bounce (+ the PSFPT's blend cells), one fill bit per cell
		const size_t L = ctx->opt.max_path_length;
		ctx->log_mask_words = uint32_t(((for_psfpt ? 4 : 3) * L + 31) / 32);
		ctx->log_emissive.alloc(planes ? n * L : 0);
		ctx->log_nee[0].alloc(planes && view->dir_lights_count ? n * L * 2 : 0);
		ctx->log_nee[1].alloc(planes ? n * L * 2 : 0);
		ctx->log_blend.alloc(planes && for_psfpt ? n * L * 3 : 0);
		ctx->log_mask.alloc(planes ? n * ctx->log_mask_words : 0);
		if (ctx->log_mask.ptr) FPT_HIP_CHECK(hipMemsetAsync(ctx->log_mask.ptr, 0, ctx->log_mask.count * sizeof(uint32_t), ctx->stream));
		ctx->h_fused.clear();                    // the resolve blocks name these buffers
		for (auto& X : ctx->extra_lanes) X->h_fused.clear();
		FPT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
		ctx->max_batch = max_passes;
		if (ctx->defer_kind != DEFER_BPT && ctx->defer_max > max_passes) ctx->defer_max = max_passes;      // a smaller batch than the deferral was sized for
	});
}
int fpt_pt_set_batch(fpt_context* ctx, uint32_t max_passes, const fpt_rendering_context_view* view) { return fpt_internal_set_batch(ctx, max_passes, view, false); }
int fpt_device_memory(fpt_context* ctx, uint64_t* free_bytes, uint64_t* total_bytes)
{
	return guarded(ctx, [&] {
		size_t f = 0, t = 0;
		FPT_HIP_CHECK(hipMemGetInfo(&f, &t));
		if (free_bytes) *free_bytes = f;
		if (total_bytes) *total_bytes = t;
	});
}
int fpt_bytes_per_path_in_flight(fpt_context* ctx, uint32_t renderer, const fpt_rendering_context_view* view, uint64_t* bytes)
{
	return guarded(ctx, [&] {
		require(view && bytes, "fpt_bytes_per_path_in_flight: null argument");
		const uint64_t dir = view->dir_lights_count ? 1 : 0;
		if (renderer == 2)
		{
			// fpt_bpt_set_batch: four ray/weight queues, the light-vertex store (64-B record + position per vertex), connection queue, albedo planes, 20-byte log cells
			const uint64_t L = ctx->bpt.opt.max_path_length ? ctx->bpt.opt.max_path_length : 9, cells = ctx->bpt.opt.single_connection ? 2 : L + 1;
			*bytes = 2 * 84 + 84 + 8 + L * 80 + 32 + L * cells * 20 + 8 * L + 16;
			return;
		}
		const uint64_t L = ctx->opt.max_path_length ? ctx->opt.max_path_length : 9;
		// two path queues (rays 32 incl. PixelInfo and pass offset, hit 16, weight 16, cone 8), the shadow queue(s) (ray 32 incl. PixelInfo, two weights 32 incl. pass offset),
		// two albedo planes, the log: emission 16 + mesh-light 32 (+ directional 32) bytes per bounce, fill bits
		uint64_t b = 2 * 72 + 64 * (1 + dir) + 32 + L * (48 + 32 * dir) + 4 * ((3 * L + 31) / 32);
		if (renderer == 1) b += 2 * 4 + 2 * 4 + L * 48 + (L + 1) * 44 + 4;      // PSFPT: cache-info words, blend cells, the reference queue, (its pass tables are sized per pass, not per path)
		*bytes = b;
	});
}

int fpt_pt_get_stats(fpt_context* ctx, fpt_pt_stats* h_out)
{ return guarded(ctx, [&] { flush_deferred(ctx); require(h_out != nullptr, "fpt_pt_get_stats: null output"); FPT_HIP_CHECK(hipStreamSynchronize(ctx->stream)); *h_out = ctx->stats; }); }
int fpt_pt_set_profiling(fpt_context* ctx, int level)
{
	return guarded(ctx, [&] { flush_deferred(ctx);
		ctx->profiling = (level == 1);
		ctx->profiling_level = level;
		if (level == 2 && ctx->ev_pool.empty())
		{
			ctx->ev_pool.resize(16384);
			for (hipEvent_t& e : ctx->ev_pool) FPT_HIP_CHECK(hipEventCreate(&e));
		}
		ctx->ev_cursor = 0; ctx->timed_launches.clear();
		if (level == 2)
		{
			// the common time base of the render lanes' events
			if (!ctx->ev_ref) FPT_HIP_CHECK(hipEventCreate(&ctx->ev_ref));
			FPT_HIP_CHECK(hipEventRecord(ctx->ev_ref, ctx->stream));
		}
	});
}
int fpt_pt_collect_timings(fpt_context* ctx, float* h_ms, uint32_t* h_launches)
{
	return guarded(ctx, [&] { flush_deferred(ctx);
		FPT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
		for (auto& X : ctx->extra_lanes) FPT_HIP_CHECK(hipStreamSynchronize(X->stream));
		for (int b = 0; b < 5; ++b) { h_ms[b] = 0.0f; h_launches[b] = 0; ctx->last_union_ms[b] = 0.0f; }
		std::vector<std::pair<float, float>> iv[5];
		for (const fpt_context::TimedLaunch& t : ctx->timed_launches)
		{
			float ms = 0.0f;
			FPT_HIP_CHECK(hipEventElapsedTime(&ms, ctx->ev_pool[t.e0], ctx->ev_pool[t.e1]));
			h_ms[t.bucket] += ms; h_launches[t.bucket]++;
			if (ctx->ev_ref)
			{
				float t0 = 0.0f;
				FPT_HIP_CHECK(hipEventElapsedTime(&t0, ctx->ev_ref, ctx->ev_pool[t.e0]));
				iv[t.bucket].push_back(std::make_pair(t0, t0 + ms));
			}
		}
		// per bucket, the time during which at least one of its launches was running: with several render lanes launches overlap, and
		// the sum of their durations is no longer the time the chip spent on them
		iv[4].clear();                                    // slot 4: all traversal launches together (buckets 0, 1, 2)
		for (int b = 0; b < 3; ++b) iv[4].insert(iv[4].end(), iv[b].begin(), iv[b].end());
		for (int b = 0; b < 5; ++b)
		{
			std::sort(iv[b].begin(), iv[b].end());
			float end = -1.0e30f, total = 0.0f;
			for (const auto& x : iv[b]) { if (x.first > end) { total += x.second - x.first; end = x.second; } else if (x.second > end) { total += x.second - end; end = x.second; } }
			ctx->last_union_ms[b] = total;
		}
		ctx->ev_cursor = 0; ctx->timed_launches.clear();
	});
}
// level-2 read-out, launch by launch (in issue order): bucket and duration of up to `cap` launches since the last collect; does not reset anything
int fpt_pt_launch_list(fpt_context* ctx, uint32_t cap, int* h_bucket, float* h_ms, uint32_t* h_count)
{
	return guarded(ctx, [&] { flush_deferred(ctx);
		require(cap == 0 || (h_bucket && h_ms), "fpt_pt_launch_list: null output array");
		FPT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
		uint32_t n = 0;
		for (const fpt_context::TimedLaunch& t : ctx->timed_launches)
		{
			if (n >= cap) break;
			float ms = 0.0f;
			FPT_HIP_CHECK(hipEventSynchronize(ctx->ev_pool[t.e1]));          // a launch of an extra render lane sits on that lane's stream, not on ctx->stream
			FPT_HIP_CHECK(hipEventElapsedTime(&ms, ctx->ev_pool[t.e0], ctx->ev_pool[t.e1]));
			h_bucket[n] = t.bucket; h_ms[n] = ms; ++n;
		}
		if (h_count) *h_count = n;
	});
}
int fpt_pt_last_union_ms(fpt_context* ctx, float* h_ms)
{ return guarded(ctx, [&] { for (int b = 0; b < 5; ++b) h_ms[b] = ctx->last_union_ms[b]; }); }
int fpt_pt_lane_count(fpt_context* ctx) { return ctx ? int(1 + ctx->extra_lanes.size()) : 0; }
int fpt_pt_set_lanes(fpt_context* ctx, uint32_t n_lanes)
{
	return guarded(ctx, [&] { flush_deferred(ctx);
		require(ctx->pt_ready, "fpt_pt_set_lanes: fpt_pt_init has not been called");
		require(n_lanes >= 1 && n_lanes <= 16, "fpt_pt_set_lanes: 1..16 lanes");
		FPT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
		for (auto& X : ctx->extra_lanes) { if (X->stream) { (void)hipStreamSynchronize(X->stream); (void)hipStreamDestroy(X->stream); } if (X->done) (void)hipEventDestroy(X->done); }
		ctx->extra_lanes.clear();
		if (!ctx->lane_start) FPT_HIP_CHECK(hipEventCreateWithFlags(&ctx->lane_start, hipEventDisableTiming));
		for (uint32_t j = 1; j < n_lanes; ++j)
		{
			std::unique_ptr<fpt_context::PtLane> X(new fpt_context::PtLane());
			FPT_HIP_CHECK(hipStreamCreateWithFlags(&X->stream, hipStreamNonBlocking));
			FPT_HIP_CHECK(hipEventCreateWithFlags(&X->done, hipEventDisableTiming));
			X->counters.alloc(CNT_TOTAL);
			ctx->extra_lanes.push_back(std::move(X));
		}
		if (n_lanes > 1 && !ctx->d_pixels && ctx->d_identity.count != ctx->n_local)
		{
			std::vector<uint32_t> id(ctx->n_local);
			for (uint32_t i = 0; i < ctx->n_local; ++i) id[i] = i;
			ctx->d_identity.upload(id.data(), id.size(), ctx->stream);
		}
	});
}
int fpt_pt_set_counting(fpt_context* ctx, int enabled)
{
	return guarded(ctx, [&] { flush_deferred(ctx);
		ctx->counting = enabled != 0;
		FPT_HIP_CHECK(hipMemsetAsync(ctx->d_trace_stats.ptr, 0, 8 * sizeof(unsigned long long), ctx->stream));
	});
}
int fpt_pt_get_trace_counters(fpt_context* ctx, fpt_trace_counters* h_closest, fpt_trace_counters* h_shadow)
{
	return guarded(ctx, [&] { flush_deferred(ctx);
		unsigned long long s[8];
		ctx->d_trace_stats.download(s, 8, ctx->stream);
		if (h_closest) { h_closest->nodes_visited = s[0]; h_closest->tris_tested = s[1]; h_closest->rays = s[2]; }
		if (h_shadow)  { h_shadow->nodes_visited = s[4];  h_shadow->tris_tested = s[5];  h_shadow->rays = s[6]; }
	});
}
int fpt_pt_set_capture(fpt_context* ctx, int bounce) { return guarded(ctx, [&] { flush_deferred(ctx); ctx->capture_bounce = bounce; }); }
int fpt_pt_get_captured(fpt_context* ctx, uint32_t* count, fpt_ray* h_rays, fpt_hit* h_hits, float* h_weights, uint32_t* h_pixel_info, float* h_cones)
{
	return guarded(ctx, [&] { flush_deferred(ctx);
		const uint32_t n = ctx->captured_count;
		if (count) *count = n;
		if (h_rays && n) std::memcpy(h_rays, ctx->cap_rays.data(), size_t(n) * sizeof(fpt_ray));
		if (h_hits && n) std::memcpy(h_hits, ctx->cap_hits.data(), size_t(n) * sizeof(fpt_hit));
		if (h_weights && n) std::memcpy(h_weights, ctx->cap_weights.data(), size_t(n) * 16);
		if (h_pixel_info && n) std::memcpy(h_pixel_info, ctx->cap_pixels.data(), size_t(n) * 4);
		if (h_cones && n) std::memcpy(h_cones, ctx->cap_cones.data(), size_t(n) * 8);
	});
}

// host-side probe of the acceleration-structure builder (no GPU, no context): builds the structure the traversal kernels walk from HOST
// arrays and copies it out, so that tests can check the layout, the encodings and the conservativeness of the boxes independently
int fpt_debug_build_bvh(uint32_t tri_count, const int32_t* h_idx, uint32_t vertex_count, const float* h_vtx, uint32_t* n_nodes, uint32_t* n_records,
                        uint32_t* depth, uint32_t* node_words, uint32_t* h_nodes, float* h_records, fpt_bvh_stats* stats)
{
	try
	{
		HostBvh2 b;
		build_acceleration(tri_count, h_idx, vertex_count, h_vtx, b, trace_stack_entries());
		if (stats) fill_bvh_stats(b, stats);
		if (n_nodes) *n_nodes = uint32_t(b.nodes8.size());
		if (n_records) *n_records = uint32_t(b.tris8.size());
		if (depth) *depth = b.wide_depth;
		if (node_words) *node_words = uint32_t(sizeof(BvhNode8) / 4);
		if (h_nodes && !b.nodes8.empty()) std::memcpy(h_nodes, b.nodes8.data(), b.nodes8.size() * sizeof(BvhNode8));
		if (h_records && !b.tris8.empty()) std::memcpy(h_records, b.tris8.data(), b.tris8.size() * sizeof(BvhTriangle));
		return 0;
	}
	catch (const std::exception& e) { g_create_error = e.what(); return 1; }
}

// the emitter-table builder without a context (CPU tests against the oracle, tools/time_emitters.py)
int fpt_debug_build_emitter_tables(uint32_t n_vpls, const fpt_mesh_view* h_mesh, const fpt_texture* h_textures, uint32_t instance, fpt_vpl* h_vpls,
                                   float* h_vpl_cdf, float* h_mesh_cdf, float* h_mesh_inv_area, float* norm, uint32_t* n_out)
{
	try
	{
		if (!h_mesh) throw std::runtime_error("fpt_debug_build_emitter_tables: null mesh");
		EmitterTables e;
		build_emitter_tables(n_vpls, *h_mesh, h_textures, instance, e);
		if (n_out) *n_out = uint32_t(e.vpls.size());
		if (h_vpls && !e.vpls.empty()) std::memcpy(h_vpls, e.vpls.data(), e.vpls.size() * sizeof(fpt_vpl));
		if (h_vpl_cdf && !e.vpl_cdf.empty()) std::memcpy(h_vpl_cdf, e.vpl_cdf.data(), e.vpl_cdf.size() * sizeof(float));
		if (h_mesh_cdf && !e.mesh_cdf.empty()) std::memcpy(h_mesh_cdf, e.mesh_cdf.data(), e.mesh_cdf.size() * sizeof(float));
		if (h_mesh_inv_area && !e.mesh_inv_area.empty()) std::memcpy(h_mesh_inv_area, e.mesh_inv_area.data(), e.mesh_inv_area.size() * sizeof(float));
		if (norm) *norm = e.norm;
		return 0;
	}
	catch (const std::exception& e) { g_create_error = e.what(); return 1; }
}

// the same probe for the refit: builds over h_vtx0, refits to h_vtx1 (same indices) and copies the refitted structure out
int fpt_debug_refit_bvh(uint32_t tri_count, const int32_t* h_idx, uint32_t vertex_count, const float* h_vtx0, const float* h_vtx1, uint32_t* n_nodes, uint32_t* n_records,
                        uint32_t* depth, uint32_t* h_nodes, float* h_records, fpt_bvh_stats* stats)
{
	try
	{
		HostBvh2 b;
		build_acceleration(tri_count, h_idx, vertex_count, h_vtx0, b, trace_stack_entries());
		refit_wide8(tri_count, h_idx, vertex_count, h_vtx1, b);
		if (stats) fill_bvh_stats(b, stats);
		if (n_nodes) *n_nodes = uint32_t(b.nodes8.size());
		if (n_records) *n_records = uint32_t(b.tris8.size());
		if (depth) *depth = b.wide_depth;
		if (h_nodes && !b.nodes8.empty()) std::memcpy(h_nodes, b.nodes8.data(), b.nodes8.size() * sizeof(BvhNode8));
		if (h_records && !b.tris8.empty()) std::memcpy(h_records, b.tris8.data(), b.tris8.size() * sizeof(BvhTriangle));
		return 0;
	}
	catch (const std::exception& e) { g_create_error = e.what(); return 1; }
}

int fpt_debug_math(fpt_context* ctx, int op, uint32_t n, const float* d_in0, const float* d_in1, float* d_out0, float* d_out1)
{ return guarded(ctx, [&] { launch_debug_math(op, n, d_in0, d_in1, d_out0, d_out1, ctx->stream); FPT_HIP_CHECK(hipGetLastError()); FPT_HIP_CHECK(hipStreamSynchronize(ctx->stream)); }); }

} // extern "C"
