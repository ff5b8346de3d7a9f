// fpt_bpt_api.cpp — C-ABI of the bidirectional path tracer: BPT::init / BPT::render (src/renderers/bpt.cu:44-104,
// src/renderers/bpt_impl.h:198-258) with the control flow of src/bpt_control.h:290-600.
//
// As in the path tracer nothing is read back inside a pass: queue sizes stay in device memory (one pre-zeroed counter per
// queue per bounce), every kernel bounds itself by them, and the launches of a pass are enqueued on one stream:
//   light sub-paths : primary, then per bounce { closest-hit trace, light_vertices }
//   eye sub-paths   : primary, then per bounce { closest-hit trace, eye_vertices, any-hit trace of the connection rays, eye_resolve }
//   light tracing   : connect_camera, any-hit trace, splat, splat_resolve
// The connection rays are occlusion queries (the reference sends them through RTContext::trace and only tests hit.t < 0,
// src/bpt_control.h:352-382 + src/bpt_kernels.h:899-916): they run on the any-hit kernel with an empty triangle mask.
#include "fpt_host.h"
#include <algorithm>
#include <cstring>

using namespace fpt;

namespace {

// counters: [0, TICKETS) trace ticket dispensers (one 128-byte-strided group per launch), then one word per queue per bounce
enum { B_TICKET_STRIDE = 8 * 32, B_MAX_LAUNCHES = 3 * 34, B_QUEUES = B_TICKET_STRIDE * B_MAX_LAUNCHES, B_PER_BOUNCE = 64, B_LIGHT = 0, B_EYE = 32,
       B_SHADOW_BASE = B_QUEUES + B_PER_BOUNCE * 35, B_TOTAL = B_SHADOW_BASE + 32 * 36 };

BptQueue queue_view(fpt_context::BptState& b, int which, uint32_t* size)
{
	BptQueue q; q.rays = b.q_rays[which].ptr; q.hits = b.q_hits[which].ptr; q.weights = b.q_weights[which].ptr; q.path_weights = b.q_pw[which].ptr; q.pixels = b.q_pixels[which].ptr; q.chan = b.q_chan[which].ptr; q.size = size;
	return q;
}

uint32_t read_u32(fpt_context* ctx, const uint32_t* d)
{
	uint32_t v = 0;
	FPT_HIP_CHECK(hipMemcpyAsync(&v, d, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
	FPT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
	return v;
}

FrameBufferDev plane_view(fpt_context::BptState& b, const FrameBufferDev& real)
{
	// passes in flight: only the two albedo channels keep a plane per pass (one term per pass and pixel); every other term goes to the eye path's cell of the log
	FrameBufferDev fb = real;
	for (int c = 0; c < 6; ++c) fb.ch[c] = nullptr;
	fb.ch[FPT_FB_DIFFUSE_A] = b.acc[FPT_FB_DIFFUSE_A].ptr; fb.ch[FPT_FB_SPECULAR_A] = b.acc[FPT_FB_SPECULAR_A].ptr;
	return fb;
}
BptLog log_view(fpt_context::BptState& b)
{
	BptLog g; g.val = b.log_val.ptr; g.chan = b.log_chan.ptr; g.mask = b.log_mask.ptr;
	g.cap = uint32_t(size_t(b.n_paths) * b.max_batch); g.conn_cells = b.opt.single_connection ? 1u : b.opt.max_path_length;
	g.mask_words = (b.opt.max_path_length * (1u + g.conn_cells) + 31u) / 32u;
	return g;
}

// fold the light-tracing splat sums into the frame (one pass: straight into the frame buffer); a batch: the merge applies, pass by pass and in the
// sequential order, the albedo planes, the eye paths' cells and the splat sums
void resolve_splats(fpt_context* ctx, const fpt_rendering_context_view* view)
{
	fpt_context::BptState& b = ctx->bpt;
	const uint32_t n_passes = b.pending_n > 1 ? b.pending_n : 1u;
	const FrameBufferDev real = fb_dev(view->fb);
	if (n_passes > 1)
	{
		launch_bpt_merge_exact(real, b.acc[FPT_FB_DIFFUSE_A].ptr, b.acc[FPT_FB_SPECULAR_A].ptr, log_view(b), b.splat_ptr(), b.d_pixels, b.n_local, b.n_paths, b.pending_first, n_passes,
		                       b.opt.max_path_length, ctx->stream);
		// the sums of every pixel of the batch, this rank's or not (all-reduced sums cover the whole frame)
		FPT_HIP_CHECK(hipMemsetAsync(b.splat_ptr(), 0, size_t(b.n_paths) * n_passes * 3 * sizeof(long long), ctx->stream));
	}
	else
	{
		BptParams P; std::memset(&P, 0, sizeof(P));
		P.fb = real; P.splat = b.splat_ptr(); P.res_x = view->res_x; P.res_y = view->res_y;
		P.n_paths = b.n_paths; P.n_passes = 1; P.plane_stride = 0u; P.instance = b.pending_first;
		launch_bpt_splat_resolve(P, ctx->stream);
	}
	b.pending_n = 0;
	FPT_HIP_CHECK(hipGetLastError());
}

// device storage for `passes` passes in flight
void alloc_storage(fpt_context* ctx, uint32_t passes)
{
	fpt_context::BptState& b = ctx->bpt;
	const uint32_t L = b.opt.max_path_length;
	const size_t nl = size_t(b.n_local) * passes, np = size_t(b.n_paths) * passes;
	for (int k = 0; k < 2; ++k)
	{
		b.q_rays[k].alloc(nl * 2); b.q_hits[k].alloc(nl); b.q_weights[k].alloc(nl); b.q_pw[k].alloc(nl); b.q_pixels[k].alloc(nl); b.q_chan[k].alloc(nl);
	}
	const size_t n_shadow = nl * L;           // an eye vertex connects to at most L light vertices; a light path splats at most L-1
	b.s_rays.alloc(n_shadow * 2); b.s_hits.alloc(n_shadow); b.s_weights.alloc(n_shadow); b.s_pixels.alloc(n_shadow); b.s_chan.alloc(n_shadow); b.conn.alloc(nl);
	const size_t nv = np * L;
	b.v_pos.alloc(nv); b.v_rec.alloc(nv); b.v_counts.alloc(np);
	if (b.opt.single_connection)
	{
		// the flat vertex list: at most one entry per store slot; per-block counts of the scan (4096 elements per block); per-pass bounds
		b.flat.alloc(nv); b.flat_block_sums.alloc((nv + 4095) / 4096 + 1); b.flat_meta.alloc(2 * size_t(passes) + 2);
	}
	FPT_HIP_CHECK(hipMemsetAsync(b.v_counts.ptr, 0, np * sizeof(uint32_t), ctx->stream));
	b.splat.alloc(np * 3);
	FPT_HIP_CHECK(hipMemsetAsync(b.splat.ptr, 0, np * 3 * sizeof(long long), ctx->stream));
	for (int c = 0; c < 6; ++c)
	{
		const bool want = passes > 1 && (c == FPT_FB_DIFFUSE_A || c == FPT_FB_SPECULAR_A);
		b.acc[c].alloc(want ? np : 0);
		if (want) FPT_HIP_CHECK(hipMemsetAsync(b.acc[c].ptr, 0, np * sizeof(float4), ctx->stream));
	}
	// the eye paths' contribution log (BptLog, fpt_bpt.h): per bounce an emission cell + 1 (-sc 1) or L (-sc 0) connection cells
	{
		const size_t cells = size_t(L) * (1u + (b.opt.single_connection ? 1u : L)), words = (cells + 31) / 32;
		b.log_val.alloc(passes > 1 ? np * cells : 0); b.log_chan.alloc(passes > 1 ? np * cells : 0); b.log_mask.alloc(passes > 1 ? np * words : 0);
		if (passes > 1) FPT_HIP_CHECK(hipMemsetAsync(b.log_mask.ptr, 0, np * words * sizeof(uint32_t), ctx->stream));
	}
	b.max_batch = passes;
}

} // namespace

extern "C" {

int fpt_bpt_init(fpt_context* ctx, const fpt_bpt_options* opts, const fpt_rendering_context_view* view, const char* h_samples_dir,
                 const uint32_t* d_pixels, uint32_t n_local_pixels)
{
	return guarded(ctx, [&] {
		flush_deferred(ctx);
		require(opts && view, "fpt_bpt_init: null argument");
		require(opts->max_path_length >= 1 && opts->max_path_length <= 15, "fpt_bpt_init: max_path_length out of range [1,15] (the s,t technique ids are 4-bit)");
		require(uint64_t(view->res_x) * view->res_y < (1ull << 24), "fpt_bpt_init: light-vertex ids hold 24-bit path indices");
		require(ctx->has_emitters, "fpt_bpt_init: fpt_mesh_lights_init has not been called");
		require(!opts->use_vpls || !ctx->emitters.vpls.empty(), "fpt_bpt_init: -use-vpls needs a VPL set");
		// light_primary_kernel reads vpls[pixel index]: the reference allocates one VPL per light path by construction (src/renderers/bpt.cu:53)
		require(!opts->use_vpls || ctx->emitters.vpls.size() >= size_t(view->res_x) * view->res_y, "fpt_bpt_init: -use-vpls needs one VPL per pixel (fpt_mesh_lights_init with n_vpls >= res_x * res_y)");
		fpt_context::BptState& b = ctx->bpt;
		b.opt = *opts;
		b.n_paths = view->res_x * view->res_y;
		b.n_local = d_pixels ? n_local_pixels : b.n_paths;
		b.d_pixels = d_pixels;
		require(b.n_local > 0, "fpt_bpt_init: empty pixel set");
		// compensate for the amount of light vs eye sub-paths (src/renderers/bpt.cu:75): both equal the pixel count here
		b.light_tracing = opts->light_tracing * (float(b.n_paths) / float(b.n_paths));
		const uint32_t L = opts->max_path_length;
		alloc_storage(ctx, 1);
		if (ctx->defer_kind == DEFER_BPT) ctx->defer_max = 1;          // storage for one pass: fpt_bpt_set_deferred sizes it again
		b.counters.alloc(B_TOTAL);
		// sampler: (L+1)*2*6 dimensions (src/renderers/bpt.cu:83-87); consumes the context's rand() stream after whatever ran before
		std::vector<float> shifts;
		b.seq_dims = (L + 1) * 2 * 6;
		build_shift_table(256, b.seq_dims, h_samples_dir, ctx->crt_rand, shifts);
		b.d_shifts.upload(shifts.data(), shifts.size(), ctx->stream);
		b.ready = true;
	});
}

int fpt_bpt_set_profiling(fpt_context* ctx, int on) { return guarded(ctx, [&] { flush_deferred(ctx); ctx->bpt.profiling = on != 0; }); }
int fpt_bpt_get_stats(fpt_context* ctx, fpt_bpt_stats* out) { return guarded(ctx, [&] { flush_deferred(ctx); require(out != nullptr, "fpt_bpt_get_stats: null"); *out = ctx->bpt.stats; }); }
int64_t* fpt_bpt_splat_buffer(fpt_context* ctx) { return ctx ? reinterpret_cast<int64_t*>(ctx->bpt.splat_ptr()) : nullptr; }
int fpt_bpt_use_splat_buffer(fpt_context* ctx, int64_t* d_splats) { return guarded(ctx, [&] { ctx->bpt.splat_external = reinterpret_cast<long long*>(d_splats); }); }
int fpt_bpt_set_deferred_splats(fpt_context* ctx, int deferred) { return guarded(ctx, [&] { flush_deferred(ctx); ctx->bpt.deferred_splats = deferred != 0; }); }
int fpt_bpt_resolve_splats(fpt_context* ctx, const fpt_rendering_context_view* view)
{ return guarded(ctx, [&] { flush_deferred(ctx); require(ctx->bpt.ready, "fpt_bpt_resolve_splats: fpt_bpt_init has not been called"); resolve_splats(ctx, view); }); }

int fpt_bpt_download_light_vertices(fpt_context* ctx, float* h_pos, uint32_t* h_input, uint32_t* h_gbuffer, float* h_weights, uint32_t* h_path_id, uint32_t* h_counts)
{
	return guarded(ctx, [&] {
		fpt_context::BptState& b = ctx->bpt;
		flush_deferred(ctx);
		require(b.ready, "fpt_bpt_download_light_vertices: fpt_bpt_init has not been called");
		const size_t nv = size_t(b.n_paths) * b.opt.max_path_length;
		FPT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
		std::vector<LightVertexRecord> rec(nv);          // the store is an array of 64-byte records on the device; the caller gets the reference's five arrays
		if (nv) FPT_HIP_CHECK(hipMemcpy(rec.data(), b.v_rec.ptr, nv * sizeof(LightVertexRecord), hipMemcpyDeviceToHost));
		for (size_t i = 0; i < nv; ++i)
		{
			const LightVertexRecord& r = rec[i];
			if (h_pos) std::memcpy(h_pos + 4 * i, &r.pos, 16);
			if (h_input) std::memcpy(h_input + 2 * i, &r.input, 8);
			if (h_gbuffer) std::memcpy(h_gbuffer + 4 * i, &r.gbuffer, 16);
			if (h_weights) std::memcpy(h_weights + 2 * i, &r.weights, 8);
			if (h_path_id) h_path_id[i] = r.path_id;
		}
		if (h_counts) FPT_HIP_CHECK(hipMemcpy(h_counts, b.v_counts.ptr, size_t(b.n_paths) * 4, hipMemcpyDeviceToHost));
	});
}

// One render call = the light sub-paths, then everything that reads the light-vertex store (the -sc 1 vertex list, the eye sub-paths and their
// connections, light tracing).  A tile-sharded -sc 1 context with shared light vertices (fpt_bpt_set_shared_light_vertices) stops between the two, so
// that the ranks can hand each other the vertices of their light paths (fpt_bpt_exchange_light_vertices, or export / import) before fpt_bpt_finish.
struct BptRun
{
	fpt_context* ctx; fpt_context::BptState& b; BptParams P; uint32_t* cnt; hipStream_t s; uint32_t n_launch, L; bool prof;
	BptRun(fpt_context* c, uint32_t instance, uint32_t n_passes, const fpt_rendering_context_view* view) : ctx(c), b(c->bpt)
	{
		const bool batched = n_passes > 1;
		n_launch = b.n_local * n_passes; s = ctx->stream; L = b.opt.max_path_length; cnt = b.counters.ptr; prof = b.profiling;
		std::memset(&P, 0, sizeof(P));
		P.store.rec = b.v_rec.ptr; P.store.pos = b.v_pos.ptr; P.store.counts = b.v_counts.ptr;
		P.conn = b.conn.ptr; P.splat = b.splat_ptr();
		P.flat = b.flat.ptr; P.flat_meta = b.flat_meta.ptr; P.flat_block_sums = b.flat_block_sums.ptr;
		P.seq.shifts = b.d_shifts.ptr; P.seq.n_dims = b.seq_dims; P.seq.tile_size = 256;
		P.mesh = view->mesh; P.textures = view->d_textures; P.table = view->d_glossy_reflectance; P.shade_records = ensure_shade_records(ctx, view, s);
		EmitterView em;
		em.n_prims = uint32_t(ctx->emitters.mesh_cdf.size()); em.prims_cdf = ctx->d_mesh_cdf.ptr; em.prims_inv_area = ctx->d_mesh_inv_area.ptr;
		em.n_vpls = b.opt.use_vpls ? uint32_t(ctx->emitters.vpls.size()) : 0u; em.vpls = b.opt.use_vpls ? ctx->d_vpls.ptr : nullptr; em.norm = ctx->emitters.norm; em.vpl_points = nullptr;
		P.emitters = em;
		const FrameBufferDev real_fb = fb_dev(view->fb);
		P.fb = batched ? plane_view(b, real_fb) : real_fb;
		if (batched) P.log = log_view(b);
		P.opt = b.opt; P.pixels = b.d_pixels; P.n_local = b.n_local; P.n_paths = b.n_paths;
		P.res_x = view->res_x; P.res_y = view->res_y; P.instance = instance;
		P.n_passes = n_passes; P.n_store = b.n_paths * n_passes; P.plane_stride = batched ? b.n_paths : 0u;
		P.light_tracing = b.light_tracing;
		P.eye = mk3(view->camera.eye[0], view->camera.eye[1], view->camera.eye[2]);
		camera_frame(view->camera, view->aspect, P.U, P.V, P.W);
		P.W_len = length(P.W);
		{ const float tn = tanf(view->camera.fov / 2); P.sq_focal = (1.0f / 4.0f) / (tn * tn); }           // Camera::square_screen_focal_length, src/camera.h:132-136
	}
	void trace(const float4* rays, float4* hits, const uint32_t* count_ptr, bool any_hit)
	{
		TraceParams tp = base_trace_params(ctx);
		tp.rays = rays; tp.hits = hits; tp.count_ptr = count_ptr; tp.work_counter = cnt + B_TICKET_STRIDE * (b.ticket++); tp.stats = ctx->d_trace_stats.ptr;
		if (any_hit) timed_launch(ctx, 2, s, [&] { launch_trace_shadow(tp, false, ctx->counting, ctx->trace_blocks(), s); });
		else         timed_launch(ctx, 0, s, [&] { launch_trace_closest(tp, ctx->counting, ctx->trace_blocks(), s); });
	}
	uint32_t* qcount(uint32_t bounce, uint32_t which) { return cnt + B_QUEUES + B_PER_BOUNCE * bounce + which; }

	// ---- sample_light_subpaths (src/bpt_control.h:290-350) ----
	void light_phase()
	{
		fpt_bpt_stats& st = b.stats;
		if (prof) std::memset(&st, 0, sizeof(st));
		b.ticket = 0;
		FPT_HIP_CHECK(hipMemsetAsync(cnt, 0, B_TOTAL * sizeof(uint32_t), s));
		// shared light vertices: the store holds the other ranks' vertices of the previous batch -- forget them
		if (b.shared_lv) FPT_HIP_CHECK(hipMemsetAsync(b.v_counts.ptr, 0, size_t(P.n_store) * sizeof(uint32_t), s));
		int cur = 0;
		P.out = queue_view(b, cur, qcount(0, B_LIGHT));
		launch_bpt_light_primary(P, s);
		for (uint32_t bounce = 0; bounce + 1 < L; ++bounce)
		{
			P.bounce = bounce;
			P.in = queue_view(b, cur, qcount(bounce, B_LIGHT));
			P.out = queue_view(b, cur ^ 1, qcount(bounce + 1, B_LIGHT));
			trace(P.in.rays, P.in.hits, P.in.size, false);
			timed_launch(ctx, 3, s, [&] { launch_bpt_light_vertices(P, n_launch, s); });
			if (prof) { st.light_queue[bounce] = read_u32(ctx, P.in.size); if (st.light_queue[bounce]) st.n_bounces_light = bounce + 1; }
			cur ^= 1;
		}
	}
	void rest(const fpt_rendering_context_view* view)
	{
		fpt_bpt_stats& st = b.stats;
		// -sc 1: the eye vertices draw from the list of ALL light vertices of their pass (VertexOrdering::kRandomOrdering)
		if (b.opt.single_connection) launch_bpt_build_flat_list(P, s);
		// ---- sample_eye_subpaths (src/bpt_control.h:384-470) ----
		int cur = 0;
		P.out = queue_view(b, cur, qcount(0, B_EYE));
		launch_bpt_eye_primary(P, s);
		BptParams P_prev = P;
		for (uint32_t bounce = 0; bounce < L; ++bounce)
		{
			P.bounce = bounce;
			P.in = queue_view(b, cur, qcount(bounce, B_EYE));
			P.out = queue_view(b, cur ^ 1, qcount(bounce + 1, B_EYE));
			P.shadow.rays = b.s_rays.ptr; P.shadow.hits = b.s_hits.ptr; P.shadow.weights = b.s_weights.ptr; P.shadow.pixels = b.s_pixels.ptr; P.shadow.chan = b.s_chan.ptr;
			P.shadow.size = cnt + B_SHADOW_BASE + 32 * bounce;
			// the connections of bounce b-1 ride in the launch that finds the hits of bounce b (one traversal launch per bounce instead of two: a
			// launch cannot end before its longest ray); they are added -- by the previous bounce's parameter block -- before this bounce's
			// vertices touch the frame or reuse the connection queue, i.e. in the order of the unfused sequence
			if (bounce == 0) trace(P.in.rays, P.in.hits, P.in.size, false);
			else
			{
				TraceParams tp = base_trace_params(ctx);
				tp.rays = P.in.rays; tp.hits = P.in.hits; tp.count_ptr = P.in.size; tp.work_counter = cnt + B_TICKET_STRIDE * (b.ticket++); tp.stats = ctx->d_trace_stats.ptr;
				tp.shadow_rays = P_prev.shadow.rays; tp.shadow_size = P_prev.shadow.size;
				timed_launch(ctx, 0, s, [&] { launch_trace_mixed_hits(tp, P_prev.shadow.hits, ctx->counting, ctx->trace_blocks(), s); });
				timed_launch(ctx, 3, s, [&] { launch_bpt_eye_resolve(P_prev, n_launch, s); });
			}
			timed_launch(ctx, 3, s, [&] { launch_bpt_eye_vertices(P, n_launch, s); });
			if (bounce + 1 == L)
			{
				trace(P.shadow.rays, P.shadow.hits, P.shadow.size, true);
				timed_launch(ctx, 3, s, [&] { launch_bpt_eye_resolve(P, n_launch, s); });
			}
			P_prev = P;
			if (prof)
			{
				st.eye_queue[bounce] = read_u32(ctx, P.in.size); st.shadow_eye[bounce] = read_u32(ctx, P.shadow.size);
				if (st.eye_queue[bounce]) st.n_bounces_eye = bounce + 1;
			}
			cur ^= 1;
		}
		// ---- light_tracing (src/bpt_control.h:572-600): this rank's own light paths ----
		if (b.light_tracing)
		{
			P.shadow.size = cnt + B_SHADOW_BASE + 32 * L;
			timed_launch(ctx, 3, s, [&] { launch_bpt_connect_camera(P, s); });
			trace(P.shadow.rays, P.shadow.hits, P.shadow.size, true);
			launch_bpt_splat(P, uint32_t(std::min<size_t>(size_t(n_launch) * (L > 1 ? L - 1 : 1), 0xFFFFFFFFu)), s);
			if (prof) st.shadow_light_tracing = read_u32(ctx, P.shadow.size);
		}
		if (prof)
		{
			std::vector<uint32_t> counts(size_t(b.n_paths) * P.n_passes);
			FPT_HIP_CHECK(hipMemcpyAsync(counts.data(), b.v_counts.ptr, counts.size() * 4, hipMemcpyDeviceToHost, s));
			FPT_HIP_CHECK(hipStreamSynchronize(s));
			uint64_t total = 0; for (uint32_t c : counts) total += c;
			st.n_light_vertices = uint32_t(total);
		}
		// the frame: light-tracing splats, then (batch) the planes in pass order.  Deferred mode leaves both to fpt_bpt_resolve_splats
		b.pending_first = P.instance; b.pending_n = P.n_passes;
		if (!b.deferred_splats || !b.light_tracing) resolve_splats(ctx, view);
		FPT_HIP_CHECK(hipGetLastError());
	}
};

static void render_impl(fpt_context* ctx, uint32_t instance, uint32_t n_passes, const fpt_rendering_context_view* view)
{
	fpt_context::BptState& b = ctx->bpt;
	require(b.ready, "fpt_bpt_render: fpt_bpt_init has not been called");
	require(ctx->has_geometry, "fpt_bpt_render: create_geometry has not been called");
	require(view->res_x * view->res_y == b.n_paths, "fpt_bpt_render: the view's resolution differs from fpt_bpt_init's");
	require(n_passes >= 1 && n_passes <= b.max_batch, "fpt_bpt_render_batch: more passes than fpt_bpt_set_batch sized the storage for");
	require(b.pending_n == 0, "fpt_bpt_render: the previous batch's splats have not been resolved (fpt_bpt_resolve_splats)");
	require(!b.light_pending, "fpt_bpt_render: the previous batch waits for fpt_bpt_finish (shared light vertices)");
	// renderer.multiply_frame(instance / (instance + 1)) over this rank's pixels; a batch scales pass by pass when its planes are merged
	if (n_passes == 1) launch_rescale(fb_dev(view->fb), b.d_pixels, b.n_local, float(instance) / float(instance + 1), ctx->stream);
	BptRun run(ctx, instance, n_passes, view);
	run.light_phase();
	if (b.shared_lv) { b.light_pending = true; b.light_instance = instance; b.light_passes = n_passes; FPT_HIP_CHECK(hipGetLastError()); return; }
	run.rest(view);
}

} // extern "C"
namespace fpt { void bpt_render_passes(fpt_context* ctx, uint32_t first, uint32_t n, const fpt_rendering_context_view* view) { render_impl(ctx, first, n, view); } }
extern "C" {

int fpt_bpt_render(fpt_context* ctx, uint32_t instance, const fpt_rendering_context_view* view)
{
	return guarded(ctx, [&] {
		require(view != nullptr, "fpt_bpt_render: null view");
		fpt_context::BptState& b = ctx->bpt;
		// deferred (fpt_bpt_set_deferred) unless the caller steps in between the phases of a pass: splat sums to all-reduce, light vertices to exchange
		if (ctx->defer_kind != DEFER_BPT || ctx->defer_max <= 1 || b.profiling || b.shared_lv || (b.deferred_splats && b.light_tracing)) { flush_deferred(ctx); render_impl(ctx, instance, 1, view); return; }
		defer_pass(ctx, DEFER_BPT, instance, view);
	});
}
/* deferred fpt_bpt_render: as fpt_pt_set_deferred (the BPT's passes in flight are bit-identical to sequential passes) */
int fpt_bpt_set_deferred(fpt_context* ctx, uint32_t max_passes)
{
	{ const int st = guarded(ctx, [&] { flush_deferred(ctx); require(max_passes >= 1, "fpt_bpt_set_deferred: max_passes must be >= 1"); }); if (st != 0) return st; }
	if (max_passes > ctx->bpt.max_batch) { const int st = fpt_bpt_set_batch(ctx, max_passes); if (st != 0) return st; }
	return guarded(ctx, [&] { ctx->defer_max = max_passes; ctx->defer_kind = DEFER_BPT; });
}

/* passes in flight: instance .. instance + n_passes - 1 as ONE wavefront (storage from fpt_bpt_set_batch) */
int fpt_bpt_set_batch(fpt_context* ctx, uint32_t max_passes)
{
	return guarded(ctx, [&] {
		fpt_context::BptState& b = ctx->bpt;
		flush_deferred(ctx);
		require(b.ready, "fpt_bpt_set_batch: fpt_bpt_init has not been called");
		// virtual path ids and light-vertex slots (virtual id + depth x passes x pixels) are 32-bit words: the bound is memory long before it is this
		require(max_passes >= 1 && uint64_t(max_passes) * b.n_paths * b.opt.max_path_length < (1ull << 32),
		        "fpt_bpt_set_batch: passes x pixels x max_path_length must stay below 2^32 (light-vertex slots are 32-bit)");
		require(b.pending_n == 0, "fpt_bpt_set_batch: a batch is waiting for fpt_bpt_resolve_splats");
		FPT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
		alloc_storage(ctx, max_passes);
		if (ctx->defer_kind == DEFER_BPT && ctx->defer_max > max_passes) ctx->defer_max = max_passes;
	});
}
int fpt_bpt_render_batch(fpt_context* ctx, uint32_t first_instance, uint32_t n_passes, const fpt_rendering_context_view* view)
{ return guarded(ctx, [&] { flush_deferred(ctx); render_impl(ctx, first_instance, n_passes, view); }); }

/* -sc 1 under tile sharding with the SAME image for any number of ranks: the connection vertices are drawn from the light vertices of ALL light paths,
 * so every rank needs every rank's.  With shared light vertices on, fpt_bpt_render / fpt_bpt_render_batch stop after the light sub-paths; the ranks
 * exchange their vertices (80-byte records: store slot + the 64-byte vertex); fpt_bpt_finish builds the vertex list over all paths and goes on. */
int fpt_bpt_set_shared_light_vertices(fpt_context* ctx, int on)
{
	return guarded(ctx, [&] {
		flush_deferred(ctx);
		require(ctx->bpt.ready && !ctx->bpt.light_pending, "fpt_bpt_set_shared_light_vertices: fpt_bpt_init first; no batch may be waiting for fpt_bpt_finish");
		ctx->bpt.shared_lv = on != 0;
	});
}
} // extern "C"
namespace fpt {
uint32_t bpt_pack_own_vertices(fpt_context* ctx)
{
	fpt_context::BptState& b = ctx->bpt;
	require(b.light_pending, "no light sub-paths are waiting (fpt_bpt_set_shared_light_vertices, then fpt_bpt_render)");
	const size_t cap = size_t(b.n_local) * b.light_passes * b.opt.max_path_length;
	b.lv_send.alloc(std::max<size_t>(b.lv_send.count, cap));
	b.lv_count.alloc(32);
	FPT_HIP_CHECK(hipMemsetAsync(b.lv_count.ptr, 0, 32 * sizeof(uint32_t), ctx->stream));
	launch_bpt_pack_light_vertices(b.v_rec.ptr, b.v_counts.ptr, b.d_pixels, b.n_local, b.n_paths, b.light_passes, b.lv_send.ptr, b.lv_count.ptr, ctx->stream);
	return read_u32(ctx, b.lv_count.ptr);
}
void bpt_import_vertices(fpt_context* ctx, const LightVertexWire* d_records, uint32_t count)
{
	fpt_context::BptState& b = ctx->bpt;
	require(b.light_pending, "no light sub-paths are waiting (fpt_bpt_set_shared_light_vertices, then fpt_bpt_render)");
	if (count) launch_bpt_unpack_light_vertices(d_records, count, b.v_rec.ptr, b.v_pos.ptr, b.v_counts.ptr, b.n_paths * b.light_passes, ctx->stream);
	FPT_HIP_CHECK(hipGetLastError());
}
} // namespace fpt
extern "C" {
/* this rank's vertices of the batch in flight as FPT_BPT_VERTEX_RECORD_BYTES-byte records in device memory owned by the library (valid until the next
 * render call); a host that moves the records itself hands them to the other ranks' fpt_bpt_import_light_vertices */
int fpt_bpt_export_light_vertices(fpt_context* ctx, const void** d_records, uint32_t* count)
{
	return guarded(ctx, [&] {
		const uint32_t n = bpt_pack_own_vertices(ctx);
		if (d_records) *d_records = ctx->bpt.lv_send.ptr;
		if (count) *count = n;
	});
}
int fpt_bpt_import_light_vertices(fpt_context* ctx, const void* d_records, uint32_t count)
{ return guarded(ctx, [&] { bpt_import_vertices(ctx, static_cast<const LightVertexWire*>(d_records), count); }); }
int fpt_bpt_finish(fpt_context* ctx, const fpt_rendering_context_view* view)
{
	return guarded(ctx, [&] {
		fpt_context::BptState& b = ctx->bpt;
		require(b.light_pending, "fpt_bpt_finish: no light sub-paths are waiting");
		BptRun run(ctx, b.light_instance, b.light_passes, view);
		b.light_pending = false;
		run.rest(view);
	});
}

} // extern "C"
