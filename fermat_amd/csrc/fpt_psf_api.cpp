// fpt_psf_api.cpp — C-ABI of the path-space-filtering path tracer: PSFPT::init / PSFPT::render (src/renderers/psfpt_impl.h:184-420).
// The pass is the path tracer's loop with the PSFPT vertex processor (shade_kernel<true>), a persistent hash table of cache cells,
// a reference queue and the final blending / clamp kernels.  Shadow rays are traced unfused (any-hit kernel writing hit records) and
// resolved by psf_resolve_kernel, because an unoccluded sample goes to a cache cell, to the frame buffer, or to both.
#include "fpt_host.h"
#include <algorithm>
#include <cstring>

using namespace fpt;

namespace {
enum { P_TICKET_STRIDE = 8 * 32, P_MAX_LAUNCHES = 4 * 34, P_QUEUES = P_TICKET_STRIDE * P_MAX_LAUNCHES, P_PER_BOUNCE = 96, P_PATH = 0, P_SHADOW_DIR = 32, P_SHADOW = 64,
       P_TOTAL = P_QUEUES + P_PER_BOUNCE * 34 };
}

namespace {
// the cache as the kernels of a SHARDED context see it: pass table in keys / cells, global table in g_keys / g_cells
PsfDev sharded_view(fpt_context* ctx)
{
	fpt_context::PsfState& ps = ctx->psf;
	PsfDev psf; std::memset(&psf, 0, sizeof(psf));
	psf.keys = ps.p_keys.ptr; psf.cells = ps.p_cells.ptr; psf.log2_size = ps.log2_size;
	psf.g_keys = ps.keys.ptr; psf.g_cells = ps.cells.ptr; psf.touched = ps.touched.ptr; psf.touched_n = ps.touched_n.ptr; psf.g_log2_size = ps.log2_size;
	psf.ref_pixels = ps.ref_pixels.ptr; psf.ref_cache = ps.ref_cache.ptr; psf.ref_wd = ps.ref_wd.ptr; psf.ref_wg = ps.ref_wg.ptr; psf.ref_size = ps.ref_size.ptr; psf.ref_k = ps.ref_k.ptr;
	psf.depth = ps.opt.psf_depth; psf.width = ps.opt.psf_width; psf.max_prob = ps.opt.psf_max_prob; psf.firefly = ps.opt.firefly_filter;
	return psf;
}
// psf_blending, update_variances, clamp_frame(100): the tail of PSFPT::render (src/renderers/psfpt_impl.h:275-284, 402-420)
void finish_pass(fpt_context* ctx, const PsfDev& psf, const FrameBufferDev& fb, uint32_t instance, uint32_t bounces_run)
{
	fpt_context::PsfState& ps = ctx->psf;
	hipStream_t s = ctx->stream;
	const uint32_t n = ctx->n_local;
	const float frame_weight = 1.0f / float(instance + 1);
	for (uint32_t bounce = ps.opt.psf_depth; bounce < bounces_run; ++bounce)
	{
		PsfDev pb = psf;
		pb.ref_pixels = psf.ref_pixels + size_t(bounce) * n; pb.ref_cache = psf.ref_cache + size_t(bounce) * n; pb.ref_k = psf.ref_k + size_t(bounce) * n;
		pb.ref_wd = psf.ref_wd + size_t(bounce) * n; pb.ref_wg = psf.ref_wg + size_t(bounce) * n; pb.ref_size = psf.ref_size + bounce;
		launch_psf_blend(pb, fb, frame_weight, n, s);
	}
	launch_variance(fb, ctx->d_pixels, n, instance + 1, s);
	launch_clamp_frame(fb, ctx->d_pixels, n, 100.0f, s);
	FPT_HIP_CHECK(hipGetLastError());
}
}

extern "C" {

int fpt_psfpt_init(fpt_context* ctx, const fpt_pt_options* opts, const fpt_psf_options* psf, const fpt_rendering_context_view* view,
                   const char* h_samples_dir, const uint32_t* d_pixels, uint32_t n_local_pixels)
{
	const int st = fpt_pt_init(ctx, opts, view, h_samples_dir, d_pixels, n_local_pixels);
	if (st != 0) return st;
	return guarded(ctx, [&] {
		require(psf != nullptr, "fpt_psfpt_init: null options");
		require(psf->psf_temporal_reuse >= 1, "fpt_psfpt_init: psf_temporal_reuse must be >= 1");
		fpt_context::PsfState& s = ctx->psf;
		s.opt = *psf;
		const uint32_t n = ctx->n_local;
		// per-path vertex info and per-shadow-sample hit records
		ctx->q_a.vinfo.alloc(n); ctx->q_b.vinfo.alloc(n);
		ctx->q_shadow.vinfo.alloc(n); ctx->q_shadow.hits.alloc(n);
		ctx->q_shadow_dir.vinfo.alloc(view->dir_lights_count ? n : 1); ctx->q_shadow_dir.hits.alloc(view->dir_lights_count ? n : 1);
		// cache: the reference allocates 64M cells (HASH_SIZE, src/renderers/psfpt_impl.h:46); a cell is addressed by key, so the table size only
		// has to exceed the number of live cells: 2^26 for frames beyond a megapixel, 2^24 below
		s.log2_size = uint64_t(view->res_x) * view->res_y > (1u << 20) ? 26u : 24u;
		s.keys.alloc(size_t(1) << s.log2_size); s.cells.alloc((size_t(1) << s.log2_size) * 4);
		FPT_HIP_CHECK(hipMemsetAsync(s.keys.ptr, 0xFF, (size_t(1) << s.log2_size) * sizeof(unsigned long long), ctx->stream));
		FPT_HIP_CHECK(hipMemsetAsync(s.cells.ptr, 0, (size_t(1) << s.log2_size) * 4 * sizeof(long long), ctx->stream));
		// PSFRefQueue: n_pixels * (L + 1) entries as in the reference (:176-181).  A path can open a new cache vertex after every glossy bounce,
		// i.e. own several references; they are kept per bounce (<= 1 per path and bounce) and blended bounce by bounce, which makes the
		// per-pixel order of the blend the order of creation without any atomics
		const size_t nr = size_t(n) * (opts->max_path_length + 1);
		s.ref_pixels.alloc(nr); s.ref_k.alloc(nr); s.ref_cache.alloc(nr); s.ref_wd.alloc(nr); s.ref_wg.alloc(nr); s.ref_size.alloc(32);
		// m_bbox = renderer.compute_bbox(): over the mesh vertices (src/renderer.cu:1086-1097)
		std::vector<float> vtx(size_t(view->mesh.num_vertices) * 4);
		FPT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
		if (!vtx.empty()) FPT_HIP_CHECK(hipMemcpy(vtx.data(), view->mesh.vertex_data, vtx.size() * sizeof(float), hipMemcpyDeviceToHost));
		float lo[3] = { 1.0e30f, 1.0e30f, 1.0e30f }, hi[3] = { -1.0e30f, -1.0e30f, -1.0e30f };
		for (int i = 0; i < view->mesh.num_vertices; ++i) for (int c = 0; c < 3; ++c)
		{
			const float v = vtx[size_t(i) * 4 + c];
			lo[c] = lo[c] < v ? lo[c] : v; hi[c] = hi[c] > v ? hi[c] : v;
		}
		for (int c = 0; c < 3; ++c) { s.bbox[c] = lo[c]; s.bbox[3 + c] = hi[c]; }
		ctx->d_counters.alloc(std::max<size_t>(ctx->d_counters.count, size_t(P_TOTAL)));
		s.ready = true;
	});
}

int fpt_psfpt_download_cells(fpt_context* ctx, uint64_t* h_keys, uint64_t* h_counts, int64_t* h_sums, uint32_t max_cells, uint32_t* n_cells)
{
	return guarded(ctx, [&] {
		flush_deferred(ctx);
		fpt_context::PsfState& s = ctx->psf;
		require(s.ready, "fpt_psfpt_download_cells: fpt_psfpt_init has not been called");
		const size_t n = size_t(1) << s.log2_size;
		std::vector<unsigned long long> keys(n); std::vector<long long> cells(n * 4);
		FPT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
		FPT_HIP_CHECK(hipMemcpy(keys.data(), s.keys.ptr, n * 8, hipMemcpyDeviceToHost));
		FPT_HIP_CHECK(hipMemcpy(cells.data(), s.cells.ptr, n * 32, hipMemcpyDeviceToHost));
		uint32_t k = 0;
		for (size_t i = 0; i < n; ++i)
		{
			if (keys[i] == ~0ull) continue;
			if (k < max_cells && h_keys) { h_keys[k] = keys[i]; h_counts[k] = uint64_t(cells[4 * i + 3]); h_sums[3 * size_t(k)] = cells[4 * i]; h_sums[3 * size_t(k) + 1] = cells[4 * i + 1]; h_sums[3 * size_t(k) + 2] = cells[4 * i + 2]; }
			++k;
		}
		if (n_cells) *n_cells = k;
	});
}

// PSFPT::render for n_passes >= 1 passes in flight (instances instance .. instance + n_passes - 1).  One pass: the reference's sequence.  A batch:
// the passes of a batch are independent until the blend (nothing on a path reads the cache), so they run as one wavefront like the path tracer's
// batches, each pass accumulating into its own pass table and its own frame planes; the tables are then folded into the global table IN PASS ORDER,
// each pass's table taking the state of the cache after that pass, the references are blended from their pass's table into their pass's plane, and
// merge_passes applies rescale / planes / variances / clamp_frame pass by pass.  The cache is bit-identical to n sequential calls, the frame to rounding.
static void render_psf(fpt_context* ctx, uint32_t instance, uint32_t n_passes, const fpt_rendering_context_view* view)
{
		fpt_context::PsfState& ps = ctx->psf;
		require(ps.ready && ctx->pt_ready, "fpt_psfpt_render: fpt_psfpt_init has not been called");
		require(ctx->has_geometry && ctx->has_emitters, "fpt_psfpt_render: geometry / mesh lights are not initialised");
		const bool batched = n_passes > 1;
		require(!batched || (n_passes <= ps.max_batch && !ps.sharded), "fpt_psfpt_render_batch: more passes than fpt_psfpt_set_batch sized the storage for (or a sharded context)");
		// the queues and the log are shared with the plain path tracer: a later fpt_pt_set_batch / fpt_pt_set_deferred may have re-shaped them (no blend cells, fewer
		// mask bits, smaller queues), and a view with directional lights needs their log cells
		require(!batched || (n_passes <= ctx->max_batch && ctx->log_blend.ptr != nullptr && ctx->log_mask_words == uint32_t((4 * size_t(ctx->opt.max_path_length) + 31) / 32)),
		        "fpt_psfpt_render_batch: the shared queues / contribution log were re-sized by fpt_pt_set_batch since fpt_psfpt_set_batch: call fpt_psfpt_set_batch again");
		require(!batched || !view->dir_lights_count || ctx->log_nee[0].ptr != nullptr, "fpt_psfpt_render_batch: the view has directional lights but the log was sized without them: call fpt_psfpt_set_batch with this view");
		hipStream_t s = ctx->stream;
		const fpt_pt_options& opt = ctx->opt;
		const FrameBufferDev real_fb = fb_dev(view->fb);
		FrameBufferDev fb = real_fb;
		ContribLog log; std::memset(&log, 0, sizeof(log));
		if (batched)
		{
			// as in the path tracer (fpt_api.cpp render_lane): the albedo channels keep a plane per pass, every other sample goes to the path's cell of the log
			for (int c = 0; c < 6; ++c) fb.ch[c] = nullptr;
			fb.ch[FPT_FB_DIFFUSE_A] = reinterpret_cast<float4*>(ctx->d_acc[FPT_FB_DIFFUSE_A].ptr);
			fb.ch[FPT_FB_SPECULAR_A] = reinterpret_cast<float4*>(ctx->d_acc[FPT_FB_SPECULAR_A].ptr);
			log = lane_log(ctx, 0);
		}
		const uint32_t n = ctx->n_local * n_passes;          // paths of the launch chain
		uint32_t* cnt = ctx->d_counters.ptr;
		if (!batched) launch_rescale(fb, ctx->d_pixels, n, float(instance) / float(instance + 1), s);
		FPT_HIP_CHECK(hipMemsetAsync(cnt, 0, P_TOTAL * sizeof(uint32_t), s));
		auto reset_cache = [&] {          // initialize the shading cache (src/renderers/psfpt_impl.h:385-387)
			FPT_HIP_CHECK(hipMemsetAsync(ps.keys.ptr, 0xFF, (size_t(1) << ps.log2_size) * sizeof(unsigned long long), s));
			FPT_HIP_CHECK(hipMemsetAsync(ps.cells.ptr, 0, (size_t(1) << ps.log2_size) * 4 * sizeof(long long), s));
		};
		if (!batched && (instance % ps.opt.psf_temporal_reuse) == 0) reset_cache();
		FPT_HIP_CHECK(hipMemsetAsync(ps.ref_size.ptr, 0, 32 * sizeof(uint32_t), s));

		PassInfo pass; pass.base_instance = instance; pass.n_passes = n_passes;
		if (batched) { pass.n_slot = ctx->n_local; pass.acc_stride = ctx->n_local; pass.pixels = ctx->d_pixels; }
		else         { pass.n_slot = view->res_x * view->res_y; pass.acc_stride = pass.n_slot; pass.pixels = nullptr; }
		SequenceView seq; seq.shifts = ctx->d_shifts.ptr; seq.n_dims = ctx->seq_dims; seq.tile_size = ctx->seq_tile;
		PsfDev psf;
		psf.keys = ps.keys.ptr; psf.cells = ps.cells.ptr; psf.log2_size = ps.log2_size;
		psf.ref_pixels = ps.ref_pixels.ptr; psf.ref_cache = ps.ref_cache.ptr; psf.ref_wd = ps.ref_wd.ptr; psf.ref_wg = ps.ref_wg.ptr; psf.ref_size = ps.ref_size.ptr; psf.ref_k = ps.ref_k.ptr;
		psf.bbox_lo = mk3(ps.bbox[0], ps.bbox[1], ps.bbox[2]); psf.bbox_hi = mk3(ps.bbox[3], ps.bbox[4], ps.bbox[5]);
		psf.depth = ps.opt.psf_depth; psf.width = ps.opt.psf_width; psf.max_prob = ps.opt.psf_max_prob; psf.firefly = ps.opt.firefly_filter; psf.instance = instance;
		psf.g_keys = nullptr; psf.g_cells = nullptr; psf.touched = nullptr; psf.touched_n = nullptr; psf.pass_stride = 0; psf.g_log2_size = ps.log2_size;
		if (ps.sharded)
		{
			require(!ps.pending, "fpt_psfpt_render: the previous pass has not been finished (fpt_psfpt_exchange_cells / fpt_psfpt_import_cells, then fpt_psfpt_finish)");
			psf.g_keys = ps.keys.ptr; psf.g_cells = ps.cells.ptr; psf.keys = ps.p_keys.ptr; psf.cells = ps.p_cells.ptr;
			psf.touched = ps.touched.ptr; psf.touched_n = ps.touched_n.ptr;
		}
		if (batched)
		{
			// one pass table per pass of the batch (2^b_log2 slots each, enough for every cell a pass can create), their slot lists, the global table
			psf.g_keys = ps.keys.ptr; psf.g_cells = ps.cells.ptr; psf.keys = ps.b_keys.ptr; psf.cells = ps.b_cells.ptr;
			psf.touched = ps.b_touched.ptr; psf.touched_n = ps.b_touched_n.ptr; psf.log2_size = ps.b_log2; psf.pass_stride = 1u << ps.b_log2;
		}

		auto counter = [&](uint32_t bounce, uint32_t which) { return cnt + P_QUEUES + P_PER_BOUNCE * bounce + which; };
		uint32_t ticket = 0;
		// the rays are those of the renderer's own queues (fpt_device.h PathQueue / ShadowQueue): the .w words carry PixelInfo / pass offset, the intervals are the queues'
		auto trace = [&](const float4* rays, float4* hits, const uint32_t* count_ptr, bool any_hit, bool primary = false)
		{
			TraceParams tp = base_trace_params(ctx);
			tp.rays = rays; tp.hits = hits; tp.count_ptr = count_ptr; tp.work_counter = cnt + P_TICKET_STRIDE * (ticket++); tp.stats = ctx->d_trace_stats.ptr;
			if (any_hit) timed_launch(ctx, 2, s, [&] { launch_trace_shadow_queue(tp, ctx->counting, ctx->trace_blocks(), s); });
			else         timed_launch(ctx, 0, s, [&] { launch_trace_closest_queue(tp, primary, ctx->counting, ctx->trace_blocks(), s); });
		};
		QueueStorage* qa = &ctx->q_a; QueueStorage* qb = &ctx->q_b;
		PathQueue qin = qa->view(counter(0, P_PATH)), qout = qb->view(counter(1, P_PATH));
		{
			PrimaryParams pp;
			pp.out = qin; pp.seq = seq; pp.pixels = ctx->d_pixels; pp.n_pixels = ctx->n_local; pp.res_x = view->res_x; pp.res_y = view->res_y; pp.pass = pass;
			pp.eye = mk3(view->camera.eye[0], view->camera.eye[1], view->camera.eye[2]);
			camera_frame(view->camera, view->aspect, pp.U, pp.V, pp.W);
			pp.W_len = length(pp.W);
			const float tn = tanf(view->camera.fov / 2);
			pp.sq_focal = (float(view->res_x * view->res_y) / 4.0f) / (tn * tn);
			launch_primary_rays(pp, s);
		}
		ShadeParams sh; std::memset(&sh, 0, sizeof(sh));
		sh.seq = seq; sh.shade_records = ensure_shade_records(ctx, view, s); sh.mesh = view->mesh; sh.textures = view->d_textures; sh.table = view->d_glossy_reflectance;
		sh.dir_lights = view->d_dir_lights; sh.n_dir_lights = view->dir_lights_count;
		EmitterView em;
		em.n_prims = uint32_t(ctx->emitters.mesh_cdf.size()); em.prims_cdf = ctx->d_mesh_cdf.ptr; em.prims_inv_area = ctx->d_mesh_inv_area.ptr;
		em.n_vpls = opt.nee_type == 1 ? uint32_t(ctx->emitters.vpls.size()) : 0u; em.vpls = opt.nee_type == 1 ? ctx->d_vpls.ptr : nullptr; em.norm = ctx->emitters.norm;
		em.vpl_points = opt.nee_type == 1 ? ensure_vpl_points(ctx, view, s) : nullptr;
		sh.emitters = em; sh.fb = fb; sh.log = log; sh.gbuffer = real_fb; sh.opt = opt; sh.res_x = view->res_x; sh.res_y = view->res_y; sh.pass = pass; sh.psf = psf;
		const uint32_t total_vpls = uint32_t(ctx->emitters.vpls.size());
		const float frame_weight = 1.0f / float(instance + 1);

		// what the fused resolve of a MIXED launch needs, one block per bounce; nothing in it changes from pass to pass (the frame weight
		// travels as the launch's first instance), so it is uploaded only when a pointer changed
		{
			std::vector<ResolveParams> blocks(opt.max_path_length);
			std::memset(blocks.data(), 0, blocks.size() * sizeof(ResolveParams));
			for (uint32_t b = 0; b < opt.max_path_length; ++b)
			{
				ResolveParams& r = blocks[b];
				r.q = ctx->q_shadow.view(nullptr); r.fb = fb; r.bounce = b; r.psf = psf; r.psf.instance = 0; r.log = log; r.kind = 1;
				r.pass = pass; r.pass.base_instance = 0;          // the launch's first instance travels as a kernel argument
			}
			if (ps.h_resolve.size() != blocks.size() || std::memcmp(ps.h_resolve.data(), blocks.data(), blocks.size() * sizeof(ResolveParams)) != 0)
			{
				ps.d_resolve.upload(blocks.data(), blocks.size(), s);
				ps.h_resolve = blocks;
			}
		}
		uint32_t bounces_run = 0;
		trace(qin.rays, qin.hits, qin.size, false, true);
		for (uint32_t bounce = 0; bounce < opt.max_path_length; ++bounce)
		{
			sh.bounce = bounce;
			sh.do_nee = (total_vpls && (bounce + 2 <= opt.max_path_length) && ((bounce == 0 && opt.direct_lighting_nee && opt.direct_lighting) || (bounce > 0 && opt.indirect_lighting_nee))) ? 1u : 0u;
			sh.do_emissive = ((bounce == 0 && opt.visible_lights) || (bounce == 1 && opt.direct_lighting_bsdf && opt.direct_lighting) || (bounce > 1 && opt.indirect_lighting_bsdf)) ? 1u : 0u;
			const uint32_t max_path_vertices = opt.max_path_length + (((opt.max_path_length == 2 && opt.direct_lighting_bsdf) || (opt.max_path_length > 2 && opt.indirect_lighting_bsdf)) ? 1u : 0u);
			sh.do_scatter = (bounce + 2 < max_path_vertices) ? 1u : 0u;
			ShadowQueue qsd = ctx->q_shadow_dir.view(counter(bounce, P_SHADOW_DIR)), qs = ctx->q_shadow.view(counter(bounce, P_SHADOW));
			sh.in = qin; sh.scatter = qout; sh.shadow_dir = qsd; sh.shadow = qs;
			sh.psf.ref_pixels = psf.ref_pixels + size_t(bounce) * n; sh.psf.ref_cache = psf.ref_cache + size_t(bounce) * n; sh.psf.ref_k = psf.ref_k + size_t(bounce) * n;
			sh.psf.ref_wd = psf.ref_wd + size_t(bounce) * n; sh.psf.ref_wg = psf.ref_wg + size_t(bounce) * n; sh.psf.ref_size = psf.ref_size + bounce;
			timed_launch(ctx, 3, s, [&] { launch_shade_psf(sh, n, s); });
			++bounces_run;
			ResolveParams rp; std::memset(&rp, 0, sizeof(rp));
			rp.fb = fb; rp.bounce = bounce; rp.pass = pass; rp.psf = psf; rp.frame_weight = frame_weight; rp.log = log;
			if (view->dir_lights_count && (bounce + 2 <= opt.max_path_length) && (bounce > 0 || opt.direct_lighting))
			{
				trace(qsd.rays, ctx->q_shadow_dir.hits.ptr, qsd.size, true);
				rp.q = qsd; rp.hits = ctx->q_shadow_dir.hits.ptr; rp.kind = 0;
				launch_psf_resolve(rp, n, s);
			}
			if (sh.do_nee && sh.do_scatter)
			{
				// MIXED launch as in the path tracer: the closest-hit rays of the next bounce together with this bounce's shadow rays, whose
				// cache-aware resolve (PSFPTVertexProcessor::accumulate_nee) is fused into the retirement of the any-hit lanes
				TraceParams mp = base_trace_params(ctx);
				mp.rays = qout.rays; mp.hits = qout.hits; mp.count_ptr = qout.size; mp.work_counter = cnt + P_TICKET_STRIDE * (ticket++); mp.stats = ctx->d_trace_stats.ptr;
				mp.shadow_rays = qs.rays; mp.shadow_size = qs.size; mp.base_instance = instance;
				mp.fused = reinterpret_cast<const FusedResolve*>(ps.d_resolve.ptr + bounce);      // MIXED_PSF reads a ResolveParams there
				timed_launch(ctx, 0, s, [&] { launch_trace_mixed_psf(mp, ctx->counting, ctx->trace_blocks(), s); });
			}
			else
			{
				if (sh.do_nee)
				{
					trace(qs.rays, ctx->q_shadow.hits.ptr, qs.size, true);
					rp.q = qs; rp.hits = ctx->q_shadow.hits.ptr; rp.kind = 1;
					launch_psf_resolve(rp, n, s);
				}
				if (!sh.do_scatter) break;
				trace(qout.rays, qout.hits, qout.size, false);
			}
			if (!sh.do_scatter) break;
			std::swap(qa, qb);
			qin = qa->view(counter(bounce + 1, P_PATH)); qout = qb->view(counter(bounce + 2, P_PATH));
		}
		if (ps.sharded)
		{
			// this rank's cells of the pass -> records; the frame is completed by fpt_psfpt_finish once every rank's records are in the global table
			launch_psf_collect(psf, ps.records.ptr, s);
			ps.pending = true; ps.pending_instance = instance; ps.pending_bounces = bounces_run;
			FPT_HIP_CHECK(hipGetLastError());
			return;
		}
		if (batched)
		{
			// the cache after pass k, for k = 0, 1, ...: global += pass table k (the reset of a reuse window falls between two passes), pass table k := global
			for (uint32_t k = 0; k < n_passes; ++k)
			{
				if (((instance + k) % ps.opt.psf_temporal_reuse) == 0) reset_cache();
				launch_psf_prefix(psf, k, s);
			}
			for (uint32_t bounce = ps.opt.psf_depth; bounce < bounces_run; ++bounce)
			{
				PsfDev pb = psf;
				pb.ref_pixels = psf.ref_pixels + size_t(bounce) * n; pb.ref_cache = psf.ref_cache + size_t(bounce) * n; pb.ref_k = psf.ref_k + size_t(bounce) * n;
				pb.ref_wd = psf.ref_wd + size_t(bounce) * n; pb.ref_wg = psf.ref_wg + size_t(bounce) * n; pb.ref_size = psf.ref_size + bounce;
				launch_psf_blend_batch(pb, log, bounce, pass, n, s);
			}
			// rescale -> albedos -> the pass's samples in order -> its blends bounce by bounce -> variances -> clamp_frame(100), pass by pass: the sequential frame
			launch_merge_passes_exact(real_fb, fb.ch[FPT_FB_DIFFUSE_A], fb.ch[FPT_FB_SPECULAR_A], log, ctx->d_pixels, ctx->n_local, pass, s, true, ps.opt.firefly_filter, 100.0f);
			for (uint32_t k = 0; k < n_passes; ++k)
			{
				PsfDev t = psf;
				t.keys = psf.keys + size_t(k) * psf.pass_stride; t.cells = psf.cells + 4 * size_t(k) * psf.pass_stride;
				t.touched = psf.touched + size_t(k) * psf.pass_stride; t.touched_n = psf.touched_n + k;
				launch_psf_clear_pass(t, s);
			}
			FPT_HIP_CHECK(hipMemsetAsync(ps.b_touched_n.ptr, 0, ps.b_touched_n.count * sizeof(uint32_t), s));
			FPT_HIP_CHECK(hipGetLastError());
			return;
		}
		finish_pass(ctx, psf, fb, instance, bounces_run);
}

} // extern "C"
namespace fpt { void psf_render_passes(fpt_context* ctx, uint32_t first, uint32_t n, const fpt_rendering_context_view* view) { render_psf(ctx, first, n, view); } }
extern "C" {

int fpt_psfpt_render(fpt_context* ctx, uint32_t instance, const fpt_rendering_context_view* view)
{
	return guarded(ctx, [&] {
		require(view != nullptr, "fpt_psfpt_render: null view");
		if (ctx->defer_kind != DEFER_PSFPT || ctx->defer_max <= 1 || ctx->psf.sharded) { flush_deferred(ctx); render_psf(ctx, instance, 1, view); return; }
		defer_pass(ctx, DEFER_PSFPT, instance, view);          // fpt_psfpt_set_deferred: consecutive instances of the same view, as fpt_pt_render
	});
}
int fpt_psfpt_render_batch(fpt_context* ctx, uint32_t first_instance, uint32_t n_passes, const fpt_rendering_context_view* view)
{ return guarded(ctx, [&] { require(n_passes >= 1, "fpt_psfpt_render_batch: n_passes must be >= 1"); flush_deferred(ctx); render_psf(ctx, first_instance, n_passes, view); }); }
/* deferred fpt_psfpt_render: as fpt_pt_set_deferred (the PSFPT's passes in flight are bit-identical to sequential passes too) */
int fpt_psfpt_set_deferred(fpt_context* ctx, uint32_t max_passes, const fpt_rendering_context_view* view)
{
	{ const int st = guarded(ctx, [&] { flush_deferred(ctx); require(max_passes >= 1, "fpt_psfpt_set_deferred: max_passes must be >= 1"); }); if (st != 0) return st; }
	if (max_passes > ctx->psf.max_batch) { const int st = fpt_psfpt_set_batch(ctx, max_passes, view); if (st != 0) return st; }
	return guarded(ctx, [&] { ctx->defer_max = max_passes; ctx->defer_kind = DEFER_PSFPT; });
}

// storage for `max_passes` passes in flight: the path tracer's queues and planes (fpt_pt_set_batch), the PSFPT's per-path words and reference
// queue, and one pass table per pass of 2^b slots, 2^b >= (pixels rendered here) x (max_path_length + 1) >= the cells a pass can create, so a pass
// table fills up only if the global table (2^24 or 2^26 slots) would
int fpt_psfpt_set_batch(fpt_context* ctx, uint32_t max_passes, const fpt_rendering_context_view* view)
{
	const int st = fpt_internal_set_batch(ctx, max_passes, view, true);       // queues + albedo planes + the contribution log with blend cells
	if (st != 0) return st;
	return guarded(ctx, [&] {
		fpt_context::PsfState& ps = ctx->psf;
		require(ps.ready, "fpt_psfpt_set_batch: fpt_psfpt_init has not been called");
		require(!ps.sharded || max_passes == 1, "fpt_psfpt_set_batch: a sharded context renders one pass at a time");
		const size_t n = size_t(ctx->n_local) * max_passes;
		ctx->q_a.vinfo.alloc(n); ctx->q_b.vinfo.alloc(n);
		ctx->q_shadow.vinfo.alloc(n); ctx->q_shadow.hits.alloc(n);
		ctx->q_shadow_dir.vinfo.alloc(view->dir_lights_count ? n : 1); ctx->q_shadow_dir.hits.alloc(view->dir_lights_count ? n : 1);
		const size_t nr = n * (ctx->opt.max_path_length + 1);
		ps.ref_pixels.alloc(nr); ps.ref_k.alloc(nr); ps.ref_cache.alloc(nr); ps.ref_wd.alloc(nr); ps.ref_wg.alloc(nr);
		ps.max_batch = max_passes;
		ps.h_resolve.clear();          // the blocks of the fused resolve name these buffers
		if (max_passes > 1)
		{
			uint32_t b = 10;
			while ((size_t(1) << b) < size_t(ctx->n_local) * (ctx->opt.max_path_length + 1) && b < ps.log2_size) ++b;
			ps.b_log2 = b;
			const size_t slots = (size_t(1) << b) * max_passes;
			ps.b_keys.alloc(slots); ps.b_cells.alloc(slots * 4); ps.b_touched.alloc(slots); ps.b_touched_n.alloc(std::max<size_t>(max_passes, 32));
			FPT_HIP_CHECK(hipMemsetAsync(ps.b_keys.ptr, 0xFF, slots * sizeof(unsigned long long), ctx->stream));
			FPT_HIP_CHECK(hipMemsetAsync(ps.b_cells.ptr, 0, slots * 4 * sizeof(long long), ctx->stream));
			FPT_HIP_CHECK(hipMemsetAsync(ps.b_touched_n.ptr, 0, ps.b_touched_n.count * sizeof(uint32_t), ctx->stream));
		}
		FPT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
	});
}

int fpt_psfpt_set_sharded(fpt_context* ctx, int on)
{
	return guarded(ctx, [&] {
		flush_deferred(ctx);
		fpt_context::PsfState& ps = ctx->psf;
		require(ps.ready, "fpt_psfpt_set_sharded: fpt_psfpt_init has not been called");
		require(!ps.pending, "fpt_psfpt_set_sharded: a pass is pending");
		ps.sharded = on != 0;
		if (!ps.sharded) return;
		const size_t n = size_t(1) << ps.log2_size;
		ps.p_keys.alloc(n); ps.p_cells.alloc(n * 4); ps.touched.alloc(n); ps.touched_n.alloc(32); ps.records.alloc(n);
		FPT_HIP_CHECK(hipMemsetAsync(ps.p_keys.ptr, 0xFF, n * sizeof(unsigned long long), ctx->stream));
		FPT_HIP_CHECK(hipMemsetAsync(ps.p_cells.ptr, 0, n * 4 * sizeof(long long), ctx->stream));
		FPT_HIP_CHECK(hipMemsetAsync(ps.touched_n.ptr, 0, 32 * sizeof(uint32_t), ctx->stream));
	});
}

int fpt_psfpt_export_cells(fpt_context* ctx, const void** d_records, uint32_t* n_records)
{
	return guarded(ctx, [&] {
		fpt_context::PsfState& ps = ctx->psf;
		require(ps.sharded && ps.pending, "fpt_psfpt_export_cells: no sharded pass is pending");
		require(d_records && n_records, "fpt_psfpt_export_cells: null argument");
		FPT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
		FPT_HIP_CHECK(hipMemcpy(n_records, ps.touched_n.ptr, sizeof(uint32_t), hipMemcpyDeviceToHost));
		*d_records = ps.records.ptr;
	});
}

int fpt_psfpt_import_cells(fpt_context* ctx, const void* d_records, uint32_t n_records)
{
	return guarded(ctx, [&] {
		fpt_context::PsfState& ps = ctx->psf;
		require(ps.sharded && ps.pending, "fpt_psfpt_import_cells: no sharded pass is pending");
		require(d_records || n_records == 0, "fpt_psfpt_import_cells: null records");
		if (!n_records) return;
		launch_psf_merge(sharded_view(ctx), static_cast<const PsfRecord*>(d_records), nullptr, n_records, ctx->stream);
		FPT_HIP_CHECK(hipGetLastError());
	});
}

int fpt_psfpt_finish(fpt_context* ctx, const fpt_rendering_context_view* view)
{
	return guarded(ctx, [&] {
		fpt_context::PsfState& ps = ctx->psf;
		require(ps.sharded && ps.pending, "fpt_psfpt_finish: no sharded pass is pending");
		PsfDev psf = sharded_view(ctx);
		psf.instance = ps.pending_instance;
		finish_pass(ctx, psf, fb_dev(view->fb), ps.pending_instance, ps.pending_bounces);
		launch_psf_clear_pass(psf, ctx->stream);
		FPT_HIP_CHECK(hipMemsetAsync(ps.touched_n.ptr, 0, sizeof(uint32_t), ctx->stream));
		ps.pending = false;
		FPT_HIP_CHECK(hipGetLastError());
	});
}

} // extern "C"
