// fpt_renderer.cpp — host-side mirror of RenderingContext / PathTracer over the C-ABI (see renderer_interface.h).
// Error behaviour follows the reference: every failure is terminal — the reference prints and exit()s
// (src/renderer.cu:1051-1055, src/rt.cpp catch blocks); here a std::runtime_error carrying fpt_last_error() is thrown so
// an embedding host can decide, and the C test hooks at the bottom translate it into an error string.
#include "renderer_interface.h"
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstdio>
#include <algorithm>
#include <cstring>
#include <stdexcept>

namespace fermat {

namespace {
void check(fpt_context* ctx, int status, const char* what)
{
	if (status != 0) throw std::runtime_error(std::string(what) + ": " + fpt_last_error(ctx));
}
void hip_check(hipError_t e, const char* what) { if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e)); }

// How many passes the library may keep in flight behind render(instance) (`-batch N`; 0 = the renderer's default).  Passes in flight cost memory -- queues, albedo
// planes and the contribution log, fpt_bytes_per_path_in_flight x pixels x passes: 25-50 GB at 1080p for 32 passes -- so a DEFAULT is sized to at most a quarter of
// the device memory that is free right now, and whatever was asked for is halved until the set-up call succeeds (ending at one pass per render(), the
// reference's own mode, which needs none of it): a plain `-pt` run on a small or shared GPU renders more slowly instead of failing at init.
// set_up(n): the renderer's set-up call for n > 1 passes in flight (0 = success); restore(): re-sizes the queues for one pass after a failed attempt
template <typename SetUp, typename Restore>
uint32_t choose_passes_in_flight(fpt_context* ctx, uint32_t asked, uint32_t dflt, uint32_t renderer, const fpt_rendering_context_view& v, uint64_t n_here, uint64_t hard_cap, SetUp&& set_up, Restore&& restore)
{
	bool failed = false;
	uint64_t n = asked ? asked : dflt;
	n = std::max<uint64_t>(1, std::min<uint64_t>(n, hard_cap));
	if (!asked && n > 1)
	{
		uint64_t free_b = 0, total_b = 0, per_path = 0;
		if (fpt_device_memory(ctx, &free_b, &total_b) == 0 && fpt_bytes_per_path_in_flight(ctx, renderer, &v, &per_path) == 0 && per_path)
			n = std::max<uint64_t>(1, std::min<uint64_t>(n, (free_b / 4) / std::max<uint64_t>(1, per_path * n_here)));
	}
	while (n > 1 && set_up(uint32_t(n)) != 0)
	{
		std::fprintf(stderr, "[fermat_hip] %u passes in flight do not fit (%s): trying %u\n", unsigned(n), fpt_last_error(ctx), unsigned(n / 2));
		n /= 2; failed = true;
	}
	if (failed && n == 1) check(ctx, restore(), "one pass per render() after the passes in flight did not fit");
	return uint32_t(n);
}

template <typename T>
T* upload(std::vector<void*>& allocs, const T* h, size_t n)
{
	if (n == 0) return nullptr;
	void* d = nullptr;
	hip_check(hipMalloc(&d, n * sizeof(T)), "hipMalloc");
	allocs.push_back(d);
	hip_check(hipMemcpy(d, h, n * sizeof(T), hipMemcpyHostToDevice), "hipMemcpy");
	return static_cast<T*>(d);
}
} // namespace

// ---- RTContext ---------------------------------------------------------------------------------------------------------------
void RTContext::create_geometry(const uint32 tri_count, const int* index_ptr, const uint32 vertex_count, const float* vertex_ptr,
                                const int*, const float*, const int*, const float*, const int*)
{ check(ctx, fpt_rt_create_geometry(ctx, tri_count, index_ptr, vertex_count, vertex_ptr), "RTContext::create_geometry"); }
void RTContext::trace(const uint32 count, const Ray* rays, Hit* hits)
{ check(ctx, fpt_rt_trace(ctx, count, reinterpret_cast<const fpt_ray*>(rays), reinterpret_cast<fpt_hit*>(hits)), "RTContext::trace"); }
void RTContext::trace(const uint32 count, const MaskedRay* rays, Hit* hits)
{ check(ctx, fpt_rt_trace(ctx, count, reinterpret_cast<const fpt_ray*>(rays), reinterpret_cast<fpt_hit*>(hits)), "RTContext::trace"); }
void RTContext::trace_shadow(const uint32 count, const MaskedRay* rays, Hit* hits)
{ check(ctx, fpt_rt_trace_shadow(ctx, count, reinterpret_cast<const fpt_ray*>(rays), reinterpret_cast<fpt_hit*>(hits)), "RTContext::trace_shadow"); }
void RTContext::trace_shadow(const uint32 count, const MaskedRay* rays, uint32* bits)
{ check(ctx, fpt_rt_trace_shadow_bits(ctx, count, reinterpret_cast<const fpt_ray*>(rays), bits), "RTContext::trace_shadow"); }

// ---- RenderingContext --------------------------------------------------------------------------------------------------------
RenderingContext::RenderingContext() : m_ctx(nullptr), m_renderer(nullptr), m_res_x(1600), m_res_y(900), m_shading_mode(FPT_SHADING_SHADED), m_aspect(0.0f), m_exposure(1.0f), m_gamma(2.2f)
{ std::memset(&m_scene, 0, sizeof(m_scene)); std::memset(&m_view, 0, sizeof(m_view)); }

RenderingContext::~RenderingContext()
{
	if (m_renderer) m_renderer->destroy();
	m_rt_context.reset();
	if (m_ctx) { fpt_synchronize(m_ctx); }
	for (void* p : m_device_allocs) (void)hipFree(p);
	if (m_ctx) fpt_destroy(m_ctx);
}

uint32 RenderingContext::register_renderer(const char* name, RendererFactoryFunction factory)
{
	m_renderer_names.push_back(name);
	m_renderer_factories.push_back(factory);
	return uint32(m_renderer_factories.size() - 1);
}

void RenderingContext::init(int argc, char** argv, const SceneArrays& scene)
{
	// built-in renderer table (src/renderer.cu:471-477) — only "pt" exists in this build
	register_renderer("pt", &HipPathTracer::factory);
	register_renderer("bpt", &HipBPT::factory);
	register_renderer("psfpt", &HipPSFPT::factory);
	int device = 0;
	uint32 renderer_type = 0;
	for (int i = 0; i < argc; ++i)                           // flag loop, src/renderer.cu:493-539 (unknown flags are ignored)
	{
		if (std::strcmp(argv[i], "-r") == 0 || std::strcmp(argv[i], "-res") == 0) { m_res_x = uint32(std::atoi(argv[++i])); m_res_y = uint32(std::atoi(argv[++i])); }
		else if (std::strcmp(argv[i], "-a") == 0) m_aspect = float(std::atof(argv[++i]));
		else if (std::strcmp(argv[i], "-device") == 0) device = std::atoi(argv[++i]);
		else if (std::strcmp(argv[i], "-filtered") == 0) m_shading_mode = FPT_SHADING_FILTERED;          // the viewer's 'f' key (src/glut_viewer.cu:298)
		else if (std::strcmp(argv[i], "-shading-mode") == 0) m_shading_mode = uint32(std::atoi(argv[++i]));
		else if (std::strcmp(argv[i], "-bvh") == 0 && i + 1 < argc) { m_build_mode = std::strcmp(argv[++i], "fast") == 0 ? 1u : 0u; }      // no counterpart in the reference (OptiX picks its builder): fast = built on the device
		else if (argv[i][0] == '-')
			for (uint32 r = 0; r < m_renderer_names.size(); ++r) if (m_renderer_names[r] == argv[i] + 1) renderer_type = r;
	}
	if (m_aspect == 0.0f) m_aspect = float(m_res_x) / float(m_res_y);
	m_scene = scene;

	if (m_world > 1) device = m_rank;                      // one process per GPU: rank r drives device r of the node
	if (fpt_create(device, &m_ctx) != 0) throw std::runtime_error(std::string("fpt_create: ") + fpt_last_error(nullptr));
	if (m_world > 1)
	{
		check(m_ctx, fpt_comm_init(m_ctx, m_rank, m_world, m_comm_id.data()), "fpt_comm_init");
		// interleaved scanlines (tile = one row, rows round-robin over the ranks; the same rule as fermat_amd.api.tile_pixel_lists(W, H, N, (W, 1)))
		m_shards.assign(size_t(m_world), std::vector<uint32>());
		for (uint32 y = 0; y < m_res_y; ++y) for (uint32 x = 0; x < m_res_x; ++x) m_shards[size_t(y % uint32(m_world))].push_back(y * m_res_x + x);
		m_d_shard = upload(m_device_allocs, m_shards[size_t(m_rank)].data(), m_shards[size_t(m_rank)].size());
	}

	// device copy of the scene (m_mesh_d = m_mesh, src/renderer.cu:912) and of the texture views
	fpt_rendering_context_view& v = m_view;
	v.camera = scene.camera;
	v.mesh = scene.mesh;
	v.mesh.vertex_indices = upload(m_device_allocs, scene.mesh.vertex_indices, size_t(scene.mesh.num_triangles) * 4);
	v.mesh.vertex_data = upload(m_device_allocs, scene.mesh.vertex_data, size_t(scene.mesh.num_vertices) * 4);
	v.mesh.texture_indices_comp = scene.mesh.texture_indices_comp ? upload(m_device_allocs, scene.mesh.texture_indices_comp, size_t(scene.mesh.num_triangles) * 4) : nullptr;
	v.mesh.material_indices = upload(m_device_allocs, scene.mesh.material_indices, size_t(scene.mesh.num_triangles));
	v.mesh.materials = upload(m_device_allocs, scene.mesh.materials, size_t(scene.mesh.num_materials));
	v.mesh.texture_data = nullptr;                      // host-only stream (mesh-light builder)
	std::vector<fpt_texture> tex(scene.num_textures ? scene.num_textures : 1);
	std::memset(tex.data(), 0, tex.size() * sizeof(fpt_texture));
	for (uint32 t = 0; t < scene.num_textures; ++t)
	{
		tex[t] = scene.textures[t];
		if (scene.textures[t].texels) tex[t].texels = upload(m_device_allocs, scene.textures[t].texels, size_t(scene.textures[t].res_x) * scene.textures[t].res_y * 4);
	}
	v.d_textures = upload(m_device_allocs, tex.data(), tex.size());
	v.num_textures = scene.num_textures;
	v.d_dir_lights = upload(m_device_allocs, scene.dir_lights, scene.dir_lights_count);
	v.dir_lights_count = scene.dir_lights_count;
	v.d_glossy_reflectance = upload(m_device_allocs, scene.glossy_reflectance, size_t(32) * 32 * 32 * 32);
	v.res_x = m_res_x; v.res_y = m_res_y; v.aspect = m_aspect; v.exposure = m_exposure; v.gamma = m_gamma;
	// frame buffer: 8 channels + gbuffer, cleared (src/renderer.cu:609-626)
	const size_t n = size_t(m_res_x) * m_res_y;
	for (int c = 0; c < FPT_FB_NUM_CHANNELS; ++c)
	{
		void* d = nullptr; hip_check(hipMalloc(&d, n * 16), "hipMalloc"); m_device_allocs.push_back(d); hip_check(hipMemset(d, 0, n * 16), "hipMemset");
		v.fb.channels[c] = static_cast<float*>(d);
	}
	{
		void* d = nullptr;
		hip_check(hipMalloc(&d, n * 16), "hipMalloc"); m_device_allocs.push_back(d); v.fb.gbuffer_geo = static_cast<float*>(d);
		hip_check(hipMalloc(&d, n * 16), "hipMalloc"); m_device_allocs.push_back(d); v.fb.gbuffer_uv = static_cast<float*>(d);
		hip_check(hipMalloc(&d, n * 4), "hipMalloc");  m_device_allocs.push_back(d); v.fb.gbuffer_tri = static_cast<uint32_t*>(d);
		hip_check(hipMalloc(&d, n * 4), "hipMalloc");  m_device_allocs.push_back(d); v.fb.gbuffer_depth = static_cast<float*>(d);
	}
	// ray-tracing context over the device mesh (src/renderer.cu:920-933)
	m_rt_context.reset(new RTContext(m_ctx));
	check(m_ctx, fpt_rt_set_build_mode(m_ctx, m_build_mode), "fpt_rt_set_build_mode");
	m_rt_context->create_geometry(uint32(scene.mesh.num_triangles), v.mesh.vertex_indices, uint32(scene.mesh.num_vertices), v.mesh.vertex_data, 0, 0, 0, 0, v.mesh.material_indices);
	// the context's own 72-dimensional sequence: unused by the PT but it advances rand() (src/renderer.cu:949-953)
	check(m_ctx, fpt_sequence_setup(m_ctx, 72, 256, scene.samples_dir), "m_sequence.setup");
	m_renderer = m_renderer_factories[renderer_type]();
	m_renderer->init(argc, argv, *this);
}

void RenderingContext::set_sharding(int rank, int world_size, const char* comm_id)
{
	m_rank = rank; m_world = world_size;
	m_comm_id.assign(comm_id, comm_id + FPT_COMM_ID_BYTES);
}

void RenderingContext::gather_frame(int root, uint32 channel_mask)
{
	if (m_world <= 1) return;
	const size_t nw = size_t(m_world);
	std::vector<const uint32*> lists(nw); std::vector<uint32> counts(nw);
	for (int r = 0; r < m_world; ++r) { lists[size_t(r)] = m_shards[size_t(r)].data(); counts[size_t(r)] = uint32(m_shards[size_t(r)].size()); }
	check(m_ctx, fpt_gather_framebuffer(m_ctx, &m_view, root, channel_mask, lists.data(), counts.data()), "gather_frame");
	check(m_ctx, fpt_synchronize(m_ctx), "gather_frame");
}

fpt_rendering_context_view RenderingContext::view(const uint32) { return m_view; }
void RenderingContext::clear()
{
	check(m_ctx, fpt_synchronize(m_ctx), "clear");
	const size_t n = size_t(m_res_x) * m_res_y;
	for (int c = 0; c < FPT_FB_NUM_CHANNELS; ++c) hip_check(hipMemset(m_view.fb.channels[c], 0, n * 16), "clear");
}
void RenderingContext::multiply_frame(const float scale) { check(m_ctx, fpt_multiply_frame(m_ctx, &m_view, scale), "multiply_frame"); }
void RenderingContext::clamp_frame(const float max_value) { check(m_ctx, fpt_clamp_frame(m_ctx, &m_view, max_value), "clamp_frame"); }
uint8_t* RenderingContext::get_device_rgba_buffer()
{
	if (!m_d_rgba) { void* d = nullptr; hip_check(hipMalloc(&d, size_t(m_res_x) * m_res_y * 4), "hipMalloc"); m_device_allocs.push_back(d); m_d_rgba = static_cast<uint8_t*>(d); }
	return m_d_rgba;
}
fpt_mesh_lights_view RenderingContext::get_mesh_lights() { fpt_mesh_lights_view v; check(m_ctx, fpt_mesh_lights_device_view(m_ctx, &v), "get_mesh_lights"); return v; }
RenderingContext::SequenceView RenderingContext::get_sequence()
{ SequenceView v; check(m_ctx, fpt_sequence_device_view(m_ctx, &v.d_shifts, &v.n_dimensions, &v.tile_size), "get_sequence"); return v; }
void RenderingContext::compute_bbox(float lo[3], float hi[3]) const
{
	for (int k = 0; k < 3; ++k) { lo[k] = 3.0e38f; hi[k] = -3.0e38f; }
	for (int i = 0; i < m_scene.mesh.num_vertices; ++i)
		for (int k = 0; k < 3; ++k) { const float x = m_scene.mesh.vertex_data[size_t(i) * 4 + k]; lo[k] = std::min(lo[k], x); hi[k] = std::max(hi[k], x); }
}
void RenderingContext::rescale_frame(const uint32 instance) { check(m_ctx, fpt_rescale_frame(m_ctx, &m_view, instance), "rescale_frame"); }
void RenderingContext::update_variances(const uint32 instance) { check(m_ctx, fpt_update_variances(m_ctx, &m_view, instance), "update_variances"); }

void RenderingContext::update_model(const float* h_vertex_data, bool refit)
{
	// passes still pending behind a deferred render() belong to the scene as it was: render them before the vertices change under them
	check(m_ctx, fpt_synchronize(m_ctx), "update_model: synchronize");
	const size_t n = size_t(m_scene.mesh.num_vertices) * 4;
	if (h_vertex_data && n)
	{
		hip_check(hipMemcpy(const_cast<float*>(m_view.mesh.vertex_data), h_vertex_data, n * sizeof(float), hipMemcpyHostToDevice), "update_model: vertex upload");
		if (m_scene.mesh.vertex_data != h_vertex_data)
		{
			// the host view the emitter builder reads must show the same vertices: from now on the context's own copy (the caller's original array is not written to)
			m_host_vertices.assign(h_vertex_data, h_vertex_data + n);
			m_scene.mesh.vertex_data = m_host_vertices.data();
		}
	}
	else if (n)
	{
		// the documented route for a mesh edited in place on the device (get_device_mesh()): the emitter builder reads the HOST view, so it has to follow -- areas, the
		// triangle CDF and the VPL selection would otherwise come from the old vertices while shading records and VPL points are derived from the new ones
		m_host_vertices.resize(n);
		hip_check(hipMemcpy(m_host_vertices.data(), m_view.mesh.vertex_data, n * sizeof(float), hipMemcpyDeviceToHost), "update_model: vertex download");
		m_scene.mesh.vertex_data = m_host_vertices.data();
	}
	if (refit) check(m_ctx, fpt_rt_refit_geometry(m_ctx, uint32(m_scene.mesh.num_triangles), m_view.mesh.vertex_indices, uint32(m_scene.mesh.num_vertices), m_view.mesh.vertex_data), "update_model: refit");
	else m_rt_context->create_geometry(uint32(m_scene.mesh.num_triangles), m_view.mesh.vertex_indices, uint32(m_scene.mesh.num_vertices), m_view.mesh.vertex_data, 0, 0, 0, 0,
	                                   m_view.mesh.material_indices);
	m_renderer->update_scene(*this);
}

void RenderingContext::render(const uint32 instance)
{
	// gbuffer.clear(): 0xFF fill (src/framebuffer.h:178-185; src/renderer.cu:1039) -- takes its place among the passes a deferred render() still holds back
	check(m_ctx, fpt_clear_gbuffer(m_ctx, &m_view), "gbuffer clear");
	m_renderer->render(instance, *this);
	if (m_shading_mode == FPT_SHADING_FILTERED) filter(instance);          // src/renderer.cu:1045-1047 (with -batch only the call that completes a batch sees a new frame; the filter is stateless)
}

void RenderingContext::filter(const uint32 instance) { check(m_ctx, fpt_filter(m_ctx, &m_view, instance), "filter"); }

void RenderingContext::download_channel(uint32 channel, float* h_out)
{
	check(m_ctx, fpt_synchronize(m_ctx), "synchronize");
	hip_check(hipMemcpy(h_out, m_view.fb.channels[channel], size_t(m_res_x) * m_res_y * 16, hipMemcpyDeviceToHost), "download_channel");
}
void RenderingContext::download_rgba(uint8_t* h_out)
{
	const size_t n = size_t(m_res_x) * m_res_y;
	void* d = nullptr; hip_check(hipMalloc(&d, n * 4), "hipMalloc");
	const int st = fpt_to_rgba_mode(m_ctx, &m_view, m_shading_mode, static_cast<uint8_t*>(d));
	if (st == 0) { fpt_synchronize(m_ctx); hip_check(hipMemcpy(h_out, d, n * 4, hipMemcpyDeviceToHost), "download_rgba"); }
	(void)hipFree(d);
	check(m_ctx, st, "to_rgba");
}

// ---- HipPathTracer -----------------------------------------------------------------------------------------------------------
void HipPathTracer::init(int argc, char** argv, RenderingContext& renderer)
{
	// PTOptions defaults + parse (src/renderers/pathtracer.h:186-249)
	fpt_pt_options& o = m_options;
	o.max_path_length = 6; o.direct_lighting = 1; o.direct_lighting_nee = 1; o.direct_lighting_bsdf = 1; o.indirect_lighting_nee = 1; o.indirect_lighting_bsdf = 1;
	o.visible_lights = 1; o.diffuse_scattering = 1; o.glossy_scattering = 1; o.indirect_glossy = 0; o.rr = 1; o.nee_type = 1;
	for (int i = 0; i < argc; ++i)
	{
		auto is = [&](const char* f) { return std::strcmp(argv[i], f) == 0; };
		if (is("-pl") || is("-path-length") || is("-max-path-length")) o.max_path_length = uint32(std::atoi(argv[++i]));
		else if (is("-bounces")) o.max_path_length = uint32(std::atoi(argv[++i]) + 1);
		else if (is("-nee")) o.direct_lighting_nee = o.indirect_lighting_nee = std::atoi(argv[++i]) > 0;
		else if (is("-bsdf")) o.direct_lighting_bsdf = o.indirect_lighting_bsdf = std::atoi(argv[++i]) > 0;
		else if (is("-direct-nee")) o.direct_lighting_nee = std::atoi(argv[++i]) > 0;
		else if (is("-direct-bsdf")) o.direct_lighting_bsdf = std::atoi(argv[++i]) > 0;
		else if (is("-indirect-nee")) o.indirect_lighting_nee = std::atoi(argv[++i]) > 0;
		else if (is("-indirect-bsdf")) o.indirect_lighting_bsdf = std::atoi(argv[++i]) > 0;
		else if (is("-visible-lights")) o.visible_lights = std::atoi(argv[++i]) > 0;
		else if (is("-direct-lighting")) o.direct_lighting = std::atoi(argv[++i]) > 0;
		else if (is("-indirect-glossy")) o.indirect_glossy = std::atoi(argv[++i]) > 0;
		else if (is("-diffuse")) o.diffuse_scattering = std::atoi(argv[++i]) > 0;
		else if (is("-glossy")) o.glossy_scattering = std::atoi(argv[++i]) > 0;
		else if (is("-rr")) o.rr = std::atoi(argv[++i]) > 0;
		else if (is("-batch") && i + 1 < argc) m_batch = uint32(std::max(1, std::atoi(argv[++i])));
		else if (is("-passes") && i + 1 < argc) m_last_pass = uint32(std::max(0, std::atoi(argv[++i])));
		else if ((is("-nee-algorithm") || is("-nee-alg")) && i + 1 < argc)
		{
			if (std::strcmp(argv[i + 1], "mesh") == 0) o.nee_type = 0;
			else if (std::strcmp(argv[i + 1], "vpl") == 0) o.nee_type = 1;
			else if (std::strcmp(argv[i + 1], "rl") == 0) throw std::runtime_error("HipPathTracer: -nee-alg rl is outside the scope of this build");
			++i;
		}
	}
	fpt_context* ctx = renderer.get_hip_context();
	const fpt_rendering_context_view v = renderer.view(0);
	const SceneArrays& h = renderer.get_host_scene();
	// PathTracer::init (src/renderers/pathtracer_impl.h:99-178) sets up queues + sampler, then the mesh lights with
	// n_vpls = n_pixels; the two use independent generators (rand() vs LFSR), so the emitters are built first here to let
	// fpt_pt_init apply the "no emitters -> mesh NEE" rule (:165-166) in one call.
	check(ctx, fpt_mesh_lights_init(ctx, v.res_x * v.res_y, &h.mesh, h.textures, 0), "mesh_lights.init");
	check(ctx, fpt_pt_init(ctx, &o, &v, h.samples_dir, renderer.shard_pixels(), renderer.shard_count()), "PathTracer::init");
	// Passes in flight behind the unchanged render(instance) calls: the library collects consecutive instances and renders them as one wavefront when
	// `-batch N` of them are pending or when anything reads the frame (fpt_synchronize, fpt_to_rgba, fpt_filter, the gather ...).  The frame is
	// bit-identical to rendering the passes one by one, so this is the default (N = 32, fewer for large frames); `-batch 1` switches it off.
	const uint64_t n_here = renderer.shard_pixels() ? renderer.shard_count() : uint64_t(v.res_x) * v.res_y;
	const bool m_batch_asked = m_batch != 0;
	if (m_batch == 0)
	{
		m_batch = 32;
		for (int i = 0; i < argc; ++i) if (std::strcmp(argv[i], "-benchmark") == 0) m_batch = 1;      // the per-pass kernel timings of dump_speed_stats need one pass per render()
	}
	const uint32_t asked = m_batch_asked ? m_batch : 0u;
	uint64_t cap = ((1ull << 32) - 1) / std::max<uint64_t>(n_here, 1);      // 2^32 paths in flight: memory binds long before
	if (m_last_pass != 0xFFFFFFFFu) cap = std::min<uint64_t>(cap, uint64_t(m_last_pass) + 1);
	m_batch = choose_passes_in_flight(ctx, asked, m_batch, 0, v, n_here, cap, [&](uint32_t n) { return fpt_pt_set_deferred(ctx, n, &v); }, [&] { return fpt_pt_set_batch(ctx, 1, &v); });
}

// update_scene of all three renderers: what is still pending behind a deferred render() belongs to the scene as it was and is rendered first (fpt_synchronize flushes the
// PT's, the PSFPT's and the BPT's deferred passes alike); then the emitter tables are built again from the context's host mesh
static void rebuild_emitters(RenderingContext& renderer, const char* who)
{
	fpt_context* ctx = renderer.get_hip_context();
	const fpt_rendering_context_view v = renderer.view(0);
	const SceneArrays& h = renderer.get_host_scene();
	check(ctx, fpt_synchronize(ctx), who);
	check(ctx, fpt_mesh_lights_update(ctx, v.res_x * v.res_y, &h.mesh, h.textures, 0, nullptr), who);          // rebuilt only when an emitting triangle moved
}
void HipPathTracer::update_scene(RenderingContext& renderer) { rebuild_emitters(renderer, "PathTracer::update_scene"); }
void HipPSFPT::update_scene(RenderingContext& renderer) { rebuild_emitters(renderer, "PSFPT::update_scene"); }
void HipBPT::update_scene(RenderingContext& renderer) { rebuild_emitters(renderer, "BPT::update_scene"); }

void HipPathTracer::render(const uint32 instance, RenderingContext& renderer)
{
	fpt_context* ctx = renderer.get_hip_context();
	const fpt_rendering_context_view v = renderer.view(instance);
	if (m_batch > 1) { check(ctx, fpt_pt_render(ctx, instance, &v), "PathTracer::render (deferred)"); return; }      // per-pass timings would flush every pass
	check(ctx, fpt_pt_render(ctx, instance, &v), "PathTracer::render");
	fpt_pt_stats st;
	check(ctx, fpt_pt_get_stats(ctx, &st), "PathTracer stats");
	if (instance)          // frame 0 skipped, src/renderers/pathtracer_impl.h:315-322
	{
		m_sum_ms[0] += st.primary_rt_ms; m_sum_ms[1] += st.path_rt_ms; m_sum_ms[2] += st.shadow_rt_ms; m_sum_ms[3] += st.path_shade_ms; m_sum_ms[4] += st.shadow_shade_ms;
		m_timed_passes++;
	}
}

void HipPathTracer::dump_speed_stats(FILE* stats)
{
	const double n = m_timed_passes ? double(m_timed_passes) : 1.0;      // src/renderers/pathtracer_impl.h:342-350
	std::fprintf(stats, "%f, %f, %f, %f, %f\n", m_sum_ms[0] / n, m_sum_ms[1] / n, m_sum_ms[2] / n, m_sum_ms[3] / n, m_sum_ms[4] / n);
}

// ---- HipPSFPT ----------------------------------------------------------------------------------------------------------------
void HipPSFPT::init(int argc, char** argv, RenderingContext& renderer)
{
	// PTOptions::parse + PSFPTOptions::parse (src/renderers/psfpt.h:57-77)
	fpt_pt_options& o = m_options;
	o.max_path_length = 6; o.direct_lighting = 1; o.direct_lighting_nee = 1; o.direct_lighting_bsdf = 1; o.indirect_lighting_nee = 1; o.indirect_lighting_bsdf = 1;
	o.visible_lights = 1; o.diffuse_scattering = 1; o.glossy_scattering = 1; o.indirect_glossy = 0; o.rr = 1; o.nee_type = 1;
	fpt_psf_options& p = m_psf_options;
	p.psf_depth = 1; p.psf_width = 3.0f; p.psf_min_dist = 0.1f; p.psf_max_prob = 32.0f; p.psf_temporal_reuse = 64; p.firefly_filter = 100.0f;
	for (int i = 0; i < argc; ++i)
	{
		auto is = [&](const char* f) { return std::strcmp(argv[i], f) == 0; };
		if (is("-pl") || is("-path-length") || is("-max-path-length")) o.max_path_length = uint32(std::atoi(argv[++i]));
		else if (is("-bounces")) o.max_path_length = uint32(std::atoi(argv[++i]) + 1);
		else if (is("-nee")) o.direct_lighting_nee = o.indirect_lighting_nee = std::atoi(argv[++i]) > 0;
		else if (is("-bsdf")) o.direct_lighting_bsdf = o.indirect_lighting_bsdf = std::atoi(argv[++i]) > 0;
		else if (is("-direct-nee")) o.direct_lighting_nee = std::atoi(argv[++i]) > 0;
		else if (is("-direct-bsdf")) o.direct_lighting_bsdf = std::atoi(argv[++i]) > 0;
		else if (is("-indirect-nee")) o.indirect_lighting_nee = std::atoi(argv[++i]) > 0;
		else if (is("-indirect-bsdf")) o.indirect_lighting_bsdf = std::atoi(argv[++i]) > 0;
		else if (is("-visible-lights")) o.visible_lights = std::atoi(argv[++i]) > 0;
		else if (is("-direct-lighting")) o.direct_lighting = std::atoi(argv[++i]) > 0;
		else if (is("-rr")) o.rr = std::atoi(argv[++i]) > 0;
		else if ((is("-nee-algorithm") || is("-nee-alg")) && i + 1 < argc)
		{
			if (std::strcmp(argv[i + 1], "mesh") == 0) o.nee_type = 0;
			else if (std::strcmp(argv[i + 1], "vpl") == 0) o.nee_type = 1;
			else if (std::strcmp(argv[i + 1], "rl") == 0) throw std::runtime_error("HipPSFPT: -nee-alg rl is outside the scope of this build");
			++i;
		}
		else if (is("-filter-depth")) p.psf_depth = uint32(std::atoi(argv[++i]));
		else if (is("-filter-width")) p.psf_width = float(std::atof(argv[++i]));
		else if (is("-filter-min-dist")) p.psf_min_dist = float(std::atof(argv[++i]));
		else if (is("-filter-max-prob")) p.psf_max_prob = float(std::atof(argv[++i]));
		else if (is("-temporal-reuse")) p.psf_temporal_reuse = uint32(std::atoi(argv[++i]));
		else if (is("-firefly-filter") || is("-ff")) p.firefly_filter = float(std::atof(argv[++i]));
		else if (is("-batch") && i + 1 < argc) m_batch = uint32(std::max(1, std::atoi(argv[++i])));          // passes in flight, as for -pt (no reference counterpart)
		else if (is("-passes") && i + 1 < argc) m_last_pass = uint32(std::max(0, std::atoi(argv[++i])));
	}
	if (m_batch > 1 && renderer.world_size() > 1)
		throw std::runtime_error("HipPSFPT: -batch N > 1 cannot be combined with -gpus: a tile-sharded PSFPT exchanges its cache cells after every pass");
	fpt_context* ctx = renderer.get_hip_context();
	const fpt_rendering_context_view v = renderer.view(0);
	const SceneArrays& h = renderer.get_host_scene();
	check(ctx, fpt_mesh_lights_init(ctx, v.res_x * v.res_y, &h.mesh, h.textures, 0), "mesh_lights.init");
	check(ctx, fpt_psfpt_init(ctx, &o, &p, &v, h.samples_dir, renderer.shard_pixels(), renderer.shard_count()), "PSFPT::init");
	// tile sharding: the cache is shared by every pixel, so the ranks exchange the cells they touched after every pass (integer sums merged by key)
	m_sharded = renderer.world_size() > 1;
	if (m_sharded) check(ctx, fpt_psfpt_set_sharded(ctx, 1), "PSFPT::set_sharded");
	// passes in flight behind render(instance), as for -pt: cache and frame are bit-identical to sequential passes, so the library batches the calls by
	// default on one GPU (`-batch 1` switches it off; a sharded context exchanges its cells after every pass and renders pass by pass)
	{
		const uint32_t asked = m_batch;
		const uint64_t n_here = renderer.shard_pixels() ? renderer.shard_count() : uint64_t(v.res_x) * v.res_y;
		uint64_t cap = ((1ull << 32) - 1) / std::max<uint64_t>(n_here, 1);
		if (m_last_pass != 0xFFFFFFFFu) cap = std::min<uint64_t>(cap, uint64_t(m_last_pass) + 1);
		m_batch = choose_passes_in_flight(ctx, asked, m_sharded ? 1u : 32u, 1, v, n_here, cap, [&](uint32_t n) { return fpt_psfpt_set_deferred(ctx, n, &v); }, [&] { return fpt_psfpt_set_batch(ctx, 1, &v); });
	}
}

void HipPSFPT::render(const uint32 instance, RenderingContext& renderer)
{
	fpt_context* ctx = renderer.get_hip_context();
	const fpt_rendering_context_view v = renderer.view(instance);
	check(ctx, fpt_psfpt_render(ctx, instance, &v), "PSFPT::render");
	if (m_sharded)
	{
		check(ctx, fpt_psfpt_exchange_cells(ctx), "PSFPT::exchange_cells");
		check(ctx, fpt_psfpt_finish(ctx, &v), "PSFPT::finish");
	}
}

// ---- HipBPT ------------------------------------------------------------------------------------------------------------------
void HipBPT::init(int argc, char** argv, RenderingContext& renderer)
{
	// BPTOptionsBase + BPTOptions defaults and parse (src/bpt_options.h:42-92, src/renderers/bpt.h:47-72)
	fpt_bpt_options& o = m_options;
	o.max_path_length = 6; o.direct_lighting_nee = 1; o.direct_lighting_bsdf = 1; o.indirect_lighting_nee = 1; o.indirect_lighting_bsdf = 1;
	o.visible_lights = 1; o.use_vpls = 0; o.rr = 1; o.light_tracing = 1.0f; o.single_connection = 1;     // single_connection(true), src/renderers/bpt.h:62
	for (int i = 0; i < argc; ++i)
	{
		auto is = [&](const char* f) { return std::strcmp(argv[i], f) == 0; };
		if (is("-pl") || is("-path-length") || is("-max-path-length")) o.max_path_length = uint32(std::atoi(argv[++i]));
		else if (is("-bounces")) o.max_path_length = uint32(std::atoi(argv[++i]) + 1);
		else if (is("-nee")) o.direct_lighting_nee = o.indirect_lighting_nee = std::atoi(argv[++i]) > 0;
		else if (is("-direct-nee")) o.direct_lighting_nee = std::atoi(argv[++i]) > 0;
		else if (is("-direct-bsdf")) o.direct_lighting_bsdf = std::atoi(argv[++i]) > 0;
		else if (is("-indirect-nee")) o.indirect_lighting_nee = std::atoi(argv[++i]) > 0;
		else if (is("-indirect-bsdf")) o.indirect_lighting_bsdf = std::atoi(argv[++i]) > 0;
		else if (is("-batch") && i + 1 < argc) m_batch = uint32(std::max(1, std::atoi(argv[++i])));
		else if (is("-passes") && i + 1 < argc) m_last_pass = uint32(std::max(0, std::atoi(argv[++i])));
		else if (is("-visible-lights")) o.visible_lights = std::atoi(argv[++i]) > 0;
		else if (is("-use-vpls")) o.use_vpls = std::atoi(argv[++i]) > 0;
		else if (is("-light-tracing")) o.light_tracing = float(std::atof(argv[++i]));
		else if (is("-rr") || is("-RR")) o.rr = std::atoi(argv[++i]) > 0;
		else if ((is("-single-connection") || is("-sc")) && i + 1 < argc)
		{
			o.single_connection = std::atoi(argv[++i]) > 0;
		}
	}
	fpt_context* ctx = renderer.get_hip_context();
	const fpt_rendering_context_view v = renderer.view(0);
	const SceneArrays& h = renderer.get_host_scene();
	check(ctx, fpt_mesh_lights_init(ctx, v.res_x * v.res_y, &h.mesh, h.textures, 0), "mesh_lights.init");     // src/renderers/bpt.cu:53
	check(ctx, fpt_bpt_init(ctx, &o, &v, h.samples_dir, renderer.shard_pixels(), renderer.shard_count()), "BPT::init");
	const bool several = renderer.world_size() > 1;
	{
		// (-sc 0 logs L + 1 cells per eye vertex: fewer passes for the same memory; light-vertex slots are 32-bit: passes x pixels x L < 2^32)
		const uint32_t asked = m_batch;
		const uint64_t n_here = renderer.shard_pixels() ? renderer.shard_count() : uint64_t(v.res_x) * v.res_y;
		uint64_t cap = std::max<uint64_t>(1, ((1ull << 32) - 1) / std::max<uint64_t>(uint64_t(v.res_x) * v.res_y * o.max_path_length, 1));
		if (m_last_pass != 0xFFFFFFFFu) cap = std::min<uint64_t>(cap, uint64_t(m_last_pass) + 1);
		m_batch = choose_passes_in_flight(ctx, asked, several ? 1u : (o.single_connection ? 32u : 8u), 2, v, n_here, cap,
		                                  [&](uint32_t n) { return several ? fpt_bpt_set_batch(ctx, n) : fpt_bpt_set_deferred(ctx, n); }, [&] { return fpt_bpt_set_batch(ctx, 1); });
		m_deferred = !several && m_batch > 1;
	}
	// tile sharding: every rank's light sub-paths splat onto arbitrary pixels, so the splat sums are all-reduced before they are folded in
	m_sharded = renderer.world_size() > 1 && o.light_tracing != 0.0f;
	if (m_sharded) check(ctx, fpt_bpt_set_deferred_splats(ctx, 1), "BPT::init (sharded)");
	// -sc 1 draws the connections from the light vertices of ALL light paths: the ranks exchange theirs after the light sub-paths, so that the image
	// does not depend on the number of GPUs
	m_shared_lv = renderer.world_size() > 1 && o.single_connection;
	if (m_shared_lv) check(ctx, fpt_bpt_set_shared_light_vertices(ctx, 1), "BPT::init (shared light vertices)");
}

// the integer all-reduce of the light-tracing splat sums (3 x int64 per pixel per pass in flight), then the fold into the frame
void HipBPT::finish_sharded_pass(fpt_context* ctx, const fpt_rendering_context_view& v, uint32 passes_in_flight)
{
	check(ctx, fpt_bpt_allreduce_splats(ctx, uint64_t(3) * v.res_x * v.res_y * passes_in_flight), "BPT::render (splat all-reduce)");
	check(ctx, fpt_bpt_resolve_splats(ctx, &v), "BPT::render (splat resolve)");
}

void HipBPT::render(const uint32 instance, RenderingContext& renderer)
{
	fpt_context* ctx = renderer.get_hip_context();
	const fpt_rendering_context_view v = renderer.view(instance);
	if (m_deferred) { check(ctx, fpt_bpt_render(ctx, instance, &v), "BPT::render (deferred)"); return; }
	if (m_batch > 1)
	{
		if ((instance + 1) % m_batch == 0 || instance >= m_last_pass)
			for (; m_next_pass <= instance; )
			{
				const uint32 n = std::min(m_batch, instance + 1 - m_next_pass);
				if (n > 1) check(ctx, fpt_bpt_render_batch(ctx, m_next_pass, n, &v), "BPT::render (-batch)");
				else       check(ctx, fpt_bpt_render(ctx, m_next_pass, &v), "BPT::render");
				if (m_shared_lv) { check(ctx, fpt_bpt_exchange_light_vertices(ctx), "BPT::render (light-vertex exchange)"); check(ctx, fpt_bpt_finish(ctx, &v), "BPT::render (finish)"); }
				if (m_sharded) finish_sharded_pass(ctx, v, std::max(m_batch, 1u));
				m_next_pass += n;
			}
		return;
	}
	check(ctx, fpt_bpt_render(ctx, instance, &v), "BPT::render");
	if (m_shared_lv) { check(ctx, fpt_bpt_exchange_light_vertices(ctx), "BPT::render (light-vertex exchange)"); check(ctx, fpt_bpt_finish(ctx, &v), "BPT::render (finish)"); }
	if (m_sharded) finish_sharded_pass(ctx, v, 1);
}

} // namespace fermat

extern "C" uint32_t register_plugin(fermat::RenderingContext& renderer)
{
	return renderer.register_renderer("hip-pt", &fermat::HipPathTracer::factory);
}

// ---- C hooks so the C++ mirror can be driven from a test harness without a C++ toolchain on the other side ---------------------
extern "C" {
static thread_local std::string g_host_error;
void* fpt_host_context_create(int argc, char** argv, const fermat::SceneArrays* scene)
{
	try { fermat::RenderingContext* c = new fermat::RenderingContext(); try { c->init(argc, argv, *scene); } catch (...) { delete c; throw; } return c; }
	catch (const std::exception& e) { g_host_error = e.what(); return nullptr; }
}
int fpt_host_context_render(void* h, uint32_t instance)
{
	try { static_cast<fermat::RenderingContext*>(h)->render(instance); return 0; } catch (const std::exception& e) { g_host_error = e.what(); return 1; }
}
int fpt_host_context_download(void* h, uint32_t channel, float* out)
{
	try { static_cast<fermat::RenderingContext*>(h)->download_channel(channel, out); return 0; } catch (const std::exception& e) { g_host_error = e.what(); return 1; }
}
int fpt_host_context_download_rgba(void* h, uint8_t* out)
{
	try { static_cast<fermat::RenderingContext*>(h)->download_rgba(out); return 0; } catch (const std::exception& e) { g_host_error = e.what(); return 1; }
}
int fpt_host_context_update_model(void* h, const float* h_vertex_data, int refit)
{
	try { static_cast<fermat::RenderingContext*>(h)->update_model(h_vertex_data, refit != 0); return 0; } catch (const std::exception& e) { g_host_error = e.what(); return 1; }
}
void fpt_host_context_destroy(void* h) { delete static_cast<fermat::RenderingContext*>(h); }
const char* fpt_host_last_error() { return g_host_error.c_str(); }
}
