// renderer_interface.h — C++ host mirror of Fermat's renderer plugin surface for the -pt path.
//
//   RendererInterface, RendererFactoryFunction   src/renderer_interface.h:38-88 (same 10 virtuals, same defaults)
//   register_plugin                               src/renderers/hellopt_plugin.cpp:35-39, src/renderer.cu:441-460
//   RenderingContext (the subset a renderer may call)   src/renderer.h:52-228
//   RTContext                                     src/rt.h:55-105
// The classes are written against the C-ABI of include/fermat_pt_hip.h; they own no kernels.  Scene import (OBJ/.fa/...)
// lives in scene_io.{h,cpp}; the context is initialised from host arrays in MeshView layout.
#pragma once
#include "../../../include/fermat_host.h"
#include <cstdio>
#include <memory>
#include <string>
#include <vector>

namespace fermat {

typedef uint32_t uint32;

struct RenderingContext;
struct FBufferStorage;
struct RendererInterface;

typedef RendererInterface* (*RendererFactoryFunction)();

// The vtable is exactly the reference's: ten virtuals in its order and NO virtual destructor (src/renderer_interface.h:45-88; the
// reference destroys renderers through destroy(), "delete this", src/renderers/pathtracer.h:286), so a host compiled against the
// reference's header calls the same slots.
struct RendererInterface
{
	virtual uint32 auxiliary_channel_count() { return 0; }
	virtual void register_auxiliary_channels(FBufferStorage& fbuffer, const uint32 channel_offset) {}
	virtual void init(int argc, char** argv, RenderingContext& renderer) {}
	virtual void update_scene(RenderingContext& renderer) {}
	virtual void render(const uint32 instance, RenderingContext& renderer) {}
	virtual void keyboard(unsigned char character, int x, int y, bool& invalidate) {}
	virtual void destroy() {}
	virtual void mouse(RenderingContext& renderer, int button, int state, int x, int y) {}
	virtual void draw(RenderingContext& renderer) {}
	virtual void dump_speed_stats(FILE* stats) {}
};

// src/ray.h:42-76
struct Ray       { float origin[3]; float tmin;  float dir[3]; float tmax; };
struct MaskedRay { float origin[3]; uint32 mask; float dir[3]; float tmax; };
struct Hit       { float t; int32_t triId; float u, v; };
static_assert(sizeof(Ray) == 32 && sizeof(MaskedRay) == 32 && sizeof(Hit) == 16 && sizeof(fpt_ray) == 32 && sizeof(fpt_hit) == 16, "ray / hit layouts (SURVEY Appendix B)");

// RTContext (src/rt.h:55-105): the calls the PT makes, forwarded to the HIP traversal kernels
struct RTContext
{
	explicit RTContext(fpt_context* c) : ctx(c) {}
	void create_geometry(const uint32 tri_count, const int* index_ptr, const uint32 vertex_count, const float* vertex_ptr,
	                     const int* normal_index_ptr, const float* normal_vertex_ptr, const int* tex_index_ptr, const float* tex_vertex_ptr,
	                     const int* material_index_ptr);
	// the reference's four overloads (src/rt.h:99-102): Ray and MaskedRay are the same 32 bytes (src/ray.h:42-68) and the closest-hit
	// trace reads MaskedRay::mask as tmin either way (src/rt.cpp:558-609), so both forward to fpt_rt_trace
	void trace(const uint32 count, const Ray* rays, Hit* hits);
	void trace(const uint32 count, const MaskedRay* rays, Hit* hits);
	void trace_shadow(const uint32 count, const MaskedRay* rays, Hit* hits);
	void trace_shadow(const uint32 count, const MaskedRay* rays, uint32* binary_hits);
	fpt_context* ctx;
};

// host arrays in MeshView layout + camera etc.: what RenderingContextImpl::init has after loading and pre-processing a scene
typedef fpt_scene_arrays SceneArrays;

struct RenderingContext
{
	RenderingContext();
	~RenderingContext();

	// RenderingContext::init (src/renderer.cu:467-991): parses -r W H, -a aspect, -<renderer name>, registers built-ins,
	// uploads the scene, creates the RTContext geometry, sets up the 72-dimensional context sequence, inits the renderer.
	void init(int argc, char** argv, const SceneArrays& scene);
	void render(const uint32 instance);                                   // src/renderer.cu:1029-1056
	// RenderingContextImpl::update_model (src/renderer.cu:999-1017): the acceleration structure is built again over the DEVICE mesh (whose vertex data a host has
	// edited -- get_device_mesh() -- or hands over here as host float4s, which also refreshes the context's host copy), then the renderer's update_scene runs
	// refit = true: the vertices moved and nothing else changed -- the tree is refitted in place (fpt_rt_refit_geometry: tens of milliseconds) instead of built again
	void update_model(const float* h_vertex_data = nullptr, bool refit = false);
	uint32 register_renderer(const char* name, RendererFactoryFunction factory);   // :1020-1025

	struct uint2v { uint32 x, y; };
	uint2v res() const { uint2v r; r.x = m_res_x; r.y = m_res_y; return r; }
	fpt_rendering_context_view view(const uint32 instance);               // :1058-1084
	void filter(const uint32 instance);                                   // EAW denoiser -> FILTERED_C, :1099-1151
	uint32& get_shading_mode() { return m_shading_mode; }                 // ShadingMode (src/renderer_view.h:61-76), :1373
	void clear();                                                         // zero the frame buffer, :381-389
	void multiply_frame(const float scale);                               // :391-401
	void rescale_frame(const uint32 instance);                            // :403-416
	void clamp_frame(const float max_value);                              // :418-427
	fpt_camera& get_camera() { return m_view.camera; }                    // src/renderer.h:61-73
	void set_aspect_ratio(const float v) { m_aspect = m_view.aspect = v; }
	float get_aspect_ratio() const { return m_aspect; }
	void set_exposure(const float v) { m_exposure = m_view.exposure = v; }
	float get_exposure() const { return m_exposure; }
	void set_gamma(const float v) { m_gamma = m_view.gamma = v; }
	float get_gamma() const { return m_gamma; }
	// the storage accessors of src/renderer.h:117-173, as the flat views of this boundary: frame buffer (device channels + gbuffer), the RGBA output
	// buffer (device, 4 bytes per pixel, allocated on first use), the mesh (host arrays / device arrays in MeshView layout), the texture views, the
	// emitter tables, the context's tiled sequence and the scene's bounding box
	fpt_framebuffer_view& get_frame_buffer() { return m_view.fb; }
	uint8_t* get_device_rgba_buffer();
	const fpt_mesh_view& get_host_mesh() const { return m_scene.mesh; }
	const fpt_mesh_view& get_device_mesh() const { return m_view.mesh; }
	const fpt_texture* get_host_texture_views() const { return m_scene.textures; }
	const fpt_texture* get_device_texture_views() const { return m_view.d_textures; }
	fpt_mesh_lights_view get_mesh_lights();
	struct SequenceView { const float* d_shifts; uint32 n_dimensions, tile_size; };
	SequenceView get_sequence();
	void compute_bbox(float lo[3], float hi[3]) const;                    // cugar::Bbox3f compute_bbox(), src/renderer.cu:1359-1371
	void update_variances(const uint32 instance);                         // :431-437
	RTContext* get_rt_context() const { return m_rt_context.get(); }
	fpt_context* get_hip_context() const { return m_ctx; }
	// multi-GPU (one process per GPU; no counterpart in the single-GPU reference): call set_sharding before init.  The frame is
	// split by interleaved scanlines (row y belongs to rank y % world), every rank renders its rows with absolute pixel coordinates, and
	// gather_frame completes the root's frame buffer over RCCL (fpt_gather_framebuffer) before the image is read.
	void set_sharding(int rank, int world_size, const char* comm_id /*[FPT_COMM_ID_BYTES]*/);
	int rank() const { return m_rank; }
	int world_size() const { return m_world; }
	const uint32* shard_pixels() const { return m_world > 1 ? m_d_shard : nullptr; }       // device list of this rank's pixels, NULL = the whole frame
	uint32 shard_count() const { return m_world > 1 ? uint32(m_shards[size_t(m_rank)].size()) : m_res_x * m_res_y; }
	void gather_frame(int root = 0, uint32 channel_mask = 1u << FPT_FB_COMPOSITED_C);
	const SceneArrays& get_host_scene() const { return m_scene; }
	void download_channel(uint32 channel, float* h_out);                  // float4 per pixel
	void download_rgba(uint8_t* h_out);                                   // to_rgba, :83-106

	fpt_context* m_ctx;
	std::unique_ptr<RTContext> m_rt_context;
	RendererInterface* m_renderer;
	std::vector<std::string> m_renderer_names;
	std::vector<RendererFactoryFunction> m_renderer_factories;
	SceneArrays m_scene;
	std::vector<float> m_host_vertices;          // update_model: the host mesh's vertex data once a caller has handed over new ones
	uint32 m_res_x, m_res_y;
	uint32 m_shading_mode;               // FPT_SHADING_*; kFiltered makes render() run filter() before to_rgba (:1045-1049)
	float m_aspect, m_exposure, m_gamma;
	std::vector<void*> m_device_allocs;
	fpt_rendering_context_view m_view;
	int m_rank = 0, m_world = 1;
	std::string m_comm_id;
	std::vector<std::vector<uint32>> m_shards;      // every rank's pixel list (the gather needs all of them on every rank)
	uint32* m_d_shard = nullptr;
	uint8_t* m_d_rgba = nullptr;
	uint32 m_build_mode = 0;          // `-bvh fast|quality`: fpt_rt_set_build_mode (quality = the host SAH builder, the default; fast = Morton radix tree + collapse on the device)
};

// the MI355X path tracer behind RendererInterface (PathTracer, src/renderers/pathtracer.h:255-305)
struct HipPathTracer final : RendererInterface
{
	void init(int argc, char** argv, RenderingContext& renderer) override;
	// the reference's PathTracer inherits the empty default and leaves "TODO: update m_mesh_lights if needed!" (src/renderer.cu:1011); here the passes still pending
	// behind render() are rendered against the OLD scene first, and the emitter tables are built again from the context's host mesh, so a moved emitter is followed
	void update_scene(RenderingContext& renderer) override;
	void render(const uint32 instance, RenderingContext& renderer) override;
	void destroy() override { delete this; }
	void dump_speed_stats(FILE* stats) override;
	static RendererInterface* factory() { return new HipPathTracer(); }

	fpt_pt_options m_options;
	double m_sum_ms[5] = { 0, 0, 0, 0, 0 };
	uint32 m_timed_passes = 0;
	// `-batch N` (no counterpart in the reference): the library keeps up to N passes in flight behind render() (fpt_pt_set_deferred): render(i)
	// returns at once, and the pending passes are rendered as one wavefront when N of them wait or when anything looks at the frame.  The frame is
	// bit-identical to one pass per call in every shading mode; `-benchmark` reads per-pass kernel timings and turns it off.
	uint32 m_batch = 0;                   // `-batch N`: passes the library may keep in flight behind render() (fpt_pt_set_deferred); 0 = default (32), 1 = off
	uint32 m_next_pass = 0;
	uint32 m_last_pass = 0xFFFFFFFFu;     // `-passes`: the last instance the CLI loop will ask for (src/main.cu:167 runs i = 0..passes)
};

// the MI355X path-space-filtering path tracer behind RendererInterface (PSFPT, src/renderers/psfpt.h:80-130); `-psfpt`
struct HipPSFPT final : RendererInterface
{
	void init(int argc, char** argv, RenderingContext& renderer) override;
	void update_scene(RenderingContext& renderer) override;      // as HipPathTracer's: pending passes first, then the emitter tables again
	void render(const uint32 instance, RenderingContext& renderer) override;
	void destroy() override { delete this; }
	static RendererInterface* factory() { return new HipPSFPT(); }

	fpt_pt_options m_options;
	fpt_psf_options m_psf_options;
	bool m_sharded = false;      // RenderingContext::set_sharding: the ranks exchange their cache cells after every pass
	uint32 m_batch = 0, m_next_pass = 0, m_last_pass = 0xFFFFFFFFu;      // -batch N: passes the library may keep in flight behind render() (fpt_psfpt_set_deferred); 0 = default (32), 1 = off
};

// the MI355X bidirectional path tracer behind RendererInterface (BPT, src/renderers/bpt.h:74-108); `-bpt` on the command line.
// `-sc 1` (one connection per eye vertex into the list of all light vertices) is the default, as in the reference; `-sc 0` = all connections.
struct HipBPT final : RendererInterface
{
	void init(int argc, char** argv, RenderingContext& renderer) override;
	void update_scene(RenderingContext& renderer) override;      // as HipPathTracer's
	void render(const uint32 instance, RenderingContext& renderer) override;
	void destroy() override { delete this; }
	static RendererInterface* factory() { return new HipBPT(); }

	fpt_bpt_options m_options;
	uint32 m_batch = 0;          // `-batch N`, as in HipPathTracer; 0 = default: 32 (-sc 1) or 8 (-sc 0) on one GPU, 1 on several
	bool m_deferred = false;     // one GPU: the library batches behind render() (fpt_bpt_set_deferred); several: render() runs the phases of a batch itself
	uint32 m_next_pass = 0, m_last_pass = 0xFFFFFFFFu;
	bool m_shared_lv = false;    // -sc 1 on several GPUs: the ranks exchange their light vertices (fpt_bpt_exchange_light_vertices)
	bool m_sharded = false;      // tile-sharded run with light tracing: splat sums are all-reduced over the ranks after every render
	void finish_sharded_pass(fpt_context* ctx, const fpt_rendering_context_view& v, uint32 passes_in_flight);
};

} // namespace fermat

// plugin entry point with the reference's name and meaning (src/renderers/hellopt_plugin.cpp:35-39)
extern "C" uint32_t register_plugin(fermat::RenderingContext& renderer);
