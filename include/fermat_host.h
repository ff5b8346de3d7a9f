/* fermat_host.h — C hooks of the C++ host mirror (fermat_amd/csrc/host/): scene front-end + rendering context.
 *
 * These are NOT the hot-path boundary (that is fermat_pt_hip.h); they let a harness without a C++ toolchain drive the
 * classes that mirror the reference's host program:
 *   RenderingContext::init / render / get_device_rgba_buffer      src/renderer.cu:467-991, 1029-1056 ; src/renderer.h:52-228
 *   load_scene / loadModel / texture set-up                        src/mesh/fermat_loader.cpp:46-360 ; src/mesh/MeshStorage.cpp:129-244 ; src/renderer.cu:690-870
 *   -c camera files, write_tga                                     src/renderer.cu:508-522 ; contrib/cugar/image/tga.cpp
 * Errors: NULL / non-zero return, message from fpt_host_last_error() / fpt_host_scene_last_error().
 */
#ifndef FERMAT_HOST_H
#define FERMAT_HOST_H
#include "fermat_pt_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* host arrays in MeshView layout + camera etc. (what RenderingContextImpl::init holds after loading and pre-processing a scene) */
typedef struct fpt_scene_arrays
{
	fpt_mesh_view mesh;                 /* HOST pointers */
	const fpt_texture* textures; uint32_t num_textures;          /* HOST texel pointers */
	const fpt_dir_light* dir_lights; uint32_t dir_lights_count;
	const float* glossy_reflectance;    /* 32^4 floats */
	fpt_camera camera;
	const char* samples_dir;            /* directory holding samples-<z>.dat */
} fpt_scene_arrays;

/* scene front-end: .fa scripts and .obj/.mtl models, TGA/PFM textures; data_dir holds glossy_reflectance.dat and samples-N.dat */
void*       fpt_host_scene_load(const char* filename, const char* data_dir);
const char* fpt_host_scene_last_error(void);
void        fpt_host_scene_free(void* scene);
int         fpt_host_scene_arrays(const void* scene, const fpt_camera* override_camera, fpt_scene_arrays* out);   /* pointers live as long as the scene */
int         fpt_host_scene_counts(const void* scene, uint32_t out[4]);     /* cameras, directional lights, textures, groups */
const char* fpt_host_scene_texture_name(const void* scene, uint32_t i);
const char* fpt_host_scene_material_name(const void* scene, uint32_t i);
const char* fpt_host_scene_group_name(const void* scene, uint32_t i, int32_t* first_triangle, int32_t* end_triangle);
int         fpt_host_load_camera(const char* filename, fpt_camera* out);
int         fpt_host_write_tga(const char* filename, int width, int height, const unsigned char* pixels, int channels);

/* rendering context (argv as the reference's command line: -r W H, -a aspect, -pt, -pl/-bounces/... PT flags) */
void*       fpt_host_context_create(int argc, char** argv, const fpt_scene_arrays* scene);
int         fpt_host_context_render(void* context, uint32_t instance);
/* RenderingContextImpl::update_model (src/renderer.cu:999-1017): new vertex data (float4 per vertex, or NULL when the device mesh was edited in place) -> the
 * acceleration structure is built again (or, refit = 1, refitted in place: only vertices moved), then RendererInterface::update_scene (slot 3) runs; passes still pending behind render() are rendered first */
int         fpt_host_context_update_model(void* context, const float* h_vertex_data, int refit /* 1: fpt_rt_refit_geometry instead of a new build */);
int         fpt_host_context_download(void* context, uint32_t channel, float* out /* float4 per pixel */);
int         fpt_host_context_download_rgba(void* context, uint8_t* out);
void        fpt_host_context_destroy(void* context);
const char* fpt_host_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
