// scene_io.h — Linux scene/asset front-end of the host mirror: the step BEFORE the hot path.
//
//   MeshStorage, loadModel, loadMaterials          src/mesh/MeshStorage.{h,cpp}:129-244 ; OBJ/MTL semantics src/mesh/MeshBase.cpp:492-1400
//   merge / transform / add_per_triangle_*          src/mesh/MeshStorage.cpp:449-648
//   compress_tex / unify_vertex_attributes / apply_material_flags   src/mesh/MeshStorage.cpp:246-331, 430-445, 651-840
//   load_scene (.fa scripts)                        src/mesh/fermat_loader.cpp:46-360
//   load_tga / write_tga / load_pfm                 contrib/cugar/image/{tga,pfm}.cpp
//   texture set-up, glossy_reflectance.dat          src/renderer.cu:646-660, 780-870
// Everything here is host code that runs once per scene; its output is the SceneArrays block (MeshView layout) that
// RenderingContext::init uploads.  Importers that need third-party libraries (Assimp, the pbrt parser, rply) are not built.
#pragma once
#include "renderer_interface.h"
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace fermat {

struct MeshException : std::runtime_error { explicit MeshException(const std::string& s) : std::runtime_error(s) {} };

struct MeshTextureMap { std::string name; float scaling[2] = { 1.0f, 1.0f }; };

// MeshMaterialParams (src/mesh/MeshBase.h:85-140, defaults src/mesh/MeshBase.cpp:354-412)
struct MeshMaterialParams
{
	std::string name = "null-material";
	float diffuse[3] = { 0.7f, 0.7f, 0.7f }, diffuse_trans[3] = { 0, 0, 0 }, ambient[3] = { 0.2f, 0.2f, 0.2f }, specular[3] = { 0, 0, 0 };
	float emissive[3] = { 0, 0, 0 }, reflectivity[3] = { 0, 0, 0 };
	float phong_exponent = 0.0f, index_of_refraction = 1.0f, opacity = 1.0f;
	int flags = 0, shading_type = 0;
	MeshTextureMap ambient_map, diffuse_map, diffuse_trans_map, specular_map, emissive_map, opacity_map, bump_map;
};

// host mesh in the reference's layout: 4 ints per triangle for each index stream (w of the vertex stream = material flags)
struct MeshStorage
{
	int num_triangles = 0, num_vertices = 0, num_normals = 0, num_texture_coordinates = 0;
	std::vector<int>   vertex_indices, normal_indices, texture_indices, material_indices;
	std::vector<float> vertex_data;        // float4 per vertex (w: packed normal after unify_vertex_attributes)
	std::vector<float> normal_data;        // float3 per normal
	std::vector<float> texture_data;       // float2 per texture coordinate
	std::vector<int>   texture_indices_comp;
	float tex_bias[2] = { 0, 0 }, tex_scale[2] = { 1, 1 };
	std::vector<fpt_material> materials;
	std::vector<std::string>  material_names;
	std::vector<std::string>  textures;                  // texture file names, in first-use order
	std::map<std::string, uint32> textures_map;
	std::vector<std::string>  group_names;
	std::vector<int>          group_offsets;             // num_groups + 1

	void compress_tex();
};

void loadModel(const std::string& filename, MeshStorage& mesh);                 // .obj (+ its mtllib)
void loadMaterials(const std::string& filename, MeshStorage& mesh);             // appends a .mtl library
void merge(MeshStorage& mesh, const MeshStorage& other);
void transform(MeshStorage& mesh, const float mat[16]);
void add_per_triangle_normals(MeshStorage& mesh);
void add_per_triangle_texture_coordinates(MeshStorage& mesh);
void unify_vertex_attributes(MeshStorage& mesh);
void apply_material_flags(MeshStorage& mesh);

void load_scene(const char* filename, MeshStorage& mesh, std::vector<fpt_camera>& cameras, std::vector<fpt_dir_light>& dir_lights,
                std::vector<std::string>& dirs, std::vector<std::string>& scene_dirs);

// images: RGB(A) bytes, rows as stored in the file
unsigned char* load_tga(const char* filename, int* width, int* height, int* bits);      // delete[] the result
bool write_tga(const char* filename, int width, int height, const unsigned char* pixdata, int channels /*3 = RGB, 4 = RGBA*/);
float* load_pfm(const char* filename, uint32* xres, uint32* yres);                       // delete[] the result

bool load_camera_file(const char* filename, fpt_camera& camera);                        // -c file, src/renderer.cu:508-522

// everything RenderingContextImpl::init holds after loading a scene (src/renderer.cu:690-870)
struct HostScene
{
	MeshStorage mesh;
	std::vector<fpt_camera> cameras;
	std::vector<fpt_dir_light> dir_lights;
	std::vector<std::vector<float>> texels;     // float4 texels per texture (empty = texture missing: n_levels == 0)
	std::vector<fpt_texture> texture_views;
	std::vector<float> glossy_reflectance;
	std::string data_dir;
	float bbox[6];

	// load + pre-process (compress_normals/compress_tex/unify/apply flags) + textures + glossy table
	void load(const char* filename, const char* data_dir);
	SceneArrays arrays(const fpt_camera* override_camera) const;
};

} // namespace fermat
