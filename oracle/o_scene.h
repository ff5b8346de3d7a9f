// ORACLE — TEST INFRASTRUCTURE ONLY (see o_math.h header).
//
// o_scene.h : CPU restatement of the scene "views" the -pt kernels read, and of the per-vertex set-up code:
//   mesh view / differential geometry : src/mesh/MeshView.h:96-145, src/mesh_utils.h:184-310, src/mesh/MeshCompression.h:52-68
//   texture lookup                    : src/texture_view.h:107-118,170-202
//   lights                            : src/lights.h:59-76,249-294,299-431 ; src/direct_lighting_mesh.h:41-111
//   eye vertex                        : src/bpt_utils.h:585-642
//   camera                            : src/camera.h:120-128,141-171,231-252
#pragma once
#include "o_bsdf.h"
#include <vector>

namespace orc {

// host mesh view after compress_normals / compress_tex / unify_vertex_attributes / apply_material_flags
// (src/renderer.cu:735-744). Only the arrays setup_differential_geometry touches in the UNIFIED_VERTEX_ATTRIBUTES,
// TEX_COORD_COMPRESSION build are kept.
struct Mesh
{
	i32 num_triangles;
	i32 num_vertices;
	const i32*   vertex_indices;        // int4 per triangle; .w = material flags
	const float* vertex_data;           // float4 per vertex;  .w = bits(pack_normal)
	const i32*   texture_indices_comp;  // int4 per triangle (packed half2 per corner, -1 = missing) or NULL
	const i32*   material_indices;      // int per triangle
	const Material* materials;
	i32 num_materials;
	float tex_bias[2], tex_scale[2];
	const float* texture_data;          // float2 per vertex after unify (host light builder only) or NULL
};

struct Texture { const float* texels; u32 res_x, res_y; };      // float4 texels, LOD 0 only (src/texture_view.h:57-84)

// src/vertex.h:105-140 (VertexGeometry = DifferentialGeometry + position + texture coords)
struct VertexGeometry : Frame
{
	V3 position;
	V4 texture_coords;
};

inline V3 load_vertex(const Mesh& m, i32 i) { const float* v = m.vertex_data + 4 * i; return V3(v[0], v[1], v[2]); }

// src/mesh/MeshCompression.h:52-68 (TEX_COORD_COMPRESSION_HALF)
inline V2 decompress_tex_coord(const Mesh& m, u32 packed)
{
	const float tx = h2f(uint16_t(packed & 0xffffu)), ty = h2f(uint16_t(packed >> 16));
	V2 r; r.x = tx * m.tex_scale[0] + m.tex_bias[0]; r.y = ty * m.tex_scale[1] + m.tex_bias[1];
	return r;
}

// src/mesh_utils.h:184-310 (lightmap coordinates are not read by the PT and are omitted)
inline void setup_differential_geometry(const Mesh& mesh, u32 tri_id, float u, float v, VertexGeometry* geom, float* pdf = 0)
{
	const i32* tri = mesh.vertex_indices + 4 * tri_id;
	const float* p0 = mesh.vertex_data + 4 * tri[0];
	const float* p1 = mesh.vertex_data + 4 * tri[1];
	const float* p2 = mesh.vertex_data + 4 * tri[2];
	const V3 vp0(p0[0], p0[1], p0[2]), vp1(p1[0], p1[1], p1[2]), vp2(p2[0], p2[1], p2[2]);

	geom->position = vp2 * (1.0f - u - v) + vp0 * u + vp1 * v;
	const V3 dp_du = vp0 - vp2;
	const V3 dp_dv = vp1 - vp2;
	geom->normal_g = normalize(cross(dp_du, dp_dv));
	if (pdf) *pdf = 2.0f / length(cross(dp_du, dp_dv));

	const V3 vn0 = unpack_normal(f2bits(p0[3]));
	const V3 vn1 = unpack_normal(f2bits(p1[3]));
	const V3 vn2 = unpack_normal(f2bits(p2[3]));
	const V3 N = normalize(vn2 * (1.0f - u - v) + vn0 * u + vn1 * v);
	geom->normal_s = N;
	geom->tangent  = orthogonal(N);
	geom->binormal = cross(N, geom->tangent);

	if (mesh.texture_indices_comp)
	{
		const i32* t = mesh.texture_indices_comp + 4 * tri_id;
		V2 vt0, vt1, vt2;
		if (t[0] >= 0) vt0 = decompress_tex_coord(mesh, u32(t[0])); else { vt0.x = 1.0f; vt0.y = 0.0f; }
		if (t[1] >= 0) vt1 = decompress_tex_coord(mesh, u32(t[1])); else { vt1.x = 0.0f; vt1.y = 1.0f; }
		if (t[2] >= 0) vt2 = decompress_tex_coord(mesh, u32(t[2])); else { vt2.x = 0.0f; vt2.y = 0.0f; }
		const float w = 1.0f - u - v;
		const float sx = vt2.x * w + vt0.x * u + vt1.x * v;
		const float sy = vt2.y * w + vt0.y * u + vt1.y * v;
		geom->texture_coords = V4(sx, sy, 0.0f, 0.0f);
	}
	else
		geom->texture_coords = V4(u, v, 0.0f, 0.0f);
}

// src/mesh_utils.h:322-337
inline V3 interpolate_position(const Mesh& mesh, u32 tri_id, float u, float v)
{
	const i32* tri = mesh.vertex_indices + 4 * tri_id;
	const V3 vp0 = load_vertex(mesh, tri[0]), vp1 = load_vertex(mesh, tri[1]), vp2 = load_vertex(mesh, tri[2]);
	return vp2 * (1.0f - u - v) + vp0 * u + vp1 * v;
}

inline V4 texel(const Texture& t, u32 x, u32 y) { const float* c = t.texels + 4 * (size_t(y) * t.res_x + x); return V4(c[0], c[1], c[2], c[3]); }

// src/texture_view.h:170-202 (+ :107-118)
inline V4 bilinear_texture_lookup(V4 st, const TexRef& ref, const Texture* textures, V4 default_value)
{
	if (ref.texture == 0xFFFFFFFFu || textures[ref.texture].texels == 0) return default_value;
	st.x *= ref.sx;
	st.y *= ref.sy;
	st.x = mod1(st.x);
	st.y = mod1(st.y);
	const Texture& tex = textures[ref.texture];
	const u32 x = minu(f2u(st.x * float(tex.res_x)), tex.res_x - 1);
	const u32 y = minu(f2u(st.y * float(tex.res_y)), tex.res_y - 1);
	const u32 xx = (x + 1) % tex.res_x;
	const u32 yy = (y + 1) % tex.res_y;
	const V4 q0 = texel(tex, x, y), q1 = texel(tex, xx, y), q2 = texel(tex, x, yy), q3 = texel(tex, xx, yy);
	const float u = mod1(st.x * float(tex.res_x));
	const float v = mod1(st.y * float(tex.res_y));
	return (q0 * (1 - u) + q1 * u) * (1 - v) + (q2 * (1 - u) + q3 * u) * v;
}

// src/lights.h:59-76 — declared here as a packed 16-byte POD (the inherited layout is ABI-dependent)
struct VPL { float u, v; u32 prim_id; float E; };
static_assert(sizeof(VPL) == 16, "VPL layout");
inline float vpl_pdf(V4 E) { return maxf(fabsf(E.x), maxf(fabsf(E.y), fabsf(E.z))); }

// src/lights.h:249-294
struct DirectionalLight { V3 dir; V3 color; };

// contrib/cugar/basic/algorithms.h:138-199
inline u32 upper_bound_index(float x, const float* begin, u32 n)
{
	const float* b = begin;
	u32 count = n;
	while (count > 0)
	{
		const u32 step = count / 2;
		const float* it = b + step;
		if (!(x < *it)) { b = it + 1; count -= step + 1; }
		else count = step;
	}
	return u32(b - begin);
}

// src/lights.h:299-431 (MeshLight in its two instantiations: VPL set when n_vpls != 0, triangle CDF otherwise)
struct MeshLight
{
	u32 n_prims; const float* prims_cdf; const float* prims_inv_area;
	const Mesh* mesh; const Texture* textures;
	u32 n_vpls; const VPL* vpls; float norm;

	// map_impl(prim, uv, geom*, pdf, edf) : :363-398
	void map(u32 prim_id, float u, float v, VertexGeometry* geom, float* pdf, Edf* edf) const
	{
		setup_differential_geometry(*mesh, prim_id, u, v, geom);
		map_geom(prim_id, *geom, pdf, edf);
	}
	// map_impl(prim, uv, const geom&, pdf, edf) : :400-424
	void map_geom(u32 prim_id, const VertexGeometry& geom, float* pdf, Edf* edf) const
	{
		if (n_vpls || n_prims)
		{
			Material material = mesh->materials[mesh->material_indices[prim_id]];
			material.emissive = material.emissive * bilinear_texture_lookup(geom.texture_coords, material.emissive_map, textures, V4(1, 1, 1, 1));
			if (n_vpls) *pdf = vpl_pdf(material.emissive) / norm;
			else        *pdf = (prims_cdf[prim_id] - (prim_id ? prims_cdf[prim_id - 1] : 0)) * prims_inv_area[prim_id];
			edf->color = V3(material.emissive.x, material.emissive.y, material.emissive.z);
		}
		else { *pdf = 1.0f; edf->color = V3(0.0f); }
	}
	// sample_impl : :309-355
	void sample(const float* Z, u32* prim_id, float* u, float* v, VertexGeometry* geom, float* pdf, Edf* edf) const
	{
		const float one = bits2f(0x3F7FFFFFu);
		if (n_vpls)
		{
			const u32 l = minu(f2u(Z[2] * float(n_vpls)), n_vpls - 1);
			*prim_id = vpls[l].prim_id; *u = vpls[l].u; *v = vpls[l].v;
			map(*prim_id, *u, *v, geom, pdf, edf);
		}
		else if (n_prims)
		{
			const u32 tri_id = upper_bound_index(minf(Z[2], one), prims_cdf, n_prims);
			*prim_id = tri_id; *u = Z[0]; *v = Z[1];
			if (*u + *v > 1.0f) { *u = 1.0f - *u; *v = 1.0f - *v; }
			map(*prim_id, *u, *v, geom, pdf, edf);
		}
		else { *prim_id = 0; *u = 0; *v = 0; *pdf = 1.0f; edf->color = V3(0.0f); }
	}
};

// src/camera.h:46-52
struct Camera { V3 eye, aim, up, dx; float fov; };

// src/camera.h:141-171 (host-side: tanf is libm on both oracle and product host code)
inline void camera_frame(const Camera& c, float aspect, V3& U, V3& V, V3& W)
{
	W = V3(c.aim.x - c.eye.x, c.aim.y - c.eye.y, c.aim.z - c.eye.z);
	const float wlen = sqrtf(dot(W, W));
	U = normalize(cross(W, c.up));
	V = normalize(cross(U, W));
	const float ulen = wlen * tanf(c.fov / 2.0f);
	U = V3(U.x * ulen, U.y * ulen, U.z * ulen);
	const float vlen = ulen / aspect;
	V = V3(V.x * vlen, V.y * vlen, V.z * vlen);
}
// src/camera.h:120-128
inline float square_pixel_focal_length(const Camera& c, u32 res_x, u32 res_y)
{
	const float t = tanf(c.fov / 2);
	return (float(res_x * res_y) / 4.0f) / (t * t);
}
// src/camera.h:231-252 (projected == false)
inline float camera_direction_pdf(V3 U, V3 V, V3 W, float W_len, float sq_focal, V3 out)
{
	const float t = dot(out, W) / (W_len * W_len);
	if (t < 0.0f) return 0.0f;
	const V3 I = out / t - W;
	const float Ix = dot(I, U) / dot(U, U);
	const float Iy = dot(I, V) / dot(V, V);
	if (Ix >= -1.0f && Ix <= 1.0f && Iy >= -1.0f && Iy <= 1.0f)
	{
		const float cos_theta = dot(out, W) / W_len;
		return sq_focal / (cos_theta * cos_theta * cos_theta);
	}
	return 0.0f;
}

// rays / hits : src/ray.h:42-76
struct Ray { float ox, oy, oz; u32 mask_or_tmin; float dx, dy, dz, tmax; };   // MaskedRay; closest-hit trace reads .mask as tmin
struct Hit { float t; i32 triId; float u, v; };
static_assert(sizeof(Ray) == 32 && sizeof(Hit) == 16, "ray/hit layout");

// the scene as shade_vertex sees it (subset of RenderingContextView, src/renderer_view.h:80-131)
struct SceneView
{
	Camera camera;
	u32 dir_lights_count; const DirectionalLight* dir_lights;
	Mesh mesh;
	MeshLight mesh_light;      // triangle-CDF instantiation  (n_vpls = 0)
	MeshLight mesh_vpls;       // VPL instantiation
	const Texture* textures;
	const float* glossy_reflectance;
	u32 res_x, res_y; float aspect, exposure, gamma;
};

// src/bpt_utils.h:585-642 — the parts the PT consumes (MIS bookkeeping for BPT omitted: weights are zero, :797)
struct EyeVertex
{
	VertexGeometry geom;
	V3 in;
	Material material;
	Bsdf bsdf;
	float prev_G_prime;

	void setup(const Ray& ray, const Hit& hit, const SceneView& r)
	{
		setup_differential_geometry(r.mesh, u32(hit.triId), hit.u, hit.v, &geom);
		geom.position = V3(ray.ox, ray.oy, ray.oz) + hit.t * V3(ray.dx, ray.dy, ray.dz);
		material = r.mesh.materials[r.mesh.material_indices[hit.triId]];
		const V4 one(1, 1, 1, 1);
		material.diffuse       = material.diffuse       * bilinear_texture_lookup(geom.texture_coords, material.diffuse_map, r.textures, one);
		material.specular      = material.specular      * bilinear_texture_lookup(geom.texture_coords, material.specular_map, r.textures, one);
		material.emissive      = material.emissive      * bilinear_texture_lookup(geom.texture_coords, material.emissive_map, r.textures, one);
		material.diffuse_trans = material.diffuse_trans * bilinear_texture_lookup(geom.texture_coords, material.diffuse_trans_map, r.textures, one);
		in = -normalize(V3(ray.dx, ray.dy, ray.dz));
		bsdf.setup(material, r.glossy_reflectance);
		prev_G_prime = fabsf(dot(in, geom.normal_s)) / (hit.t * hit.t);
	}
};

} // namespace orc
