// fpt_shading.h — per-vertex set-up shared by the shading kernels (device) and the mesh-light builder (host).
//
//   surface point from (triangle, barycentrics)   src/mesh_utils.h:184-310, src/mesh/MeshCompression.h:52-68
//   LOD-0 bilinear texture fetch                   src/texture_view.h:107-118,170-202
//   emitter lookup (VPL set or triangle CDF)       src/lights.h:299-431, src/direct_lighting_mesh.h:41-111
#pragma once
#include "fpt_bsdf.h"
#include "../../include/fermat_pt_hip.h"

namespace fpt {

struct f4 { float x, y, z, w; };
FPT_HD f4 mk4(float x, float y, float z, float w) { f4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
FPT_HD f4 operator*(f4 a, f4 b) { return mk4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
FPT_HD f4 operator*(f4 a, float s) { return mk4(a.x * s, a.y * s, a.z * s, a.w * s); }
FPT_HD f4 operator+(f4 a, f4 b) { return mk4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
FPT_HD f4 load4(const float* p) { const float4 v = *reinterpret_cast<const float4*>(p); return mk4(v.x, v.y, v.z, v.w); }
FPT_HD void store4(float* p, f4 v) { *reinterpret_cast<float4*>(p) = make_float4(v.x, v.y, v.z, v.w); }
FPT_HD f3 xyz(f4 v) { return mk3(v.x, v.y, v.z); }

struct SurfacePoint
{
	ShadingFrame frame;
	f3 position;
	float s, t;          // interpolated texture coordinates
};

#ifndef FPT_VPL_POINT_STRIDE
#define FPT_VPL_POINT_STRIDE 3
#endif
static constexpr uint32_t VPL_POINT_STRIDE = FPT_VPL_POINT_STRIDE;       // float4 per record of EmitterView::vpl_points

// emitter tables (MeshLight, src/lights.h:299-307)
struct EmitterView
{
	uint32_t n_prims; const float* prims_cdf; const float* prims_inv_area;
	uint32_t n_vpls;  const fpt_vpl* vpls; float norm;
	// The VPLs' light points, tabulated (round 4): what emitter_sample computes for VPL l -- the surface point of its triangle at its barycentrics, the
	// emitted radiance there (material x emissive texture) and its pdf -- depends on the VPL alone, not on the vertex that draws it, yet every shaded vertex
	// recomputed it: six scattered fetches (VPL, index quad, three vertices, material) and ~500 instructions.  vpl_points_kernel runs the SAME device
	// functions once per VPL and stores three float4 per VPL: {position, pdf}, {shading normal, -}, {radiance, -}; a vertex then reads 48 contiguous bytes.
	// Same operations on the same values: every bit of the sample is unchanged.  NULL: emitter_sample computes it per vertex (mesh-CDF emitters, the BPT).
	// (Records padded to 64 B, so that none straddles two of the 128-byte lines memory is fetched in, were measured no better: shading 0.687-0.691 vs 0.685 ms per
	// step on the bathroom2 stand-in, 0.338 vs 0.329 on rounds 1-3's scene -- the table is a third larger and the pick is uniformly random over it.)
	const float4* vpl_points;
};

FPT_HD f3 mesh_position(const fpt_mesh_view& mesh, int32_t v) { const float* p = mesh.vertex_data + 4 * size_t(v); return mk3(p[0], p[1], p[2]); }

// geometry at (u, v) of a triangle given its three vertex records (position + packed normal) and its three packed texture coordinates (the int4 of
// texture_indices_comp; < 0 = the vertex has none): u weights vertex 0, v vertex 1 (src/kernels/optix_base_shaders.h:50-57)
FPT_HD void surface_point_of(const f4 a, const f4 b, const f4 c, bool has_texcoords, int32_t tc_x, int32_t tc_y, int32_t tc_z, const float tex_scale[2], const float tex_bias[2],
                             float u, float v, SurfacePoint& sp, float* area_pdf = nullptr)
{
	const f3 p0 = xyz(a), p1 = xyz(b), p2 = xyz(c);
	const float w = 1.0f - u - v;
	sp.position = p2 * w + p0 * u + p1 * v;
	const f3 du = p0 - p2, dv = p1 - p2;
	const f3 gx = cross(du, dv);
	sp.frame.ng = normalize(gx);
	if (area_pdf) *area_pdf = 2.0f / length(gx);
	const f3 n0 = unpack_normal(as_u32(a.w)), n1 = unpack_normal(as_u32(b.w)), n2 = unpack_normal(as_u32(c.w));
	const f3 N = normalize(n2 * w + n0 * u + n1 * v);
	sp.frame.n = N;
	sp.frame.t = orthogonal(N);
	sp.frame.b = cross(N, sp.frame.t);
	if (has_texcoords)
	{
		float s0 = 1.0f, t0 = 0.0f, s1 = 0.0f, t1 = 1.0f, s2 = 0.0f, t2 = 0.0f;
		if (tc_x >= 0) { s0 = half_bits_to_float(uint32_t(tc_x) & 0xffffu) * tex_scale[0] + tex_bias[0]; t0 = half_bits_to_float(uint32_t(tc_x) >> 16) * tex_scale[1] + tex_bias[1]; }
		if (tc_y >= 0) { s1 = half_bits_to_float(uint32_t(tc_y) & 0xffffu) * tex_scale[0] + tex_bias[0]; t1 = half_bits_to_float(uint32_t(tc_y) >> 16) * tex_scale[1] + tex_bias[1]; }
		if (tc_z >= 0) { s2 = half_bits_to_float(uint32_t(tc_z) & 0xffffu) * tex_scale[0] + tex_bias[0]; t2 = half_bits_to_float(uint32_t(tc_z) >> 16) * tex_scale[1] + tex_bias[1]; }
		sp.s = s2 * w + s0 * u + s1 * v;
		sp.t = t2 * w + t0 * u + t1 * v;
	}
	else { sp.s = u; sp.t = v; }
}
// ... from the mesh view's arrays (MeshView, src/mesh/MeshView.h): the index quad, three vertices, the texture-index quad -- five scattered fetches
FPT_HD void surface_point(const fpt_mesh_view& mesh, uint32_t tri, float u, float v, SurfacePoint& sp, float* area_pdf = nullptr)
{
	const int4 idx = *reinterpret_cast<const int4*>(mesh.vertex_indices + 4 * size_t(tri));
	const f4 a = load4(mesh.vertex_data + 4 * size_t(idx.x));
	const f4 b = load4(mesh.vertex_data + 4 * size_t(idx.y));
	const f4 c = load4(mesh.vertex_data + 4 * size_t(idx.z));
	int4 tc = make_int4(-1, -1, -1, -1);
	if (mesh.texture_indices_comp) tc = *reinterpret_cast<const int4*>(mesh.texture_indices_comp + 4 * size_t(tri));
	surface_point_of(a, b, c, mesh.texture_indices_comp != nullptr, tc.x, tc.y, tc.z, mesh.tex_scale, mesh.tex_bias, u, v, sp, area_pdf);
}
// ... and from the triangle's SHADING RECORD (round 4): the same fifteen words -- three vertex records, three packed texture coordinates, the material index --
// gathered once per triangle into 64 contiguous bytes (shade_records_kernel), so that a shaded vertex fetches one record instead of six scattered
// sectors.  On the bathroom2 stand-in the shading kernel moved 465 B per vertex beyond L2 (4.9 TB/s: at the memory system's limit, VALU 0.73 busy); the same
// values feed the same arithmetic, so every bit of the result is unchanged.
struct ShadeRecord { float4 a, b, c, d; };          // a, b, c: vertex_data of the three vertices; d: texture_indices_comp.xyz, material index
FPT_HD void surface_point(const ShadeRecord& r, const fpt_mesh_view& mesh, float u, float v, SurfacePoint& sp)
{
	surface_point_of(mk4(r.a.x, r.a.y, r.a.z, r.a.w), mk4(r.b.x, r.b.y, r.b.z, r.b.w), mk4(r.c.x, r.c.y, r.c.z, r.c.w), mesh.texture_indices_comp != nullptr,
	                 int32_t(as_u32(r.d.x)), int32_t(as_u32(r.d.y)), int32_t(as_u32(r.d.z)), mesh.tex_scale, mesh.tex_bias, u, v, sp);
}

FPT_HD f3 surface_position_only(const fpt_mesh_view& mesh, uint32_t tri, float u, float v)       // src/mesh_utils.h:322-337
{
	const int32_t* idx = mesh.vertex_indices + 4 * size_t(tri);
	const f3 p0 = mesh_position(mesh, idx[0]), p1 = mesh_position(mesh, idx[1]), p2 = mesh_position(mesh, idx[2]);
	return p2 * (1.0f - u - v) + p0 * u + p1 * v;
}

// bilinear, wrap-around, LOD 0; `fallback` when the reference is invalid or the texture has no levels
FPT_HD f4 sample_texture(const fpt_texture* textures, const fpt_texture_ref& ref, float s, float t, f4 fallback)
{
	if (ref.texture == 0xFFFFFFFFu) return fallback;
	const fpt_texture tex = textures[ref.texture];
	if (tex.texels == nullptr) return fallback;
	s = mod1(s * ref.scaling[0]);
	t = mod1(t * ref.scaling[1]);
	const float fx = s * float(tex.res_x), fy = t * float(tex.res_y);
	const uint32_t x = sel_min(to_u32_sat(fx), tex.res_x - 1);
	const uint32_t y = sel_min(to_u32_sat(fy), tex.res_y - 1);
	const uint32_t xx = (x + 1) % tex.res_x;
	const uint32_t yy = (y + 1) % tex.res_y;
	const f4 q0 = load4(tex.texels + 4 * (size_t(y) * tex.res_x + x));
	const f4 q1 = load4(tex.texels + 4 * (size_t(y) * tex.res_x + xx));
	const f4 q2 = load4(tex.texels + 4 * (size_t(yy) * tex.res_x + x));
	const f4 q3 = load4(tex.texels + 4 * (size_t(yy) * tex.res_x + xx));
	const float u = mod1(fx);
	const float v = mod1(fy);
	return (q0 * (1 - u) + q1 * u) * (1 - v) + (q2 * (1 - u) + q3 * u) * v;
}

FPT_HD float emission_pdf_measure(f4 E) { return sel_max(fabsf(E.x), sel_max(fabsf(E.y), fabsf(E.z))); }      // VPL::pdf, src/lights.h:75

// emitted radiance and area pdf at a known surface point of triangle `tri` (MeshLight::map_impl, src/lights.h:400-424)
FPT_HD void emitter_at(const EmitterView& em, const fpt_mesh_view& mesh, const fpt_texture* textures, uint32_t tri, float s, float t,
                       f3& radiance, float& pdf)
{
	if (em.n_vpls || em.n_prims)
	{
		const fpt_material* mat = mesh.materials + mesh.material_indices[tri];
		const f4 e = load4(mat->emissive) * sample_texture(textures, mat->emissive_map, s, t, mk4(1, 1, 1, 1));
		if (em.n_vpls) pdf = emission_pdf_measure(e) / em.norm;
		else           pdf = (em.prims_cdf[tri] - (tri ? em.prims_cdf[tri - 1] : 0)) * em.prims_inv_area[tri];
		radiance = xyz(e);
	}
	else { pdf = 1.0f; radiance = splat3(0.0f); }
}

FPT_HD uint32_t upper_bound(const float* a, uint32_t n, float x)           // contrib/cugar/basic/algorithms.h:138-199
{
	uint32_t lo = 0, count = n;
	while (count > 0)
	{
		const uint32_t step = count / 2;
		if (!(x < a[lo + step])) { lo += step + 1; count -= step + 1; }
		else count = step;
	}
	return lo;
}

// what light_sample needs of an emitter sample: the point, its shading normal, the radiance it emits and the pdf of having drawn it
struct LightPoint { f3 position, normal, radiance; float pdf; };

// draw an emitter point (MeshLight::sample_impl, src/lights.h:309-355)
FPT_HD void emitter_sample(const EmitterView& em, const fpt_mesh_view& mesh, const fpt_texture* textures, float z0, float z1, float z2,
                           SurfacePoint& lp, f3& radiance, float& pdf)
{
	uint32_t tri; float u, v;
	if (em.n_vpls)
	{
		const uint32_t l = sel_min(to_u32_sat(z2 * float(em.n_vpls)), em.n_vpls - 1);
		const fpt_vpl vp = em.vpls[l];
		tri = vp.prim_id; u = vp.uv[0]; v = vp.uv[1];
	}
	else if (em.n_prims)
	{
		tri = upper_bound(em.prims_cdf, em.n_prims, sel_min(z2, as_f32(0x3F7FFFFFu)));
		u = z0; v = z1;
		if (u + v > 1.0f) { u = 1.0f - u; v = 1.0f - v; }
	}
	else { pdf = 1.0f; radiance = splat3(0.0f); lp.position = splat3(0.0f); lp.frame.n = lp.frame.ng = mk3(0, 0, 1); lp.frame.t = mk3(1, 0, 0); lp.frame.b = mk3(0, 1, 0); lp.s = lp.t = 0; return; }
	surface_point(mesh, tri, u, v, lp);
	emitter_at(em, mesh, textures, tri, lp.s, lp.t, radiance, pdf);
}

// the same as a LightPoint; VPL emitters read the tabulated point (EmitterView::vpl_points) when there is one
FPT_HD LightPoint emitter_light_point(const EmitterView& em, const fpt_mesh_view& mesh, const fpt_texture* textures, float z0, float z1, float z2)
{
	LightPoint r;
	if (em.n_vpls && em.vpl_points)
	{
		const uint32_t l = sel_min(to_u32_sat(z2 * float(em.n_vpls)), em.n_vpls - 1);          // the index emitter_sample draws
		const float4* rec = em.vpl_points + VPL_POINT_STRIDE * size_t(l);
		const float4 a = rec[0], b = rec[1], c = rec[2];
		r.position = mk3(a.x, a.y, a.z); r.pdf = a.w; r.normal = mk3(b.x, b.y, b.z); r.radiance = mk3(c.x, c.y, c.z);
		return r;
	}
	SurfacePoint lp;
	emitter_sample(em, mesh, textures, z0, z1, z2, lp, r.radiance, r.pdf);
	r.position = lp.position; r.normal = lp.frame.n;
	return r;
}

} // namespace fpt
