// ORACLE — TEST INFRASTRUCTURE ONLY (see o_math.h header).
//
// o_lights.h : CPU restatement of the host-side mesh-light / VPL builder.
//   src/mesh_lights.cu:164-424 (MeshLightsStorageImpl::init) ; contrib/cugar/sampling/lfsr.h:66-281 ;
//   VPL re-ordering = stable sort by 60-bit Morton code (contrib/cugar/bvh/cuda/lbvh_builder_inline.h:76-116,
//   contrib/cugar/bits/morton.h:260-285) — the LBVH itself only feeds the RL sampler and is out of scope.
#pragma once
#include "o_scene.h"
#include <algorithm>
#include <map>

namespace orc {

// contrib/cugar/sampling/lfsr.h:60-281 (m = 32, GOOD_PROJECTIONS)
struct LFSRMatrix
{
	u32 m_m;
	u32 m_f[32];
	explicit LFSRMatrix(u32 m = 32, bool good_projections = true) : m_m(m)
	{
		static const u32 pp_table[30] = {
			(1 << 1) | 1, (1 << 1) | 1, (1 << 2) | 1, (1 << 1) | 1, (1 << 1) | 1, (1 << 4) | (1 << 3) | (1 << 2) | 1, (1 << 4) | 1, (1 << 3) | 1,
			(1 << 2) | 1, (1 << 6) | (1 << 4) | (1 << 1) | 1, (1 << 4) | (1 << 3) | (1 << 1) | 1, (1 << 5) | (1 << 3) | (1 << 1) | 1, (1 << 1) | 1,
			(1 << 5) | (1 << 3) | (1 << 2) | 1, (1 << 3) | 1, (1 << 7) | 1, (1 << 5) | (1 << 2) | (1 << 1) | 1, (1 << 3) | 1, (1 << 2) | 1, (1 << 1) | 1,
			(1 << 5) | 1, (1 << 4) | (1 << 3) | (1 << 1) | 1, (1 << 3) | 1, (1 << 6) | (1 << 2) | (1 << 1) | 1, (1 << 5) | (1 << 2) | (1 << 1) | 1,
			(1 << 3) | 1, (1 << 2) | 1, (1 << 6) | (1 << 4) | (1 << 1) | 1, (1 << 3) | 1, (1 << 7) | (1 << 6) | (1 << 2) | 1 };
		static const u32 offsets[30][2] = {
			{1,1},{2,2},{15,15},{8,8},{4,4},{41,41},{113,113},{115,226},{291,520},{172,1583},{267,2242},{332,2312},{388,38},{283,13981},
			{514,514},{698,698},{706,706},{1304,1304},{920,920},{1336,1336},{1236,1236},{1511,1511},{1445,1445},{1906,1906},{1875,1875},
			{2573,2573},{2633,2633},{2423,2423},{3573,3573},{3632,3632} };
		u32 matrix[32];
		matrix[m - 1] = 0;
		u32 pp = pp_table[m - 3];
		for (u32 i = 1; i < m; ++i, pp >>= 1)
		{
			matrix[m - 1] |= (pp & 1u) << (m - i);
			matrix[i - 1] = 1u << (m - i - 1);
		}
		u32 r0[32], r1[32];
		for (u32 i = 0; i < m; ++i) r0[i] = matrix[i];
		u32* in = r0; u32* out = r1;
		const u32 offset = offsets[m - 3][good_projections ? 1 : 0];
		for (u32 it = 1; it < offset; ++it)
		{
			for (u32 y = 0; y < m; ++y)
			{
				out[y] = 0;
				for (u32 x = 0; x < m; ++x)
					for (u32 i = 0; i < m; ++i)
						out[y] ^= (((in[y] >> i) & (matrix[m - i - 1] >> x)) & 1u) << x;
			}
			std::swap(in, out);
		}
		for (u32 y = 0; y < m; ++y)
		{
			m_f[y] = 0;
			for (u32 x = 0; x < m; ++x)
				m_f[y] |= ((in[x] >> y) & 1u) << (m - x - 1);
		}
	}
	// lfsr.h:256-275 (FLT_EPSILON from <float.h> is defined there: 1.1920929e-7)
	float next(u32 scramble, u32* state) const
	{
		u32 result = 0;
		u32 s = *state;
		for (u32 i = 0; s; ++i, s >>= 1)
			if (s & 1u) result ^= m_f[i];
		*state = result;
		result = (m_m == 32 ? result : (result << (32 - m_m))) ^ scramble;
		const float fr = float(result) * (1.f / float(u64(1) << 32));
		const float lim = 1.0f - 1.1920928955078125e-7f;
		return fr <= lim ? fr : lim;
	}
};
struct LFSRStream
{
	const LFSRMatrix* m; u32 state, scramble;
	LFSRStream(const LFSRMatrix* _m, u32 _state, u32 _scramble) : m(_m), state(_state ? _state : 0xFFFFFFFFu), scramble(_scramble) {}
	float next() { return m->next(scramble, &state); }
};

struct MeshLightsStorage
{
	std::vector<float> mesh_cdf, mesh_inv_area, vpl_cdf;
	std::vector<VPL> vpls;
	float normalization_coeff;

	// MipMapStorage::generate_mips / downsample (src/texture.h:222-258): level l+1 = 2x2 box filter of level l, res/2 (floor)
	struct Mips { std::vector<std::vector<V4>> levels; std::vector<u32> res_x, res_y; };
	static Mips generate_mips(const Texture& tex)
	{
		Mips m;
		m.levels.push_back(std::vector<V4>(size_t(tex.res_x) * tex.res_y));
		for (size_t p = 0; p < m.levels[0].size(); ++p) m.levels[0][p] = V4(tex.texels[4 * p], tex.texels[4 * p + 1], tex.texels[4 * p + 2], tex.texels[4 * p + 3]);
		m.res_x.push_back(tex.res_x); m.res_y.push_back(tex.res_y);
		u32 l_res_x = tex.res_x / 2, l_res_y = tex.res_y / 2;
		while (l_res_x >= 1 && l_res_y >= 1)
		{
			const std::vector<V4>& src = m.levels.back(); const u32 src_res_x = m.res_x.back();
			std::vector<V4> dst(size_t(l_res_x) * l_res_y);
			for (u32 y = 0; y < l_res_y; ++y)
				for (u32 x = 0; x < l_res_x; ++x)
				{
					V4 t(0, 0, 0, 0);
					for (u32 j = 0; j < 2; ++j)
						for (u32 i = 0; i < 2; ++i)
							t = t + src[size_t(y * 2 + j) * src_res_x + (x * 2 + i)];
					dst[size_t(y) * l_res_x + x] = V4(t.x / 4.0f, t.y / 4.0f, t.z / 4.0f, t.w / 4.0f);
				}
			m.levels.push_back(dst); m.res_x.push_back(l_res_x); m.res_y.push_back(l_res_y);
			l_res_x /= 2; l_res_y /= 2;
		}
		return m;
	}
	// cugar::log2(uint32) (contrib/cugar/basic/numbers.h:618-627)
	static u32 ilog2(u32 n)
	{
		u32 c = 0;
		if (n & 0xffff0000u) { n >>= 16; c |= 16; }
		if (n & 0xff00u) { n >>= 8; c |= 8; }
		if (n & 0xf0u) { n >>= 4; c |= 4; }
		if (n & 0xcu) { n >>= 2; c |= 2; }
		if (n & 0x2u) c |= 1;
		return c;
	}

	// src/mesh_lights.cu:164-424.  Emissive *textured* triangles (:186-243): 10 point samples of the mip level whose texel
	// matches the triangle's texture-space footprint; needs the raw per-vertex texture coordinates (Mesh::texture_data) — when
	// the caller does not provide them the untextured estimate is used and only the 20 LFSR draws are spent.
	void init(u32 n_vpls, const Mesh& mesh, const Texture* textures, u32 instance = 0)
	{
		const u32 nt = u32(mesh.num_triangles);
		mesh_cdf.assign(nt, 0.0f); mesh_inv_area.assign(nt, 0.0f);
		vpls.clear(); vpl_cdf.clear(); normalization_coeff = 0.0f;
		double sum = 0.0;
		std::map<u32, Mips> mips;
		LFSRMatrix generator(32, true);
		LFSRStream random(&generator, 1u, hash(1351u + instance));

		for (u32 i = 0; i < nt; ++i)
		{
			const i32* tri = mesh.vertex_indices + 4 * i;
			const V3 vp0 = load_vertex(mesh, tri[0]), vp1 = load_vertex(mesh, tri[1]), vp2 = load_vertex(mesh, tri[2]);
			const float area = 0.5f * length(cross(vp0 - vp2, vp1 - vp2));
			const Material material = mesh.materials[mesh.material_indices[i]];
			if (material.emissive_map.texture != 0xFFFFFFFFu && textures[material.emissive_map.texture].texels)
			{
				if (!mesh.texture_data)
				{
					for (u32 s = 0; s < 10; ++s) { random.next(); random.next(); }
					sum += double(vpl_pdf(material.emissive) * area);
				}
				else
				{
					// after unify_vertex_attributes the texture triangle is the vertex triangle
					const float* td = mesh.texture_data;
					const V2 vt0 = { td[2 * tri[0]], td[2 * tri[0] + 1] }, vt1 = { td[2 * tri[1]], td[2 * tri[1] + 1] }, vt2 = { td[2 * tri[2]], td[2 * tri[2] + 1] };
					const V2 dst_du = { vt0.x - vt2.x, vt0.y - vt2.y }, dst_dv = { vt1.x - vt2.x, vt1.y - vt2.y };
					const float n_samples = 10;
					if (mips.find(material.emissive_map.texture) == mips.end()) mips[material.emissive_map.texture] = generate_mips(textures[material.emissive_map.texture]);
					const Mips& mipmap = mips[material.emissive_map.texture];
					float max_edge = fmaxf(
						fmaxf(fabsf(dst_du.x), fabsf(dst_dv.x)) * material.emissive_map.sx * float(mipmap.res_x[0]),
						fmaxf(fabsf(dst_du.y), fabsf(dst_dv.y)) * material.emissive_map.sy * float(mipmap.res_y[0]));
					max_edge /= sqrtf(n_samples);
					const u32 lod = std::min(ilog2(f2u(max_edge)), u32(mipmap.levels.size()) - 1);
					const std::vector<V4>& texture = mipmap.levels[lod];
					const u32 res_x = mipmap.res_x[lod], res_y = mipmap.res_y[lod];
					V4 avg(0, 0, 0, 0);
					for (u32 s = 0; s < u32(n_samples); ++s)
					{
						float u = random.next();
						float v = random.next();
						if (u + v > 1.0f) { u = 1.0f - u; v = 1.0f - v; }
						const float st_x = mod1(((vt2.x * (1.0f - u - v) + vt0.x * u) + vt1.x * v) * material.emissive_map.sx);
						const float st_y = mod1(((vt2.y * (1.0f - u - v) + vt0.y * u) + vt1.y * v) * material.emissive_map.sy);
						const u32 x = std::min(f2u(st_x * float(res_x)), res_x - 1);
						const u32 y = std::min(f2u(st_y * float(res_y)), res_y - 1);
						avg = avg + texture[size_t(y) * res_x + x];
					}
					avg = V4(avg.x / n_samples, avg.y / n_samples, avg.z / n_samples, avg.w / n_samples);
					sum += double(vpl_pdf(material.emissive * avg) * area);
				}
			}
			else
				sum += double(vpl_pdf(material.emissive) * area);
			mesh_cdf[i] = float(sum);
			mesh_inv_area[i] = 1.0f / area;
		}
		if (sum)
		{
			for (u32 i = 0; i < nt; ++i) mesh_cdf[i] = float(double(mesh_cdf[i]) / double(sum));
			if (mesh_cdf[nt - 1] != 1.0f)
			{
				const float last = mesh_cdf[nt - 1];
				for (i32 i = i32(nt) - 1; i >= 0; --i) { if (mesh_cdf[i] == last) mesh_cdf[i] = 1.0f; else break; }
			}
		}
		else
		{
			for (u32 i = 0; i < nt; ++i) mesh_cdf[i] = float(i + 1) / float(nt);
			return;    // no emissive surfaces : n_vpls stays 0
		}

		std::vector<VPL> h_vpls(n_vpls);
		const float one = nexttowardf(1.0f, 0.0L);
		for (u32 i = 0; i < n_vpls; ++i)
		{
			const float r = (float(i) + random.next()) / float(n_vpls);
			const u32 tri_id = minu(upper_bound_index(minf(r, one), mesh_cdf.data(), nt), nt - 1);
			float u = random.next();
			float v = random.next();
			if (u + v > 1.0f) { u = 1.0f - u; v = 1.0f - v; }
			VertexGeometry geom; float pdf;
			setup_differential_geometry(mesh, tri_id, u, v, &geom, &pdf);
			pdf *= mesh_cdf[tri_id] - (tri_id ? mesh_cdf[tri_id - 1] : 0.0f);
			Material material = mesh.materials[mesh.material_indices[tri_id]];
			material.emissive = material.emissive * bilinear_texture_lookup(geom.texture_coords, material.emissive_map, textures, V4(1, 1, 1, 1));
			V4 E = material.emissive;
			E = V4(E.x / pdf, E.y / pdf, E.z / pdf, E.w / pdf);
			h_vpls[i].prim_id = tri_id; h_vpls[i].u = u; h_vpls[i].v = v; h_vpls[i].E = vpl_pdf(E);
			normalization_coeff += h_vpls[i].E;
		}
		normalization_coeff /= float(n_vpls);

		vpl_cdf.assign(n_vpls, 0.0f);
		{
			float s = 0.0f;
			for (u32 i = 0; i < n_vpls; ++i)
			{
				h_vpls[i].E /= normalization_coeff;
				s += h_vpls[i].E / float(n_vpls);
				vpl_cdf[i] = s;
			}
		}
		std::vector<VPL> resampled(n_vpls);
		std::vector<V3> pos(n_vpls);
		V3 bmin(1.0e30f), bmax(-1.0e30f);     // cugar::Bbox3f default (contrib/cugar/linalg/bbox.h field limits, numbers.h:1084-1085)
		for (u32 i = 0; i < n_vpls; ++i)
		{
			const float r = (float(i) + random.next()) / float(n_vpls);
			const u32 id = minu(upper_bound_index(minf(r, one), vpl_cdf.data(), n_vpls), n_vpls - 1u);
			resampled[i] = h_vpls[id];
			pos[i] = interpolate_position(mesh, resampled[i].prim_id, resampled[i].u, resampled[i].v);
			bmin = V3(minf(bmin.x, pos[i].x), minf(bmin.y, pos[i].y), minf(bmin.z, pos[i].z));
			bmax = V3(maxf(bmax.x, pos[i].x), maxf(bmax.y, pos[i].y), maxf(bmax.z, pos[i].z));
		}
		// Morton order (stable) : contrib/cugar/bits/morton.h:260-285 ; 0*inf = NaN quantises to 0 (numbers.h:600-603)
		const V3 inv(1.0f / (bmax.x - bmin.x), 1.0f / (bmax.y - bmin.y), 1.0f / (bmax.z - bmin.z));
		std::vector<std::pair<u64, u32> > keys(n_vpls);
		for (u32 i = 0; i < n_vpls; ++i)
		{
			const u32 x = quantize((pos[i].x - bmin.x) * inv.x, 1u << 20);
			const u32 y = quantize((pos[i].y - bmin.y) * inv.y, 1u << 20);
			const u32 z = quantize((pos[i].z - bmin.z) * inv.z, 1u << 20);
			keys[i] = std::make_pair(morton60(x, y, z), i);
		}
		std::stable_sort(keys.begin(), keys.end(), [](const std::pair<u64, u32>& a, const std::pair<u64, u32>& b) { return a.first < b.first; });
		vpls.resize(n_vpls);
		for (u32 i = 0; i < n_vpls; ++i) vpls[i] = resampled[keys[i].second];
	}
};

} // namespace orc
