// fpt_lights.cpp — host builder of the mesh-emitter sampling tables: the triangle CDF used by "-nee-alg mesh" and the
// emission-distributed VPL set used by the default "-nee-alg vpl" (MeshLightsStorageImpl::init, src/mesh_lights.cu:164-424).
//
// Randomness: a 32-bit Markov-chain QMC LFSR stream (contrib/cugar/sampling/lfsr.h:66-281, "good projections" offsets,
// state 1, scramble hash(1351 + instance)).  The VPL array is finally ordered by the 60-bit Morton code of the VPL position
// (the order the reference's LBVH builder sorts them into, contrib/cugar/bvh/cuda/lbvh_builder_inline.h:76-116); the LBVH
// itself only serves the out-of-scope RL sampler.
// Emissive *textured* triangles: their energy is estimated with 10 point samples of a box-filtered mip level chosen from the
// triangle's texture-space footprint (src/mesh_lights.cu:186-243; mip pyramid = 2x2 box filter, src/texture.h:222-258), read
// through the uncompressed per-vertex texture coordinates (fpt_mesh_view::texture_data).  Without texture_data the untextured
// emission is used, and the LFSR stream still advances by the 20 draws such a triangle costs, so every later draw lines up.
#include "fpt_host.h"
#include <algorithm>
#include <cmath>
#include <map>
#include <utility>

namespace fpt {
namespace {

// transition-matrix power of a primitive polynomial over GF(2), m = 32
struct Lfsr32
{
	uint32_t col[32];
	Lfsr32()
	{
		const uint32_t m = 32;
		const uint32_t poly = (1u << 7) | (1u << 6) | (1u << 2) | 1u;       // x^32 + x^7 + x^6 + x^2 + 1
		const uint32_t offset = 3632;                                          // m = 32 row of the offset table
		uint32_t step[32];
		step[m - 1] = 0;
		uint32_t pp = poly;
		for (uint32_t i = 1; i < m; ++i, pp >>= 1)
		{
			step[m - 1] |= (pp & 1u) << (m - i);
			step[i - 1] = 1u << (m - i - 1);
		}
		uint32_t a[32], b[32];
		for (uint32_t i = 0; i < m; ++i) a[i] = step[i];
		uint32_t* cur = a; uint32_t* nxt = b;
		for (uint32_t it = 1; it < offset; ++it)
		{
			for (uint32_t y = 0; y < m; ++y)
			{
				uint32_t acc = 0;
				for (uint32_t x = 0; x < m; ++x)
					for (uint32_t i = 0; i < m; ++i)
						acc ^= (((cur[y] >> i) & (step[m - i - 1] >> x)) & 1u) << x;
				nxt[y] = acc;
			}
			std::swap(cur, nxt);
		}
		for (uint32_t y = 0; y < m; ++y)
		{
			col[y] = 0;
			for (uint32_t x = 0; x < m; ++x) col[y] |= ((cur[x] >> y) & 1u) << (m - x - 1);
		}
	}
};
struct LfsrStream
{
	const Lfsr32& gen; uint32_t state, scramble;
	float next()
	{
		uint32_t r = 0;
		for (uint32_t i = 0, s = state; s; ++i, s >>= 1) if (s & 1u) r ^= gen.col[i];
		state = r;
		const float f = float(r ^ scramble) * (1.f / 4294967296.0f);
		const float cap = 1.0f - 1.1920928955078125e-7f;
		return f <= cap ? f : cap;
	}
};

// box-filtered mip pyramid of one float4 texture, built lazily for emissive maps only
struct MipPyramid
{
	std::vector<std::vector<float>> level; std::vector<uint32_t> rx, ry;
	void build(const fpt_texture& t)
	{
		uint32_t w = t.res_x, h = t.res_y;
		level.emplace_back(t.texels, t.texels + size_t(w) * h * 4); rx.push_back(w); ry.push_back(h);
		for (w /= 2, h /= 2; w >= 1 && h >= 1; w /= 2, h /= 2)
		{
			const std::vector<float>& src = level.back(); const uint32_t sw = rx.back();
			std::vector<float> dst(size_t(w) * h * 4);
			for (uint32_t y = 0; y < h; ++y) for (uint32_t x = 0; x < w; ++x) for (int c = 0; c < 4; ++c)
			{
				float acc = 0.0f;
				for (uint32_t j = 0; j < 2; ++j) for (uint32_t i = 0; i < 2; ++i) acc += src[(size_t(y * 2 + j) * sw + (x * 2 + i)) * 4 + c];
				dst[(size_t(y) * w + x) * 4 + c] = acc / 4.0f;
			}
			level.push_back(std::move(dst)); rx.push_back(w); ry.push_back(h);
		}
	}
};
uint32_t floor_log2(uint32_t n) { uint32_t c = 0; while (n > 1) { n >>= 1; ++c; } return c; }

} // namespace

void build_emitter_tables(uint32_t n_vpls, const fpt_mesh_view& mesh, const fpt_texture* textures, uint32_t instance, EmitterTables& out)
{
	const uint32_t nt = uint32_t(mesh.num_triangles);
	out.mesh_cdf.assign(nt, 0.0f); out.mesh_inv_area.assign(nt, 0.0f);
	out.vpl_cdf.clear(); out.vpls.clear(); out.norm = 0.0f;
	static const Lfsr32 generator;
	LfsrStream random{ generator, 1u, hash32(1351u + instance) };

	// emission-weighted triangle CDF, accumulated in double (src/mesh_lights.cu:169-277)
	double total = 0.0;
	std::map<uint32_t, MipPyramid> pyramids;
	for (uint32_t t = 0; t < nt; ++t)
	{
		const int32_t* ix = mesh.vertex_indices + 4 * size_t(t);
		const f3 p0 = mesh_position(mesh, ix[0]), p1 = mesh_position(mesh, ix[1]), p2 = mesh_position(mesh, ix[2]);
		const float area = 0.5f * length(cross(p0 - p2, p1 - p2));
		const fpt_material& mat = mesh.materials[mesh.material_indices[t]];
		f4 emission = load4(mat.emissive);
		if (mat.emissive_map.texture != 0xFFFFFFFFu && textures && textures[mat.emissive_map.texture].texels)
		{
			const uint32_t n_samples = 10;
			if (!mesh.texture_data) { for (uint32_t k = 0; k < 2 * n_samples; ++k) random.next(); }
			else
			{
				MipPyramid& mip = pyramids[mat.emissive_map.texture];
				if (mip.level.empty()) mip.build(textures[mat.emissive_map.texture]);
				const float* td = mesh.texture_data;
				const float s0 = td[2 * size_t(ix[0])], t0 = td[2 * size_t(ix[0]) + 1], s1 = td[2 * size_t(ix[1])], t1 = td[2 * size_t(ix[1]) + 1];
				const float s2 = td[2 * size_t(ix[2])], t2 = td[2 * size_t(ix[2]) + 1];
				const float sx = mat.emissive_map.scaling[0], sy = mat.emissive_map.scaling[1];
				// footprint of the triangle in texels of level 0, per sample
				float edge = sel_max(sel_max(fabsf(s0 - s2), fabsf(s1 - s2)) * sx * float(mip.rx[0]), sel_max(fabsf(t0 - t2), fabsf(t1 - t2)) * sy * float(mip.ry[0]));
				edge /= sqrtf(float(n_samples));
				const uint32_t lod = sel_min(floor_log2(to_u32_sat(edge)), uint32_t(mip.level.size()) - 1u);
				const std::vector<float>& tex = mip.level[lod]; const uint32_t rx = mip.rx[lod], ry = mip.ry[lod];
				f4 avg = mk4(0, 0, 0, 0);
				for (uint32_t k = 0; k < n_samples; ++k)
				{
					float u = random.next(), v = random.next();
					if (u + v > 1.0f) { u = 1.0f - u; v = 1.0f - v; }
					const float w = 1.0f - u - v;
					const float s = mod1(((s2 * w + s0 * u) + s1 * v) * sx), t = mod1(((t2 * w + t0 * u) + t1 * v) * sy);
					const uint32_t x = sel_min(to_u32_sat(s * float(rx)), rx - 1), y = sel_min(to_u32_sat(t * float(ry)), ry - 1);
					const float* px = &tex[(size_t(y) * rx + x) * 4];
					avg = avg + mk4(px[0], px[1], px[2], px[3]);
				}
				const float inv = float(n_samples);
				emission = emission * mk4(avg.x / inv, avg.y / inv, avg.z / inv, avg.w / inv);
			}
		}
		total += double(emission_pdf_measure(emission) * area);
		out.mesh_cdf[t] = float(total);
		out.mesh_inv_area[t] = 1.0f / area;
	}
	if (total == 0.0)
	{
		for (uint32_t t = 0; t < nt; ++t) out.mesh_cdf[t] = float(t + 1) / float(nt);
		return;                                           // no emitters: the VPL set stays empty, NEE is disabled
	}
	for (uint32_t t = 0; t < nt; ++t) out.mesh_cdf[t] = float(double(out.mesh_cdf[t]) / total);
	if (out.mesh_cdf[nt - 1] != 1.0f)
	{
		const float last = out.mesh_cdf[nt - 1];
		for (int32_t t = int32_t(nt) - 1; t >= 0 && out.mesh_cdf[t] == last; --t) out.mesh_cdf[t] = 1.0f;
	}

	// stratified draw of n_vpls surface points through the CDF (:301-340)
	const float below_one = std::nexttoward(1.0f, 0.0L);
	std::vector<fpt_vpl> first_pass(n_vpls);
	float norm = 0.0f;
	for (uint32_t i = 0; i < n_vpls; ++i)
	{
		const float r = (float(i) + random.next()) / float(n_vpls);
		const uint32_t tri = sel_min(upper_bound(out.mesh_cdf.data(), nt, sel_min(r, below_one)), nt - 1);
		float u = random.next();
		float v = random.next();
		if (u + v > 1.0f) { u = 1.0f - u; v = 1.0f - v; }
		SurfacePoint sp; float pdf;
		surface_point(mesh, tri, u, v, sp, &pdf);
		pdf *= out.mesh_cdf[tri] - (tri ? out.mesh_cdf[tri - 1] : 0.0f);
		const fpt_material& mat = mesh.materials[mesh.material_indices[tri]];
		const f4 e = load4(mat.emissive) * sample_texture(textures, mat.emissive_map, sp.s, sp.t, mk4(1, 1, 1, 1));
		first_pass[i].prim_id = tri; first_pass[i].uv[0] = u; first_pass[i].uv[1] = v;
		first_pass[i].E = emission_pdf_measure(mk4(e.x / pdf, e.y / pdf, e.z / pdf, e.w / pdf));
		norm += first_pass[i].E;
	}
	norm /= float(n_vpls);
	out.norm = norm;

	// per-VPL CDF, then resample so the set is distributed exactly by emission (:346-377)
	out.vpl_cdf.resize(n_vpls);
	{
		float acc = 0.0f;
		for (uint32_t i = 0; i < n_vpls; ++i)
		{
			first_pass[i].E /= norm;
			acc += first_pass[i].E / float(n_vpls);
			out.vpl_cdf[i] = acc;
		}
	}
	std::vector<fpt_vpl> picked(n_vpls);
	std::vector<f3> where(n_vpls);
	f3 lo = splat3(1.0e30f), hi = splat3(-1.0e30f);
	for (uint32_t i = 0; i < n_vpls; ++i)
	{
		const float r = (float(i) + random.next()) / float(n_vpls);
		const uint32_t k = sel_min(upper_bound(out.vpl_cdf.data(), n_vpls, sel_min(r, below_one)), n_vpls - 1u);
		picked[i] = first_pass[k];
		where[i] = surface_position_only(mesh, picked[i].prim_id, picked[i].uv[0], picked[i].uv[1]);
		lo = mk3(sel_min(lo.x, where[i].x), sel_min(lo.y, where[i].y), sel_min(lo.z, where[i].z));
		hi = mk3(sel_max(hi.x, where[i].x), sel_max(hi.y, where[i].y), sel_max(hi.z, where[i].z));
	}
	// spatial order: stable sort by 60-bit Morton code over the VPL bounding box (:391-424)
	const f3 inv = mk3(1.0f / (hi.x - lo.x), 1.0f / (hi.y - lo.y), 1.0f / (hi.z - lo.z));
	std::vector<std::pair<uint64_t, uint32_t> > keyed(n_vpls);
	for (uint32_t i = 0; i < n_vpls; ++i)
	{
		const uint32_t x = quantize((where[i].x - lo.x) * inv.x, 1u << 20);
		const uint32_t y = quantize((where[i].y - lo.y) * inv.y, 1u << 20);
		const uint32_t z = quantize((where[i].z - lo.z) * inv.z, 1u << 20);
		keyed[i] = std::make_pair(morton60(x, y, z), i);
	}
	std::stable_sort(keyed.begin(), keyed.end(), [](const std::pair<uint64_t, uint32_t>& a, const std::pair<uint64_t, uint32_t>& b) { return a.first < b.first; });
	out.vpls.resize(n_vpls);
	for (uint32_t i = 0; i < n_vpls; ++i) out.vpls[i] = picked[keyed[i].second];
}

} // namespace fpt
