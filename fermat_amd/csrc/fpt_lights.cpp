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
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <map>
#include <system_error>
#include <thread>
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
		// step^offset.  Rows are words; bit i of a row selects row m-1-i of the right factor: product(A, B)[y] = XOR over the set bits i of A[y] of B[m-1-i].
		// (Until round 5 the power was taken by offset - 1 = 3631 multiplications, bit by bit: three seconds of every context's creation.  The product is
		// associative, so square-and-multiply gives the same matrix.)
		auto product = [&](const uint32_t* A, const uint32_t* B, uint32_t* C) {
			for (uint32_t y = 0; y < m; ++y) { uint32_t acc = 0; for (uint32_t i = 0, r = A[y]; r; ++i, r >>= 1) if (r & 1u) acc ^= B[m - i - 1]; C[y] = acc; } };
		uint32_t result[32], base[32], tmp[32];
		for (uint32_t i = 0; i < m; ++i) { result[i] = 1u << (m - i - 1); base[i] = step[i]; }          // identity: row y selects row y
		for (uint32_t e = offset; e; e >>= 1)
		{
			if (e & 1u) { product(result, base, tmp); for (uint32_t i = 0; i < m; ++i) result[i] = tmp[i]; }
			product(base, base, tmp); for (uint32_t i = 0; i < m; ++i) base[i] = tmp[i];
		}
		const uint32_t* cur = result;
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
	// the stream as it will be after `count` more draws: the transition is linear over GF(2), so the jump is one matrix power (square-and-multiply on the
	// columns) -- what lets the threads of build_emitter_tables each start in the middle of the sequence the serial loop would have drawn
	LfsrStream skipped(uint64_t count) const
	{
		auto apply = [](const uint32_t* M, uint32_t v) { uint32_t r = 0; for (uint32_t i = 0; v; ++i, v >>= 1) if (v & 1u) r ^= M[i]; return r; };
		uint32_t base[32], tmp[32], st = state;
		for (uint32_t i = 0; i < 32; ++i) base[i] = gen.col[i];
		for (uint64_t e = count; e; e >>= 1)
		{
			if (e & 1u) st = apply(base, st);
			for (uint32_t i = 0; i < 32; ++i) tmp[i] = apply(base, base[i]);
			for (uint32_t i = 0; i < 32; ++i) base[i] = tmp[i];
		}
		return LfsrStream{ gen, st, scramble };
	}
};

// the builder's threads (the same rule as the acceleration structure's: the GPU boxes show 256 hardware threads and grant ~16)
uint32_t table_threads()
{
	uint32_t n = std::thread::hardware_concurrency();
	n = n == 0 ? 1u : std::min(n, 16u);
	if (const char* e = std::getenv("FPT_BUILD_THREADS")) n = uint32_t(std::max(1, std::min(64, std::atoi(e))));
	return n;
}
// f(begin, end, slice) over contiguous slices of [0, n); exceptions are rethrown on the caller's thread; threads that cannot be created: their slices run here
// (`grain`: below this many elements the loop is not worth a thread; 1 for loops over a handful of heavy tasks)
template <class F> void slices(size_t n, uint32_t count, F f, size_t grain = 4096)
{
	if (count <= 1 || n < grain) { f(size_t(0), n, 0u); return; }
	std::vector<std::exception_ptr> error(count);
	auto run = [&](uint32_t t) { try { f(n * t / count, n * (t + 1) / count, t); } catch (...) { error[t] = std::current_exception(); } };
	std::vector<std::thread> pool;
	uint32_t started = 1;
	try { for (uint32_t t = 1; t < count; ++t) { pool.emplace_back(run, t); started = t + 1; } } catch (const std::system_error&) {}
	run(0u);
	for (uint32_t t = started; t < count; ++t) run(t);
	for (std::thread& t : pool) t.join();
	for (const std::exception_ptr& e : error) if (e) std::rethrow_exception(e);
}

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

// Round 5: on all threads, with the serial loop's results bit for bit.  What is sequential in the reference's algorithm is kept sequential -- the one random stream
// (threads jump to their place in it), every float accumulation (the emission total in double, `norm`, the VPL CDF: summed by one thread in index order over values
// the threads computed), the stable order of equal Morton codes -- and everything else (areas, CDF look-ups, surface points, texture fetches, codes, the sort's
// runs) is per element.  1.44 M VPLs over 1.82 M triangles: 0.5 s -> 0.06 s on the box's 16 threads; it is most of what update_scene costs after a refit.
// What the emitter tables of a mesh depend on besides materials, textures and texture coordinates: the positions of the triangles that emit (a non-zero emission colour
// or an emissive map).  A 64-bit fingerprint of exactly those -- per slice of the triangle array, combined in slice order -- lets update_scene tell a moved chair from
// a moved lamp: the tables (0.09 s of host work for 1.44 M VPLs) are rebuilt only for the lamp.
uint64_t emitter_fingerprint(const fpt_mesh_view& mesh, const fpt_texture* textures)
{
	const uint32_t nt = uint32_t(mesh.num_triangles), th = table_threads();
	std::vector<uint64_t> part(std::max(th, 1u), 0ull);
	std::vector<uint64_t> count(std::max(th, 1u), 0ull);
	slices(nt, th, [&](size_t tb, size_t te, uint32_t sl) {
		uint64_t h = 1469598103934665603ull, n = 0;
		for (size_t t = tb; t < te; ++t)
		{
			const fpt_material& mat = mesh.materials[mesh.material_indices[t]];
			const bool mapped = mat.emissive_map.texture != 0xFFFFFFFFu && textures && textures[mat.emissive_map.texture].texels;
			if (!mapped && mat.emissive[0] == 0.0f && mat.emissive[1] == 0.0f && mat.emissive[2] == 0.0f) continue;
			const int32_t* ix = mesh.vertex_indices + 4 * t;
			uint32_t w[13]; w[0] = uint32_t(t);
			for (int c = 0; c < 3; ++c) std::memcpy(&w[1 + 3 * c], mesh.vertex_data + 4 * size_t(ix[c]), 12);
			std::memcpy(&w[10], mat.emissive, 12);
			for (int k = 0; k < 13; ++k) { h ^= w[k]; h *= 1099511628211ull; }
			++n;
		}
		part[sl] = h; count[sl] = n; });
	uint64_t h = 1469598103934665603ull;
	for (size_t i = 0; i < part.size(); ++i) { h ^= part[i]; h *= 1099511628211ull; h ^= count[i]; h *= 1099511628211ull; }
	return h ^ (uint64_t(th) << 56);          // (the slicing depends on the thread count, which is fixed for a process)
}

void build_emitter_tables(uint32_t n_vpls, const fpt_mesh_view& mesh, const fpt_texture* textures, uint32_t instance, EmitterTables& out)
{
	const auto clock = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
	const double t_start = clock();
	const uint32_t nt = uint32_t(mesh.num_triangles);
	const uint32_t th = table_threads();
	out.mesh_cdf.assign(nt, 0.0f); out.mesh_inv_area.assign(nt, 0.0f);
	out.vpl_cdf.clear(); out.vpls.clear(); out.norm = 0.0f;
	static const Lfsr32 generator;
	LfsrStream random{ generator, 1u, hash32(1351u + instance) };

	// emission-weighted triangle CDF, accumulated in double (src/mesh_lights.cu:169-277)
	std::vector<float> weight(nt);          // emission_pdf_measure(emission) * area
	std::vector<uint32_t> mapped;           // triangles whose material has an emissive map with texels: they draw from the stream, in triangle order
	{
		std::vector<std::vector<uint32_t>> part(th);
		slices(nt, th, [&](size_t tb, size_t te, uint32_t sl) {
			for (size_t t = tb; t < te; ++t)
			{
				const int32_t* ix = mesh.vertex_indices + 4 * t;
				const f3 p0 = mesh_position(mesh, ix[0]), p1 = mesh_position(mesh, ix[1]), p2 = mesh_position(mesh, ix[2]);
				const float area = 0.5f * length(cross(p0 - p2, p1 - p2));
				const fpt_material& mat = mesh.materials[mesh.material_indices[t]];
				out.mesh_inv_area[t] = 1.0f / area;
				if (mat.emissive_map.texture != 0xFFFFFFFFu && textures && textures[mat.emissive_map.texture].texels) { part[sl].push_back(uint32_t(t)); weight[t] = area; }
				else weight[t] = emission_pdf_measure(load4(mat.emissive)) * area;
			} });
		for (const std::vector<uint32_t>& p : part) mapped.insert(mapped.end(), p.begin(), p.end());
	}
	std::map<uint32_t, MipPyramid> pyramids;
	for (const uint32_t t : mapped)
	{
		const int32_t* ix = mesh.vertex_indices + 4 * size_t(t);
		const float area = weight[t];
		const fpt_material& mat = mesh.materials[mesh.material_indices[t]];
		f4 emission = load4(mat.emissive);
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
				const float s = mod1(((s2 * w + s0 * u) + s1 * v) * sx), tt = mod1(((t2 * w + t0 * u) + t1 * v) * sy);
				const uint32_t x = sel_min(to_u32_sat(s * float(rx)), rx - 1), y = sel_min(to_u32_sat(tt * float(ry)), ry - 1);
				const float* px = &tex[(size_t(y) * rx + x) * 4];
				avg = avg + mk4(px[0], px[1], px[2], px[3]);
			}
			const float inv = float(n_samples);
			emission = emission * mk4(avg.x / inv, avg.y / inv, avg.z / inv, avg.w / inv);
		}
		weight[t] = emission_pdf_measure(emission) * area;
	}
	double total = 0.0;
	for (uint32_t t = 0; t < nt; ++t) { total += double(weight[t]); out.mesh_cdf[t] = float(total); }
	if (total == 0.0)
	{
		for (uint32_t t = 0; t < nt; ++t) out.mesh_cdf[t] = float(t + 1) / float(nt);
		return;                                           // no emitters: the VPL set stays empty, NEE is disabled
	}
	slices(nt, th, [&](size_t tb, size_t te, uint32_t) { for (size_t t = tb; t < te; ++t) out.mesh_cdf[t] = float(double(out.mesh_cdf[t]) / total); });
	if (out.mesh_cdf[nt - 1] != 1.0f)
	{
		const float last = out.mesh_cdf[nt - 1];
		for (int32_t t = int32_t(nt) - 1; t >= 0 && out.mesh_cdf[t] == last; --t) out.mesh_cdf[t] = 1.0f;
	}

	const double t_cdf = clock();
	// stratified draw of n_vpls surface points through the CDF (:301-340): three draws per point
	const float below_one = std::nexttoward(1.0f, 0.0L);
	std::vector<fpt_vpl> first_pass(n_vpls);
	slices(n_vpls, th, [&](size_t ib, size_t ie, uint32_t) {
		LfsrStream rnd = random.skipped(3ull * ib);
		for (size_t i = ib; i < ie; ++i)
		{
			const float r = (float(uint32_t(i)) + rnd.next()) / float(n_vpls);
			const uint32_t tri = sel_min(upper_bound(out.mesh_cdf.data(), nt, sel_min(r, below_one)), nt - 1);
			float u = rnd.next();
			float v = rnd.next();
			if (u + v > 1.0f) { u = 1.0f - u; v = 1.0f - v; }
			SurfacePoint sp; float pdf;
			surface_point(mesh, tri, u, v, sp, &pdf);
			pdf *= out.mesh_cdf[tri] - (tri ? out.mesh_cdf[tri - 1] : 0.0f);
			const fpt_material& mat = mesh.materials[mesh.material_indices[tri]];
			const f4 e = load4(mat.emissive) * sample_texture(textures, mat.emissive_map, sp.s, sp.t, mk4(1, 1, 1, 1));
			first_pass[i].prim_id = tri; first_pass[i].uv[0] = u; first_pass[i].uv[1] = v;
			first_pass[i].E = emission_pdf_measure(mk4(e.x / pdf, e.y / pdf, e.z / pdf, e.w / pdf));
		} });
	random.state = random.skipped(3ull * n_vpls).state;
	float norm = 0.0f;
	for (uint32_t i = 0; i < n_vpls; ++i) norm += first_pass[i].E;
	norm /= float(n_vpls);
	out.norm = norm;

	const double t_first = clock();
	// per-VPL CDF, then resample so the set is distributed exactly by emission (:346-377): one draw per point
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
	{
		std::vector<f3> plo(th, lo), phi(th, hi);
		slices(n_vpls, th, [&](size_t ib, size_t ie, uint32_t sl) {
			LfsrStream rnd = random.skipped(uint64_t(ib));
			f3 l = splat3(1.0e30f), h = splat3(-1.0e30f);
			for (size_t i = ib; i < ie; ++i)
			{
				const float r = (float(uint32_t(i)) + rnd.next()) / float(n_vpls);
				const uint32_t k = sel_min(upper_bound(out.vpl_cdf.data(), n_vpls, sel_min(r, below_one)), n_vpls - 1u);
				picked[i] = first_pass[k];
				where[i] = surface_position_only(mesh, picked[i].prim_id, picked[i].uv[0], picked[i].uv[1]);
				l = mk3(sel_min(l.x, where[i].x), sel_min(l.y, where[i].y), sel_min(l.z, where[i].z));
				h = mk3(sel_max(h.x, where[i].x), sel_max(h.y, where[i].y), sel_max(h.z, where[i].z));
			}
			plo[sl] = l; phi[sl] = h; });
		for (uint32_t t = 0; t < th; ++t)
		{
			lo = mk3(sel_min(lo.x, plo[t].x), sel_min(lo.y, plo[t].y), sel_min(lo.z, plo[t].z));
			hi = mk3(sel_max(hi.x, phi[t].x), sel_max(hi.y, phi[t].y), sel_max(hi.z, phi[t].z));
		}
	}
	const double t_second = clock();
	// spatial order: stable sort by 60-bit Morton code over the VPL bounding box (:391-424).  Big sets: a least-significant-digit radix sort, five passes of 12
	// bits, each slice counting and then scattering its own elements in index order -- stable by construction, i.e. the one order std::stable_sort gives, and
	// balanced whatever the codes look like (the VPLs of two small emitters share most of their bits).
	const f3 inv = mk3(1.0f / (hi.x - lo.x), 1.0f / (hi.y - lo.y), 1.0f / (hi.z - lo.z));
	typedef std::pair<uint64_t, uint32_t> Keyed;
	std::vector<Keyed> keyed(n_vpls), other;
	slices(n_vpls, th, [&](size_t ib, size_t ie, uint32_t) {
		for (size_t i = ib; i < ie; ++i)
		{
			const uint32_t x = quantize((where[i].x - lo.x) * inv.x, 1u << 20);
			const uint32_t y = quantize((where[i].y - lo.y) * inv.y, 1u << 20);
			const uint32_t z = quantize((where[i].z - lo.z) * inv.z, 1u << 20);
			keyed[i] = std::make_pair(morton60(x, y, z), uint32_t(i));
		} });
	std::vector<Keyed>* src = &keyed;
	const uint32_t parts = (th > 1 && n_vpls >= 65536u) ? th : 1u;
	if (parts == 1) std::stable_sort(keyed.begin(), keyed.end(), [](const Keyed& a, const Keyed& b) { return a.first < b.first; });
	else
	{
		other.resize(n_vpls);
		std::vector<Keyed>* dst = &other;
		const uint32_t kBits = 12, kDigits = 1u << kBits;
		std::vector<uint32_t> hist(size_t(parts) * kDigits);
		for (uint32_t shift = 0; shift < 60; shift += kBits)
		{
			slices(parts, parts, [&](size_t pb, size_t pe, uint32_t) {
				for (size_t t = pb; t < pe; ++t)
				{
					uint32_t* h = &hist[t * kDigits];
					for (uint32_t d = 0; d < kDigits; ++d) h[d] = 0;
					for (size_t i = size_t(n_vpls) * t / parts; i < size_t(n_vpls) * (t + 1) / parts; ++i) h[((*src)[i].first >> shift) & (kDigits - 1)]++;
				} }, 1);
			uint32_t running = 0; bool one_digit = false;
			for (uint32_t d = 0; d < kDigits; ++d)
			{
				const uint32_t before = running;
				for (uint32_t t = 0; t < parts; ++t) { const uint32_t c = hist[size_t(t) * kDigits + d]; hist[size_t(t) * kDigits + d] = running; running += c; }
				if (running - before == n_vpls) one_digit = true;          // summed over the parts: no single part ever holds all the elements
			}
			if (one_digit) continue;          // every code has this digit: the pass would copy the array
			slices(parts, parts, [&](size_t pb, size_t pe, uint32_t) {
				for (size_t t = pb; t < pe; ++t)
				{
					uint32_t* h = &hist[t * kDigits];
					for (size_t i = size_t(n_vpls) * t / parts; i < size_t(n_vpls) * (t + 1) / parts; ++i) { const Keyed& k = (*src)[i]; (*dst)[h[(k.first >> shift) & (kDigits - 1)]++] = k; }
				} }, 1);
			std::swap(src, dst);
		}
	}
	out.vpls.resize(n_vpls);
	slices(n_vpls, th, [&](size_t ib, size_t ie, uint32_t) { for (size_t i = ib; i < ie; ++i) out.vpls[i] = picked[(*src)[i].second]; });
	if (std::getenv("FPT_BVH_TIMERS"))
		std::fprintf(stderr, "build_emitter_tables: triangle CDF %.3f s, first draw %.3f, resampling %.3f, Morton order %.3f (%u threads)\n", t_cdf - t_start, t_first - t_cdf, t_second - t_first, clock() - t_second, th);
}

} // namespace fpt
