// tools/bvh_walk.cpp -- a CPU model of the traversal ORDER of fpt_trace.hip over the 8-wide compressed BVH (fpt_bvh.h BvhNode8), used to
// evaluate builder changes without a GPU: it counts node steps and triangle tests per ray exactly as the kernel would take them (octant
// order through clz on the hit bits, one stack entry per node group, culling against the best hit so far) and models SIMD coherence by
// running 64 consecutive rays in lock step ("wave iterations": a wave pays for an iteration while any of its lanes is busy).
// Beside the kernel's own order (bvh8_walk) it holds the what-ifs DESIGN.md 5 quotes, run by `tools/bvh_stats.py --what-if`: bvh8_walk_policy (when a
// stacked node group is taken), bvh8_walk_cull (entry distances kept with stacked groups / children), bvh8_walk_set_exact (fp32 child boxes),
// bvh8_walk_sorted (strictly nearest-first), bvh8_walk_pairs (two rays per lane).
// Not part of the product; built by tools/bvh_stats.py with g++ -O2 -fopenmp into tools/_build/.
#include <stdint.h>
#include <string.h>
#include <math.h>
#include <algorithm>
#include <vector>

namespace {
int g_policy = 1;      // bvh8_walk_policy: 1 = the kernel's order (a node group on top of the stack is taken while triangles are still in hand); 0 = round 2's (pop only with nothing in hand); 2 = also pop parked triangles while only nodes are in hand
struct Ray { float o[3]; uint32_t mask; float d[3]; float tmax; };

inline float as_f32(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
inline float rcp_guard(float d) { const float a = fabsf(d); const float g = (a < 1.0e-20f) ? (d < 0.0f ? -1.0e-20f : 1.0e-20f) : d; return 1.0f / g; }

struct Lane
{
	bool have = false;
	float o[3], d[3], idir[3], tmin, tmax, best_t; int32_t best_id;
	uint32_t oct_inv; bool neg[3];
	uint32_t gx, gy; uint32_t tri_base, tri_bits;
	uint32_t stack[64][2]; int sp;
	// what-if (bvh8_walk_cull): entry distances kept with the groups -- g_t / tri_t: the nearest entry of the node group / triangle group in hand, st[]: of the
	// stacked ones, ct[]: per child slot of the group in hand, sct[][]: of the stacked groups
	float g_t = 0.0f, tri_t = 0.0f, st[64], ct[8], sct[64][8];
	uint64_t n_nodes = 0, n_tris = 0, n_loose = 0;
	bool any, occluded; uint32_t ray_mask;
};

// what-if (bvh8_walk_set_exact): per node 8 x {lo xyz, hi xyz} fp32 child boxes that replace the quantised ones -- prices what the 8-bit grid costs
const float* g_exact = nullptr; const uint32_t* g_nodes_base = nullptr;
int g_cull = 0;          // bvh8_walk_cull: 0 = the kernel (no distances on the stack); 1 = one distance per group; 2 = one per child; 3 = one per group, and the
                         // leaves of a node that start behind its nearest inner child are parked until that child's subtree is done
float g_last_tn[8];      // entry distances of the last test_node call, per child slot (3e38: not hit)
#pragma omp threadprivate(g_last_tn)
uint32_t test_node(const uint32_t* w, const Lane& L)
{
	const uint8_t* b = reinterpret_cast<const uint8_t*>(w);
	float A[3], B[3];
	for (int k = 0; k < 3; ++k)
	{
		A[k] = as_f32(uint32_t(b[12 + k]) << 23) * L.idir[k];
		B[k] = (as_f32(w[k]) - L.o[k]) * L.idir[k];
	}
	uint32_t hits = 0;
	for (int s = 0; s < 8; ++s)
	{
		g_last_tn[s] = 3.0e38f;
		const uint32_t m = b[24 + s];
		if (!m) continue;
		float tn = L.tmin, tf = L.best_t;
		for (int k = 0; k < 3; ++k)
		{
			float lo = fmaf(float(b[32 + 8 * k + s]), A[k], B[k]), hi = fmaf(float(b[56 + 8 * k + s]), A[k], B[k]);
			if (g_exact)
			{
				const float* e = g_exact + (size_t(w - g_nodes_base) / 20) * 48 + size_t(s) * 6;
				lo = (e[k] - L.o[k]) * L.idir[k]; hi = (e[3 + k] - L.o[k]) * L.idir[k];
			}
			tn = std::max(tn, L.neg[k] ? hi : lo); tf = std::min(tf, L.neg[k] ? lo : hi);
		}
		if (!(tn <= tf)) continue;
		g_last_tn[s] = tn;
		const bool inner = (m >> 5) == 1 && (m & 0x1F) >= 24;
		if (inner) hits |= 1u << (24 + (uint32_t(s) ^ L.oct_inv));
		else hits |= (m >> 5) << (m & 0x1F);
	}
	return hits;
}

// the three parts of one loop iteration of the kernel for one lane: the node step, the triangle test, the stack
int step_node(Lane& L, const uint32_t* nodes)
{
	int did = 0;
	if (g_cull == 1 && (L.gy & 0xFF000000u) && L.g_t > L.best_t) L.gy &= 0x00FFFFFFu;          // nothing left in the group can beat the hit in hand
	if (g_cull == 2)
		while (L.gy & 0xFF000000u)
		{
			const uint32_t bit = 31u - uint32_t(__builtin_clz(L.gy));
			if (L.ct[(bit - 24u) ^ L.oct_inv] > L.best_t) L.gy &= ~(1u << bit); else break;
		}
	if (L.gy & 0xFF000000u)
	{
		const uint32_t bit = 31u - uint32_t(__builtin_clz(L.gy));
		const uint32_t rest = L.gy & ~(1u << bit);
		if (rest & 0xFF000000u)
		{
			L.stack[L.sp][0] = L.gx; L.stack[L.sp][1] = rest; L.st[L.sp] = L.g_t; memcpy(L.sct[L.sp], L.ct, sizeof(L.ct)); L.sp++;
		}
		const uint32_t slot = (bit - 24u) ^ L.oct_inv;
		const uint32_t rel = uint32_t(__builtin_popcount(L.gy & ~(0xFFFFFFFFu << slot) & 0xFFu));
		const uint32_t* w = nodes + 20 * size_t(L.gx + rel);
		L.n_nodes++; did |= 1;
		const uint32_t hits = test_node(w, L);
		L.gx = w[4]; L.gy = (hits & 0xFF000000u) | (w[3] >> 24);
		float ti = 3.0e38f, tl = 3.0e38f;
		{
			const uint8_t* b = reinterpret_cast<const uint8_t*>(w);
			for (int s = 0; s < 8; ++s)
			{
				L.ct[s] = g_last_tn[s];
				const uint32_t m = b[24 + s];
				if (!m || g_last_tn[s] > 1.0e38f) continue;
				if ((m >> 5) == 1 && (m & 0x1F) >= 24) ti = std::min(ti, g_last_tn[s]); else tl = std::min(tl, g_last_tn[s]);
			}
		}
		L.g_t = ti;
		uint32_t tri_hits = hits & 0x00FFFFFFu;
		if (g_cull == 3 && tri_hits && (hits & 0xFF000000u))
		{
			// leaves that start behind the nearest inner child wait on the stack (with their distance) until that child's subtree is done
			const uint8_t* b = reinterpret_cast<const uint8_t*>(w);
			uint32_t far_bits = 0; float tfar = 3.0e38f; tl = 3.0e38f;
			for (int s = 0; s < 8; ++s)
			{
				const uint32_t m = b[24 + s];
				if (!m || g_last_tn[s] > 1.0e38f || ((m >> 5) == 1 && (m & 0x1F) >= 24)) continue;
				const uint32_t bits = (m >> 5) << (m & 0x1F);
				if (g_last_tn[s] > ti) { far_bits |= bits; tfar = std::min(tfar, g_last_tn[s]); } else tl = std::min(tl, g_last_tn[s]);
			}
			if (far_bits) { L.stack[L.sp][0] = w[5]; L.stack[L.sp][1] = far_bits; L.st[L.sp] = tfar; L.sp++; tri_hits &= ~far_bits; }
		}
		if (tri_hits)
		{
			if (L.tri_bits) { L.stack[L.sp][0] = L.tri_base; L.stack[L.sp][1] = L.tri_bits; L.st[L.sp] = L.tri_t; L.sp++; }
			L.tri_base = w[5]; L.tri_bits = tri_hits; L.tri_t = tl;
		}
	}
	return did;
}
int step_tri(Lane& L, const float* recs)
{
	int did = 0;
	if (g_cull && L.tri_bits && L.tri_t > L.best_t) L.tri_bits = 0;
	if (L.tri_bits)
	{
		const uint32_t k = uint32_t(__builtin_ctz(L.tri_bits));
		L.tri_bits &= L.tri_bits - 1u;
		const float* t = recs + 12 * size_t(L.tri_base + k);
		uint32_t tmask; memcpy(&tmask, t + 10, 4);
		const bool skip = L.any && (L.ray_mask & tmask);
		if (!skip) { L.n_tris++; }
		did |= 2;
		{
			// would the triangle's own fp32 box (what an uncompressed leaf would store) have culled this test?
			float tn = L.tmin, tf = L.best_t;
			for (int a = 0; a < 3; ++a)
			{
				const float p0 = t[a], p1 = t[a] + t[3 + a], p2 = t[a] + t[6 + a];
				const float lo = std::min(p0, std::min(p1, p2)) - 1e-5f, hi = std::max(p0, std::max(p1, p2)) + 1e-5f;
				const float t0 = (lo - L.o[a]) * L.idir[a], t1 = (hi - L.o[a]) * L.idir[a];
				tn = std::max(tn, std::min(t0, t1)); tf = std::min(tf, std::max(t0, t1));
			}
			if (!(tn <= tf)) L.n_loose++;
		}
		// Moeller-Trumbore (double: statistics only)
		const double v0[3] = { t[0], t[1], t[2] }, e1[3] = { t[3], t[4], t[5] }, e2[3] = { t[6], t[7], t[8] };
		const double d[3] = { L.d[0], L.d[1], L.d[2] };
		const double p[3] = { d[1] * e2[2] - d[2] * e2[1], d[2] * e2[0] - d[0] * e2[2], d[0] * e2[1] - d[1] * e2[0] };
		const double det = e1[0] * p[0] + e1[1] * p[1] + e1[2] * p[2];
		if (det != 0.0 && !skip)
		{
			const double inv = 1.0 / det;
			const double s[3] = { L.o[0] - v0[0], L.o[1] - v0[1], L.o[2] - v0[2] };
			const double bu = (s[0] * p[0] + s[1] * p[1] + s[2] * p[2]) * inv;
			const double q[3] = { s[1] * e1[2] - s[2] * e1[1], s[2] * e1[0] - s[0] * e1[2], s[0] * e1[1] - s[1] * e1[0] };
			const double bv = (d[0] * q[0] + d[1] * q[1] + d[2] * q[2]) * inv;
			const double tt = (e2[0] * q[0] + e2[1] * q[1] + e2[2] * q[2]) * inv;
			int32_t id; memcpy(&id, t + 9, 4);
			if (bu >= 0 && bu <= 1 && bv >= 0 && bu + bv <= 1 && tt > L.tmin && tt < L.tmax)
			{
				if (L.any) L.occluded = true;
				else if (L.best_id < 0 || float(tt) < L.best_t || (float(tt) == L.best_t && id < L.best_id)) { L.best_t = float(tt); L.best_id = id; }
			}
		}
	}
	return did;
}
void step_end(Lane& L)
{
	if (L.any && L.occluded) { L.have = false; return; }
	if (g_policy >= 1 && L.sp > 0 && !(L.gy & 0xFF000000u) && L.tri_bits && (L.stack[L.sp - 1][1] & 0xFF000000u))
	{
		L.sp--; L.gx = L.stack[L.sp][0]; L.gy = L.stack[L.sp][1]; L.g_t = L.st[L.sp]; memcpy(L.ct, L.sct[L.sp], sizeof(L.ct));
	}
	else if (g_policy >= 2 && L.sp > 0 && (L.gy & 0xFF000000u) && !L.tri_bits && !(L.stack[L.sp - 1][1] & 0xFF000000u))
	{
		L.sp--; L.tri_base = L.stack[L.sp][0]; L.tri_bits = L.stack[L.sp][1]; L.tri_t = L.st[L.sp];
	}
	if (!(L.gy & 0xFF000000u) && !L.tri_bits)
	{
		if (L.sp == 0) L.have = false;
		else
		{
			L.sp--;
			if (L.stack[L.sp][1] & 0xFF000000u) { L.gx = L.stack[L.sp][0]; L.gy = L.stack[L.sp][1]; L.g_t = L.st[L.sp]; memcpy(L.ct, L.sct[L.sp], sizeof(L.ct)); }
			else { L.tri_base = L.stack[L.sp][0]; L.tri_bits = L.stack[L.sp][1]; L.tri_t = L.st[L.sp]; }
		}
	}
}
// one loop iteration; returns what the lane did: bit 0 = node step, bit 1 = triangle test
int step(Lane& L, const uint32_t* nodes, const float* recs)
{
	int did = step_node(L, nodes);
	did |= step_tri(L, recs);
	step_end(L);
	return did;
}

void start(Lane& L, const Ray& r, bool any)
{
	L.have = true; L.any = any; L.occluded = false; L.ray_mask = r.mask;
	for (int k = 0; k < 3; ++k) { L.o[k] = r.o[k]; L.d[k] = r.d[k]; L.idir[k] = rcp_guard(r.d[k]); L.neg[k] = L.idir[k] < 0.0f; }
	L.oct_inv = 7u - ((L.neg[0] ? 4u : 0u) | (L.neg[1] ? 2u : 0u) | (L.neg[2] ? 1u : 0u));
	L.tmin = any ? 0.0f : as_f32(r.mask); L.tmax = r.tmax; L.best_t = r.tmax; L.best_id = -1;
	L.gx = 0; L.gy = 0x80000000u; L.sp = 0; L.tri_bits = 0; L.tri_base = 0; L.g_t = 0.0f; L.tri_t = 0.0f; for (int s = 0; s < 8; ++s) L.ct[s] = 0.0f;
	L.n_nodes = L.n_tris = L.n_loose = 0;
}
} // namespace

// out[0] node steps, out[1] triangle tests, out[2] wave iterations (64-ray groups in lock step, no refill), out[3] lane-iterations with work,
// out[4] max stack depth, out[5] wave iterations with refill modelled (a wave takes new rays when >= 32 lanes idle), out[6] / out[7] those of
// them in which some lane took a node step / tested a triangle (the wave pays ~228 / ~100 VALU instructions for them)
extern "C" void bvh8_walk_policy(int p) { g_policy = p; }
extern "C" void bvh8_walk_cull(int c) { g_cull = c; }
extern "C" void bvh8_walk_set_exact(const float* boxes, const uint32_t* nodes) { g_exact = boxes; g_nodes_base = nodes; }
// lane-level picture of the refill model: out[0] lane-iterations without a ray, [1] node step only, [2] triangle only, [3] both, [4] neither (a pop)
static uint64_t g_lane_stats[5];
extern "C" void bvh8_walk_lane_stats(uint64_t* out) { for (int i = 0; i < 5; ++i) out[i] = g_lane_stats[i]; }
extern "C" void bvh8_walk(const uint32_t* nodes, const float* recs, const Ray* rays, uint32_t n, int any_hit, uint64_t* out, int32_t* hit_ids, float* hit_t)
{
	uint64_t tn = 0, tt = 0, tw = 0, tl = 0, tdepth = 0, twr = 0, twn = 0, twt = 0, tloose = 0;
	const uint32_t n_waves = (n + 63) / 64;
	// (1) lock-step waves without refill
	#pragma omp parallel for schedule(dynamic, 16) reduction(+ : tn, tt, tw, tl, tloose) reduction(max : tdepth)
	for (uint32_t w = 0; w < n_waves; ++w)
	{
		Lane* lanes = new Lane[64];
		const uint32_t base = w * 64, cnt = std::min(64u, n - base);
		for (uint32_t l = 0; l < cnt; ++l) start(lanes[l], rays[base + l], any_hit != 0);
		for (;;)
		{
			int busy = 0;
			for (uint32_t l = 0; l < cnt; ++l)
				if (lanes[l].have) { busy++; step(lanes[l], nodes, recs); tdepth = std::max<uint64_t>(tdepth, uint64_t(lanes[l].sp)); }
			if (!busy) break;
			tw++; tl += uint64_t(busy);
		}
		for (uint32_t l = 0; l < cnt; ++l)
		{
			tn += lanes[l].n_nodes; tt += lanes[l].n_tris; tloose += lanes[l].n_loose;
			if (hit_ids) hit_ids[base + l] = any_hit ? (lanes[l].occluded ? 1 : -1) : lanes[l].best_id;
			if (hit_t) hit_t[base + l] = lanes[l].best_t;
		}
		delete[] lanes;
	}
	// (2) persistent waves with refill at >= 32 idle lanes: 256 model waves share the queue in contiguous chunks of 1024 rays
	{
		const uint32_t chunk = 1024; const uint32_t n_chunks = (n + chunk - 1) / chunk;
		uint64_t ls0 = 0, ls1 = 0, ls2 = 0, ls3 = 0, ls4 = 0;
		#pragma omp parallel for schedule(dynamic, 1) reduction(+ : twr, twn, twt, ls0, ls1, ls2, ls3, ls4)
		for (uint32_t c = 0; c < n_chunks; ++c)
		{
			Lane* lanes = new Lane[64];
			uint32_t next = c * chunk; const uint32_t end = std::min(n, next + chunk);
			for (;;)
			{
				int idle = 0; for (int l = 0; l < 64; ++l) idle += lanes[l].have ? 0 : 1;
				if (next < end && idle >= 32) for (int l = 0; l < 64 && next < end; ++l) if (!lanes[l].have) start(lanes[l], rays[next++], any_hit != 0);
				int busy = 0;
				int did = 0;
				uint64_t k[5] = { 0, 0, 0, 0, 0 };
				for (int l = 0; l < 64; ++l)
				{
					if (!lanes[l].have) { k[0]++; continue; }
					busy++; const int d = step(lanes[l], nodes, recs); did |= d;
					k[d == 1 ? 1 : d == 2 ? 2 : d == 3 ? 3 : 4]++;
				}
				if (!busy) break;
				ls0 += k[0]; ls1 += k[1]; ls2 += k[2]; ls3 += k[3]; ls4 += k[4];
				twr++; twn += (did & 1) ? 1 : 0; twt += (did & 2) ? 1 : 0;
			}
			delete[] lanes;
		}
		g_lane_stats[0] = ls0; g_lane_stats[1] = ls1; g_lane_stats[2] = ls2; g_lane_stats[3] = ls3; g_lane_stats[4] = ls4;
	}
	out[0] = tn; out[1] = tt; out[2] = tw; out[3] = tl; out[4] = tdepth; out[5] = twr; out[6] = twn; out[7] = twt; out[8] = tloose;
}

// What-if model (not the kernel): a wave owns a POOL of `pool` rays whose traversal state lives in LDS; the node half of an iteration takes up to 64 pool rays that
// have a node group to open, the triangle half up to 64 that have a triangle in hand (verdict r3 #3: "node tests and triangle tests each run on full waves").
// out[0] wave iterations with a node half, [1] with a triangle half, [2] node steps, [3] triangle tests, [4] / [5] lanes filled in the node / triangle halves;
// `refill_idle` = empty pool slots at which the wave takes new rays.
extern "C" void bvh8_walk_pool(const uint32_t* nodes, const float* recs, const Ray* rays, uint32_t n, int any_hit, int pool, int refill_idle, uint64_t* out)
{
	uint64_t twn = 0, twt = 0, tn = 0, tt = 0, fn = 0, ft = 0;
	const uint32_t chunk = 1024; const uint32_t n_chunks = (n + chunk - 1) / chunk;
	#pragma omp parallel for schedule(dynamic, 1) reduction(+ : twn, twt, tn, tt, fn, ft)
	for (uint32_t c = 0; c < n_chunks; ++c)
	{
		Lane* slot = new Lane[pool];
		uint32_t next = c * chunk; const uint32_t end = std::min(n, next + chunk);
		for (;;)
		{
			int idle = 0; for (int l = 0; l < pool; ++l) idle += slot[l].have ? 0 : 1;
			if (next < end && idle >= refill_idle) for (int l = 0; l < pool && next < end; ++l) if (!slot[l].have) start(slot[l], rays[next++], any_hit != 0);
			int busy = 0; for (int l = 0; l < pool; ++l) busy += slot[l].have ? 1 : 0;
			if (!busy) break;
			// one half per iteration: the one with more candidates (a ray that is not taken waits at no cost -- its state is in LDS, not in a lane)
			int cn = 0, ct = 0;
			for (int l = 0; l < pool; ++l) if (slot[l].have) { cn += (slot[l].gy & 0xFF000000u) ? 1 : 0; ct += slot[l].tri_bits ? 1 : 0; }
			if (cn >= ct && cn > 0)
			{
				int k = 0;
				for (int l = 0; l < pool && k < 64; ++l)
					if (slot[l].have && (slot[l].gy & 0xFF000000u)) { step_node(slot[l], nodes); step_end(slot[l]); ++k; }
				twn++; fn += uint64_t(k);
			}
			else if (ct > 0)
			{
				int m = 0;
				for (int l = 0; l < pool && m < 64; ++l)
					if (slot[l].have && slot[l].tri_bits) { step_tri(slot[l], recs); step_end(slot[l]); ++m; }
				twt++; ft += uint64_t(m);
			}
			else for (int l = 0; l < pool; ++l) if (slot[l].have) step_end(slot[l]);      // (cannot happen: a live ray always has something in hand after step_end)
		}
		for (int l = 0; l < pool; ++l) { tn += slot[l].n_nodes; tt += slot[l].n_tris; }
		delete[] slot;
	}
	out[0] = twn; out[1] = twt; out[2] = tn; out[3] = tt; out[4] = fn; out[5] = ft;
}

// The same with a ONE-BATCH PIPELINE: batch k+1 is chosen (and its node / triangle fetches issued) before batch k is processed, from the pool rays not in batch k --
// what a wave that hides its own memory latency has to do.  out as bvh8_walk_pool.
extern "C" void bvh8_walk_pool_pipelined(const uint32_t* nodes, const float* recs, const Ray* rays, uint32_t n, int any_hit, int pool, int refill_idle, uint64_t* out)
{
	uint64_t twn = 0, twt = 0, tn = 0, tt = 0, fn = 0, ft = 0;
	const uint32_t chunk = 1024; const uint32_t n_chunks = (n + chunk - 1) / chunk;
	#pragma omp parallel for schedule(dynamic, 1) reduction(+ : twn, twt, tn, tt, fn, ft)
	for (uint32_t c = 0; c < n_chunks; ++c)
	{
		Lane* slot = new Lane[pool];
		std::vector<uint8_t> busy(pool, 0);
		uint32_t next = c * chunk; const uint32_t end = std::min(n, next + chunk);
		std::vector<int> cur, nxt; int cur_type = 0, nxt_type = 0;      // 1 = node half, 2 = triangle half
		auto select = [&](std::vector<int>& b, int& type)
		{
			b.clear(); type = 0;
			int cn = 0, ct = 0;
			for (int l = 0; l < pool; ++l) if (slot[l].have && !busy[l]) { cn += (slot[l].gy & 0xFF000000u) ? 1 : 0; ct += slot[l].tri_bits ? 1 : 0; }
			if (cn >= ct && cn > 0) type = 1; else if (ct > 0) type = 2; else return;
			for (int l = 0; l < pool && b.size() < 64; ++l)
				if (slot[l].have && !busy[l] && (type == 1 ? (slot[l].gy & 0xFF000000u) != 0u : slot[l].tri_bits != 0u)) { b.push_back(l); busy[l] = 1; }
		};
		for (;;)
		{
			int idle = 0; for (int l = 0; l < pool; ++l) idle += slot[l].have ? 0 : 1;
			if (next < end && idle >= refill_idle) for (int l = 0; l < pool && next < end; ++l) if (!slot[l].have) start(slot[l], rays[next++], any_hit != 0);
			if (cur.empty()) select(cur, cur_type);
			if (cur.empty()) { int live = 0; for (int l = 0; l < pool; ++l) live += slot[l].have ? 1 : 0; if (!live && next >= end) break; if (!live) continue; break; }
			select(nxt, nxt_type);
			for (int l : cur) { if (cur_type == 1) step_node(slot[l], nodes); else step_tri(slot[l], recs); step_end(slot[l]); busy[l] = 0; }
			if (cur_type == 1) { twn++; fn += cur.size(); } else { twt++; ft += cur.size(); }
			cur.swap(nxt); cur_type = nxt_type; nxt.clear();
		}
		for (int l = 0; l < pool; ++l) { tn += slot[l].n_nodes; tt += slot[l].n_tris; }
		delete[] slot;
	}
	out[0] = twn; out[1] = twt; out[2] = tn; out[3] = tt; out[4] = fn; out[5] = ft;
}

// What-if model (not the kernel): every lane of a wave holds TWO rays; the node half of an iteration serves whichever of them has a node group to
// open, the triangle half whichever has a triangle in hand.  out[0] wave iterations, [1] of them with a node step, [2] with a triangle test,
// [3] node steps, [4] triangle tests; `refill_idle` = idle ray slots (of 128) at which the wave takes new rays.
extern "C" void bvh8_walk_pairs(const uint32_t* nodes, const float* recs, const Ray* rays, uint32_t n, int any_hit, int refill_idle, uint64_t* out)
{
	uint64_t tw = 0, twn = 0, twt = 0, tn = 0, tt = 0;
	const uint32_t chunk = 1024; const uint32_t n_chunks = (n + chunk - 1) / chunk;
	#pragma omp parallel for schedule(dynamic, 1) reduction(+ : tw, twn, twt, tn, tt)
	for (uint32_t c = 0; c < n_chunks; ++c)
	{
		Lane* slot = new Lane[128];          // lane l holds slot[2 l] and slot[2 l + 1]
		uint32_t next = c * chunk; const uint32_t end = std::min(n, next + chunk);
		for (;;)
		{
			int idle = 0; for (int l = 0; l < 128; ++l) idle += slot[l].have ? 0 : 1;
			if (next < end && idle >= refill_idle) for (int l = 0; l < 128 && next < end; ++l) if (!slot[l].have) start(slot[l], rays[next++], any_hit != 0);
			int busy = 0, did = 0;
			for (int l = 0; l < 64; ++l)
			{
				Lane& A = slot[2 * l]; Lane& B = slot[2 * l + 1];
				if (!A.have && !B.have) continue;
				busy++;
				Lane* X = (A.have && (A.gy & 0xFF000000u)) ? &A : (B.have && (B.gy & 0xFF000000u)) ? &B : nullptr;
				if (X) did |= step_node(*X, nodes);
				Lane* Y = (A.have && A.tri_bits) ? &A : (B.have && B.tri_bits) ? &B : nullptr;
				if (Y) did |= step_tri(*Y, recs);
				if (A.have) step_end(A);
				if (B.have) step_end(B);
			}
			if (!busy) break;
			tw++; twn += (did & 1) ? 1 : 0; twt += (did & 2) ? 1 : 0;
		}
		for (int l = 0; l < 128; ++l) { tn += slot[l].n_nodes; tt += slot[l].n_tris; }
		delete[] slot;
	}
	out[0] = tw; out[1] = twn; out[2] = twt; out[3] = tn; out[4] = tt;
}

// What-if (not the kernel's order): every ray visits the hit children of a node strictly nearest-first by the entry distance of their boxes -- inner
// children and leaves in one order, one stack entry per child.  out[0] node steps, out[1] triangle tests: how far the octant slot order is from a sort.
extern "C" void bvh8_walk_sorted(const uint32_t* nodes, const float* recs, const Ray* rays, uint32_t n, uint64_t* out)
{
	uint64_t tn = 0, tt = 0;
	#pragma omp parallel for schedule(dynamic, 256) reduction(+ : tn, tt)
	for (uint32_t r = 0; r < n; ++r)
	{
		Lane L; start(L, rays[r], false);
		struct E { float t; uint32_t kind, a, b; };          // kind 0: node index a; kind 1: triangles a .. a + b - 1
		E stack[256]; int sp = 0;
		stack[sp++] = E{ L.tmin, 0u, 0u, 0u };
		while (sp)
		{
			const E e = stack[--sp];
			if (e.t > L.best_t) continue;
			if (e.kind == 1)
			{
				for (uint32_t k = 0; k < e.b; ++k) { L.tri_base = e.a; L.tri_bits = 1u << k; step_tri(L, recs); }
				L.tri_bits = 0;
				continue;
			}
			const uint32_t* w = nodes + 20 * size_t(e.a);
			const uint8_t* b = reinterpret_cast<const uint8_t*>(w);
			L.n_nodes++;
			float A[3], B[3];
			for (int k = 0; k < 3; ++k) { A[k] = as_f32(uint32_t(b[12 + k]) << 23) * L.idir[k]; B[k] = (as_f32(w[k]) - L.o[k]) * L.idir[k]; }
			E found[8]; int nf = 0; uint32_t rel = 0;
			for (int s = 0; s < 8; ++s)
			{
				const uint32_t m = b[24 + s];
				if (!m) continue;
				const bool inner = (m >> 5) == 1 && (m & 0x1F) >= 24;
				const uint32_t my_rel = rel; if (inner) rel++;
				float t0 = L.tmin, t1 = L.best_t;
				for (int k = 0; k < 3; ++k)
				{
					const float lo = fmaf(float(b[32 + 8 * k + s]), A[k], B[k]), hi = fmaf(float(b[56 + 8 * k + s]), A[k], B[k]);
					t0 = std::max(t0, L.neg[k] ? hi : lo); t1 = std::min(t1, L.neg[k] ? lo : hi);
				}
				if (!(t0 <= t1)) continue;
				if (inner) found[nf++] = E{ t0, 0u, w[4] + my_rel, 0u };
				else { const uint32_t c = (m >> 5) == 1 ? 1u : (m >> 5) == 3 ? 2u : 3u; found[nf++] = E{ t0, 1u, w[5] + (m & 0x1Fu), c }; }
			}
			std::sort(found, found + nf, [](const E& x, const E& y) { return x.t > y.t; });      // farthest first: the nearest ends on top of the stack
			for (int i = 0; i < nf && sp < 256; ++i) stack[sp++] = found[i];
		}
		tn += L.n_nodes; tt += L.n_tris;
	}
	out[0] = tn; out[1] = tt;
}

// What-if (round 5, verdict r4 #4): EIGHT LANES PER RAY -- a ray owns a row of 8 lanes, one child of the node per lane (Embree's single-ray BVH8 traversal laid on a
// SIMD row); a wave carries 8 rays.  Children are ordered by entry distance across the row (the order is free here: a sorting network over 8 lanes), every hit
// child gets its own stack entry with its distance, stacked entries behind the hit in hand are dropped when popped; a leaf group's triangles (<= 3) are tested one
// per lane, and all the triangle groups that are NEXT on the stack and nearer than the nearest stacked node are taken together (up to 8 triangles per triangle half).
// A wave iteration runs a node half if any of its 8 rays opens a node and a triangle half if any tests triangles; a ray slot that retires is refilled at once.
// out[0] wave iterations, [1] node halves, [2] triangle halves, [3] node steps, [4] triangle tests, [5] ray-slots active in node halves, [6] in triangle halves
extern "C" void bvh8_walk_lanes8(const uint32_t* nodes, const float* recs, const Ray* rays, uint32_t n, uint64_t* out)
{
	uint64_t tw = 0, twn = 0, twt = 0, sn = 0, st = 0, un = 0, ut = 0;
	const uint32_t per_wave = 8, chunk = 256;          // a wave draws its rays from a contiguous chunk of the queue, as the kernel's waves do
	#pragma omp parallel for schedule(dynamic, 16) reduction(+ : tw, twn, twt, sn, st, un, ut)
	for (uint32_t c0 = 0; c0 < n; c0 += chunk)
	{
		const uint32_t c1 = c0 + chunk < n ? c0 + chunk : n;
		struct E { float t; uint32_t kind, a, b; };
		struct Slot { Lane L; E stack[128]; int sp; bool live; };
		std::vector<Slot> S(per_wave);
		uint32_t next = c0;
		auto refill = [&](Slot& s) { if (next < c1) { start(s.L, rays[next++], false); s.sp = 0; s.stack[s.sp++] = E{ s.L.tmin, 0u, 0u, 0u }; s.live = true; } else s.live = false; };
		for (Slot& s : S) refill(s);
		for (;;)
		{
			bool any_live = false, node_half = false, tri_half = false; uint32_t an = 0, at = 0;
			for (Slot& s : S)
			{
				if (!s.live) continue;
				// drop entries behind the hit in hand
				while (s.sp && s.stack[s.sp - 1].t > s.L.best_t) --s.sp;
				if (!s.sp) { sn += s.L.n_nodes; st += s.L.n_tris; refill(s); if (!s.live) continue; }
				any_live = true;
				const E e = s.stack[s.sp - 1];
				if (e.kind == 1)
				{
					// all the triangle groups on top of the stack, up to 8 triangles
					uint32_t lanes = 0;
					while (s.sp && s.stack[s.sp - 1].kind == 1 && lanes + s.stack[s.sp - 1].b <= 8 && s.stack[s.sp - 1].t <= s.L.best_t)
					{
						const E g = s.stack[--s.sp];
						for (uint32_t k = 0; k < g.b; ++k) { s.L.tri_base = g.a; s.L.tri_bits = 1u << k; step_tri(s.L, recs); }
						s.L.tri_bits = 0; lanes += g.b;
					}
					tri_half = true; ++at;
					continue;
				}
				--s.sp;
				const uint32_t* w = nodes + 20 * size_t(e.a);
				const uint8_t* b = reinterpret_cast<const uint8_t*>(w);
				s.L.n_nodes++;
				float A[3], B[3];
				for (int k = 0; k < 3; ++k) { A[k] = as_f32(uint32_t(b[12 + k]) << 23) * s.L.idir[k]; B[k] = (as_f32(w[k]) - s.L.o[k]) * s.L.idir[k]; }
				E found[8]; int nf = 0; uint32_t rel = 0;
				for (int q = 0; q < 8; ++q)
				{
					const uint32_t m = b[24 + q];
					if (!m) continue;
					const bool inner = (m >> 5) == 1 && (m & 0x1F) >= 24;
					const uint32_t my_rel = rel; if (inner) rel++;
					float t0 = s.L.tmin, t1 = s.L.best_t;
					for (int k = 0; k < 3; ++k)
					{
						const float lo = fmaf(float(b[32 + 8 * k + q]), A[k], B[k]), hi = fmaf(float(b[56 + 8 * k + q]), A[k], B[k]);
						t0 = std::max(t0, s.L.neg[k] ? hi : lo); t1 = std::min(t1, s.L.neg[k] ? lo : hi);
					}
					if (!(t0 <= t1)) continue;
					if (inner) found[nf++] = E{ t0, 0u, w[4] + my_rel, 0u };
					else { const uint32_t cnt = (m >> 5) == 1 ? 1u : (m >> 5) == 3 ? 2u : 3u; found[nf++] = E{ t0, 1u, w[5] + (m & 0x1Fu), cnt }; }
				}
				std::sort(found, found + nf, [](const E& x, const E& y) { return x.t > y.t; });
				for (int i = 0; i < nf && s.sp < 128; ++i) s.stack[s.sp++] = found[i];
				node_half = true; ++an;
			}
			if (!any_live) break;
			++tw; if (node_half) { ++twn; un += an; } if (tri_half) { ++twt; ut += at; }
		}
	}
	out[0] = tw; out[1] = twn; out[2] = twt; out[3] = sn; out[4] = st; out[5] = un; out[6] = ut;
}

// What-if (round 4): VOTE-scheduled halves.  The wave runs the node half of an iteration only when at least `tn` lanes have a node group to open (or no
// lane has a triangle in hand), and the triangle half only when at least `tt` lanes have a triangle in hand (or the node half does not run); a lane whose
// half is skipped keeps its work for a later iteration.  `tri_rounds` > 1 repeats the triangle half while at least `tt` lanes still have triangles.
// out[0] wave iterations, [1] node halves run, [2] triangle halves run, [3] node steps, [4] triangle tests, [5] lane-slots of node halves that did a node step,
// [6] lane-slots of triangle halves that tested a triangle
extern "C" void bvh8_walk_vote(const uint32_t* nodes, const float* recs, const Ray* rays, uint32_t n, int any_hit, int tn, int tt, int tri_rounds, int refill_idle, uint64_t* out)
{
	uint64_t tw = 0, twn = 0, twt = 0, sn = 0, st = 0, un = 0, ut = 0;
	const uint32_t chunk = 1024; const uint32_t n_chunks = (n + chunk - 1) / chunk;
	#pragma omp parallel for schedule(dynamic, 1) reduction(+ : tw, twn, twt, sn, st, un, ut)
	for (uint32_t c = 0; c < n_chunks; ++c)
	{
		Lane* lanes = new Lane[64];
		uint32_t next = c * chunk; const uint32_t end = std::min(n, next + chunk);
		for (;;)
		{
			int idle = 0; for (int l = 0; l < 64; ++l) idle += lanes[l].have ? 0 : 1;
			if (next < end && idle >= refill_idle) for (int l = 0; l < 64 && next < end; ++l) if (!lanes[l].have) start(lanes[l], rays[next++], any_hit != 0);
			int busy = 0, n_node = 0, n_tri = 0;
			for (int l = 0; l < 64; ++l) if (lanes[l].have) { busy++; n_node += (lanes[l].gy & 0xFF000000u) ? 1 : 0; n_tri += lanes[l].tri_bits ? 1 : 0; }
			if (!busy) break;
			bool do_node = n_node >= tn || n_tri == 0;
			bool do_tri = n_tri >= tt || !do_node;
			if (!do_node && n_tri == 0) do_node = true;
			tw++;
			if (do_node) { twn++; for (int l = 0; l < 64; ++l) if (lanes[l].have) un += step_node(lanes[l], nodes) ? 1 : 0; }
			if (do_tri)
			{
				for (int round = 0; round < tri_rounds; ++round)
				{
					int cnt = 0; for (int l = 0; l < 64; ++l) if (lanes[l].have && lanes[l].tri_bits && !(lanes[l].any && lanes[l].occluded)) cnt++;
					if (round > 0 && cnt < tt) break;
					if (cnt == 0) break;
					twt++;
					for (int l = 0; l < 64; ++l) if (lanes[l].have && !(lanes[l].any && lanes[l].occluded)) ut += step_tri(lanes[l], recs) ? 1 : 0;
				}
			}
			for (int l = 0; l < 64; ++l) if (lanes[l].have) { Lane& L = lanes[l]; const bool was = L.have; step_end(L); if (was && !L.have) { sn += L.n_nodes; st += L.n_tris; } }
		}
		delete[] lanes;
	}
	out[0] = tw; out[1] = twn; out[2] = twt; out[3] = sn; out[4] = st; out[5] = un; out[6] = ut;
}
