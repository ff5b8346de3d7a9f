// ORACLE — TEST INFRASTRUCTURE ONLY (see o_math.h header).
//
// o_bvh.h : the "CUGAR host BVH" CPU ray caster = BASELINE.json's CPU baseline and the hit oracle.
//   builder   : contrib/cugar/bvh/bvh_sah_builder.h:43-164, bvh_sah_builder_inline.h:35-484:
//               three axis-sorted index arrays, full-sweep SAH, max_leaf_size 4, force_splitting.
//   traversal : CUGAR has none for 3-D rays; an ordered stack walk is supplied here.
//   hit rules : src/rt.cpp:558-635, src/kernels/optix_rt.cu:45-82,133-164, optix_base_shaders.h:42-91,
//               optix_base_shadow_shaders.h:42-72, optix_payload.h:75-78.
// OptiX's own intersector is closed-source: PARITY UNPINNED at this boundary.  The build's specification
// ("fpt-MT") is: Moeller-Trumbore in the operation order below, open interval tmin < t < tmax,
// edge-inclusive barycentric test, closest hit = minimum t, ties -> lowest triangle id; any-hit returns a boolean.
// Results are therefore independent of BVH topology and traversal order (boxes are conservatively padded).
#pragma once
#include <cstdlib>
#include "o_scene.h"
#include <algorithm>
#include <vector>

namespace orc {

struct Aabb { V3 lo, hi; };
inline Aabb aabb_empty() { Aabb b; b.lo = V3(1.0e30f); b.hi = V3(-1.0e30f); return b; }
inline void aabb_grow(Aabb& b, const Aabb& o)
{
	b.lo = V3(minf(b.lo.x, o.lo.x), minf(b.lo.y, o.lo.y), minf(b.lo.z, o.lo.z));
	b.hi = V3(maxf(b.hi.x, o.hi.x), maxf(b.hi.y, o.hi.y), maxf(b.hi.z, o.hi.z));
}
// bvh_sah_builder_inline.h:260-264 (half surface area)
inline float aabb_area(const Aabb& b)
{
	const float ex = b.hi.x - b.lo.x, ey = b.hi.y - b.lo.y, ez = b.hi.z - b.lo.z;
	return ex * ey + ez * (ex + ey);
}

// node : children are an adjacent pair (k, k+1); leaf = [begin, end) into `index`
struct BvhNode { Aabb box; u32 child; u32 begin, end; bool leaf; };

struct HostBvh
{
	std::vector<BvhNode> nodes;
	std::vector<u32> index;          // leaf primitive order (builder.index(i), bvh_sah_builder.h:105)

	void build(const std::vector<Aabb>& boxes, u32 max_leaf_size = 4)
	{
		const u32 n = u32(boxes.size());
		nodes.clear(); index.clear();
		if (n == 0) return;
		// pre-sort : bvh_sah_builder_inline.h:35-55,70-83 — by (min+max)[d], ties by entity index
		std::vector<u32> order[3];
		for (int d = 0; d < 3; ++d)
		{
			order[d].resize(n);
			for (u32 i = 0; i < n; ++i) order[d][i] = i;
			std::sort(order[d].begin(), order[d].end(), [&](u32 a, u32 b) {
				const float ca = (d == 0 ? boxes[a].lo.x + boxes[a].hi.x : d == 1 ? boxes[a].lo.y + boxes[a].hi.y : boxes[a].lo.z + boxes[a].hi.z);
				const float cb = (d == 0 ? boxes[b].lo.x + boxes[b].hi.x : d == 1 ? boxes[b].lo.y + boxes[b].hi.y : boxes[b].lo.z + boxes[b].hi.z);
				return ca < cb || (ca == cb && a < b); });
		}
		std::vector<Aabb> right(n);
		std::vector<uint8_t> tag(n);
		std::vector<u32> tmp(n);
		struct Task { u32 node, begin, end; };
		std::vector<Task> stack;
		BvhNode root; root.box = aabb_empty(); for (u32 i = 0; i < n; ++i) aabb_grow(root.box, boxes[i]);
		root.begin = 0; root.end = n; root.leaf = true; root.child = 0;
		nodes.push_back(root);
		stack.push_back(Task{ 0, 0, n });
		while (!stack.empty())
		{
			const Task t = stack.back(); stack.pop_back();
			const u32 cnt = t.end - t.begin;
			if (cnt <= max_leaf_size) { nodes[t.node].leaf = true; nodes[t.node].begin = t.begin; nodes[t.node].end = t.end; continue; }
			// find_best_split : bvh_sah_builder_inline.h:373-483 (strict <, first axis/position wins ties)
			float best = 3.0e38f; int best_axis = -1; u32 best_pos = 0;
			for (int d = 0; d < 3; ++d)
			{
				const u32* ord = order[d].data() + t.begin;
				Aabb acc = aabb_empty();
				for (u32 i = cnt - 1; i > 0; --i) { aabb_grow(acc, boxes[ord[i]]); right[i] = acc; }
				Aabb left = aabb_empty();
				for (u32 i = 1; i < cnt; ++i)
				{
					aabb_grow(left, boxes[ord[i - 1]]);
					const float cost = aabb_area(left) * float(i) + aabb_area(right[i]) * float(cnt - i);
					if (cost < best) { best = cost; best_axis = d; best_pos = i; }
				}
			}
			if (best_axis < 0) { best_axis = 0; best_pos = cnt / 2; }     // force_splitting fallback: median (:141-156)
			// partition : tag by winning axis, stable re-partition of the other two (:454-481)
			const u32* ord = order[best_axis].data() + t.begin;
			for (u32 i = 0; i < cnt; ++i) tag[ord[i]] = (i < best_pos) ? 0 : 1;
			for (int d = 0; d < 3; ++d)
			{
				if (d == best_axis) continue;
				u32* o = order[d].data() + t.begin;
				u32 l = 0, r = 0;
				for (u32 i = 0; i < cnt; ++i) { if (tag[o[i]] == 0) o[l++] = o[i]; else tmp[r++] = o[i]; }
				for (u32 i = 0; i < r; ++i) o[l + i] = tmp[i];
			}
			const u32 k = u32(nodes.size());
			BvhNode a, b;
			a.box = aabb_empty(); b.box = aabb_empty();
			for (u32 i = 0; i < best_pos; ++i) aabb_grow(a.box, boxes[ord[i]]);
			for (u32 i = best_pos; i < cnt; ++i) aabb_grow(b.box, boxes[ord[i]]);
			a.leaf = b.leaf = true; a.child = b.child = 0;
			a.begin = t.begin; a.end = t.begin + best_pos; b.begin = a.end; b.end = t.end;
			nodes.push_back(a); nodes.push_back(b);
			nodes[t.node].leaf = false; nodes[t.node].child = k;
			stack.push_back(Task{ k + 1, b.begin, b.end });   // right pushed first, left processed first (:110-131)
			stack.push_back(Task{ k, a.begin, a.end });
		}
		index = order[0];
	}
};

// ---- fpt-MT : the intersector specification shared with the HIP kernels -------------------------------------------
// Round 5, the BOX clause: a hit counts only if the point the ray reaches at the computed t, y = (o - v0) + t d, lies in the triangle's own bounding box
// [min(0, e1, e2), max(0, e1, e2)] widened per component by tol = vpad + 4e-7 (|y| + |t d|), vpad = 5e-7 (|triangle|max + |scene|max) (1e-6 in round 5).  For a ray that grazes the
// triangle's plane (det -> 0) the computed t is noise -- off by more than the boxes are padded -- and it can fall inside (tmin, tmax) when the true crossing does not
// (found on the water_caustic stand-in: one connection ray in 10^8, ending ON the surface it grazes, whose "hit" at t = 0.99988 of tmax = 0.9999 one tree reached and two
// others culled).  WHETHER such a triangle is tested at all then depends on the acceleration structure.  With the clause an accepted hit's point is inside the
// triangle's padded box (2e-6 (...) = 4 vpad) with margin, so every conservative traversal reaches it for the parameter t: the answer is a function of the ray and the triangles
// alone, whatever the tree and the order.  A true hit's computed point is off the triangle's box by rounding only (measured: <= 0.25 vpad over 1e6 rays of the bench
// scenes; the 4e-7 term keeps that true for origins far outside the scene), so the clause rejects next to no true hit: with round 6's t and HALF of round 5's constant tolerance, 0 of 4.31 M / 4.08 M / 3.29 M / 3.25 M closest-hit rays of real 1600x900 passes on four scenes, and still 0 of 4.31 M on the bench scene with the constant halved again (round 5's t at round 5's tolerance lost one there; profiles/r06_clause_rate.txt).  (A first form of the clause compared the ray's point
// with the point the barycentrics name: on sliver triangles bu and bv carry errors of 1e-4 of an edge, and it rejected 1-2 % of TRUE hits on the bench scene.  Bit-exact
// parity with the kernel cannot see that -- both sides did it; tools/diag_clause_rate.py is the check that does.)
// the clause can be switched off for ONE purpose: measuring, on the rays of real passes, that it changes nothing there (tests/test_oracle.py, tools/diag_clause_rate.py)
inline bool& box_clause_enabled() { static bool on = true; return on; }
struct TriHit { float t, bu, bv; };   // bu, bv = weights of vertex 1 and 2
inline bool intersect_tri(V3 o, V3 d, V3 v0, V3 v1, V3 v2, float tmin, float tmax, float vpad, TriHit* h)
{
	const V3 e1 = v1 - v0;
	const V3 e2 = v2 - v0;
	// round 6: two cross products instead of three, and the determinant and t from ONE normal.  With n = e1 x e2 and c = s x d (s = o - v0):
	//   det = e1 . (d x e2) = -(d . n)      bu = s . (d x e2) / det = (e2 . c) / det      bv = d . (s x e1) / det = -(e1 . c) / det      t = e2 . (s x e1) / det = (s . n) / det
	// t is then the exact crossing of the ray with A plane through v0 whose normal is n as computed: the error of n (large on sliver triangles, where e1 x e2 cancels) tilts
	// that plane by ~1e-5 rad about v0 and moves the crossing by 1e-5 of the TRIANGLE's size.  Until round 5 numerator and denominator were two separately rounded triple
	// products, and on a sliver 42 units from the eye t came out wrong by 1.5e-4 -- the one true hit in 17 M the box clause below rejected (profiles/r05_clause_rate.txt).
	const V3 n = cross(e1, e2);
	const float det = 0.0f - dot(d, n);
	if (det == 0.0f) return false;
	const float inv = 1.0f / det;
	const V3 s = o - v0;
	const V3 c = cross(s, d);
	const float bu = dot(e2, c) * inv;
	if (!(bu >= 0.0f && bu <= 1.0f)) return false;
	const float bv = (0.0f - dot(e1, c)) * inv;
	if (!(bv >= 0.0f && bu + bv <= 1.0f)) return false;
	const float t = dot(s, n) * inv;
	if (!(t > tmin && t < tmax)) return false;
	const V3 td = t * d;
	const V3 y = s + td;
	const float lo[3] = { minf(minf(0.0f, e1.x), e2.x), minf(minf(0.0f, e1.y), e2.y), minf(minf(0.0f, e1.z), e2.z) };
	const float hi[3] = { maxf(maxf(0.0f, e1.x), e2.x), maxf(maxf(0.0f, e1.y), e2.y), maxf(maxf(0.0f, e1.z), e2.z) };
	const float yy[3] = { y.x, y.y, y.z }, tt[3] = { td.x, td.y, td.z };
	static const float vscale = std::getenv("ORC_VPAD_SCALE") ? float(std::atof(std::getenv("ORC_VPAD_SCALE"))) : 1.0f;      // tools/diag_clause_rate.py: how much margin the tolerance has
	for (int k = 0; k < 3 && box_clause_enabled(); ++k)
	{
		const float tol = vpad * vscale + 4.0e-7f * (fabsf(yy[k]) + fabsf(tt[k]));
		if (!(yy[k] >= lo[k] - tol && yy[k] <= hi[k] + tol)) return false;
	}
	h->t = t; h->bu = bu; h->bv = bv;
	return true;
}

struct RayCaster
{
	HostBvh bvh;
	const Mesh* mesh;
	u64 nodes_visited, tris_tested;
	std::vector<float> vpad;          // per triangle: the constant part of the tolerance of fpt-MT's box clause, 5e-7 (|triangle|max + |scene|max)

	void build(const Mesh& m)
	{
		mesh = &m;
		float scene_mag = 0.0f;
		for (i32 v = 0; v < m.num_vertices; ++v) for (int k = 0; k < 3; ++k) scene_mag = maxf(scene_mag, fabsf(m.vertex_data[4 * size_t(v) + k]));
		vpad.assign(size_t(m.num_triangles), 0.0f);
		std::vector<Aabb> boxes(m.num_triangles);
		for (i32 i = 0; i < m.num_triangles; ++i)
		{
			const i32* tri = m.vertex_indices + 4 * i;
			Aabb b = aabb_empty();
			for (int k = 0; k < 3; ++k) { const V3 p = load_vertex(m, tri[k]); Aabb pb; pb.lo = p; pb.hi = p; aabb_grow(b, pb); }
			// conservative padding so that rounding in the slab test can never cull a triangle the fpt-MT test accepts
			const float m0 = maxf(maxf(fabsf(b.lo.x), fabsf(b.hi.x)), maxf(maxf(fabsf(b.lo.y), fabsf(b.hi.y)), maxf(fabsf(b.lo.z), fabsf(b.hi.z))));
			const float pad = (m0 + scene_mag) * 2.0e-6f + 1.0e-30f;          // four times the box clause's constant tolerance: an accepted hit point is inside with margin
			vpad[size_t(i)] = (m0 + scene_mag) * 5.0e-7f;
			b.lo = b.lo - V3(pad); b.hi = b.hi + V3(pad);
			boxes[i] = b;
		}
		bvh.build(boxes, 4);
		nodes_visited = tris_tested = 0;
	}
	static bool slab(const Aabb& b, V3 o, V3 id, float tmin, float tmax, float* tn)
	{
		float t0 = (b.lo.x - o.x) * id.x, t1 = (b.hi.x - o.x) * id.x;
		float lo = minf(t0, t1), hi = maxf(t0, t1);
		t0 = (b.lo.y - o.y) * id.y; t1 = (b.hi.y - o.y) * id.y;
		lo = maxf(lo, minf(t0, t1)); hi = minf(hi, maxf(t0, t1));
		t0 = (b.lo.z - o.z) * id.z; t1 = (b.hi.z - o.z) * id.z;
		lo = maxf(lo, minf(t0, t1)); hi = minf(hi, maxf(t0, t1));
		// widen by 2 ulp-ish factor: conservative
		lo = lo - fabsf(lo) * 1.0e-6f; hi = hi + fabsf(hi) * 1.0e-6f;
		*tn = lo;
		return !(lo > hi) && !(hi < tmin) && !(lo > tmax);    // NaN-tolerant: NaN comparisons are false -> visit
	}
	// closest hit : RTContext::trace, ray type 2 (no masking), src/kernels/optix_rt.cu:45-82
	Hit trace(const Ray& r) { return trace(r, nodes_visited, tris_tested); }
	Hit trace(const Ray& r, u64& nv, u64& tt) const
	{
		const V3 o(r.ox, r.oy, r.oz), d(r.dx, r.dy, r.dz);
		const float tmin = bits2f(r.mask_or_tmin);
		float best_t = r.tmax; i32 best_id = -1; float best_bu = 0, best_bv = 0;
		Hit h; h.t = -1.0f; h.triId = -1; h.u = 0.0f; h.v = 0.0f;
		if (bvh.nodes.empty()) return h;
		const V3 id(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
		u32 stack[128]; int sp = 0; stack[sp++] = 0;
		while (sp)
		{
			const BvhNode& n = bvh.nodes[stack[--sp]];
			float tn; nv++;
			if (!slab(n.box, o, id, tmin, best_t, &tn)) continue;
			if (n.leaf)
			{
				for (u32 i = n.begin; i < n.end; ++i)
				{
					const u32 tri_id = bvh.index[i];
					const i32* tri = mesh->vertex_indices + 4 * tri_id;
					TriHit th; tt++;
					// upper bound is inclusive of the current best so that equal-t ties can be resolved by id
					if (intersect_tri(o, d, load_vertex(*mesh, tri[0]), load_vertex(*mesh, tri[1]), load_vertex(*mesh, tri[2]), tmin, r.tmax, vpad[tri_id], &th))
					{
						if (best_id < 0 || th.t < best_t || (th.t == best_t && i32(tri_id) < best_id))
						{ best_t = th.t; best_id = i32(tri_id); best_bu = th.bu; best_bv = th.bv; }
					}
				}
			}
			else
			{
				float ta, tb;
				const bool ha = slab(bvh.nodes[n.child].box, o, id, tmin, best_t, &ta);
				const bool hb = slab(bvh.nodes[n.child + 1].box, o, id, tmin, best_t, &tb);
				if (ha && hb) { if (ta <= tb) { stack[sp++] = n.child + 1; stack[sp++] = n.child; } else { stack[sp++] = n.child; stack[sp++] = n.child + 1; } }
				else if (ha) stack[sp++] = n.child;
				else if (hb) stack[sp++] = n.child + 1;
			}
		}
		if (best_id >= 0)
		{
			// optix_base_shaders.h:50-57 : u = 1 - bx - by (vertex 0), v = bx (vertex 1); then fp16 round trip (optix_payload.h:75-78)
			const float u = 1.0f - best_bu - best_bv;
			const float v = best_bu;
			h.t = best_t; h.triId = best_id; h.u = h2f(f2h(u)); h.v = h2f(f2h(v));
		}
		return h;
	}
	// any hit with triangle masking : RTContext::trace_shadow, src/kernels/optix_rt.cu:133-164, optix_base_shadow_shaders.h:54-72
	Hit trace_shadow(const Ray& r) { return trace_shadow(r, nodes_visited, tris_tested); }
	Hit trace_shadow(const Ray& r, u64& nv, u64& tt) const
	{
		const V3 o(r.ox, r.oy, r.oz), d(r.dx, r.dy, r.dz);
		const u32 mask = r.mask_or_tmin;
		bool occluded = false;
		if (!bvh.nodes.empty())
		{
			const V3 id(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
			u32 stack[128]; int sp = 0; stack[sp++] = 0;
			while (sp && !occluded)
			{
				const BvhNode& n = bvh.nodes[stack[--sp]];
				float tn; nv++;
				if (!slab(n.box, o, id, 0.0f, r.tmax, &tn)) continue;
				if (n.leaf)
				{
					for (u32 i = n.begin; i < n.end && !occluded; ++i)
					{
						const u32 tri_id = bvh.index[i];
						const i32* tri = mesh->vertex_indices + 4 * tri_id;
						if (mask & u32(tri[3])) continue;
						TriHit th; tt++;
						if (intersect_tri(o, d, load_vertex(*mesh, tri[0]), load_vertex(*mesh, tri[1]), load_vertex(*mesh, tri[2]), 0.0f, r.tmax, vpad[tri_id], &th))
							occluded = true;
					}
				}
				else { stack[sp++] = n.child + 1; stack[sp++] = n.child; }
			}
		}
		Hit h;
		if (occluded) { h.t = 1.0f; h.triId = 1; } else { h.t = -1.0f; h.triId = -1; }
		h.u = 0.0f; h.v = 0.0f;
		return h;
	}
};

} // namespace orc
