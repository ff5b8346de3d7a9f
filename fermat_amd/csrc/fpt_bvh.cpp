// fpt_bvh.cpp — host-side builder of the 8-wide compressed BVH: a binned-SAH BVH2 with spatial splits for static scenes (the GPU build is
// host-side by design: scenes are static across passes, SURVEY §2.2 "cugar/bvh"), then the 8-wide collapse.  Topology is irrelevant to
// results (closest-t / lowest-id rule), so this builder is free to differ from the oracle's CUGAR full-sweep restatement.
#include "fpt_bvh.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

namespace fpt {
namespace {

struct Box
{
	float lo[3], hi[3];
	void reset() { for (int k = 0; k < 3; ++k) { lo[k] = 3.0e38f; hi[k] = -3.0e38f; } }
	void grow(const Box& o) { for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], o.lo[k]); hi[k] = std::max(hi[k], o.hi[k]); } }
	void grow(const float* p) { for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], p[k]); hi[k] = std::max(hi[k], p[k]); } }
	float half_area() const
	{
		const float ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
		return (ex < 0 || ey < 0 || ez < 0) ? 0.0f : ex * ey + ez * (ex + ey);
	}
};

// A triangle reference: the triangle and the bounds of the part of it this subtree is responsible for (padded, see build_bvh2)
struct Ref { uint32_t tri; Box box; };

// Binned SAH with spatial splits (Stich, Friedrich, Dietrich: Spatial Splits in Bounding Volume Hierarchies, HPG 2009).  A node is split
// either by partitioning its references (object split, 32 centroid bins per axis) or by a plane that CUTS the references it crosses (spatial
// split, 32 bins over the node's extent; a cut triangle is referenced from both sides with clipped bounds; opt-in, see `spatial` below).
// Duplicated references change no result: the same triangle tested twice gives the same (t, id).
struct Builder
{
	static const int kBins = 32;
	uint32_t kLeaf = 3;             // max triangles per leaf (the wide collapse keeps a unary count in three meta bits)
	const int32_t* idx; const float* vtx; const std::vector<float>& pad;
	std::vector<BvhNode>& nodes;
	std::vector<BvhTriangle>& tris; // leaf records, appended leaf by leaf
	uint32_t max_depth = 0;
	float cost = 0.0f;
	float root_area = 1.0f;
	size_t n_refs = 0, ref_budget = 0;      // references alive in leaves so far + still to be placed; spatial splits stop at the budget
	bool spatial = false;           // measured on the two bench scenes (FPT_BVH_SPATIAL_SPLITS=1): stand-in 1477 vs 1556 Msample/s (node steps 5.1 -> 4.5 per ray but triangle
	                                // tests 4.4 -> 6.2: the cut room walls are tested from many leaves), testball-room 749 vs 738: off by default
	float spatial_alpha = 1.0e-5f;  // Stich et al.'s alpha: spatial splits are considered where the object split's children overlap by more than this share of the root's area

	Builder(const int32_t* i, const float* v, const std::vector<float>& p, std::vector<BvhNode>& n, std::vector<BvhTriangle>& t) : idx(i), vtx(v), pad(p), nodes(n), tris(t) {}

	// bounds of (triangle  intersected with  lo <= x[axis] <= hi), padded; false when the intersection is empty
	bool clip(uint32_t tri, int axis, double lo, double hi, Box& out) const
	{
		double poly[8][3], tmp[8][3]; int n = 3;
		for (int c = 0; c < 3; ++c) for (int k = 0; k < 3; ++k) poly[c][k] = double(vtx[4 * size_t(idx[4 * size_t(tri) + c]) + k]);
		for (int side = 0; side < 2 && n > 0; ++side)
		{
			const double plane = side ? hi : lo, sgn = side ? -1.0 : 1.0;      // keep sgn * (x - plane) >= 0
			int m = 0;
			for (int i = 0; i < n; ++i)
			{
				const double* A = poly[i]; const double* B = poly[(i + 1) % n];
				const double da = sgn * (A[axis] - plane), db = sgn * (B[axis] - plane);
				if (da >= 0.0) { for (int k = 0; k < 3; ++k) tmp[m][k] = A[k]; ++m; }
				if ((da > 0.0 && db < 0.0) || (da < 0.0 && db > 0.0))
				{
					const double t = da / (da - db);
					for (int k = 0; k < 3; ++k) tmp[m][k] = A[k] + t * (B[k] - A[k]);
					tmp[m][axis] = plane; ++m;
				}
			}
			n = m;
			for (int i = 0; i < n; ++i) for (int k = 0; k < 3; ++k) poly[i][k] = tmp[i][k];
		}
		if (n == 0) return false;
		const float pd = pad[tri];
		for (int k = 0; k < 3; ++k)
		{
			double mn = poly[0][k], mx = poly[0][k];
			for (int i = 1; i < n; ++i) { mn = std::min(mn, poly[i][k]); mx = std::max(mx, poly[i][k]); }
			// outward to fp32 (the interpolated points carry double rounding only), then the usual pad
			float flo = float(mn), fhi = float(mx);
			if (double(flo) > mn) flo = std::nextafter(flo, -3.0e38f);
			if (double(fhi) < mx) fhi = std::nextafter(fhi, 3.0e38f);
			out.lo[k] = flo - pd; out.hi[k] = fhi + pd;
		}
		return true;
	}
	static void intersect(Box& a, const Box& b) { for (int k = 0; k < 3; ++k) { a.lo[k] = std::max(a.lo[k], b.lo[k]); a.hi[k] = std::min(a.hi[k], b.hi[k]); } }

	int32_t make_leaf(const std::vector<Ref>& refs, const Box& box)
	{
		const uint32_t first = uint32_t(tris.size()), n = uint32_t(refs.size());
		if (first >= (1u << 28)) throw std::runtime_error("fpt: too many triangle references for the leaf encoding");
		for (const Ref& r : refs)
		{
			const int32_t* ix = idx + 4 * size_t(r.tri);
			const float* p0 = vtx + 4 * size_t(ix[0]); const float* p1 = vtx + 4 * size_t(ix[1]); const float* p2 = vtx + 4 * size_t(ix[2]);
			BvhTriangle t;
			for (int k = 0; k < 3; ++k) { t.v0[k] = p0[k]; t.e1[k] = p1[k] - p0[k]; t.e2[k] = p2[k] - p0[k]; }
			t.tri_id = int32_t(r.tri); t.mask = uint32_t(ix[3]); t.pad = 0;
			tris.push_back(t);
		}
		cost += box.half_area() / root_area * float(n);
		return ~int32_t((first << 3) | n);
	}

	// returns the child reference for the references in `refs` (consumed); `box` receives their bounds
	int32_t build(std::vector<Ref>& refs, Box& box, uint32_t depth)
	{
		box.reset();
		for (const Ref& r : refs) box.grow(r.box);
		max_depth = std::max(max_depth, depth);
		const uint32_t n = uint32_t(refs.size());
		if (n <= kLeaf) return make_leaf(refs, box);

		// ---- object split candidates: centroid bins ----
		float clo[3] = { 3.0e38f, 3.0e38f, 3.0e38f }, chi[3] = { -3.0e38f, -3.0e38f, -3.0e38f };
		for (const Ref& r : refs)
			for (int k = 0; k < 3; ++k) { const float c = 0.5f * (r.box.lo[k] + r.box.hi[k]); clo[k] = std::min(clo[k], c); chi[k] = std::max(chi[k], c); }
		float best = 3.0e38f; int best_axis = -1; int best_bin = 0; float best_overlap = 0.0f;
		for (int a = 0; a < 3; ++a)
		{
			const float ext = chi[a] - clo[a];
			if (!(ext > 0.0f)) continue;
			const float scale = float(kBins) / ext;
			Box bb[kBins]; uint32_t cnt[kBins];
			for (int k = 0; k < kBins; ++k) { bb[k].reset(); cnt[k] = 0; }
			for (const Ref& r : refs)
			{
				int k = int((0.5f * (r.box.lo[a] + r.box.hi[a]) - clo[a]) * scale); k = k < 0 ? 0 : (k >= kBins ? kBins - 1 : k);
				bb[k].grow(r.box); cnt[k]++;
			}
			Box rbox[kBins]; uint32_t rcnt[kBins];
			Box acc; acc.reset(); uint32_t c = 0;
			for (int k = kBins - 1; k > 0; --k) { acc.grow(bb[k]); c += cnt[k]; rbox[k] = acc; rcnt[k] = c; }
			acc.reset(); c = 0;
			for (int k = 1; k < kBins; ++k)
			{
				acc.grow(bb[k - 1]); c += cnt[k - 1];
				if (c == 0 || rcnt[k] == 0) continue;
				const float sc = acc.half_area() * float(c) + rbox[k].half_area() * float(rcnt[k]);
				if (sc < best)
				{
					best = sc; best_axis = a; best_bin = k;
					Box ov = acc; intersect(ov, rbox[k]); best_overlap = ov.half_area();
				}
			}
		}
		// ---- spatial split candidates: only where the object split leaves the children overlapping noticeably ----
		float sbest = 3.0e38f; int s_axis = -1; float s_plane = 0.0f;
		if (spatial && depth <= 30 && best_axis >= 0 && best_overlap / root_area > spatial_alpha && n_refs + n < ref_budget)
		{
			for (int a = 0; a < 3; ++a)
			{
				const float lo = box.lo[a], ext = box.hi[a] - box.lo[a];
				if (!(ext > 0.0f)) continue;
				const float scale = float(kBins) / ext;
				Box bb[kBins]; uint32_t enter[kBins], leave[kBins];
				for (int k = 0; k < kBins; ++k) { bb[k].reset(); enter[k] = leave[k] = 0; }
				for (const Ref& r : refs)
				{
					int k0 = int((r.box.lo[a] - lo) * scale), k1 = int((r.box.hi[a] - lo) * scale);
					k0 = k0 < 0 ? 0 : (k0 >= kBins ? kBins - 1 : k0); k1 = k1 < k0 ? k0 : (k1 >= kBins ? kBins - 1 : k1);
					enter[k0]++; leave[k1]++;
					if (k0 == k1) { bb[k0].grow(r.box); continue; }
					for (int k = k0; k <= k1; ++k)
					{
						Box c;
						const double plo = double(lo) + double(ext) * double(k) / kBins, phi = double(lo) + double(ext) * double(k + 1) / kBins;
						if (!clip(r.tri, a, plo, phi, c)) continue;
						intersect(c, r.box);
						if (c.lo[0] <= c.hi[0] && c.lo[1] <= c.hi[1] && c.lo[2] <= c.hi[2]) bb[k].grow(c);
					}
				}
				Box rbox[kBins]; uint32_t rcnt[kBins];
				Box acc; acc.reset(); uint32_t c = 0;
				for (int k = kBins - 1; k > 0; --k) { acc.grow(bb[k]); c += leave[k]; rbox[k] = acc; rcnt[k] = c; }
				acc.reset(); c = 0;
				for (int k = 1; k < kBins; ++k)
				{
					acc.grow(bb[k - 1]); c += enter[k - 1];
					if (c == 0 || rcnt[k] == 0 || c == n || rcnt[k] == n) continue;
					const float sc = acc.half_area() * float(c) + rbox[k].half_area() * float(rcnt[k]);
					if (sc < sbest) { sbest = sc; s_axis = a; s_plane = float(double(lo) + double(ext) * double(k) / kBins); }
				}
			}
		}
		std::vector<Ref> left, right;
		bool done = false;
		if (s_axis >= 0 && sbest < best)
		{
			// cut the references the plane crosses
			left.reserve(n); right.reserve(n);
			for (const Ref& r : refs)
			{
				if (r.box.hi[s_axis] <= s_plane) left.push_back(r);
				else if (r.box.lo[s_axis] >= s_plane) right.push_back(r);
				else
				{
					Ref L = r, R = r; Box c;
					bool hl = clip(r.tri, s_axis, -1.0e300, double(s_plane), c);
					if (hl) { intersect(c, r.box); hl = c.lo[0] <= c.hi[0] && c.lo[1] <= c.hi[1] && c.lo[2] <= c.hi[2]; if (hl) { L.box = c; left.push_back(L); } }
					bool hr = clip(r.tri, s_axis, double(s_plane), 1.0e300, c);
					if (hr) { intersect(c, r.box); hr = c.lo[0] <= c.hi[0] && c.lo[1] <= c.hi[1] && c.lo[2] <= c.hi[2]; if (hr) { R.box = c; right.push_back(R); } }
					if (!hl && !hr) left.push_back(r);          // numerically degenerate: keep it whole on one side
				}
			}
			if (!left.empty() && !right.empty() && left.size() < n && right.size() < n)
			{
				done = true;
				n_refs += left.size() + right.size() - n;
			}
			else { left.clear(); right.clear(); }
		}
		if (!done)
		{
			size_t mid;
			if (depth > 30 || best_axis < 0)
			{
				// a deep chain (strongly non-uniform scales peel off one primitive per level) or coinciding centroids: split at the object median of
				// the widest centroid axis, so that the depth stays below 30 + log2(n) < the traversal stack whatever the input
				int a = 0;
				for (int k = 1; k < 3; ++k) if (chi[k] - clo[k] > chi[a] - clo[a]) a = k;
				mid = n / 2;
				std::nth_element(refs.begin(), refs.begin() + mid, refs.end(), [&](const Ref& x, const Ref& y) { return x.box.lo[a] + x.box.hi[a] < y.box.lo[a] + y.box.hi[a]; });
			}
			else
			{
				const float scale = float(kBins) / (chi[best_axis] - clo[best_axis]);
				const float lo = clo[best_axis]; const int a = best_axis;
				auto m = std::partition(refs.begin(), refs.end(), [&](const Ref& r) {
					int k = int((0.5f * (r.box.lo[a] + r.box.hi[a]) - lo) * scale); k = k < 0 ? 0 : (k >= kBins ? kBins - 1 : k);
					return k < best_bin; });
				mid = size_t(m - refs.begin());
				if (mid == 0 || mid == n) mid = n / 2;
			}
			left.assign(refs.begin(), refs.begin() + mid); right.assign(refs.begin() + mid, refs.end());
		}
		std::vector<Ref>().swap(refs);          // release the parent's list before recursing
		const uint32_t self = uint32_t(nodes.size());
		nodes.push_back(BvhNode());
		Box b0, b1;
		const int32_t c0 = build(left, b0, depth + 1);
		const int32_t c1 = build(right, b1, depth + 1);
		BvhNode& nd = nodes[self];
		for (int k = 0; k < 3; ++k) { nd.lo0[k] = b0.lo[k]; nd.hi0[k] = b0.hi[k]; nd.lo1[k] = b1.lo[k]; nd.hi1[k] = b1.hi[k]; }
		nd.child0 = c0; nd.child1 = c1; nd.pad0 = nd.pad1 = 0;
		cost += box.half_area() / root_area * 1.0f;
		return int32_t(self);
	}
};

} // namespace

void build_bvh2(uint32_t tri_count, const int32_t* idx, uint32_t vertex_count, const float* vtx, HostBvh2& out, uint32_t max_leaf)
{
	out.nodes.clear(); out.tris.clear(); out.max_depth = 0; out.sah_cost = 0.0f;
	if (tri_count >= (1u << 28)) throw std::runtime_error("fpt: too many triangles for the leaf reference encoding");
	// scene magnitude for the conservative padding (rounding in the slab test must never cull a
	// triangle that the fpt-MT intersector accepts)
	float scene_mag = 0.0f;
	for (uint32_t v = 0; v < vertex_count; ++v)
		for (int k = 0; k < 3; ++k) scene_mag = std::max(scene_mag, std::fabs(vtx[4 * size_t(v) + k]));
	std::vector<Ref> refs(tri_count);
	std::vector<float> pads(tri_count);
	for (uint32_t t = 0; t < tri_count; ++t)
	{
		Box b; b.reset();
		float m0 = 0.0f;
		for (int c = 0; c < 3; ++c)
		{
			const int32_t vi = idx[4 * size_t(t) + c];
			if (vi < 0 || uint32_t(vi) >= vertex_count) throw std::runtime_error("fpt: vertex index out of range in create_geometry");
			const float* p = vtx + 4 * size_t(vi);
			b.grow(p);
			for (int k = 0; k < 3; ++k) m0 = std::max(m0, std::fabs(p[k]));
		}
		const float pad = 2.0e-6f * (m0 + scene_mag) + 1.0e-30f;
		for (int k = 0; k < 3; ++k) { b.lo[k] -= pad; b.hi[k] += pad; }
		refs[t].tri = t; refs[t].box = b; pads[t] = pad;
	}
	if (tri_count == 0)
	{
		// an empty scene still gets one node whose children are empty leaves, so kernels need no special case
		BvhNode n; std::memset(&n, 0, sizeof(n));
		for (int k = 0; k < 3; ++k) { n.lo0[k] = n.lo1[k] = 3.0e38f; n.hi0[k] = n.hi1[k] = -3.0e38f; }
		n.child0 = ~0; n.child1 = ~0;
		out.nodes.push_back(n);
		return;
	}
	Builder bld(idx, vtx, pads, out.nodes, out.tris);
	bld.kLeaf = std::max(1u, std::min(max_leaf, 4u));
	if (const char* e = std::getenv("FPT_BVH_MAX_LEAF")) bld.kLeaf = std::max(1u, std::min(uint32_t(std::atoi(e)), bld.kLeaf));      // tuning aid
	bld.n_refs = tri_count; bld.ref_budget = size_t(tri_count) + size_t(tri_count) / 2 + 64;       // at most ~50 % duplicated references
	if (const char* e = std::getenv("FPT_BVH_SPATIAL_SPLITS")) bld.spatial = std::atoi(e) != 0;      // tuning aid: 0 = object splits only
	if (const char* e = std::getenv("FPT_BVH_SPATIAL_ALPHA")) bld.spatial_alpha = float(std::atof(e));
	out.tris.reserve(bld.ref_budget);
	{
		Box rb; rb.reset(); for (uint32_t t = 0; t < tri_count; ++t) rb.grow(refs[t].box);
		bld.root_area = std::max(rb.half_area(), 1.0e-30f);
	}
	Box root_box;
	const int32_t root = bld.build(refs, root_box, 1);
	if (root < 0)
	{
		// a handful of triangles: wrap the single leaf in a node with an empty sibling
		BvhNode n; std::memset(&n, 0, sizeof(n));
		for (int k = 0; k < 3; ++k) { n.lo0[k] = root_box.lo[k]; n.hi0[k] = root_box.hi[k]; n.lo1[k] = 3.0e38f; n.hi1[k] = -3.0e38f; }
		n.child0 = root; n.child1 = ~0;
		out.nodes.push_back(n);
	}
	else if (root != 0) throw std::runtime_error("fpt: internal BVH builder error (root is not node 0)");
	out.max_depth = bld.max_depth;
	out.sah_cost = bld.cost;
}

// ---- 8-wide collapse ------------------------------------------------------------------------------------------------------------
namespace {
struct WideChild { int32_t ref; Box box; };
inline float center(const Box& b, int k) { return 0.5f * (b.lo[k] + b.hi[k]); }
} // namespace

void build_wide8(HostBvh2& bvh)
{
	bvh.nodes8.clear(); bvh.tris8.clear(); bvh.wide_depth = 0;
	std::vector<int32_t> queue;       // wide node i is the collapse of the binary subtree rooted at queue[i]
	std::vector<uint32_t> depth;
	queue.push_back(0); depth.push_back(1);
	auto child_of = [&](const BvhNode& n, int which) { WideChild c; c.ref = which ? n.child1 : n.child0; for (int k = 0; k < 3; ++k) { c.box.lo[k] = which ? n.lo1[k] : n.lo0[k]; c.box.hi[k] = which ? n.hi1[k] : n.hi0[k]; } return c; };
	for (size_t wi = 0; wi < queue.size(); ++wi)
	{
		bvh.wide_depth = std::max(bvh.wide_depth, depth[wi]);
		// greedy collapse: open the inner child with the largest surface area until there are eight children or only leaves
		std::vector<WideChild> ch;
		{
			const BvhNode& root = bvh.nodes[size_t(queue[wi])];
			ch.push_back(child_of(root, 0)); ch.push_back(child_of(root, 1));
		}
		while (ch.size() < 8)
		{
			int best = -1; float best_area = -1.0f;
			for (size_t i = 0; i < ch.size(); ++i)
				if (ch[i].ref >= 0 && ch[i].box.half_area() > best_area) { best_area = ch[i].box.half_area(); best = int(i); }
			if (best < 0) break;
			const BvhNode& n = bvh.nodes[size_t(ch[size_t(best)].ref)];
			ch[size_t(best)] = child_of(n, 0);
			ch.push_back(child_of(n, 1));
		}
		// empty leaves (padding of tiny scenes) carry nothing
		{
			std::vector<WideChild> kept;
			for (const WideChild& c : ch) if (c.ref >= 0 || (uint32_t(~c.ref) & 7u) != 0u) kept.push_back(c);
			ch.swap(kept);
		}
		Box nb; nb.reset();
		for (const WideChild& c : ch) nb.grow(c.box);
		if (ch.empty()) { for (int k = 0; k < 3; ++k) { nb.lo[k] = 0.0f; nb.hi[k] = 0.0f; } }
		// slot assignment: slot s looks along (s&4 ? +x : -x, s&2 ? +y : -y, s&1 ? +z : -z); greedily give each slot the child whose centre
		// lies furthest that way, so that (slot ^ (7 - octant)) descending visits near children first for every ray octant
		int slot_of[8]; bool slot_used[8] = { false, false, false, false, false, false, false, false };
		{
			std::vector<bool> done(ch.size(), false);
			for (size_t round = 0; round < ch.size(); ++round)
			{
				float best = -3.0e38f; int bc = -1, bs = -1;
				for (size_t c = 0; c < ch.size(); ++c)
				{
					if (done[c]) continue;
					for (int s = 0; s < 8; ++s)
					{
						if (slot_used[s]) continue;
						float cost = 0.0f;
						for (int k = 0; k < 3; ++k) cost += (center(ch[c].box, k) - center(nb, k)) * (((s >> (2 - k)) & 1) ? 1.0f : -1.0f);
						if (cost > best) { best = cost; bc = int(c); bs = s; }
					}
				}
				done[size_t(bc)] = true; slot_used[bs] = true; slot_of[bc] = bs;
			}
		}
		int child_in_slot[8] = { -1, -1, -1, -1, -1, -1, -1, -1 };
		for (size_t c = 0; c < ch.size(); ++c) child_in_slot[slot_of[c]] = int(c);

		BvhNode8 node; std::memset(&node, 0, sizeof(node));
		uint8_t* bytes = reinterpret_cast<uint8_t*>(node.w);
		std::memcpy(&node.w[0], &nb.lo[0], 4); std::memcpy(&node.w[1], &nb.lo[1], 4); std::memcpy(&node.w[2], &nb.lo[2], 4);
		// node-local grid: the smallest power-of-two cell that spans the node in 255 steps
		int ex[3];
		for (int k = 0; k < 3; ++k)
		{
			const double ext = double(nb.hi[k]) - double(nb.lo[k]);
			int e = -100;
			if (ext > 0.0)
			{
				e = int(std::ceil(std::log2(ext / 255.0)));
				while (ext / std::ldexp(1.0, e) > 255.0) ++e;
				while (e > -100 && ext / std::ldexp(1.0, e - 1) <= 255.0) --e;
			}
			e = std::max(-100, std::min(e, 120));
			ex[k] = e;
			bytes[12 + k] = uint8_t(e + 127);
		}
		uint32_t imask = 0;
		const uint32_t child_base = uint32_t(queue.size()), tri_base = uint32_t(bvh.tris8.size());
		node.w[4] = child_base; node.w[5] = tri_base;
		for (int s = 0; s < 8; ++s)
		{
			uint8_t* qlo[3] = { bytes + 32 + s, bytes + 40 + s, bytes + 48 + s };
			uint8_t* qhi[3] = { bytes + 56 + s, bytes + 64 + s, bytes + 72 + s };
			if (child_in_slot[s] < 0) { for (int k = 0; k < 3; ++k) { *qlo[k] = 255; *qhi[k] = 0; } continue; }      // empty slot: meta 0, inverted box
			const WideChild& c = ch[size_t(child_in_slot[s])];
			for (int k = 0; k < 3; ++k)
			{
				const double p = nb.lo[k], cell = std::ldexp(1.0, ex[k]);
				double lo = std::floor((double(c.box.lo[k]) - p) / cell); lo = lo < 0.0 ? 0.0 : (lo > 255.0 ? 255.0 : lo);
				while (lo > 0.0 && !(p + lo * cell <= double(c.box.lo[k]))) lo -= 1.0;
				double hi = std::ceil((double(c.box.hi[k]) - p) / cell); hi = hi < 0.0 ? 0.0 : (hi > 255.0 ? 255.0 : hi);
				while (hi < 255.0 && !(p + hi * cell >= double(c.box.hi[k]))) hi += 1.0;
				if (!(p + lo * cell <= double(c.box.lo[k])) || !(p + hi * cell >= double(c.box.hi[k]))) throw std::runtime_error("fpt: internal wide-BVH quantisation error");
				*qlo[k] = uint8_t(lo); *qhi[k] = uint8_t(hi);
			}
			if (c.ref >= 0)
			{
				imask |= 1u << s;
				bytes[24 + s] = uint8_t(0x20u | (24u + uint32_t(s)));
				queue.push_back(c.ref); depth.push_back(depth[wi] + 1);
			}
			else
			{
				const uint32_t leaf = uint32_t(~c.ref), first = leaf >> 3, count = leaf & 7u;
				if (count > 3) throw std::runtime_error("fpt: wide-BVH leaves hold at most 3 triangles");
				const uint32_t offset = uint32_t(bvh.tris8.size()) - tri_base;
				if (offset + count > 24) throw std::runtime_error("fpt: internal wide-BVH error (triangle range)");
				bytes[24 + s] = uint8_t((((1u << count) - 1u) << 5) | offset);
				for (uint32_t t = 0; t < count; ++t) bvh.tris8.push_back(bvh.tris[first + t]);
			}
		}
		bytes[15] = uint8_t(imask);
		bvh.nodes8.push_back(node);
	}
	if (bvh.tris8.empty()) { BvhTriangle z; std::memset(&z, 0, sizeof(z)); bvh.tris8.push_back(z); }
}

} // namespace fpt
