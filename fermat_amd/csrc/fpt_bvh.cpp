// fpt_bvh.cpp — host-side builder of the 8-wide compressed BVH: a binned-SAH BVH2 for static scenes (the GPU build is host-side by design: scenes
// are static across passes, SURVEY §2.2 "cugar/bvh").  Topology is irrelevant to results (closest-t / lowest-id rule,
// DESIGN.md §5), so this builder is free to differ from the oracle's CUGAR full-sweep restatement: it bins centroids
// into 32 buckets per axis, which is O(n) per level and handles multi-million triangle scenes in seconds.
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

struct Builder
{
	static const int kBins = 32;
	uint32_t kLeaf = 4;             // max triangles per leaf (<= 7: 3-bit count in the leaf reference)
	const std::vector<Box>& boxes;
	std::vector<float> cx, cy, cz;
	std::vector<uint32_t> order;
	std::vector<BvhNode>& nodes;
	std::vector<uint32_t> leaf_first, leaf_count;   // filled through child refs
	uint32_t max_depth = 0;
	float cost = 0.0f;
	float root_area = 1.0f;

	Builder(const std::vector<Box>& b, std::vector<BvhNode>& n) : boxes(b), nodes(n)
	{
		const size_t N = b.size();
		cx.resize(N); cy.resize(N); cz.resize(N); order.resize(N);
		for (size_t i = 0; i < N; ++i)
		{
			cx[i] = 0.5f * (b[i].lo[0] + b[i].hi[0]); cy[i] = 0.5f * (b[i].lo[1] + b[i].hi[1]); cz[i] = 0.5f * (b[i].lo[2] + b[i].hi[2]);
			order[i] = uint32_t(i);
		}
	}
	const float* centroid_axis(int a) const { return a == 0 ? cx.data() : a == 1 ? cy.data() : cz.data(); }

	Box range_box(uint32_t b, uint32_t e) const { Box r; r.reset(); for (uint32_t i = b; i < e; ++i) r.grow(boxes[order[i]]); return r; }

	// returns the child reference for the range [b,e); `box` receives its bounds
	int32_t build(uint32_t b, uint32_t e, Box& box, uint32_t depth)
	{
		box = range_box(b, e);
		max_depth = std::max(max_depth, depth);
		const uint32_t n = e - b;
		if (n <= kLeaf)
		{
			cost += box.half_area() / root_area * float(n);
			return ~int32_t((b << 3) | n);
		}
		// centroid bounds
		float clo[3] = { 3.0e38f, 3.0e38f, 3.0e38f }, chi[3] = { -3.0e38f, -3.0e38f, -3.0e38f };
		for (uint32_t i = b; i < e; ++i)
		{
			const uint32_t t = order[i];
			const float c[3] = { cx[t], cy[t], cz[t] };
			for (int k = 0; k < 3; ++k) { clo[k] = std::min(clo[k], c[k]); chi[k] = std::max(chi[k], c[k]); }
		}
		float best = 3.0e38f; int best_axis = -1; int best_bin = 0;
		for (int a = 0; a < 3; ++a)
		{
			const float ext = chi[a] - clo[a];
			if (!(ext > 0.0f)) continue;
			const float scale = float(kBins) / ext;
			Box bb[kBins]; uint32_t cnt[kBins];
			for (int k = 0; k < kBins; ++k) { bb[k].reset(); cnt[k] = 0; }
			const float* ca = centroid_axis(a);
			for (uint32_t i = b; i < e; ++i)
			{
				const uint32_t t = order[i];
				int k = int((ca[t] - clo[a]) * scale); k = k < 0 ? 0 : (k >= kBins ? kBins - 1 : k);
				bb[k].grow(boxes[t]); cnt[k]++;
			}
			float rarea[kBins]; uint32_t rcnt[kBins];
			Box acc; acc.reset(); uint32_t c = 0;
			for (int k = kBins - 1; k > 0; --k) { acc.grow(bb[k]); c += cnt[k]; rarea[k] = acc.half_area(); rcnt[k] = c; }
			acc.reset(); c = 0;
			for (int k = 1; k < kBins; ++k)
			{
				acc.grow(bb[k - 1]); c += cnt[k - 1];
				if (c == 0 || rcnt[k] == 0) continue;
				const float s = acc.half_area() * float(c) + rarea[k] * float(rcnt[k]);
				if (s < best) { best = s; best_axis = a; best_bin = k; }
			}
		}
		uint32_t mid;
		if (depth > 30)
		{
			// a deep chain (strongly non-uniform scales peel off one primitive per level): from here on split at the object median of the
			// widest centroid axis, so the depth stays below 30 + log2(n) <= 58 < the 64-entry traversal stack whatever the input
			int a = 0;
			for (int k = 1; k < 3; ++k) if (chi[k] - clo[k] > chi[a] - clo[a]) a = k;
			const float* ca = centroid_axis(a);
			mid = b + n / 2;
			std::nth_element(order.begin() + b, order.begin() + mid, order.begin() + e, [&](uint32_t x, uint32_t y) { return ca[x] < ca[y]; });
		}
		else if (best_axis < 0)
			mid = b + n / 2;           // all centroids coincide: split the run in half
		else
		{
			const float* ca = centroid_axis(best_axis);
			const float scale = float(kBins) / (chi[best_axis] - clo[best_axis]);
			const float lo = clo[best_axis];
			uint32_t* first = order.data() + b; uint32_t* last = order.data() + e;
			uint32_t* m = std::partition(first, last, [&](uint32_t t) {
				int k = int((ca[t] - lo) * scale); k = k < 0 ? 0 : (k >= kBins ? kBins - 1 : k);
				return k < best_bin; });
			mid = uint32_t(m - order.data());
			if (mid == b || mid == e) mid = b + n / 2;
		}
		const uint32_t self = uint32_t(nodes.size());
		nodes.push_back(BvhNode());
		Box b0, b1;
		const int32_t c0 = build(b, mid, b0, depth + 1);
		const int32_t c1 = build(mid, e, b1, depth + 1);
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
	std::vector<Box> boxes(tri_count);
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
		boxes[t] = b;
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
	Builder bld(boxes, out.nodes);
	bld.kLeaf = std::max(1u, std::min(max_leaf, 4u));
	{
		Box rb; rb.reset(); for (uint32_t t = 0; t < tri_count; ++t) rb.grow(boxes[t]);
		bld.root_area = std::max(rb.half_area(), 1.0e-30f);
	}
	Box root_box;
	const int32_t root = bld.build(0, tri_count, root_box, 1);
	if (root < 0)
	{
		// <= 4 triangles: wrap the single leaf in a node with an empty sibling
		BvhNode n; std::memset(&n, 0, sizeof(n));
		for (int k = 0; k < 3; ++k) { n.lo0[k] = root_box.lo[k]; n.hi0[k] = root_box.hi[k]; n.lo1[k] = 3.0e38f; n.hi1[k] = -3.0e38f; }
		n.child0 = root; n.child1 = ~0;
		out.nodes.push_back(n);
	}
	else if (root != 0) throw std::runtime_error("fpt: internal BVH builder error (root is not node 0)");
	out.max_depth = bld.max_depth;
	out.sah_cost = bld.cost;
	// triangle records in leaf order
	out.tris.resize(tri_count);
	for (uint32_t i = 0; i < tri_count; ++i)
	{
		const uint32_t t = bld.order[i];
		const int32_t* ix = idx + 4 * size_t(t);
		const float* p0 = vtx + 4 * size_t(ix[0]); const float* p1 = vtx + 4 * size_t(ix[1]); const float* p2 = vtx + 4 * size_t(ix[2]);
		BvhTriangle& r = out.tris[i];
		for (int k = 0; k < 3; ++k) { r.v0[k] = p0[k]; r.e1[k] = p1[k] - p0[k]; r.e2[k] = p2[k] - p0[k]; }
		r.tri_id = int32_t(t); r.mask = uint32_t(ix[3]); r.pad = 0;
	}
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
