// fpt_bvh.cpp — host-side builder of the 8-wide compressed BVH: a binned-SAH BVH2 down to single triangles (multi-threaded; scenes are static
// across passes, SURVEY §2.2 "cugar/bvh"), then the SAH-optimal 8-wide collapse.  Topology is irrelevant to results (closest-t / lowest-id
// rule), so this builder is free to differ from the oracle's CUGAR full-sweep restatement.
#include "fpt_bvh.h"
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <atomic>
#include <exception>
#include <stdexcept>
#include <thread>
#include <system_error>

namespace fpt {
namespace {

struct Box
{
	float lo[3], hi[3];
	void reset() { for (int k = 0; k < 3; ++k) { lo[k] = 3.0e38f; hi[k] = -3.0e38f; } }
	void grow(const Box& o) { for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], o.lo[k]); hi[k] = std::max(hi[k], o.hi[k]); } }
	void grow(const float* p) { for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], p[k]); hi[k] = std::max(hi[k], p[k]); } }
	double half_area() const          // in double: extents of 1e19 are legal input and their products overflow fp32
	{
		const double ex = double(hi[0]) - double(lo[0]), ey = double(hi[1]) - double(lo[1]), ez = double(hi[2]) - double(lo[2]);
		return (ex < 0 || ey < 0 || ez < 0) ? 0.0 : ex * ey + ez * (ex + ey);
	}
};

// A triangle reference: the triangle and its (padded, see build_bvh2) bounds
struct Ref { uint32_t tri; Box box; };

static constexpr int32_t kDeferred = 0x40000000;      // child reference of a subtree handed to a worker: kDeferred + task index (node indices stay below)

struct Task
{
	std::vector<Ref> refs; uint32_t depth = 0;
	std::vector<BvhNode> nodes; std::vector<uint32_t> prims;
	int32_t root = 0; uint32_t max_depth = 0; double cost = 0.0;
};

// runs f(begin, end, slice) over contiguous slices of [0, n) on `threads` threads; the slices are a function of n and `slices` only
template <class F> void parallel_slices(size_t n, uint32_t slices, F f)
{
	if (slices <= 1) { f(size_t(0), n, 0u); return; }
	std::vector<std::thread> pool;
	std::vector<std::exception_ptr> error(slices);          // an exception must not leave a thread (std::terminate): it is rethrown on the caller's
	auto run = [&](uint32_t t) { try { f(n * t / slices, n * (t + 1) / slices, t); } catch (...) { error[t] = std::current_exception(); } };
	uint32_t started = 1;
	try { for (uint32_t t = 1; t < slices; ++t) { pool.emplace_back(run, t); started = t + 1; } } catch (const std::system_error&) {}
	run(0u);
	for (uint32_t t = started; t < slices; ++t) run(t);      // the slices of threads that could not be created (cgroup pid limit, EAGAIN): done here
	for (std::thread& t : pool) t.join();
	for (const std::exception_ptr& e : error) if (e) std::rethrow_exception(e);
}

// Binned SAH (32 centroid bins per axis), one triangle per leaf.  Below depth 30 (strongly non-uniform scales peel off one primitive per level)
// and where all centroids coincide the split is the object median of the widest axis, so the depth is bounded by 30 + log2(n) for any input.
// The large nodes at the top of the tree are binned and partitioned by all threads (slices in index order: the result is that of the serial code).
struct Builder
{
	static const int kBins = 32;        // 16 ... 512 measured with the traversal model: the tree's cost moves by +-3 % in either direction, and after the
	                                    // re-insertion pass the trees of 32 and 128 bins cost the same
	std::vector<BvhNode>& nodes;
	std::vector<uint32_t>& prims;   // triangle ids, appended leaf by leaf
	std::vector<Task>* defer;       // top phase: subtrees of at most `grain` references become tasks
	size_t grain = 0;
	uint32_t sah_depth = 30;        // SAH splits down to this depth, object medians below
	uint32_t threads = 1;           // top phase: threads for the big nodes
	uint32_t max_depth = 0;
	double cost = 0.0;
	double root_area = 1.0;

	Builder(std::vector<BvhNode>& n, std::vector<uint32_t>& p, std::vector<Task>* d) : nodes(n), prims(p), defer(d) {}

	int32_t make_leaf(const std::vector<Ref>& refs, const Box& box)
	{
		const uint32_t first = uint32_t(prims.size()), n = uint32_t(refs.size());
		for (const Ref& r : refs) prims.push_back(r.tri);
		cost += double(box.half_area()) / root_area * double(n);
		return ~int32_t((first << 3) | n);
	}

	struct Bins { Box bb[3][kBins]; uint32_t cnt[3][kBins]; };

	// returns the child reference for the references in `refs` (consumed); `box` receives their bounds
	int32_t build(std::vector<Ref>& refs, Box& box, uint32_t depth)
	{
		const uint32_t n = uint32_t(refs.size());
		const uint32_t slices = (defer && threads > 1 && n >= 65536u) ? threads : 1u;
		// bounds of the boxes and of their centres
		Box cb;
		{
			std::vector<Box> part(2 * size_t(slices));
			parallel_slices(n, slices, [&](size_t b, size_t e, uint32_t t) {
				Box bx, cx; bx.reset(); cx.reset();
				for (size_t i = b; i < e; ++i)
				{
					const Ref& r = refs[i];
					bx.grow(r.box);
					const float c[3] = { 0.5f * (r.box.lo[0] + r.box.hi[0]), 0.5f * (r.box.lo[1] + r.box.hi[1]), 0.5f * (r.box.lo[2] + r.box.hi[2]) };
					cx.grow(c);
				}
				part[2 * size_t(t)] = bx; part[2 * size_t(t) + 1] = cx; });
			box.reset(); cb.reset();
			for (uint32_t t = 0; t < slices; ++t) { box.grow(part[2 * size_t(t)]); cb.grow(part[2 * size_t(t) + 1]); }
		}
		if (defer && n <= grain && n > 1)
		{
			defer->emplace_back();
			defer->back().refs.swap(refs); defer->back().depth = depth;
			return kDeferred + int32_t(defer->size() - 1);
		}
		max_depth = std::max(max_depth, depth);
		if (n <= 1) return make_leaf(refs, box);

		const float* clo = cb.lo; const float* chi = cb.hi;
		double best = 1.0e300; int best_axis = -1; int best_bin = 0;
		if (depth <= sah_depth)
		{
			float scale[3]; bool live[3];
			for (int a = 0; a < 3; ++a) { const float ext = chi[a] - clo[a]; live[a] = ext > 0.0f; scale[a] = live[a] ? float(kBins) / ext : 0.0f; }
			std::vector<Bins> part(slices);
			parallel_slices(n, slices, [&](size_t b, size_t e, uint32_t t) {
				Bins& B = part[t];
				for (int a = 0; a < 3; ++a) for (int k = 0; k < kBins; ++k) { B.bb[a][k].reset(); B.cnt[a][k] = 0; }
				for (size_t i = b; i < e; ++i)
				{
					const Ref& r = refs[i];
					for (int a = 0; a < 3; ++a)
					{
						if (!live[a]) continue;
						int k = int((0.5f * (r.box.lo[a] + r.box.hi[a]) - clo[a]) * scale[a]); k = k < 0 ? 0 : (k >= kBins ? kBins - 1 : k);
						B.bb[a][k].grow(r.box); B.cnt[a][k]++;
					}
				} });
			Bins& B = part[0];
			for (uint32_t t = 1; t < slices; ++t)
				for (int a = 0; a < 3; ++a) for (int k = 0; k < kBins; ++k) { B.bb[a][k].grow(part[t].bb[a][k]); B.cnt[a][k] += part[t].cnt[a][k]; }
			for (int a = 0; a < 3; ++a)
			{
				if (!live[a]) continue;
				Box rbox[kBins]; uint32_t rcnt[kBins];
				Box acc; acc.reset(); uint32_t c = 0;
				for (int k = kBins - 1; k > 0; --k) { acc.grow(B.bb[a][k]); c += B.cnt[a][k]; rbox[k] = acc; rcnt[k] = c; }
				acc.reset(); c = 0;
				for (int k = 1; k < kBins; ++k)
				{
					acc.grow(B.bb[a][k - 1]); c += B.cnt[a][k - 1];
					if (c == 0 || rcnt[k] == 0) continue;
					const double sc = acc.half_area() * double(c) + rbox[k].half_area() * double(rcnt[k]);
					if (sc < best) { best = sc; best_axis = a; best_bin = k; }
				}
			}
		}
		std::vector<Ref> left, right;
		if (best_axis >= 0)
		{
			const float scl = float(kBins) / (chi[best_axis] - clo[best_axis]);
			const float lo = clo[best_axis]; const int a = best_axis;
			auto goes_left = [&](const Ref& r) {
				int k = int((0.5f * (r.box.lo[a] + r.box.hi[a]) - lo) * scl); k = k < 0 ? 0 : (k >= kBins ? kBins - 1 : k);
				return k < best_bin; };
			// stable partition, slice by slice
			std::vector<std::vector<Ref>> L(slices), R(slices);
			parallel_slices(n, slices, [&](size_t b, size_t e, uint32_t t) {
				L[t].reserve(e - b); R[t].reserve(e - b);
				for (size_t i = b; i < e; ++i) (goes_left(refs[i]) ? L[t] : R[t]).push_back(refs[i]); });
			if (slices == 1) { left.swap(L[0]); right.swap(R[0]); }
			else
			{
				size_t nl = 0, nr = 0;
				for (uint32_t t = 0; t < slices; ++t) { nl += L[t].size(); nr += R[t].size(); }
				left.reserve(nl); right.reserve(nr);
				for (uint32_t t = 0; t < slices; ++t) { left.insert(left.end(), L[t].begin(), L[t].end()); right.insert(right.end(), R[t].begin(), R[t].end()); }
			}
			if (left.empty() || right.empty()) { left.clear(); right.clear(); best_axis = -1; }
		}
		if (best_axis < 0)
		{
			int a = 0;
			for (int k = 1; k < 3; ++k) if (chi[k] - clo[k] > chi[a] - clo[a]) a = k;
			const size_t mid = n / 2;
			std::nth_element(refs.begin(), refs.begin() + mid, refs.end(), [&](const Ref& x, const Ref& y) {
				const float cx = x.box.lo[a] + x.box.hi[a], cy = y.box.lo[a] + y.box.hi[a];
				return cx < cy || (cx == cy && x.tri < y.tri); });
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
		cost += double(box.half_area()) / root_area;
		return int32_t(self);
	}
};

uint32_t builder_threads()
{
	// the GPU boxes show every hardware thread of the host but grant a cgroup quota of ~16: more threads than that only contend
	uint32_t n = std::thread::hardware_concurrency();
	n = n == 0 ? 1u : std::min(n, 16u);
	if (const char* e = std::getenv("FPT_BUILD_THREADS")) n = uint32_t(std::max(1, std::min(64, std::atoi(e))));
	return n;
}

// the constant part of the tolerance of fpt-MT's box clause for one triangle (fpt_trace.hip intersect_record, oracle/o_bvh.h intersect_tri): 1e-6 (|triangle|max + |scene|max)
float triangle_vpad(const float* p0, const float* p1, const float* p2, float scene_mag)
{
	float m0 = 0.0f;
	for (int k = 0; k < 3; ++k) m0 = std::max(m0, std::max(std::fabs(p0[k]), std::max(std::fabs(p1[k]), std::fabs(p2[k]))));
	return (m0 + scene_mag) * 1.0e-6f;
}

double now_seconds() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

} // namespace

void build_bvh2(uint32_t tri_count, const int32_t* idx, uint32_t vertex_count, const float* vtx, HostBvh2& out, uint32_t sah_depth)
{
	const double t0 = now_seconds();
	out.nodes.clear(); out.prims.clear(); out.max_depth = 0; out.sah_cost = 0.0f;
	if (tri_count >= (1u << 28)) throw std::runtime_error("fpt: too many triangles for the leaf reference encoding");
	// scene magnitude for the conservative padding (rounding in the slab test must never cull a
	// triangle that the fpt-MT intersector accepts)
	float scene_mag = 0.0f;
	for (uint32_t v = 0; v < vertex_count; ++v)
		for (int k = 0; k < 3; ++k) scene_mag = std::max(scene_mag, std::fabs(vtx[4 * size_t(v) + k]));
	out.scene_mag = scene_mag;
	std::vector<Ref> refs(tri_count);
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
		const float pad = (m0 + scene_mag) * 4.0e-6f + 1.0e-30f;          // four times the constant tolerance of fpt-MT's box clause (triangle_vpad): an accepted hit lies inside with margin
		for (int k = 0; k < 3; ++k) { b.lo[k] -= pad; b.hi[k] += pad; }
		refs[t].tri = t; refs[t].box = b;
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
	double root_area;
	{
		Box rb; rb.reset(); for (uint32_t t = 0; t < tri_count; ++t) rb.grow(refs[t].box);
		root_area = std::max(rb.half_area(), 1.0e-300);
	}
	const uint32_t n_threads = builder_threads();
	out.threads = n_threads;
	const double t_refs = now_seconds();
	out.prims.reserve(tri_count);
	// top phase (serial): split until the subtrees hold at most `grain` references; those become tasks
	std::vector<Task> tasks;
	Builder top(out.nodes, out.prims, &tasks);
	top.root_area = root_area; top.threads = n_threads; top.sah_depth = sah_depth;
	top.grain = (n_threads > 1 && tri_count >= 20000u) ? std::max<size_t>(4096, size_t(tri_count) / (size_t(n_threads) * 8)) : 0;
	Box root_box;
	int32_t root = top.build(refs, root_box, 1);
	double cost = top.cost; uint32_t max_depth = top.max_depth;
	const double t_top = now_seconds(); double t_tasks = t_top;
	if (!tasks.empty())
	{
		std::atomic<size_t> next(0);
		std::atomic<bool> failed(false);
		auto worker = [&]() {
			for (;;)
			{
				const size_t i = next.fetch_add(1);
				if (i >= tasks.size() || failed.load()) return;
				try
				{
					Task& T = tasks[i];
					Builder b(T.nodes, T.prims, nullptr);
					b.root_area = root_area; b.sah_depth = sah_depth;
					Box box;
					T.root = b.build(T.refs, box, T.depth);
					T.max_depth = b.max_depth; T.cost = b.cost;
				}
				catch (...) { failed.store(true); }
			}
		};
		// a thread that cannot be created (cgroup pid limit, EAGAIN) is not an error: the threads that did start, and this one, share its tasks
		std::vector<std::thread> pool;
		try { for (uint32_t t = 1; t < n_threads; ++t) pool.emplace_back(worker); } catch (const std::system_error&) {}
		worker();
		for (std::thread& t : pool) t.join();
		if (failed.load()) throw std::runtime_error("fpt: BVH builder worker failed (out of memory?)");
		t_tasks = now_seconds();
		// stitch the subtrees behind the top nodes in task order: the result does not depend on which thread built what
		std::vector<int32_t> task_root(tasks.size());
		for (size_t i = 0; i < tasks.size(); ++i)
		{
			Task& T = tasks[i];
			const int32_t node_base = int32_t(out.nodes.size()); const uint32_t prim_base = uint32_t(out.prims.size());
			auto fix = [&](int32_t ref) {
				if (ref >= 0) return ref + node_base;
				const uint32_t leaf = uint32_t(~ref);
				return ~int32_t((((leaf >> 3) + prim_base) << 3) | (leaf & 7u));
			};
			for (BvhNode n : T.nodes) { n.child0 = fix(n.child0); n.child1 = fix(n.child1); out.nodes.push_back(n); }
			out.prims.insert(out.prims.end(), T.prims.begin(), T.prims.end());
			task_root[i] = fix(T.root);
			cost += T.cost; max_depth = std::max(max_depth, T.max_depth);
			std::vector<BvhNode>().swap(T.nodes); std::vector<uint32_t>().swap(T.prims);
		}
		if (out.nodes.size() >= size_t(kDeferred)) throw std::runtime_error("fpt: too many BVH nodes");
		if (root >= kDeferred) root = task_root[size_t(root - kDeferred)];
		for (BvhNode& n : out.nodes)
		{
			if (n.child0 >= kDeferred) n.child0 = task_root[size_t(n.child0 - kDeferred)];
			if (n.child1 >= kDeferred) n.child1 = task_root[size_t(n.child1 - kDeferred)];
		}
	}
	if (root < 0)
	{
		// a single triangle: wrap the leaf in a node with an empty sibling
		BvhNode n; std::memset(&n, 0, sizeof(n));
		for (int k = 0; k < 3; ++k) { n.lo0[k] = root_box.lo[k]; n.hi0[k] = root_box.hi[k]; n.lo1[k] = 3.0e38f; n.hi1[k] = -3.0e38f; }
		n.child0 = root; n.child1 = ~0;
		out.nodes.push_back(n);
	}
	else if (root != 0) throw std::runtime_error("fpt: internal BVH builder error (root is not node 0)");
	out.max_depth = max_depth;
	out.sah_cost = float(cost);
	out.seconds_bvh2 = float(now_seconds() - t0);
	if (std::getenv("FPT_BVH_TIMERS")) std::fprintf(stderr, "build_bvh2: references %.3f s, top phase %.3f (%zu tasks), tasks %.3f, stitch %.3f\n", t_refs - t0, t_top - t_refs, tasks.size(), t_tasks - t_top, now_seconds() - t_tasks);
}

// ---- insertion-based optimisation of the binary tree -------------------------------------------------------------------------------
// Bittner, Hapala, Havran: Fast Insertion-Based Optimization of Bounding Volume Hierarchies (CGF 2013).  The top-down SAH build is greedy; this pass
// repeatedly takes the inner nodes that bound their children worst (large, with small or very unequal children), removes them, and re-inserts their two
// subtrees where they increase the tree's surface area least (branch and bound over the tree with the induced cost of the ancestors as the bound).
// Topology only: leaves keep their (padded) boxes, results of the intersector do not depend on it.
namespace {

struct ONode
{
	Box box; double area;
	int32_t parent, child[2];      // leaf: child[0] = -1
	uint32_t tri;
};

struct Optimizer
{
	std::vector<ONode> n;
	int32_t root = 0;
	uint32_t n_inner = 0;
	struct Item { double induced; int32_t node; bool operator<(const Item& o) const { return induced > o.induced; } };      // min-heap on the induced cost
	std::vector<Item> heap;
	// work bound: a search normally opens a few hundred nodes, but where everything overlaps everything (coincident geometry) the bound prunes nothing;
	// once the budget is spent a search settles for the best position seen so far (any position is valid) and the pass ends with the batch
	uint64_t visits = 0, budget = ~0ull;

	static Box merged(const Box& a, const Box& b) { Box r = a; r.grow(b); return r; }
	bool is_leaf(int32_t i) const { return n[size_t(i)].child[0] < 0; }

	// recompute boxes from node i up to the root (stops when nothing changes)
	void refit(int32_t i)
	{
		while (i >= 0)
		{
			ONode& X = n[size_t(i)];
			const Box b = merged(n[size_t(X.child[0])].box, n[size_t(X.child[1])].box);
			if (std::memcmp(&b, &X.box, sizeof(Box)) == 0) break;
			X.box = b; X.area = b.half_area();
			i = X.parent;
		}
	}
	void replace_child(int32_t parent, int32_t old_child, int32_t new_child)
	{
		if (parent < 0) { root = new_child; n[size_t(new_child)].parent = -1; return; }
		ONode& P = n[size_t(parent)];
		P.child[P.child[0] == old_child ? 0 : 1] = new_child;
		n[size_t(new_child)].parent = parent;
	}
	// the node next to which subtree x (detached) costs least: minimises S(X u x) + sum over the ancestors A of X of S(A u x) - S(A)
	// `hint`: a node of the tree whose position gives the search its first upper bound (the subtree's former neighbour: putting it back costs what it cost before);
	// the search then prunes from its first step instead of only after it has descended to a good candidate.  The result is the minimum either way; where several
	// positions cost exactly the same the hint wins.
	int32_t find_position(int32_t x, int32_t hint = -1)
	{
		const Box bx = n[size_t(x)].box; const double ax = n[size_t(x)].area;
		heap.clear();
		heap.push_back(Item{ 0.0, root });
		double best = 1.0e300; int32_t best_node = root;
		if (hint >= 0)
		{
			double total = merged(n[size_t(hint)].box, bx).half_area();
			for (int32_t a = n[size_t(hint)].parent; a >= 0; a = n[size_t(a)].parent) total += merged(n[size_t(a)].box, bx).half_area() - n[size_t(a)].area;
			best = total; best_node = hint;
		}
		while (!heap.empty())
		{
			std::pop_heap(heap.begin(), heap.end()); const Item it = heap.back(); heap.pop_back();
			if (it.induced + ax >= best || ++visits > budget) break;
			const ONode& X = n[size_t(it.node)];
			const double direct = merged(X.box, bx).half_area();
			const double total = it.induced + direct;
			if (total < best) { best = total; best_node = it.node; }
			const double below = total - X.area;          // induced cost for anything under X
			if (X.child[0] >= 0 && below + ax < best)
			{
				heap.push_back(Item{ below, X.child[0] }); std::push_heap(heap.begin(), heap.end());
				heap.push_back(Item{ below, X.child[1] }); std::push_heap(heap.begin(), heap.end());
			}
		}
		return best_node;
	}

	void insert(int32_t x, int32_t free_node, int32_t hint = -1)
	{
		const int32_t b = find_position(x, hint);
		ONode& F = n[size_t(free_node)];
		const int32_t bp = n[size_t(b)].parent;
		F.child[0] = b; F.child[1] = x;
		F.box = merged(n[size_t(b)].box, n[size_t(x)].box); F.area = F.box.half_area();
		replace_child(bp, b, free_node);
		n[size_t(b)].parent = free_node; n[size_t(x)].parent = free_node;
		refit(F.parent);
	}
	double cost() const
	{
		double c = 0.0;
		for (uint32_t i = 0; i < n_inner; ++i) c += n[i].area;
		return c / n[size_t(root)].area;
	}
	// one batch: the `count` worst inner nodes are removed and their children re-inserted
	double t_select = 0, t_apply = 0;
	uint32_t threads = 1;
	void batch(size_t count, std::vector<std::pair<double, int32_t>>& order)
	{
		const double tt0 = now_seconds();
		// the measure of every inner node, on all threads (slices in index order, concatenated in slice order: the list is that of the serial loop)
		const uint32_t th = n_inner >= 65536u ? std::max(1u, threads) : 1u;
		std::vector<std::vector<std::pair<double, int32_t>>> part(th);
		parallel_slices(size_t(n_inner), th, [&](size_t b, size_t e, uint32_t t) {
			std::vector<std::pair<double, int32_t>>& o = part[t];
			o.reserve((e - b) / 2 + 16);
			for (size_t i = b; i < e; ++i)
			{
				const ONode& X = n[i];
				if (int32_t(i) == root || X.parent == root) continue;
				const double a0 = n[size_t(X.child[0])].area, a1 = n[size_t(X.child[1])].area;
				const double amin = std::max(std::min(a0, a1), 1.0e-300), asum = std::max(0.5 * (a0 + a1), 1.0e-300);
				if (!(X.area > amin)) continue;          // a node no larger than either child (coincident geometry) has nothing to gain, and where every position
				                                         // costs the same the search would string such subtrees into a chain
				o.emplace_back(-(X.area / asum) * (X.area / amin) * X.area, int32_t(i));
			}
		});
		order.clear();
		for (const auto& o : part) order.insert(order.end(), o.begin(), o.end());
		count = std::min(count, order.size());
		if (count == 0) return;
		std::nth_element(order.begin(), order.begin() + (count - 1), order.end());
		std::sort(order.begin(), order.begin() + count);
		const double tt1 = now_seconds(); t_select += tt1 - tt0;
		for (size_t k = 0; k < count && visits <= budget; ++k)
		{
			const int32_t N = order[k].second;
			const int32_t P = n[size_t(N)].parent;
			if (N == root || P < 0 || P == root) continue;           // (the tree changes while the batch runs)
			const int32_t G = n[size_t(P)].parent;
			const int32_t S = n[size_t(P)].child[n[size_t(P)].child[0] == N ? 1 : 0];
			int32_t L = n[size_t(N)].child[0], R = n[size_t(N)].child[1];
			replace_child(G, P, S);
			refit(G);
			if (n[size_t(L)].area < n[size_t(R)].area) std::swap(L, R);
			static const bool use_hint = std::getenv("FPT_BVH_NO_HINT") == nullptr;
			insert(L, N, use_hint ? S : -1);
			insert(R, P, use_hint ? S : -1);
		}
		t_apply += now_seconds() - tt1;
	}
};

} // namespace

void optimize_bvh2(HostBvh2& bvh, uint32_t max_iterations, double batch_fraction)
{
	const double t0 = now_seconds();
	const size_t ni = bvh.nodes.size();
	if (ni < 4 || max_iterations == 0) return;
	for (const BvhNode& N : bvh.nodes) for (int32_t r : { N.child0, N.child1 }) if (r < 0 && (uint32_t(~r) & 7u) != 1u) return;      // built for one triangle per leaf
	Optimizer O; O.n_inner = uint32_t(ni); O.threads = builder_threads();
	O.n.resize(2 * ni + 1);
	size_t n_nodes = ni;
	// Node numbers are the pre-order ranks of the binary tree (inner nodes 0 .. ni-1, leaves behind them in the order a depth-first walk meets them), NOT the
	// positions build_bvh2 left them at: those depend on how the tree was cut into tasks, i.e. on the number of threads, and the pass breaks ties between equal
	// measures and equal costs by node number (on the 1.8 M-triangle bench scene 3 and 8 threads gave trees of 180 885 and 180 884 wide nodes until round 5)
	std::vector<int32_t> id_of(ni, -1), leaf_id(2 * ni, -1);
	{
		std::vector<int32_t> stack; stack.push_back(0);
		int32_t next_inner = 0;
		while (!stack.empty())
		{
			const int32_t i = stack.back(); stack.pop_back();
			id_of[size_t(i)] = next_inner++;
			const BvhNode& N = bvh.nodes[size_t(i)];
			if (N.child0 < 0) leaf_id[2 * size_t(i)] = int32_t(n_nodes++);
			if (N.child1 < 0) leaf_id[2 * size_t(i) + 1] = int32_t(n_nodes++);          // (a leaf under child1 is met after child0's whole subtree; its number only has to be canonical)
			if (N.child1 >= 0) stack.push_back(N.child1);
			if (N.child0 >= 0) stack.push_back(N.child0);
		}
		if (size_t(next_inner) != ni) return;          // not a tree over all its nodes: leave it alone
	}
	for (size_t i = 0; i < ni; ++i) O.n[i].parent = -1;
	for (size_t i = 0; i < ni; ++i)
	{
		const BvhNode& N = bvh.nodes[i];
		const int32_t ref[2] = { N.child0, N.child1 };
		const int32_t me = id_of[i];
		Box cb[2];
		for (int k = 0; k < 3; ++k) { cb[0].lo[k] = N.lo0[k]; cb[0].hi[k] = N.hi0[k]; cb[1].lo[k] = N.lo1[k]; cb[1].hi[k] = N.hi1[k]; }
		for (int c = 0; c < 2; ++c)
		{
			int32_t id;
			if (ref[c] >= 0) id = id_of[size_t(ref[c])];
			else
			{
				id = leaf_id[2 * i + size_t(c)];
				ONode& Lf = O.n[size_t(id)];
				Lf.child[0] = Lf.child[1] = -1; Lf.tri = bvh.prims[uint32_t(~ref[c]) >> 3];
			}
			O.n[size_t(id)].box = cb[c]; O.n[size_t(id)].area = cb[c].half_area(); O.n[size_t(id)].parent = me;
			O.n[size_t(me)].child[c] = id;
		}
	}
	O.n.resize(n_nodes);
	{ ONode& R = O.n[0]; R.box = Optimizer::merged(O.n[size_t(R.child[0])].box, O.n[size_t(R.child[1])].box); R.area = R.box.half_area(); R.parent = -1; }
	O.root = 0;
	std::vector<std::pair<double, int32_t>> order; order.reserve(ni);
	const size_t per_batch = std::max<size_t>(1, size_t(double(ni) * batch_fraction));
	double best = O.cost(); uint32_t stale = 0;
	bvh.opt_cost_before = float(best);
	uint32_t it = 0;
	// (nearly all of the gain comes with the first batches -- the few thousand nodes a centroid-binned build gets badly wrong, large triangles filed
	//  among small ones; pseudo-random batches after the measure-driven ones stall were tried and find nothing more)
	O.budget = 64ull * uint64_t(n_nodes);          // the bench scenes use 2-3 node visits per node of the tree and batch
	for (; it < max_iterations && stale < 2 && O.visits <= O.budget; ++it)
	{
		O.batch(per_batch, order);
		const double c = O.cost();
		if (c < best * (1.0 - 1.0e-3)) stale = 0; else ++stale;
		best = std::min(best, c);
	}
	bvh.opt_cost_after = float(O.cost()); bvh.opt_iterations = it;
	if (std::getenv("FPT_BVH_TIMERS")) std::fprintf(stderr, "optimize: setup+loop %.3f s, select %.3f, apply %.3f, visits %llu\n", now_seconds() - t0, O.t_select, O.t_apply, (unsigned long long)O.visits);
	// back into the array form: pre-order, parents before children (build_wide8's bottom-up pass walks the array backwards), leaves in the order met
	std::vector<BvhNode> out; out.reserve(ni);
	std::vector<uint32_t> prims; prims.reserve(bvh.prims.size());
	struct Todo { int32_t node; int32_t out_parent; int which; uint32_t depth; };
	std::vector<Todo> stack; stack.push_back(Todo{ O.root, -1, 0, 1 });
	uint32_t max_depth = 0; double cost = 0.0; const double root_area = std::max(O.n[size_t(O.root)].area, 1.0e-300);
	while (!stack.empty())
	{
		const Todo t = stack.back(); stack.pop_back();
		const ONode& X = O.n[size_t(t.node)];
		int32_t ref;
		if (X.child[0] < 0)
		{
			ref = ~int32_t((uint32_t(prims.size()) << 3) | 1u); prims.push_back(X.tri);
			cost += X.area / root_area;
		}
		else
		{
			ref = int32_t(out.size());
			BvhNode N; std::memset(&N, 0, sizeof(N));
			out.push_back(N);
			max_depth = std::max(max_depth, t.depth);
			cost += X.area / root_area;
			stack.push_back(Todo{ X.child[1], ref, 1, t.depth + 1 });
			stack.push_back(Todo{ X.child[0], ref, 0, t.depth + 1 });
		}
		if (t.out_parent >= 0)
		{
			BvhNode& P = out[size_t(t.out_parent)];
			if (t.which == 0) { P.child0 = ref; for (int k = 0; k < 3; ++k) { P.lo0[k] = X.box.lo[k]; P.hi0[k] = X.box.hi[k]; } }
			else              { P.child1 = ref; for (int k = 0; k < 3; ++k) { P.lo1[k] = X.box.lo[k]; P.hi1[k] = X.box.hi[k]; } }
		}
	}
	bvh.nodes.swap(out); bvh.prims.swap(prims);
	bvh.max_depth = max_depth; bvh.sah_cost = float(cost);
	bvh.seconds_opt = float(now_seconds() - t0);
}

// ---- 8-wide collapse ------------------------------------------------------------------------------------------------------------
namespace {

// SAH-optimal collapse (Ylitie, Karras, Laine 2017, section 3.1).  For every binary node n and i = 1..7:
//   C(n, 1) = min( C_leaf(n), C_internal(n) )                      n is the root of ONE wide-node child: a leaf or a wide node
//   C(n, i) = min( C_distribute(n, i), C(n, i-1) )                 the subtree of n is represented by at most i child slots of some wide node
//   C_leaf(n) = A_n P_n c_prim  (P_n <= 3 triangles)               C_internal(n) = C_distribute(n, 8) + A_n c_node
//   C_distribute(n, j) = min over 0 < k < j of C(left, k) + C(right, j - k)
// A_n = surface area relative to the root's.  c_prim / c_node is the price of a triangle test against a node step in fpt_trace.hip
// (100 against 228 VALU instructions).
struct Collapse
{
	static constexpr float c_node = 1.0f;
	float c_prim = 0.45f;          // swept 0.2 .. 1.0 on the two bench scenes with tools/bvh_stats.py: the traversal cost model moves by < 1.5 %
	struct Cell { float c[8]; uint8_t k[8]; uint8_t k8; uint8_t leaf; uint8_t count; };      // index 1..7 used; count = min(P_n, 255)
	const std::vector<BvhNode>& nodes;
	std::vector<Cell> cell;
	double root_area = 1.0;

	explicit Collapse(const std::vector<BvhNode>& n) : nodes(n) {}

	static Box box_of(const BvhNode& n, int which)
	{
		Box b;
		for (int k = 0; k < 3; ++k) { b.lo[k] = which ? n.lo1[k] : n.lo0[k]; b.hi[k] = which ? n.hi1[k] : n.hi0[k]; }
		return b;
	}
	static uint32_t leaf_count(int32_t ref) { return uint32_t(~ref) & 7u; }

	// cost row of a child reference (inner node: its cell; binary leaf: the same price for every i)
	void row(int32_t ref, const Box& b, float* c, uint32_t& count) const
	{
		if (ref >= 0) { for (int i = 1; i <= 7; ++i) c[i] = cell[size_t(ref)].c[i]; count = cell[size_t(ref)].count; return; }
		count = leaf_count(ref);
		const float v = float(b.half_area() / root_area) * float(count) * c_prim;
		for (int i = 1; i <= 7; ++i) c[i] = v;
	}

	void solve_node(size_t n)
	{
		const BvhNode& N = nodes[n];
		const Box b0 = box_of(N, 0), b1 = box_of(N, 1);
		Box nb = b0; nb.grow(b1);
		const float area = float(nb.half_area() / root_area);
		float cl[8], cr[8]; uint32_t pl, pr;
		row(N.child0, b0, cl, pl); row(N.child1, b1, cr, pr);
		Cell& X = cell[n];
		const uint32_t P = pl + pr;
		X.count = uint8_t(std::min(P, 255u));
		float dist[9]; uint8_t dk[9];
		for (int j = 2; j <= 8; ++j)
		{
			dist[j] = 3.0e38f; dk[j] = 1;
			for (int k = 1; k < j; ++k)
			{
				if (k > 7 || j - k > 7) continue;
				const float v = cl[k] + cr[j - k];
				if (v < dist[j]) { dist[j] = v; dk[j] = uint8_t(k); }
			}
		}
		const float c_internal = dist[8] + area * c_node;
		const float c_leaf = (P >= 1 && P <= 3) ? area * float(P) * c_prim : 3.0e38f;
		X.k8 = dk[8];
		X.leaf = c_leaf <= c_internal ? 1 : 0;
		X.c[0] = 0.0f; X.k[0] = 0;
		X.c[1] = X.leaf ? c_leaf : c_internal; X.k[1] = 0;
		for (int i = 2; i <= 7; ++i)
		{
			if (dist[i] < X.c[i - 1]) { X.c[i] = dist[i]; X.k[i] = dk[i]; }
			else { X.c[i] = X.c[i - 1]; X.k[i] = 0; }
		}
	}
	// children have larger indices than their parents, so the array is solved backwards.  When it is in PRE-ORDER (optimize_bvh2 writes it so: a subtree is a
	// contiguous range [root, end)), disjoint subtrees are solved on all threads and the nodes above them afterwards: a cell depends on its children's cells only,
	// so the result is that of the backward loop.
	void solve(uint32_t threads)
	{
		cell.resize(nodes.size());
		{
			Box rb = box_of(nodes[0], 0); rb.grow(box_of(nodes[0], 1));
			root_area = std::max(rb.half_area(), 1.0e-300);
		}
		const size_t N = nodes.size();
		struct Range { size_t begin, end; };
		std::vector<Range> work;              // disjoint subtrees
		std::vector<size_t> above;            // the nodes above them, in increasing index order
		bool preorder = threads > 1 && N >= 65536;
		if (preorder)
		{
			std::vector<Range> todo; todo.push_back(Range{ 0, N });
			const size_t grain = std::max<size_t>(4096, N / (size_t(threads) * 16));
			while (!todo.empty() && preorder)
			{
				const Range r = todo.back(); todo.pop_back();
				if (r.end - r.begin <= grain) { work.push_back(r); continue; }
				const BvhNode& X = nodes[r.begin];
				above.push_back(r.begin);
				// pre-order: an inner child0 is the next node, an inner child1 follows child0's subtree and runs to the end of the range
				size_t next = r.begin + 1;
				if (X.child0 >= 0)
				{
					if (size_t(X.child0) != next) { preorder = false; break; }
					const size_t e0 = X.child1 >= 0 ? size_t(X.child1) : r.end;
					if (e0 <= next || e0 > r.end) { preorder = false; break; }
					todo.push_back(Range{ next, e0 }); next = e0;
				}
				if (X.child1 >= 0)
				{
					if (size_t(X.child1) != next) { preorder = false; break; }
					todo.push_back(Range{ next, r.end }); next = r.end;
				}
				if (next != r.end) { preorder = false; break; }
			}
		}
		if (!preorder) { for (size_t n = N; n-- > 0;) solve_node(n); return; }
		// every reference inside a range must stay inside it (checked while solving: a child outside its parent's range means the array is not what it seemed)
		std::atomic<size_t> next_task(0); std::atomic<bool> bad(false);
		parallel_slices(size_t(threads), threads, [&](size_t, size_t, uint32_t) {
			for (;;)
			{
				const size_t t = next_task.fetch_add(1);
				if (t >= work.size()) return;
				const Range r = work[t];
				for (size_t n = r.end; n-- > r.begin;)
				{
					const BvhNode& X = nodes[n];
					if ((X.child0 >= 0 && (size_t(X.child0) <= n || size_t(X.child0) >= r.end)) || (X.child1 >= 0 && (size_t(X.child1) <= n || size_t(X.child1) >= r.end))) { bad.store(true); return; }
					solve_node(n);
				}
			}
		});
		if (bad.load()) { for (size_t n = N; n-- > 0;) solve_node(n); return; }
		std::sort(above.begin(), above.end());
		for (size_t i = above.size(); i-- > 0;) solve_node(above[i]);
	}
};

struct WideChild { int32_t ref; Box box; uint32_t n_prims; uint32_t prim[3]; };      // ref >= 0: the binary node that roots an inner child; < 0: a leaf of n_prims triangles
inline float center(const Box& b, int k) { return 0.5f * (b.lo[k] + b.hi[k]); }

// exact assignment of <= 8 children to the 8 slots maximising the summed score (Kuhn-Munkres on the 8 x 8 matrix, rows padded with zeros)
void assign_slots(const double score[8][8], int n_children, int slot_of[8])
{
	const int N = 8;
	double a[N + 1][N + 1];
	for (int i = 1; i <= N; ++i) for (int j = 1; j <= N; ++j) a[i][j] = (i <= n_children) ? -score[i - 1][j - 1] : 0.0;
	double u[N + 1] = { 0 }, v[N + 1] = { 0 }; int p[N + 1] = { 0 }, way[N + 1] = { 0 };
	for (int i = 1; i <= N; ++i)
	{
		p[0] = i; int j0 = 0;
		double minv[N + 1]; bool used[N + 1];
		for (int j = 0; j <= N; ++j) { minv[j] = 1.0e300; used[j] = false; }
		do
		{
			used[j0] = true;
			const int i0 = p[j0]; double delta = 1.0e300; int j1 = 0;
			for (int j = 1; j <= N; ++j)
				if (!used[j])
				{
					const double cur = a[i0][j] - u[i0] - v[j];
					if (cur < minv[j]) { minv[j] = cur; way[j] = j0; }
					if (minv[j] < delta) { delta = minv[j]; j1 = j; }
				}
			for (int j = 0; j <= N; ++j)
				if (used[j]) { u[p[j]] += delta; v[j] -= delta; } else minv[j] -= delta;
			j0 = j1;
		} while (p[j0] != 0);
		do { const int j1 = way[j0]; p[j0] = p[j1]; j0 = j1; } while (j0);
	}
	for (int j = 1; j <= N; ++j) if (p[j] >= 1 && p[j] <= n_children) slot_of[p[j] - 1] = j - 1;
}

} // namespace

void build_wide8(uint32_t tri_count, const int32_t* idx, const float* vtx, HostBvh2& bvh)
{
	const double t0 = now_seconds();
	bvh.nodes8.clear(); bvh.tris8.clear(); bvh.wide_depth = 0; bvh.stack_need = 0; bvh.wide_cost = 0.0f;
	bvh.n_inner_children = bvh.n_leaf_children = 0;
	for (int k = 0; k < 9; ++k) bvh.slot_hist[k] = 0;
	Collapse dp(bvh.nodes);
	dp.solve(builder_threads());
	const double t_dp = now_seconds();
	bvh.wide_cost = dp.cell[0].c[1];

	auto leaf_child = [&](int32_t ref, const Box& b) {
		WideChild c; c.ref = -1; c.box = b; c.n_prims = 0; c.prim[0] = c.prim[1] = c.prim[2] = 0;
		const uint32_t leaf = uint32_t(~ref), first = leaf >> 3, count = leaf & 7u;
		for (uint32_t t = 0; t < count && c.n_prims < 3; ++t) c.prim[c.n_prims++] = bvh.prims[first + t];
		return c;
	};
	// the triangles below a binary subtree that the collapse turns into one leaf (<= 3)
	struct Gather { const HostBvh2& B; WideChild& c; void run(int32_t ref) {
		if (ref < 0) { const uint32_t leaf = uint32_t(~ref), first = leaf >> 3, count = leaf & 7u;
		               for (uint32_t t = 0; t < count; ++t) { if (c.n_prims >= 3) throw std::runtime_error("fpt: internal wide-BVH error (leaf size)"); c.prim[c.n_prims++] = B.prims[first + t]; } return; }
		run(B.nodes[size_t(ref)].child0); run(B.nodes[size_t(ref)].child1); } };
	// the children that the subtree behind (ref, box) contributes to a wide node when it may use at most `budget` slots
	struct Collect { const HostBvh2& B; const Collapse& dp; std::vector<WideChild>& out; decltype(leaf_child)& mk_leaf;
		void run(int32_t ref, const Box& b, int budget)
		{
			if (ref < 0) { if ((uint32_t(~ref) & 7u) != 0u) out.push_back(mk_leaf(ref, b)); return; }      // empty leaves (padding of tiny scenes) carry nothing
			const Collapse::Cell& X = dp.cell[size_t(ref)];
			int i = budget;
			while (i > 1 && X.k[i] == 0) --i;
			if (i == 1)
			{
				WideChild c; c.box = b; c.n_prims = 0; c.prim[0] = c.prim[1] = c.prim[2] = 0;
				if (X.leaf) { c.ref = -1; Gather g{ B, c }; g.run(ref); }
				else c.ref = ref;
				out.push_back(c);
				return;
			}
			const BvhNode& N = B.nodes[size_t(ref)];
			run(N.child0, Collapse::box_of(N, 0), int(X.k[i]));
			run(N.child1, Collapse::box_of(N, 1), i - int(X.k[i]));
		} };

	// Wide nodes are numbered in breadth-first order (inner children of a node contiguous, in slot order).  A wide node's content depends only on the binary subtree it
	// collapses, so the nodes of a LEVEL are worked out on all threads (children, slot assignment, quantised boxes, the triangles of their leaves); a serial pass over
	// the level then hands out child and triangle bases in order -- the arrays are those of a node-by-node loop, whatever the number of threads.
	struct Emit { BvhNode8 node; uint32_t n_children, n_inner, n_leaf, n_tris; int32_t inner_ref[8]; uint32_t tri[24]; };
	auto emit_node = [&](int32_t binary_root, Emit& E, std::vector<WideChild>& ch)
	{
		ch.clear();
		{
			const BvhNode& root = bvh.nodes[size_t(binary_root)];
			const Collapse::Cell& X = dp.cell[size_t(binary_root)];
			Collect col{ bvh, dp, ch, leaf_child };
			col.run(root.child0, Collapse::box_of(root, 0), int(X.k8));
			col.run(root.child1, Collapse::box_of(root, 1), 8 - int(X.k8));
			if (ch.size() > 8) throw std::runtime_error("fpt: internal wide-BVH error (more than eight children)");
		}
		E.n_children = uint32_t(ch.size()); E.n_inner = E.n_leaf = E.n_tris = 0;
		Box nb; nb.reset();
		for (const WideChild& c : ch) nb.grow(c.box);
		if (ch.empty()) { for (int k = 0; k < 3; ++k) { nb.lo[k] = 0.0f; nb.hi[k] = 0.0f; } }
		// slot assignment: slot s looks along (s&4 ? +x : -x, s&2 ? +y : -y, s&1 ? +z : -z); the assignment maximises the sum over children of
		// (child centre - node centre) . direction of its slot, so that (slot ^ (7 - octant)) descending visits near children first for every ray octant
		int slot_of[8] = { 0, 1, 2, 3, 4, 5, 6, 7 };
		{
			double score[8][8];
			for (size_t c = 0; c < ch.size(); ++c)
				for (int sl = 0; sl < 8; ++sl)
				{
					double v = 0.0;
					for (int k = 0; k < 3; ++k) v += (double(center(ch[c].box, k)) - double(center(nb, k))) * (((sl >> (2 - k)) & 1) ? 1.0 : -1.0);
					score[c][sl] = v;
				}
			assign_slots(score, int(ch.size()), slot_of);
		}
		int child_in_slot[8] = { -1, -1, -1, -1, -1, -1, -1, -1 };
		for (size_t c = 0; c < ch.size(); ++c) child_in_slot[slot_of[c]] = int(c);

		BvhNode8& node = E.node; std::memset(&node, 0, sizeof(node));
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
		for (int sl = 0; sl < 8; ++sl)
		{
			uint8_t* qlo[3] = { bytes + 32 + sl, bytes + 40 + sl, bytes + 48 + sl };
			uint8_t* qhi[3] = { bytes + 56 + sl, bytes + 64 + sl, bytes + 72 + sl };
			if (child_in_slot[sl] < 0) { for (int k = 0; k < 3; ++k) { *qlo[k] = 255; *qhi[k] = 0; } continue; }      // empty slot: meta 0, inverted box
			const WideChild& c = ch[size_t(child_in_slot[sl])];
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
				imask |= 1u << sl;
				bytes[24 + sl] = uint8_t(0x20u | (24u + uint32_t(sl)));
				E.inner_ref[E.n_inner++] = c.ref;
			}
			else
			{
				const uint32_t count = c.n_prims;
				if (count < 1 || count > 3) throw std::runtime_error("fpt: wide-BVH leaves hold 1..3 triangles");
				const uint32_t offset = E.n_tris;          // relative to the node's triangle base
				if (offset + count > 24) throw std::runtime_error("fpt: internal wide-BVH error (triangle range)");
				bytes[24 + sl] = uint8_t((((1u << count) - 1u) << 5) | offset);
				for (uint32_t t = 0; t < count; ++t) E.tri[E.n_tris++] = c.prim[t];
				E.n_leaf++;
			}
		}
		bytes[15] = uint8_t(imask);
	};
	auto write_records = [&](const Emit& E, BvhTriangle* out)
	{
		for (uint32_t t = 0; t < E.n_tris; ++t)
		{
			const uint32_t tri = E.tri[t];
			const int32_t* ix = idx + 4 * size_t(tri);
			const float* p0 = vtx + 4 * size_t(ix[0]); const float* p1 = vtx + 4 * size_t(ix[1]); const float* p2 = vtx + 4 * size_t(ix[2]);
			BvhTriangle r;
			for (int k = 0; k < 3; ++k) { r.v0[k] = p0[k]; r.e1[k] = p1[k] - p0[k]; r.e2[k] = p2[k] - p0[k]; }
			r.tri_id = int32_t(tri); r.mask = uint32_t(ix[3]); r.vpad = triangle_vpad(p0, p1, p2, bvh.scene_mag);
			out[t] = r;
		}
	};

	std::vector<int32_t> queue;       // wide node i is the collapse of the binary subtree rooted at queue[i]
	queue.push_back(0);
	const uint32_t n_threads = builder_threads();
	std::vector<Emit> level;
	std::vector<size_t> tri_base_of;
	size_t tri_total = 0;
	bvh.tris8.clear();
	std::vector<BvhTriangle> records(size_t(tri_count) + 1);          // sized once: every triangle lands in exactly one leaf
	for (size_t lb = 0, depth = 1; lb < queue.size(); ++depth)
	{
		const size_t le = queue.size(), n_level = le - lb;
		bvh.wide_depth = std::max(bvh.wide_depth, uint32_t(depth));
		level.resize(n_level);
		const uint32_t th = n_level >= 512 ? n_threads : 1u;
		parallel_slices(n_level, th, [&](size_t b, size_t e, uint32_t) { std::vector<WideChild> ch; for (size_t i = b; i < e; ++i) emit_node(queue[lb + i], level[i], ch); });
		tri_base_of.resize(n_level);
		for (size_t i = 0; i < n_level; ++i)
		{
			Emit& E = level[i];
			bvh.slot_hist[E.n_children]++;
			E.node.w[4] = uint32_t(queue.size()); E.node.w[5] = uint32_t(tri_total);
			tri_base_of[i] = tri_total; tri_total += E.n_tris;
			if (tri_total > records.size()) throw std::runtime_error("fpt: internal wide-BVH error (more leaf triangles than triangles)");
			for (uint32_t c = 0; c < E.n_inner; ++c) queue.push_back(E.inner_ref[c]);
			bvh.n_inner_children += E.n_inner; bvh.n_leaf_children += E.n_leaf;
			bvh.nodes8.push_back(E.node);
		}
		parallel_slices(n_level, th, [&](size_t b, size_t e, uint32_t) { for (size_t i = b; i < e; ++i) write_records(level[i], records.data() + tri_base_of[i]); });
		lb = le;
	}
	records.resize(tri_total);
	bvh.tris8.swap(records);
	if (bvh.tris8.empty()) { BvhTriangle z; std::memset(&z, 0, sizeof(z)); bvh.tris8.push_back(z); }
	// upper bound of the traversal stack a ray can need (fpt_trace.hip pushes, per node step, at most the rest of the node group it came from -- when that
	// group has more than one inner child -- and at most one parked triangle group -- when the node has leaf children): bottom-up over the BFS order
	{
		std::vector<uint32_t> need(bvh.nodes8.size(), 0);
		for (size_t n = bvh.nodes8.size(); n-- > 0;)
		{
			const BvhNode8& N = bvh.nodes8[n];
			const uint32_t imask = N.w[3] >> 24; const uint32_t n_inner = uint32_t(__builtin_popcount(imask));
			const uint8_t* meta = reinterpret_cast<const uint8_t*>(N.w) + 24;
			bool has_leaf = false;
			for (int s = 0; s < 8; ++s) if (meta[s] && !((imask >> s) & 1u)) has_leaf = true;
			uint32_t below = 0;
			for (uint32_t c = 0; c < n_inner; ++c) below = std::max(below, need[size_t(N.w[4]) + c]);
			need[n] = (has_leaf ? 1u : 0u) + (n_inner ? (n_inner >= 2 ? 1u : 0u) + below : 0u);
		}
		bvh.stack_need = need.empty() ? 0u : need[0];
	}
	bvh.seconds_wide = float(now_seconds() - t0);
	if (std::getenv("FPT_BVH_TIMERS")) std::fprintf(stderr, "build_wide8: dp %.3f s, emission + stack bound %.3f s\n", t_dp - t0, now_seconds() - t_dp);
}

// The whole builder.  The kernel's stack pushes are unchecked, so the bound computed from the tree itself (rest-of-group + parked-triangle entries along
// the deepest path) must fit `stack_limit`: a degenerate input whose tree is too deep is built again without the optimisation (which may deepen a tree:
// where every position costs the same -- coincident triangles -- re-insertion strings the subtrees into a chain) and then with shallower SAH limits, down
// to the balanced object-median tree.  The caller checks out.stack_need.
// Refit (round 5): the vertices moved, the topology stays -- the triangle records are recomputed from the new positions, every wide node's box becomes the union of
// its children's EXACT boxes (leaves: their triangles' padded boxes, inner children: the box their node was given) bottom-up, and the children are quantised again
// on the node's new grid, outward, with build_wide8's arithmetic.  Slots, leaves and numbering are untouched, so the traversal-stack bound holds; what degrades with
// large motion is only the tree's quality (its boxes grow), never a result -- the intersector's answer does not depend on the tree.  Tens of milliseconds where a
// build takes most of a second: what RenderingContext::update_model uses when asked to (the reference rebuilds: src/renderer.cu:999-1017).
void refit_wide8(uint32_t tri_count, const int32_t* idx, uint32_t vertex_count, const float* vtx, HostBvh2& bvh)
{
	const double t0 = now_seconds();
	if (bvh.nodes8.empty() || tri_count == 0) return;
	if (bvh.tris8.size() != size_t(tri_count)) throw std::runtime_error("fpt: refit needs the geometry the tree was built over (triangle count differs)");
	const uint32_t n_threads = builder_threads();
	float scene_mag = 0.0f;
	for (uint32_t v = 0; v < vertex_count; ++v) for (int k = 0; k < 3; ++k) scene_mag = std::max(scene_mag, std::fabs(vtx[4 * size_t(v) + k]));
	// triangle records and their padded boxes (the padding rule of build_bvh2)
	std::vector<Box> tri_box(bvh.tris8.size());
	parallel_slices(bvh.tris8.size(), bvh.tris8.size() >= 65536 ? n_threads : 1u, [&](size_t b, size_t e, uint32_t) {
		for (size_t i = b; i < e; ++i)
		{
			BvhTriangle& r = bvh.tris8[i];
			const uint32_t tri = uint32_t(r.tri_id);
			if (tri >= tri_count) throw std::runtime_error("fpt: refit found a triangle record outside the mesh");
			const int32_t* ix = idx + 4 * size_t(tri);
			Box bx; bx.reset(); float m0 = 0.0f;
			for (int c = 0; c < 3; ++c)
			{
				if (ix[c] < 0 || uint32_t(ix[c]) >= vertex_count) throw std::runtime_error("fpt: vertex index out of range in refit");
				const float* p = vtx + 4 * size_t(ix[c]);
				bx.grow(p);
				for (int k = 0; k < 3; ++k) m0 = std::max(m0, std::fabs(p[k]));
			}
			const float pad = (m0 + scene_mag) * 4.0e-6f + 1.0e-30f;
			for (int k = 0; k < 3; ++k) { bx.lo[k] -= pad; bx.hi[k] += pad; }
			tri_box[i] = bx;
			const float* p0 = vtx + 4 * size_t(ix[0]); const float* p1 = vtx + 4 * size_t(ix[1]); const float* p2 = vtx + 4 * size_t(ix[2]);
			for (int k = 0; k < 3; ++k) { r.v0[k] = p0[k]; r.e1[k] = p1[k] - p0[k]; r.e2[k] = p2[k] - p0[k]; }
			r.mask = uint32_t(ix[3]); r.vpad = triangle_vpad(p0, p1, p2, scene_mag);
		} });
	bvh.scene_mag = scene_mag;
	// nodes bottom-up: children have larger indices than their parent (breadth-first numbering)
	std::vector<Box> node_box(bvh.nodes8.size());
	for (size_t n = bvh.nodes8.size(); n-- > 0;)
	{
		BvhNode8& node = bvh.nodes8[n];
		uint8_t* bytes = reinterpret_cast<uint8_t*>(node.w);
		const uint32_t imask = bytes[15], child_base = node.w[4], tri_base = node.w[5];
		Box cb[8]; bool used[8];
		Box nb; nb.reset();
		for (int sl = 0; sl < 8; ++sl)
		{
			const uint32_t m = bytes[24 + sl];
			used[sl] = m != 0;
			if (!m) continue;
			if ((imask >> sl) & 1u) cb[sl] = node_box[size_t(child_base) + uint32_t(__builtin_popcount(imask & ((1u << sl) - 1u)))];
			else
			{
				const uint32_t count = (m >> 5) == 1 ? 1u : ((m >> 5) == 3 ? 2u : 3u), first = tri_base + (m & 0x1Fu);
				cb[sl].reset();
				for (uint32_t t = 0; t < count; ++t) cb[sl].grow(tri_box[size_t(first) + t]);
			}
			nb.grow(cb[sl]);
		}
		bool any = false; for (int sl = 0; sl < 8; ++sl) any = any || used[sl];
		if (!any) { for (int k = 0; k < 3; ++k) { nb.lo[k] = 0.0f; nb.hi[k] = 0.0f; } }
		node_box[n] = nb;
		std::memcpy(&node.w[0], &nb.lo[0], 4); std::memcpy(&node.w[1], &nb.lo[1], 4); std::memcpy(&node.w[2], &nb.lo[2], 4);
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
		for (int sl = 0; sl < 8; ++sl)
		{
			uint8_t* qlo[3] = { bytes + 32 + sl, bytes + 40 + sl, bytes + 48 + sl };
			uint8_t* qhi[3] = { bytes + 56 + sl, bytes + 64 + sl, bytes + 72 + sl };
			if (!used[sl]) { for (int k = 0; k < 3; ++k) { *qlo[k] = 255; *qhi[k] = 0; } continue; }
			for (int k = 0; k < 3; ++k)
			{
				const double p = nb.lo[k], cell = std::ldexp(1.0, ex[k]);
				double lo = std::floor((double(cb[sl].lo[k]) - p) / cell); lo = lo < 0.0 ? 0.0 : (lo > 255.0 ? 255.0 : lo);
				while (lo > 0.0 && !(p + lo * cell <= double(cb[sl].lo[k]))) lo -= 1.0;
				double hi = std::ceil((double(cb[sl].hi[k]) - p) / cell); hi = hi < 0.0 ? 0.0 : (hi > 255.0 ? 255.0 : hi);
				while (hi < 255.0 && !(p + hi * cell >= double(cb[sl].hi[k]))) hi += 1.0;
				if (!(p + lo * cell <= double(cb[sl].lo[k])) || !(p + hi * cell >= double(cb[sl].hi[k]))) throw std::runtime_error("fpt: internal wide-BVH quantisation error (refit)");
				*qlo[k] = uint8_t(lo); *qhi[k] = uint8_t(hi);
			}
		}
	}
	bvh.seconds_refit = float(now_seconds() - t0);
}

void build_acceleration(uint32_t tri_count, const int32_t* idx, uint32_t vertex_count, const float* vtx, HostBvh2& out, uint32_t stack_limit)
{
	build_bvh2(tri_count, idx, vertex_count, vtx, out);
	optimize_bvh2(out);
	build_wide8(tri_count, idx, vtx, out);
	for (uint32_t sah_depth = 30; out.stack_need > stack_limit; sah_depth = sah_depth > 12 ? 12 : (sah_depth >= 6 ? sah_depth - 6 : 0))
	{
		const float t_opt = out.seconds_opt;
		build_bvh2(tri_count, idx, vertex_count, vtx, out, sah_depth);
		out.opt_iterations = 0; out.opt_cost_before = out.opt_cost_after = 0.0f; out.seconds_opt = t_opt;      // (the time was spent; its tree was not kept)
		build_wide8(tri_count, idx, vtx, out);
		if (sah_depth == 0) break;
	}
}

} // namespace fpt
