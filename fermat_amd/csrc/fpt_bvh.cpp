// fpt_bvh.cpp — host-side builder of the 8-wide compressed BVH: a binned-SAH BVH2 down to single triangles (multi-threaded; scenes are static
// across passes, SURVEY §2.2 "cugar/bvh"), an insertion-based optimisation of it, then the SAH-optimal 8-wide collapse.  Topology is irrelevant to results
// (closest-t / lowest-id rule + the intersector's box clause), so this builder is free to differ from the oracle's CUGAR full-sweep restatement.
#include "fpt_bvh.h"
#include "fpt_cw8_slots.h"
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
#include <mutex>
#include <condition_variable>
#include <system_error>
#include <emmintrin.h>          // SSE2: the host of an MI355X is x86-64

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

// A triangle reference: the triangle and its (padded, see build_bvh2) bounds; 32 bytes, partitioned in place
struct alignas(32) Ref { float lo[3]; uint32_t tri; float hi[3]; uint32_t pad; };
static_assert(sizeof(Ref) == 32, "Ref must be 32 bytes");
inline void grow(Box& b, const Ref& r) { for (int k = 0; k < 3; ++k) { b.lo[k] = std::min(b.lo[k], r.lo[k]); b.hi[k] = std::max(b.hi[k], r.hi[k]); } }
inline Box box_of(const Ref& r) { Box b; for (int k = 0; k < 3; ++k) { b.lo[k] = r.lo[k]; b.hi[k] = r.hi[k]; } return b; }

// A build's helper threads: created once per build and woken per parallel loop (the top of the tree alone runs some 250 parallel loops; creating 15 threads for each
// of them cost more than the loops).  The pool belongs to the thread that built it (thread_local): the workers themselves, and host threads building other scenes,
// do not see it.
class SlicePool
{
public:
	explicit SlicePool(uint32_t workers)
	{
		try { for (uint32_t w = 0; w < workers; ++w) th.emplace_back([this, w]() { work(w); }); }
		catch (const std::system_error&) {}          // a thread that cannot be created (cgroup pid limit, EAGAIN) is not an error: its slices run on the caller
	}
	~SlicePool()
	{
		{ std::lock_guard<std::mutex> g(m); stop = true; ++generation; }
		cv_start.notify_all();
		for (std::thread& t : th) t.join();
	}
	uint32_t workers() const { return uint32_t(th.size()); }
	// job(ctx, t) for t = 1 .. slices - 1 on the workers (t - 1 = worker; slices beyond the pool are left to the caller, who is told from where)
	uint32_t start(void (*fn)(void*, uint32_t), void* c, uint32_t slices)
	{
		const uint32_t mine = std::min(slices - 1u, workers());
		{ std::lock_guard<std::mutex> g(m); job = fn; ctx = c; job_slices = mine + 1u; pending = mine; ++generation; }
		cv_start.notify_all();
		return mine + 1u;
	}
	void wait()
	{
		std::unique_lock<std::mutex> g(m);
		cv_done.wait(g, [this]() { return pending == 0; });
	}
private:
	void work(uint32_t w)
	{
		uint64_t seen = 0;
		for (;;)
		{
			void (*fn)(void*, uint32_t); void* c; uint32_t ns;
			{
				std::unique_lock<std::mutex> g(m);
				cv_start.wait(g, [&]() { return generation != seen; });
				seen = generation;
				if (stop) return;
				fn = job; c = ctx; ns = job_slices;
			}
			if (w + 1u < ns)
			{
				fn(c, w + 1u);
				bool last;
				{ std::lock_guard<std::mutex> g(m); last = --pending == 0; }
				if (last) cv_done.notify_one();
			}
		}
	}
	std::vector<std::thread> th;
	std::mutex m; std::condition_variable cv_start, cv_done;
	uint64_t generation = 0; uint32_t pending = 0, job_slices = 0; bool stop = false;
	void (*job)(void*, uint32_t) = nullptr; void* ctx = nullptr;
};
thread_local SlicePool* tl_pool = nullptr;

// runs f(begin, end, slice) over contiguous slices of [0, n) on `slices` threads; the slices are a function of n and `slices` only
template <class F> void parallel_slices(size_t n, uint32_t slices, F f)
{
	if (slices <= 1) { f(size_t(0), n, 0u); return; }
	std::vector<std::exception_ptr> error(slices);          // an exception must not leave a thread (std::terminate): it is rethrown on the caller's
	struct Ctx { F* f; size_t n; uint32_t slices; std::exception_ptr* error; } ctx = { &f, n, slices, error.data() };
	auto run = [](void* p, uint32_t t) {
		Ctx& c = *static_cast<Ctx*>(p);
		try { (*c.f)(c.n * t / c.slices, c.n * (t + 1) / c.slices, t); } catch (...) { c.error[t] = std::current_exception(); } };
	if (tl_pool)
	{
		const uint32_t next = tl_pool->start(run, &ctx, slices);
		run(&ctx, 0u);
		for (uint32_t t = next; t < slices; ++t) run(&ctx, t);      // slices beyond the pool's workers
		tl_pool->wait();
	}
	else
	{
		std::vector<std::thread> pool;
		uint32_t started = 1;
		try { for (uint32_t t = 1; t < slices; ++t) { pool.emplace_back(run, &ctx, t); started = t + 1; } } catch (const std::system_error&) {}
		run(&ctx, 0u);
		for (uint32_t t = started; t < slices; ++t) run(&ctx, t);      // the slices of threads that could not be created (cgroup pid limit, EAGAIN): done here
		for (std::thread& t : pool) t.join();
	}
	for (const std::exception_ptr& e : error) if (e) std::rethrow_exception(e);
}

// Binned SAH (32 centroid bins per axis), one triangle per leaf.  Below depth 30 (strongly non-uniform scales peel off one primitive per level)
// and where all centroids coincide the split is the object median of the widest axis, so the depth is bounded by 30 + log2(n) for any input.
//
// Round 5 (second half): no allocation per node.  The references are ONE array partitioned in place; a subtree over m references has exactly m - 1 nodes, so
// the node of a range is known before it is built -- a node's left subtree starts right behind it, its right subtree nl nodes behind it: the array comes out in
// pre-order, the leaf order IS the reference order, and subtrees built by other threads need no stitching.  The big ranges at the top are binned and partitioned
// by all threads (stable, through a second array; the children then live there and the two arrays swap roles); ranges of at most `grain` references become tasks.
// Every decision is a function of the SET of references in a range (bin bounds, counts, a total order for the median), so the tree does not depend on the
// number of threads or on the order in which a partition leaves the references.
// two SSE registers: a box; lane 3 is kept at zero (a Ref carries integers there).  min / max are spelled so that ties and signs of zero fall as std::min / std::max
// of (accumulator, value) would: the trees of the scalar builder of rounds 2-5 are reproduced bit for bit.
struct VBox
{
	__m128 lo, hi;
	void reset() { lo = _mm_setr_ps(3.0e38f, 3.0e38f, 3.0e38f, 0.0f); hi = _mm_setr_ps(-3.0e38f, -3.0e38f, -3.0e38f, 0.0f); }
	void grow(__m128 l, __m128 h) { lo = _mm_min_ps(l, lo); hi = _mm_max_ps(h, hi); }
	void grow(const VBox& o) { grow(o.lo, o.hi); }
	void grow_point(__m128 c) { grow(c, c); }
	Box box() const { alignas(16) float l[4], h[4]; _mm_store_ps(l, lo); _mm_store_ps(h, hi); Box b; for (int k = 0; k < 3; ++k) { b.lo[k] = l[k]; b.hi[k] = h[k]; } return b; }
	double half_area() const          // = Box::half_area of the same box, bit for bit (differences and products in double, no contraction), without the trip through memory
	{
		const __m128d e01 = _mm_sub_pd(_mm_cvtps_pd(hi), _mm_cvtps_pd(lo));
		const __m128d e2 = _mm_sub_pd(_mm_cvtps_pd(_mm_movehl_ps(hi, hi)), _mm_cvtps_pd(_mm_movehl_ps(lo, lo)));
		const double ex = _mm_cvtsd_f64(e01), ey = _mm_cvtsd_f64(_mm_unpackhi_pd(e01, e01)), ez = _mm_cvtsd_f64(e2);
		return (ex < 0 || ey < 0 || ez < 0) ? 0.0 : ex * ey + ez * (ex + ey);
	}
	bool same_bits(const VBox& o) const
	{
		return _mm_movemask_epi8(_mm_and_si128(_mm_cmpeq_epi32(_mm_castps_si128(lo), _mm_castps_si128(o.lo)), _mm_cmpeq_epi32(_mm_castps_si128(hi), _mm_castps_si128(o.hi)))) == 0xFFFF;
	}
	static VBox from(const float* l, const float* h) { VBox b; b.lo = _mm_setr_ps(l[0], l[1], l[2], 0.0f); b.hi = _mm_setr_ps(h[0], h[1], h[2], 0.0f); return b; }
	void store(float* l, float* h) const { alignas(16) float a[4], c[4]; _mm_store_ps(a, lo); _mm_store_ps(c, hi); for (int k = 0; k < 3; ++k) { l[k] = a[k]; h[k] = c[k]; } }
};
inline __m128 lane_mask() { return _mm_castsi128_ps(_mm_setr_epi32(-1, -1, -1, 0)); }
inline __m128 ref_lo(const Ref& r) { return _mm_and_ps(_mm_load_ps(r.lo), lane_mask()); }
inline __m128 ref_hi(const Ref& r) { return _mm_and_ps(_mm_load_ps(r.hi), lane_mask()); }
inline __m128 ref_centre(__m128 l, __m128 h) { return _mm_mul_ps(_mm_set1_ps(0.5f), _mm_add_ps(l, h)); }

struct Builder
{
	static const int kSmall = 32;       // ranges of at most this many references find their split without bins (same decisions)
	static const int kBins = 32;        // 16 ... 512 measured with the traversal model: the tree's cost moves by +-3 % in either direction, and after the
	                                    // re-insertion pass the trees of 32 and 128 bins cost the same
	BvhNode* nodes = nullptr;       // tri_count - 1 of them
	uint32_t* prims = nullptr;      // triangle ids in leaf order = reference order
	size_t grain = 0;               // top phase: ranges of at most this many references become tasks (0: none)
	uint32_t sah_depth = 30;        // SAH splits down to this depth, object medians below
	uint32_t threads = 1;           // top phase: threads for the big ranges
	double root_area = 1.0;
	struct Task { Ref* buf; Ref* other; uint32_t begin, end, node, depth; VBox box, cb; uint32_t max_depth; double cost; };
	std::vector<Task> tasks;
	struct Stats { uint32_t max_depth = 0; double cost = 0.0; };

	struct Bins { VBox bb[3][kBins]; uint32_t cnt[3][kBins]; };

	// bounds of the boxes and of their centres
	static void bounds(const Ref* r0, uint32_t n, uint32_t slices, VBox& box, VBox& cb)
	{
		VBox part[2 * 64];
		parallel_slices(n, slices, [&](size_t sb, size_t se, uint32_t t) {
			VBox bx, cx; bx.reset(); cx.reset();
			for (size_t i = sb; i < se; ++i) { const __m128 l = ref_lo(r0[i]), h = ref_hi(r0[i]); bx.grow(l, h); cx.grow_point(ref_centre(l, h)); }
			part[2 * size_t(t)] = bx; part[2 * size_t(t) + 1] = cx; });
		box.reset(); cb.reset();
		for (uint32_t t = 0; t < slices; ++t) { box.grow(part[2 * size_t(t)]); cb.grow(part[2 * size_t(t) + 1]); }
	}

	// builds the subtree over a[begin, end) (b: the second array, same indices, free to use) whose root is node `node`; `box` / `cb`: the bounds of the range's
	// boxes and of their centres (the caller has them from its partition pass); returns the child reference
	int32_t build(Ref* a, Ref* b, uint32_t begin, uint32_t end, uint32_t node, uint32_t depth, const VBox& box, const VBox& cb, bool top, Stats& st)
	{
		const uint32_t n = end - begin;
		const uint32_t slices = (top && threads > 1 && n >= 32768u) ? std::min(threads, n / 16384u) : 1u;
		Ref* const r0 = a + begin;
		if (top && n <= grain && n > 1)
		{
			tasks.push_back(Task{ a, b, begin, end, node, depth, box, cb, 0u, 0.0 });
			return int32_t(node);
		}
		st.max_depth = std::max(st.max_depth, depth);
		if (n <= 1)
		{
			prims[begin] = r0[0].tri;
			st.cost += box.half_area() / root_area;
			return ~int32_t((begin << 3) | 1u);
		}
		alignas(16) float clo[4], chi[4];
		_mm_store_ps(clo, cb.lo); _mm_store_ps(chi, cb.hi);
		double best = 1.0e300; int best_axis = -1; int best_bin = 0;
		std::vector<uint32_t> slice_all;          // top phase: every slice's bin counts (3 x kBins each), kept for the partition's offsets
		if (depth <= sah_depth && n <= uint32_t(kSmall))
		{
			// few references: the same candidates, costs and tie-breaks as the binned sweep below, found from the references instead of from 3 x 32 bins
			// (most inner nodes of a tree hold a handful of triangles; resetting and sweeping 96 bins for each of them was three quarters of the build).
			// Per axis: bin numbers, references ordered by bin, the boxes of the suffixes; a candidate wherever the bin number changes -- "everything in
			// a lower bin goes left" -- which is the binned sweep's candidate k = that bin number; its candidates in between repeat a cost and never win.
			for (int x = 0; x < 3; ++x)
			{
				const float ext = chi[x] - clo[x];
				if (!(ext > 0.0f)) continue;
				const float scale = float(kBins) / ext;
				uint8_t key[kSmall], ord[kSmall];
				for (uint32_t i = 0; i < n; ++i)
				{
					const Ref& r = r0[i];
					int k = int((0.5f * (r.lo[x] + r.hi[x]) - clo[x]) * scale); k = k < 0 ? 0 : (k >= kBins ? kBins - 1 : k);
					key[i] = uint8_t(k);
					uint32_t j = i;
					while (j > 0 && key[ord[j - 1]] > k) { ord[j] = ord[j - 1]; --j; }
					ord[j] = uint8_t(i);
				}
				VBox rbox[kSmall];
				VBox acc; acc.reset();
				for (uint32_t j = n - 1; j > 0; --j) { acc.grow(ref_lo(r0[ord[j]]), ref_hi(r0[ord[j]])); rbox[j] = acc; }
				acc.reset();
				for (uint32_t j = 1; j < n; ++j)
				{
					acc.grow(ref_lo(r0[ord[j - 1]]), ref_hi(r0[ord[j - 1]]));
					if (key[ord[j]] == key[ord[j - 1]]) continue;
					const double sc = acc.half_area() * double(j) + rbox[j].half_area() * double(n - j);
					if (sc < best) { best = sc; best_axis = x; best_bin = key[ord[j]]; }
				}
			}
		}
		else if (depth <= sah_depth)
		{
			alignas(16) float scale[4] = { 0, 0, 0, 0 }; bool live[3];
			for (int x = 0; x < 3; ++x) { const float ext = chi[x] - clo[x]; live[x] = ext > 0.0f; scale[x] = live[x] ? float(kBins) / ext : 0.0f; }
			const __m128 vclo = cb.lo, vscale = _mm_load_ps(scale);
			std::vector<Bins> extra(slices - 1);
			Bins B0;
			parallel_slices(n, slices, [&](size_t sb, size_t se, uint32_t t) {
				Bins& B = t == 0 ? B0 : extra[t - 1];
				for (int x = 0; x < 3; ++x) for (int k = 0; k < kBins; ++k) { B.bb[x][k].reset(); B.cnt[x][k] = 0; }
				for (size_t i = sb; i < se; ++i)
				{
					const __m128 l = ref_lo(r0[i]), h = ref_hi(r0[i]);
					alignas(16) int32_t k[4];
					_mm_store_si128(reinterpret_cast<__m128i*>(k), _mm_cvttps_epi32(_mm_mul_ps(_mm_sub_ps(ref_centre(l, h), vclo), vscale)));
					for (int x = 0; x < 3; ++x)
					{
						if (!live[x]) continue;
						const int kk = k[x] < 0 ? 0 : (k[x] >= kBins ? kBins - 1 : k[x]);
						B.bb[x][kk].grow(l, h); B.cnt[x][kk]++;
					}
				} });
			Bins& B = B0;
			if (slices > 1) { slice_all.resize(size_t(slices) * 3 * kBins); for (uint32_t t = 0; t < slices; ++t) std::memcpy(&slice_all[size_t(t) * 3 * kBins], t == 0 ? &B0.cnt[0][0] : &extra[t - 1].cnt[0][0], sizeof(uint32_t) * 3 * kBins); }
			for (uint32_t t = 1; t < slices; ++t)
				for (int x = 0; x < 3; ++x) for (int k = 0; k < kBins; ++k) { B.bb[x][k].grow(extra[t - 1].bb[x][k]); B.cnt[x][k] += extra[t - 1].cnt[x][k]; }
			for (int x = 0; x < 3; ++x)
			{
				if (!live[x]) continue;
				VBox rbox[kBins]; uint32_t rcnt[kBins];
				VBox acc; acc.reset(); uint32_t c = 0;
				for (int k = kBins - 1; k > 0; --k) { acc.grow(B.bb[x][k]); c += B.cnt[x][k]; rbox[k] = acc; rcnt[k] = c; }
				acc.reset(); c = 0;
				for (int k = 1; k < kBins; ++k)
				{
					if (B.cnt[x][k - 1] == 0) continue;          // the same left set as the candidate before: the same cost, which never wins
					acc.grow(B.bb[x][k - 1]); c += B.cnt[x][k - 1];
					if (rcnt[k] == 0) continue;
					const double sc = acc.half_area() * double(c) + rbox[k].half_area() * double(rcnt[k]);
					if (sc < best) { best = sc; best_axis = x; best_bin = k; }
				}
			}
		}
		std::vector<uint32_t> slice_cnt;          // the chosen axis' bin counts per slice
		if (best_axis >= 0 && slices > 1) { slice_cnt.resize(size_t(slices) * kBins); for (uint32_t t = 0; t < slices; ++t) std::memcpy(&slice_cnt[size_t(t) * kBins], &slice_all[(size_t(t) * 3 + size_t(best_axis)) * kBins], sizeof(uint32_t) * kBins); }
		uint32_t nl = 0;
		bool moved = false;          // the children live in b
		VBox lb, lc, rb, rc;         // the children's bounds
		if (best_axis >= 0)
		{
			const float scl = float(kBins) / (chi[best_axis] - clo[best_axis]);
			const float lo = clo[best_axis]; const int x = best_axis;
			auto goes_left = [&](const Ref& r) {
				int k = int((0.5f * (r.lo[x] + r.hi[x]) - lo) * scl); k = k < 0 ? 0 : (k >= kBins ? kBins - 1 : k);
				return k < best_bin; };
			if (slices == 1)
			{
				// in place: the two ends walk towards each other; every reference is classified once and counted into its side's bounds
				lb.reset(); lc.reset(); rb.reset(); rc.reset();
				auto to_left = [&](const Ref& r) { const __m128 l = ref_lo(r), h = ref_hi(r); lb.grow(l, h); lc.grow_point(ref_centre(l, h)); };
				auto to_right = [&](const Ref& r) { const __m128 l = ref_lo(r), h = ref_hi(r); rb.grow(l, h); rc.grow_point(ref_centre(l, h)); };
				uint32_t i = 0, j = n;
				for (;;)
				{
					while (i < j && goes_left(r0[i])) { to_left(r0[i]); ++i; }
					while (i < j && !goes_left(r0[j - 1])) { to_right(r0[j - 1]); --j; }
					if (i >= j) break;
					std::swap(r0[i], r0[j - 1]);
					to_left(r0[i]); to_right(r0[j - 1]); ++i; --j;
				}
				nl = i;
			}
			else
			{
				// through the second array, slice by slice: counts, offsets, scatter
				uint32_t lefts[64];          // per slice: what its own bins counted below the split (the binning pass ran over the same slices)
				for (uint32_t t = 0; t < slices; ++t) { uint32_t c = 0; for (int k = 0; k < best_bin; ++k) c += slice_cnt[size_t(t) * kBins + size_t(k)]; lefts[t] = c; nl += c; }
				if (nl != 0 && nl != n)
				{
					uint32_t lofs[64], rofs[64]; uint32_t l = 0, r = nl;
					for (uint32_t t = 0; t < slices; ++t) { lofs[t] = l; rofs[t] = r; l += lefts[t]; r += uint32_t(size_t(n) * (t + 1) / slices - size_t(n) * t / slices) - lefts[t]; }
					Ref* const d0 = b + begin;
					VBox part[4 * 64];
					parallel_slices(n, slices, [&](size_t sb, size_t se, uint32_t t) {
						uint32_t l2 = lofs[t], r2 = rofs[t];
						VBox plb, plc, prb, prc; plb.reset(); plc.reset(); prb.reset(); prc.reset();
						for (size_t i = sb; i < se; ++i)
						{
							const Ref& rf = r0[i];
							const __m128 lo4 = ref_lo(rf), hi4 = ref_hi(rf);
							if (goes_left(rf)) { d0[l2++] = rf; plb.grow(lo4, hi4); plc.grow_point(ref_centre(lo4, hi4)); }
							else               { d0[r2++] = rf; prb.grow(lo4, hi4); prc.grow_point(ref_centre(lo4, hi4)); }
						}
						part[4 * size_t(t)] = plb; part[4 * size_t(t) + 1] = plc; part[4 * size_t(t) + 2] = prb; part[4 * size_t(t) + 3] = prc; });
					lb.reset(); lc.reset(); rb.reset(); rc.reset();
					for (uint32_t t = 0; t < slices; ++t) { lb.grow(part[4 * size_t(t)]); lc.grow(part[4 * size_t(t) + 1]); rb.grow(part[4 * size_t(t) + 2]); rc.grow(part[4 * size_t(t) + 3]); }
					moved = true;
				}
			}
			if (nl == 0 || nl == n) best_axis = -1;
		}
		if (best_axis < 0)
		{
			int x = 0;
			for (int k = 1; k < 3; ++k) if (chi[k] - clo[k] > chi[x] - clo[x]) x = k;
			nl = n / 2;
			std::nth_element(r0, r0 + nl, r0 + n, [&](const Ref& p, const Ref& q) {
				const float cp = p.lo[x] + p.hi[x], cq = q.lo[x] + q.hi[x];
				return cp < cq || (cp == cq && p.tri < q.tri); });
			bounds(r0, nl, 1u, lb, lc); bounds(r0 + nl, n - nl, 1u, rb, rc);
		}
		Ref* const ca = moved ? b : a; Ref* const cbuf = moved ? a : b;
		const int32_t c0 = build(ca, cbuf, begin, begin + nl, node + 1u, depth + 1, lb, lc, top, st);
		const int32_t c1 = build(ca, cbuf, begin + nl, end, node + nl, depth + 1, rb, rc, top, st);
		BvhNode& nd = nodes[node];
		const Box b0 = lb.box(), b1 = rb.box();
		for (int k = 0; k < 3; ++k) { nd.lo0[k] = b0.lo[k]; nd.hi0[k] = b0.hi[k]; nd.lo1[k] = b1.lo[k]; nd.hi1[k] = b1.hi[k]; }
		nd.child0 = c0; nd.child1 = c1; nd.pad0 = nd.pad1 = 0;
		st.cost += box.half_area() / root_area;
		return int32_t(node);
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

// installs a pool for the calling thread for the duration of a build step (nested scopes reuse the outer one)
struct PoolScope
{
	std::unique_ptr<SlicePool> own;
	PoolScope() { if (!tl_pool) { const uint32_t n = builder_threads(); if (n > 1) { own.reset(new SlicePool(n - 1)); tl_pool = own.get(); } } }
	~PoolScope() { if (own) tl_pool = nullptr; }
	PoolScope(const PoolScope&) = delete; PoolScope& operator=(const PoolScope&) = delete;
};

// the constant part of the tolerance of fpt-MT's box clause for one triangle (fpt_trace.hip intersect_record, oracle/o_bvh.h intersect_tri): 5e-7 (|triangle|max + |scene|max)
float triangle_vpad(const float* p0, const float* p1, const float* p2, float scene_mag)
{
	float m0 = 0.0f;
	for (int k = 0; k < 3; ++k) m0 = std::max(m0, std::max(std::fabs(p0[k]), std::max(std::fabs(p1[k]), std::fabs(p2[k]))));
	return (m0 + scene_mag) * 5.0e-7f;
}

double now_seconds() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

} // namespace

void build_bvh2(uint32_t tri_count, const int32_t* idx, uint32_t vertex_count, const float* vtx, HostBvh2& out, uint32_t sah_depth)
{
	PoolScope pool_scope;
	const double t0 = now_seconds();
	out.nodes.clear(); out.prims.clear(); out.max_depth = 0; out.sah_cost = 0.0f;
	if (tri_count >= (1u << 28)) throw std::runtime_error("fpt: too many triangles for the leaf reference encoding");
	const uint32_t n_threads = builder_threads();
	out.threads = n_threads;
	const uint32_t wide = tri_count >= 65536u ? n_threads : 1u;
	// scene magnitude for the conservative padding (rounding in the slab test must never cull a hit that the fpt-MT intersector accepts)
	float scene_mag = 0.0f;
	{
		float part[64] = { 0.0f };
		parallel_slices(vertex_count, vertex_count >= 65536u ? n_threads : 1u, [&](size_t b, size_t e, uint32_t t) {
			float m = 0.0f;
			for (size_t v = b; v < e; ++v) for (int k = 0; k < 3; ++k) m = std::max(m, std::fabs(vtx[4 * v + k]));
			part[t] = m; });
		for (float m : part) scene_mag = std::max(scene_mag, m);
	}
	out.scene_mag = scene_mag;
	if (tri_count == 0)
	{
		// an empty scene still gets one node whose children are empty leaves, so kernels need no special case
		BvhNode n; std::memset(&n, 0, sizeof(n));
		for (int k = 0; k < 3; ++k) { n.lo0[k] = n.lo1[k] = 3.0e38f; n.hi0[k] = n.hi1[k] = -3.0e38f; }
		n.child0 = ~0; n.child1 = ~0;
		out.nodes.push_back(n);
		return;
	}
	NoInitVector<Ref> refs(tri_count), second(tri_count >= 32768u && n_threads > 1 ? tri_count : 0u);
	Box root_bounds; root_bounds.reset();
	{
		Box part[64];
		parallel_slices(tri_count, wide, [&](size_t tb, size_t te, uint32_t s) {
			Box all; all.reset();
			for (size_t t = tb; t < te; ++t)
			{
				Box b; b.reset();
				float m0 = 0.0f;
				for (int c = 0; c < 3; ++c)
				{
					const int32_t vi = idx[4 * t + c];
					if (vi < 0 || uint32_t(vi) >= vertex_count) throw std::runtime_error("fpt: vertex index out of range in create_geometry");
					const float* p = vtx + 4 * size_t(vi);
					b.grow(p);
					for (int k = 0; k < 3; ++k) m0 = std::max(m0, std::fabs(p[k]));
				}
				const float pad = (m0 + scene_mag) * 2.0e-6f + 1.0e-30f;          // four times the constant tolerance of fpt-MT's box clause (triangle_vpad): an accepted hit lies inside with margin
				Ref& r = refs[t];
				for (int k = 0; k < 3; ++k) { r.lo[k] = b.lo[k] - pad; r.hi[k] = b.hi[k] + pad; }
				r.tri = uint32_t(t); r.pad = 0;
				grow(all, r);
			}
			part[s] = all; });
		for (uint32_t s = 0; s < wide; ++s) root_bounds.grow(part[s]);
	}
	const double root_area = std::max(root_bounds.half_area(), 1.0e-300);
	const double t_refs = now_seconds();
	out.nodes.resize(std::max<size_t>(1, size_t(tri_count) - 1));
	out.prims.resize(tri_count);
	if (out.nodes.size() >= size_t(0x40000000)) throw std::runtime_error("fpt: too many BVH nodes");
	// top phase: the big ranges on all threads, until the subtrees hold at most `grain` references; those become tasks
	Builder top;
	top.nodes = out.nodes.data(); top.prims = out.prims.data();
	top.root_area = root_area; top.threads = second.empty() ? 1u : n_threads; top.sah_depth = sah_depth;
	top.grain = (n_threads > 1 && tri_count >= 20000u) ? std::max<size_t>(4096, size_t(tri_count) / (size_t(n_threads) * 8)) : 0;
	VBox root_vbox, root_cb;
	Builder::bounds(refs.data(), tri_count, wide, root_vbox, root_cb);
	const Box root_box = root_vbox.box();
	Builder::Stats top_stats;
	const int32_t root = top.build(refs.data(), second.empty() ? refs.data() : second.data(), 0u, tri_count, 0u, 1u, root_vbox, root_cb, true, top_stats);
	double cost = top_stats.cost; uint32_t max_depth = top_stats.max_depth;
	const double t_top = now_seconds();
	if (!top.tasks.empty())
	{
		std::atomic<size_t> next(0);
		std::atomic<bool> failed(false);
		auto worker = [&]() {
			for (;;)
			{
				const size_t i = next.fetch_add(1);
				if (i >= top.tasks.size() || failed.load()) return;
				try
				{
					Builder::Task& T = top.tasks[i];
					Builder::Stats st;
					top.build(T.buf, T.other, T.begin, T.end, T.node, T.depth, T.box, T.cb, false, st);
					T.max_depth = st.max_depth; T.cost = st.cost;
				}
				catch (...) { failed.store(true); }
			}
		};
		parallel_slices(n_threads, n_threads, [&](size_t, size_t, uint32_t) { worker(); });          // (threads that could not be created: the others share their tasks)
		if (failed.load()) throw std::runtime_error("fpt: BVH builder worker failed (out of memory?)");
		for (const Builder::Task& T : top.tasks) { cost += T.cost; max_depth = std::max(max_depth, T.max_depth); }          // in task order: a function of the tree and the grain
	}
	if (root < 0)
	{
		// a single triangle: wrap the leaf in a node with an empty sibling
		BvhNode n; std::memset(&n, 0, sizeof(n));
		for (int k = 0; k < 3; ++k) { n.lo0[k] = root_box.lo[k]; n.hi0[k] = root_box.hi[k]; n.lo1[k] = 3.0e38f; n.hi1[k] = -3.0e38f; }
		n.child0 = root; n.child1 = ~0;
		out.nodes[0] = n;
	}
	else if (root != 0) throw std::runtime_error("fpt: internal BVH builder error (root is not node 0)");
	out.max_depth = max_depth;
	out.sah_cost = float(cost);
	out.seconds_bvh2 = float(now_seconds() - t0);
	if (std::getenv("FPT_BVH_TIMERS")) std::fprintf(stderr, "build_bvh2: references %.3f s, top phase %.3f (%zu tasks), tasks %.3f\n", t_refs - t0, t_top - t_refs, top.tasks.size(), now_seconds() - t_top);
}

// ---- insertion-based optimisation of the binary tree -------------------------------------------------------------------------------
// Bittner, Hapala, Havran: Fast Insertion-Based Optimization of Bounding Volume Hierarchies (CGF 2013).  The top-down SAH build is greedy; this pass
// repeatedly takes the inner nodes that bound their children worst (large, with small or very unequal children), removes them, and re-inserts their two
// subtrees where they increase the tree's surface area least (branch and bound over the tree with the induced cost of the ancestors as the bound).
// Topology only: leaves keep their (padded) boxes, results of the intersector do not depend on it.
namespace {

struct alignas(64) ONode          // one cache line
{
	VBox box; double area;
	int32_t parent, child[2];      // leaf: child[0] = -1
	uint32_t tri;
};
static_assert(sizeof(ONode) == 64, "ONode must be one 64-byte line");

struct Optimizer
{
	NoInitVector<ONode> n;
	int32_t root = 0;
	uint32_t n_inner = 0;
	struct Item { double induced; int32_t node; bool operator<(const Item& o) const { return induced > o.induced; } };      // min-heap on the induced cost
	std::vector<Item> heap;
	// work bound: a search normally opens a few hundred nodes, but where everything overlaps everything (coincident geometry) the bound prunes nothing;
	// once the budget is spent a search settles for the best position seen so far (any position is valid) and the pass ends with the batch
	uint64_t visits = 0, budget = ~0ull;

	static VBox merged(const VBox& a, const VBox& b) { VBox r = a; r.grow(b); return r; }
	bool is_leaf(int32_t i) const { return n[size_t(i)].child[0] < 0; }

	// recompute boxes from node i up to the root (stops when nothing changes)
	void refit(int32_t i)
	{
		while (i >= 0)
		{
			ONode& X = n[size_t(i)];
			const VBox b = merged(n[size_t(X.child[0])].box, n[size_t(X.child[1])].box);
			if (b.same_bits(X.box)) break;
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
		const VBox bx = n[size_t(x)].box; const double ax = n[size_t(x)].area;
		heap.clear();
		heap.push_back(Item{ 0.0, -1 });          // an item stands for the two children of `node` (-1: for the root), which share their induced cost: half the heap traffic
		double best = 1.0e300; int32_t best_node = root;
		if (hint >= 0)
		{
			double total = merged(n[size_t(hint)].box, bx).half_area();
			for (int32_t a = n[size_t(hint)].parent; a >= 0; a = n[size_t(a)].parent) total += merged(n[size_t(a)].box, bx).half_area() - n[size_t(a)].area;
			best = total; best_node = hint;
		}
		bool done = false;
		while (!heap.empty() && !done)
		{
			std::pop_heap(heap.begin(), heap.end()); const Item it = heap.back(); heap.pop_back();
			const int32_t pair[2] = { it.node < 0 ? root : n[size_t(it.node)].child[0], it.node < 0 ? -1 : n[size_t(it.node)].child[1] };
			for (int c = 0; c < 2 && pair[c] >= 0; ++c)
			{
				if (it.induced + ax >= best || ++visits > budget) { done = true; break; }          // (the heap's smallest: nothing left can do better)
				const ONode& X = n[size_t(pair[c])];
				const double direct = merged(X.box, bx).half_area();
				const double total = it.induced + direct;
				if (total < best) { best = total; best_node = pair[c]; }
				const double below = total - X.area;          // induced cost for anything under X
				if (X.child[0] >= 0 && below + ax < best)
				{
					_mm_prefetch(reinterpret_cast<const char*>(&n[size_t(X.child[0])]), _MM_HINT_T0);          // the search is bound by the latency of these lines (233 MB of nodes)
					_mm_prefetch(reinterpret_cast<const char*>(&n[size_t(X.child[1])]), _MM_HINT_T0);
					heap.push_back(Item{ below, pair[c] }); std::push_heap(heap.begin(), heap.end());
				}
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
		// partial sums over 64 fixed chunks, added in order: the value (which decides when the pass stops) does not depend on the number of threads
		double part[64];
		const size_t ni = n_inner;
		parallel_slices(64, std::min(64u, ni >= 65536 ? std::max(1u, threads) : 1u), [&](size_t cb, size_t ce, uint32_t) {
			for (size_t ch = cb; ch < ce; ++ch)
			{
				double c = 0.0;
				for (size_t i = ni * ch / 64; i < ni * (ch + 1) / 64; ++i) c += n[i].area;
				part[ch] = c;
			} });
		double c = 0.0;
		for (double x : part) c += x;
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
		// the `count` worst of all = the `count` worst of the slices' `count` worst each (measure, node number): a total order, whatever the slices
		parallel_slices(size_t(th), th, [&](size_t b, size_t e, uint32_t) {
			for (size_t t = b; t < e; ++t)
				if (part[t].size() > count) { std::nth_element(part[t].begin(), part[t].begin() + (count - 1), part[t].end()); part[t].resize(count); } });
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
			insert(L, N, S);
			insert(R, P, S);
		}
		t_apply += now_seconds() - tt1;
	}
};

} // namespace

void optimize_bvh2(HostBvh2& bvh, uint32_t max_iterations, double batch_fraction)
{
	PoolScope pool_scope;
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
	// every node is written by its parent (box, area, parent) and by itself (children): no two threads write the same field
	parallel_slices(ni, ni >= 65536 ? O.threads : 1u, [&](size_t ib, size_t ie, uint32_t) {
		for (size_t i = ib; i < ie; ++i)
		{
			const BvhNode& N = bvh.nodes[i];
			const int32_t ref[2] = { N.child0, N.child1 };
			const int32_t me = id_of[i];
			const VBox cb[2] = { VBox::from(N.lo0, N.hi0), VBox::from(N.lo1, N.hi1) };
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
				ONode& C = O.n[size_t(id)];
				C.box = cb[c]; C.area = cb[c].half_area(); C.parent = me;
				if (ref[c] >= 0) C.tri = 0;
				O.n[size_t(me)].child[c] = id;
			}
		} });
	O.n.resize(n_nodes);
	const double t_setup = now_seconds();
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
		if (std::getenv("FPT_BVH_TIMERS")) std::fprintf(stderr, "  batch %u: area %.4f, visits %llu\n", it, c, (unsigned long long)O.visits);          // (bench scene: 67.44 -> 58.87, 58.24, 58.15, 58.07, 58.07, 57.92, 57.91, 57.94;
		                                                                                                                                              //  the traversal model prices 1 / 2 / 3 / 8 batches at 114.8 / 114.2 / 114.0 / 113.2 wave instructions per ray)
		if (c < best * (1.0 - 1.0e-3)) stale = 0; else ++stale;
		best = std::min(best, c);
	}
	bvh.opt_cost_after = float(O.cost()); bvh.opt_iterations = it;
	const double t_loop = now_seconds();
	// back into the array form: pre-order, parents before children (build_wide8's bottom-up pass walks the array backwards), leaves in the order met.
	// The walk is bound by the latency of 3.6 M scattered 64-byte nodes, so it runs on all threads: the top of the tree (depth <= kCut) is walked once to find
	// the subtrees below it, their sizes are counted in parallel, a prefix sum over them in pre-order gives every subtree its place, and they are written in parallel.
	// The cut is a constant: the sums (cost) are added in the same order whatever the number of threads.
	NoInitVector<BvhNode> out(ni);
	NoInitVector<uint32_t> prims(bvh.prims.size());
	struct Todo { int32_t node; int32_t out_parent; int which; uint32_t depth; };
	struct Piece { int32_t node; int32_t parent_piece; int which; uint32_t depth; bool sub; uint32_t n_inner, n_leaf, max_depth; double cost; uint32_t out_base, prim_base; };
	const uint32_t kCut = 11;
	const double root_area = std::max(O.n[size_t(O.root)].area, 1.0e-300);
	std::vector<Piece> pieces;
	{
		struct Top { int32_t node; int32_t parent_piece; int which; uint32_t depth; };
		std::vector<Top> stack; stack.push_back(Top{ O.root, -1, 0, 1 });
		while (!stack.empty())
		{
			const Top t = stack.back(); stack.pop_back();
			const ONode& X = O.n[size_t(t.node)];
			const bool sub = X.child[0] < 0 || t.depth > kCut;
			pieces.push_back(Piece{ t.node, t.parent_piece, t.which, t.depth, sub, 0u, 0u, 0u, 0.0, 0u, 0u });
			if (!sub)
			{
				const int32_t me = int32_t(pieces.size() - 1);
				stack.push_back(Top{ X.child[1], me, 1, t.depth + 1 });
				stack.push_back(Top{ X.child[0], me, 0, t.depth + 1 });
			}
		}
	}
	// one subtree: counted (write = false) or written at its place; returns through the piece
	auto walk = [&](Piece& pc, bool write) {
		std::vector<Todo> stack; stack.push_back(Todo{ pc.node, -1, 0, pc.depth });
		uint32_t n_out = 0, n_pr = 0, md = 0; double cost = 0.0;
		while (!stack.empty())
		{
			const Todo t = stack.back(); stack.pop_back();
			const ONode& X = O.n[size_t(t.node)];
			int32_t ref;
			if (X.child[0] < 0)
			{
				ref = ~int32_t(((pc.prim_base + n_pr) << 3) | 1u);
				if (write) prims[size_t(pc.prim_base) + n_pr] = X.tri;
				++n_pr;
				cost += X.area / root_area;
			}
			else
			{
				ref = int32_t(pc.out_base + n_out);
				if (write) { BvhNode N; std::memset(&N, 0, sizeof(N)); out[size_t(ref)] = N; }
				++n_out;
				md = std::max(md, t.depth);
				cost += X.area / root_area;
				_mm_prefetch(reinterpret_cast<const char*>(&O.n[size_t(X.child[1])]), _MM_HINT_T0);
				stack.push_back(Todo{ X.child[1], ref, 1, t.depth + 1 });
				stack.push_back(Todo{ X.child[0], ref, 0, t.depth + 1 });
			}
			if (write && t.out_parent >= 0)
			{
				BvhNode& P = out[size_t(t.out_parent)];
				if (t.which == 0) { P.child0 = ref; X.box.store(P.lo0, P.hi0); }
				else              { P.child1 = ref; X.box.store(P.lo1, P.hi1); }
			}
		}
		pc.n_inner = n_out; pc.n_leaf = n_pr; pc.max_depth = md; pc.cost = cost;
	};
	std::vector<uint32_t> subs;
	for (uint32_t i = 0; i < pieces.size(); ++i) if (pieces[i].sub) subs.push_back(i);
	const uint32_t wth = ni >= 65536 ? std::max(1u, O.threads) : 1u;
	std::atomic<size_t> next_piece(0);
	auto over_subs = [&](bool write) {
		next_piece.store(0);
		parallel_slices(wth, wth, [&](size_t, size_t, uint32_t) { for (;;) { const size_t k = next_piece.fetch_add(1); if (k >= subs.size()) return; walk(pieces[subs[k]], write); } }); };
	over_subs(false);
	uint32_t max_depth = 0; double cost = 0.0;
	{
		uint32_t o = 0, pr = 0;
		for (Piece& pc : pieces)          // pre-order: a top node takes one place, a subtree as many as it has
		{
			pc.out_base = o; pc.prim_base = pr;
			if (pc.sub) { o += pc.n_inner; pr += pc.n_leaf; max_depth = std::max(max_depth, pc.max_depth); cost += pc.cost; }
			else { o += 1; max_depth = std::max(max_depth, pc.depth); cost += O.n[size_t(pc.node)].area / root_area; }
		}
		if (size_t(o) != ni || size_t(pr) != prims.size()) throw std::runtime_error("fpt: internal BVH optimiser error (node count changed)");
	}
	for (const Piece& pc : pieces) if (!pc.sub) { BvhNode N; std::memset(&N, 0, sizeof(N)); out[pc.out_base] = N; }
	over_subs(true);
	for (const Piece& pc : pieces)
	{
		if (pc.parent_piece < 0) continue;
		const ONode& X = O.n[size_t(pc.node)];
		const int32_t ref = X.child[0] < 0 ? ~int32_t((pc.prim_base << 3) | 1u) : int32_t(pc.out_base);
		BvhNode& P = out[pieces[size_t(pc.parent_piece)].out_base];
		if (pc.which == 0) { P.child0 = ref; X.box.store(P.lo0, P.hi0); }
		else               { P.child1 = ref; X.box.store(P.lo1, P.hi1); }
	}
	bvh.nodes.swap(out); bvh.prims.swap(prims);
	bvh.max_depth = max_depth; bvh.sah_cost = float(cost);
	bvh.seconds_opt = float(now_seconds() - t0);
	if (std::getenv("FPT_BVH_TIMERS")) std::fprintf(stderr, "optimize: set-up %.3f s, batches %.3f (select %.3f, apply %.3f, visits %llu), write-back %.3f\n", t_setup - t0, t_loop - t_setup, O.t_select, O.t_apply, (unsigned long long)O.visits, now_seconds() - t_loop);
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
	uint32_t max_leaf = CW8_MAX_LEAF;      // triangles a leaf child may hold (FPT_BVH_MAX_LEAF=1: experiments)
	float c_prim = 0.6f;           // a triangle test's issue time over a node step's: 409 / 716 cycles since round 6 (tools/isa_classes.py; 0.45 = 100 / 228 instructions until round 5).  Swept on the
	                               // GPU, bench scene, driver's form (profiles/r06_ab_scheduling.txt): 0.3 -> 539.2, 0.45 -> 540.7, 0.6 -> 543.6, 0.8 -> 537.7 Msample/s; the traversal cost model moves by < 1.5 % over 0.2 .. 1.0
	struct Cell { float c[8]; uint8_t k[8]; uint8_t k8; uint8_t leaf; uint8_t count; };      // index 1..7 used; count = min(P_n, 255)
	const NoInitVector<BvhNode>& nodes;
	std::vector<Cell> cell;
	double root_area = 1.0;

	explicit Collapse(const NoInitVector<BvhNode>& n) : nodes(n) {}

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
		const float c_leaf = (P >= 1 && P <= max_leaf) ? area * float(P) * c_prim : 3.0e38f;
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

} // namespace

void build_wide8(uint32_t tri_count, const int32_t* idx, const float* vtx, HostBvh2& bvh)
{
	PoolScope pool_scope;
	const double t0 = now_seconds();
	bvh.nodes8.clear(); bvh.tris8.clear(); bvh.wide_depth = 0; bvh.stack_need = 0; bvh.wide_cost = 0.0f;
	bvh.n_inner_children = bvh.n_leaf_children = 0;
	for (int k = 0; k < 9; ++k) bvh.slot_hist[k] = 0;
	Collapse dp(bvh.nodes);
	if (const char* e = std::getenv("FPT_BVH_MAX_LEAF")) dp.max_leaf = uint32_t(std::max(1, std::min(int(CW8_MAX_LEAF), std::atoi(e))));
	if (const char* e = std::getenv("FPT_BVH_C_PRIM")) dp.c_prim = float(std::atof(e));
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
		const double cell_of[3] = { std::ldexp(1.0, ex[0]), std::ldexp(1.0, ex[1]), std::ldexp(1.0, ex[2]) };
		uint32_t imask = 0;
		for (int sl = 0; sl < 8; ++sl)
		{
			uint8_t* qlo[3] = { bytes + 32 + sl, bytes + 40 + sl, bytes + 48 + sl };
			uint8_t* qhi[3] = { bytes + 56 + sl, bytes + 64 + sl, bytes + 72 + sl };
			if (child_in_slot[sl] < 0) { for (int k = 0; k < 3; ++k) { *qlo[k] = 255; *qhi[k] = 0; } continue; }      // empty slot: no valid bits, inverted box
			const WideChild& c = ch[size_t(child_in_slot[sl])];
			for (int k = 0; k < 3; ++k)
			{
				const double p = nb.lo[k], cell = cell_of[k];
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
				E.inner_ref[E.n_inner++] = c.ref;
			}
			else
			{
				const uint32_t count = c.n_prims;
				if (count < 1 || count > CW8_MAX_LEAF) throw std::runtime_error("fpt: wide-BVH leaves hold 1..2 triangles");
				if (E.n_tris + count > 16) throw std::runtime_error("fpt: internal wide-BVH error (triangle range)");
				node.w[6] |= ((1u << count) - 1u) << (2 * sl);          // the records follow in slot order: record = tri_base + popcount(valid bits below)
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
	NoInitVector<BvhTriangle> records(size_t(tri_count) + 1);          // sized once: every triangle lands in exactly one leaf
	bvh.level_begin.clear();
	for (size_t lb = 0, depth = 1; lb < queue.size(); ++depth)
	{
		const size_t le = queue.size(), n_level = le - lb;
		bvh.level_begin.push_back(uint32_t(lb));
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
	bvh.level_begin.push_back(uint32_t(bvh.nodes8.size()));
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
			const bool has_leaf = (N.w[6] & 0xFFFFu) != 0u;
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
	PoolScope pool_scope;
	const double t0 = now_seconds();
	if (bvh.nodes8.empty() || tri_count == 0) return;
	if (bvh.tris8.size() != size_t(tri_count)) throw std::runtime_error("fpt: refit needs the geometry the tree was built over (triangle count differs)");
	const uint32_t n_threads = builder_threads();
	float scene_mag = 0.0f;
	{
		float part[64] = { 0.0f };
		parallel_slices(vertex_count, vertex_count >= 65536u ? n_threads : 1u, [&](size_t b, size_t e, uint32_t t) {
			float m = 0.0f;
			for (size_t v = b; v < e; ++v) for (int k = 0; k < 3; ++k) m = std::max(m, std::fabs(vtx[4 * v + k]));
			part[t] = m; });
		for (float m : part) scene_mag = std::max(scene_mag, m);
	}
	// nothing is written before every record's triangle id and vertex indices have been checked: a refused refit leaves the tree as it was
	for (size_t i = 0; i < (tri_count ? bvh.tris8.size() : 0); ++i)
	{
		const uint32_t tri = uint32_t(bvh.tris8[i].tri_id);
		if (tri >= tri_count) throw std::runtime_error("fpt: refit found a triangle record outside the mesh");
		const int32_t* ix = idx + 4 * size_t(tri);
		for (int c = 0; c < 3; ++c) if (ix[c] < 0 || uint32_t(ix[c]) >= vertex_count) throw std::runtime_error("fpt: vertex index out of range in refit");
	}
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
			const float pad = (m0 + scene_mag) * 2.0e-6f + 1.0e-30f;
			for (int k = 0; k < 3; ++k) { bx.lo[k] -= pad; bx.hi[k] += pad; }
			tri_box[i] = bx;
			const float* p0 = vtx + 4 * size_t(ix[0]); const float* p1 = vtx + 4 * size_t(ix[1]); const float* p2 = vtx + 4 * size_t(ix[2]);
			for (int k = 0; k < 3; ++k) { r.v0[k] = p0[k]; r.e1[k] = p1[k] - p0[k]; r.e2[k] = p2[k] - p0[k]; }
			r.mask = uint32_t(ix[3]); r.vpad = triangle_vpad(p0, p1, p2, scene_mag);
		} });
	bvh.scene_mag = scene_mag;
	// nodes bottom-up: children have larger indices than their parent (breadth-first numbering: a level is a contiguous range, the next level the children of
	// its nodes in order), so the levels are found front to back and worked back to front, each on all threads
	std::vector<Box> node_box(bvh.nodes8.size());
	std::vector<size_t> level_begin;
	for (size_t lb = 0, le = 1; lb < le && lb < bvh.nodes8.size();)
	{
		level_begin.push_back(lb);
		size_t children = 0;
		for (size_t n = lb; n < le; ++n) children += size_t(__builtin_popcount(bvh.nodes8[n].w[3] >> 24));
		lb = le; le = std::min(bvh.nodes8.size(), le + children);
	}
	level_begin.push_back(bvh.nodes8.size());
	auto refit_node = [&](size_t n)
	{
		BvhNode8& node = bvh.nodes8[n];
		uint8_t* bytes = reinterpret_cast<uint8_t*>(node.w);
		const uint32_t imask = bytes[15], child_base = node.w[4], tri_base = node.w[5];
		Box cb[8]; bool used[8];
		Box nb; nb.reset();
		for (int sl = 0; sl < 8; ++sl)
		{
			const uint32_t kind = cw8_slot_kind(node, sl);
			used[sl] = kind != 0;
			if (!kind) continue;
			if (kind == 1) cb[sl] = node_box[size_t(child_base) + uint32_t(__builtin_popcount(imask & ((1u << sl) - 1u)))];
			else
			{
				const uint32_t count = cw8_leaf_count(node, sl), first = cw8_leaf_first(node, sl);
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
	};
	for (size_t L = level_begin.size() - 1; L-- > 0;)
	{
		const size_t lb = level_begin[L], n_level = level_begin[L + 1] - lb;
		parallel_slices(n_level, n_level >= 512 ? n_threads : 1u, [&](size_t b, size_t e, uint32_t) { for (size_t i = b; i < e; ++i) refit_node(lb + i); });
	}
	bvh.seconds_refit = float(now_seconds() - t0);
}

void build_acceleration(uint32_t tri_count, const int32_t* idx, uint32_t vertex_count, const float* vtx, HostBvh2& out, uint32_t stack_limit)
{
	PoolScope pool_scope;
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
	// the binary tree has served: a context keeps only what the kernel walks and the refit rewrites (123 MB less per 1.8 M triangles)
	NoInitVector<BvhNode>().swap(out.nodes); NoInitVector<uint32_t>().swap(out.prims);
}

} // namespace fpt
