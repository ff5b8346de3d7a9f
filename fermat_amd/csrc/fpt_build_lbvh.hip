// fpt_build_lbvh.hip — the FAST build mode of the acceleration structure, entirely on the device (round 6): Morton-order binary radix tree -> SAH-optimal 8-wide
// collapse -> the CW8 nodes and triangle records fpt_trace.hip walks.  The reference builds on the GPU too (OptiX "Trbvh", src/rt.cpp:307-322; its own GPU builder is
// contrib/cugar/bvh/cuda/lbvh_builder.h, lbvh_builder_inline.h:76-116: Morton codes, a radix sort, a binary radix tree over the sorted codes), and its update_model hands
// DEVICE pointers over (src/renderer.cu:999-1017).  The host builder (fpt_bvh.cpp: binned SAH + re-insertion + the same collapse) stays the QUALITY mode and the default;
// this is what update_model(rebuild) costs when the topology changes every frame: milliseconds instead of 0.34 s + two PCIe copies, for a tree that traverses slower
// (the numbers are in DESIGN.md 5).
//
// Stages, all on the context's stream (host reads back 8 bytes per level of the wide tree + two small status words):
//   1 refs      per triangle: validation, the padded box of build_bvh2 (2e-6 (|tri|max + |scene|max)), bounds of the boxes and of their centres     [streaming, HBM]
//   2 codes     63-bit Morton code of each box centre on the grid of the centre bounds; rocPRIM radix sort of (code, triangle)                      [HBM, 4 passes]
//   3 tree      Karras (HPG 2012): every inner node of the binary radix tree finds its range and split independently; ties broken by position        [latency]
//   4 fit + DP  bottom-up in rounds (a node is done a round after its children; kernel boundaries are the only synchronisation): boxes, and the collapse's cost rows
//               C(n, 1..7) of Ylitie et al. 2017 (fpt_bvh.cpp Collapse)
//   5 emission  level by level from the root: a thread per wide node gathers its <= 8 children from the DP's decisions, assigns them to octant slots (the same exact
//               8 x 8 assignment as the host builder, fpt_cw8_slots.h), snaps their boxes outward onto the node's 8-bit grid and writes the 80-byte node; a scan of
//               the level's child and triangle counts hands out child_base / tri_base in node order, so the tree is the same whatever the scheduling; records follow
//   6 bound     the traversal-stack bound of the tree, bottom-up by level (fpt_rt_create_geometry refuses a tree the kernel's stack cannot hold -> host builder)
// rocPRIM (the ROCm-native primitives library) supplies the radix sort and the scans; everything else is here.  No MFMA: integer / pointer work.
#include "fpt_device.h"
#include "fpt_bvh.h"
#include "fpt_cw8_slots.h"
#include "fpt_host.h"
#include <rocprim/rocprim.hpp>

namespace fpt {

struct LbvhBox { float lo[3], hi[3]; };
// the collapse's cell of one binary node (fpt_bvh.cpp Collapse::Cell): c[i - 1] = the cheapest way to represent the subtree by at most i child slots, k[i - 1] = how many
// of them go to the left child (0 = no split at this i: use i - 1), k8 = the split of a full wide node's 8 slots, leaf = the subtree is cheapest as one leaf (<= 2 triangles)
struct LbvhCell { float c[7]; uint8_t k[7]; uint8_t k8, leaf, count; };
static constexpr float C_PRIM = 0.6f, C_NODE = 1.0f;           // fpt_bvh.cpp Collapse

__device__ __forceinline__ float hmin(float a, float b) { return (b < a) ? b : a; }          // std::min / std::max as the host builder applies them (NaN operands ignored)
__device__ __forceinline__ float hmax(float a, float b) { return (a < b) ? b : a; }
__device__ __forceinline__ int ordered(float f) { const int i = __float_as_int(f); return i ^ ((i >> 31) & 0x7FFFFFFF); }
__device__ __forceinline__ float unordered(int i) { return __int_as_float(i ^ ((i >> 31) & 0x7FFFFFFF)); }
__device__ __forceinline__ double half_area(const LbvhBox& b)
{
	const double ex = double(b.hi[0]) - double(b.lo[0]), ey = double(b.hi[1]) - double(b.lo[1]), ez = double(b.hi[2]) - double(b.lo[2]);
	return (ex < 0 || ey < 0 || ez < 0) ? 0.0 : ex * ey + ez * (ex + ey);
}

// ---- 1: references ---------------------------------------------------------------------------------------------------------
// status[0] = error bits (2 vertex index out of range); bounds[0..5] = ordered-int min / max of the boxes, [6..11] of the box centres
// (bounds of a block go to partials[block][12]: a chip-wide atomic per wave on twelve words of one cache line cost 3.8 ms for 1.8 M triangles -- ~90 atomics per microsecond --
//  where the kernel streams its data in 0.1 ms; lbvh_bounds_kernel folds the partials)
__global__ __launch_bounds__(256) void lbvh_refs_kernel(uint32_t n, const int4* __restrict__ idx, uint32_t n_verts, const float4* __restrict__ vtx, const uint32_t* __restrict__ scan,
                                                       LbvhBox* __restrict__ refs, int* __restrict__ partials, uint32_t* __restrict__ status)
{
	__shared__ int sh_lo[4][6], sh_hi[4][6];
	const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
	int lo[6], hi[6];
	#pragma unroll
	for (int k = 0; k < 6; ++k) { lo[k] = 0x7FFFFFFF; hi[k] = int(0x80000000u); }
	uint32_t err = 0;
	if (t < n)
	{
		const int4 ix = idx[t];
		LbvhBox b;
		if (ix.x < 0 || uint32_t(ix.x) >= n_verts || ix.y < 0 || uint32_t(ix.y) >= n_verts || ix.z < 0 || uint32_t(ix.z) >= n_verts) { err = 2u; for (int k = 0; k < 3; ++k) { b.lo[k] = 0.0f; b.hi[k] = 0.0f; } }
		else
		{
			const float4 q0 = vtx[ix.x], q1 = vtx[ix.y], q2 = vtx[ix.z];
			const float p[3][3] = { { q0.x, q0.y, q0.z }, { q1.x, q1.y, q1.z }, { q2.x, q2.y, q2.z } };
			float m0 = 0.0f;
			#pragma unroll
			for (int k = 0; k < 3; ++k) { b.lo[k] = 3.0e38f; b.hi[k] = -3.0e38f; }
			#pragma unroll
			for (int c = 0; c < 3; ++c)
				#pragma unroll
				for (int k = 0; k < 3; ++k) { b.lo[k] = hmin(b.lo[k], p[c][k]); b.hi[k] = hmax(b.hi[k], p[c][k]); m0 = hmax(m0, fabsf(p[c][k])); }
			const float pad = (m0 + as_f32(scan[0])) * 2.0e-6f + 1.0e-30f;
			#pragma unroll
			for (int k = 0; k < 3; ++k) { b.lo[k] -= pad; b.hi[k] += pad; }
		}
		refs[t] = b;
		#pragma unroll
		for (int k = 0; k < 3; ++k)
		{
			lo[k] = ordered(b.lo[k]); hi[k] = ordered(b.hi[k]);
			const float c = 0.5f * b.lo[k] + 0.5f * b.hi[k];
			if (c == c && fabsf(c) < 3.0e38f) { lo[3 + k] = hi[3 + k] = ordered(c); }
		}
	}
	for (int off = 32; off > 0; off >>= 1)
	{
		err |= __shfl_down(err, off);
		#pragma unroll
		for (int k = 0; k < 6; ++k) { lo[k] = min(lo[k], __shfl_down(lo[k], off)); hi[k] = max(hi[k], __shfl_down(hi[k], off)); }
	}
	if ((threadIdx.x & 63u) == 0u)
	{
		if (err) atomicOr(status, err);
		for (int k = 0; k < 6; ++k) { sh_lo[threadIdx.x >> 6][k] = lo[k]; sh_hi[threadIdx.x >> 6][k] = hi[k]; }
	}
	__syncthreads();
	if (threadIdx.x < 6)
	{
		const int k = threadIdx.x;
		const int l = min(min(sh_lo[0][k], sh_lo[1][k]), min(sh_lo[2][k], sh_lo[3][k])), h = max(max(sh_hi[0][k], sh_hi[1][k]), max(sh_hi[2][k], sh_hi[3][k]));
		// layout of a partial = layout of `bounds`: [0..2] box min, [3..5] box max, [6..8] centre min, [9..11] centre max
		int* out = partials + size_t(blockIdx.x) * 12;
		if (k < 3) { out[k] = l; out[3 + k] = h; } else { out[6 + (k - 3)] = l; out[9 + (k - 3)] = h; }
	}
}
__global__ __launch_bounds__(256) void lbvh_bounds_kernel(uint32_t n_blocks, const int* __restrict__ partials, int* __restrict__ bounds)
{
	__shared__ int sh[256];
	for (int k = 0; k < 12; ++k)
	{
		const bool is_min = (k % 6) < 3;
		int v = is_min ? 0x7FFFFFFF : int(0x80000000u);
		for (uint32_t b = threadIdx.x; b < n_blocks; b += 256) { const int x = partials[size_t(b) * 12 + k]; v = is_min ? min(v, x) : max(v, x); }
		sh[threadIdx.x] = v; __syncthreads();
		for (int off = 128; off > 0; off >>= 1) { if (int(threadIdx.x) < off) sh[threadIdx.x] = is_min ? min(sh[threadIdx.x], sh[threadIdx.x + off]) : max(sh[threadIdx.x], sh[threadIdx.x + off]); __syncthreads(); }
		if (threadIdx.x == 0) bounds[k] = sh[0];
		__syncthreads();
	}
}

// ---- 2: Morton codes -------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long spread21(uint32_t v)
{
	unsigned long long x = v & 0x1FFFFFull;
	x = (x | x << 32) & 0x1F00000000FFFFull; x = (x | x << 16) & 0x1F0000FF0000FFull; x = (x | x << 8) & 0x100F00F00F00F00Full;
	x = (x | x << 4) & 0x10C30C30C30C30C3ull; x = (x | x << 2) & 0x1249249249249249ull;
	return x;
}
__global__ __launch_bounds__(256) void lbvh_codes_kernel(uint32_t n, const LbvhBox* __restrict__ refs, const int* __restrict__ bounds, unsigned long long* __restrict__ keys, uint32_t* __restrict__ vals)
{
	const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= n) return;
	const LbvhBox b = refs[t];
	uint32_t q[3];
	#pragma unroll
	for (int k = 0; k < 3; ++k)
	{
		const float lo = unordered(bounds[6 + k]), hi = unordered(bounds[9 + k]);
		const float c = 0.5f * b.lo[k] + 0.5f * b.hi[k];
		const double ext = double(hi) - double(lo);
		double f = ext > 0.0 ? (double(c) - double(lo)) / ext : 0.0;
		f = (f == f) ? (f < 0.0 ? 0.0 : (f > 1.0 ? 1.0 : f)) : 0.0;
		const double s = f * 2097152.0;
		q[k] = s >= 2097151.0 ? 2097151u : uint32_t(s);
	}
	keys[t] = (spread21(q[0]) << 2) | (spread21(q[1]) << 1) | spread21(q[2]);          // (bits handed to the longest remaining extent instead of x, y, z in turn: measured, no better -- EXPERIMENTS B2)
	vals[t] = t;
}

// ---- 3: the binary radix tree (Karras 2012) ----------------------------------------------------------------------------------
// child references: >= 0 an inner node, < 0 ~position of a leaf in the sorted order
__device__ __forceinline__ int lbvh_delta(const unsigned long long* __restrict__ keys, uint32_t n, int i, long long j)
{
	if (j < 0 || j >= (long long)n) return -1;
	const unsigned long long a = keys[i], b = keys[j];
	return a == b ? 64 + __clz(uint32_t(i) ^ uint32_t(j)) : __clzll((long long)(a ^ b));
}
__global__ __launch_bounds__(256) void lbvh_tree_kernel(uint32_t n, const unsigned long long* __restrict__ keys, int* __restrict__ left, int* __restrict__ right)
{
	const int i = int(blockIdx.x * blockDim.x + threadIdx.x);
	if (i >= int(n) - 1) return;
	const int d = (lbvh_delta(keys, n, i, i + 1) - lbvh_delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
	const int dmin = lbvh_delta(keys, n, i, i - d);
	long long lmax = 2;
	while (lbvh_delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
	long long l = 0;
	for (long long t = lmax / 2; t >= 1; t /= 2) if (lbvh_delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
	const int j = int(i + l * d);
	const int dnode = lbvh_delta(keys, n, i, j);
	long long s = 0;
	for (long long t = (l + 1) / 2, span = l; ; )
	{
		if (lbvh_delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
		if (t == 1) break;
		span = t; t = (span + 1) / 2;
	}
	const int split = int(i + s * d + min(d, 0));
	const int first = min(i, j), last = max(i, j);
	const int l_ref = (first == split) ? ~split : split, r_ref = (last == split + 1) ? ~(split + 1) : split + 1;
	left[i] = l_ref; right[i] = r_ref;
}

// ---- 4: boxes and the collapse's cost rows, bottom-up ------------------------------------------------------------------------
__device__ __forceinline__ void lbvh_row(int ref, const LbvhBox& b, double inv_root_area, const LbvhCell* __restrict__ cells, float* c, uint32_t& count)
{
	if (ref >= 0) { const LbvhCell X = cells[ref]; for (int i = 0; i < 7; ++i) c[i] = X.c[i]; count = X.count; return; }
	count = 1u;
	const float v = float(half_area(b) * inv_root_area) * C_PRIM;
	for (int i = 0; i < 7; ++i) c[i] = v;
}
// One ROUND of the bottom-up pass: every inner node whose two children were finished in an EARLIER round (leaves always are) computes its box and its cost row and stamps
// itself with the round's number.  A kernel boundary is the only synchronisation: within a round a node never reads what the round writes (a stamp equal to the current round
// does not count, a stale "not yet" only postpones the node by a round), so there are no fences and no atomics -- the first form of this pass (one thread per leaf walking up
// behind an atomic flag per node, two agent-scope fences per step: each a write-back / invalidate of the XCD's L2) took 13.7 ms of a 21 ms build for 1.8 M triangles.
// Rounds needed = the height of the radix tree (40-60 for Morton codes); the host launches them in groups and reads the root's stamp back.
__global__ __launch_bounds__(256) void lbvh_fit_round_kernel(uint32_t n, uint32_t round, const LbvhBox* __restrict__ refs, const uint32_t* __restrict__ vals, const int* __restrict__ left,
                                                            const int* __restrict__ right, uint32_t* __restrict__ stamp, LbvhBox* __restrict__ node_box, LbvhCell* __restrict__ cells,
                                                            const int* __restrict__ bounds)
{
	const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
	if (p + 1 >= n || stamp[p] != 0u) return;
	const int l = left[p], r = right[p];
	if ((l >= 0 && (stamp[l] == 0u || stamp[l] >= round)) || (r >= 0 && (stamp[r] == 0u || stamp[r] >= round))) return;
	LbvhBox root; for (int k = 0; k < 3; ++k) { root.lo[k] = unordered(bounds[k]); root.hi[k] = unordered(bounds[3 + k]); }
	const double ra = half_area(root);
	const double inv_root_area = 1.0 / (ra > 1.0e-300 ? ra : 1.0e-300);
	const LbvhBox b0 = l >= 0 ? node_box[l] : refs[vals[~l]], b1 = r >= 0 ? node_box[r] : refs[vals[~r]];
	LbvhBox nb;
	for (int k = 0; k < 3; ++k) { nb.lo[k] = hmin(b0.lo[k], b1.lo[k]); nb.hi[k] = hmax(b0.hi[k], b1.hi[k]); }
	node_box[p] = nb;
	// Collapse::solve_node (fpt_bvh.cpp)
	const float area = float(half_area(nb) * inv_root_area);
	float cl[7], cr[7]; uint32_t pl, pr;
	lbvh_row(l, b0, inv_root_area, cells, cl, pl); lbvh_row(r, b1, inv_root_area, cells, cr, pr);
	const uint32_t P = pl + pr;
	LbvhCell X; X.count = uint8_t(P < 255u ? P : 255u);
	float dist[9]; uint8_t dk[9];
	for (int j = 2; j <= 8; ++j)
	{
		dist[j] = 3.0e38f; dk[j] = 1;
		for (int k = 1; k < j; ++k)
		{
			if (k > 7 || j - k > 7) continue;
			const float v = cl[k - 1] + cr[j - k - 1];
			if (v < dist[j]) { dist[j] = v; dk[j] = uint8_t(k); }
		}
	}
	const float c_internal = dist[8] + area * C_NODE;
	const float c_leaf = (P >= 1u && P <= CW8_MAX_LEAF) ? area * float(P) * C_PRIM : 3.0e38f;
	X.k8 = dk[8]; X.leaf = c_leaf <= c_internal ? 1 : 0;
	X.c[0] = X.leaf ? c_leaf : c_internal; X.k[0] = 0;
	for (int i = 2; i <= 7; ++i)
	{
		if (dist[i] < X.c[i - 2]) { X.c[i - 1] = dist[i]; X.k[i - 1] = dk[i]; }
		else { X.c[i - 1] = X.c[i - 2]; X.k[i - 1] = 0; }
	}
	cells[p] = X;
	stamp[p] = round;
}

// ---- 5: emission of a level of wide nodes ------------------------------------------------------------------------------------
struct LbvhEmitTmp { int inner_ref[8]; uint32_t tri[16]; };
__device__ __forceinline__ int lbvh_grid_exponent(double ext)
{
	int e = -100;
	if (ext > 0.0)
	{
		e = int(ceil(log2(ext / 255.0)));
		while (ext / ldexp(1.0, e) > 255.0) ++e;
		while (e > -100 && ext / ldexp(1.0, e - 1) <= 255.0) --e;
	}
	return e < -100 ? -100 : (e > 120 ? 120 : e);
}
__global__ __launch_bounds__(64) void lbvh_emit_kernel(uint32_t n_level, const int* __restrict__ queue, const LbvhBox* __restrict__ refs, const uint32_t* __restrict__ vals,
                                                      const int* __restrict__ left, const int* __restrict__ right, const LbvhBox* __restrict__ node_box, const LbvhCell* __restrict__ cells,
                                                      BvhNode8* __restrict__ nodes, LbvhEmitTmp* __restrict__ tmp, uint2* __restrict__ counts, uint32_t* __restrict__ status)
{
	const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= n_level) return;
	const int root = queue[t];
	// the children this binary subtree contributes to a wide node (Collect of fpt_bvh.cpp build_wide8): depth-first, left before right
	struct Child { int ref; uint32_t n_prims; uint32_t prim[2]; LbvhBox box; };
	Child ch[8]; int n_ch = 0;
	int st_ref[16], st_budget[16], sp = 0;
	{
		const LbvhCell X = cells[root];
		st_ref[sp] = right[root]; st_budget[sp++] = 8 - int(X.k8);
		st_ref[sp] = left[root]; st_budget[sp++] = int(X.k8);
	}
	bool bad = false;
	while (sp > 0)
	{
		const int ref = st_ref[--sp]; const int budget = st_budget[sp];
		if (n_ch >= 8) { bad = true; break; }
		if (ref < 0) { Child& c = ch[n_ch++]; c.ref = -1; c.n_prims = 1; c.prim[0] = vals[~ref]; c.prim[1] = 0; c.box = refs[c.prim[0]]; continue; }
		const LbvhCell X = cells[ref];
		int i = budget;
		while (i > 1 && X.k[i - 1] == 0) --i;
		if (i <= 1)
		{
			Child& c = ch[n_ch++]; c.box = node_box[ref]; c.n_prims = 0; c.prim[0] = c.prim[1] = 0;
			if (X.leaf)
			{
				// a leaf of two triangles: both children of the binary node are leaves
				const int l = left[ref], r = right[ref];
				if (l >= 0 || r >= 0) { bad = true; break; }
				c.ref = -1; c.n_prims = 2; c.prim[0] = vals[~l]; c.prim[1] = vals[~r];
			}
			else c.ref = ref;
			continue;
		}
		if (sp + 2 > 16) { bad = true; break; }
		st_ref[sp] = right[ref]; st_budget[sp++] = i - int(X.k[i - 1]);
		st_ref[sp] = left[ref]; st_budget[sp++] = int(X.k[i - 1]);
	}
	if (bad) { atomicOr(status, 8u); n_ch = 0; }
	LbvhBox nb; for (int k = 0; k < 3; ++k) { nb.lo[k] = 3.0e38f; nb.hi[k] = -3.0e38f; }
	for (int c = 0; c < n_ch; ++c) for (int k = 0; k < 3; ++k) { nb.lo[k] = hmin(nb.lo[k], ch[c].box.lo[k]); nb.hi[k] = hmax(nb.hi[k], ch[c].box.hi[k]); }
	if (n_ch == 0) { for (int k = 0; k < 3; ++k) { nb.lo[k] = 0.0f; nb.hi[k] = 0.0f; } }
	int slot_of[8] = { 0, 1, 2, 3, 4, 5, 6, 7 };
	{
		double score[8][8];
		for (int c = 0; c < n_ch; ++c)
			for (int sl = 0; sl < 8; ++sl)
			{
				double v = 0.0;
				for (int k = 0; k < 3; ++k) v += (double(0.5f * (ch[c].box.lo[k] + ch[c].box.hi[k])) - double(0.5f * (nb.lo[k] + nb.hi[k]))) * (((sl >> (2 - k)) & 1) ? 1.0 : -1.0);
				score[c][sl] = v;
			}
		for (int c = n_ch; c < 8; ++c) for (int sl = 0; sl < 8; ++sl) score[c][sl] = 0.0;
		assign_slots(score, n_ch, slot_of);
	}
	int child_in_slot[8] = { -1, -1, -1, -1, -1, -1, -1, -1 };
	for (int c = 0; c < n_ch; ++c) child_in_slot[slot_of[c]] = c;
	BvhNode8 node;
	for (int w = 0; w < 20; ++w) node.w[w] = 0u;
	node.w[0] = as_u32(nb.lo[0]); node.w[1] = as_u32(nb.lo[1]); node.w[2] = as_u32(nb.lo[2]);
	int ex[3]; uint32_t ew = 0;
	for (int k = 0; k < 3; ++k) { ex[k] = lbvh_grid_exponent(double(nb.hi[k]) - double(nb.lo[k])); ew |= uint32_t(ex[k] + 127) << (8 * k); }
	uint32_t imask = 0, valid = 0, n_inner = 0, n_tris = 0;
	LbvhEmitTmp T;
	uint32_t q[12] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
	for (int sl = 0; sl < 8; ++sl)
	{
		const int c = child_in_slot[sl];
		for (int k = 0; k < 3; ++k)
		{
			double lo = 255.0, hi = 0.0;
			if (c >= 0)
			{
				const double p = nb.lo[k], cell = ldexp(1.0, ex[k]);
				const double clo = double(ch[c].box.lo[k]), chi = double(ch[c].box.hi[k]);
				lo = floor((clo - p) / cell); lo = lo < 0.0 ? 0.0 : (lo > 255.0 ? 255.0 : lo);
				while (lo > 0.0 && !(p + lo * cell <= clo)) lo -= 1.0;
				hi = ceil((chi - p) / cell); hi = hi < 0.0 ? 0.0 : (hi > 255.0 ? 255.0 : hi);
				while (hi < 255.0 && !(p + hi * cell >= chi)) hi += 1.0;
				if (!(p + lo * cell <= clo) || !(p + hi * cell >= chi)) bad = true;
			}
			q[2 * k + (sl >> 2)] |= uint32_t(lo) << (8 * (sl & 3));
			q[6 + 2 * k + (sl >> 2)] |= uint32_t(hi) << (8 * (sl & 3));
		}
		if (c < 0) continue;
		if (ch[c].ref >= 0) { imask |= 1u << sl; T.inner_ref[n_inner++] = ch[c].ref; }
		else
		{
			valid |= ((1u << ch[c].n_prims) - 1u) << (2 * sl);
			for (uint32_t j = 0; j < ch[c].n_prims; ++j) T.tri[n_tris++] = ch[c].prim[j];
		}
	}
	if (bad) atomicOr(status, 4u);          // non-finite vertices: a box that cannot be quantised
	node.w[3] = ew | (imask << 24); node.w[6] = valid;
	for (int w = 0; w < 12; ++w) node.w[8 + w] = q[w];
	nodes[t] = node;
	tmp[t] = T;
	counts[t] = make_uint2(n_inner, n_tris);
}
struct Uint2Plus { __host__ __device__ uint2 operator()(const uint2& a, const uint2& b) const { return make_uint2(a.x + b.x, a.y + b.y); } };
// bases in node order, the next level's queue, the records
__global__ __launch_bounds__(256) void lbvh_finish_kernel(uint32_t n_level, BvhNode8* __restrict__ nodes, const LbvhEmitTmp* __restrict__ tmp, const uint2* __restrict__ counts,
                                                         const uint2* __restrict__ offsets, uint32_t next_base, uint32_t tri_base, int* __restrict__ next_queue,
                                                         BvhTriangle* __restrict__ records, const int4* __restrict__ idx, const float4* __restrict__ vtx, const uint32_t* __restrict__ scan,
                                                         uint2* __restrict__ totals)
{
	const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= n_level) return;
	const uint2 cnt = counts[t], off = offsets[t];
	nodes[t].w[4] = next_base + off.x; nodes[t].w[5] = tri_base + off.y;
	const LbvhEmitTmp T = tmp[t];
	for (uint32_t j = 0; j < cnt.x; ++j) next_queue[off.x + j] = T.inner_ref[j];
	const float scene_mag = as_f32(scan[0]);
	for (uint32_t j = 0; j < cnt.y; ++j)
	{
		const uint32_t tri = T.tri[j];
		const int4 ix = idx[tri];
		const float4 q0 = vtx[ix.x], q1 = vtx[ix.y], q2 = vtx[ix.z];
		const float p[3][3] = { { q0.x, q0.y, q0.z }, { q1.x, q1.y, q1.z }, { q2.x, q2.y, q2.z } };
		BvhTriangle r;
		float mv = 0.0f;
		for (int k = 0; k < 3; ++k)
		{
			r.v0[k] = p[0][k]; r.e1[k] = p[1][k] - p[0][k]; r.e2[k] = p[2][k] - p[0][k];
			mv = hmax(mv, hmax(fabsf(p[0][k]), hmax(fabsf(p[1][k]), fabsf(p[2][k]))));
		}
		r.tri_id = int32_t(tri); r.mask = uint32_t(ix.w); r.vpad = (mv + scene_mag) * 5.0e-7f;
		records[tri_base + off.y + j] = r;
	}
	if (t == n_level - 1) *totals = make_uint2(off.x + cnt.x, off.y + cnt.y);
}

// ---- 6: the traversal-stack bound (fpt_bvh.cpp build_wide8), one level per launch, bottom-up ------------------------------------
__global__ __launch_bounds__(256) void lbvh_need_kernel(const BvhNode8* __restrict__ nodes, uint32_t begin, uint32_t count, uint32_t* __restrict__ need)
{
	const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= count) return;
	const uint32_t n = begin + t;
	const uint32_t imask = nodes[n].w[3] >> 24, n_inner = uint32_t(__popc(imask)), child_base = nodes[n].w[4];
	const bool has_leaf = (nodes[n].w[6] & 0xFFFFu) != 0u;
	uint32_t below = 0;
	for (uint32_t c = 0; c < n_inner; ++c) below = max(below, need[child_base + c]);
	need[n] = (has_leaf ? 1u : 0u) + (n_inner ? (n_inner >= 2u ? 1u : 0u) + below : 0u);
}

// occupancy of the finished tree (fpt_rt_bvh_stats): wide nodes by number of used slots, inner and leaf children; hist[0..8] slots, [9] inner children, [10] leaf children
__global__ __launch_bounds__(256) void lbvh_hist_kernel(const BvhNode8* __restrict__ nodes, uint32_t n, uint32_t* __restrict__ hist)
{
	__shared__ uint32_t sh[11];
	if (threadIdx.x < 11) sh[threadIdx.x] = 0u;
	__syncthreads();
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n)
	{
		const uint32_t inner = uint32_t(__popc(nodes[i].w[3] >> 24)), leaf = uint32_t(__popc(nodes[i].w[6] & 0x5555u));
		atomicAdd(&sh[inner + leaf], 1u); atomicAdd(&sh[9], inner); atomicAdd(&sh[10], leaf);
	}
	__syncthreads();
	if (threadIdx.x < 11 && sh[threadIdx.x]) atomicAdd(hist + threadIdx.x, sh[threadIdx.x]);
}

// ---- the driver --------------------------------------------------------------------------------------------------------------
template <class T> static T* carve(uint8_t*& p, size_t n) { T* r = reinterpret_cast<T*>(p); p += (n * sizeof(T) + 255) & ~size_t(255); return r; }

// Builds the tree over the DEVICE mesh into ctx->d_nodes / ctx->d_tris; fills the host-side bookkeeping (level ranges, counts, the stack bound).  Returns false -- nothing
// touched -- when the tree cannot be used (its stack bound exceeds `stack_limit`: a degenerate input) and the caller should fall back to the host builder; throws on bad input.
bool build_acceleration_device(fpt_context* ctx, uint32_t n, const int32_t* d_idx, uint32_t n_verts, const float* d_vtx, uint32_t stack_limit)
{
	hipStream_t s = ctx->stream;
	const double t0 = wall_seconds();
	// |scene|max (the same kernel the refit uses; no records to validate yet)
	ctx->d_refit_scan.alloc(2);
	FPT_HIP_CHECK(hipMemsetAsync(ctx->d_refit_scan.ptr, 0, 2 * sizeof(uint32_t), s));
	launch_refit_scan(0, d_idx, n_verts, d_vtx, 0, nullptr, ctx->d_refit_scan.ptr, s);
	// scratch, one allocation
	size_t sort_bytes = 0, scan_bytes = 0;
	{
		rocprim::double_buffer<unsigned long long> k(nullptr, nullptr); rocprim::double_buffer<uint32_t> v(nullptr, nullptr);
		FPT_HIP_CHECK(rocprim::radix_sort_pairs(nullptr, sort_bytes, k, v, n, 0, 63, s));
		FPT_HIP_CHECK(rocprim::exclusive_scan(nullptr, scan_bytes, (uint2*)nullptr, (uint2*)nullptr, make_uint2(0, 0), size_t(n), Uint2Plus(), s));
	}
	const size_t cap = n;          // a level of the wide tree holds fewer nodes than there are triangles
	size_t total = 0;
	auto sz = [&](size_t bytes) { total += (bytes + 255) & ~size_t(255); };
	sz(n * sizeof(LbvhBox)); sz(16 * sizeof(int)); sz((size_t((n + 255u) / 256u) * 12 + 12) * sizeof(int)); sz(n * 8); sz(n * 8); sz(n * 4); sz(n * 4); sz(n * 4); sz(n * 4); sz(n * 4); sz(n * 4); sz(n * 4); sz(n * sizeof(LbvhBox)); sz(n * sizeof(LbvhCell));
	sz(cap * 4); sz(cap * 4); sz(cap * sizeof(LbvhEmitTmp)); sz(cap * 8); sz(cap * 8); sz(64); sz(sort_bytes); sz(scan_bytes); sz(cap * sizeof(BvhNode8)); sz((size_t(n) + 1) * sizeof(BvhTriangle)); sz(cap * 4);
	// the scratch stays with the context (a host that rebuilds every frame allocates it once; fpt_destroy or a smaller build's reuse keeps it)
	if (ctx->d_build_scratch.count < total) ctx->d_build_scratch.alloc(total);
	uint8_t* p = ctx->d_build_scratch.ptr;
	LbvhBox* refs = carve<LbvhBox>(p, n); int* bounds = carve<int>(p, 16); int* partials = carve<int>(p, size_t((n + 255u) / 256u) * 12 + 12);
	unsigned long long* keys0 = carve<unsigned long long>(p, n); unsigned long long* keys1 = carve<unsigned long long>(p, n);
	uint32_t* vals0 = carve<uint32_t>(p, n); uint32_t* vals1 = carve<uint32_t>(p, n);
	int* left = carve<int>(p, n); int* right = carve<int>(p, n); uint32_t* flags = carve<uint32_t>(p, n);
	LbvhBox* node_box = carve<LbvhBox>(p, n); LbvhCell* cells = carve<LbvhCell>(p, n);
	int* queue0 = carve<int>(p, cap); int* queue1 = carve<int>(p, cap); LbvhEmitTmp* tmp = carve<LbvhEmitTmp>(p, cap);
	uint2* counts = carve<uint2>(p, cap); uint2* offsets = carve<uint2>(p, cap); uint32_t* status = carve<uint32_t>(p, 16);
	uint8_t* sort_tmp = carve<uint8_t>(p, sort_bytes); uint8_t* scan_tmp = carve<uint8_t>(p, scan_bytes);
	BvhNode8* nodes = carve<BvhNode8>(p, cap); BvhTriangle* records = carve<BvhTriangle>(p, size_t(n) + 1); uint32_t* need = carve<uint32_t>(p, cap);
	uint2* totals = reinterpret_cast<uint2*>(status + 4);

	FPT_HIP_CHECK(hipMemsetAsync(status, 0, 64, s));
	FPT_HIP_CHECK(hipMemsetAsync(flags, 0, size_t(n) * 4, s));
	const bool timers = std::getenv("FPT_BVH_TIMERS") != nullptr;
	double t_stage = wall_seconds(); double ms_stage[6] = { 0, 0, 0, 0, 0, 0 };
	auto stage = [&](int k) { if (timers) { FPT_HIP_CHECK(hipStreamSynchronize(s)); const double t = wall_seconds(); ms_stage[k] = (t - t_stage) * 1e3; t_stage = t; } };
	stage(0);          // |scene|max + scratch
	const dim3 B(256), G((n + 255u) / 256u);
	hipLaunchKernelGGL(lbvh_refs_kernel, G, B, 0, s, n, reinterpret_cast<const int4*>(d_idx), n_verts, reinterpret_cast<const float4*>(d_vtx), ctx->d_refit_scan.ptr, refs, partials, status);
	hipLaunchKernelGGL(lbvh_bounds_kernel, dim3(1), B, 0, s, (n + 255u) / 256u, partials, bounds);
	hipLaunchKernelGGL(lbvh_codes_kernel, G, B, 0, s, n, refs, bounds, keys0, vals0);
	stage(1);          // references + codes
	rocprim::double_buffer<unsigned long long> kb(keys0, keys1); rocprim::double_buffer<uint32_t> vb(vals0, vals1);
	FPT_HIP_CHECK(rocprim::radix_sort_pairs(sort_tmp, sort_bytes, kb, vb, n, 0, 63, s));
	const unsigned long long* keys = kb.current(); const uint32_t* vals = vb.current();
	stage(2);          // sort
	hipLaunchKernelGGL(lbvh_tree_kernel, G, B, 0, s, n, keys, left, right);
	stage(3);          // radix tree
	// bottom-up in rounds (lbvh_fit_round_kernel); `flags` holds the stamps.  Eight rounds per read-back of the root's stamp
	uint32_t rounds = 0;
	for (uint32_t root_stamp = 0; root_stamp == 0u;)
	{
		for (int k = 0; k < 8; ++k) { ++rounds; hipLaunchKernelGGL(lbvh_fit_round_kernel, G, B, 0, s, n, rounds, refs, vals, left, right, flags, node_box, cells, bounds); }
		FPT_HIP_CHECK(hipMemcpyAsync(&root_stamp, flags, 4, hipMemcpyDeviceToHost, s));
		FPT_HIP_CHECK(hipStreamSynchronize(s));
		require(rounds <= 4096, "fpt: internal device-build error (the bottom-up pass does not terminate)");
	}
	stage(4);          // boxes + cost rows
	uint32_t h_status[8] = { 0 };
	FPT_HIP_CHECK(hipMemcpyAsync(h_status, status, 4, hipMemcpyDeviceToHost, s));
	FPT_HIP_CHECK(hipStreamSynchronize(s));
	require(!(h_status[0] & 2u), "fpt: vertex index out of range");
	const double t_tree = wall_seconds();

	// emission, level by level
	std::vector<uint32_t> level_begin;
	const int root_ref = 0;
	FPT_HIP_CHECK(hipMemcpyAsync(queue0, &root_ref, 4, hipMemcpyHostToDevice, s));
	int* q_cur = queue0; int* q_next = queue1;
	uint32_t n_level = 1, level_base = 0, tri_total = 0;
	while (n_level)
	{
		require(size_t(level_base) + n_level <= cap, "fpt: internal device-build error (more wide nodes than triangles)");
		level_begin.push_back(level_base);
		hipLaunchKernelGGL(lbvh_emit_kernel, dim3((n_level + 63u) / 64u), dim3(64), 0, s, n_level, q_cur, refs, vals, left, right, node_box, cells, nodes + level_base, tmp, counts, status);
		size_t sb = scan_bytes;
		FPT_HIP_CHECK(rocprim::exclusive_scan(scan_tmp, sb, counts, offsets, make_uint2(0, 0), size_t(n_level), Uint2Plus(), s));
		hipLaunchKernelGGL(lbvh_finish_kernel, dim3((n_level + 255u) / 256u), B, 0, s, n_level, nodes + level_base, tmp, counts, offsets, level_base + n_level, tri_total, q_next, records,
		                   reinterpret_cast<const int4*>(d_idx), reinterpret_cast<const float4*>(d_vtx), ctx->d_refit_scan.ptr, totals);
		uint2 tot;
		FPT_HIP_CHECK(hipMemcpyAsync(&tot, totals, sizeof(tot), hipMemcpyDeviceToHost, s));
		FPT_HIP_CHECK(hipStreamSynchronize(s));
		level_base += n_level; tri_total += tot.y; n_level = tot.x;
		std::swap(q_cur, q_next);
		require(level_begin.size() <= 4096, "fpt: internal device-build error (runaway depth)");
	}
	const uint32_t n_nodes = level_base;
	level_begin.push_back(n_nodes);
	require(tri_total == n, "fpt: internal device-build error (a triangle was lost or doubled)");
	for (size_t L = level_begin.size() - 1; L-- > 0;)
		hipLaunchKernelGGL(lbvh_need_kernel, dim3((level_begin[L + 1] - level_begin[L] + 255u) / 256u), B, 0, s, nodes, level_begin[L], level_begin[L + 1] - level_begin[L], need);
	uint32_t h_hist[11] = { 0 };
	uint32_t* hist = reinterpret_cast<uint32_t*>(offsets);          // the level scan's offsets are no longer needed: 11 words of them hold the histogram
	FPT_HIP_CHECK(hipMemsetAsync(hist, 0, sizeof(h_hist), s));
	hipLaunchKernelGGL(lbvh_hist_kernel, dim3((n_nodes + 255u) / 256u), B, 0, s, nodes, n_nodes, hist);
	FPT_HIP_CHECK(hipMemcpyAsync(h_hist, hist, sizeof(h_hist), hipMemcpyDeviceToHost, s));
	uint32_t h_need = 0, h_scan[2] = { 0, 0 };
	FPT_HIP_CHECK(hipMemcpyAsync(&h_need, need, 4, hipMemcpyDeviceToHost, s));
	FPT_HIP_CHECK(hipMemcpyAsync(h_status, status, 4, hipMemcpyDeviceToHost, s));
	FPT_HIP_CHECK(hipMemcpyAsync(h_scan, ctx->d_refit_scan.ptr, 8, hipMemcpyDeviceToHost, s));
	FPT_HIP_CHECK(hipStreamSynchronize(s));
	FPT_HIP_CHECK(hipGetLastError());
	require(!(h_status[0] & 8u), "fpt: internal device-build error (collapse)");
	require(!(h_status[0] & 4u), "fpt: internal wide-BVH quantisation error: non-finite vertices?");
	if (h_need > stack_limit) return false;
	// the tree in exact-size arrays
	ctx->d_nodes.alloc(n_nodes); ctx->d_tris.alloc(tri_total);
	FPT_HIP_CHECK(hipMemcpyAsync(ctx->d_nodes.ptr, nodes, size_t(n_nodes) * sizeof(BvhNode8), hipMemcpyDeviceToDevice, s));
	FPT_HIP_CHECK(hipMemcpyAsync(ctx->d_tris.ptr, records, size_t(tri_total) * sizeof(BvhTriangle), hipMemcpyDeviceToDevice, s));
	FPT_HIP_CHECK(hipStreamSynchronize(s));
	HostBvh2& H = ctx->host_bvh;
	H = HostBvh2();
	H.level_begin = level_begin; H.wide_depth = uint32_t(level_begin.size() - 1); H.stack_need = h_need;
	H.device_nodes = n_nodes; H.device_records = tri_total; H.built_on_device = true;
	for (int k = 0; k < 9; ++k) H.slot_hist[k] = h_hist[k];
	H.n_inner_children = h_hist[9]; H.n_leaf_children = h_hist[10];
	std::memcpy(&H.scene_mag, &h_scan[0], 4);
	H.seconds_bvh2 = float(t_tree - t0); H.seconds_wide = float(wall_seconds() - t_tree); H.threads = 0;
	if (timers) std::fprintf(stderr, "build_acceleration_device: %u triangles -> %u wide nodes in %zu levels, stack bound %u; to the binary tree %.3f ms (|scene|max + scratch %.3f, references + codes %.3f, "
	                                 "sort %.3f, radix tree %.3f, boxes + cost rows %.3f), emission + bound + copy %.3f ms\n",
	                                 n, n_nodes, level_begin.size() - 1, h_need, H.seconds_bvh2 * 1e3, ms_stage[0], ms_stage[1], ms_stage[2], ms_stage[3], ms_stage[4], H.seconds_wide * 1e3);
	return true;
}

} // namespace fpt
