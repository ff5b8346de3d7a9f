// fpt_bvh.h — the acceleration structure that replaces OptiX's "Trbvh" + RTX triangles (src/rt.cpp:284-331): an 8-wide compressed BVH
// ("CW8") built on the host as the SAH-optimal collapse of a binned-SAH binary tree that an insertion-based optimisation pass has improved.
//
// Device layout, chosen for CDNA4:
//   * one 80-byte node holds EIGHT children's boxes on a node-local 8-bit grid + what is needed to find them (BvhNode8 below): a ray
//     needs a third of the dependent fetches of a binary tree, and the tree is a quarter of the size;
//   * leaves reference runs of 1..2 pre-transformed 48-byte triangle records {v0, e1 = v1-v0, e2 = v2-v0, id, mask}; the edges are
//     computed on the host in fp32 exactly as the intersector would, so results are unchanged.
// BvhNode (fp32, two children) is the builder's intermediate: child reference >= 0 inner node index; < 0 leaf, ~ref = (first_prim << 3) | count.
#pragma once
#include <stdint.h>
#include <memory>
#include <vector>

namespace fpt {

// std::vector that leaves trivially constructible elements uninitialised on resize(): the builder sizes its arrays first and fills them from many threads
// (value-initialising 116 MB of nodes on one thread cost as much as a level of the build)
template <class T> struct NoInitAllocator : std::allocator<T>
{
	template <class U> struct rebind { typedef NoInitAllocator<U> other; };
	NoInitAllocator() = default;
	template <class U> NoInitAllocator(const NoInitAllocator<U>&) {}
	template <class U> void construct(U* p) { ::new (static_cast<void*>(p)) U; }
	template <class U, class... A> void construct(U* p, A&&... a) { ::new (static_cast<void*>(p)) U(static_cast<A&&>(a)...); }
};
template <class T> using NoInitVector = std::vector<T, NoInitAllocator<T>>;

struct alignas(64) BvhNode
{
	float lo0[3], hi0[3];
	float lo1[3], hi1[3];
	int32_t child0, child1;
	int32_t pad0, pad1;
};
static_assert(sizeof(BvhNode) == 64, "BVH2 node must be one 64-byte record");

struct alignas(16) BvhTriangle
{
	float v0[3], e1[3], e2[3];
	int32_t tri_id;
	uint32_t mask;
	float vpad;          // constant part of the tolerance of the intersector's box clause for this triangle: 5e-7 (|triangle|max + |scene|max)
};
static_assert(sizeof(BvhTriangle) == 48, "triangle record must be 48 bytes");      // (64-byte records that also carry e1 x e2 -- nine instructions less per test, no record straddling a
                                                                                    //  cache line -- were built and measured in round 6: 1 - 1.5 % SLOWER, +29 MB; EXPERIMENTS B2)

// 8-wide compressed node ("CW8", after Ylitie, Karras, Laine: Efficient Incoherent Ray Traversal on GPUs Through Compressed Wide BVHs,
// HPG 2017), 80 bytes = five 16-byte loads for eight children:
//   w[0..2]  p          : node origin (the lower corner of the node's box), fp32
//   w[3]     e.x | e.y << 8 | e.z << 16 | imask << 24 : per-axis exponent BYTES of the node-local grid (cell = 2^(e - 127)) and the
//                         bit mask of the slots that hold inner children
//   w[4]     child_base : index of the first inner child (inner children are stored contiguously in slot order)
//   w[5]     tri_base   : index of the node's first triangle record (the leaf children's records follow each other in slot order, packed)
//   w[6]     valid      : bits 0..15, two per slot: bit 2s = the leaf in slot s has a first triangle, bit 2s + 1 = it has a second one (leaves hold 1 or 2
//                         triangles; inner and empty slots: 00); bits 16..31 zero.  The record of the triangle at bit k is tri_base + popcount(valid below k).
//   w[7]     0 (spare)
//   w[8..19] qlo.x[8] qlo.y[8] qlo.z[8] qhi.x[8] qhi.y[8] qhi.z[8] : child boxes on the node-local 8-bit grid, snapped outward; empty slots: lo 255, hi 0
// Slots are assigned so that visiting them in the order (slot ^ (7 - ray octant)) descending is roughly front to back for every octant:
// the traversal needs no sorting, and one stack entry (child_base, hit bits) stands for all the hit children of a node.
// (Until round 5: w[6..7] = eight `meta` bytes -- inner 0x20 | 24 + slot, leaf unary count << 5 | offset -- from which the kernel shifted every child's bits into
//  place, 4 slow-class instructions per child; leaves held up to 3 triangles.  Two-triangle leaves cost nothing: 11.09 / 8.00 node steps / triangle tests per ray
//  against 11.06 / 8.08 on the bench scene's rays, tools/bvh_stats.py.)
struct alignas(16) BvhNode8 { uint32_t w[20]; };
static_assert(sizeof(BvhNode8) == 80, "CW8 node must be 80 bytes");      // (padding the record to a 128-byte line was measured: no gain)
static constexpr uint32_t CW8_MAX_LEAF = 2;
// host-side decoding of a slot: 0 = empty, 1 = inner, 2 = leaf
inline uint32_t cw8_slot_kind(const BvhNode8& n, int slot) { return ((n.w[3] >> 24) >> slot) & 1u ? 1u : (((n.w[6] >> (2 * slot)) & 1u) ? 2u : 0u); }
inline uint32_t cw8_leaf_count(const BvhNode8& n, int slot) { return ((n.w[6] >> (2 * slot)) & 1u) + ((n.w[6] >> (2 * slot + 1)) & 1u); }
inline uint32_t cw8_leaf_first(const BvhNode8& n, int slot) { return n.w[5] + uint32_t(__builtin_popcount(n.w[6] & ((1u << (2 * slot)) - 1u))); }
inline uint32_t cw8_inner_child(const BvhNode8& n, int slot) { return n.w[4] + uint32_t(__builtin_popcount((n.w[3] >> 24) & ((1u << slot) - 1u))); }

struct HostBvh2
{
	NoInitVector<BvhNode> nodes;             // the binary SAH tree (builder intermediate), one triangle per leaf: the collapse forms the leaves
	NoInitVector<uint32_t> prims;            // triangle ids in leaf order
	uint32_t max_depth = 0;
	float sah_cost = 0.0f;
	// the 8-wide collapse of the same tree (build_wide8): what the traversal kernel walks
	std::vector<BvhNode8> nodes8;
	NoInitVector<BvhTriangle> tris8;         // triangle records grouped per wide node
	uint32_t wide_depth = 0;
	std::vector<uint32_t> level_begin;       // wide nodes are numbered breadth-first: level L = [level_begin[L], level_begin[L + 1]); what a refit walks bottom-up
	uint32_t stack_need = 0;                 // upper bound of the traversal-stack entries a ray can need in this tree (see build_wide8)
	uint32_t slot_hist[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };      // wide nodes by number of used child slots
	uint32_t n_inner_children = 0, n_leaf_children = 0;
	float wide_cost = 0.0f;                  // SAH cost of the collapse (c_node = 1 per wide node, c_prim per triangle, areas relative to the root)
	float seconds_bvh2 = 0.0f, seconds_wide = 0.0f, seconds_opt = 0.0f, seconds_refit = 0.0f;
	float opt_cost_before = 0.0f, opt_cost_after = 0.0f;      // optimize_bvh2: sum of the inner nodes' areas relative to the root's
	uint32_t opt_iterations = 0;
	uint32_t threads = 1;
	uint32_t device_nodes = 0, device_records = 0;      // what the device holds (a device-side build keeps no host copy: nodes8 / tris8 stay empty)
	bool built_on_device = false;
	float scene_mag = 0.0f;                  // largest |coordinate| of the vertex array the tree was built (or refitted) over
};

// idx: int4 per triangle (x,y,z vertex ids, w shadow mask); vtx: float4 per vertex.  Multi-threaded (std::thread, one pool per build): the big ranges at the top
// of the tree are binned and partitioned by all threads, the subtrees below are handed out; references are partitioned in place, nodes come out in pre-order without
// stitching (a subtree over m triangles has m - 1 nodes); the result does not depend on the number of threads.
// sah_depth: SAH splits down to that depth, object-median splits below (depth <= sah_depth + log2(n) for any input); 0 = a balanced median tree
void build_bvh2(uint32_t tri_count, const int32_t* idx, uint32_t vertex_count, const float* vtx, HostBvh2& out, uint32_t sah_depth = 30);
// insertion-based optimisation of the binary tree (Bittner et al. 2013): batches of the worst inner nodes are removed and their subtrees re-inserted
// where they cost least; stops after max_iterations batches or when the cost no longer falls.  Call between build_bvh2 and build_wide8.
void optimize_bvh2(HostBvh2& bvh, uint32_t max_iterations = 16, double batch_fraction = 0.01);
// collapses out.nodes / out.prims into out.nodes8 / out.tris8: the SAH-optimal 8-wide collapse (dynamic programme of Ylitie et al. 2017, section 3:
// which binary nodes become wide nodes, which subtrees of <= 2 triangles become leaves), octant-ordered slots by an exact 8x8 assignment,
// outward 8-bit quantisation checked in double
void build_wide8(uint32_t tri_count, const int32_t* idx, const float* vtx, HostBvh2& bvh);
// the vertices moved, the topology stays: triangle records and every node's boxes recomputed in place (nodes8 / tris8), bottom-up; nothing else changes
void refit_wide8(uint32_t tri_count, const int32_t* idx, uint32_t vertex_count, const float* vtx, HostBvh2& bvh);
// build_bvh2 + optimize_bvh2 + build_wide8; a tree whose traversal-stack bound (bvh.stack_need) exceeds stack_limit is built again without the
// optimisation and then with shallower SAH limits.  The caller checks bvh.stack_need against its kernel.
void build_acceleration(uint32_t tri_count, const int32_t* idx, uint32_t vertex_count, const float* vtx, HostBvh2& bvh, uint32_t stack_limit);

} // namespace fpt
