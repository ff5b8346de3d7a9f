// fpt_bvh.h — BVH2 acceleration structure that replaces OptiX's "Trbvh" + RTX triangles (src/rt.cpp:284-331).
//
// Device layout, chosen for CDNA4:
//   * one node = BOTH children's boxes + both child references, so that one lane-private fetch decides the next step for the two
//     subtrees; built as a 64-byte fp32 record (BvhNode), shipped to the device as its 32-byte quantised twin (BvhNode32);
//   * leaves reference runs of 1..4 pre-transformed 48-byte triangle records {v0, e1 = v1-v0, e2 = v2-v0, id, mask};
//     the edges are computed on the host in fp32 exactly as the intersector would, so results are unchanged.
// Child reference: >= 0 inner node index; < 0 leaf, ~ref = (first_record << 3) | count.
#pragma once
#include <stdint.h>
#include <vector>

namespace fpt {

struct alignas(64) BvhNode
{
	float lo0[3], hi0[3];
	float lo1[3], hi1[3];
	int32_t child0, child1;
	int32_t pad0, pad1;
};
static_assert(sizeof(BvhNode) == 64, "BVH2 node must be one 64-byte record");

// The record the traversal kernel actually fetches: the same node with both boxes snapped OUTWARD onto a 16-bit grid over the
// scene's bounds (decoded coordinate = grid_base + q * grid_step, never inside the fp32 box), 32 bytes = two 16-byte loads per
// lane instead of four.  Traversal is bound by the per-CU address/L1 pipeline (one 16-byte lane request per clock for divergent
// lanes), so halving the requests per node is what counts; looser boxes can only add visits, never change a hit.
struct alignas(32) BvhNode32
{
	uint16_t q[12];          // lo0.xyz hi0.xyz lo1.xyz hi1.xyz
	int32_t  child0, child1;
};
static_assert(sizeof(BvhNode32) == 32, "quantised BVH2 node must be one 32-byte record");

struct alignas(16) BvhTriangle
{
	float v0[3], e1[3], e2[3];
	int32_t tri_id;
	uint32_t mask;
	uint32_t pad;
};
static_assert(sizeof(BvhTriangle) == 48, "triangle record must be 48 bytes");

struct HostBvh2
{
	std::vector<BvhNode> nodes;
	std::vector<BvhNode32> nodes32;          // same topology and numbering as `nodes`
	float grid_base[3] = { 0, 0, 0 }, grid_step[3] = { 1, 1, 1 };
	std::vector<BvhTriangle> tris;
	uint32_t max_depth = 0;
	float sah_cost = 0.0f;
};

// idx: int4 per triangle (x,y,z vertex ids, w shadow mask); vtx: float4 per vertex
void build_bvh2(uint32_t tri_count, const int32_t* idx, uint32_t vertex_count, const float* vtx, HostBvh2& out);

} // namespace fpt
