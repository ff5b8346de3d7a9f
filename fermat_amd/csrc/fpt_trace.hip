// fpt_trace.hip — hand-written gfx950 traversal kernels over the 8-wide compressed BVH (fpt_bvh.h BvhNode8): the replacement for OptiX
// behind RTContext::trace / trace_shadow (src/rt.cpp:558-659, src/kernels/optix_rt.cu:45-82,133-204, optix_base_shaders.h:42-91,
// optix_base_shadow_shaders.h:42-72).
//
// CDNA4 design:
//   * persistent waves with sharded ticket counters, chunked hand-out and partial-wave refill (a single device-scope counter sustains
//     only ~90 atomics/us on MI355X, and atomics to one 128-B line serialise chip-wide);
//   * the tree is the 8-wide collapse of the SAH BVH2 with 80-byte compressed nodes: a ray needs a third of the dependent fetches of
//     the binary tree (that kernel was bound by dependent-fetch latency x occupancy, not by the VALU: 45-50 % VALU busy at 8 waves/SIMD),
//     the tree is 4x smaller, and the eight slab tests of a node step are the same straight-line code in every lane;
//   * octant-ordered slots: children are visited in the order (slot ^ (7 - ray octant)) descending, fixed at build time, so there is no
//     sorting and ONE 8-byte stack entry (child_base, hit bits | imask) stands for all the hit children of a node; the stack lives in
//     LDS as [level][thread] uint2 (ds_read/write_b64, conflict-free), deeper levels spill to scratch;
//   * a node step decodes the 8-bit child boxes with v_cvt_f32_ubyteN + one FMA per plane (t = q * (2^e / d) + (p - o) / d); the near /
//     far planes are selected per axis from the ray's direction signs on the packed words, four children at a time; a child's test ends at
//     one subtraction and one v_alignbit (the sign of exit - entry shifted into an 8-bit miss mask), and the octant order of the inner
//     children and the triangle bits of the leaves come from two LDS tables looked up once per node step (round 6: gfx950 issues compares,
//     selects, left shifts and bit-field extracts in the slow class, 4.4 cycles per wave against 2.7 -- the per-child compare + select +
//     two shifts of rounds 2-5 were a fifth of the step);
//   * slab tests use FMAs (conservative: boxes are padded and snapped outward by the builders); the triangle test is the fixed-order "fpt-MT"
//     Moeller-Trumbore -- two cross products, determinant and t from one normal -- whose results must equal the CPU oracle bit for bit
//     (no FMA contraction, IEEE divide);
//   * closest hit = minimum t, ties -> lowest triangle id; barycentrics rounded through fp16 like OptiX's payload
//     (src/kernels/optix_payload.h:75-78); any-hit honours the per-triangle shadow mask (optix_base_shadow_shaders.h:54-59): results
//     are independent of the tree and of the traversal order, bit for bit;
//   * MIXED mode: one launch serves the closest-hit rays of bounce b+1 AND the shadow rays of bounce b (fused with
//     solve_occlusion).  A launch cannot end before its longest ray, so halving the number of launches per pass halves those tails.
// No MFMA: a pointer chase, not a contraction.
#include "fpt_device.h"
#include "fpt_bvh.h"
#include "fpt_psf.h"

namespace fpt {

#ifndef FPT_LDS_STACK
#define FPT_LDS_STACK 8            // uint2 entries: 8 levels x 256 threads x 8 B = 16 KB of LDS per block
#endif
#ifndef FPT_TRACE_MIN_WAVES
#define FPT_TRACE_MIN_WAVES 7      // 72 VGPRs, no vector spills.  Until round 4: 8 waves at 64 VGPRs with 5 VGPR + 26 SGPR spills (launch constants parked in the prologue and
                                   // reloaded at every refill); the two were equal within noise then (traversal ms per step on the bathroom2 stand-in, driver's form: 8 waves
                                   // 1.960-1.977, 7 waves 1.960-1.968, 6 waves 2.018-2.027, 5 waves 2.152-2.173; round 5, two runs each: 8 waves 1.967 / 1.961, 7 waves 1.956 / 1.956),
                                   // and round 5's queue layout (bookkeeping in the rays' .w words, fpt_device.h) costs the 64-register build 17 spilled VGPRs.  The register
                                   // cliff round 4 documented is still there one wave lower: any extra value live across the refill (a straggler-slot test, a restart flag, a
                                   // second exit condition) parked six of the ray's registers in scratch around every burst and cost 10-15 % (profiles/r04_exp_carry_over_launches.txt);
                                   // tools/isa_stats.py lists the scratch instructions block by block.  (Round 2, BVH2-era sweep on the bounce-1 rays: 8 waves 0.60 ms, 6: 0.70, 4: 0.71.)
#endif
#ifndef FPT_REFILL_MIN
#define FPT_REFILL_MIN 16          // round 4, on the bathroom2 stand-in (11 node steps per ray: a refill costs less of a ray) 32 -> 498, 24 -> 507, 16 -> 508, 8 -> 496 Msample/s; testball-room
                                   // 864 -> 887; rounds 1-3 scene (3.3 node steps per ray) 1637 vs 1635: no longer 32 (round 2, on that scene: 32 -> 1550, 16 -> 1533)
#endif
#ifndef FPT_CHUNK_MAX
#define FPT_CHUNK_MAX 256          // rays a wave draws per ticket: the last chunk a wave holds is the imbalance at the end of a launch.  Measured, Msample/s in the
#endif                             // driver's form / at 64 in flight: 1024 -> 1455 / 1678, 512 -> 1495 / 1710, 256 -> 1530 / 1723, 128 -> 1532 / 1704, 64 -> 1481 / 1637
static constexpr int TRACE_BLOCK = 256;
static constexpr int LDS_STACK   = FPT_LDS_STACK;        // levels x 256 threads x 4 B of LDS per block
static constexpr int OVF_STACK   = 48 - FPT_LDS_STACK;   // scratch overflow: 48 entries in all (fpt_rt_create_geometry checks the tree's stack bound against it)
static constexpr int REFILL_MIN  = FPT_REFILL_MIN;       // refill a wave once this many lanes are idle
static constexpr uint32_t TICKET_SHARDS = 8;             // one ticket counter per XCD-sized share of the waves
static constexpr uint32_t TICKET_PAD    = 32;            // counters sit 128 B apart: atomics on one cache line serialise chip-wide

enum TraceMode { MODE_CLOSEST = 0, MODE_ANY = 1, MODE_ANY_FUSED = 2, MODE_MIXED = 3, MODE_MIXED_PSF = 4, MODE_MIXED_HITS = 5, MODE_CLOSEST_QP = 6, MODE_CLOSEST_QS = 7, MODE_ANY_Q = 8 };
// *_QP / *_QS / ANY_Q (round 5): the rays of a renderer's own queues (fpt_device.h PathQueue / ShadowQueue), whose .w words carry PixelInfo and the pass offset instead of
// tmin / tmax: primary rays (0, 1e34), scattered rays (1e-3, 1e8), shadow rays (mask, 0.9999).  MIXED, MIXED_PSF and ANY_FUSED read such queues too.
constexpr bool closest_from_queue(int m) { return m == MODE_MIXED || m == MODE_MIXED_PSF || m == MODE_CLOSEST_QP || m == MODE_CLOSEST_QS; }
constexpr bool any_from_queue(int m) { return m == MODE_ANY_FUSED || m == MODE_MIXED || m == MODE_MIXED_PSF || m == MODE_ANY_Q; }
// MIXED_PSF: MIXED with the path-space-filtering resolve (`fused` points to a ResolveParams); MIXED_HITS: the any-hit rays' results are WRITTEN
// (`fused` points to their float4 Hit array) instead of resolved -- the bidirectional path tracer's connections, which its own kernel adds in order
constexpr bool mode_is_mixed(int m) { return m == MODE_MIXED || m == MODE_MIXED_PSF || m == MODE_MIXED_HITS; }

struct LaneRay
{
	f3 o, d;
	f3 idir;             // guarded reciprocal of d
	float tmin, tmax;
};

__device__ __forceinline__ float guarded_rcp(float d)
{
	const float a = fabsf(d);
	const float g = (a < 1.0e-20f) ? (d < 0.0f ? -1.0e-20f : 1.0e-20f) : d;
	return 1.0f / g;
}

// v_max_f32 / v_min_f32 / v_max3 / v_min3 on operands known to be ordinary numbers or infinities: spelled as instructions so that the
// compiler neither re-quiets loop-invariant operands nor splits the three-operand forms
__device__ __forceinline__ float raw_max(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float raw_min(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float raw_max3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float raw_min3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
// byte K of a packed word as a float (v_cvt_f32_ubyteK)
template <int K> __device__ __forceinline__ float ubyte(uint32_t w) { return float((w >> (8 * K)) & 0xFFu); }

// One node step: the eight slab tests of a CW8 node.  Returns the MISS bits of the eight slots (bit s set = the ray misses the child in slot s; empty slots hold an
// inverted box and always miss).  Round 6: the per-child work ends at ONE fast-class subtraction and ONE v_alignbit that shifts the sign of (exit - entry) into the
// mask -- no compare, no select, no per-child shift; the octant order of the inner children and the triangle bits of the leaves come from two small LDS tables
// (lut_perm, lut_pair) looked up once per node step.  gfx950 issues fp32 FMA / MUL / ADD / SUB / MOV, v_bitop3, AND / OR / XOR, right shifts and integer add / sub in
// ~2.7 cycles per wave and everything else -- conversions, min / max, compares (!), v_cndmask, left shifts, v_bfe, v_or3 ... -- in ~4.4 (tools/micro/issue_model2.hip,
// profiles/r06_micro_issue_model2.txt): until round 5 a child cost 14 slow + 6 fast instructions, now 11 + 7.
struct NodeWords { uint4 a, b, c, d, e; };
template <int K>
__device__ __forceinline__ uint32_t child_miss(uint32_t miss, uint32_t lx, uint32_t ly, uint32_t lz, uint32_t hx, uint32_t hy, uint32_t hz, const f3 A, const f3 B, float tmin, float tlimit)
{
	const float tlx = __builtin_fmaf(ubyte<K>(lx), A.x, B.x), tly = __builtin_fmaf(ubyte<K>(ly), A.y, B.y), tlz = __builtin_fmaf(ubyte<K>(lz), A.z, B.z);
	const float thx = __builtin_fmaf(ubyte<K>(hx), A.x, B.x), thy = __builtin_fmaf(ubyte<K>(hy), A.y, B.y), thz = __builtin_fmaf(ubyte<K>(hz), A.z, B.z);
	const float tn = raw_max3(tlx, tly, raw_max(tlz, tmin));
	const float tf = raw_min3(thx, thy, raw_min(thz, tlimit));
	// hit <=> tn <= tf <=> the sign bit of tf - tn is clear (x - x = +0; a box that ends exactly where the interval begins with tf = -0, tn = +0 holds no point with
	// t > tmin >= 0 and may be missed; inf - inf = the positive quiet NaN: a hit, as inf <= inf is)
	return __builtin_amdgcn_alignbit(miss, as_u32(tf - tn), 31);          // (miss << 1) | sign
}
__device__ __forceinline__ uint32_t test_node(const NodeWords& n, const LaneRay& r, float tlimit, bool neg_x, bool neg_y, bool neg_z)
{
	// node-local grid -> ray parameter: t = q * A + B, A = 2^e / d, B = (p - o) / d
	const uint32_t ew = n.a.w;
	const f3 A = mk3(as_f32((ew & 0xFFu) << 23) * r.idir.x, as_f32(((ew >> 8) & 0xFFu) << 23) * r.idir.y, as_f32(((ew >> 16) & 0xFFu) << 23) * r.idir.z);
	const f3 B = mk3((as_f32(n.a.x) - r.o.x) * r.idir.x, (as_f32(n.a.y) - r.o.y) * r.idir.y, (as_f32(n.a.z) - r.o.z) * r.idir.z);
	uint32_t miss = 0;
	// slots 7 .. 0: the first sign shifted in ends up highest, so that bit s is slot s
	#pragma unroll
	for (int half = 1; half >= 0; --half)
	{
		// words of this group of four children: lo.xyz, hi.xyz
		const uint32_t qlx = half ? n.c.y : n.c.x, qly = half ? n.c.w : n.c.z, qlz = half ? n.d.y : n.d.x;
		const uint32_t qhx = half ? n.d.w : n.d.z, qhy = half ? n.e.y : n.e.x, qhz = half ? n.e.w : n.e.z;
		// entry / exit planes by direction sign
		const uint32_t lx = neg_x ? qhx : qlx, hx = neg_x ? qlx : qhx;
		const uint32_t ly = neg_y ? qhy : qly, hy = neg_y ? qly : qhy;
		const uint32_t lz = neg_z ? qhz : qlz, hz = neg_z ? qlz : qhz;
		miss = child_miss<3>(miss, lx, ly, lz, hx, hy, hz, A, B, r.tmin, tlimit);
		miss = child_miss<2>(miss, lx, ly, lz, hx, hy, hz, A, B, r.tmin, tlimit);
		miss = child_miss<1>(miss, lx, ly, lz, hx, hy, hz, A, B, r.tmin, tlimit);
		miss = child_miss<0>(miss, lx, ly, lz, hx, hy, hz, A, B, r.tmin, tlimit);
	}
	return miss;
}

// fpt-MT: fixed-order Moeller-Trumbore on a pre-transformed record; bu, bv weight vertices 1 and 2.  Evaluated without early
// exits: in a divergent wave some lane nearly always survives each test, so the exits save no VALU work and only cost exec-mask
// bookkeeping on the scalar unit; a rejected triangle's values are simply never used (det == 0 gives inf/NaN, which fail the
// comparisons exactly as the explicit test does).
__device__ __forceinline__ bool intersect_record(const float4 a, const float4 b, const float4 c, const LaneRay& r, float& t, float& bu, float& bv)
{
	const f3 v0 = mk3(a.x, a.y, a.z);
	const f3 e1 = mk3(a.w, b.x, b.y);
	const f3 e2 = mk3(b.z, b.w, c.x);
	// (round 6) two cross products instead of three, determinant and t from ONE normal: n = e1 x e2, c = s x d; det = e1 . (d x e2) = -(d . n), bu = s . (d x e2) / det = (e2 . c) / det,
	// bv = d . (s x e1) / det = -(e1 . c) / det, t = e2 . (s x e1) / det = (s . n) / det -- t is then the exact crossing with a plane through v0 tilted by n's rounding error, which on a
	// sliver moves it by 1e-5 of the triangle's size instead of 1e-5 of the ray's length (oracle/o_bvh.h intersect_tri)
	const f3 n = cross(e1, e2);
	const float det = 0.0f - dot(r.d, n);
	const float inv = 1.0f / det;
	const f3 s = r.o - v0;
	const f3 cc = cross(s, r.d);
	bu = dot(e2, cc) * inv;
	bv = (0.0f - dot(e1, cc)) * inv;
	t = dot(s, n) * inv;
	// the box clause (round 5; oracle/o_bvh.h intersect_tri has the reasoning): the point the ray reaches at t, relative to v0, must lie in the triangle's own box
	// [min(0, e1, e2), max(0, e1, e2)] widened by tol = c.w + 4e-7 (|y| + |t d|), c.w = 5e-7 (|triangle|max + |scene|max).  For a grazing ray (det -> 0) t is noise and
	// can land inside (tmin, tmax) when the true crossing does not; whether such a triangle is tested at all depends on the tree.  With the clause an accepted hit's
	// point lies inside the triangle's padded box, which every conservative traversal reaches.  27 fp32 MUL / ADD / compares of the cheap issue class + 6 min3 / max3.
	const f3 td = t * r.d;
	const f3 y = s + td;
	const float vpad = c.w;
	const float tolx = vpad + 4.0e-7f * (fabsf(y.x) + fabsf(td.x)), toly = vpad + 4.0e-7f * (fabsf(y.y) + fabsf(td.y)), tolz = vpad + 4.0e-7f * (fabsf(y.z) + fabsf(td.z));
	const int in_box = int(y.x >= raw_min3(0.0f, e1.x, e2.x) - tolx) & int(y.x <= raw_max3(0.0f, e1.x, e2.x) + tolx) &
	                   int(y.y >= raw_min3(0.0f, e1.y, e2.y) - toly) & int(y.y <= raw_max3(0.0f, e1.y, e2.y) + toly) &
	                   int(y.z >= raw_min3(0.0f, e1.z, e2.z) - tolz) & int(y.z <= raw_max3(0.0f, e1.z, e2.z) + tolz);
	return bool(int(det != 0.0f) & int(bu >= 0.0f) & int(bu <= 1.0f) & int(bv >= 0.0f) & int(bu + bv <= 1.0f) & int(t > r.tmin) & int(t < r.tmax) & in_box);
}

// stack pop: always a ds_read (clamped level), the scratch overflow only for the lanes that are that deep -- written this way so that
// the compiler does not merge the two address spaces into one flat_load, which would run every pop through the slower flat path
__device__ __forceinline__ uint2 pop_entry(uint2 (*lds_stack)[256], const uint2* ovf, int sp, uint32_t tid)
{
	typedef const volatile __attribute__((address_space(3))) uint32_t* lds_ptr;      // explicit LDS address space + volatile: stays a ds_read
	lds_ptr q = (lds_ptr)&lds_stack[sp < LDS_STACK ? sp : LDS_STACK - 1][tid];
	uint2 v = make_uint2(q[0], q[1]);
	if (__builtin_expect(sp >= LDS_STACK, 0)) v = ovf[sp - LDS_STACK];
	return v;
}

template <int MODE, bool COUNTED>
__global__ __launch_bounds__(TRACE_BLOCK, FPT_TRACE_MIN_WAVES)
void trace_kernel(const TraceParams P)
{
	__shared__ uint2 lds_stack[LDS_STACK][TRACE_BLOCK];
	__shared__ uint8_t  lut_perm[8 * 256];      // [7 - ray octant][inner hit byte in slot order] -> the byte in visiting order: slot s at bit (s ^ (7 - octant))
	__shared__ uint16_t lut_pair[256];          // [leaf hit byte] -> the slots' triangle PAIRS: bit s -> bits 2s, 2s + 1 (masked with the node's valid word afterwards)
	uint2 ovf[OVF_STACK];

	const uint32_t tid  = threadIdx.x;
	const uint32_t lane = tid & 63u;
	for (uint32_t i = tid; i < 8u * 256u; i += TRACE_BLOCK)
	{
		const uint32_t o = i >> 8;
		uint32_t v = 0;
		#pragma unroll
		for (uint32_t sl = 0; sl < 8; ++sl) v |= ((i >> sl) & 1u) << (sl ^ o);
		lut_perm[i] = uint8_t(v);
	}
	{
		uint32_t v = 0;
		#pragma unroll
		for (uint32_t sl = 0; sl < 8; ++sl) v |= ((tid >> sl) & 1u) * (3u << (2u * sl));
		lut_pair[tid] = uint16_t(v);
	}
	__syncthreads();
	// index space: [0, n_first) = the primary ray array (closest-hit rays, or the any-hit rays in MODE_ANY*),
	//              [n_first, n_rays) = the fused shadow queue (MODE_MIXED only)
	const uint32_t n_first = (MODE == MODE_ANY_FUSED) ? *P.shadow_size : (P.count_ptr ? *P.count_ptr : P.count);
	const uint32_t n_rays  = mode_is_mixed(MODE) ? n_first + *P.shadow_size : n_first;

	const uint32_t shard_size = (n_rays + TICKET_SHARDS - 1) / TICKET_SHARDS;
	const uint32_t total_waves = gridDim.x * (TRACE_BLOCK / 64);
	uint32_t chunk = ((n_rays / (total_waves * 2u)) + 63u) & ~63u;
	chunk = chunk < 64u ? 64u : (chunk > uint32_t(FPT_CHUNK_MAX) ? uint32_t(FPT_CHUNK_MAX) : chunk);
	const uint32_t wave_id = blockIdx.x * (TRACE_BLOCK / 64) + (tid >> 6);
	uint32_t shard = wave_id % TICKET_SHARDS;      // (tying the shard to the block's XCD, blockIdx % 8, was measured: 1536 vs 1554 Msample/s)
	uint32_t c_next = 0, c_end = 0;   // wave-uniform: the chunk being handed out
	// small queues (later bounces): every wave owns one fixed 64-ray batch, no atomics at all
	const bool static_batches = n_rays <= total_waves * 64u;
	if (static_batches) { c_next = wave_id * 64u; c_end = (c_next + 64u < n_rays) ? c_next + 64u : n_rays; if (c_next >= n_rays) { c_next = c_end = 0; } }

	bool     have = false;          // this lane owns a ray
	bool     dry  = false;          // wave-uniform: every shard is exhausted
	bool     any  = (MODE == MODE_ANY || MODE == MODE_ANY_FUSED || MODE == MODE_ANY_Q);     // this lane's ray is an any-hit (shadow) ray
	uint32_t ray_index = 0;
	LaneRay  r;
	uint32_t ray_mask = 0;
	uint2    grp = make_uint2(0u, 0u);      // current node group: .x = index of the first inner child, .y = hit bits (24..31) | imask (0..7)
	uint32_t oct_off = 0;                   // (7 - ray octant) << 8: the ray's row of lut_perm
	bool     neg_x = false, neg_y = false, neg_z = false;
	int      sp = 0;
	uint32_t tri_base = 0, tri_bits = 0;    // the triangle group in hand (persists over iterations): first record of its node; bits 0..15 the triangles still to test in
	                                        // the node's slot-pair layout (bit 2s + j = triangle j of the leaf in slot s), bits 16..31 the node's valid word (which of those exist)
	float    best_t = 0.0f, best_bu = 0.0f, best_bv = 0.0f;
	int32_t  best_id = -1;
	bool     occluded = false;
	unsigned long long cnt[6] = { 0, 0, 0, 0, 0, 0 };      // COUNTED: {nodes, tris, rays} for closest, then for any-hit rays

	for (;;)
	{
		// ---- refill idle lanes from the wave's current chunk ----
		const unsigned long long idle = __ballot(!have);
		const int n_idle = __popcll(idle);
		if (!dry && (n_idle == 64 || n_idle >= REFILL_MIN))
		{
			if (c_next >= c_end && static_batches) dry = true;
			else if (c_next >= c_end)
			{
				// draw a new chunk: one atomic per wave per CHUNK rays, on the shard this wave started on; steal from the others when dry
				uint32_t lo = 0, hi = 0;
				if (lane == 0)
				{
					for (uint32_t tried = 0; tried < TICKET_SHARDS; ++tried)
					{
						const uint32_t sb = shard_size * shard, se = (shard + 1 == TICKET_SHARDS) ? n_rays : shard_size * (shard + 1);
						// (guided self-scheduling -- chunks that shrink with the work left -- was measured and dropped: 1367 vs 1542 Msample/s; the
						//  extra atomics on the small chunks cost more than the shorter tail saves.  Round 3, on top of 256-ray chunks: draws of 64 / 128
						//  rays once a shard is nearly empty: 1487-1539 vs 1508-1556 in the driver's form, 1690 vs 1712 at 64 in flight: what is left of a
						//  launch's tail is its longest rays, not the hand-out)
						const uint32_t base = sb + atomicAdd(P.work_counter + shard * TICKET_PAD, chunk);
						if (base < se) { lo = base; hi = (base + chunk < se) ? base + chunk : se; break; }
						shard = (shard + 1 == TICKET_SHARDS) ? 0u : shard + 1;
					}
				}
				c_next = __shfl(lo, 0); c_end = __shfl(hi, 0); shard = __shfl(shard, 0);
				if (c_next >= c_end) dry = true;
			}
			if (!dry)
			{
				const uint32_t avail = c_end - c_next;
				const uint32_t rank = __popcll(idle & ((1ull << lane) - 1ull));
				if (!have && rank < avail)
				{
					const uint32_t i = c_next + rank;
					if (mode_is_mixed(MODE)) any = i >= n_first;
					const float4* src = (MODE == MODE_ANY_FUSED) ? P.shadow_rays + 2 * size_t(i)
					                  : (mode_is_mixed(MODE) && any) ? P.shadow_rays + 2 * size_t(i - n_first) : P.rays + 2 * size_t(i);
					const float4 ro = src[0];
					const float4 rd = src[1];
					r.o = mk3(ro.x, ro.y, ro.z);
					r.d = mk3(rd.x, rd.y, rd.z);
					r.idir = mk3(guarded_rcp(rd.x), guarded_rcp(rd.y), guarded_rcp(rd.z));
					neg_x = r.idir.x < 0.0f; neg_y = r.idir.y < 0.0f; neg_z = r.idir.z < 0.0f;
					oct_off = (7u - ((neg_x ? 4u : 0u) | (neg_y ? 2u : 0u) | (neg_z ? 1u : 0u))) << 8;
					ray_mask = as_u32(ro.w);
					// closest-hit trace reads .mask as tmin (src/pathtracer_kernels.h:343); the rays of a renderer's queues carry bookkeeping in the .w words and
					// have the same interval throughout a queue (fpt_device.h)
					const float q_tmin = (MODE == MODE_CLOSEST_QP) ? QUEUE_PRIMARY_TMIN : QUEUE_SCATTER_TMIN, q_tmax = (MODE == MODE_CLOSEST_QP) ? QUEUE_PRIMARY_TMAX : QUEUE_SCATTER_TMAX;
					r.tmin = any ? 0.0f : (closest_from_queue(MODE) ? q_tmin : ro.w);
					r.tmax = any ? (any_from_queue(MODE) ? QUEUE_SHADOW_TMAX : rd.w) : (closest_from_queue(MODE) ? q_tmax : rd.w);
					best_t = r.tmax; best_id = -1; best_bu = 0.0f; best_bv = 0.0f; occluded = false;
					ray_index = (mode_is_mixed(MODE) && any) ? i - n_first : i;
					grp = make_uint2(0u, 0x80000000u);           // the root: "child 0 of base 0", no siblings
					sp = 0; have = true; tri_bits = 0;
					if (COUNTED) cnt[any ? 5 : 2]++;
					// a ray with a non-finite origin or direction can hit nothing (every comparison of fpt-MT fails) but would walk the
					// whole tree, because NaN slab bounds cull nothing: give it an empty interval instead
					if (!(all_finite(r.o) && all_finite(r.d))) { r.tmin = 1.0f; r.tmax = 0.0f; best_t = 0.0f; }
				}
				c_next += (uint32_t(n_idle) < avail) ? uint32_t(n_idle) : avail;
			}
		}
		if (!__any(have)) break;

		// ---- traversal burst: wave-uniform loop, idle lanes are predicated off inside ----
		for (;;)
		{
			if (have)
			{
				bool alive = true;
				// ---- node step: take the nearest hit child of the current group, leave its siblings on the stack ----
				if (grp.y & 0xFF000000u)
				{
					const uint32_t bit = 31u - uint32_t(__builtin_clz(grp.y));
					const uint32_t rest = grp.y & ~(1u << bit);
					if (rest & 0xFF000000u)
					{
						const uint2 e = make_uint2(grp.x, rest);
						if (sp < LDS_STACK) lds_stack[sp][tid] = e; else ovf[sp - LDS_STACK] = e;
						sp++;
					}
					const uint32_t slot = (bit - 24u) ^ (oct_off >> 8);
					const uint32_t rel = uint32_t(__builtin_popcount(grp.y & ~(0xFFFFFFFFu << slot) & 0xFFu));
					const uint4* np = P.bvh.nodes + 5 * size_t(grp.x + rel);          // 80-byte nodes
					NodeWords n; n.a = np[0]; n.b = np[1]; n.c = np[2]; n.d = np[3]; n.e = np[4];
					if (COUNTED) cnt[any ? 3 : 0]++;
					const uint32_t miss = test_node(n, r, best_t, neg_x, neg_y, neg_z);
					const uint32_t imask = n.a.w >> 24;
					const uint32_t inner_hits = ~miss & imask, leaf_hits = ~miss & ~imask & 0xFFu;
					grp = make_uint2(n.b.x, (uint32_t(lut_perm[oct_off | inner_hits]) << 24) | imask);
					const uint32_t tris = uint32_t(lut_pair[leaf_hits]) & n.b.z;          // n.b.z: the valid word in bits 0..15, zero above
					// (touching the next node here -- a load nothing waits for, so that its lines travel during the triangle test -- was measured: 1490-1499 vs
					//  1536-1567 Msample/s in the driver's form, no change in the one-pass mode: a step of a lone wave is not waiting for that line)
					if (tris)
					{
						// new (nearer) triangles: they go first; an older group still in hand is parked on the stack
						if (tri_bits & 0xFFFFu)
						{
							const uint2 e = make_uint2(tri_base, tri_bits);
							if (sp < LDS_STACK) lds_stack[sp][tid] = e; else ovf[sp - LDS_STACK] = e;
							sp++;
						}
						tri_base = n.b.y; tri_bits = tris | (n.b.z << 16);
					}
				}
				// ---- ONE triangle of the group in hand ----
				if (tri_bits & 0xFFFFu)
				{
					// the lowest pending bit; its record is the node's first + the number of EXISTING triangles below it (the records of a node are packed in slot order)
					const uint32_t k = uint32_t(__builtin_ctz(tri_bits));
					const uint32_t rank = uint32_t(__builtin_popcount((tri_bits >> 16) & ~(0xFFFFFFFFu << k)));
					tri_bits &= tri_bits - 1u;
					const float4* tp = P.bvh.tris + 3 * size_t(tri_base + rank);
					const float4 a = tp[0], b = tp[1], c = tp[2];
					const bool skip = any && (ray_mask & as_u32(c.z));
					if (COUNTED) cnt[any ? 4 : 1] += skip ? 0u : 1u;
					float t, bu, bv;
					const bool hit = intersect_record(a, b, c, r, t, bu, bv) && !skip;
					const int32_t id = int32_t(as_u32(c.y));
					const bool better = bool(int(hit) & int(!any) & (int(best_id < 0) | int(t < best_t) | (int(t == best_t) & int(id < best_id))));
					best_t = better ? t : best_t; best_id = better ? id : best_id; best_bu = better ? bu : best_bu; best_bv = better ? bv : best_bv;
					occluded = occluded || (hit && any);
				}
				// ---- the next entry of the stack.  Nothing in hand: whatever is on top (a node group or a parked triangle group).  Triangles still in hand
				//      but no node group: a node group on top is taken NOW, so that the next iteration's node step has work while the triangles are tested --
				//      results do not depend on the order, and a wave pays for both halves of an iteration anyway (the CPU model of tools/bvh_walk.cpp: 6.6 % /
				//      8.9 % fewer wave instructions on the two bench scenes; measured: traversal -3.0 % / -3.7 %.  Also taking parked triangles while only
				//      nodes are in hand adds nothing: 1737 vs 1732 Msample/s) ----
				if (any && occluded) alive = false;
				else if (!(grp.y & 0xFF000000u))
				{
					if (sp == 0) alive = (tri_bits & 0xFFFFu) != 0u;
					else
					{
						const uint2 e = pop_entry(lds_stack, ovf, sp - 1, tid);
						// a node group holds nothing in bits 8..23; a parked triangle group always does (pending bits in 8..15 or valid bits in 16..23: a group whose
						// valid bits all sit in 24..31 has its pending bits in 8..15, and only groups with something pending are parked)
						const bool is_grp = (e.y & 0x00FFFF00u) == 0u;
						if (is_grp) { grp = e; sp--; }
						else if (!(tri_bits & 0xFFFFu)) { tri_base = e.x; tri_bits = e.y; sp--; }
					}
				}
				if (!alive)
				{
					// ---- retire the ray ----
					if (any)
					{
						if (MODE == MODE_MIXED_PSF)
						{
							// PSFPTVertexProcessor::accumulate_nee fused: the sample goes to its cache cell and / or the frame
							if (!occluded) psf_resolve_sample(*reinterpret_cast<const ResolveParams*>(P.fused), P.base_instance, ray_index);
						}
						else if (MODE == MODE_MIXED_HITS)
						{
							float4* shadow_hits = reinterpret_cast<float4*>(const_cast<FusedResolve*>(P.fused));
							shadow_hits[ray_index] = occluded ? make_float4(1.0f, as_f32(1u), 0.0f, 0.0f) : make_float4(-1.0f, as_f32(0xFFFFFFFFu), 0.0f, 0.0f);
						}
						else if (MODE == MODE_ANY_FUSED || MODE == MODE_MIXED)
						{
							// solve_occlusion (src/pathtracer_kernels.h:248-280) fused: accumulate the light sample when unoccluded
							if (!occluded)
							{
								const FusedResolve* F = P.fused;
								const float4 wd = F->w_d[ray_index], wg = F->w_g[ray_index];
								const uint32_t pixel_info = as_u32(P.shadow_rays[2 * size_t(ray_index) + 1].w);          // ShadowQueue: dir | PixelInfo, w_d.w = pass offset
								PassInfo ps = F->pass; ps.base_instance = P.base_instance;
								accumulate_nee(F->fb, ps, F->log, F->kind, pixel_info, ps.n_passes > 1 ? as_u32(wd.w) : 0u, F->bounce, mk3(wd.x, wd.y, wd.z), mk3(wg.x, wg.y, wg.z));
							}
						}
						else
						{
							if (P.hits) P.hits[ray_index] = occluded ? make_float4(1.0f, as_f32(1u), 0.0f, 0.0f) : make_float4(-1.0f, as_f32(0xFFFFFFFFu), 0.0f, 0.0f);
							if (P.bits && occluded) atomicOr(P.bits + (ray_index >> 5), 1u << (ray_index & 31u));
						}
					}
					else
					{
						float4 h = make_float4(-1.0f, as_f32(0xFFFFFFFFu), 0.0f, 0.0f);
						if (best_id >= 0)
						{
							const float u = 1.0f - best_bu - best_bv;        // weight of vertex 0 (optix_base_shaders.h:50-57)
							h = make_float4(best_t, as_f32(uint32_t(best_id)), round_through_half(u), round_through_half(best_bu));
						}
						P.hits[ray_index] = h;
					}
					have = false;
				}
			}
			// every lane of the wave reaches this point: decide (uniformly) whether to keep traversing or go refill
			const int n_busy = __popcll(__ballot(have));
			if (n_busy == 0) break;
			if (!dry && (64 - n_busy) >= REFILL_MIN) break;
		}
	}
	if (COUNTED)
	{
		// wave-level reduction, then one atomic per counter per wave
		#pragma unroll
		for (int k = 0; k < 6; ++k)
		{
			unsigned long long v = cnt[k];
			for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
			if (lane == 0 && v) atomicAdd(P.stats + (k < 3 ? k : k + 1), v);       // closest -> stats[0..2], any-hit -> stats[4..6]
		}
	}
}

template <int MODE>
static void launch_mode(const TraceParams& p, bool counted, uint32_t n_blocks, hipStream_t stream)
{
	if (counted) hipLaunchKernelGGL((trace_kernel<MODE, true>), dim3(n_blocks), dim3(TRACE_BLOCK), 0, stream, p);
	else         hipLaunchKernelGGL((trace_kernel<MODE, false>), dim3(n_blocks), dim3(TRACE_BLOCK), 0, stream, p);
}

uint32_t trace_blocks_per_cu() { return FPT_TRACE_MIN_WAVES; }
uint32_t trace_stack_entries() { return uint32_t(LDS_STACK + OVF_STACK); }
void launch_trace_closest(const TraceParams& p, bool counted, uint32_t n_blocks, hipStream_t stream) { launch_mode<MODE_CLOSEST>(p, counted, n_blocks, stream); }
void launch_trace_shadow(const TraceParams& p, bool fused_resolve, bool counted, uint32_t n_blocks, hipStream_t stream)
{
	if (fused_resolve) launch_mode<MODE_ANY_FUSED>(p, counted, n_blocks, stream);
	else               launch_mode<MODE_ANY>(p, counted, n_blocks, stream);
}
void launch_trace_closest_queue(const TraceParams& p, bool primary, bool counted, uint32_t n_blocks, hipStream_t stream)
{
	if (primary) launch_mode<MODE_CLOSEST_QP>(p, counted, n_blocks, stream);
	else         launch_mode<MODE_CLOSEST_QS>(p, counted, n_blocks, stream);
}
void launch_trace_shadow_queue(const TraceParams& p, bool counted, uint32_t n_blocks, hipStream_t stream) { launch_mode<MODE_ANY_Q>(p, counted, n_blocks, stream); }
void launch_trace_mixed(const TraceParams& p, bool counted, uint32_t n_blocks, hipStream_t stream) { launch_mode<MODE_MIXED>(p, counted, n_blocks, stream); }
void launch_trace_mixed_psf(const TraceParams& p, bool counted, uint32_t n_blocks, hipStream_t stream) { launch_mode<MODE_MIXED_PSF>(p, counted, n_blocks, stream); }
void launch_trace_mixed_hits(const TraceParams& p, float4* shadow_hits, bool counted, uint32_t n_blocks, hipStream_t stream)
{ TraceParams q = p; q.fused = reinterpret_cast<const FusedResolve*>(shadow_hits); launch_mode<MODE_MIXED_HITS>(q, counted, n_blocks, stream); }

} // namespace fpt
