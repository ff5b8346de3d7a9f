// fpt_trace.hip — hand-written gfx950 BVH2 traversal kernels: the replacement for OptiX behind RTContext::trace /
// trace_shadow (src/rt.cpp:558-659, src/kernels/optix_rt.cu:45-82,133-204, optix_base_shaders.h:42-91,
// optix_base_shadow_shaders.h:42-72).
//
// CDNA4 design:
//   * persistent waves: the grid is sized to the chip (CUs x resident blocks) and every 64-lane wave pulls work itself.
//     Work distribution is built around one measured fact: a single device-scope counter sustains only ~90 atomics/us on
//     MI355X (and atomics to one 128-B line serialise chip-wide).  So (i) the ray range is cut into 8 shards whose ticket
//     counters sit on separate cache lines, (ii) a wave draws a CHUNK of rays per atomic and feeds its lanes from it,
//     refilling idle lanes by ballot/popcount rank without further atomics, (iii) queues no larger than one 64-ray batch per
//     wave (the late bounces) are assigned statically with no atomics at all;
//   * per-lane traversal stack in LDS, laid out [level][thread] so a wave's accesses are conflict-free whatever the per-lane
//     depth; entries beyond LDS_STACK spill to a private scratch array (rare);
//   * 32-byte nodes (both children's boxes on a 16-bit scene grid + both child references, fpt_bvh.h) fetched with two 16-byte
//     loads; the slab test runs directly on grid coordinates (t = q*A + B with per-ray A = step/d, B = (base - o)/d), so decoding
//     costs twelve SDWA conversions and no extra arithmetic; near child first, far child pushed.  Against 64-byte fp32 nodes:
//     +1.6 % samples/s, 40 % less HBM traffic, half the footprint (the kernel is latency- and VALU-bound, not request-bound);
//   * slab tests use FMAs (conservative: boxes are padded and snapped outward on the host); the triangle test is the fixed-order "fpt-MT"
//     Moeller-Trumbore whose results must equal the CPU oracle bit for bit (no FMA contraction, IEEE divide);
//   * closest hit = minimum t, ties -> lowest triangle id; barycentrics rounded through fp16 like OptiX's payload
//     (src/kernels/optix_payload.h:75-78); any-hit honours the per-triangle shadow mask (optix_base_shadow_shaders.h:54-59);
//   * MIXED mode: one launch serves the closest-hit rays of bounce b+1 AND the shadow rays of bounce b (fused with
//     solve_occlusion).  A launch cannot end before its longest ray (~100 dependent fetches ~ 0.1 ms), so halving the
//     number of launches per pass halves the number of such tails.
// No MFMA: this is a latency/bandwidth-bound pointer chase, not a contraction.  An LDS-resident copy of the top of the tree
// was measured and dropped: those nodes already hit in L1, while the LDS it takes costs occupancy.
#include "fpt_device.h"
#include "fpt_psf.h"

namespace fpt {

#ifndef FPT_LDS_STACK
#define FPT_LDS_STACK 16
#endif
#ifndef FPT_TRACE_MIN_WAVES
#define FPT_TRACE_MIN_WAVES 8      // 64 VGPRs, 9 of them spilled: with 32-byte nodes and the branch-free leaf loop full occupancy wins again (8: 0.623, 7: 0.643, 6: 0.687 ms/pass)
#endif
#ifndef FPT_REFILL_MIN
#define FPT_REFILL_MIN 32
#endif
static constexpr int TRACE_BLOCK = 256;
static constexpr int LDS_STACK   = FPT_LDS_STACK;        // levels x 256 threads x 4 B of LDS per block
static constexpr int OVF_STACK   = 64 - FPT_LDS_STACK;   // scratch overflow: total depth 64
static constexpr int REFILL_MIN  = FPT_REFILL_MIN;       // refill a wave once this many lanes are idle
static constexpr uint32_t TICKET_SHARDS = 8;             // one ticket counter per XCD-sized share of the waves
static constexpr uint32_t TICKET_PAD    = 32;            // counters sit 128 B apart: atomics on one cache line serialise chip-wide

enum TraceMode { MODE_CLOSEST = 0, MODE_ANY = 1, MODE_ANY_FUSED = 2, MODE_MIXED = 3, MODE_MIXED_PSF = 4 };    // MIXED_PSF: MIXED with the path-space-filtering resolve (`fused` points to a ResolveParams)

struct LaneRay
{
	f3 o, d;
	f3 A, B;             // slab test on grid coordinates: t = q*A + B, A = grid_step/d, B = (grid_base - o)/d (guarded reciprocal)
	float tmin, tmax;
};

__device__ __forceinline__ float guarded_rcp(float d)
{
	const float a = fabsf(d);
	const float g = (a < 1.0e-20f) ? (d < 0.0f ? -1.0e-20f : 1.0e-20f) : d;
	return 1.0f / g;
}

// both-children slab test on a 32-byte quantised node (fpt_bvh.h BvhNode32); returns hit flags and entry distances
__device__ __forceinline__ float q_lo16(uint32_t w) { return float(w & 0xFFFFu); }
__device__ __forceinline__ float q_hi16(uint32_t w) { return float(w >> 16); }
// v_max_f32 / v_min_f32 on operands known to be ordinary numbers: spelled as instructions so that the compiler does not re-quiet the
// loop-invariant interval ends (a v_max_f32 x, x each) on every node step
__device__ __forceinline__ float raw_max(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float raw_min(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ void test_children(const uint4 w0, const uint4 w1, const LaneRay& r, float tlimit,
                                              bool& h0, float& t0, bool& h1, float& t1)
{
	// q[]: lo0.x lo0.y | lo0.z hi0.x | hi0.y hi0.z | lo1.x lo1.y || lo1.z hi1.x | hi1.y hi1.z | child0 | child1
	{
		const float ax = __builtin_fmaf(q_lo16(w0.x), r.A.x, r.B.x), bx = __builtin_fmaf(q_hi16(w0.y), r.A.x, r.B.x);
		const float ay = __builtin_fmaf(q_hi16(w0.x), r.A.y, r.B.y), by = __builtin_fmaf(q_lo16(w0.z), r.A.y, r.B.y);
		const float az = __builtin_fmaf(q_lo16(w0.y), r.A.z, r.B.z), bz = __builtin_fmaf(q_hi16(w0.z), r.A.z, r.B.z);
		const float tn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), raw_max(fminf(az, bz), r.tmin));
		const float tf = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), raw_min(fmaxf(az, bz), tlimit));
		h0 = tn <= tf; t0 = tn;
	}
	{
		const float ax = __builtin_fmaf(q_lo16(w0.w), r.A.x, r.B.x), bx = __builtin_fmaf(q_hi16(w1.x), r.A.x, r.B.x);
		const float ay = __builtin_fmaf(q_hi16(w0.w), r.A.y, r.B.y), by = __builtin_fmaf(q_lo16(w1.y), r.A.y, r.B.y);
		const float az = __builtin_fmaf(q_lo16(w1.x), r.A.z, r.B.z), bz = __builtin_fmaf(q_hi16(w1.y), r.A.z, r.B.z);
		const float tn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), raw_max(fminf(az, bz), r.tmin));
		const float tf = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), raw_min(fmaxf(az, bz), tlimit));
		h1 = tn <= tf; t1 = tn;
	}
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
	const f3 p = cross(r.d, e2);
	const float det = dot(e1, p);
	const float inv = 1.0f / det;
	const f3 s = r.o - v0;
	bu = dot(s, p) * inv;
	const f3 q = cross(s, e1);
	bv = dot(r.d, q) * inv;
	t = dot(e2, q) * inv;
	return bool(int(det != 0.0f) & int(bu >= 0.0f) & int(bu <= 1.0f) & int(bv >= 0.0f) & int(bu + bv <= 1.0f) & int(t > r.tmin) & int(t < r.tmax));
}

// stack pop: always a ds_read (clamped level), the scratch overflow only for the lanes that are that deep -- written this way so that
// the compiler does not merge the two address spaces into one flat_load, which would run every pop through the slower flat path
__device__ __forceinline__ int32_t pop_entry(uint32_t (*lds_stack)[256], const uint32_t* ovf, int sp, uint32_t tid)
{
	typedef const volatile __attribute__((address_space(3))) uint32_t* lds_ptr;      // explicit LDS address space + volatile: stays a ds_read
	uint32_t v = *(lds_ptr)&lds_stack[sp < LDS_STACK ? sp : LDS_STACK - 1][tid];
	if (__builtin_expect(sp >= LDS_STACK, 0)) v = ovf[sp - LDS_STACK];
	return int32_t(v);
}

template <int MODE, bool COUNTED>
__global__ __launch_bounds__(TRACE_BLOCK, FPT_TRACE_MIN_WAVES)
void trace_kernel(const TraceParams P)
{
	__shared__ uint32_t lds_stack[LDS_STACK][TRACE_BLOCK];
	uint32_t ovf[OVF_STACK];

	const uint32_t tid  = threadIdx.x;
	const uint32_t lane = tid & 63u;
	// index space: [0, n_first) = the primary ray array (closest-hit rays, or the any-hit rays in MODE_ANY*),
	//              [n_first, n_rays) = the fused shadow queue (MODE_MIXED only)
	const uint32_t n_first = (MODE == MODE_ANY_FUSED) ? *P.shadow_size : (P.count_ptr ? *P.count_ptr : P.count);
	const uint32_t n_rays  = (MODE == MODE_MIXED || MODE == MODE_MIXED_PSF) ? n_first + *P.shadow_size : n_first;

	const uint32_t shard_size = (n_rays + TICKET_SHARDS - 1) / TICKET_SHARDS;
	const uint32_t total_waves = gridDim.x * (TRACE_BLOCK / 64);
	uint32_t chunk = ((n_rays / (total_waves * 2u)) + 63u) & ~63u;
	chunk = chunk < 64u ? 64u : (chunk > 1024u ? 1024u : chunk);
	const uint32_t wave_id = blockIdx.x * (TRACE_BLOCK / 64) + (tid >> 6);
	uint32_t shard = wave_id % TICKET_SHARDS;
	uint32_t c_next = 0, c_end = 0;   // wave-uniform: the chunk being handed out
	// small queues (later bounces): every wave owns one fixed 64-ray batch, no atomics at all
	const bool static_batches = n_rays <= total_waves * 64u;
	if (static_batches) { c_next = wave_id * 64u; c_end = (c_next + 64u < n_rays) ? c_next + 64u : n_rays; if (c_next >= n_rays) { c_next = c_end = 0; } }

	bool     have = false;          // this lane owns a ray
	bool     dry  = false;          // wave-uniform: every shard is exhausted
	bool     any  = (MODE == MODE_ANY || MODE == MODE_ANY_FUSED);     // this lane's ray is an any-hit (shadow) ray
	uint32_t ray_index = 0;
	LaneRay  r;
	uint32_t ray_mask = 0;
	int32_t  cur = 0;               // node reference being visited
	int      sp = 0;
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
					if (MODE == MODE_MIXED || MODE == MODE_MIXED_PSF) any = i >= n_first;
					const float4* src = (MODE == MODE_ANY_FUSED) ? P.shadow_rays + 2 * size_t(i)
					                  : ((MODE == MODE_MIXED || MODE == MODE_MIXED_PSF) && any) ? P.shadow_rays + 2 * size_t(i - n_first) : P.rays + 2 * size_t(i);
					const float4 ro = src[0];
					const float4 rd = src[1];
					r.o = mk3(ro.x, ro.y, ro.z);
					r.d = mk3(rd.x, rd.y, rd.z);
					{
						const float ix = guarded_rcp(rd.x), iy = guarded_rcp(rd.y), iz = guarded_rcp(rd.z);
						r.A = mk3(P.bvh.grid_step[0] * ix, P.bvh.grid_step[1] * iy, P.bvh.grid_step[2] * iz);
						r.B = mk3(__builtin_fmaf(P.bvh.grid_base[0], ix, -(ro.x * ix)), __builtin_fmaf(P.bvh.grid_base[1], iy, -(ro.y * iy)),
						          __builtin_fmaf(P.bvh.grid_base[2], iz, -(ro.z * iz)));
					}
					ray_mask = as_u32(ro.w);
					r.tmin = any ? 0.0f : ro.w;                  // closest-hit trace reads .mask as tmin (src/pathtracer_kernels.h:343)
					r.tmax = rd.w;
					best_t = rd.w; best_id = -1; best_bu = 0.0f; best_bv = 0.0f; occluded = false;
					ray_index = ((MODE == MODE_MIXED || MODE == MODE_MIXED_PSF) && any) ? i - n_first : i;
					cur = 0; sp = 0; have = true;
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
				// descend through inner nodes
				while (alive && cur >= 0)
				{
					const uint4* np = P.bvh.nodes + 2 * size_t(cur);
					const uint4 w0 = np[0], w1 = np[1];
					if (COUNTED) cnt[any ? 3 : 0]++;
					bool h0, h1; float t0, t1;
					test_children(w0, w1, r, best_t, h0, t0, h1, t1);
					const int32_t c0 = int32_t(w1.z), c1 = int32_t(w1.w);
					if (h0 && h1)
					{
						const bool first0 = t0 <= t1;
						const int32_t nearc = first0 ? c0 : c1, farc = first0 ? c1 : c0;
						if (sp < LDS_STACK) lds_stack[sp][tid] = uint32_t(farc); else ovf[sp - LDS_STACK] = uint32_t(farc);
						sp++;
						cur = nearc;
					}
					else if (h0) cur = c0;
					else if (h1) cur = c1;
					else
					{
						if (sp == 0) alive = false;
						else { sp--; cur = pop_entry(lds_stack, ovf, sp, tid); }
					}
				}
				// leaf
				if (alive)
				{
					const uint32_t ref = uint32_t(~cur);
					const uint32_t first = ref >> 3, n_tri = ref & 7u;
					for (uint32_t k = 0; k < n_tri; ++k)
					{
						const float4* tp = P.bvh.tris + 3 * size_t(first + k);
						const float4 a = tp[0], b = tp[1], c = tp[2];
						const bool skip = any && (ray_mask & as_u32(c.z));
						if (COUNTED) cnt[any ? 4 : 1] += skip ? 0u : 1u;
						float t, bu, bv;
						const bool hit = intersect_record(a, b, c, r, t, bu, bv) && !skip;
						const int32_t id = int32_t(as_u32(c.y));
						const bool better = bool(int(hit) & int(!any) & (int(best_id < 0) | int(t < best_t) | (int(t == best_t) & int(id < best_id))));
						best_t = better ? t : best_t; best_id = better ? id : best_id; best_bu = better ? bu : best_bu; best_bv = better ? bv : best_bv;
						occluded = occluded || (hit && any);
						if (occluded) break;
					}
					if ((any && occluded) || sp == 0) alive = false;
					else { sp--; cur = pop_entry(lds_stack, ovf, sp, tid); }
				}
				if (!alive)
				{
					// ---- retire the ray ----
					if (any)
					{
						if (MODE == MODE_MIXED_PSF)
						{
							// PSFPTVertexProcessor::accumulate_nee fused: the sample goes to its cache cell and / or the frame
							if (!occluded) psf_resolve_sample(*reinterpret_cast<const ResolveParams*>(P.fused), 1.0f / float(P.base_instance + 1), ray_index);
						}
						else if (MODE == MODE_ANY_FUSED || MODE == MODE_MIXED)
						{
							// solve_occlusion (src/pathtracer_kernels.h:248-280) fused: accumulate the light sample when unoccluded
							if (!occluded)
							{
								const FusedResolve* F = P.fused;
								const float4 wd = F->w_d[ray_index], wg = F->w_g[ray_index];
								PassInfo ps = F->pass; ps.base_instance = P.base_instance;
								accumulate_nee(F->fb, ps, F->pixels[ray_index], F->bounce, mk3(wd.x, wd.y, wd.z), mk3(wg.x, wg.y, wg.z));
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

// resident 256-thread blocks per CU = the waves/SIMD the kernels are compiled for: the size of the persistent grid
uint32_t trace_blocks_per_cu() { return FPT_TRACE_MIN_WAVES; }
void launch_trace_closest(const TraceParams& p, bool counted, uint32_t n_blocks, hipStream_t stream) { launch_mode<MODE_CLOSEST>(p, counted, n_blocks, stream); }
void launch_trace_shadow(const TraceParams& p, bool fused_resolve, bool counted, uint32_t n_blocks, hipStream_t stream)
{
	if (fused_resolve) launch_mode<MODE_ANY_FUSED>(p, counted, n_blocks, stream);
	else               launch_mode<MODE_ANY>(p, counted, n_blocks, stream);
}
void launch_trace_mixed(const TraceParams& p, bool counted, uint32_t n_blocks, hipStream_t stream) { launch_mode<MODE_MIXED>(p, counted, n_blocks, stream); }
void launch_trace_mixed_psf(const TraceParams& p, bool counted, uint32_t n_blocks, hipStream_t stream) { launch_mode<MODE_MIXED_PSF>(p, counted, n_blocks, stream); }

} // namespace fpt
