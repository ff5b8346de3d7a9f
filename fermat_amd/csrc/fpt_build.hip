// fpt_build.hip — device-side maintenance of the acceleration structure (round 6): the REFIT behind fpt_rt_refit_geometry / RenderingContext::update_model(refit),
// i.e. what OptiX's Trbvh refit does on the GPU in the reference (src/rt.cpp:284-331 builds there; src/renderer.cu:999-1017 update_model hands DEVICE pointers over).
// Until round 5 the mesh went to the host, fpt_bvh.cpp refit_wide8 ran on CPU threads and the tree came back over PCIe (0.02 s + two copies of 130 MB).
//
// Three kernels, all on the context's stream, no host memory touched but one 8-byte read-back of {|scene|max, error bits}:
//   refit_scan_kernel     |scene|max over the vertices (the tolerance of the intersector's box clause and the leaf padding scale with it) and validation of every
//                         record's triangle id and vertex indices BEFORE anything is written (a refused refit leaves the tree as it was);
//   refit_records_kernel  the 48-byte triangle records {v0, e1, e2, id, mask, delta} and each triangle's padded box, one thread per record;
//   refit_level_kernel    one launch per level of the tree, deepest first (wide nodes are numbered breadth-first: a level is a contiguous range and a node's children
//                         come behind it): a node's box = union of its children's exact boxes, then origin, per-axis power-of-two cell and the eight children's boxes
//                         snapped outward onto the 8-bit grid, one thread per node.
// The arithmetic is fpt_bvh.cpp refit_wide8's, operation for operation (fp32 min / max that ignore NaNs the way std::min / std::max do, the grid in double, the same
// outward-snapping loops): the device tree equals the host refit byte for byte (tests/test_gpu_parity.py::test_device_refit_equals_the_host_refit).  HBM-bound streaming
// work: 1.82 M records x (48 B written + 16 B indices + 3 x 16 B vertices gathered + 24 B box) + 0.19 M nodes x (80 B read + 80 B written + children's boxes).
#include "fpt_device.h"
#include "fpt_bvh.h"

namespace fpt {

struct RefitBox { float lo[3], hi[3]; };

// std::min / std::max as the host builder applies them: (b < a) ? b : a and (a < b) ? b : a -- a NaN second operand is ignored
__device__ __forceinline__ float host_min(float a, float b) { return (b < a) ? b : a; }
__device__ __forceinline__ float host_max(float a, float b) { return (a < b) ? b : a; }

// out[0]: bits of |scene|max (non-negative floats order like their bit patterns), out[1]: error bits (1 record's triangle id out of range, 2 vertex index out of range)
__global__ __launch_bounds__(256) void refit_scan_kernel(uint32_t n_tris, const int4* __restrict__ idx, uint32_t n_verts, const float4* __restrict__ vtx,
                                                        uint32_t n_records, const BvhTriangle* __restrict__ records, uint32_t* __restrict__ out)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	float m = 0.0f;
	if (i < n_verts) { const float4 v = vtx[i]; m = host_max(host_max(host_max(m, fabsf(v.x)), fabsf(v.y)), fabsf(v.z)); }
	uint32_t err = 0;
	if (i < n_records && n_tris)
	{
		const uint32_t tri = uint32_t(records[i].tri_id);
		if (tri >= n_tris) err = 1u;
		else { const int4 ix = idx[tri]; if (ix.x < 0 || uint32_t(ix.x) >= n_verts || ix.y < 0 || uint32_t(ix.y) >= n_verts || ix.z < 0 || uint32_t(ix.z) >= n_verts) err = 2u; }
	}
	for (int off = 32; off > 0; off >>= 1) { m = host_max(m, __shfl_down(m, off)); err |= __shfl_down(err, off); }
	if ((threadIdx.x & 63u) == 0u) { if (m > 0.0f) atomicMax(out, as_u32(m)); if (err) atomicOr(out + 1, err); }
}

__global__ __launch_bounds__(256) void refit_records_kernel(uint32_t n_records, BvhTriangle* __restrict__ records, const int4* __restrict__ idx, const float4* __restrict__ vtx,
                                                           const uint32_t* __restrict__ scan, RefitBox* __restrict__ tri_box)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_records) return;
	const float scene_mag = as_f32(scan[0]);
	BvhTriangle r = records[i];
	const int4 ix = idx[uint32_t(r.tri_id)];
	const float4 q0 = vtx[ix.x], q1 = vtx[ix.y], q2 = vtx[ix.z];
	const float p[3][3] = { { q0.x, q0.y, q0.z }, { q1.x, q1.y, q1.z }, { q2.x, q2.y, q2.z } };
	RefitBox bx; float m0 = 0.0f;
	#pragma unroll
	for (int k = 0; k < 3; ++k) { bx.lo[k] = 3.0e38f; bx.hi[k] = -3.0e38f; }
	#pragma unroll
	for (int c = 0; c < 3; ++c)
		#pragma unroll
		for (int k = 0; k < 3; ++k) { bx.lo[k] = host_min(bx.lo[k], p[c][k]); bx.hi[k] = host_max(bx.hi[k], p[c][k]); }
	#pragma unroll
	for (int c = 0; c < 3; ++c)
		#pragma unroll
		for (int k = 0; k < 3; ++k) m0 = host_max(m0, fabsf(p[c][k]));
	const float pad = (m0 + scene_mag) * 2.0e-6f + 1.0e-30f;          // fpt_bvh.cpp build_bvh2 / refit_wide8: four times the box clause's constant tolerance
	#pragma unroll
	for (int k = 0; k < 3; ++k) { bx.lo[k] -= pad; bx.hi[k] += pad; }
	tri_box[i] = bx;
	#pragma unroll
	for (int k = 0; k < 3; ++k) { r.v0[k] = p[0][k]; r.e1[k] = p[1][k] - p[0][k]; r.e2[k] = p[2][k] - p[0][k]; }
	r.mask = uint32_t(ix.w);
	// triangle_vpad (fpt_bvh.cpp): max over the components of max(|p0|, max(|p1|, |p2|)), then 5e-7 (that + |scene|max)
	float mv = 0.0f;
	#pragma unroll
	for (int k = 0; k < 3; ++k) mv = host_max(mv, host_max(fabsf(p[0][k]), host_max(fabsf(p[1][k]), fabsf(p[2][k]))));
	r.vpad = (mv + scene_mag) * 5.0e-7f;
	records[i] = r;
}

// the smallest power-of-two cell that spans `ext` in 255 steps, as an exponent in [-100, 120] (fpt_bvh.cpp build_wide8 / refit_wide8: the loops make the answer
// independent of log2's rounding)
__device__ __forceinline__ int grid_exponent(double ext)
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

__global__ __launch_bounds__(128) void refit_level_kernel(BvhNode8* __restrict__ nodes, RefitBox* __restrict__ node_box, const RefitBox* __restrict__ tri_box,
                                                         uint32_t begin, uint32_t count, uint32_t* __restrict__ scan)
{
	const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= count) return;
	const uint32_t n = begin + t;
	BvhNode8 node = nodes[n];
	const uint32_t imask = node.w[3] >> 24, valid = node.w[6], child_base = node.w[4], tri_base = node.w[5];
	RefitBox cb[8]; RefitBox nb;
	#pragma unroll
	for (int k = 0; k < 3; ++k) { nb.lo[k] = 3.0e38f; nb.hi[k] = -3.0e38f; }
	uint32_t used = 0;
	#pragma unroll
	for (int sl = 0; sl < 8; ++sl)
	{
		const uint32_t pair = (valid >> (2 * sl)) & 3u;
		const bool inner = (imask >> sl) & 1u;
		if (!inner && !pair) continue;
		used |= 1u << sl;
		if (inner) cb[sl] = node_box[child_base + uint32_t(__popc(imask & ((1u << sl) - 1u)))];
		else
		{
			const uint32_t first = tri_base + uint32_t(__popc(valid & ((1u << (2 * sl)) - 1u)));
			RefitBox b = tri_box[first];          // reset + grow: the first record's box through min / max with the empty box, i.e. itself unless it holds NaNs
			#pragma unroll
			for (int k = 0; k < 3; ++k) { b.lo[k] = host_min(3.0e38f, b.lo[k]); b.hi[k] = host_max(-3.0e38f, b.hi[k]); }
			if (pair == 3u) { const RefitBox c = tri_box[first + 1]; for (int k = 0; k < 3; ++k) { b.lo[k] = host_min(b.lo[k], c.lo[k]); b.hi[k] = host_max(b.hi[k], c.hi[k]); } }
			cb[sl] = b;
		}
		#pragma unroll
		for (int k = 0; k < 3; ++k) { nb.lo[k] = host_min(nb.lo[k], cb[sl].lo[k]); nb.hi[k] = host_max(nb.hi[k], cb[sl].hi[k]); }
	}
	if (!used) { for (int k = 0; k < 3; ++k) { nb.lo[k] = 0.0f; nb.hi[k] = 0.0f; } }
	node_box[n] = nb;
	node.w[0] = as_u32(nb.lo[0]); node.w[1] = as_u32(nb.lo[1]); node.w[2] = as_u32(nb.lo[2]);
	int ex[3]; uint32_t ew = imask << 24;
	#pragma unroll
	for (int k = 0; k < 3; ++k) { ex[k] = grid_exponent(double(nb.hi[k]) - double(nb.lo[k])); ew |= uint32_t(ex[k] + 127) << (8 * k); }
	node.w[3] = ew;
	uint32_t q[12] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };      // words 8..19: qlo.x[8] qlo.y[8] qlo.z[8] qhi.x[8] qhi.y[8] qhi.z[8]
	bool bad = false;
	#pragma unroll
	for (int sl = 0; sl < 8; ++sl)
	{
		#pragma unroll
		for (int k = 0; k < 3; ++k)
		{
			double lo = 255.0, hi = 0.0;          // empty slot: the inverted box no ray hits
			if ((used >> sl) & 1u)
			{
				const double p = nb.lo[k], cell = ldexp(1.0, ex[k]);
				const double clo = double(cb[sl].lo[k]), chi = double(cb[sl].hi[k]);
				lo = floor((clo - p) / cell); lo = lo < 0.0 ? 0.0 : (lo > 255.0 ? 255.0 : lo);
				while (lo > 0.0 && !(p + lo * cell <= clo)) lo -= 1.0;
				hi = ceil((chi - p) / cell); hi = hi < 0.0 ? 0.0 : (hi > 255.0 ? 255.0 : hi);
				while (hi < 255.0 && !(p + hi * cell >= chi)) hi += 1.0;
				if (!(p + lo * cell <= clo) || !(p + hi * cell >= chi)) bad = true;          // a NaN or infinite vertex: the host throws "quantisation error (refit)"
			}
			q[2 * k + (sl >> 2)] |= uint32_t(lo) << (8 * (sl & 3));
			q[6 + 2 * k + (sl >> 2)] |= uint32_t(hi) << (8 * (sl & 3));
		}
	}
	#pragma unroll
	for (int w = 0; w < 12; ++w) node.w[8 + w] = q[w];
	nodes[n] = node;
	if (bad) atomicOr(scan + 1, 4u);
}

void launch_refit_scan(uint32_t n_tris, const int32_t* d_idx, uint32_t n_verts, const float* d_vtx, uint32_t n_records, const BvhTriangle* d_records, uint32_t* d_scan, hipStream_t s)
{
	const uint32_t n = n_verts > n_records ? n_verts : n_records;
	if (n) hipLaunchKernelGGL(refit_scan_kernel, dim3((n + 255u) / 256u), dim3(256), 0, s, n_tris, reinterpret_cast<const int4*>(d_idx), n_verts, reinterpret_cast<const float4*>(d_vtx),
	                          n_records, d_records, d_scan);
}
void launch_refit_records(uint32_t n_records, BvhTriangle* d_records, const int32_t* d_idx, const float* d_vtx, const uint32_t* d_scan, void* d_tri_box, hipStream_t s)
{
	if (n_records) hipLaunchKernelGGL(refit_records_kernel, dim3((n_records + 255u) / 256u), dim3(256), 0, s, n_records, d_records, reinterpret_cast<const int4*>(d_idx),
	                                  reinterpret_cast<const float4*>(d_vtx), d_scan, static_cast<RefitBox*>(d_tri_box));
}
void launch_refit_level(BvhNode8* d_nodes, void* d_node_box, const void* d_tri_box, uint32_t begin, uint32_t count, uint32_t* d_scan, hipStream_t s)
{
	if (count) hipLaunchKernelGGL(refit_level_kernel, dim3((count + 127u) / 128u), dim3(128), 0, s, d_nodes, static_cast<RefitBox*>(d_node_box), static_cast<const RefitBox*>(d_tri_box),
	                              begin, count, d_scan);
}

} // namespace fpt
