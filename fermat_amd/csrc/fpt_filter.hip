// fpt_filter.hip — the post-process step after the path tracer (kFiltered shading mode): variance box filter, edge-avoiding
// à-trous wavelet (EAW) steps and the per-ShadingMode to_rgba.
//
//   EAW_kernel / EAW_mad_kernel             src/eaw.cu:45-252           (one kernel here, MAD as a template flag, FilterOp bits at run time)
//   filter_variance_kernel                  src/renderer.cu:366-399
//   to_rgba_kernel                          src/renderer.cu:83-282
//   GBufferView unpack                      src/framebuffer.h:92-111 ; contrib/cugar/spherical/mappings_inline.h:162-172
//
// Roofline: by bytes a streaming stencil -- algorithmic bytes per pixel per EAW step = 16 (img) + 16 (gbuffer geo) + 16 (normals) +
// 4 (variance) + 16 (dst) [+ 16 weight image, + 16 dst read in add mode]; the 25 taps of neighbouring pixels overlap, so everything
// beyond the compulsory bytes is served by L2 (steps 1..8) or by the 256 MB Infinity Cache (the whole 1600x900 working set is ~90 MB)
// -- but measured it is VALU-bound: each of the 25 taps costs ~80 instructions (the deterministic exp, the reference's double-precision
// exponent sum), 2.9e9 lane-instructions per step = 0.07 ms at full rate against the 0.02 ms the compulsory bytes would take; the
// step runs in 0.10-0.13 ms.
// Launch: 64x4 tiles, one wavefront per 64-pixel row segment => every tap row is one coalesced 1 KB float4 load; the block
// index is taken XCD-major (blockIdx % 8 selects the XCD) so that vertically adjacent tiles, which share taps, meet in one L2.
#include "fpt_kernels.h"
#include "fpt_shading.h"

namespace fpt {

namespace {

enum { OP_MODULATE_IN = 0x1u, OP_DEMODULATE_IN = 0x2u, OP_MODULATE_OUT = 0x4u, OP_DEMODULATE_OUT = 0x8u, OP_ADD = 0x10u };

__device__ __forceinline__ float det_exp(float x) { return det_exp2(x * 1.44269504088896341f); }
__device__ __forceinline__ bool gb_miss(float4 g) { return (as_u32(g.w) & 0x80000000u) != 0u; }
__device__ __forceinline__ f3 gb_normal(float4 g)
{
	const uint32_t n = as_u32(g.w) & 0x7fffffffu;
	const float ux = float(n & 32767u) / 32767.0f, uy = float(n >> 15) / 32767.0f;
	const float ct = uy * 2.0f - 1.0f;
	const float st = sqrtf(ieee_max(1.0f - ct * ct, 0.0f));
	float s, c; det_sincos(ux * (2.0f * kPi), s, c);
	return mk3(c * st, s * st, ct);
}
__device__ __forceinline__ f4 clamp_below(float4 v, float m) { return mk4(sel_max(v.x, m), sel_max(v.y, m), sel_max(v.z, m), sel_max(v.w, m)); }
__device__ __forceinline__ f4 to_f4(float4 v) { return mk4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ f4 div4(f4 a, f4 b) { return mk4(a.x / b.x, a.y / b.y, a.z / b.z, a.w / b.w); }

// tile index -> (tile x, tile y): consecutive block ids round-robin over the 8 XCDs; give each XCD a contiguous band of tile rows
__device__ __forceinline__ void tile_of_block(uint32_t bid, uint32_t tiles_x, uint32_t tiles_y, uint32_t& tx, uint32_t& ty)
{
	const uint32_t n = tiles_x * tiles_y;
	const uint32_t per_xcd = (n + 7u) / 8u;
	uint32_t t = (bid % 8u) * per_xcd + bid / 8u;
	if (bid / 8u >= per_xcd || t >= n) { tx = tiles_x; ty = tiles_y; return; }       // padding block
	tx = t % tiles_x; ty = t / tiles_x;
}

// normals unpacked once per frame for the 14 steps of fpt_filter (the unpack costs a sin/cos + sqrt per tap otherwise)
__global__ void unpack_normals_kernel(const float4* __restrict__ geo, float4* __restrict__ nrm, uint32_t n)
{
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	if (i >= n) return;
	const f3 v = gb_normal(geo[i]);
	nrm[i] = make_float4(v.x, v.y, v.z, 0.0f);
}

// NRM: per-pixel normals were unpacked beforehand (same values, computed once)
template <bool MAD, bool NRM>
__global__ void __launch_bounds__(256) eaw_kernel(float4* __restrict__ dst, uint32_t op, const float4* __restrict__ w_img, float w_min, const float4* __restrict__ img,
                                                     const float4* __restrict__ geo, const float4* __restrict__ nrm, const float* __restrict__ var, EawParams prm, uint32_t step,
                                                     uint32_t res_x, uint32_t res_y, uint32_t tiles_x, uint32_t tiles_y)
{
	uint32_t tx, ty;
	tile_of_block(blockIdx.x, tiles_x, tiles_y, tx, ty);
	const uint32_t x = tx * 64u + (threadIdx.x & 63u), y = ty * 4u + (threadIdx.x >> 6);
	if (tx >= tiles_x || x >= res_x || y >= res_y) return;
	const size_t pc = size_t(y) * res_x + x;
	const float kw[3] = { 1.0f, float(2.0 / 3.0), float(1.0 / 6.0) };

	const f4 w_c = MAD ? clamp_below(w_img[pc], w_min) : mk4(1, 1, 1, 1);
	const f4 img_c = to_f4(img[pc]);
	const f4 col_c = !MAD ? img_c : (op & OP_MODULATE_IN) ? img_c * w_c : (op & OP_DEMODULATE_IN) ? div4(img_c, w_c) : img_c;
	const float4 geo_c = geo[pc];
	f4 result = col_c;
	if (!gb_miss(geo_c))
	{
		f3 n_c;
		if (NRM) { const float4 t = nrm[pc]; n_c = mk3(t.x, t.y, t.z); } else n_c = gb_normal(geo_c);
		const f3 p_c = mk3(geo_c.x, geo_c.y, geo_c.z);
		const f3 rel = MAD ? p_c : p_c - prm.E;           // as written in the reference: only the plain kernel subtracts the eye
		const float radius = 20 * sel_min(length(prm.U) / float(res_x), length(prm.V) / float(res_y)) * dot(rel, prm.W) / dot(prm.W, prm.W);
		const float variance = var ? var[pc] : 1.0f;
		const float phi_n = prm.phi_normal * float(step) * float(step);
		const float phi_p = prm.phi_position / (radius * radius);
		const float phi_c = prm.phi_color / sel_max(1.0e-3f, variance * variance);
		float sum_w = 0.0f;
		f3 sum_c = mk3(0, 0, 0);
		for (int yy = -2; yy <= 2; ++yy)
		{
			const int py = int(y) + yy * int(step);
			if (py < 0 || py >= int(res_y)) continue;
			#pragma unroll
			for (int xx = -2; xx <= 2; ++xx)
			{
				const int px = int(x) + xx * int(step);
				if (px < 0 || px >= int(res_x)) continue;
				const size_t pp = size_t(py) * res_x + size_t(px);
				const float4 geo_p = geo[pp];
				f4 col_p = to_f4(img[pp]);
				if (MAD)
				{
					const f4 w_p = clamp_below(w_img[pp], w_min);
					col_p = (op & OP_MODULATE_IN) ? col_p * w_p : (op & OP_DEMODULATE_IN) ? div4(col_p, w_p) : col_p;
				}
				if (gb_miss(geo_p)) continue;
				const f3 dc = mk3(col_p.x - col_c.x, col_p.y - col_c.y, col_p.z - col_c.z);
				const float w_col = dot(dc, dc) * phi_c;
				f3 n_p;
				if (NRM) { const float4 t = nrm[pp]; n_p = mk3(t.x, t.y, t.z); } else n_p = gb_normal(geo_p);
				const float w_nrm = (1.0f - sel_max(1e-8f, dot(n_p, n_c))) * phi_n;
				const f3 dp = mk3(geo_p.x, geo_p.y, geo_p.z) - p_c;
				const float w_pos = dot(dp, dp) * phi_p;
				// the reference writes expf(0.0 - a - b - c): the double literal promotes the sum
				const double e = ((0.0 - double(sel_max(w_pos, 0.0f))) - double(sel_max(w_nrm, 0.0f))) - double(sel_max(w_col, 0.0f));
				const float w = (kw[xx < 0 ? -xx : xx] * kw[yy < 0 ? -yy : yy]) * det_exp(float(e));
				sum_w += w;
				sum_c = sum_c + w * mk3(col_p.x, col_p.y, col_p.z);
			}
		}
		if (sum_w) result = mk4(sum_c.x / sum_w, sum_c.y / sum_w, sum_c.z / sum_w, col_c.w);
	}
	if (MAD)
	{
		f4 r = (op & OP_ADD) ? to_f4(dst[pc]) : mk4(0, 0, 0, 0);
		r = r + ((op & OP_MODULATE_OUT) ? result * w_c : (op & OP_DEMODULATE_OUT) ? div4(result, w_c) : result);
		result = r;
	}
	dst[pc] = make_float4(result.x, result.y, result.z, result.w);
}

__global__ void __launch_bounds__(256) filter_variance_kernel(const float4* __restrict__ img, float* __restrict__ var, uint32_t FW, uint32_t res_x, uint32_t res_y)
{
	const uint32_t x = blockIdx.x * 64u + (threadIdx.x & 63u), y = blockIdx.y * 4u + (threadIdx.x >> 6);
	if (x >= res_x || y >= res_y) return;
	const int lx = x > FW ? int(x - FW) : 0, rx = x + FW < res_x ? int(x + FW) : int(res_x) - 1;
	const int ly = y > FW ? int(y - FW) : 0, ry = y + FW < res_y ? int(y + FW) : int(res_y) - 1;
	float v = 0.0f;
	for (int yy = ly; yy <= ry; ++yy)
		for (int xx = lx; xx <= rx; ++xx)
			v += img[size_t(yy) * res_x + size_t(xx)].w;
	v /= float((ry - ly + 1) * (rx - lx + 1));
	var[size_t(y) * res_x + x] = v;
}

__device__ __forceinline__ uint32_t pack_bytes(const float c[4])
{
	uint32_t r = 0;
	#pragma unroll
	for (int k = 0; k < 4; ++k) r |= (to_u32_sat(ieee_min(c[k] * 256.0f, 255.0f)) & 0xffu) << (8 * k);
	return r;
}

__global__ void rgba_mode_kernel(FrameBufferDev fb, uint32_t mode, uint32_t n, float exposure, float inv_gamma,
                                 uint32_t* __restrict__ rgba)
{
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	if (i >= n) return;
	float c[4] = { 0, 0, 0, 0 };
	int tonemap_channel = -1;
	switch (mode)
	{
	case FPT_SHADING_SHADED: tonemap_channel = FPT_FB_COMPOSITED_C; break;
	case FPT_SHADING_FILTERED: tonemap_channel = FPT_FB_FILTERED_C; break;
	case FPT_SHADING_DIFFUSE_COLOR: tonemap_channel = FPT_FB_DIFFUSE_C; break;
	case FPT_SHADING_SPECULAR_COLOR: tonemap_channel = FPT_FB_SPECULAR_C; break;
	case FPT_SHADING_DIRECT_LIGHTING: tonemap_channel = FPT_FB_DIRECT_C; break;
	case FPT_SHADING_ALBEDO: { const float4 a = fb.ch[FPT_FB_DIFFUSE_A][i], b = fb.ch[FPT_FB_SPECULAR_A][i]; c[0] = a.x + b.x; c[1] = a.y + b.y; c[2] = a.z + b.z; c[3] = a.w + b.w; break; }
	case FPT_SHADING_DIFFUSE_ALBEDO: { const float4 a = fb.ch[FPT_FB_DIFFUSE_A][i]; c[0] = a.x; c[1] = a.y; c[2] = a.z; c[3] = a.w; break; }
	case FPT_SHADING_SPECULAR_ALBEDO: { const float4 a = fb.ch[FPT_FB_SPECULAR_A][i]; c[0] = a.x; c[1] = a.y; c[2] = a.z; c[3] = a.w; break; }
	case FPT_SHADING_VARIANCE:
	{
		float v = fb.ch[FPT_FB_COMPOSITED_C][i].w * exposure;
		v = v / (v + 1);
		v = det_pow(v, inv_gamma);
		c[0] = c[1] = c[2] = c[3] = v; break;
	}
	case FPT_SHADING_UV: { const float4 uv = fb.gb_uv[i]; c[0] = uv.z; c[1] = uv.w; c[2] = 0.5f; c[3] = 0.0f; break; }
	case FPT_SHADING_NORMAL:
	{
		const f3 nrm = gb_normal(fb.gb_geo[i]);
		rgba[i] = (to_u32_sat(ieee_min(nrm.x * 128.0f + 128.0f, 255.0f)) & 0xffu) | ((to_u32_sat(ieee_min(nrm.y * 128.0f + 128.0f, 255.0f)) & 0xffu) << 8)
		        | ((to_u32_sat(ieee_min(nrm.z * 128.0f + 128.0f, 255.0f)) & 0xffu) << 16);
		return;
	}
	default: break;
	}
	if (tonemap_channel >= 0)
	{
		const float4 s = fb.ch[tonemap_channel][i];
		const float v[4] = { s.x * exposure, s.y * exposure, s.z * exposure, s.w * exposure };
		#pragma unroll
		for (int k = 0; k < 4; ++k) c[k] = det_pow(v[k] / (v[k] + 1.0f), inv_gamma);
	}
	rgba[i] = pack_bytes(c);
}

} // namespace

void launch_eaw(float4* dst, int op, const float4* w_img, float w_min, const float4* img, const float4* geo, const float4* nrm, const float* var, const EawParams& prm, uint32_t step,
                uint32_t res_x, uint32_t res_y, hipStream_t s)
{
	const uint32_t tiles_x = (res_x + 63u) / 64u, tiles_y = (res_y + 3u) / 4u;
	const uint32_t blocks = ((tiles_x * tiles_y + 7u) / 8u) * 8u;
	#define FPT_EAW_LAUNCH(MAD, NRM, OP) hipLaunchKernelGGL((eaw_kernel<MAD, NRM>), dim3(blocks), dim3(256), 0, s, dst, OP, w_img, w_min, img, geo, nrm, var, prm, step, res_x, res_y, tiles_x, tiles_y)
	if (op < 0) { if (nrm) FPT_EAW_LAUNCH(false, true, 0u); else FPT_EAW_LAUNCH(false, false, 0u); }
	else        { if (nrm) FPT_EAW_LAUNCH(true, true, uint32_t(op)); else FPT_EAW_LAUNCH(true, false, uint32_t(op)); }
	#undef FPT_EAW_LAUNCH
}
void launch_unpack_normals(const float4* geo, float4* nrm, uint32_t n, hipStream_t s)
{ hipLaunchKernelGGL(unpack_normals_kernel, dim3((n + 255u) / 256u), dim3(256), 0, s, geo, nrm, n); }
void launch_filter_variance(const float4* img, float* var, uint32_t FW, uint32_t res_x, uint32_t res_y, hipStream_t s)
{ hipLaunchKernelGGL(filter_variance_kernel, dim3((res_x + 63u) / 64u, (res_y + 3u) / 4u), dim3(256), 0, s, img, var, FW, res_x, res_y); }
void launch_rgba_mode(const FrameBufferDev& fb, uint32_t mode, uint32_t n, float exposure, float inv_gamma, uint32_t* rgba, hipStream_t s)
{ hipLaunchKernelGGL(rgba_mode_kernel, dim3((n + 255u) / 256u), dim3(256), 0, s, fb, mode, n, exposure, inv_gamma, rgba); }

} // namespace fpt
