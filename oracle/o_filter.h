// TEST INFRASTRUCTURE ONLY (see oracle/README in DESIGN.md §3): CPU restatement of Fermat's post-process path, the step AFTER the
// path tracer in `kFiltered` shading mode.  Parity unpinned at image level (the reference ships no filtered images).
//
//   EAW_kernel, EAW_mad_kernel, norm_diff, the three EAW drivers     src/eaw.cu:45-368 ; EAWParams src/eaw.h ; FilterOp src/filters.h:44-57
//   filter_variance_kernel                                            src/renderer.cu:366-399
//   RenderingContextImpl::filter                                      src/renderer.cu:1099-1151
//   GBufferView::{is_miss,unpack_pos,unpack_normal}                   src/framebuffer.h:92-111
//   uniform_square_to_sphere, unpack_vector                           contrib/cugar/spherical/mappings_inline.h:162-172 ; linalg/vector_inl.h:464-472
//   to_rgba_kernel, every ShadingMode except kUVStretch/kCharts/kAux  src/renderer.cu:83-282 ; enum src/renderer_view.h:61-76
// Floating point: expf/cosf/sinf/powf are the deterministic "detmath v1" polynomials shared with the product.
#pragma once
#include "o_pt.h"

namespace orc {

enum { kFilterOpNone = 0x0u, kFilterOpModulateInput = 0x1u, kFilterOpDemodulateInput = 0x2u, kFilterOpModulateOutput = 0x4u,
       kFilterOpDemodulateOutput = 0x8u, kFilterOpAddMode = 0x10u, kFilterOpReplaceMode = 0x20u };

enum { kShaded = 0, kUV = 1, kUVStretch = 2, kCharts = 3, kAlbedo = 4, kDiffuseAlbedo = 5, kSpecularAlbedo = 6, kDiffuseColor = 7,
       kSpecularColor = 8, kDirectLighting = 9, kFiltered = 10, kVariance = 11, kNormal = 12, kAux0 = 13 };

struct EAWParams { float phi_normal, phi_position, phi_color; V3 E, U, V, W; };

inline float det_exp(float x) { return det_exp2(x * 1.44269504088896341f); }

inline bool gb_is_miss(const float* geo) { return (f2bits(geo[3]) & (1u << 31)) != 0; }
inline V3 gb_unpack_pos(const float* geo) { return V3(geo[0], geo[1], geo[2]); }
inline V3 gb_unpack_normal(const float* geo)
{
	const u32 n_i = f2bits(geo[3]) & ~(1u << 31);
	const u32 MAXV = (1u << 15) - 1u;
	const float ux = float(n_i & MAXV) / float(MAXV), uy = float(n_i >> 15) / float(MAXV);
	const float cosTheta = uy * 2.0f - 1.0f;
	const float sinTheta = sqrtf(fmax_ieee(1.0f - cosTheta * cosTheta, 0.0f));
	const float phi = ux * (2.0f * PI_F);
	float s, c; det_sincos(phi, &s, &c);
	return V3(c * sinTheta, s * sinTheta, cosTheta);
}

// src/eaw.cu:36-42
inline float norm_diff(V3 a, V3 b) { const float d = maxf(1e-8f, dot(a, b)); return 1.0f - d; }

struct Image { float* data; u32 res_x, res_y; V4 at(u32 x, u32 y) const { const float* f = data + 4 * (size_t(y) * res_x + x); return V4(f[0], f[1], f[2], f[3]); }
               void put(u32 x, u32 y, V4 v) { float* f = data + 4 * (size_t(y) * res_x + x); f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w; } };

inline V4 vmax(V4 a, float m) { return V4(maxf(a.x, m), maxf(a.y, m), maxf(a.z, m), maxf(a.w, m)); }
inline V4 vdiv(V4 a, V4 b) { return V4(a.x / b.x, a.y / b.y, a.z / b.z, a.w / b.w); }

// one à-trous step.  op == -1 selects EAW_kernel (src/eaw.cu:45-123); op >= 0 EAW_mad_kernel (:125-252) with the FilterOp bits
inline void eaw_step(Image dst, int op, Image w_img, float w_min, Image img, const float* gb_geo, const float* var, const EAWParams& params, u32 step_size)
{
	const float kernelWeights[3] = { float(1.0), float(2.0 / 3.0), float(1.0 / 6.0) };
	const bool mad = op >= 0;
	std::vector<float> out(size_t(dst.res_x) * dst.res_y * 4);
	for (u32 y = 0; y < dst.res_y; ++y)
		for (u32 x = 0; x < dst.res_x; ++x)
		{
			float* o = &out[4 * (size_t(y) * dst.res_x + x)];
			auto store = [&](V4 v) { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; };
			const V4 weightCenter = mad ? vmax(w_img.at(x, y), w_min) : V4(1, 1, 1, 1);
			const V4 imgCenter = img.at(x, y);
			const V4 colorCenter = !mad ? imgCenter : (op & kFilterOpModulateInput) ? imgCenter * weightCenter : (op & kFilterOpDemodulateInput) ? vdiv(imgCenter, weightCenter) : imgCenter;
			const float* geoC = gb_geo + 4 * (size_t(y) * img.res_x + x);
			const V3 normalCenter = gb_unpack_normal(geoC);
			const V3 positionCenter = gb_unpack_pos(geoC);
			auto finish = [&](V4 c)
			{
				if (!mad) { store(c); return; }
				V4 r = (op & kFilterOpAddMode) ? dst.at(x, y) : V4(0, 0, 0, 0);
				r = r + ((op & kFilterOpModulateOutput) ? c * weightCenter : (op & kFilterOpDemodulateOutput) ? vdiv(c, weightCenter) : c);
				store(r);
			};
			if (gb_is_miss(geoC)) { finish(colorCenter); continue; }
			// the plain kernel measures the depth from the eye, the mad kernel from the origin (as written, :62-64 vs :169-171)
			const V3 rel = mad ? positionCenter : positionCenter - params.E;
			const float posRadius = 20 * minf(length(params.U) / float(img.res_x), length(params.V) / float(img.res_y)) * dot(rel, params.W) / dot(params.W, params.W);
			const float variance = var ? var[x + size_t(y) * img.res_x] : 1.0f;
			const float phiNormal = params.phi_normal * float(step_size) * float(step_size);
			const float phiPosition = params.phi_position / (posRadius * posRadius);
			const float phiColor = params.phi_color / maxf(1.0e-3f, variance * variance);
			float sumWeight = 0.0f;
			V3 sumColor(0, 0, 0);
			for (int yy = -2; yy <= 2; yy++)
				for (int xx = -2; xx <= 2; xx++)
				{
					const int px = int(x) + xx * int(step_size), py = int(y) + yy * int(step_size);
					const bool inside = (px >= 0 && py >= 0) && (px < int(img.res_x) && py < int(img.res_y));
					const float kernel = kernelWeights[xx < 0 ? -xx : xx] * kernelWeights[yy < 0 ? -yy : yy];
					if (!inside) continue;
					V4 colorP = img.at(u32(px), u32(py));
					if (mad)
					{
						const V4 weightP = vmax(w_img.at(u32(px), u32(py)), w_min);
						colorP = (op & kFilterOpModulateInput) ? colorP * weightP : (op & kFilterOpDemodulateInput) ? vdiv(colorP, weightP) : colorP;
					}
					const float* geoP = gb_geo + 4 * (size_t(py) * img.res_x + px);
					if (gb_is_miss(geoP)) continue;
					const V3 normalP = gb_unpack_normal(geoP), positionP = gb_unpack_pos(geoP);
					const V3 diffCol = colorP.xyz() - colorCenter.xyz();
					const float wColor = dot(diffCol, diffCol) * phiColor;
					const float wNormal = norm_diff(normalP, normalCenter) * phiNormal;
					const V3 diffPosition = positionP - positionCenter;
					const float wPosition = dot(diffPosition, diffPosition) * phiPosition;
					// "expf(0.0 - a - b - c)": the literal 0.0 promotes the sum to double (:98-102)
					const double e = ((0.0 - double(maxf(wPosition, 0.0f))) - double(maxf(wNormal, 0.0f))) - double(maxf(wColor, 0.0f));
					const float w = kernel * det_exp(float(e));
					sumWeight += w;
					sumColor = sumColor + w * colorP.xyz();
				}
			finish(sumWeight ? V4(sumColor.x / sumWeight, sumColor.y / sumWeight, sumColor.z / sumWeight, colorCenter.w) : colorCenter);
		}
	std::memcpy(dst.data, out.data(), out.size() * sizeof(float));
}

// src/renderer.cu:366-399
inline void filter_variance(Image img, float* var, u32 FW)
{
	for (u32 y = 0; y < img.res_y; ++y)
		for (u32 x = 0; x < img.res_x; ++x)
		{
			const i32 lx = x > FW ? i32(x - FW) : 0, rx = x + FW < img.res_x ? i32(x + FW) : i32(img.res_x) - 1;
			const i32 ly = y > FW ? i32(y - FW) : 0, ry = y + FW < img.res_y ? i32(y + FW) : i32(img.res_y) - 1;
			float variance = 0.0f;
			for (i32 yy = ly; yy <= ry; yy++)
				for (i32 xx = lx; xx <= rx; xx++)
					variance += img.at(u32(xx), u32(yy)).w;
			variance /= float((ry - ly + 1) * (rx - lx + 1));
			var[x + size_t(y) * img.res_x] = variance;
		}
}

// the weighted multi-iteration driver, src/eaw.cu:320-368: dst += w_img * eaw^n(img / w_img)
inline void eaw_weighted(u32 n_iterations, Image dst, Image w_img, Image img, const float* gb_geo, const float* var, const EAWParams& params, Image pingpong[2])
{
	u32 in_buffer = 0;
	for (u32 i = 0; i < n_iterations; ++i)
	{
		const u32 out_buffer = in_buffer ? 0 : 1;
		if (i == n_iterations - 1) eaw_step(dst, int(kFilterOpModulateOutput | kFilterOpAddMode), w_img, 1.0e-4f, i == 0 ? img : pingpong[in_buffer], gb_geo, var, params, 1u << i);
		else if (i == 0)           eaw_step(pingpong[out_buffer], int(kFilterOpDemodulateInput | kFilterOpReplaceMode), w_img, 1.0e-4f, img, gb_geo, var, params, 1u << i);
		else                       eaw_step(pingpong[out_buffer], -1, w_img, 0.0f, pingpong[in_buffer], gb_geo, var, params, 1u << i);
		in_buffer = out_buffer;
	}
}

// RenderingContextImpl::filter (src/renderer.cu:1099-1151): FILTERED_C = DIRECT_C + eaw(DIFFUSE_C | DIFFUSE_A) + eaw(SPECULAR_C | SPECULAR_A)
inline void filter_frame(FrameBuffer& fb, const SceneView& scene, u32 instance)
{
	const size_t n = size_t(fb.res_x) * fb.res_y;
	std::memcpy(fb.channels[FB_FILTERED_C], fb.channels[FB_DIRECT_C], n * 16);
	EAWParams p;
	p.phi_normal = 2.0f; p.phi_position = 1.0f; p.phi_color = float(instance * instance + 1) / 10000.0f;
	p.E = scene.camera.eye;
	camera_frame(scene.camera, scene.aspect, p.U, p.V, p.W);
	std::vector<float> t0(n * 4), t1(n * 4), var(n);
	Image pingpong[2] = { { t0.data(), fb.res_x, fb.res_y }, { t1.data(), fb.res_x, fb.res_y } };
	Image output = { fb.channels[FB_FILTERED_C], fb.res_x, fb.res_y };
	const u32 pairs[2][2] = { { FB_DIFFUSE_C, FB_DIFFUSE_A }, { FB_SPECULAR_C, FB_SPECULAR_A } };
	for (int k = 0; k < 2; ++k)
	{
		Image input = { fb.channels[pairs[k][0]], fb.res_x, fb.res_y }, weight = { fb.channels[pairs[k][1]], fb.res_x, fb.res_y };
		filter_variance(input, var.data(), 2);
		eaw_weighted(7, output, weight, input, fb.gb_geo, var.data(), p, pingpong);
	}
}

// to_rgba_kernel (src/renderer.cu:83-282) for a given ShadingMode
inline void to_rgba_mode(const FrameBuffer& fb, const SceneView& scene, u32 mode, uint8_t* rgba)
{
	const u32 np = fb.res_x * fb.res_y;
	auto put = [&](u32 p, const float c[4]) { for (int i = 0; i < 4; ++i) rgba[4 * size_t(p) + i] = uint8_t(f2u(fmin_ieee(c[i] * 256.0f, 255.0f))); };
	auto tonemapped = [&](V4 c, float out[4])
	{
		c = c * scene.exposure;
		const float v[4] = { c.x / (c.x + 1.0f), c.y / (c.y + 1.0f), c.z / (c.z + 1.0f), c.w / (c.w + 1.0f) };
		for (int i = 0; i < 4; ++i) out[i] = det_pow(v[i], 1.0f / scene.gamma);
	};
	for (u32 p = 0; p < np; ++p)
	{
		float c[4] = { 0, 0, 0, 0 };
		switch (mode)
		{
		case kShaded:         tonemapped(fb.get(FB_COMPOSITED_C, p), c); put(p, c); break;
		case kFiltered:       tonemapped(fb.get(FB_FILTERED_C, p), c); put(p, c); break;
		case kDiffuseColor:   tonemapped(fb.get(FB_DIFFUSE_C, p), c); put(p, c); break;
		case kSpecularColor:  tonemapped(fb.get(FB_SPECULAR_C, p), c); put(p, c); break;
		case kDirectLighting: tonemapped(fb.get(FB_DIRECT_C, p), c); put(p, c); break;
		case kAlbedo:         { const V4 a = fb.get(FB_DIFFUSE_A, p) + fb.get(FB_SPECULAR_A, p); c[0] = a.x; c[1] = a.y; c[2] = a.z; c[3] = a.w; put(p, c); break; }
		case kDiffuseAlbedo:  { const V4 a = fb.get(FB_DIFFUSE_A, p); c[0] = a.x; c[1] = a.y; c[2] = a.z; c[3] = a.w; put(p, c); break; }
		case kSpecularAlbedo: { const V4 a = fb.get(FB_SPECULAR_A, p); c[0] = a.x; c[1] = a.y; c[2] = a.z; c[3] = a.w; put(p, c); break; }
		case kVariance:
		{
			float v = fb.get(FB_COMPOSITED_C, p).w * scene.exposure;
			v = v / (v + 1);
			v = det_pow(v, 1.0f / scene.gamma);
			c[0] = c[1] = c[2] = c[3] = v; put(p, c); break;
		}
		case kUV:             { const float* uv = fb.gb_uv + 4 * size_t(p); c[0] = uv[2]; c[1] = uv[3]; c[2] = 0.5f; c[3] = 0.0f; put(p, c); break; }
		case kNormal:
		{
			const V3 nrm = gb_unpack_normal(fb.gb_geo + 4 * size_t(p));
			rgba[4 * size_t(p) + 0] = uint8_t(f2u(fmin_ieee(nrm.x * 128.0f + 128.0f, 255.0f)));
			rgba[4 * size_t(p) + 1] = uint8_t(f2u(fmin_ieee(nrm.y * 128.0f + 128.0f, 255.0f)));
			rgba[4 * size_t(p) + 2] = uint8_t(f2u(fmin_ieee(nrm.z * 128.0f + 128.0f, 255.0f)));
			rgba[4 * size_t(p) + 3] = 0;
			break;
		}
		default: put(p, c); break;
		}
	}
}

} // namespace orc
