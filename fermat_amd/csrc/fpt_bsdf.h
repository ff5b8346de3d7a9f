// fpt_bsdf.h — Fermat's layered surface model for the gfx950 shading kernels.
//
// Model (src/bsdf.h:123-1281 with USE_GGX_SMITH / USE_APPROX_SMITH / USE_EFFICIENT_SAMPLER_WITH_APPROXIMATE_PDFS):
// four inner lobes — Lambert reflection/transmission (contrib/cugar/bsdf/lambert.h:64-140, lambert_trans.h:51-140)
// and GGX reflection/transmission with the approximate height-correlated Smith term and VNDF sampling
// (contrib/cugar/bsdf/ggx_smith.h:203-690, ggx_common.h:50-110,265-290) — under a specular clearcoat
// (src/bsdf.h:1202-1232).  Only what shade_vertex reaches with default options is implemented: projected-solid-angle
// measure, all components enabled, Russian roulette on (src/pathtracer_core.h:906,1024,1190).
//
// Layout choice: the lobe objects of the reference (5 structs, ~100 B) are flattened into one register-resident
// parameter block; lobes are free functions selected by a `transmissive` flag so the compiler can share the GGX body.
#pragma once
#include "fpt_math.h"

namespace fpt {

struct ShadingFrame { f3 n, ng, t, b; };         // shading normal, geometric normal, tangent (not normalised), binormal

FPT_HD f3 to_local(const ShadingFrame& fr, f3 v) { return mk3(dot(v, fr.t), dot(v, fr.b), dot(v, fr.n)); }
FPT_HD f3 from_local(const ShadingFrame& fr, f3 v) { return v.x * fr.t + v.y * fr.b + v.z * fr.n; }

enum : uint32_t { LOBE_DIFF_R = 0, LOBE_DIFF_T = 1, LOBE_GLOSSY_R = 2, LOBE_GLOSSY_T = 3 };
enum : uint32_t { COMP_ABSORB = 0u, COMP_DIFF_R = 1u, COMP_DIFF_T = 2u, COMP_GLOSSY_R = 4u, COMP_GLOSSY_T = 8u, COMP_COAT = 0x10u,
                  COMP_DIFFUSE_MASK = 3u, COMP_GLOSSY_MASK = 0xCu };

struct SurfaceModel
{
	f3 kd;            // diffuse / pi                  (src/bsdf.h:226)
	f3 kdt;           // diffuse transmission / pi     (:227)
	f3 ks;            // "fresnel" = specular / pi     (:234)
	f3 coat;          // clearcoat normal-incidence reflectivity (:235)
	float alpha;      // GGX roughness (used unsquared, ggx_smith.h:234,251)
	float inv_alpha;
	float ior, opacity, coat_ior;
	const float* table;   // 32^4 directional-albedo table (src/bsdf.h:1254-1268)
};

// ---- small pieces -----------------------------------------------------------------------------------------------------
// concentric square->disk map then lift to the cosine hemisphere (contrib/cugar/spherical/mappings_inline.h:56-87,119-126)
FPT_HD f3 cosine_hemisphere(float u0, float u1)
{
	const float a = 2 * u0 - 1;
	const float b = 2 * u1 - 1;
	float phi, r;
	if (a > -b)
	{
		if (a > b) { r = a;  phi = (kPi / 4) * (b / a); }
		else       { r = b;  phi = (kPi / 4) * (2 - (a / b)); }
	}
	else
	{
		if (a < b) { r = -a; phi = (kPi / 4) * (4 + (b / a)); }
		else       { r = -b; phi = b != 0 ? (kPi / 4) * (6 - (a / b)) : 0; }
	}
	float s, c;
	det_sincos(phi, s, c);
	const float dx = r * c, dy = r * s;
	const float r2 = dx * dx + dy * dy;
	return mk3(dx, dy, sqrtf(sel_max(1.0f - r2, 0.0f)));
}

// Schlick Fresnel with TIR guard (contrib/cugar/bsdf/refraction.h:91-118)
FPT_HD f3 schlick(float cos_i, float eta, f3 base)
{
	cos_i = saturate(fabsf(cos_i));
	const float cos_t2 = saturate(1.f - eta * eta * (1.f - cos_i * cos_i));
	if (cos_t2 < 0.0f) return splat3(1.0f);
	const float ct = eta > 1.0f ? sqrtf(cos_t2) : cos_i;
	const float w = 1 - ct;
	const float w2 = w * w;
	const float Fc = w2 * w2 * w;
	return splat3(Fc) + (1 - Fc) * base;
}

// half vector conventions of the two call sites (contrib/cugar/bsdf/ggx_common.h:50-84)
FPT_HD f3 half_vector_n(f3 V, f3 L, f3 N, float inv_eta)       // "microfacet": oriented with N, degenerate -> N
{
	f3 H = (dot(V, N) * dot(L, N) >= 0.0f) ? V + L : V + L * inv_eta;
	if (dot(H, H) == 0.0f) return N;
	if (dot(N, H) < 0.0f) H = -H;
	return normalize(H);
}
FPT_HD f3 half_vector_v(f3 V, f3 L, f3 N, float inv_eta)       // "vndf_microfacet": oriented with V, |H|^2 < 1e-12 -> N
{
	f3 H = (dot(V, N) * dot(L, N) >= 0.0f) ? V + L : V + L * inv_eta;
	if (dot(H, H) < 1.0e-12f) return N;
	if (dot(V, H) < 0.0f) H = -H;
	return normalize(H);
}

FPT_HD float ggx_ndf(float inv_alpha, float nh, float ht, float hb)     // ggx_common.h:86-106, isotropic
{
	const float x = ht * inv_alpha;
	const float y = hb * inv_alpha;
	const float aniso = x * x + y * y;
	const float f = aniso + nh * nh;
	return (1.0f / kPi) * inv_alpha * inv_alpha / (f * f);
}
FPT_HD float clamp_pdf(float p) { return (!is_finite(p) || p != p) ? 1.0e8f : sel_max(p, 0.0f); }     // ggx_smith.h:228

// visible-normal sampling in the local frame (ggx_common.h:265-290)
FPT_HD f3 sample_vndf(float u0, float u1, float alpha, f3 Vin)
{
	const f3 V = normalize(mk3(alpha * Vin.x, alpha * Vin.y, Vin.z));
	const f3 T1 = (V.z < 0.9999f) ? normalize(cross(V, mk3(0, 0, 1))) : mk3(1, 0, 0);
	const f3 T2 = cross(T1, V);
	const float a = 1.0f / (1.0f + V.z);
	const float r = sqrtf(u0);
	const float phi = (u1 < a) ? u1 / a * kPi : kPi + (u1 - a) / (1.0f - a) * kPi;
	float sp, cp;
	det_sincos(phi, sp, cp);
	const float P1 = r * cp;
	const float P2 = r * sp * ((u1 < a) ? 1.0f : V.z);
	f3 N = P1 * T1 + P2 * T2 + sqrtf(sel_max(0.0f, 1.0f - P1 * P1 - P2 * P2)) * V;
	N = normalize(mk3(alpha * N.x, alpha * N.y, sel_max(0.0f, N.z)));
	return N;
}

// ---- GGX lobe (reflective: int_ior = ext_ior = -1 ; transmissive: int_ior = material ior, ext_ior = 1) -----------------
struct GgxLobe { float alpha, inv_alpha, int_ior, ext_ior; };
FPT_HD GgxLobe ggx_reflective(float alpha) { GgxLobe l; l.alpha = alpha; l.inv_alpha = 1.0f / alpha; l.int_ior = -1.0f; l.ext_ior = -1.0f; return l; }
FPT_HD GgxLobe ggx_transmissive(float alpha, float ior) { GgxLobe l; l.alpha = alpha; l.inv_alpha = 1.0f / alpha; l.int_ior = ior; l.ext_ior = 1.0f; return l; }

FPT_HD float ggx_vis_joint(const GgxLobe& l, float NoV, float NoL)      // ggx_smith.h:232-241
{
	const float a = l.alpha;
	const float vv = NoL * (NoV * (1 - a) + a);
	const float vl = NoV * (NoL * (1 - a) + a);
	return 0.5f * 1.0f / (vv + vl);
}
FPT_HD float ggx_vis_g1(const GgxLobe& l, float NoV, float NoL)         // :255-264
{
	const float a2 = l.alpha * l.alpha;
	const float G_V = NoV + sqrtf((NoV - NoV * a2) * NoV + a2);
	return 0.5f / (G_V * NoL);
}
FPT_HD float ggx_refraction_jacobian(float VoH, float LoH, float eta, float inv_eta)    // :311-330
{
	const float ci = fabsf(VoH);
	const float ct2 = 1.f - eta * eta * (1.f - ci * ci);
	if (ct2 < 0.0f) return 0.0f;
	const float sd = VoH + inv_eta * LoH;
	return 4 * inv_eta * inv_eta * fabsf(VoH * LoH) / (sd * sd);
}
// common tail of f_and_p / sample: given V, L, H produce (G*D*T, G1*D*T, G/G1) or report "no exchange"
FPT_HD bool ggx_terms(const GgxLobe& l, const ShadingFrame& fr, f3 V, f3 L, f3 H, float NoV, float eta, float inv_eta,
                      float& fval, float& pval, float& ratio)
{
	const bool trans = l.int_ior > 0.0f;
	const float NoL = dot(fr.n, L);
	const float NoH = dot(fr.n, H);
	const float sgn = trans ? -1.0f : 1.0f;
	if (sgn * NoL * NoV <= 0.0f || NoH == 0.0f) return false;
	const float D  = ggx_ndf(l.inv_alpha, fabsf(NoH), dot(fr.t, H), dot(fr.b, H));
	const float G  = ggx_vis_joint(l, fabsf(NoV), fabsf(NoL));
	const float G1 = ggx_vis_g1(l, fabsf(NoV), fabsf(NoL));
	float tf = 1.0f;
	if (trans) tf = ggx_refraction_jacobian(dot(V, H), dot(L, H), eta, inv_eta);
	fval = clamp_pdf(G * D * tf);
	pval = clamp_pdf(G1 * D * tf);
	ratio = clamp_pdf(G / G1);
	return true;
}
// f and projected pdf for a given pair of directions (ggx_smith.h:414-467)
FPT_HD void ggx_eval(const GgxLobe& l, const ShadingFrame& fr, f3 V, f3 L, float& f, float& p)
{
	const float NoV = dot(fr.n, V);
	const float eta     = NoV >= 0.0f ? l.ext_ior / l.int_ior : l.int_ior / l.ext_ior;
	const float inv_eta = NoV >= 0.0f ? l.int_ior / l.ext_ior : l.ext_ior / l.int_ior;
	const f3 H = half_vector_v(V, L, fr.n, inv_eta);
	float ratio;
	if (!ggx_terms(l, fr, V, L, H, NoV, eta, inv_eta, f, p, ratio)) { f = 0.0f; p = 0.0f; }
}
// outgoing direction for a given microfacet normal (ggx_smith.h:503-578); returns weight g = f/p_proj, p, p_proj
FPT_HD void ggx_sample_given_h(const GgxLobe& l, const ShadingFrame& fr, f3 H, f3 V, f3& L, float& g, float& p, float& p_proj)
{
	const bool trans = l.int_ior > 0.0f;
	const float NoV = dot(fr.n, V);
	const float eta     = NoV >= 0.0f ? l.ext_ior / l.int_ior : l.int_ior / l.ext_ior;
	const float inv_eta = NoV >= 0.0f ? l.int_ior / l.ext_ior : l.ext_ior / l.int_ior;
	g = 0.0f; p = 0.0f; p_proj = 0.0f;
	if (NoV == 0.0f) return;
	if (!trans)
		L = 2 * dot(V, H) * H - V;
	else
	{
		const float ci = dot(V, H);
		const float ct2 = 1.f - eta * eta * (1.f - ci * ci);
		if (ct2 < 0.0f) { L = 2 * dot(V, H) * H - V; return; }
		const float ct = (ci >= 0.0f ? 1.0f : -1.0f) * sqrtf(ct2);
		L = (eta * ci - ct) * H - eta * V;
	}
	float fval, pval, ratio;
	if (!ggx_terms(l, fr, V, L, H, NoV, eta, inv_eta, fval, pval, ratio)) return;
	p_proj = pval;
	p = p_proj * fabsf(dot(fr.n, L));
	g = ratio;
}

// ---- composite -------------------------------------------------------------------------------------------------------
// constructor (src/bsdf.h:218-243) from the texture-modulated material colours
FPT_HD SurfaceModel make_surface_model(f3 diffuse, f3 diffuse_trans, f3 specular, f3 reflectivity, float roughness, float ior, float opacity, const float* table)
{
	SurfaceModel m;
	m.kd = diffuse / kPi;
	m.kdt = diffuse_trans / kPi;
	m.ks = specular / kPi;
	m.coat = reflectivity;
	m.alpha = sel_max(roughness * 1.0f + 0.0f, 0.0f);
	m.inv_alpha = 1.0f / m.alpha;
	m.ior = ior;
	m.opacity = opacity;
	const float R0 = sel_min(max_comp(reflectivity), 0.95f);
	m.coat_ior = (1 + sqrtf(R0)) / (1 - sqrtf(R0));
	m.table = table;
	return m;
}

FPT_HD float directional_albedo(const SurfaceModel& m, float cos_theta)        // src/bsdf.h:1254-1268
{
	const uint32_t S = 32;
	const float eta = cos_theta > 0.0f ? 1.0f / m.ior : m.ior;
	const uint32_t ci = sel_min(S - 1u, to_u32_sat(fabsf(cos_theta) * float(S - 1)));
	const uint32_t bi = sel_min(S - 1u, to_u32_sat(max_comp(m.ks) * float(S - 1)));
	const uint32_t ei = sel_min(S - 1u, to_u32_sat((eta / 2.0f) * float(S - 1)));
	const uint32_t ri = sel_min(S - 1u, to_u32_sat(m.alpha * float(S - 1)));
	return m.table[ei * S * S * S + bi * S * S + ri * S + ci];
}

// clearcoat interface (src/bsdf.h:1202-1232 with contrib/cugar/bsdf/refraction.h:49-66,144-168): Fresnel reflection Fc and
// transmission Tc for direction w_i; false on total internal reflection
FPT_HD bool coat_interface(const SurfaceModel& m, const ShadingFrame& fr, f3 w_i, float& cos_i, f3& Fc, f3& Tc)
{
	const float R0 = sel_min(max_comp(m.coat), 0.95f);
	const float eta = 1.0f / m.coat_ior;
	cos_i = dot(w_i, fr.n);
	float F;
	if (eta == 1.0f) F = 0.0f;
	else
	{
		const float ct2 = 1.f - eta * eta * (1.f - cos_i * cos_i);
		if (ct2 < 0.0f) { Fc = splat3(1.0f); Tc = splat3(0.0f); return false; }
		const float ct = (cos_i >= 0.0f ? -1.0f : 1.0f) * sqrtf(ct2);
		const float a = fabsf(cos_i), b = fabsf(ct);
		const float Rs = (a - eta * b) / (a + eta * b);
		const float Rp = (eta * a - b) / (eta * a + b);
		F = 0.5f * (Rs * Rs + Rp * Rp);
	}
	Fc = lerp(m.coat, splat3(1.0f), sel_max(F - R0, 0.0f) / (1 - R0));
	Tc = splat3(1.0f) - Fc;
	return true;
}

FPT_HD float radiance_compression(const SurfaceModel& m, const ShadingFrame& fr, f3 w_i, f3 w_o)   // src/bsdf.h:1237-1251
{
	if (m.ior != 0.0f)
	{
		const float NoV = dot(w_i, fr.n), NoL = dot(w_o, fr.n);
		if (NoV * NoL < 0.0f) return sqr(NoV > 0.0f ? m.ior : 1.0f / m.ior);
	}
	return 1.0f;
}

// a-priori lobe selection weights (src/bsdf.h:530-586)
FPT_HD void lobe_prior_weights(const SurfaceModel& m, const ShadingFrame& fr, f3 V, float w[4])
{
	float r, t;
	if (m.ior == 0) { r = 0.0f; t = 1.0f; }
	else { r = directional_albedo(m, dot(fr.n, V)); t = 1.0f - r; }
	w[LOBE_GLOSSY_R] = r;
	w[LOBE_GLOSSY_T] = (1 - m.opacity) * t;
	w[LOBE_DIFF_R]   = m.opacity * max_comp(splat3(t) * m.kd) * kPi;
	w[LOBE_DIFF_T]   = m.opacity * max_comp(splat3(t) * m.kdt) * kPi;
}

// Fresnel split between the glossy layer and what it lets through (src/bsdf.h:632-664)
FPT_HD void layer_fresnel(const SurfaceModel& m, float VoH, float eta, f3& r, f3& t)
{
	if (eta == 0.0f) { r = splat3(0.0f); t = splat3(1.0f); }
	else { r = schlick(VoH, eta, m.ks); t = splat3(1.0f - max_comp(r)); }
}

// true per-lobe weights for a pair of directions, before the clearcoat factor (src/bsdf.h:666-743)
FPT_HD void inner_lobe_weights(const SurfaceModel& m, const ShadingFrame& fr, f3 V, f3 L, f3 w[4])
{
	float eta = 0.0f, inv_eta = 0.0f, VoH = 0.0f;
	if (m.ior != 0.0f)
	{
		const bool front = dot(fr.n, V) > 0.0f;
		eta     = front ? 1.0f / m.ior : m.ior;
		inv_eta = front ? m.ior : 1.0f / m.ior;
		const f3 H = half_vector_n(V, L, fr.n, inv_eta);
		VoH = dot(V, H);
	}
	f3 r, t;
	layer_fresnel(m, VoH, eta, r, t);
	const float dw = (1.0f - directional_albedo(m, dot(fr.n, V))) * (1.0f - directional_albedo(m, dot(fr.n, L)));
	w[LOBE_GLOSSY_R] = r;
	w[LOBE_GLOSSY_T] = t * (1 - m.opacity);
	w[LOBE_DIFF_R]   = t * m.opacity * dw;
	w[LOBE_DIFF_T]   = t * m.opacity * dw;
}

// per-lobe value and projected-solid-angle pdf (src/bsdf.h:366-412)
FPT_HD void surface_f_and_p(const SurfaceModel& m, const ShadingFrame& fr, f3 w_i, f3 w_o, f3 f[4], float p[4])
{
	float cos_i; f3 Fc, Tc, w[4];
	if (coat_interface(m, fr, w_i, cos_i, Fc, Tc))
	{
		inner_lobe_weights(m, fr, w_i, w_o, w);
		const f3 T12 = Tc * (splat3(1.0f) - splat3(0.0f));
		for (int i = 0; i < 4; ++i) w[i] = w[i] * T12;
	}
	else
		w[0] = w[1] = w[2] = w[3] = splat3(0.0f);
	const float coat_T = 1.0f - average(Fc);

	const float NoL = dot(fr.n, w_o), NoV = dot(fr.n, w_i);
	const bool same_side = NoL * NoV > 0.0f, opp_side = NoL * NoV < 0.0f;
	const f3 f_d  = same_side ? m.kd : splat3(0.0f);
	const f3 f_dt = opp_side ? m.kdt : splat3(0.0f);
	const float p_d  = same_side ? 1.0f / kPi : 0.0f;
	const float p_dt = opp_side ? 1.0f / kPi : 0.0f;
	float f_g, p_g, f_gt, p_gt;
	ggx_eval(ggx_reflective(m.alpha), fr, w_i, w_o, f_g, p_g);
	ggx_eval(ggx_transmissive(m.alpha, m.ior), fr, w_i, w_o, f_gt, p_gt);

	float wp[4];
	lobe_prior_weights(m, fr, w_i, wp);
	p[LOBE_DIFF_R]   = p_d  * (wp[LOBE_DIFF_R] * coat_T);
	p[LOBE_DIFF_T]   = p_dt * (wp[LOBE_DIFF_T] * coat_T);
	p[LOBE_GLOSSY_R] = p_g  * (wp[LOBE_GLOSSY_R] * coat_T);
	p[LOBE_GLOSSY_T] = p_gt * (wp[LOBE_GLOSSY_T] * coat_T);

	const float factor = radiance_compression(m, fr, w_i, w_o);
	f[LOBE_DIFF_R]   = f_d  * w[LOBE_DIFF_R] * factor;
	f[LOBE_DIFF_T]   = f_dt * w[LOBE_DIFF_T] * factor;
	f[LOBE_GLOSSY_R] = splat3(f_g)  * w[LOBE_GLOSSY_R] * factor;
	f[LOBE_GLOSSY_T] = splat3(f_gt) * w[LOBE_GLOSSY_T] * factor;
}

// scattering with Russian roulette (src/bsdf.h:921-1199).  Returns the component id (COMP_ABSORB when the path dies).
FPT_HD uint32_t surface_sample(const SurfaceModel& m, const ShadingFrame& fr, float z0, float z1, float z2, f3 in,
                               f3& out, float& out_p, float& out_p_proj, f3& out_g)
{
	out = splat3(0.0f); out_p = 0.0f; out_p_proj = 0.0f; out_g = splat3(0.0f);
	float cos_i; f3 Fc, Tc;
	if (!coat_interface(m, fr, in, cos_i, Fc, Tc)) return COMP_ABSORB;
	const float coat_R = average(Fc);
	const float coat_T = 1.0f - coat_R;

	float wp[4];
	lobe_prior_weights(m, fr, in, wp);
	// one visible-normal sample refines the priors with the Fresnel term at the sampled microfacet (:996-1035)
	const f3 Vl = to_local(fr, in);
	const float sg = Vl.z >= 0.0f ? 1.0f : -1.0f;
	f3 Hl = sample_vndf(z0, z1, m.alpha, mk3(Vl.x, Vl.y, Vl.z * sg));
	Hl.z *= sg;
	const f3 H = from_local(fr, Hl);
	f3 r, t;
	layer_fresnel(m, dot(Vl, Hl), Vl.z > 0.0f ? 1.0f / m.ior : m.ior, r, t);
	wp[LOBE_GLOSSY_R] = (wp[LOBE_GLOSSY_R] + max_comp(r)) * 0.5f;
	wp[LOBE_GLOSSY_T] = (wp[LOBE_GLOSSY_T] + (1 - m.opacity) * max_comp(t)) * 0.5f;
	wp[LOBE_DIFF_R]   = (wp[LOBE_DIFF_R] + m.opacity * max_comp(t * m.kd) * kPi) * 0.5f;
	wp[LOBE_DIFF_T]   = (wp[LOBE_DIFF_T] + m.opacity * max_comp(t * m.kdt) * kPi) * 0.5f;
	const float s0 = wp[LOBE_DIFF_R] * coat_T, s1 = wp[LOBE_GLOSSY_R] * coat_T, s2 = wp[LOBE_DIFF_T] * coat_T, s3 = wp[LOBE_GLOSSY_T] * coat_T;

	// thresholds are accumulated left to right exactly like the reference's else-if chain (:1041-1125)
	uint32_t comp; float p_comp;
	if      (z2 < s0)                        { comp = COMP_DIFF_R;   p_comp = s0; }
	else if (z2 < s0 + s1)                   { comp = COMP_GLOSSY_R; p_comp = s1; }
	else if (z2 < s0 + s1 + s2)              { comp = COMP_DIFF_T;   p_comp = s2; }
	else if (z2 < s0 + s1 + s2 + s3)         { comp = COMP_GLOSSY_T; p_comp = s3; }
	else if (z2 < s0 + s1 + s2 + s3 + coat_R){ comp = COMP_COAT;     p_comp = coat_R; }
	else return COMP_ABSORB;

	if (comp == COMP_COAT)
	{
		out = 2 * cos_i * fr.n - in;
		out_g = (Fc / p_comp) * radiance_compression(m, fr, in, out);
		out_p = inf_f(); out_p_proj = inf_f();
		return comp;
	}

	f3 L = splat3(0.0f), g; float p, p_proj;
	if (comp & COMP_DIFFUSE_MASK)
	{
		f3 l = cosine_hemisphere(z0, z1);
		const float NoV = dot(in, fr.n);
		if ((comp == COMP_DIFF_R) ? (NoV < 0.0f) : (NoV > 0.0f)) l.z = -l.z;
		L = l.x * fr.t + l.y * fr.b + l.z * fr.n;
		g = ((comp == COMP_DIFF_R) ? m.kd : m.kdt) * kPi;
		p = fabsf(l.z) / kPi;
		p_proj = 1.0f / kPi;
	}
	else
	{
		float gs;
		ggx_sample_given_h((comp == COMP_GLOSSY_R) ? ggx_reflective(m.alpha) : ggx_transmissive(m.alpha, m.ior), fr, H, in, L, gs, p, p_proj);
		g = splat3(gs);
	}
	g = g * (Tc * (splat3(1.0f) - splat3(0.0f)));
	out = L;
	f3 w[4];
	inner_lobe_weights(m, fr, in, out, w);
	g = g * ((comp & COMP_GLOSSY_R) ? w[LOBE_GLOSSY_R] : (comp & COMP_GLOSSY_T) ? w[LOBE_GLOSSY_T] : (comp & COMP_DIFF_R) ? w[LOBE_DIFF_R] : w[LOBE_DIFF_T]);
	g = g / p_comp;
	out_p = p * p_comp;
	out_p_proj = p_proj * p_comp;
	out_g = g * radiance_compression(m, fr, in, out);
	return comp;
}


// ---- the path tracer evaluates the BSDF twice per vertex with the same incoming direction (next-event estimation, then scattering):
// everything that depends on that direction alone -- the clearcoat interface, the a-priori lobe weights, the directional albedo
// towards the viewer -- is computed once here and handed to both.  Same operations on the same values: results are unchanged.
struct ViewTerms { bool coat_ok; float cos_i; f3 Fc, Tc; float wp[4]; float albedo_v; };
FPT_HD ViewTerms view_terms(const SurfaceModel& m, const ShadingFrame& fr, f3 V)
{
	ViewTerms v;
	v.coat_ok = coat_interface(m, fr, V, v.cos_i, v.Fc, v.Tc);
	lobe_prior_weights(m, fr, V, v.wp);
	v.albedo_v = directional_albedo(m, dot(fr.n, V));
	return v;
}
// inner_lobe_weights with the viewer-side albedo supplied
FPT_HD void inner_lobe_weights(const SurfaceModel& m, const ShadingFrame& fr, f3 V, f3 L, float albedo_v, f3 w[4])
{
	float eta = 0.0f, inv_eta = 0.0f, VoH = 0.0f;
	if (m.ior != 0.0f)
	{
		const bool front = dot(fr.n, V) > 0.0f;
		eta     = front ? 1.0f / m.ior : m.ior;
		inv_eta = front ? m.ior : 1.0f / m.ior;
		const f3 H = half_vector_n(V, L, fr.n, inv_eta);
		VoH = dot(V, H);
	}
	f3 r, t;
	layer_fresnel(m, VoH, eta, r, t);
	const float dw = (1.0f - albedo_v) * (1.0f - directional_albedo(m, dot(fr.n, L)));
	w[LOBE_GLOSSY_R] = r;
	w[LOBE_GLOSSY_T] = t * (1 - m.opacity);
	w[LOBE_DIFF_R]   = t * m.opacity * dw;
	w[LOBE_DIFF_T]   = t * m.opacity * dw;
}

// surface_f_and_p with the view terms supplied
FPT_HD void surface_f_and_p(const SurfaceModel& m, const ShadingFrame& fr, const ViewTerms& vt, f3 w_i, f3 w_o, f3 f[4], float p[4])
{
	const f3 Fc = vt.Fc, Tc = vt.Tc; f3 w[4];
	if (vt.coat_ok)
	{
		inner_lobe_weights(m, fr, w_i, w_o, vt.albedo_v, w);
		const f3 T12 = Tc * (splat3(1.0f) - splat3(0.0f));
		for (int i = 0; i < 4; ++i) w[i] = w[i] * T12;
	}
	else
		w[0] = w[1] = w[2] = w[3] = splat3(0.0f);
	const float coat_T = 1.0f - average(Fc);

	const float NoL = dot(fr.n, w_o), NoV = dot(fr.n, w_i);
	const bool same_side = NoL * NoV > 0.0f, opp_side = NoL * NoV < 0.0f;
	const f3 f_d  = same_side ? m.kd : splat3(0.0f);
	const f3 f_dt = opp_side ? m.kdt : splat3(0.0f);
	const float p_d  = same_side ? 1.0f / kPi : 0.0f;
	const float p_dt = opp_side ? 1.0f / kPi : 0.0f;
	float f_g, p_g, f_gt = 0.0f, p_gt = 0.0f;
	ggx_eval(ggx_reflective(m.alpha), fr, w_i, w_o, f_g, p_g);
	// An opaque surface (opacity == 1, nearly every material) gives the transmissive GGX lobe the weight t * (1 - 1) = +0 and the prior (1 - 1) * t = +0, and the
	// lobe's value and pdf are finite and non-negative (clamp_pdf): both products below are exactly +0 whatever the lobe evaluates to.  Skipping the evaluation
	// (a second half-vector, D, G, G1, the refraction Jacobian: ~8 IEEE divisions) changes no bit; a wave skips it when all of its surfaces are opaque.
#ifndef FPT_PROF_OPAQUE
	if (m.opacity != 1.0f) ggx_eval(ggx_transmissive(m.alpha, m.ior), fr, w_i, w_o, f_gt, p_gt);
#endif

	const float* wp = vt.wp;
	p[LOBE_DIFF_R]   = p_d  * (wp[LOBE_DIFF_R] * coat_T);
	p[LOBE_DIFF_T]   = p_dt * (wp[LOBE_DIFF_T] * coat_T);
	p[LOBE_GLOSSY_R] = p_g  * (wp[LOBE_GLOSSY_R] * coat_T);
	p[LOBE_GLOSSY_T] = p_gt * (wp[LOBE_GLOSSY_T] * coat_T);

	const float factor = radiance_compression(m, fr, w_i, w_o);
	f[LOBE_DIFF_R]   = f_d  * w[LOBE_DIFF_R] * factor;
	f[LOBE_DIFF_T]   = f_dt * w[LOBE_DIFF_T] * factor;
	f[LOBE_GLOSSY_R] = splat3(f_g)  * w[LOBE_GLOSSY_R] * factor;
	f[LOBE_GLOSSY_T] = splat3(f_gt) * w[LOBE_GLOSSY_T] * factor;
}

// surface_sample with the view terms supplied
FPT_HD uint32_t surface_sample(const SurfaceModel& m, const ShadingFrame& fr, const ViewTerms& vt, float z0, float z1, float z2, f3 in,
                               f3& out, float& out_p, float& out_p_proj, f3& out_g)
{
	out = splat3(0.0f); out_p = 0.0f; out_p_proj = 0.0f; out_g = splat3(0.0f);
	if (!vt.coat_ok) return COMP_ABSORB;
	const float cos_i = vt.cos_i; const f3 Fc = vt.Fc, Tc = vt.Tc;
	const float coat_R = average(Fc);
	const float coat_T = 1.0f - coat_R;

	float wp[4] = { vt.wp[0], vt.wp[1], vt.wp[2], vt.wp[3] };
	// one visible-normal sample refines the priors with the Fresnel term at the sampled microfacet (:996-1035)
	const f3 Vl = to_local(fr, in);
	const float sg = Vl.z >= 0.0f ? 1.0f : -1.0f;
	f3 Hl = sample_vndf(z0, z1, m.alpha, mk3(Vl.x, Vl.y, Vl.z * sg));
	Hl.z *= sg;
	const f3 H = from_local(fr, Hl);
	f3 r, t;
	layer_fresnel(m, dot(Vl, Hl), Vl.z > 0.0f ? 1.0f / m.ior : m.ior, r, t);
	wp[LOBE_GLOSSY_R] = (wp[LOBE_GLOSSY_R] + max_comp(r)) * 0.5f;
	wp[LOBE_GLOSSY_T] = (wp[LOBE_GLOSSY_T] + (1 - m.opacity) * max_comp(t)) * 0.5f;
	wp[LOBE_DIFF_R]   = (wp[LOBE_DIFF_R] + m.opacity * max_comp(t * m.kd) * kPi) * 0.5f;
	wp[LOBE_DIFF_T]   = (wp[LOBE_DIFF_T] + m.opacity * max_comp(t * m.kdt) * kPi) * 0.5f;
	const float s0 = wp[LOBE_DIFF_R] * coat_T, s1 = wp[LOBE_GLOSSY_R] * coat_T, s2 = wp[LOBE_DIFF_T] * coat_T, s3 = wp[LOBE_GLOSSY_T] * coat_T;

	// thresholds are accumulated left to right exactly like the reference's else-if chain (:1041-1125)
	uint32_t comp; float p_comp;
	if      (z2 < s0)                        { comp = COMP_DIFF_R;   p_comp = s0; }
	else if (z2 < s0 + s1)                   { comp = COMP_GLOSSY_R; p_comp = s1; }
	else if (z2 < s0 + s1 + s2)              { comp = COMP_DIFF_T;   p_comp = s2; }
	else if (z2 < s0 + s1 + s2 + s3)         { comp = COMP_GLOSSY_T; p_comp = s3; }
	else if (z2 < s0 + s1 + s2 + s3 + coat_R){ comp = COMP_COAT;     p_comp = coat_R; }
	else return COMP_ABSORB;

	if (comp == COMP_COAT)
	{
		out = 2 * cos_i * fr.n - in;
		out_g = (Fc / p_comp) * radiance_compression(m, fr, in, out);
		out_p = inf_f(); out_p_proj = inf_f();
		return comp;
	}

	f3 L = splat3(0.0f), g; float p, p_proj;
	if (comp & COMP_DIFFUSE_MASK)
	{
		f3 l = cosine_hemisphere(z0, z1);
		const float NoV = dot(in, fr.n);
		if ((comp == COMP_DIFF_R) ? (NoV < 0.0f) : (NoV > 0.0f)) l.z = -l.z;
		L = l.x * fr.t + l.y * fr.b + l.z * fr.n;
		g = ((comp == COMP_DIFF_R) ? m.kd : m.kdt) * kPi;
		p = fabsf(l.z) / kPi;
		p_proj = 1.0f / kPi;
	}
	else
	{
		float gs;
		ggx_sample_given_h((comp == COMP_GLOSSY_R) ? ggx_reflective(m.alpha) : ggx_transmissive(m.alpha, m.ior), fr, H, in, L, gs, p, p_proj);
		g = splat3(gs);
	}
	g = g * (Tc * (splat3(1.0f) - splat3(0.0f)));
	out = L;
	f3 w[4];
	inner_lobe_weights(m, fr, in, out, vt.albedo_v, w);
	g = g * ((comp & COMP_GLOSSY_R) ? w[LOBE_GLOSSY_R] : (comp & COMP_GLOSSY_T) ? w[LOBE_GLOSSY_T] : (comp & COMP_DIFF_R) ? w[LOBE_DIFF_R] : w[LOBE_DIFF_T]);
	g = g / p_comp;
	out_p = p * p_comp;
	out_p_proj = p_proj * p_comp;
	out_g = g * radiance_compression(m, fr, in, out);
	return comp;
}



// ---- variants used by the bidirectional path tracer -------------------------------------------------------------------------
// TransportType (src/bsdf.h:81-87): the (eta_t/eta_i)^2 radiance compression applies to eye vertices only
FPT_HD float transport_factor(const SurfaceModel& m, const ShadingFrame& fr, f3 w_i, f3 w_o, bool particle)
{ return particle ? 1.0f : radiance_compression(m, fr, w_i, w_o); }

// stored light vertices rebuild their BSDF from 16 packed bytes (second constructor, src/bsdf.h:246-273); the reference reads an
// uninitialised reflectivity there — defined as zero (no clearcoat) here and in the oracle
FPT_HD SurfaceModel make_surface_model_unpacked(f3 diffuse, f3 specular, float roughness, f3 diffuse_trans, float opacity, float ior, const float* table)
{
	SurfaceModel m;
	m.kd = diffuse / kPi; m.kdt = diffuse_trans / kPi; m.ks = specular / kPi; m.coat = splat3(0.0f);
	m.alpha = roughness; m.inv_alpha = 1.0f / m.alpha; m.ior = ior; m.opacity = opacity;
	const float R0 = sel_min(max_comp(m.coat), 0.95f);
	m.coat_ior = (1 + sqrtf(R0)) / (1 - sqrtf(R0));
	m.table = table;
	return m;
}

// normalize_sampling_weights with the RR switch (src/bsdf.h:591-627): without Russian roulette everything is renormalised
FPT_HD void scale_lobe_weights(float wp[4], float& coat_R, float& coat_T, bool RR)
{
	wp[LOBE_DIFF_R] *= coat_T; wp[LOBE_DIFF_T] *= coat_T; wp[LOBE_GLOSSY_R] *= coat_T; wp[LOBE_GLOSSY_T] *= coat_T;
	if (!RR)
	{
		const float inv_sum = 1.0f / (wp[LOBE_DIFF_R] + wp[LOBE_DIFF_T] + wp[LOBE_GLOSSY_R] + wp[LOBE_GLOSSY_T] + coat_R);
		wp[LOBE_DIFF_R] *= inv_sum; wp[LOBE_DIFF_T] *= inv_sum; wp[LOBE_GLOSSY_R] *= inv_sum; wp[LOBE_GLOSSY_T] *= inv_sum;
		coat_R *= inv_sum; coat_T *= inv_sum;
	}
}

struct LobeEval { f3 f_d, f_dt; float f_g, f_gt, p_d, p_dt, p_g, p_gt; };
FPT_HD LobeEval eval_lobes(const SurfaceModel& m, const ShadingFrame& fr, f3 w_i, f3 w_o)
{
	LobeEval e;
	const float NoL = dot(fr.n, w_o), NoV = dot(fr.n, w_i);
	const bool same_side = NoL * NoV > 0.0f, opp_side = NoL * NoV < 0.0f;
	e.f_d  = same_side ? m.kd : splat3(0.0f);
	e.f_dt = opp_side ? m.kdt : splat3(0.0f);
	e.p_d  = same_side ? 1.0f / kPi : 0.0f;
	e.p_dt = opp_side ? 1.0f / kPi : 0.0f;
	ggx_eval(ggx_reflective(m.alpha), fr, w_i, w_o, e.f_g, e.p_g);
	ggx_eval(ggx_transmissive(m.alpha, m.ior), fr, w_i, w_o, e.f_gt, e.p_gt);
	return e;
}
FPT_HD void coated_lobe_weights(const SurfaceModel& m, const ShadingFrame& fr, f3 w_i, f3 w_o, f3& Fc, f3 w[4])
{
	float cos_i; f3 Tc;
	if (coat_interface(m, fr, w_i, cos_i, Fc, Tc))
	{
		inner_lobe_weights(m, fr, w_i, w_o, w);
		const f3 T12 = Tc * (splat3(1.0f) - splat3(0.0f));
		for (int i = 0; i < 4; ++i) w[i] = w[i] * T12;
	}
	else
		w[0] = w[1] = w[2] = w[3] = splat3(0.0f);
}
// summed value and pdf (src/bsdf.h:417-464)
FPT_HD void surface_f_and_p_sum(const SurfaceModel& m, const ShadingFrame& fr, f3 w_i, f3 w_o, bool RR, bool particle, f3& f, float& p)
{
	f3 Fc, w[4];
	coated_lobe_weights(m, fr, w_i, w_o, Fc, w);
	float coat_R = average(Fc);
	float coat_T = 1.0f - coat_R;
	const LobeEval e = eval_lobes(m, fr, w_i, w_o);
	float wp[4];
	lobe_prior_weights(m, fr, w_i, wp);
	scale_lobe_weights(wp, coat_R, coat_T, RR);
	p = e.p_d * wp[LOBE_DIFF_R] + e.p_dt * wp[LOBE_DIFF_T] + e.p_g * wp[LOBE_GLOSSY_R] + e.p_gt * wp[LOBE_GLOSSY_T];
	const float factor = transport_factor(m, fr, w_i, w_o, particle);
	f = e.f_d * w[LOBE_DIFF_R] * factor + e.f_dt * w[LOBE_DIFF_T] * factor + splat3(e.f_g) * w[LOBE_GLOSSY_R] * factor + splat3(e.f_gt) * w[LOBE_GLOSSY_T] * factor;
}
// value only (src/bsdf.h:296-318)
FPT_HD f3 surface_f_sum(const SurfaceModel& m, const ShadingFrame& fr, f3 w_i, f3 w_o, bool particle)
{
	f3 Fc, w[4];
	coated_lobe_weights(m, fr, w_i, w_o, Fc, w);
	const float factor = transport_factor(m, fr, w_i, w_o, particle);
	const LobeEval e = eval_lobes(m, fr, w_i, w_o);
	return e.f_d * w[LOBE_DIFF_R] * factor + e.f_dt * w[LOBE_DIFF_T] * factor + splat3(e.f_g) * w[LOBE_GLOSSY_R] * factor + splat3(e.f_gt) * w[LOBE_GLOSSY_T] * factor;
}
// pdf only (src/bsdf.h:466-528)
FPT_HD float surface_p_sum(const SurfaceModel& m, const ShadingFrame& fr, f3 w_i, f3 w_o, bool RR)
{
	float cos_i; f3 Fc, Tc;
	if (!coat_interface(m, fr, w_i, cos_i, Fc, Tc)) return 0.0f;
	float coat_R = average(Fc);
	float coat_T = 1.0f - coat_R;
	float wp[4];
	lobe_prior_weights(m, fr, w_i, wp);
	scale_lobe_weights(wp, coat_R, coat_T, RR);
	const LobeEval e = eval_lobes(m, fr, w_i, w_o);
	return e.p_d * wp[LOBE_DIFF_R] + e.p_dt * wp[LOBE_DIFF_T] + e.p_g * wp[LOBE_GLOSSY_R] + e.p_gt * wp[LOBE_GLOSSY_T];
}

// scattering with the RR switch and the "evaluate the full BSDF" option of the bidirectional tracer (src/bsdf.h:921-1199)
FPT_HD uint32_t surface_sample_ex(const SurfaceModel& m, const ShadingFrame& fr, float z0, float z1, float z2, f3 in, bool RR, bool full, bool particle,
                                  f3& out, float& out_p, float& out_p_proj, f3& out_g)
{
	out = splat3(0.0f); out_p = 0.0f; out_p_proj = 0.0f; out_g = splat3(0.0f);
	float cos_i; f3 Fc, Tc;
	if (!coat_interface(m, fr, in, cos_i, Fc, Tc)) return COMP_ABSORB;
	float coat_R = average(Fc);
	float coat_T = 1.0f - coat_R;

	float wp[4];
	lobe_prior_weights(m, fr, in, wp);
	const f3 Vl = to_local(fr, in);
	const float sg = Vl.z >= 0.0f ? 1.0f : -1.0f;
	f3 Hl = sample_vndf(z0, z1, m.alpha, mk3(Vl.x, Vl.y, Vl.z * sg));
	Hl.z *= sg;
	const f3 H = from_local(fr, Hl);
	f3 r, t;
	layer_fresnel(m, dot(Vl, Hl), Vl.z > 0.0f ? 1.0f / m.ior : m.ior, r, t);
	wp[LOBE_GLOSSY_R] = (wp[LOBE_GLOSSY_R] + max_comp(r)) * 0.5f;
	wp[LOBE_GLOSSY_T] = (wp[LOBE_GLOSSY_T] + (1 - m.opacity) * max_comp(t)) * 0.5f;
	wp[LOBE_DIFF_R]   = (wp[LOBE_DIFF_R] + m.opacity * max_comp(t * m.kd) * kPi) * 0.5f;
	wp[LOBE_DIFF_T]   = (wp[LOBE_DIFF_T] + m.opacity * max_comp(t * m.kdt) * kPi) * 0.5f;
	scale_lobe_weights(wp, coat_R, coat_T, RR);
	const float s0 = wp[LOBE_DIFF_R], s1 = wp[LOBE_GLOSSY_R], s2 = wp[LOBE_DIFF_T], s3 = wp[LOBE_GLOSSY_T];

	uint32_t comp; float p_comp;
	if      (z2 < s0)                        { comp = COMP_DIFF_R;   p_comp = s0; }
	else if (z2 < s0 + s1)                   { comp = COMP_GLOSSY_R; p_comp = s1; }
	else if (z2 < s0 + s1 + s2)              { comp = COMP_DIFF_T;   p_comp = s2; }
	else if (z2 < s0 + s1 + s2 + s3)         { comp = COMP_GLOSSY_T; p_comp = s3; }
	else if (z2 < s0 + s1 + s2 + s3 + coat_R){ comp = COMP_COAT;     p_comp = coat_R; }
	else return COMP_ABSORB;

	if (comp == COMP_COAT)
	{
		out = 2 * cos_i * fr.n - in;
		out_g = (Fc / p_comp) * transport_factor(m, fr, in, out, particle);
		out_p = inf_f(); out_p_proj = inf_f();
		return comp;
	}

	f3 L = splat3(0.0f), g; float p, p_proj;
	if (comp & COMP_DIFFUSE_MASK)
	{
		f3 l = cosine_hemisphere(z0, z1);
		const float NoV = dot(in, fr.n);
		if ((comp == COMP_DIFF_R) ? (NoV < 0.0f) : (NoV > 0.0f)) l.z = -l.z;
		L = l.x * fr.t + l.y * fr.b + l.z * fr.n;
		g = ((comp == COMP_DIFF_R) ? m.kd : m.kdt) * kPi;
		p = fabsf(l.z) / kPi;
		p_proj = 1.0f / kPi;
	}
	else
	{
		float gs;
		ggx_sample_given_h((comp == COMP_GLOSSY_R) ? ggx_reflective(m.alpha) : ggx_transmissive(m.alpha, m.ior), fr, H, in, L, gs, p, p_proj);
		g = splat3(gs);
	}
	g = g * (Tc * (splat3(1.0f) - splat3(0.0f)));
	out = L;
	if (full)
	{
		// every lobe's pdf for the sampled direction; the lobe probabilities get the coat transmission a second time, as the reference has it (:1147-1162)
		const LobeEval e = eval_lobes(m, fr, in, out);
		p_proj = e.p_d * wp[LOBE_DIFF_R] * coat_T + e.p_dt * wp[LOBE_DIFF_T] * coat_T + e.p_g * wp[LOBE_GLOSSY_R] * coat_T + e.p_gt * wp[LOBE_GLOSSY_T] * coat_T;
		p = p_proj * fabsf(dot(out, fr.n));
		g = surface_f_sum(m, fr, in, out, particle) / p_proj;
	}
	else
	{
		f3 w[4];
		inner_lobe_weights(m, fr, in, out, w);
		g = g * ((comp & COMP_GLOSSY_R) ? w[LOBE_GLOSSY_R] : (comp & COMP_GLOSSY_T) ? w[LOBE_GLOSSY_T] : (comp & COMP_DIFF_R) ? w[LOBE_DIFF_R] : w[LOBE_DIFF_T]);
		g = g / p_comp;
		p = p * p_comp;
		p_proj = p_proj * p_comp;
	}
	out_p = p;
	out_p_proj = p_proj;
	out_g = g * transport_factor(m, fr, in, out, particle);
	return comp;
}

} // namespace fpt
