// ORACLE — TEST INFRASTRUCTURE ONLY (see o_math.h header).
//
// o_bsdf.h : CPU restatement of Fermat's layered BSDF as used by the -pt renderer.
//   inner lobes : contrib/cugar/bsdf/{lambert.h, lambert_trans.h, ggx_smith.h, ggx_common.h, refraction.h}
//   composite   : src/bsdf.h (USE_GGX_SMITH, USE_APPROX_SMITH, USE_EFFICIENT_SAMPLER_WITH_APPROXIMATE_PDFS = 1)
//   emitter     : contrib/cugar/bsdf/lambert_edf.h, src/edf.h
// Only the kProjectedSolidAngle measure and the kAllComponents mask are restated: these are the only
// ones shade_vertex (src/pathtracer_core.h:906,1024,1190) reaches with default -diffuse/-glossy flags.
#pragma once
#include "o_math.h"

namespace orc {

// contrib/cugar/bsdf/differential_geometry.h:54-89
struct Frame
{
	V3 normal_s, normal_g, tangent, binormal;
	V3 to_local(V3 v) const { return V3(dot(v, tangent), dot(v, binormal), dot(v, normal_s)); }
	V3 from_local(V3 v) const { return v.x * tangent + v.y * binormal + v.z * normal_s; }
};

// contrib/cugar/spherical/mappings_inline.h:56-87
inline V2 square_to_unit_disk(float sx, float sy)
{
	float phi, r;
	const float a = 2 * sx - 1;
	const float b = 2 * sy - 1;
	if (a > -b)
	{
		if (a > b) { r = a;  phi = (PI_F / 4) * (b / a); }
		else       { r = b;  phi = (PI_F / 4) * (2 - (a / b)); }
	}
	else
	{
		if (a < b) { r = -a; phi = (PI_F / 4) * (4 + (b / a)); }
		else       { r = -b; phi = b != 0 ? (PI_F / 4) * (6 - (a / b)) : 0; }
	}
	float s, c;
	det_sincos(phi, &s, &c);
	V2 d; d.x = r * c; d.y = r * s;
	return d;
}
// contrib/cugar/spherical/mappings_inline.h:119-126
inline V3 square_to_cosine_hemisphere(float sx, float sy)
{
	const V2 d = square_to_unit_disk(sx, sy);
	const float r2 = d.x * d.x + d.y * d.y;
	return V3(d.x, d.y, sqrtf(maxf(1.0f - r2, 0.0f)));
}

// contrib/cugar/bsdf/refraction.h:91-94
inline float pow5(float x) { const float x2 = x * x; return x2 * x2 * x; }
// contrib/cugar/bsdf/refraction.h:96-118
inline V3 fresnel_schlick(float cos_theta_i, float eta, V3 base)
{
	cos_theta_i = saturate(fabsf(cos_theta_i));
	const float cos_theta_t2 = saturate(1.f - eta * eta * (1.f - cos_theta_i * cos_theta_i));
	if (cos_theta_t2 < 0.0f) return V3(1.0f);
	const float cos_theta = eta > 1.0f ? sqrtf(cos_theta_t2) : cos_theta_i;
	const float Fc = pow5(1 - cos_theta);
	return V3(Fc) + (1 - Fc) * base;
}
// contrib/cugar/bsdf/refraction.h:49-66 (scalar eta)
inline float fresnel_dielectric(float ci, float ct, float eta)
{
	if (eta == 1.0f) return 0.0f;
	const float Rs = (ci - eta * ct) / (ci + eta * ct);
	const float Rp = (eta * ci - ct) / (eta * ci + ct);
	return 0.5f * (Rs * Rs + Rp * Rp);
}
// contrib/cugar/bsdf/refraction.h:144-168
inline bool refract(V3 w_i, V3 N, float cos_theta_i, float eta, V3* out, float* F)
{
	if (eta == 1.0f) { *out = -w_i; *F = 0.0f; return true; }
	const float cos_theta_t2 = 1.f - eta * eta * (1.f - cos_theta_i * cos_theta_i);
	if (cos_theta_t2 < 0.0f) return false;
	const float cos_theta_t = (cos_theta_i >= 0.0f ? -1.0f : 1.0f) * sqrtf(cos_theta_t2);
	*F = fresnel_dielectric(fabsf(cos_theta_i), fabsf(cos_theta_t), eta);
	*out = (eta * cos_theta_i + cos_theta_t) * N - eta * w_i;
	return true;
}

// contrib/cugar/bsdf/ggx_common.h:50-66
inline V3 microfacet(V3 V, V3 L, V3 N, float inv_eta)
{
	V3 H = (dot(V, N) * dot(L, N) >= 0.0f) ? V + L : V + L * inv_eta;
	if (dot(H, H) == 0.0f) return N;
	if (dot(N, H) < 0.0f) H = -H;
	return normalize(H);
}
// contrib/cugar/bsdf/ggx_common.h:68-84
inline V3 vndf_microfacet(V3 V, V3 L, V3 N, float inv_eta)
{
	V3 H = (dot(V, N) * dot(L, N) >= 0.0f) ? V + L : V + L * inv_eta;
	if (dot(H, H) < 1.0e-12f) return N;
	if (dot(V, H) < 0.0f) H = -H;
	return normalize(H);
}
// contrib/cugar/bsdf/ggx_common.h:86-106 (isotropic: inv_alpha.x == inv_alpha.y)
inline float hvd_ggx_eval(float inv_alpha, float nh, float ht, float hb)
{
	const float x = ht * inv_alpha;
	const float y = hb * inv_alpha;
	const float aniso = x * x + y * y;
	const float f = aniso + nh * nh;
	return (1.0f / PI_F) * inv_alpha * inv_alpha / (f * f);
}
// contrib/cugar/bsdf/ggx_common.h:265-290
inline V3 vndf_ggx_smith_sample(float u0, float u1, float alpha, V3 _V)
{
	V3 V = normalize(V3(alpha * _V.x, alpha * _V.y, _V.z));
	V3 T1 = (V.z < 0.9999f) ? normalize(cross(V, V3(0, 0, 1))) : V3(1, 0, 0);
	V3 T2 = cross(T1, V);
	const float a = 1.0f / (1.0f + V.z);
	const float r = sqrtf(u0);
	const float phi = (u1 < a) ? u1 / a * PI_F : PI_F + (u1 - a) / (1.0f - a) * PI_F;
	float sp, cp;
	det_sincos(phi, &sp, &cp);
	const float P1 = r * cp;
	const float P2 = r * sp * ((u1 < a) ? 1.0f : V.z);
	V3 N = P1 * T1 + P2 * T2 + sqrtf(maxf(0.0f, 1.0f - P1 * P1 - P2 * P2)) * V;
	N = normalize(V3(alpha * N.x, alpha * N.y, maxf(0.0f, N.z)));
	return N;
}

// contrib/cugar/bsdf/ggx_common.h:296-392: the inverse of vndf_ggx_smith_sample -- (u0, u1) such that the sampler returns the micro-normal _N.
// Off the render path: restated for the reference's own unit test (contrib/cugar/bsdf/bsdf_test.h:64-91), which round-trips sample -> invert -> sample.
inline V2 vndf_ggx_smith_invert(V3 _N, float alpha, V3 _V)
{
	V3 V = normalize(V3(alpha * _V.x, alpha * _V.y, _V.z));
	V3 T1 = (V.z < 0.9999f) ? normalize(cross(V, V3(0, 0, 1))) : V3(1, 0, 0);
	V3 T2 = cross(T1, V);
	// the point of the ray <N> on the ellipsoid (alpha X)^2 + (alpha Y)^2 + Z^2 = 1
	V3 N = normalize(V3(_N.x / alpha, _N.y / alpha, _N.z));
	const float P1 = dot(N, T1), P2 = dot(N, T2);
	const float a = 1.0f / (1.0f + V.z);
	const float TWO_PI = 2.0f * PI_F;
	V2 smp;
	// N projects along V either onto the half disk orthogonal to V (u1 < a) or onto the tangent half disk (u1 >= a)
	const V3 PN = N - dot(N, V) * V;
	if (PN.z > 0.0f)
	{
		const float r2 = minf(P1 * P1 + P2 * P2, 1.0f), r = sqrtf(r2);
		float phi = r > 1.0e-6f ? det_atan2(P2 / r, P1 / r) : 0.0f;
		if (phi < 0.0f) phi += TWO_PI;
		smp.x = r2; smp.y = phi * a / PI_F;
		if (phi < PI_F) return smp;
	}
	{
		const float r2 = minf(P1 * P1 + P2 * P2 / (V.z * V.z), 1.0f), r = sqrtf(r2);
		float phi = (r > 1.0e-6f && V.z > 1.0e-6f) ? det_atan2(P2 / (r * V.z), P1 / r) : 0.0f;
		if (phi < 0.0f) phi += TWO_PI;
		smp.x = r2; smp.y = (phi - PI_F) * (1.0f - a) / PI_F + a;
	}
	return smp;
}

// contrib/cugar/bsdf/lambert.h:49-210 and lambert_trans.h:51-140 (TRANS flips the hemisphere tests)
struct Lambert
{
	V3 color;
	bool trans;
	void f_and_p(const Frame& g, V3 V, V3 L, V3& f, float& p) const
	{
		const float NoL = dot(g.normal_s, L), NoV = dot(g.normal_s, V);
		const bool on = trans ? (NoL * NoV < 0.0f) : (NoL * NoV > 0.0f);
		f = on ? color : V3(0.0f);
		p = on ? 1.0f / PI_F : 0.0f;
	}
	void sample(float u0, float u1, const Frame& g, V3 V, V3& L, V3& gg, float& p, float& p_proj) const
	{
		V3 l = square_to_cosine_hemisphere(u0, u1);
		const float NoV = dot(V, g.normal_s);
		if (trans ? (NoV > 0.0f) : (NoV < 0.0f)) l.z = -l.z;
		L = l.x * g.tangent + l.y * g.binormal + l.z * g.normal_s;
		gg = color * PI_F;
		p = fabsf(l.z) / PI_F;
		p_proj = 1.0f / PI_F;
	}
};

// contrib/cugar/bsdf/ggx_smith.h:54-201 (distribution) and :203-690 (bsdf)
struct GGXSmith
{
	float roughness, inv_roughness, int_ior, ext_ior;
	GGXSmith() {}
	GGXSmith(float r, bool transmission = false, float _int = 1.0f, float _ext = 1.0f) :
		roughness(r), inv_roughness(1.0f / r), int_ior(transmission ? _int : -1.0f), ext_ior(transmission ? _ext : -1.0f) {}
	bool transmissive() const { return int_ior > 0.0f; }
	float eta(float NoV) const { return NoV >= 0.0f ? ext_ior / int_ior : int_ior / ext_ior; }
	float inv_eta(float NoV) const { return NoV >= 0.0f ? int_ior / ext_ior : ext_ior / int_ior; }
	static float clamp_inf(float p) { return (!finite_f(p) || isnan_f(p)) ? 1.0e8f : maxf(p, 0.0f); }   // :228

	float smith_joint_approx(float NoV, float NoL) const    // :232-241
	{
		const float a = roughness;
		const float vv = NoL * (NoV * (1 - a) + a);
		const float vl = NoV * (NoL * (1 - a) + a);
		return 0.5f * 1.0f / (vv + vl);
	}
	float smith_g1v(float NoV, float NoL) const             // :255-264
	{
		const float a2 = roughness * roughness;
		const float G_V = NoV + sqrtf((NoV - NoV * a2) * NoV + a2);
		return 0.5f / (G_V * NoL);
	}
	float dwo_dh(float VoH, float LoH, float eta_, float inv_eta_) const   // :311-330
	{
		const float ci = fabsf(VoH);
		const float ct2 = 1.f - eta_ * eta_ * (1.f - ci * ci);
		if (ct2 < 0.0f) return 0.0f;
		const float sd = VoH + inv_eta_ * LoH;
		return 4 * inv_eta_ * inv_eta_ * fabsf(VoH * LoH) / (sd * sd);
	}
	// :414-467
	void f_and_p(const Frame& g, V3 V, V3 L, V3& f, float& p) const
	{
		const V3 N = g.normal_s;
		const float NoL = dot(N, L), NoV = dot(N, V);
		const float e = eta(NoV), ie = inv_eta(NoV);
		const V3 H = vndf_microfacet(V, L, N, ie);
		const float NoH = dot(N, H);
		const float sgn = transmissive() ? -1.0f : 1.0f;
		if (sgn * NoL * NoV <= 0.0f || NoH == 0.0f) { p = 0.0f; f = V3(0.0f); return; }
		const float D = hvd_ggx_eval(inv_roughness, fabsf(NoH), dot(g.tangent, H), dot(g.binormal, H));
		const float G = smith_joint_approx(fabsf(NoV), fabsf(NoL));
		const float G1 = smith_g1v(fabsf(NoV), fabsf(NoL));
		float tf = 1.0f;
		if (transmissive()) tf = dwo_dh(dot(V, H), dot(L, H), e, ie);
		f = V3(clamp_inf(G * D * tf));
		p = clamp_inf(G1 * D * tf);
	}
	// distribution().sample(u, V_local) : :109-131 ("V.z >= 0" sign rule, local frame)
	V3 sample_h_local(float u0, float u1, V3 Vl) const
	{
		const float sgn = Vl.z >= 0.0f ? 1.0f : -1.0f;
		V3 H = vndf_ggx_smith_sample(u0, u1, roughness, V3(Vl.x, Vl.y, Vl.z * sgn));
		H.z *= sgn;
		return H;
	}
	// sample L given H : :503-578
	void sample_given_h(const Frame& g, V3 H, V3 V, V3& L, V3& gg, float& p, float& p_proj) const
	{
		const V3 N = g.normal_s;
		const float NoV = dot(N, V);
		const float e = eta(NoV), ie = inv_eta(NoV);
		if (NoV == 0.0f) { p = 0.0f; p_proj = 0.0f; gg = V3(0.0f); return; }
		if (!transmissive())
			L = 2 * dot(V, H) * H - V;
		else
		{
			const float VoH = dot(V, H);
			const float ci = VoH;
			const float ct2 = 1.f - e * e * (1.f - ci * ci);
			if (ct2 < 0.0f) { L = 2 * dot(V, H) * H - V; p = 0.0f; p_proj = 0.0f; gg = V3(0.0f); return; }
			const float ct = (ci >= 0.0f ? 1.0f : -1.0f) * sqrtf(ct2);
			L = (e * ci - ct) * H - e * V;
		}
		const float NoL = dot(N, L), NoH = dot(N, H);
		const float sgn = transmissive() ? -1.0f : 1.0f;
		if (sgn * NoL * NoV <= 0.0f || NoH == 0.0f) { p = 0.0f; p_proj = 0.0f; gg = V3(0.0f); return; }
		const float D = hvd_ggx_eval(inv_roughness, fabsf(NoH), dot(g.tangent, H), dot(g.binormal, H));
		const float G = smith_joint_approx(fabsf(NoV), fabsf(NoL));
		const float G1 = smith_g1v(fabsf(NoV), fabsf(NoL));
		float tf = 1.0f;
		if (transmissive()) tf = dwo_dh(dot(V, H), dot(L, H), e, ie);
		p_proj = clamp_inf(G1 * D * tf);
		p = p_proj * fabsf(NoL);
		gg = V3(clamp_inf(G / G1));
	}
	// invert(geometry, V, L, random, z, p, p_proj) : :733-812 -- z.xy such that sample(z) returns L; p, p_proj are the RECIPROCAL densities
	bool invert(const Frame& g, V3 V, V3 L, float& z0, float& z1, float& p, float& p_proj) const
	{
		const V3 N = g.normal_s;
		const float NoV = dot(N, V), NoL = dot(N, L);
		const float e = eta(NoV), ie = inv_eta(NoV);
		const float sgn_N = NoV > 0.0f ? 1.0f : -1.0f;
		const V3 Vl(dot(V, g.tangent), dot(V, g.binormal), sgn_N * NoV);
		const V3 H = vndf_microfacet(V, L, N, ie);
		const float NoH = dot(N, H);
		const V3 Hl(dot(H, g.tangent), dot(H, g.binormal), sgn_N * NoH);
		const V2 uv = vndf_ggx_smith_invert(Hl, roughness, Vl);
		z0 = uv.x; z1 = uv.y;
		const float sgn = transmissive() ? -1.0f : 1.0f;
		if (sgn * NoL * NoV <= 0.0f || NoH == 0.0f) { p = 0.0f; p_proj = 0.0f; }
		else
		{
			const float D = hvd_ggx_eval(inv_roughness, fabsf(NoH), Hl.x, Hl.y);
			const float G1 = smith_g1v(fabsf(NoV), fabsf(NoL));
			float tf = 1.0f;
			if (transmissive()) tf = dwo_dh(dot(V, H), dot(L, H), e, ie);
			p_proj = clamp_inf(G1 * D * tf);
			p = p_proj * fabsf(NoL);
		}
		p_proj = clamp_inf(1.0f / p_proj);
		p = clamp_inf(1.0f / p);
		return true;
	}
	// full sample(u, geometry, V, ...) : :580-668 — used by the table generator (src/bsdf.cu:78-85) and the KAT
	void sample(float u0, float u1, const Frame& g, V3 V, V3& L, V3& gg, float& p, float& p_proj) const
	{
		const V3 N = g.normal_s;
		const float NoV = dot(N, V);
		const float e = eta(NoV), ie = inv_eta(NoV);
		const float sgn_V = NoV > 0.0f ? 1.0f : -1.0f;
		const V3 Vl(dot(V, g.tangent), dot(V, g.binormal), sgn_V * NoV);
		if (NoV == 0.0f) { p = 0.0f; p_proj = 0.0f; gg = V3(0.0f); return; }
		V3 H = vndf_ggx_smith_sample(u0, u1, roughness, Vl);
		H = H.x * g.tangent + H.y * g.binormal + H.z * g.normal_s * sgn_V;
		if (!transmissive())
			L = 2 * dot(V, H) * H - V;
		else
		{
			const float ci = dot(V, H);
			const float ct2 = 1.f - e * e * (1.f - ci * ci);
			if (ct2 < 0.0f) { L = 2 * dot(V, H) * H - V; p = 0.0f; p_proj = 0.0f; gg = V3(0.0f); return; }
			const float ct = -(ci >= 0.0f ? 1.0f : -1.0f) * sqrtf(ct2);
			L = (e * ci + ct) * H - e * V;
		}
		const float NoL = dot(N, L), NoH = dot(N, H);
		const float sgn = transmissive() ? -1.0f : 1.0f;
		if (sgn * NoL * NoV <= 0.0f || NoH == 0.0f) { p = 0.0f; p_proj = 0.0f; gg = V3(0.0f); return; }
		const float D = hvd_ggx_eval(inv_roughness, fabsf(NoH), dot(g.tangent, H), dot(g.binormal, H));
		const float G = smith_joint_approx(fabsf(NoV), fabsf(NoL));
		const float G1 = smith_g1v(fabsf(NoV), fabsf(NoL));
		float tf = 1.0f;
		if (transmissive()) tf = dwo_dh(dot(V, H), dot(L, H), e, ie);
		p_proj = clamp_inf(G1 * D * tf);
		p = p_proj * fabsf(NoL);
		gg = V3(clamp_inf(G / G1));
	}
};

// material record : src/mesh/MeshView.h:55-74 (208 bytes; TextureReference = {u32 texture; float2 scaling} 16 B)
struct TexRef { u32 texture; u32 _pad; float sx, sy; };
struct Material
{
	V4 diffuse, diffuse_trans, ambient, specular, emissive, reflectivity;
	float roughness, index_of_refraction, opacity;
	i32 flags;
	TexRef ambient_map, diffuse_map, diffuse_trans_map, specular_map, emissive_map, bump_map;
};
static_assert(sizeof(TexRef) == 16, "TextureReference layout");
static_assert(sizeof(Material) == 208, "MeshMaterial layout");

// component ids : src/bsdf.h:125-154
enum { kDiffR = 0, kDiffT = 1, kGlossR = 2, kGlossT = 3 };
enum { kAbsorption = 0u, kDiffuseReflection = 1u, kDiffuseTransmission = 2u, kGlossyReflection = 4u, kGlossyTransmission = 8u,
       kClearcoatReflection = 0x10u, kDiffuseMask = 3u, kGlossyMask = 0xCu };

// src/bsdf.h:123-1281
struct Bsdf
{
	Lambert  diffuse, diffuse_trans;
	GGXSmith glossy, glossy_trans;
	V3 fresnel, reflectivity;
	float ior, opacity, clearcoat_ior;
	const float* table;     // 32^4 glossy reflectance table (vs/fermat/glossy_reflectance.dat; regenerated, see tools/)
	bool particle_transport = false;      // TransportType (src/bsdf.h:81-87): kRadianceTransport for eye vertices, kParticleTransport for light vertices

	// ctor : src/bsdf.h:218-243 (mollification factor 1, bias 0, min_roughness 0 : src/bpt_utils.h:633-635)
	void setup(const Material& m, const float* _table)
	{
		diffuse.color = V3(m.diffuse.x, m.diffuse.y, m.diffuse.z) / PI_F; diffuse.trans = false;
		diffuse_trans.color = V3(m.diffuse_trans.x, m.diffuse_trans.y, m.diffuse_trans.z) / PI_F; diffuse_trans.trans = true;
		glossy = GGXSmith(maxf(m.roughness * 1.0f + 0.0f, 0.0f));
		glossy_trans = GGXSmith(m.roughness, true, m.index_of_refraction, 1.0f);
		fresnel = V3(m.specular.x, m.specular.y, m.specular.z) / PI_F;
		reflectivity = V3(m.reflectivity.x, m.reflectivity.y, m.reflectivity.z);
		ior = m.index_of_refraction;
		opacity = m.opacity;
		table = _table;
		const float R0 = minf(max_comp(reflectivity), 0.95f);
		clearcoat_ior = (1 + sqrtf(R0)) / (1 - sqrtf(R0));
	}

	// second ctor : src/bsdf.h:246-273, used by unpack_bsdf for stored light vertices (src/bpt_utils.h:244-260).  The reference leaves
	// m_reflectivity UNINITIALISED there and then reads it for the clearcoat IOR; this restatement defines it as zero (no clearcoat).
	void setup_unpacked(V3 diffuse_c, V3 specular_c, float roughness, V3 diffuse_trans_c, float opacity_, float ior_, const float* _table, bool particle)
	{
		diffuse.color = diffuse_c / PI_F; diffuse.trans = false;
		diffuse_trans.color = diffuse_trans_c / PI_F; diffuse_trans.trans = true;
		glossy = GGXSmith(roughness);
		glossy_trans = GGXSmith(roughness, true, ior_, 1.0f);
		fresnel = specular_c / PI_F;
		reflectivity = V3(0.0f);
		ior = ior_; opacity = opacity_; table = _table; particle_transport = particle;
		const float R0 = minf(max_comp(reflectivity), 0.95f);
		clearcoat_ior = (1 + sqrtf(R0)) / (1 - sqrtf(R0));
	}

	// src/bsdf.h:1254-1268
	float glossy_reflectance(float cos_theta) const
	{
		const u32 S = 32;
		const float eta = cos_theta > 0.0f ? 1.0f / ior : ior;
		const u32 ci = minu(S - 1u, f2u(fabsf(cos_theta) * float(S - 1)));
		const u32 bi = minu(S - 1u, f2u(max_comp(fresnel) * float(S - 1)));
		const u32 ei = minu(S - 1u, f2u((eta / 2.0f) * float(S - 1)));
		const u32 ri = minu(S - 1u, f2u(glossy.roughness * float(S - 1)));
		return table[ei * S * S * S + bi * S * S + ri * S + ci];
	}
	// src/bsdf.h:1202-1232
	bool clearcoat_transmission(const Frame& g, V3 w_i, V3& H, float& cos_theta_i, V3& Fc_1, V3& Tc_1) const
	{
		const float R0 = minf(max_comp(reflectivity), 0.95f);
		const float eta_c = 1.0f / clearcoat_ior;
		H = g.normal_s;
		cos_theta_i = dot(w_i, H);
		V3 w_t; float F;
		if (!refract(w_i, H, cos_theta_i, eta_c, &w_t, &F)) { Fc_1 = V3(1.0f); Tc_1 = V3(0.0f); return false; }
		Fc_1 = lerp3(reflectivity, V3(1.0f), maxf(F - R0, 0.0f) / (1 - R0));
		Tc_1 = V3(1.0f) - Fc_1;
		return true;
	}
	// src/bsdf.h:1237-1251
	float compression_factor(const Frame& g, V3 w_i, V3 w_o) const
	{
		if (!particle_transport && ior != 0.0f)
		{
			const float NoV = dot(w_i, g.normal_s), NoL = dot(w_o, g.normal_s);
			if (NoV * NoL < 0.0f) return sqr(NoV > 0.0f ? ior : 1.0f / ior);
		}
		return 1.0f;
	}
	// src/bsdf.h:530-586
	void sampling_weights(const Frame& g, V3 V, float* w) const
	{
		const float NoV_signed = dot(g.normal_s, V);
		V3 r, t;
		if (ior == 0) { r = V3(0.0f); t = V3(1.0f); }
		else { r = V3(glossy_reflectance(NoV_signed)); t = V3(1.0f - max_comp(r)); }
		w[kGlossR] = max_comp(r);
		w[kGlossT] = (1 - opacity) * max_comp(t);
		w[kDiffR]  = opacity * max_comp(t * diffuse.color) * PI_F;
		w[kDiffT]  = opacity * max_comp(t * diffuse_trans.color) * PI_F;
	}
	// src/bsdf.h:591-627 with components == kAllComponents, RR == true
	static void normalize_sampling_weights(float* w, float coat_T)
	{
		w[kDiffR] *= coat_T; w[kDiffT] *= coat_T; w[kGlossR] *= coat_T; w[kGlossT] *= coat_T;
	}
	// src/bsdf.h:591-627 with components == kAllComponents and the RR switch (RR == false renormalises everything, the coat probabilities included)
	static void normalize_sampling_weights(float* w, float& coat_R, float& coat_T, bool RR)
	{
		w[kDiffR] *= coat_T; w[kDiffT] *= coat_T; w[kGlossR] *= coat_T; w[kGlossT] *= coat_T;
		if (RR == false)
		{
			const float inv_sum = 1.0f / (w[kDiffR] + w[kDiffT] + w[kGlossR] + w[kGlossT] + coat_R);
			w[kDiffR] *= inv_sum; w[kDiffT] *= inv_sum; w[kGlossR] *= inv_sum; w[kGlossT] *= inv_sum;
			coat_R *= inv_sum; coat_T *= inv_sum;
		}
	}
	// src/bsdf.h:632-664
	void fresnel_weights(float VoH, float eta, V3& r, V3& t) const
	{
		if (eta == 0.0f) { r = V3(0.0f); t = V3(1.0f); }
		else { r = fresnel_schlick(VoH, eta, fresnel); t = V3(1.0f - max_comp(r)); }
	}
	// src/bsdf.h:666-695
	void fresnel_weights(const Frame& g, V3 V, V3 L, V3& r, V3& t) const
	{
		float eta = 0.0f, inv_eta = 0.0f, VoH = 0.0f;
		if (ior != 0.0f)
		{
			const V3 N = g.normal_s;
			eta     = dot(N, V) > 0.0f ? 1.0f / ior : ior;
			inv_eta = dot(N, V) > 0.0f ? ior : 1.0f / ior;
			const V3 H = microfacet(V, L, N, inv_eta);
			VoH = dot(V, H);
		}
		fresnel_weights(VoH, eta, r, t);
	}
	// src/bsdf.h:722-743
	void inner_component_weights(const Frame& g, V3 V, V3 L, V3* w) const
	{
		V3 r, t;
		fresnel_weights(g, V, L, r, t);
		const float dw = (1.0f - glossy_reflectance(dot(g.normal_s, V))) * (1.0f - glossy_reflectance(dot(g.normal_s, L)));
		w[kGlossR] = r;
		w[kGlossT] = t * (1 - opacity);
		w[kDiffR]  = t * opacity * dw;
		w[kDiffT]  = t * opacity * dw;
	}
	// src/bsdf.h:748-792
	void component_weights(const Frame& g, V3 w_i, V3 w_o, V3& Fc_1, V3& Tc_1, V3* w) const
	{
		V3 H_c; float ci;
		if (!clearcoat_transmission(g, w_i, H_c, ci, Fc_1, Tc_1))
		{
			w[0] = w[1] = w[2] = w[3] = V3(0.0f);
			return;
		}
		const V3 Tc_2 = V3(1.0f) - V3(0.0f);
		inner_component_weights(g, w_i, w_o, w);
		for (int i = 0; i < 4; ++i) w[i] = w[i] * (Tc_1 * Tc_2);
	}
	// per-lobe f_and_p : src/bsdf.h:366-412 (measure = projected solid angle, RR = true)
	void f_and_p(const Frame& g, V3 w_i, V3 w_o, V3* f, float* p) const
	{
		V3 Fc_1, Tc_1, w[4];
		component_weights(g, w_i, w_o, Fc_1, Tc_1, w);
		const float coat_R = average(Fc_1);
		const float coat_T = 1.0f - coat_R;
		V3 f_d, f_g, f_dt, f_gt; float p_d, p_g, p_dt, p_gt;
		diffuse.f_and_p(g, w_i, w_o, f_d, p_d);
		diffuse_trans.f_and_p(g, w_i, w_o, f_dt, p_dt);
		glossy.f_and_p(g, w_i, w_o, f_g, p_g);
		glossy_trans.f_and_p(g, w_i, w_o, f_gt, p_gt);
		float wp[4];
		sampling_weights(g, w_i, wp);
		normalize_sampling_weights(wp, coat_T);
		p[kDiffR] = p_d * wp[kDiffR];
		p[kDiffT] = p_dt * wp[kDiffT];
		p[kGlossR] = p_g * wp[kGlossR];
		p[kGlossT] = p_gt * wp[kGlossT];
		const float factor = compression_factor(g, w_i, w_o);
		f[kDiffR]  = f_d  * w[kDiffR]  * factor;
		f[kDiffT]  = f_dt * w[kDiffT]  * factor;
		f[kGlossR] = f_g  * w[kGlossR] * factor;
		f[kGlossT] = f_gt * w[kGlossT] * factor;
	}
	// summed f_and_p : src/bsdf.h:417-464 (projected solid angle, all components)
	void f_and_p_sum(const Frame& g, V3 w_i, V3 w_o, V3& f, float& p, bool RR) const
	{
		V3 Fc_1, Tc_1, w[4];
		component_weights(g, w_i, w_o, Fc_1, Tc_1, w);
		float coat_R = average(Fc_1);
		float coat_T = 1.0f - coat_R;
		V3 f_d, f_g, f_dt, f_gt; float p_d, p_g, p_dt, p_gt;
		diffuse.f_and_p(g, w_i, w_o, f_d, p_d);
		diffuse_trans.f_and_p(g, w_i, w_o, f_dt, p_dt);
		glossy.f_and_p(g, w_i, w_o, f_g, p_g);
		glossy_trans.f_and_p(g, w_i, w_o, f_gt, p_gt);
		float wp[4];
		sampling_weights(g, w_i, wp);
		normalize_sampling_weights(wp, coat_R, coat_T, RR);
		p = p_d * wp[kDiffR] + p_dt * wp[kDiffT] + p_g * wp[kGlossR] + p_gt * wp[kGlossT];
		const float factor = compression_factor(g, w_i, w_o);
		f = f_d * w[kDiffR] * factor + f_dt * w[kDiffT] * factor + f_g * w[kGlossR] * factor + f_gt * w[kGlossT] * factor;
	}
	// f : src/bsdf.h:296-318 (all components)
	V3 f_sum(const Frame& g, V3 w_i, V3 w_o) const
	{
		V3 Fc_1, Tc_1, w[4];
		component_weights(g, w_i, w_o, Fc_1, Tc_1, w);
		const float factor = compression_factor(g, w_i, w_o);
		V3 f_d, f_g, f_dt, f_gt; float pd;
		diffuse.f_and_p(g, w_i, w_o, f_d, pd);
		diffuse_trans.f_and_p(g, w_i, w_o, f_dt, pd);
		glossy.f_and_p(g, w_i, w_o, f_g, pd);
		glossy_trans.f_and_p(g, w_i, w_o, f_gt, pd);
		return f_d * w[kDiffR] * factor + f_dt * w[kDiffT] * factor + f_g * w[kGlossR] * factor + f_gt * w[kGlossT] * factor;
	}
	// p : src/bsdf.h:466-528 (projected solid angle, all components)
	float p_sum(const Frame& g, V3 w_i, V3 w_o, bool RR) const
	{
		V3 H_c, Fc_1, Tc_1; float ci;
		if (!clearcoat_transmission(g, w_i, H_c, ci, Fc_1, Tc_1)) return 0.0f;
		float coat_R = average(Fc_1);
		float coat_T = 1.0f - coat_R;
		float wp[4];
		sampling_weights(g, w_i, wp);
		normalize_sampling_weights(wp, coat_R, coat_T, RR);
		V3 fd; float p_d, p_g, p_dt, p_gt;
		diffuse.f_and_p(g, w_i, w_o, fd, p_d);
		diffuse_trans.f_and_p(g, w_i, w_o, fd, p_dt);
		glossy.f_and_p(g, w_i, w_o, fd, p_g);
		glossy_trans.f_and_p(g, w_i, w_o, fd, p_gt);
		return p_d * wp[kDiffR] + p_dt * wp[kDiffT] + p_g * wp[kGlossR] + p_gt * wp[kGlossT];
	}

	// src/bsdf.h:921-1199 with RR = true, evaluate_full_bsdf = false, components = kAllComponents
	bool sample(const Frame& g, const float z[3], V3 in, u32& out_comp, V3& out, float& out_p, float& out_p_proj, V3& out_g) const
	{ return sample_ex(g, z, in, out_comp, out, out_p, out_p_proj, out_g, true, false); }
	// src/bsdf.h:921-1199 with components = kAllComponents
	bool sample_ex(const Frame& g, const float z[3], V3 in, u32& out_comp, V3& out, float& out_p, float& out_p_proj, V3& out_g, bool RR, bool evaluate_full_bsdf) const
	{
		V3 gg(0.0f); float p = 0.0f, p_proj = 0.0f, p_comp = 0.0f;
		V3 w_i = in, w_o(0.0f);   // NB: reference leaves w_o uninitialised; it is only observable when p == 0 (path dies)
		V3 H_c, Fc_1, Tc_1; float cos_theta_i;
		if (!clearcoat_transmission(g, in, H_c, cos_theta_i, Fc_1, Tc_1))
		{
			out = V3(0.0f); out_p = 0.0f; out_p_proj = 0.0f; out_g = V3(0.0f); out_comp = kAbsorption;
			return false;
		}
		float coat_R = average(Fc_1);
		float coat_T = 1.0f - coat_R;
		float wp[4];
		sampling_weights(g, in, wp);
		// "efficient sampler" : :996-1035
		const V3 V_local = g.to_local(w_i);
		const V3 H_local = glossy.sample_h_local(z[0], z[1], V_local);
		const V3 H = g.from_local(H_local);
		V3 r, t;
		const float eta = V_local.z > 0.0f ? 1.0f / ior : ior;
		fresnel_weights(dot(V_local, H_local), eta, r, t);
		wp[kGlossR] = (wp[kGlossR] + max_comp(r)) * 0.5f;
		wp[kGlossT] = (wp[kGlossT] + (1 - opacity) * max_comp(t)) * 0.5f;
		wp[kDiffR]  = (wp[kDiffR] + opacity * max_comp(t * diffuse.color) * PI_F) * 0.5f;
		wp[kDiffT]  = (wp[kDiffT] + opacity * max_comp(t * diffuse_trans.color) * PI_F) * 0.5f;
		normalize_sampling_weights(wp, coat_R, coat_T, RR);

		// lobe selection order: diffR, glossR, diffT, glossT, coat, absorb : :1041-1125
		if (z[2] < wp[kDiffR])
		{
			const float zn = z[2] / wp[kDiffR]; (void)zn;
			p_comp = wp[kDiffR];
			diffuse.sample(z[0], z[1], g, w_i, w_o, gg, p, p_proj);
			out_comp = kDiffuseReflection;
		}
		else if (z[2] < wp[kDiffR] + wp[kGlossR])
		{
			p_comp = wp[kGlossR];
			glossy.sample_given_h(g, H, w_i, w_o, gg, p, p_proj);
			out_comp = kGlossyReflection;
		}
		else if (z[2] < wp[kDiffR] + wp[kGlossR] + wp[kDiffT])
		{
			p_comp = wp[kDiffT];
			diffuse_trans.sample(z[0], z[1], g, w_i, w_o, gg, p, p_proj);
			out_comp = kDiffuseTransmission;
		}
		else if (z[2] < wp[kDiffR] + wp[kGlossR] + wp[kDiffT] + wp[kGlossT])
		{
			p_comp = wp[kGlossT];
			glossy_trans.sample_given_h(g, H, w_i, w_o, gg, p, p_proj);
			out_comp = kGlossyTransmission;
		}
		else if (z[2] < wp[kDiffR] + wp[kGlossR] + wp[kDiffT] + wp[kGlossT] + coat_R)
		{
			p_comp = coat_R;
			out = 2 * cos_theta_i * H_c - in;
			gg = Fc_1 / p_comp;
			p_proj = finf();
			p = finf();
			out_comp = kClearcoatReflection;
		}
		else
			out_comp = kAbsorption;

		if (out_comp != kAbsorption && out_comp != kClearcoatReflection)
		{
			const V3 Tc_2 = V3(1.0f) - V3(0.0f);
			gg = gg * (Tc_1 * Tc_2);
			out = w_o;
		}
		if (out_comp != kAbsorption)
		{
			if (out_comp != kClearcoatReflection && evaluate_full_bsdf)
			{
				// :1147-1162 — NB the lobe probabilities are multiplied by coat_transmission_prob a second time, as written
				V3 fd; float p_d, p_dt, p_g, p_gt;
				diffuse.f_and_p(g, in, out, fd, p_d);
				diffuse_trans.f_and_p(g, in, out, fd, p_dt);
				glossy.f_and_p(g, in, out, fd, p_g);
				glossy_trans.f_and_p(g, in, out, fd, p_gt);
				p_proj = p_d * wp[kDiffR] * coat_T + p_dt * wp[kDiffT] * coat_T + p_g * wp[kGlossR] * coat_T + p_gt * wp[kGlossT] * coat_T;
				p = p_proj * fabsf(dot(out, g.normal_s));
				gg = f_sum(g, in, out) / p_proj;
			}
			else if (out_comp != kClearcoatReflection)
			{
				V3 w[4];
				inner_component_weights(g, in, out, w);
				gg = gg * ((out_comp & kGlossyReflection) ? w[kGlossR] :
				           (out_comp & kGlossyTransmission) ? w[kGlossT] :
				           (out_comp & kDiffuseReflection) ? w[kDiffR] : w[kDiffT]);
				gg = gg / p_comp;
				p = p * p_comp;
				p_proj = p_proj * p_comp;
			}
			const float factor = compression_factor(g, in, out);
			out_p = p; out_p_proj = p_proj; out_g = gg * factor;
			return true;
		}
		out = V3(0.0f); out_p = 0.0f; out_p_proj = 0.0f; out_g = V3(0.0f);
		return false;
	}
};

// contrib/cugar/bsdf/lambert_edf.h:56-62, src/edf.h:49-65
struct Edf
{
	V3 color;
	V3 f(const Frame& g, V3 out) const { return dot(g.normal_s, out) > 0.0f ? color : V3(0.0f); }
};

// one cell of the 32^4 glossy reflectance table : src/bsdf.cu:36-102 (all S^4 cells)
inline float glossy_reflectance_cell(u32 cell_index)
{
	const u32 S = 32;
	Frame g;
	g.tangent = V3(1, 0, 0); g.binormal = V3(0, 1, 0); g.normal_s = g.normal_g = V3(0, 0, 1);
	const u32 eta_i = cell_index / (S * S * S);
	const u32 base_i = (cell_index / (S * S)) % S;
	const u32 rough_i = (cell_index / S) % S;
	const u32 theta_i = cell_index % S;
	const float eta = 2.0f * float(float(eta_i) + 0.5f) / float(S);
	const float base = float(base_i) / float(S - 1);
	const float rough = sqr(float(rough_i) / float(S - 1));
	const float ct = float(theta_i) / float(S - 1);
	GGXSmith bsdf(rough);
	const V3 V(sqrtf(1.0f - ct * ct), 0.0f, ct);
	float sum = 0.0f;
	const u32 M = 4 * 32, N = M * M;
	for (u32 s = 0; s < N; ++s)
	{
		float ux, uy;
		correlated_multijitter(s, M, M, cell_index, ux, uy);
		V3 L(0.0f), gg(0.0f); float p, pp;
		bsdf.sample(ux, uy, g, V, L, gg, p, pp);
		const V3 H = normalize(V + L);
		const float VoH = dot(V, H);
		const float F = max_comp(fresnel_schlick(VoH, eta, V3(base)));
		sum += F * gg.x;
	}
	return sum / float(N);
}

} // namespace orc
