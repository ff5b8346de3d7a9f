// cugar_kat_driver.cpp -- OUR driver around the reference's header-only math layer (contrib/cugar), used by tests/golden/make_cugar_kat.py in the BUILD container
// only: it calls the reference's own randfloat / hash / permute / correlated_multijitter / LFSRRandomStream / pack_normal / orthogonal /
// square_to_cosine_hemisphere / fresnel_* / refract / LambertBsdf / LambertTransBsdf / GGXSmithBsdf on seeded inputs and prints inputs and outputs as text.
//
// HONEST LABEL (VERDICT r4 task 8): the cugar headers include CUDA's <vector_types.h>, <vector_functions.h>, <cuda_fp16.h>, <cuda_runtime.h>, which this image
// lacks; make_cugar_kat.py writes four one-line stand-in headers into a temporary directory (each includes the HIP header that defines the same float3 / half
// types) and a temporary copy of cugar/linalg/vector{,_inl}.h with ONE parameter renamed (it shadows its template parameter, which only MSVC accepts).  By this
// task's rules a reference build that needs stand-in headers does not count as "the reference built here", so these vectors do NOT pin parity; they are 300+ known
// answers from the reference's own arithmetic instead of the two the survey recorded, nothing more.  Only tests/golden/cugar_kat.npz travels.
#define _finite(x) std::isfinite(x)
#define _isnan(x) std::isnan(x)
#include <cmath>
#include <cstdio>
#include <cstdint>
using std::isfinite; using std::isnan;
#include <cugar/basic/numbers.h>
#include <cugar/linalg/vector.h>
#include <cugar/spherical/mappings.h>
#include <cugar/sampling/lfsr.h>
#include <cugar/sampling/multijitter.h>
#include <cugar/bsdf/differential_geometry.h>
#include <cugar/bsdf/refraction.h>
#include <cugar/bsdf/lambert.h>
#include <cugar/bsdf/lambert_trans.h>
#include <cugar/bsdf/ggx_smith.h>

using cugar::Vector3f;
static uint32_t lcg = 12345u;
static float U() { lcg = lcg * 1664525u + 1013904223u; return float(lcg >> 8) * (1.0f / 16777216.0f); }
static Vector3f unit(float zmin = -1.0f)
{
	for (;;)
	{
		Vector3f v(2 * U() - 1, 2 * U() - 1, 2 * U() - 1);
		const float l = sqrtf(v.x * v.x + v.y * v.y + v.z * v.z);
		if (l > 0.2f && l <= 1.0f) { v = v / l; if (v.z >= zmin) return v; }
	}
}
static void P(const char* tag, int n, const double* v) { printf("%s", tag); for (int i = 0; i < n; ++i) printf(" %.17g", v[i]); printf("\n"); }
#define ROW(tag, ...) do { const double _v[] = { __VA_ARGS__ }; P(tag, int(sizeof(_v) / sizeof(double)), _v); } while (0)

int main()
{
	cugar::DifferentialGeometry g;
	g.tangent = Vector3f(1, 0, 0); g.binormal = Vector3f(0, 1, 0); g.normal_s = Vector3f(0, 0, 1); g.normal_g = Vector3f(0, 0, 1);
	// integer paths (exact)
	for (uint32_t i = 0; i < 24; ++i) { const uint32_t a = i * 2654435761u + 17u; ROW("hash", double(a), double(cugar::hash(a))); }
	for (uint32_t i = 0; i < 24; ++i) { const uint32_t d = i * 7u, p = i * 977u + 1u; ROW("randfloat", double(d), double(p), double(cugar::randfloat(d, p))); }
	for (uint32_t i = 0; i < 24; ++i) { const uint32_t l = 3u + i * 37u, k = (i * 131u) % l, p = i * 7919u + 5u; ROW("permute", double(k), double(l), double(p), double(cugar::permute(k, l, p))); }
	for (uint32_t i = 0; i < 24; ++i)
	{
		const uint32_t m = 4 + (i % 5), n = 3 + (i % 7), s = (i * 13u) % (m * n), p = i * 104729u + 3u;
		const float2 r = cugar::correlated_multijitter(s, m, n, p);
		ROW("cmj", double(s), double(m), double(n), double(p), double(r.x), double(r.y));
	}
	{
		cugar::LFSRGeneratorMatrix mat(32, cugar::LFSRGeneratorMatrix::GOOD_PROJECTIONS);
		for (uint32_t inst = 0; inst < 2; ++inst)
		{
			cugar::LFSRRandomStream st(&mat, 1u, cugar::hash(1351u + inst));
			for (uint32_t i = 0; i < 16; ++i) ROW("lfsr", double(inst), double(i), double(st.next()));
		}
	}
	for (uint32_t i = 0; i < 24; ++i) { const Vector3f n = unit(); ROW("pack_normal", double(n.x), double(n.y), double(n.z), double(cugar::pack_normal(n))); }
	for (uint32_t i = 0; i < 24; ++i) { const uint32_t p = cugar::hash(i + 99u) & 0x3FFFFFFFu; const Vector3f n = cugar::unpack_normal(p); ROW("unpack_normal", double(p), double(n.x), double(n.y), double(n.z)); }
	for (uint32_t i = 0; i < 24; ++i) { const Vector3f v = unit(); const Vector3f o = cugar::orthogonal(v); ROW("orthogonal", double(v.x), double(v.y), double(v.z), double(o.x), double(o.y), double(o.z)); }
	// float paths
	for (uint32_t i = 0; i < 32; ++i) { const float a = U(), b = U(); const Vector3f d = cugar::square_to_cosine_hemisphere(cugar::Vector2f(a, b)); ROW("cos_hemi", double(a), double(b), double(d.x), double(d.y), double(d.z)); }
	for (uint32_t i = 0; i < 24; ++i)
	{
		const float c = 2 * U() - 1, eta = (i & 1) ? 1.0f / (1.1f + U()) : (1.1f + U()); const Vector3f base(U(), U(), U());
		const Vector3f f = cugar::fresnel_schlick(c, eta, base);
		ROW("fresnel_schlick", double(c), double(eta), double(base.x), double(base.y), double(base.z), double(f.x), double(f.y), double(f.z));
	}
	for (uint32_t i = 0; i < 24; ++i) { const float ci = U(), ct = U(), eta = i % 6 == 0 ? 1.0f : 0.5f + 1.5f * U(); ROW("fresnel_dielectric", double(ci), double(ct), double(eta), double(cugar::fresnel_dielectric(ci, ct, eta))); }
	for (uint32_t i = 0; i < 32; ++i)
	{
		const Vector3f w = unit(), N(0, 0, 1); const float eta = i % 8 == 0 ? 1.0f : ((i & 1) ? 1.0f / (1.05f + U()) : (1.05f + U()));
		Vector3f out(0.0f); float F = 0.0f;
		const bool ok = cugar::refract(w, N, cugar::dot(w, N), eta, &out, &F);
		ROW("refract", double(w.x), double(w.y), double(w.z), double(eta), double(ok ? 1 : 0), double(out.x), double(out.y), double(out.z), double(F));
	}
	for (uint32_t i = 0; i < 24; ++i)
	{
		const Vector3f V = unit(), L = unit(), color(U(), U(), U());
		cugar::LambertBsdf b(color); cugar::LambertTransBsdf t(color);
		Vector3f f, ft; float p, pt;
		b.f_and_p(g, V, L, f, p, cugar::kProjectedSolidAngle); t.f_and_p(g, V, L, ft, pt, cugar::kProjectedSolidAngle);
		ROW("lambert_f_and_p", double(V.x), double(V.y), double(V.z), double(L.x), double(L.y), double(L.z), double(color.x), double(color.y), double(color.z),
		    double(f.x), double(f.y), double(f.z), double(p), double(ft.x), double(ft.y), double(ft.z), double(pt));
	}
	for (uint32_t i = 0; i < 24; ++i)
	{
		const Vector3f V = unit(), color(U(), U(), U()), u(U(), U(), U());
		cugar::LambertBsdf b(color); cugar::LambertTransBsdf t(color);
		Vector3f L, gg, Lt, ggt; float p, pp, pt, ppt;
		b.sample(u, g, V, L, gg, p, pp); t.sample(u, g, V, Lt, ggt, pt, ppt);
		ROW("lambert_sample", double(V.x), double(V.y), double(V.z), double(color.x), double(color.y), double(color.z), double(u.x), double(u.y),
		    double(L.x), double(L.y), double(L.z), double(gg.x), double(gg.y), double(gg.z), double(p), double(pp),
		    double(Lt.x), double(Lt.y), double(Lt.z), double(ggt.x), double(ggt.y), double(ggt.z), double(pt), double(ppt));
	}
	// GGX-Smith: reflection and rough-dielectric configurations (roughness, transmission, interior / exterior index)
	const float cfg[6][4] = { { 0.2f, 0, 1, 1 }, { 0.05f, 0, 1, 1 }, { 0.7f, 0, 1, 1 }, { 0.2f, 1, 1.5f, 1.0f }, { 0.4f, 1, 1.33f, 1.0f }, { 0.1f, 1, 1.0f, 1.5f } };
	for (int c = 0; c < 6; ++c)
	{
		for (uint32_t i = 0; i < 12; ++i)
		{
			const Vector3f V = unit(c < 3 ? 0.05f : -1.0f), u(U(), U(), U());
			Vector3f L(0.0f), gg(0.0f); float p = 0, pp = 0;
			if (cfg[c][1] != 0.0f) { cugar::GGXSmithBsdf b(cfg[c][0], true, cfg[c][2], cfg[c][3]); b.sample(u, g, V, L, gg, p, pp); }
			else                   { cugar::GGXSmithBsdf b(cfg[c][0]); b.sample(u, g, V, L, gg, p, pp); }
			ROW("ggx_sample", double(cfg[c][0]), double(cfg[c][1]), double(cfg[c][2]), double(cfg[c][3]), double(V.x), double(V.y), double(V.z), double(u.x), double(u.y), double(u.z),
			    double(L.x), double(L.y), double(L.z), double(gg.x), double(p), double(pp));
			const Vector3f L2 = unit(c < 3 ? 0.05f : -1.0f);
			Vector3f f(0.0f); float q = 0;
			if (cfg[c][1] != 0.0f) { cugar::GGXSmithBsdf b(cfg[c][0], true, cfg[c][2], cfg[c][3]); b.f_and_p(g, V, L2, f, q, cugar::kProjectedSolidAngle); }
			else                   { cugar::GGXSmithBsdf b(cfg[c][0]); b.f_and_p(g, V, L2, f, q, cugar::kProjectedSolidAngle); }
			ROW("ggx_f_and_p", double(cfg[c][0]), double(cfg[c][1]), double(cfg[c][2]), double(cfg[c][3]), double(V.x), double(V.y), double(V.z), double(L2.x), double(L2.y), double(L2.z),
			    double(f.x), double(q));
		}
	}
	return 0;
}
