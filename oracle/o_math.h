// ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is shipped or measured as product;
// only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
//
// o_math.h : CPU restatement of the cugar math layer the -pt hot path uses.
// Citations are relative to /root/reference (NVlabs/fermat).
//
// PARITY STATUS: the reference is not buildable here (MSVC + CUDA 10 + OptiX 6; even the cugar headers
// need CUDA's <vector_types.h>/<cuda_fp16.h>), so this restatement is pinned only by
//   (a) the two known-answer values the survey recorded from the cugar headers,
//   (b) the property tests of contrib/cugar/bsdf/bsdf_test.h:49-149,
//   (c) integer paths (hash / randfloat / LFSR / MSVC LCG / Morton) which are exact by construction.
// Image-level parity is UNPINNED by the reference (it ships no golden images).
//
// Floating-point contract shared with the HIP product ("fpt detmath v1"):
//   * fp32 everywhere, no FMA contraction (-ffp-contract=off), IEEE div/sqrt, denormals kept;
//   * dot(a,b) = (a.x*b.x + a.y*b.y) + a.z*b.z;
//   * sin/cos/atan2/pow are the polynomial kernels below (NOT libm / CUDA intrinsics, which differ by ulps
//     between vendors); CUDA's rsqrtf() is restated as 1/sqrtf();
//   * float->uint32 conversion saturates like CUDA's cvt.rzi.u32.f32 (NaN -> 0).
#pragma once
#include <cstdint>
#include <cmath>
#include <cstring>

namespace orc {

typedef uint32_t u32;
typedef int32_t  i32;
typedef uint64_t u64;

static const float PI_F     = 3.14159265358979323846f;   // contrib/cugar/basic/numbers.h:102-104 (M_PIf)
static const float TWO_PI_F = 6.28318530717958647693f;

inline u32   f2bits(float f) { u32 u; std::memcpy(&u, &f, 4); return u; }
inline float bits2f(u32 u)   { float f; std::memcpy(&f, &u, 4); return f; }

// contrib/cugar/basic/numbers.h:121
inline float finf() { return bits2f(0x7f800000u); }
// contrib/cugar/basic/numbers.h:76-83
inline bool  finite_f(float x) { return (f2bits(x) & 0x7f800000u) != 0x7f800000u; }
inline bool  isnan_f(float x)  { return x != x; }

// contrib/cugar/basic/numbers.h:536,540 — comparison-based min/max (NaN in `b` wins / NaN in `a` loses)
inline float minf(float a, float b) { return a < b ? a : b; }
inline float maxf(float a, float b) { return a > b ? a : b; }
inline u32   minu(u32 a, u32 b) { return a < b ? a : b; }
// IEEE fmaxf/fminf as used verbatim at src/pathtracer_core.h:898,1016,1126 (NaN operand is dropped)
inline float fmax_ieee(float a, float b) { if (a != a) return b; if (b != b) return a; return a > b ? a : b; }
inline float fmin_ieee(float a, float b) { if (a != a) return b; if (b != b) return a; return a < b ? a : b; }
// contrib/cugar/basic/numbers.h:636-644 (device path ::saturate: NaN -> 0)
inline float saturate(float x) { if (x != x) return 0.0f; return x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x); }
inline float sqr(float x) { return x * x; }

// CUDA float->uint32 semantics (round toward zero, saturating, NaN->0); used wherever the reference
// writes uint32(float) in device code, e.g. src/bsdf.h:1261-1264, src/texture_view.h:186-187, src/lights.h:321
inline u32 f2u(float x)
{
	if (!(x > 0.0f)) return 0u;               // negatives, zero, NaN
	if (x >= 4294967296.0f) return 0xFFFFFFFFu;
	return (u32)x;
}
// CUDA float->int32 semantics (saturating, NaN->0)
inline i32 f2i(float x)
{
	if (x != x) return 0;
	if (x >= 2147483648.0f) return 0x7FFFFFFF;
	if (x <= -2147483648.0f) return (i32)0x80000000;
	return (i32)x;
}

// contrib/cugar/basic/numbers.h:600-603
inline u32 quantize(float x, u32 n)
{
	i32 v = f2i(x * float(n));
	i32 hi = i32(n - 1);
	if (v > hi) v = hi;
	if (v < 0) v = 0;
	return (u32)v;
}

// fmodf(x,1) for any finite x >= 0 : exact (x - trunc(x) is representable)
inline float fmod1_pos(float x) { return x - truncf(x); }
// contrib/cugar/basic/numbers.h:606 — cugar::mod(x,1)
inline float mod1(float x) { return x > 0.0f ? fmod1_pos(x) : 1.0f - fmod1_pos(-x); }

// contrib/cugar/basic/numbers.h:648-657
inline u32 hash(u32 a)
{
	a = (a + 0x7ed55d16) + (a << 12);
	a = (a ^ 0xc761c23c) ^ (a >> 19);
	a = (a + 0x165667b1) + (a << 5);
	a = (a + 0xd3a2646c) ^ (a << 9);
	a = (a + 0xfd7046c5) + (a << 3);
	a = (a ^ 0xb55a4f09) ^ (a >> 16);
	return a;
}

// contrib/cugar/basic/numbers.h:752-763
inline float randfloat(u32 i, u32 p)
{
	i ^= p;
	i ^= i >> 17;
	i ^= i >> 10; i *= 0xb36534e5u;
	i ^= i >> 12;
	i ^= i >> 21; i *= 0x93fc4795u;
	i ^= 0xdf6e307fu;
	i ^= i >> 17; i *= 1u | p >> 18;
	return float(i) * (1.0f / 4294967808.0f);
}

// contrib/cugar/basic/numbers.h:836-860 (Kensler's permute)
inline u32 permute(u32 i, u32 l, u32 p)
{
	u32 w = l - 1;
	w |= w >> 1; w |= w >> 2; w |= w >> 4; w |= w >> 8; w |= w >> 16;
	do {
		i ^= p;             i *= 0xe170893du;
		i ^= p >> 16;
		i ^= (i & w) >> 4;
		i ^= p >> 8;        i *= 0x0929eb3fu;
		i ^= p >> 23;
		i ^= (i & w) >> 1;  i *= 1u | p >> 27;
		                    i *= 0x6935fa69u;
		i ^= (i & w) >> 11; i *= 0x74dcb303u;
		i ^= (i & w) >> 2;  i *= 0x9e501cc3u;
		i ^= (i & w) >> 2;  i *= 0xc860a3dfu;
		i &= w;
		i ^= i >> 5;
	} while (i >= l);
	return (i + p) % l;
}

// contrib/cugar/sampling/multijitter.h:63-90 (unordered branch only)
inline void correlated_multijitter(u32 s, u32 m, u32 n, u32 p, float& ox, float& oy)
{
	s = permute(s, m * n, p * 0x51633e2du);
	const u32 x = s % m, y = s / m;
	const u32 sx = permute(x, m, p * 0x68bc21ebu);
	const u32 sy = permute(y, n, p * 0x02e5be93u);
	const float jx = randfloat(s, p * 0x967a889bu);
	const float jy = randfloat(s, p * 0x368cc8b7u);
	ox = (float(sx) + (float(sy) + jx) / float(n)) / float(m);
	oy = (float(y) + (float(x) + jy) / float(m)) / float(n);
}

// ---------------------------------------------------------------------------------------------
// deterministic transcendental kernels ("fpt detmath v1").  Plain +,-,*,/ only, fixed order.
// They replace CUDA's sinf/cosf (contrib/cugar/bsdf/ggx_common.h:283-284,
// contrib/cugar/spherical/mappings_inline.h:83-85), atan2f (mappings_inline.h:181) and powf
// (src/renderer.cu:99-102), whose bit patterns no CPU libm reproduces.
// ---------------------------------------------------------------------------------------------
inline void det_sincos(float x, float* s, float* c)
{
	// quadrant reduction, 3-term Cody-Waite split of pi/2; valid for |x| < ~1e4, callers stay within [-pi, 2pi]
	const float k  = floorf(x * 0.636619772367581343f + 0.5f);
	const float C1 = 1.5703125f;
	const float C2 = 4.837512969970703125e-4f;
	const float C3 = 7.54978995489188216e-8f;
	float r = x - k * C1;
	r = r - k * C2;
	r = r - k * C3;
	const float z = r * r;
	// Cephes single-precision minimax polynomials on [-pi/4, pi/4]
	float ps = -1.9515295891e-4f * z + 8.3321608736e-3f;
	ps = ps * z - 1.6666654611e-1f;
	const float sr = r + r * z * ps;
	float pc = 2.443315711809948e-5f * z - 1.388731625493765e-3f;
	pc = pc * z + 4.166664568298827e-2f;
	const float cr = (1.0f - 0.5f * z) + z * z * pc;
	const int q = int(k) & 3;
	switch (q)
	{
	case 0: *s = sr;  *c = cr;  break;
	case 1: *s = cr;  *c = -sr; break;
	case 2: *s = -sr; *c = -cr; break;
	default:*s = -cr; *c = sr;  break;
	}
}
inline float det_sin(float x) { float s, c; det_sincos(x, &s, &c); return s; }
inline float det_cos(float x) { float s, c; det_sincos(x, &s, &c); return c; }

// atan on [0,inf) by Cephes reduction, then atan2 by quadrant
inline float det_atan_pos(float t)
{
	float y0, x;
	if (t > 2.414213562373095f)      { y0 = 1.5707963267948966f; x = -1.0f / t; }
	else if (t > 0.4142135623730950f){ y0 = 0.7853981633974483f; x = (t - 1.0f) / (t + 1.0f); }
	else                             { y0 = 0.0f;                x = t; }
	const float z = x * x;
	float p = 8.05374449538e-2f * z - 1.38776856032e-1f;
	p = p * z + 1.99777106478e-1f;
	p = p * z - 3.33329491539e-1f;
	return y0 + (p * z * x + x);
}
inline float det_atan2(float y, float x)
{
	if (x != x || y != y) return 0.0f;
	if (x == 0.0f && y == 0.0f) return 0.0f;
	const float ay = y < 0.0f ? -y : y;
	const float ax = x < 0.0f ? -x : x;
	float a;
	if (ax == 0.0f) a = 1.5707963267948966f;
	else a = det_atan_pos(ay / ax);
	if (x < 0.0f) a = PI_F - a;
	return y < 0.0f ? -a : a;
}

// log2 / exp2 / pow for the tonemapper only (x > 0)
inline float det_log2(float x)
{
	u32 b = f2bits(x);
	int e = int((b >> 23) & 0xff) - 127;
	float m = bits2f((b & 0x007fffffu) | 0x3f800000u);     // [1,2)
	if (m > 1.41421356237f) { m = m * 0.5f; e += 1; }      // [sqrt(.5), sqrt(2))
	const float t = (m - 1.0f) / (m + 1.0f);
	const float t2 = t * t;
	// 2/ln2 * atanh(t) series
	float p = 0.2222222222f * t2 + 0.2857142857f;
	p = p * t2 + 0.4f;
	p = p * t2 + 0.6666666667f;
	p = p * t2 + 2.0f;
	return float(e) + (p * t) * 1.44269504088896341f;
}
inline float det_exp2(float x)
{
	if (x < -126.0f) return 0.0f;
	if (x > 127.0f) return finf();
	const float fl = floorf(x);
	const float f = x - fl;                                  // [0,1)
	const float y = f * 0.693147180559945309f;
	float p = 1.3888889e-3f * y + 8.3333333e-3f;
	p = p * y + 4.1666667e-2f;
	p = p * y + 1.6666667e-1f;
	p = p * y + 0.5f;
	p = p * y + 1.0f;
	p = p * y + 1.0f;
	return p * bits2f(u32(int(fl) + 127) << 23);
}
inline float det_pow(float x, float y)
{
	if (x != x) return x;
	if (!(x > 0.0f)) return 0.0f;
	if (!finite_f(x)) return x;
	return det_exp2(y * det_log2(x));
}

// ---------------------------------------------------------------------------------------------
// vectors  (contrib/cugar/linalg/vector_inl.h)
// ---------------------------------------------------------------------------------------------
struct V2 { float x, y; };
struct V3 { float x, y, z; V3() {} V3(float a, float b, float c) : x(a), y(b), z(c) {} explicit V3(float a) : x(a), y(a), z(a) {} };
struct V4 { float x, y, z, w; V4() {} V4(float a, float b, float c, float d) : x(a), y(b), z(c), w(d) {} V3 xyz() const { return V3(x, y, z); } };

inline V3 operator+(V3 a, V3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator*(V3 a, V3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline V3 operator*(V3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
inline V3 operator*(float s, V3 a) { return V3(s * a.x, s * a.y, s * a.z); }
inline V3 operator/(V3 a, float s) { return V3(a.x / s, a.y / s, a.z / s); }   // vector_inl.h:161-168: per-component divide
inline V3 operator-(V3 a) { return V3(-a.x, -a.y, -a.z); }
inline V4 operator*(V4 a, V4 b) { return V4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
inline V4 operator*(V4 a, float s) { return V4(a.x * s, a.y * s, a.z * s, a.w * s); }
inline V4 operator+(V4 a, V4 b) { return V4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

inline float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }        // vector_inl.h:319-327
inline V3 cross(V3 a, V3 b) { return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); } // :353-359
inline float length(V3 a) { return sqrtf(dot(a, a)); }
inline V3 normalize(V3 a) { const float l = length(a); return l > 0.0f ? a / l : a; }  // :345-349
inline float max_comp(V3 a) { return maxf(a.x, maxf(a.y, a.z)); }                    // :530-533, numbers.h:944-947
inline float average(V3 a) { return ((0.0f + a.x) + a.y + a.z) / 3.0f; }             // :692-699
inline bool  finite3(V3 a) { return finite_f(a.x) && finite_f(a.y) && finite_f(a.z); }
inline V3 lerp3(V3 a, V3 b, float u) { return a * (1.0f - u) + b * u; }               // :717-720

// contrib/cugar/linalg/vector_inl.h:391-420  (NOT normalised)
inline V3 orthogonal(V3 v)
{
	if (v.x * v.x < v.y * v.y)
	{
		if (v.x * v.x < v.z * v.z) return V3(0.0f, -v.z, v.y);
		else                       return V3(-v.y, v.x, 0.0f);
	}
	else
	{
		if (v.y * v.y < v.z * v.z) return V3(v.z, 0.0f, -v.x);
		else                       return V3(-v.y, v.x, 0.0f);
	}
}

// contrib/cugar/linalg/vector_inl.h:748-797 : 10:10:10 normal packing
inline u32 pack_normal(V3 n)
{
	const float ex = saturate(n.x * 0.5f + 0.5f), ey = saturate(n.y * 0.5f + 0.5f), ez = saturate(n.z * 0.5f + 0.5f);
	return f2u(ex * 1023.0f) | (f2u(ey * 1023.0f) << 10) | (f2u(ez * 1023.0f) << 20);
}
inline V3 unpack_normal(u32 p)
{
	const float x = float(p & 0x3ffu) / 1023.0f, y = float((p >> 10) & 0x3ffu) / 1023.0f, z = float((p >> 20) & 0x3ffu) / 1023.0f;
	return V3(x * 2.0f - 1.0f, y * 2.0f - 1.0f, z * 2.0f - 1.0f);
}

// IEEE binary16 <-> binary32, round-to-nearest-even (CUDA __floats2half2_rn / __half22float2,
// src/kernels/optix_payload.h:75-78, src/mesh/MeshCompression.h:36-68)
inline uint16_t f2h(float f)
{
	const u32 x = f2bits(f);
	const u32 sign = (x >> 16) & 0x8000u;
	const u32 ax = x & 0x7fffffffu;
	if (ax >= 0x7f800000u) return uint16_t(sign | 0x7c00u | ((ax > 0x7f800000u) ? 0x200u : 0u));
	if (ax >= 0x477ff000u) return uint16_t(sign | 0x7c00u);             // rounds to >= 65520 -> inf
	if (ax < 0x33000001u) return uint16_t(sign);                         // < 2^-25 (or == ) -> 0
	int e = int(ax >> 23) - 127;
	u32 m = (ax & 0x7fffffu) | 0x800000u;
	int shift;
	u32 he;
	if (e < -14) { shift = 13 + (-14 - e); he = 0; }
	else         { shift = 13; he = u32(e + 15); }
	u32 hm = m >> shift;
	const u32 rem = m & ((1u << shift) - 1u);
	const u32 half = 1u << (shift - 1);
	if (rem > half || (rem == half && (hm & 1u))) hm++;
	u32 h;
	if (he == 0) h = hm;                       // subnormal (hm may carry into exponent bit 10: correct)
	else         h = ((he << 10) + (hm - 0x400u));   // hm includes the implicit bit (0x400); carry propagates
	return uint16_t(sign | h);
}
inline float h2f(uint16_t h)
{
	const u32 sign = u32(h & 0x8000u) << 16;
	const u32 e = (h >> 10) & 0x1fu;
	const u32 m = h & 0x3ffu;
	if (e == 0)
	{
		if (m == 0) return bits2f(sign);
		const float v = float(m) * (1.0f / 16777216.0f);     // m * 2^-24
		return bits2f(f2bits(v) | sign);
	}
	if (e == 31) return bits2f(sign | 0x7f800000u | (m << 13));
	return bits2f(sign | ((e + 112u) << 23) | (m << 13));
}

// contrib/cugar/bits/morton.h:84-107,139-154
inline u32 morton3_10(u32 x, u32 y, u32 z)
{
	x = (x | (x << 16)) & 0x030000FFu; x = (x | (x << 8)) & 0x0300F00Fu; x = (x | (x << 4)) & 0x030C30C3u; x = (x | (x << 2)) & 0x09249249u;
	y = (y | (y << 16)) & 0x030000FFu; y = (y | (y << 8)) & 0x0300F00Fu; y = (y | (y << 4)) & 0x030C30C3u; y = (y | (y << 2)) & 0x09249249u;
	z = (z | (z << 16)) & 0x030000FFu; z = (z | (z << 8)) & 0x0300F00Fu; z = (z | (z << 4)) & 0x030C30C3u; z = (z | (z << 2)) & 0x09249249u;
	return x | (y << 1) | (z << 2);
}
inline u64 morton60(u32 x, u32 y, u32 z)
{
	return (u64(morton3_10(x >> 10, y >> 10, z >> 10)) << 30) | u64(morton3_10(x & 1023u, y & 1023u, z & 1023u));
}

} // namespace orc
