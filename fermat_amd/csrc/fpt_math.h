// fpt_math.h — scalar/vector math for the gfx950 path-tracing kernels (and the host-side builders that must agree
// with them bit for bit).  Everything is fp32 with a fixed operation order; the translation unit is compiled with
// -ffp-contract=off and IEEE-rounded divide/sqrt, so that a kernel result can be compared for EQUALITY with a CPU
// evaluation of the same formulae ("fpt detmath v1").
//
// What each group stands in for in the reference (paths relative to NVlabs/fermat):
//   bit casts, saturating float->int      contrib/cugar/basic/numbers.h:121,600-606,636-644 and CUDA cvt semantics
//   hash / randfloat                       contrib/cugar/basic/numbers.h:648-657,752-763
//   sincos / atan2 / pow kernels           CUDA sinf,cosf (ggx_common.h:283-284, mappings_inline.h:83-85), atan2f
//                                          (mappings_inline.h:181), powf (src/renderer.cu:99-102)
//   float3 algebra                         contrib/cugar/linalg/vector_inl.h:161-168,319-359,391-420,530-533,692-720
//   normal / half packing                  contrib/cugar/linalg/vector_inl.h:748-797, CUDA __floats2half2_rn
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define FPT_HD __host__ __device__ __forceinline__

namespace fpt {

static constexpr float kPi    = 3.14159265358979323846f;
static constexpr float kTwoPi = 6.28318530717958647693f;

FPT_HD uint32_t as_u32(float f) { return __builtin_bit_cast(uint32_t, f); }
FPT_HD float    as_f32(uint32_t u) { return __builtin_bit_cast(float, u); }
FPT_HD float    inf_f() { return as_f32(0x7f800000u); }
FPT_HD bool     is_finite(float x) { return (as_u32(x) & 0x7f800000u) != 0x7f800000u; }

// comparison-based select min/max (cugar::min/max), as opposed to IEEE fminf/fmaxf below
FPT_HD float sel_min(float a, float b) { return a < b ? a : b; }
FPT_HD float sel_max(float a, float b) { return a > b ? a : b; }
FPT_HD uint32_t sel_min(uint32_t a, uint32_t b) { return a < b ? a : b; }
FPT_HD float ieee_max(float a, float b) { return (a != a) ? b : ((b != b) ? a : (a > b ? a : b)); }
FPT_HD float ieee_min(float a, float b) { return (a != a) ? b : ((b != b) ? a : (a < b ? a : b)); }
FPT_HD float saturate(float x) { return (x != x) ? 0.0f : (x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x)); }
FPT_HD float sqr(float x) { return x * x; }

// float -> uint32 / int32 with CUDA's saturating, NaN->0 behaviour (written out: C++ leaves the out-of-range cast undefined)
FPT_HD uint32_t to_u32_sat(float x)
{
	if (!(x > 0.0f)) return 0u;
	if (x >= 4294967296.0f) return 0xFFFFFFFFu;
	return (uint32_t)x;
}
FPT_HD int32_t to_i32_sat(float x)
{
	if (x != x) return 0;
	if (x >= 2147483648.0f) return 0x7FFFFFFF;
	if (x <= -2147483648.0f) return (int32_t)0x80000000;
	return (int32_t)x;
}
FPT_HD uint32_t quantize(float x, uint32_t n)
{
	int32_t v = to_i32_sat(x * float(n));
	const int32_t hi = int32_t(n - 1);
	v = v > hi ? hi : v;
	v = v < 0 ? 0 : v;
	return uint32_t(v);
}
// x mod 1 : exact for any finite x
FPT_HD float frac_pos(float x) { return x - truncf(x); }
FPT_HD float mod1(float x) { return x > 0.0f ? frac_pos(x) : 1.0f - frac_pos(-x); }

FPT_HD uint32_t hash32(uint32_t a)
{
	a = (a + 0x7ed55d16u) + (a << 12);
	a = (a ^ 0xc761c23cu) ^ (a >> 19);
	a = (a + 0x165667b1u) + (a << 5);
	a = (a + 0xd3a2646cu) ^ (a << 9);
	a = (a + 0xfd7046c5u) + (a << 3);
	a = (a ^ 0xb55a4f09u) ^ (a >> 16);
	return a;
}
FPT_HD float randfloat(uint32_t i, uint32_t p)
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

// ---- detmath v1 ------------------------------------------------------------------------------------------------------
// sin & cos together: quadrant k = floor(x*2/pi + 1/2), r = x - k*pi/2 by a 3-constant split, then the classic
// single-precision minimax polynomials on [-pi/4, pi/4].  Argument range in the renderer: [-pi/4, 2*pi].
FPT_HD void det_sincos(float x, float& s, float& c)
{
	const float k = floorf(x * 0.636619772367581343f + 0.5f);
	float r = x - k * 1.5703125f;
	r = r - k * 4.837512969970703125e-4f;
	r = r - k * 7.54978995489188216e-8f;
	const float z = r * r;
	float ps = -1.9515295891e-4f * z + 8.3321608736e-3f;
	ps = ps * z - 1.6666654611e-1f;
	const float sr = r + r * z * ps;
	float pc = 2.443315711809948e-5f * z - 1.388731625493765e-3f;
	pc = pc * z + 4.166664568298827e-2f;
	const float cr = (1.0f - 0.5f * z) + z * z * pc;
	const int q = int(k) & 3;
	s = (q == 0) ? sr : (q == 1) ? cr : (q == 2) ? -sr : -cr;
	c = (q == 0) ? cr : (q == 1) ? -sr : (q == 2) ? -cr : sr;
}
FPT_HD float det_atan_pos(float t)
{
	float y0, x;
	if (t > 2.414213562373095f)       { y0 = 1.5707963267948966f; x = -1.0f / t; }
	else if (t > 0.4142135623730950f) { y0 = 0.7853981633974483f; x = (t - 1.0f) / (t + 1.0f); }
	else                              { y0 = 0.0f;                x = t; }
	const float z = x * x;
	float p = 8.05374449538e-2f * z - 1.38776856032e-1f;
	p = p * z + 1.99777106478e-1f;
	p = p * z - 3.33329491539e-1f;
	return y0 + (p * z * x + x);
}
FPT_HD float det_atan2(float y, float x)
{
	if (x != x || y != y) return 0.0f;
	if (x == 0.0f && y == 0.0f) return 0.0f;
	const float ay = y < 0.0f ? -y : y;
	const float ax = x < 0.0f ? -x : x;
	float a = (ax == 0.0f) ? 1.5707963267948966f : det_atan_pos(ay / ax);
	if (x < 0.0f) a = kPi - a;
	return y < 0.0f ? -a : a;
}
FPT_HD float det_log2(float x)
{
	const uint32_t b = as_u32(x);
	int e = int((b >> 23) & 0xffu) - 127;
	float m = as_f32((b & 0x007fffffu) | 0x3f800000u);
	if (m > 1.41421356237f) { m = m * 0.5f; e += 1; }
	const float t = (m - 1.0f) / (m + 1.0f);
	const float t2 = t * t;
	float p = 0.2222222222f * t2 + 0.2857142857f;
	p = p * t2 + 0.4f;
	p = p * t2 + 0.6666666667f;
	p = p * t2 + 2.0f;
	return float(e) + (p * t) * 1.44269504088896341f;
}
FPT_HD float det_exp2(float x)
{
	if (x < -126.0f) return 0.0f;
	if (x > 127.0f) return inf_f();
	const float fl = floorf(x);
	const float y = (x - fl) * 0.693147180559945309f;
	float p = 1.3888889e-3f * y + 8.3333333e-3f;
	p = p * y + 4.1666667e-2f;
	p = p * y + 1.6666667e-1f;
	p = p * y + 0.5f;
	p = p * y + 1.0f;
	p = p * y + 1.0f;
	return p * as_f32(uint32_t(int(fl) + 127) << 23);
}
FPT_HD float det_pow(float x, float y)
{
	if (x != x) return x;
	if (!(x > 0.0f)) return 0.0f;
	if (!is_finite(x)) return x;
	return det_exp2(y * det_log2(x));
}

// ---- float3 ----------------------------------------------------------------------------------------------------------
struct f3 { float x, y, z; };
FPT_HD f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
FPT_HD f3 splat3(float a) { return mk3(a, a, a); }
FPT_HD f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
FPT_HD f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
FPT_HD f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
FPT_HD f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
FPT_HD f3 operator*(float s, f3 a) { return mk3(s * a.x, s * a.y, s * a.z); }
FPT_HD f3 operator/(f3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
FPT_HD f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
FPT_HD float dot(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
FPT_HD f3 cross(f3 a, f3 b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
FPT_HD float length(f3 a) { return sqrtf(dot(a, a)); }
FPT_HD f3 normalize(f3 a) { const float l = length(a); return l > 0.0f ? a / l : a; }
FPT_HD float max_comp(f3 a) { return sel_max(a.x, sel_max(a.y, a.z)); }
FPT_HD float average(f3 a) { return (((0.0f + a.x) + a.y) + a.z) / 3.0f; }
FPT_HD bool all_finite(f3 a) { return is_finite(a.x) && is_finite(a.y) && is_finite(a.z); }
FPT_HD f3 lerp(f3 a, f3 b, float u) { return a * (1.0f - u) + b * u; }

// a vector orthogonal to v, not normalised (Fermat's tangent frames are not orthonormal: src/mesh_utils.h:216-217)
FPT_HD f3 orthogonal(f3 v)
{
	const float xx = v.x * v.x, yy = v.y * v.y, zz = v.z * v.z;
	if (xx < yy) return (xx < zz) ? mk3(0.0f, -v.z, v.y) : mk3(-v.y, v.x, 0.0f);
	else         return (yy < zz) ? mk3(v.z, 0.0f, -v.x) : mk3(-v.y, v.x, 0.0f);
}

FPT_HD uint32_t pack_normal(f3 n)
{
	const uint32_t x = to_u32_sat(saturate(n.x * 0.5f + 0.5f) * 1023.0f);
	const uint32_t y = to_u32_sat(saturate(n.y * 0.5f + 0.5f) * 1023.0f);
	const uint32_t z = to_u32_sat(saturate(n.z * 0.5f + 0.5f) * 1023.0f);
	return x | (y << 10) | (z << 20);
}
FPT_HD f3 unpack_normal(uint32_t p)
{
	const float x = float(p & 0x3ffu) / 1023.0f;
	const float y = float((p >> 10) & 0x3ffu) / 1023.0f;
	const float z = float((p >> 20) & 0x3ffu) / 1023.0f;
	return mk3(x * 2.0f - 1.0f, y * 2.0f - 1.0f, z * 2.0f - 1.0f);
}

// binary16 conversions, round-to-nearest-even, done in integer arithmetic so host and device agree
FPT_HD uint32_t float_to_half_bits(float f)
{
	const uint32_t x = as_u32(f);
	const uint32_t sign = (x >> 16) & 0x8000u;
	const uint32_t ax = x & 0x7fffffffu;
	if (ax >= 0x7f800000u) return sign | 0x7c00u | ((ax > 0x7f800000u) ? 0x200u : 0u);
	if (ax >= 0x477ff000u) return sign | 0x7c00u;
	if (ax < 0x33000001u) return sign;
	const int e = int(ax >> 23) - 127;
	const uint32_t m = (ax & 0x7fffffu) | 0x800000u;
	const int shift = (e < -14) ? (13 + (-14 - e)) : 13;
	const uint32_t he = (e < -14) ? 0u : uint32_t(e + 15);
	uint32_t hm = m >> shift;
	const uint32_t rem = m & ((1u << shift) - 1u);
	const uint32_t half = 1u << (shift - 1);
	if (rem > half || (rem == half && (hm & 1u))) hm++;
	const uint32_t h = (he == 0u) ? hm : ((he << 10) + (hm - 0x400u));
	return sign | h;
}
FPT_HD float half_bits_to_float(uint32_t h)
{
	const uint32_t sign = (h & 0x8000u) << 16;
	const uint32_t e = (h >> 10) & 0x1fu;
	const uint32_t m = h & 0x3ffu;
	if (e == 0u)
	{
		if (m == 0u) return as_f32(sign);
		return as_f32(as_u32(float(m) * (1.0f / 16777216.0f)) | sign);
	}
	if (e == 31u) return as_f32(sign | 0x7f800000u | (m << 13));
	return as_f32(sign | ((e + 112u) << 23) | (m << 13));
}
FPT_HD float round_through_half(float f) { return half_bits_to_float(float_to_half_bits(f)); }

// 60-bit Morton code of three 20-bit coordinates (VPL ordering, contrib/cugar/bits/morton.h:84-107,139-154)
FPT_HD uint32_t spread10(uint32_t v)
{
	v = (v | (v << 16)) & 0x030000FFu;
	v = (v | (v << 8)) & 0x0300F00Fu;
	v = (v | (v << 4)) & 0x030C30C3u;
	v = (v | (v << 2)) & 0x09249249u;
	return v;
}
FPT_HD uint64_t morton60(uint32_t x, uint32_t y, uint32_t z)
{
	const uint32_t hi = spread10(x >> 10) | (spread10(y >> 10) << 1) | (spread10(z >> 10) << 2);
	const uint32_t lo = spread10(x & 1023u) | (spread10(y & 1023u) << 1) | (spread10(z & 1023u) << 2);
	return (uint64_t(hi) << 30) | uint64_t(lo);
}

} // namespace fpt
