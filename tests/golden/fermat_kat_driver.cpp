// fermat_kat_driver.cpp -- OUR driver around two headers of the reference's own Fermat layer (round 6, VERDICT r5 task 5), used by tests/golden/make_cugar_kat.py in the
// BUILD container only: src/tiled_sampling.h (mj_3d / build_tiled_samples_3d: the Cranley-Patterson shift layers every renderer's TiledSequence is built from, with the C
// library's rand() bound to the Microsoft CRT's generator the reference was built against) and src/mis_utils.h (balance / power / cutoff heuristics).
// Same HONEST LABEL as cugar_kat_driver.cpp: <types.h> includes <cuda_runtime.h>, for which the generator writes a one-line stand-in -- not "the reference built here".
// Not reachable this way: src/bsdf.h (includes renderer_view.h -> mesh storage on thrust device vectors, OptiX's optixu_matrix.h through camera.h) and src/camera.h
// (optix::Matrix): they would need stand-ins for OptiX and thrust, i.e. stand-ins with behaviour; the layered Bsdf stays anchored one level down (its cugar lobes) and by
// the replay of contrib/cugar/bsdf/bsdf_test.h.
#define _finite(x) std::isfinite(x)
#define _isnan(x) std::isnan(x)
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
using std::isfinite; using std::isnan;
// the Microsoft CRT's rand(): state0 = 1, state = state * 214013 + 2531011, rand() = (state >> 16) & 0x7fff, RAND_MAX = 0x7fff
static uint32_t g_crt_state = 1u;
static int crt_rand() { g_crt_state = g_crt_state * 214013u + 2531011u; return int((g_crt_state >> 16) & 0x7fffu); }
#include <algorithm>
#include <stdlib.h>
#include <types.h>
#include <cugar/basic/numbers.h>
#include <mis_utils.h>
// every header tiled_sampling.h includes has been read by now: the two names below are rebound for ITS text only.  `random` collides with glibc's long random(void)
// (the Microsoft CRT has no such function), so the reference's own `inline float random()` gets another name here -- a rename, not a replacement.
#undef RAND_MAX
#define RAND_MAX 0x7fff
#define rand crt_rand
#define random fermat_random
#include <tiled_sampling.h>
#undef rand
#undef random

static void P(const char* tag, int n, const double* v) { printf("%s", tag); for (int i = 0; i < n; ++i) printf(" %.17g", v[i]); printf("\n"); }
#define ROW(tag, ...) do { const double _v[] = { __VA_ARGS__ }; P(tag, int(sizeof(_v) / sizeof(double)), _v); } while (0)

int main()
{
	// two set-ups in a row on ONE generator state, as RenderingContextImpl::init (setup(72, 256)) followed by a renderer's init consume it; small tiles here
	const uint32_t cases[2][3] = { { 16, 16, 4 }, { 8, 8, 6 } };
	for (int c = 0; c < 2; ++c)
	{
		const uint32_t X = cases[c][0], Y = cases[c][1], Z = cases[c][2];
		std::vector<float> s(size_t(X) * Y * Z * 3, 0.0f);
		build_tiled_samples_3d(X, Y, Z, s.data());
		for (size_t i = 0; i < s.size(); i += 8)
			ROW(c == 0 ? "tiled_16_16_4" : "tiled_8_8_6", double(i), double(s[i]), double(s[i + 1]), double(s[i + 2]), double(s[i + 3]), double(s[i + 4]), double(s[i + 5]), double(s[i + 6]), double(s[i + 7]));
	}
	for (int i = 0; i < 8; ++i) ROW("crt_rand_after", double(i), double(crt_rand()));
	uint32_t lcg = 777u;
	auto U = [&]() { lcg = lcg * 1664525u + 1013904223u; return float(lcg >> 8) * (1.0f / 16777216.0f); };
	for (int i = 0; i < 48; ++i)
	{
		float p1 = U() * (i % 3 == 0 ? 100.0f : 1.0f), p2 = U() * (i % 5 == 0 ? 1.0e-3f : 1.0f);
		if (i == 7) p1 = INFINITY; if (i == 11) p2 = INFINITY; if (i == 13) { p1 = INFINITY; p2 = INFINITY; } if (i == 17) p1 = 0.0f; if (i == 19) p2 = 0.0f;
		ROW("heuristics", double(p1), double(p2), double(balance_heuristic(p1, p2)), double(power_heuristic(p1, p2)), double(cutoff_heuristic(p1, p2)));
	}
	return 0;
}
