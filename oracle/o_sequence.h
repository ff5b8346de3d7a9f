// ORACLE — TEST INFRASTRUCTURE ONLY (see o_math.h header).
//
// o_sequence.h : CPU restatement of the tiled Cranley-Patterson QMC sequence.
//   src/tiled_sequence.h:53-157, src/tiled_sequence.cu:37-110, src/tiled_sampling.h:44-55,92-160,287-337
// The shift layers >= 7 come from MSVC's rand() (SURVEY Fact 10 / Appendix A.16): state0 = 1,
// state = state*214013 + 2531011 (mod 2^32), rand() = (state >> 16) & 0x7fff, RAND_MAX = 0x7fff.  This LCG is the
// documented behaviour of the Microsoft CRT, an external fact (not in the reference tree).
#pragma once
#include "o_math.h"
#include <vector>
#include <string>
#include <cstdio>
#include <utility>

namespace orc {

struct MsvcRand
{
	u32 state;
	MsvcRand() : state(1u) {}
	int next() { state = state * 214013u + 2531011u; return int((state >> 16) & 0x7fffu); }
	// src/tiled_sampling.h:44-47
	float random() { return float(next()) / float(0x7fff); }
	// src/tiled_sampling.h:51-55
	u32 irandom(u32 N) { const float r = random(); return minu(f2u(r * float(N)), N - 1); }
};

// layout : point (x,y,z) component i lives at  z*X*Y*3 + i*X*Y + y*X + x   (src/tiled_sampling.h:69-88)
inline size_t ss_index(u32 X, u32 Y, u32 x, u32 y, u32 z, u32 comp) { return size_t(z) * X * Y * 3 + size_t(comp) * X * Y + size_t(X) * y + x; }

// src/tiled_sampling.h:92-160 (the `#if 1 / #if 1` branch)
inline void mj_3d(u32 X, u32 Y, u32 Z, float* s, MsvcRand& rng)
{
	for (u32 k = 0; k < Z; ++k)
		for (u32 j = 0; j < Y; ++j)
			for (u32 i = 0; i < X; ++i)
			{
				// NB: integer operands are converted to float one addition at a time, as the C++ expression does
				s[ss_index(X, Y, i, j, k, 0)] = (float(i) + (float(j) + (float(k) + rng.random()) / float(Z)) / float(Y)) / float(X);
				s[ss_index(X, Y, i, j, k, 1)] = (float(j) + (float(k) + (float(i) + rng.random()) / float(X)) / float(Z)) / float(Y);
				s[ss_index(X, Y, i, j, k, 2)] = (float(k) + (float(i) + (float(j) + rng.random()) / float(Y)) / float(X)) / float(Z);
			}
	// exchange the points among Z slices
	for (u32 k = 0; k < Z; ++k)
		for (u32 j = 0; j < Y; ++j)
			for (u32 i = 0; i < X; ++i)
			{
				const u32 r = k + rng.irandom(Z - k);
				for (u32 c = 0; c < 3; ++c)
					std::swap(s[ss_index(X, Y, i, j, k, c)], s[ss_index(X, Y, i, j, r, c)]);
			}
	for (u32 k = 0; k < Z; ++k)
	{
		// exchange the X components among Y columns
		for (u32 j = 0; j < Y; ++j)
		{
			const u32 r = j + rng.irandom(Y - j);
			for (u32 i = 0; i < X; ++i)
				std::swap(s[ss_index(X, Y, i, j, k, 0)], s[ss_index(X, Y, i, r, k, 0)]);
		}
		// exchange the Y components among X rows
		for (u32 i = 0; i < X; ++i)
		{
			const u32 r = i + rng.irandom(X - i);
			for (u32 j = 0; j < Y; ++j)
				std::swap(s[ss_index(X, Y, i, j, k, 1)], s[ss_index(X, Y, r, j, k, 1)]);
		}
	}
}

// src/tiled_sampling.h:287-308
inline void build_tiled_samples_3d(u32 X, u32 Y, u32 Z, float* samples, MsvcRand& rng)
{
	mj_3d(X, Y, Z, samples, rng);
	const u32 SLICE = X * Y;
	for (u32 z = 0; z < Z; ++z)
		for (u32 i = 0; i < SLICE; ++i)
		{
			const u32 r = i + rng.irandom(SLICE - i);
			for (u32 c = 0; c < 3; ++c)
				std::swap(samples[size_t(z) * SLICE * 3 + i + size_t(c) * SLICE], samples[size_t(z) * SLICE * 3 + r + size_t(c) * SLICE]);
		}
}

// src/tiled_sampling.h:312-337 — overwrite layers with <dir>/samples-<z>.dat while they exist (AoS float3 -> SoA)
inline u32 load_samples(const char* dir, u32 X, u32 Y, u32 Z, float* samples)
{
	u32 z = 0;
	for (; z < Z; ++z)
	{
		char name[1024];
		std::snprintf(name, sizeof(name), "%s/samples-%u.dat", dir, z);
		FILE* f = std::fopen(name, "rb");
		if (!f) break;
		std::vector<float> slice(size_t(X) * Y * 3);
		const size_t got = std::fread(slice.data(), sizeof(float) * 3, size_t(X) * Y, f);
		std::fclose(f);
		if (got != size_t(X) * Y) break;
		for (u32 i = 0; i < X * Y; ++i)
			for (u32 c = 0; c < 3; ++c)
				samples[size_t(z) * X * Y * 3 + i + size_t(c) * X * Y] = slice[size_t(i) * 3 + c];
	}
	return z;
}

// TiledSequence : src/tiled_sequence.h:109-157, src/tiled_sequence.cu:62-110
struct TiledSequence
{
	u32 n_dimensions, tile_size;
	std::vector<float> shifts;     // [n_dims][tile*tile]
	std::vector<float> samples;    // [n_dims][tile*tile]

	// `rng` carries the process-wide rand() state: RenderingContextImpl::init consumes it first with setup(72,256)
	// (src/renderer.cu:949-953), then PathTracer::init with setup(6*(L+1),256) (src/renderers/pathtracer_impl.h:148-150).
	void setup(u32 n_dims, u32 tile, const char* samples_dir, MsvcRand& rng)
	{
		n_dimensions = n_dims; tile_size = tile;
		shifts.assign(size_t(tile) * tile * n_dims, 0.0f);
		build_tiled_samples_3d(tile, tile, n_dims / 3, shifts.data(), rng);
		if (samples_dir) load_samples(samples_dir, tile, tile, n_dims / 3, shifts.data());
		samples.assign(shifts.size(), 0.0f);
	}
	void set_instance(u32 instance)
	{
		const size_t T = size_t(tile_size) * tile_size;
		for (u32 d = 0; d < n_dimensions; ++d)
		{
			const float s = randfloat(d, instance + 1);
			for (size_t p = 0; p < T; ++p)
				samples[p + d * T] = fmod1_pos(s + shifts[p + d * T]);
		}
	}
	// src/tiled_sequence.h:93-105
	float sample_2d(u32 px, u32 py, u32 dim) const
	{
		const size_t T = size_t(tile_size) * tile_size;
		const u32 shift = (px & (tile_size - 1)) + (py & (tile_size - 1)) * tile_size;
		const u32 tile  = ((px / tile_size) & (tile_size - 1)) + ((py / tile_size) & (tile_size - 1)) * tile_size;
		return fmod1_pos(samples[dim * T + (shift & (T - 1))] + shifts[dim * T + (tile & (T - 1))]);
	}
};

} // namespace orc
