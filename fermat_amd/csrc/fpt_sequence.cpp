// fpt_sequence.cpp — host builder of the tiled QMC shift table (the device part is sequence_kernel in fpt_pt.hip).
//
// What it reproduces (paths relative to NVlabs/fermat): TiledSequence::setup builds a 256x256 tile of N-dimensional
// Cranley-Patterson shifts as N/3 layers of 3-d multi-jittered points (src/tiled_sampling.h:92-160), shuffles each layer
// (:287-308), then overwrites as many leading layers as there are samples-<z>.dat blue-noise files (:312-337,
// src/tiled_sequence.cu:62-98).  All randomness comes from the C runtime's rand() — in the reference build that is the
// Microsoft CRT's LCG, restated in CrtRand (fpt_host.h) — and the stream is shared process-wide, so the caller threads one
// CrtRand through every setup() in call order.
#include "fpt_host.h"
#include <cstdio>
#include <utility>

namespace fpt {
namespace {

struct Lattice
{
	uint32_t X, Y;
	float* data;
	// component c of point (x,y) in layer z: layers are 3 consecutive planes of X*Y floats
	float& at(uint32_t x, uint32_t y, uint32_t z, uint32_t c) { return data[(size_t(z) * 3 + c) * X * Y + size_t(y) * X + x]; }
};

inline float unit_random(CrtRand& g) { return float(g.next()) / float(0x7fff); }                // random(), RAND_MAX = 32767
inline uint32_t index_random(CrtRand& g, uint32_t n)                                             // irandom()
{
	const float v = unit_random(g) * float(n);
	const uint32_t k = (v >= 4294967296.0f) ? 0xFFFFFFFFu : (v > 0.0f ? uint32_t(v) : 0u);
	return k < n - 1 ? k : n - 1;
}

void multi_jitter_layers(Lattice L, uint32_t Z, CrtRand& g)
{
	const float fX = float(L.X), fY = float(L.Y), fZ = float(Z);
	// canonical multi-jittered arrangement: three draws per lattice point, innermost x
	for (uint32_t k = 0; k < Z; ++k)
		for (uint32_t j = 0; j < L.Y; ++j)
			for (uint32_t i = 0; i < L.X; ++i)
			{
				const float a = unit_random(g); L.at(i, j, k, 0) = (float(i) + (float(j) + (float(k) + a) / fZ) / fY) / fX;
				const float b = unit_random(g); L.at(i, j, k, 1) = (float(j) + (float(k) + (float(i) + b) / fX) / fZ) / fY;
				const float c = unit_random(g); L.at(i, j, k, 2) = (float(k) + (float(i) + (float(j) + c) / fY) / fX) / fZ;
			}
	// shuffle whole points across layers
	for (uint32_t k = 0; k < Z; ++k)
		for (uint32_t j = 0; j < L.Y; ++j)
			for (uint32_t i = 0; i < L.X; ++i)
			{
				const uint32_t r = k + index_random(g, Z - k);
				for (uint32_t c = 0; c < 3; ++c) std::swap(L.at(i, j, k, c), L.at(i, j, r, c));
			}
	// within each layer: x components shuffled across rows (one draw per row), y components across columns
	for (uint32_t k = 0; k < Z; ++k)
	{
		for (uint32_t j = 0; j < L.Y; ++j)
		{
			const uint32_t r = j + index_random(g, L.Y - j);
			for (uint32_t i = 0; i < L.X; ++i) std::swap(L.at(i, j, k, 0), L.at(i, r, k, 0));
		}
		for (uint32_t i = 0; i < L.X; ++i)
		{
			const uint32_t r = i + index_random(g, L.X - i);
			for (uint32_t j = 0; j < L.Y; ++j) std::swap(L.at(i, j, k, 1), L.at(r, j, k, 1));
		}
	}
}

} // namespace

void build_shift_table(uint32_t tile, uint32_t n_dims, const char* samples_dir, CrtRand& rng, std::vector<float>& shifts)
{
	const uint32_t Z = n_dims / 3;
	const size_t plane = size_t(tile) * tile;
	shifts.assign(plane * n_dims, 0.0f);
	Lattice L{ tile, tile, shifts.data() };
	multi_jitter_layers(L, Z, rng);
	// Fisher-Yates over the points of each layer
	for (uint32_t z = 0; z < Z; ++z)
	{
		float* layer = shifts.data() + size_t(z) * 3 * plane;
		for (uint32_t i = 0; i < plane; ++i)
		{
			const uint32_t r = i + index_random(rng, uint32_t(plane) - i);
			for (uint32_t c = 0; c < 3; ++c) std::swap(layer[c * plane + i], layer[c * plane + r]);
		}
	}
	// blue-noise layers shipped with the renderer (array-of-float3 files) replace the leading layers
	if (samples_dir)
	{
		std::vector<float> aos(plane * 3);
		for (uint32_t z = 0; z < Z; ++z)
		{
			char path[2048];
			std::snprintf(path, sizeof(path), "%s/samples-%u.dat", samples_dir, z);
			FILE* f = std::fopen(path, "rb");
			if (!f) break;
			const size_t got = std::fread(aos.data(), 3 * sizeof(float), plane, f);
			std::fclose(f);
			if (got != plane) break;
			float* layer = shifts.data() + size_t(z) * 3 * plane;
			for (size_t i = 0; i < plane; ++i)
				for (uint32_t c = 0; c < 3; ++c) layer[c * plane + i] = aos[3 * i + c];
		}
	}
}

} // namespace fpt
