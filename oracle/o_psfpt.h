// TEST INFRASTRUCTURE ONLY: CPU restatement of Fermat's path-space-filtering path tracer (`-psfpt`): the PT loop
// of o_pt.h driven by the PSFPT vertex processor.  Parity unpinned at image level.
//
//   PSFPTVertexProcessor (CacheInfo, preprocess_vertex, compute_nee_weights, compute_scattering_weights,
//   accumulate_emissive, accumulate_nee)                    src/psfpt_vertex_processor.h:44-476
//   spatial_hash (the 10-argument overload)                 src/spatial_hash.h:86-167
//   PSFPTOptions, PSFPT::render / render_pass, psf_blending_kernel, PSFRefQueue   src/renderers/psfpt.h:39-85 ; src/renderers/psfpt_impl.h:38-125,275-420
//   clamp_frame                                             src/renderer.cu:314-331,418-428
//   modulate / demodulate                                   src/filters.h:63-82
//
// DEFINED HERE (the reference is order-dependent): the cache cells are float4 accumulated with float atomics by every path that maps
// to them (src/psfpt_vertex_processor.h:170,338-340,382-384); here a cell holds 2^-32 fixed-point 64-bit sums and an integer sample
// count, so that its value does not depend on the order of the additions.  A cell is identified by its 64-bit key; the slot number the
// reference threads through the queues is an implementation detail (insertion order) and does not influence any result.
#pragma once
#include "o_scene.h"
#include "o_sequence.h"
#include <unordered_map>
#include <cmath>

namespace orc {

struct PSFOptions { u32 psf_depth; float psf_width, psf_min_dist, psf_max_prob; u32 psf_temporal_reuse; float firefly_filter; };

static const u32 PSF_INVALID = 0xFFFFFFFFu, PSF_INVALID_SLOT = (1u << 29) - 1u;
inline u32 cache_info(u32 slot, u32 comp, u32 new_entry) { return (slot & PSF_INVALID_SLOT) | ((comp & 3u) << 29) | ((new_entry & 1u) << 31); }
inline u32 ci_slot(u32 c) { return c & PSF_INVALID_SLOT; }
inline u32 ci_comp(u32 c) { return (c >> 29) & 3u; }
inline u32 ci_new(u32 c) { return c >> 31; }
inline bool ci_valid(u32 c) { return ci_slot(c) != PSF_INVALID_SLOT; }

inline float det_log2f(float x) { return det_log2(x); }

inline float cugar_round(float x) { const i32 y = x > 0.0f ? f2i(x) : f2i(x) - 1; return (x - float(y) > 0.5f) ? float(y) + 1.0f : float(y); }      // numbers.h:512-516

// src/spatial_hash.h:86-167
inline u64 spatial_hash(V3 P, V3 N, V3 T, V3 B, V3 bbox_lo, V3 bbox_hi, const float samples[6], float cone_radius, float filter_radius, u32 normal_bits = 4)
{
	const V3 ext = bbox_hi - bbox_lo;
	const float world_extent = max_comp(ext);
	const float float_grid_size = maxf(world_extent / (2.0f * cone_radius), 1.0f);
	const float flog_grid_size = det_log2f(float_grid_size);
	const u32 log_grid_size = f2u(flog_grid_size);
	const float rlog_grid_size = flog_grid_size - float(log_grid_size);
	const u32 log_grid_size_i = log_grid_size + (samples[5] < rlog_grid_size ? 1u : 0u);
	const u32 grid_size = 1u << log_grid_size_i;
	const V2 disk = square_to_unit_disk(samples[0], samples[1]);
	const float rs = filter_radius * cone_radius;
	const float rx = rs * disk.x, ry = rs * disk.y;
	const V3 q = ((P + T * rx) + B * ry) - bbox_lo;
	const V3 loc = float(grid_size) * q / world_extent;
	const u32 lx = f2u(maxf(cugar_round(loc.x), 0.0f)), ly = f2u(maxf(cugar_round(loc.y), 0.0f)), lz = f2u(maxf(cugar_round(loc.z), 0.0f));
	const float nj = float(1u << (normal_bits / 2));
	float phi;
	if (fabsf(N.z) >= 1.0f - 1.0e-5f) phi = 0.0f;
	else { phi = det_atan2(N.y, N.x); phi = phi < 0.0f ? phi + 2.0f * PI_F : phi; }
	float nu = phi / (2.0f * PI_F), nv = (N.z + 1.0f) * 0.5f;
	nu = mod1(nu + samples[3] / nj);
	nv = minf(nv + samples[4] / nj, 1.0f);
	const u32 MAXV = (1u << (normal_bits / 2)) - 1u;
	const u32 normal_i = quantize(nu, MAXV) | (quantize(nv, MAXV) << (normal_bits / 2));
	const u32 comp_mask = (1u << 17) - 1u;
	return (u64(lx & comp_mask) << 0) | (u64(ly & comp_mask) << 17) | (u64(lz & comp_mask) << 34) | (u64(log_grid_size_i) << 51) | (u64(normal_i) << 56);
}

struct PsfState
{
	PSFOptions options;
	bool whatif_nee_vertex_info = false;      // TEST-ONLY (orc_psf_set_whatif): shadow samples carry compute_nee_weights' out_vertex_info; never set by a renderer
	V3 bbox_lo, bbox_hi;
	struct Cell { long long x, y, z; u64 count; };
	std::unordered_map<u64, u32> index;      // key -> slot
	std::vector<u64> keys;
	std::vector<Cell> cells;
	struct Ref { u32 pixel_info, cache; V4 w_d, w_g; };
	std::vector<Ref> refs;

	static long long fixed(float v) { return (long long)rint(double(v) * 4294967296.0); }
	void add(u32 slot, V3 v) { Cell& c = cells[slot]; c.x += fixed(v.x); c.y += fixed(v.y); c.z += fixed(v.z); }
	V3 clamp_sample(V3 v) const { return finite3(v) ? V3(minf(v.x, options.firefly_filter), minf(v.y, options.firefly_filter), minf(v.z, options.firefly_filter)) : V3(0.0f); }
	void clear() { index.clear(); keys.clear(); cells.clear(); }
	u32 insert(u64 key)
	{
		auto it = index.find(key);
		if (it != index.end()) return it->second;
		const u32 slot = u32(cells.size());
		index[key] = slot; keys.push_back(key); cells.push_back(Cell{ 0, 0, 0, 0 });
		return slot;
	}
};

} // namespace orc
