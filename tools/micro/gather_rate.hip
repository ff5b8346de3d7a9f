// micro-benchmark (round 5): how many 16-byte gathers per clock a CU's vector-memory path serves when every lane of a wave reads a DIFFERENT record -- the access
// pattern of the traversal kernel's node step (five 16-byte loads of an 80-byte node per lane) -- against the same bytes fetched cooperatively (five adjacent
// lanes read the five 16-byte pieces of ONE record, so an instruction touches 13 records instead of 64).  The records form a random cycle (a pointer chase, like
// the traversal: the next record's index comes out of the loaded data), the array is sized to stay in L2 / Infinity Cache, 8 waves per SIMD, no arithmetic.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/gather_rate tools/micro/gather_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>
// MODE 0..: per-lane gather of LOADS x 16 B of an 80-byte record (LOADS = 1, 3, 4, 5; 8 reads 128-byte records);  MODE 100: cooperative, 5 lanes per 80-byte record
template <int LOADS, int STRIDE16>
__global__ __launch_bounds__(256) void gather(const uint4* __restrict__ rec, uint32_t n_rec, uint32_t* out, int iters)
{
	uint32_t idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u % n_rec;
	uint32_t acc = 0;
	for (int it = 0; it < iters; ++it)
	{
		const uint4* p = rec + size_t(idx) * STRIDE16;
		uint4 v[LOADS];
		#pragma unroll
		for (int k = 0; k < LOADS; ++k) v[k] = p[k];
		#pragma unroll
		for (int k = 1; k < LOADS; ++k) acc ^= v[k].y;
		idx = v[0].x;          // the next record
	}
	out[blockIdx.x * 256 + threadIdx.x] = acc ^ idx;
}
// five adjacent lanes fetch one 80-byte record: lane l of a group of 5 reads piece l; the group leader's piece 0 holds the next index, broadcast inside the group
__global__ __launch_bounds__(256) void gather_coop(const uint4* __restrict__ rec, uint32_t n_rec, uint32_t* out, int iters)
{
	const uint32_t lane = threadIdx.x & 63u, grp = lane / 5u, piece = lane - grp * 5u;
	const bool live = lane < 60u;          // 12 groups of 5; lanes 60..63 idle
	uint32_t idx = ((blockIdx.x * 256u + (threadIdx.x & ~63u)) + grp) * 2654435761u % n_rec;
	uint32_t acc = 0;
	for (int it = 0; it < iters; ++it)
	{
		uint4 v = make_uint4(0, 0, 0, 0);
		if (live) v = rec[size_t(idx) * 5 + piece];
		acc ^= v.y;
		idx = __shfl(v.x, int(grp * 5u));          // piece 0's first word
	}
	out[blockIdx.x * 256 + threadIdx.x] = acc ^ idx;
}
template <typename F> float timed(F launch)
{
	hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
	launch(8); (void)hipEventRecord(a); launch(-1); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
	float ms = 0; (void)hipEventElapsedTime(&ms, a, b); return ms;
}
// calibration of rocprofv3's FETCH_SIZE on THIS access pattern (MI355X_MICROARCH.md, HBM: "calibrate on a known byte count in your own access pattern"): a 1 GB
// array (L2 holds 0.4 % of it), every lane at a different record, reading 16 / 64 / 80 / 128 bytes of a 128-byte-aligned record and 80 bytes of packed 80-byte
// records.  tools/calibrate_fetch_size.py runs this under `rocprofv3 --pmc FETCH_SIZE` and divides.
static int calibrate()
{
	const int iters = 200;
	const double lanes = 2048.0 * 256.0 * iters;
	for (int stride16 : { 8, 5 })
	{
		const uint32_t n_rec = stride16 == 8 ? 8000000u : 12800000u;          // 1.02 GB either way
		std::vector<uint32_t> perm(n_rec); std::iota(perm.begin(), perm.end(), 0u);
		std::mt19937 rng(11); std::shuffle(perm.begin(), perm.end(), rng);
		std::vector<uint4> h(size_t(n_rec) * stride16);
		for (uint32_t i = 0; i < n_rec; ++i) for (int k = 0; k < stride16; ++k) h[size_t(perm[i]) * stride16 + k] = make_uint4(perm[(i + 1) % n_rec], i * 7u + k, 0u, 0u);
		uint4* d; uint32_t* o;
		if (hipMalloc(&d, h.size() * sizeof(uint4)) != hipSuccess || hipMalloc(&o, 2048 * 256 * 4) != hipSuccess) return 1;
		(void)hipMemcpy(d, h.data(), h.size() * sizeof(uint4), hipMemcpyHostToDevice);
		auto report = [&](const char* kernel, int bytes_read, float ms) {
			printf("CALIB kernel=%s record_stride=%d bytes_read_per_record=%d record_fetches=%.0f ms=%.3f  -> %.2f G records/s, %.2f TB/s of 128-byte lines if every fetch moves %s\n", kernel,
			       stride16 * 16, bytes_read, lanes, ms, lanes / (ms * 1e-3) / 1e9, lanes * (stride16 == 8 ? 128.0 : 192.0) / (ms * 1e-3) / 1e12, stride16 == 8 ? "one line" : "1.5 lines"); };
		if (stride16 == 8)
		{
			report("gather<1,8>", 16, timed([&](int n) { hipLaunchKernelGGL((gather<1, 8>), dim3(2048), dim3(256), 0, 0, d, n_rec, o, n < 0 ? iters : 0); }));
			report("gather<4,8>", 64, timed([&](int n) { hipLaunchKernelGGL((gather<4, 8>), dim3(2048), dim3(256), 0, 0, d, n_rec, o, n < 0 ? iters : 0); }));
			report("gather<5,8>", 80, timed([&](int n) { hipLaunchKernelGGL((gather<5, 8>), dim3(2048), dim3(256), 0, 0, d, n_rec, o, n < 0 ? iters : 0); }));
			report("gather<8,8>", 128, timed([&](int n) { hipLaunchKernelGGL((gather<8, 8>), dim3(2048), dim3(256), 0, 0, d, n_rec, o, n < 0 ? iters : 0); }));
		}
		else
			report("gather<5,5>", 80, timed([&](int n) { hipLaunchKernelGGL((gather<5, 5>), dim3(2048), dim3(256), 0, 0, d, n_rec, o, n < 0 ? iters : 0); }));
		(void)hipFree(d); (void)hipFree(o);
	}
	return 0;
}
int main(int argc, char** argv)
{
	if (argc > 1 && argv[1][0] == 'c') return calibrate();
	const int iters = 2000;
	for (uint32_t n_rec : { 40000u, 1280000u })          // 3.2 MB (one XCD's L2 holds it) and 102 MB (the bench scene's node array: Infinity Cache)
	{
		for (int stride16 : { 5, 8 })
		{
			std::vector<uint32_t> perm(n_rec); std::iota(perm.begin(), perm.end(), 0u);
			std::mt19937 rng(7); std::shuffle(perm.begin(), perm.end(), rng);
			std::vector<uint4> h(size_t(n_rec) * stride16);
			for (uint32_t i = 0; i < n_rec; ++i) for (int k = 0; k < stride16; ++k) h[size_t(perm[i]) * stride16 + k] = make_uint4(perm[(i + 1) % n_rec], i * 7u + k, 0u, 0u);
			uint4* d; uint32_t* o;
			if (hipMalloc(&d, h.size() * sizeof(uint4)) != hipSuccess || hipMalloc(&o, 2048 * 256 * 4) != hipSuccess) return 1;
			(void)hipMemcpy(d, h.data(), h.size() * sizeof(uint4), hipMemcpyHostToDevice);
			auto report = [&](const char* name, float ms, double records, double requests) {
				printf("%8u records x %3d B  %-44s %8.3f ms  %7.2f G records/s  %6.3f lane-requests (16 B) per clock per CU at 2.4 GHz\n", n_rec, stride16 * 16, name, ms,
				       records / (ms * 1e-3) / 1e9, requests / (ms * 1e-3) / 2.4e9 / 256.0); };
			const double lanes = 2048.0 * 256.0 * iters;
			if (stride16 == 5)
			{
				report("per-lane gather, 1 x 16 B", timed([&](int n) { hipLaunchKernelGGL((gather<1, 5>), dim3(2048), dim3(256), 0, 0, d, n_rec, o, n < 0 ? iters : n); }), lanes, lanes * 1);
				report("per-lane gather, 3 x 16 B", timed([&](int n) { hipLaunchKernelGGL((gather<3, 5>), dim3(2048), dim3(256), 0, 0, d, n_rec, o, n < 0 ? iters : n); }), lanes, lanes * 3);
				report("per-lane gather, 4 x 16 B", timed([&](int n) { hipLaunchKernelGGL((gather<4, 5>), dim3(2048), dim3(256), 0, 0, d, n_rec, o, n < 0 ? iters : n); }), lanes, lanes * 4);
				report("per-lane gather, 5 x 16 B (the node step)", timed([&](int n) { hipLaunchKernelGGL((gather<5, 5>), dim3(2048), dim3(256), 0, 0, d, n_rec, o, n < 0 ? iters : n); }), lanes, lanes * 5);
				report("cooperative, 5 lanes x 16 B per record", timed([&](int n) { hipLaunchKernelGGL(gather_coop, dim3(2048), dim3(256), 0, 0, d, n_rec, o, n < 0 ? iters : n); }), lanes * 12.0 / 64.0, lanes * 60.0 / 64.0);
			}
			else
			{
				report("per-lane gather, 5 x 16 B of a 128-B record", timed([&](int n) { hipLaunchKernelGGL((gather<5, 8>), dim3(2048), dim3(256), 0, 0, d, n_rec, o, n < 0 ? iters : n); }), lanes, lanes * 5);
				report("per-lane gather, 8 x 16 B of a 128-B record", timed([&](int n) { hipLaunchKernelGGL((gather<8, 8>), dim3(2048), dim3(256), 0, 0, d, n_rec, o, n < 0 ? iters : n); }), lanes, lanes * 8);
			}
			(void)hipFree(d); (void)hipFree(o);
		}
	}
	return 0;
}
