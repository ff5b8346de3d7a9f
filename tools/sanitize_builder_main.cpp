#include "fpt_bvh.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <thread>
#include <string>
namespace fpt { void build_acceleration(uint32_t, const int32_t*, uint32_t, const float*, HostBvh2&, uint32_t); }
template <class T> std::vector<T> load(const std::string& p) { FILE* f = fopen(p.c_str(), "rb"); fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET); std::vector<T> v(n / sizeof(T)); fread(v.data(), 1, n, f); fclose(f); return v; }
int main(int argc, char** argv)
{
	std::string name = argc > 1 ? argv[1] : "glossy";
	auto idx = load<int32_t>("tools/_build/sanitize/" + name + "_idx.bin"); auto vtx = load<float>("tools/_build/sanitize/" + name + "_vtx.bin");
	uint32_t nt = uint32_t(idx.size() / 4), nv = uint32_t(vtx.size() / 4);
	// two host threads build at the same time (each with its own pool), then a refit
	auto job = [&](int k) {
		fpt::HostBvh2 b; fpt::build_acceleration(nt, idx.data(), nv, vtx.data(), b, 48);
		std::vector<float> v2 = vtx; for (size_t i = 0; i < v2.size(); i += 4) v2[i] += 0.01f * float(k + 1);
		fpt::refit_wide8(nt, idx.data(), nv, v2.data(), b);
		printf("thread %d: %zu nodes, %zu records, stack %u\n", k, b.nodes8.size(), b.tris8.size(), b.stack_need);
	};
	std::thread a(job, 0), c(job, 1); a.join(); c.join();
	return 0;
}
