// fpt_cw8_slots.h — the exact 8 x 8 slot assignment of a wide node's children, shared by the host builder (fpt_bvh.cpp build_wide8) and the device builder
// (fpt_build_lbvh.hip): slot s looks along (s&4 ? +x : -x, s&2 ? +y : -y, s&1 ? +z : -z), and the assignment maximises the sum over children of
// (child centre - node centre) . direction of its slot, so that (slot ^ (7 - ray octant)) descending visits near children first for every ray octant.
#pragma once
#ifdef __HIPCC__
#define FPT_SLOTS_HD __host__ __device__
#else
#define FPT_SLOTS_HD
#endif

namespace fpt {

// exact assignment of <= 8 children to the 8 slots maximising the summed score (Kuhn-Munkres on the 8 x 8 matrix, rows padded with zeros)
FPT_SLOTS_HD inline void assign_slots(const double score[8][8], int n_children, int slot_of[8])
{
	const int N = 8;
	double a[N + 1][N + 1];
	for (int i = 1; i <= N; ++i) for (int j = 1; j <= N; ++j) a[i][j] = (i <= n_children) ? -score[i - 1][j - 1] : 0.0;
	double u[N + 1] = { 0 }, v[N + 1] = { 0 }; int p[N + 1] = { 0 }, way[N + 1] = { 0 };
	for (int i = 1; i <= N; ++i)
	{
		p[0] = i; int j0 = 0;
		double minv[N + 1]; bool used[N + 1];
		for (int j = 0; j <= N; ++j) { minv[j] = 1.0e300; used[j] = false; }
		do
		{
			used[j0] = true;
			const int i0 = p[j0]; double delta = 1.0e300; int j1 = 0;
			for (int j = 1; j <= N; ++j)
				if (!used[j])
				{
					const double cur = a[i0][j] - u[i0] - v[j];
					if (cur < minv[j]) { minv[j] = cur; way[j] = j0; }
					if (minv[j] < delta) { delta = minv[j]; j1 = j; }
				}
			for (int j = 0; j <= N; ++j)
				if (used[j]) { u[p[j]] += delta; v[j] -= delta; } else minv[j] -= delta;
			j0 = j1;
		} while (p[j0] != 0);
		do { const int j1 = way[j0]; p[j0] = p[j1]; j0 = j1; } while (j0);
	}
	for (int j = 1; j <= N; ++j) if (p[j] >= 1 && p[j] <= n_children) slot_of[p[j] - 1] = j - 1;
}


} // namespace fpt
