// micro-benchmark (round 5): what a wave64 VALU instruction costs on gfx950 as a function of the dependency distance -- the same instruction
// (a) in eight independent self-dependent chains per wave (distance 8), (b) with destinations nobody reads (no dependency at all), (c) in ONE chain
// (distance 1); and the node step's plane arithmetic for two children with its instructions grouped by kind instead of chained pair by pair.
// 8 waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/issue_model tools/micro/issue_model.hip
#include <hip/hip_runtime.h>
#include <cstdio>
enum { OP_FMA, OP_CVT, OP_MAX, OP_PERM, OP_PKFMA16, OP_PKMAX16, OP_MUL, OP_ADD, OP_MIN3, OP_AND, OP_LSHL, OP_MUL_SDWA, OP_CNDMASK_S, OP_CMP, OP_SUB, OP_ASHR, OP_BITOP3, OP_BFE, OP_OR3, OP_MOV, OP_CVT_SDWA_ADD, OP_MAD24, OP_COUNT };
static const char* OPN[OP_COUNT] = { "v_fma_f32", "v_cvt_f32_ubyte1", "v_max_f32", "v_perm_b32", "v_pk_fma_f16", "v_pk_max_f16", "v_mul_f32", "v_add_f32", "v_min3_f32", "v_and_b32", "v_lshlrev_b32", "v_mul_f32_sdwa byte1 * 2^127", "v_cndmask_b32_e64 (sgpr mask)", "v_cmp_le_f32 vcc", "v_sub_f32", "v_ashrrev_i32", "v_bitop3_b32", "v_bfe_u32", "v_or3_b32", "v_mov_b32", "v_add_f32_sdwa byte2", "v_mad_u32_u24" };
template <int OP> __device__ __forceinline__ void op(float& d, float s, float A, float B, uint32_t q)
{
	if (OP == OP_FMA)     asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(A), "v"(B), "v"(s));
	if (OP == OP_CVT)     asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(d) : "v"(s));
	if (OP == OP_MAX)     asm volatile("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(A), "v"(s));
	if (OP == OP_PERM)    asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(d) : "v"(s), "v"(A), "v"(q));
	if (OP == OP_PKFMA16) asm volatile("v_pk_fma_f16 %0, %1, %2, %3" : "=v"(d) : "v"(A), "v"(B), "v"(s));
	if (OP == OP_PKMAX16) asm volatile("v_pk_max_f16 %0, %1, %2" : "=v"(d) : "v"(A), "v"(s));
	if (OP == OP_MUL)     asm volatile("v_mul_f32 %0, %1, %2" : "=v"(d) : "v"(A), "v"(s));
	if (OP == OP_ADD)     asm volatile("v_add_f32 %0, %1, %2" : "=v"(d) : "v"(A), "v"(s));
	if (OP == OP_MIN3)    asm volatile("v_min3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(A), "v"(B), "v"(s));
	if (OP == OP_AND)     asm volatile("v_and_b32 %0, %1, %2" : "=v"(d) : "v"(A), "v"(s));
	if (OP == OP_LSHL)    asm volatile("v_lshlrev_b32 %0, 3, %1" : "=v"(d) : "v"(s));
	if (OP == OP_MUL_SDWA) asm volatile("v_mul_f32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(d) : "v"(s), "v"(B));
	if (OP == OP_CVT_SDWA_ADD) asm volatile("v_add_f32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "=v"(d) : "v"(s), "v"(B));
	if (OP == OP_CNDMASK_S) asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(d) : "v"(s), "v"(A), "s"(0x5555aaaa5555aaaaull));
	if (OP == OP_CMP)     asm volatile("v_cmp_le_f32_e32 vcc, %1, %2\n\tv_mov_b32 %0, %1" : "=v"(d) : "v"(s), "v"(A) : "vcc");
	if (OP == OP_SUB)     asm volatile("v_sub_f32 %0, %1, %2" : "=v"(d) : "v"(A), "v"(s));
	if (OP == OP_ASHR)    asm volatile("v_ashrrev_i32 %0, 31, %1" : "=v"(d) : "v"(s));
	if (OP == OP_BITOP3)  asm volatile("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xc8" : "=v"(d) : "v"(s), "v"(A), "v"(B));
	if (OP == OP_BFE)     asm volatile("v_bfe_u32 %0, %1, 5, 3" : "=v"(d) : "v"(s));
	if (OP == OP_OR3)     asm volatile("v_or3_b32 %0, %1, %2, %3" : "=v"(d) : "v"(s), "v"(A), "v"(B));
	if (OP == OP_MOV)     asm volatile("v_mov_b32 %0, %1" : "=v"(d) : "v"(s));
	if (OP == OP_MAD24)   asm volatile("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(d) : "v"(s), "v"(A), "v"(B));
}
// DIST: 8 = eight chains, 0 = no dependency (source is a loop invariant), 1 = one chain
template <int OP, int DIST>
__global__ __launch_bounds__(256) void k(float* out, float A, float B, int iters)
{
	float acc[8]; for (int i = 0; i < 8; ++i) acc[i] = float(threadIdx.x + i);
	const uint32_t q = 0x0c040c05u + (threadIdx.x & 1u);
	const float inv = float(threadIdx.x) * 0.25f;
	for (int it = 0; it < iters; ++it)
	{
		#pragma unroll
		for (int i = 0; i < 8; ++i)
		{
			if (DIST == 8) op<OP>(acc[i], acc[i], A, B, q);
			if (DIST == 0) op<OP>(acc[i], inv, A, B, q);
			if (DIST == 1) op<OP>(acc[0], acc[0], A, B, q);
		}
	}
	float s = 0; for (int i = 0; i < 8; ++i) s += acc[i];
	out[blockIdx.x * 256 + threadIdx.x] = s;
}
// plane arithmetic of two children, grouped by kind (what a scheduler may do with it): GROUPED fp32 = 12 cvt, 12 fma, 4 max/max3, 4 min/min3; f16 = 6 perm, 6 pk_fma, 3 pk_max, 3 pk_min, pk_add
template <int KIND>
__global__ __launch_bounds__(256) void node(float* out, float A, float B, int iters)
{
	float acc[8]; for (int i = 0; i < 8; ++i) acc[i] = float(threadIdx.x + i);
	uint32_t q = 0x0c040c05u + (threadIdx.x & 1u), ah = 0x3c003c00u + threadIdx.x, bh = 0x38003800u + threadIdx.x;
	for (int it = 0; it < iters; ++it)
	{
		if (KIND == 0)
		{
			float t[12];
			#pragma unroll
			for (int p = 0; p < 6; ++p) asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(t[p]) : "v"(acc[p]));
			#pragma unroll
			for (int p = 0; p < 6; ++p) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(t[6 + p]) : "v"(acc[p]));
			#pragma unroll
			for (int p = 0; p < 12; ++p) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(t[p]) : "v"(A), "v"(B));
			float tn0, tn1, tf0, tf1;
			asm volatile("v_max_f32 %0, %1, %2" : "=v"(tn0) : "v"(t[2]), "v"(B));
			asm volatile("v_max_f32 %0, %1, %2" : "=v"(tn1) : "v"(t[8]), "v"(B));
			asm volatile("v_min_f32 %0, %1, %2" : "=v"(tf0) : "v"(t[5]), "v"(A));
			asm volatile("v_min_f32 %0, %1, %2" : "=v"(tf1) : "v"(t[11]), "v"(A));
			asm volatile("v_max3_f32 %0, %1, %2, %0" : "+v"(tn0) : "v"(t[0]), "v"(t[1]));
			asm volatile("v_max3_f32 %0, %1, %2, %0" : "+v"(tn1) : "v"(t[6]), "v"(t[7]));
			asm volatile("v_min3_f32 %0, %1, %2, %0" : "+v"(tf0) : "v"(t[3]), "v"(t[4]));
			asm volatile("v_min3_f32 %0, %1, %2, %0" : "+v"(tf1) : "v"(t[9]), "v"(t[10]));
			asm volatile("v_sub_f32 %0, %1, %2" : "=v"(acc[6]) : "v"(tf0), "v"(tn0));
			asm volatile("v_sub_f32 %0, %1, %2" : "=v"(acc[7]) : "v"(tf1), "v"(tn1));
		}
		else
		{
			uint32_t t[6];
			#pragma unroll
			for (int p = 0; p < 6; ++p) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(t[p]) : "v"(*reinterpret_cast<uint32_t*>(&acc[p])), "v"(q), "v"(q));
			#pragma unroll
			for (int p = 0; p < 6; ++p) asm volatile("v_pk_fma_f16 %0, %0, %1, %2 op_sel_hi:[1,0,1]" : "+v"(t[p]) : "v"(ah), "v"(bh));
			uint32_t tn, tf, tn2, tf2;
			asm volatile("v_pk_max_f16 %0, %1, %2" : "=v"(tn) : "v"(t[2]), "v"(bh));
			asm volatile("v_pk_min_f16 %0, %1, %2" : "=v"(tf) : "v"(t[5]), "v"(ah));
			asm volatile("v_pk_max_f16 %0, %1, %2" : "=v"(tn2) : "v"(t[0]), "v"(t[1]));
			asm volatile("v_pk_min_f16 %0, %1, %2" : "=v"(tf2) : "v"(t[3]), "v"(t[4]));
			asm volatile("v_pk_max_f16 %0, %1, %0" : "+v"(tn) : "v"(tn2));
			asm volatile("v_pk_min_f16 %0, %1, %0" : "+v"(tf) : "v"(tf2));
			asm volatile("v_pk_add_f16 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(*reinterpret_cast<uint32_t*>(&acc[6])) : "v"(tf), "v"(tn));
		}
	}
	float s = 0; for (int i = 0; i < 8; ++i) s += acc[i];
	out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename F> float timed(F launch)
{
	hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
	launch(16); (void)hipEventRecord(a); launch(-1); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
	float ms = 0; (void)hipEventElapsedTime(&ms, a, b); return ms;
}
template <int OP> void row(float* d, int iters)
{
	const float m8 = timed([&](int n) { hipLaunchKernelGGL((k<OP, 8>), dim3(2048), dim3(256), 0, 0, d, 1.0001f, 0.5f, n < 0 ? iters : n); });
	const float m0 = timed([&](int n) { hipLaunchKernelGGL((k<OP, 0>), dim3(2048), dim3(256), 0, 0, d, 1.0001f, 0.5f, n < 0 ? iters : n); });
	const float m1 = timed([&](int n) { hipLaunchKernelGGL((k<OP, 1>), dim3(2048), dim3(256), 0, 0, d, 1.0001f, 0.5f, n < 0 ? iters : n); });
	const double w = 8192.0 * iters * 8 / 1024.0;
	printf("%-20s  8 chains %5.2f   independent %5.2f   one chain %5.2f   cycles per wave-instruction per SIMD at 2.4 GHz (8 waves per SIMD)\n", OPN[OP],
	       m8 * 1e-3 * 2.4e9 / w, m0 * 1e-3 * 2.4e9 / w, m1 * 1e-3 * 2.4e9 / w);
}
int main()
{
	float* d; if (hipMalloc(&d, 2048 * 256 * 4) != hipSuccess) return 1;
	const int iters = 20000;
	row<OP_FMA>(d, iters); row<OP_MUL>(d, iters); row<OP_ADD>(d, iters); row<OP_CVT>(d, iters); row<OP_MAX>(d, iters); row<OP_MIN3>(d, iters); row<OP_PERM>(d, iters);
	row<OP_PKFMA16>(d, iters); row<OP_PKMAX16>(d, iters); row<OP_AND>(d, iters); row<OP_LSHL>(d, iters);
	row<OP_MUL_SDWA>(d, iters); row<OP_CVT_SDWA_ADD>(d, iters); row<OP_CNDMASK_S>(d, iters); row<OP_CMP>(d, iters); row<OP_SUB>(d, iters); row<OP_ASHR>(d, iters); row<OP_BITOP3>(d, iters);
	row<OP_BFE>(d, iters); row<OP_OR3>(d, iters); row<OP_MOV>(d, iters); row<OP_MAD24>(d, iters);
	const float n32 = timed([&](int n) { hipLaunchKernelGGL(node<0>, dim3(2048), dim3(256), 0, 0, d, 1.0001f, 0.5f, n < 0 ? iters / 4 : n); });
	const float n16 = timed([&](int n) { hipLaunchKernelGGL(node<1>, dim3(2048), dim3(256), 0, 0, d, 1.0001f, 0.5f, n < 0 ? iters / 4 : n); });
	const double wn = 8192.0 * (iters / 4) / 1024.0;
	printf("two children, grouped by kind: fp32 (34 instructions) %6.1f cycles, packed f16 (19 instructions) %6.1f cycles\n", n32 * 1e-3 * 2.4e9 / wn, n16 * 1e-3 * 2.4e9 / wn);
	return 0;
}
