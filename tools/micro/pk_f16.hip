// micro-benchmark (round 5, verdict r4 #3): issue rate on gfx950 of the instructions a packed-fp16 slab test would be made of --
// v_perm_b32 (two plane bytes -> a half pair 1024+q), v_pk_fma_f16, v_pk_max_f16 / v_pk_min_f16, v_pk_add_f16, v_cvt_pkrtz_f16_f32 --
// against the fp32 instructions of the node step as built (v_cvt_f32_ubyteN, v_fma_f32, v_max_f32, v_max3_f32) and the integer
// instructions that assemble the hit bits (v_bfe_u32, v_lshl_or_b32, v_and_or_b32, v_cndmask_b32).
// Eight independent dependency chains per wave, 8 waves per SIMD, so the figure is issue throughput, not latency.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/pk_f16 tools/micro/pk_f16.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

enum { M_FMA32, M_CVT_UB, M_MAX32, M_MAX3_32, M_MIN3_32, M_PERM, M_PK_FMA16, M_PK_MAX16, M_PK_MIN16, M_PK_ADD16, M_CVT_PKRTZ, M_BFE, M_LSHL_OR, M_AND_OR, M_CNDMASK,
       M_PK_FMA32, M_FMA_MIX, M_PK_MUL16, M_PK_ASHR16, M_MIX_NODE32, M_MIX_NODE16, M_COUNT };
static const char* NAMES[M_COUNT] = { "v_fma_f32", "v_cvt_f32_ubyte1", "v_max_f32", "v_max3_f32", "v_min3_f32", "v_perm_b32", "v_pk_fma_f16", "v_pk_max_f16", "v_pk_min_f16",
                                      "v_pk_add_f16", "v_cvt_pkrtz_f16_f32", "v_bfe_u32", "v_lshl_or_b32", "v_and_or_b32", "v_cndmask_b32", "v_pk_fma_f32", "v_fma_mix_f32",
                                      "v_pk_mul_f16", "v_pk_ashrrev_i16",
                                      "mix: 2 children fp32 (12 cvt+12 fma+2 max3+2 max+2 min3+2 min)", "mix: 2 children pk-f16 (6 perm+6 pk_fma+3 pk_max+3 pk_min+1 pk_add)" };
static const int INSTS[M_COUNT] = { 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 32, 19 };

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, uint32_t w, float A, float B, int iters)
{
	float acc[8]; for (int i = 0; i < 8; ++i) acc[i] = float(threadIdx.x + i);
	uint32_t q = w + threadIdx.x, sel = 0x0c040c05u + (threadIdx.x & 1u);
	uint32_t ah = 0x3c003c00u + threadIdx.x, bh = 0x38003800u + threadIdx.x;
	for (int it = 0; it < iters; ++it)
	{
		#pragma unroll
		for (int i = 0; i < 8; ++i)
		{
			uint32_t& u = *reinterpret_cast<uint32_t*>(&acc[i]);
			if (MODE == M_FMA32)     asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(A), "v"(B));
			if (MODE == M_CVT_UB)    asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(acc[i]));
			if (MODE == M_MAX32)     asm volatile("v_max_f32 %0, %1, %0" : "+v"(acc[i]) : "v"(A));
			if (MODE == M_MAX3_32)   asm volatile("v_max3_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(A), "v"(B));
			if (MODE == M_MIN3_32)   asm volatile("v_min3_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(A), "v"(B));
			if (MODE == M_PERM)      asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u) : "v"(q), "v"(sel));
			if (MODE == M_PK_FMA16)  asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(u) : "v"(ah), "v"(bh));
			if (MODE == M_PK_MAX16)  asm volatile("v_pk_max_f16 %0, %1, %0" : "+v"(u) : "v"(ah));
			if (MODE == M_PK_MIN16)  asm volatile("v_pk_min_f16 %0, %1, %0" : "+v"(u) : "v"(ah));
			if (MODE == M_PK_ADD16)  asm volatile("v_pk_add_f16 %0, %1, %0 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(u) : "v"(ah));
			if (MODE == M_PK_MUL16)  asm volatile("v_pk_mul_f16 %0, %1, %0" : "+v"(u) : "v"(ah));
			if (MODE == M_PK_ASHR16) asm volatile("v_pk_ashrrev_i16 %0, 15, %0" : "+v"(u));
			if (MODE == M_CVT_PKRTZ) asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(A));
			if (MODE == M_BFE)       asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(u));
			if (MODE == M_LSHL_OR)   asm volatile("v_lshl_or_b32 %0, %1, %2, %0" : "+v"(u) : "v"(q), "v"(sel));
			if (MODE == M_AND_OR)    asm volatile("v_and_or_b32 %0, %1, %2, %0" : "+v"(u) : "v"(q), "v"(sel));
			if (MODE == M_CNDMASK)   asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u) : "v"(q) : );
			if (MODE == M_PK_FMA32)  asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(*reinterpret_cast<double*>(&acc[i & 6])) : "v"(*reinterpret_cast<double*>(&A)), "v"(*reinterpret_cast<double*>(&B)));
			if (MODE == M_FMA_MIX)   asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(acc[i]) : "v"(q), "v"(A));
		}
		if (MODE == M_MIX_NODE32)
		{
			// the plane arithmetic of two children as the kernel does it today: 12 conversions, 12 FMAs, entry = max3 + max, exit = min3 + min
			#pragma unroll
			for (int c = 0; c < 2; ++c)
			{
				float t[6];
				#pragma unroll
				for (int p = 0; p < 6; ++p)
				{
					if (c == 0) asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(t[p]) : "v"(*reinterpret_cast<uint32_t*>(&acc[p])));
					else        asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(t[p]) : "v"(*reinterpret_cast<uint32_t*>(&acc[p])));
					asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(t[p]) : "v"(A), "v"(B));
				}
				float tn, tf;
				asm volatile("v_max_f32 %0, %1, %2" : "=v"(tn) : "v"(t[2]), "v"(B));
				asm volatile("v_max3_f32 %0, %1, %2, %0" : "+v"(tn) : "v"(t[0]), "v"(t[1]));
				asm volatile("v_min_f32 %0, %1, %2" : "=v"(tf) : "v"(t[5]), "v"(A));
				asm volatile("v_min3_f32 %0, %1, %2, %0" : "+v"(tf) : "v"(t[3]), "v"(t[4]));
				asm volatile("v_sub_f32 %0, %1, %2" : "=v"(acc[6 + c]) : "v"(tf), "v"(tn));
			}
		}
		if (MODE == M_MIX_NODE16)
		{
			// the same two children as packed halves: 6 byte permutes, 6 packed FMAs, 3 packed max, 3 packed min, 1 packed subtract
			uint32_t t[6];
			#pragma unroll
			for (int p = 0; p < 6; ++p)
			{
				asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(t[p]) : "v"(*reinterpret_cast<uint32_t*>(&acc[p])), "v"(q), "v"(sel));
				asm volatile("v_pk_fma_f16 %0, %0, %1, %2 op_sel_hi:[1,0,1]" : "+v"(t[p]) : "v"(ah), "v"(bh));
			}
			uint32_t tn, tf;
			asm volatile("v_pk_max_f16 %0, %1, %2" : "=v"(tn) : "v"(t[2]), "v"(bh));
			asm volatile("v_pk_max_f16 %0, %1, %0" : "+v"(tn) : "v"(t[0]));
			asm volatile("v_pk_max_f16 %0, %1, %0" : "+v"(tn) : "v"(t[1]));
			asm volatile("v_pk_min_f16 %0, %1, %2" : "=v"(tf) : "v"(t[5]), "v"(ah));
			asm volatile("v_pk_min_f16 %0, %1, %0" : "+v"(tf) : "v"(t[3]));
			asm volatile("v_pk_min_f16 %0, %1, %0" : "+v"(tf) : "v"(t[4]));
			asm volatile("v_pk_add_f16 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(*reinterpret_cast<uint32_t*>(&acc[6])) : "v"(tf), "v"(tn));
		}
	}
	float s = 0; for (int i = 0; i < 8; ++i) s += acc[i];
	out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> float run(float* d, int iters)
{
	hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
	hipLaunchKernelGGL(k<MODE>, dim3(256 * 8), dim3(256), 0, 0, d, 0x3c004000u, 1.0001f, 0.5f, 16);
	hipEventRecord(a);
	hipLaunchKernelGGL(k<MODE>, dim3(256 * 8), dim3(256), 0, 0, d, 0x3c004000u, 1.0001f, 0.5f, iters);
	hipEventRecord(b); hipEventSynchronize(b);
	float ms = 0; hipEventElapsedTime(&ms, a, b); return ms;
}
template <int MODE> void report(float* d, int iters, double base_ms)
{
	const float ms = run<MODE>(d, iters);
	// 2048 blocks x 4 waves = 8192 waves = 8 per SIMD
	const bool mix = MODE == M_MIX_NODE32 || MODE == M_MIX_NODE16;
	const double winst = 8192.0 * iters * (mix ? INSTS[MODE] : 8);
	printf("%-80s %9.3f ms  %5.2f cycles per wave-instruction per SIMD at 2.4 GHz%s\n", NAMES[MODE], ms, ms * 1e-3 * 2.4e9 / (winst / 1024.0),
	       mix ? "  (per two children)" : "");
	(void)base_ms;
}
int main()
{
	float* d; CHECK(hipMalloc(&d, 256 * 8 * 256 * 4));
	const int iters = 20000;
	report<M_FMA32>(d, iters, 0); report<M_CVT_UB>(d, iters, 0); report<M_MAX32>(d, iters, 0); report<M_MAX3_32>(d, iters, 0); report<M_MIN3_32>(d, iters, 0);
	report<M_PERM>(d, iters, 0); report<M_PK_FMA16>(d, iters, 0); report<M_PK_MAX16>(d, iters, 0); report<M_PK_MIN16>(d, iters, 0); report<M_PK_ADD16>(d, iters, 0);
	report<M_PK_MUL16>(d, iters, 0); report<M_PK_ASHR16>(d, iters, 0);
	report<M_CVT_PKRTZ>(d, iters, 0); report<M_BFE>(d, iters, 0); report<M_LSHL_OR>(d, iters, 0); report<M_AND_OR>(d, iters, 0); report<M_CNDMASK>(d, iters, 0);
	report<M_PK_FMA32>(d, iters, 0); report<M_FMA_MIX>(d, iters, 0);
	report<M_MIX_NODE32>(d, iters / 4, 0); report<M_MIX_NODE16>(d, iters / 4, 0);
	return 0;
}
