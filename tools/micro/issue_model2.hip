// micro-benchmark (round 6): the issue cost of the wave64 VALU instructions the traversal kernel uses or could use instead -- the table behind tools/isa_classes.py and the
// `valu` roof of bench.py's roofline object.  Same method as issue_model.hip (round 5): 2048 blocks x 256 threads = 8 waves per SIMD, every instruction in 8 independent
// self-dependent chains per wave; cycles per wave-instruction per SIMD at the nominal 2.4 GHz.  New here: compares on their own (round 5's row timed v_cmp + v_mov), VOP2 / VOP3
// selects, carry ops, alignbit, integer min / max, conversions, the IEEE division's pieces, packed fp32, partial EXEC masks (does a wave with 16 or 32 live lanes issue faster?),
// and VALU + SALU / VALU + LDS pairs (do scalar or LDS instructions of the same wave take VALU issue slots?).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/issue_model2 tools/micro/issue_model2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

// %0 = d (VGPR, read and written), %1 = a VGPR pair (read and written by the packed rows), %2 = A, %3 = B (VGPR floats), %4 = q (VGPR uint), %5 = M (SGPR pair), %6 = a VGPR pair
#define OPS(X) \
	X(FMA,        "v_fma_f32 %0, %2, %3, %0") \
	X(MUL,        "v_mul_f32 %0, %2, %0") \
	X(ADD,        "v_add_f32 %0, %2, %0") \
	X(SUB,        "v_sub_f32 %0, %2, %0") \
	X(MOV,        "v_mov_b32 %0, %0") \
	X(FMA_MOD,    "v_fma_f32 %0, |%2|, -%3, %0") \
	X(MUL_LIT,    "v_mul_f32 %0, 0x3a83126f, %0") \
	X(BITOP3,     "v_bitop3_b32 %0, %0, %2, %3 bitop3:0xc8") \
	X(AND,        "v_and_b32 %0, %2, %0") \
	X(OR,         "v_or_b32 %0, %2, %0") \
	X(XOR,        "v_xor_b32 %0, %2, %0") \
	X(AND_OR,     "v_and_or_b32 %0, %0, %2, %3") \
	X(OR3,        "v_or3_b32 %0, %0, %2, %3") \
	X(LSHL,       "v_lshlrev_b32 %0, 3, %0") \
	X(LSHR,       "v_lshrrev_b32 %0, 3, %0") \
	X(ASHR,       "v_ashrrev_i32 %0, 31, %0") \
	X(LSHL_V,     "v_lshlrev_b32 %0, %4, %0") \
	X(LSHL_OR,    "v_lshl_or_b32 %0, %0, 3, %2") \
	X(LSHL_ADD,   "v_lshl_add_u32 %0, %0, 3, %2") \
	X(ALIGNBIT,   "v_alignbit_b32 %0, %0, %2, 31") \
	X(ALIGNBYTE,  "v_alignbyte_b32 %0, %0, %2, 1") \
	X(BFE,        "v_bfe_u32 %0, %0, 5, 3") \
	X(BFI,        "v_bfi_b32 %0, %4, %0, %2") \
	X(PERM,       "v_perm_b32 %0, %0, %2, %4") \
	X(ADD_U32,    "v_add_u32 %0, %2, %0") \
	X(SUB_U32,    "v_sub_u32 %0, %2, %0") \
	X(ADD3_U32,   "v_add3_u32 %0, %0, %2, %3") \
	X(ADDC,       "v_addc_co_u32 %0, vcc, %0, %0, vcc") \
	X(MUL_LO,     "v_mul_lo_u32 %0, %0, %4") \
	X(MUL_U24,    "v_mul_u32_u24 %0, %0, %4") \
	X(MAD_U24,    "v_mad_u32_u24 %0, %0, %4, %2") \
	X(BCNT,       "v_bcnt_u32_b32 %0, %0, %2") \
	X(FFBH,       "v_ffbh_u32 %0, %0") \
	X(FFBL,       "v_ffbl_b32 %0, %0") \
	X(MAX_F32,    "v_max_f32 %0, %2, %0") \
	X(MIN_F32,    "v_min_f32 %0, %2, %0") \
	X(MAX3_F32,   "v_max3_f32 %0, %0, %2, %3") \
	X(MED3_F32,   "v_med3_f32 %0, %0, %2, %3") \
	X(MAX_U32,    "v_max_u32 %0, %2, %0") \
	X(MIN_I32,    "v_min_i32 %0, %2, %0") \
	X(MAX3_U32,   "v_max3_u32 %0, %0, %2, %3") \
	X(CVT_UB0,    "v_cvt_f32_ubyte0 %0, %0") \
	X(CVT_UB3,    "v_cvt_f32_ubyte3 %0, %0") \
	X(CVT_F32_U32,"v_cvt_f32_u32 %0, %0") \
	X(CVT_F32_I32,"v_cvt_f32_i32 %0, %0") \
	X(CVT_U32_F32,"v_cvt_u32_f32 %0, %0") \
	X(CVT_F16,    "v_cvt_f32_f16 %0, %0") \
	X(CVT_PKRTZ,  "v_cvt_pkrtz_f16_f32 %0, %0, %2") \
	X(CMP_VCC,    "v_cmp_le_f32 vcc, %2, %0") \
	X(CMP_SGPR,   "v_cmp_le_f32 s[20:21], %2, %0") \
	X(CMP_U32,    "v_cmp_lt_u32 vcc, %4, %0") \
	X(CMP_CLASS,  "v_cmp_class_f32 vcc, %0, %4") \
	X(CNDMASK_E32,"v_cndmask_b32 %0, %0, %2, vcc") \
	X(CNDMASK_E64,"v_cndmask_b32 %0, %0, %2, %5") \
	X(CNDMASK_INL,"v_cndmask_b32 %0, 0, 16, %5") \
	X(RCP,        "v_rcp_f32 %0, %0") \
	X(SQRT,       "v_sqrt_f32 %0, %0") \
	X(DIV_SCALE,  "v_div_scale_f32 %0, vcc, %0, %2, %0") \
	X(DIV_FMAS,   "v_div_fmas_f32 %0, %0, %2, %3") \
	X(DIV_FIXUP,  "v_div_fixup_f32 %0, %0, %2, %3") \
	X(PK_MUL_F32, "v_pk_mul_f32 %1, %1, %6") \
	X(PK_ADD_F32, "v_pk_add_f32 %1, %1, %6") \
	X(PK_FMA_F32, "v_pk_fma_f32 %1, %1, %6, %6") \
	X(FMA_MIX,    "v_fma_mix_f32 %0, %0, %2, %3 op_sel_hi:[1,0,0]") \
	X(DOT4_U8,    "v_dot4_u32_u8 %0, %0, %4, %0") \
	X(SAD_U8,     "v_sad_u8 %0, %0, %4, %0") \
	X(MBCNT,      "v_mbcnt_lo_u32_b32 %0, %0, %4") \
	X(READLANE,   "v_readfirstlane_b32 s22, %0") \
	X(MOV_DPP,    "v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf") \
	X(FMA_SALU,   "v_fma_f32 %0, %2, %3, %0\n\ts_and_b64 s[20:21], s[20:21], %5") \
	X(FMA_2SALU,  "v_fma_f32 %0, %2, %3, %0\n\ts_and_b64 s[20:21], s[20:21], %5\n\ts_or_b64 s[22:23], s[22:23], %5") \
	X(MAX_SALU,   "v_max_f32 %0, %2, %0\n\ts_and_b64 s[20:21], s[20:21], %5")

enum {
#define X(n, s) OP_##n,
	OPS(X)
#undef X
	OP_COUNT };
static const char* OPN[OP_COUNT] = {
#define X(n, s) s,
	OPS(X)
#undef X
};

template <int OP> __device__ __forceinline__ void op(float& d, float A, float B, uint32_t q, unsigned long long M, double& pd, double pa);
#define ASM_ROW(n, s) \
	template <> __device__ __forceinline__ void op<OP_##n>(float& d, float A, float B, uint32_t q, unsigned long long M, double& pd, double pa) \
	{ asm volatile(s : "+v"(d), "+v"(pd) : "v"(A), "v"(B), "v"(q), "s"(M), "v"(pa) : "vcc", "scc", "s20", "s21", "s22", "s23"); }
OPS(ASM_ROW)

template <int OP>
__global__ __launch_bounds__(256) void k(float* out, float A, float B, int iters, unsigned long long M)
{
	float acc[8]; double pd[8];
	for (int i = 0; i < 8; ++i) { acc[i] = float(threadIdx.x + i) + 1.0f; pd[i] = double(threadIdx.x + i); }
	const uint32_t q = 0x0c040c05u + (threadIdx.x & 1u);
	const double pa = double(threadIdx.x) + 0.5;
	for (int it = 0; it < iters; ++it)
	{
		#pragma unroll
		for (int i = 0; i < 8; ++i) op<OP>(acc[i], A, B, q, M, pd[i], pa);
	}
	float s = 0; for (int i = 0; i < 8; ++i) s += acc[i] + float(pd[i]);
	out[blockIdx.x * 256 + threadIdx.x] = s;
}

// partial EXEC: the same FMA / MAX chains with only some lanes live (set inside the kernel, restored before the store)
template <int OP>
__global__ __launch_bounds__(256) void k_exec(float* out, float A, float B, int iters, unsigned long long live)
{
	float acc[8]; for (int i = 0; i < 8; ++i) acc[i] = float(threadIdx.x + i) + 1.0f;
	if ((live >> (threadIdx.x & 63u)) & 1ull)
	{
		for (int it = 0; it < iters; ++it)
		{
			#pragma unroll
			for (int i = 0; i < 8; ++i)
			{
				if (OP == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(A), "v"(B));
				if (OP == 1) asm volatile("v_max_f32 %0, %1, %0" : "+v"(acc[i]) : "v"(A));
				if (OP == 2) asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(acc[i]));
			}
		}
	}
	float s = 0; for (int i = 0; i < 8; ++i) s += acc[i];
	out[blockIdx.x * 256 + threadIdx.x] = s;
}

// VALU + LDS: one FMA chain step + one ds_read_u8 / ds_read_b64 at a lane-dependent address per step
template <int KIND>
__global__ __launch_bounds__(256) void k_lds(float* out, float A, float B, int iters)
{
	__shared__ uint32_t tab[4096];
	for (int i = threadIdx.x; i < 4096; i += 256) tab[i] = i * 2654435761u;
	__syncthreads();
	float acc[8]; for (int i = 0; i < 8; ++i) acc[i] = float(threadIdx.x + i) + 1.0f;
	uint32_t a = threadIdx.x * 37u, sum = 0;
	for (int it = 0; it < iters; ++it)
	{
		#pragma unroll
		for (int i = 0; i < 8; ++i)
		{
			asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(A), "v"(B));
			if (KIND == 1 && (i & 3) == 0) { sum += reinterpret_cast<const volatile uint8_t*>(tab)[(a + i * 97u + it) & 2047u]; }
			if (KIND == 2 && (i & 3) == 0) { sum += tab[(a + i * 97u + it) & 4095u]; }
		}
	}
	float s = float(sum); for (int i = 0; i < 8; ++i) s += acc[i];
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
	const float m = timed([&](int n) { hipLaunchKernelGGL((k<OP>), dim3(2048), dim3(256), 0, 0, d, 1.0001f, 0.5f, n < 0 ? iters : n, 0x5555aaaa5555aaaaull); });
	const double w = 8192.0 * iters * 8 / 1024.0;
	char name[96]; int j = 0; for (const char* p = OPN[OP]; *p && j < 94; ++p) name[j++] = (*p == '\n' || *p == '\t') ? ' ' : *p; name[j] = 0;
	printf("%-78s %5.2f\n", name, m * 1e-3 * 2.4e9 / w); fflush(stdout);
}
template <int OP> void rows(float* d, int iters) { row<OP>(d, iters); if constexpr (OP + 1 < OP_COUNT) rows<OP + 1>(d, iters); }

int main()
{
	float* d; if (hipMalloc(&d, 2048 * 256 * 4) != hipSuccess) return 1;
	const int iters = 10000;
	printf("cycles per wave-instruction (per asm row) per SIMD at 2.4 GHz, 8 waves per SIMD, 8 independent chains per wave; operands: %%0 d, %%1 / %%6 register pairs, %%2 A, %%3 B, %%4 q (uint), %%5 SGPR pair\n");
	rows<0>(d, iters);
	const double w = 8192.0 * iters * 8 / 1024.0;
	const unsigned long long masks[6] = { ~0ull, 0xFFFFFFFFull, 0xFFFFull, 0x000F000F000F000Full, 0x5555555555555555ull, 0xFFFF0000FFFF0000ull };
	const char* mn[6] = { "64 lanes", "lanes 0-31", "lanes 0-15", "4 lanes of every 16", "every other lane", "lanes 16-31 and 48-63" };
	for (int o = 0; o < 3; ++o)
		for (int m = 0; m < 6; ++m)
		{
			float ms = 0;
			if (o == 0) ms = timed([&](int n) { hipLaunchKernelGGL((k_exec<0>), dim3(2048), dim3(256), 0, 0, d, 1.0001f, 0.5f, n < 0 ? iters : n, masks[m]); });
			if (o == 1) ms = timed([&](int n) { hipLaunchKernelGGL((k_exec<1>), dim3(2048), dim3(256), 0, 0, d, 1.0001f, 0.5f, n < 0 ? iters : n, masks[m]); });
			if (o == 2) ms = timed([&](int n) { hipLaunchKernelGGL((k_exec<2>), dim3(2048), dim3(256), 0, 0, d, 1.0001f, 0.5f, n < 0 ? iters : n, masks[m]); });
			printf("partial EXEC  %-18s %-24s %5.2f\n", o == 0 ? "v_fma_f32" : (o == 1 ? "v_max_f32" : "v_cvt_f32_ubyte1"), mn[m], ms * 1e-3 * 2.4e9 / w); fflush(stdout);
		}
	const float l0 = timed([&](int n) { hipLaunchKernelGGL((k_lds<0>), dim3(2048), dim3(256), 0, 0, d, 1.0001f, 0.5f, n < 0 ? iters : n); });
	const float l1 = timed([&](int n) { hipLaunchKernelGGL((k_lds<1>), dim3(2048), dim3(256), 0, 0, d, 1.0001f, 0.5f, n < 0 ? iters : n); });
	const float l2 = timed([&](int n) { hipLaunchKernelGGL((k_lds<2>), dim3(2048), dim3(256), 0, 0, d, 1.0001f, 0.5f, n < 0 ? iters : n); });
	printf("8 FMA per step: alone %5.2f cycles per FMA; + 2 ds_read_u8 (scattered) + address arithmetic %5.2f; + 2 ds_read_b32 (scattered) %5.2f\n",
	       l0 * 1e-3 * 2.4e9 / w, l1 * 1e-3 * 2.4e9 / w, l2 * 1e-3 * 2.4e9 / w);
	return 0;
}
