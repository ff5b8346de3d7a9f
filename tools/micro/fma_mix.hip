// micro-benchmark: issue rate of v_fma_mix_f32 (f16 operand folded into an fp32 FMA) against v_fma_f32 and against v_cvt_f32_ubyte + v_fma_f32 on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, uint32_t w, float A, float B, int iters)
{
	float acc[8]; for (int i = 0; i < 8; ++i) acc[i] = float(threadIdx.x + i);
	uint32_t q = w + threadIdx.x;
	for (int it = 0; it < iters; ++it)
	{
		#pragma unroll
		for (int i = 0; i < 8; ++i)
		{
			if (MODE == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(A), "v"(B));
			if (MODE == 1) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(acc[i]) : "v"(q), "v"(A));
			if (MODE == 2) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(acc[i]) : "v"(q), "v"(A));
			if (MODE == 3) { float t; asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(t) : "v"(q)); asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(t), "v"(A)); }
			if (MODE == 4) asm volatile("v_max3_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(A), "v"(B));
			if (MODE == 5) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(*reinterpret_cast<double*>(&acc[i & 6])) : "v"(*reinterpret_cast<double*>(&A)), "v"(*reinterpret_cast<double*>(&B)));
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
int main()
{
	float* d; CHECK(hipMalloc(&d, 256 * 8 * 256 * 4));
	const int iters = 20000;
	const char* names[6] = { "v_fma_f32", "v_fma_mix_f32 (lo half)", "v_fma_mix_f32 (hi half)", "v_cvt_f32_ubyte1 + v_fma_f32", "v_max3_f32", "v_pk_fma_f32 (2 flops/lane)" };
	float ms[6] = { run<0>(d, iters), run<1>(d, iters), run<2>(d, iters), run<3>(d, iters), run<4>(d, iters), run<5>(d, iters) };
	// 2048 blocks x 4 waves = 8192 waves = 8 per SIMD; per wave iters x 8 instructions (x2 for mode 3)
	for (int m = 0; m < 6; ++m)
	{
		const double winst = 8192.0 * iters * 8 * (m == 3 ? 2 : 1);
		printf("%-32s %8.3f ms  %.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", names[m], ms[m], ms[m] * 1e-3 * 2.4e9 / (winst / 1024.0));
	}
	return 0;
}
