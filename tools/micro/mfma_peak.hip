// Developer micro-benchmark: bare v_mfma_f32_16x16x32_bf16 loop on random operands, 1 or 2 waves per SIMD on every CU.
// Gives the matrix-pipe ceiling the chip actually sustains under DVFS (MI355X_MICROARCH.md "DVFS give-back").
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int NACC>
__global__ __launch_bounds__(512) void k(const bf16x8* in, float* out, int iters, unsigned long long* clk) {
  bf16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = in[threadIdx.x + 512 * i]; b[i] = in[threadIdx.x + 512 * (4 + i)]; }
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
  unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

int main() {
  bf16x8* in; float* out; unsigned long long* clk;
  hipMalloc(&in, 512 * 8 * 16); hipMalloc(&out, 256 * 512 * 4 * 2); hipMalloc(&clk, 16);
  unsigned short* h = (unsigned short*)malloc(512 * 8 * 16);
  for (int i = 0; i < 512 * 8 * 8; ++i) { float f = (rand() / (float)RAND_MAX) * 2 - 1; unsigned u; memcpy(&u, &f, 4); h[i] = u >> 16; }
  hipMemcpy(in, h, 512 * 8 * 16, hipMemcpyHostToDevice);
  for (int threads : {256, 512}) {
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k<16>, dim3(256), dim3(threads), 0, 0, in, out, iters, clk);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      unsigned long long c[2]; hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost);
      double fl = 2.0 * 16 * 16 * 32 * 16.0 * iters * (threads / 64) * 256;
      printf("threads %d: %.3f ms  %.1f TFLOP/s   s_memtime ticks/MFMA %.2f  memtime/realtime %.2f\n", threads, ms, fl / ms / 1e9,
             (double)c[0] / (16.0 * iters), (double)c[0] / (double)c[1]);
    }
  }
  return 0;
}
