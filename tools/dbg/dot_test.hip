#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <cstring>
typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ float dot8(const u32x4 a, const u32x4 b, float acc) {
#pragma unroll
  for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, a[i]), __builtin_bit_cast(bf2, b[i]), acc, false);
  return acc;
}
__global__ void k(const u32x4* a, const u32x4* b, float* o, float* o2) {
  const int l = threadIdx.x;
  float d = dot8(a[l], b[l], 0.f);
  o[l] = d;
  d += __shfl_xor(d, 1, 64);
  d += __shfl_xor(d, 2, 64);
  d += __shfl_xor(d, 4, 64);
  d += __shfl_xor(d, 8, 64);
  o2[l] = d;
}
static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)((u + 0x7FFF + ((u >> 16) & 1)) >> 16); }
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
int main() {
  uint16_t ha[64 * 8], hb[64 * 8];
  for (int i = 0; i < 512; ++i) { ha[i] = f2bf(sinf(i * 0.37f)); hb[i] = f2bf(cosf(i * 0.11f) * 2.f); }
  void *da, *db; float *d_o, *do2;
  hipMalloc(&da, 1024); hipMalloc(&db, 1024); hipMalloc(&d_o, 256); hipMalloc(&do2, 256);
  hipMemcpy(da, ha, 1024, hipMemcpyHostToDevice); hipMemcpy(db, hb, 1024, hipMemcpyHostToDevice);
  k<<<1, 64>>>((const u32x4*)da, (const u32x4*)db, d_o, do2);
  float o[64], o2[64];
  hipMemcpy(o, d_o, 256, hipMemcpyDeviceToHost); hipMemcpy(o2, do2, 256, hipMemcpyDeviceToHost);
  double maxe = 0, maxe2 = 0;
  double ref[64];
  for (int l = 0; l < 64; ++l) {
    double r = 0;
    for (int e = 0; e < 8; ++e) r += (double)bf2f(ha[l * 8 + e]) * bf2f(hb[l * 8 + e]);
    ref[l] = r;
    maxe = fmax(maxe, fabs(r - o[l]));
  }
  for (int l = 0; l < 64; ++l) {
    double r = 0;
    for (int j = 0; j < 16; ++j) r += ref[(l & ~15) + j];
    maxe2 = fmax(maxe2, fabs(r - o2[l]));
  }
  printf("dot8 max err %g  (lane0 got %g want %g); group reduce max err %g\n", maxe, o[0], ref[0], maxe2);
  return 0;
}
