// Micro-benchmark: HBM write/read rate of the access shapes used by the GEMM epilogue / operand loads.
// Rows are 10240 B apart (a [M, 5120] bf16 activation), M = 16448.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

// A: fully coalesced 16 B per lane over the whole buffer
__global__ void st_coalesced(u32x4* p, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) p[i] = u32x4{1, 2, 3, 4};
}
// B: MFMA C-layout store: wave covers 16 rows x (4 x BYTES) bytes per instruction; a block of 4 waves writes a
// 128-row x 160-col tile like the GEMM epilogue (tiles enumerated over the [M, 5120] output).
template <int BYTES>
__global__ void st_mfma(char* p, int M, int ldb /*row bytes*/, int tiles_n) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, fr = lane & 15, kg = lane >> 4;
  const int wn = wid & 1, wm = wid >> 1;
  for (int t = blockIdx.x; t < ((M + 127) / 128) * tiles_n; t += gridDim.x) {
    const int mt = t / tiles_n, nt = t % tiles_n;
    for (int j = 0; j < 4; ++j) {
      const int m = mt * 128 + wm * 64 + j * 16 + fr;
      if (m >= M) continue;
      for (int i = 0; i < 5; ++i) {
        const int n = nt * 160 + wn * 80 + i * 16 + kg * 4;   // element index, BYTES/4 bytes per element
        char* q = p + (size_t)m * ldb + (size_t)n * (BYTES / 4);
        if (BYTES == 8) *(u32x2*)q = u32x2{1, 2};
        else *(u32x4*)q = u32x4{1, 2, 3, 4};
      }
    }
  }
}
// C: same tile, but each wave writes whole row pieces: 16 B per lane, consecutive lanes consecutive addresses
__global__ void st_rows(char* p, int M, int ldb, int tiles_n, int tile_row_bytes) {
  const int tid = threadIdx.x;
  const int cpr = tile_row_bytes / 16;
  for (int t = blockIdx.x; t < ((M + 127) / 128) * tiles_n; t += gridDim.x) {
    const int mt = t / tiles_n, nt = t % tiles_n;
    for (int q = tid; q < 128 * cpr; q += 256) {
      const int r = q / cpr, c = q % cpr;
      const int m = mt * 128 + r;
      if (m < M) *(u32x4*)(p + (size_t)m * ldb + (size_t)nt * tile_row_bytes + c * 16) = u32x4{1, 2, 3, 4};
    }
  }
}
// D: loads: fragment-shaped (16 rows x 64 B per instruction) vs staged (8 rows x 128 B)
template <int MODE>
__global__ void ld_pattern(const char* p, int M, int ldb, int kbytes, unsigned* sink) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  unsigned acc = 0;
  for (int mt = blockIdx.x; mt < (M + 127) / 128; mt += gridDim.x) {
    for (int k0 = 0; k0 < kbytes; k0 += 128) {
      if (MODE == 0) {  // fragment: wave rows = 32, lane (fr,kg): 2 ks x 2 j loads of 16 B
        const int fr = lane & 15, kg = lane >> 4;
        for (int j = 0; j < 2; ++j)
          for (int ks = 0; ks < 2; ++ks) {
            const int m = mt * 128 + wid * 32 + j * 16 + fr;
            if (m < M) { u32x4 v = *(const u32x4*)(p + (size_t)m * ldb + k0 + ks * 64 + kg * 16); acc += v[0] ^ v[3]; }
          }
      } else {          // staged: thread (r_in = tid>>3, kc = tid&7) rows r_in + 32 i
        const int r_in = threadIdx.x >> 3, kc = threadIdx.x & 7;
        for (int i = 0; i < 4; ++i) {
          const int m = mt * 128 + r_in + 32 * i;
          if (m < M) { u32x4 v = *(const u32x4*)(p + (size_t)m * ldb + k0 + kc * 16); acc += v[0] ^ v[3]; }
        }
      }
    }
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

template <typename F>
float timeit(F f, int reps = 20) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) f();
  hipEventRecord(a);
  for (int i = 0; i < reps; ++i) f();
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms * 1000.f / reps;
}

int main() {
  const int M = 16448;
  const size_t bytes_bf = (size_t)M * 5120 * 2, bytes_f32 = (size_t)M * 1280 * 4;
  char* buf; hipMalloc(&buf, bytes_bf);
  unsigned* sink; hipMalloc(&sink, 4);
  hipMemset(buf, 1, bytes_bf);
  float us;
  us = timeit([&] { st_coalesced<<<4096, 256>>>((u32x4*)buf, bytes_bf / 16); });
  printf("store coalesced 16B/lane            : %7.1f us  %6.2f TB/s\n", us, bytes_bf / us / 1e6);
  us = timeit([&] { st_mfma<8><<<2048, 256>>>(buf, M, 10240, 32); });
  printf("store MFMA-layout 8B/lane (bf16 out): %7.1f us  %6.2f TB/s\n", us, bytes_bf / us / 1e6);
  us = timeit([&] { st_rows<<<2048, 256>>>(buf, M, 10240, 32, 320); });
  printf("store row pieces 320B, 16B/lane     : %7.1f us  %6.2f TB/s\n", us, bytes_bf / us / 1e6);
  us = timeit([&] { st_mfma<16><<<2048, 256>>>(buf, M, 5120, 8); });
  printf("store MFMA-layout 16B/lane (f32 out): %7.1f us  %6.2f TB/s\n", us, bytes_f32 / us / 1e6);
  us = timeit([&] { st_rows<<<2048, 256>>>(buf, M, 5120, 8, 640); });
  printf("store row pieces 640B f32           : %7.1f us  %6.2f TB/s\n", us, bytes_f32 / us / 1e6);
  // loads over an [M, 1280] bf16 activation (2560 B rows), reading one 320-elem irrep slab (640 B) per row
  us = timeit([&] { ld_pattern<0><<<1024, 256>>>(buf, M, 2560, 640, sink); });
  printf("load fragment-shaped (16 rows x 64B): %7.1f us  %6.2f TB/s\n", us, (double)M * 640 / us / 1e6);
  us = timeit([&] { ld_pattern<1><<<1024, 256>>>(buf, M, 2560, 640, sink); });
  printf("load staged (8 rows x 128B)         : %7.1f us  %6.2f TB/s\n", us, (double)M * 640 / us / 1e6);
  return 0;
}
