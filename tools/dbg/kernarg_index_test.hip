// Developer probe: dynamic indexing of arrays inside a by-value kernel argument struct (gfx950, ROCm 7.2).
// Finding: with a 2-byte-element and a 4-byte-element array indexed by the same wave-uniform index, hipcc forms
//   base' = kernarg + 2 x ;  s_load_dword dst, base', soffset = 2 x, offset:...   (4 x in total, but base' is not
// dword aligned for odd x: the scalar load drops the low address bits and returns element x - 1).
//   hipcc --offload-arch=gfx950 -O3 tools/dbg/kernarg_index_test.hip -o /tmp/kit && /tmp/kit
#include <hip/hip_runtime.h>
#include <stdio.h>
struct Args {
  long pad[10];
  int n;
  short s[8];
  int i[8];
  int j[8];
};
__global__ void probe(Args a, int* out) {
  const int x = blockIdx.x & 7;            // wave-uniform index
  const int t = threadIdx.x & 7;           // per-lane index
  if (threadIdx.x == 0) {
    out[blockIdx.x * 4 + 0] = a.s[x];
    out[blockIdx.x * 4 + 1] = a.i[x];
    out[blockIdx.x * 4 + 2] = a.j[x];
  }
  if (threadIdx.x < 8) out[64 + blockIdx.x * 8 + threadIdx.x] = a.i[t] + 1000 * a.s[t];
}
__global__ void probe_int_only(Args a, int* out) {   // the same loads without the 2-byte array in the picture
  const int x = blockIdx.x & 7;
  if (threadIdx.x == 0) {
    out[blockIdx.x * 4 + 1] = a.i[x];
    out[blockIdx.x * 4 + 2] = a.j[x];
  }
}
int main() {
  Args a = {};
  for (int k = 0; k < 8; ++k) { a.s[k] = (short)(10 + k); a.i[k] = 100 + k; a.j[k] = 200 + k; }
  int* d;
  hipMalloc(&d, 4096);
  hipMemset(d, 0, 4096);
  probe<<<8, 64>>>(a, d);
  int h[1024];
  hipMemcpy(h, d, 4096, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int b = 0; b < 8; ++b) {
    printf("block %d: s %d i %d j %d | per-lane:", b, h[b * 4], h[b * 4 + 1], h[b * 4 + 2]);
    for (int t = 0; t < 8; ++t) printf(" %d", h[64 + b * 8 + t]);
    printf("\n");
    bad += h[b * 4] != 10 + b || h[b * 4 + 1] != 100 + b || h[b * 4 + 2] != 200 + b;
    for (int t = 0; t < 8; ++t) bad += h[64 + b * 8 + t] != 100 + t + 1000 * (10 + t);
  }
  printf("%s\n", bad ? "MISMATCH (short + int arrays indexed by one uniform index)" : "all values correct");
  hipMemset(d, 0, 4096);
  probe_int_only<<<8, 64>>>(a, d);
  hipMemcpy(h, d, 4096, hipMemcpyDeviceToHost);
  int bad2 = 0;
  for (int b = 0; b < 8; ++b) bad2 += h[b * 4 + 1] != 100 + b || h[b * 4 + 2] != 200 + b;
  printf("int arrays only: %s\n", bad2 ? "MISMATCH" : "all values correct");
  return bad != 0;
}
