// Argument block shared by the LinearD8 GEMM kernels (gemm.hip, gemm_wreg.hip).
#pragma once
#include "octic_common.hpp"

namespace octic {

struct GemmGroup {
  const char* a;      // A rows
  int64_t a_ld;       // row stride (elements)
  const char* w;      // [N, K] row-major
  char* y;
  int64_t y_ld;
  const char* resid;  // optional (same addressing as y unless lift)
  int64_t r_ld;
  const float* bias;  // optional [N]
  const float* cs;    // optional [N]
  int64_t rows;       // M (or 2M for the E pair group)
  int K, N;
  int pair;           // 1: row mm -> token mm>>1, half mm&1 (adjacent halves of width K / N)
  int n_tiles, m_tiles;
  int chunk, n_chunks;  // n-tiles walked by one block, number of such chunks
  int tile_begin;     // first linear work-item id of this group
  int wgs;            // W-stationary kernel: workgroups that share one column chunk's rows
};

struct GemmArgs {
  GemmGroup g[5];
  int ngroups;
  int total_tiles;
  const float* rs;   // optional per-sample scale
  int64_t rps;       // token rows per sample for rs
  // lift mode: output row = (row / lift_np) * (lift_np + lift_tok0) + lift_tok0 + row % lift_np ; resid row = row % lift_np
  int64_t lift_np;
  int lift_tok0;
  // ring kernel dispatch plan (gemm.hip plan_ring): 0 = groups spread evenly over the XCDs; 1 = per XCD x (workgroups
  // x, x + 8, ... in dispatch order): plan_a[x] items of the short groups, plan_e[x] items of group 0, then short items
  int plan_mode;
  int plan_nshort;    // short groups (1 .. ngroups-1), equal item counts
  int plan_a[8], plan_e[8];
  int plan_eb[8], plan_sb[8];   // first long / short item of XCD x
};

// MFMA operand reads as inline asm with hand-counted waits: inside the GEMM loops hipcc protects every MFMA group with
// `s_waitcnt lgkmcnt(0)` right after the ds_read that feeds it (it does not count LDS returns across the loops' blocks),
// exposing one LDS latency per group.  LDS returns are in order, so "at most N younger reads outstanding" is exact as
// long as no scalar load is in flight (keep kernel-argument reads out of the loops).
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)p; }
__device__ __forceinline__ bf16x8 lds_read16(unsigned addr) {
  bf16x8 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
template <int N>
__device__ __forceinline__ void landed(bf16x8& v) {   // at most N younger LDS reads outstanding => v has arrived
  asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(v) : "n"(N) : "memory");
}

// W-stationary streaming kernel (gemm_wreg.hip); returns -100 when the problem does not qualify.
int launch_wreg(GemmArgs& a, int out_dtype, hipStream_t s);

}  // namespace octic
