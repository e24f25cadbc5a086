// Dense bf16 GEMMs of the standard half (deit/vit.py:14-56,90-134: qkv / proj / fc1 / fc2 of Layer_scale_init_Block),
// hand-written for gfx950 with the block's elementwise tails fused into the epilogue.  SURVEY §8f-3.
//
//   NT problem:  C[M,N] = A[M,K] · B[N,K]^T        (A: token rows, B: nn.Linear weight or its transposed copy;
//                                                    both K-contiguous; forward and input-gradient GEMMs)
//
// One workgroup = 8 waves (2 along M x 4 along N) owns a 256 x 256 output tile; a wave owns 128 x 64 of it as
// 8 x 4 MFMA tiles of v_mfma_f32_16x16x32_bf16 (128 accumulator registers).  Operands are swapped (B rows are the
// MFMA "A" operand) so a lane ends up with 4 consecutive output columns of one token row.
//
// Data movement: K is walked in 64-wide tiles.  A K-tile is cut into four 16 KiB "units" of 128 rows x 128 B, chosen
// so that a unit is exactly what ONE phase of the compute schedule starts to need:
//     unit 0: A rows of every wave's first 64-row half      unit 3: A rows of the second halves
//     unit 1: B rows of the 32-column halves used first      unit 2: the other 32-column halves
// Units stream through an 8-slot LDS ring (128 KiB) by global_load_lds (16 B per lane, source-side XOR swizzle, no
// VGPR staging); unit g+6 is issued in phase g, i.e. a load has ~5 phases (~2.5k cycles) to land.  Completion is
// tracked with counted `s_waitcnt vmcnt(6)` + one raw `s_barrier` per phase - nothing drains the queue in the loop.
// A K-tile is four phases of 16 MFMAs per wave (quadrants of the wave's tile in snake order, so only one operand
// changes between phases); the fragments of phase g+1 are read from LDS while the MFMAs of phase g run (two register
// sets per operand).  Consecutive K-tiles alternate which 32-column half goes first so the prefetch never targets a
// live register set.
//
// Scheduling: tiles are dealt to XCDs in contiguous chunks walked in groups of 8 row-panels (operand panels stay in the
// XCD's L2).  M = 16 448 gives 65 x {5,15,20} tiles on 256 CUs: the last partial round would idle most of the chip, so
// the tiles of that round are split along K over `split` workgroups each (f32 partial slabs + a ticket; the last
// arriver of a tile reduces and runs the epilogue) - see dense_plan().
#include <type_traits>
#include "octic_common.hpp"

namespace octic {

constexpr int DG_BM = 256, DG_BN = 256, DG_BK = 64;
constexpr int DG_UNIT = 128 * 128;            // bytes per unit
constexpr int DG_SLOTS = 8;
constexpr int DG_D = 6;                       // prefetch distance in units
constexpr int DG_LDS = 8 * 128 * 144;         // max(ring 8 x 16 KiB, epilogue tiles 8 x 18 KiB)

struct DgArgs {
  const bf16* A;      // [M, K], row stride lda
  const bf16* B;      // [N, K], row stride ldb
  int64_t lda, ldb;
  int M, N, K;
  // epilogue operands (see DgMode)
  bf16* C;            // [M, N] bf16 primary output (row stride ldc)
  bf16* C2;           // GELU mode: gelu(C)
  int64_t ldc;
  const float* bias;  // [N] or null
  const float* gamma; // [N] or null          (RESID)
  const float* rs;    // [M / rps] or null    (RESID)
  int64_t rps;
  const float* X;     // [M, N] f32 residual stream in  (RESID)
  float* OUT;         // [M, N] f32 residual stream out (RESID)
  const bf16* H;      // [M, N] saved pre-activation     (DGELU)
  float* colsum;      // [gridDim-row-panels, N] partial column sums of the output (DGELU, optional)
  // schedule
  int tiles_m, tiles_n;
  int full_tiles;     // tiles computed by one workgroup each
  int split;          // K-split factor of the remaining tiles (>= 1)
  float* slabs;       // [(tiles - full_tiles) * split] x 256 x 256 f32 partials
  int* tickets;       // [(tiles - full_tiles)] arrival counters (zeroed by the host per launch)
};

enum DgMode { DG_PLAIN = 0, DG_GELU = 1, DG_RESID = 2, DG_DGELU = 3 };

__device__ inline void dg_wait_vmcnt(int n) {
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

template <int MODE>
__global__ __launch_bounds__(512, 1) void dense_nt_kernel(DgArgs a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];   // DG_SLOTS x 16 KiB ring; re-used as 8 x 18 KiB epilogue tiles

  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wid >> 2, wc = wid & 3;
  const bool hi = wr != 0;              // waves 4-7 run one barrier interval behind waves 0-3 (see the loop)
  const int fr = lane & 15, kg = lane >> 4;

  // ---- work item -> (tile, k-range)
  const int bid = blockIdx.x;
  int tile, kt_begin, kt_end, part = 0, rem_idx = -1;
  const int nkt_all = a.K / DG_BK;
  if (bid < a.full_tiles) {
    // XCD-aware bijective remap over the full tiles
    const int nf = a.full_tiles;
    const int xcd = bid & 7, q8 = nf >> 3, r8 = nf & 7;
    tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    kt_begin = 0;
    kt_end = nkt_all;
  } else {
    const int r = bid - a.full_tiles;
    rem_idx = r / a.split;
    part = r - rem_idx * a.split;
    tile = a.full_tiles + rem_idx;
    kt_begin = (int)(((int64_t)nkt_all * part) / a.split);
    kt_end = (int)(((int64_t)nkt_all * (part + 1)) / a.split);
  }
  // tile -> (tm, tn): groups of 8 row panels, column-major inside a group
  int tm, tn;
  {
    const int G = 8;
    const int per_group = G * a.tiles_n;
    const int gidx = tile / per_group;
    const int first_m = gidx * G;
    const int gsz = (a.tiles_m - first_m) < G ? (a.tiles_m - first_m) : G;
    const int in_g = tile - gidx * per_group;
    tm = first_m + in_g % gsz;
    tn = in_g / gsz;
  }
  const int m0 = tm * DG_BM, n0 = tn * DG_BN;
  const int nkt = kt_end - kt_begin;          // >= 2
  const int nunits = 4 * nkt;

  // ---- DMA sources.  A wave issues 2 instructions per unit: unit rows 16*wid + 8*j + (lane>>3), LDS chunk position
  // (lane&7) which must hold source chunk (lane&7) ^ (row&7) = (lane&7) ^ (lane>>3).
  const int drow = lane >> 3;
  const int dch = (lane & 7) ^ drow;
  // Buffer-addressed DMA: one 32-bit per-lane byte offset per operand (row of instruction 0, first half); the K-tile,
  // the second instruction (+8 rows) and the second half (+64 / +32 rows) are uniform and go into the scalar offset.
  // The descriptors carry the exact byte sizes, so rows past M / N read as zeros (no clamping, no out-of-bounds access).
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a.A, 0, (int)((int64_t)a.M * a.lda * 2), 0x27000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)a.B, 0, (int)((int64_t)a.N * a.ldb * 2), 0x27000);
  const int r0 = 16 * wid + drow;                                   // unit row of instruction 0
  const unsigned voA = (unsigned)(((int64_t)(m0 + (r0 >> 6) * 128 + (r0 & 63)) * a.lda + dch * 8) * 2);
  const unsigned voB = (unsigned)(((int64_t)(n0 + (r0 >> 5) * 64 + (r0 & 31)) * a.ldb + dch * 8) * 2);
  const int rowA8 = (int)(a.lda * 16), rowB8 = (int)(a.ldb * 16);   // +8 rows, bytes
  const int halfA = (int)(a.lda * 128), halfB = (int)(a.ldb * 64);  // +64 rows of A, +32 rows of B, bytes
  const int kbase = kt_begin * (DG_BK * 2);

  int u_issue = 0;                      // next unit to issue
  // KIND 0 / 3: first / second 64-row halves of A; KIND 1 / 2: first / second 32-column halves of B (compile-time)
  auto issue_unit = [&](auto kind_c) {
    constexpr int KIND = decltype(kind_c)::value;
    constexpr bool isA = KIND == 0 || KIND == 3;
    constexpr bool second = KIND >= 2;
    char* dst = lds + (u_issue & (DG_SLOTS - 1)) * DG_UNIT + wid * 2048;
    const int so = kbase + (u_issue >> 2) * (DG_BK * 2) + (second ? (isA ? halfA : halfB) : 0);
    if constexpr (isA) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)dst, 16, voA, so, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(dst + 1024), 16, voA, so + rowA8, 0, 0);
    } else {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void*)dst, 16, voB, so, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void*)(dst + 1024), 16, voB, so + rowB8, 0, 0);
    }
    ++u_issue;
  };
#define DG_IC(v) std::integral_constant<int, v>()

  // ---- fragment read offsets inside a unit
  const int sw = lane & 7;
  const int rdo0 = fr * 128 + ((kg ^ sw) << 4);
  const int rdo1 = fr * 128 + (((4 + kg) ^ sw) << 4);
  const int a_row0 = wr * 64;          // unit rows of this wave inside units 0 / 3
  const int b_row0 = wc * 32;          // inside units 1 / 2

  bf16x8 Af[2][2][4];                  // [row half][kstep][m-tile]
  bf16x8 Bf[2][2][2];                  // [n-half][kstep][n-tile]
  f32x4 acc[2][2][4][2];               // [m-half][n-half][m-tile][n-tile]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 2; ++q) acc[i][j][p][q] = f32x4{0, 0, 0, 0};

  auto readA = [&](int mh, int unit) {
    const char* base = lds + (unit & (DG_SLOTS - 1)) * DG_UNIT + a_row0 * 128;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      Af[mh][0][mi] = *(const bf16x8*)(base + mi * 2048 + rdo0);
      Af[mh][1][mi] = *(const bf16x8*)(base + mi * 2048 + rdo1);
    }
  };
  auto readB = [&](int nh, int unit) {
    const char* base = lds + (unit & (DG_SLOTS - 1)) * DG_UNIT + b_row0 * 128;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      Bf[nh][0][ni] = *(const bf16x8*)(base + ni * 2048 + rdo0);
      Bf[nh][1][ni] = *(const bf16x8*)(base + ni * 2048 + rdo1);
    }
  };
  // 16 MFMAs of one quadrant; the DMA of the next unit is issued from inside the block (the matrix pipe is busy for 16
  // cycles per MFMA, the issue slots in between are free), which keeps the R intervals short
  auto mma = [&](int mh, int nh, auto kind_c) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
          acc[mh][nh][mi][ni] =
              __builtin_amdgcn_mfma_f32_16x16x32_bf16(Bf[nh][ks][ni], Af[mh][ks][mi], acc[mh][nh][mi][ni], 0, 0, 0);
      if (ks == 0) {
        __builtin_amdgcn_sched_barrier(0);
        if (u_issue < nunits) issue_unit(kind_c);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };

  // The two wave groups alternate roles between consecutive barriers: while waves 0-3 run the MFMAs of phase g (an
  // "M" interval), waves 4-7 issue DMA and read the fragments of the same phase from LDS (their "R" interval), and
  // vice versa in the next interval - the matrix pipe of every SIMD always has one wave feeding it.  Every wave executes
  // barrier, R_g, barrier, M_g, ...; waves 4-7 execute one extra barrier first, which puts them one interval behind.
  // Protocol (unit v is read in phase <= v; R_g reads units <= g+1):
  //   landed : before the barrier that ends its interval, a wave waits until its share of units <= g+2 has landed
  //            (waves 0-3 at the end of M_g, having issued units <= g+6: vmcnt(8); waves 4-7 at the end of R_g, having
  //            issued units <= g+5: vmcnt(6))
  //   reuse  : unit g+6 (issued inside M_g) overwrites unit g-2, whose last reads (waves 4-7, phase <= g-2) completed
  //            more than two barriers earlier
  int g = 0;
  auto wait_landed = [&]() {
    int need = g + 2;
    need = need < nunits - 1 ? need : nunits - 1;
    const int ok = (u_issue - 1) - need;
    if (ok == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (ok == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else dg_wait_vmcnt(ok > 0 ? 2 * ok : 0);
  };

  // ---- prologue: 6 units in flight; units 0, 1 landed before anyone reads
  issue_unit(DG_IC(0));
  issue_unit(DG_IC(1));
  issue_unit(DG_IC(2));
  issue_unit(DG_IC(3));
  issue_unit(DG_IC(0));                // nkt >= 2: units 4, 5 exist
  issue_unit(DG_IC(1));
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  if (hi) __builtin_amdgcn_s_barrier();

#pragma unroll 1
  for (int t = 0; t < nkt; ++t) {
    const int u0 = 4 * t;
    // phase 0: first row half x first column half   (reads per R interval: 4 / 4 / 8 / 8)
    __builtin_amdgcn_s_barrier();
    if (t == 0) readA(0, 0);
    readB(0, u0 + 1);
    if (hi) wait_landed();
    __builtin_amdgcn_s_barrier();
    mma(0, 0, DG_IC(2));
    if (!hi) wait_landed();
    ++g;
    // phase 1: first row half x second column half
    __builtin_amdgcn_s_barrier();
    readB(1, u0 + 2);
    if (hi) wait_landed();
    __builtin_amdgcn_s_barrier();
    mma(0, 1, DG_IC(3));
    if (!hi) wait_landed();
    ++g;
    // phase 2: second row half x second column half
    __builtin_amdgcn_s_barrier();
    readA(1, u0 + 3);
    if (hi) wait_landed();
    __builtin_amdgcn_s_barrier();
    mma(1, 1, DG_IC(0));
    if (!hi) wait_landed();
    ++g;
    // phase 3: second row half x first column half; the first row half of the next K-tile is read meanwhile
    __builtin_amdgcn_s_barrier();
    if (t + 1 < nkt) readA(0, u0 + 4);
    if (hi) wait_landed();
    __builtin_amdgcn_s_barrier();
    mma(1, 0, DG_IC(1));
    if (!hi) wait_landed();
    ++g;
  }
  if (!hi) __builtin_amdgcn_s_barrier();   // re-align the two groups
  __builtin_amdgcn_s_barrier();            // every wave is done with the ring: LDS is free for the epilogue

  // ---- split-K tail tiles: publish the partial tile, the last arriver of the tile reduces (MI355X guide, split-K recipe:
  // plain stores -> every wave drains -> barrier -> lane 0 agent release -> ticket; reducer: agent acquire -> barrier)
  if (rem_idx >= 0 && a.split > 1) {
    float* slab = a.slabs + ((int64_t)rem_idx * a.split + part) * (DG_BM * DG_BN);
    // slab layout: [wave][acc register index][lane] float4 -> fully coalesced 16-byte stores and loads
    f32x4* sw4 = (f32x4*)slab + (int64_t)wid * 32 * 64 + lane;
#pragma unroll
    for (int mh = 0; mh < 2; ++mh)
#pragma unroll
      for (int nh = 0; nh < 2; ++nh)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) sw4[(((mh * 2 + nh) * 4 + mi) * 2 + ni) * 64] = acc[mh][nh][mi][ni];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int* flag = (int*)lds;                 // the ring is idle now
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const int old = __hip_atomic_fetch_add(a.tickets + rem_idx, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      flag[0] = (old == a.split - 1) ? 1 : 0;
      if (old == a.split - 1) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
    __syncthreads();
    const int last = flag[0];
    __syncthreads();                       // flag word is part of the staging area below
    if (last == 0) return;
    for (int p = 0; p < a.split; ++p) {
      if (p == part) continue;
      const f32x4* o4 = (const f32x4*)(a.slabs + ((int64_t)rem_idx * a.split + p) * (DG_BM * DG_BN)) +
                        (int64_t)wid * 32 * 64 + lane;
#pragma unroll
      for (int mh = 0; mh < 2; ++mh)
#pragma unroll
        for (int nh = 0; nh < 2; ++nh)
#pragma unroll
          for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) acc[mh][nh][mi][ni] += o4[(((mh * 2 + nh) * 4 + mi) * 2 + ni) * 64];
    }
  }

  // ---- epilogue.  Each wave stages its 128 x 64 block (+ bias, rounded to bf16) in its own LDS tile in the MFMA layout
  // (lane (fr, kg) of tile (m-tile, n-tile): token row fr, output columns 4 kg .. 4 kg + 3) and reads it back row-wise:
  // a global instruction then moves 8 rows x 128 contiguous bytes (16 B per lane) instead of 16 rows x 32 bytes.
  constexpr int SRS = 144;                   // staged row stride: 128 B of data + 16 B (bank spread of the 8-byte writes)
  char* const stg = lds + wid * (128 * SRS);
#pragma unroll
  for (int nh = 0; nh < 2; ++nh)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int nl = nh * 32 + ni * 16 + kg * 4;
      const int n = n0 + wc * 64 + nl;
      f32x4 bv = {0, 0, 0, 0};
      if (a.bias && n < a.N) bv = *(const f32x4*)(a.bias + n);
#pragma unroll
      for (int mh = 0; mh < 2; ++mh)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          const f32x4 v = acc[mh][nh][mi][ni] + bv;
          const bf16x4 cb = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
          *(bf16x4*)(stg + (mh * 64 + mi * 16 + fr) * SRS + nl * 2) = cb;
        }
    }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // a wave's LDS operations complete in order; the tile is private
  const int srow = lane >> 3, sch = lane & 7;
  const int n = n0 + wc * 64 + sch * 8;
  const bool nok = n < a.N;
  f32x4 gm0 = {1, 1, 1, 1}, gm1 = {1, 1, 1, 1};
  if (MODE == DG_RESID && a.gamma && nok) {
    gm0 = *(const f32x4*)(a.gamma + n);
    gm1 = *(const f32x4*)(a.gamma + n + 4);
  }
#pragma unroll 4
  for (int it = 0; it < 16; ++it) {
    const int row = it * 8 + srow;
    const int m = m0 + wr * 128 + row;
    if (m >= a.M || !nok) continue;
    const u32x4 raw = *(const u32x4*)(stg + row * SRS + sch * 16);
    const bf16x8 cb = __builtin_bit_cast(bf16x8, raw);
    bf16* cp = a.C + (int64_t)m * a.ldc + n;
    if (MODE == DG_PLAIN) {
      *(u32x4*)cp = raw;
    } else if (MODE == DG_GELU) {
      *(u32x4*)cp = raw;                     // pre-activation, kept for the backward
      bf16x8 y;
#pragma unroll
      for (int e = 0; e < 8; ++e) y[e] = (bf16)gelu_exact((float)cb[e]);        // F.gelu of the bf16-rounded h
      *(bf16x8*)(a.C2 + (int64_t)m * a.ldc + n) = y;
    } else if (MODE == DG_RESID) {
      *(u32x4*)cp = raw;                     // branch output, needed for d gamma
      const float rsv = a.rs ? a.rs[m / a.rps] : 1.0f;
      const float* xp = a.X + (int64_t)m * a.N + n;
      f32x4 x0 = *(const f32x4*)xp, x1 = *(const f32x4*)(xp + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        x0[e] += rsv * gm0[e] * (float)cb[e];
        x1[e] += rsv * gm1[e] * (float)cb[4 + e];
      }
      float* op = a.OUT + (int64_t)m * a.N + n;
      *(f32x4*)op = x0;
      *(f32x4*)(op + 4) = x1;
    } else {                                 // DG_DGELU: dh = gelu'(h) * g
      const bf16x8 h = *(const bf16x8*)(a.H + (int64_t)m * a.ldc + n);
      bf16x8 d;
#pragma unroll
      for (int e = 0; e < 8; ++e) d[e] = (bf16)(gelu_grad((float)h[e]) * (float)cb[e]);
      *(bf16x8*)cp = d;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Schedule: `full` tiles get one workgroup each; the remaining r = tiles - full tiles (the partial last round on the
// `cus` workgroup slots) are split along K over `split` workgroups each, so the last round also fills the chip.
struct DgPlan { int tiles_m, tiles_n, full, rem, split, grid; };

inline DgPlan dense_plan(int M, int N, int K, int cus) {
  DgPlan p;
  p.tiles_m = (M + DG_BM - 1) / DG_BM;
  p.tiles_n = (N + DG_BN - 1) / DG_BN;
  const int tiles = p.tiles_m * p.tiles_n;
  const int rounds = tiles / cus;
  p.full = rounds * cus;
  p.rem = tiles - p.full;
  p.split = 1;
  if (p.rem > 0) {
    const int nkt = K / DG_BK;
    int s = cus / p.rem;                                   // workgroup slots per remaining tile
    const int smax = nkt / 4 > 0 ? nkt / 4 : 1;            // keep >= 4 K-tiles per part
    s = s < 1 ? 1 : (s > smax ? smax : s);
    s = s > 8 ? 8 : s;
    // a tail that nearly fills a round is cheaper unsplit (no slab traffic)
    if (p.rem * 10 >= cus * 8) s = 1;
    p.split = s;
  }
  p.grid = p.full + p.rem * p.split;
  return p;
}

}  // namespace octic

using namespace octic;

extern "C" {

int64_t octic_dense_gemm_workspace_bytes(int M, int N, int K) {
  const DgPlan p = dense_plan(M, N, K, 256);
  if (p.split <= 1) return 256;
  return (int64_t)p.rem * p.split * DG_BM * DG_BN * 4 + (int64_t)p.rem * 4 + 256;
}

// mode: 0 plain (C = A B^T + bias), 1 GELU (C = pre-activation, C2 = gelu(C)), 2 RESID (C = branch, OUT = X + rs*gamma*C),
// 3 DGELU (C = gelu'(H) * (A B^T)).
int octic_dense_gemm_nt(const void* A, const void* B, int M, int N, int K, int64_t lda, int64_t ldb, int mode, void* C,
                        void* C2, int64_t ldc, const float* bias, const float* gamma, const float* rs, int64_t rps,
                        const float* X, float* OUT, const void* H, void* workspace, void* stream) {
  if (!A || !B || !C) return OCTIC_ENULL;
  if (M <= 0 || N <= 0 || K <= 0 || (K % DG_BK) || K < 2 * DG_BK || (N % 8) || (lda % 8) || (ldb % 8) || (ldc % 4)) return OCTIC_ESHAPE;
  if ((((uintptr_t)A) | ((uintptr_t)B)) & 15) return OCTIC_EALIGN;
  if (mode == DG_GELU && !C2) return OCTIC_ENULL;
  if (mode == DG_RESID && (!X || !OUT || (rs && rps <= 0))) return OCTIC_ENULL;
  if (mode == DG_DGELU && !H) return OCTIC_ENULL;
  DgArgs a = {};
  a.A = (const bf16*)A; a.B = (const bf16*)B; a.lda = lda; a.ldb = ldb; a.M = M; a.N = N; a.K = K;
  a.C = (bf16*)C; a.C2 = (bf16*)C2; a.ldc = ldc; a.bias = bias; a.gamma = gamma; a.rs = rs; a.rps = rs ? rps : 1;
  a.X = X; a.OUT = OUT; a.H = (const bf16*)H;
  const DgPlan p = dense_plan(M, N, K, 256);
  a.tiles_m = p.tiles_m; a.tiles_n = p.tiles_n; a.full_tiles = p.full; a.split = p.split;
  hipStream_t s = (hipStream_t)stream;
  if (p.split > 1) {
    if (!workspace) return OCTIC_ENULL;
    a.tickets = (int*)workspace;
    a.slabs = (float*)((char*)workspace + (((int64_t)p.rem * 4 + 255) & ~(int64_t)255));
    (void)hipMemsetAsync(a.tickets, 0, (size_t)p.rem * 4, s);
  }
  const int smem = DG_LDS;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)dense_nt_kernel<DG_PLAIN>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    (void)hipFuncSetAttribute((const void*)dense_nt_kernel<DG_GELU>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    (void)hipFuncSetAttribute((const void*)dense_nt_kernel<DG_RESID>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    (void)hipFuncSetAttribute((const void*)dense_nt_kernel<DG_DGELU>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    (void)hipGetLastError();
    attr_done = true;
  }
  switch (mode) {
    case DG_PLAIN: dense_nt_kernel<DG_PLAIN><<<p.grid, 512, smem, s>>>(a); break;
    case DG_GELU: dense_nt_kernel<DG_GELU><<<p.grid, 512, smem, s>>>(a); break;
    case DG_RESID: dense_nt_kernel<DG_RESID><<<p.grid, 512, smem, s>>>(a); break;
    case DG_DGELU: dense_nt_kernel<DG_DGELU><<<p.grid, 512, smem, s>>>(a); break;
    default: return OCTIC_ESHAPE;
  }
  return launch_status();
}

}  // extern "C"
