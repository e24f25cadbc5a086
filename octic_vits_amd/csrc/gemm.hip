// Irrep-blocked GEMM on MFMA for gfx950 (LinearD8 forward, input gradient, lift patch-embed).
//
// One launch covers the five sub-problems of a LinearD8 (A1,A2,B1,B2: [M,c]x[c,c'] ; E: [2M,2c]x[2c,2c'],
// the two E rows of a token are two GEMM rows sharing W_E).  Tile = 128 rows x (32*NT) outputs,
// 256 threads = 4 waves (2 along N x 2 along M), each wave NT x 4 MFMA tiles of 16x16.
// Operands are swapped on purpose: W is the MFMA "A" operand and X^T the "B" operand, so every lane
// ends up with 4 CONSECUTIVE output channels of ONE token -> 8/16-byte stores into the token row and a
// fused epilogue (bias, layer-scale, drop-path mask, residual) that reads/writes whole vectors.
// LDS tiles are [rows][128 B] (64 bf16 / 32 f32 of K) with the 16-byte chunk index XOR-swizzled by
// (row & 7): conflict-free ds_read_b128 for the 16-lane MFMA operand pattern.  Global->LDS goes through
// registers (16 B per lane, 128 B contiguous per 8 lanes) with the next K tile's loads issued before
// the current tile's MFMAs (software pipeline, one barrier per K tile).  K is short (160/320 at
// ViT-H), so the kernel is bound by HBM/epilogue traffic, not MFMA issue; see DESIGN.md §kernels.
// bf16: v_mfma_f32_16x16x32_bf16.  f32: v_mfma_f32_16x16x4_f32 (exact f32 fma chain) for the
// reference's fp32 tolerances.
#include <stdlib.h>
#include <type_traits>
#include <mutex>
#include <vector>
#include "gemm_args.hpp"

namespace octic {

template <typename T> struct Elem;
template <> struct Elem<float> { static constexpr int EPC = 4; typedef f32x4 frag; };
template <> struct Elem<bf16> { static constexpr int EPC = 8; typedef bf16x8 frag; };

template <typename TOUT>
__device__ inline void store_out4(TOUT* p, f32x4 v);
template <>
__device__ inline void store_out4<float>(float* p, f32x4 v) { *(f32x4*)p = v; }
template <>
__device__ inline void store_out4<bf16>(bf16* p, f32x4 v) {
  bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
  *(bf16x4*)p = o;
}
template <typename TOUT>
__device__ inline f32x4 load_out4(const TOUT* p);
template <>
__device__ inline f32x4 load_out4<float>(const float* p) { return *(const f32x4*)p; }
template <>
__device__ inline f32x4 load_out4<bf16>(const bf16* p) {
  bf16x4 a = *(const bf16x4*)p;
  return f32x4{(float)a[0], (float)a[1], (float)a[2], (float)a[3]};
}

__device__ inline f32x4 mfma_step(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ inline f32x4 mfma_step(f32x4 a, f32x4 b, f32x4 c) {
  // 16 k values per (lane-group, chunk): element s of lane-group kg is k = 4*kg + s.  A and B use the
  // same assignment, and a sum over k does not care about the order.
#pragma unroll
  for (int s = 0; s < 4; ++s) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[s], c, 0, 0, 0);
  return c;
}

constexpr int kBM = 128;

// Zero a staged chunk without a branch (loads are always issued from a clamped, valid address).
__device__ inline u32x4 keep_if(u32x4 v, bool ok) {
  u32x4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) r[i] = ok ? v[i] : 0u;
  return r;
}

// The kernel is written so that the per-step instruction stream is almost only {9 global loads, 9 LDS
// writes, 18 LDS reads, 40 MFMAs}: every address is a per-lane constant plus an immediate, the (n-tile,
// k-tile) position is tracked with counters (no divisions), row offsets / drop-path scales are computed once
// per block, and zero-filling of tile tails is only compiled into the path of blocks that have a tail.
// (An ablation with loads, MFMAs and stores disabled showed the first version spent 1.25 us per step on
// address arithmetic alone — see DESIGN.md.)
template <typename TIN, typename TOUT, int NT>
__global__ __launch_bounds__(256, 2) void linear_d8_kernel(GemmArgs args) {
  constexpr int EPC = Elem<TIN>::EPC;   // elements per 16-byte chunk
  constexpr int BKE = 8 * EPC;          // K elements per tile (128 B rows)
  constexpr int BN = 32 * NT;
  constexpr int MT = 4;
  typedef typename Elem<TIN>::frag frag;
  extern __shared__ __attribute__((aligned(16))) char lds[];  // 2 stages x (BM + BN) rows x 128 B
  constexpr int STAGE = (kBM + BN) * 128;

  // ---- XCD-aware, bijective block remap
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
  const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);

  int gi = 0;
#pragma unroll
  for (int i = 1; i < 5; ++i)
    if (i < args.ngroups && tile >= args.g[i].tile_begin) gi = i;
  const GemmGroup& G = args.g[gi];
  const int lt = tile - G.tile_begin;
  const int mt = lt / G.n_chunks, nc = lt - mt * G.n_chunks;
  const int64_t m0 = (int64_t)mt * kBM;
  const int nt_begin = nc * G.chunk;
  const int nt_count = (G.n_tiles - nt_begin) < G.chunk ? (G.n_tiles - nt_begin) : G.chunk;
  const int K = G.K, N = G.N;
  const int nkt = (K + BKE - 1) / BKE;
  const int steps = nt_count * nkt;
  const int k_rem = K - (nkt - 1) * BKE;                       // valid K elements in the last k-tile
  const bool last_half_only = k_rem <= 4 * EPC;                // second 4-chunk half of the last k-tile is empty
  // does this block touch any partial tile?  (wave-uniform)
  const bool tails = (m0 + kBM > G.rows) || ((nt_begin + nt_count) * BN > N) || (k_rem != BKE && (k_rem % (4 * EPC)) != 0);

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wn = wid & 1, wm = wid >> 1;
  const int fr = lane & 15, kg = lane >> 4;

  // ---- staging constants: thread owns chunk column kc of rows r_in + 32*i
  const int kc = tid & 7, r_in = tid >> 3;
  const int kcol = kc * EPC;                                   // element offset of the chunk inside a k-tile
  const int st_off = r_in * 128 + ((kc ^ (r_in & 7)) << 4);    // LDS byte offset (+ i*4096 per 32 rows)
  const TIN* xp[4];
  unsigned xmask = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int64_t mm = m0 + r_in + 32 * i;
    if (mm < G.rows) xmask |= 1u << i;
    mm = mm < G.rows ? mm : G.rows - 1;
    const int64_t off = G.pair ? (mm >> 1) * G.a_ld + (mm & 1) * (int64_t)K : mm * G.a_ld;
    xp[i] = (const TIN*)G.a + off;
  }
  // load-stream position (runs two steps ahead of the compute stream)
  int l_nt = nt_begin, l_kt = 0;
  const TIN* wp[NT];
  unsigned wmask = 0;
  auto set_w = [&](int nt) {
    wmask = 0;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      int n = nt * BN + r_in + 32 * i;
      if (n < N) wmask |= 1u << i;
      n = n < N ? n : N - 1;
      wp[i] = (const TIN*)G.w + (int64_t)n * K;
    }
  };
  set_w(l_nt);

  // a register set = 9 chunks + the validity mask that goes with them (only consulted when `tails`)
  auto gload = [&](u32x4 (&rx)[4], u32x4 (&rw)[NT], unsigned& mask) {
    int k = l_kt * BKE + kcol;
    const bool kok = k < K;
    k = kok ? k : 0;
    mask = kok ? (xmask | (wmask << 4)) : 0u;
#pragma unroll
    for (int i = 0; i < 4; ++i) rx[i] = *(const u32x4*)(xp[i] + k);
#pragma unroll
    for (int i = 0; i < NT; ++i) rw[i] = *(const u32x4*)(wp[i] + k);
    if (++l_kt == nkt) {
      l_kt = 0;
      ++l_nt;
      if (l_nt < nt_begin + nt_count) set_w(l_nt);
    }
  };
  auto lstore = [&](int stage, const u32x4 (&rx)[4], const u32x4 (&rw)[NT], unsigned mask) {
    char* xs = lds + stage * STAGE + st_off;
    char* ws = xs + kBM * 128;
    if (!tails) {
#pragma unroll
      for (int i = 0; i < 4; ++i) *(u32x4*)(xs + i * 4096) = rx[i];
#pragma unroll
      for (int i = 0; i < NT; ++i) *(u32x4*)(ws + i * 4096) = rw[i];
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) *(u32x4*)(xs + i * 4096) = keep_if(rx[i], (mask >> i) & 1u);
#pragma unroll
      for (int i = 0; i < NT; ++i) *(u32x4*)(ws + i * 4096) = keep_if(rw[i], (mask >> (4 + i)) & 1u);
    }
  };

  f32x4 acc[NT][MT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0, 0, 0, 0};

  // fragment read offsets: row = base + 16*i + fr  =>  (row & 7) == (fr & 7); tile rows advance by immediates
  const int sw = fr & 7;
  const int rd_w = kBM * 128 + (wn * (NT * 16) + fr) * 128;    // + i*2048
  const int rd_x = (wm * 64 + fr) * 128;                       // + j*2048
  const int ch0 = (kg ^ sw) << 4, ch1 = ((4 + kg) ^ sw) << 4;
  int c_kt = 0;                                                // k-tile of the compute stream
  auto compute = [&](int stage) {
    const char* base = lds + stage * STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if (ks == 1 && last_half_only && c_kt == nkt - 1) break;  // wave-uniform: skip an all-zero half tile
      const int ch = ks ? ch1 : ch0;
      frag af[NT], bfr[MT];
#pragma unroll
      for (int i = 0; i < NT; ++i) af[i] = *(const frag*)(base + rd_w + i * 2048 + ch);
#pragma unroll
      for (int j = 0; j < MT; ++j) bfr[j] = *(const frag*)(base + rd_x + j * 2048 + ch);
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = mfma_step(af[i], bfr[j], acc[i][j]);
    }
  };

  // ---- epilogue constants (once per block): per-j output row offset, residual row offset, drop-path scale
  int64_t yoff[MT], roff[MT];
  float rsv[MT];
  unsigned rowok = 0;
#pragma unroll
  for (int j = 0; j < MT; ++j) {
    const int64_t mm = m0 + wm * 64 + j * 16 + fr;
    const bool ok = mm < G.rows;
    if (ok) rowok |= 1u << j;
    const int64_t mc = ok ? mm : 0;
    const int64_t token = G.pair ? (mc >> 1) : mc;
    if (args.lift_np > 0) {
      const int64_t b = mc / args.lift_np, p = mc - b * args.lift_np;
      yoff[j] = (b * (args.lift_np + args.lift_tok0) + args.lift_tok0 + p) * G.y_ld;
      roff[j] = p * G.r_ld;
    } else {
      yoff[j] = G.pair ? (mc >> 1) * G.y_ld + (mc & 1) * (int64_t)N : mc * G.y_ld;
      roff[j] = G.pair ? (mc >> 1) * G.r_ld + (mc & 1) * (int64_t)N : mc * G.r_ld;
    }
    rsv[j] = args.rs ? args.rs[token / args.rps] : 1.0f;
  }
  const int n_lane = wn * (NT * 16) + kg * 4;                  // + i*16 + n0
  const bool has_bias = G.bias != nullptr, has_cs = G.cs != nullptr, has_rs = args.rs != nullptr,
             has_res = G.resid != nullptr;
  int e_nt = nt_begin;
  auto epilogue = [&]() {
    const int n0 = e_nt * BN + n_lane;
    ++e_nt;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int n = n0 + i * 16;
      if (n < N) {
        f32x4 bv = {0, 0, 0, 0}, sv = {1, 1, 1, 1};
        if (has_bias) bv = *(const f32x4*)(G.bias + n);
        if (has_cs) sv = *(const f32x4*)(G.cs + n);
#pragma unroll
        for (int j = 0; j < MT; ++j) {
          if ((rowok >> j) & 1u) {
            f32x4 v = acc[i][j];
            if (has_bias) v += bv;
            if (has_cs) v *= sv;
            if (has_rs) v *= rsv[j];
            if (has_res) v += load_out4<TOUT>((const TOUT*)G.resid + roff[j] + n);
            store_out4<TOUT>((TOUT*)G.y + yoff[j] + n, v);
          }
        }
      }
#pragma unroll
      for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    }
  };

  // ---- main pipeline.  Two register sets: while step s is computed from LDS, the loads of step s+1 are
  // landing in one set and the loads of step s+2 are issued into the other (issue early / write late; the
  // sched_barriers keep hipcc from hoisting the vmcnt wait + LDS writes above the MFMA block).
  u32x4 rxA[4], rwA[NT], rxB[4], rwB[NT];
  unsigned mA = 0, mB = 0;
  gload(rxA, rwA, mA);
  if (steps > 1) gload(rxB, rwB, mB);
  lstore(0, rxA, rwA, mA);
  __syncthreads();
  for (int s = 0; s < steps; s += 2) {
    if (s + 2 < steps) gload(rxA, rwA, mA);
    __builtin_amdgcn_sched_barrier(0);
    compute(0);
    __builtin_amdgcn_sched_barrier(0);
    if (++c_kt == nkt) {
      c_kt = 0;
      epilogue();
    }
    if (s + 1 < steps) lstore(1, rxB, rwB, mB);
    __syncthreads();
    if (s + 1 >= steps) break;
    if (s + 3 < steps) gload(rxB, rwB, mB);
    __builtin_amdgcn_sched_barrier(0);
    compute(1);
    __builtin_amdgcn_sched_barrier(0);
    if (++c_kt == nkt) {
      c_kt = 0;
      epilogue();
    }
    if (s + 2 < steps) lstore(0, rxA, rwA, mA);
    __syncthreads();
  }
}

// ================================================================================================
// LDS-DMA ring variant (the fast path).  Same math and epilogue as linear_d8_kernel, different data movement:
//   * tiles go HBM/L2 -> LDS with global_load_lds (16 B per lane, 1 KiB per wave-instruction, no VGPR staging,
//     no ds_write); the XOR swizzle is applied on the per-lane SOURCE address (the LDS image of a DMA is
//     lane-linear), the reads use the same involution;
//   * a 3-stage ring keeps TWO k-tiles in flight behind the one being multiplied; completion is tracked with
//     counted `s_waitcnt vmcnt(N)` (N = this wave's DMA instructions per tile) and raw `s_barrier`s, so
//     nothing drains the queue inside the loop (PMC on the register-staged kernel: 51 % of wave cycles in
//     SQ_WAIT_ANY, ~1 tile in flight);
//   * tile 128 rows x 80 outputs, 4 waves each owning 32 rows x 80 columns (5 x 2 MFMA tiles): each wave DMAs
//     exactly the X rows it consumes plus a share of the W rows; 3 x 26 KiB of LDS -> 2 workgroups per CU.
// Requirements (else the register-staged kernel runs): K a multiple of one MFMA k-step (32 bf16 / 16 f32) so a
// partially valid k-step never has to be zero-filled — out-of-range ROWS are simply clamped (a garbage row only
// feeds outputs that are never stored).
// ================================================================================================
constexpr int kRingBN = 80;
constexpr int kRingS = 3;
constexpr int kRingStage = (kBM + kRingBN) * 128;

// NT = 16-column MFMA tiles per wave (tile width BN = 16 NT: 80 or 160), S = ring stages.  The wide tile (NT = 10,
// S = 2: 72 KiB, still two workgroups per CU) is for the long-K / narrow-N problems (fc2, input gradients of qkv and
// fc1: N = 160 | 320): with 80-column tiles their X panel went through the L2->LDS path once per n-tile and each X
// k-tile (16 KiB) fed only 20 MFMAs per wave - above the ~70 GB/s per CU that path sustains.
#ifdef OCTIC_RING_TRACE
// developer-only timeline (build with -DOCTIC_RING_TRACE, tools/ring_trace.py): clock stamps of the first 2048 workgroups
__device__ unsigned long long g_ring_trace[2048 * 4 * 64];
extern "C" void* octic_dbg_ring_trace(void) {
  void* p = nullptr;
  (void)hipGetSymbolAddress(&p, HIP_SYMBOL(g_ring_trace));
  return p;
}
#define RTRACE(slot)                                                                                                  \
  do {                                                                                                                \
    if (tile < 2048 && lane == 0 && wid < 4 && (slot) < 63) g_ring_trace[(tile * 4 + wid) * 64 + (slot)] = __builtin_readcyclecounter(); \
  } while (0)
#else
#define RTRACE(slot) do {} while (0)
#endif

// Work item of workgroup `bid` of a ring launch of `nwg` workgroups: group gi, item lt of the group (m-tile x n-chunk).
// Workgroups are dealt round-robin to the eight XCDs; `tile` walks the launch XCD by XCD.
__host__ __device__ inline void ring_item_of(const GemmArgs& args, int nwg, int bid, int& gi, int& lt) {
  const int xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
  const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  gi = 0;
  lt = 0;
  if (args.plan_mode == 1) {
    // planned order (plan_ring below): XCD x runs, in dispatch order, a few short items, its share of the long group, then
    // short items - so that the workgroup slots that must take a third item are slots that started with a short one
    const int l = bid >> 3;
    int a_x = 0, e_x = 0, eb_x = 0, sb_x = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i)          // constant indices + scalar selects (no dynamic indexing of the argument block)
      if (xcd == i) {
        a_x = args.plan_a[i];
        e_x = args.plan_e[i];
        eb_x = args.plan_eb[i];
        sb_x = args.plan_sb[i];
      }
    if (l >= a_x && l < a_x + e_x) {
      gi = 0;
      lt = eb_x + (l - a_x);
    } else {
      const int sidx = sb_x + (l < a_x ? l : l - e_x);
      gi = 1 + sidx % args.plan_nshort;
      lt = sidx / args.plan_nshort;
    }
  } else {
    int T[5];
#pragma unroll
    for (int i = 0; i < 5; ++i)
      T[i] = i < args.ngroups ? (i + 1 < args.ngroups ? args.g[i + 1].tile_begin : nwg) - args.g[i].tile_begin : 0;
    auto cum = [&](int k) {
      int c = 0;
#pragma unroll
      for (int i = 0; i < 5; ++i) c += (int)(((int64_t)k * T[i]) >> 3);
      return c;
    };
    int k = (int)(((int64_t)tile * 8) / nwg);
    k = k > 7 ? 7 : k;
    while (k < 7 && cum(k + 1) <= tile) ++k;
    while (k > 0 && cum(k) > tile) --k;
    int r = tile - cum(k);
    bool found = false;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int s0 = (int)(((int64_t)k * T[i]) >> 3), sz = (int)(((int64_t)(k + 1) * T[i]) >> 3) - s0;
      if (!found && r < sz) {
        gi = i;
        lt = s0 + r;
        found = true;
      }
      if (!found) r -= sz;
    }
  }
}

template <typename TIN, typename TOUT, int EPI, int NT, int S, int BM = 128>
__global__ __launch_bounds__(BM * 2, 2) void linear_d8_ring_kernel(GemmArgs args) {
  constexpr int NW = BM / 32;                 // waves: each owns 32 rows x BN columns
  constexpr int EPC = Elem<TIN>::EPC;
  constexpr int BKE = 8 * EPC;
  constexpr int MT = 2;
  constexpr int BN = 16 * NT;
  constexpr int STAGE = (BM + BN) * 128;
  constexpr int WI = BN / 8;                  // W DMA instructions per tile (8 rows each)
  constexpr int WQ = (WI + NW - 1) / NW;      // per wave, at most
  typedef typename Elem<TIN>::frag frag;
  extern __shared__ __attribute__((aligned(16))) char lds[];  // S stages x (128 + BN) rows x 128 B

  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
  const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  // `tile` walks the launch XCD by XCD.  Every group's items are spread evenly over the eight XCDs, the E irrep's
  // first: an E item takes 1.6x the time of a one-dimensional one, and with the groups laid out one after the other
  // XCDs 0-3 ran only E items and XCDs 4-7 only short ones (HW_ID timeline, tools/ring_trace.py: per-CU spans of
  // 95 k .. 200 k cycles for one launch).  Segment k of the order = [share k of group 0 | share k of group 1 | ...].
  (void)tile;                                 // (the timeline build indexes its stamps with it)
  int gi, lt;
  ring_item_of(args, nwg, bid, gi, lt);
  const GemmGroup& G = args.g[gi];
  // work item = (m-tile, chunk of consecutive n-tiles); the DMA ring runs continuously over its (n-tile, k-tile) steps
  const int mt = lt / G.n_chunks, nc = lt - mt * G.n_chunks;
  const int64_t m0 = (int64_t)mt * BM;
  const int nt_begin = nc * G.chunk;
  const int nt_count = (G.n_tiles - nt_begin) < G.chunk ? (G.n_tiles - nt_begin) : G.chunk;
  const int K = G.K, N = G.N;
  const int nkt = (K + BKE - 1) / BKE;
  const int steps = nt_count * nkt;
  const bool last_half_only = (K - (nkt - 1) * BKE) <= 4 * EPC;

  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int fr = lane & 15, kg = lane >> 4;
  RTRACE(0);
#ifdef OCTIC_RING_TRACE
  {
    unsigned xcc, hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    if (tile < 2048 && lane == 0 && wid < 4) g_ring_trace[(tile * 4 + wid) * 64 + 63] = ((unsigned long long)hwid << 32) | xcc;
  }
#endif

  // ---- DMA sources.  A wave-instruction fills 8 LDS rows: lane -> row (lane>>3), chunk position (lane&7), which
  // must hold source chunk (lane&7) ^ (row&7) = (lane&7) ^ (lane>>3).
  const int drow = lane >> 3;
  const int dkc = (lane & 7) ^ drow;
  const char* xsrc[4];   // this wave's 32 X rows: instruction q covers rows 32*wid + 8q .. +7
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    int64_t mm = m0 + wid * 32 + q * 8 + drow;
    mm = mm < G.rows ? mm : G.rows - 1;
    const int64_t off = G.pair ? (mm >> 1) * G.a_ld + (mm & 1) * (int64_t)K : mm * G.a_ld;
    xsrc[q] = G.a + off * (int64_t)sizeof(TIN);
  }
  // W rows: WI instructions over 4 waves (10: waves 0,1 take 3, waves 2,3 take 2; 20: 5 each)
  const int w_rem = WI % NW;
  const int w_cnt = WI / NW + (wid < w_rem ? 1 : 0);
  const int w_first = wid * (WI / NW) + (wid < w_rem ? wid : w_rem);
  const int dma_cnt = 4 + w_cnt;              // this wave's DMA instructions per step
  const char* wsrc[WQ];
  auto set_w = [&](int nt) {
#pragma unroll
    for (int q = 0; q < WQ; ++q) {
      int n = nt * BN + (w_first + q) * 8 + drow;
      n = n < N ? n : N - 1;
      wsrc[q] = G.w + (int64_t)n * K * (int64_t)sizeof(TIN);
    }
  };
  int l_nt = nt_begin, l_kt = 0, l_stage = 0;   // DMA stream position
  set_w(l_nt);
  auto issue = [&]() {
    int k = l_kt * BKE + dkc * EPC;
    k = k < K ? k : 0;                        // chunks past K belong to a skipped k-step: any valid address
    const int kb = k * (int)sizeof(TIN);
    char* st = lds + l_stage * STAGE;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xsrc[q] + kb),
                                       (__attribute__((address_space(3))) void*)(st + (wid * 4 + q) * 1024), 16, 0, 0);
#pragma unroll
    for (int q = 0; q < WQ; ++q)
      if (q < w_cnt)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc[q] + kb),
                                         (__attribute__((address_space(3))) void*)(st + BM * 128 + (w_first + q) * 1024),
                                         16, 0, 0);
    l_stage = l_stage == S - 1 ? 0 : l_stage + 1;
    if (++l_kt == nkt) {
      l_kt = 0;
      ++l_nt;
      if (l_nt < nt_begin + nt_count) set_w(l_nt);
    }
  };

  f32x4 acc[NT][MT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
  const int sw = fr & 7;
  const int rd_w = BM * 128 + fr * 128;                // + i*2048
  const int rd_x = (wid * 32 + fr) * 128;               // + j*2048
  const int ch0 = (kg ^ sw) << 4, ch1 = ((4 + kg) ^ sw) << 4;

  // ---- epilogue constants: drop-path scale of this lane's two MFMA rows
  float rsv[MT];
#pragma unroll
  for (int j = 0; j < MT; ++j) {
    const int64_t mm = m0 + wid * 32 + j * 16 + fr;
    const int64_t mc = mm < G.rows ? mm : 0;
    const int64_t token = G.pair ? (mc >> 1) : mc;
    rsv[j] = (EPI == 1 && args.rs) ? args.rs[token / args.rps] : 1.0f;
  }
  const bool has_bias = G.bias != nullptr;
  const bool has_cs = EPI == 1 && G.cs != nullptr, has_rs = EPI == 1 && args.rs != nullptr,
             has_res = EPI == 1 && G.resid != nullptr;
  // ---- epilogue (once per item: a workgroup owns ONE output tile).  Storing straight from the MFMA layout - a lane holds 4
  // consecutive channels of one token, so a wave-instruction touches 16 rows x 64 B (f32) or 32 B (bf16) - is bound by the
  // address path (tools/ring_trace.py: 7.8 k cycles per item, a fifth of a short item).  The tile goes through the idle
  // ring instead: per pass a wave parks 16 rows x BN f32 (bias / column scale / row scale applied) in its own LDS strip
  // and walks them row-wise, 16 contiguous bytes per lane for the residual load and the store.  Same arithmetic in the same
  // order as the direct form: results are bit-identical.
  auto epilogue = [&]() {
    constexpr int RS = BN * 4 + 16;                   // strip row stride: the 16 B skew keeps ds_write_b128 conflict-free
    constexpr int OPC = 16 / (int)sizeof(TOUT);       // output columns per 16-byte piece
    constexpr int CPR = BN / OPC;                     // pieces per row
    constexpr int PIECES = 16 * CPR;                  // per wave and pass
    static_assert(NW * 16 * RS + 2 * BN * 4 <= S * STAGE, "the strips and the column vectors must fit the ring");
    static_assert(BN / 4 <= BM * 2, "one thread per four columns");
    const int nb = nt_begin * BN;
    // bias and column scale of the tile's BN columns go through LDS as well (per-n-tile global loads in the MFMA layout
    // compile to twenty serial round trips)
    f32x4 bias_r = {0, 0, 0, 0}, cs_r = {1, 1, 1, 1};
    const int cvi = threadIdx.x;                      // this thread's four columns
    if (cvi < BN / 4 && nb + 4 * cvi < N) {
      if (has_bias) bias_r = *(const f32x4*)(G.bias + nb + 4 * cvi);
      if (has_cs) cs_r = *(const f32x4*)(G.cs + nb + 4 * cvi);
    }
    __syncthreads();                                  // every wave is done with the ring
    char* strip = lds + wid * (16 * RS);
    const char* colv = lds + NW * 16 * RS;            // [bias BN f32 | column scale BN f32]
    if (cvi < BN / 4) {
      *(f32x4*)(lds + NW * 16 * RS + cvi * 16) = bias_r;
      *(f32x4*)(lds + NW * 16 * RS + BN * 4 + cvi * 16) = cs_r;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      // (compile-time variants: a run-time `if (has_bias)` per n-tile puts every LDS read in its own block behind a wait)
      auto park = [&](auto hb_c, auto hc_c) {
#pragma unroll
        for (int i = 0; i < NT; ++i) {
          f32x4 v = acc[i][j];
          if constexpr (decltype(hb_c)::value) v += *(const f32x4*)(colv + (i * 16 + kg * 4) * 4);
          if constexpr (decltype(hc_c)::value) v *= *(const f32x4*)(colv + BN * 4 + (i * 16 + kg * 4) * 4);
          if constexpr (EPI == 1) v *= rsv[j];          // 1.0f without a row scale: exact
          *(f32x4*)(strip + fr * RS + (i * 16 + kg * 4) * 4) = v;
        }
      };
      if (has_bias && has_cs) park(std::true_type(), std::true_type());
      else if (has_bias) park(std::true_type(), std::false_type());
      else if (has_cs) park(std::false_type(), std::true_type());
      else park(std::false_type(), std::false_type());
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      // row pass: every residual load of the pass is requested (from clamped, always valid addresses) before the first
      // use; the stores are predicated
      constexpr int NIT = (PIECES + 63) / 64;
      int64_t yo[NIT];
      bool okp[NIT];
      typedef typename std::conditional<sizeof(TOUT) == 4, f32x4, bf16x8>::type piece_t;
      piece_t rv[NIT];
      const int pshift = G.pair ? 1 : 0;               // pair groups: GEMM row mm = token mm >> 1, half mm & 1
      auto request = [&](auto lift_c) {                // (the lift addressing divides: its own straight-line variant)
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
          const int p = it * 64 + lane;
          const int r = p / CPR, c = p - r * CPR;
          const int64_t mm = m0 + wid * 32 + j * 16 + r;
          const int n = nb + c * OPC;
          okp[it] = p < PIECES && mm < G.rows && n < N;
          const int64_t mc = okp[it] ? mm : 0;
          const int ncl = okp[it] ? n : 0;
          int64_t roff;
          if constexpr (decltype(lift_c)::value) {
            const unsigned b = (unsigned)mc / (unsigned)args.lift_np, q = (unsigned)mc - b * (unsigned)args.lift_np;   // rows < 2^31
            yo[it] = ((int64_t)b * (args.lift_np + args.lift_tok0) + args.lift_tok0 + q) * G.y_ld + ncl;
            roff = (int64_t)q * G.r_ld + ncl;
          } else {
            const int64_t tok = mc >> pshift, half = mc & pshift;
            yo[it] = tok * G.y_ld + half * N + ncl;
            roff = tok * G.r_ld + half * N + ncl;
          }
          if constexpr (EPI == 1) rv[it] = *(const piece_t*)((const TOUT*)(has_res ? G.resid : G.y) + (has_res ? roff : yo[it]));
        }
      };
      if (args.lift_np > 0) request(std::true_type());
      else request(std::false_type());
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int p = it * 64 + lane;
        const int r = p / CPR, c = p - r * CPR;
        const char* src = strip + (p < PIECES ? r * RS + c * (OPC * 4) : 0);
        if constexpr (sizeof(TOUT) == 4) {
          f32x4 v = *(const f32x4*)src;
          if constexpr (EPI == 1) {
            if (has_res) v += rv[it];
          }
          if (okp[it]) *(f32x4*)((float*)G.y + yo[it]) = v;
        } else {
          f32x4 v0 = *(const f32x4*)src, v1 = *(const f32x4*)(src + 16);
          if constexpr (EPI == 1) {
            if (has_res) {
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                v0[e] += (float)rv[it][e];
                v1[e] += (float)rv[it][4 + e];
              }
            }
          }
          const bf16x8 o = {(bf16)v0[0], (bf16)v0[1], (bf16)v0[2], (bf16)v0[3], (bf16)v1[0], (bf16)v1[1], (bf16)v1[2], (bf16)v1[3]};
          if (okp[it]) *(bf16x8*)((bf16*)G.y + yo[it]) = o;
        }
      }
      __builtin_amdgcn_wave_barrier();                // the strip is rewritten by the next pass
    }
  };

  // ---- ring.  VMEM program order per step s:  [wait tile s][barrier] DMA(s+S-1)  compute(s)
  issue();
  if (S > 2 && steps > 1) issue();
  RTRACE(1);
  int c_kt = 0, c_stage = 0;
  for (int s = 0; s < steps; ++s) {
    // S = 3: DMA runs two tiles ahead (tile s+1 may still be in flight at this wait: dma_cnt = 6 | 7 younger
    // instructions; a fixed immediate, two scalar compares - a `switch` over all counts compiles to a ~500-cycle
    // compare ladder); S = 2: one tile ahead, nothing is younger than DMA(s)
    if (S > 2 && s + 1 < steps && dma_cnt >= 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    RTRACE(2 + 3 * s);
    __builtin_amdgcn_s_barrier();             // every wave's share of tile s has landed; the stage of tile s-1 is free
    RTRACE(3 + 3 * s);
    if (s + S - 1 < steps) issue();
    const char* base = lds + c_stage * STAGE;
    c_stage = c_stage == S - 1 ? 0 : c_stage + 1;
    const bool half_tile = last_half_only && c_kt == nkt - 1;
    if constexpr (sizeof(TIN) == 2) {
      // bf16: every fragment of the k-tile is requested before the first MFMA (one exposed LDS latency per step).  Left
      // to itself hipcc emits  2 x ds_read, s_waitcnt lgkmcnt(0), 4 x MFMA  ten times per step: ten exposed latencies,
      // ~2000 cycles for 640 cycles of MFMA (tools/ring_trace.py).
      frag af[2][NT], bfr[2][MT];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int ch = ks ? ch1 : ch0;
#pragma unroll
        for (int j = 0; j < MT; ++j) bfr[ks][j] = *(const frag*)(base + rd_x + j * 2048 + ch);
#pragma unroll
        for (int i = 0; i < NT; ++i) af[ks][i] = *(const frag*)(base + rd_w + i * 2048 + ch);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        if (ks == 1 && half_tile) break;       // the second half holds chunks past K: loaded (valid LDS), never multiplied
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
          for (int j = 0; j < MT; ++j) acc[i][j] = mfma_step(af[ks][i], bfr[ks][j], acc[i][j]);
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        if (ks == 1 && half_tile) break;
        const int ch = ks ? ch1 : ch0;
        frag af[NT], bfr[MT];
#pragma unroll
        for (int i = 0; i < NT; ++i) af[i] = *(const frag*)(base + rd_w + i * 2048 + ch);
#pragma unroll
        for (int j = 0; j < MT; ++j) bfr[j] = *(const frag*)(base + rd_x + j * 2048 + ch);
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
          for (int j = 0; j < MT; ++j) acc[i][j] = mfma_step(af[i], bfr[j], acc[i][j]);
      }
    }
    ++c_kt;
    RTRACE(4 + 3 * s);
  }
  epilogue();
  RTRACE(62);
}

// ---- dispatch plan of a ring launch.  One long group (E: twice the K steps) + equal short groups, two workgroups per CU:
// at ViT-H that is 514 long + 516 short items on 512 slots - six items more than two rounds.  Workgroups are dispatched in
// blockIdx order, round-robin over the XCDs, to whichever slot of the XCD frees first; with "long items first" the six
// leftovers land as THIRD items behind a long and a short one (tools/ring_trace.py: six CUs end at 150 k cycles, the other
// 250 at 101 k - the launch is half again as long as its mean).  The plan gives every XCD its own list
// [a short items | e long items | short items]: an XCD with an odd item starts one slot on a short item, which then has
// time for three short ones; XCDs without odd items take the long items the others gave up (a slot with two long items
// ends about when a three-short slot does).  e[x] and a[x] come from a small DP over simulated in-order dispatch; the
// plan depends on the launch shape only and is cached.
struct RingPlan {
  int key[6];
  short a[8], e[8];
};
inline int sim_xcd(int n, int a, int e, int slots, int c_short, int c_long) {
  // makespan of the list [a short | e long | n - a - e short] dispatched in order to `slots` slots
  int t[256];
  slots = slots > 256 ? 256 : slots;
  for (int i = 0; i < slots; ++i) t[i] = 0;
  int mk = 0;
  for (int i = 0; i < n; ++i) {
    int best = 0;
    for (int j = 1; j < slots; ++j)
      if (t[j] < t[best]) best = j;
    t[best] += (i >= a && i < a + e) ? c_long : c_short;
    mk = t[best] > mk ? t[best] : mk;
  }
  return mk;
}
// routing override OCTIC_ROUTE_RING_EVEN: 1 = the even spread of round 3
inline bool plan_ring(GemmArgs& a, int nwg, int slots_per_xcd) {
  a.plan_mode = 0;
  if (route(OCTIC_ROUTE_RING_EVEN) || a.ngroups < 2 || nwg < 16) return false;
  if (nwg > 16 * 8 * slots_per_xcd) return false;   // many rounds: a leftover item is noise, and the DP below is O(items^2)
  const int bke = 64;   // (cost model in K steps of the bf16 kernel; only ratios matter)
  auto items = [&](int g) { return a.g[g].n_chunks * a.g[g].m_tiles; };
  auto cost = [&](int g) { return (a.g[g].K + bke - 1) / bke + 2; };       // + prologue and epilogue, about two steps
  const int c_long = cost(0), c_short = cost(1), n_short1 = items(1);
  if (c_long <= c_short) return false;
  for (int g = 2; g < a.ngroups; ++g)
    if (cost(g) != c_short || items(g) != n_short1) return false;
  const int E = items(0), nshort = a.ngroups - 1;
  // (the key space is small for a fixed model, but a run with varying batch sizes visits dozens of launch shapes: a plan
  // costs milliseconds of host time, so none is ever computed twice)
  static std::vector<RingPlan> cache;
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  const int key[6] = {nwg, slots_per_xcd, E, c_long, c_short, nshort};
  const RingPlan* hit = nullptr;
  for (size_t i = 0; i < cache.size() && !hit; ++i) {
    bool same = true;
    for (int j = 0; j < 6; ++j) same = same && cache[i].key[j] == key[j];
    if (same) hit = &cache[i];
  }
  if (!hit) {
    const int q8 = nwg >> 3, r8 = nwg & 7;
    constexpr int AMAX = 4;
    // mk[type][e][a]: type 0 = XCDs with q8 + 1 workgroups, 1 = q8
    const int nmax = q8 + 2;
    int* mk = (int*)malloc(sizeof(int) * 2 * nmax * AMAX);
    for (int ty = 0; ty < 2; ++ty)
      for (int e = 0; e < nmax; ++e)
        for (int aa = 0; aa < AMAX; ++aa) {
          const int n = q8 + (ty == 0 ? 1 : 0);
          mk[(ty * nmax + e) * AMAX + aa] = (e + aa <= n) ? sim_xcd(n, aa, e, slots_per_xcd, c_short, c_long) : (1 << 30);
        }
    // f[x][u]: best (max makespan, then sum) for XCDs 0..x-1 using u long items
    // (ties: the most even spread of the long items, then the smallest sum of the XCD makespans)
    struct Cell { int mx, maxe; long long sum; short e, a; };
    Cell* f = (Cell*)malloc(sizeof(Cell) * 9 * (E + 1));
    for (int u = 0; u <= E; ++u) f[u] = Cell{u == 0 ? 0 : (1 << 30), 0, 0, 0, 0};
    for (int x = 0; x < 8; ++x) {
      const int ty = x < r8 ? 0 : 1, n = q8 + (ty == 0 ? 1 : 0);
      for (int u = 0; u <= E; ++u) {
        Cell best{1 << 30, 0, 0, 0, 0};
        for (int e = 0; e <= n && e <= u; ++e) {
          const Cell& prev = f[x * (E + 1) + (u - e)];
          if (prev.mx >= (1 << 30)) continue;
          for (int aa = 0; aa < AMAX; ++aa) {
            const int m = mk[(ty * nmax + e) * AMAX + aa];
            if (m >= (1 << 30)) continue;
            const int mx = m > prev.mx ? m : prev.mx, maxe = e > prev.maxe ? e : prev.maxe;
            const long long sum = prev.sum + m;
            if (mx < best.mx || (mx == best.mx && (maxe < best.maxe || (maxe == best.maxe && sum < best.sum))))
              best = Cell{mx, maxe, sum, (short)e, (short)aa};
          }
        }
        f[(x + 1) * (E + 1) + u] = best;
      }
    }
    RingPlan pl;
    for (int j = 0; j < 6; ++j) pl.key[j] = key[j];
    bool ok = f[8 * (E + 1) + E].mx < (1 << 30);
    int u = E;
    for (int x = 7; x >= 0 && ok; --x) {
      const Cell& c = f[(x + 1) * (E + 1) + u];
      pl.e[x] = c.e;
      pl.a[x] = c.a;
      u -= c.e;
    }
    free(f);
    free(mk);
    if (!ok) return false;
    cache.push_back(pl);
    hit = &cache.back();
  }
  const int q8 = nwg >> 3, r8 = nwg & 7;
  int eb = 0, sb = 0;
  for (int x = 0; x < 8; ++x) {
    const int n = q8 + (x < r8 ? 1 : 0);
    a.plan_a[x] = hit->a[x];
    a.plan_e[x] = hit->e[x];
    a.plan_eb[x] = eb;
    a.plan_sb[x] = sb;
    eb += hit->e[x];
    sb += n - hit->e[x];
  }
  a.plan_nshort = nshort;
  a.plan_mode = 1;
  return true;
}

template <typename TIN, typename TOUT, int NT, int S, int BM = 128>
int launch_ring_nt(GemmArgs& a, hipStream_t s) {
  constexpr int BN = 16 * NT;
  int t = 0;
  bool fused = a.rs != nullptr || a.lift_np > 0;
  for (int i = 0; i < a.ngroups; ++i) {
    a.g[i].n_tiles = (a.g[i].N + BN - 1) / BN;
    a.g[i].m_tiles = (int)((a.g[i].rows + BM - 1) / BM);
    a.g[i].chunk = 1;                  // one output tile per workgroup (measured best in situ on MI355X; the staged epilogue
    a.g[i].n_chunks = a.g[i].n_tiles;  // of the kernel relies on it: the ring is idle when the tile is complete)
    a.g[i].tile_begin = t;
    t += a.g[i].n_chunks * a.g[i].m_tiles;
    fused = fused || a.g[i].cs || a.g[i].resid;
  }
  a.total_tiles = t;
  const size_t smem = (size_t)S * (BM + BN) * 128;
  {
    const int per_cu = (int)(160 * 1024 / smem) < 2 ? (int)(160 * 1024 / smem) : 2;   // __launch_bounds__(BM * 2, 2)
    plan_ring(a, t, device_cus() / 8 * (per_cu > 0 ? per_cu : 1));
  }
  static DeviceOnce once;
  if (once.first()) {
    (void)hipFuncSetAttribute((const void*)linear_d8_ring_kernel<TIN, TOUT, 0, NT, S, BM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    (void)hipFuncSetAttribute((const void*)linear_d8_ring_kernel<TIN, TOUT, 1, NT, S, BM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    (void)hipGetLastError();
  }
  if (fused) linear_d8_ring_kernel<TIN, TOUT, 1, NT, S, BM><<<t, BM * 2, smem, s>>>(a);
  else linear_d8_ring_kernel<TIN, TOUT, 0, NT, S, BM><<<t, BM * 2, smem, s>>>(a);
  return launch_status();
}

template <typename TIN, typename TOUT>
int launch_ring(GemmArgs& a, hipStream_t s) {
  // wide tile when every group's N is a multiple of 160 and the problem is long in K (the X panel dominates)
  bool ok = sizeof(TIN) == 2 && a.lift_np == 0;
  for (int i = 0; i < a.ngroups; ++i) ok = ok && (a.g[i].N % 160) == 0 && a.g[i].K >= 2 * a.g[i].N;
  if (ok) return launch_ring_nt<TIN, TOUT, 10, 2>(a, s);
  return launch_ring_nt<TIN, TOUT, 5, 3>(a, s);
}

inline bool ring_ok(const GemmArgs& a, int dtype) {
  const int kstep = dtype == OCTIC_BF16 ? 32 : 16;
  for (int i = 0; i < a.ngroups; ++i)
    if (a.g[i].K % kstep) return false;
  return true;
}

inline int pick_nt(const GemmArgs& a) {
  // minimise padded work; prefer the wider tile on ties (fewer re-reads of the token panel)
  int best = 2;
  double best_cost = 1e30;
  for (int nt = 2; nt <= 5; ++nt) {
    const int bn = 32 * nt;
    double cost = 0;
    for (int i = 0; i < a.ngroups; ++i) {
      const double tiles = (a.g[i].N + bn - 1) / bn;
      cost += tiles * bn * (double)a.g[i].K * (double)a.g[i].rows;
    }
    if (cost <= best_cost * 1.0001) {
      best_cost = cost < best_cost ? cost : best_cost;
      best = nt;
    }
  }
  return best;
}

template <typename TIN, typename TOUT>
int launch_gemm(GemmArgs& a, hipStream_t s) {
  const int nt = pick_nt(a);
  const int bn = 32 * nt;
  int t = 0;
  for (int i = 0; i < a.ngroups; ++i) {
    a.g[i].n_tiles = (a.g[i].N + bn - 1) / bn;
    a.g[i].m_tiles = (int)((a.g[i].rows + kBM - 1) / kBM);
    // ~16 pipeline steps per block: long enough to amortise pipeline fill/drain, short enough to balance
    const int bke = 128 / (int)sizeof(TIN);
    const int nkt = (a.g[i].K + bke - 1) / bke;
    // pipeline steps per workgroup; 1 = one output tile per workgroup (measured best in situ on MI355X)
    constexpr int target_steps = 1;
    int chunk = target_steps / nkt;
    chunk = chunk < 1 ? 1 : (chunk > a.g[i].n_tiles ? a.g[i].n_tiles : chunk);
    a.g[i].chunk = chunk;
    a.g[i].n_chunks = (a.g[i].n_tiles + chunk - 1) / chunk;
    a.g[i].tile_begin = t;
    t += a.g[i].n_chunks * a.g[i].m_tiles;
  }
  a.total_tiles = t;
  const size_t smem = (size_t)2 * (kBM + bn) * 128;
  // gfx950 has 160 KiB of LDS per CU; anything above the 64 KiB default must be opted into once.
  static DeviceOnce once;
  if (once.first()) {
    hipFuncSetAttribute((const void*)linear_d8_kernel<TIN, TOUT, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (kBM + 64) * 128);
    hipFuncSetAttribute((const void*)linear_d8_kernel<TIN, TOUT, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (kBM + 96) * 128);
    hipFuncSetAttribute((const void*)linear_d8_kernel<TIN, TOUT, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (kBM + 128) * 128);
    hipFuncSetAttribute((const void*)linear_d8_kernel<TIN, TOUT, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (kBM + 160) * 128);
    (void)hipGetLastError();
  }
  switch (nt) {
    case 2: linear_d8_kernel<TIN, TOUT, 2><<<t, 256, smem, s>>>(a); break;
    case 3: linear_d8_kernel<TIN, TOUT, 3><<<t, 256, smem, s>>>(a); break;
    case 4: linear_d8_kernel<TIN, TOUT, 4><<<t, 256, smem, s>>>(a); break;
    default: linear_d8_kernel<TIN, TOUT, 5><<<t, 256, smem, s>>>(a); break;
  }
  return launch_status();
}

inline int dispatch_gemm(GemmArgs& a, int dtype, int out_dtype, hipStream_t s) {
  if (dtype == OCTIC_BF16) {
    const int rw = launch_wreg(a, out_dtype, s);   // short-K problems: W-stationary streaming kernel (gemm_wreg.hip)
    if (rw != -100) return rw;
  }
  if (ring_ok(a, dtype)) {
    if (dtype == OCTIC_F32 && out_dtype == OCTIC_F32) return launch_ring<float, float>(a, s);
    if (dtype == OCTIC_BF16 && out_dtype == OCTIC_BF16) return launch_ring<bf16, bf16>(a, s);
    if (dtype == OCTIC_BF16 && out_dtype == OCTIC_F32) return launch_ring<bf16, float>(a, s);
    return OCTIC_EDTYPE;
  }
  if (dtype == OCTIC_F32 && out_dtype == OCTIC_F32) return launch_gemm<float, float>(a, s);
  if (dtype == OCTIC_BF16 && out_dtype == OCTIC_BF16) return launch_gemm<bf16, bf16>(a, s);
  if (dtype == OCTIC_BF16 && out_dtype == OCTIC_F32) return launch_gemm<bf16, float>(a, s);
  return OCTIC_EDTYPE;
}

}  // namespace octic

using namespace octic;

// Host-only query (no device call): the item order of a ring launch.  items[g] / ksteps[g] describe
// group g (group 0 = the long one); out_group / out_item receive, per workgroup in blockIdx order, the item it would run.
// Returns the plan mode (1 = planned order, 0 = even spread) or a negative error.
extern "C" int octic_linear_d8_ring_order(int ngroups, const int* items, const int* ksteps, int slots_per_xcd, int* out_group,
                                    int* out_item) {
  if (ngroups < 1 || ngroups > 5 || !items || !ksteps || !out_group || !out_item) return OCTIC_ENULL;
  GemmArgs a = {};
  a.ngroups = ngroups;
  int t = 0;
  for (int g = 0; g < ngroups; ++g) {
    a.g[g].K = ksteps[g] * 64;
    a.g[g].n_chunks = 1;
    a.g[g].chunk = 1;
    a.g[g].n_tiles = 1;
    a.g[g].m_tiles = items[g];
    a.g[g].tile_begin = t;
    t += items[g];
  }
  a.total_tiles = t;
  plan_ring(a, t, slots_per_xcd);
  for (int bid = 0; bid < t; ++bid) ring_item_of(a, t, bid, out_group[bid], out_item[bid]);
  return a.plan_mode;
}

extern "C" {

int octic_linear_d8_fwd(const octic_view* x, const void* const w[5], const float* bias, const octic_view* y,
                        const octic_view* resid, const float* rs, int64_t rows_per_sample, const float* const cs[5],
                        int64_t M, int cin, int cout, int dtype, int out_dtype, void* stream) {
  int e;
  if ((e = check_c_dt(cin, dtype)) || (e = check_c_dt(cout, dtype)) || (e = check_c_dt(cout, out_dtype))) return e;
  if ((e = check_view(x, cin, dtype)) || (e = check_view(y, cout, out_dtype))) return e;
  if (resid && (e = check_view(resid, cout, out_dtype))) return e;
  if (!w) return OCTIC_ENULL;
  for (int i = 0; i < 5; ++i)
    if (!w[i] || (((uintptr_t)w[i]) & 15)) return w[i] ? OCTIC_EALIGN : OCTIC_ENULL;
  if (M <= 0 || (rs && rows_per_sample <= 0)) return OCTIC_ESHAPE;
  if (cs)
    for (int i = 0; i < 5; ++i)
      if (!cs[i]) return OCTIC_ENULL;
  GemmArgs a = {};
  a.ngroups = 5;
  a.rs = rs;
  a.rps = rs ? rows_per_sample : 1;
  a.lift_np = 0;
  a.lift_tok0 = 0;
  // group order: E first (its tiles carry 2x the K work), then the four one-dimensional irreps
  for (int gidx = 0; gidx < 5; ++gidx) {
    const int irrep = gidx == 0 ? 4 : gidx - 1;
    GemmGroup& g = a.g[gidx];
    const bool isE = irrep == 4;
    g.a = (const char*)x->ptr[irrep];
    g.a_ld = x->ld[irrep];
    g.w = (const char*)w[irrep];
    g.y = (char*)y->ptr[irrep];
    g.y_ld = y->ld[irrep];
    g.resid = resid ? (const char*)resid->ptr[irrep] : nullptr;
    g.r_ld = resid ? resid->ld[irrep] : 0;
    g.bias = (irrep == 0) ? bias : nullptr;
    g.cs = cs ? cs[irrep] : nullptr;
    g.rows = isE ? 2 * M : M;
    g.K = isE ? 2 * cin : cin;
    g.N = isE ? 2 * cout : cout;
    g.pair = isE ? 1 : 0;
  }
  if (cs)
    for (int i = 0; i < 5; ++i)
      if (!cs[i]) return OCTIC_ENULL;
  return dispatch_gemm(a, dtype, out_dtype, (hipStream_t)stream);
}

int octic_linear_d8_tile_n(int64_t M, int cin, int cout) {
  GemmArgs a = {};
  a.ngroups = 5;
  for (int i = 0; i < 5; ++i) {
    a.g[i].rows = i == 0 ? 2 * M : M;
    a.g[i].K = i == 0 ? 2 * cin : cin;
    a.g[i].N = i == 0 ? 2 * cout : cout;
  }
  return 32 * pick_nt(a);
}

int octic_lift_gemm(const void* patches, const void* w, const float* bias, const float* pos, float* out, int64_t B,
                    int64_t n_patches, int tok0, int Kpad, int D, int dtype, void* stream) {
  if (!patches || !w || !out) return OCTIC_ENULL;
  if (B <= 0 || n_patches <= 0 || tok0 < 0 || Kpad <= 0 || (Kpad % 8) || D <= 0 || (D % 8)) return OCTIC_ESHAPE;
  if ((((uintptr_t)patches) | ((uintptr_t)w) | ((uintptr_t)out)) & 15) return OCTIC_EALIGN;
  GemmArgs a = {};
  a.ngroups = 1;
  a.rs = nullptr;
  a.rps = 1;
  a.lift_np = n_patches;
  a.lift_tok0 = tok0;
  GemmGroup& g = a.g[0];
  g.a = (const char*)patches;
  g.a_ld = Kpad;
  g.w = (const char*)w;
  g.y = (char*)out;
  g.y_ld = D;
  g.resid = (const char*)pos;
  g.r_ld = D;
  g.bias = bias;  // length D (zero outside the A1 block)
  g.cs = nullptr;
  g.rows = B * n_patches;
  g.K = Kpad;
  g.N = D;
  g.pair = 0;
  return dispatch_gemm(a, dtype, OCTIC_F32, (hipStream_t)stream);
}

}  // extern "C"
