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
// (A PERSISTENT variant - one workgroup per CU walking items, the next item's DMA prologue issued before the current
// epilogue, the epilogue staged 16 rows at a time behind the ring - was built in round 3: per item the prologue + launch
// gap + store acknowledgement went 2.3 -> 0.7 us and the 16-row epilogue cost 1.1 us more; 142 vs 147 us in isolation on
// qkv, but 0.2-0.7 ms per step SLOWER in the train step with either tail order, so it is not kept.)
// Scheduling: tiles are dealt to XCDs in contiguous chunks walked in groups of 8 row-panels (operand panels stay in the
// XCD's L2).  M = 16 448 gives 65 x {5,15,20} tiles on 256 CUs: the last partial round would idle most of the chip, so
// the tiles of that round are split along K over `split` workgroups each, placed at the FRONT of the grid (f32 partial
// slabs + a ticket; the last arriver of a tile reduces and runs the epilogue) where that pays - see dense_plan().
//
// Round 4: the tile WIDTH is a template parameter (NT = MFMA n-tiles per wave).  NT = 4 is the 256 x 256 tile above.
// NT = 5 is a 256 x 320 tile for the N = 1280 problems (proj, fc2, the input gradients of proj and fc1): 65 x 5 = 325
// tiles of 256 x 256 are 1.27 rounds on 256 CUs, 64 x 4 = 256 tiles of 256 x 320 are exactly ONE (plus the 64-row last
// panel, whose 4 tiles go through the split-K front of the grid with the MFMAs of empty row halves skipped).  A wave then
// owns 128 x 80 (160 accumulator registers, 242 VGPRs in all); the column sets of a wave are 3 + 2 n-tiles, so the four
// units of a K-tile are 16 / 24 / 16 / 16 KiB (2 / 3 / 2 / 2 DMA instructions per wave; the four units in flight behind
// every landed-wait are always one of each kind: vmcnt(9)), two K-tiles of 72 KiB fill the ring, and the epilogue stages
// the tile in two 64-row passes.
//
// Round 6: per-image row panels.  The token rows of ViT-H/14 come as B images x 257 tokens (1 class token + 256 patches), so
// M = 16 448 = 64 x 256 + 64: classic panels (rows 256 tm ..) leave a 64-row last panel whose tiles go through the split-K
// front of the grid and cost their CUs 17-22 us (tools/dense_phases.py) - on every one of the 80 N = 1280 launches of a step,
// whose 256 full tiles are otherwise exactly one round.  With `tokens` = 257 known (octic_dense_gemm_nt_tokens) panel tm is
// the 256 PATCH rows of image tm (rows 257 tm + 1 ..): N = 1280 is exactly 64 x 4 tiles, N = 5120 exactly five rounds, no
// panel is ragged, nothing is split - and the B class-token rows (row stride 257 lda) are a skinny [B, K] x [K, N] problem
// for dense_cls_kernel below (a few microseconds: one 16 x 16 MFMA tile per wave, K cut over the eight waves of a
// workgroup, the same epilogue math per element).  dense_plan_tokens() takes it where its launch model is shorter.
#include <cmath>
#include <type_traits>
#include "octic_common.hpp"

namespace octic {

constexpr int DG_BM = 256, DG_BK = 64;
constexpr int DG_UNIT = 128 * 128;            // bytes per 128-row unit
constexpr int DG_SLOTS = 8;                   // ring = two K-tiles of four units
constexpr int DG_D = 6;                 // prefetch distance in units: unit g+D is issued in R_g (D <= slots - 2)
static_assert(DG_D >= 4 && DG_D <= DG_SLOTS - 2, "prefetch distance");
// Per tile width (NT = n-tiles of 16 columns per wave; 4 waves along N):
//   column sets of a wave: NA = NT - 2 tiles first, 2 tiles second; unit kinds 0 / 3 = A row halves (16 KiB),
//   kind 1 = first column sets (NA x 8 KiB), kind 2 = second column sets (16 KiB)
template <int NT> struct DgGeom {
  static constexpr int NA = NT - 2;
  static constexpr int BN = 64 * NT;
  static constexpr int WCOLS = 16 * NT;                       // columns per wave
  static constexpr int U1 = NA * 8192;                        // bytes of the kind-1 unit
  static constexpr int KT = 3 * DG_UNIT + U1;                 // bytes of a K-tile in the ring (64 / 72 KiB)
  static constexpr int RING = 2 * KT;
  static constexpr int SRS = WCOLS * 2 + 16;                  // staged epilogue row stride (data + 16 B bank spread)
  static constexpr int EPI_PASSES = NT == 4 ? 1 : 2;          // the staged tile must fit the LDS: 8 x (128 / passes) rows
  static constexpr int EPI_ROWS = 128 / EPI_PASSES;
  static constexpr int EPI = 8 * EPI_ROWS * SRS;
  static constexpr int LDS = RING > EPI ? RING : EPI;
  static constexpr int INFLIGHT = 2 * (DG_D - 2) + (NA - 2);  // DMA instructions of the D - 2 youngest units (one of each kind when D = 6)
  static_assert(DG_D == 6 || NT == 4, "unequal units: the steady vmcnt assumes the 4 youngest units are one of each kind");
  __host__ __device__ static constexpr int unit_off(int kind) { return kind == 0 ? 0 : kind == 1 ? DG_UNIT : kind == 2 ? DG_UNIT + U1 : 2 * DG_UNIT + U1; }
  __host__ __device__ static constexpr int unit_instr(int kind) { return kind == 1 ? NA : 2; }
};

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
  float* colsum;      // [tiles_m * 2][N] partial column sums of the output (DGELU, optional; see the epilogue)
  // schedule
  int m_stride, m_base;   // first token row of row panel tm = tm * m_stride + m_base: (256, 0), or (tokens, 1) for per-image panels
  int tiles_m, tiles_n;
  int full_tiles;     // tiles computed by one workgroup each
  int split;          // K-split factor of the remaining tiles (>= 1)
  int tail_pad;       // workgroups in front of the full tiles: the split parts of the remaining tiles, padded to 8
  float* slabs;       // [(tiles - full_tiles) * split] x 256 x 256 f32 partials
  int* tickets;       // [(tiles - full_tiles)] arrival counters (zeroed by the host per launch)
};

enum DgMode { DG_PLAIN = 0, DG_GELU = 1, DG_RESID = 2, DG_DGELU = 3,
              DG_GELUF = 4,    // like GELU, but C = gelu'(pre-activation) in bf16 (the factor the backward multiplies by) instead of it
              DG_DFACT = 5,    // like DGELU with H = that stored factor: C = H * acc, no transcendental in the epilogue
              DG_GELUO = 6 };  // GELU only: C = gelu(pre-activation), nothing kept for a backward (inference passes)

__device__ inline void dg_wait_vmcnt(int n) {      // n even, wave-uniform; anything unexpected drains (always safe)
  switch (n) {
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
    case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

#ifdef DG_TRACE
// developer-only timeline (tools/dense_trace.py builds with -DDG_TRACE): s_memtime stamps of workgroup 0
__device__ unsigned long long g_dg_trace[8 * 256];
extern "C" void* octic_dbg_dense_trace(void) {
  void* p = nullptr;
  (void)hipGetSymbolAddress(&p, HIP_SYMBOL(g_dg_trace));
  return p;
}
#define DGT()                                                                                        \
  do {                                                                                               \
    if (blockIdx.x == 0 && lane == 0 && dgt_i < 256) g_dg_trace[wid * 256 + dgt_i++] = __builtin_readcyclecounter(); \
    if (blockIdx.x == 0 && lane == 0 && dgt_i == 200) g_dg_trace[wid * 256 + 255] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define DGT() do {} while (0)
#endif
#ifdef DG_TRACE2
// developer-only (tools/dense_phases.py builds with -DDG_TRACE2): per workgroup, wave 0 stamps s_memrealtime (100 MHz) at
// kernel start / first K-tile / end of the K loop / end of the epilogue, plus XCC_ID and HW_ID
__device__ unsigned long long g_dg_trace2[4096 * 8];
extern "C" void* octic_dbg_dense_trace2(void) {
  void* p = nullptr;
  (void)hipGetSymbolAddress(&p, HIP_SYMBOL(g_dg_trace2));
  return p;
}
#define DGT2(slot)                                                                                            \
  do {                                                                                                        \
    if (wid == 0 && lane == 0 && blockIdx.x < 4096) g_dg_trace2[blockIdx.x * 8 + (slot)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define DGT2(slot) do {} while (0)
#endif

// GELU for the fused epilogues.  The epilogue runs after the main loop with nothing to hide its VALU work behind (one
// workgroup per CU), and libm's erff costs ~40 instructions per element (~100 us for the 84 M elements of fc1).  erf by
// Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, one v_exp + one v_rcp + 6 fma) is far inside the bf16 output's 2^-9.
// (Round 4: the math of the GELU tail is 3.8 us of a tile's 8.3 us epilogue - round-4 ablation build - and
// VALU-throughput bound: a transcendental-free form, Phi(x) = 1/2 + x P12(0.08 x^2 - 1) on |x| <= 5, thirteen packed fmas and
// 3.9e-7 of error, was built and timed EQUAL (245 vs 246 us per launch): the quarter-rate v_exp / v_rcp overlap the packed
// fmas of the other wave, so the form with fewer full-rate instructions stays.)
__device__ inline float dg_erf(float x) {
  const float ax = fabsf(x);
  // v_rcp_f32 (1 ulp): __frcp_rn / 1.0f/x expand to the IEEE division sequence (div_scale x2, rcp, 4 fma, div_fmas,
  // div_fixup) - ten instructions per element in a VALU-bound epilogue
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float r = 1.0f - poly * __expf(-ax * ax);
  return copysignf(r, x);
}
__device__ inline float dg_gelu(float x) { return 0.5f * x * (1.0f + dg_erf(x * kSqrt1Over2)); }
__device__ inline float dg_gelu_grad(float x) {
  return 0.5f * (1.0f + dg_erf(x * kSqrt1Over2)) + x * kInvSqrt2Pi * __expf(-0.5f * x * x);
}
// gelu(x) and gelu'(x) from ONE erf evaluation: Phi(x) = (1 + erf(x / sqrt 2)) / 2, and the exp(-x^2 / 2) inside the erf
// approximation is the density term of the derivative (DG_GELUF: the fc1 epilogue that also leaves the backward's factor)
__device__ inline void dg_gelu_both(float x, float& g, float& d) {
  const float ax = fabsf(x) * kSqrt1Over2;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float e = __expf(-ax * ax);                     // exp(-x^2 / 2)
  const float phi = 0.5f * (1.0f + copysignf(1.0f - poly * e, x));
  g = x * phi;
  d = phi + x * kInvSqrt2Pi * e;
}

template <int N>
__device__ inline void dg_wait_imm() {
  if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if constexpr (N == 14) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
  else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int MODE, int NT>
__global__ __launch_bounds__(512, 1) void dense_nt_kernel(DgArgs a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];   // two K-tiles of four units; re-used as 8 staged epilogue tiles
  using GE = DgGeom<NT>;
  constexpr int NA = GE::NA, DG_BN = GE::BN;
  static_assert(NT == 4 || MODE == DG_PLAIN, "the fused tails exist for the 256-wide tile only");

  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wid >> 2, wc = wid & 3;
  const bool hi = wr != 0;              // waves 4-7 run one barrier interval behind waves 0-3 (see the loop)
  const int fr = lane & 15, kg = lane >> 4;
#ifdef DG_TRACE
  int dgt_i = 0;
  if (blockIdx.x == 0 && lane == 0) g_dg_trace[wid * 256 + 254] = __builtin_amdgcn_s_memrealtime();
#endif
  DGT();
  DGT2(0);
#ifdef DG_TRACE2
  if (wid == 0 && lane == 0 && blockIdx.x < 4096) {
    unsigned xcc, hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    g_dg_trace2[blockIdx.x * 8 + 7] = ((unsigned long long)hwid << 32) | xcc;
  }
#endif

  // ---- work item -> (tile, k-range)
  const int bid = blockIdx.x;
  int tile, kt_begin, kt_end, part = 0, rem_idx = -1;
  const int nkt_all = a.K / DG_BK;
  if (bid >= a.tail_pad) {
    // XCD-aware bijective remap over the full tiles (tail_pad is a multiple of 8: j & 7 is still the XCD of this workgroup)
    const int j = bid - a.tail_pad;
    const int nf = a.full_tiles;
    const int xcd = j & 7, q8 = nf >> 3, r8 = nf & 7;
    tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (j >> 3);
    kt_begin = 0;
    kt_end = nkt_all;
  } else {
    // The split parts of the last partial round go FIRST: the short items and their slab round trip (publish, last arriver
    // re-reads `split` partial tiles) then run beside everybody's full tiles instead of forming the end of the launch
    // (tools/dense_phases.py: N 5120 K 1280 203.5 -> 197.7 us; with K 5120 the reducers' CUs still end the launch).
    const int r = bid;
    if (r >= (a.tiles_m * a.tiles_n - a.full_tiles) * a.split) return;       // padding
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
  const int m0 = tm * a.m_stride + a.m_base, n0 = tn * DG_BN;
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
  const int am0 = m0, an0 = n0;
  const unsigned voA = (unsigned)(((int64_t)(am0 + (r0 >> 6) * 128 + (r0 & 63)) * a.lda + dch * 8) * 2);
  // B: a wave's DMA rows of the first column sets are unit rows 8 NA wid + 8 j + drow (j < NA), of the second sets
  // 16 wid + 8 j + drow; unit row blocks of 16 NA (32) rows belong to column wave wc = block index
  const int rb1 = 8 * NA * wid + drow;                              // kind 1: unit row of instruction 0
  const unsigned voB1 = (unsigned)(((int64_t)(an0 + (rb1 / (16 * NA)) * GE::WCOLS + (rb1 % (16 * NA))) * a.ldb + dch * 8) * 2);
  const unsigned voB2 = (unsigned)(((int64_t)(an0 + (r0 >> 5) * GE::WCOLS + 16 * NA + (r0 & 31)) * a.ldb + dch * 8) * 2);
  const int rowA8 = (int)(a.lda * 16), rowB8 = (int)(a.ldb * 16);   // +8 rows, bytes
  const int halfA = (int)(a.lda * 128);                             // +64 rows of A, bytes
  const int kbase = kt_begin * (DG_BK * 2);

  int u_issue = 0;                      // next unit to issue
  // KIND 0 / 3: first / second 64-row halves of A; KIND 1 / 2: first / second 32-column halves of B (compile-time)
  auto issue_unit = [&](auto kind_c) {
    constexpr int KIND = decltype(kind_c)::value;
    constexpr bool isA = KIND == 0 || KIND == 3;
    char* dst = lds + ((u_issue >> 2) & 1) * GE::KT + GE::unit_off(KIND) + wid * (GE::unit_instr(KIND) * 1024);
    const int so = kbase + (u_issue >> 2) * (DG_BK * 2) + (KIND == 3 ? halfA : 0);
    if constexpr (isA) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)dst, 16, voA, so, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(dst + 1024), 16, voA, so + rowA8, 0, 0);
    } else if constexpr (KIND == 1) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void*)dst, 16, voB1, so, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void*)(dst + 1024), 16, voB1, so + rowB8, 0, 0);
      if constexpr (NA == 3)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void*)(dst + 2048), 16, voB1, so + 2 * rowB8, 0, 0);
    } else {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void*)dst, 16, voB2, so, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void*)(dst + 1024), 16, voB2, so + rowB8, 0, 0);
    }
    ++u_issue;
  };
  // DMA instructions of this wave that belong to the `n` youngest issued units (kind-1 units count NA, the others 2)
  auto young_instr = [&](int n) {
    const int k1 = ((u_issue + 2) >> 2) - ((u_issue - n + 2) >> 2);   // units u in [u_issue - n, u_issue) with u % 4 == 1
    return 2 * n + (NA - 2) * k1;
  };
#define DG_IC(v) std::integral_constant<int, v>()

  // ---- fragment read offsets inside a unit
  const int sw = lane & 7;
  const int rdo0 = fr * 128 + ((kg ^ sw) << 4);
  const int rdo1 = fr * 128 + (((4 + kg) ^ sw) << 4);
  const int a_row0 = wr * 64;          // unit rows of this wave inside units 0 / 3

  bf16x8 Af[1][2][4];    // [one row half at a time][kstep][m-tile]
  bf16x8 Bf0[2][NA], Bf1[2][2];        // first / second column set: [kstep][n-tile]
  f32x4 acc0[2][4][NA], acc1[2][4][2]; // first / second column set: [m-half][m-tile][n-tile]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int p = 0; p < 4; ++p) {
#pragma unroll
      for (int q = 0; q < NA; ++q) acc0[i][p][q] = f32x4{0, 0, 0, 0};
#pragma unroll
      for (int q = 0; q < 2; ++q) acc1[i][p][q] = f32x4{0, 0, 0, 0};
    }
  // row tiles (mh * 4 + mi) of this wave with rows below M: the MFMAs of a row half without real rows are skipped (the
  // 64-row last panel of M = 16 448 keeps one half of one wave row busy - its tiles cost a third of a full one)
  int rt_hi = (a.M - m0 - wr * 128 + 15) >> 4;
  rt_hi = rt_hi < 0 ? 0 : (rt_hi > 8 ? 8 : rt_hi);
  const int live_mask = __builtin_amdgcn_readfirstlane((rt_hi > 0 ? 1 : 0) | (rt_hi > 4 ? 2 : 0));   // scalar: s_bitcmp + s_cbranch

  auto readA = [&](int mh, int unit) {
    const char* base = lds + ((unit >> 2) & 1) * GE::KT + GE::unit_off(mh ? 3 : 0) + a_row0 * 128;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      Af[0][0][mi] = *(const bf16x8*)(base + mi * 2048 + rdo0);
      Af[0][1][mi] = *(const bf16x8*)(base + mi * 2048 + rdo1);
    }
  };
  auto readB = [&](int nh, int unit) {
    if (nh == 0) {
      const char* base = lds + ((unit >> 2) & 1) * GE::KT + GE::unit_off(1) + wc * (16 * NA) * 128;
#pragma unroll
      for (int ni = 0; ni < NA; ++ni) {
        Bf0[0][ni] = *(const bf16x8*)(base + ni * 2048 + rdo0);
        Bf0[1][ni] = *(const bf16x8*)(base + ni * 2048 + rdo1);
      }
    } else {
      const char* base = lds + ((unit >> 2) & 1) * GE::KT + GE::unit_off(2) + wc * 32 * 128;
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        Bf1[0][ni] = *(const bf16x8*)(base + ni * 2048 + rdo0);
        Bf1[1][ni] = *(const bf16x8*)(base + ni * 2048 + rdo1);
      }
    }
  };
  // 16 MFMAs of one quadrant; the DMA of the next unit is issued from inside the block (the matrix pipe is busy for 16
  // cycles per MFMA, the issue slots in between are free), which keeps the R intervals short
  auto mma = [&](int mh, int nh, auto kind_c) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (live_mask & (mh ? 2 : 1)) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        if (nh == 0) {
#pragma unroll
          for (int ni = 0; ni < NA; ++ni)
            acc0[mh][mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Bf0[ks][ni], Af[0][ks][mi], acc0[mh][mi][ni], 0, 0, 0);
        } else {
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
            acc1[mh][mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Bf1[ks][ni], Af[0][ks][mi], acc1[mh][mi][ni], 0, 0, 0);
        }
      }
    }
    }
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
    if (ok == DG_D - 2) dg_wait_imm<GE::INFLIGHT>();         // steady state: one compare
    else dg_wait_vmcnt(ok > 0 ? young_instr(ok) : 0);
  };

  // ---- prologue: the whole ring in flight
  // (a K range shorter than the ring simply has fewer units: the guards keep u_issue <= nunits)
#define DG_PRO(i) if constexpr ((i) < DG_D) { if (u_issue < nunits) issue_unit(DG_IC((i) & 3)); }
  DG_PRO(0) DG_PRO(1) DG_PRO(2) DG_PRO(3) DG_PRO(4) DG_PRO(5) DG_PRO(6) DG_PRO(7) DG_PRO(8) DG_PRO(9)
#undef DG_PRO
  {
    // units 0, 1 landed before anyone reads.  Unit 0 of the first K-tile is read here by everybody (all later "first
    // row half" units are read in phase 3 of the previous K-tile): with the ring completely in flight (D = slots) the
    // slot of unit v is refilled in M_v, which is only safe because every unit is read in a phase < v.
    const int ok = (u_issue - 1) - 1;
    dg_wait_vmcnt(ok > 0 ? young_instr(ok) : 0);
  }
  if (hi) __builtin_amdgcn_s_barrier();
  DGT2(1);

  // One K-tile = four phases.  STEADY: every unit issued here exists and the ring is full, so the DMA issue and the
  // landed-wait need no conditions (one immediate vmcnt); the last K-tiles of the range take the guarded path.
  auto ktile = [&](int t, auto steady_c) {
    constexpr bool STEADY = decltype(steady_c)::value != 0;
    const int u0 = 4 * t;
    auto dma = [&](auto kind_c) {
      if (STEADY || u_issue < nunits) issue_unit(kind_c);
    };
    auto landed = [&]() {
      if constexpr (STEADY) dg_wait_imm<GE::INFLIGHT>();
      else wait_landed();
    };
#define DG_RP(x) do { __builtin_amdgcn_s_setprio(x); } while (0)
#define DG_R(q, reads) do { DG_RP(2); dma(DG_IC(((q) + DG_D) & 3)); reads; DG_RP(0); } while (0)
    // phase 0: first row half x first column half
    __builtin_amdgcn_s_barrier();
    DGT();
    DG_R(0, readA(0, u0); readB(0, u0 + 1));
    if (hi) landed();
    DGT();
    __builtin_amdgcn_s_barrier();
    DGT();
    mma(0, 0, DG_IC((0 + DG_D) & 3));
    DGT();
    if (!hi) landed();
    ++g;
    // phase 1: first row half x second column half
    __builtin_amdgcn_s_barrier();
    DGT();
    DG_R(1, readB(1, u0 + 2));
    if (hi) landed();
    DGT();
    __builtin_amdgcn_s_barrier();
    DGT();
    mma(0, 1, DG_IC((1 + DG_D) & 3));
    DGT();
    if (!hi) landed();
    ++g;
    // phase 2: second row half x second column half
    __builtin_amdgcn_s_barrier();
    DGT();
    DG_R(2, readA(1, u0 + 3));
    if (hi) landed();
    DGT();
    __builtin_amdgcn_s_barrier();
    DGT();
    mma(1, 1, DG_IC((2 + DG_D) & 3));
    DGT();
    if (!hi) landed();
    ++g;
    // phase 3: second row half x first column half (fragments already in registers)
    __builtin_amdgcn_s_barrier();
    DGT();
    DG_R(3, (void)0);
    if (hi) landed();
    DGT();
    __builtin_amdgcn_s_barrier();
    DGT();
    mma(1, 0, DG_IC((3 + DG_D) & 3));
    DGT();
    if (!hi) landed();
    ++g;
#undef DG_R
#undef DG_RP
  };
  // steady while the last unit issued in the K-tile, 4 t + 3 + D, exists
  const int t_steady = (nunits - 4 - DG_D) >= 0 ? (nunits - 4 - DG_D) / 4 + 1 : 0;
  int t = 0;
#pragma unroll 1
  for (; t < t_steady; ++t) ktile(t, DG_IC(1));
#pragma unroll 1
  for (; t < nkt; ++t) ktile(t, DG_IC(0));
  if (!hi) __builtin_amdgcn_s_barrier();   // re-align the two groups
  __builtin_amdgcn_s_barrier();            // every wave is done with the ring: LDS is free for the epilogue
  DGT2(2);

  // ---- split-K tail tiles: publish the partial tile, the last arriver of the tile reduces (MI355X guide, split-K recipe:
  // plain stores -> every wave drains -> barrier -> lane 0 agent release -> ticket; reducer: agent acquire -> barrier).
  // Only row tiles with rows below M travel: the tail round of M = 16 448 is mostly the 64-row last panel, a quarter of
  // whose slabs is data.  What a split costs is this slab round trip (tools/dense_phases.py, 207 parts of full tiles:
  // publish 9 us = 53 MB of f32 partials at HBM write rate, agent release / acquire 6-8 us of L2 write-back, reducer
  // 15-20 us reading split x 256 KiB through one CU) - dense_plan() splits only where the K loop it saves is longer.
  // (A cooperative variant - the parts meet at a counter and each reduces 1/split of the tile - measured no faster even
  // with the parts at the front of the grid: the wait replaces the reducer's reads, the fences and the 53 MB stay; and
  // it would forbid two of these GEMMs on two streams.)
  const int rt_lo = 0;
  // accumulator tile (mh, mi, n-tile j of the wave's NT) as ONE indexable thing for the slab / epilogue code
  auto ACC = [&](int mh, int mi, int j) -> f32x4& { return j < NA ? acc0[mh][mi][j] : acc1[mh][mi][j - NA]; };
  if (rem_idx >= 0 && a.split > 1) {
    float* slab = a.slabs + ((int64_t)rem_idx * a.split + part) * (DG_BM * DG_BN);
    // slab layout: [wave][acc register index][lane] float4 -> fully coalesced 16-byte stores and loads
    f32x4* sw4 = (f32x4*)slab + (int64_t)wid * (8 * NT) * 64 + lane;
#pragma unroll
    for (int mh = 0; mh < 2; ++mh)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        if (mh * 4 + mi >= rt_hi) continue;
#pragma unroll
        for (int j = 0; j < NT; ++j) sw4[((mh * 4 + mi) * NT + j) * 64] = ACC(mh, mi, j);
      }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    DGT2(5);
    int* flag = (int*)lds;                 // the ring is idle now
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const int old = __hip_atomic_fetch_add(a.tickets + rem_idx, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      flag[0] = (old == a.split - 1) ? 1 : 0;
      if (old == a.split - 1) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // every slice of the tile has arrived: re-arm the ticket for the next launch (the workspace is zeroed once, when
        // it is allocated; launches that share it are stream-ordered) - no per-launch memset node
        __hip_atomic_store(a.tickets + rem_idx, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    __syncthreads();
    const int last = flag[0];
    __syncthreads();                       // flag word is part of the staging area below
    if (last == 0) return;
    DGT2(6);
    // fixed summation order slab 0 + slab 1 + ... whichever part arrived last (its own partial is re-read from its
    // slab): the result does not depend on the arrival order, so identical launches give identical bits
    for (int p = 0; p < a.split; ++p) {
      const f32x4* o4 = (const f32x4*)(a.slabs + ((int64_t)rem_idx * a.split + p) * (DG_BM * DG_BN)) +
                        (int64_t)wid * (8 * NT) * 64 + lane;
#pragma unroll
      for (int mh = 0; mh < 2; ++mh)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          if (mh * 4 + mi >= rt_hi) continue;
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            const f32x4 o = o4[((mh * 4 + mi) * NT + j) * 64];
            ACC(mh, mi, j) = p == 0 ? o : ACC(mh, mi, j) + o;
          }
        }
    }
  }

  // ---- epilogue.  Each wave stages its 128 x 16 NT block (+ bias, rounded to bf16) in its own LDS tile in the MFMA layout
  // (lane (fr, kg) of tile (m-tile, n-tile): token row fr, output columns 4 kg .. 4 kg + 3) and reads it back row-wise:
  // a global instruction then moves whole row pieces (16 B per lane) instead of 16 rows x 32 bytes.
  if constexpr (NT != 4) {
    // 320-wide tile, plain epilogue: two passes of 64 rows (the staged 128 x 80 blocks of 8 waves exceed the LDS);
    // read-back: a row piece is 160 B = 10 chunks, an instruction moves 6 rows (lanes 60..63 idle)
    constexpr int SRS = GE::SRS;
    char* const stg = lds + wid * (GE::EPI_ROWS * SRS);
    const int er = lane / 10, ech = lane - er * 10;
    const int n = n0 + wc * GE::WCOLS + ech * 8;
    const bool nok = n < a.N && lane < 60;
#pragma unroll
    for (int mh = 0; mh < 2; ++mh) {
      if (mh * 4 >= rt_hi) break;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int nl = j * 16 + kg * 4;
        const int nb = n0 + wc * GE::WCOLS + nl;
        f32x4 bv = {0, 0, 0, 0};
        if (a.bias && nb < a.N) bv = *(const f32x4*)(a.bias + nb);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          if (mh * 4 + mi >= rt_hi) continue;
          const f32x4 v = ACC(mh, mi, j) + bv;
          const bf16x4 cb = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
          *(bf16x4*)(stg + (mi * 16 + fr) * SRS + nl * 2) = cb;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // a wave's LDS operations complete in order; the tile is private
#pragma unroll
      for (int it = 0; it < 11; ++it) {
        const int row = it * 6 + er;
        const int m = m0 + wr * 128 + mh * 64 + row;
        if (row >= 64 || m >= a.M || !nok || mh * 4 + (row >> 4) >= rt_hi) continue;
        const u32x4 raw = *(const u32x4*)(stg + row * SRS + ech * 16);
        *(u32x4*)(a.C + (int64_t)m * a.ldc + n) = raw;
      }
      // the next pass re-uses the staging tile: its ds_writes follow these ds_reads in program order (in-order LDS)
    }
  } else {
  constexpr int SRS = GE::SRS;               // staged row stride: 128 B of data + 16 B (bank spread of the 8-byte writes)
  char* const stg = lds + wid * (128 * SRS);
#pragma unroll
  for (int j = 0; j < NT; ++j) {
      const int nl = j * 16 + kg * 4;
      const int n = n0 + wc * 64 + nl;
      f32x4 bv = {0, 0, 0, 0};
      if (a.bias && n < a.N) bv = *(const f32x4*)(a.bias + n);
#pragma unroll
      for (int mh = 0; mh < 2; ++mh)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          if (mh * 4 + mi < rt_lo || mh * 4 + mi >= rt_hi) continue;
          const f32x4 v = ACC(mh, mi, j) + bv;
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
  // the accumulators are dead now: every global operand of the row-wise pass is requested up front (16 rows x 32 B of x
  // or 16 B of h per lane in flight), so the pass pays the memory latency once instead of once per row
  f32x4 xin[MODE == DG_RESID ? 16 : 1][2];
  bf16x8 hin[(MODE == DG_DGELU || MODE == DG_DFACT) ? 16 : 1];
  float csum[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // DGELU: column sums of this lane's rows (bias gradient of fc1)
  if (MODE == DG_RESID || MODE == DG_DGELU || MODE == DG_DFACT) {
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int m = m0 + wr * 128 + it * 8 + srow;
      const bool ok = m < a.M && nok && (it >> 1) >= rt_lo && (it >> 1) < rt_hi;
      if (MODE == DG_RESID) {
        const float* xp = a.X + (int64_t)(ok ? m : 0) * a.N + (ok ? n : 0);
        xin[it][0] = *(const f32x4*)xp;
        xin[it][1] = *(const f32x4*)(xp + 4);
      } else {
        hin[it] = *(const bf16x8*)(a.H + (int64_t)(ok ? m : 0) * a.ldc + (ok ? n : 0));
      }
    }
  }
#pragma unroll
  for (int it = 0; it < 16; ++it) {
    const int row = it * 8 + srow;
    const int m = m0 + wr * 128 + row;
    if (m >= a.M || !nok || (it >> 1) < rt_lo || (it >> 1) >= rt_hi) continue;
    const u32x4 raw = *(const u32x4*)(stg + row * SRS + sch * 16);
    const bf16x8 cb = __builtin_bit_cast(bf16x8, raw);
    bf16* cp = a.C + (int64_t)m * a.ldc + n;
    if (MODE == DG_PLAIN) {
      *(u32x4*)cp = raw;
    } else if (MODE == DG_GELU) {
      *(u32x4*)cp = raw;                     // pre-activation, kept for the backward
      bf16x8 y;
#pragma unroll
      for (int e = 0; e < 8; ++e) y[e] = (bf16)dg_gelu((float)cb[e]);           // F.gelu of the bf16-rounded h
      *(bf16x8*)(a.C2 + (int64_t)m * a.ldc + n) = y;
    } else if (MODE == DG_GELUO) {
      bf16x8 y;
#pragma unroll
      for (int e = 0; e < 8; ++e) y[e] = (bf16)dg_gelu((float)cb[e]);           // F.gelu of the bf16-rounded h, as mode 1
      *(bf16x8*)cp = y;
    } else if (MODE == DG_GELUF) {
      bf16x8 y, f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float gv, dv;
        dg_gelu_both((float)cb[e], gv, dv);    // of the bf16-rounded pre-activation, as the two-pass form sees it
        y[e] = (bf16)gv;
        f[e] = (bf16)dv;
      }
      *(bf16x8*)cp = f;                      // gelu'(h): what the fc2 input gradient multiplies by (DG_DFACT)
      *(bf16x8*)(a.C2 + (int64_t)m * a.ldc + n) = y;
    } else if (MODE == DG_RESID) {
      *(u32x4*)cp = raw;                     // branch output, needed for d gamma
      const float rsv = a.rs ? a.rs[m / a.rps] : 1.0f;
      f32x4 x0 = xin[it][0], x1 = xin[it][1];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        x0[e] += rsv * gm0[e] * (float)cb[e];
        x1[e] += rsv * gm1[e] * (float)cb[4 + e];
      }
      float* op = a.OUT + (int64_t)m * a.N + n;
      *(f32x4*)op = x0;
      *(f32x4*)(op + 4) = x1;
    } else {                                 // DG_DGELU: dh = gelu'(h) * g;  DG_DFACT: dh = factor * g
      const bf16x8 h = hin[it];
      bf16x8 d;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        d[e] = (bf16)((MODE == DG_DFACT ? (float)h[e] : dg_gelu_grad((float)h[e])) * (float)cb[e]);
        csum[e] += (float)d[e];                // sum what the GEMMs downstream actually see
      }
      *(bf16x8*)cp = d;
    }
  }
  // DGELU column sums: slab row 2 tm + wr holds the sums over this wave's rows (the tile's epilogue runs in exactly one
  // workgroup): every element of the [tiles_m * 2, N] buffer is written once per launch, octic_dense_finish adds the rows
  if ((MODE == DG_DGELU || MODE == DG_DFACT) && a.colsum != nullptr) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float v = csum[e];
      v += __shfl_xor(v, 8, 64);
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      csum[e] = v;
    }
    if (srow == 0 && nok) {
      float* base = a.colsum + (int64_t)(tm * 2 + wr) * a.N + n;
      *(f32x4*)base = f32x4{csum[0], csum[1], csum[2], csum[3]};
      *(f32x4*)(base + 4) = f32x4{csum[4], csum[5], csum[6], csum[7]};
    }
  }
  }   // NT == 4
#ifdef DG_TRACE2
  DGT2(3);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  DGT2(4);
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// The class-token rows of per-image launches: C[b * tokens][n] for b < nrows.  One workgroup = 16 rows x 16 columns; its
// blockDim / 64 waves (1 .. 16, chosen by the host so that a wave contracts 320 of K where K allows: 4 / 12 / 16 waves for
// K = 1280 / 3840 / 5120) each contract K / waves (a multiple of 32) straight from global memory - operands are tiny, no LDS
// staging, ONE batch of loads per wave so a workgroup lives one memory round trip -, partial tiles are summed through LDS in
// wave order (fixed order: bitwise reproducible), wave 0 runs the mode's epilogue with the same per-element math as
// dense_nt_kernel (bias added in f32, one rounding to bf16, GELU / factor on the rounded value; its operands are requested
// before the contraction).  The grid is (N / 16) x ceil(nrows / 16) workgroups = 320 for N = 1280, B = 64: all resident at once.
template <int MODE>
__global__ __launch_bounds__(1024) void dense_cls_kernel(DgArgs a, int tokens, int nrows, int colsum_row0) {
  __shared__ f32x4 part[16][64];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nw = blockDim.x >> 6;
  const int fr = lane & 15, kg = lane >> 4;
  const int ntn = a.N >> 4;
  const int tn = blockIdx.x % ntn, tb = blockIdx.x / ntn;
  const int n0 = tn * 16, b0 = tb * 16;
  const int kw = a.K / nw;                                  // K range of this wave (host: a multiple of 32)
  const int nks = kw >> 5;
  int brow = b0 + fr;
  const bool rok = brow < nrows;
  brow = rok ? brow : nrows - 1;                            // rows past the batch read a valid row and are dropped at the end
  const bf16* wp = a.B + (int64_t)(n0 + fr) * a.ldb + wid * kw + kg * 8;
  const bf16* xp = a.A + (int64_t)brow * tokens * a.lda + wid * kw + kg * 8;
  // epilogue operands of wave 0: lane (fr, kg) = token row b0 + fr, output columns n0 + 4 kg .. + 3
  const int n = n0 + 4 * kg;
  const int64_t m = (int64_t)(b0 + fr) * tokens;
  f32x4 bv = {0, 0, 0, 0};
  bf16x4 hv = {0, 0, 0, 0};
  if (wid == 0) {
    if (a.bias) bv = *(const f32x4*)(a.bias + n);
    if ((MODE == DG_DGELU || MODE == DG_DFACT) && rok) hv = *(const bf16x4*)(a.H + m * a.ldc + n);
  }
  f32x4 acc = {0, 0, 0, 0};
  // k-steps per batch of loads.  Static register indices only: a runtime-selected double buffer put the fragments in scratch
  // memory (first version of this kernel: 34-49 us per launch)
  constexpr int U = 10;
  bf16x8 wf[U], xf[U];
#pragma unroll 1
  for (int ks0 = 0; ks0 < nks; ks0 += U) {
#pragma unroll
    for (int j = 0; j < U; ++j)
      if (ks0 + j < nks) {
        wf[j] = *(const bf16x8*)(wp + (ks0 + j) * 32);
        xf[j] = *(const bf16x8*)(xp + (ks0 + j) * 32);
      }
#pragma unroll
    for (int j = 0; j < U; ++j)
      if (ks0 + j < nks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[j], acc, 0, 0, 0);
  }
  part[wid][lane] = acc;
  __syncthreads();
  if (wid != 0) return;
  f32x4 v = part[0][lane];
  for (int w = 1; w < nw; ++w) v = v + part[w][lane];
  v = v + bv;
  const bf16x4 cb = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
  bf16* cp = a.C + m * a.ldc + n;
  float cs[4] = {0, 0, 0, 0};
  if (rok) {
    if (MODE == DG_PLAIN) {
      *(bf16x4*)cp = cb;
    } else if (MODE == DG_GELU || MODE == DG_GELUO || MODE == DG_GELUF) {
      bf16x4 y, f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float gv, dv;
        if (MODE == DG_GELUF) dg_gelu_both((float)cb[e], gv, dv);
        else { gv = dg_gelu((float)cb[e]); dv = 0.0f; }
        y[e] = (bf16)gv;
        f[e] = (bf16)dv;
      }
      if (MODE == DG_GELUO) *(bf16x4*)cp = y;
      else {
        *(bf16x4*)cp = MODE == DG_GELUF ? f : cb;
        *(bf16x4*)(a.C2 + m * a.ldc + n) = y;
      }
    } else if (MODE == DG_DGELU || MODE == DG_DFACT) {
      const bf16x4 h = hv;
      bf16x4 d;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        d[e] = (bf16)((MODE == DG_DFACT ? (float)h[e] : dg_gelu_grad((float)h[e])) * (float)cb[e]);
        cs[e] = (float)d[e];
      }
      *(bf16x4*)cp = d;
    }
  }
  if ((MODE == DG_DGELU || MODE == DG_DFACT) && a.colsum != nullptr) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float t = cs[e];
      t += __shfl_xor(t, 1, 64);
      t += __shfl_xor(t, 2, 64);
      t += __shfl_xor(t, 4, 64);
      t += __shfl_xor(t, 8, 64);
      cs[e] = t;
    }
    if (fr == 0) *(f32x4*)(a.colsum + (int64_t)(colsum_row0 + tb) * a.N + n) = f32x4{cs[0], cs[1], cs[2], cs[3]};
  }
}


// Long K (>= 3840): the class-token GEMM as TWO launches - K cut over workgroups, f32 partial tiles, a second launch that sums
// them in slice order and runs the plain epilogue.  Why: a 16 x 16 tile pulls (16 + 16) K bytes through its CU and a CU
// sustains ~25 GB/s of first-touch loads, so at K = 5120 the single launch above is 21 us however its waves are arranged;
// here a workgroup owns 64 rows x 64 columns x 320 of K (82 KB of operands, one batch of loads, W fragments re-used over four row
// tiles), 320 workgroups for N = 1280, K = 5120.  A kernel boundary is the (cheap) cross-workgroup reduction: no tickets, no
// fences.  Plain mode only (fc2, the input gradients of fc1 / qkv).
constexpr int CLS_KSLICE = 320;
__global__ __launch_bounds__(320) void dense_cls_part_kernel(DgArgs a, int tokens, int nrows, int rows_pad, float* part) {
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int fr = lane & 15, kg = lane >> 4;
  const int bw = (blockDim.x >> 6) * 16;                    // columns per workgroup: 64 or 80 (one 16-column tile per wave)
  const int nnb = a.N / bw, nks = a.K / CLS_KSLICE;
  const int nb = blockIdx.x % nnb, ks = (blockIdx.x / nnb) % nks, rb = blockIdx.x / (nnb * nks);
  const int n0 = nb * bw + wid * 16;
  const bf16* wp = a.B + (int64_t)(n0 + fr) * a.ldb + ks * CLS_KSLICE + kg * 8;
  const bf16* xp[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int r = rb * 64 + i * 16 + fr;
    r = r < nrows ? r : nrows - 1;                          // rows past the batch: a valid row, dropped by the second launch
    xp[i] = a.A + (int64_t)r * tokens * a.lda + ks * CLS_KSLICE + kg * 8;
  }
  constexpr int U = CLS_KSLICE / 32;                        // 10 k-steps: ONE batch of 50 loads per wave
  bf16x8 wf[U], xf[4][U];
#pragma unroll
  for (int j = 0; j < U; ++j) {
    wf[j] = *(const bf16x8*)(wp + j * 32);
#pragma unroll
    for (int i = 0; i < 4; ++i) xf[i][j] = *(const bf16x8*)(xp[i] + j * 32);
  }
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
  for (int j = 0; j < U; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i][j], acc[i], 0, 0, 0);
  // lane (fr, kg) of row tile i: row rb * 64 + 16 i + fr, columns n0 + 4 kg .. + 3
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = rb * 64 + i * 16 + fr;
    *(f32x4*)(part + ((int64_t)ks * rows_pad + r) * a.N + n0 + 4 * kg) = acc[i];
  }
}

__global__ __launch_bounds__(256) void dense_cls_sum_kernel(DgArgs a, int tokens, int nrows, int rows_pad, const float* part) {
  const int n4 = a.N >> 2;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const int r = idx / n4, n = (idx - r * n4) * 4;
  if (r >= nrows) return;
  const int nks = a.K / CLS_KSLICE;
  f32x4 v = *(const f32x4*)(part + (int64_t)r * a.N + n);
  for (int ks = 1; ks < nks; ++ks) v = v + *(const f32x4*)(part + ((int64_t)ks * rows_pad + r) * a.N + n);
  if (a.bias) v = v + *(const f32x4*)(a.bias + n);
  *(bf16x4*)(a.C + (int64_t)r * tokens * a.ldc + n) = bf16x4{(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
}

// ---------------------------------------------------------------------------------------------------------------
// Schedule: `full` tiles get one workgroup each; the remaining r = tiles - full tiles (the partial last round on the
// `cus` workgroup slots) are split along K over `split` workgroups each, so the last round also fills the chip.
struct DgPlan { int nt, tiles_m, tiles_n, full, rem, split, tail_pad, grid; double cost; };

// `cost` = launch length in units of one K-tile of the 256-wide tile (a model, used only to choose the tile width):
// whole rounds of full tiles + the split parts in front of the grid.
inline DgPlan dense_plan_nt(int M, int N, int K, int cus, int nt, int force_split = 0, bool front_unsplit = false) {
  DgPlan p;
  const int bn = 64 * nt;
  const double wf = nt / 4.0;                              // K-tile cost and slab size relative to the 256-wide tile
  p.nt = nt;
  p.tiles_m = (M + DG_BM - 1) / DG_BM;
  p.tiles_n = (N + bn - 1) / bn;
  const int tiles = p.tiles_m * p.tiles_n;
  const int rounds = tiles / cus;
  const int nkt = K / DG_BK;
  p.full = rounds * cus;
  p.rem = tiles - p.full;
  p.split = 1;
  double f = 1.0;                                          // share of real rows in the remaining tiles
  if (p.rem > 0) {
    int s = cus / p.rem;                                   // workgroup slots per remaining tile
    const int smax = nkt / 4 > 0 ? nkt / 4 : 1;            // keep >= 4 K-tiles per part
    s = s < 1 ? 1 : (s > smax ? smax : s);
    s = s > 8 ? 8 : s;
    // a tail that nearly fills a round is cheaper unsplit (no slab traffic)
    if (p.rem * 10 >= cus * 8) s = 1;
    // ... and so is a short K: a split tile pays the slab round trip (f32 partial out, `s` partials back through the
    // reducer's CU: ~26 us = 17 K-tiles of this kernel for a tile of 256 real rows, tools/dense_phases.py), in proportion
    // to the rows that exist (the tail of M = 16 448 is mostly the 64-row last panel).  Split only if K/s + that < K.
    {
      int64_t rows = 0;
      for (int tile = p.full; tile < tiles; ++tile) {
        const int G = 8, per_group = G * p.tiles_n, gidx = tile / per_group, first_m = gidx * G;
        const int gsz = (p.tiles_m - first_m) < G ? (p.tiles_m - first_m) : G;
        const int tm = first_m + (tile - gidx * per_group) % gsz;
        const int left = M - tm * DG_BM;
        rows += left < DG_BM ? left : DG_BM;
      }
      f = (double)rows / ((double)p.rem * DG_BM);
    }
    if (s > 1 && (double)nkt / s + 17.0 * f * wf + 1.0 >= (double)nkt) s = 1;
    if (force_split > 0) s = force_split > smax ? smax : force_split;
    p.split = s;
    // an unsplit tail is just more full tiles - unless it is a thin (mostly empty) panel: those tiles are short and stay
    // in FRONT of the grid, where they run beside the first round instead of after it
    if (s == 1 && !(front_unsplit && f < 0.5 && rounds >= 1)) {
      p.full = tiles;
      p.rem = 0;
    }
  }
  p.tail_pad = (p.rem * p.split + 7) & ~7;
  p.grid = p.tail_pad + p.full;
  const double tile_cost = nkt * wf + 3.0;                 // + prologue / epilogue / launch gap (~5 us)
  p.cost = (double)((p.full + cus - 1) / cus) * tile_cost;
  if (p.rem > 0) {
    // rows that do not exist cost no MFMAs (row halves without real rows are skipped): a part of a mostly empty tile
    // is bounded by its DMA / barrier skeleton, about half a full K-tile
    const double kt = (f < 0.5 ? 0.5 : 1.0) * wf;
    p.cost += (double)nkt / p.split * kt + 17.0 * f * wf + 3.0;
  }
  return p;
}

// The workspace of one (M, N, K) is shared by every mode and tile width of that problem: the ticket words (zero between
// launches: the last arriver re-arms its ticket) live in a region of FIXED size in front of the slabs, so no plan's slabs
// can land on another plan's tickets.
constexpr int DG_TICKET_BYTES = 8192;

// routing overrides (octic_route_override): OCTIC_ROUTE_DENSE_TILE 0 = choose by the cost model, 4 / 5 = force;
// OCTIC_ROUTE_DENSE_SPLIT 0 = plan's choice, n = split of the remaining tiles (1 = unsplit, kept in front)
static inline int dense_force_nt() { const int v = route(OCTIC_ROUTE_DENSE_TILE); return (v == 4 || v == 5) ? v : 0; }
static inline int dense_force_split() { const int v = route(OCTIC_ROUTE_DENSE_SPLIT); return v > 0 ? v : 0; }

// The 320-wide tile serves plain-epilogue problems whose N is a multiple of 320 when the model says its launch is shorter
// (ViT-H: N = 1280 - one round of 256 tiles instead of 1.27 rounds of 325).
inline DgPlan dense_plan(int M, int N, int K, int cus, int mode) {
  const DgPlan p4 = dense_plan_nt(M, N, K, cus, 4, dense_force_split(), dense_force_split() == 1);
  if (mode != DG_PLAIN || (N % 320) != 0 || dense_force_nt() == 4) return p4;
  const DgPlan p5 = dense_plan_nt(M, N, K, cus, 5, dense_force_split(), dense_force_split() == 1);
  if (dense_force_nt() == 5) return p5;
  return p5.cost < p4.cost ? p5 : p4;
}

// Waves of a class-token workgroup: K / waves a multiple of 32, 320 per wave where K allows, at most 16 waves.
inline int dense_cls_waves(int K) {
  if (K % 320 == 0 && K / 320 <= 16) return K / 320;
  for (int w = 16; w > 1; --w)
    if (K % (32 * w) == 0) return w;
  return 1;
}

// Per-image panels (see the file header): taken when the batch is whole images of 256 q + 1 tokens with q = 1 (ViT-H/14's 257),
// the class-token kernel takes the shape (K % 128 == 0, N % 16 == 0, no fused residual tail) and the launch model says so:
// the panels' plan + ~4 K-tile units for the class-token launch against the classic plan.
// OCTIC_ROUTE_DENSE_IMAGE: 0 = by the model, 1 = always where legal, 2 = never, 3 = by the model for plain launches only (A/B).
struct DgTokPlan { DgPlan p; bool image; int images; };
#ifndef DG_CLS2_US
#define DG_CLS2_US 15.0    // measured (rocprofv3, B = 64): part 9.0-10.4 us + sum 5.0 us at K = 3840 / 5120
#endif
// What the class-token launch costs, in the plan's unit (one K-tile of the 256-wide tile, ~1.54 us): measured on MI355X at
// B = 64 (rocprofv3, round 6): 7.0 us for N = K = 1280, 11.4 (N 3840), 13.7-16.2 (N 5120), 16.2 (K 3840), 21.1 us (K 5120) -
// bound by L2 -> CU operand traffic of 16 x 16 output tiles ((16 + 16) K bytes per 512 flops) and by its one memory round trip;
// fitted as 7 us x (N / 1280)^0.5 x (K / 1280)^0.8, scaled by the batch.
inline bool dense_cls_two_launches(int mode, int N, int K) {
  const int v = route(OCTIC_ROUTE_DENSE_CLS2);              // 0 = by shape, 1 = never, 2 = wherever legal
  const bool legal = mode == DG_PLAIN && (K % CLS_KSLICE) == 0 && (N % 64) == 0;
  return legal && v != 1 && (v == 2 || K >= 3840);
}
inline double dense_cls_cost(int B, int N, int K, int mode) {
  double us = 7.0 * sqrt((double)N / 1280.0) * pow((double)K / 1280.0, 0.8);
  if (dense_cls_two_launches(mode, N, K)) us = DG_CLS2_US;   // (two launches, measured: see DG_CLS2_US)
  return us * (B <= 64 ? 1.0 : (double)B / 64.0) / 1.54;
}
inline DgTokPlan dense_plan_tokens(int M, int N, int K, int cus, int mode, int tokens) {
  DgTokPlan t;
  t.p = dense_plan(M, N, K, cus, mode);
  t.image = false;
  t.images = 0;
  const int force = route(OCTIC_ROUTE_DENSE_IMAGE);
  if (tokens != 257 || M % tokens != 0 || mode == DG_RESID || (K % 128) != 0 || (N % 16) != 0 || force == 2) return t;
  const int B = M / tokens;
  const DgPlan pi = dense_plan(B * DG_BM, N, K, cus, mode);
  // (modes 3 / 5 - the fc2 input gradient with its column sums - measured no shorter on per-image panels: 221.2 against
  // 221.8 us at ViT-H; their class-token launch would be a pure loss)
  // (fc1 with its GELU tails: the panels save 20 us, the class-token launch costs 14-16 - and in the step the pair measured
  // 0.13 ms per step SLOWER than leaving fc1 on classic panels (A/B/C of alternating processes, round 6): the margin keeps it out)
  const double margin = 1.0;
  const bool pays = mode != DG_DGELU && mode != DG_DFACT && pi.cost + dense_cls_cost(B, N, K, mode) + margin < t.p.cost;
  if (force == 1 || (force == 0 && pays) || (force == 3 && pays && mode == DG_PLAIN)) {
    t.p = pi;
    t.image = true;
    t.images = B;
  }
  return t;
}

}  // namespace octic

using namespace octic;

extern "C" {

// CUs of the current device (the split-K parts of a tile wait for each other: the plan must not assume more workgroup
// slots than the device has); 256 when no device is visible (host-only callers sizing a workspace)
static int dense_cus() { return device_cus(); }

// enough for either tile width (the plain mode may pick the 320-wide tile, the fused tails use the 256-wide one)
int64_t octic_dense_gemm_workspace_bytes(int M, int N, int K) {
  int64_t need = 256;
  // (the per-image panels of a 257-token batch are planned as 256 rows per image: room for either plan, and for the f32
  // partial tiles of the two-launch class-token path behind the ticket words)
  if ((M % 257) == 0 && (K % CLS_KSLICE) == 0) {
    const int64_t rows_pad = ((int64_t)(M / 257) + 63) / 64 * 64;
    need = DG_TICKET_BYTES + (int64_t)(K / CLS_KSLICE) * rows_pad * N * 4 + 256;
  }
  const int Ms[2] = {M, (M % 257) == 0 ? M / 257 * DG_BM : 0};
  for (int i = 0; i < 2; ++i) {
    if (Ms[i] <= 0) continue;
    for (int nt = 4; nt <= 5; ++nt) {
      if (nt == 5 && (N % 320) != 0) continue;
      DgPlan p = dense_plan_nt(Ms[i], N, K, dense_cus(), nt);
      if (dense_force_split() > 1) p = dense_plan_nt(Ms[i], N, K, dense_cus(), nt, 8);     // developer switch: room for any split
      if (p.split <= 1) continue;
      const int64_t b = (int64_t)p.rem * p.split * DG_BM * (64 * nt) * 4 + DG_TICKET_BYTES + 256;
      need = b > need ? b : need;
    }
  }
  return need;
}

// out[0] = tile width (256 | 320), out[1] = colsum slab rows of modes 3 / 5, out[2] = 1 if the launch uses per-image panels
// + the class-token kernel, out[3] = workgroups of the main launch
int octic_dense_gemm_plan(int M, int N, int K, int mode, int tokens, int* out) {
  if (!out) return OCTIC_ENULL;
  if (M <= 0 || N <= 0 || K < 2 * DG_BK) { out[0] = 256; out[1] = 0; out[2] = 0; out[3] = 0; return OCTIC_OK; }
  const DgTokPlan t = dense_plan_tokens(M, N, K, dense_cus(), mode, tokens);
  out[0] = t.p.nt == 5 ? 320 : 256;
  out[1] = t.image ? 2 * t.images + (t.images + 15) / 16 : 2 * ((M + DG_BM - 1) / DG_BM);
  out[2] = t.image ? 1 : 0;
  out[3] = t.p.grid;
  return OCTIC_OK;
}

int octic_dense_gemm_tile(int M, int N, int K, int mode) {
  if (M <= 0 || N <= 0 || K < 2 * DG_BK) return 256;
  return dense_plan(M, N, K, dense_cus(), mode).nt == 5 ? 320 : 256;
}

int octic_dense_gemm_colsum_rows(int M, int N, int K) {
  (void)N; (void)K;
  return 2 * ((M + DG_BM - 1) / DG_BM);
}

// mode: 0 plain (C = A B^T + bias), 1 GELU (C = pre-activation, C2 = gelu(C)), 2 RESID (C = branch, OUT = X + rs*gamma*C),
// 3 DGELU (C = gelu'(H) * (A B^T); colsum != NULL: octic_dense_gemm_colsum_rows() slabs [N] of column sums of C),
// 4 GELUF (C = gelu'(pre-activation), C2 = gelu(pre-activation)), 5 DFACT (C = H * (A B^T), H = the factor of mode 4),
// 6 GELUO (C = gelu(pre-activation) only: passes without a backward).
int octic_dense_gemm_nt_tokens(const void* A, const void* B, int M, int N, int K, int64_t lda, int64_t ldb, int mode, void* C,
                               void* C2, int64_t ldc, const float* bias, const float* gamma, const float* rs, int64_t rps,
                               const float* X, float* OUT, const void* H, float* colsum, void* workspace, int tokens,
                               void* stream) {
  if (!A || !B || !C) return OCTIC_ENULL;
  if (M <= 0 || N <= 0 || K <= 0 || (K % DG_BK) || K < 2 * DG_BK || (N % 8) || (lda % 8) || (ldb % 8) || (ldc % 4)) return OCTIC_ESHAPE;
  // buffer descriptors and per-lane offsets are 32-bit: operands of 2 GiB or more are refused (callers fall back to
  // the BLAS library) instead of wrapping
  if ((int64_t)M * lda * 2 >= (1ll << 31) || (int64_t)N * ldb * 2 >= (1ll << 31)) return OCTIC_ESHAPE;
  if ((((uintptr_t)A) | ((uintptr_t)B)) & 15) return OCTIC_EALIGN;
  if ((mode == DG_GELU || mode == DG_GELUF) && !C2) return OCTIC_ENULL;
  if (mode == DG_RESID && (!X || !OUT || (rs && rps <= 0))) return OCTIC_ENULL;
  if ((mode == DG_DGELU || mode == DG_DFACT) && !H) return OCTIC_ENULL;
  DgArgs a = {};
  a.A = (const bf16*)A; a.B = (const bf16*)B; a.lda = lda; a.ldb = ldb; a.M = M; a.N = N; a.K = K;
  a.C = (bf16*)C; a.C2 = (bf16*)C2; a.ldc = ldc; a.bias = bias; a.gamma = gamma; a.rs = rs; a.rps = rs ? rps : 1;
  a.X = X; a.OUT = OUT; a.H = (const bf16*)H; a.colsum = (mode == DG_DGELU || mode == DG_DFACT) ? colsum : nullptr;
  const DgTokPlan tp = dense_plan_tokens(M, N, K, dense_cus(), mode, tokens);
  const DgPlan p = tp.p;
  a.tiles_m = p.tiles_m; a.tiles_n = p.tiles_n; a.full_tiles = p.full; a.split = p.split; a.tail_pad = p.tail_pad;
  a.m_stride = tp.image ? tokens : DG_BM;
  a.m_base = tp.image ? 1 : 0;
  hipStream_t s = (hipStream_t)stream;
  if (p.split > 1) {
    if (!workspace) return OCTIC_ENULL;
    if (p.rem * 4 > DG_TICKET_BYTES) return OCTIC_ESHAPE;
    a.tickets = (int*)workspace;
    a.slabs = (float*)((char*)workspace + DG_TICKET_BYTES);
  }
  const int smem = DgGeom<4>::LDS, smem5 = DgGeom<5>::LDS;
  static DeviceOnce once;
  if (once.first()) {
    (void)hipFuncSetAttribute((const void*)dense_nt_kernel<DG_PLAIN, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    (void)hipFuncSetAttribute((const void*)dense_nt_kernel<DG_PLAIN, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, smem5);
    (void)hipFuncSetAttribute((const void*)dense_nt_kernel<DG_GELU, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    (void)hipFuncSetAttribute((const void*)dense_nt_kernel<DG_RESID, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    (void)hipFuncSetAttribute((const void*)dense_nt_kernel<DG_DGELU, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    (void)hipFuncSetAttribute((const void*)dense_nt_kernel<DG_GELUF, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    (void)hipFuncSetAttribute((const void*)dense_nt_kernel<DG_DFACT, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    (void)hipFuncSetAttribute((const void*)dense_nt_kernel<DG_GELUO, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    (void)hipGetLastError();
  }
  switch (mode) {
    case DG_PLAIN:
      if (p.nt == 5) dense_nt_kernel<DG_PLAIN, 5><<<p.grid, 512, smem5, s>>>(a);
      else dense_nt_kernel<DG_PLAIN, 4><<<p.grid, 512, smem, s>>>(a);
      break;
    case DG_GELU: dense_nt_kernel<DG_GELU, 4><<<p.grid, 512, smem, s>>>(a); break;
    case DG_RESID: dense_nt_kernel<DG_RESID, 4><<<p.grid, 512, smem, s>>>(a); break;
    case DG_DGELU: dense_nt_kernel<DG_DGELU, 4><<<p.grid, 512, smem, s>>>(a); break;
    case DG_GELUF: dense_nt_kernel<DG_GELUF, 4><<<p.grid, 512, smem, s>>>(a); break;
    case DG_DFACT: dense_nt_kernel<DG_DFACT, 4><<<p.grid, 512, smem, s>>>(a); break;
    case DG_GELUO: dense_nt_kernel<DG_GELUO, 4><<<p.grid, 512, smem, s>>>(a); break;
    default: return OCTIC_ESHAPE;
  }
  if (tp.image && dense_cls_two_launches(mode, N, K) && workspace && p.split <= 1) {
    // (the partial tiles live where split-K slabs would: only with a main launch that splits nothing - N = 1280 at any batch
    // that takes per-image panels; otherwise the single launch below)
    const int rows_pad = (tp.images + 63) / 64 * 64;
    float* part = (float*)((char*)workspace + DG_TICKET_BYTES);
    // 64 or 80 columns per workgroup: whichever keeps the grid within ONE workgroup per CU (a CU sustains ~25 GB/s of
    // first-touch loads: 320 workgroups on 256 CUs took 14.1 us where 240 took 8.2)
    const int cus = dense_cus(), per64 = (N / 64) * (K / CLS_KSLICE) * (rows_pad / 64);
    const int nw = (per64 > cus && (N % 80) == 0 && (N / 80) * (K / CLS_KSLICE) * (rows_pad / 64) <= cus) ? 5 : 4;
    dense_cls_part_kernel<<<(N / (16 * nw)) * (K / CLS_KSLICE) * (rows_pad / 64), 64 * nw, 0, s>>>(a, tokens, tp.images, rows_pad, part);
    dense_cls_sum_kernel<<<(tp.images * (N / 4) + 255) / 256, 256, 0, s>>>(a, tokens, tp.images, rows_pad, part);
  } else if (tp.image) {
    // the class-token rows: row b * tokens of A / C / C2 / H, colsum slab rows behind the panels' 2 * images
    const int grid = (N / 16) * ((tp.images + 15) / 16), cr0 = 2 * tp.images, thr = 64 * dense_cls_waves(K);
    switch (mode) {
      case DG_PLAIN: dense_cls_kernel<DG_PLAIN><<<grid, thr, 0, s>>>(a, tokens, tp.images, cr0); break;
      case DG_GELU: dense_cls_kernel<DG_GELU><<<grid, thr, 0, s>>>(a, tokens, tp.images, cr0); break;
      case DG_DGELU: dense_cls_kernel<DG_DGELU><<<grid, thr, 0, s>>>(a, tokens, tp.images, cr0); break;
      case DG_GELUF: dense_cls_kernel<DG_GELUF><<<grid, thr, 0, s>>>(a, tokens, tp.images, cr0); break;
      case DG_DFACT: dense_cls_kernel<DG_DFACT><<<grid, thr, 0, s>>>(a, tokens, tp.images, cr0); break;
      case DG_GELUO: dense_cls_kernel<DG_GELUO><<<grid, thr, 0, s>>>(a, tokens, tp.images, cr0); break;
      default: return OCTIC_ESHAPE;
    }
  }
  return launch_status();
}

int octic_dense_gemm_nt(const void* A, const void* B, int M, int N, int K, int64_t lda, int64_t ldb, int mode, void* C,
                        void* C2, int64_t ldc, const float* bias, const float* gamma, const float* rs, int64_t rps,
                        const float* X, float* OUT, const void* H, float* colsum, void* workspace, void* stream) {
  return octic_dense_gemm_nt_tokens(A, B, M, N, K, lda, ldb, mode, C, C2, ldc, bias, gamma, rs, rps, X, OUT, H, colsum,
                                    workspace, 0, stream);
}

}  // extern "C"
