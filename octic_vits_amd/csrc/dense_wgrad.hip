// Weight gradients of the standard half's nn.Linear layers (deit/vit.py:14-56 Attention.qkv / .proj, timm Mlp.fc1 / .fc2):
//
//   TN problem:  dW[N,K] = dY[M,N]^T · X[M,K]
//
// bf16 operands exactly as they lie in HBM (token rows M are the slow, strided dimension of BOTH operands), f32 result
// in the nn.Linear layout [N,K] — no transposed copy of the activations, no bf16 round trip of the gradient.
//
// Same machinery as the forward kernel (csrc/dense_gemm.hip): 256 x 256 output tile per workgroup, 8 waves as 2 x 4 with
// 128 (n) x 64 (k) per wave on v_mfma_f32_16x16x32_bf16, reduction walked in steps of 64 token rows, each step cut into
// four 16 KiB units that stream through an 8-slot LDS ring by buffer-addressed LDS-DMA with counted vmcnt waits, the
// two wave groups alternating between "read fragments + issue DMA" and "16 MFMAs" intervals.  What differs:
//   * a unit is 64 token rows x 128 columns (256-byte rows); MFMA operands need 8 consecutive token rows per lane, i.e. a
//     COLUMN of the staged tile: fragments come from ds_read_b64_tr_b16 (transposing read, 4 rows x 16 columns per
//     16-lane group).  32-byte slots are XOR-swizzled with ((row & 3) | ((row >> 3) & 1) << 2) on the DMA source side and
//     in the read address, so the 8 (row, slot) pieces a half-wave touches land in 8 different bank groups;
//   * the output is small (N x K) and the reduction long (M = 16 448 rows = 257 steps): there are only 25-100 tiles for
//     256 CUs, so the reduction is cut into S ROW SLABS and a workgroup owns one (slab, tile) item.  All tiles of a slab
//     are dealt to consecutive workgroups and walk the same token rows in lockstep, tiles that share a dY column panel
//     sit on one XCD: an operand row is fetched from HBM once per slab and then served out of L2 / the Infinity Cache.
//     (Round 2's stream-K split - equal contiguous ranges of tiles x steps - gave concurrent workgroups disjoint token
//     rows: every workgroup streamed its own operands, 1.7 GB of HBM traffic per launch instead of 0.2 GB, and the
//     kernel ran at the HBM rate, 0.53 PFLOP/s.)  Every item leaves an f32 partial slab; the LAST workgroup to finish
//     a tile (agent-scope ticket) adds the S slabs in slab order and writes dW (bitwise reproducible).
//     (Bias gradients come from the row kernels that already stream dY: csrc/dense.hip.)
#include <type_traits>
#include "octic_common.hpp"

namespace octic {

constexpr int DW_T = 256;                     // output tile side along n (and along k for the 256-wide tile)
constexpr int DW_BR = 64;                     // token rows per reduction step
constexpr int DW_UNIT = 64 * 256;             // bytes per unit: 64 rows x 128 bf16 columns
constexpr int DW_UNIT_X2 = 64 * 128;          // the third X piece of the 320-wide tile: 64 rows x 64 columns
// Tile width along k (KW = 16-column MFMA k-tiles per wave; 4 waves along k): KW = 4 is the 256 x 256 tile, KW = 5 a
// 256 x 320 one for K % 320 == 0 (round 4; see dw_plan() for where it is used).  A wave's 80 columns are k-sets of 2 + 3
// tiles; the units of a step are Y0 | X0 | X1 (+ X2) | Y1 = 16 / 16 / 24 / 16 KiB, two steps fill the ring.
template <int KW> struct DwGeom {
  static constexpr int BK = 64 * KW;                           // tile width along k
  static constexpr int WK = 16 * KW;                           // columns per wave
  static constexpr int U2 = DW_UNIT + (KW == 5 ? DW_UNIT_X2 : 0);
  static constexpr int STEP = 3 * DW_UNIT + U2;                // bytes of one step in the ring (64 / 72 KiB)
  static constexpr int RING = 2 * STEP;
  __host__ __device__ static constexpr int unit_off(int kind) { return kind == 0 ? 0 : kind == 1 ? DW_UNIT : kind == 2 ? 2 * DW_UNIT : 2 * DW_UNIT + U2; }
  __host__ __device__ static constexpr int unit_instr(int kind) { return kind == 2 && KW == 5 ? 3 : 2; }
  static constexpr int INFLIGHT = KW == 5 ? 9 : 8;             // DMA instructions of the 4 youngest units (one of each kind)
};
#ifndef DW_DIST
#define DW_DIST 6
#endif
#ifndef DW_MAP
#define DW_MAP 0        // 0: one slab spread over all XCDs (each XCD a contiguous run of the tn-major tile list);
                        // 1: an XCD works on 32 consecutive (slab, tile) items per round.  With the DMA ring actually
                        // running ahead (inline-asm LDS-DMA) 0 is 3-10 % faster at the same slab count.
#endif
#ifndef DW_AUX
#define DW_AUX 0        // cache policy bits of the LDS-DMA loads (2 = nt)
#endif
constexpr int DW_SLOTS = 8;                   // ring = two steps of four units
constexpr int DW_D = DW_DIST;
static_assert(DW_D == 6, "the counted waits assume a prefetch distance of 6 units");

struct DwArgs {
  const bf16* Y;      // dY [M, N]
  const bf16* X;      // X  [M, K]
  int64_t ldy, ldx;
  int M, N, K;
  float* W;           // dW [N, K]
  int tiles_k;        // K / tile width
  int tiles;          // (N / 256) * (K / 256), tile = tn * tiles_k + tk
  int tiles8;         // tiles rounded up to a multiple of 8: items per slab (the padding items exit at once)
  int steps;          // ceil(M / 64) reduction steps per tile
  int S;              // row slabs: slab s covers steps [steps s / S, steps (s + 1) / S)
  float* slabs;       // [tiles][S] x 256 x (256 | 320) f32 partial tiles
  int* tickets;       // [tiles], zero when the workspace is created; the last arriver of a tile re-arms its ticket
  // A SECOND problem with the same M and K in the same launch (round 5: octic_dense_wgrad_tn_pair): tiles [tiles0, tiles) of
  // the tile list belong to it.  tiles0 = 0: one problem.  (1280 x 1280 alone is 25 tiles x 8 slabs and runs at 0.6 PF; next
  // to the 75 tiles of 3840 x 1280 the pair is the 100-tile, two-slab shape of the MLP weights.)
  const bf16* Y1; const bf16* X1;
  int64_t ldy1, ldx1;
  int N1;
  float* W1;
  int tiles0;
};

__device__ inline void dw_wait_vmcnt(int n) {
  switch (n) {
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

template <int N> __device__ inline void dw_wait_steady() {          // steady state: all but the 4 youngest units landed
  static_assert(N == 8 || N == 9, "add the immediate");
  if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
}

template <int KW>
__global__ __launch_bounds__(512, 1) void dense_tn_kernel(DwArgs a) {
  typedef DwGeom<KW> G;
  constexpr int KB = KW - 2;                  // k-tiles of a wave's second k-set (the first has 2)
  extern __shared__ __attribute__((aligned(16))) char lds[];   // two steps of G::STEP bytes

  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wid >> 2, wc = wid & 3;      // wr: 128-wide n half of the tile, wc: k quarter (64 | 80 columns)
  const bool hi = wr != 0;
  const int fr = lane & 15, kg = lane >> 4;

  typedef __attribute__((ext_vector_type(8))) short s16x8;
  // buffer descriptors (raw, 32-bit offsets; rows past M read as zero) for the inline-asm LDS-DMA below
  typedef __attribute__((ext_vector_type(4))) int i32x4;
  auto make_rs = [](const void* base, int64_t bytes) {
    const uint64_t q = (uint64_t)base;
    return i32x4{(int)(uint32_t)q, (int)(uint32_t)((q >> 32) & 0xFFFF), (int)bytes, 0x27000};
  };
  // LDS-DMA as inline asm: hipcc tracks the buffer_load_lds BUILTIN as a pending LDS write and drains the whole ring with
  // `s_waitcnt vmcnt(0)` in front of every transposing read (that is what held this kernel at 0.45-0.61 PFLOP/s in
  // round 3's first attempt; see dma16_to_lds in octic_common.hpp)
  auto dma16 = [](unsigned lds_dst, unsigned voff, int soff, const i32x4 rs) {
    lds_dst = __builtin_amdgcn_readfirstlane(lds_dst);
    unsigned keep;   // M0 saved / restored inside the statement (octic_common.hpp: dma16_to_lds)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_dst), "v"(voff), "s"(rs), "s"(soff) : "memory");
  };
  const unsigned lds0 = lds_offset(lds);

  // ---- this workgroup's (slab, tile) item.  Items of a slab are consecutive; item j of a slab runs on XCD j % 8
  // (workgroups are dealt round-robin over the XCDs), which owns a contiguous chunk of the tn-major tile list: the
  // tiles of one dY column panel (same tn) share an L2, and a tile's S partial slabs are written and reduced there.
#if DW_MAP == 1
  // Items are numbered slab-major, tiles tn-major inside a slab.  Workgroups are dealt round-robin over the 8 XCDs and
  // 32 of them are resident per XCD: XCD x takes items [32 (8 r + x), + 32) in round r - a compact patch of ~6 dY
  // column panels x all X panels of ONE slab, so an XCD fetches ~11 operand panels per step for 32 tiles (0.36 panel
  // fetches per tile-step; spreading a slab over all XCDs costs 0.6 and made the kernel wait on the fabric).
  int slab_i, tile;
  {
    const int x = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int item = ((slot >> 5) * 8 + x) * 32 + (slot & 31);
    if (item >= a.tiles * a.S) return;                       // padding item
    slab_i = item / a.tiles;
    tile = item - slab_i * a.tiles;
  }
#else
  const int slab_i = blockIdx.x / a.tiles8;
  int tile;
  {
    const int j = blockIdx.x - slab_i * a.tiles8;
    const int x = j & 7, l = j >> 3, q8 = a.tiles >> 3, r8 = a.tiles & 7;
    if (l >= q8 + (x < r8 ? 1 : 0)) return;                  // padding item
    tile = (x < r8 ? x * (q8 + 1) : r8 * (q8 + 1) + (x - r8) * q8) + l;
  }
#endif
  // which problem (wave-uniform selects; the tile list is [problem 0 | problem 1])
  const bool second = a.tiles0 > 0 && tile >= a.tiles0;
  const bf16* const Yp = second ? a.Y1 : a.Y;
  const bf16* const Xp = second ? a.X1 : a.X;
  const int64_t ldy_ = second ? a.ldy1 : a.ldy, ldx_ = second ? a.ldx1 : a.ldx;
  const int N_ = second ? a.N1 : a.N;
  float* const W_ = second ? a.W1 : a.W;
  const int tile_l = second ? tile - a.tiles0 : tile;
  const i32x4 rsY = make_rs(Yp, (int64_t)a.M * ldy_ * 2), rsX = make_rs(Xp, (int64_t)a.M * ldx_ * 2);

  // ---- DMA lane constants.  A wave-instruction fills 4 unit rows (256 B each): lane -> row (lane >> 4), 16-byte chunk
  // position (lane & 15) = 32-byte slot (lane >> 1) & 7, half (lane & 1); the slot holds source slot ^ f(row), with
  // f(row) = (row & 3) | ((row >> 3) & 1) << 2 and row = 8 * wid + 4 * j + (lane >> 4) for instruction j of the wave.
  // Unit column c (0..127) of dY-units = tile column (c >> 6) * 128 + (c & 63) (+64 for the second halves); of X-units =
  // (c >> 5) * WK + (c & 31) (+32): the first two 16-column k-tiles of every wave, then the next two.
  // KW = 5: the fifth k-tile of the four waves is a third X piece of 64 rows x 64 columns (128-byte rows, 8 per
  // wave-instruction: lane -> row (lane >> 3), slot (lane >> 1) & 3, f(row) = ((row >> 1) & 1) | ((row >> 3) & 1) << 1),
  // column c (0..63) = tile column (c >> 4) * 80 + 64 + (c & 15).
  const int drow = lane >> 4;                 // 0..3
  const int dpos = lane & 15;
  unsigned voY[2], voX[2], voX2 = 0;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = 8 * wid + 4 * j + drow;                         // unit row 0..63
    const int f = (row & 3) | (((row >> 3) & 1) << 2);
    const int src_slot = ((dpos >> 1) & 7) ^ f;
    const int c = src_slot * 16 + (dpos & 1) * 8;                    // first unit column of this lane's 8 bf16
    voY[j] = (unsigned)(((int64_t)row * ldy_ + (c >> 6) * 128 + (c & 63)) * 2);
    voX[j] = (unsigned)(((int64_t)row * ldx_ + (c >> 5) * G::WK + (c & 31)) * 2);
  }
  if constexpr (KW == 5) {
    const int row = 8 * wid + (lane >> 3);
    const int f = ((row >> 1) & 1) | (((row >> 3) & 1) << 1);
    const int src_slot = ((lane >> 1) & 3) ^ f;
    const int c = src_slot * 16 + (lane & 1) * 8;
    voX2 = (unsigned)(((int64_t)row * ldx_ + (c >> 4) * 80 + 64 + (c & 15)) * 2);
  }

  // ---- fragment read constants (transposing reads): lane fr = 4 q + p of a 16-lane group addresses row q, columns
  // 4 p .. 4 p + 3 of a 4 x 16 block and receives column fr of its 4 rows
  const int frow = kg * 8 + (fr >> 2);
  const int ff = ((fr >> 2) & 3) | ((kg & 1) << 2);                  // f(row) for rows frow (+4) (+32 ks)
  int offY[4], offX[2], offX2 = 0;                                   // byte offsets inside a unit, k-step 0, "lo" rows
#pragma unroll
  for (int j = 0; j < 4; ++j) offY[j] = frow * 256 + (((wr * 4 + j) ^ ff) << 5) + (fr & 3) * 8;
#pragma unroll
  for (int i = 0; i < 2; ++i) offX[i] = frow * 256 + (((wc * 2 + i) ^ ff) << 5) + (fr & 3) * 8;
  if constexpr (KW == 5) offX2 = frow * 128 + ((wc ^ (((fr >> 3) & 1) | ((kg & 1) << 1))) << 5) + (fr & 3) * 8;

  {
    const int s0 = (int)((int64_t)a.steps * slab_i / a.S);
    const int s1 = (int)((int64_t)a.steps * (slab_i + 1) / a.S);
    const int tn = tile_l / a.tiles_k, tk = tile_l - tn * a.tiles_k;
    const int n0 = tn * DW_T, k0 = tk * G::BK;
    const int nkt = s1 - s0;                                 // >= 1 (the launcher keeps S <= max(1, steps / 2))
    const int nunits = 4 * nkt;

    f32x4 acc[2][KW][4];                 // [n-half][k-tile][n-tile]   (k-tiles 0, 1 = first k-set, 2 .. KW-1 = second)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int q = 0; q < KW; ++q)
#pragma unroll
        for (int p = 0; p < 4; ++p) acc[i][q][p] = f32x4{0, 0, 0, 0};

    // scalar offsets of this segment: first token row, tile columns
    const int sbY = (int)(((int64_t)s0 * DW_BR * ldy_ + n0) * 2);
    const int sbX = (int)(((int64_t)s0 * DW_BR * ldx_ + k0) * 2);
    const int stepY = (int)(ldy_ * DW_BR * 2), stepX = (int)(ldx_ * DW_BR * 2);

    int u_issue = 0;
    // KIND 0 / 3: first / second 64-column halves of the dY tile halves; KIND 1 / 2: first / second k-sets of X
    auto issue_unit = [&](auto kind_c) {
      constexpr int KIND = decltype(kind_c)::value;
      constexpr bool isY = KIND == 0 || KIND == 3;
      const unsigned dst = lds0 + ((u_issue >> 2) & 1) * G::STEP + G::unit_off(KIND) + wid * 2048;
      const int t = u_issue >> 2;
      if constexpr (isY) {
        const int so = sbY + t * stepY + (KIND == 3 ? 128 : 0);
        dma16(dst, voY[0], so, rsY);
        dma16(dst + 1024, voY[1], so, rsY);
      } else {
        const int so = sbX + t * stepX + (KIND == 2 ? 64 : 0);
        dma16(dst, voX[0], so, rsX);
        dma16(dst + 1024, voX[1], so, rsX);
        if constexpr (KIND == 2 && KW == 5)
          dma16(lds0 + ((u_issue >> 2) & 1) * G::STEP + G::unit_off(2) + DW_UNIT + wid * 1024, voX2, sbX + t * stepX, rsX);
      }
      ++u_issue;
    };
#define DW_IC(v) std::integral_constant<int, v>()

    bf16x8 Yf[2][4];                     // [k-step][n-tile]   (the n half in use)
    bf16x8 XfA[2][2];                    // first k-set:  [k-step][k-tile]
    bf16x8 XfB[2][KB];                   // second k-set
    auto tr8s = [&](const char* p, int hi_off) {
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)p);
      const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p + hi_off));
      const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi4[0], hi4[1], hi4[2], hi4[3]};
      return __builtin_bit_cast(bf16x8, v);
    };
    auto tr8 = [&](const char* p) { return tr8s(p, 4 * 256); };
    auto unit_base = [&](int unit) { return lds + ((unit >> 2) & 1) * G::STEP; };   // + G::unit_off(kind)
    auto readY = [&](int unit, auto kind_c) {
      const char* base = unit_base(unit) + G::unit_off(decltype(kind_c)::value);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 4; ++j) Yf[ks][j] = tr8(base + ks * (32 * 256) + offY[j]);
    };
    auto readXA = [&](int unit) {
      const char* base = unit_base(unit) + G::unit_off(1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 2; ++i) XfA[ks][i] = tr8(base + ks * (32 * 256) + offX[i]);
    };
    auto readXB = [&](int unit) {
      const char* base = unit_base(unit) + G::unit_off(2);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int i = 0; i < 2; ++i) XfB[ks][i] = tr8(base + ks * (32 * 256) + offX[i]);
        if constexpr (KW == 5) XfB[ks][2] = tr8s(base + DW_UNIT + ks * (32 * 128) + offX2, 4 * 128);
      }
    };
    auto mma = [&](int nh, auto set_c) {
      constexpr int SET = decltype(set_c)::value;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if constexpr (SET == 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
              acc[nh][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(XfA[ks][i], Yf[ks][j], acc[nh][i][j], 0, 0, 0);
          } else {
#pragma unroll
            for (int i = 0; i < KB; ++i)
              acc[nh][2 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(XfB[ks][i], Yf[ks][j], acc[nh][2 + i][j], 0, 0, 0);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    };

    int g = 0;
    auto younger = [&](int units) {          // DMA instructions of this wave in the `units` youngest issued units
      int n = 0;
      for (int v = u_issue - units; v < u_issue; ++v) n += (KW == 5 && (v & 3) == 2) ? 3 : 2;
      return n;
    };
    auto wait_landed = [&]() {
      int need = g + 2;
      need = need < nunits - 1 ? need : nunits - 1;
      const int ok = (u_issue - 1) - need;
      if (ok == DW_D - 2) dw_wait_steady<G::INFLIGHT>();
      else dw_wait_vmcnt(ok > 0 ? younger(ok) : 0);
    };

    // ---- prologue (see csrc/dense_gemm.hip for the protocol)
#define DW_PRO(i) if (u_issue < nunits) issue_unit(DW_IC((i) & 3));
    DW_PRO(0) DW_PRO(1) DW_PRO(2) DW_PRO(3)
    if constexpr (DW_D > 4) { DW_PRO(4) DW_PRO(5) }
    if constexpr (DW_D > 6) { DW_PRO(6) DW_PRO(7) }
#undef DW_PRO
    {
      const int ok = (u_issue - 1) - 1;
      dw_wait_vmcnt(ok > 0 ? younger(ok) : 0);
    }
    if (hi) __builtin_amdgcn_s_barrier();

    auto ktile = [&](int t, auto steady_c) {
      constexpr bool STEADY = decltype(steady_c)::value != 0;
      const int b0 = 4 * t;
      auto dma = [&](auto kind_c) {
        if (STEADY || u_issue < nunits) issue_unit(kind_c);
      };
      auto landed = [&]() {
        if constexpr (STEADY) dw_wait_steady<G::INFLIGHT>();
        else wait_landed();
      };
      // phase 0: first n half x first k half
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_s_setprio(2);
      dma(DW_IC((0 + DW_D) & 3));
      readY(b0, DW_IC(0));
      readXA(b0 + 1);
      __builtin_amdgcn_s_setprio(0);
      if (hi) landed();
      __builtin_amdgcn_s_barrier();
      mma(0, DW_IC(0));
      if (!hi) landed();
      ++g;
      // phase 1: first n half x second k half
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_s_setprio(2);
      dma(DW_IC((1 + DW_D) & 3));
      readXB(b0 + 2);
      __builtin_amdgcn_s_setprio(0);
      if (hi) landed();
      __builtin_amdgcn_s_barrier();
      mma(0, DW_IC(1));
      if (!hi) landed();
      ++g;
      // phase 2: second n half x second k half
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_s_setprio(2);
      dma(DW_IC((2 + DW_D) & 3));
      readY(b0 + 3, DW_IC(3));
      __builtin_amdgcn_s_setprio(0);
      if (hi) landed();
      __builtin_amdgcn_s_barrier();
      mma(1, DW_IC(1));
      if (!hi) landed();
      ++g;
      // phase 3: second n half x first k half
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_s_setprio(2);
      dma(DW_IC((3 + DW_D) & 3));
      __builtin_amdgcn_s_setprio(0);
      if (hi) landed();
      __builtin_amdgcn_s_barrier();
      mma(1, DW_IC(0));
      if (!hi) landed();
      ++g;
    };
    const int t_steady = (nunits - 4 - DW_D) >= 0 ? (nunits - 4 - DW_D) / 4 + 1 : 0;
    int t = 0;
#pragma unroll 1
    for (; t < t_steady; ++t) ktile(t, DW_IC(1));
#pragma unroll 1
    for (; t < nkt; ++t) ktile(t, DW_IC(0));
    if (!hi) __builtin_amdgcn_s_barrier();   // re-align the two groups
    __builtin_amdgcn_s_barrier();            // the ring is idle

    // ---- partial tiles.  Ticket FIRST (one word per tile: arrivals in the low half, completed publications in the high
    // half): the first S - 1 workgroups to arrive publish their f32 partial (plain stores -> drain -> barrier -> agent
    // release -> count) and leave; the LAST arriver publishes nothing - it waits until the others' partials are complete
    // (they all hold a ticket, i.e. they are running: the wait cannot deadlock), then sums the S partials in slab order with
    // its own taken from the accumulator registers, half a tile at a time (the running sum lives in the registers the
    // operand fragments used).  Slab traffic is (S - 1) / S of "everyone publishes, the last arriver re-reads all S"
    // (tools/ab_tn_tile.py --slabs: the round trip was 52 us of a 219 us launch at S = 2, in proportion to the bytes).
    if (a.S == 1) {
#pragma unroll
      for (int nh = 0; nh < 2; ++nh)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int n = n0 + wr * 128 + nh * 64 + j * 16 + fr;
#pragma unroll
          for (int q = 0; q < KW; ++q) {
            const int k = k0 + wc * G::WK + q * 16 + kg * 4;
            if (n < N_ && k < a.K) *(f32x4*)(W_ + (int64_t)n * a.K + k) = acc[nh][q][j];
          }
        }
    } else {
      int* flag = (int*)lds;
      if (threadIdx.x == 0) {
        const unsigned old = (unsigned)__hip_atomic_fetch_add(a.tickets + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        flag[0] = ((int)(old & 0xFFFFu) == a.S - 1) ? 1 : 0;
      }
      __syncthreads();
      const bool last = flag[0] != 0;
      if (!last) {
        float* slab = a.slabs + ((int64_t)tile * a.S + slab_i) * (DW_T * G::BK);
        f32x4* sw4 = (f32x4*)slab + (int64_t)wid * (8 * KW) * 64 + lane;
#pragma unroll
        for (int nh = 0; nh < 2; ++nh)
#pragma unroll
          for (int q = 0; q < KW; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) sw4[((nh * KW + q) * 4 + j) * 64] = acc[nh][q][j];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __hip_atomic_fetch_add(a.tickets + tile, 0x10000, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      } else {
        if (threadIdx.x == 0) {
          while (((unsigned)__hip_atomic_load(a.tickets + tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 16) != (unsigned)(a.S - 1))
            __builtin_amdgcn_s_sleep(8);
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __hip_atomic_store(a.tickets + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm for the next launch
        }
        __syncthreads();
#pragma unroll
        for (int nh = 0; nh < 2; ++nh)
#pragma unroll
          for (int jh = 0; jh < 2; ++jh) {                 // a quarter of the wave's tile at a time: 2 n-tiles x KW k-tiles
            f32x4 t[KW][2];
            for (int w = 0; w < a.S; ++w) {
              if (w == slab_i) {
#pragma unroll
                for (int q = 0; q < KW; ++q)
#pragma unroll
                  for (int j = 0; j < 2; ++j) t[q][j] = w == 0 ? acc[nh][q][jh * 2 + j] : t[q][j] + acc[nh][q][jh * 2 + j];
              } else {
                const f32x4* o4 = (const f32x4*)(a.slabs + ((int64_t)tile * a.S + w) * (DW_T * G::BK)) + (int64_t)wid * (8 * KW) * 64 + lane;
#pragma unroll
                for (int q = 0; q < KW; ++q)
#pragma unroll
                  for (int j = 0; j < 2; ++j) {
                    const f32x4 o = o4[((nh * KW + q) * 4 + jh * 2 + j) * 64];
                    t[q][j] = w == 0 ? o : t[q][j] + o;
                  }
              }
            }
            // lane (fr, kg) of MFMA tile (n-tile j, k-tile q) holds dW[n = .. + fr][k = .. + 4 kg .. + 3]
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const int n = n0 + wr * 128 + nh * 64 + (jh * 2 + j) * 16 + fr;
#pragma unroll
              for (int q = 0; q < KW; ++q) {
                const int k = k0 + wc * G::WK + q * 16 + kg * 4;
                if (n < N_ && k < a.K) *(f32x4*)(W_ + (int64_t)n * a.K + k) = t[q][j];
              }
            }
          }
      }
    }
  }
}

}  // namespace octic

using namespace octic;

template <int KW>
static int dw_launch(DwArgs& a, hipStream_t s) {
  const int smem = DwGeom<KW>::RING;
  static DeviceOnce once;
  if (once.first()) {
    (void)hipFuncSetAttribute((const void*)dense_tn_kernel<KW>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    (void)hipGetLastError();
  }
#if DW_MAP == 1
  dense_tn_kernel<KW><<<(a.tiles * a.S + 255) / 256 * 256, 512, smem, s>>>(a);
#else
  dense_tn_kernel<KW><<<a.tiles8 * a.S, 512, smem, s>>>(a);
#endif
  return launch_status();
}

extern "C" {

// Row slabs: the items (tiles8 x S) run in ceil(items / CUs) rounds of ceil(steps / S) reduction steps each; every item
// also pays a fixed prologue + slab epilogue (about 8 steps' worth).  Pick the S with the shortest estimate.
// routing overrides (octic_route_override): OCTIC_ROUTE_WGRAD_SLABS forces the number of row slabs, OCTIC_ROUTE_WGRAD_TILE
// (256 | 320) the tile width where the shape allows it; 0 = automatic
static int dw_slabs(int tiles, int steps) {
  if (route(OCTIC_ROUTE_WGRAD_SLABS) > 0) return route(OCTIC_ROUTE_WGRAD_SLABS) <= steps / 2 ? route(OCTIC_ROUTE_WGRAD_SLABS) : (steps / 2 > 0 ? steps / 2 : 1);
  const int cus = device_cus();
  const int tiles8 = (tiles + 7) / 8 * 8;
  const int per_slab = DW_MAP == 1 ? tiles : tiles8;
  // One round of workgroups whenever the tiles fit: the most slabs that still give every item its own CU (measured on
  // MI355X: 5120x1280 -> 2, 3840x1280 -> 3, 1280x1280 -> 8; a second round never paid for its slab traffic and ramp-up).
  if (per_slab <= cus) {
    int S = cus / per_slab;
    S = S > 16 ? 16 : S;
    S = S > steps / 2 ? steps / 2 : S;
    return S < 1 ? 1 : S;
  }
  int best = 1;
  double best_cost = 1e30;
  for (int S = 1; S <= 16 && S <= steps / 2; ++S) {
    const int rounds = (per_slab * S + cus - 1) / cus;
    const double cost = rounds * ((steps + S - 1) / S + (S > 1 ? 8.0 : 2.0)) * (1.0 + 0.25 * (rounds - 1));
    if (cost < best_cost - 1e-9) { best_cost = cost; best = S; }
  }
  return best;
}

// Tile width along k.  The 320-wide tile was built for occupancy (at M = 16 448: 5120 x 1280 and 1280 x 5120 are 100 tiles
// x 2 slabs = 200 workgroups at 256 wide, 80 x 3 = 240 at 320) and measured NO faster on any ViT-H shape
// (tools/ab_tn_tile.py, same process, us at 256 | 320: 5120 x 1280 200-213 | 198-207, 1280 x 5120 210-214 | 209-212,
// 3840 x 1280 154 | 169, 1280 x 1280 90 | 109): with 240 instead of 200 CUs busy every step gets slower - the launch is
// bound by what the whole chip sustains (~1.05 PFLOP/s for this kernel), not by idle CUs.  So the 256-wide tile is used
// wherever K allows it; the 320-wide one serves K % 320 == 0 && K % 256 != 0 (and the developer knob).
struct DwPlan { int kw, tiles_k, tiles, S; };
static DwPlan dw_plan(int M, int N, int K) {
  const int steps = (M + DW_BR - 1) / DW_BR;
  auto make = [&](int kw) {
    DwPlan p;
    p.kw = kw;
    p.tiles_k = K / (64 * kw);
    p.tiles = (N / DW_T) * p.tiles_k;
    p.S = dw_slabs(p.tiles, steps);
    return p;
  };
  const bool ok4 = K % 256 == 0, ok5 = K % 320 == 0;          // (the entry points refuse K that fits neither)
  if (ok5 && (!ok4 || route(OCTIC_ROUTE_WGRAD_TILE) == 320)) return make(5);
  return make(4);
}

int octic_dense_wgrad_tile(int M, int N, int K) {      // tile width the launch uses (256 | 320)
  if (M <= 0 || N <= 0 || K <= 0 || (N % DW_T) || ((K % 256) && (K % 320))) return 0;
  return dw_plan(M, N, K).kw * 64;
}

int64_t octic_dense_wgrad_workspace_bytes(int M, int N, int K) {
  // room for either width and any forced slab count: [4 KiB tickets (<= 1024 tiles) | slabs]
  const int steps = (M + DW_BR - 1) / DW_BR;
  int64_t need = 0;
  for (int kw = 4; kw <= 5; ++kw) {
    if (K % (64 * kw)) continue;
    const int tiles = (N / DW_T) * (K / (64 * kw));
    const int S = route(OCTIC_ROUTE_WGRAD_SLABS) > 0 ? 16 : dw_slabs(tiles, steps);
    const int64_t b = (int64_t)tiles * S * (DW_T * 64 * kw) * 4;
    need = b > need ? b : need;
  }
  return need + 4096;
}

int octic_dense_wgrad_tn(const void* dY, const void* X, int M, int N, int K, int64_t ldy, int64_t ldx, float* dW,
                         void* workspace, void* stream) {
  if (!dY || !X || !dW || !workspace) return OCTIC_ENULL;
  if (M <= 0 || N <= 0 || K <= 0 || (N % DW_T) || ((K % 256) && (K % 320)) || (ldy % 8) || (ldx % 8)) return OCTIC_ESHAPE;
  if ((int64_t)M * ldy * 2 >= (1ll << 31) || (int64_t)M * ldx * 2 >= (1ll << 31)) return OCTIC_ESHAPE;   // 32-bit buffer offsets
  if ((((uintptr_t)dY) | ((uintptr_t)X) | ((uintptr_t)dW)) & 15) return OCTIC_EALIGN;
  const DwPlan pl = dw_plan(M, N, K);
  if (pl.tiles > 1024) return OCTIC_ESHAPE;                   // ticket region
  DwArgs a = {};
  a.Y = (const bf16*)dY; a.X = (const bf16*)X; a.ldy = ldy; a.ldx = ldx; a.M = M; a.N = N; a.K = K;
  a.W = dW;
  a.tiles_k = pl.tiles_k;
  a.tiles = pl.tiles;
  a.tiles8 = (a.tiles + 7) / 8 * 8;
  a.steps = (M + DW_BR - 1) / DW_BR;
  a.S = pl.S;
  a.tickets = (int*)workspace;
  char* p = (char*)workspace + 4096;         // fixed ticket region: a workspace shared by several shapes keeps its zeros
  a.slabs = (float*)p;
  hipStream_t s = (hipStream_t)stream;
  return pl.kw == 5 ? dw_launch<5>(a, s) : dw_launch<4>(a, s);
}


// Two weight gradients with the same token rows M and the same K as ONE launch: dW0[N0,K] = dY0^T X0, dW1[N1,K] = dY1^T X1.
// The tile list is [tiles of problem 0 | tiles of problem 1], row slabs chosen for the sum: the qkv and proj weight gradients
// of a standard block (3840 x 1280 and 1280 x 1280 at ViT-H) become one 100-tile, two-slab launch - the shape of an MLP
// weight gradient - instead of a 75-tile one and a 25-tile one that needs eight slabs to fill the chip.
int64_t octic_dense_wgrad_pair_workspace_bytes(int M, int N0, int N1, int K) {
  return octic_dense_wgrad_workspace_bytes(M, N0 + N1, K);
}

int octic_dense_wgrad_tn_pair(const void* dY0, const void* X0, int N0, int64_t ldy0, int64_t ldx0, float* dW0,
                              const void* dY1, const void* X1, int N1, int64_t ldy1, int64_t ldx1, float* dW1, int M, int K,
                              void* workspace, void* stream) {
  if (!dY0 || !X0 || !dW0 || !dY1 || !X1 || !dW1 || !workspace) return OCTIC_ENULL;
  if (M <= 0 || N0 <= 0 || N1 <= 0 || K <= 0 || (N0 % DW_T) || (N1 % DW_T) || ((K % 256) && (K % 320)) || (ldy0 % 8) ||
      (ldx0 % 8) || (ldy1 % 8) || (ldx1 % 8))
    return OCTIC_ESHAPE;
  const int64_t lim = 1ll << 31;
  if ((int64_t)M * ldy0 * 2 >= lim || (int64_t)M * ldx0 * 2 >= lim || (int64_t)M * ldy1 * 2 >= lim || (int64_t)M * ldx1 * 2 >= lim)
    return OCTIC_ESHAPE;
  if ((((uintptr_t)dY0) | ((uintptr_t)X0) | ((uintptr_t)dW0) | ((uintptr_t)dY1) | ((uintptr_t)X1) | ((uintptr_t)dW1)) & 15) return OCTIC_EALIGN;
  const DwPlan pl = dw_plan(M, N0 + N1, K);                   // tiles and slabs of the joint tile list
  if (pl.tiles > 1024) return OCTIC_ESHAPE;
  DwArgs a = {};
  a.Y = (const bf16*)dY0; a.X = (const bf16*)X0; a.ldy = ldy0; a.ldx = ldx0; a.M = M; a.N = N0; a.K = K; a.W = dW0;
  a.Y1 = (const bf16*)dY1; a.X1 = (const bf16*)X1; a.ldy1 = ldy1; a.ldx1 = ldx1; a.N1 = N1; a.W1 = dW1;
  a.tiles_k = pl.tiles_k;
  a.tiles = pl.tiles;
  a.tiles0 = (N0 / DW_T) * pl.tiles_k;
  a.tiles8 = (a.tiles + 7) / 8 * 8;
  a.steps = (M + DW_BR - 1) / DW_BR;
  a.S = pl.S;
  a.tickets = (int*)workspace;
  a.slabs = (float*)((char*)workspace + 4096);
  hipStream_t s = (hipStream_t)stream;
  return pl.kw == 5 ? dw_launch<5>(a, s) : dw_launch<4>(a, s);
}

}  // extern "C"
