// Weight gradient of the irrep-blocked linear for gfx950:  G_g[n,k] = sum_rows dY_g[row,n] X_g[row,k].
//
// The reduction runs over token rows, which are the STRIDED dimension of both operands.  Tiles of
// X [rows x k-cols] and dY [rows x n-cols] are staged row-major in LDS exactly as they lie in HBM
// (coalesced 16-byte loads), and the MFMA operands (which need 8 consecutive reduction indices per
// lane) are fetched with the CDNA4 transposing LDS read ds_read_b64_tr_b16 — no transposed copy of
// the activations is ever materialised.  f32 uses v_mfma_f32_16x16x4_f32 whose operands are single
// floats, read with ds_read_b32.
// D[i=k][j=n] orientation: a lane ends with 4 consecutive k of one n -> 16-byte stores into the
// [N,K] (nn.Linear) layout.  The row range is split over `splits` workgroup sets writing f32 slabs;
// the finish kernel sums the slabs in a fixed order (bitwise reproducible), applies the layer-scale
// chain rule (see octic_hip.h) and writes dW / dcs / dbias.
#include <stdlib.h>
#include "octic_common.hpp"

namespace octic {

struct WgGroup {
  const char* x;   // [rows, K] rows
  int64_t x_ld;
  const char* dy;  // [rows, N] rows
  int64_t dy_ld;
  int64_t rows;
  int K, N;
  int pair;
  int k_tiles, n_tiles;
  int wg_begin;      // first logical workgroup id of this group
  int splits;        // row splits of this group (the two-dimensional irrep has twice the rows and gets twice the splits)
  int64_t slab_off;  // element offset of this group's [N,K] block inside a slab
  float* colsum;     // ring kernel only: [splits][N] partial column sums of dY (bias gradient), or null
};
struct WgArgs {
  WgGroup g[5];
  int ngroups;
  int wgs;         // workgroups in all
  int64_t slab_elems;
  float* slabs;
};

template <typename T> struct WElem;
template <> struct WElem<float> { static constexpr int EPC = 4; static constexpr int BMR = 32; };
template <> struct WElem<bf16> { static constexpr int EPC = 8; static constexpr int BMR = 64; };

template <typename TIN, int TT>
__global__ __launch_bounds__(256) void wgrad_kernel(WgArgs args) {
  constexpr int EPC = WElem<TIN>::EPC;
  constexpr int BMR = WElem<TIN>::BMR;          // reduction rows per LDS tile
  constexpr int BW = 32 * TT;                   // tile width in columns (k and n alike)
  constexpr int CPR = BW / EPC;                 // 16-byte chunks per tile row
  constexpr int RS = BW * (int)sizeof(TIN) + 16;  // LDS row stride (bytes), +16 B pad
  constexpr int TILE = BMR * RS;
  extern __shared__ __attribute__((aligned(16))) char lds[];  // 2 stages x (X tile + dY tile)

  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
  const int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  // logical id -> (group, split, tile): tiles of one (group, split) are adjacent so they share the
  // same row panels of X and dY through one XCD's L2.
  int gi = 0;
#pragma unroll
  for (int i = 1; i < 5; ++i)
    if (i < args.ngroups && lid >= args.g[i].wg_begin) gi = i;
  const WgGroup& G = args.g[gi];
  const int gt = G.k_tiles * G.n_tiles;
  const int rel = lid - G.wg_begin;
  const int split = rel / gt, lt = rel - split * gt;
  const int kt = lt / G.n_tiles, nt = lt - kt * G.n_tiles;
  const int k0 = kt * BW, n0 = nt * BW;
  const int K = G.K, N = G.N;
  int64_t chunk = (G.rows + G.splits - 1) / G.splits;
  chunk = (chunk + BMR - 1) / BMR * BMR;
  const int64_t r0 = (int64_t)split * chunk;
  const int64_t r1 = r0 + chunk < G.rows ? r0 + chunk : G.rows;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wk = wid & 1, wn = wid >> 1;
  const int fr = lane & 15, kg = lane >> 4;

  f32x4 acc[TT][TT];
#pragma unroll
  for (int i = 0; i < TT; ++i)
#pragma unroll
    for (int j = 0; j < TT; ++j) acc[i][j] = f32x4{0, 0, 0, 0};

  u32x4 rx[TT], ry[TT];
  auto gload = [&](int64_t rbase) {
#pragma unroll
    for (int i = 0; i < TT; ++i) {
      const int q = tid + 256 * i;
      const int row = q / CPR, cc = q - row * CPR;
      const int64_t mm = rbase + row;
      const bool rok = mm < r1;
      const int64_t xo = G.pair ? (mm >> 1) * G.x_ld + (mm & 1) * (int64_t)K : mm * G.x_ld;
      const int64_t yo = G.pair ? (mm >> 1) * G.dy_ld + (mm & 1) * (int64_t)N : mm * G.dy_ld;
      const int kcol = k0 + cc * EPC, ncol = n0 + cc * EPC;
      rx[i] = (rok && kcol < K) ? *(const u32x4*)((const TIN*)G.x + xo + kcol) : u32x4{0, 0, 0, 0};
      ry[i] = (rok && ncol < N) ? *(const u32x4*)((const TIN*)G.dy + yo + ncol) : u32x4{0, 0, 0, 0};
    }
  };
  auto lstore = [&](int stage) {
    char* xs = lds + stage * 2 * TILE;
    char* ys = xs + TILE;
#pragma unroll
    for (int i = 0; i < TT; ++i) {
      const int q = tid + 256 * i;
      const int row = q / CPR, cc = q - row * CPR;
      *(u32x4*)(xs + row * RS + cc * 16) = rx[i];
      *(u32x4*)(ys + row * RS + cc * 16) = ry[i];
    }
  };

  if (r0 < r1) {
    const int64_t nrt = (r1 - r0 + BMR - 1) / BMR;
    gload(r0);
    lstore(0);
    __syncthreads();
    for (int64_t rt = 0; rt < nrt; ++rt) {
      if (rt + 1 < nrt) gload(r0 + (rt + 1) * BMR);
      const char* xs = lds + (int)(rt & 1) * 2 * TILE;
      const char* ys = xs + TILE;
      if constexpr (sizeof(TIN) == 2) {
#pragma unroll
        for (int ks = 0; ks < BMR / 32; ++ks) {
          bf16x8 af[TT], bfr[TT];
          // transposing reads: lane fr=4q+p of a 16-lane group addresses row q, columns 4p..4p+3 of a
          // 4x16 block and receives column fr of its 4 rows (cdna_hip_programming.md T10)
          const int rbase = ks * 32 + kg * 8 + (fr >> 2);
          const int cb = (fr & 3) * 4;
#pragma unroll
          for (int i = 0; i < TT; ++i) {
            const int col = wk * (TT * 16) + i * 16 + cb;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (s16x4 __attribute__((address_space(3)))*)(xs + rbase * RS + col * 2));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (s16x4 __attribute__((address_space(3)))*)(xs + (rbase + 4) * RS + col * 2));
            typedef __attribute__((ext_vector_type(8))) short s16x8;
            s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            af[i] = __builtin_bit_cast(bf16x8, v);
          }
#pragma unroll
          for (int j = 0; j < TT; ++j) {
            const int col = wn * (TT * 16) + j * 16 + cb;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (s16x4 __attribute__((address_space(3)))*)(ys + rbase * RS + col * 2));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (s16x4 __attribute__((address_space(3)))*)(ys + (rbase + 4) * RS + col * 2));
            typedef __attribute__((ext_vector_type(8))) short s16x8;
            s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            bfr[j] = __builtin_bit_cast(bf16x8, v);
          }
#pragma unroll
          for (int i = 0; i < TT; ++i)
#pragma unroll
            for (int j = 0; j < TT; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int s4 = 0; s4 < BMR / 4; ++s4) {
          float af[TT], bfr[TT];
          const int row = s4 * 4 + kg;
#pragma unroll
          for (int i = 0; i < TT; ++i) af[i] = *(const float*)(xs + row * RS + (wk * (TT * 16) + i * 16 + fr) * 4);
#pragma unroll
          for (int j = 0; j < TT; ++j) bfr[j] = *(const float*)(ys + row * RS + (wn * (TT * 16) + j * 16 + fr) * 4);
#pragma unroll
          for (int i = 0; i < TT; ++i)
#pragma unroll
            for (int j = 0; j < TT; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
      }
      if (rt + 1 < nrt) lstore((int)((rt + 1) & 1));
      __syncthreads();
    }
  }

  // slab store: lane holds k = kb + kg*4 + (0..3) of row n = nb + fr
  float* slab = args.slabs + (int64_t)split * args.slab_elems + G.slab_off;
#pragma unroll
  for (int j = 0; j < TT; ++j) {
    const int n = n0 + wn * (TT * 16) + j * 16 + fr;
    if (n >= N) continue;
#pragma unroll
    for (int i = 0; i < TT; ++i) {
      const int k = k0 + wk * (TT * 16) + i * 16 + kg * 4;
      if (k >= K) continue;
      *(f32x4*)(slab + (int64_t)n * K + k) = acc[i][j];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// bf16, 160 x 160 tiles, K and N multiples of 160 (every shape of the ViT-H block): LDS-DMA ring.
// Three stages of 32 reduction rows; a stage is the X tile and the dY tile as they lie in HBM (32 rows x 320 B each,
// no padding: global_load_lds writes 1 KiB per wave-instruction linearly).  Rows whose index has bit 3 set are stored
// rotated by two 16-byte chunks (the rotation is applied on the SOURCE side: lane l of a DMA instruction fetches the
// chunk that belongs at its LDS position), which makes the transposing reads of rows r and r+8 hit disjoint banks.
// 60 KiB of LDS and <= 128 VGPRs: two workgroups per CU (the register-staged kernel above fits one).
// Row tails: rows >= r1 fetch from a zero page, so they add nothing.
#ifndef OCTIC_WG_STAGES
#define OCTIC_WG_STAGES 3
#endif
constexpr int kWgS = OCTIC_WG_STAGES, kWgR = 32;   // ring stages (tiles in flight behind the one being multiplied: kWgS - 1)
constexpr int kWgTile = kWgR * 320;
constexpr int kWgStage = 2 * kWgTile;
__device__ __attribute__((aligned(256))) unsigned char g_wg_zero[512];

// wait until this wave's share of the oldest tile in flight has landed: `ahead` younger tiles (5 DMA instructions each)
// may stay outstanding
__device__ inline void wg_wait_tiles(int ahead) {
  if (ahead <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if (ahead == 1) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
  else if (ahead == 2) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
}

__global__ __launch_bounds__(256, 2) void wgrad_ring_kernel(WgArgs args) {
  constexpr int TT = 5, BW = 160;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
  const int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  int gi = 0;
#pragma unroll
  for (int i = 1; i < 5; ++i)
    if (i < args.ngroups && lid >= args.g[i].wg_begin) gi = i;
  const WgGroup& G = args.g[gi];
  const int gt = G.k_tiles * G.n_tiles;
  const int rel = lid - G.wg_begin;
  const int split = rel / gt, lt = rel - split * gt;
  const int kt = lt / G.n_tiles, nt = lt - kt * G.n_tiles;
  const int k0 = kt * BW, n0 = nt * BW;
  const int K = G.K, N = G.N;
  int64_t chunk = (G.rows + G.splits - 1) / G.splits;
  chunk = (chunk + kWgR - 1) / kWgR * kWgR;
  const int64_t r0 = (int64_t)split * chunk;
  const int64_t r1 = r0 + chunk < G.rows ? r0 + chunk : G.rows;
  const int nsteps = r0 < r1 ? (int)((r1 - r0 + kWgR - 1) / kWgR) : 0;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wk = wid & 1, wn = wid >> 1;
  const int fr = lane & 15, kg = lane >> 4;

  // ---- DMA descriptors: instruction j of this wave is tile chunk q = wid + 4j (0-9: X, 10-19: dY)
  const char* src[5];
  int64_t inc[5];
  int row_in_tile[5], cbyte[5], ldst[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const int q = wid + 4 * j;
    const bool isY = q >= 10;
    const int qq = isY ? q - 10 : q;
    const int o = qq * 1024 + lane * 16;
    const int row = o / 320;
    const int cpos = (o - row * 320) >> 4;                       // chunk position inside the LDS row
    const int c = ((row & 8) ? cpos + 18 : cpos) % 20;           // the chunk that lives there (rotation by 2)
    const int64_t mm = r0 + row;
    const int64_t ld = isY ? G.dy_ld : G.x_ld;
    const int width = isY ? N : K;
    const int64_t ro = G.pair ? (mm >> 1) * ld + (mm & 1) * (int64_t)width : mm * ld;
    src[j] = (isY ? G.dy : G.x) + (ro + (isY ? n0 : k0)) * 2 + c * 16;
    inc[j] = (G.pair ? 16 : 32) * ld * 2;
    row_in_tile[j] = row;
    cbyte[j] = c * 16;
    ldst[j] = (isY ? kWgTile : 0) + qq * 1024;
  }
  int l_step = 0, l_stage = 0;
  auto issue = [&]() {
    const unsigned st = lds_offset(lds) + l_stage * kWgStage;
    const bool tail = (int64_t)(l_step + 1) * kWgR > r1 - r0;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const char* p = src[j];
      if (tail && r0 + (int64_t)l_step * kWgR + row_in_tile[j] >= r1) p = (const char*)g_wg_zero + cbyte[j];
      dma16_to_lds(st + ldst[j], p);
      src[j] += inc[j];
    }
    l_stage = l_stage == kWgS - 1 ? 0 : l_stage + 1;
    ++l_step;
  };

  f32x4 acc[TT][TT];
#pragma unroll
  for (int i = 0; i < TT; ++i)
#pragma unroll
    for (int j = 0; j < TT; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
  // bias gradient for free: the column sums of dY are one more MFMA row with an all-ones X fragment.  Only the
  // invariant irrep has a bias; its k-tile-0 workgroups (waves wk == 0) carry the extra 5 MFMAs per step.
  const bool do_cs = G.colsum != nullptr && kt == 0 && wk == 0;
  f32x4 accs[TT];
#pragma unroll
  for (int j = 0; j < TT; ++j) accs[j] = f32x4{0, 0, 0, 0};
  const bf16 one = (bf16)1.0f;
  const bf16x8 ones = {one, one, one, one, one, one, one, one};

  // ---- fragment addresses inside a tile (transposing reads: lane fr = 4q+p of a 16-lane group addresses row q,
  // columns 4p..4p+3 of a 4x16 block); rows kg*8 + (fr>>2) (+4): bit 3 of the row = kg & 1 -> rotation
  const int frow = kg * 8 + (fr >> 2);
  const int rot = (kg & 1) * 2;
  int offa[TT], offb[TT];
#pragma unroll
  for (int i = 0; i < TT; ++i) {
    const int ca = wk * 10 + i * 2 + ((fr & 3) >> 1), cb = wn * 10 + i * 2 + ((fr & 3) >> 1);
    offa[i] = frow * 320 + ((ca + rot) % 20) * 16 + (fr & 1) * 8;
    offb[i] = kWgTile + frow * 320 + ((cb + rot) % 20) * 16 + (fr & 1) * 8;
  }

#pragma unroll
  for (int pz = 0; pz < kWgS - 1; ++pz)
    if (pz < nsteps) issue();
  int c_stage = 0;
  for (int sidx = 0; sidx < nsteps; ++sidx) {
    const int left = nsteps - 1 - sidx;      // tiles after this one; kWgS - 2 of them are in flight in steady state
    wg_wait_tiles(left < kWgS - 2 ? left : kWgS - 2);
    __builtin_amdgcn_s_barrier();          // every wave's share of tile sidx has landed; stage of tile sidx-1 is free
    if (sidx + kWgS - 1 < nsteps) issue();
    const char* base = lds + c_stage * kWgStage;
    c_stage = c_stage == kWgS - 1 ? 0 : c_stage + 1;
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    bf16x8 af[TT], bfr[TT];
#pragma unroll
    for (int i = 0; i < TT; ++i) {
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(base + offa[i]));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(base + offa[i] + 4 * 320));
      const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      af[i] = __builtin_bit_cast(bf16x8, v);
    }
#pragma unroll
    for (int j = 0; j < TT; ++j) {
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(base + offb[j]));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(base + offb[j] + 4 * 320));
      const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      bfr[j] = __builtin_bit_cast(bf16x8, v);
    }
#pragma unroll
    for (int i = 0; i < TT; ++i)
#pragma unroll
      for (int j = 0; j < TT; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    if (do_cs) {
#pragma unroll
      for (int j = 0; j < TT; ++j) accs[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, bfr[j], accs[j], 0, 0, 0);
    }
  }

  // slab store: lane holds k = kb + kg*4 + (0..3) of row n = nb + fr
  float* slab = args.slabs + (int64_t)split * args.slab_elems + G.slab_off;
#pragma unroll
  for (int j = 0; j < TT; ++j) {
    const int n = n0 + wn * (TT * 16) + j * 16 + fr;
#pragma unroll
    for (int i = 0; i < TT; ++i) {
      const int k = k0 + wk * (TT * 16) + i * 16 + kg * 4;
      *(f32x4*)(slab + (int64_t)n * K + k) = acc[i][j];
    }
    if (do_cs && kg == 0) G.colsum[(int64_t)split * N + n] = accs[j][0];   // all 16 output rows are equal
  }
}

// One wave per weight row n of a group: G[n,:] = sum_splits slab ; dW = cs[n]*G ; dcs[n] = <W[n,:],G[n,:]> (+ bias term)
struct FinGroup {
  int64_t slab_off;
  int K, N;
  int row_begin;
  int splits;
  const float* w32;
  const float* cs;
  float* dw;
  float* dcs;
};
struct FinArgs {
  FinGroup g[5];
  int ngroups;
  int rows_total;
  int64_t slab_elems;
  const float* slabs;
  const float* bias;
  const float* dysum;
  const float* colsum_part;   // [splits of group 0][N of group 0] partial column sums left by the ring kernel, or null
  float* dbias;
};

__device__ __forceinline__ void wgrad_finish_body(const FinArgs& a) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.rows_total) return;
  int gi = 0;
#pragma unroll
  for (int i = 1; i < 5; ++i)
    if (i < a.ngroups && row >= a.g[i].row_begin) gi = i;
  const FinGroup& G = a.g[gi];
  const int n = row - G.row_begin;
  const float s = G.cs ? G.cs[n] : 1.0f;
  float dot = 0.f;
  for (int k = lane * 4; k < G.K; k += 256) {
    // slabs summed in a fixed order (bitwise reproducible), eight loads in flight at a time
    f32x4 v = {0, 0, 0, 0};
    const float* sp0 = a.slabs + G.slab_off + (int64_t)n * G.K + k;
    int sp = 0;
    for (; sp + 8 <= G.splits; sp += 8) {
      f32x4 t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = *(const f32x4*)(sp0 + (int64_t)(sp + u) * a.slab_elems);
#pragma unroll
      for (int u = 0; u < 8; ++u) v += t[u];
    }
    for (; sp < G.splits; ++sp) v += *(const f32x4*)(sp0 + (int64_t)sp * a.slab_elems);
    if (G.cs) {
      const f32x4 w = *(const f32x4*)(G.w32 + (int64_t)n * G.K + k);
      dot += w[0] * v[0] + w[1] * v[1] + w[2] * v[2] + w[3] * v[3];
    }
    if (G.dw) *(f32x4*)(G.dw + (int64_t)n * G.K + k) = v * s;
  }
  float ysum = 0.f;
  bool have_ysum = false;
  if (gi == 0) {
    if (a.dysum) {
      ysum = a.dysum[n];
      have_ysum = true;
    } else if (a.colsum_part) {
      for (int sp = 0; sp < G.splits; ++sp) ysum += a.colsum_part[(int64_t)sp * G.N + n];
      have_ysum = true;
    }
  }
  if (G.cs) {
    dot = wave_sum(dot);
    if (lane == 0 && G.dcs) {
      if (gi == 0 && a.bias && have_ysum) dot += a.bias[n] * ysum;
      G.dcs[n] = dot;
    }
  }
  if (gi == 0 && lane == 0 && a.dbias && have_ysum) a.dbias[n] = s * ysum;
}

__global__ __launch_bounds__(256) void wgrad_finish_kernel(FinArgs a) { wgrad_finish_body(a); }

// up to 8 finishes in one launch (blockIdx.y = job; see octic_dense_finish_batch in csrc/dense.hip)
struct FinPack {
  FinArgs a[8];
};
__global__ __launch_bounds__(256) void wgrad_finish_batch_kernel(FinPack pack) {
  const FinArgs& a = pack.a[blockIdx.y];
  if ((int)blockIdx.x * 4 >= a.rows_total) return;
  wgrad_finish_body(a);
}

inline int pick_tt(const WgArgs& a) {
  int best = 2;
  double best_cost = 1e30;
  for (int tt = 2; tt <= 5; ++tt) {
    const int bw = 32 * tt;
    double cost = 0;
    for (int i = 0; i < a.ngroups; ++i) {
      const double kt = (a.g[i].K + bw - 1) / bw, nt = (a.g[i].N + bw - 1) / bw;
      cost += kt * nt * bw * bw * (double)a.g[i].rows;
    }
    if (cost <= best_cost * 1.0001) {
      best_cost = cost < best_cost ? cost : best_cost;
      best = tt;
    }
  }
  return best;
}

template <typename TIN, int TT>
int launch_wgrad_tt(WgArgs& a, hipStream_t s) {
  constexpr int BW = 32 * TT;
  int t = 0;
  for (int i = 0; i < a.ngroups; ++i) {
    a.g[i].k_tiles = (a.g[i].K + BW - 1) / BW;
    a.g[i].n_tiles = (a.g[i].N + BW - 1) / BW;
    a.g[i].wg_begin = t;
    t += a.g[i].k_tiles * a.g[i].n_tiles * a.g[i].splits;
  }
  a.wgs = t;
  constexpr size_t smem = (size_t)2 * 2 * WElem<TIN>::BMR * (BW * sizeof(TIN) + 16);
  static DeviceOnce once;
  if (once.first()) {
    (void)hipFuncSetAttribute((const void*)wgrad_kernel<TIN, TT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    (void)hipGetLastError();
  }
  wgrad_kernel<TIN, TT><<<t, 256, smem, s>>>(a);
  return launch_status();
}

inline bool wgrad_ring_ok(const WgArgs& a) {
  for (int i = 0; i < a.ngroups; ++i) {
    const WgGroup& g = a.g[i];
    if ((g.K % 160) || (g.N % 160)) return false;
    if ((((uintptr_t)g.x) | ((uintptr_t)g.dy)) & 15) return false;
  }
  return true;
}

inline int launch_wgrad_ring(WgArgs& a, hipStream_t s) {
  int t = 0;
  for (int i = 0; i < a.ngroups; ++i) {
    a.g[i].k_tiles = a.g[i].K / 160;
    a.g[i].n_tiles = a.g[i].N / 160;
    a.g[i].wg_begin = t;
    t += a.g[i].k_tiles * a.g[i].n_tiles * a.g[i].splits;
  }
  a.wgs = t;
  static DeviceOnce once;
  if (kWgS * kWgStage > 64 * 1024 && once.first()) {
    (void)hipFuncSetAttribute((const void*)wgrad_ring_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kWgS * kWgStage);
    (void)hipGetLastError();
  }
  wgrad_ring_kernel<<<t, 256, kWgS * kWgStage, s>>>(a);
  return launch_status();
}

template <typename TIN>
int launch_wgrad(WgArgs& a, hipStream_t s) {
  if (sizeof(TIN) == 2 && wgrad_ring_ok(a)) return launch_wgrad_ring(a, s);
  switch (pick_tt(a)) {
    case 2: return launch_wgrad_tt<TIN, 2>(a, s);
    case 3: return launch_wgrad_tt<TIN, 3>(a, s);
    case 4: return launch_wgrad_tt<TIN, 4>(a, s);
    default: return launch_wgrad_tt<TIN, 5>(a, s);
  }
}

inline int64_t linear_slab_elems(int cin, int cout) { return (int64_t)8 * cin * cout; }
// `splits` of the C ABI is the split count of the two-dimensional irrep's [2M]-row problem; the one-dimensional
// irreps have half the rows and take half the splits, so every workgroup reduces about the same number of rows
inline int group_splits(int splits, bool isE) { return isE ? splits : (splits + 1) / 2; }

}  // namespace octic

using namespace octic;

extern "C" {

// behind the slabs: [splits][cout] partial column sums of the invariant irrep's dY (ring kernel)
inline int64_t colsum_off_elems(int cin, int cout, int splits) { return linear_slab_elems(cin, cout) * (int64_t)splits; }

int64_t octic_linear_d8_wgrad_workspace_bytes(int cin, int cout, int splits) {
  return (colsum_off_elems(cin, cout, splits) + (int64_t)splits * cout) * 4;
}

int octic_linear_d8_wgrad_has_colsum(int cin, int cout, int dtype) {
  return dtype == OCTIC_BF16 && (cin % 160) == 0 && (cout % 160) == 0;
}

int octic_linear_d8_wgrad_tile(int64_t M, int cin, int cout) {
  WgArgs a = {};
  a.ngroups = 5;
  for (int i = 0; i < 5; ++i) {
    a.g[i].K = i == 0 ? 2 * cin : cin;
    a.g[i].N = i == 0 ? 2 * cout : cout;
    a.g[i].rows = i == 0 ? 2 * M : M;
  }
  return 32 * pick_tt(a);
}

int octic_linear_d8_wgrad_splits(int64_t M, int cin, int cout) {
  // enough row-splits to give every CU two workgroups, bounded so the f32 slabs stay a fraction of the activation
  // bytes the kernel has to read anyway
  WgArgs a = {};
  a.ngroups = 5;
  for (int i = 0; i < 5; ++i) {
    a.g[i].K = i == 0 ? 2 * cin : cin;
    a.g[i].N = i == 0 ? 2 * cout : cout;
    a.g[i].rows = i == 0 ? 2 * M : M;
  }
  const int bw = 32 * pick_tt(a);
  const int tiles_e = ((2 * cin + bw - 1) / bw) * ((2 * cout + bw - 1) / bw);
  const int tiles_1 = 4 * ((cin + bw - 1) / bw) * ((cout + bw - 1) / bw);
#ifndef OCTIC_WG_TARGET
#define OCTIC_WG_TARGET 512.0
#endif
  constexpr double target = OCTIC_WG_TARGET;
  int s = (int)((target / (tiles_e + 0.5 * tiles_1)) + 0.5);
  const int64_t max_by_rows = (2 * M + 255) / 256;
  if (s > max_by_rows) s = (int)max_by_rows;
  if (s > 32) s = 32;
  if (s < 1) s = 1;
  return s;
}

int octic_linear_d8_wgrad(const octic_view* x, const octic_view* dy, int64_t M, int cin, int cout, int dtype,
                          float* workspace, int splits, void* stream) {
  int e;
  if ((e = check_c_dt(cin, dtype)) || (e = check_c_dt(cout, dtype))) return e;
  if ((e = check_view(x, cin, dtype)) || (e = check_view(dy, cout, dtype))) return e;
  if (!workspace) return OCTIC_ENULL;
  if (M <= 0 || splits <= 0) return OCTIC_ESHAPE;
  WgArgs a = {};
  a.ngroups = 5;
  a.slab_elems = linear_slab_elems(cin, cout);
  a.slabs = workspace;
  // slab layout: [W_A1 | W_A2 | W_B1 | W_B2 | W_E]; launch order E first
  for (int gidx = 0; gidx < 5; ++gidx) {
    const int irrep = gidx == 0 ? 4 : gidx - 1;
    const bool isE = irrep == 4;
    WgGroup& g = a.g[gidx];
    g.x = (const char*)x->ptr[irrep];
    g.x_ld = x->ld[irrep];
    g.dy = (const char*)dy->ptr[irrep];
    g.dy_ld = dy->ld[irrep];
    g.rows = isE ? 2 * M : M;
    g.K = isE ? 2 * cin : cin;
    g.N = isE ? 2 * cout : cout;
    g.pair = isE ? 1 : 0;
    g.splits = group_splits(splits, isE);
    g.slab_off = isE ? (int64_t)4 * cin * cout : (int64_t)irrep * cin * cout;
    g.colsum = irrep == 0 ? workspace + colsum_off_elems(cin, cout, splits) : nullptr;
  }
  if (dtype == OCTIC_F32) return launch_wgrad<float>(a, (hipStream_t)stream);
  if (dtype == OCTIC_BF16) return launch_wgrad<bf16>(a, (hipStream_t)stream);
  return OCTIC_EDTYPE;
}

static int fin_args(FinArgs& a, const float* workspace, int splits, int cin, int cout, const float* const w32[5],
                    const float* const cs[5], const float* bias, const float* dysum, float* const dw[5], float* const dcs[5],
                    float* dbias) {
  if (!workspace || !dw) return OCTIC_ENULL;
  if (splits <= 0 || check_c_dt(cin, OCTIC_F32) || check_c_dt(cout, OCTIC_F32)) return OCTIC_ESHAPE;
  if (cs && (!w32 || !dcs)) return OCTIC_ENULL;
  a = FinArgs{};
  a.ngroups = 5;
  a.slab_elems = linear_slab_elems(cin, cout);
  a.slabs = workspace;
  a.bias = bias;
  a.dysum = dysum;
  // no explicit dysum but a bias gradient wanted: the caller vouches (octic_linear_d8_wgrad_has_colsum) that the
  // wgrad launch left the partial column sums behind the slabs
  a.colsum_part = (!dysum && (dbias || (bias && cs))) ? workspace + colsum_off_elems(cin, cout, splits) : nullptr;
  a.dbias = dbias;
  int row = 0;
  for (int irrep = 0; irrep < 5; ++irrep) {  // group 0 must be A1 (bias terms)
    const bool isE = irrep == 4;
    FinGroup& g = a.g[irrep];
    g.K = isE ? 2 * cin : cin;
    g.N = isE ? 2 * cout : cout;
    g.slab_off = isE ? (int64_t)4 * cin * cout : (int64_t)irrep * cin * cout;
    g.row_begin = row;
    g.splits = group_splits(splits, isE);
    row += g.N;
    g.w32 = cs ? w32[irrep] : nullptr;
    g.cs = cs ? cs[irrep] : nullptr;
    g.dw = dw[irrep];
    g.dcs = cs ? dcs[irrep] : nullptr;
    if (cs && (!g.w32 || !g.cs || !g.dcs)) return OCTIC_ENULL;
  }
  a.rows_total = row;
  return OCTIC_OK;
}

int octic_linear_d8_wgrad_finish(const float* workspace, int splits, int cin, int cout, const float* const w32[5],
                                 const float* const cs[5], const float* bias, const float* dysum, float* const dw[5],
                                 float* const dcs[5], float* dbias, void* stream) {
  FinArgs a;
  const int e = fin_args(a, workspace, splits, cin, cout, w32, cs, bias, dysum, dw, dcs, dbias);
  if (e) return e;
  wgrad_finish_kernel<<<(a.rows_total + 3) / 4, 256, 0, (hipStream_t)stream>>>(a);
  return launch_status();
}

int octic_linear_d8_wgrad_finish_batch(const octic_wgrad_finish_job* jobs, int njobs, void* stream) {
  if (!jobs) return OCTIC_ENULL;
  if (njobs < 0) return OCTIC_ESHAPE;
  for (int i0 = 0; i0 < njobs; i0 += 8) {
    const int n = njobs - i0 < 8 ? njobs - i0 : 8;
    FinPack pack;
    int rmax = 0;
    for (int i = 0; i < n; ++i) {
      const octic_wgrad_finish_job& j = jobs[i0 + i];
      const int e = fin_args(pack.a[i], j.workspace, j.splits, j.cin, j.cout, j.has_cs ? j.w32 : nullptr,
                             j.has_cs ? j.cs : nullptr, j.bias, j.dysum, j.dw, j.has_cs ? j.dcs : nullptr, j.dbias);
      if (e) return e;
      rmax = pack.a[i].rows_total > rmax ? pack.a[i].rows_total : rmax;
    }
    wgrad_finish_batch_kernel<<<dim3((rmax + 3) / 4, n), 256, 0, (hipStream_t)stream>>>(pack);
  }
  return launch_status();
}

int64_t octic_lift_wgrad_workspace_bytes(int Kpad, int D, int splits) { return (int64_t)Kpad * D * splits * 4; }

int octic_lift_wgrad(const void* patches, const void* dout, float* dw, float* workspace, int splits, int64_t rows,
                     int Kpad, int D, int dtype, void* stream) {
  if (!patches || !dout || !dw || !workspace) return OCTIC_ENULL;
  if (rows <= 0 || splits <= 0 || Kpad <= 0 || (Kpad % 8) || D <= 0 || (D % 8)) return OCTIC_ESHAPE;
  WgArgs a = {};
  a.ngroups = 1;
  a.slab_elems = (int64_t)Kpad * D;
  a.slabs = workspace;
  WgGroup& g = a.g[0];
  g.x = (const char*)patches;
  g.x_ld = Kpad;
  g.dy = (const char*)dout;
  g.dy_ld = D;
  g.rows = rows;
  g.K = Kpad;
  g.N = D;
  g.pair = 0;
  g.splits = splits;
  g.slab_off = 0;
  int e = dtype == OCTIC_F32 ? launch_wgrad<float>(a, (hipStream_t)stream)
                             : (dtype == OCTIC_BF16 ? launch_wgrad<bf16>(a, (hipStream_t)stream) : OCTIC_EDTYPE);
  if (e) return e;
  FinArgs f = {};
  f.ngroups = 1;
  f.slab_elems = a.slab_elems;
  f.slabs = workspace;
  f.g[0].K = Kpad;
  f.g[0].N = D;
  f.g[0].slab_off = 0;
  f.g[0].row_begin = 0;
  f.g[0].splits = splits;
  f.g[0].dw = dw;
  f.rows_total = D;
  wgrad_finish_kernel<<<(D + 3) / 4, 256, 0, (hipStream_t)stream>>>(f);
  return launch_status();
}

}  // extern "C"
