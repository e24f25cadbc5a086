// Shared pieces of the persistent head_dim-80 attention kernels (csrc/attn80.hip: forward; csrc/attn80_bwd.hip: the
// single-pass backward): 32-row LDS tile images filled by LDS-DMA, their swizzle, the per-lane fragment addresses of
// the row reads (ds_read_b128) and transposing reads (ds_read_b64_tr_b16), the stagers for plain and packed rows.
#pragma once
#include "attn_common.hpp"

namespace octic {
namespace a80 {

constexpr int KS = 5, DT = 3, HD = 80;
constexpr int TILE_B = 5120, TAIL_OFF = 4096;          // bytes per 32-row tile: 32 x 128 (main) + 32 x 32 (tail)
constexpr int MAXNT = 9, WAVES = 8, GROUP = 3;         // tiles per release group
constexpr unsigned OOR = 0x7FFFFFF0u;                  // beyond every descriptor: the DMA writes zeros
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

// chunk swizzle of row r (16-byte chunks 0..7 of the 128-byte main part).  With t = (r >> 1) & 7:
//   * rows of one parity take 8 different values -> the 16 rows a ds_read_b128 lane group touches hit 16 different
//     16-byte bank slots;
//   * bit 2 flips between row pairs (r, r+1) and (r+2, r+3) -> the 4 rows x 64 bytes of a transposing read cover the
//     four 64-byte quarters of the 256-byte bank row.
__device__ __forceinline__ int swz(int r) { const int t = (r >> 1) & 7; return ((t & 1) << 2) | (t >> 1); }

#ifndef A80_DMA_POLICY
#define A80_DMA_POLICY ""          // cache policy of the LDS-DMA loads (developer A/B: " sc1", " nt", " sc0 sc1")
#endif
__device__ __forceinline__ void dma16(unsigned lds_dst, unsigned vo, const i32x4 rs) {
  unsigned keep;   // M0 saved / restored inside the statement (octic_common.hpp: dma16_to_lds)
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 4\n\tbuffer_load_dwordx4 %2, %3, 0 offen" A80_DMA_POLICY " lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(lds_dst), "v"(vo), "s"(rs) : "memory");
}
__device__ __forceinline__ void dma4(unsigned lds_dst, unsigned vo, const i32x4 rs) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 4\n\tbuffer_load_dword %2, %3, 0 offen" A80_DMA_POLICY " lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(lds_dst), "v"(vo), "s"(rs) : "memory");
}

// descriptor of one (batch element, head) of a tensor: rows 0..T-1, everything past the last row reads as zero
__device__ __forceinline__ i32x4 make_rs(const bf16* base, int64_t off, int64_t sT, int T, int cv, int drop = 0) {
  const uint64_t p = (uint64_t)(base + off);
  const int rec = drop ? 0 : (int)((T - 1) * sT * 2 + (cv ? 16 * cv : HD * 2));
  return i32x4{(int)(uint32_t)p, (int)(uint32_t)((p >> 32) & 0xFFFF), rec, 0x27000};
}

// ---- staging: who fills what ----------------------------------------------------------------------------------------
// A tile of an image takes 5 wave-instructions from plain rows (4 x 8 rows x 128 B + one of 32 rows x 32 B) and 8 from
// packed rows (the tail's 8 dwords per row are 4-byte pieces of six different irrep pieces).  The 10 / 16 jobs of a
// tile step (two images) are dealt to the 8 waves: wave w runs jobs w and w + 8.
struct Stager {
  // rows of `tile` of both images of one head -> LDS.  ts = bytes per 32 rows, bs = head term in elements, cv = irrep
  // block width of a packed row (0: plain rows).  W waves share the jobs; the per-lane source offsets are recomputed
  // per job (a dozen integer instructions against the hundreds of cycles of a tile step).
  __device__ __forceinline__ static void issue(int wid, int W, int lane, int tile, int nt, int T, unsigned lds0,
                                               unsigned lds1, const i32x4 rs0, const i32x4 rs1, int bs0, int bs1,
                                               int64_t sT0, int64_t sT1, int cv0, int cv1, int slot = -1,
                                               int only = 0) {
    if (slot < 0) slot = tile;                       // slot: which 5 KiB tile slot of the images receives the rows
                                                     // only: 0 both images, 1 the first, 2 the second
    asm volatile("" : "+v"(lane));                   // per-lane offsets are recomputed per call: hoisted out of the
                                                     // loops they would cost dozens of registers
    const bool packed = cv0 > 0;
    const int per = packed ? 8 : 5;
    const int rows_last = T - 32 * (nt - 1);
    const int jb = only == 2 ? per : 0, je = only == 1 ? per : 2 * per;
    for (int job = jb + wid; job < je; job += W) {
      const bool second = job >= per;
      const int k = second ? job - per : job;
      const int sT = (int)(second ? sT1 : sT0), cv = second ? cv1 : cv0, bs = second ? bs1 : bs0;
      const unsigned dst = (second ? lds1 : lds0) + slot * TILE_B;
      int row;
      unsigned v;
      if (k < 4) {                                   // main: 8 rows x 8 chunks of 16 B
        row = k * 8 + (lane >> 3);
        const int g = (lane & 7) ^ swz(row);
        if (!packed) v = (unsigned)(((tile * 32 + row) * sT) * 2 + g * 16);
        else v = (unsigned)(((tile * 32 + row) * sT + (g < 4 ? g * cv + bs : (4 + 2 * ((g - 4) >> 1)) * cv + ((g - 4) & 1) * 8 + 2 * bs)) * 2);
        if (tile == nt - 1 && row >= rows_last) v = OOR;
        if (second) dma16(dst + k * 1024, v, rs1); else dma16(dst + k * 1024, v, rs0);
      } else if (!packed) {                          // tail, plain rows: 32 rows x 2 chunks of 16 B
        row = lane >> 1;
        v = (unsigned)(((tile * 32 + row) * sT) * 2 + (8 + (lane & 1)) * 16);
        if (tile == nt - 1 && row >= rows_last) v = OOR;
        if (second) dma16(dst + TAIL_OFF, v, rs1); else dma16(dst + TAIL_OFF, v, rs0);
      } else {                                       // tail, packed rows: 8 rows x 8 pieces of 4 B
        const int m = k - 4, d = lane & 7;
        row = m * 8 + (lane >> 3);
        v = (unsigned)(((tile * 32 + row) * sT + (d < 4 ? d * cv + 8 + bs : (4 + 2 * ((d - 4) >> 1)) * cv + 16 + 2 * ((d - 4) & 1) + 2 * bs)) * 2);
        if (tile == nt - 1 && row >= rows_last) v = OOR;
        if (second) dma4(dst + TAIL_OFF + m * 256, v, rs1); else dma4(dst + TAIL_OFF + m * 256, v, rs0);
      }
    }
  }
  // all jobs of ONE tile of ONE image by the calling wave (row tiles of a compute wave's IO slot): `dst` = LDS address
  // of the slot, rows of tile `tile` of the tensor behind `rs`
  __device__ __forceinline__ static void issue_one(int lane, int tile, int nt, int T, unsigned dst, const i32x4 rs, int bs,
                                                   int64_t sT_, int cv) {
    asm volatile("" : "+v"(lane));
    const bool packed = cv > 0;
    const int per = packed ? 8 : 5;
    const int rows_last = T - 32 * (nt - 1);
    const int sT = (int)sT_;
    for (int k = 0; k < per; ++k) {
      int row;
      unsigned v;
      if (k < 4) {
        row = k * 8 + (lane >> 3);
        const int g = (lane & 7) ^ swz(row);
        if (!packed) v = (unsigned)(((tile * 32 + row) * sT) * 2 + g * 16);
        else v = (unsigned)(((tile * 32 + row) * sT + (g < 4 ? g * cv + bs : (4 + 2 * ((g - 4) >> 1)) * cv + ((g - 4) & 1) * 8 + 2 * bs)) * 2);
        if (tile == nt - 1 && row >= rows_last) v = OOR;
        dma16(dst + k * 1024, v, rs);
      } else if (!packed) {
        row = lane >> 1;
        v = (unsigned)(((tile * 32 + row) * sT) * 2 + (8 + (lane & 1)) * 16);
        if (tile == nt - 1 && row >= rows_last) v = OOR;
        dma16(dst + TAIL_OFF, v, rs);
      } else {
        const int m = k - 4, d = lane & 7;
        row = m * 8 + (lane >> 3);
        v = (unsigned)(((tile * 32 + row) * sT + (d < 4 ? d * cv + 8 + bs : (4 + 2 * ((d - 4) >> 1)) * cv + 16 + 2 * ((d - 4) & 1) + 2 * bs)) * 2);
        if (tile == nt - 1 && row >= rows_last) v = OOR;
        dma4(dst + TAIL_OFF + m * 256, v, rs);
      }
    }
  }
};

// One image at a time, per-lane offsets precomputed once per kernel: a tile of one image is 5 / 8 jobs, wave w runs job
// w (and w + W).  An issue is then four or five instructions - cheap enough to sit between the MFMAs of an unrolled sweep.
struct LeanStager {
  unsigned vo[2];       // per lane: byte offset inside (batch element, head) for tile 0, head term excluded
  int vrow[2];          // per lane: row inside the tile
  int hmul[2];          // per lane: bytes per element of the head term bs
  int kind[2], ldsoff[2];
  bool on[2];
  int tstride, rows_last, nt;
  __device__ __forceinline__ void setup(int wid, int W, int lane, int64_t sT, int cv, int nt_, int T) {
    const bool packed = cv > 0;
    const int per = packed ? 8 : 5;
    nt = nt_;
    tstride = (int)(32 * sT * 2);
    rows_last = T - 32 * (nt - 1);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int k = wid + W * s;
      on[s] = k < per;
      vo[s] = 0; vrow[s] = 0; hmul[s] = 0; kind[s] = 0; ldsoff[s] = 0;
      if (!on[s]) continue;
      if (k < 4) {
        const int row = k * 8 + (lane >> 3), g = (lane & 7) ^ swz(row);
        kind[s] = 0; ldsoff[s] = k * 1024; vrow[s] = row;
        if (!packed) vo[s] = (unsigned)((row * sT) * 2 + g * 16);
        else {
          vo[s] = (unsigned)((row * sT + (g < 4 ? g * cv : (4 + 2 * ((g - 4) >> 1)) * cv + ((g - 4) & 1) * 8)) * 2);
          hmul[s] = g < 4 ? 2 : 4;
        }
      } else if (!packed) {
        const int row = lane >> 1;
        kind[s] = 1; ldsoff[s] = TAIL_OFF; vrow[s] = row;
        vo[s] = (unsigned)((row * sT) * 2 + (8 + (lane & 1)) * 16);
      } else {
        const int m = k - 4, row = m * 8 + (lane >> 3), d = lane & 7;
        kind[s] = 2; ldsoff[s] = TAIL_OFF + m * 256; vrow[s] = row;
        vo[s] = (unsigned)((row * sT + (d < 4 ? d * cv + 8 : (4 + 2 * ((d - 4) >> 1)) * cv + 16 + 2 * ((d - 4) & 1))) * 2);
        hmul[s] = d < 4 ? 2 : 4;
      }
    }
  }
  // rows of `tile` of one image -> tile slot `slot` (default: the same index) of the image buffer at LDS address `img`
  __device__ __forceinline__ void issue(int tile, unsigned img, const i32x4 rs, int bs, int slot = -1) const {
    if (slot < 0) slot = tile;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      if (!on[s]) continue;
      unsigned v = vo[s] + (unsigned)(hmul[s] * bs) + (unsigned)(tile * tstride);
      if (tile == nt - 1 && vrow[s] >= rows_last) v = OOR;
      const unsigned dst = img + slot * TILE_B + ldsoff[s];
      if (kind[s] == 2) dma4(dst, v, rs); else dma16(dst, v, rs);
    }
  }
};

// ---- per-lane fragment addresses inside a tile ----------------------------------------------------------------------
struct FragAddr {
  int rb[5];     // row read: row r = lane & 31, logical chunk 2 ks + half (ks = 4: the tail)
  int tb[5];     // transposing read: [d-tile 0 lo, 0 hi, 1 lo, 1 hi, tail lo]; +2048 (tail: +512) for keys 16..31, tail hi: +256
  __device__ __forceinline__ void setup(int lane) {
    const int r = lane & 31, half = lane >> 5;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) rb[ks] = r * 128 + (((2 * ks + half) ^ swz(r)) << 4);
    rb[4] = TAIL_OFF + r * 32 + half * 16;
    const int i = lane & 15, g = lane >> 4, q4 = i >> 2, p = i & 3;
    const int rlo = 4 * (g >> 1) + q4, rhi = rlo + 8;
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      const int ch = d * 4 + (g & 1) * 2 + (p >> 1);
      tb[2 * d] = rlo * 128 + ((ch ^ swz(rlo)) << 4) + (p & 1) * 8;
      tb[2 * d + 1] = rhi * 128 + ((ch ^ swz(rhi)) << 4) + (p & 1) * 8;
    }
    tb[4] = TAIL_OFF + rlo * 32 + p * 8;       // elements 64..79; the lanes of columns 80..95 read the same bytes (rows discarded)
  }
};

__device__ __forceinline__ bf16x8 rowfrag(const char* tile, const FragAddr& fa, int ks) {
  return *(const bf16x8*)(tile + fa.rb[ks]);
}
__device__ __forceinline__ bf16x8 trfrag(const char* tile, const FragAddr& fa, int d, int khalf) {
  const char* lo;
  const char* hi;
  if (d < 2) {
    lo = tile + fa.tb[2 * d] + khalf * 2048;
    hi = tile + fa.tb[2 * d + 1] + khalf * 2048;
  } else {
    lo = tile + fa.tb[4] + khalf * 512;
    hi = lo + 256;
  }
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)lo);
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)hi);
  const s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(bf16x8, v);
}

// own rows of a wave (B operand of the swapped products): lane (r, half) <- chunk 2 ks + half of row tile*32 + r
__device__ __forceinline__ void load_rows(bf16x8 (&f)[KS], const bf16* base, int64_t st, int tile, int T, int lane,
                                          const HeadMap m) {
  const int r = lane & 31, half = lane >> 5;
  const int i = tile * 32 + r;
  const int ic = i < T ? i : T - 1;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) f[ks] = __builtin_bit_cast(bf16x8, hm_load16(base + (int64_t)ic * st, 2 * ks + half, m));
}
// the same fragments for the extra rows (row 32 W + min(r, nx - 1)) out of their LDS copy [nx][160 B]
__device__ __forceinline__ void xrow_frags(bf16x8 (&f)[KS], const char* xr, int nx, int lane) {
  const int r = lane & 31, half = lane >> 5;
  const char* p = xr + (r < nx ? r : nx - 1) * (HD * 2) + half * 16;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) f[ks] = *(const bf16x8*)(p + ks * 32);
}

// 16-byte row stores: accumulator set (rows = d on registers, lane = token) -> bf16 token rows.  A lane holds elements
// 8 g + 4 half .. + 3 of group g; exchanging halves between the two half-waves (v_permlane32_swap) gives lanes 0-31
// the whole even group and lanes 32-63 the whole odd group of a pair: one 16-byte store per lane and pair.
__device__ __forceinline__ void store_rows16(bf16* row, const f32x16 (&acc)[DT], float f, int half, const HeadMap m) {
  asm volatile("" : "+v"(half));                    // keep the piece offsets out of the persistent loop's preheader
#pragma unroll
  for (int pr = 0; pr < 5; ++pr) {                  // groups (2 pr, 2 pr + 1)
    u32x2 a, b;
    {
      const int g = 2 * pr, d = g >> 2, k4 = g & 3;
      const bf16x4 v = {(bf16)(acc[d][4 * k4] * f), (bf16)(acc[d][4 * k4 + 1] * f), (bf16)(acc[d][4 * k4 + 2] * f), (bf16)(acc[d][4 * k4 + 3] * f)};
      a = __builtin_bit_cast(u32x2, v);
    }
    {
      const int g = 2 * pr + 1, d = g >> 2, k4 = g & 3;
      const bf16x4 v = {(bf16)(acc[d][4 * k4] * f), (bf16)(acc[d][4 * k4 + 1] * f), (bf16)(acc[d][4 * k4 + 2] * f), (bf16)(acc[d][4 * k4 + 3] * f)};
      b = __builtin_bit_cast(u32x2, v);
    }
    // lanes 32-63 of `a` <-> lanes 0-31 of `b`
    const auto r0 = __builtin_amdgcn_permlane32_swap(a[0], b[0], false, false);
    const auto r1 = __builtin_amdgcn_permlane32_swap(a[1], b[1], false, false);
    // lanes 0-31: (own a | upper's a) = group 2 pr; lanes 32-63: (lower's b | own b) = group 2 pr + 1
    const u32x4 v = {r0[0], r1[0], r0[1], r1[1]};
    hm_store16(row, 2 * pr + half, v, m);
  }
}

}  // namespace a80
}  // namespace octic
