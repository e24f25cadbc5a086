// Shared pieces of the attention kernels (csrc/attention.hip: generic shapes; csrc/attn80.hip: the persistent
// head_dim-80 kernels): argument structs, head-vector addressing on packed LinearD8 rows, accumulator helpers.
#pragma once
#include "octic_common.hpp"

namespace octic {

typedef __attribute__((ext_vector_type(16))) float f32x16;

struct AttnArgs {
  const bf16* q; const bf16* k; const bf16* v;   // element (b,h,t,d) at base + b*sB + h*sH + t*sT + d
  int64_t sB, sH, sT;
  bf16* o; int64_t oB, oH, oT;
  float* lse;            // [B,H,T] log2-domain log-sum-exp of the scaled scores
  int H, T, hd;
  float scale_log2;      // softmax scale * log2(e)
  // packed mode (cv_in > 0): q == k == v is the packed LinearD8 output [B,T,3*8c] (row stride sT, sH = 0), o the packed
  // [B,T,8c] input of the output projection; cv = irrep block width of a row (3c / c), c = channels per irrep.
  int cv_in, cv_out, c;
  int dbg = 0;           // developer probes (csrc/attn80.hip): 1 = K / V descriptors with zero records (their DMA is dropped)
};

// ---- head-vector addressing ------------------------------------------------------------------------------------------
// Plain (cv == 0): the hd elements of (b,h,t) are contiguous.  Packed: the head vector of tensor s, head h is six pieces
// of the token row - w = c/H channels of each one-dimensional irrep and 2w of each row of E (reference
// d8_layers.py:631-643, 650-656) - so AttentionD8 needs no pack / unpack passes.  The dot products do not care about the
// order of the hd elements as long as q, k (and v, o) agree, so the pieces are visited in an order that keeps 16-byte
// groups inside a piece where possible (w = 10, hd = 80 = ten groups of 8): groups 0-3 = first 8 channels of A1, A2, B1,
// B2; groups 4-7 = E0[0:8], E0[8:16], E1[0:8], E1[8:16]; group 8 = the four 2-channel remainders of the 1-D pieces;
// group 9 = the two 4-channel remainders of the E pieces.  Pieces start on 4-byte boundaries (20 h bytes).
struct HeadMap { int cv, bs; };        // bs = s*c + h*w
typedef unsigned u32x4_u __attribute__((ext_vector_type(4), aligned(4)));
typedef unsigned u32x2_u __attribute__((ext_vector_type(2), aligned(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int hm_off8(const HeadMap m, int g) {
  return g < 4 ? g * m.cv + m.bs : (4 + 2 * ((g - 4) >> 1)) * m.cv + 2 * m.bs + ((g - 4) & 1) * 8;
}
__device__ __forceinline__ u32x4 hm_load16(const bf16* row, int c, const HeadMap m) {
  if (m.cv == 0) return *(const u32x4*)(row + c * 8);
  if (c < 8) return *(const u32x4_u*)(row + hm_off8(m, c));
  if (c == 8) {
    const bf16* p = row + m.bs + 8;
    return u32x4{*(const unsigned*)p, *(const unsigned*)(p + m.cv), *(const unsigned*)(p + 2 * m.cv),
                 *(const unsigned*)(p + 3 * m.cv)};
  }
  const bf16* p = row + 4 * m.cv + 2 * m.bs + 16;
  const u32x2_u e0 = *(const u32x2_u*)p, e1 = *(const u32x2_u*)(p + 2 * m.cv);
  return u32x4{e0[0], e0[1], e1[0], e1[1]};
}
__device__ __forceinline__ void hm_store16(bf16* row, int c, const u32x4 v, const HeadMap m) {
  if (m.cv == 0) { *(u32x4*)(row + c * 8) = v; return; }
  if (c < 8) { *(u32x4_u*)(row + hm_off8(m, c)) = v; return; }
  if (c == 8) {
    bf16* p = row + m.bs + 8;
    *(unsigned*)p = v[0]; *(unsigned*)(p + m.cv) = v[1]; *(unsigned*)(p + 2 * m.cv) = v[2]; *(unsigned*)(p + 3 * m.cv) = v[3];
    return;
  }
  bf16* p = row + 4 * m.cv + 2 * m.bs + 16;
  *(u32x2_u*)p = u32x2_u{v[0], v[1]};
  *(u32x2_u*)(p + 2 * m.cv) = u32x2_u{v[2], v[3]};
}
// four consecutive elements: half `sub` of group g
__device__ __forceinline__ void hm_store8(bf16* row, int g, int sub, const u32x2 v, const HeadMap m) {
  if (m.cv == 0) { *(u32x2*)(row + g * 8 + sub * 4) = v; return; }
  if (g < 8) { *(u32x2_u*)(row + hm_off8(m, g) + sub * 4) = u32x2_u{v[0], v[1]}; return; }
  if (g == 8) {
    bf16* p = row + 2 * sub * m.cv + m.bs + 8;
    *(unsigned*)p = v[0]; *(unsigned*)(p + m.cv) = v[1];
    return;
  }
  *(u32x2_u*)(row + (4 + 2 * sub) * m.cv + 2 * m.bs + 16) = u32x2_u{v[0], v[1]};
}
// the head maps of one (b, h): tensor s of the packed projection output / the packed single-tensor rows
struct HeadMaps { HeadMap q, k, v, o; };
template <typename A>
__device__ __forceinline__ HeadMaps head_maps(const A& a, int h) {
  HeadMaps m;
  const int w = a.cv_in > 0 ? a.c / a.H : 0;
  m.q = HeadMap{a.cv_in, h * w};
  m.k = HeadMap{a.cv_in, a.c + h * w};
  m.v = HeadMap{a.cv_in, 2 * a.c + h * w};
  m.o = HeadMap{a.cv_out, h * w};
  return m;
}
// Workgroup / work-item index -> (b, h) unit.  When the heads of a token share cache lines (packed rows: 20-byte pieces;
// the standard block's fused [B,T,3,H,hd] projection: 160-byte pieces) the units are dealt so that each XCD works on
// whole batches: the lines a head leaves partly used are consumed by its neighbours out of the same L2 instead of
// being fetched once per XCD.  Bijective for any unit count.
__device__ __forceinline__ int unit_of(int idx, int units, bool shared_rows) {
  if (!shared_rows) return idx;
  const int xcd = idx & 7, local = idx >> 3, q8 = units >> 3, r8 = units & 7;
  return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + local;
}

// element e of the head vector -> element offset in the row (scalar accesses of the merge paths)
__device__ __forceinline__ int hm_elem(int e, const HeadMap m) {
  if (m.cv == 0) return e;
  const int g = e >> 3, j = e & 7;
  if (g < 8) return hm_off8(m, g) + j;
  if (g == 8) return (j >> 1) * m.cv + m.bs + 8 + (j & 1);
  return (4 + 2 * (j >> 2)) * m.cv + 2 * m.bs + 16 + (j & 3);
}

__device__ inline int acc_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

__device__ inline bf16x8 pack8(const float* p) {
  bf16x8 r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r[i] = (bf16)p[i];
  return r;
}

// V^T (or any row-major [key][col] LDS image) fragment of the A operand for  Y = A X  where X is an accumulator tile:
// lane (r = lane&31 -> column c0 + r, half) gets, for k-step s, keys 16s + 4*half + {0..3} and + 8 + {0..3}.
__device__ inline bf16x8 tr_frag(const char* img, int row_bytes, int key0, int c0, int lane) {
  const int i = lane & 15, g = lane >> 4;
  const int q4 = i >> 2, p = i & 3, half = g >> 1;
  const char* a = img + (size_t)(key0 + 4 * half + q4) * row_bytes + (c0 + (g & 1) * 16 + 4 * p) * 2;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)a);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a + 8 * row_bytes));
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

constexpr int kPartPad = 4;    // f32 row pad of the partial-result images (conflict-free 16-byte accesses)

struct AttnBwdArgs {
  const bf16* q; const bf16* k; const bf16* v; int64_t sB, sH, sT;      // inputs
  const bf16* o; const bf16* dout; int64_t oB, oH, oT;                   // forward output and its cotangent
  const float* lse; float* delta;                                         // [B,H,T] f32
  bf16* dq; bf16* dk; bf16* dv; int64_t gB, gH, gT;                     // gradients
  int H, T, hd;
  float scale, scale_log2;
  int cv_in, cv_out, c;                                                   // packed mode, see AttnArgs
};


// accumulator tile set -> bf16 rows of the output (lane r = row `row`, 4 consecutive columns per store)
template <int DT>
__device__ __forceinline__ void store_rows(bf16* row, const f32x16 (&acc)[DT], float f, int hd, int half,
                                           const HeadMap m = HeadMap{0, 0}) {
#pragma unroll
  for (int d = 0; d < DT; ++d)
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) {
      const int g = d * 4 + k4;                  // elements 8 g + 4 half .. + 3
      if (g * 8 < hd) {
        bf16x4 ov = {(bf16)(acc[d][4 * k4] * f), (bf16)(acc[d][4 * k4 + 1] * f), (bf16)(acc[d][4 * k4 + 2] * f),
                     (bf16)(acc[d][4 * k4 + 3] * f)};
        hm_store8(row, g, half, __builtin_bit_cast(u32x2, ov), m);
      }
    }
}

// 16-byte row stores (head_dim % 16 == 0): a lane holds elements 8 g + 4 half .. + 3 of group g; exchanging halves
// between the two half-waves (v_permlane32_swap) gives lanes 0-31 the whole even group and lanes 32-63 the whole odd
// group of a pair - one 16-byte store per lane and pair instead of two 8-byte ones (the store tail of these kernels is
// bound by the number of store instructions, not by bytes: MI355X guide, T21).
template <int DT>
__device__ __forceinline__ void store_rows_wide(bf16* row, const f32x16 (&acc)[DT], float f, int hd, int half,
                                                const HeadMap m = HeadMap{0, 0}) {
#pragma unroll
  for (int pr = 0; pr < DT * 2; ++pr) {                // groups (2 pr, 2 pr + 1)
    if (pr * 16 < hd) {
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
      const auto r0 = __builtin_amdgcn_permlane32_swap(a[0], b[0], false, false);   // lanes 32-63 of a <-> lanes 0-31 of b
      const auto r1 = __builtin_amdgcn_permlane32_swap(a[1], b[1], false, false);
      hm_store16(row, 2 * pr + half, u32x4{r0[0], r1[0], r0[1], r1[1]}, m);
    }
  }
}

template <int DT>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[DT]) {
#pragma unroll
  for (int d = 0; d < DT; ++d)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[d][i] = 0.f;
}


// csrc/attn80.hip: persistent head_dim-80 kernels (all operands by LDS-DMA, one head ahead)
int attn80_fwd_ok(const AttnArgs& a);
int attn80_fwd_launch(const AttnArgs& a, int64_t B, hipStream_t s);
// csrc/attn80_bwd.hip: single-pass backward (dq, dk, dv from ONE recomputation of P) for head_dim 80, T = 257
int attn80_bwd_ok(const AttnBwdArgs& a);
int attn80_bwd_launch(const AttnBwdArgs& a, int64_t B, hipStream_t s);

}  // namespace octic
