// Softmax attention core for gfx950, bf16, short sequences (T <= 512: ViT token counts 197 / 257).
// Replaces F.scaled_dot_product_attention in AttentionD8 (reference octic_vits/d8_layers.py:645-648) and in the
// standard blocks (deit/vit.py:41-45).  On MI355X the stock SDPA backward for head_dim 80 takes ~640 us per call
// (64 x 16 heads x 257 tokens); the sequence is short enough for a much simpler structure than flash attention:
//
//   * one workgroup per (batch, head), one wave per 32 queries (9 waves at T = 257); K and V of the head are loaded
//     ONCE into LDS (K rows padded to an odd number of 16-byte slots -> conflict-free ds_read_b128; V rows sized so
//     the transposing reads are conflict-free);
//   * swapped product  X = K Q^T  on v_mfma_f32_32x32x16_bf16: the accumulator has the QUERY on the lane and the keys
//     in registers, so max / sum of the softmax are in-lane reductions plus one exchange between the half-waves;
//   * P = exp2(X - m) is used straight from the accumulator registers as the B operand of  O^T = V^T P  (an accumulator
//     tile is a valid operand for a product that sums over its row index; the k order inside a step is permuted and
//     the V^T fragments are fetched in the same order with ds_read_b64_tr_b16 from the row-major V image);
//   * O^T keeps the query on the lane too, so the online-softmax rescale and the final 1/l are lane-local.
// head_dim must be a multiple of 16 (<= 128).  Bound: MFMA/VALU mix, not HBM (q,k,v,o are 4 x 42 MB per call).
#include <stdlib.h>
#include "attn_common.hpp"

namespace octic {

inline int attn_rsk(int hd) { return hd * 2 + 16; }                       // K image row bytes (odd # of 16-B slots)
inline int attn_rsv(int dp) { int r = dp * 2; return ((r / 4) % 32 == 0) ? r + 64 : r; }   // V image row bytes

// Stage two row images (zero-filled beyond T rows / kcA, kcB source chunks per row) with all global loads of a batch
// in flight before the first LDS store.  The s_memtime timeline showed 10-20k cycles (a third of the kernel) in
// the one-load-one-store loops this replaces: every trip waited out a full HBM round trip.
#ifndef OCTIC_STAGE_BATCH
#define OCTIC_STAGE_BATCH 5
#endif
constexpr int kStageBatch = OCTIC_STAGE_BATCH;
// No integer division anywhere in here: the first version turned a linear item index into (row, chunk) with `q / wc`
// (and `q / T` for the packed pieces) twice per item - ~28 divisions of ~30 VALU instructions per thread; the s_memtime
// timeline showed the staging at 12-13 k cycles per image pair (HALF of the backward kernels), VALU-bound, not memory-bound.
// Now a thread owns chunk column tid & 15 and walks rows (wc <= 16 chunks per image row).
__device__ __forceinline__ void stage_two(char* imgA, int rsA, const bf16* srcA, int64_t stA, int kcA, int wcA,
                                          char* imgB, int rsB, const bf16* srcB, int64_t stB, int kcB, int wcB,
                                          int T, int Tp, int tid, int nthr, const HeadMap mA = HeadMap{0, 0},
                                          const HeadMap mB = HeadMap{0, 0}) {
  const int c = tid & 15, t0 = tid >> 4, tstep = nthr >> 4;
  if (mA.cv != 0 && kcA == 10 && kcB == 10) {
    // Packed rows at head_dim 80: stage by PIECES - a thread fetches one whole piece (20 bytes of a one-dimensional irrep: 16 + 4, or
    // 40 bytes of an E row: 16 + 16 + 8) and spreads it over the row image's groups.  One cache-line request per
    // piece instead of one per 16-byte group (6 instead of 16 per row).
    // pads first: rows >= T, and groups >= hd / 8 of every row
    for (int t = t0; t < Tp; t += tstep) {
      if (c < wcA && (t >= T || c >= kcA)) *(u32x4*)(imgA + (size_t)t * rsA + c * 16) = u32x4{0, 0, 0, 0};
      if (c < wcB && (t >= T || c >= kcB)) *(u32x4*)(imgB + (size_t)t * rsB + c * 16) = u32x4{0, 0, 0, 0};
    }
    // a thread takes row t = tid, tid + nthr, ...; all six pieces of the row of both images are requested before the
    // first LDS write
    for (int t = tid; t < T; t += nthr) {
      u32x4 a4[4], b4[4], ae[2][2], be[2][2];
      unsigned a1[4], b1[4];
      u32x2 a2[2], b2[2];
      const bf16* rowa = srcA + (int64_t)t * stA;
      const bf16* rowb = srcB + (int64_t)t * stB;
#pragma unroll
      for (int pz = 0; pz < 4; ++pz) {
        const bf16* ra = rowa + pz * mA.cv + mA.bs;
        const bf16* rb = rowb + pz * mB.cv + mB.bs;
        a4[pz] = *(const u32x4_u*)ra; a1[pz] = *(const unsigned*)(ra + 8);
        b4[pz] = *(const u32x4_u*)rb; b1[pz] = *(const unsigned*)(rb + 8);
      }
#pragma unroll
      for (int pz = 0; pz < 2; ++pz) {
        const bf16* ra = rowa + (4 + 2 * pz) * mA.cv + 2 * mA.bs;
        const bf16* rb = rowb + (4 + 2 * pz) * mB.cv + 2 * mB.bs;
        ae[pz][0] = *(const u32x4_u*)ra; ae[pz][1] = *(const u32x4_u*)(ra + 8);
        be[pz][0] = *(const u32x4_u*)rb; be[pz][1] = *(const u32x4_u*)(rb + 8);
        const u32x2_u ta = *(const u32x2_u*)(ra + 16), tb = *(const u32x2_u*)(rb + 16);
        a2[pz] = u32x2{ta[0], ta[1]}; b2[pz] = u32x2{tb[0], tb[1]};
      }
      char* la = imgA + (size_t)t * rsA;
      char* lb = imgB + (size_t)t * rsB;
#pragma unroll
      for (int pz = 0; pz < 4; ++pz) {
        *(u32x4*)(la + pz * 16) = a4[pz]; *(unsigned*)(la + 8 * 16 + pz * 4) = a1[pz];
        *(u32x4*)(lb + pz * 16) = b4[pz]; *(unsigned*)(lb + 8 * 16 + pz * 4) = b1[pz];
      }
#pragma unroll
      for (int pz = 0; pz < 2; ++pz) {
        *(u32x4*)(la + (4 + 2 * pz) * 16) = ae[pz][0]; *(u32x4*)(la + (5 + 2 * pz) * 16) = ae[pz][1];
        *(u32x2*)(la + 9 * 16 + pz * 8) = a2[pz];
        *(u32x4*)(lb + (4 + 2 * pz) * 16) = be[pz][0]; *(u32x4*)(lb + (5 + 2 * pz) * 16) = be[pz][1];
        *(u32x2*)(lb + 9 * 16 + pz * 8) = b2[pz];
      }
    }
    return;
  }
  // wc = chunks written per row (>= kc: the extra ones are zeros), kc = chunks that exist in the source row
  for (int base = t0; base < Tp; base += kStageBatch * tstep) {
    u32x4 va[kStageBatch], vb[kStageBatch];
#pragma unroll
    for (int it = 0; it < kStageBatch; ++it) {
      const int t = base + it * tstep;
      va[it] = u32x4{0, 0, 0, 0};
      vb[it] = u32x4{0, 0, 0, 0};
      if (t < T && c < kcA) va[it] = hm_load16(srcA + (int64_t)t * stA, c, mA);
      if (t < T && c < kcB) vb[it] = hm_load16(srcB + (int64_t)t * stB, c, mB);
    }
#pragma unroll
    for (int it = 0; it < kStageBatch; ++it) {
      const int t = base + it * tstep;
      if (t < Tp && c < wcA) *(u32x4*)(imgA + (size_t)t * rsA + c * 16) = va[it];
      if (t < Tp && c < wcB) *(u32x4*)(imgB + (size_t)t * rsB + c * 16) = vb[it];
    }
  }
}

// ---- the same staging split into "request" and "write to LDS", so that a kernel can have EVERY global load of its
// prologue in flight at once.  The backward kernels need two image pairs one after the other (the wave's own rows come
// out of the first pair, then the pair is overwritten); staged pair by pair, each in two batches, the prologue was a
// chain of five dependent memory round trips of ~3 us under load: 25-29 k of a workgroup's 50 k cycles
// (tools/attn_trace.py).  Register cost: 72 VGPRs per pair on plain rows, 80 on packed rows - free in a prologue.
constexpr int kStageRows = 9;      // rows per thread: Tp / (threads / 16) = 8, or 9 for nine tiles on eight waves
struct StagePlain { u32x4 a[kStageRows], b[kStageRows]; };
// Which 16-byte chunk of which rows a thread stages.  A row of the head is kc chunks (10 at head_dim 80); mapping a thread
// to (row tid / kc, chunk tid % kc) fills 60 of a wave's 64 lanes with work (6 rows per wave-instruction, 6 instructions
// per thread and tensor at T = 257) where (tid >> 4, tid & 15) filled 40 (4 rows, 9 instructions): the prologues are
// bound by the CU's vector-memory instruction rate, so lanes that address nothing are time.
struct StageMap { int c, t0, tstep; };
__device__ __forceinline__ StageMap stage_map(int tid, int nthr, int kc) {
  StageMap m;
  m.tstep = nthr / kc;
  m.t0 = tid / kc;
  m.c = tid - m.t0 * kc;
  if (m.t0 >= m.tstep) {      // the last nthr % kc threads: no rows
    m.t0 = 1 << 20;
    m.c = 0;
  }
  return m;
}
__device__ __forceinline__ void stage_request(StagePlain& R, const bf16* srcA, int64_t stA, int kcA, const bf16* srcB,
                                              int64_t stB, int kcB, int T, int tid, int nthr,
                                              const HeadMap mA = HeadMap{0, 0}, const HeadMap mB = HeadMap{0, 0}) {
  const StageMap sm = stage_map(tid, nthr, kcA > kcB ? kcA : kcB);
  const int c = sm.c, t0 = sm.t0, tstep = sm.tstep;
#pragma unroll
  for (int it = 0; it < kStageRows; ++it) {
    const int t = t0 + it * tstep;
    R.a[it] = u32x4{0, 0, 0, 0};
    R.b[it] = u32x4{0, 0, 0, 0};
    if (t < T && c < kcA) R.a[it] = hm_load16(srcA + (int64_t)t * stA, c, mA);       // (cv == 0: the plain 16-byte chunk)
    if (t < T && c < kcB) R.b[it] = hm_load16(srcB + (int64_t)t * stB, c, mB);
  }
}
__device__ __forceinline__ void stage_write(const StagePlain& R, char* imgA, int rsA, int wcA, char* imgB, int rsB, int wcB,
                                            int Tp, int tid, int nthr) {
  const StageMap sm = stage_map(tid, nthr, wcA > wcB ? wcA : wcB);
  const int c = sm.c, t0 = sm.t0, tstep = sm.tstep;
#pragma unroll
  for (int it = 0; it < kStageRows; ++it) {
    const int t = t0 + it * tstep;
    if (t < Tp && c < wcA) *(u32x4*)(imgA + (size_t)t * rsA + c * 16) = R.a[it];
    if (t < Tp && c < wcB) *(u32x4*)(imgB + (size_t)t * rsB + c * 16) = R.b[it];
  }
}
// packed rows: a thread owns row tid (T <= threads in every backward launch) and fetches its six pieces per image
struct StagePacked {
  u32x4 a4[4], b4[4], ae[2][2], be[2][2];
  unsigned a1[4], b1[4];
  u32x2 a2[2], b2[2];
};
__device__ __forceinline__ void stage_request(StagePacked& R, const bf16* srcA, int64_t stA, const bf16* srcB, int64_t stB,
                                              int T, int tid, const HeadMap mA, const HeadMap mB) {
  const int t = tid < T ? tid : T - 1;
  const bf16* rowa = srcA + (int64_t)t * stA;
  const bf16* rowb = srcB + (int64_t)t * stB;
#pragma unroll
  for (int pz = 0; pz < 4; ++pz) {
    const bf16* ra = rowa + pz * mA.cv + mA.bs;
    const bf16* rb = rowb + pz * mB.cv + mB.bs;
    R.a4[pz] = *(const u32x4_u*)ra; R.a1[pz] = *(const unsigned*)(ra + 8);
    R.b4[pz] = *(const u32x4_u*)rb; R.b1[pz] = *(const unsigned*)(rb + 8);
  }
#pragma unroll
  for (int pz = 0; pz < 2; ++pz) {
    const bf16* ra = rowa + (4 + 2 * pz) * mA.cv + 2 * mA.bs;
    const bf16* rb = rowb + (4 + 2 * pz) * mB.cv + 2 * mB.bs;
    R.ae[pz][0] = *(const u32x4_u*)ra; R.ae[pz][1] = *(const u32x4_u*)(ra + 8);
    R.be[pz][0] = *(const u32x4_u*)rb; R.be[pz][1] = *(const u32x4_u*)(rb + 8);
    const u32x2_u ta = *(const u32x2_u*)(ra + 16), tb = *(const u32x2_u*)(rb + 16);
    R.a2[pz] = u32x2{ta[0], ta[1]}; R.b2[pz] = u32x2{tb[0], tb[1]};
  }
}
__device__ __forceinline__ void stage_write(const StagePacked& R, char* imgA, int rsA, int kcA, int wcA, char* imgB, int rsB,
                                            int kcB, int wcB, int T, int Tp, int tid, int nthr) {
  const int c = tid & 15;
  for (int t = tid >> 4; t < Tp; t += nthr >> 4) {       // pads: rows >= T, and groups >= hd / 8 of every row
    if (c < wcA && (t >= T || c >= kcA)) *(u32x4*)(imgA + (size_t)t * rsA + c * 16) = u32x4{0, 0, 0, 0};
    if (c < wcB && (t >= T || c >= kcB)) *(u32x4*)(imgB + (size_t)t * rsB + c * 16) = u32x4{0, 0, 0, 0};
  }
  if (tid < T) {
    char* la = imgA + (size_t)tid * rsA;
    char* lb = imgB + (size_t)tid * rsB;
#pragma unroll
    for (int pz = 0; pz < 4; ++pz) {
      *(u32x4*)(la + pz * 16) = R.a4[pz]; *(unsigned*)(la + 8 * 16 + pz * 4) = R.a1[pz];
      *(u32x4*)(lb + pz * 16) = R.b4[pz]; *(unsigned*)(lb + 8 * 16 + pz * 4) = R.b1[pz];
    }
#pragma unroll
    for (int pz = 0; pz < 2; ++pz) {
      *(u32x4*)(la + (4 + 2 * pz) * 16) = R.ae[pz][0]; *(u32x4*)(la + (5 + 2 * pz) * 16) = R.ae[pz][1];
      *(u32x2*)(la + 9 * 16 + pz * 8) = R.a2[pz];
      *(u32x4*)(lb + (4 + 2 * pz) * 16) = R.be[pz][0]; *(u32x4*)(lb + (5 + 2 * pz) * 16) = R.be[pz][1];
      *(u32x2*)(lb + 9 * 16 + pz * 8) = R.b2[pz];
    }
  }
}
__device__ __forceinline__ float dot8(const u32x4 x, const u32x4 y) {
  const bf16x8 a = __builtin_bit_cast(bf16x8, x), b = __builtin_bit_cast(bf16x8, y);
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) s += (float)a[j] * (float)b[j];
  return s;
}

// ---- work split ------------------------------------------------------------------------------------------------
// The sequence is cut into nt = ceil(T/32) tiles.  A wave owns one tile of its own dimension (queries in the forward
// and dq kernels, keys in the dkv kernel) and loops over all tiles of the other one.  Nine tiles (T = 257: 16x16
// patches + cls) would need nine waves - three on one SIMD, which then sets the pace and caps every wave at 170
// registers.  Instead eight waves (two per SIMD, 256 registers) own tiles 0-7 and share the ninth: each wave runs
// it against its 1-2 tiles of the other dimension and the partial results are combined through LDS (the K/V images
// are dead by then).
#ifdef OCTIC_ATTN_TRACE
// developer-only timeline (build with -DOCTIC_ATTN_TRACE=<first workgroup>): [kernel 0 fwd / 1 dq / 2 dkv][256 workgroups][10 waves][16]
__device__ unsigned long long g_attn_trace[3 * 256 * 10 * 16];
extern "C" void* octic_dbg_attn_trace(void) {
  void* p = nullptr;
  (void)hipGetSymbolAddress(&p, HIP_SYMBOL(g_attn_trace));
  return p;
}
#define ATRACE(kern, slot)                                                                              \
  do {                                                                                                  \
    if (blockIdx.x >= OCTIC_ATTN_TRACE && blockIdx.x < OCTIC_ATTN_TRACE + 256 && (threadIdx.x & 63) == 0 && (slot) < 16)     \
      g_attn_trace[(((kern) * 256 + blockIdx.x - OCTIC_ATTN_TRACE) * 10 + (threadIdx.x >> 6)) * 16 + (slot)] = __builtin_readcyclecounter(); \
  } while (0)
#else
#define ATRACE(kern, slot) do {} while (0)
#endif
constexpr int kAttnWaves = 8;
inline int attn_waves(int nt) { return nt == kAttnWaves + 1 ? kAttnWaves : nt; }

// One pass of the forward over key tiles kt0, kt0+kstep, ... < nt for query tile `qtile`; online softmax state
// (m, l: per lane = per query, l still split between the half-waves) and O^T accumulators are updated in place.
// lane (r, half) holds Q[query][16 ks + 8 half .. +7] = B operand of K Q^T
template <int KS>
__device__ __forceinline__ void load_rows8(bf16x8 (&f)[KS], const bf16* base, int64_t st, int tile, int T, int lane,
                                           const HeadMap m = HeadMap{0, 0}) {
  const int r = lane & 31, half = lane >> 5;
  const int i = tile * 32 + r;
  const int ic = i < T ? i : T - 1;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
    f[ks] = __builtin_bit_cast(bf16x8, hm_load16(base + (int64_t)ic * st, 2 * ks + half, m));
}

template <int KS, int DT>
__device__ __forceinline__ void fwd_pass(const AttnArgs& a, const char* Ks, const char* Vs, int rsk, int rsv,
                                         const bf16x8 (&qf)[KS], int kt0, int kstep, int nt, int lane, float& m,
                                         float& l, f32x16 (&ot)[DT]) {
  const int T = a.T;
  const int r = lane & 31, half = lane >> 5;
  for (int kt = kt0; kt < nt; kt += kstep) {
    f32x16 x;
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = 0.f;
    const char* krow = Ks + (size_t)(kt * 32 + r) * rsk + half * 16;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const bf16x8 kf = *(const bf16x8*)(krow + ks * 32);
      x = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], x, 0, 0, 0);
    }
    // raw scores of query `qi` against keys kt*32 + acc_row(reg, half); the softmax scale (> 0) is applied inside
    // the exponent's fma, the running maximum m is kept in the scaled log2 domain
    if (kt == nt - 1) {
#pragma unroll
      for (int i = 0; i < 16; ++i)
        if (kt * 32 + acc_row(i, half) >= T) x[i] = -INFINITY;
    }
    float mx = fmaxf(fmaxf(x[0], x[1]), x[2]);
#pragma unroll
    for (int i = 3; i < 15; i += 2) mx = fmaxf(fmaxf(mx, x[i]), x[i + 1]);
    mx = fmaxf(mx, x[15]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m, mx * a.scale_log2);   // finite: every key tile has at least one real key
    if (__builtin_amdgcn_ballot_w64(m_new > m)) {      // rescale only when some query's maximum moved
      const float alpha = __builtin_amdgcn_exp2f(m - m_new);
      l *= alpha;
#pragma unroll
      for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int i = 0; i < 16; ++i) ot[d][i] *= alpha;
      m = m_new;
    }
    float ps[16];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      ps[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(x[i], a.scale_log2, -m));
      sum += ps[i];
    }
    l += sum;
    const bf16x8 pb0 = pack8(ps), pb1 = pack8(ps + 8);
#pragma unroll
    for (int d = 0; d < DT; ++d) {
      const bf16x8 v0 = tr_frag(Vs, rsv, kt * 32, d * 32, lane);
      ot[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0, pb0, ot[d], 0, 0, 0);
      const bf16x8 v1 = tr_frag(Vs, rsv, kt * 32 + 16, d * 32, lane);
      ot[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1, pb1, ot[d], 0, 0, 0);
    }
  }
}

// accumulator tile set [DT][16] (row index = d, lane = r) -> f32 image part[r][d] of one wave
template <int DT>
__device__ __forceinline__ void store_partial(float* part, const f32x16 (&acc)[DT], int lane, int nrows) {
  const int r = lane & 31, half = lane >> 5;
  if (r >= nrows) return;
  float* row = part + (size_t)r * (DT * 32 + kPartPad);
#pragma unroll
  for (int d = 0; d < DT; ++d)
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4)
      *(f32x4*)(row + d * 32 + 8 * k4 + 4 * half) =
          f32x4{acc[d][4 * k4], acc[d][4 * k4 + 1], acc[d][4 * k4 + 2], acc[d][4 * k4 + 3]};
}

// Sum the waves' partial images (optionally weighted per wave and row) and store rows [row0, T) as bf16.
// wgt: [waves][32] weights or nullptr; rowscale: [32] final factor per row or nullptr; scale: uniform factor.
template <int DT>
__device__ __forceinline__ void combine_store(const float* parts, int waves, const float* wgt, const float* rowscale,
                                              float scale, bf16* out, int64_t st, int row0, int T, int hd, int tid,
                                              int nthr, const HeadMap m = HeadMap{0, 0}) {
  const int DC = DT * 32 + kPartPad;
  const int c8 = hd / 8;
  const int nrows = T - row0 < 32 ? T - row0 : 32;
  for (int q = tid; q < nrows * c8; q += nthr) {
    const int r = q / c8, c = q - r * c8;
    f32x4 s0 = {0, 0, 0, 0}, s1 = {0, 0, 0, 0};
    for (int w = 0; w < waves; ++w) {
      const float* p = parts + ((size_t)w * 32 + r) * DC + c * 8;
      const float g = wgt ? wgt[w * 32 + r] : 1.f;
      s0 += *(const f32x4*)p * g;
      s1 += *(const f32x4*)(p + 4) * g;
    }
    const float f = scale * (rowscale ? rowscale[r] : 1.f);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      o[j] = (bf16)(s0[j] * f);
      o[4 + j] = (bf16)(s1[j] * f);
    }
    hm_store16(out + (int64_t)(row0 + r) * st, c, __builtin_bit_cast(u32x4, o), m);
  }
}

// DT = ceil(hd / 32) d-tiles of the output, KS = hd / 16 k-steps of the score product
template <int KS, int DT, int MAXT>
__global__ __launch_bounds__(MAXT) void attn_fwd_kernel(AttnArgs a, int rsk, int rsv, int nt) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int T = a.T, hd = a.hd;
  const int W = blockDim.x >> 6;             // waves: one query tile each, + a shared one if nt == W + 1
  const int Tp = nt * 32;
  char* Ks = smem;
  char* Vs = smem + (size_t)Tp * rsk;
  const int bh = unit_of(blockIdx.x, gridDim.x, a.sH < a.sT), b = bh / a.H, h = bh - b * a.H;
  const bf16* qb = a.q + b * a.sB + h * a.sH;
  const bf16* kb = a.k + b * a.sB + h * a.sH;
  const bf16* vb = a.v + b * a.sB + h * a.sH;
  bf16* ob = a.o + b * a.oB + h * a.oH;
  float* lseb = a.lse ? a.lse + ((int64_t)b * a.H + h) * T : nullptr;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = lane & 31, half = lane >> 5;
  ATRACE(0, 0);

  // ---- this wave's query rows first (their loads overlap the staging), then K and V of this head (zero-filled
  // pads: padded keys are masked, padded V rows meet P = 0)
  bf16x8 qf[KS], qfs[KS];                       // own query tile; the shared tile's rows are fetched up front too
  const HeadMaps hm = head_maps(a, h);
  load_rows8<KS>(qf, qb, a.sT, wid, T, lane, hm.q);
  if (nt != W) load_rows8<KS>(qfs, qb, a.sT, W, T, lane, hm.q);
  stage_two(Ks, rsk, kb, a.sT, hd / 8, hd / 8, Vs, rsv, vb, a.sT, hd / 8, (DT * 32) / 8, T, Tp, tid, blockDim.x, hm.k,
            hm.v);
  ATRACE(0, 1);
  __syncthreads();
  ATRACE(0, 2);

  f32x16 ot[DT];
  zero_acc<DT>(ot);
  float m = -INFINITY, l = 0.f;
  fwd_pass<KS, DT>(a, Ks, Vs, rsk, rsv, qf, 0, 1, nt, lane, m, l, ot);
  ATRACE(0, 3);
  l += __shfl_xor(l, 32, 64);
  {
    const int qi = wid * 32 + r;
    if (qi < T) {
      if (half == 0 && lseb) lseb[qi] = m + log2f(l);
      store_rows_wide<DT>(ob + (int64_t)qi * a.oT, ot, 1.0f / l, hd, half, hm.o);
    }
  }
  ATRACE(0, 4);
  if (nt == W) return;

  // ---- the shared last query tile: this wave's share of the keys, then a log-sum-exp merge of the W partials
  zero_acc<DT>(ot);
  m = -INFINITY;
  l = 0.f;
  fwd_pass<KS, DT>(a, Ks, Vs, rsk, rsv, qfs, wid, W, nt, lane, m, l, ot);
  l += __shfl_xor(l, 32, 64);
  __syncthreads();                                   // every wave is done with the K / V images
  const int nrows = T - W * 32;                      // real queries of the shared tile
  float* parts = (float*)smem;                       // [W][32][DT*32 + pad]
  float* ml = parts + (size_t)W * 32 * (DT * 32 + kPartPad);   // m[W][32] | weight[W][32] | 1/L[32]
  store_partial<DT>(parts + (size_t)wid * 32 * (DT * 32 + kPartPad), ot, lane, nrows);
  if (half == 0) {
    ml[wid * 32 + r] = m;
    ml[(W + wid) * 32 + r] = l;
  }
  __syncthreads();
  if (tid < nrows) {
    float M = -INFINITY;
    for (int w = 0; w < W; ++w) M = fmaxf(M, ml[w * 32 + tid]);
    float L = 0.f;
    for (int w = 0; w < W; ++w) {
      const float g = __builtin_amdgcn_exp2f(ml[w * 32 + tid] - M);
      L += ml[(W + w) * 32 + tid] * g;
      ml[(W + w) * 32 + tid] = g;                    // l is consumed: the slot now holds the wave's weight
    }
    ml[2 * W * 32 + tid] = 1.0f / L;
    const int qi = W * 32 + tid;
    if (qi < T && lseb) lseb[qi] = M + log2f(L);
  }
  __syncthreads();
  combine_store<DT>(parts, W, ml + W * 32, ml + 2 * W * 32, 1.0f, ob, a.oT, W * 32, T, hd, tid, blockDim.x, hm.o);
}

// ---------------------------------------------------------------------------------------------------------------
// Persistent forward for nt <= 9 tiles (T <= 258 with at most two rows in the ninth tile: the ViT token counts).
// The kernel above spends more than a third of a workgroup's life loading the head's rows with idle matrix cores
// (s_memtime timeline at (64,16,257,80): load 11k cycles, barrier 1.3k, passes 15.5k, store 1.8k) and the 106 KB of
// images allow no second workgroup on the CU to fill the gap.  Here one workgroup per CU walks over its heads and
// fetches the NEXT head while it computes the current one:
//   * K: two images, the next one filled by LDS-DMA (buffer_load ... lds, 16 B per lane; a wave-instruction fills
//     1 KiB of the padded image, the lanes that fall on the pad chunk or beyond row T read out of the descriptor's
//     range and get zeros) - no registers, no instructions besides the issue;
//   * V: one image (its reads need the zero pad columns), the next head's rows wait in registers (<= 6 x 16 B per
//     lane) from the start of the pass until every wave is done with the current image;
//   * Q rows of the next head are requested right after the last use of the current ones.
// Eight waves (two per SIMD, one query tile each); the ninth tile's one or two queries are shared: wave w runs them
// against key tiles w, w + 8 and wave 0 merges the eight partial rows.  Two barriers per head.
template <int KS, int DT>
__global__ __launch_bounds__(512) void attn_fwd_persist_kernel(AttnArgs a, int rsk, int rsv, int nt, int units) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int T = a.T, hd = a.hd;
  const int W = blockDim.x >> 6;             // min(nt, 8) waves
  const int nthr = blockDim.x;
  const int Tp = nt * 32;
  const int kimg = (Tp * rsk + 1023) & ~1023;          // K image, rounded up to whole DMA instructions
  char* const Vs = smem + 2 * kimg;
  constexpr int DC = DT * 32 + kPartPad;
  const int nrows = T - W * 32;                        // queries of the shared tile (<= 0: none)
  float* const parts = (float*)(Vs + (size_t)Tp * rsv);          // [W][nrows][DC]
  float* const ml = parts + (size_t)W * (nrows > 0 ? nrows : 0) * DC;   // m[W][nrows] | l[W][nrows]
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, half = lane >> 5;
  const int G = gridDim.x;
  constexpr int kc = 2 * KS;                           // 16-byte chunks of a row (hd / 8)

  // ---- K DMA: wave w issues instructions w, w + W, ...; instruction i covers image bytes [1024 i, 1024 i + 1024)
  constexpr int MAXI = 8;
  const int ninstr = kimg >> 10;
  const int cpr = rsk >> 4;                            // chunks per image row (data + pad)
  const bool packed = a.cv_in > 0;                     // head vectors gathered from packed LinearD8 rows (HeadMap)
  const bool ktail_on = packed && kc > 8;              // head_dim 80: groups 8, 9 are gathered remainders (hd 64: none)
  unsigned voff[MAXI];
  unsigned emask = 0;                                  // packed mode: bit j = instruction j of this lane reads an E piece
#pragma unroll
  for (int j = 0; j < MAXI; ++j) {
    const int idx = (wid + j * W) * 64 + lane;
    const int row = idx / cpr, ch = idx - row * cpr;
    if (!packed) {
      voff[j] = (ch < kc && row < T) ? (unsigned)(row * (int)a.sT * 2 + ch * 16) : 0x7FFFFFF0u;
    } else {
      // groups 0-7 are one 16-byte piece each (start + bs or 2 bs elements, added per head); groups 8, 9 are gathered
      // through registers (below), the DMA leaves zeros there
      const int g0 = hm_off8(HeadMap{a.cv_in, 0}, ch < 8 ? ch : 0);
      voff[j] = (ch < 8 && row < T) ? (unsigned)((row * (int)a.sT + g0) * 2) : 0x7FFFFFF0u;
      if (ch >= 4 && ch < 8) emask |= 1u << j;
    }
  }
  const int krec = packed ? (int)((T - 1) * a.sT * 2 + 16 * a.cv_in) : (int)((T - 1) * a.sT * 2 + hd * 2);
  // The DMA goes out as inline assembly on purpose: the compiler protects every transposing LDS read that follows a
  // buffer_load ... lds builtin with s_waitcnt vmcnt(0) (it cannot tell the two K images apart), which would put the
  // arrival of the NEXT head in front of the first P V product of the current one.  The waits that matter are
  // written out below (vmcnt(0) in front of the barrier that hands the image over).
  typedef __attribute__((ext_vector_type(4))) int i32x4;
  auto issue_k = [&](const bf16* kb, unsigned lds_dst, int bs) {
    const uint64_t p = (uint64_t)kb;
    const i32x4 rs = {(int)(uint32_t)p, (int)(uint32_t)((p >> 32) & 0xFFFF), krec, 0x27000};
#pragma unroll
    for (int j = 0; j < MAXI; ++j) {
      const int i = wid + j * W;
      if (i < ninstr) {
        const unsigned vo = voff[j] == 0x7FFFFFF0u ? voff[j] : voff[j] + (unsigned)(bs * (((emask >> j) & 1) ? 4 : 2));
        unsigned keep;   // M0 saved / restored inside the statement (octic_common.hpp: dma16_to_lds)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 1\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "s"(lds_dst + i * 1024), "v"(vo), "s"(rs) : "memory");
      }
    }
  };
  // packed mode: the gathered groups 8 and 9 of the K rows (two 16-byte chunks per row) travel through registers
  u32x4 ktail[2];
  auto load_ktail = [&](const bf16* kb, const HeadMap mk) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      int idx = it * nthr + tid;
      asm volatile("" : "+v"(idx));
      const int row = idx >> 1;
      ktail[it] = u32x4{0, 0, 0, 0};
      if (row < T) ktail[it] = hm_load16(kb + (int64_t)row * a.sT, 8 + (idx & 1), mk);
    }
  };
  auto write_ktail = [&](char* Kimg) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      int idx = it * nthr + tid;
      asm volatile("" : "+v"(idx));
      const int row = idx >> 1;
      if (row < T) *(u32x4*)(Kimg + (size_t)row * rsk + (8 + (idx & 1)) * 16) = ktail[it];
    }
  };
  const unsigned smem_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  // ---- V through registers
  constexpr int NV = 6;
  const int nv = (T * kc + nthr - 1) / nthr;           // <= NV (checked by the launcher)
  u32x4 vreg[NV];
  auto load_v = [&](const bf16* vb, const HeadMap mv) {
#pragma unroll
    for (int it = 0; it < NV; ++it)
      if (it < nv) {
        int idx = it * nthr + tid;
        asm volatile("" : "+v"(idx));                    // recomputed per head: hoisted addresses would cost 2 NV registers
        const int row = idx / kc, c = idx - row * kc;
        vreg[it] = u32x4{0, 0, 0, 0};
        if (row < T) vreg[it] = hm_load16(vb + (int64_t)row * a.sT, c, mv);
      }
  };
  auto write_v = [&]() {
#pragma unroll
    for (int it = 0; it < NV; ++it)
      if (it < nv) {
        int idx = it * nthr + tid;
        asm volatile("" : "+v"(idx));
        const int row = idx / kc, c = idx - row * kc;
        if (row < T) *(u32x4*)(Vs + (size_t)row * rsv + c * 16) = vreg[it];
      }
  };
  const bool shared_rows = a.sH < a.sT;
  auto head_off = [&](int idx, int64_t& in_off, int64_t& o_off, int64_t& st_off, int& h) {
    // work item idx of this launch -> unit; the padded index space keeps `idx + G` on the same XCD
    const int u = unit_of(idx, units, shared_rows);
    const int b = u / a.H;
    h = u - b * a.H;
    in_off = b * a.sB + h * a.sH;
    o_off = b * a.oB + h * a.oH;
    st_off = ((int64_t)b * a.H + h) * T;
  };

  int u = blockIdx.x;
  int64_t in_off, o_off, st_off;
  int hh;
  head_off(u, in_off, o_off, st_off, hh);
  HeadMaps hm = head_maps(a, hh);
  bf16x8 qf[KS], qfs[KS];
  issue_k(a.k + in_off, smem_lds, hm.k.bs);
  if (ktail_on) load_ktail(a.k + in_off, hm.k);
  load_v(a.v + in_off, hm.v);
  load_rows8<KS>(qf, a.q + in_off, a.sT, wid, T, lane, hm.q);
  if (nrows > 0) load_rows8<KS>(qfs, a.q + in_off, a.sT, W, T, lane, hm.q);
  // the V image's pad (columns >= hd, rows >= T) is written once: zeros
  for (int o = tid * 16; o < Tp * rsv; o += nthr * 16) *(u32x4*)(Vs + o) = u32x4{0, 0, 0, 0};
  __syncthreads();
  write_v();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (ktail_on) write_ktail(smem);
  __syncthreads();

  int cur = 0;
  for (; u < units; u += G) {
    const int un = u + G;
    const bool has_next = un < units;
    const char* Ks = smem + cur * kimg;
    const bool tr_on = u == (int)blockIdx.x + G;
    if (tr_on) ATRACE(0, 0);
    int64_t n_in = 0, n_o = 0, n_st = 0;
    // The Q rows were requested a pass ago: settle them HERE, before the prefetch goes out.  Left to the compiler,
    // the wait lands in front of the first MFMA with a count it cannot know (the prefetch instructions are
    // predicated), i.e. vmcnt(0) - and the pass would start only after the next head has arrived.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      asm volatile("" : "+v"(qf[ks]));
      asm volatile("" : "+v"(qfs[ks]));
    }
    int nh = 0;
    HeadMaps nhm = hm;
    if (has_next) {
      head_off(un, n_in, n_o, n_st, nh);
      nhm = head_maps(a, nh);
      issue_k(a.k + n_in, smem_lds + (cur ^ 1) * kimg, nhm.k.bs);
      if (ktail_on) load_ktail(a.k + n_in, nhm.k);
      load_v(a.v + n_in, nhm.v);
    }
    if (tr_on) ATRACE(0, 1);
    bf16* ob = a.o + o_off;
    float* lseb = a.lse ? a.lse + st_off : nullptr;

    f32x16 ot[DT];
    zero_acc<DT>(ot);
    float m = -INFINITY, l = 0.f;
    fwd_pass<KS, DT>(a, Ks, Vs, rsk, rsv, qf, 0, 1, nt, lane, m, l, ot);
    if (tr_on) ATRACE(0, 2);
    l += __shfl_xor(l, 32, 64);
    {
      const int qi = wid * 32 + r;
      if (qi < T) {
        if (half == 0 && lseb) lseb[qi] = m + log2f(l);
        store_rows_wide<DT>(ob + (int64_t)qi * a.oT, ot, 1.0f / l, hd, half, hm.o);
      }
    }
    if (has_next) load_rows8<KS>(qf, a.q + n_in, a.sT, wid, T, lane, nhm.q);
    if (tr_on) ATRACE(0, 3);
    if (nrows > 0) {
      // the shared tile against this wave's key tiles; the partial rows meet in LDS
      zero_acc<DT>(ot);
      m = -INFINITY;
      l = 0.f;
      fwd_pass<KS, DT>(a, Ks, Vs, rsk, rsv, qfs, wid, W, nt, lane, m, l, ot);
      l += __shfl_xor(l, 32, 64);
      if (has_next) load_rows8<KS>(qfs, a.q + n_in, a.sT, W, T, lane, nhm.q);
      if (r < nrows) {
        float* row = parts + ((size_t)wid * nrows + r) * DC;
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
          for (int k4 = 0; k4 < 4; ++k4)
            *(f32x4*)(row + d * 32 + 8 * k4 + 4 * half) =
                f32x4{ot[d][4 * k4], ot[d][4 * k4 + 1], ot[d][4 * k4 + 2], ot[d][4 * k4 + 3]};
        if (half == 0) {
          ml[wid * nrows + r] = m;
          ml[(W + wid) * nrows + r] = l;
        }
      }
    }
    if (tr_on) ATRACE(0, 4);
    __syncthreads();                                   // every wave is done with K[cur] and V
    if (tr_on) ATRACE(0, 5);
    if (has_next) write_v();
    if (tr_on) ATRACE(0, 6);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the K DMA of the next head has landed
    if (ktail_on && has_next) write_ktail(smem + (cur ^ 1) * kimg);
    if (tr_on) ATRACE(0, 7);
    __syncthreads();
    if (tr_on) ATRACE(0, 8);
    if (nrows > 0 && wid == 0) {
      for (int rr = 0; rr < nrows; ++rr) {
        float M = -INFINITY;
        for (int w = 0; w < W; ++w) M = fmaxf(M, ml[w * nrows + rr]);
        float L = 0.f, o0 = 0.f, o1 = 0.f;
        for (int w = 0; w < W; ++w) {
          const float g = __builtin_amdgcn_exp2f(ml[w * nrows + rr] - M);
          L += ml[(W + w) * nrows + rr] * g;
          const float* row = parts + ((size_t)w * nrows + rr) * DC;
          o0 += g * row[lane];
          if (lane + 64 < DT * 32) o1 += g * row[lane + 64];
        }
        const int qi = W * 32 + rr;
        const float inv = 1.0f / L;
        bf16* orow = ob + (int64_t)qi * a.oT;
        if (lane < hd) orow[hm_elem(lane, hm.o)] = (bf16)(o0 * inv);
        if (lane + 64 < hd) orow[hm_elem(lane + 64, hm.o)] = (bf16)(o1 * inv);
        if (lane == 0 && lseb) lseb[qi] = M + log2f(L);
      }
    }
    if (tr_on) ATRACE(0, 9);
    cur ^= 1;
    in_off = n_in; o_off = n_o; st_off = n_st;
    hm = nhm;
  }
}

// routing override OCTIC_ROUTE_ATTN_LEGACY: 1 = round-2 kernels for every shape

template <int KS, int DT>
static int attn_fwd_launch(const AttnArgs& a, int64_t B, hipStream_t s) {
  // the forward is light on registers: one wave per tile even for nine tiles (three waves on one SIMD hide the
  // softmax latency better than the shared-tile split does: 78 vs 85 us at T = 257); the kernel supports both
  const int rsk = attn_rsk(a.hd), rsv = attn_rsv(DT * 32);
  if (!route(OCTIC_ROUTE_ATTN_LEGACY) && attn80_fwd_ok(a)) return attn80_fwd_launch(a, B, s);
  {
    const int nt = (a.T + 31) / 32, W = nt < 8 ? nt : 8;
    const int nrows = a.T - W * 32;
    const int kimg = (nt * 32 * rsk + 1023) & ~1023;
    const size_t need = 2 * (size_t)kimg + (size_t)nt * 32 * rsv +
                        (nrows > 0 ? (size_t)W * nrows * (DT * 32 + kPartPad + 2) * sizeof(float) : 0);
    const bool fits = nt <= 9 && nrows <= 2 && need <= 160 * 1024 && kimg / 1024 <= 8 * W &&
                      (a.T * 2 * KS + W * 64 - 1) / (W * 64) <= 6 && (a.T - 1) * a.sT * 2 + a.hd * 2 < 0x7FFFFFF0ll;
    if (fits) {
      static DeviceOnce once;
      const int cus = device_cus();
      if (once.first()) {
        (void)hipFuncSetAttribute((const void*)attn_fwd_persist_kernel<KS, DT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipGetLastError();
      }
      const int units = (int)(B * a.H);
      attn_fwd_persist_kernel<KS, DT><<<units < cus ? units : cus, W * 64, need, s>>>(a, rsk, rsv, nt, units);
      return launch_status();
    }
  }
  const int nt = (a.T + 31) / 32, W = nt;
  size_t smem = (size_t)nt * 32 * (rsk + rsv);
  const size_t comb = ((size_t)W * 32 * (DT * 32 + kPartPad) + (2 * W + 1) * 32) * sizeof(float);
  if (nt != W && comb > smem) smem = comb;
  if (smem > 160 * 1024) return OCTIC_ESHAPE;
  static DeviceOnce once;
  if (once.first()) {
    (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<KS, DT, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<KS, DT, 640>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipGetLastError();
  }
  if (W <= 8) attn_fwd_kernel<KS, DT, 512><<<(int)(B * a.H), W * 64, smem, s>>>(a, rsk, rsv, nt);
  else attn_fwd_kernel<KS, DT, 640><<<(int)(B * a.H), W * 64, smem, s>>>(a, rsk, rsv, nt);
  return launch_status();
}


// =================================================================================================
// Backward.  P is recomputed from q, k and the saved log-sum-exp (no T x T tensor is ever stored).
//   dV = P^T dO ;  dP = dO V^T ;  dS = P * (dP - delta),  delta[q] = sum_d dO[q,d] O[q,d] ;  dQ = scale dS K ;  dK = scale dS^T Q
// Two kernels so that no gradient needs a cross-wave reduction (beyond the shared ninth tile):
//   attn_bwd_dq_kernel : wave owns 32 QUERIES (same swapped layout as the forward); K and V rows in LDS.
//   attn_bwd_dkv_kernel: wave owns 32 KEYS; Q and dO rows (+ lse, delta) in LDS; un-swapped scores X'[q][key]
//                        keep the key on the lane, so dK^T and dV^T accumulate lane-locally.
// =================================================================================================
// per-query operands of the dq kernel for one query tile: Q and dO fragments, log-sum-exp, delta = <dO, O>
template <int KS>
struct DqRows {
  bf16x8 qf[KS], dof[KS], of[KS];
  float lse, delta;
};
// issue the loads only: the staging loads follow right behind, so the two memory round trips overlap ...
template <int KS>
__device__ __forceinline__ void load_dq_rows(DqRows<KS>& R, const AttnBwdArgs& a, int64_t in_off, int64_t o_off,
                                             int64_t stat_off, int qtile, int lane, const HeadMaps& hm) {
  const int T = a.T;
  const int r = lane & 31, half = lane >> 5;
  const int qi = qtile * 32 + r;
  const int qc = qi < T ? qi : T - 1;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    R.qf[ks] = __builtin_bit_cast(bf16x8, hm_load16(a.q + in_off + (int64_t)qc * a.sT, 2 * ks + half, hm.q));
    R.dof[ks] = __builtin_bit_cast(bf16x8, hm_load16(a.dout + o_off + (int64_t)qc * a.oT, 2 * ks + half, hm.o));
    R.of[ks] = __builtin_bit_cast(bf16x8, hm_load16(a.o + o_off + (int64_t)qc * a.oT, 2 * ks + half, hm.o));
  }
  R.lse = a.lse[stat_off + qc];
}
// ... and delta = <dO, O> once everything has landed
template <int KS>
__device__ __forceinline__ void finish_dq_rows(DqRows<KS>& R, const AttnBwdArgs& a, int64_t stat_off, int qtile,
                                               int lane, bool write_delta) {
  const int r = lane & 31, half = lane >> 5;
  const int qi = qtile * 32 + r;
  float delta = 0.f;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int j = 0; j < 8; ++j) delta += (float)R.dof[ks][j] * (float)R.of[ks][j];
  delta += __shfl_xor(delta, 32, 64);
  R.delta = delta;
  if (write_delta && qi < a.T && half == 0) a.delta[stat_off + qi] = delta;
}

template <int KS, int DT>
__device__ __forceinline__ void dq_pass(const AttnBwdArgs& a, const char* Ks, const char* Vs, int rs,
                                        const DqRows<KS>& R, int kt0, int kstep, int nt, int lane,
                                        f32x16 (&dqt)[DT]) {
  const int T = a.T;
  const int r = lane & 31, half = lane >> 5;
  const float lse = R.lse, delta = R.delta;
  const bf16x8 (&qf)[KS] = R.qf;
  const bf16x8 (&dof)[KS] = R.dof;
  for (int kt = kt0; kt < nt; kt += kstep) {
    f32x16 x, dp;
#pragma unroll
    for (int i = 0; i < 16; ++i) { x[i] = 0.f; dp[i] = 0.f; }
    const char* krow = Ks + (size_t)(kt * 32 + r) * rs + half * 16;
    const char* vrow = Vs + (size_t)(kt * 32 + r) * rs + half * 16;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      x = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(krow + ks * 32), qf[ks], x, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(vrow + ks * 32), dof[ks], dp, 0, 0, 0);
    }
    float ds[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      float p = __builtin_amdgcn_exp2f(x[i] * a.scale_log2 - lse);
      if (kt == nt - 1 && kt * 32 + acc_row(i, half) >= T) p = 0.f;
      ds[i] = p * (dp[i] - delta);
    }
    const bf16x8 b0 = pack8(ds), b1 = pack8(ds + 8);
#pragma unroll
    for (int d = 0; d < DT; ++d) {
      dqt[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(Ks, rs, kt * 32, d * 32, lane), b0, dqt[d], 0, 0, 0);
      dqt[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(Ks, rs, kt * 32 + 16, d * 32, lane), b1, dqt[d], 0, 0, 0);
    }
  }
}

template <int KS, int DT, int MAXT>
__global__ __launch_bounds__(MAXT) void attn_bwd_dq_kernel(AttnBwdArgs a, int rs, int nt) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int T = a.T, hd = a.hd;
  const int W = blockDim.x >> 6, Tp = nt * 32;
  char* Ks = smem;
  char* Vs = smem + (size_t)Tp * rs;
  const int bh = unit_of(blockIdx.x, gridDim.x, a.sH < a.sT), b = bh / a.H, h = bh - b * a.H;
  const int64_t in_off = b * a.sB + h * a.sH, o_off = b * a.oB + h * a.oH;
  const int64_t stat_off = ((int64_t)b * a.H + h) * T;
  bf16* dqb = a.dq + b * a.gB + h * a.gH;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = lane & 31, half = lane >> 5;
  ATRACE(1, 0);
  DqRows<KS> mine, shared;                          // the shared tile's rows are fetched up front too
  const HeadMaps hm = head_maps(a, h);
  // Prologue in ONE memory round trip: Q, dO, O, K and V of the head are all requested before anything is used.
  //  * the wave's own 32 query rows (Q, dO fragments; delta = <dO, O>) come out of LDS images of Q and dO staged with
  //    row-contiguous requests, not from fragment-shaped global loads (lane (r, half) taking 16 bytes of row r = 32
  //    partial lines per instruction, 45 instructions per wave);
  //  * delta: the thread that stages chunk (row, c) of dO also holds chunk (row, c) of O: partial dot products -> LDS;
  //  * the images are then overwritten with K and V, which have been waiting in registers.
  // The shared ninth tile has one real row: its clamped fragment loads touch two lines.
  float* part = (float*)(smem + (size_t)2 * Tp * rs);          // [Tp][hd / 8] delta partials
  const int kc = hd / 8;
  if (hm.q.cv == 0 || hd != 80) {                    // plain rows, or packed rows whose 16-byte groups are whole pieces (hd 64)
    StagePlain qd, kv;
    u32x4 oo[kStageRows];
    const StageMap sm = stage_map(tid, blockDim.x, kc);
    const int c = sm.c, t0 = sm.t0, tstep = sm.tstep;
    stage_request(qd, a.q + in_off, a.sT, kc, a.dout + o_off, a.oT, kc, T, tid, blockDim.x, hm.q, hm.o);
#pragma unroll
    for (int it = 0; it < kStageRows; ++it) {
      const int t = t0 + it * tstep;
      oo[it] = u32x4{0, 0, 0, 0};
      if (t < T && c < kc) oo[it] = hm_load16(a.o + o_off + (int64_t)t * a.oT, c, hm.o);
    }
    stage_request(kv, a.k + in_off, a.sT, kc, a.v + in_off, a.sT, kc, T, tid, blockDim.x, hm.k, hm.v);
    stage_write(qd, Ks, rs, kc, Vs, rs, kc, Tp, tid, blockDim.x);
#pragma unroll
    for (int it = 0; it < kStageRows; ++it) {
      const int t = t0 + it * tstep;
      if (t < T && c < kc) part[t * kc + c] = dot8(qd.b[it], oo[it]);
    }
    __syncthreads();
    if (nt != W) load_dq_rows<KS>(shared, a, in_off, o_off, stat_off, W, lane, hm);   // (the staging registers are free now)
    {
      const int qi = wid * 32 + r, qc = qi < T ? qi : T - 1;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        mine.qf[ks] = *(const bf16x8*)(Ks + (size_t)qc * rs + (2 * ks + half) * 16);
        mine.dof[ks] = *(const bf16x8*)(Vs + (size_t)qc * rs + (2 * ks + half) * 16);
      }
      mine.lse = a.lse[stat_off + qc];
      float delta = 0.f;
      for (int cc = 0; cc < kc; ++cc) delta += part[qc * kc + cc];
      mine.delta = delta;
      if (qi < T && half == 0) a.delta[stat_off + qi] = delta;
    }
    __syncthreads();                                 // every wave has its fragments: the images may be overwritten
    stage_write(kv, Ks, rs, kc, Vs, rs, kc, Tp, tid, blockDim.x);
    if (nt != W) finish_dq_rows<KS>(shared, a, stat_off, W, lane, wid == 0);
  } else {
    StagePacked qd, kv, oo;                          // oo.a*: the O row of this thread (its b half is unused)
    stage_request(qd, a.q + in_off, a.sT, a.dout + o_off, a.oT, T, tid, hm.q, hm.o);
    stage_request(oo, a.o + o_off, a.oT, a.o + o_off, a.oT, T, tid, hm.o, hm.o);
    stage_request(kv, a.k + in_off, a.sT, a.v + in_off, a.sT, T, tid, hm.k, hm.v);
    stage_write(qd, Ks, rs, kc, kc, Vs, rs, kc, kc, T, Tp, tid, blockDim.x);
    if (tid < T) {                                    // the whole row of dO and of O sits in this thread
      float sum = 0.f;
#pragma unroll
      for (int pz = 0; pz < 4; ++pz) {
        sum += dot8(qd.b4[pz], oo.a4[pz]);
        const bf16 *x = (const bf16*)&qd.b1[pz], *y = (const bf16*)&oo.a1[pz];
        sum += (float)x[0] * (float)y[0] + (float)x[1] * (float)y[1];
      }
#pragma unroll
      for (int pz = 0; pz < 2; ++pz) {
        sum += dot8(qd.be[pz][0], oo.ae[pz][0]) + dot8(qd.be[pz][1], oo.ae[pz][1]);
        const bf16 *x = (const bf16*)&qd.b2[pz], *y = (const bf16*)&oo.a2[pz];
#pragma unroll
        for (int j = 0; j < 4; ++j) sum += (float)x[j] * (float)y[j];
      }
      part[tid] = sum;
    }
    __syncthreads();
    if (nt != W) load_dq_rows<KS>(shared, a, in_off, o_off, stat_off, W, lane, hm);   // (the staging registers are free now)
    {
      const int qi = wid * 32 + r, qc = qi < T ? qi : T - 1;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        mine.qf[ks] = *(const bf16x8*)(Ks + (size_t)qc * rs + (2 * ks + half) * 16);
        mine.dof[ks] = *(const bf16x8*)(Vs + (size_t)qc * rs + (2 * ks + half) * 16);
      }
      mine.lse = a.lse[stat_off + qc];
      mine.delta = part[qc];
      if (qi < T && half == 0) a.delta[stat_off + qi] = mine.delta;
    }
    __syncthreads();
    stage_write(kv, Ks, rs, kc, kc, Vs, rs, kc, kc, T, Tp, tid, blockDim.x);
    if (nt != W) finish_dq_rows<KS>(shared, a, stat_off, W, lane, wid == 0);
  }
  ATRACE(1, 1);
  __syncthreads();
  ATRACE(1, 2);

  f32x16 dqt[DT];
  zero_acc<DT>(dqt);
  dq_pass<KS, DT>(a, Ks, Vs, rs, mine, 0, 1, nt, lane, dqt);
  ATRACE(1, 3);
  if (wid * 32 + r < T) store_rows_wide<DT>(dqb + (int64_t)(wid * 32 + r) * a.gT, dqt, a.scale, hd, half, hm.q);
  ATRACE(1, 4);
  if (nt == W) return;

  zero_acc<DT>(dqt);
  dq_pass<KS, DT>(a, Ks, Vs, rs, shared, wid, W, nt, lane, dqt);
  ATRACE(1, 5);
  __syncthreads();
  ATRACE(1, 6);
  float* parts = (float*)smem;
  store_partial<DT>(parts + (size_t)wid * 32 * (DT * 32 + kPartPad), dqt, lane, T - W * 32);
  __syncthreads();
  combine_store<DT>(parts, W, nullptr, nullptr, a.scale, dqb, a.gT, W * 32, T, hd, tid, blockDim.x, hm.q);
  ATRACE(1, 7);
}

// dK^T, dV^T of key tile `ktile` accumulated over query tiles qt0, qt0+qstep, ...
// the wave's key rows: lane (r, half) holds K[key][16 ks + 8 half ..] and V[key][..] = B operands (key on the lane)
template <int KS>
struct KvRows {
  bf16x8 kf[KS], vf[KS];
};
template <int KS>
__device__ __forceinline__ void load_kv_rows(KvRows<KS>& R, const AttnBwdArgs& a, int64_t in_off, int ktile, int lane,
                                             const HeadMaps& hm) {
  const int r = lane & 31, half = lane >> 5;
  const int ki = ktile * 32 + r;
  const int kcl = ki < a.T ? ki : a.T - 1;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    R.kf[ks] = __builtin_bit_cast(bf16x8, hm_load16(a.k + in_off + (int64_t)kcl * a.sT, 2 * ks + half, hm.k));
    R.vf[ks] = __builtin_bit_cast(bf16x8, hm_load16(a.v + in_off + (int64_t)kcl * a.sT, 2 * ks + half, hm.v));
  }
}

template <int KS, int DT>
__device__ __forceinline__ void dkv_pass(const AttnBwdArgs& a, const char* Qs, const char* Ds, const float* lse_s,
                                         const float* del_s, int rs, const KvRows<KS>& R, int qt0, int qstep, int nt,
                                         int lane, f32x16 (&dkt)[DT], f32x16 (&dvt)[DT]) {
  const int r = lane & 31, half = lane >> 5;
  const bf16x8 (&kf)[KS] = R.kf;
  const bf16x8 (&vf)[KS] = R.vf;
  for (int qt = qt0; qt < nt; qt += qstep) {
    f32x16 x, dp;
#pragma unroll
    for (int i = 0; i < 16; ++i) { x[i] = 0.f; dp[i] = 0.f; }
    const char* qrow = Qs + (size_t)(qt * 32 + r) * rs + half * 16;
    const char* drow = Ds + (size_t)(qt * 32 + r) * rs + half * 16;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      x = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(qrow + ks * 32), kf[ks], x, 0, 0, 0);    // X'[q][key]
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(drow + ks * 32), vf[ks], dp, 0, 0, 0);  // dP[q][key]
    }
    float ps[16], ds[16];
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int q0 = qt * 32 + 8 * g4 + 4 * half;            // accumulator rows 4*g4 .. 4*g4+3 are queries q0 .. q0+3
      const f32x4 l4 = *(const f32x4*)(lse_s + q0), d4 = *(const f32x4*)(del_s + q0);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int i = 4 * g4 + j;
        const float p = __builtin_amdgcn_exp2f(x[i] * a.scale_log2 - l4[j]);
        ps[i] = p;
        ds[i] = p * (dp[i] - d4[j]);
      }
    }
    const bf16x8 p0 = pack8(ps), p1 = pack8(ps + 8), s0 = pack8(ds), s1 = pack8(ds + 8);
#pragma unroll
    for (int d = 0; d < DT; ++d) {
      dvt[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(Ds, rs, qt * 32, d * 32, lane), p0, dvt[d], 0, 0, 0);
      dvt[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(Ds, rs, qt * 32 + 16, d * 32, lane), p1, dvt[d], 0, 0, 0);
      dkt[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(Qs, rs, qt * 32, d * 32, lane), s0, dkt[d], 0, 0, 0);
      dkt[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(Qs, rs, qt * 32 + 16, d * 32, lane), s1, dkt[d], 0, 0, 0);
    }
  }
}

template <int KS, int DT, int MAXT>
__global__ __launch_bounds__(MAXT) void attn_bwd_dkv_kernel(AttnBwdArgs a, int rs, int nt) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int T = a.T, hd = a.hd;
  const int W = blockDim.x >> 6, Tp = nt * 32;
  char* Qs = smem;
  char* Ds = smem + (size_t)Tp * rs;
  float* lse_s = (float*)(smem + (size_t)2 * Tp * rs);
  float* del_s = lse_s + Tp;
  const int bh = unit_of(blockIdx.x, gridDim.x, a.sH < a.sT), b = bh / a.H, h = bh - b * a.H;
  const int64_t in_off = b * a.sB + h * a.sH, o_off = b * a.oB + h * a.oH;
  bf16* dkb = a.dk + b * a.gB + h * a.gH;
  bf16* dvb = a.dv + b * a.gB + h * a.gH;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = lane & 31, half = lane >> 5;
  ATRACE(2, 0);
  KvRows<KS> kv;
  const HeadMaps hm = head_maps(a, h);
  // Prologue in one memory round trip (see attn_bwd_dq_kernel): K, V, Q and dO are all requested up front; the own key
  // rows' fragments are read from the LDS images of K and V, which are then overwritten with Q and dO.
  const int kc = hd / 8;
  if (hm.q.cv == 0 || hd != 80) {
    StagePlain kvr, qdr;
    stage_request(kvr, a.k + in_off, a.sT, kc, a.v + in_off, a.sT, kc, T, tid, blockDim.x, hm.k, hm.v);
    stage_request(qdr, a.q + in_off, a.sT, kc, a.dout + o_off, a.oT, kc, T, tid, blockDim.x, hm.q, hm.o);
    stage_write(kvr, Qs, rs, kc, Ds, rs, kc, Tp, tid, blockDim.x);
    __syncthreads();
    {
      const int ki = wid * 32 + r, kcl = ki < T ? ki : T - 1;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        kv.kf[ks] = *(const bf16x8*)(Qs + (size_t)kcl * rs + (2 * ks + half) * 16);
        kv.vf[ks] = *(const bf16x8*)(Ds + (size_t)kcl * rs + (2 * ks + half) * 16);
      }
    }
    __syncthreads();
    stage_write(qdr, Qs, rs, kc, Ds, rs, kc, Tp, tid, blockDim.x);
  } else {
    StagePacked kvr, qdr;
    stage_request(kvr, a.k + in_off, a.sT, a.v + in_off, a.sT, T, tid, hm.k, hm.v);
    stage_request(qdr, a.q + in_off, a.sT, a.dout + o_off, a.oT, T, tid, hm.q, hm.o);
    stage_write(kvr, Qs, rs, kc, kc, Ds, rs, kc, kc, T, Tp, tid, blockDim.x);
    __syncthreads();
    {
      const int ki = wid * 32 + r, kcl = ki < T ? ki : T - 1;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        kv.kf[ks] = *(const bf16x8*)(Qs + (size_t)kcl * rs + (2 * ks + half) * 16);
        kv.vf[ks] = *(const bf16x8*)(Ds + (size_t)kcl * rs + (2 * ks + half) * 16);
      }
    }
    __syncthreads();
    stage_write(qdr, Qs, rs, kc, kc, Ds, rs, kc, kc, T, Tp, tid, blockDim.x);
  }
  for (int t = tid; t < Tp; t += blockDim.x) {
    const int64_t stat = ((int64_t)b * a.H + h) * T + t;
    lse_s[t] = t < T ? a.lse[stat] : INFINITY;    // padded queries: P = exp2(x - inf) = 0
    del_s[t] = t < T ? a.delta[stat] : 0.f;
  }
  ATRACE(2, 1);
  __syncthreads();
  ATRACE(2, 2);

  f32x16 dkt[DT], dvt[DT];
  zero_acc<DT>(dkt);
  zero_acc<DT>(dvt);
  dkv_pass<KS, DT>(a, Qs, Ds, lse_s, del_s, rs, kv, 0, 1, nt, lane, dkt, dvt);
  ATRACE(2, 3);
  if (wid * 32 + r < T) {
    store_rows_wide<DT>(dkb + (int64_t)(wid * 32 + r) * a.gT, dkt, a.scale, hd, half, hm.k);
    store_rows_wide<DT>(dvb + (int64_t)(wid * 32 + r) * a.gT, dvt, 1.0f, hd, half, hm.v);
  }
  ATRACE(2, 4);
  if (nt == W) return;

  zero_acc<DT>(dkt);
  zero_acc<DT>(dvt);
  load_kv_rows<KS>(kv, a, in_off, W, lane, hm);
  dkv_pass<KS, DT>(a, Qs, Ds, lse_s, del_s, rs, kv, wid, W, nt, lane, dkt, dvt);
  ATRACE(2, 5);
  __syncthreads();
  ATRACE(2, 6);
  float* parts = (float*)smem;
  float* mine = parts + (size_t)wid * 32 * (DT * 32 + kPartPad);
  store_partial<DT>(mine, dkt, lane, T - W * 32);
  __syncthreads();
  combine_store<DT>(parts, W, nullptr, nullptr, a.scale, dkb, a.gT, W * 32, T, hd, tid, blockDim.x, hm.k);
  __syncthreads();
  store_partial<DT>(mine, dvt, lane, T - W * 32);
  __syncthreads();
  combine_store<DT>(parts, W, nullptr, nullptr, 1.0f, dvb, a.gT, W * 32, T, hd, tid, blockDim.x, hm.v);
  ATRACE(2, 7);
}

template <int KS, int DT>
static int attn_bwd_launch(const AttnBwdArgs& a, int64_t B, int phase, hipStream_t s) {
  // dkv (two accumulator sets) needs the 256-register budget of the eight-wave split (105 vs 156 us with spills);
  // dq runs the same either way (~106 us) and follows it
  // both gradients asked for at once: the single-pass kernel (csrc/attn80_bwd.hip) where its shape applies
  if (phase == 3 && KS == 5 && !route(OCTIC_ROUTE_ATTN_LEGACY) && attn80_bwd_ok(a)) return attn80_bwd_launch(a, B, s);
  const int nt = (a.T + 31) / 32, W = attn_waves(nt), Wq = W;
  // the row images are read both by rows (ds_read_b128) and transposed (ds_read_b64_tr_b16); rows are padded so the
  // b128 reads are conflict-free, the transposed reads then see at most 2-way conflicts.  The tr fragments reach
  // DT*32 columns, so rows must hold that many (the pad columns meet zero accumulator columns / are discarded).
  const int cols = DT * 32 > a.hd ? DT * 32 : a.hd;
  const int rs = cols * 2 + 16;
  const size_t img = (size_t)2 * nt * 32 * rs;
  size_t smem_dq = img + (size_t)nt * 32 * (a.hd / 8) * sizeof(float);      // + the delta partials [Tp][hd / 8]
  size_t smem_kv = img + (size_t)2 * nt * 32 * sizeof(float);
  const size_t comb = (size_t)W * 32 * (DT * 32 + kPartPad) * sizeof(float);
  if (nt != W && comb > smem_dq) smem_dq = comb;
  if (nt != W && comb > smem_kv) smem_kv = comb;
  if (smem_kv > 160 * 1024 || smem_dq > 160 * 1024) return OCTIC_ESHAPE;
  static DeviceOnce once;
  if (once.first()) {
    (void)hipFuncSetAttribute((const void*)attn_bwd_dq_kernel<KS, DT, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel<KS, DT, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)attn_bwd_dq_kernel<KS, DT, 640>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel<KS, DT, 640>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipGetLastError();
  }
  if (phase & 1) {
    if (Wq <= 8) attn_bwd_dq_kernel<KS, DT, 512><<<(int)(B * a.H), Wq * 64, smem_dq, s>>>(a, rs, nt);
    else attn_bwd_dq_kernel<KS, DT, 640><<<(int)(B * a.H), Wq * 64, smem_dq, s>>>(a, rs, nt);
  }
  if (phase & 2) {
    if (W <= 8) attn_bwd_dkv_kernel<KS, DT, 512><<<(int)(B * a.H), W * 64, smem_kv, s>>>(a, rs, nt);
    else attn_bwd_dkv_kernel<KS, DT, 640><<<(int)(B * a.H), W * 64, smem_kv, s>>>(a, rs, nt);
  }
  return launch_status();
}

}  // namespace octic

using namespace octic;

extern "C" {

int octic_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse, int64_t B, int H, int T, int hd,
                   int64_t sB, int64_t sH, int64_t sT, int64_t oB, int64_t oH, int64_t oT, float scale, void* stream) {
  if (!q || !k || !v || !o) return OCTIC_ENULL;
  if (B <= 0 || H <= 0 || T <= 0 || T > 320 || hd <= 0 || (hd % 16) || hd > 128) return OCTIC_ESHAPE;
  if ((((uintptr_t)q) | ((uintptr_t)k) | ((uintptr_t)v) | ((uintptr_t)o)) & 15) return OCTIC_EALIGN;
  if ((sB | sH | sT | oB | oH | oT) & 7) return OCTIC_EALIGN;   // rows must stay 16-byte aligned (8 bf16)
  AttnArgs a;
  a.q = (const bf16*)q; a.k = (const bf16*)k; a.v = (const bf16*)v;
  a.sB = sB; a.sH = sH; a.sT = sT;
  a.o = (bf16*)o; a.oB = oB; a.oH = oH; a.oT = oT;
  a.lse = lse;
  a.H = H; a.T = T; a.hd = hd;
  a.scale_log2 = scale * 1.4426950408889634f;
  a.cv_in = a.cv_out = a.c = 0;
  hipStream_t s = (hipStream_t)stream;
  switch (hd / 16) {
    case 1: return attn_fwd_launch<1, 1>(a, B, s);
    case 2: return attn_fwd_launch<2, 1>(a, B, s);
    case 3: return attn_fwd_launch<3, 2>(a, B, s);
    case 4: return attn_fwd_launch<4, 2>(a, B, s);
    case 5: return attn_fwd_launch<5, 3>(a, B, s);
    case 6: return attn_fwd_launch<6, 3>(a, B, s);
    case 7: return attn_fwd_launch<7, 4>(a, B, s);
    default: return attn_fwd_launch<8, 4>(a, B, s);
  }
}

int octic_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
                   float* delta, void* dq, void* dk, void* dv, int64_t B, int H, int T, int hd, int64_t sB, int64_t sH,
                   int64_t sT, int64_t oB, int64_t oH, int64_t oT, int64_t gB, int64_t gH, int64_t gT, float scale,
                   int phase, void* stream) {
  if (!q || !k || !v || !o || !dout || !lse || !delta || !dq || !dk || !dv) return OCTIC_ENULL;
  if (phase < 1 || phase > 3) return OCTIC_ESHAPE;
  if (B <= 0 || H <= 0 || T <= 0 || T > 320 || hd <= 0 || (hd % 16) || hd > 128) return OCTIC_ESHAPE;
  if ((((uintptr_t)q) | ((uintptr_t)k) | ((uintptr_t)v) | ((uintptr_t)o) | ((uintptr_t)dout) | ((uintptr_t)dq) |
       ((uintptr_t)dk) | ((uintptr_t)dv)) & 15)
    return OCTIC_EALIGN;
  if ((sB | sH | sT | oB | oH | oT | gB | gH | gT) & 7) return OCTIC_EALIGN;
  AttnBwdArgs a;
  a.q = (const bf16*)q; a.k = (const bf16*)k; a.v = (const bf16*)v; a.sB = sB; a.sH = sH; a.sT = sT;
  a.o = (const bf16*)o; a.dout = (const bf16*)dout; a.oB = oB; a.oH = oH; a.oT = oT;
  a.lse = lse; a.delta = delta;
  a.dq = (bf16*)dq; a.dk = (bf16*)dk; a.dv = (bf16*)dv; a.gB = gB; a.gH = gH; a.gT = gT;
  a.H = H; a.T = T; a.hd = hd;
  a.scale = scale;
  a.scale_log2 = scale * 1.4426950408889634f;
  a.cv_in = a.cv_out = a.c = 0;
  hipStream_t s = (hipStream_t)stream;
  switch (hd / 16) {
    case 1: return attn_bwd_launch<1, 1>(a, B, phase, s);
    case 2: return attn_bwd_launch<2, 1>(a, B, phase, s);
    case 3: return attn_bwd_launch<3, 2>(a, B, phase, s);
    case 4: return attn_bwd_launch<4, 2>(a, B, phase, s);
    case 5: return attn_bwd_launch<5, 3>(a, B, phase, s);
    case 6: return attn_bwd_launch<6, 3>(a, B, phase, s);
    case 7: return attn_bwd_launch<7, 4>(a, B, phase, s);
    default: return attn_bwd_launch<8, 4>(a, B, phase, s);
  }
}

// AttentionD8 on PACKED rows (reference d8_layers.py:631-656 without the pack / unpack copies): qkv is the LinearD8
// output [B, T, 3*8c] (row stride ld_qkv elements), o the packed [B, T, 8c] input of the output projection (row stride
// ld_o); head h of tensor s takes c/H channels of every one-dimensional irrep and 2c/H of each E row.  c/H must be 10
// (head_dim 80: the piece schedule of HeadMap) or 8 (head_dim 64, ViT-L/16: every 16-byte group of a head vector is a
// whole piece - A1, A2, B1, B2, two halves of each E row - so only the g < 8 branch of HeadMap is used), bf16.
int octic_attn_fwd_packed(const void* qkv, void* o, float* lse, int64_t B, int H, int T, int c, int64_t ld_qkv,
                          int64_t ld_o, float scale, void* stream) {
  if (!qkv || !o) return OCTIC_ENULL;
  if (B <= 0 || H <= 0 || T <= 0 || T > 320 || c <= 0 || (c != 10 * H && c != 8 * H)) return OCTIC_ESHAPE;
  if (((((uintptr_t)qkv) | ((uintptr_t)o)) & 15) || ((ld_qkv | ld_o) & 7) || ld_qkv < 24 * c || ld_o < 8 * c) return OCTIC_EALIGN;
  AttnArgs a;
  a.q = a.k = a.v = (const bf16*)qkv;
  a.sB = (int64_t)T * ld_qkv; a.sH = 0; a.sT = ld_qkv;
  a.o = (bf16*)o; a.oB = (int64_t)T * ld_o; a.oH = 0; a.oT = ld_o;
  a.lse = lse;
  a.H = H; a.T = T; a.hd = 8 * (c / H);
  a.scale_log2 = scale * 1.4426950408889634f;
  a.cv_in = 3 * c; a.cv_out = c; a.c = c;
  if (a.hd == 64) return attn_fwd_launch<4, 2>(a, B, (hipStream_t)stream);
  return attn_fwd_launch<5, 3>(a, B, (hipStream_t)stream);
}

// Backward of the above: dqkv (packed like qkv, row stride ld_g) receives dq | dk | dv; dout packed like o.
int octic_attn_bwd_packed(const void* qkv, const void* o, const void* dout, const float* lse, float* delta, void* dqkv,
                          int64_t B, int H, int T, int c, int64_t ld_qkv, int64_t ld_o, int64_t ld_g, float scale,
                          int phase, void* stream) {
  if (!qkv || !o || !dout || !lse || !delta || !dqkv) return OCTIC_ENULL;
  if (phase < 1 || phase > 3) return OCTIC_ESHAPE;
  if (B <= 0 || H <= 0 || T <= 0 || T > 320 || c <= 0 || (c != 10 * H && c != 8 * H)) return OCTIC_ESHAPE;
  if (((((uintptr_t)qkv) | ((uintptr_t)o) | ((uintptr_t)dout) | ((uintptr_t)dqkv)) & 15) || ((ld_qkv | ld_o | ld_g) & 7) ||
      ld_qkv < 24 * c || ld_g < 24 * c || ld_o < 8 * c)
    return OCTIC_EALIGN;
  AttnBwdArgs a;
  a.q = a.k = a.v = (const bf16*)qkv; a.sB = (int64_t)T * ld_qkv; a.sH = 0; a.sT = ld_qkv;
  a.o = (const bf16*)o; a.dout = (const bf16*)dout; a.oB = (int64_t)T * ld_o; a.oH = 0; a.oT = ld_o;
  a.lse = lse; a.delta = delta;
  a.dq = a.dk = a.dv = (bf16*)dqkv; a.gB = (int64_t)T * ld_g; a.gH = 0; a.gT = ld_g;
  a.H = H; a.T = T; a.hd = 8 * (c / H);
  a.scale = scale;
  a.scale_log2 = scale * 1.4426950408889634f;
  a.cv_in = 3 * c; a.cv_out = c; a.c = c;
  if (a.hd == 64) return attn_bwd_launch<4, 2>(a, B, phase, (hipStream_t)stream);
  return attn_bwd_launch<5, 3>(a, B, phase, (hipStream_t)stream);
}

}  // extern "C"
