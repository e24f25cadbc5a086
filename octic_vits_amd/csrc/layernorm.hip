// LayerNormD8 (+AffineD8) forward / backward for gfx950.  One wave64 per token row; the row
// (8c f32 = 5 KiB at ViT-H) lives in registers between the statistics pass and the normalise pass,
// so HBM traffic is exactly one read of x and one write of y (fwd), read g,x(,dres) + write dx (bwd).
// Everything that depends only on the lane (which irrep segment each of its 16-byte chunks belongs to,
// the chunk's address inside the token row, its alpha/beta values) is computed ONCE before the row loop
// and kept in registers; the chunk count per lane is a template parameter (NV = 5 at ViT-H) so nothing
// is predicated at run time.  Bound: HBM.
// Reference: octic_vits/d8_layers.py:161-186 (forward), backward derived in SURVEY.md §10.3.
#include <stdlib.h>
#include <type_traits>
#include "octic_common.hpp"

namespace octic {

template <typename T>
__device__ inline f32x4 load4(const T* p);
template <>
__device__ inline f32x4 load4<float>(const float* p) { return *(const f32x4*)p; }
template <>
__device__ inline f32x4 load4<bf16>(const bf16* p) {
  bf16x4 a = *(const bf16x4*)p;
  f32x4 r = {(float)a[0], (float)a[1], (float)a[2], (float)a[3]};
  return r;
}
template <typename T>
__device__ inline void store4(T* p, f32x4 v);
template <>
__device__ inline void store4<float>(float* p, f32x4 v) { *(f32x4*)p = v; }
template <>
__device__ inline void store4<bf16>(bf16* p, f32x4 v) {
  bf16x4 a = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
  *(bf16x4*)p = a;
}

// Per-lane, row-independent description of chunk i (logical packed columns e..e+3, e = 4*(lane+64 i)).
template <int NV>
struct LaneMap {
  int seg[NV];        // 0..5 (A1,A2,B1,B2,E_row0,E_row1); -1 = beyond the row
  int grp[NV];        // view tensor 0..4
  int off[NV];        // element offset inside that tensor's token row
  int aidx[NV];       // index into the segment's alpha
};

template <int NV>
__device__ inline void make_lane_map(LaneMap<NV>& lm, int lane, int c) {
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int e = (lane + 64 * i) * 4;
    if (e >= 8 * c) {
      lm.seg[i] = -1; lm.grp[i] = 0; lm.off[i] = 0; lm.aidx[i] = 0;
    } else if (e < 4 * c) {
      const int g = e / c;
      lm.seg[i] = g; lm.grp[i] = g; lm.off[i] = e - g * c; lm.aidx[i] = e - g * c;
    } else {
      const int o = e - 4 * c, row = o / (2 * c);
      lm.seg[i] = 4 + row; lm.grp[i] = 4; lm.off[i] = o; lm.aidx[i] = o - row * 2 * c;
    }
  }
}

__device__ inline float pick6(const float v[6], int seg) {
  float r = v[0];
#pragma unroll
  for (int k = 1; k < 6; ++k) r = (seg == k) ? v[k] : r;
  return r;
}

// PK: both views are one packed [M, 8c] tensor each (p[g] = p[0] + column offset, equal row strides) - the layout the
// engine itself always uses - so a chunk's address is row base + lane*16 B + i*1 KiB and nothing per-chunk has to
// be kept in registers; the generic variant serves foreign 5-tuple views.
template <typename TOUT, int NV, bool PK>
__global__ __launch_bounds__(256, PK ? 4 : 2) void ln_fwd_kernel(View x, View y, const float* a0, const float* a1, const float* a2,
                                                     const float* a3, const float* a4, const float* beta,
                                                     float* stats, int64_t M, int c, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  LaneMap<NV> lm;
  make_lane_map<NV>(lm, lane, c);
  const float* alpha[5] = {a0, a1, a2, a3, a4};
  f32x4 av[NV], bv[NV];
  const float* xb[NV];
  TOUT* yb[NV];
  int64_t xld[NV], yld[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    av[i] = f32x4{1.f, 1.f, 1.f, 1.f};
    bv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int g = lm.grp[i];
    if (lm.seg[i] >= 0) {
      if (a0) av[i] = *(const f32x4*)(alpha[g] + lm.aidx[i]);
      if (beta && lm.seg[i] == 0) bv[i] = *(const f32x4*)(beta + lm.aidx[i]);
    }
    if constexpr (PK) {
      xb[i] = (const float*)x.p[0] + (lane + 64 * i) * 4;
      yb[i] = (TOUT*)y.p[0] + (lane + 64 * i) * 4;
      xld[i] = x.ld[0];
      yld[i] = y.ld[0];
    } else {
      xb[i] = (const float*)x.p[g] + lm.off[i];
      yb[i] = (TOUT*)y.p[g] + lm.off[i];
      xld[i] = x.ld[g];
      yld[i] = y.ld[g];
    }
  }
  const float inv_n1 = 1.0f / (float)c, inv_n2 = 1.0f / (float)(2 * c);
  for (int64_t m = wave; m < M; m += nwaves) {
    f32x4 xv[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i)
      xv[i] = lm.seg[i] >= 0 ? *(const f32x4*)(xb[i] + m * xld[i]) : f32x4{0.f, 0.f, 0.f, 0.f};
    float s[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const float t = xv[i][0] + xv[i][1] + xv[i][2] + xv[i][3];
#pragma unroll
      for (int k = 0; k < 6; ++k) s[k] += (lm.seg[i] == k) ? t : 0.f;
    }
    float mean[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) mean[k] = wave_sum(s[k]) * (k < 4 ? inv_n1 : inv_n2);
    float q[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const float mu = pick6(mean, lm.seg[i]);
      float t = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        xv[i][j] -= mu;  // centred from here on
        t += xv[i][j] * xv[i][j];
      }
#pragma unroll
      for (int k = 0; k < 6; ++k) q[k] += (lm.seg[i] == k) ? t : 0.f;
    }
    float S = eps;
#pragma unroll
    for (int k = 0; k < 6; ++k) S += wave_sum(q[k]) * (k < 4 ? inv_n1 : 0.5f * inv_n2);
    const float rstd = 1.0f / (kSqrt2Over4 * sqrtf(S));
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if (lm.seg[i] >= 0) {
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = xv[i][j] * rstd * av[i][j] + bv[i][j];
        store4<TOUT>(yb[i] + m * yld[i], o);
      }
    }
    if (stats && lane < 8) {
      float v = (lane == 6) ? rstd : 0.f;
#pragma unroll
      for (int k = 0; k < 6; ++k) v = (lane == k) ? mean[k] : v;
      stats[m * 8 + lane] = v;
    }
  }
}

// Backward.  Per row:  ghat = alpha*g ; xhat = (x-mu)*rstd ;
//   dstd = -rstd * sum(ghat*xhat) ; dS = dstd*rstd/16 ;
//   dx_s = (ghat_s - mean(ghat_s))*rstd + w_s*dS*2*(x_s-mu_s)/n_s       (w = 1 | 1/2, n = c | 2c)
// Parameter partials: dalpha += g*xhat, dbeta += g, accumulated per wave in registers over its
// rows, then reduced over the block's 4 waves through LDS into partials[blk][2][8c].
#ifndef OCTIC_LNBWD_WAVES
#define OCTIC_LNBWD_WAVES 8
#endif
#ifndef OCTIC_LNBWD_OCC
#define OCTIC_LNBWD_OCC 4
#endif
constexpr int kLnBwdWaves = OCTIC_LNBWD_WAVES;   // 512 slabs x 8 waves = 4 waves per SIMD: enough rows in flight to cover HBM latency

template <typename TG, int NV, bool PK>
__global__ __launch_bounds__(kLnBwdWaves * 64, PK ? OCTIC_LNBWD_OCC : 2) void ln_bwd_kernel(View g, View x, const float* stats, const float* a0,
                                                     const float* a1, const float* a2, const float* a3,
                                                     const float* a4, View dres, int has_dres, View dx,
                                                     float* partials, int64_t M, int c) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [2][8c]
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int64_t wave = (int64_t)blockIdx.x * kLnBwdWaves + wid;
  const int64_t nwaves = (int64_t)gridDim.x * kLnBwdWaves;
  const int D = 8 * c;
  LaneMap<NV> lm;
  make_lane_map<NV>(lm, lane, c);
  const float* alpha[5] = {a0, a1, a2, a3, a4};
  f32x4 av[NV], pa[NV];
  f32x4 pb = {0, 0, 0, 0};   // sum of g over rows, A1 columns only (c <= 32 NV <= 256: they all sit in chunk 0)
  float coefw[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    av[i] = f32x4{1.f, 1.f, 1.f, 1.f};
    if (a0 && lm.seg[i] >= 0) av[i] = *(const f32x4*)(alpha[lm.grp[i]] + lm.aidx[i]);
    pa[i] = f32x4{0, 0, 0, 0};
    coefw[i] = (lm.seg[i] < 4) ? 2.0f / (float)c : 1.0f / (float)(2 * c);  // w_s*2/n_s
  }
  const float inv_n1 = 1.0f / (float)c, inv_n2 = 1.0f / (float)(2 * c);
  for (int64_t m = wave; m < M; m += nwaves) {
    float mean[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) mean[k] = stats[m * 8 + k];
    const float rstd = stats[m * 8 + 6];
    f32x4 xc[NV], gh[NV];
    float sg[6] = {0, 0, 0, 0, 0, 0};
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      xc[i] = f32x4{0, 0, 0, 0};
      gh[i] = f32x4{0, 0, 0, 0};
      if (lm.seg[i] >= 0) {
        const int gidx = PK ? 0 : lm.grp[i];
        const int eoff = PK ? (lane + 64 * i) * 4 : lm.off[i];
        const f32x4 xv = *(const f32x4*)((const float*)x.p[gidx] + m * x.ld[gidx] + eoff);
        const f32x4 gv = load4<TG>((const TG*)g.p[gidx] + m * g.ld[gidx] + eoff);
        const float mu = pick6(mean, lm.seg[i]);
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          xc[i][j] = xv[j] - mu;
          const float xh = xc[i][j] * rstd;
          gh[i][j] = av[i][j] * gv[j];
          pa[i][j] += gv[j] * xh;
          if (i == 0) pb[j] += gv[j];
          dot += gh[i][j] * xh;
          t += gh[i][j];
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) sg[k] += (lm.seg[i] == k) ? t : 0.f;
      }
    }
    dot = wave_sum(dot);
#pragma unroll
    for (int k = 0; k < 6; ++k) sg[k] = wave_sum(sg[k]) * (k < 4 ? inv_n1 : inv_n2);
    const float dS = -rstd * dot * rstd * (1.0f / 16.0f);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if (lm.seg[i] >= 0) {
        const int gidx = PK ? 0 : lm.grp[i];
        const int eoff = PK ? (lane + 64 * i) * 4 : lm.off[i];
        const float mg = pick6(sg, lm.seg[i]);
        const float coef = coefw[i] * dS;
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (gh[i][j] - mg) * rstd + coef * xc[i][j];
        if (has_dres) {
          const f32x4 r = *(const f32x4*)((const float*)dres.p[gidx] + m * dres.ld[gidx] + eoff);
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] += r[j];
        }
        *(f32x4*)((float*)dx.p[gidx] + m * dx.ld[gidx] + eoff) = o;
      }
    }
  }
  // block reduction of the parameter partials: the waves add their registers into one [2][8c] image in turn
  // (fixed order -> bitwise reproducible), then the block writes its slab
  for (int w = 0; w < kLnBwdWaves; ++w) {
    if (wid == w) {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int e = (lane + 64 * i) * 4;
        if (lm.seg[i] >= 0) {
          f32x4* qa = (f32x4*)(smem + e);
          f32x4* qb = (f32x4*)(smem + D + e);
          if (w == 0) {
            *qa = pa[i];
            if (i == 0) *qb = pb;
          } else {
            *qa += pa[i];
            if (i == 0) *qb += pb;
          }
        }
      }
    }
    __syncthreads();
  }
  float* outp = partials + (size_t)blockIdx.x * 2 * D;
  // plane 0: all 8c columns; plane 1: only its first 256 columns (>= c) carry data
  for (int e = threadIdx.x * 4; e < D + 256 && e < 2 * D; e += kLnBwdWaves * 64 * 4)
    *(f32x4*)(outp + e) = *(const f32x4*)(smem + e);
}

// Reduce partials[nblk][2][8c] over blocks.  256 threads = 16 outputs x 16 block-lanes; the 16 lanes of an
// output are combined through LDS in a fixed order (bitwise reproducible).  Output j in [0,7c):
//   j < 4c : dalpha of A1..B2 = plane0[j] ; 4c <= j < 6c : dalpha_E[o] = plane0[4c+o] + plane0[6c+o] (both E rows
//   share alpha_E) ; j >= 6c : dbeta[o] = plane1[o]  (A1 columns of the sum-of-g plane).
__device__ __forceinline__ void ln_bwd_finish_body(const float* partials, int nblk, int c, float* d0, float* d1, float* d2,
                                                   float* d3, float* d4, float* dbeta) {
  __shared__ float red[16][17];
  const int D = 8 * c;
  const int cl = threadIdx.x & 15, bl = threadIdx.x >> 4;
  const int j = blockIdx.x * 16 + cl;
  float s = 0.f;
  if (j < 7 * c) {
    const int c0 = j < 6 * c ? j : D + (j - 6 * c);
    const int c1 = (j >= 4 * c && j < 6 * c) ? j + 2 * c : -1;
    int b = bl;                                   // fixed summation order, four slabs in flight
    for (; b + 48 < nblk; b += 64) {
      float t[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float* row = partials + (size_t)(b + 16 * u) * 2 * D;
        t[u] = row[c0] + (c1 >= 0 ? row[c1] : 0.f);
      }
      s += t[0]; s += t[1]; s += t[2]; s += t[3];
    }
    for (; b < nblk; b += 16) {
      const float* row = partials + (size_t)b * 2 * D;
      s += row[c0] + (c1 >= 0 ? row[c1] : 0.f);
    }
  }
  red[bl][cl] = s;
  __syncthreads();
  if (bl == 0 && j < 7 * c) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][cl];
    if (j < 4 * c) {
      float* d = j < c ? d0 : (j < 2 * c ? d1 : (j < 3 * c ? d2 : d3));
      if (d) d[j % c] = t;
    } else if (j < 6 * c) {
      if (d4) d4[j - 4 * c] = t;
    } else if (dbeta) {
      dbeta[j - 6 * c] = t;
    }
  }
}

__global__ __launch_bounds__(256) void ln_bwd_finish_kernel(const float* partials, int nblk, int c, float* d0, float* d1,
                                                            float* d2, float* d3, float* d4, float* dbeta) {
  ln_bwd_finish_body(partials, nblk, c, d0, d1, d2, d3, d4, dbeta);
}

// up to 48 of these reductions in one launch (blockIdx.y = job; see octic_dense_finish_batch in csrc/dense.hip): the backward
// of an octic block ends in two of them, 32 five-microsecond launches per ViT-H step
struct LnFinishPack {
  octic_ln_finish_job j[48];
};
__global__ __launch_bounds__(256) void ln_bwd_finish_batch_kernel(LnFinishPack pack) {
  const octic_ln_finish_job& job = pack.j[blockIdx.y];
  if ((int)blockIdx.x * 16 >= 7 * job.c) return;
  ln_bwd_finish_body(job.partials, job.nblk, job.c, job.dalpha[0], job.dalpha[1], job.dalpha[2], job.dalpha[3], job.dalpha[4],
                     job.dbeta);
}

// ---------------------------------------------------------------------------------------------------------------
// Lane-group variants (packed rows, c % 32 == 0).  Lanes 0-7 own segment A1, 8-15 A2, 16-23 B1, 24-31 B2, 32-47 the
// first E row, 48-63 the second: every lane's NV = c/32 chunks belong to ONE segment, so a segment mean is a sum
// over 8 (16) neighbouring lanes - three (four) DPP adds, no LDS permutes, no per-chunk segment selects - and the
// row-wide terms are one 64-lane DPP reduction.  A wave-load still covers whole 128-byte lines (8 lanes x 16 B).
// The row is eight octets of c columns (A1 A2 B1 B2 | E row 0: 2 octets | E row 1: 2 octets); octet o belongs to
// lanes 8o..8o+7 and chunk i of a lane covers packed columns col0 + 32 i .. +3: one uniform 128-byte stride, so every
// load / store of a row uses the same base register with an immediate offset.
struct LaneGroup {
  bool isE;
  int seg, col0, acol0;          // acol0 + 32 i = index into this segment's alpha
  static constexpr int step4 = 32;
};
__device__ inline LaneGroup lane_group(int lane, int c) {
  LaneGroup lg;
  const int o = lane >> 3, j = lane & 7;
  lg.isE = o >= 4;
  lg.seg = lg.isE ? 4 + ((o - 4) >> 1) : o;
  lg.col0 = o * c + j * 4;
  lg.acol0 = (lg.isE ? (o & 1) * c : 0) + j * 4;
  return lg;
}
__device__ inline const float* pick_alpha(const LaneGroup& lg, const float* a0, const float* a1, const float* a2,
                                          const float* a3, const float* a4) {
  const float* p = a0;
  p = lg.seg == 1 ? a1 : p;
  p = lg.seg == 2 ? a2 : p;
  p = lg.seg == 3 ? a3 : p;
  return lg.seg >= 4 ? a4 : p;
}

template <typename TOUT, int NV>
__global__ __launch_bounds__(256, 4) void ln_fwd_g8_kernel(const float* __restrict__ x, int64_t ldx,
                                                           TOUT* __restrict__ y, int64_t ldy, const float* a0,
                                                           const float* a1, const float* a2, const float* a3,
                                                           const float* a4, const float* beta,
                                                           float* __restrict__ stats, int64_t M, int c, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  const LaneGroup lg = lane_group(lane, c);
  const float* ap = pick_alpha(lg, a0, a1, a2, a3, a4);
  f32x4 av[NV], bv[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    av[i] = a0 ? *(const f32x4*)(ap + lg.acol0 + i * lg.step4) : f32x4{1.f, 1.f, 1.f, 1.f};
    bv[i] = (beta && lg.seg == 0) ? *(const f32x4*)(beta + lg.acol0 + i * lg.step4) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const float inv_n = lg.isE ? 1.0f / (float)(2 * c) : 1.0f / (float)c;
  const float wq = lg.isE ? 0.5f * inv_n : inv_n;      // weight of this lane's squares in S
  const int stat_idx = (lg.acol0 == 0) ? lg.seg : (lane == 1 ? 6 : -1);
  for (int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); m < M; m += nwaves) {
    const float* xr = x + m * ldx + lg.col0;
    f32x4 xv[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      xv[i] = *(const f32x4*)(xr + i * lg.step4);
      s += (xv[i][0] + xv[i][1]) + (xv[i][2] + xv[i][3]);
    }
    const float s8 = sum8(s);
    const float s16 = sum16_from8(s8);
    const float mean = (lg.isE ? s16 : s8) * inv_n;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      xv[i] -= mean;
      const f32x4 sq = xv[i] * xv[i];
      q += (sq[0] + sq[1]) + (sq[2] + sq[3]);
    }
    const float S = eps + wave_total(q * wq);
    const float rstd = 1.0f / (kSqrt2Over4 * sqrtf(S));
    TOUT* yr = y + m * ldy + lg.col0;
#pragma unroll
    for (int i = 0; i < NV; ++i) store4<TOUT>(yr + i * lg.step4, xv[i] * rstd * av[i] + bv[i]);
    if (stats && stat_idx >= 0) stats[m * 8 + stat_idx] = (stat_idx == 6) ? rstd : mean;
  }
}

// Backward.  Register budget (128 for 4 waves/SIMD) goes to the row in flight and the d alpha partials; alpha itself
// and the d beta partials (A1 lanes only) live in LDS - each lane re-reads / updates only its own addresses.
// LDS: [2][8c] slab image | [8c] alpha by packed column | [waves][c] d beta partials.
// WIDE (NV <= 5, bf16 cotangent: the train step): the 3 NV loads of a row (x, g, dres) are requested before the first use
// and the kernel is built for 2 waves per SIMD (256 registers) instead of 4.  The 128-register build has no room to
// hold a row's loads: its ISA is load / load / s_waitcnt pairs in the first loop and load / vmcnt(0) / store per chunk
// in the second - two loads in flight per wave, 16 waves per CU = 32 KiB in flight against the ~60 KiB that cover the
// memory latency of a CU.  Eight waves holding 15 KiB each do (csrc/dense.hip dense_ln_bwd_wide_kernel: same finding).
#ifndef OCTIC_LNBWD_WIDE
#define OCTIC_LNBWD_WIDE 1
#endif
template <typename TG, int NV, bool WIDE = false>
__global__ __launch_bounds__(kLnBwdWaves * 64, WIDE ? 2 : OCTIC_LNBWD_OCC) void ln_bwd_g8_kernel(
    const TG* __restrict__ g, int64_t ldg, const float* __restrict__ x, int64_t ldx, const float* __restrict__ stats,
    const float* a0, const float* a1, const float* a2, const float* a3, const float* a4,
    const float* __restrict__ dres, int64_t ldr, float* __restrict__ dx, int64_t ldd, float* __restrict__ partials,
    int64_t M, int c, const float* __restrict__ crs = nullptr, int64_t crps = 1, bf16* __restrict__ cg = nullptr,
    int64_t ldc = 0) {
  // cg (WIDE only): also store bf16(crs[row / crps] * dx) - the drop-path-scaled bf16 cotangent the backward of the
  // residual-fused LinearD8 in front of this norm needs (what cast_rowscale_kernel makes of dx in a pass of its own)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int64_t nwaves = (int64_t)gridDim.x * kLnBwdWaves;
  const int D = 8 * c;
  float* alds = smem + 2 * D;
  float* pbl = smem + 3 * D + wid * c;
  const LaneGroup lg = lane_group(lane, c);
  for (int e = threadIdx.x; e < D; e += kLnBwdWaves * 64) {
    float v = 1.f;
    if (a0) {
      const int sg = e < 4 * c ? e / c : 4;
      const float* ap = sg == 0 ? a0 : (sg == 1 ? a1 : (sg == 2 ? a2 : (sg == 3 ? a3 : a4)));
      v = ap[sg < 4 ? e - sg * c : (e - 4 * c) % (2 * c)];
    }
    alds[e] = v;
  }
  f32x4 pa[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    pa[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (lg.seg == 0) *(f32x4*)(pbl + lg.acol0 + i * 32) = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  __syncthreads();
  const float inv_n = lg.isE ? 1.0f / (float)(2 * c) : 1.0f / (float)c;
  const float coefw = lg.isE ? inv_n : 2.0f * inv_n;    // w_s*2/n_s
  for (int64_t m = (int64_t)blockIdx.x * kLnBwdWaves + wid; m < M; m += nwaves) {
    const float mu = stats[m * 8 + lg.seg], rstd = stats[m * 8 + 6];
    f32x4 xc[NV], gh[NV];
    f32x4 dr[WIDE ? NV : 1];
    typedef typename std::conditional<std::is_same<TG, float>::value, f32x4, bf16x4>::type graw;
    graw gr[WIDE ? NV : 1];
    if constexpr (WIDE) {
#pragma unroll
      for (int i = 0; i < NV; ++i) xc[i] = *(const f32x4*)(x + m * ldx + lg.col0 + i * lg.step4);
#pragma unroll
      for (int i = 0; i < NV; ++i) gr[i] = *(const graw*)(g + m * ldg + lg.col0 + i * lg.step4);
#pragma unroll
      for (int i = 0; i < NV; ++i)
        dr[i] = dres ? *(const f32x4*)(dres + m * ldr + lg.col0 + i * lg.step4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    float dot = 0.f, t = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      f32x4 gv;
      if constexpr (WIDE) {
        xc[i] = xc[i] - mu;
        if constexpr (std::is_same<TG, float>::value) gv = gr[i];
        else gv = f32x4{(float)gr[i][0], (float)gr[i][1], (float)gr[i][2], (float)gr[i][3]};
      } else {
        xc[i] = *(const f32x4*)(x + m * ldx + lg.col0 + i * lg.step4) - mu;
        gv = load4<TG>(g + m * ldg + lg.col0 + i * lg.step4);
      }
      const f32x4 xh = xc[i] * rstd;
      gh[i] = *(const f32x4*)(alds + lg.col0 + i * lg.step4) * gv;
      pa[i] += gv * xh;
      if (lg.seg == 0) *(f32x4*)(pbl + lg.acol0 + i * 32) += gv;
      const f32x4 d = gh[i] * xh;
      dot += (d[0] + d[1]) + (d[2] + d[3]);
      t += (gh[i][0] + gh[i][1]) + (gh[i][2] + gh[i][3]);
    }
    const float t8 = sum8(t);
    const float mg = (lg.isE ? sum16_from8(t8) : t8) * inv_n;
    const float dS = -rstd * wave_total(dot) * rstd * (1.0f / 16.0f);
    const float coef = coefw * dS;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      f32x4 o = (gh[i] - mg) * rstd + xc[i] * coef;
      if constexpr (WIDE) {
        o += dr[i];
        if (cg) {
          const float cs_ = crs ? crs[m / crps] : 1.0f;
          const f32x4 sc = o * cs_;
          *(bf16x4*)(cg + m * ldc + lg.col0 + i * lg.step4) = bf16x4{(bf16)sc[0], (bf16)sc[1], (bf16)sc[2], (bf16)sc[3]};
        }
      } else {
        if (dres) o += *(const f32x4*)(dres + m * ldr + lg.col0 + i * lg.step4);
      }
      *(f32x4*)(dx + m * ldd + lg.col0 + i * lg.step4) = o;
    }
  }
  // slab: plane 0 = sum g*xhat for all 8c columns, plane 1 = sum g, A1 columns only (all the finish kernel reads)
  for (int w = 0; w < kLnBwdWaves; ++w) {
    if (wid == w) {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        f32x4* qa = (f32x4*)(smem + lg.col0 + i * lg.step4);
        if (w == 0) *qa = pa[i];
        else *qa += pa[i];
        if (lg.seg == 0) {
          f32x4* qb = (f32x4*)(smem + D + lg.acol0 + i * 32);
          const f32x4 mine = *(const f32x4*)(pbl + lg.acol0 + i * 32);
          if (w == 0) *qb = mine;
          else *qb += mine;
        }
      }
    }
    __syncthreads();
  }
  float* outp = partials + (size_t)blockIdx.x * 2 * D;
  for (int e = threadIdx.x * 4; e < D + c; e += kLnBwdWaves * 64 * 4) *(f32x4*)(outp + e) = *(const f32x4*)(smem + e);
}

inline int ln_blocks(int64_t M) {   // backward: one partial slab per block, 8 waves each
  int64_t b = (M + kLnBwdWaves - 1) / kLnBwdWaves;
#ifndef OCTIC_LNBWD_CAP
#define OCTIC_LNBWD_CAP 256     // one workgroup per CU: the wide kernel (1 per CU by registers) then runs ONE round (56 vs 63 us)
#endif
  const int64_t cap = OCTIC_LNBWD_CAP;
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}
inline int ln_fwd_blocks(int64_t M) {   // forward: 4 waves per block, 4 rows per wave (per-lane constants amortised)
  int64_t b = (M + 15) / 16;
  return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

inline int pick_nv(int c) {
  const int need = (2 * c + 63) / 64;  // float4 chunks per lane
  return need <= 2 ? 2 : (need <= 4 ? 4 : (need <= 5 ? 5 : (need <= 8 ? 8 : 0)));
}

// one packed tensor behind the five pointers?
inline bool view_is_packed(const View& v, int c, int es) {
  for (int g = 1; g < 5; ++g)
    if (v.ld[g] != v.ld[0] || v.p[g] != v.p[0] + (int64_t)g * c * es) return false;
  return true;
}

template <typename TOUT, bool PK>
void launch_ln_fwd(int nv, int grid, hipStream_t s, View vx, View vy, const float* const a[5], const float* beta,
                   float* stats, int64_t M, int c, float eps) {
  switch (nv) {
    case 2: ln_fwd_kernel<TOUT, 2, PK><<<grid, 256, 0, s>>>(vx, vy, a[0], a[1], a[2], a[3], a[4], beta, stats, M, c, eps); break;
    case 4: ln_fwd_kernel<TOUT, 4, PK><<<grid, 256, 0, s>>>(vx, vy, a[0], a[1], a[2], a[3], a[4], beta, stats, M, c, eps); break;
    case 5: ln_fwd_kernel<TOUT, 5, PK><<<grid, 256, 0, s>>>(vx, vy, a[0], a[1], a[2], a[3], a[4], beta, stats, M, c, eps); break;
    default: ln_fwd_kernel<TOUT, 8, PK><<<grid, 256, 0, s>>>(vx, vy, a[0], a[1], a[2], a[3], a[4], beta, stats, M, c, eps); break;
  }
}

template <typename TG, bool PK>
void launch_ln_bwd(int nv, int grid, size_t smem, hipStream_t s, View vg, View vx, const float* stats,
                   const float* const a[5], View vr, int has_dres, View vd, float* partials, int64_t M, int c) {
  switch (nv) {
    case 2: ln_bwd_kernel<TG, 2, PK><<<grid, kLnBwdWaves * 64, smem, s>>>(vg, vx, stats, a[0], a[1], a[2], a[3], a[4], vr, has_dres, vd, partials, M, c); break;
    case 4: ln_bwd_kernel<TG, 4, PK><<<grid, kLnBwdWaves * 64, smem, s>>>(vg, vx, stats, a[0], a[1], a[2], a[3], a[4], vr, has_dres, vd, partials, M, c); break;
    case 5: ln_bwd_kernel<TG, 5, PK><<<grid, kLnBwdWaves * 64, smem, s>>>(vg, vx, stats, a[0], a[1], a[2], a[3], a[4], vr, has_dres, vd, partials, M, c); break;
    default: ln_bwd_kernel<TG, 8, PK><<<grid, kLnBwdWaves * 64, smem, s>>>(vg, vx, stats, a[0], a[1], a[2], a[3], a[4], vr, has_dres, vd, partials, M, c); break;
  }
}

}  // namespace octic

using namespace octic;

extern "C" {

int octic_layernorm_d8_fwd(const octic_view* x, const octic_view* y, const float* const alpha[5], const float* beta,
                           float* stats, int64_t M, int c, float eps, int out_dtype, void* stream) {
  int e;
  if ((e = check_c(c)) || (e = check_view(x, c, OCTIC_F32)) || (e = check_view(y, c, out_dtype))) return e;
  const int nv = pick_nv(c);
  if (M <= 0 || nv == 0) return OCTIC_ESHAPE;
  const float* a[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  if (alpha && alpha[0]) {
    for (int i = 0; i < 5; ++i) {
      if (!alpha[i]) return OCTIC_ENULL;
      a[i] = alpha[i];
    }
  }
  View vx = make_view<void>(x), vy = make_view<void>(y);
  const int grid = ln_fwd_blocks(M);
  hipStream_t s = (hipStream_t)stream;
  if (out_dtype != OCTIC_F32 && out_dtype != OCTIC_BF16) return OCTIC_EDTYPE;
  const bool pk = view_is_packed(vx, c, 4) && view_is_packed(vy, c, elem_size(out_dtype));
  if (pk && (c % 32) == 0 && c <= 256) {
    const float* xp = (const float*)vx.p[0];
#define LN_FWD_G8(T, N) ln_fwd_g8_kernel<T, N><<<grid, 256, 0, s>>>(xp, vx.ld[0], (T*)vy.p[0], vy.ld[0], a[0], a[1], a[2], a[3], a[4], beta, stats, M, c, eps)
#define LN_FWD_G8_NV(T) switch (c / 32) { case 1: LN_FWD_G8(T, 1); break; case 2: LN_FWD_G8(T, 2); break; case 3: LN_FWD_G8(T, 3); break; \
    case 4: LN_FWD_G8(T, 4); break; case 5: LN_FWD_G8(T, 5); break; case 6: LN_FWD_G8(T, 6); break; case 7: LN_FWD_G8(T, 7); break; default: LN_FWD_G8(T, 8); break; }
    if (out_dtype == OCTIC_F32) { LN_FWD_G8_NV(float) } else { LN_FWD_G8_NV(bf16) }
    return launch_status();
  }
  if (out_dtype == OCTIC_F32) {
    if (pk) launch_ln_fwd<float, true>(nv, grid, s, vx, vy, a, beta, stats, M, c, eps);
    else launch_ln_fwd<float, false>(nv, grid, s, vx, vy, a, beta, stats, M, c, eps);
  } else {
    if (pk) launch_ln_fwd<bf16, true>(nv, grid, s, vx, vy, a, beta, stats, M, c, eps);
    else launch_ln_fwd<bf16, false>(nv, grid, s, vx, vy, a, beta, stats, M, c, eps);
  }
  return launch_status();
}

int octic_layernorm_d8_bwd_blocks(int64_t M) { return ln_blocks(M); }

static int ln_bwd_impl(const octic_view* g, const octic_view* x, const float* stats, const float* const alpha[5],
                       const octic_view* dres, const octic_view* dx, float* partials, int64_t M, int c, int g_dtype,
                       void* stream, const float* crs, int64_t crps, void* cg) {
  int e;
  if ((e = check_c(c)) || (e = check_view(g, c, g_dtype)) || (e = check_view(x, c, OCTIC_F32)) ||
      (e = check_view(dx, c, OCTIC_F32)))
    return e;
  if (dres && (e = check_view(dres, c, OCTIC_F32))) return e;
  if (!stats || !partials) return OCTIC_ENULL;
  const int nv = pick_nv(c);
  if (M <= 0 || nv == 0) return OCTIC_ESHAPE;
  const float* a[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  if (alpha && alpha[0]) {
    for (int i = 0; i < 5; ++i) {
      if (!alpha[i]) return OCTIC_ENULL;
      a[i] = alpha[i];
    }
  }
  View vg = make_view<void>(g), vx = make_view<void>(x), vd = make_view<void>(dx);
  View vr = dres ? make_view<void>(dres) : vd;
  const int grid = ln_blocks(M);
  const size_t smem = (size_t)2 * 8 * c * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
  if (g_dtype != OCTIC_F32 && g_dtype != OCTIC_BF16) return OCTIC_EDTYPE;
  const bool pk = view_is_packed(vg, c, elem_size(g_dtype)) && view_is_packed(vx, c, 4) && view_is_packed(vd, c, 4) &&
                  view_is_packed(vr, c, 4);
  const int hd = dres ? 1 : 0;
  if (pk && (c % 32) == 0 && c <= 256) {
    const float* rp = dres ? (const float*)vr.p[0] : nullptr;
    const size_t smem_g8 = (size_t)(3 * 8 * c + kLnBwdWaves * c) * sizeof(float);
#define LN_BWD_G8(T, N) ln_bwd_g8_kernel<T, N><<<grid, kLnBwdWaves * 64, smem_g8, s>>>((const T*)vg.p[0], vg.ld[0], (const float*)vx.p[0], vx.ld[0], stats, a[0], a[1], a[2], a[3], a[4], rp, vr.ld[0], (float*)vd.p[0], vd.ld[0], partials, M, c)
#define LN_BWD_G8_NV(T) switch (c / 32) { case 1: LN_BWD_G8(T, 1); break; case 2: LN_BWD_G8(T, 2); break; case 3: LN_BWD_G8(T, 3); break; \
    case 4: LN_BWD_G8(T, 4); break; case 5: LN_BWD_G8(T, 5); break; case 6: LN_BWD_G8(T, 6); break; case 7: LN_BWD_G8(T, 7); break; default: LN_BWD_G8(T, 8); break; }
    if (OCTIC_LNBWD_WIDE && g_dtype == OCTIC_BF16 && c / 32 <= 5) {
#define LN_BWD_G8W(N) ln_bwd_g8_kernel<bf16, N, true><<<grid, kLnBwdWaves * 64, smem_g8, s>>>((const bf16*)vg.p[0], vg.ld[0], (const float*)vx.p[0], vx.ld[0], stats, a[0], a[1], a[2], a[3], a[4], rp, vr.ld[0], (float*)vd.p[0], vd.ld[0], partials, M, c, crs, crs ? crps : 1, (bf16*)cg, (int64_t)8 * c)
      switch (c / 32) { case 1: LN_BWD_G8W(1); break; case 2: LN_BWD_G8W(2); break; case 3: LN_BWD_G8W(3); break;
                        case 4: LN_BWD_G8W(4); break; default: LN_BWD_G8W(5); break; }
#undef LN_BWD_G8W
      return launch_status();
    }
    if (cg) return OCTIC_ESHAPE;          // the scaled bf16 copy exists in the wide kernel only
    if (g_dtype == OCTIC_F32) { LN_BWD_G8_NV(float) } else { LN_BWD_G8_NV(bf16) }
    return launch_status();
  }
  if (cg) return OCTIC_ESHAPE;
  if (g_dtype == OCTIC_F32) {
    if (pk) launch_ln_bwd<float, true>(nv, grid, smem, s, vg, vx, stats, a, vr, hd, vd, partials, M, c);
    else launch_ln_bwd<float, false>(nv, grid, smem, s, vg, vx, stats, a, vr, hd, vd, partials, M, c);
  } else {
    if (pk) launch_ln_bwd<bf16, true>(nv, grid, smem, s, vg, vx, stats, a, vr, hd, vd, partials, M, c);
    else launch_ln_bwd<bf16, false>(nv, grid, smem, s, vg, vx, stats, a, vr, hd, vd, partials, M, c);
  }
  return launch_status();
}

int octic_layernorm_d8_bwd(const octic_view* g, const octic_view* x, const float* stats, const float* const alpha[5],
                           const octic_view* dres, const octic_view* dx, float* partials, int64_t M, int c, int g_dtype,
                           void* stream) {
  return ln_bwd_impl(g, x, stats, alpha, dres, dx, partials, M, c, g_dtype, stream, nullptr, 1, nullptr);
}

int octic_layernorm_d8_bwd_cast(const octic_view* g, const octic_view* x, const float* stats, const float* const alpha[5],
                                const octic_view* dres, const octic_view* dx, float* partials, int64_t M, int c,
                                const float* rs, int64_t rows_per_sample, void* gcast, void* stream) {
  if (!gcast) return OCTIC_ENULL;
  if (rs && rows_per_sample <= 0) return OCTIC_ESHAPE;
  if (((uintptr_t)gcast) & 15) return OCTIC_EALIGN;
  return ln_bwd_impl(g, x, stats, alpha, dres, dx, partials, M, c, OCTIC_BF16, stream, rs, rows_per_sample, gcast);
}

int octic_layernorm_d8_bwd_finish(const float* partials, int nblk, int c, float* const dalpha[5], float* dbeta,
                                  void* stream) {
  if (!partials) return OCTIC_ENULL;
  if (nblk <= 0 || check_c(c)) return OCTIC_ESHAPE;
  float* d[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  if (dalpha)
    for (int i = 0; i < 5; ++i) d[i] = dalpha[i];
  ln_bwd_finish_kernel<<<(7 * c + 15) / 16, 256, 0, (hipStream_t)stream>>>(partials, nblk, c, d[0], d[1], d[2], d[3], d[4], dbeta);
  return launch_status();
}

int octic_layernorm_d8_bwd_finish_batch(const octic_ln_finish_job* jobs, int njobs, void* stream) {
  if (!jobs) return OCTIC_ENULL;
  if (njobs < 0) return OCTIC_ESHAPE;
  for (int i = 0; i < njobs; ++i) {
    if (!jobs[i].partials) return OCTIC_ENULL;
    if (jobs[i].nblk <= 0 || check_c(jobs[i].c)) return OCTIC_ESHAPE;
  }
  for (int i0 = 0; i0 < njobs; i0 += 48) {
    const int n = njobs - i0 < 48 ? njobs - i0 : 48;
    LnFinishPack pack = {};
    int cmax = 0;
    for (int i = 0; i < n; ++i) {
      pack.j[i] = jobs[i0 + i];
      cmax = jobs[i0 + i].c > cmax ? jobs[i0 + i].c : cmax;
    }
    ln_bwd_finish_batch_kernel<<<dim3((7 * cmax + 15) / 16, n), 256, 0, (hipStream_t)stream>>>(pack);
  }
  return launch_status();
}

}  // extern "C"
