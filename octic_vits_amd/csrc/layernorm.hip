// LayerNormD8 (+AffineD8) forward / backward for gfx950.  One wave64 per token row; the row
// (8c f32 = 5 KiB at ViT-H) lives in registers between the statistics pass and the normalise pass,
// so HBM traffic is exactly one read of x and one write of y (fwd), read g,x(,dres) + write dx (bwd).
// Everything that depends only on the lane (which irrep segment each of its 16-byte chunks belongs to,
// the chunk's address inside the token row, its alpha/beta values) is computed ONCE before the row loop
// and kept in registers; the chunk count per lane is a template parameter (NV = 5 at ViT-H) so nothing
// is predicated at run time.  Bound: HBM.
// Reference: octic_vits/d8_layers.py:161-186 (forward), backward derived in SURVEY.md §10.3.
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

template <typename TOUT, int NV>
__global__ __launch_bounds__(256) void ln_fwd_kernel(View x, View y, const float* a0, const float* a1, const float* a2,
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
    xb[i] = (const float*)x.p[g] + lm.off[i];
    yb[i] = (TOUT*)y.p[g] + lm.off[i];
    xld[i] = x.ld[g];
    yld[i] = y.ld[g];
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
template <typename TG, int NV>
__global__ __launch_bounds__(256) void ln_bwd_kernel(View g, View x, const float* stats, const float* a0,
                                                     const float* a1, const float* a2, const float* a3,
                                                     const float* a4, View dres, int has_dres, View dx,
                                                     float* partials, int64_t M, int c) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [4 waves][2][8c]
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int64_t wave = (int64_t)blockIdx.x * 4 + wid;
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  const int D = 8 * c;
  LaneMap<NV> lm;
  make_lane_map<NV>(lm, lane, c);
  const float* alpha[5] = {a0, a1, a2, a3, a4};
  f32x4 av[NV], pa[NV], pb[NV];
  float coefw[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    av[i] = f32x4{1.f, 1.f, 1.f, 1.f};
    if (a0 && lm.seg[i] >= 0) av[i] = *(const f32x4*)(alpha[lm.grp[i]] + lm.aidx[i]);
    pa[i] = f32x4{0, 0, 0, 0};
    pb[i] = f32x4{0, 0, 0, 0};
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
        const int gidx = lm.grp[i];
        const f32x4 xv = *(const f32x4*)((const float*)x.p[gidx] + m * x.ld[gidx] + lm.off[i]);
        const f32x4 gv = load4<TG>((const TG*)g.p[gidx] + m * g.ld[gidx] + lm.off[i]);
        const float mu = pick6(mean, lm.seg[i]);
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          xc[i][j] = xv[j] - mu;
          const float xh = xc[i][j] * rstd;
          gh[i][j] = av[i][j] * gv[j];
          pa[i][j] += gv[j] * xh;
          pb[i][j] += gv[j];
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
        const int gidx = lm.grp[i];
        const float mg = pick6(sg, lm.seg[i]);
        const float coef = coefw[i] * dS;
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (gh[i][j] - mg) * rstd + coef * xc[i][j];
        if (has_dres) {
          const f32x4 r = *(const f32x4*)((const float*)dres.p[gidx] + m * dres.ld[gidx] + lm.off[i]);
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] += r[j];
        }
        *(f32x4*)((float*)dx.p[gidx] + m * dx.ld[gidx] + lm.off[i]) = o;
      }
    }
  }
  // block reduction of the parameter partials
  float* mine = smem + (size_t)wid * 2 * D;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int e = (lane + 64 * i) * 4;
    if (lm.seg[i] >= 0) {
      *(f32x4*)(mine + e) = pa[i];
      *(f32x4*)(mine + D + e) = pb[i];
    }
  }
  __syncthreads();
  float* outp = partials + (size_t)blockIdx.x * 2 * D;
  for (int e = threadIdx.x; e < 2 * D; e += 256)
    outp[e] = smem[e] + smem[2 * D + e] + smem[4 * D + e] + smem[6 * D + e];
}

// Reduce partials[nblk][2][8c] over blocks.  256 threads = 16 outputs x 16 block-lanes; the 16 lanes of an
// output are combined through LDS in a fixed order (bitwise reproducible).  Output j in [0,7c):
//   j < 4c : dalpha of A1..B2 = plane0[j] ; 4c <= j < 6c : dalpha_E[o] = plane0[4c+o] + plane0[6c+o] (both E rows
//   share alpha_E) ; j >= 6c : dbeta[o] = plane1[o]  (A1 columns of the sum-of-g plane).
__global__ __launch_bounds__(256) void ln_bwd_finish_kernel(const float* partials, int nblk, int c, float* d0, float* d1,
                                                            float* d2, float* d3, float* d4, float* dbeta) {
  __shared__ float red[16][17];
  const int D = 8 * c;
  const int cl = threadIdx.x & 15, bl = threadIdx.x >> 4;
  const int j = blockIdx.x * 16 + cl;
  float s = 0.f;
  if (j < 7 * c) {
    const int c0 = j < 6 * c ? j : D + (j - 6 * c);
    const int c1 = (j >= 4 * c && j < 6 * c) ? j + 2 * c : -1;
    for (int b = bl; b < nblk; b += 16) {
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

inline int ln_blocks(int64_t M) {
  int64_t b = (M + 3) / 4;
  const int64_t cap = 256 * 2;  // 2 blocks (8 waves) per CU x register-resident rows; grid-stride beyond
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

inline int pick_nv(int c) {
  const int need = (2 * c + 63) / 64;  // float4 chunks per lane
  return need <= 2 ? 2 : (need <= 4 ? 4 : (need <= 5 ? 5 : (need <= 8 ? 8 : 0)));
}

template <typename TOUT>
void launch_ln_fwd(int nv, int grid, hipStream_t s, View vx, View vy, const float* const a[5], const float* beta,
                   float* stats, int64_t M, int c, float eps) {
  switch (nv) {
    case 2: ln_fwd_kernel<TOUT, 2><<<grid, 256, 0, s>>>(vx, vy, a[0], a[1], a[2], a[3], a[4], beta, stats, M, c, eps); break;
    case 4: ln_fwd_kernel<TOUT, 4><<<grid, 256, 0, s>>>(vx, vy, a[0], a[1], a[2], a[3], a[4], beta, stats, M, c, eps); break;
    case 5: ln_fwd_kernel<TOUT, 5><<<grid, 256, 0, s>>>(vx, vy, a[0], a[1], a[2], a[3], a[4], beta, stats, M, c, eps); break;
    default: ln_fwd_kernel<TOUT, 8><<<grid, 256, 0, s>>>(vx, vy, a[0], a[1], a[2], a[3], a[4], beta, stats, M, c, eps); break;
  }
}

template <typename TG>
void launch_ln_bwd(int nv, int grid, size_t smem, hipStream_t s, View vg, View vx, const float* stats,
                   const float* const a[5], View vr, int has_dres, View vd, float* partials, int64_t M, int c) {
  switch (nv) {
    case 2: ln_bwd_kernel<TG, 2><<<grid, 256, smem, s>>>(vg, vx, stats, a[0], a[1], a[2], a[3], a[4], vr, has_dres, vd, partials, M, c); break;
    case 4: ln_bwd_kernel<TG, 4><<<grid, 256, smem, s>>>(vg, vx, stats, a[0], a[1], a[2], a[3], a[4], vr, has_dres, vd, partials, M, c); break;
    case 5: ln_bwd_kernel<TG, 5><<<grid, 256, smem, s>>>(vg, vx, stats, a[0], a[1], a[2], a[3], a[4], vr, has_dres, vd, partials, M, c); break;
    default: ln_bwd_kernel<TG, 8><<<grid, 256, smem, s>>>(vg, vx, stats, a[0], a[1], a[2], a[3], a[4], vr, has_dres, vd, partials, M, c); break;
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
  const int grid = ln_blocks(M);
  hipStream_t s = (hipStream_t)stream;
  if (out_dtype == OCTIC_F32) launch_ln_fwd<float>(nv, grid, s, vx, vy, a, beta, stats, M, c, eps);
  else if (out_dtype == OCTIC_BF16) launch_ln_fwd<bf16>(nv, grid, s, vx, vy, a, beta, stats, M, c, eps);
  else return OCTIC_EDTYPE;
  return launch_status();
}

int octic_layernorm_d8_bwd_blocks(int64_t M) { return ln_blocks(M); }

int octic_layernorm_d8_bwd(const octic_view* g, const octic_view* x, const float* stats, const float* const alpha[5],
                           const octic_view* dres, const octic_view* dx, float* partials, int64_t M, int c, int g_dtype,
                           void* stream) {
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
  const size_t smem = (size_t)4 * 2 * 8 * c * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
  if (g_dtype == OCTIC_F32) launch_ln_bwd<float>(nv, grid, smem, s, vg, vx, stats, a, vr, dres ? 1 : 0, vd, partials, M, c);
  else if (g_dtype == OCTIC_BF16) launch_ln_bwd<bf16>(nv, grid, smem, s, vg, vx, stats, a, vr, dres ? 1 : 0, vd, partials, M, c);
  else return OCTIC_EDTYPE;
  return launch_status();
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

}  // extern "C"
