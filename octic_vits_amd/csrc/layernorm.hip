// LayerNormD8 (+AffineD8) forward / backward for gfx950.  One wave64 per token row; the row
// (8c f32 = 5 KiB at ViT-H) lives in registers between the statistics pass and the normalise pass,
// so HBM traffic is exactly one read of x and one write of y (fwd), read g,x + write dx (bwd).
// Reference: octic_vits/d8_layers.py:161-186 (forward), backward derived in SURVEY.md §10.3.
#include "octic_common.hpp"

namespace octic {

constexpr int kMaxV = 8;  // float4 chunks cached per lane: rows up to 8*64*4 = 2048 channels stay in registers

// segment (0..5: A1,A2,B1,B2,E_row0,E_row1), element pointer and alpha index of logical column e
struct Col {
  int seg;
  int aidx;  // index into alpha_seg (E rows share alpha_E)
};
__device__ inline Col col_of(int e, int c) {
  Col r;
  if (e < 4 * c) {
    r.seg = e / c;
    r.aidx = e - r.seg * c;
  } else {
    const int o = e - 4 * c;
    const int row = o / (2 * c);
    r.seg = 4 + row;
    r.aidx = o - row * 2 * c;
  }
  return r;
}

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

// Per-row statistics from the register-resident row.  nseg[] = channels per segment.
__device__ inline void row_stats(const f32x4 xv[kMaxV], int nv, int lane, int c, float eps, float mean[6], float& rstd) {
  float s[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < kMaxV; ++i) {
    if (i < nv) {
      const int e = (lane + 64 * i) * 4;
      if (e < 8 * c) {
        const int seg = col_of(e, c).seg;
        const float t = xv[i][0] + xv[i][1] + xv[i][2] + xv[i][3];
#pragma unroll
        for (int k = 0; k < 6; ++k) s[k] += (seg == k) ? t : 0.f;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    s[k] = wave_sum(s[k]);
    mean[k] = s[k] / (float)(k < 4 ? c : 2 * c);
  }
  float q[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < kMaxV; ++i) {
    if (i < nv) {
      const int e = (lane + 64 * i) * 4;
      if (e < 8 * c) {
        const int seg = col_of(e, c).seg;
        float mu = 0.f;
#pragma unroll
        for (int k = 0; k < 6; ++k) mu = (seg == k) ? mean[k] : mu;
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float d = xv[i][j] - mu;
          t += d * d;
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) q[k] += (seg == k) ? t : 0.f;
      }
    }
  }
  float S = eps;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    q[k] = wave_sum(q[k]);
    S += (k < 4) ? q[k] / (float)c : 0.5f * q[k] / (float)(2 * c);
  }
  rstd = 1.0f / (kSqrt2Over4 * sqrtf(S));
}

template <typename TOUT>
__global__ __launch_bounds__(256) void ln_fwd_kernel(View x, View y, const float* a0, const float* a1, const float* a2,
                                                     const float* a3, const float* a4, const float* beta,
                                                     float* stats, int64_t M, int c, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  const int D = 8 * c;
  const int nv = (D / 4 + 63) / 64;
  const float* alpha[6] = {a0, a1, a2, a3, a4, a4};
  for (int64_t m = wave; m < M; m += nwaves) {
    f32x4 xv[kMaxV];
#pragma unroll
    for (int i = 0; i < kMaxV; ++i) {
      const int e = (lane + 64 * i) * 4;
      if (i < nv && e < D) xv[i] = *(const f32x4*)view_ptr<float>(x, m, e, c);
    }
    float mean[6], rstd;
    row_stats(xv, nv, lane, c, eps, mean, rstd);
#pragma unroll
    for (int i = 0; i < kMaxV; ++i) {
      const int e = (lane + 64 * i) * 4;
      if (i < nv && e < D) {
        const Col cl = col_of(e, c);
        float mu = 0.f;
#pragma unroll
        for (int k = 0; k < 6; ++k) mu = (cl.seg == k) ? mean[k] : mu;
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (xv[i][j] - mu) * rstd;
        if (a0) {
          const float* ap = a4;
#pragma unroll
          for (int k = 0; k < 4; ++k) ap = (cl.seg == k) ? alpha[k] : ap;
          const f32x4 av = *(const f32x4*)(ap + cl.aidx);
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] *= av[j];
        }
        if (beta && cl.seg == 0) {
          const f32x4 bv = *(const f32x4*)(beta + cl.aidx);
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] += bv[j];
        }
        store4<TOUT>(view_ptr<TOUT>(y, m, e, c), o);
      }
    }
    if (stats && lane < 8) {
      float v = lane < 6 ? 0.f : (lane == 6 ? rstd : 0.f);
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
template <typename TG>
__global__ __launch_bounds__(256) void ln_bwd_kernel(View g, View x, const float* stats, const float* a0,
                                                     const float* a1, const float* a2, const float* a3,
                                                     const float* a4, View dres, int has_dres, View dx,
                                                     float* partials, int64_t M, int c) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [4 waves][2][8c]
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int64_t wave = (int64_t)blockIdx.x * 4 + wid;
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  const int D = 8 * c;
  const int nv = (D / 4 + 63) / 64;
  const float* alpha[6] = {a0, a1, a2, a3, a4, a4};
  f32x4 pa[kMaxV], pb[kMaxV];
#pragma unroll
  for (int i = 0; i < kMaxV; ++i) {
    pa[i] = f32x4{0, 0, 0, 0};
    pb[i] = f32x4{0, 0, 0, 0};
  }
  for (int64_t m = wave; m < M; m += nwaves) {
    float mean[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) mean[k] = stats[m * 8 + k];
    const float rstd = stats[m * 8 + 6];
    f32x4 xc[kMaxV], gh[kMaxV];  // centred x, alpha*g
    float sg[6] = {0, 0, 0, 0, 0, 0};
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < kMaxV; ++i) {
      const int e = (lane + 64 * i) * 4;
      if (i < nv && e < D) {
        const Col cl = col_of(e, c);
        float mu = 0.f;
#pragma unroll
        for (int k = 0; k < 6; ++k) mu = (cl.seg == k) ? mean[k] : mu;
        const f32x4 xv = *(const f32x4*)view_ptr<float>(x, m, e, c);
        const f32x4 gv = load4<TG>(view_ptr<TG>(g, m, e, c));
        f32x4 av = {1.f, 1.f, 1.f, 1.f};
        if (a0) {
          const float* ap = a4;
#pragma unroll
          for (int k = 0; k < 4; ++k) ap = (cl.seg == k) ? alpha[k] : ap;
          av = *(const f32x4*)(ap + cl.aidx);
        }
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          xc[i][j] = xv[j] - mu;
          const float xh = xc[i][j] * rstd;
          gh[i][j] = av[j] * gv[j];
          pa[i][j] += gv[j] * xh;
          pb[i][j] += gv[j];
          dot += gh[i][j] * xh;
          t += gh[i][j];
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) sg[k] += (cl.seg == k) ? t : 0.f;
      }
    }
    dot = wave_sum(dot);
#pragma unroll
    for (int k = 0; k < 6; ++k) sg[k] = wave_sum(sg[k]) / (float)(k < 4 ? c : 2 * c);
    const float dS = -rstd * dot * rstd * (1.0f / 16.0f);
#pragma unroll
    for (int i = 0; i < kMaxV; ++i) {
      const int e = (lane + 64 * i) * 4;
      if (i < nv && e < D) {
        const Col cl = col_of(e, c);
        float mg = 0.f;
#pragma unroll
        for (int k = 0; k < 6; ++k) mg = (cl.seg == k) ? sg[k] : mg;
        const float coef = (cl.seg < 4) ? 2.0f * dS / (float)c : dS / (float)(2 * c);  // w_s*2/n_s*dS
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (gh[i][j] - mg) * rstd + coef * xc[i][j];
        if (has_dres) {
          const f32x4 r = *(const f32x4*)view_ptr<float>(dres, m, e, c);
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] += r[j];
        }
        *(f32x4*)view_ptr<float>(dx, m, e, c) = o;
      }
    }
  }
  // block reduction of the parameter partials
  float* mine = smem + (size_t)wid * 2 * D;
#pragma unroll
  for (int i = 0; i < kMaxV; ++i) {
    const int e = (lane + 64 * i) * 4;
    if (i < nv && e < D) {
      *(f32x4*)(mine + e) = pa[i];
      *(f32x4*)(mine + D + e) = pb[i];
    }
  }
  __syncthreads();
  float* outp = partials + (size_t)blockIdx.x * 2 * D;
  for (int e = threadIdx.x; e < 2 * D; e += 256)
    outp[e] = smem[e] + smem[2 * D + e] + smem[4 * D + e] + smem[6 * D + e];
}

// dalpha_seg[j] = sum_blk partial[blk][0][col] (E: both rows), dbeta[j] = sum_blk partial[blk][1][col<c]
__global__ __launch_bounds__(256) void ln_bwd_finish_kernel(const float* partials, int nblk, int c, float* d0, float* d1,
                                                            float* d2, float* d3, float* d4, float* dbeta) {
  const int D = 8 * c;
  const int j = blockIdx.x * 256 + threadIdx.x;  // 0 .. 6c (alpha entries) then c beta entries
  if (j >= 7 * c) return;
  float s = 0.f;
  if (j < 4 * c) {
    for (int b = 0; b < nblk; ++b) s += partials[(size_t)b * 2 * D + j];
    float* d = j < c ? d0 : (j < 2 * c ? d1 : (j < 3 * c ? d2 : d3));
    if (d) d[j % c] = s;
  } else if (j < 6 * c) {
    const int o = j - 4 * c;
    for (int b = 0; b < nblk; ++b) s += partials[(size_t)b * 2 * D + 4 * c + o] + partials[(size_t)b * 2 * D + 6 * c + o];
    if (d4) d4[o] = s;
  } else {
    const int o = j - 6 * c;
    for (int b = 0; b < nblk; ++b) s += partials[(size_t)b * 2 * D + D + o];
    if (dbeta) dbeta[o] = s;
  }
}

inline int ln_blocks(int64_t M) {
  int64_t b = (M + 3) / 4;
  const int64_t cap = 256 * 4;  // 4 blocks (16 waves) per CU; each wave then owns ~M/4096 rows
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace octic

using namespace octic;

extern "C" {

int octic_layernorm_d8_fwd(const octic_view* x, const octic_view* y, const float* const alpha[5], const float* beta,
                           float* stats, int64_t M, int c, float eps, int out_dtype, void* stream) {
  int e;
  if ((e = check_c(c)) || (e = check_view(x, c, OCTIC_F32)) || (e = check_view(y, c, out_dtype))) return e;
  if (M <= 0 || 8 * c > kMaxV * 256) return OCTIC_ESHAPE;
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
  if (out_dtype == OCTIC_F32)
    ln_fwd_kernel<float><<<grid, 256, 0, s>>>(vx, vy, a[0], a[1], a[2], a[3], a[4], beta, stats, M, c, eps);
  else if (out_dtype == OCTIC_BF16)
    ln_fwd_kernel<bf16><<<grid, 256, 0, s>>>(vx, vy, a[0], a[1], a[2], a[3], a[4], beta, stats, M, c, eps);
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
  if (M <= 0 || 8 * c > kMaxV * 256) return OCTIC_ESHAPE;
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
  if (g_dtype == OCTIC_F32)
    ln_bwd_kernel<float><<<grid, 256, smem, s>>>(vg, vx, stats, a[0], a[1], a[2], a[3], a[4], vr, dres ? 1 : 0, vd, partials, M, c);
  else if (g_dtype == OCTIC_BF16)
    ln_bwd_kernel<bf16><<<grid, 256, smem, s>>>(vg, vx, stats, a[0], a[1], a[2], a[3], a[4], vr, dres ? 1 : 0, vd, partials, M, c);
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
  ln_bwd_finish_kernel<<<(7 * c + 255) / 256, 256, 0, (hipStream_t)stream>>>(partials, nblk, c, d[0], d[1], d[2], d[3], d[4], dbeta);
  return launch_status();
}

}  // extern "C"
