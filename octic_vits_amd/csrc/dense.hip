// Row kernels of the standard (non-equivariant) half of a hybrid octic ViT: plain LayerNorm over dense rows and the
// block tail  x + rs*gamma*y  (layer scale, stochastic depth, residual).  All of them are HBM-bound: one wave owns a
// row, a lane owns the same NV 16-byte column chunks of every row it visits, so the per-column parameters and the
// per-column gradient partial sums live in registers for the whole launch.
//
// Reference: deit/models_v2.py Layer_scale_init_Block (norm1/norm2 = nn.LayerNorm(eps=1e-6),
// x = x + drop_path(gamma_1 * attn(norm1(x)))), timm drop_path (per-sample mask / keep_prob).
#include "octic_common.hpp"

namespace octic {

constexpr int kDenseWaves = 16;         // waves per workgroup of the backward kernels (16 x 256 = the same 4096 waves as 8 x 512, half the slabs)
constexpr int kDenseMaxBlocks = 256;    // partial-sum slabs per launch

template <typename T> struct Row4;
template <> struct Row4<float> {
  typedef f32x4 vec;
  static __device__ inline f32x4 load(const float* p) { return *(const f32x4*)p; }
  static __device__ inline f32x4 load_raw(const float* p) { return *(const f32x4*)p; }
  static __device__ inline f32x4 widen(f32x4 v) { return v; }
  static __device__ inline void store(float* p, f32x4 v) { *(f32x4*)p = v; }
};
template <> struct Row4<bf16> {
  typedef bf16x4 vec;
  static __device__ inline f32x4 load(const bf16* p) {
    bf16x4 h = *(const bf16x4*)p;
    return f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
  }
  static __device__ inline bf16x4 load_raw(const bf16* p) { return *(const bf16x4*)p; }
  static __device__ inline f32x4 widen(bf16x4 h) { return f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]}; }
  static __device__ inline void store(bf16* p, f32x4 v) {
    *(bf16x4*)p = bf16x4{(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
  }
};

__device__ inline float hsum(f32x4 v) { return (v[0] + v[1]) + (v[2] + v[3]); }

// ------------------------------------------------------------------------------------------ LayerNorm forward
template <typename TOUT, int NV>
__global__ __launch_bounds__(256) void dense_ln_fwd_kernel(const float* __restrict__ x, TOUT* __restrict__ y,
                                                           const float* __restrict__ w, const float* __restrict__ b,
                                                           float* __restrict__ stats, long rows, int d, float eps,
                                                           const int* __restrict__ rowmap = nullptr,
                                                           float* __restrict__ xcopy = nullptr) {
  // rowmap: row r of y / stats / xcopy is row rowmap[r] of x (batch-subset stochastic depth: the kept samples of a stream);
  // xcopy: the rows as read, compact (what the backward needs once the stream has been edited in place)
  const int lane = threadIdx.x & 63;
  const long nw = (long)gridDim.x * 4;
  f32x4 wv[NV], bv[NV];
  bool ok[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int col = (i * 64 + lane) * 4;
    ok[i] = col < d;
    wv[i] = (ok[i] && w) ? *(const f32x4*)(w + col) : f32x4{1.f, 1.f, 1.f, 1.f};
    bv[i] = (ok[i] && b) ? *(const f32x4*)(b + col) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const float inv_d = 1.0f / (float)d;
  for (long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += nw) {
    const float* xr = x + (rowmap ? (long)rowmap[r] : r) * d;
    f32x4 xv[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      xv[i] = ok[i] ? *(const f32x4*)(xr + (i * 64 + lane) * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
      s += hsum(xv[i]);
    }
    if (xcopy) {
#pragma unroll
      for (int i = 0; i < NV; ++i)
        if (ok[i]) *(f32x4*)(xcopy + r * d + (i * 64 + lane) * 4) = xv[i];
    }
    const float mean = wave_total(s) * inv_d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      f32x4 c = xv[i] - mean;
      xv[i] = c;
      if (ok[i]) q += hsum(c * c);
    }
    const float rstd = rsqrtf(wave_total(q) * inv_d + eps);
    TOUT* yr = y + r * d;
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if (ok[i]) Row4<TOUT>::store(yr + (i * 64 + lane) * 4, xv[i] * rstd * wv[i] + bv[i]);
    if (lane == 0) {
      stats[2 * r] = mean;
      stats[2 * r + 1] = rstd;
    }
  }
}

// Block tail + the LayerNorm that follows it, one row pass: xout = x + rs[row / rps] * gamma * yb (the residual stream after
// the branch: same expression as scale_residual_fwd_kernel), y = LayerNorm(xout).  The stream is written once and not read
// back (the two separate kernels move 210 + 126 MB per launch at ViT-H, this one 252).
template <typename TY, typename TOUT, int NV>
__global__ __launch_bounds__(256) void dense_resid_ln_fwd_kernel(const float* __restrict__ x, const TY* __restrict__ yb,
                                                                 const float* __restrict__ gamma, const float* __restrict__ rs,
                                                                 long rps, float* __restrict__ xout, TOUT* __restrict__ y,
                                                                 const float* __restrict__ w, const float* __restrict__ b,
                                                                 float* __restrict__ stats, long rows, int d, float eps) {
  const int lane = threadIdx.x & 63;
  const long nw = (long)gridDim.x * 4;
  f32x4 wv[NV], bv[NV], gm[NV];
  bool ok[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int col = (i * 64 + lane) * 4;
    ok[i] = col < d;
    wv[i] = (ok[i] && w) ? *(const f32x4*)(w + col) : f32x4{1.f, 1.f, 1.f, 1.f};
    bv[i] = (ok[i] && b) ? *(const f32x4*)(b + col) : f32x4{0.f, 0.f, 0.f, 0.f};
    gm[i] = (ok[i] && gamma) ? *(const f32x4*)(gamma + col) : f32x4{1.f, 1.f, 1.f, 1.f};
  }
  const float inv_d = 1.0f / (float)d;
  for (long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += nw) {
    const float* xr = x + r * d;
    const TY* yr_in = yb + r * d;
    float* xo = xout + r * d;
    f32x4 xv[NV];
    float s = 0.f;
    const float rsv = rs ? rs[r / rps] : 1.f;
    if (NV <= 6 && d == NV * 256) {   // every lane has all chunks: the 2 NV loads of the row go out before the first use
      typename Row4<TY>::vec yv[NV];
#pragma unroll
      for (int i = 0; i < NV; ++i) xv[i] = *(const f32x4*)(xr + (i * 64 + lane) * 4);
#pragma unroll
      for (int i = 0; i < NV; ++i) yv[i] = Row4<TY>::load_raw(yr_in + (i * 64 + lane) * 4);
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        f32x4 sc = gm[i];
        if (rs) sc *= rsv;
        xv[i] = xv[i] + sc * Row4<TY>::widen(yv[i]);
        *(f32x4*)(xo + (i * 64 + lane) * 4) = xv[i];
        s += hsum(xv[i]);
      }
    } else {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        xv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (ok[i]) {
          const int col = (i * 64 + lane) * 4;
          f32x4 sc = gm[i];
          if (rs) sc *= rsv;
          xv[i] = *(const f32x4*)(xr + col) + sc * Row4<TY>::load(yr_in + col);
          *(f32x4*)(xo + col) = xv[i];
        }
        s += hsum(xv[i]);
      }
    }
    const float mean = wave_total(s) * inv_d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      f32x4 c = xv[i] - mean;
      xv[i] = c;
      if (ok[i]) q += hsum(c * c);
    }
    const float rstd = rsqrtf(wave_total(q) * inv_d + eps);
    TOUT* yr = y + r * d;
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if (ok[i]) Row4<TOUT>::store(yr + (i * 64 + lane) * 4, xv[i] * rstd * wv[i] + bv[i]);
    if (lane == 0) {
      stats[2 * r] = mean;
      stats[2 * r + 1] = rstd;
    }
  }
}

// Fold the kDenseWaves per-wave register partials of a workgroup into its slab [2][d] (deterministic order).
template <int NV>
__device__ inline void slab_reduce(float* lds, float* slab, const f32x4 (&p0)[NV], const f32x4 (&p1)[NV],
                                   const bool (&ok)[NV], int d) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  for (int wv = 0; wv < nwaves; ++wv) {
    if (wave == wv) {
#pragma unroll
      for (int i = 0; i < NV; ++i)
        if (ok[i]) {
          const int col = (i * 64 + lane) * 4;
          f32x4* a = (f32x4*)(lds + col);
          f32x4* c = (f32x4*)(lds + d + col);
          if (wv == 0) {
            *a = p0[i];
            *c = p1[i];
          } else {
            *a += p0[i];
            *c += p1[i];
          }
        }
    }
    __syncthreads();
  }
  for (int j = threadIdx.x * 4; j < 2 * d; j += blockDim.x * 4) *(f32x4*)(slab + j) = *(const f32x4*)(lds + j);
}

// ------------------------------------------------------------------------------------------ LayerNorm backward
// dx = rstd*(g - mean(g) - xhat*mean(g*xhat)) + dres,  g = gy*w;  slab partials: dw += gy*xhat, db += gy.
template <typename TG, int NV>
__global__ __launch_bounds__(kDenseWaves * 64, 4) void dense_ln_bwd_kernel(
    const TG* __restrict__ gy, const float* __restrict__ x, const float* __restrict__ w,
    const float* __restrict__ stats, const float* __restrict__ dres, float* __restrict__ dx,
    float* __restrict__ partials, long rows, int d, const int* __restrict__ rowmap = nullptr) {
  // rowmap: dres and dx are rows rowmap[r] of a larger tensor (the stream's cotangent, edited in place when dres == dx)
  extern __shared__ float lds[];             // [2][d] slab image | [d] weights (kept out of the register budget)
  const int lane = threadIdx.x & 63;
  const long nw = (long)gridDim.x * kDenseWaves;
  float* wl = lds + 2 * d;
  for (int j = threadIdx.x; j < d; j += kDenseWaves * 64) wl[j] = w ? w[j] : 1.f;
  f32x4 pw[NV], pb[NV];
  bool ok[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    ok[i] = (i * 64 + lane) * 4 < d;
    pw[i] = pb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  __syncthreads();
  const float inv_d = 1.0f / (float)d;
  for (long r = (long)blockIdx.x * kDenseWaves + (threadIdx.x >> 6); r < rows; r += nw) {
    const float mean = stats[2 * r], rstd = stats[2 * r + 1];
    f32x4 xh[NV], g[NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const long o = r * d + (i * 64 + lane) * 4;
      if (ok[i]) {
        xh[i] = (*(const f32x4*)(x + o) - mean) * rstd;
        f32x4 gyv = Row4<TG>::load(gy + o);
        pw[i] += gyv * xh[i];
        pb[i] += gyv;
        g[i] = gyv * *(const f32x4*)(wl + (i * 64 + lane) * 4);
        s1 += hsum(g[i]);
        s2 += hsum(g[i] * xh[i]);
      }
    }
    const float m1 = wave_total(s1) * inv_d, m2 = wave_total(s2) * inv_d;
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if (ok[i]) {
        const long o = (rowmap ? (long)rowmap[r] : r) * d + (i * 64 + lane) * 4;
        f32x4 v = (g[i] - m1 - xh[i] * m2) * rstd;
        if (dres) v += *(const f32x4*)(dres + o);
        *(f32x4*)(dx + o) = v;
      }
  }
  if (partials) slab_reduce<NV>(lds, partials + (long)blockIdx.x * 2 * d, pw, pb, ok, d);
}

// The same for rows of exactly 256 NV columns (d = 1280 at ViT-H), eight waves per workgroup: every lane has all NV
// chunks, so nothing is predicated and ALL loads of a row (x, gy, dres: 3 NV wave-instructions = 15 KiB per wave at NV = 5)
// are requested before the first use.  The predicated kernel above compiles to load / s_waitcnt vmcnt(0) pairs - two
// loads in flight per wave, ten serial round trips per row, 32 KiB in flight per CU against the ~60 KiB a CU needs to
// cover the memory latency - and its 128-register budget (16 waves) has no room to hold a row's loads; 8 waves with
// 256 registers do, and 120 KiB in flight per CU.
template <typename TG, int NV>
#ifndef OCTIC_DLNBWD_WAVES
#define OCTIC_DLNBWD_WAVES 8
#endif
__global__ __launch_bounds__(OCTIC_DLNBWD_WAVES * 64, OCTIC_DLNBWD_WAVES / 4) void dense_ln_bwd_wide_kernel(
    const TG* __restrict__ gy, const float* __restrict__ x, const float* __restrict__ w,
    const float* __restrict__ stats, const float* __restrict__ dres, float* __restrict__ dx,
    float* __restrict__ partials, long rows, int d, const int* __restrict__ rowmap = nullptr) {
  extern __shared__ float lds[];             // [2][d] slab image | [d] weights
  const int lane = threadIdx.x & 63, nwaves = blockDim.x >> 6;
  const long nw = (long)gridDim.x * nwaves;
  float* wl = lds + 2 * d;
  for (int j = threadIdx.x; j < d; j += blockDim.x) wl[j] = w ? w[j] : 1.f;
  f32x4 pw[NV], pb[NV];
  bool ok[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    ok[i] = true;
    pw[i] = pb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  __syncthreads();
  const float inv_d = 1.0f / (float)d;
  for (long r = (long)blockIdx.x * nwaves + (threadIdx.x >> 6); r < rows; r += nw) {
    const long o = r * d + lane * 4;
    const long om = (rowmap ? (long)rowmap[r] : r) * d + lane * 4;      // row of dres / dx
    f32x4 xh[NV], g[NV], dr[NV];
    typename Row4<TG>::vec gr[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) xh[i] = *(const f32x4*)(x + o + i * 256);
#pragma unroll
    for (int i = 0; i < NV; ++i) gr[i] = Row4<TG>::load_raw(gy + o + i * 256);
#pragma unroll
    for (int i = 0; i < NV; ++i) dr[i] = dres ? *(const f32x4*)(dres + om + i * 256) : f32x4{0.f, 0.f, 0.f, 0.f};
    const float mean = stats[2 * r], rstd = stats[2 * r + 1];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      xh[i] = (xh[i] - mean) * rstd;
      const f32x4 gyv = Row4<TG>::widen(gr[i]);
      pw[i] += gyv * xh[i];
      pb[i] += gyv;
      g[i] = gyv * *(const f32x4*)(wl + (i * 64 + lane) * 4);
      s1 += hsum(g[i]);
      s2 += hsum(g[i] * xh[i]);
    }
    const float m1 = wave_total(s1) * inv_d, m2 = wave_total(s2) * inv_d;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      f32x4 v = (g[i] - m1 - xh[i] * m2) * rstd;
      if (dres) v += dr[i];
      *(f32x4*)(dx + om + i * 256) = v;
    }
  }
  if (partials) slab_reduce<NV>(lds, partials + (long)blockIdx.x * 2 * d, pw, pb, ok, d);
}

// LayerNorm backward + the backward of the residual tail in front of it, one row pass (whole-chunk rows, eight waves):
//   dx  = LN'(gy) + dres                      (cotangent of the stream that entered the norm: f32, written once)
//   gyb = rs * gamma * dx                     (cotangent of the branch output yb that was added to make that stream)
//   slabs: dw += gy*xhat, db += gy  |  d gamma += rs*dx*yb, bias gradient / gamma += rs*dx
// The two separate kernels (dense_ln_bwd_wide + scale_residual_bwd) write dx and read it back: 463 MB per launch at
// ViT-H, this one 379.  Arithmetic per element is that of the two kernels (dx is used as stored).
template <typename TG, int NV>
__global__ __launch_bounds__(512, 2) void dense_ln_bwd_tail_kernel(
    const TG* __restrict__ gy, const float* __restrict__ x, const float* __restrict__ w,
    const float* __restrict__ stats, const float* __restrict__ dres, float* __restrict__ dx,
    float* __restrict__ partials, const TG* __restrict__ yb, const float* __restrict__ gamma,
    const float* __restrict__ rs, long rps, TG* __restrict__ gyb, float* __restrict__ partials2, long rows, int d) {
  extern __shared__ float lds[];             // [2][d] slab image | [d] LayerNorm weights | [d] layer scale
  const int lane = threadIdx.x & 63, nwaves = blockDim.x >> 6;
  const long nw = (long)gridDim.x * nwaves;
  float* wl = lds + 2 * d;
  float* gl = lds + 3 * d;
  for (int j = threadIdx.x; j < d; j += blockDim.x) {
    wl[j] = w ? w[j] : 1.f;
    gl[j] = gamma ? gamma[j] : 1.f;
  }
  f32x4 pw[NV], pb[NV], p0[NV], p1[NV];
  bool ok[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    ok[i] = true;
    pw[i] = pb[i] = p0[i] = p1[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  __syncthreads();
  const float inv_d = 1.0f / (float)d;
  for (long r = (long)blockIdx.x * nwaves + (threadIdx.x >> 6); r < rows; r += nw) {
    const long o = r * d + lane * 4;
    f32x4 xh[NV], g[NV], dr[NV];
    typename Row4<TG>::vec gr[NV], yr[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) xh[i] = *(const f32x4*)(x + o + i * 256);
#pragma unroll
    for (int i = 0; i < NV; ++i) gr[i] = Row4<TG>::load_raw(gy + o + i * 256);
#pragma unroll
    for (int i = 0; i < NV; ++i) dr[i] = dres ? *(const f32x4*)(dres + o + i * 256) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NV; ++i) yr[i] = Row4<TG>::load_raw(yb + o + i * 256);
    const float mean = stats[2 * r], rstd = stats[2 * r + 1];
    const float sc = rs ? rs[r / rps] : 1.f;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      xh[i] = (xh[i] - mean) * rstd;
      const f32x4 gyv = Row4<TG>::widen(gr[i]);
      pw[i] += gyv * xh[i];
      pb[i] += gyv;
      g[i] = gyv * *(const f32x4*)(wl + (i * 64 + lane) * 4);
      s1 += hsum(g[i]);
      s2 += hsum(g[i] * xh[i]);
    }
    const float m1 = wave_total(s1) * inv_d, m2 = wave_total(s2) * inv_d;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      f32x4 v = (g[i] - m1 - xh[i] * m2) * rstd;
      if (dres) v += dr[i];
      *(f32x4*)(dx + o + i * 256) = v;
      const f32x4 t = v * sc;                                  // scale_residual_bwd_kernel: g = gout * s
      p1[i] += t;
      p0[i] += t * Row4<TG>::widen(yr[i]);
      Row4<TG>::store(gyb + o + i * 256, t * *(const f32x4*)(gl + (i * 64 + lane) * 4));
    }
  }
  if (partials) slab_reduce<NV>(lds, partials + (long)blockIdx.x * 2 * d, pw, pb, ok, d);
  __syncthreads();
  if (partials2) slab_reduce<NV>(lds, partials2 + (long)blockIdx.x * 2 * d, p0, p1, ok, d);
}

// ------------------------------------------------------------------------------------------ block tail
// out = x + rs[row / rps] * gamma[col] * y
template <typename TY>
__global__ __launch_bounds__(256) void scale_residual_fwd_kernel(const float* __restrict__ x, const TY* __restrict__ y,
                                                                 const float* __restrict__ gamma,
                                                                 const float* __restrict__ rs, long rps,
                                                                 float* __restrict__ out, long rows, int d,
                                                                 const int* __restrict__ rowmap = nullptr) {
  // rowmap: row r of the result is row rowmap[r] of `out` (the kept samples written back into the stream)
  const int d4 = d >> 2;
  const long n4 = rows * d4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const long r = i / d4;
    const int col = (int)(i - r * d4) * 4;
    f32x4 s = gamma ? *(const f32x4*)(gamma + col) : f32x4{1.f, 1.f, 1.f, 1.f};
    if (rs) s *= rs[r / rps];
    const long io = rowmap ? (long)rowmap[r] * d + col : i * 4;
    *(f32x4*)(out + io) = *(const f32x4*)(x + i * 4) + s * Row4<TY>::load(y + i * 4);
  }
}

// gy = rs*gamma*gout (dtype of y);  slab partials: p0 += rs*gout*y (d gamma), p1 += rs*gout (bias grad / gamma).
template <typename TY, int NV>
__global__ __launch_bounds__(kDenseWaves * 64) void scale_residual_bwd_kernel(
    const float* __restrict__ gout, const TY* __restrict__ y, const float* __restrict__ gamma,
    const float* __restrict__ rs, long rps, TY* __restrict__ gy, float* __restrict__ partials, long rows, int d,
    const int* __restrict__ rowmap = nullptr) {
  // rowmap: row r of the cotangent is row rowmap[r] of gout (the stream's cotangent; y, gy and rs stay compact)
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63;
  const long nw = (long)gridDim.x * kDenseWaves;
  f32x4 gm[NV], p0[NV], p1[NV];
  bool ok[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int col = (i * 64 + lane) * 4;
    ok[i] = col < d;
    gm[i] = (ok[i] && gamma) ? *(const f32x4*)(gamma + col) : f32x4{1.f, 1.f, 1.f, 1.f};
    p0[i] = p1[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  if (NV <= 5 && d == NV * 256 && partials) {      // (wider rows would spill at the 128-register budget of 16 waves)
    // every lane has all NV chunks (d = 1280 at ViT-H): no per-chunk predicate, so the 2 NV loads of a row are all
    // requested before the first use (the predicated loop below compiles to load / vmcnt(0) / load / vmcnt(0) / store per
    // chunk: ten serial round trips per row, hidden only by occupancy)
    for (long r = (long)blockIdx.x * kDenseWaves + (threadIdx.x >> 6); r < rows; r += nw) {
      const float s = rs ? rs[r / rps] : 1.f;
      const long o = r * d + lane * 4;
      const long og = (rowmap ? (long)rowmap[r] : r) * d + lane * 4;
      f32x4 g[NV];
      typename Row4<TY>::vec yv[NV];
#pragma unroll
      for (int i = 0; i < NV; ++i) g[i] = *(const f32x4*)(gout + og + i * 256);
#pragma unroll
      for (int i = 0; i < NV; ++i) yv[i] = Row4<TY>::load_raw(y + o + i * 256);
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        g[i] *= s;
        p1[i] += g[i];
        p0[i] += g[i] * Row4<TY>::widen(yv[i]);
        Row4<TY>::store(gy + o + i * 256, g[i] * gm[i]);
      }
    }
  } else {
    for (long r = (long)blockIdx.x * kDenseWaves + (threadIdx.x >> 6); r < rows; r += nw) {
      const float s = rs ? rs[r / rps] : 1.f;
#pragma unroll
      for (int i = 0; i < NV; ++i)
        if (ok[i]) {
          const long o = r * d + (i * 64 + lane) * 4;
          f32x4 g = *(const f32x4*)(gout + (rowmap ? (long)rowmap[r] : r) * d + (i * 64 + lane) * 4) * s;
          p1[i] += g;
          if (partials) p0[i] += g * Row4<TY>::load(y + o);
          Row4<TY>::store(gy + o, g * gm[i]);
        }
    }
  }
  if (partials) slab_reduce<NV>(lds, partials + (long)blockIdx.x * 2 * d, p0, p1, ok, d);
}

// out0[j] = sum_b partials[b][j], out1[j] = scale1[j] * sum_b partials[b][d + j]   (16 columns x 16 slab groups)
__global__ __launch_bounds__(256) void dense_finish_kernel(const float* __restrict__ partials, int nblocks, int d,
                                                           float* __restrict__ out0, float* __restrict__ out1,
                                                           const float* __restrict__ scale1) {
  __shared__ float red[16][17];
  const int cx = threadIdx.x & 15, gy = threadIdx.x >> 4;
  const int j = blockIdx.x * 16 + cx;
  float s = 0.f;
  if (j < 2 * d) {
    int b = gy;                                   // fixed summation order, four loads in flight
    for (; b + 48 < nblocks; b += 64) {
      const float t0 = partials[(long)b * 2 * d + j], t1 = partials[(long)(b + 16) * 2 * d + j],
                  t2 = partials[(long)(b + 32) * 2 * d + j], t3 = partials[(long)(b + 48) * 2 * d + j];
      s += t0; s += t1; s += t2; s += t3;
    }
    for (; b < nblocks; b += 16) s += partials[(long)b * 2 * d + j];
  }
  red[gy][cx] = s;
  __syncthreads();
  if (gy == 0 && j < 2 * d) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][cx];
    if (j < d) {
      if (out0) out0[j] = t;
    } else if (out1) {
      out1[j - d] = scale1 ? t * scale1[j - d] : t;
    }
  }
}

// The same reduction for up to 64 slab sets in one launch (blockIdx.y = job): the backward of a standard block ends in six
// of these 5-us reductions (LayerNorm weight / bias, layer-scale gamma, projection and MLP bias gradients), 96 launches per
// ViT-H step whose results nothing reads before the optimizer.  Job fields are 8- and 4-byte members only: hipcc mis-addresses
// a wave-uniform index into a 4-byte kernel-argument array when a 2-byte array shares the index (tools/dbg/kernarg_index_test.hip).
struct FinishPack {
  octic_finish_job j[64];
};
__global__ __launch_bounds__(256) void dense_finish_batch_kernel(FinishPack pack) {
  __shared__ float red[16][17];
  const octic_finish_job& job = pack.j[blockIdx.y];
  const float* __restrict__ partials = job.partials;
  const int nblocks = job.nblocks, d = job.d;
  if ((int)blockIdx.x * 16 >= 2 * d) return;                    // (uniform over the workgroup)
  const int cx = threadIdx.x & 15, gy = threadIdx.x >> 4;
  const int j = blockIdx.x * 16 + cx;
  float s = 0.f;
  if (j < 2 * d) {
    int b = gy;                                   // the summation order of dense_finish_kernel: results are bit-identical
    for (; b + 48 < nblocks; b += 64) {
      const float t0 = partials[(long)b * 2 * d + j], t1 = partials[(long)(b + 16) * 2 * d + j],
                  t2 = partials[(long)(b + 32) * 2 * d + j], t3 = partials[(long)(b + 48) * 2 * d + j];
      s += t0; s += t1; s += t2; s += t3;
    }
    for (; b < nblocks; b += 16) s += partials[(long)b * 2 * d + j];
  }
  red[gy][cx] = s;
  __syncthreads();
  if (gy == 0 && j < 2 * d) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][cx];
    if (j < d) {
      if (job.out0) job.out0[j] = t;
    } else if (job.out1) {
      job.out1[j - d] = job.scale1 ? t * job.scale1[j - d] : t;
    }
  }
}

// ------------------------------------------------------------------------------------------ GELU backward (+ bias grad)
// dh = gelu'(h) * g for dense bf16 [rows, d] (the standard MLP's hidden activations, d = 4 * dim) and, on the side,
// the column sums of dh = gradient of the bias of the projection that produced h.  A workgroup is 64 column chunks
// (8 columns = 16 bytes each) x 4 row phases and walks a strip of rows; per-thread partial sums, one LDS fold over
// the row phases, slab [row_block][d].
constexpr int kGeluRowBlocks = 256;    // x 10 column blocks at d = 5120: 10 waves per SIMD's worth of work in flight
__global__ __launch_bounds__(256) void dense_gelu_bwd_kernel(const bf16* __restrict__ h, const bf16* __restrict__ g,
                                                             bf16* __restrict__ dh, float* __restrict__ partials,
                                                             long rows, int d) {
  __shared__ float red[4][64][8];
  const int cl = threadIdx.x & 63, ph = threadIdx.x >> 6;
  const int col = (blockIdx.x * 64 + cl) * 8;
  const long per = (rows + kGeluRowBlocks - 1) / kGeluRowBlocks;
  const long r0 = (long)blockIdx.y * per, r1 = r0 + per < rows ? r0 + per : rows;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (col < d) {
    long r = r0 + ph;
    for (; r + 4 < r1; r += 8) {          // two rows per trip: four 16-byte loads in flight per lane
      const bf16x8 hv0 = *(const bf16x8*)(h + r * d + col), hv1 = *(const bf16x8*)(h + (r + 4) * d + col);
      const bf16x8 gv0 = *(const bf16x8*)(g + r * d + col), gv1 = *(const bf16x8*)(g + (r + 4) * d + col);
      bf16x8 o0, o1;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        o0[e] = (bf16)(gelu_grad((float)hv0[e]) * (float)gv0[e]);
        o1[e] = (bf16)(gelu_grad((float)hv1[e]) * (float)gv1[e]);
        acc[e] += (float)o0[e] + (float)o1[e];      // sum what the GEMMs downstream actually see
      }
      *(bf16x8*)(dh + r * d + col) = o0;
      *(bf16x8*)(dh + (r + 4) * d + col) = o1;
    }
    for (; r < r1; r += 4) {
      const bf16x8 hv = *(const bf16x8*)(h + r * d + col);
      const bf16x8 gv = *(const bf16x8*)(g + r * d + col);
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        o[e] = (bf16)(gelu_grad((float)hv[e]) * (float)gv[e]);
        acc[e] += (float)o[e];
      }
      *(bf16x8*)(dh + r * d + col) = o;
    }
  }
  if (!partials) return;
#pragma unroll
  for (int e = 0; e < 8; ++e) red[ph][cl][e] = acc[e];
  __syncthreads();
  if (ph == 0 && col < d) {
    float* out = partials + (long)blockIdx.y * d + col;
#pragma unroll
    for (int e = 0; e < 8; ++e) out[e] = (red[0][cl][e] + red[1][cl][e]) + (red[2][cl][e] + red[3][cl][e]);
  }
}

// Column sums of a bf16 [rows, d] tensor (row stride ld) in f32: kGeluRowBlocks slabs [d] in a fixed order, reduced by
// dense_finish_kernel.  The bias gradient of the fused-qkv projection (deit/vit.py:33; autograd's `grad.sum(0)`), whose
// cotangent comes out of the attention backward and is read by nothing else row-wise.
__global__ __launch_bounds__(256) void dense_colsum_kernel(const bf16* __restrict__ g, float* __restrict__ partials,
                                                           long rows, int d, long ld) {
  __shared__ float red[4][64][8];
  const int cl = threadIdx.x & 63, ph = threadIdx.x >> 6;
  const int col = (blockIdx.x * 64 + cl) * 8;
  const long per = (rows + kGeluRowBlocks - 1) / kGeluRowBlocks;
  const long r0 = (long)blockIdx.y * per, r1 = r0 + per < rows ? r0 + per : rows;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (col < d) {
    long r = r0 + ph;
    for (; r + 12 < r1; r += 16) {        // four rows per trip: four 16-byte loads in flight per lane
      const bf16x8 v0 = *(const bf16x8*)(g + r * ld + col), v1 = *(const bf16x8*)(g + (r + 4) * ld + col);
      const bf16x8 v2 = *(const bf16x8*)(g + (r + 8) * ld + col), v3 = *(const bf16x8*)(g + (r + 12) * ld + col);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += ((float)v0[e] + (float)v1[e]) + ((float)v2[e] + (float)v3[e]);
    }
    for (; r < r1; r += 4) {
      const bf16x8 v = *(const bf16x8*)(g + r * ld + col);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += (float)v[e];
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[ph][cl][e] = acc[e];
  __syncthreads();
  if (ph == 0 && col < d) {
    float* out = partials + (long)blockIdx.y * d + col;
#pragma unroll
    for (int e = 0; e < 8; ++e) out[e] = (red[0][cl][e] + red[1][cl][e]) + (red[2][cl][e] + red[3][cl][e]);
  }
}

static inline int dense_nv(int d) { return (d + 255) / 256; }
static inline int dense_check(long rows, int d) {
  if (rows < 0 || d <= 0 || (d & 3) || d > 2048) return OCTIC_ESHAPE;
  return OCTIC_OK;
}
static inline int dense_blocks(long rows) {
  long b = (rows + kDenseWaves - 1) / kDenseWaves;
  return (int)(b < 1 ? 1 : (b > kDenseMaxBlocks ? kDenseMaxBlocks : b));
}

#define DENSE_NV_SWITCH(nv, CALL) \
  switch (nv) {                   \
    case 1: { constexpr int NV = 1; CALL; } break; \
    case 2: { constexpr int NV = 2; CALL; } break; \
    case 3: { constexpr int NV = 3; CALL; } break; \
    case 4: { constexpr int NV = 4; CALL; } break; \
    case 5: { constexpr int NV = 5; CALL; } break; \
    case 6: { constexpr int NV = 6; CALL; } break; \
    case 7: { constexpr int NV = 7; CALL; } break; \
    default: { constexpr int NV = 8; CALL; } break; \
  }

// Compute-dtype copies of all nn.Linear weights of the standard half in ONE launch, run after the optimizer step:
// wb = bf16(W) [N,K] (forward operand) and wt = bf16(W)^T [K,N] (operand of the input-gradient GEMM, which is then an
// NT problem like the forward).  One 64 x 64 tile per workgroup, transposed through LDS.
template <typename TS>
__global__ __launch_bounds__(256) void dense_prep_batch_kernel(const octic_dense_prep_item* __restrict__ items, int n_items) {
  __shared__ float tile[64][65];
  const int b = blockIdx.x;
  // items are sorted by block_begin: this block's item = (number of items starting at or before it) - 1, counted by the
  // whole workgroup at once (a per-thread scan of the table was most of a 64 x 64 tile's time)
  int cnt = 0;
  for (int i0 = 0; i0 < n_items; i0 += 256) {
    const int i = i0 + (int)threadIdx.x;
    cnt += __syncthreads_count(i < n_items && b >= items[i].block_begin);
  }
  const int it = cnt > 0 ? cnt - 1 : 0;
  const octic_dense_prep_item& I = items[it];
  const int N = I.N, K = I.K;
  const int lt = b - I.block_begin;
  const int kt = (K + 63) / 64;
  const int n0 = (lt / kt) * 64, k0 = (lt % kt) * 64;
  const TS* w = (const TS*)I.src;
  bf16* wb = (bf16*)I.wb;
  bf16* wt = (bf16*)I.wt;
  const bool vec = (N % 8) == 0 && (K % 8) == 0;             // 16-byte rows both ways (every nn.Linear of the models)
  if (vec) {
    // read: 8 consecutive k of one row n per thread (two rows per thread), write: 8 consecutive n of one row k
    const int c8 = (threadIdx.x & 7) * 8, r = threadIdx.x >> 3;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int n = n0 + r + 32 * h, k = k0 + c8;
      float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      if (n < N && k < K) {
        load8<TS>(w + (int64_t)n * K + k, v);
        if (wb) store8<bf16>(wb + (int64_t)n * K + k, v);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) tile[r + 32 * h][c8 + e] = v[e];
    }
    __syncthreads();
    if (wt) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int k = k0 + r + 32 * h, n = n0 + c8;
        if (k < K && n < N) {
          float v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = tile[c8 + e][r + 32 * h];
          store8<bf16>(wt + (int64_t)k * N + n, v);
        }
      }
    }
    return;
  }
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll 4
  for (int r = ty; r < 64; r += 4) {
    const int n = n0 + r, k = k0 + tx;
    float v = 0.f;
    if (n < N && k < K) {
      v = (float)w[(int64_t)n * K + k];
      if (wb) wb[(int64_t)n * K + k] = (bf16)v;
    }
    tile[r][tx] = v;
  }
  __syncthreads();
  if (wt) {
#pragma unroll 4
    for (int r = ty; r < 64; r += 4) {
      const int k = k0 + r, n = n0 + tx;
      if (k < K && n < N) wt[(int64_t)k * N + n] = (bf16)tile[tx][r];
    }
  }
}

}  // namespace octic

using namespace octic;

extern "C" {

int octic_dense_layernorm_fwd(const float* x, void* y, int y_dtype, const float* w, const float* b, float* stats,
                              int64_t rows, int d, float eps, void* stream) {
  return octic_dense_layernorm_fwd_rows(x, y, y_dtype, w, b, stats, rows, d, eps, nullptr, nullptr, stream);
}

int octic_dense_layernorm_fwd_rows(const float* x, void* y, int y_dtype, const float* w, const float* b, float* stats,
                                   int64_t rows, int d, float eps, const int* rowmap, float* xcopy, void* stream) {
  if (rows == 0) return OCTIC_OK;
  if (!x || !y || !stats) return OCTIC_ENULL;
  if (int e = dense_check(rows, d)) return e;
  if (y_dtype != OCTIC_F32 && y_dtype != OCTIC_BF16) return OCTIC_EDTYPE;
  long blocks = (rows + 15) / 16;            // 4 rows per wave: parameters are re-read once per 4 rows
  if (blocks > 4096) blocks = 4096;
  hipStream_t s = (hipStream_t)stream;
  if (y_dtype == OCTIC_BF16) {
    DENSE_NV_SWITCH(dense_nv(d), (dense_ln_fwd_kernel<bf16, NV><<<dim3((unsigned)blocks), dim3(256), 0, s>>>(x, (bf16*)y, w, b, stats, rows, d, eps, rowmap, xcopy)));
  } else {
    DENSE_NV_SWITCH(dense_nv(d), (dense_ln_fwd_kernel<float, NV><<<dim3((unsigned)blocks), dim3(256), 0, s>>>(x, (float*)y, w, b, stats, rows, d, eps, rowmap, xcopy)));
  }
  return launch_status();
}

int octic_dense_resid_layernorm_fwd(const float* x, const void* yb, int yb_dtype, const float* gamma, const float* rs,
                                    int64_t rows_per_scale, float* xout, void* y, int y_dtype, const float* w,
                                    const float* b, float* stats, int64_t rows, int d, float eps, void* stream) {
  if (rows == 0) return OCTIC_OK;
  if (!x || !yb || !xout || !y || !stats) return OCTIC_ENULL;
  if (int e = dense_check(rows, d)) return e;
  if ((yb_dtype != OCTIC_F32 && yb_dtype != OCTIC_BF16) || (y_dtype != OCTIC_F32 && y_dtype != OCTIC_BF16)) return OCTIC_EDTYPE;
  if (rs && rows_per_scale <= 0) return OCTIC_ESHAPE;
  long blocks = (rows + 7) / 8;              // 2 rows per wave (41-42 us at ViT-H against 46 with 4: tools/bench_rln.py)
  if (blocks > 8192) blocks = 8192;
  hipStream_t s = (hipStream_t)stream;
  const long rps = rs ? rows_per_scale : 1;
#define OCTIC_RLN(TYB, TO) \
  DENSE_NV_SWITCH(dense_nv(d), (dense_resid_ln_fwd_kernel<TYB, TO, NV><<<dim3((unsigned)blocks), dim3(256), 0, s>>>( \
      x, (const TYB*)yb, gamma, rs, rps, xout, (TO*)y, w, b, stats, rows, d, eps)))
  if (yb_dtype == OCTIC_BF16 && y_dtype == OCTIC_BF16) { OCTIC_RLN(bf16, bf16); }
  else if (yb_dtype == OCTIC_BF16) { OCTIC_RLN(bf16, float); }
  else if (y_dtype == OCTIC_BF16) { OCTIC_RLN(float, bf16); }
  else { OCTIC_RLN(float, float); }
#undef OCTIC_RLN
  return launch_status();
}

int octic_dense_blocks(int64_t rows) { return dense_blocks(rows); }

int octic_dense_layernorm_bwd(const void* gy, int g_dtype, const float* x, const float* w, const float* stats,
                              const float* dres, float* dx, float* partials, int64_t rows, int d, void* stream) {
  return octic_dense_layernorm_bwd_rows(gy, g_dtype, x, w, stats, dres, dx, partials, rows, d, nullptr, stream);
}

int octic_dense_layernorm_bwd_rows(const void* gy, int g_dtype, const float* x, const float* w, const float* stats,
                                   const float* dres, float* dx, float* partials, int64_t rows, int d, const int* rowmap,
                                   void* stream) {
  if (rows == 0) return OCTIC_OK;
  if (!gy || !x || !stats || !dx) return OCTIC_ENULL;
  if (int e = dense_check(rows, d)) return e;
  if (g_dtype != OCTIC_F32 && g_dtype != OCTIC_BF16) return OCTIC_EDTYPE;
  const int blocks = dense_blocks(rows);
  const size_t lds = (size_t)3 * d * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
  if (d == dense_nv(d) * 256 && dense_nv(d) <= 6) {      // whole-chunk rows: the unpredicated eight-wave kernel
    if (g_dtype == OCTIC_BF16) {
      DENSE_NV_SWITCH(dense_nv(d), (dense_ln_bwd_wide_kernel<bf16, NV><<<dim3(blocks), dim3(OCTIC_DLNBWD_WAVES * 64), lds, s>>>((const bf16*)gy, x, w, stats, dres, dx,
                                                            partials, rows, d, rowmap)));
    } else {
      DENSE_NV_SWITCH(dense_nv(d), (dense_ln_bwd_wide_kernel<float, NV><<<dim3(blocks), dim3(OCTIC_DLNBWD_WAVES * 64), lds, s>>>((const float*)gy, x, w, stats, dres, dx,
                                                             partials, rows, d, rowmap)));
    }
    return launch_status();
  }
  if (g_dtype == OCTIC_BF16) {
    DENSE_NV_SWITCH(dense_nv(d), (dense_ln_bwd_kernel<bf16, NV><<<dim3(blocks), dim3(kDenseWaves * 64), lds, s>>>((const bf16*)gy, x, w, stats, dres,
                                                     dx, partials, rows, d, rowmap)));
  } else {
    DENSE_NV_SWITCH(dense_nv(d), (dense_ln_bwd_kernel<float, NV><<<dim3(blocks), dim3(kDenseWaves * 64), lds, s>>>((const float*)gy, x, w, stats,
                                                     dres, dx, partials, rows, d, rowmap)));
  }
  return launch_status();
}

int octic_dense_layernorm_bwd_tail(const void* gy, const float* x, const float* w, const float* stats, const float* dres,
                                   float* dx, float* partials, const void* yb, const float* gamma, const float* rs,
                                   int64_t rows_per_scale, void* gyb, float* partials2, int64_t rows, int d, void* stream) {
  if (rows == 0) return OCTIC_OK;
  if (!gy || !x || !stats || !dx || !yb || !gyb) return OCTIC_ENULL;
  if (int e = dense_check(rows, d)) return e;
  if (d != dense_nv(d) * 256 || dense_nv(d) > 5) return OCTIC_ESHAPE;     // whole 256-column chunks only (callers fall back)
  if (rs && rows_per_scale <= 0) return OCTIC_ESHAPE;
  const int blocks = dense_blocks(rows);
  const size_t lds = (size_t)4 * d * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
  const long rps = rs ? rows_per_scale : 1;
  DENSE_NV_SWITCH(dense_nv(d), (dense_ln_bwd_tail_kernel<bf16, NV><<<dim3(blocks), dim3(512), lds, s>>>(
      (const bf16*)gy, x, w, stats, dres, dx, partials, (const bf16*)yb, gamma, rs, rps, (bf16*)gyb, partials2, rows, d)));
  return launch_status();
}

int octic_dense_finish(const float* partials, int nblocks, int d, float* out0, float* out1, const float* scale1,
                       void* stream) {
  if (!partials) return OCTIC_ENULL;
  if (nblocks <= 0 || d <= 0) return OCTIC_ESHAPE;
  hipLaunchKernelGGL(dense_finish_kernel, dim3((2 * d + 15) / 16), dim3(256), 0, (hipStream_t)stream, partials, nblocks,
                     d, out0, out1, scale1);
  return launch_status();
}

int octic_dense_finish_batch(const octic_finish_job* jobs, int njobs, void* stream) {
  if (!jobs) return OCTIC_ENULL;
  if (njobs < 0) return OCTIC_ESHAPE;
  for (int i = 0; i < njobs; ++i) {
    if (!jobs[i].partials) return OCTIC_ENULL;
    if (jobs[i].nblocks <= 0 || jobs[i].d <= 0) return OCTIC_ESHAPE;
  }
  for (int i0 = 0; i0 < njobs; i0 += 64) {
    const int n = njobs - i0 < 64 ? njobs - i0 : 64;
    FinishPack pack = {};
    int dmax = 0;
    for (int i = 0; i < n; ++i) {
      pack.j[i] = jobs[i0 + i];
      dmax = jobs[i0 + i].d > dmax ? jobs[i0 + i].d : dmax;
    }
    hipLaunchKernelGGL(dense_finish_batch_kernel, dim3((2 * dmax + 15) / 16, n), dim3(256), 0, (hipStream_t)stream, pack);
  }
  return launch_status();
}

int octic_dense_gelu_blocks(void) { return kGeluRowBlocks; }

int octic_dense_colsum(const void* g, int64_t rows, int d, int64_t ld, float* partials, void* stream) {
  if (!g || !partials) return OCTIC_ENULL;
  if (rows <= 0 || d <= 0 || (d & 7) || (ld & 7) || ld < d) return OCTIC_ESHAPE;
  if (((uintptr_t)g) & 15) return OCTIC_EALIGN;
  dense_colsum_kernel<<<dim3((d / 8 + 63) / 64, kGeluRowBlocks), dim3(256), 0, (hipStream_t)stream>>>(
      (const bf16*)g, partials, rows, d, ld);
  return launch_status();
}

int octic_dense_gelu_bwd(const void* h, const void* g, void* dh, float* partials, int64_t rows, int d, void* stream) {
  if (rows == 0) return OCTIC_OK;
  if (!h || !g || !dh) return OCTIC_ENULL;
  if (rows < 0 || d <= 0 || (d & 7)) return OCTIC_ESHAPE;
  if ((((uintptr_t)h) | ((uintptr_t)g) | ((uintptr_t)dh)) & 15) return OCTIC_EALIGN;
  dense_gelu_bwd_kernel<<<dim3((d / 8 + 63) / 64, kGeluRowBlocks), dim3(256), 0, (hipStream_t)stream>>>(
      (const bf16*)h, (const bf16*)g, (bf16*)dh, partials, rows, d);
  return launch_status();
}

int octic_scale_residual_fwd(const float* x, const void* y, int y_dtype, const float* gamma, const float* rs,
                             int64_t rows_per_scale, float* out, int64_t rows, int d, void* stream) {
  return octic_scale_residual_fwd_rows(x, y, y_dtype, gamma, rs, rows_per_scale, out, rows, d, nullptr, stream);
}

int octic_scale_residual_fwd_rows(const float* x, const void* y, int y_dtype, const float* gamma, const float* rs,
                                  int64_t rows_per_scale, float* out, int64_t rows, int d, const int* rowmap, void* stream) {
  if (rows == 0) return OCTIC_OK;
  if (!x || !y || !out) return OCTIC_ENULL;
  if (int e = dense_check(rows, d)) return e;
  if (y_dtype != OCTIC_F32 && y_dtype != OCTIC_BF16) return OCTIC_EDTYPE;
  if (rs && rows_per_scale <= 0) return OCTIC_ESHAPE;
  const long n4 = rows * (d >> 2);
  long blocks = (n4 + 1023) / 1024;          // 4 chunks per thread
  if (blocks > 8192) blocks = 8192;
  hipStream_t s = (hipStream_t)stream;
  if (y_dtype == OCTIC_BF16)
    scale_residual_fwd_kernel<bf16><<<dim3((unsigned)blocks), dim3(256), 0, s>>>(x, (const bf16*)y,
                       gamma, rs, rows_per_scale, out, rows, d, rowmap);
  else
    scale_residual_fwd_kernel<float><<<dim3((unsigned)blocks), dim3(256), 0, s>>>(x, (const float*)y,
                       gamma, rs, rows_per_scale, out, rows, d, rowmap);
  return launch_status();
}

int octic_scale_residual_bwd(const float* gout, const void* y, int y_dtype, const float* gamma, const float* rs,
                             int64_t rows_per_scale, void* gy, float* partials, int64_t rows, int d, void* stream) {
  return octic_scale_residual_bwd_rows(gout, y, y_dtype, gamma, rs, rows_per_scale, gy, partials, rows, d, nullptr, stream);
}

int octic_scale_residual_bwd_rows(const float* gout, const void* y, int y_dtype, const float* gamma, const float* rs,
                                  int64_t rows_per_scale, void* gy, float* partials, int64_t rows, int d, const int* rowmap,
                                  void* stream) {
  if (rows == 0) return OCTIC_OK;
  if (!gout || !gy || (partials && !y)) return OCTIC_ENULL;
  if (int e = dense_check(rows, d)) return e;
  if (y_dtype != OCTIC_F32 && y_dtype != OCTIC_BF16) return OCTIC_EDTYPE;
  if (rs && rows_per_scale <= 0) return OCTIC_ESHAPE;
  const int blocks = dense_blocks(rows);
  const size_t lds = (size_t)2 * d * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
  if (y_dtype == OCTIC_BF16) {
    DENSE_NV_SWITCH(dense_nv(d), (scale_residual_bwd_kernel<bf16, NV><<<dim3(blocks), dim3(kDenseWaves * 64), lds, s>>>(gout, (const bf16*)y, gamma, rs,
                                                     rows_per_scale, (bf16*)gy, partials, rows, d, rowmap)));
  } else {
    DENSE_NV_SWITCH(dense_nv(d), (scale_residual_bwd_kernel<float, NV><<<dim3(blocks), dim3(kDenseWaves * 64), lds, s>>>(gout, (const float*)y, gamma, rs,
                                                     rows_per_scale, (float*)gy, partials, rows, d, rowmap)));
  }
  return launch_status();
}

int octic_dense_prep_batch_blocks(int N, int K) { return ((N + 63) / 64) * ((K + 63) / 64); }

int octic_dense_prep_batch(const octic_dense_prep_item* items_dev, int n_items, int total_blocks, int src_dtype, void* stream) {
  if (!items_dev) return OCTIC_ENULL;
  if (n_items <= 0 || total_blocks <= 0) return OCTIC_ESHAPE;
  if (src_dtype == OCTIC_F32) dense_prep_batch_kernel<float><<<total_blocks, 256, 0, (hipStream_t)stream>>>(items_dev, n_items);
  else if (src_dtype == OCTIC_BF16) dense_prep_batch_kernel<bf16><<<total_blocks, 256, 0, (hipStream_t)stream>>>(items_dev, n_items);
  else return OCTIC_EDTYPE;
  return launch_status();
}

}  // extern "C"
