// Row kernels of the DINOv2 objective on the prototype axis (K = 65 536 in the recipe): the teacher's centred softmax
// (dinov2/loss/dino_clstoken_loss.py:43-51, ibot_patch_loss.py:63-77) and the soft-target cross entropy
// -sum_k t_k log_softmax(s / temp)_k with its gradient (dino_clstoken_loss.py:78-92, ibot_patch_loss.py:26-34).
// HBM-bound by construction: the eager composition walks the [rows, K] logits nine times in f32 (subtract, divide, softmax,
// divide, log_softmax, multiply, sum, and the two backward passes); here the teacher's probabilities cost one read (+ one
// re-read that hits L2 / MALL) and one write, the cross entropy ONE read of the student logits and the probabilities, its
// gradient one read of each and one write.  One workgroup per row, 16-byte loads, online max / sum (no second pass for the
// log-sum-exp), f32 arithmetic throughout (expf / logf of the device library, not the fast intrinsics); the student logits are taken in their storage dtype (bf16 under autocast: the
// reference divides by the temperature in bf16 first - this path skips that rounding).
#include "octic_common.hpp"

namespace octic {

constexpr int SL_THREADS = 512;

struct MaxSum { float m, e; };
__device__ __forceinline__ void ms_add8(MaxSum& a, const float v[8]) {
  float mx = v[0];
#pragma unroll
  for (int i = 1; i < 8; ++i) mx = fmaxf(mx, v[i]);
  if (mx > a.m) { a.e *= expf(a.m - mx); a.m = mx; }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += expf(v[i] - a.m);
  a.e += s;
}
// workgroup-wide combine of per-thread (max, sum of exp relative to it); every thread gets the result
__device__ __forceinline__ MaxSum ms_block(MaxSum a, float* red) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = SL_THREADS / 64;
  float m = a.m;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  const float e = wave_total(a.e * (a.m == -INFINITY ? 0.f : expf(a.m - m)));
  if (lane == 0) { red[2 * w] = m; red[2 * w + 1] = e; }
  __syncthreads();
  float M = red[0];
  for (int i = 1; i < nw; ++i) M = fmaxf(M, red[2 * i]);
  float E = 0.f;
  for (int i = 0; i < nw; ++i) E += red[2 * i + 1] * (red[2 * i] == -INFINITY ? 0.f : expf(red[2 * i] - M));
  __syncthreads();
  return MaxSum{M, E};
}
__device__ __forceinline__ float sum_block(float v, float* red) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = SL_THREADS / 64;
  v = wave_total(v);
  if (lane == 0) red[w] = v;
  __syncthreads();
  float s = 0.f;
  for (int i = 0; i < nw; ++i) s += red[i];
  __syncthreads();
  return s;
}

// out[r, :] = softmax((t[r, :] - center) * inv_temp)
template <typename TS>
__global__ __launch_bounds__(SL_THREADS) void softmax_center_kernel(const TS* __restrict__ t, int64_t ldt,
                                                                     const float* __restrict__ center, float inv_temp,
                                                                     float* __restrict__ out, int K) {
  __shared__ float red[2 * SL_THREADS / 64];
  const int64_t r = blockIdx.x;
  const TS* tr = t + r * ldt;
  float* o = out + r * (int64_t)K;
  MaxSum a{-INFINITY, 0.f};
  for (int k = threadIdx.x * 8; k < K; k += SL_THREADS * 8) {
    float v[8], c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    load8<TS>(tr + k, v);
    if (center) load8<float>(center + k, c);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (v[i] - c[i]) * inv_temp;
    ms_add8(a, v);
  }
  const MaxSum A = ms_block(a, red);
  const float inv = 1.0f / A.e;
  for (int k = threadIdx.x * 8; k < K; k += SL_THREADS * 8) {
    float v[8], c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    load8<TS>(tr + k, v);
    if (center) load8<float>(center + k, c);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = expf((v[i] - c[i]) * inv_temp - A.m) * inv;
    store8<float>(o + k, v);
  }
}

// loss[r] = tsum lse - sum_k t_k z_k,  z = s[r, :] * inv_temp,  lse = log sum exp z,  t = tprob[r % t_rows, :]
template <typename TS>
__global__ __launch_bounds__(SL_THREADS) void soft_ce_fwd_kernel(const TS* __restrict__ s, int64_t lds_,
                                                                  const float* __restrict__ tprob, int64_t t_rows,
                                                                  float inv_temp, float* __restrict__ loss,
                                                                  float* __restrict__ lse, float* __restrict__ tsum, int K) {
  __shared__ float red[2 * SL_THREADS / 64];
  const int64_t r = blockIdx.x;
  const TS* sr = s + r * lds_;
  const float* tr = tprob + (r % t_rows) * (int64_t)K;
  MaxSum a{-INFINITY, 0.f};
  float dot = 0.f, ts = 0.f;
  for (int k = threadIdx.x * 8; k < K; k += SL_THREADS * 8) {
    float v[8], t[8];
    load8<TS>(sr + k, v);
    load8<float>(tr + k, t);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      v[i] *= inv_temp;
      dot = __builtin_fmaf(t[i], v[i], dot);
      ts += t[i];
    }
    ms_add8(a, v);
  }
  const MaxSum A = ms_block(a, red);
  dot = sum_block(dot, red);
  ts = sum_block(ts, red);
  if (threadIdx.x == 0) {
    const float l = A.m + logf(A.e);
    lse[r] = l;
    tsum[r] = ts;
    loss[r] = ts * l - dot;
  }
}

// ds[r, k] = g[r] inv_temp (exp(z_k - lse[r]) tsum[r] - t_k)
template <typename TS>
__global__ __launch_bounds__(SL_THREADS) void soft_ce_bwd_kernel(const TS* __restrict__ s, int64_t lds_,
                                                                  const float* __restrict__ tprob, int64_t t_rows,
                                                                  float inv_temp, const float* __restrict__ g,
                                                                  const float* __restrict__ lse,
                                                                  const float* __restrict__ tsum, TS* __restrict__ ds,
                                                                  int64_t ldd, int K) {
  const int64_t r = blockIdx.x;
  const TS* sr = s + r * lds_;
  const float* tr = tprob + (r % t_rows) * (int64_t)K;
  TS* dr = ds + r * ldd;
  const float gr = g[r] * inv_temp, l = lse[r], ts = tsum[r];
  for (int k = threadIdx.x * 8; k < K; k += SL_THREADS * 8) {
    float v[8], t[8];
    load8<TS>(sr + k, v);
    load8<float>(tr + k, t);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = gr * (expf(v[i] * inv_temp - l) * ts - t[i]);
    store8<TS>(dr + k, v);
  }
}

static int sl_check(int64_t rows, int K, int64_t ld) {
  if (rows < 0 || K <= 0) return OCTIC_ESHAPE;
  if (K % 8 || ld % 8 || ld < K) return OCTIC_ESHAPE;
  if (rows > 0x7FFFFFFF) return OCTIC_ESHAPE;
  return OCTIC_OK;
}

}  // namespace octic

using namespace octic;

extern "C" {

int octic_softmax_center(const void* t, int t_dtype, int64_t ldt, const float* center, float inv_temp, float* out,
                         int64_t rows, int K, void* stream) {
  if (rows == 0) return OCTIC_OK;
  if (!t || !out) return OCTIC_ENULL;
  if (int e = sl_check(rows, K, ldt)) return e;
  hipStream_t st = (hipStream_t)stream;
  if (t_dtype == OCTIC_BF16)
    softmax_center_kernel<bf16><<<dim3((unsigned)rows), dim3(SL_THREADS), 0, st>>>((const bf16*)t, ldt, center, inv_temp, out, K);
  else if (t_dtype == OCTIC_F32)
    softmax_center_kernel<float><<<dim3((unsigned)rows), dim3(SL_THREADS), 0, st>>>((const float*)t, ldt, center, inv_temp, out, K);
  else
    return OCTIC_EDTYPE;
  return launch_status();
}

int octic_soft_ce_fwd(const void* s, int s_dtype, int64_t lds, const float* tprob, int64_t t_rows, float inv_temp,
                      float* loss, float* lse, float* tsum, int64_t rows, int K, void* stream) {
  if (rows == 0) return OCTIC_OK;
  if (!s || !tprob || !loss || !lse || !tsum) return OCTIC_ENULL;
  if (int e = sl_check(rows, K, lds)) return e;
  if (t_rows <= 0) return OCTIC_ESHAPE;
  hipStream_t st = (hipStream_t)stream;
  if (s_dtype == OCTIC_BF16)
    soft_ce_fwd_kernel<bf16><<<dim3((unsigned)rows), dim3(SL_THREADS), 0, st>>>((const bf16*)s, lds, tprob, t_rows, inv_temp, loss, lse, tsum, K);
  else if (s_dtype == OCTIC_F32)
    soft_ce_fwd_kernel<float><<<dim3((unsigned)rows), dim3(SL_THREADS), 0, st>>>((const float*)s, lds, tprob, t_rows, inv_temp, loss, lse, tsum, K);
  else
    return OCTIC_EDTYPE;
  return launch_status();
}

int octic_soft_ce_bwd(const void* s, int s_dtype, int64_t lds, const float* tprob, int64_t t_rows, float inv_temp,
                      const float* g, const float* lse, const float* tsum, void* ds, int64_t ldd, int64_t rows, int K,
                      void* stream) {
  if (rows == 0) return OCTIC_OK;
  if (!s || !tprob || !g || !lse || !tsum || !ds) return OCTIC_ENULL;
  if (int e = sl_check(rows, K, lds)) return e;
  if (int e = sl_check(rows, K, ldd)) return e;
  if (t_rows <= 0) return OCTIC_ESHAPE;
  hipStream_t st = (hipStream_t)stream;
  if (s_dtype == OCTIC_BF16)
    soft_ce_bwd_kernel<bf16><<<dim3((unsigned)rows), dim3(SL_THREADS), 0, st>>>((const bf16*)s, lds, tprob, t_rows, inv_temp, g, lse, tsum, (bf16*)ds, ldd, K);
  else if (s_dtype == OCTIC_F32)
    soft_ce_bwd_kernel<float><<<dim3((unsigned)rows), dim3(SL_THREADS), 0, st>>>((const float*)s, lds, tprob, t_rows, inv_temp, g, lse, tsum, (float*)ds, ldd, K);
  else
    return OCTIC_EDTYPE;
  return launch_status();
}

}  // extern "C"
