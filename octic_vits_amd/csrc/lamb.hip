// Fused multi-tensor LAMB + EMA for gfx950 (the reference recipe uses apex FusedLAMB + timm ModelEma:
// experiments/train_deit.py:42, deit/main.py:344-351, deit/engine.py:77-84).
//
// All trainable tensors are described once by per-tensor pointer tables and a static chunk list
// (tensor id, offset, length <= 64Ki elements).  A step is five launches, each one streaming pass or a tiny
// reduction, every element moved with 16-byte accesses:
//   1. gradsq   : per-chunk sum g^2                                   (read g)
//   2. scalars  : global grad norm -> clip factor                      (tiny)
//   3. stage1   : m,v update, u = mhat/(sqrt(vhat)+eps) + wd*p stored over g, per-chunk |p|^2, |u|^2
//   4. ratios   : per-tensor trust ratio |p|/|u|                       (tiny, fixed-order => reproducible)
//   5. stage2   : p -= lr*ratio*u ; ema += (1-decay)(p-ema)
// Traffic 4+28+20 = 52 B/param vs ~100+ for the op-by-op foreach formulation.  Bound: HBM.
#include "octic_common.hpp"

namespace octic {

struct LambTables {
  float* const* p;      // [ntensors]
  float* const* g;
  float* const* m;
  float* const* v;
  float* const* ema;    // may be null
  bf16* const* shadow;  // may be null; entries may be null: bf16 copy of the updated parameter
  const float* wd;      // [ntensors] weight decay per tensor
  const int* chunk_tensor;   // [nchunks]
  const int64_t* chunk_off;  // [nchunks]
  const int* chunk_len;      // [nchunks]
  const int* tensor_chunk_begin;  // [ntensors+1]
};

__device__ inline float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) red[wid] = v;
  __syncthreads();
  const float r = red[0] + red[1] + red[2] + red[3];
  __syncthreads();
  return r;
}

__global__ __launch_bounds__(256) void lamb_gradsq_kernel(LambTables t, float* part_g2) {
  __shared__ float red[4];
  const int ch = blockIdx.x;
  const int ti = t.chunk_tensor[ch];
  const float* g = t.g[ti] + t.chunk_off[ch];
  const int n = t.chunk_len[ch];
  float s = 0.f;
  const int n4 = ((((uintptr_t)g) & 15) == 0) ? (n >> 2) : 0;
  for (int i = threadIdx.x; i < n4; i += 256) {
    const f32x4 a = ((const f32x4*)g)[i];
    s += a[0] * a[0] + a[1] * a[1] + a[2] * a[2] + a[3] * a[3];
  }
  for (int i = n4 * 4 + threadIdx.x; i < n; i += 256) s += g[i] * g[i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) part_g2[ch] = s;
}

// scal[0] = clip multiplier (1 / max(1, gnorm / max_norm)), scal[1] = gnorm, scal[2] = 1 when this step is skipped
// (non-finite gradient norm: parameters, moments, EMA and bf16 copies are left untouched; the reference exits before
// optimizer.step() on a non-finite loss, deit/engine.py:67-71), scal[3] = number of applied steps (the device-side
// step counter used when the host passes step = 0, so a captured hipGraph replays with the right bias correction),
// scal[4] = 1 - beta1^t, scal[5] = 1 / sqrt(1 - beta2^t), scal[6] = number of skipped steps so far.
__global__ __launch_bounds__(256) void lamb_scalars_kernel(const float* part_g2, int nchunks, float max_norm, float b1,
                                                           float b2, int host_step, float* scal) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < nchunks; i += 256) s += part_g2[i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) {
    const float gn = sqrtf(s);
    const bool ok = isfinite(gn);
    scal[1] = gn;
    scal[0] = (max_norm > 0.f && gn > max_norm) ? max_norm / gn : 1.0f;
    scal[2] = ok ? 0.f : 1.f;
    if (!ok) scal[6] += 1.f;
    float t = host_step > 0 ? (float)host_step : scal[3] + (ok ? 1.f : 0.f);
    if (host_step > 0 || ok) scal[3] = t;
    t = t < 1.f ? 1.f : t;
    scal[4] = 1.f - powf(b1, t);
    scal[5] = 1.0f / sqrtf(1.f - powf(b2, t));
  }
}

__global__ __launch_bounds__(256) void lamb_stage1_kernel(LambTables t, const float* scal, float b1, float b2, float eps,
                                                          float* part_p2, float* part_u2) {
  __shared__ float red[4];
  const int ch = blockIdx.x;
  if (scal[2] != 0.f) return;                  // skipped step (uniform over the grid)
  const float bc1 = scal[4], rsqrt_bc2_inv = scal[5];
  const int ti = t.chunk_tensor[ch];
  const int64_t off = t.chunk_off[ch];
  float* p = t.p[ti] + off;
  float* g = t.g[ti] + off;
  float* m = t.m[ti] + off;
  float* v = t.v[ti] + off;
  const int n = t.chunk_len[ch];
  const float wd = t.wd[ti], clip = scal[0];
  float sp = 0.f, su = 0.f;
  auto one = [&](float pv, float gv, float& mv, float& vv) {
    gv *= clip;
    mv = b1 * mv + (1.f - b1) * gv;
    vv = b2 * vv + (1.f - b2) * gv * gv;
    const float denom = sqrtf(vv) * rsqrt_bc2_inv + eps;   // sqrt(v)/sqrt(bc2) + eps
    float u = (mv / bc1) / denom;
    if (wd != 0.f) u += wd * pv;
    sp += pv * pv;
    su += u * u;
    return u;
  };
  const bool al = ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) == 0;
  const int n4 = al ? (n >> 2) : 0;
  for (int i = threadIdx.x; i < n4; i += 256) {
    const f32x4 pv = ((const f32x4*)p)[i];
    f32x4 gv = ((const f32x4*)g)[i], mv = ((const f32x4*)m)[i], vv = ((const f32x4*)v)[i];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float mj = mv[j], vj = vv[j];
      gv[j] = one(pv[j], gv[j], mj, vj);
      mv[j] = mj;
      vv[j] = vj;
    }
    ((f32x4*)m)[i] = mv;
    ((f32x4*)v)[i] = vv;
    ((f32x4*)g)[i] = gv;   // the update overwrites the gradient
  }
  for (int i = n4 * 4 + threadIdx.x; i < n; i += 256) {
    float mv = m[i], vv = v[i];
    g[i] = one(p[i], g[i], mv, vv);
    m[i] = mv;
    v[i] = vv;
  }
  sp = block_sum(sp, red);
  su = block_sum(su, red);
  if (threadIdx.x == 0) {
    part_p2[ch] = sp;
    part_u2[ch] = su;
  }
}

__global__ __launch_bounds__(256) void lamb_ratio_kernel(LambTables t, int ntensors, const float* part_p2,
                                                         const float* part_u2, float* ratio, int adam) {
  const int ti = blockIdx.x * 256 + threadIdx.x;
  if (ti >= ntensors) return;
  if (adam) {                                   // AdamW: no layer-wise trust ratio
    ratio[ti] = 1.0f;
    return;
  }
  float sp = 0.f, su = 0.f;
  for (int c = t.tensor_chunk_begin[ti]; c < t.tensor_chunk_begin[ti + 1]; ++c) {
    sp += part_p2[c];
    su += part_u2[c];
  }
  const float wn = sqrtf(sp), un = sqrtf(su);
  ratio[ti] = (t.wd[ti] != 0.f && wn > 0.f && un > 0.f) ? wn / un : 1.0f;
}

__global__ __launch_bounds__(256) void lamb_stage2_kernel(LambTables t, const float* scal, const float* ratio, float lr,
                                                          float ema_w) {
  const int ch = blockIdx.x;
  if (scal[2] != 0.f) return;                  // skipped step: weights, EMA and bf16 copies stay as they are
  const int ti = t.chunk_tensor[ch];
  const int64_t off = t.chunk_off[ch];
  float* p = t.p[ti] + off;
  const float* u = t.g[ti] + off;
  float* e = t.ema ? t.ema[ti] + off : nullptr;
  bf16* sh = (t.shadow && t.shadow[ti]) ? t.shadow[ti] + off : nullptr;
  const int n = t.chunk_len[ch];
  const float step = lr * ratio[ti];
  const bool al = ((((uintptr_t)p) | ((uintptr_t)u) | ((uintptr_t)e)) & 15) == 0 && (((uintptr_t)sh) & 7) == 0;
  const int n4 = al ? (n >> 2) : 0;
  for (int i = threadIdx.x; i < n4; i += 256) {
    f32x4 pv = ((const f32x4*)p)[i];
    const f32x4 uv = ((const f32x4*)u)[i];
    pv -= step * uv;
    ((f32x4*)p)[i] = pv;
    if (sh) ((bf16x4*)sh)[i] = bf16x4{(bf16)pv[0], (bf16)pv[1], (bf16)pv[2], (bf16)pv[3]};
    if (e) {
      f32x4 ev = ((const f32x4*)e)[i];
      ev += ema_w * (pv - ev);
      ((f32x4*)e)[i] = ev;
    }
  }
  for (int i = n4 * 4 + threadIdx.x; i < n; i += 256) {
    const float pv = p[i] - step * u[i];
    p[i] = pv;
    if (sh) sh[i] = (bf16)pv;
    if (e) e[i] += ema_w * (pv - e[i]);
  }
}

}  // namespace octic

using namespace octic;

extern "C" {

static int lamb_impl(void* const* p, void* const* g, void* const* m, void* const* v, void* const* ema, const float* wd,
                     const int* chunk_tensor, const int64_t* chunk_off, const int* chunk_len,
                     const int* tensor_chunk_begin, int ntensors, int nchunks, float* workspace, float lr, float beta1,
                     float beta2, float eps, float max_grad_norm, int step, float ema_decay, void* const* bf16_shadow,
                     void* stream, int adam) {
  if (!p || !g || !m || !v || !wd || !chunk_tensor || !chunk_off || !chunk_len || !tensor_chunk_begin || !workspace)
    return OCTIC_ENULL;
  if (ntensors <= 0 || nchunks <= 0 || step < 0) return OCTIC_ESHAPE;
  LambTables t;
  t.p = (float* const*)p; t.g = (float* const*)g; t.m = (float* const*)m; t.v = (float* const*)v;
  t.ema = (float* const*)ema; t.shadow = (bf16* const*)bf16_shadow; t.wd = wd;
  t.chunk_tensor = chunk_tensor; t.chunk_off = chunk_off; t.chunk_len = chunk_len;
  t.tensor_chunk_begin = tensor_chunk_begin;
  // workspace layout: [8] scalars (see lamb_scalars_kernel) | [nchunks] g2 | [nchunks] p2 | [nchunks] u2 | [ntensors] ratio
  float* scal = workspace;
  float* g2 = scal + 8;
  float* p2 = g2 + nchunks;
  float* u2 = p2 + nchunks;
  float* ratio = u2 + nchunks;
  hipStream_t s = (hipStream_t)stream;
  lamb_gradsq_kernel<<<nchunks, 256, 0, s>>>(t, g2);
  lamb_scalars_kernel<<<1, 256, 0, s>>>(g2, nchunks, max_grad_norm, beta1, beta2, step, scal);
  lamb_stage1_kernel<<<nchunks, 256, 0, s>>>(t, scal, beta1, beta2, eps, p2, u2);
  lamb_ratio_kernel<<<(ntensors + 255) / 256, 256, 0, s>>>(t, ntensors, p2, u2, ratio, adam);
  lamb_stage2_kernel<<<nchunks, 256, 0, s>>>(t, scal, ratio, lr, ema ? 1.0f - ema_decay : 0.f);
  return launch_status();
}

int octic_lamb_step(void* const* p, void* const* g, void* const* m, void* const* v, void* const* ema, const float* wd,
                    const int* chunk_tensor, const int64_t* chunk_off, const int* chunk_len,
                    const int* tensor_chunk_begin, int ntensors, int nchunks, float* workspace, float lr, float beta1,
                    float beta2, float eps, float max_grad_norm, int step, float ema_decay, void* const* bf16_shadow,
                    void* stream) {
  return lamb_impl(p, g, m, v, ema, wd, chunk_tensor, chunk_off, chunk_len, tensor_chunk_begin, ntensors, nchunks, workspace, lr,
                   beta1, beta2, eps, max_grad_norm, step, ema_decay, bf16_shadow, stream, 0);
}

int octic_adamw_step(void* const* p, void* const* g, void* const* m, void* const* v, void* const* ema, const float* wd,
                     const int* chunk_tensor, const int64_t* chunk_off, const int* chunk_len,
                     const int* tensor_chunk_begin, int ntensors, int nchunks, float* workspace, float lr, float beta1,
                     float beta2, float eps, float max_grad_norm, int step, float ema_decay, void* const* bf16_shadow,
                     void* stream) {
  return lamb_impl(p, g, m, v, ema, wd, chunk_tensor, chunk_off, chunk_len, tensor_chunk_begin, ntensors, nchunks, workspace, lr,
                   beta1, beta2, eps, max_grad_norm, step, ema_decay, bf16_shadow, stream, 1);
}

int64_t octic_lamb_workspace_floats(int ntensors, int nchunks) { return 8 + 3 * (int64_t)nchunks + ntensors; }

}  // extern "C"
