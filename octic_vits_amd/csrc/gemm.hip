// Irrep-blocked GEMM on MFMA for gfx950 (LinearD8 forward, input gradient, lift patch-embed).
//
// One launch covers the five sub-problems of a LinearD8 (A1,A2,B1,B2: [M,c]x[c,c'] ; E: [2M,2c]x[2c,2c'],
// the two E rows of a token are two GEMM rows sharing W_E).  Tile = 128 rows x (32*NT) outputs,
// 256 threads = 4 waves (2 along N x 2 along M), each wave NT x 4 MFMA tiles of 16x16.
// Operands are swapped on purpose: W is the MFMA "A" operand and X^T the "B" operand, so every lane
// ends up with 4 CONSECUTIVE output channels of ONE token -> 8/16-byte stores into the token row and a
// fused epilogue (bias, layer-scale, drop-path mask, residual) that reads/writes whole vectors.
// LDS tiles are [rows][128 B] (64 bf16 / 32 f32 of K) with the 16-byte chunk index XOR-swizzled by
// (row & 7): conflict-free ds_read_b128 for the 16-lane MFMA operand pattern.  Global->LDS goes through
// registers (16 B per lane, 128 B contiguous per 8 lanes) with the next K tile's loads issued before
// the current tile's MFMAs (software pipeline, one barrier per K tile).  K is short (160/320 at
// ViT-H), so the kernel is bound by HBM/epilogue traffic, not MFMA issue; see DESIGN.md §kernels.
// bf16: v_mfma_f32_16x16x32_bf16.  f32: v_mfma_f32_16x16x4_f32 (exact f32 fma chain) for the
// reference's fp32 tolerances.
#include "octic_common.hpp"

namespace octic {

struct GemmGroup {
  const char* a;      // A rows
  int64_t a_ld;       // row stride (elements)
  const char* w;      // [N, K] row-major
  char* y;
  int64_t y_ld;
  const char* resid;  // optional (same addressing as y unless lift)
  int64_t r_ld;
  const float* bias;  // optional [N]
  const float* cs;    // optional [N]
  int64_t rows;       // M (or 2M for the E pair group)
  int K, N;
  int pair;           // 1: row mm -> token mm>>1, half mm&1 (adjacent halves of width K / N)
  int n_tiles, m_tiles;
  int tile_begin;     // first linear tile id of this group
};

struct GemmArgs {
  GemmGroup g[5];
  int ngroups;
  int total_tiles;
  const float* rs;   // optional per-sample scale
  int64_t rps;       // token rows per sample for rs
  // lift mode: output row = (row / lift_np) * (lift_np + lift_tok0) + lift_tok0 + row % lift_np ; resid row = row % lift_np
  int64_t lift_np;
  int lift_tok0;
};

template <typename T> struct Elem;
template <> struct Elem<float> { static constexpr int EPC = 4; typedef f32x4 frag; };
template <> struct Elem<bf16> { static constexpr int EPC = 8; typedef bf16x8 frag; };

template <typename TOUT>
__device__ inline void store_out4(TOUT* p, f32x4 v);
template <>
__device__ inline void store_out4<float>(float* p, f32x4 v) { *(f32x4*)p = v; }
template <>
__device__ inline void store_out4<bf16>(bf16* p, f32x4 v) {
  bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
  *(bf16x4*)p = o;
}
template <typename TOUT>
__device__ inline f32x4 load_out4(const TOUT* p);
template <>
__device__ inline f32x4 load_out4<float>(const float* p) { return *(const f32x4*)p; }
template <>
__device__ inline f32x4 load_out4<bf16>(const bf16* p) {
  bf16x4 a = *(const bf16x4*)p;
  return f32x4{(float)a[0], (float)a[1], (float)a[2], (float)a[3]};
}

__device__ inline f32x4 mfma_step(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ inline f32x4 mfma_step(f32x4 a, f32x4 b, f32x4 c) {
  // 16 k values per (lane-group, chunk): element s of lane-group kg is k = 4*kg + s.  A and B use the
  // same assignment, and a sum over k does not care about the order.
#pragma unroll
  for (int s = 0; s < 4; ++s) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[s], c, 0, 0, 0);
  return c;
}

constexpr int kBM = 128;

template <typename TIN, typename TOUT, int NT>
__global__ __launch_bounds__(256) void linear_d8_kernel(GemmArgs args) {
  constexpr int EPC = Elem<TIN>::EPC;   // elements per 16-byte chunk
  constexpr int BKE = 8 * EPC;          // K elements per tile (128 B rows)
  constexpr int BN = 32 * NT;
  constexpr int MT = 4;
  typedef typename Elem<TIN>::frag frag;
  extern __shared__ __attribute__((aligned(16))) char lds[];  // 2 stages x (BM + BN) rows x 128 B
  constexpr int STAGE = (kBM + BN) * 128;

  // ---- XCD-aware, bijective block remap: blocks that share an XCD (bid % 8) get a contiguous range
  // of tiles, so the n-tiles that re-read one token panel hit the same L2.
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
  const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);

  int gi = 0;
#pragma unroll
  for (int i = 1; i < 5; ++i)
    if (i < args.ngroups && tile >= args.g[i].tile_begin) gi = i;
  const GemmGroup& G = args.g[gi];
  const int lt = tile - G.tile_begin;
  const int mt = lt / G.n_tiles, nt = lt - mt * G.n_tiles;
  const int64_t m0 = (int64_t)mt * kBM;
  const int n0 = nt * BN;
  const int K = G.K, N = G.N;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wn = wid & 1, wm = wid >> 1;

  // ---- staging assignment: thread owns chunk column kc of rows (tid>>3) + 32*i
  const int kc = tid & 7, r_in = tid >> 3;
  const TIN* xrow[4];
  bool xok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t mm = m0 + r_in + 32 * i;
    xok[i] = mm < G.rows;
    const int64_t off = G.pair ? (mm >> 1) * G.a_ld + (mm & 1) * (int64_t)K : mm * G.a_ld;
    xrow[i] = (const TIN*)G.a + (xok[i] ? off : 0);
  }
  const TIN* wrow[NT];
  bool wok[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    const int n = n0 + r_in + 32 * i;
    wok[i] = n < N;
    wrow[i] = (const TIN*)G.w + (wok[i] ? (int64_t)n * K : 0);
  }
  u32x4 rx[4], rw[NT];
  auto gload = [&](int k0) {
    const int k = k0 + kc * EPC;
    const bool kok = k < K;
#pragma unroll
    for (int i = 0; i < 4; ++i) rx[i] = (xok[i] && kok) ? *(const u32x4*)(xrow[i] + k) : u32x4{0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < NT; ++i) rw[i] = (wok[i] && kok) ? *(const u32x4*)(wrow[i] + k) : u32x4{0, 0, 0, 0};
  };
  auto lstore = [&](int stage) {
    char* xs = lds + stage * STAGE;
    char* ws = xs + kBM * 128;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = r_in + 32 * i;
      *(u32x4*)(xs + row * 128 + ((kc ^ (row & 7)) << 4)) = rx[i];
    }
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int row = r_in + 32 * i;
      *(u32x4*)(ws + row * 128 + ((kc ^ (row & 7)) << 4)) = rw[i];
    }
  };

  f32x4 acc[NT][MT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0, 0, 0, 0};

  const int nkt = (K + BKE - 1) / BKE;
  gload(0);
  lstore(0);
  __syncthreads();
  const int fr = lane & 15, kg = lane >> 4;
  for (int kt = 0; kt < nkt; ++kt) {
    if (kt + 1 < nkt) gload((kt + 1) * BKE);
    const char* xs = lds + (kt & 1) * STAGE;
    const char* ws = xs + kBM * 128;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if (kt * BKE + ks * 4 * EPC < K) {  // wave-uniform: skip an all-zero half tile
        const int ch = ks * 4 + kg;
        frag af[NT], bfr[MT];
#pragma unroll
        for (int i = 0; i < NT; ++i) {
          const int row = wn * (NT * 16) + i * 16 + fr;
          af[i] = *(const frag*)(ws + row * 128 + ((ch ^ (row & 7)) << 4));
        }
#pragma unroll
        for (int j = 0; j < MT; ++j) {
          const int row = wm * 64 + j * 16 + fr;
          bfr[j] = *(const frag*)(xs + row * 128 + ((ch ^ (row & 7)) << 4));
        }
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
          for (int j = 0; j < MT; ++j) acc[i][j] = mfma_step(af[i], bfr[j], acc[i][j]);
      }
    }
    if (kt + 1 < nkt) lstore((kt + 1) & 1);
    __syncthreads();
  }

  // ---- epilogue: lane holds outputs n..n+3 of token row mm for each (i,j) tile
#pragma unroll
  for (int j = 0; j < MT; ++j) {
    const int64_t mm = m0 + wm * 64 + j * 16 + fr;
    if (mm >= G.rows) continue;
    const int64_t token = G.pair ? (mm >> 1) : mm;
    int64_t yoff, roff;
    if (args.lift_np > 0) {
      const int64_t b = mm / args.lift_np, p = mm - b * args.lift_np;
      yoff = (b * (args.lift_np + args.lift_tok0) + args.lift_tok0 + p) * G.y_ld;
      roff = p * G.r_ld;
    } else {
      yoff = G.pair ? (mm >> 1) * G.y_ld + (mm & 1) * (int64_t)N : mm * G.y_ld;
      roff = G.pair ? (mm >> 1) * G.r_ld + (mm & 1) * (int64_t)N : mm * G.r_ld;
    }
    const float rsv = args.rs ? args.rs[token / args.rps] : 1.0f;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int n = n0 + wn * (NT * 16) + i * 16 + kg * 4;
      if (n >= N) continue;
      f32x4 v = acc[i][j];
      if (G.bias) {
        const f32x4 b = *(const f32x4*)(G.bias + n);
        v += b;
      }
      if (G.cs) {
        const f32x4 s = *(const f32x4*)(G.cs + n);
        v *= s;
      }
      if (args.rs) v *= rsv;
      if (G.resid) v += load_out4<TOUT>((const TOUT*)G.resid + roff + n);
      store_out4<TOUT>((TOUT*)G.y + yoff + n, v);
    }
  }
}

inline int pick_nt(const GemmArgs& a) {
  // minimise padded work; prefer the wider tile on ties (fewer re-reads of the token panel)
  int best = 2;
  double best_cost = 1e30;
  for (int nt = 2; nt <= 5; ++nt) {
    const int bn = 32 * nt;
    double cost = 0;
    for (int i = 0; i < a.ngroups; ++i) {
      const double tiles = (a.g[i].N + bn - 1) / bn;
      cost += tiles * bn * (double)a.g[i].K * (double)a.g[i].rows;
    }
    if (cost <= best_cost * 1.0001) {
      best_cost = cost < best_cost ? cost : best_cost;
      best = nt;
    }
  }
  return best;
}

template <typename TIN, typename TOUT>
int launch_gemm(GemmArgs& a, hipStream_t s) {
  const int nt = pick_nt(a);
  const int bn = 32 * nt;
  int t = 0;
  for (int i = 0; i < a.ngroups; ++i) {
    a.g[i].n_tiles = (a.g[i].N + bn - 1) / bn;
    a.g[i].m_tiles = (int)((a.g[i].rows + kBM - 1) / kBM);
    a.g[i].tile_begin = t;
    t += a.g[i].n_tiles * a.g[i].m_tiles;
  }
  a.total_tiles = t;
  const size_t smem = (size_t)2 * (kBM + bn) * 128;
  // gfx950 has 160 KiB of LDS per CU; anything above the 64 KiB default must be opted into once.
  static bool attr_done = false;
  if (!attr_done) {
    hipFuncSetAttribute((const void*)linear_d8_kernel<TIN, TOUT, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (kBM + 64) * 128);
    hipFuncSetAttribute((const void*)linear_d8_kernel<TIN, TOUT, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (kBM + 96) * 128);
    hipFuncSetAttribute((const void*)linear_d8_kernel<TIN, TOUT, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (kBM + 128) * 128);
    hipFuncSetAttribute((const void*)linear_d8_kernel<TIN, TOUT, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (kBM + 160) * 128);
    (void)hipGetLastError();
    attr_done = true;
  }
  switch (nt) {
    case 2: linear_d8_kernel<TIN, TOUT, 2><<<t, 256, smem, s>>>(a); break;
    case 3: linear_d8_kernel<TIN, TOUT, 3><<<t, 256, smem, s>>>(a); break;
    case 4: linear_d8_kernel<TIN, TOUT, 4><<<t, 256, smem, s>>>(a); break;
    default: linear_d8_kernel<TIN, TOUT, 5><<<t, 256, smem, s>>>(a); break;
  }
  return launch_status();
}

inline int dispatch_gemm(GemmArgs& a, int dtype, int out_dtype, hipStream_t s) {
  if (dtype == OCTIC_F32 && out_dtype == OCTIC_F32) return launch_gemm<float, float>(a, s);
  if (dtype == OCTIC_BF16 && out_dtype == OCTIC_BF16) return launch_gemm<bf16, bf16>(a, s);
  if (dtype == OCTIC_BF16 && out_dtype == OCTIC_F32) return launch_gemm<bf16, float>(a, s);
  return OCTIC_EDTYPE;
}

}  // namespace octic

using namespace octic;

extern "C" {

int octic_linear_d8_fwd(const octic_view* x, const void* const w[5], const float* bias, const octic_view* y,
                        const octic_view* resid, const float* rs, int64_t rows_per_sample, const float* const cs[5],
                        int64_t M, int cin, int cout, int dtype, int out_dtype, void* stream) {
  int e;
  if ((e = check_c_dt(cin, dtype)) || (e = check_c_dt(cout, dtype)) || (e = check_c_dt(cout, out_dtype))) return e;
  if ((e = check_view(x, cin, dtype)) || (e = check_view(y, cout, out_dtype))) return e;
  if (resid && (e = check_view(resid, cout, out_dtype))) return e;
  if (!w) return OCTIC_ENULL;
  for (int i = 0; i < 5; ++i)
    if (!w[i] || (((uintptr_t)w[i]) & 15)) return w[i] ? OCTIC_EALIGN : OCTIC_ENULL;
  if (M <= 0 || (rs && rows_per_sample <= 0)) return OCTIC_ESHAPE;
  GemmArgs a = {};
  a.ngroups = 5;
  a.rs = rs;
  a.rps = rs ? rows_per_sample : 1;
  a.lift_np = 0;
  a.lift_tok0 = 0;
  // group order: E first (its tiles carry 2x the K work), then the four one-dimensional irreps
  for (int gidx = 0; gidx < 5; ++gidx) {
    const int irrep = gidx == 0 ? 4 : gidx - 1;
    GemmGroup& g = a.g[gidx];
    const bool isE = irrep == 4;
    g.a = (const char*)x->ptr[irrep];
    g.a_ld = x->ld[irrep];
    g.w = (const char*)w[irrep];
    g.y = (char*)y->ptr[irrep];
    g.y_ld = y->ld[irrep];
    g.resid = resid ? (const char*)resid->ptr[irrep] : nullptr;
    g.r_ld = resid ? resid->ld[irrep] : 0;
    g.bias = (irrep == 0) ? bias : nullptr;
    g.cs = cs ? cs[irrep] : nullptr;
    g.rows = isE ? 2 * M : M;
    g.K = isE ? 2 * cin : cin;
    g.N = isE ? 2 * cout : cout;
    g.pair = isE ? 1 : 0;
  }
  if (cs)
    for (int i = 0; i < 5; ++i)
      if (!cs[i]) return OCTIC_ENULL;
  return dispatch_gemm(a, dtype, out_dtype, (hipStream_t)stream);
}

int octic_lift_gemm(const void* patches, const void* w, const float* bias, const float* pos, float* out, int64_t B,
                    int64_t n_patches, int tok0, int Kpad, int D, int dtype, void* stream) {
  if (!patches || !w || !out) return OCTIC_ENULL;
  if (B <= 0 || n_patches <= 0 || tok0 < 0 || Kpad <= 0 || (Kpad % 8) || D <= 0 || (D % 8)) return OCTIC_ESHAPE;
  if ((((uintptr_t)patches) | ((uintptr_t)w) | ((uintptr_t)out)) & 15) return OCTIC_EALIGN;
  GemmArgs a = {};
  a.ngroups = 1;
  a.rs = nullptr;
  a.rps = 1;
  a.lift_np = n_patches;
  a.lift_tok0 = tok0;
  GemmGroup& g = a.g[0];
  g.a = (const char*)patches;
  g.a_ld = Kpad;
  g.w = (const char*)w;
  g.y = (char*)out;
  g.y_ld = D;
  g.resid = (const char*)pos;
  g.r_ld = D;
  g.bias = bias;  // length D (zero outside the A1 block)
  g.cs = nullptr;
  g.rows = B * n_patches;
  g.K = Kpad;
  g.N = D;
  g.pair = 0;
  return dispatch_gemm(a, dtype, OCTIC_F32, (hipStream_t)stream);
}

}  // extern "C"
