// W-stationary streaming GEMM for the short-K LinearD8 problems (k-chunk = cin <= 160: qkv, proj, fc1 and the input
// gradient of fc2 at ViT-H) — bf16 operands.  Reference semantics: octic_vits/d8_layers.py:104-130 (LinearD8.forward),
// one launch for the five irrep sub-problems.
//
// These problems move 5-10 bytes per MFMA flop-pair less than a dense GEMM of the same rows: they are bound by the
// token rows going in and out, not by the matrix pipe.  The X-stationary kernel (gemm.hip) re-streamed the weights
// once per 128 rows (316 MB of L2->LDS traffic for a 210 MB fc1) and ran load / multiply / store as phases of a
// short-lived workgroup.  Here the roles are swapped:
//   * a workgroup owns up to 320 (1-D irreps) or 256 (E irrep) OUTPUT COLUMNS of one irrep for its whole life and keeps
//     that slice of W as MFMA A-operand fragments in registers (<= 160 VGPRs per lane), loaded once;
//   * it then streams a contiguous range of token rows: 32-row tiles of X arrive through a 4-stage LDS-DMA ring
//     (row-contiguous 16-byte pieces: every request covers whole 64-byte groups of a row, no fragment-shaped gathers),
//     each wave multiplies the tile with its columns (v_mfma_f32_16x16x32_bf16, X^T as the B operand read with
//     conflict-free ds_read_b128), stages its 32 x 80 result block in LDS and stores it row-wise, 16 bytes per lane;
//   * the E irrep's K = 2 cin goes through the ring as two k-chunks of cin, so every stage has the same shape;
//   * loads, multiplies and stores of one workgroup never wait for each other: the ring is tracked with counted
//     `s_waitcnt vmcnt(n)`, n = the exact number of younger vector-memory instructions of the wave (DMA, residual
//     loads and stores all share CDNA's one in-order counter), one `s_barrier` per stage.
// Launch = ~2 workgroups per CU for the whole problem (row ranges sized by bytes moved), sibling column chunks of one
// row range adjacent in the XCD-contiguous block order so the X rows are fetched from HBM once.
#include "gemm_args.hpp"

namespace octic {
namespace wr {

constexpr int TM = 32;     // GEMM rows per ring stage
constexpr int KSC = 5;     // MFMA k-steps per k-chunk (k-chunk <= 160 elements)
constexpr int S = 4;       // ring stages
constexpr int MAXD = 3;    // DMA wave-instructions per wave and stage (<= 10 per stage over 4 waves)

__device__ char g_sink[64 * 16];

int g_off = 0;   // developer switch: 1 = never take this kernel (A/B against the X-stationary one)

#define OCTIC_WCASE(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
__device__ __forceinline__ void wait_vm(int n) {   // n wave-uniform; anything outside the table drains (always safe)
  switch (n) {
    OCTIC_WCASE(1) OCTIC_WCASE(2) OCTIC_WCASE(3) OCTIC_WCASE(4) OCTIC_WCASE(5) OCTIC_WCASE(6) OCTIC_WCASE(7) OCTIC_WCASE(8)
    OCTIC_WCASE(9) OCTIC_WCASE(10) OCTIC_WCASE(11) OCTIC_WCASE(12) OCTIC_WCASE(13) OCTIC_WCASE(14) OCTIC_WCASE(15)
    OCTIC_WCASE(16) OCTIC_WCASE(17) OCTIC_WCASE(18) OCTIC_WCASE(19) OCTIC_WCASE(20) OCTIC_WCASE(21) OCTIC_WCASE(22)
    OCTIC_WCASE(23) OCTIC_WCASE(24) OCTIC_WCASE(25) OCTIC_WCASE(26) OCTIC_WCASE(27) OCTIC_WCASE(28) OCTIC_WCASE(29)
    OCTIC_WCASE(30) OCTIC_WCASE(31) OCTIC_WCASE(32) OCTIC_WCASE(33) OCTIC_WCASE(34) OCTIC_WCASE(35) OCTIC_WCASE(36)
    OCTIC_WCASE(37) OCTIC_WCASE(38) OCTIC_WCASE(39) OCTIC_WCASE(40) OCTIC_WCASE(41) OCTIC_WCASE(42) OCTIC_WCASE(43)
    OCTIC_WCASE(44) OCTIC_WCASE(45) OCTIC_WCASE(46) OCTIC_WCASE(47) OCTIC_WCASE(48)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}
#undef OCTIC_WCASE

template <typename TOUT>
__device__ __forceinline__ void stage_out4(char* p, f32x4 v);
template <>
__device__ __forceinline__ void stage_out4<float>(char* p, f32x4 v) { *(f32x4*)p = v; }
template <>
__device__ __forceinline__ void stage_out4<bf16>(char* p, f32x4 v) {
  bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
  *(bf16x4*)p = o;
}

template <typename TOUT>
__device__ __forceinline__ u32x4 add_resid(u32x4 v, u32x4 r) {
  if constexpr (sizeof(TOUT) == 4) {
    return __builtin_bit_cast(u32x4, __builtin_bit_cast(f32x4, v) + __builtin_bit_cast(f32x4, r));
  } else {
    const bf16x8 a = __builtin_bit_cast(bf16x8, v), c = __builtin_bit_cast(bf16x8, r);
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (bf16)((float)a[e] + (float)c[e]);
    return __builtin_bit_cast(u32x4, o);
  }
}

// One workgroup's life.  NCH = k-chunks per row tile (1: one-dimensional irreps, 2: the E pair rows), NTW = 16-column
// MFMA tiles a wave can own.
template <typename TOUT, int EPI, int NCH, int NTW>
__device__ __forceinline__ void body(const GemmArgs& args, const GemmGroup& G, const int lt, char* lds) {
  constexpr int ES = (int)sizeof(TOUT);
  constexpr int SRS = NTW * 16 * ES + 16;   // staged output row stride (bytes)
  constexpr int NSMAX = NTW * ES / 2;       // row-wise store instructions per tile when the wave owns NTW tiles

  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int fr = lane & 15, kg = lane >> 4;
  const int K = G.K, N = G.N;
  const int Kc = K / NCH, ksteps = Kc >> 5, cpr = Kc >> 3, rowb = Kc * 2, stage_b = TM * rowb;

  // ---- work: column chunk chunk_id of this irrep, row tiles [t_begin, t_end)
  const int chunk_id = lt % G.n_chunks, jw = lt / G.n_chunks;
  const int t_begin = (int)((int64_t)jw * G.m_tiles / G.wgs);
  const int t_end = (int)((int64_t)(jw + 1) * G.m_tiles / G.wgs);
  const int ntiles = t_end - t_begin;
  const int steps = ntiles * NCH;
  const int c_first = chunk_id * G.chunk;
  const int ct = (G.n_tiles - c_first) < G.chunk ? (G.n_tiles - c_first) : G.chunk;
  const int cb = ct >> 2, crem = ct & 3;
  const int ntw = cb + (wid < crem ? 1 : 0);                 // column tiles of this wave (<= NTW)
  const int wt0 = wid * cb + (wid < crem ? wid : crem);      // its first tile inside the chunk
  const int n0 = (c_first + wt0) * 16;

  char* const ring = lds;
  char* const stg = lds + S * stage_b + wid * (TM * SRS);
  float* const lbias = (float*)(lds + S * stage_b + 4 * TM * SRS);
  float* const lcs = lbias + 4 * NTW * 16;

  // XOR applied to a row's 16-byte chunk index so the B-operand reads (16 rows x one chunk column per lane group) are
  // conflict-free at a row stride of cpr chunks: two bits for cpr = 4 (mod 8), three for 8 (mod 16), four for 0 (mod 16).
  auto swz = [&](int row) {
    return (cpr & 15) == 0 ? (row & 15) : (cpr & 7) == 0 ? ((row >> 1) & 7) : ((0x1320 >> (((row >> 2) & 3) * 4)) & 3);
  };

  // ---- X DMA: stage image = [32 rows][Kc bf16] row-contiguous; wave-instruction `inst` covers chunks 64 inst .. +63
  const int n_inst = cpr >> 1;
  const int D = (n_inst - wid + 3) >> 2;     // this wave issues instructions wid, wid + 4, ...
  const int64_t a_row_b = G.a_ld * 2;
  const int64_t tile_stride = G.pair ? (TM / 2) * a_row_b : TM * a_row_b;
  int xoff[MAXD];
  auto set_xoff = [&](int rows_valid) {
#pragma unroll
    for (int q = 0; q < MAXD; ++q) {
      const int p = (wid + 4 * q) * 64 + lane;
      int row = p / cpr;
      const int lc = (p - row * cpr) ^ swz(row);
      row = row < rows_valid ? row : rows_valid - 1;   // clamped rows only feed outputs that go to the sink
      xoff[q] = (int)(G.pair ? (row >> 1) * a_row_b + (row & 1) * (int64_t)K * 2 : row * a_row_b) + lc * 16;
    }
  };
  const int64_t last_row0 = (int64_t)(G.m_tiles - 1) * TM;
  const int rv_last = (int)(G.rows - last_row0) < TM ? (int)(G.rows - last_row0) : TM;
  const bool partial_last = t_end == G.m_tiles && rv_last < TM;
  set_xoff(ntiles == 1 && partial_last ? rv_last : TM);
  const char* l_src = G.a + (int64_t)t_begin * tile_stride;
  int l_c = 0, l_u = 0, l_stage = 0;
  auto issue = [&]() {
    char* st = ring + l_stage * stage_b;
#pragma unroll
    for (int q = 0; q < MAXD; ++q)
      if (q < D)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(l_src + l_c * rowb + xoff[q]),
                                         (__attribute__((address_space(3))) void*)(st + (wid + 4 * q) * 1024), 16, 0, 0);
    l_stage = l_stage == S - 1 ? 0 : l_stage + 1;
    ++l_u;
    if (++l_c == NCH) {
      l_c = 0;
      l_src += tile_stride;
      if (partial_last && l_u == (ntiles - 1) * NCH) set_xoff(rv_last);
    }
  };
#pragma unroll
  for (int pz = 0; pz < S - 1; ++pz)
    if (pz < steps) issue();

  // ---- bias / layer-scale columns of the chunk -> LDS (zeros / ones where absent)
  for (int c = threadIdx.x; c < 4 * NTW * 16; c += 256) {
    const int col = c_first * 16 + c;
    const bool ok = c < ct * 16 && col < N;
    lbias[c] = (G.bias && ok) ? G.bias[col] : 0.f;
    if (EPI) lcs[c] = (G.cs && ok) ? G.cs[col] : 1.f;
  }

  // ---- W fragments of this wave's columns for the whole K: lane (fr, kg) holds W[n0 + 16 i + fr][32 ks + 8 kg .. +7]
  bf16x8 wf[NTW][NCH * KSC];
#pragma unroll
  for (int i = 0; i < NTW; ++i) {
    int col = n0 + i * 16 + fr;
    col = col < N ? col : N - 1;
    const bf16* wrow = (const bf16*)G.w + (int64_t)col * K + kg * 8;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int ks = 0; ks < KSC; ++ks) wf[i][c * KSC + ks] = *(const bf16x8*)(wrow + c * Kc + (ks < ksteps ? ks : 0) * 32);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  // re-define the fragments so the loop carries no pending-VMEM dependence on them (see the X-stationary kernel)
#pragma unroll
  for (int i = 0; i < NTW; ++i)
#pragma unroll
    for (int k = 0; k < NCH * KSC; ++k) asm volatile("" : "+v"(wf[i][k]));

  // ---- B-operand read addresses: row (16 jj + fr), chunk (4 ks + kg) ^ swz
  const int sw = swz(fr);
  int o4[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) o4[m] = ((4 * m + kg) ^ sw) << 4;
  const int rb0 = fr * rowb, rb1 = (16 + fr) * rowb;

  // ---- row-wise store map of the wave's staged [32][16 ntw] block
  const int cprow = ntw * ES;              // 16-byte chunks per staged row
  const int NS = (TM * cprow) >> 6;        // store instructions per tile (every lane used)
  const int64_t y_row_b = G.y_ld * ES, r_row_b = G.r_ld * ES;
  const int64_t y_tile_stride = G.pair ? (TM / 2) * y_row_b : TM * y_row_b;
  const int64_t r_tile_stride = G.pair ? (TM / 2) * r_row_b : TM * r_row_b;
  int so[NSMAX], go[NSMAX], srow[NSMAX], gr[NSMAX];
#pragma unroll
  for (int t = 0; t < NSMAX; ++t) {
    const int q = lane + 64 * t;
    const int row = cprow ? q / cprow : 0;
    const int cc = q - row * cprow;
    const int gcol = n0 + cc * (16 / ES);
    const bool ok = t < NS && gcol < N;
    so[t] = row * SRS + cc * 16;
    go[t] = (int)(G.pair ? (row >> 1) * y_row_b + (row & 1) * (int64_t)N * ES : row * y_row_b) + gcol * ES;
    gr[t] = (int)(G.pair ? (row >> 1) * r_row_b + (row & 1) * (int64_t)N * ES : row * r_row_b) + gcol * ES;
    srow[t] = ok ? row : (1 << 20);
  }
  const bool has_res = EPI == 1 && G.resid != nullptr, has_rs = EPI == 1 && args.rs != nullptr;
  const int R = (has_res ? NS : 0) + (has_rs ? 2 : 0);
  char* const sink = g_sink + lane * 16;
  char* y_t = G.y + (int64_t)t_begin * y_tile_stride;
  const char* r_t = G.resid + (int64_t)t_begin * r_tile_stride;
  const int wave_col = wt0 * 16 + kg * 4;

  // since[k]: vector-memory instructions this wave has issued after the DMA of in-flight stage k (0 = oldest)
  int since[S - 1];
#pragma unroll
  for (int k = 0; k < S - 1; ++k) since[k] = 0;
  auto bump = [&](int x) {
#pragma unroll
    for (int k = 0; k < S - 1; ++k) since[k] += x;
  };
  int c_stage = 0;
  f32x4 acc[NTW][2];
  u32x4 rr[NSMAX];
  float rsv[2] = {1.f, 1.f};

  for (int tile = 0; tile < ntiles; ++tile) {
    const int64_t row0 = (int64_t)(t_begin + tile) * TM;
    const int rows_valid = (int)(G.rows - row0) < TM ? (int)(G.rows - row0) : TM;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      wait_vm(since[0]);
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int k = 0; k < S - 2; ++k) since[k] = since[k + 1];
      if (c == 0) {
#pragma unroll
        for (int i = 0; i < NTW; ++i) {
          const f32x4 b4 = *(const f32x4*)(lbias + wave_col + i * 16);
          acc[i][0] = b4;
          acc[i][1] = b4;
        }
        if (EPI == 1) {
          if (has_rs) {
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
              int64_t mm = row0 + jj * 16 + fr;
              mm = mm < G.rows ? mm : G.rows - 1;
              rsv[jj] = args.rs[(int)(G.pair ? (mm >> 1) : mm) / (int)args.rps];
            }
          }
          if (has_res) {
#pragma unroll
            for (int t = 0; t < NSMAX; ++t)
              if (t < NS) rr[t] = *(const u32x4*)(srow[t] < rows_valid ? r_t + gr[t] : (const char*)sink);
          }
          bump(R);
        }
      }
      if (tile * NCH + c + S - 1 < steps) {
        issue();
        bump(D);
      }
      since[S - 2] = 0;
      const char* st = ring + c_stage * stage_b;
      c_stage = c_stage == S - 1 ? 0 : c_stage + 1;
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int ks = 0; ks < KSC; ++ks)
          if (ks < ksteps) {
            const bf16x8 xb = *(const bf16x8*)(st + (jj ? rb1 : rb0) + 256 * (ks >> 2) + o4[ks & 3]);
#pragma unroll
            for (int i = 0; i < NTW; ++i)
              acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i][c * KSC + ks], xb, acc[i][jj], 0, 0, 0);
          }
      if (c == NCH - 1) {
        // phase A: MFMA layout (lane: token fr, 4 consecutive channels) -> the wave's staging block
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
          for (int i = 0; i < NTW; ++i) {
            f32x4 v = acc[i][jj];
            if (EPI == 1) {
              v *= *(const f32x4*)(lcs + wave_col + i * 16);
              v *= rsv[jj];
            }
            stage_out4<TOUT>(stg + (jj * 16 + fr) * SRS + (i * 16 + kg * 4) * ES, v);
          }
        // phase B: rows of the block, 16 bytes per lane (+ residual)
#pragma unroll
        for (int t = 0; t < NSMAX; ++t)
          if (t < NS) {
            u32x4 v = *(const u32x4*)(stg + so[t]);
            if (has_res) v = add_resid<TOUT>(v, rr[t]);
            *(u32x4*)(srow[t] < rows_valid ? y_t + go[t] : sink) = v;
          }
        bump(NS);
        y_t += y_tile_stride;
        r_t += r_tile_stride;
      }
    }
  }
}

template <typename TOUT, int EPI>
__global__ __launch_bounds__(256, 2) void linear_d8_wreg_kernel(GemmArgs args) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr int NTW_A = sizeof(TOUT) == 2 ? (EPI ? 4 : 5) : 3, NTW_E = sizeof(TOUT) == 2 ? (EPI ? 3 : 4) : 3;
  // consecutive work items (the column chunks of one row range) share an XCD and its L2
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
  const int item = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  int gi = 0;
#pragma unroll
  for (int i = 1; i < 5; ++i)
    if (i < args.ngroups && item >= args.g[i].tile_begin) gi = i;
  const GemmGroup& G = args.g[gi];
  if (G.pair) body<TOUT, EPI, 2, NTW_E>(args, G, item - G.tile_begin, lds);
  else body<TOUT, EPI, 1, NTW_A>(args, G, item - G.tile_begin, lds);
}

static int cu_count() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
  }
  return n;
}

template <typename TOUT>
int launch_t(GemmArgs& a, hipStream_t s) {
  constexpr int ES = (int)sizeof(TOUT);
  if (a.lift_np > 0 || g_off) return -100;
  int Kc = 0;
  bool fused = a.rs != nullptr;
  for (int i = 0; i < a.ngroups; ++i) {
    const GemmGroup& g = a.g[i];
    const int nch = g.pair ? 2 : 1;
    if (g.K % (32 * nch) || g.rows <= 0) return -100;
    if (Kc && g.K / nch != Kc) return -100;
    Kc = g.K / nch;
    fused = fused || g.cs || g.resid;
  }
  if (Kc < 32 || Kc > 32 * KSC) return -100;
  // column tiles per wave: what fits next to the accumulators (and the residual registers of the fused epilogue)
  const int NTW_A = ES == 2 ? (fused ? 4 : 5) : 3, NTW_E = ES == 2 ? (fused ? 3 : 4) : 3, NTW_MAX = NTW_A;
  // column chunks and the bytes one workgroup of a chunk moves per row tile
  double cost[5], total = 0;
  for (int i = 0; i < a.ngroups; ++i) {
    GemmGroup& g = a.g[i];
    const int cap = 4 * (g.pair ? NTW_E : NTW_A);
    g.n_tiles = (g.N + 15) / 16;
    g.n_chunks = (g.n_tiles + cap - 1) / cap;
    g.chunk = (g.n_tiles + g.n_chunks - 1) / g.n_chunks;
    g.n_chunks = (g.n_tiles + g.chunk - 1) / g.chunk;
    g.m_tiles = (int)((g.rows + TM - 1) / TM);
    cost[i] = (double)g.m_tiles * TM * (g.chunk * 16.0 * ES + g.K * 2.0);
    total += g.n_chunks * cost[i];
  }
  const int target = 2 * cu_count();
  int used = 0;
  for (int i = 0; i < a.ngroups; ++i) {
    GemmGroup& g = a.g[i];
    int w = (int)(target * cost[i] / total);
    w = w < 1 ? 1 : (w > g.m_tiles ? g.m_tiles : w);
    g.wgs = w;
    used += w * g.n_chunks;
  }
  for (;;) {   // hand the rounding leftover to the most loaded streams
    int best = -1;
    double load = 0;
    for (int i = 0; i < a.ngroups; ++i) {
      const GemmGroup& g = a.g[i];
      if (g.wgs < g.m_tiles && used + g.n_chunks <= target && cost[i] / g.wgs > load) {
        load = cost[i] / g.wgs;
        best = i;
      }
    }
    if (best < 0) break;
    ++a.g[best].wgs;
    used += a.g[best].n_chunks;
  }
  int t = 0;
  for (int i = 0; i < a.ngroups; ++i) {
    a.g[i].tile_begin = t;
    t += a.g[i].wgs * a.g[i].n_chunks;
  }
  a.total_tiles = t;
  const size_t smem = (size_t)S * TM * Kc * 2 + 4 * TM * (NTW_MAX * 16 * ES + 16) + 2 * 4 * NTW_MAX * 16 * sizeof(float);
  static bool attr_done = false;
  if (!attr_done) {
    hipFuncSetAttribute((const void*)linear_d8_wreg_kernel<TOUT, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    hipFuncSetAttribute((const void*)linear_d8_wreg_kernel<TOUT, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    (void)hipGetLastError();
    attr_done = true;
  }
  if (fused) linear_d8_wreg_kernel<TOUT, 1><<<t, 256, smem, s>>>(a);
  else linear_d8_wreg_kernel<TOUT, 0><<<t, 256, smem, s>>>(a);
  return launch_status();
}

}  // namespace wr

int launch_wreg(GemmArgs& a, int out_dtype, hipStream_t s) {
  if (out_dtype == OCTIC_BF16) return wr::launch_t<bf16>(a, s);
  if (out_dtype == OCTIC_F32) return wr::launch_t<float>(a, s);
  return -100;
}

}  // namespace octic

// developer switch (not part of the ABI contract): 1 = route the short-K problems to the X-stationary kernel instead
extern "C" void octic_dbg_wreg_off(int off) { octic::wr::g_off = off; }
