// W-stationary streaming GEMM for the short-K LinearD8 problems (k-chunk = cin <= 160: qkv, proj, fc1 and the input
// gradient of fc2 at ViT-H) — bf16 operands.  Reference semantics: octic_vits/d8_layers.py:104-130 (LinearD8.forward),
// one launch for the five irrep sub-problems.
//
// These problems move 5-10 bytes per MFMA flop-pair less than a dense GEMM of the same rows: they are bound by the
// token rows going in and out, not by the matrix pipe.  The X-stationary kernel of rounds 1-2 (removed) re-streamed the
// weights once per 128 rows (316 MB of L2->LDS traffic for a 210 MB fc1) and ran load / multiply / store as phases of
// a short-lived workgroup (fc1 80-89 us).  Here the roles are swapped (fc1 55-58 us):
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

#ifndef OCTIC_WREG_ABL
#define OCTIC_WREG_ABL 0   // developer ablation builds (tools/wreg_variants.py): 1 no X DMA, 2 no global stores, 4 no MFMAs, 8 no W loads
#endif
#ifndef OCTIC_WREG_S
#define OCTIC_WREG_S 4
#endif
#ifndef OCTIC_WREG_WGS
#define OCTIC_WREG_WGS 2   // workgroups per CU the launch is sized for
#endif
// time model of the work partition (cycles; launch_t)
#ifndef OCTIC_WREG_FIX0
#define OCTIC_WREG_FIX0 5000.0
#endif
#ifndef OCTIC_WREG_FIX1
#define OCTIC_WREG_FIX1 300.0
#endif
#ifndef OCTIC_WREG_STEP0
#define OCTIC_WREG_STEP0 700.0
#endif
#ifndef OCTIC_WREG_EPI
#define OCTIC_WREG_EPI 900.0
#endif
#ifndef OCTIC_WREG_CPB
#define OCTIC_WREG_CPB 0.1   // cycles per byte of a workgroup at two per CU
#endif
constexpr int TM = 32;     // GEMM rows per ring stage
constexpr int KSC = 5;     // MFMA k-steps per k-chunk (k-chunk <= 160 elements)
constexpr int S = OCTIC_WREG_S;   // ring stages (3, 4, 5 measured within noise of each other)
constexpr int MAXD = 3;    // DMA wave-instructions per wave and stage (<= 10 per stage over 4 waves)

__device__ char g_sink[64 * 16];

#ifdef OCTIC_WREG_TRACE
// developer-only timeline (build with -DOCTIC_WREG_TRACE, tools/wreg_trace.py): clock stamps of every workgroup's waves
__device__ unsigned long long g_trace[1024 * 4 * 128];
#define WTRACE(slot)                                                                                                    \
  do {                                                                                                                  \
    if (trace_item < 1024 && lane == 0 && (slot) < 128) g_trace[(trace_item * 4 + wid) * 128 + (slot)] = __builtin_readcyclecounter(); \
  } while (0)
#else
#define WTRACE(slot) do {} while (0)
#endif

// routing override OCTIC_ROUTE_LINEAR_RING: 1 = never take this kernel (A/B against the ring kernel)

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
template <typename TOUT, int EPI, int NCH, int NTW, bool FULLK>
__device__ __forceinline__ void body(const GemmArgs& args, const GemmGroup& G, const int lt, char* lds) {
  constexpr int ES = (int)sizeof(TOUT);
  constexpr int SRS = NTW * 16 * ES + 16;   // staged output row stride (bytes)
  constexpr int NSMAX = NTW * ES / 2;       // row-wise store instructions per tile when the wave owns NTW tiles
  constexpr bool kPair = NCH == 2;          // the E irrep: GEMM row mm = token mm >> 1, half mm & 1

  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int fr = lane & 15, kg = lane >> 4;
#ifdef OCTIC_WREG_TRACE
  const int trace_item = lt + G.tile_begin;
#endif
  WTRACE(0);
  const int K = G.K, N = G.N;
  int g_rows = (int)G.rows;   // pinned in an SGPR below: a scalar re-load inside the loop would force lgkmcnt(0) waits
  // FULLK: the k-chunk is exactly KSC MFMA k-steps (cin = 160, ViT-H): no k-step guards, pipelined fragment reads
  const int Kc = K / NCH, ksteps = FULLK ? KSC : (Kc >> 5), cpr = Kc >> 3, rowb = Kc * 2, stage_b = TM * rowb;

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
  const int64_t tile_stride = kPair ? (TM / 2) * a_row_b : TM * a_row_b;
  int xoff[MAXD];
  auto set_xoff = [&](int rows_valid) {
#pragma unroll
    for (int q = 0; q < MAXD; ++q) {
      const int p = (wid + 4 * q) * 64 + lane;
      int row = p / cpr;
      const int lc = (p - row * cpr) ^ swz(row);
      row = row < rows_valid ? row : rows_valid - 1;   // clamped rows only feed outputs that go to the sink
      xoff[q] = (int)(kPair ? (row >> 1) * a_row_b + (row & 1) * (int64_t)K * 2 : row * a_row_b) + lc * 16;
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
      if (q < D && !(OCTIC_WREG_ABL & 1))
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
      for (int ks = 0; ks < KSC; ++ks) {
        if constexpr (OCTIC_WREG_ABL & 8) {
          wf[i][c * KSC + ks] = bf16x8{(bf16)1.f, (bf16)0.5f, (bf16)0.25f, (bf16)0.f, (bf16)1.f, (bf16)0.5f, (bf16)0.25f, (bf16)0.f};
          asm volatile("" ::"v"(wrow));
        } else
          wf[i][c * KSC + ks] = *(const bf16x8*)(wrow + c * Kc + (ks < ksteps ? ks : 0) * 32);
      }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  // re-define the fragments: hipcc's waitcnt pass cannot see that loads issued before a loop are complete after its
  // first trip and would protect every MFMA block with `s_waitcnt vmcnt(0)`, draining the DMA ring at every step
#pragma unroll
  for (int i = 0; i < NTW; ++i)
#pragma unroll
    for (int k = 0; k < NCH * KSC; ++k) asm volatile("" : "+v"(wf[i][k]));
  WTRACE(1);

  // ---- B-operand read addresses: row (16 jj + fr), chunk (4 ks + kg) ^ swz
  const int sw = swz(fr);
  int ksw16 = (kg ^ sw) << 4;   // chunk (4 m + kg) ^ sw = 4 m ^ (kg ^ sw): one register instead of four offsets
  int rb0 = fr * rowb;
  const int rb16 = 16 * rowb;

  // ---- row-wise store map of the wave's staged [32][16 ntw] block
  const int cprow = ntw * ES;              // 16-byte chunks per staged row
  const int NS = (TM * cprow) >> 6;        // store instructions per tile (every lane used)
  const int64_t y_row_b = G.y_ld * ES, r_row_b = G.r_ld * ES;
  const int64_t y_tile_stride = kPair ? (TM / 2) * y_row_b : TM * y_row_b;
  const int64_t r_tile_stride = kPair ? (TM / 2) * r_row_b : TM * r_row_b;
  // row / chunk of store instruction t of this lane are re-derived at use (q / cprow by multiply-shift, exact for
  // q < 384, cprow <= 20): precomputed offset arrays cost 20 VGPRs next to 160 of W fragments
  const int inv_cprow = cprow ? (65536 + cprow - 1) / cprow : 0;
  int lane_l = lane;   // re-defined inside the loop: keeps hipcc from hoisting every derived offset into its own VGPR
  auto pk_of = [&](int t) {
    const int q = lane_l + 64 * t;
    const int row = (q * inv_cprow) >> 16;
    const int cc = q - row * cprow;
    const bool ok = n0 + cc * (16 / ES) < N;
    return row | (cc << 8) | (ok ? 0 : (1 << 30));
  };
  const int y_rb = (int)y_row_b, r_rb = (int)r_row_b, n_b = N * ES;
  auto pk_row = [&](int v) { return v & 0xff; };
  auto pk_so = [&](int v) { return (v & 0xff) * SRS + ((v >> 8) & 0xff) * 16; };
  auto pk_go = [&](int v, int rowb_) {
    const int row = v & 0xff, colb = n0 * ES + ((v >> 8) & 0xff) * 16;
    return (kPair ? (row >> 1) * rowb_ + (row & 1) * n_b : row * rowb_) + colb;
  };
  auto pk_ok = [&](int v, int rows_valid) { return v < (1 << 30) && (v & 0xff) < rows_valid; };
  const bool has_res = EPI == 1 && G.resid != nullptr, has_rs = EPI == 1 && args.rs != nullptr;
  const int R = (has_res ? NS : 0) + (has_rs ? 2 : 0);
  char* const sink = g_sink + lane * 16;
  char* y_t = G.y + (int64_t)t_begin * y_tile_stride;
  const char* r_t = G.resid + (int64_t)t_begin * r_tile_stride;
  const int wave_col = wt0 * 16 + kg * 4;

  // ---- counted waits.  Per stage step a wave issues, in this order: [R residual / row-scale loads, first k-chunk
  // only] D DMA instructions (stage u + S - 1) [NS stores, last k-chunk only].  At the top of step u the DMA of stage u
  // (issued S - 1 steps earlier) is followed by younger[u % NCH] instructions; `s_waitcnt vmcnt(n)` with n <= that
  // count guarantees it has landed (one in-order counter for loads, DMA and stores).  The immediate is fixed at 8 (or a
  // drain when fewer are younger, during ramp-up and in the tail where no DMA is issued any more): what it additionally
  // waits for are the OLDEST of the younger instructions - stores issued S - 1 steps ago, long complete.  (The first
  // version tracked the exact count and dispatched over 48 immediates: ~550 cycles of scalar compare ladder per step.)
  bool wait8[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    int younger = 0;
#pragma unroll
    for (int k = S - 1; k >= 1; --k) {           // step u - k has k-chunk index (c - k) mod NCH
      const int ck = ((c - k) % NCH + NCH) % NCH;
      if (k < S - 1) younger += (ck == 0 ? R : 0) + D;
      younger += (ck == NCH - 1 && !(OCTIC_WREG_ABL & 2)) ? NS : 0;
    }
    wait8[c] = younger >= 8;
  }
  int c_stage = 0;
  f32x4 acc[NTW][2];
  u32x4 rr[NSMAX];
  float rsv[2] = {1.f, 1.f};
  // fast epilogue: every store instruction of the wave is whole (all NTW tiles owned, all columns inside N)
  const bool cols_whole = ntw == NTW && n0 + NTW * 16 <= N;

  asm volatile("" : "+s"(g_rows));
  for (int tile = 0; tile < ntiles; ++tile) {
    const int row0 = (t_begin + tile) * TM;
    const int rows_valid = (g_rows - row0) < TM ? (g_rows - row0) : TM;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int u = tile * NCH + c;
      asm volatile("" : "+v"(lane_l), "+v"(ksw16), "+v"(rb0));
      if (wait8[c] && u >= S - 1 && u + S - 1 < steps) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      WTRACE(2 + 3 * u);
      __builtin_amdgcn_s_barrier();
      WTRACE(3 + 3 * u);
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
              int mm = row0 + jj * 16 + fr;
              mm = mm < g_rows ? mm : g_rows - 1;
              rsv[jj] = args.rs[(kPair ? (mm >> 1) : mm) / (int)args.rps];
            }
          }
          if (has_res) {
#pragma unroll
            for (int t = 0; t < NSMAX; ++t)
              if (t < NS) rr[t] = *(const u32x4*)(pk_ok(pk_of(t), rows_valid) ? r_t + pk_go(pk_of(t), r_rb) : (const char*)sink);
          }
        }
      }
      if (u + S - 1 < steps) issue();
      const char* st = ring + c_stage * stage_b;
      c_stage = c_stage == S - 1 ? 0 : c_stage + 1;
      const unsigned st_a = lds_addr(st) + rb0;
      auto xload = [&](int jj, int ks) { return lds_read16(st_a + (jj ? rb16 : 0) + 256 * (ks >> 2) + ((64 * (ks & 3)) ^ ksw16)); };
      auto mm = [&](int jj, int ks, const bf16x8 xb) {
#pragma unroll
        for (int i = 0; i < NTW; ++i)
          if constexpr (OCTIC_WREG_ABL & 4) acc[i][jj][0] += (float)wf[i][c * KSC + ks][0] * (float)xb[0];
          else acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i][c * KSC + ks], xb, acc[i][jj], 0, 0, 0);
      };
      if constexpr (FULLK) {
        constexpr int PD = 3, NX = 2 * KSC;   // fragments in flight; fragment n = rows 16 (n / KSC) + fr, k-step n % KSC
        bf16x8 xq[PD];
#pragma unroll
        for (int n = 0; n < PD; ++n) xq[n] = xload(n / KSC, n % KSC);
#pragma unroll
        for (int n = 0; n < NX; ++n) {
          // fragments n + 1 .. min(NX, n + PD) - 1 are younger than fragment n
          if (n + PD <= NX) landed<PD - 1>(xq[n % PD]);
          else if (n + 2 == NX) landed<1>(xq[n % PD]);
          else landed<0>(xq[n % PD]);
          mm(n / KSC, n % KSC, xq[n % PD]);
          if (n + PD < NX) xq[n % PD] = xload((n + PD) / KSC, (n + PD) % KSC);
          __builtin_amdgcn_sched_barrier(0);   // keep the MFMAs between the reads
        }
      } else {
#pragma unroll
        for (int n = 0; n < 2 * KSC; ++n)
          if (n % KSC < ksteps) {
            bf16x8 xb = xload(n / KSC, n % KSC);
            landed<0>(xb);
            mm(n / KSC, n % KSC, xb);
          }
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
        // phase B: rows of the block, 16 bytes per lane (+ residual); all reads first, then the stores
        u32x4 ov[NSMAX];
        if (cols_whole && rows_valid == TM) {       // no masks, no sink, no per-instruction branches
#pragma unroll
          for (int t = 0; t < NSMAX; ++t) ov[t] = *(const u32x4*)(stg + pk_so(pk_of(t)));
#pragma unroll
          for (int t = 0; t < NSMAX; ++t) {
            u32x4 v = ov[t];
            if (has_res) v = add_resid<TOUT>(v, rr[t]);
            if constexpr (OCTIC_WREG_ABL & 2) asm volatile("" ::"v"(v));   // no store at all: a shared sink line would serialise in L2
            else *(u32x4*)(y_t + pk_go(pk_of(t), y_rb)) = v;
          }
        } else {
#pragma unroll
          for (int t = 0; t < NSMAX; ++t)
            if (t < NS) ov[t] = *(const u32x4*)(stg + pk_so(pk_of(t)));
#pragma unroll
          for (int t = 0; t < NSMAX; ++t)
            if (t < NS) {
              u32x4 v = ov[t];
              if (has_res) v = add_resid<TOUT>(v, rr[t]);
              if constexpr (OCTIC_WREG_ABL & 2) asm volatile("" ::"v"(v));
              else *(u32x4*)(pk_ok(pk_of(t), rows_valid) ? y_t + pk_go(pk_of(t), y_rb) : sink) = v;
            }
        }
        y_t += y_tile_stride;
        r_t += r_tile_stride;
      }
      WTRACE(4 + 3 * u);
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
  const bool fullk = G.K == (G.pair ? 2 : 1) * 32 * KSC;
  if (G.pair) {
    if (fullk) body<TOUT, EPI, 2, NTW_E, true>(args, G, item - G.tile_begin, lds);
    else body<TOUT, EPI, 2, NTW_E, false>(args, G, item - G.tile_begin, lds);
  } else {
    if (fullk) body<TOUT, EPI, 1, NTW_A, true>(args, G, item - G.tile_begin, lds);
    else body<TOUT, EPI, 1, NTW_A, false>(args, G, item - G.tile_begin, lds);
  }
}

static int cu_count() {
  return device_cus();
}

template <typename TOUT>
int launch_t(GemmArgs& a, hipStream_t s) {
  constexpr int ES = (int)sizeof(TOUT);
  if (a.lift_np > 0 || route(OCTIC_ROUTE_LINEAR_RING)) return -100;
  int Kc = 0;
  bool fused = a.rs != nullptr;
  for (int i = 0; i < a.ngroups; ++i) {
    const GemmGroup& g = a.g[i];
    const int nch = g.pair ? 2 : 1;
    if (g.K % (32 * nch) || g.rows <= 0 || g.rows >= (1ll << 31)) return -100;
    if (Kc && g.K / nch != Kc) return -100;
    Kc = g.K / nch;
    fused = fused || g.cs || g.resid;
  }
  if (Kc < 32 || Kc > 32 * KSC) return -100;
  // column tiles per wave: what fits next to the accumulators (and the residual registers of the fused epilogue)
  const int NTW_A = ES == 2 ? (fused ? 4 : 5) : 3, NTW_E = ES == 2 ? (fused ? 3 : 4) : 3, NTW_MAX = NTW_A;
  // column chunks, and a time model of one workgroup of a chunk in cycles (from the s_memtime timeline,
  // tools/wreg_trace.py): a fixed part - its W fragments, 16 partial lines per load instruction - and a part per row
  // tile - the MFMAs of two co-resident waves per SIMD plus the barrier / wait / epilogue overhead of a step.
  // Workgroups per chunk are chosen so that every workgroup of the launch ends at about the same time.
  double fixed[5], per_tile[5];
  for (int i = 0; i < a.ngroups; ++i) {
    GemmGroup& g = a.g[i];
    const int ntw = g.pair ? NTW_E : NTW_A, nch = g.pair ? 2 : 1, cap = 4 * ntw;
    g.n_tiles = (g.N + 15) / 16;
    g.n_chunks = (g.n_tiles + cap - 1) / cap;
    g.chunk = (g.n_tiles + g.n_chunks - 1) / g.n_chunks;
    g.n_chunks = (g.n_tiles + g.chunk - 1) / g.chunk;
    g.m_tiles = (int)((g.rows + TM - 1) / TM);
    const double frags = (double)ntw * nch * (Kc / 32);          // W load instructions = MFMAs per 16 rows, per wave
    fixed[i] = OCTIC_WREG_FIX0 + OCTIC_WREG_FIX1 * frags;
    per_tile[i] = OCTIC_WREG_STEP0 * nch + OCTIC_WREG_EPI + 2 * 2 * 16.0 * frags;
    // ... or the bytes of the tile at the workgroup's share of the HBM rate, whichever is longer (f32 + residual outputs)
    const double bytes = TM * (g.chunk * 16.0 * ES * (g.resid ? 2 : 1) + g.K * 2.0);
    per_tile[i] = per_tile[i] > OCTIC_WREG_CPB * bytes ? per_tile[i] : OCTIC_WREG_CPB * bytes;
  }
  const int target = OCTIC_WREG_WGS * cu_count();
  // smallest common end time T whose workgroup counts  ceil(m_tiles per_tile / (T - fixed))  fit the launch
  double lo = 0, hi = 0;
  for (int i = 0; i < a.ngroups; ++i) {
    lo = fixed[i] + per_tile[i] > lo ? fixed[i] + per_tile[i] : lo;
    hi = fixed[i] + per_tile[i] * a.g[i].m_tiles > hi ? fixed[i] + per_tile[i] * a.g[i].m_tiles : hi;
  }
  auto fit = [&](double T, bool commit) {
    int used = 0;
    for (int i = 0; i < a.ngroups; ++i) {
      GemmGroup& g = a.g[i];
      int w = (int)((g.m_tiles * per_tile[i]) / (T - fixed[i]) + 0.999);
      w = w < 1 ? 1 : (w > g.m_tiles ? g.m_tiles : w);
      if (commit) g.wgs = w;
      used += w * g.n_chunks;
    }
    return used;
  };
  if (fit(lo, false) <= target) hi = lo;
  for (int it = 0; it < 40 && hi - lo > 1.0; ++it) {
    const double mid = 0.5 * (lo + hi);
    if (fit(mid, false) <= target) hi = mid;
    else lo = mid;
  }
  fit(hi, true);
  int t = 0;
  for (int i = 0; i < a.ngroups; ++i) {
    a.g[i].tile_begin = t;
    t += a.g[i].wgs * a.g[i].n_chunks;
  }
  a.total_tiles = t;
  const size_t smem = (size_t)S * TM * Kc * 2 + 4 * TM * (NTW_MAX * 16 * ES + 16) + 2 * 4 * NTW_MAX * 16 * sizeof(float);
  static DeviceOnce once;
  if (once.first()) {
    hipFuncSetAttribute((const void*)linear_d8_wreg_kernel<TOUT, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    hipFuncSetAttribute((const void*)linear_d8_wreg_kernel<TOUT, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    (void)hipGetLastError();
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

#ifdef OCTIC_WREG_TRACE
extern "C" void* octic_dbg_wreg_trace(void) {
  void* p = nullptr;
  (void)hipGetSymbolAddress(&p, HIP_SYMBOL(octic::wr::g_trace));
  return p;
}
#endif

// developer switch (not part of the ABI contract): 1 = route the short-K problems to the ring kernel instead

