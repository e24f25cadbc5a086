// Persistent softmax-attention FORWARD for head_dim 80 and T <= 288 tokens (ViT-H/14 at 224x224: T = 257), bf16:
// AttentionD8's core on packed LinearD8 rows (reference octic_vits/d8_layers.py:631-656) and the standard block's fused
// [B,T,3,H,hd] projection (deit/vit.py:38-45).  Same MFMA formulation as csrc/attention.hip (swapped score product, P
// straight from accumulator registers, transposing LDS reads); new: how the operands reach the CU, and the softmax.
// (The backward stays on csrc/attention.hip: persistent dq / dkv kernels on the same tile rings were built and only
// tied the round-2 kernels - DESIGN.md section 3.3 has the measurements and the reason.)
//
// The s_memtime timelines of the round-2 kernels (tools/attn_trace.py, cycles per head and wave at (64,16,257,80)):
// dq 48k of which 24k staging (71k / 43k on packed rows), dkv 57k / 21k (76k / 31k), forward 34k / 5.5k + 3k + 5k of
// waits around a 15.5k pass.  Staging never overlapped anything: two 60 KB row images per workgroup leave no room for
// a second workgroup or a second buffer.  Here:
//   * one workgroup per CU walks its (batch, head) units (persistent);
//   * both operand images live in LDS as 32-row TILES of 5 KiB (8 x 16-byte chunks of a row XOR-swizzled so that the
//     row reads (ds_read_b128) AND the transposing reads (ds_read_b64_tr_b16) are bank-conflict free, + a 32-byte tail
//     per row for elements 64..79): 2 x 9 x 5 KiB = 90 KiB;
//   * every byte is brought in by LDS-DMA (buffer_load ... lds; 16-byte pieces, 4-byte pieces for the split remainders
//     of packed rows): no staging registers, no LDS store instructions.  All eight waves sweep the tiles together; a
//     group of tiles is released by one barrier, and its slots are refilled at once with the NEXT head's rows, which
//     therefore have a whole head's compute time to arrive (one s_waitcnt vmcnt(0) + barrier per head);
//   * the query rows of the 257th token (the tile of one row that does not fit the eight waves) are worked on FIRST,
//     from a 160-byte LDS copy, wave w against key tile w - their fragments are never live during the main sweep.
#include "attn80_common.hpp"

#ifdef A80_TRACE
// developer-only timeline: [kernel 0 fwd / 1 dq / 2 dkv][256 workgroups][8 waves][16 stamps], second unit of each workgroup
__device__ unsigned long long g_a80_trace[3 * 256 * 8 * 16];
extern "C" void* octic_dbg_a80_trace(void) {
  void* p = nullptr;
  (void)hipGetSymbolAddress(&p, HIP_SYMBOL(g_a80_trace));
  return p;
}
#define A80T(kern, slot)                                                                                  \
  do {                                                                                                    \
    if (tr_on && (threadIdx.x & 63) == 0)                                                                 \
      g_a80_trace[(((kern) * 256 + blockIdx.x) * 8 + (threadIdx.x >> 6)) * 16 + (slot)] = __builtin_readcyclecounter(); \
  } while (0)
#else
#define A80T(kern, slot) do {} while (0)
#endif

namespace octic {
namespace a80 {

// ======================================================================================================= forward
// one key tile of the online-softmax forward for the 32 queries of a wave (see fwd_pass in csrc/attention.hip)
__device__ __forceinline__ void fwd_tile(const char* kt_, const char* vt_, const FragAddr& fa, const bf16x8 (&qf)[KS],
                                         int kt, int nt, int T, float scale_log2, int half, float& m, float& l,
                                         f32x16 (&ot)[DT]) {
  f32x16 x;
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = 0.f;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) x = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rowfrag(kt_, fa, ks), qf[ks], x, 0, 0, 0);
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
  const float m_new = fmaxf(m, mx * scale_log2);
  if (__builtin_amdgcn_ballot_w64(m_new > m)) {
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
    ps[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(x[i], scale_log2, -m));
    sum += ps[i];
  }
  l += sum;
  const bf16x8 pb0 = pack8(ps), pb1 = pack8(ps + 8);
#pragma unroll
  for (int d = 0; d < DT; ++d) {
    ot[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag(vt_, fa, d, 0), pb0, ot[d], 0, 0, 0);
    ot[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag(vt_, fa, d, 1), pb1, ot[d], 0, 0, 0);
  }
}

// LDS: [K image nt tiles][V image nt tiles][extra rows nx x 160 B][partials W x nx x (DC f32)][m, l: 2 x W x nx f32]
__global__ __launch_bounds__(512) void fwd_kernel(AttnArgs a, int nt, int units) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int T = a.T;
  const int W = blockDim.x >> 6;
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, half = lane >> 5;
  const int nx = T - 32 * W > 0 ? T - 32 * W : 0;                 // extra rows (<= 2)
  constexpr int DC = DT * 32 + kPartPad;
  char* const Kimg = smem;
  char* const Vimg = smem + nt * TILE_B;
  char* const xrow = smem + 2 * nt * TILE_B;
  float* const parts = (float*)(xrow + 2 * HD * 2);
  float* const ml = parts + WAVES * 2 * DC;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned ldsK = lds0, ldsV = lds0 + nt * TILE_B;
  const int G = gridDim.x;
  const bool shared_rows = a.sH < a.sT;

  FragAddr fa;
  fa.setup(lane);

  auto head_of = [&](int idx, int64_t& in_off, int64_t& o_off, int64_t& st_off, int& h, int64_t& b_in) {
    const int u = unit_of(idx, units, shared_rows);
    const int b = u / a.H;
    h = u - b * a.H;
    b_in = b * a.sB;
    in_off = b * a.sB + h * a.sH;
    o_off = b * a.oB + h * a.oH;
    st_off = ((int64_t)b * a.H + h) * T;
  };
  auto issue_group = [&](int g0, int64_t in_off, const HeadMaps& hm) {
    const i32x4 rk = make_rs(a.k, in_off, a.sT, T, a.cv_in), rv = make_rs(a.v, in_off, a.sT, T, a.cv_in);
#pragma unroll
    for (int j = 0; j < GROUP; ++j)
      if (g0 + j < nt) Stager::issue(wid, W, lane, g0 + j, nt, T, ldsK, ldsV, rk, rv, hm.k.bs, hm.v.bs, a.sT, a.sT, a.cv_in, a.cv_in);
  };

  int u = blockIdx.x;
  int64_t in_off, o_off, st_off, b_in;
  int hh;
  head_of(u, in_off, o_off, st_off, hh, b_in);
  HeadMaps hm = head_maps(a, hh);
  for (int g0 = 0; g0 < nt; g0 += GROUP) issue_group(g0, in_off, hm);
  bf16x8 qf[KS];
  load_rows(qf, a.q + in_off, a.sT, wid, T, lane, hm.q);
  u32x4 xq = {0, 0, 0, 0};                                        // extra query rows: wave 1, lanes < 10 nx
  if (wid == 1 && lane < 10 * nx) xq = hm_load16(a.q + in_off + (int64_t)(32 * W + lane / 10) * a.sT, lane % 10, hm.q);

  for (; u < units; u += G) {
    const int un = u + G;
    const bool has_next = un < units;
    int64_t n_in = 0, n_o = 0, n_st = 0, n_b = 0;
    int nh = 0;
    HeadMaps nhm = hm;
    if (has_next) {
      head_of(un, n_in, n_o, n_st, nh, n_b);
      nhm = head_maps(a, nh);
    }
    bf16* ob = a.o + o_off;
    float* lseb = a.lse ? a.lse + st_off : nullptr;
    const bool tr_on = u == (int)blockIdx.x + G && blockIdx.x < 256;
    (void)tr_on;

    A80T(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // this head's tiles, query rows and extra rows
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[ks]));
    if (wid == 1 && lane < 10 * nx) *(u32x4*)(xrow + (lane / 10) * (HD * 2) + (lane % 10) * 16) = xq;
    A80T(0, 1);
    __syncthreads();
    A80T(0, 2);

    // ---- the extra queries first: wave w against key tile w (wave 0 also takes the last tile)
    if (nx > 0) {
      bf16x8 qx[KS];
      xrow_frags(qx, xrow, nx, lane);
      f32x16 ox[DT];
      zero_acc<DT>(ox);
      float mxx = -INFINITY, lx = 0.f;
      for (int kt = wid; kt < nt; kt += W)
        fwd_tile(Kimg + kt * TILE_B, Vimg + kt * TILE_B, fa, qx, kt, nt, T, a.scale_log2, half, mxx, lx, ox);
      lx += __shfl_xor(lx, 32, 64);
      if (r < nx) {
        float* row = parts + ((size_t)wid * nx + r) * DC;
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
          for (int k4 = 0; k4 < 4; ++k4)
            *(f32x4*)(row + d * 32 + 8 * k4 + 4 * half) = f32x4{ox[d][4 * k4], ox[d][4 * k4 + 1], ox[d][4 * k4 + 2], ox[d][4 * k4 + 3]};
        if (half == 0) {
          ml[wid * nx + r] = mxx;
          ml[(W + wid) * nx + r] = lx;
        }
      }
    }

    A80T(0, 3);
    // ---- main sweep.  The barrier in front of tile j says every wave is done with tile j - 1: its slots are refilled
    // with the next head's rows - by waves 0-3 before they compute tile j, by waves 4-7 after (a wave waits in the
    // DMA issue while the CU's memory pipeline drains, ~1 KiB per 100 cycles: its SIMD partner computes meanwhile)
    f32x16 ot[DT];
    zero_acc<DT>(ot);
    float m = -INFINITY, l = 0.f;
    const i32x4 nrk = make_rs(a.k, n_in, a.sT, T, a.cv_in, a.dbg & 1), nrv = make_rs(a.v, n_in, a.sT, T, a.cv_in, a.dbg & 1);
    const bool early = wid < 4;
    for (int j = 0; j < nt; ++j) {
      if (j > 0) {
        __syncthreads();
        if (has_next && early) Stager::issue(wid, W, lane, j - 1, nt, T, ldsK, ldsV, nrk, nrv, nhm.k.bs, nhm.v.bs, a.sT, a.sT, a.cv_in, a.cv_in);
      }
      fwd_tile(Kimg + j * TILE_B, Vimg + j * TILE_B, fa, qf, j, nt, T, a.scale_log2, half, m, l, ot);
      if (j > 0 && has_next && !early) Stager::issue(wid, W, lane, j - 1, nt, T, ldsK, ldsV, nrk, nrv, nhm.k.bs, nhm.v.bs, a.sT, a.sT, a.cv_in, a.cv_in);
    }
    A80T(0, 4);
    __syncthreads();
    A80T(0, 5);
    if (has_next) Stager::issue(wid, W, lane, nt - 1, nt, T, ldsK, ldsV, nrk, nrv, nhm.k.bs, nhm.v.bs, a.sT, a.sT, a.cv_in, a.cv_in);
    A80T(0, 6);
    // next head's own rows (the current fragments are dead), then this head's results
    if (has_next) {
      load_rows(qf, a.q + n_in, a.sT, wid, T, lane, nhm.q);
      if (wid == 1 && lane < 10 * nx) xq = hm_load16(a.q + n_in + (int64_t)(32 * W + lane / 10) * a.sT, lane % 10, nhm.q);
    }
    l += __shfl_xor(l, 32, 64);
    {
      const int qi = wid * 32 + r;
      if (qi < T) {
        if (half == 0 && lseb) lseb[qi] = m + log2f(l);
        store_rows16(ob + (int64_t)qi * a.oT, ot, 1.0f / l, half, hm.o);
      }
    }
    if (nx > 0 && wid == 0) {                                     // merge of the W partial rows (written before the sweep)
      for (int rr = 0; rr < nx; ++rr) {
        float M = -INFINITY;
        for (int w = 0; w < W; ++w) M = fmaxf(M, ml[w * nx + rr]);
        float L = 0.f, o0 = 0.f, o1 = 0.f;
        for (int w = 0; w < W; ++w) {
          const float g = __builtin_amdgcn_exp2f(ml[w * nx + rr] - M);
          L += ml[(W + w) * nx + rr] * g;
          const float* row = parts + ((size_t)w * nx + rr) * DC;
          o0 += g * row[lane];
          if (lane + 64 < HD) o1 += g * row[lane + 64];
        }
        const int qi = W * 32 + rr;
        const float inv = 1.0f / L;
        bf16* orow = ob + (int64_t)qi * a.oT;
        orow[hm_elem(lane, hm.o)] = (bf16)(o0 * inv);
        if (lane + 64 < HD) orow[hm_elem(lane + 64, hm.o)] = (bf16)(o1 * inv);
        if (lane == 0 && lseb) lseb[qi] = M + log2f(L);
      }
    }
    A80T(0, 7);
    in_off = n_in; o_off = n_o; st_off = n_st; b_in = n_b;
    hm = nhm;
  }
}


// ================================================================================ forward, one-shot softmax
// The timeline of the kernels above with every K / V byte dropped (zero-record descriptors) moves by 10 %: the forward
// is bound by its vector instructions, not by memory - 125 of them per 32 x 32 score tile against 11 MFMAs (running
// max, exchange, conditional rescale of 48 accumulators, exp, sum, packing).  T <= 288 keys is short enough to keep a
// query's WHOLE score row in registers (9 tiles x 16): phase A = all score tiles (45 MFMAs), one exact row maximum
// (72 v_max3 + one exchange), phase C = exp2 (one fma + one v_exp per score), packing, P V - no running maximum, no
// rescale.  The row sum comes out of the matrix pipe: element 80 of every V row reads as 1.0 (a constant LDS block
// feeds the lanes of columns 80..95 of the last transposing read), so row 80 of O^T is sum_k P.
// K is dead after phase A and V after phase C, so three image buffers rotate: [K(n) | V(n) | spare]; K(n+1) streams
// into the spare during phase A, V(n+1) into K(n)'s buffer during phase C.  Two barriers per head.
__device__ __forceinline__ bf16x8 trfrag_ones(const char* tile, const FragAddr& fa, int khalf, const char* ones, int lane) {
  // d-tile 2: lanes of columns 64..79 read the tail, lanes of columns 80..95 read the constant block (column 80 = 1)
  const bool hi_cols = (lane >> 4) & 1;
  const char* lo = hi_cols ? ones + (lane & 15) * 8 : tile + fa.tb[4] + khalf * 512;
  const char* hi = hi_cols ? lo : lo + 256;
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)lo);
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)hi);
  const s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(bf16x8, v);
}

// LDS: [3 image buffers x nt tiles][extra rows][ones block 128 B][P of the extra rows: 8 waves x 2 KiB][partials][m, l]
__global__ __launch_bounds__(512) void fwd_os_kernel(AttnArgs a, int units) {
  constexpr int nt = MAXNT;                                       // nine tiles exactly (T = 257, 258): static register indexing
  constexpr int W = WAVES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int T = a.T;
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, half = lane >> 5;
  const int nx = T - 32 * W;                                      // 1 or 2 extra rows
  constexpr int DC = DT * 32 + kPartPad;
  constexpr int IMG = nt * TILE_B;
  char* const xrow = smem + 3 * IMG;
  char* const ones = xrow + 2 * HD * 2;
  char* const pxs = ones + 128;                                   // [8 waves][64 lanes][32 B]
  float* const parts = (float*)(pxs + W * 2048 + 1024);          // + wave 0's block for keys 256..
  float* const ml = parts + W * 2 * DC;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int G = gridDim.x;
  const bool shared_rows = a.sH < a.sT;
  FragAddr fa;
  fa.setup(lane);
  LeanStager st;
  st.setup(wid, W, lane, a.sT, a.cv_in, nt, T);

  auto head_of = [&](int idx, int64_t& in_off, int64_t& o_off, int64_t& st_off, int& h) {
    const int u = unit_of(idx, units, shared_rows);
    const int b = u / a.H;
    h = u - b * a.H;
    in_off = b * a.sB + h * a.sH;
    o_off = b * a.oB + h * a.oH;
    st_off = ((int64_t)b * a.H + h) * T;
  };
  // merge of the W partial extra rows of a finished head (wave 0, after the barrier that follows their P V)
  auto merge_extra = [&](bf16* ob, float* lseb, const HeadMap hmo) {
    for (int rr = 0; rr < nx; ++rr) {
      float M = -INFINITY;
      for (int w = 0; w < W; ++w) M = fmaxf(M, ml[w * 2 + rr]);
      float Ls = 0.f, o0 = 0.f, o1 = 0.f;
      for (int w = 0; w < W; ++w) {
        const float g = __builtin_amdgcn_exp2f(ml[w * 2 + rr] - M);
        const float* row = parts + ((size_t)w * 2 + rr) * DC;
        Ls += g * row[HD];                                         // column 80 of the partial = its row sum (ones column)
        o0 += g * row[lane];
        if (lane + 64 < HD) o1 += g * row[lane + 64];
      }
      const int qi = W * 32 + rr;
      const float inv = 1.0f / Ls;
      bf16* orow = ob + (int64_t)qi * a.oT;
      orow[hm_elem(lane, hmo)] = (bf16)(o0 * inv);
      if (lane + 64 < HD) orow[hm_elem(lane + 64, hmo)] = (bf16)(o1 * inv);
      if (lane == 0 && lseb) lseb[qi] = M + log2f(Ls);
    }
  };
  // the 4 rows x 16 columns block whose transposed read gives "column 80 = 1, columns 81..95 = 0" for four keys
  if (tid < 32) ((unsigned*)ones)[tid] = (tid & 7) == 0 ? 0x3F80u : 0u;

  int u = blockIdx.x;
  int64_t in_off, o_off, st_off;
  int hh;
  head_of(u, in_off, o_off, st_off, hh);
  HeadMaps hm = head_maps(a, hh);
  int bK = 0, bV = 1, bS = 2;                                     // buffer roles
  {
    const i32x4 rk = make_rs(a.k, in_off, a.sT, T, a.cv_in), rv = make_rs(a.v, in_off, a.sT, T, a.cv_in);
    for (int j = 0; j < nt; ++j) {
      st.issue(j, lds0 + bK * IMG, rk, hm.k.bs);
      st.issue(j, lds0 + bV * IMG, rv, hm.v.bs);
    }
  }
  bf16x8 qf[KS];
  load_rows(qf, a.q + in_off, a.sT, wid, T, lane, hm.q);
  u32x4 xq = {0, 0, 0, 0};
  if (wid == 1 && lane < 10 * nx) xq = hm_load16(a.q + in_off + (int64_t)(32 * W + lane / 10) * a.sT, lane % 10, hm.q);
  bf16* p_ob = nullptr;                                           // the previous head (its extra rows are merged one head late)
  float* p_lse = nullptr;
  HeadMap p_hmo = hm.o;

  for (; u < units; u += G) {
    const int un = u + G;
    const bool has_next = un < units;
    int64_t n_in = 0, n_o = 0, n_st = 0;
    int nh = 0;
    HeadMaps nhm = hm;
    if (has_next) {
      head_of(un, n_in, n_o, n_st, nh);
      nhm = head_maps(a, nh);
    }
    bf16* ob = a.o + o_off;
    float* lseb = a.lse ? a.lse + st_off : nullptr;
    const char* Kimg = smem + bK * IMG;
    const char* Vimg = smem + bV * IMG;
    const bool tr_on = u == (int)blockIdx.x + G && blockIdx.x < 256;
    (void)tr_on;
    const i32x4 nrk = make_rs(a.k, n_in, a.sT, T, a.cv_in, a.dbg & 1), nrv = make_rs(a.v, n_in, a.sT, T, a.cv_in, a.dbg & 1);

    A80T(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // K(n), V(n), the query rows and the extra rows
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[ks]));
    if (wid == 1 && lane < 10 * nx) *(u32x4*)(xrow + (lane / 10) * (HD * 2) + (lane % 10) * 16) = xq;
    A80T(0, 1);
    __syncthreads();
    A80T(0, 2);
    if (wid == 0 && p_ob != nullptr) merge_extra(p_ob, p_lse, p_hmo);   // the previous head's extra rows

    // ---- The sweep runs in two chunks (key tiles 0-4, then 5-8) so that at most five score tiles (80 registers) are
    // live: scores of a chunk, its exact maximum, P and P V; between the chunks ONE unconditional rescale of O^T by
    // exp2(m0 - m1).  All nine score tiles at once need 144 + 48 registers and spilled (every reload of a spilled
    // register drains the DMA queue with s_waitcnt vmcnt(0)).
    // K(n+1) streams into the spare buffer during chunk 0, V(n+1) into K(n)'s buffer: tiles 0-4 after the barrier that
    // follows the first chunk's score tiles, tiles 5-8 after the one that follows the second chunk's.  The ninth tile holds one or two keys: only those two scores are kept (rows 0, 1 = registers 0, 1 of
    // the lower half-wave) and only its first k-step is multiplied.  The extra QUERY rows ride along: wave w multiplies
    // key tile w (wave 0 also the ninth) with them, takes a local softmax and parks P in LDS.
    float mxw = -INFINITY;                                        // local maximum of the extra rows (scaled, log2 domain)
    float x8a, x8b;
    {
      bf16x8 qx[KS];
      xrow_frags(qx, xrow, nx, lane);
      f32x16 t, sx, s8;
#pragma unroll
      for (int i = 0; i < 16; ++i) { t[i] = 0.f; sx[i] = 0.f; s8[i] = 0.f; }
      const char* k8 = Kimg + (MAXNT - 1) * TILE_B;
      const char* kw = Kimg + wid * TILE_B;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8 kf8 = rowfrag(k8, fa, ks);
        t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf8, qf[ks], t, 0, 0, 0);
        if (wid == 0) s8 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf8, qx[ks], s8, 0, 0, 0);
        sx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rowfrag(kw, fa, ks), qx[ks], sx, 0, 0, 0);
      }
      x8a = half == 0 ? t[0] : -INFINITY;
      x8b = (half == 0 && nx > 1) ? t[1] : -INFINITY;
      if (has_next) st.issue(MAXNT - 1, lds0 + bS * IMG, nrk, nhm.k.bs);
      // local softmax of the extra rows over the 32 keys of tile w (+ keys 256.. for wave 0)
      float e8a = (wid == 0 && half == 0) ? s8[0] : -INFINITY, e8b = (wid == 0 && half == 0 && nx > 1) ? s8[1] : -INFINITY;
      float mxl = fmaxf(e8a, e8b);
#pragma unroll
      for (int i = 0; i < 16; i += 2) mxl = fmaxf(fmaxf(mxl, sx[i]), sx[i + 1]);
      mxl = fmaxf(mxl, __shfl_xor(mxl, 32, 64));
      mxw = mxl * a.scale_log2;
      float ps[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) ps[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(sx[i], a.scale_log2, -mxw));
      const bf16x8 pb0 = pack8(ps), pb1 = pack8(ps + 8);
      float p8[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) p8[i] = 0.f;
      p8[0] = __builtin_amdgcn_exp2f(__builtin_fmaf(e8a, a.scale_log2, -mxw));
      p8[1] = __builtin_amdgcn_exp2f(__builtin_fmaf(e8b, a.scale_log2, -mxw));
      const bf16x8 pb8 = pack8(p8);
      char* pw = pxs + wid * 2048 + lane * 16;
      *(bf16x8*)pw = pb0;
      *(bf16x8*)(pw + 1024) = pb1;
      if (wid == 0) *(bf16x8*)(pxs + W * 2048 + lane * 16) = pb8;         // wave 0's third block (keys 256..)
    }
    f32x16 ot[DT];
    zero_acc<DT>(ot);
    constexpr int C0 = 5;                                          // tiles of chunk 0; chunk 1 = tiles 5..7 + the ninth
    float m;
    {
      f32x16 x[C0];
#pragma unroll
      for (int j = 0; j < C0; ++j) {
#pragma unroll
        for (int i = 0; i < 16; ++i) x[j][i] = 0.f;
        const char* kt_ = Kimg + j * TILE_B;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) x[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rowfrag(kt_, fa, ks), qf[ks], x[j], 0, 0, 0);
        if (has_next) st.issue(j, lds0 + bS * IMG, nrk, nhm.k.bs);
      }
      float mx = x[0][0];
#pragma unroll
      for (int j = 0; j < C0; ++j) {
#pragma unroll
        for (int i = 0; i < 16; i += 2) mx = fmaxf(fmaxf(mx, x[j][i]), x[j][i + 1]);
      }
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      m = mx * a.scale_log2;
      __syncthreads();                                            // every wave is done with K(n) tiles 0..4
#pragma unroll
      for (int j = 0; j < C0; ++j) {
        float ps[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) ps[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(x[j][i], a.scale_log2, -m));
        const bf16x8 pb0 = pack8(ps), pb1 = pack8(ps + 8);
        const char* vt_ = Vimg + j * TILE_B;
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          ot[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag(vt_, fa, d, 0), pb0, ot[d], 0, 0, 0);
          ot[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag(vt_, fa, d, 1), pb1, ot[d], 0, 0, 0);
        }
        ot[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag_ones(vt_, fa, 0, ones, lane), pb0, ot[2], 0, 0, 0);
        ot[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag_ones(vt_, fa, 1, ones, lane), pb1, ot[2], 0, 0, 0);
        if (has_next && j + C0 < MAXNT - 1) st.issue(j + C0, lds0 + bS * IMG, nrk, nhm.k.bs);   // K(n+1) tiles 5..7
        if (has_next) st.issue(j, lds0 + bK * IMG, nrv, nhm.v.bs);                              // V(n+1) tiles 0..4
      }
    }
    A80T(0, 3);
    {
      f32x16 x[MAXNT - 1 - C0];
#pragma unroll
      for (int j = 0; j < MAXNT - 1 - C0; ++j) {
#pragma unroll
        for (int i = 0; i < 16; ++i) x[j][i] = 0.f;
        const char* kt_ = Kimg + (C0 + j) * TILE_B;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) x[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rowfrag(kt_, fa, ks), qf[ks], x[j], 0, 0, 0);
      }
      float mx = fmaxf(x8a, x8b);
#pragma unroll
      for (int j = 0; j < MAXNT - 1 - C0; ++j) {
#pragma unroll
        for (int i = 0; i < 16; i += 2) mx = fmaxf(fmaxf(mx, x[j][i]), x[j][i + 1]);
      }
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m1 = fmaxf(m, mx * a.scale_log2);
      const float alpha = __builtin_amdgcn_exp2f(m - m1);
      m = m1;
#pragma unroll
      for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int i = 0; i < 16; ++i) ot[d][i] *= alpha;
      __syncthreads();                                            // every wave is done with K(n)
      A80T(0, 4);
      if (has_next) load_rows(qf, a.q + n_in, a.sT, wid, T, lane, nhm.q);   // this head's fragments are dead
#pragma unroll
      for (int j = 0; j < MAXNT - 1 - C0; ++j) {
        float ps[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) ps[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(x[j][i], a.scale_log2, -m));
        const bf16x8 pb0 = pack8(ps), pb1 = pack8(ps + 8);
        const char* vt_ = Vimg + (C0 + j) * TILE_B;
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          ot[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag(vt_, fa, d, 0), pb0, ot[d], 0, 0, 0);
          ot[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag(vt_, fa, d, 1), pb1, ot[d], 0, 0, 0);
        }
        ot[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag_ones(vt_, fa, 0, ones, lane), pb0, ot[2], 0, 0, 0);
        ot[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag_ones(vt_, fa, 1, ones, lane), pb1, ot[2], 0, 0, 0);
        if (has_next) st.issue(C0 + j, lds0 + bK * IMG, nrv, nhm.v.bs);                         // V(n+1) tiles 5..7
      }
      if (has_next) st.issue(MAXNT - 1, lds0 + bK * IMG, nrv, nhm.v.bs);
      {
        // ninth tile: keys 256 (and 257) are rows 0 (1) of its first k-step; exp2(-inf) = 0 for everything else
        float ps[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) ps[i] = 0.f;
        ps[0] = __builtin_amdgcn_exp2f(__builtin_fmaf(x8a, a.scale_log2, -m));
        ps[1] = __builtin_amdgcn_exp2f(__builtin_fmaf(x8b, a.scale_log2, -m));
        const bf16x8 pb0 = pack8(ps);
        const char* vt_ = Vimg + (MAXNT - 1) * TILE_B;
#pragma unroll
        for (int d = 0; d < 2; ++d) ot[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag(vt_, fa, d, 0), pb0, ot[d], 0, 0, 0);
        ot[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag_ones(vt_, fa, 0, ones, lane), pb0, ot[2], 0, 0, 0);
      }
    }
    A80T(0, 5);
    // row 80 of O^T (d-tile 2, accumulator row 16 = register 8 of the lower half-wave) is the row sum of P
    float l = ot[2][8];
    l = __shfl(l, r, 64);
    {
      const int qi = wid * 32 + r;
      if (half == 0 && lseb) lseb[qi] = m + log2f(l);
      store_rows16(ob + (int64_t)qi * a.oT, ot, 1.0f / l, half, hm.o);
    }
    if (has_next && wid == 1 && lane < 10 * nx) xq = hm_load16(a.q + n_in + (int64_t)(32 * W + lane / 10) * a.sT, lane % 10, nhm.q);
    A80T(0, 6);
    {
      // the extra rows' P V: V(n) stays resident until the next head's phase A
      char* pw = pxs + wid * 2048 + lane * 16;
      const bf16x8 pb0 = *(const bf16x8*)pw, pb1 = *(const bf16x8*)(pw + 1024);
      zero_acc<DT>(ot);
      const char* vt_ = Vimg + wid * TILE_B;
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        ot[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag(vt_, fa, d, 0), pb0, ot[d], 0, 0, 0);
        ot[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag(vt_, fa, d, 1), pb1, ot[d], 0, 0, 0);
      }
      ot[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag_ones(vt_, fa, 0, ones, lane), pb0, ot[2], 0, 0, 0);
      ot[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag_ones(vt_, fa, 1, ones, lane), pb1, ot[2], 0, 0, 0);
      if (wid == 0) {
        const bf16x8 pb8 = *(const bf16x8*)(pxs + W * 2048 + lane * 16);
        const char* v8 = Vimg + (MAXNT - 1) * TILE_B;
#pragma unroll
        for (int d = 0; d < 2; ++d) ot[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag(v8, fa, d, 0), pb8, ot[d], 0, 0, 0);
        ot[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag_ones(v8, fa, 0, ones, lane), pb8, ot[2], 0, 0, 0);
      }
      if (r < nx) {
        float* row = parts + ((size_t)wid * 2 + r) * DC;
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
          for (int k4 = 0; k4 < 4; ++k4)
            *(f32x4*)(row + d * 32 + 8 * k4 + 4 * half) = f32x4{ot[d][4 * k4], ot[d][4 * k4 + 1], ot[d][4 * k4 + 2], ot[d][4 * k4 + 3]};
        if (half == 0) ml[wid * 2 + r] = mxw;
      }
    }
    A80T(0, 7);
    p_ob = ob; p_lse = lseb; p_hmo = hm.o;
    in_off = n_in; o_off = n_o; st_off = n_st;
    hm = nhm;
    const int t = bK;                                             // K(n+1) is in the spare, V(n+1) in K(n)'s buffer
    bK = bS; bS = bV; bV = t;
  }
  __syncthreads();
  if (wid == 0 && p_ob != nullptr) merge_extra(p_ob, p_lse, p_hmo);
}


// ================================================================ forward, one-shot softmax, T <= 256 (round 5)
// The same plan for the sequence lengths beside ViT-H/14's 257: NT = ceil(T / 32) <= 8 key tiles = NT waves, every token
// inside a tile (no extra rows, no ninth tile); the last tile may be partial - its rows past T arrive as zeros (the DMA
// reads outside the descriptor) and their scores are set to -inf before the row maximum.  DINOv2 ViT-H/16: T = 197
// (NT = 7: chunks of 5 + 2 key tiles) and T = 37 (NT = 2: one chunk).  Three image buffers rotate as above.
template <int NT>
__global__ __launch_bounds__((NT < 4 ? 4 : NT) * 64) void fwd_oss_kernel(AttnArgs a, int units) {
  // at least four waves: a tile's 5 (plain rows) or 8 (packed rows) DMA jobs are dealt two per wave; waves past NT only stage
  constexpr int W = NT < 4 ? 4 : NT;
  constexpr int C0 = NT < 5 ? NT : 5, C1 = NT - C0;               // key tiles of the two chunks (<= 80 score registers each)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int T = a.T;
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, half = lane >> 5;
  const bool active = wid < NT;                                   // owns a query tile
  constexpr int IMG = NT * TILE_B;
  char* const ones = smem + 3 * IMG;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int G = gridDim.x;
  const bool shared_rows = a.sH < a.sT;
  FragAddr fa;
  fa.setup(lane);
  LeanStager st;
  st.setup(wid, W, lane, a.sT, a.cv_in, NT, T);
  auto head_of = [&](int idx, int64_t& in_off, int64_t& o_off, int64_t& st_off, int& h) {
    const int u = unit_of(idx, units, shared_rows);
    const int b = u / a.H;
    h = u - b * a.H;
    in_off = b * a.sB + h * a.sH;
    o_off = b * a.oB + h * a.oH;
    st_off = ((int64_t)b * a.H + h) * T;
  };
  if (tid < 32) ((unsigned*)ones)[tid] = (tid & 7) == 0 ? 0x3F80u : 0u;
  // scores of key tile j: rows past T (zeros from the DMA) must not enter the maximum or the sums
  auto mask_last = [&](f32x16& x, int j) {
    if (j == NT - 1 && (T & 31) != 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i)
        if (j * 32 + acc_row(i, half) >= T) x[i] = -INFINITY;
    }
  };

  int u = blockIdx.x;
  int64_t in_off, o_off, st_off;
  int hh;
  head_of(u, in_off, o_off, st_off, hh);
  HeadMaps hm = head_maps(a, hh);
  int bK = 0, bV = 1, bS = 2;
  {
    const i32x4 rk = make_rs(a.k, in_off, a.sT, T, a.cv_in), rv = make_rs(a.v, in_off, a.sT, T, a.cv_in);
    for (int j = 0; j < NT; ++j) {
      st.issue(j, lds0 + bK * IMG, rk, hm.k.bs);
      st.issue(j, lds0 + bV * IMG, rv, hm.v.bs);
    }
  }
  bf16x8 qf[KS];
  load_rows(qf, a.q + in_off, a.sT, active ? wid : 0, T, lane, hm.q);

  for (; u < units; u += G) {
    const int un = u + G;
    const bool has_next = un < units;
    int64_t n_in = 0, n_o = 0, n_st = 0;
    int nh = 0;
    HeadMaps nhm = hm;
    if (has_next) {
      head_of(un, n_in, n_o, n_st, nh);
      nhm = head_maps(a, nh);
    }
    bf16* ob = a.o + o_off;
    float* lseb = a.lse ? a.lse + st_off : nullptr;
    const char* Kimg = smem + bK * IMG;
    const char* Vimg = smem + bV * IMG;
    const i32x4 nrk = make_rs(a.k, n_in, a.sT, T, a.cv_in), nrv = make_rs(a.v, n_in, a.sT, T, a.cv_in);

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // K(n), V(n) and the query rows
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[ks]));
    __syncthreads();                                              // ... of every wave; everybody is done with V(n - 1)

    f32x16 ot[DT];
    zero_acc<DT>(ot);
    float m;
    auto pv_tile = [&](const f32x16& x, int j) {                  // P = exp2(scale x - m), O^T += V^T P (row 80 = row sums)
      float ps[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) ps[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(x[i], a.scale_log2, -m));
      const bf16x8 pb0 = pack8(ps), pb1 = pack8(ps + 8);
      const char* vt_ = Vimg + j * TILE_B;
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        ot[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag(vt_, fa, d, 0), pb0, ot[d], 0, 0, 0);
        ot[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag(vt_, fa, d, 1), pb1, ot[d], 0, 0, 0);
      }
      ot[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag_ones(vt_, fa, 0, ones, lane), pb0, ot[2], 0, 0, 0);
      ot[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag_ones(vt_, fa, 1, ones, lane), pb1, ot[2], 0, 0, 0);
    };
    {
      f32x16 x[C0];
#pragma unroll
      for (int j = 0; j < C0; ++j) {
#pragma unroll
        for (int i = 0; i < 16; ++i) x[j][i] = 0.f;
        const char* kt_ = Kimg + j * TILE_B;
        if (active) {
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) x[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rowfrag(kt_, fa, ks), qf[ks], x[j], 0, 0, 0);
          mask_last(x[j], j);
        }
        if (has_next) st.issue(j, lds0 + bS * IMG, nrk, nhm.k.bs);                        // K(n+1) tiles 0 .. C0-1 -> spare
      }
      float mx = x[0][0];
#pragma unroll
      for (int j = 0; j < C0; ++j) {
#pragma unroll
        for (int i = 0; i < 16; i += 2) mx = fmaxf(fmaxf(mx, x[j][i]), x[j][i + 1]);
      }
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      m = mx * a.scale_log2;
      __syncthreads();                                            // every wave is done with K(n) tiles 0 .. C0-1
      if (C1 == 0 && has_next && active) load_rows(qf, a.q + n_in, a.sT, wid, T, lane, nhm.q);      // (single chunk: the fragments are dead)
#pragma unroll
      for (int j = 0; j < C0; ++j) {
        if (active) pv_tile(x[j], j);
        if (has_next && j + C0 < NT) st.issue(j + C0, lds0 + bS * IMG, nrk, nhm.k.bs);   // K(n+1) tiles C0 ..
        if (has_next) st.issue(j, lds0 + bK * IMG, nrv, nhm.v.bs);                        // V(n+1) tiles 0 .. C0-1 -> K(n)'s buffer
      }
    }
    if constexpr (C1 > 0) {
      f32x16 x[C1];
#pragma unroll
      for (int j = 0; j < C1; ++j) {
#pragma unroll
        for (int i = 0; i < 16; ++i) x[j][i] = 0.f;
        const char* kt_ = Kimg + (C0 + j) * TILE_B;
        if (active) {
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) x[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rowfrag(kt_, fa, ks), qf[ks], x[j], 0, 0, 0);
          mask_last(x[j], C0 + j);
        }
      }
      float mx = x[0][0];
#pragma unroll
      for (int j = 0; j < C1; ++j) {
#pragma unroll
        for (int i = 0; i < 16; i += 2) mx = fmaxf(fmaxf(mx, x[j][i]), x[j][i + 1]);
      }
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m1 = fmaxf(m, mx * a.scale_log2);
      const float alpha = __builtin_amdgcn_exp2f(m - m1);
      m = m1;
#pragma unroll
      for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int i = 0; i < 16; ++i) ot[d][i] *= alpha;
      __syncthreads();                                            // every wave is done with K(n)
      if (has_next && active) load_rows(qf, a.q + n_in, a.sT, wid, T, lane, nhm.q);
#pragma unroll
      for (int j = 0; j < C1; ++j) {
        if (active) pv_tile(x[j], C0 + j);
        if (has_next) st.issue(C0 + j, lds0 + bK * IMG, nrv, nhm.v.bs);                   // V(n+1) tiles C0 ..
      }
    }
    float l = ot[2][8];                                           // row 80 of O^T: the row sum of P
    l = __shfl(l, r, 64);
    {
      const int qi = wid * 32 + r;
      if (active && qi < T) {
        if (half == 0 && lseb) lseb[qi] = m + log2f(l);
        store_rows16(ob + (int64_t)qi * a.oT, ot, 1.0f / l, half, hm.o);
      }
    }
    in_off = n_in; o_off = n_o; st_off = n_st;
    hm = nhm;
    const int t = bK;                                             // K(n+1) is in the spare, V(n+1) in K(n)'s buffer
    bK = bS; bS = bV; bV = t;
  }
}

inline size_t fwd_oss_lds(int nt) { return (size_t)3 * nt * TILE_B + 128; }

inline size_t fwd_os_lds(int nt) {
  return (size_t)3 * nt * TILE_B + 2 * HD * 2 + 128 + (WAVES * 2048 + 1024) + (size_t)(WAVES * 2 * (DT * 32 + kPartPad) + 2 * WAVES * 2) * sizeof(float);
}

inline size_t fwd_lds(int nt) {
  return (size_t)2 * nt * TILE_B + 2 * HD * 2 + (size_t)(WAVES * 2 * (DT * 32 + kPartPad) + 2 * WAVES * 2) * sizeof(float);
}

// shapes these kernels take: head_dim 80, at most 8 full tiles + 2 extra rows, 32-bit buffer offsets
inline bool shape_ok(int T, int hd, int64_t sT, int64_t oT) {
  const int nt = (T + 31) / 32, W = nt < 8 ? nt : 8;
  return hd == HD && nt <= MAXNT && T - 32 * W <= 2 && (int64_t)T * sT * 2 < 0x7FFFFFF0ll && (int64_t)T * oT * 2 < 0x7FFFFFF0ll;
}

static int cu_count() {
  return device_cus();
}

}  // namespace a80

// routing override OCTIC_ROUTE_ATTN_ONLINE: 0 = one-shot softmax where it applies, 1 = online softmax everywhere
// (+16: K / V descriptors with zero records, a timing probe of developer runs)

int attn80_fwd_ok(const AttnArgs& a) { return a80::shape_ok(a.T, a.hd, a.sT, a.oT) ? 1 : 0; }

int attn80_fwd_launch(const AttnArgs& a_, int64_t B, hipStream_t s) {
  using namespace a80;
  AttnArgs a = a_;
  a.dbg = route(OCTIC_ROUTE_ATTN_ONLINE) >> 4;
  const int nt = (a.T + 31) / 32, W = nt < 8 ? nt : 8;
  const int units = (int)(B * a.H), cus = cu_count();
  static DeviceOnce once;
  if (once.first()) {
    (void)hipFuncSetAttribute((const void*)fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)fwd_os_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipGetLastError();
  }
  const int grid = units < cus ? units : cus;
  const bool one_shot = (route(OCTIC_ROUTE_ATTN_ONLINE) & 15) == 0;
  if (one_shot && nt == MAXNT) fwd_os_kernel<<<grid, 512, fwd_os_lds(nt), s>>>(a, units);
  else if (one_shot && a.T <= 32 * nt && nt <= 8) {               // every token inside a tile: the T <= 256 one-shot kernels
    switch (nt) {
#define OSS(n) case n: { static DeviceOnce o##n; if (o##n.first()) { (void)hipFuncSetAttribute((const void*)fwd_oss_kernel<n>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); (void)hipGetLastError(); } \
                 constexpr int wv = n < 4 ? 4 : n;                   /* waves per workgroup */ \
                 const int wgs = cus * (8 / wv);                     /* short sequences: two workgroups per CU (8 waves) */ \
                 fwd_oss_kernel<n><<<units < wgs ? units : wgs, wv * 64, fwd_oss_lds(n), s>>>(a, units); break; }
      OSS(1) OSS(2) OSS(3) OSS(4) OSS(5) OSS(6) OSS(7) OSS(8)
#undef OSS
    }
  } else fwd_kernel<<<grid, W * 64, fwd_lds(nt), s>>>(a, nt, units);
  return launch_status();
}

}  // namespace octic
