// Single-pass softmax-attention BACKWARD for head_dim 80 and T = 257 tokens (ViT-H/14 at 224x224), bf16:
// AttentionD8's core on packed LinearD8 rows (reference octic_vits/d8_layers.py:631-656) and the standard block's fused
// [B,T,3,H,hd] projection (deit/vit.py:38-45, autograd of F.scaled_dot_product_attention).
//
// The round-2 pair (csrc/attention.hip: attn_bwd_dq_kernel + attn_bwd_dkv_kernel) recomputes P in both kernels
// (14 T^2 hd FLOP instead of 10), stages every operand image twice (504 MB instead of 336 per call) and spends 33-51 % of
// a workgroup's life in a prologue with idle matrix cores.  Here P and dS are computed ONCE per (query tile, key tile):
//
//   * one workgroup (8 waves) per (batch, head); wave w OWNS key tile w (keys 32 w .. 32 w + 31): its K and V rows sit in
//     registers as B operands (key on the lane) for the whole head, its dK^T and dV^T accumulate lane-locally
//     (2 x 48 registers) - no cross-wave reduction for dK / dV;
//   * the query side STREAMS: Q, dO and O arrive tile by tile (32 queries) through a three-stage LDS ring filled by
//     LDS-DMA (a80 tile format: the same swizzled 5 KiB images as the forward), two tiles ahead of the one being
//     multiplied; the whole-head images are K (every wave reads all of it for dQ) and V (a wave re-reads its own tile);
//   * per query tile a wave computes  S' = Q K_w^T, dP = dO V_w^T  (10 MFMAs 32x32x16, un-swapped: query on the
//     accumulator row, key on the lane),  P = exp2(S' scale - lse), dS = P (dP - delta)  in registers,
//     dV^T += dO^T P, dK^T += Q^T dS  (12 MFMAs, P / dS straight from the accumulator registers);
//   * dQ needs the contraction over ALL keys: every wave writes its 32 x 32 dS tile to LDS (2 KiB, packed accumulator
//     chunks, bf16 - the same rounding dK sees), one barrier, and the 80 x 32 tile dQ^T = K^T dS^T is cut into ten
//     16 x 16 blocks, each owned by ONE wave which runs the whole key range (8 MFMAs 16x16x32, K^T from the resident K
//     image and dS^T from the eight dS tiles by transposing reads): no partial sums, no f32 traffic, fixed summation order
//     (bitwise reproducible).  (First version of this kernel: 32 x 80 f32 partials per wave parked in LDS and summed by
//     320 threads - 84 KiB of LDS and 19 % of the cycles; tools/a80_bwd_trace.py.)
//   * the 257th KEY (one row that fits no wave) is worked on by the vector unit, 16 lanes per query / one 16-byte chunk
//     per lane: two dot products per query, p and dS of that key, the rank-1 term of dQ (added when a block is stored) and
//     running sums of dK[256] / dV[256] that are combined through LDS after the sweep;
//   * the 257th QUERY is the only real row of the ninth query tile (an ordinary iteration on a mostly empty tile);
//   * delta = <dO, O> is computed one tile ahead from the ring's dO and O tiles (same thread layout, DPP row sums).
// LDS: ring 45 KiB + K image 40 KiB + V image 40 KiB + dS tiles 16 KiB + statistics = 144 KiB.
#include "attn80_common.hpp"

#ifdef A80_TRACE
// developer-only (tools/a80_bwd_trace.py builds with -DA80_TRACE): per workgroup and wave, cycles summed per phase over the
// nine iterations: [0] prologue, [2] S' / dP, [3] softmax terms + dS tile, [7] landed-wait + the barrier, [8] dq stores + DMA
// issue, [4] dV / dK, [5] dQ blocks, [6] delta + key 256, [9] epilogue, [10] total
__device__ unsigned long long g_a80_bwd_trace[1024 * 8 * 16];
extern "C" void* octic_dbg_a80_bwd_trace(void) {
  void* p = nullptr;
  (void)hipGetSymbolAddress(&p, HIP_SYMBOL(g_a80_bwd_trace));
  return p;
}
#define BWT(i)                                                         \
  do {                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                 \
    const unsigned long long now_ = __builtin_readcyclecounter();      \
    bwt_ph[i] += now_ - bwt_last;                                      \
    bwt_last = now_;                                                   \
    __builtin_amdgcn_sched_barrier(0);                                 \
  } while (0)
#else
#define BWT(i) do {} while (0)
#endif

namespace octic {
namespace a80 {

constexpr int BW_T = 257, BW_NT = 9;
constexpr int BW_NSTG = 4;                        // ring stages: tile t is multiplied while t + 1, t + 2 have landed and t + 3 flies
constexpr int BW_STG = 3 * TILE_B;                // one ring stage: Q | dO | O tile
constexpr int BW_RING = BW_NSTG * BW_STG;
constexpr int BW_KIMG = 8 * TILE_B;
constexpr int BW_DST = 2048;                      // a wave's dS tile: [key 32][64 B]
constexpr int BW_DS = 2 * WAVES * BW_DST;         // double-buffered (one barrier per tile)
constexpr int BW_PQ = 2 * 2 * 4 * 64 * 16;        // f32 key quarters of dQ blocks 8 and 9, double-buffered
constexpr int BW_STAT = 2 * 288 * 4;              // lse_s, del_s
constexpr int BW_XK = 384;                        // K row 256 | V row 256 (160 B each)
constexpr int BW_PX = 2 * 2 * 32 * 4;             // p and dS of key 256 for the queries of a tile, double-buffered
constexpr int BW_ACC = 160 * 4;                   // dK[256] | dV[256] running sums
constexpr int BW_LDS = BW_RING + BW_KIMG + BW_DS + BW_PQ + BW_STAT + BW_XK + BW_PX + BW_ACC;
static_assert(BW_LDS <= 160 * 1024, "LDS budget");
static_assert(8 * TILE_B <= BW_DS + BW_PQ, "the V image borrows the dS / quarter buffers during the prologue");

__device__ __forceinline__ float bf_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xFFFF0000u); }
// <a, b> of eight bf16 pairs in f32.  (Not __builtin_amdgcn_fdot2_f32_bf16: a dependent chain of v_dot2c_f32_bf16 returned
// wrong sums on gfx950 / ROCm 7.2 - tools/dbg/dot_test.hip: 2.87 for 9.18 - so the products are plain f32 fmas.)
__device__ __forceinline__ float dot8_bf16(const u32x4 a, const u32x4 b, float acc) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    acc = __builtin_fmaf(bf_lo(a[i]), bf_lo(b[i]), acc);
    acc = __builtin_fmaf(bf_hi(a[i]), bf_hi(b[i]), acc);
  }
  return acc;
}
// four consecutive head elements (half `sub` of group g) at `base[row_off + ...]`: the base stays a uniform pointer and
// everything per-lane is a 32-bit element offset (a per-lane 64-bit row pointer kept live across the tile loop was
// spilled, and every scratch reload drains the DMA queue)
__device__ __forceinline__ void hm_store8_at(bf16* base, int row_off, int g, int sub, const u32x2 v, const HeadMap m) {
  if (m.cv == 0) { *(u32x2*)(base + (row_off + g * 8 + sub * 4)) = v; return; }
  if (g < 8) { *(u32x2_u*)(base + (row_off + hm_off8(m, g) + sub * 4)) = u32x2_u{v[0], v[1]}; return; }
  if (g == 8) {
    const int o = row_off + 2 * sub * m.cv + m.bs + 8;
    *(unsigned*)(base + o) = v[0];
    *(unsigned*)(base + (o + m.cv)) = v[1];
    return;
  }
  *(u32x2_u*)(base + (row_off + (4 + 2 * sub) * m.cv + 2 * m.bs + 16)) = u32x2_u{v[0], v[1]};
}
__device__ __forceinline__ int wid_of() { return __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); }
// chunk j (16 bytes = elements 8 j .. 8 j + 7) of row `row` of a tile image
__device__ __forceinline__ const char* tile_chunk(const char* tile, int row, int j) {
  return j < 8 ? tile + row * 128 + ((j ^ swz(row)) << 4) : tile + TAIL_OFF + row * 32 + (j - 8) * 16;
}

// NKT = key tiles (= waves that own one); XKEY: one more key / query row past them (T = 32 NKT + 1: ViT-H/14's 257).  Round 5:
// T <= 256 runs the same kernel with XKEY = false - every token inside a tile, the last tile possibly partial (its keys past T
// are masked out of P and dS, their rows arrive as zeros), waves past NKT own no key tile and only stage, reduce and compute dQ.
template <int NKT, bool XKEY>
__global__ __launch_bounds__(512) void bwd_kernel(AttnBwdArgs a) {
  constexpr int nt = XKEY ? NKT + 1 : NKT, W = WAVES;
  const int T = a.T;
  const bool owner = wid_of() < NKT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const ring = smem;
  char* const kimg = smem + BW_RING;
  char* const dsb = kimg + BW_KIMG;                  // [2][8 waves][2 KiB] dS tiles (prologue: the V image, with pq)
  char* const pq = dsb + BW_DS;                      // [2][2 blocks][4 quarters][64 lanes] f32x4
  float* const lse_s = (float*)(pq + BW_PQ);
  float* const del_s = lse_s + 288;
  char* const xk = (char*)(del_s + 288);
  float* const px = (float*)(xk + BW_XK);            // [2][p 32 | dS 32]
  float* const acc256 = px + 128;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned ldsK = lds0 + BW_RING, ldsV = ldsK + BW_KIMG;

  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, half = lane >> 5;
  const int bh = unit_of(blockIdx.x, gridDim.x, a.sH < a.sT), b = bh / a.H, h = bh - b * a.H;
  const int64_t in_off = b * a.sB + h * a.sH, o_off = b * a.oB + h * a.oH, g_off = b * a.gB + h * a.gH;
  const int64_t stat_off = ((int64_t)b * a.H + h) * T;
  const HeadMaps hm = head_maps(a, h);

#ifdef A80_TRACE
  unsigned long long bwt_ph[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long bwt_t0 = __builtin_readcyclecounter();
  unsigned long long bwt_last = bwt_t0;
#endif
  FragAddr fa;
  fa.setup(lane);
  LeanStager stq, sto;                               // rows of q / k / v (stride sT, block width cv_in) and of dO / O (oT, cv_out)
  stq.setup(wid, W, lane, a.sT, a.cv_in, nt, T);
  sto.setup(wid, W, lane, a.oT, a.cv_out, nt, T);
  const i32x4 rq = make_rs(a.q, in_off, a.sT, T, a.cv_in), rk = make_rs(a.k, in_off, a.sT, T, a.cv_in);
  const i32x4 rv = make_rs(a.v, in_off, a.sT, T, a.cv_in), rdo = make_rs(a.dout, o_off, a.oT, T, a.cv_out);
  const i32x4 ro = make_rs(a.o, o_off, a.oT, T, a.cv_out);
  auto issue_tile = [&](int t) {                     // Q, dO, O rows of query tile t -> ring stage t % 4
    const int s3 = 3 * (t % BW_NSTG);
    stq.issue(t, lds0, rq, hm.q.bs, s3);
    sto.issue(t, lds0, rdo, hm.o.bs, s3 + 1);
    sto.issue(t, lds0, ro, hm.o.bs, s3 + 2);
  };

  // thread layout of the row-wise delta: 16 lanes per query, lane j < 10 owns 16-byte chunk j
  const int qq = 4 * wid + (lane >> 4), j16 = lane & 15;
  const bool jon = j16 < 10;
  const int jc = jon ? j16 : 9;
  float* const dl = a.delta + stat_off;              // uniform base, 32-bit per-lane index
  // delta of tile t for query qq from the ring's dO and O tiles
  auto delta_of = [&](int t) {
    const char* st_ = ring + (t % BW_NSTG) * BW_STG;
    const u32x4 dc = *(const u32x4*)tile_chunk(st_ + TILE_B, qq, jc), oc = *(const u32x4*)tile_chunk(st_ + 2 * TILE_B, qq, jc);
    const float d = sum16_from8(sum8(jon ? dot8_bf16(dc, oc, 0.f) : 0.f));
    const int q = 32 * t + qq;
    if (j16 == 0) {
      del_s[q] = d;                                  // rows past T are zero rows: d = 0
      if (q < T) dl[q] = d;
    }
  };

  // ---- transposing-read geometry of the 16x16x32 products (natural k order: lane group kq holds k = 8 kq + e).
  // A 16 x 32 operand block read out of a 32-row tile image: rows 8 kq + q4 (+ 4), columns 16 db + 4 p .. + 3.
  const int kq = lane >> 4, ti = lane & 15, q4 = ti >> 2, pp = ti & 3;
  const int trow = 8 * kq + q4;
  auto blk_lo = [&](int db) { return db < 4 ? trow * 128 + (((2 * db + (pp >> 1)) ^ swz(trow)) << 4) + (pp & 1) * 8 : TAIL_OFF + trow * 32 + pp * 8; };
  auto blk_hi = [&](int db) { return db < 4 ? (trow + 4) * 128 + (((2 * db + (pp >> 1)) ^ swz(trow + 4)) << 4) + (pp & 1) * 8 : TAIL_OFF + (trow + 4) * 32 + pp * 8; };
  auto tr8 = [&](const char* lo, const char* hi) {
    const s16x4 u = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)lo);
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)hi);
    const s16x8 w = {u[0], u[1], u[2], u[3], v[0], v[1], v[2], v[3]};
    return __builtin_bit_cast(bf16x8, w);
  };

  // ---- key 256 (the row that fits no wave) on the matrix pipe with 16x16x32 products, in two parts:
  //   A (two waves, one query block of 16 each; a tile AHEAD): s'[q] = <Q[q], K[256]>, dp[q] = <dO[q], V[256]> (three
  //     k-steps each, B = the key row in every column), p and dS of that key -> px (dS is also the rank-1 term of dQ);
  //   B (five waves, one d-block of 16 each): dV[256] += dO^T p, dK[256] += Q^T dS (B = p / dS in every column) -> acc256.
  auto key256_a = [&](int u, int qb2) {
    const char* qt_ = ring + (u % BW_NSTG) * BW_STG;
    const char* dt_ = qt_ + TILE_B;
    f32x4 aS = {0, 0, 0, 0}, aD = {0, 0, 0, 0};
#pragma unroll
    for (int ks3 = 0; ks3 < 3; ++ks3) {
      const int c = 4 * ks3 + kq;                    // 16-byte chunk of the head vector (10, 11: beyond head_dim)
      u32x4 fq = {0, 0, 0, 0}, fd = {0, 0, 0, 0}, fk = {0, 0, 0, 0}, fv = {0, 0, 0, 0};
      if (c < 10) {
        fq = *(const u32x4*)tile_chunk(qt_, 16 * qb2 + ti, c);
        fd = *(const u32x4*)tile_chunk(dt_, 16 * qb2 + ti, c);
        fk = *(const u32x4*)(xk + c * 16);
        fv = *(const u32x4*)(xk + 160 + c * 16);
      }
      aS = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fq), __builtin_bit_cast(bf16x8, fk), aS, 0, 0, 0);
      aD = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fd), __builtin_bit_cast(bf16x8, fv), aD, 0, 0, 0);
    }
    const int q0 = 16 * qb2 + 4 * kq;                // lane (n, rq = kq): queries q0 .. q0 + 3 of the tile (every n alike)
    const f32x4 l4 = *(const f32x4*)(lse_s + 32 * u + q0), d4 = *(const f32x4*)(del_s + 32 * u + q0);
    f32x4 p4, s4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      p4[e] = __builtin_amdgcn_exp2f(__builtin_fmaf(aS[e], a.scale_log2, -l4[e]));
      s4[e] = p4[e] * (aD[e] - d4[e]);
    }
    if (ti == 0) {
      float* pb_ = px + (u & 1) * 64;
      *(f32x4*)(pb_ + q0) = p4;
      *(f32x4*)(pb_ + 32 + q0) = s4;
    }
  };
  auto key256_b = [&](int t, int db) {
    const char* qt_ = ring + (t % BW_NSTG) * BW_STG;
    const char* dt_ = qt_ + TILE_B;
    const float* pb_ = px + (t & 1) * 64 + 8 * kq;   // B operands: lane (n, kq) holds queries 8 kq + e, the same in every column n
    const f32x4 a0 = *(const f32x4*)pb_, a1 = *(const f32x4*)(pb_ + 4), b0 = *(const f32x4*)(pb_ + 32), b1 = *(const f32x4*)(pb_ + 36);
    const float pb[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
    const float sb[8] = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
    const int lo = blk_lo(db), hi = blk_hi(db);
    const f32x4 z = {0, 0, 0, 0};
    const f32x4 dv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr8(dt_ + lo, dt_ + hi), pack8(pb), z, 0, 0, 0);
    const f32x4 dk = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr8(qt_ + lo, qt_ + hi), pack8(sb), z, 0, 0, 0);
    if (ti == 0) {                                   // column 0 of the block: elements 16 db + 4 kq .. + 3
      float* pk = acc256 + 16 * db + 4 * kq;
      *(f32x4*)pk = *(const f32x4*)pk + dk;
      *(f32x4*)(pk + 80) = *(const f32x4*)(pk + 80) + dv;
    }
  };

  // ---------------------------------------------------------------------------------------------------- prologue
  // K image, V image (borrowing the dS / quarter buffers) and the first three ring tiles by LDS-DMA; rows 256 of K and V
#pragma unroll
  for (int jt = 0; jt < NKT; ++jt) {
    stq.issue(jt, ldsK, rk, hm.k.bs);
    stq.issue(jt, ldsV, rv, hm.v.bs);
  }
  issue_tile(0);
  if (1 < nt) issue_tile(1);
  if (2 < nt) issue_tile(2);
  {
    u32x4 xrow = {0, 0, 0, 0};
    if constexpr (XKEY) {
      if (wid == 1 && lane < 10) xrow = hm_load16(a.k + in_off + (int64_t)(32 * NKT) * a.sT, lane, hm.k);
      if (wid == 2 && lane < 10) xrow = hm_load16(a.v + in_off + (int64_t)(32 * NKT) * a.sT, lane, hm.v);
    }
    for (int t = tid; t < 288; t += 512) lse_s[t] = t < T ? a.lse[stat_off + t] : INFINITY;   // padded queries: P = 0
    if constexpr (XKEY) {
      if (tid < 160) acc256[tid] = 0.f;
      if ((wid == 1 || wid == 2) && lane < 10) *(u32x4*)(xk + (wid - 1) * 160 + lane * 16) = xrow;
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  // The V rows of the own key tile (B operand of dP) live in registers; the K rows (B operand of S') are re-read from the
  // resident K image every tile (both in registers: 13-96 spilled VGPRs, and every scratch reload inside the tile loop is
  // an `s_waitcnt vmcnt(0)`, i.e. a drain of the DMA queue)
  bf16x8 vf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) vf[ks] = rowfrag(dsb + wid * TILE_B, fa, ks);
  const char* const kt_ = kimg + wid * TILE_B;
  delta_of(0);
  if (1 < nt) delta_of(1);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                      // the V image is dead; delta of tiles 0, 1 is visible
  if constexpr (XKEY) {
    if (wid < 2) key256_a(0, wid);                   // (visible after the first tile's barrier)
  }
  // keys of this wave's tile past T (partial last tile, XKEY = false): out of P and dS
  const float kmask = (XKEY || wid * 32 + r < T) ? 1.0f : 0.0f;

  BWT(0);
  f32x16 dkt[DT], dvt[DT];
  zero_acc<DT>(dkt);
  zero_acc<DT>(dvt);
  bf16* const dqb = a.dq + g_off;
  const int gT = (int)a.gT;

  // ---- dQ^T = K^T dS^T in ten blocks of 16 d x 16 queries (block bi: d-block bi % 5, query block bi / 5).  Wave w owns
  // block w over the whole key range (8 k-steps); blocks 8 and 9 are cut into four key quarters of 2 k-steps, wave w takes
  // quarter w >> 1 of block 8 + (w & 1): ten k-steps per wave.  The quarters meet in LDS (f32, 1 KiB each) and are
  // summed in quarter order by waves 0 / 1 after the next barrier.
  //   A = K^T block out of K tile kt; B = dS^T block out of wave kt's dS tile: rows 8 kq + q4 (+ 4), 32-byte half
  //   qb ^ (row >> 3 & 1), bytes 8 p ..
  const int bi0 = wid, bi1 = 8 + (wid & 1);
  const int ka_lo0 = blk_lo(bi0 % 5), ka_hi0 = blk_hi(bi0 % 5), kb0 = trow * 64 + (((bi0 / 5) ^ (kq & 1)) * 32) + pp * 8;
  const int ka_lo1 = blk_lo(bi1 % 5), ka_hi1 = blk_hi(bi1 % 5), kb1 = trow * 64 + ((1 ^ (kq & 1)) * 32) + pp * 8;
  // a block's accumulator: lane (n = ti, rq = kq) holds elements 16 db + 4 rq .. + 3 of query position 16 qb + n of the tile
  const int qpos16 = 4 * (ti >> 3) + (ti & 3) + 8 * ((ti & 7) >> 2);       // + 16 qb
  u32x2 pend = {0, 0};                               // dq piece of the previous tile (own block), stored one tile late
  auto rank1 = [&](f32x4& acc, int t, int db, int qb) {   // dS of key 256 for this query: rank-1 term of dQ
    const float sk = px[(t & 1) * 64 + 32 + 16 * qb + qpos16];
    const u32x2 k2 = *(const u32x2*)(xk + (16 * db + 4 * kq) * 2);
    acc[0] += sk * bf_lo(k2[0]); acc[1] += sk * bf_hi(k2[0]); acc[2] += sk * bf_lo(k2[1]); acc[3] += sk * bf_hi(k2[1]);
  };
  auto pack_scaled = [&](const f32x4 v) {
    const bf16x4 o = {(bf16)(v[0] * a.scale), (bf16)(v[1] * a.scale), (bf16)(v[2] * a.scale), (bf16)(v[3] * a.scale)};
    return __builtin_bit_cast(u32x2, o);
  };
  auto store_pending = [&](int tprev) {              // dq of tile tprev: the own block, and (waves 0 / 1) blocks 8 / 9
    if (tprev < 0) return;
    const int q0 = 32 * tprev + 16 * (bi0 / 5) + qpos16, q1 = 32 * tprev + 16 + qpos16;   // (recomputed: two registers less)
    if (q0 < T) {
      const int d0 = 16 * (bi0 % 5) + 4 * kq;
      hm_store8_at(dqb, q0 * gT, d0 >> 3, (d0 >> 2) & 1, pend, hm.q);
    }
    if (wid < 2 && q1 < T) {                         // the four key quarters, in quarter order
      const f32x4* q4p = (const f32x4*)pq + (tprev & 1) * 512 + wid * 256 + lane;
      const f32x4 sum = ((q4p[0] + q4p[64]) + q4p[128]) + q4p[192];
      const int d0 = 16 * (3 + wid) + 4 * kq;
      hm_store8_at(dqb, q1 * gT, d0 >> 3, (d0 >> 2) & 1, pack_scaled(sum), hm.q);
    }
  };

#ifndef BW_HI_PRIO
#define BW_HI_PRIO 0      // A/B on MI355X (tools/ab_attn_bwd.py): 157.8 us without, 162.3 us with the static priority
#endif
#if BW_HI_PRIO
  // the second-dispatched half of the workgroup loses the vector-issue arbitration on its SIMD to its older partner in every
  // phase (MI355X guide, "Two waves per SIMD", item 4): one static s_setprio 1 for waves 4-7, no per-phase flips
  if (wid >= 4) __builtin_amdgcn_s_setprio(1);
#endif
  for (int t = 0; t < nt; ++t) {
    const char* qt_ = ring + (t % BW_NSTG) * BW_STG;
    const char* dt_ = qt_ + TILE_B;
    char* const mydst = dsb + (t & 1) * (WAVES * BW_DST) + wid * BW_DST;

    // ---- S' and dP (query on the accumulator row, key on the lane)
    bf16x8 p0, p1, s0, s1;
    if (owner) {
    f32x16 x, dp;
#pragma unroll
    for (int i = 0; i < 16; ++i) { x[i] = 0.f; dp[i] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      x = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rowfrag(qt_, fa, ks), rowfrag(kt_, fa, ks), x, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rowfrag(dt_, fa, ks), vf[ks], dp, 0, 0, 0);
    }
    BWT(2);
    float ps[16], ds[16];
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int q0 = t * 32 + 8 * g4 + 4 * half;            // accumulator rows 4 g4 .. 4 g4 + 3 are queries q0 .. q0 + 3
      const f32x4 l4 = *(const f32x4*)(lse_s + q0), d4 = *(const f32x4*)(del_s + q0);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = 4 * g4 + e;
        float p = __builtin_amdgcn_exp2f(__builtin_fmaf(x[i], a.scale_log2, -l4[e]));
        if constexpr (!XKEY) p *= kmask;
        ps[i] = p;
        ds[i] = p * (dp[i] - d4[e]);
      }
    }
    p0 = pack8(ps); p1 = pack8(ps + 8); s0 = pack8(ds); s1 = pack8(ds + 8);
    // dS tile of this wave: row = key (lane r); its 32-byte half s (queries 16 s + {4 half + 0..3, + 8}) sits at half
    // s ^ (r >> 3 & 1), so that the transposing reads of rows 0-3 and 8-11 hit disjoint banks
    {
      const int sw = (r >> 3) & 1;
      *(bf16x8*)(mydst + r * 64 + (sw * 32) + half * 16) = s0;
      *(bf16x8*)(mydst + r * 64 + ((sw ^ 1) * 32) + half * 16) = s1;
    }
    }
    BWT(3);
    // ---- [b_t] the ONE barrier of a tile.  Before it: this wave's dS tile, its share of tile t + 2 (DMA issued a tile
    // ago), delta of tile t + 1 and px of tile t are complete; after it: all of that is visible, and every wave is done
    // with tile t - 1 (its ring stage, dS buffer, quarters, px buffer may be overwritten)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    BWT(7);
    store_pending(t - 1);
    if (t + 3 < nt) issue_tile(t + 3);
    BWT(8);
    if (owner) {
#pragma unroll
    for (int d = 0; d < DT; ++d) {
      dvt[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag(dt_, fa, d, 0), p0, dvt[d], 0, 0, 0);
      dvt[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag(dt_, fa, d, 1), p1, dvt[d], 0, 0, 0);
      dkt[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag(qt_, fa, d, 0), s0, dkt[d], 0, 0, 0);
      dkt[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag(qt_, fa, d, 1), s1, dkt[d], 0, 0, 0);
    }
    }
    BWT(4);
    // ---- dQ^T: the own block over all keys (two accumulation chains), a key quarter of block 8 / 9
    {
      const char* dsr = dsb + (t & 1) * (WAVES * BW_DST);
      f32x4 e0 = {0, 0, 0, 0}, e1 = {0, 0, 0, 0};
#pragma unroll 2
      for (int k = 0; k < NKT; k += 2) {
        const char* kp = kimg + k * TILE_B;
        const char* sp = dsr + k * BW_DST + kb0;
        e0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr8(kp + ka_lo0, kp + ka_hi0), tr8(sp, sp + 256), e0, 0, 0, 0);
        if (k + 1 < NKT)
          e1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr8(kp + TILE_B + ka_lo0, kp + TILE_B + ka_hi0), tr8(sp + BW_DST, sp + BW_DST + 256), e1, 0, 0, 0);
      }
      f32x4 own = e0 + e1;
      if constexpr (XKEY) rank1(own, t, bi0 % 5, bi0 / 5);
      pend = pack_scaled(own);
      const int k2 = 2 * (wid >> 1);
      const char* kp = kimg + k2 * TILE_B;
      const char* sp = dsr + k2 * BW_DST + kb1;
      const f32x4 z = {0, 0, 0, 0};
      f32x4 part = z;
      if (k2 < NKT) part = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr8(kp + ka_lo1, kp + ka_hi1), tr8(sp, sp + 256), z, 0, 0, 0);
      if (k2 + 1 < NKT)
        part = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr8(kp + TILE_B + ka_lo1, kp + TILE_B + ka_hi1), tr8(sp + BW_DST, sp + BW_DST + 256), part, 0, 0, 0);
      if constexpr (XKEY) {
        if (wid < 2) rank1(part, t, bi1 % 5, 1);
      }
      ((f32x4*)pq)[(t & 1) * 512 + (wid & 1) * 256 + (wid >> 1) * 64 + lane] = part;
    }
    BWT(5);
    // ---- row-wise work for later tiles: delta two tiles ahead; key 256: part A one tile ahead, part B for this tile
    if (t + 2 < nt) delta_of(t + 2);
    if constexpr (XKEY) {
      const int role = (wid - 2 * t) & 7;            // rotates over the waves: 0, 1 = part A (query block), 2..6 = part B (d-block)
      if (role < 2) { if (t + 1 < nt) key256_a(t + 1, role); }
      else if (role < 7) key256_b(t, role - 2);
    }
    BWT(6);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                      // the last tile's dQ quarters and acc256 are complete
  store_pending(nt - 1);

  // ---------------------------------------------------------------------------------------------------- epilogue
  {
    const int ki = wid * 32 + r;
    if (owner && ki < T) {
      store_rows16(a.dk + g_off + (int64_t)ki * a.gT, dkt, a.scale, half, hm.k);
      store_rows16(a.dv + g_off + (int64_t)ki * a.gT, dvt, 1.0f, half, hm.v);
    }
  }
  if constexpr (XKEY) {
    if (tid < 160) {                                 // dK[256] | dV[256] (acc256: one wave per d-block and tile, in tile order)
      const bool isv = tid >= 80;
      const int e = isv ? tid - 80 : tid;
      bf16* row = (isv ? a.dv : a.dk) + g_off + (int64_t)(32 * NKT) * a.gT;
      row[hm_elem(e, isv ? hm.v : hm.k)] = (bf16)(isv ? acc256[tid] : acc256[tid] * a.scale);
    }
  }
#ifdef A80_TRACE
  BWT(9);
  bwt_ph[10] = bwt_last - bwt_t0;
  if (lane == 0 && blockIdx.x < 1024) {
#pragma unroll
    for (int i = 0; i < 11; ++i) g_a80_bwd_trace[(blockIdx.x * 8 + wid) * 16 + i] = bwt_ph[i];
  }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// Short sequences (T <= 32 NKT <= 64 tokens: the 96 x 96 local crops of DINOv2's multi-crop recipe, 37 tokens at patch 16).
// The same single pass with a workgroup of NKT waves (wave w owns key tile w), EVERYTHING resident: the Q, dO, O, K and V
// images of the head land once by LDS-DMA (no ring), delta of all query tiles is formed in the prologue, the dS tiles take over
// the O image, dQ^T's ten 16 x 16 blocks per query tile are dealt round-robin to the waves over the whole key range (no
// quarters, no extra-row machinery).  50.5 KiB of LDS at NKT = 2: three workgroups per CU.  (V rows and the delta chunks
// straight from global memory - 38.5 KiB, four workgroups per CU - timed the same on strided rows, 77 against 76 us, and lost on
// packed rows, whose 16-byte chunks straddle irrep pieces: the DMA stager's piece addressing is the cheaper path there.)
template <int NKT>
__global__ __launch_bounds__(64 * NKT) void bwd_small_kernel(AttnBwdArgs a) {
  constexpr int nt = NKT, W = NKT;
  constexpr int IMG = NKT * TILE_B;
  const int T = a.T;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const qimg = smem;
  char* const doimg = qimg + IMG;
  char* const kimg = doimg + IMG;
  char* const vimg = kimg + IMG;
  char* const oimg = vimg + IMG;                     // prologue: the O image (delta); afterwards the dS tiles [2][W][2 KiB]
  char* const dsb = oimg;
  float* const lse_s = (float*)(oimg + IMG);
  float* const del_s = lse_s + 32 * NKT;
  static_assert(2 * NKT * BW_DST <= IMG, "the dS tiles fit into the O image");
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned ldsQ = lds0, ldsDO = lds0 + IMG, ldsK = lds0 + 2 * IMG, ldsV = lds0 + 3 * IMG, ldsO = lds0 + 4 * IMG;

  const int tid = threadIdx.x, lane = tid & 63, wid = wid_of();
  const int r = lane & 31, half = lane >> 5;
  const int bh = unit_of(blockIdx.x, gridDim.x, a.sH < a.sT), b = bh / a.H, h = bh - b * a.H;
  const int64_t in_off = b * a.sB + h * a.sH, o_off = b * a.oB + h * a.oH, g_off = b * a.gB + h * a.gH;
  const int64_t stat_off = ((int64_t)b * a.H + h) * T;
  const HeadMaps hm = head_maps(a, h);
  FragAddr fa;
  fa.setup(lane);

  // ---- prologue: every image of the head by LDS-DMA, the row statistics
  {
    const i32x4 rq = make_rs(a.q, in_off, a.sT, T, a.cv_in), rk = make_rs(a.k, in_off, a.sT, T, a.cv_in);
    const i32x4 rv = make_rs(a.v, in_off, a.sT, T, a.cv_in), rdo = make_rs(a.dout, o_off, a.oT, T, a.cv_out);
    const i32x4 ro = make_rs(a.o, o_off, a.oT, T, a.cv_out);
#pragma unroll
    for (int jt = 0; jt < NKT; ++jt) {
      Stager::issue(wid, W, lane, jt, nt, T, ldsK, ldsV, rk, rv, hm.k.bs, hm.v.bs, a.sT, a.sT, a.cv_in, a.cv_in);
      Stager::issue(wid, W, lane, jt, nt, T, ldsDO, ldsO, rdo, ro, hm.o.bs, hm.o.bs, a.oT, a.oT, a.cv_out, a.cv_out);
      Stager::issue(wid, W, lane, jt, nt, T, ldsQ, ldsQ, rq, rq, hm.q.bs, hm.q.bs, a.sT, a.sT, a.cv_in, a.cv_in, -1, 1);
    }
    for (int t = tid; t < 32 * NKT; t += 64 * NKT) lse_s[t] = t < T ? a.lse[stat_off + t] : INFINITY;   // padded queries: P = 0
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  bf16x8 vf[KS];                                     // V rows of the own key tile: B operand of dP
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) vf[ks] = rowfrag(vimg + wid * TILE_B, fa, ks);
  {
    // delta = <dO, O> per query: 16 lanes per query (lane j < 10 owns 16-byte chunk j), four queries per wave and pass
    const int j16 = lane & 15;
    const bool jon = j16 < 10;
    const int jc = jon ? j16 : 9;
    float* const dl = a.delta + stat_off;
    for (int q = 4 * wid + (lane >> 4); q < 32 * NKT; q += 4 * W) {
      const int t = q >> 5, qq = q & 31;
      const u32x4 dc = *(const u32x4*)tile_chunk(doimg + t * TILE_B, qq, jc), oc = *(const u32x4*)tile_chunk(oimg + t * TILE_B, qq, jc);
      const float d = sum16_from8(sum8(jon ? dot8_bf16(dc, oc, 0.f) : 0.f));
      if (j16 == 0) {
        del_s[q] = d;                                // rows past T are zero rows: d = 0
        if (q < T) dl[q] = d;
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                      // the O image is dead (dS tiles from here on); delta is visible

  const char* const kt_ = kimg + wid * TILE_B;
  const float kmask = (wid * 32 + r < T) ? 1.0f : 0.0f;      // keys of the partial last tile past T: out of P and dS
  f32x16 dkt[DT], dvt[DT];
  zero_acc<DT>(dkt);
  zero_acc<DT>(dvt);
  bf16* const dqb = a.dq + g_off;
  const int gT = (int)a.gT;
  // transposing-read geometry of the 16 x 16 x 32 products (as in bwd_kernel)
  const int kq = lane >> 4, ti = lane & 15, q4 = ti >> 2, pp = ti & 3;
  const int trow = 8 * kq + q4;
  auto blk_lo = [&](int db) { return db < 4 ? trow * 128 + (((2 * db + (pp >> 1)) ^ swz(trow)) << 4) + (pp & 1) * 8 : TAIL_OFF + trow * 32 + pp * 8; };
  auto blk_hi = [&](int db) { return db < 4 ? (trow + 4) * 128 + (((2 * db + (pp >> 1)) ^ swz(trow + 4)) << 4) + (pp & 1) * 8 : TAIL_OFF + (trow + 4) * 32 + pp * 8; };
  auto tr8 = [&](const char* lo, const char* hi) {
    const s16x4 u = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)lo);
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)hi);
    const s16x8 w = {u[0], u[1], u[2], u[3], v[0], v[1], v[2], v[3]};
    return __builtin_bit_cast(bf16x8, w);
  };
  const int qpos16 = 4 * (ti >> 3) + (ti & 3) + 8 * ((ti & 7) >> 2);

  for (int t = 0; t < nt; ++t) {
    const char* qt_ = qimg + t * TILE_B;
    const char* dt_ = doimg + t * TILE_B;
    char* const mydst = dsb + (t & 1) * (W * BW_DST) + wid * BW_DST;
    // ---- S' and dP (query on the accumulator row, key on the lane), P and dS in registers, the dS tile to LDS
    bf16x8 p0, p1, s0, s1;
    {
      f32x16 x, dp;
#pragma unroll
      for (int i = 0; i < 16; ++i) { x[i] = 0.f; dp[i] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        x = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rowfrag(qt_, fa, ks), rowfrag(kt_, fa, ks), x, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rowfrag(dt_, fa, ks), vf[ks], dp, 0, 0, 0);
      }
      float ps[16], ds[16];
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int q0 = t * 32 + 8 * g4 + 4 * half;
        const f32x4 l4 = *(const f32x4*)(lse_s + q0), d4 = *(const f32x4*)(del_s + q0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int i = 4 * g4 + e;
          const float pe = __builtin_amdgcn_exp2f(__builtin_fmaf(x[i], a.scale_log2, -l4[e])) * kmask;
          ps[i] = pe;
          ds[i] = pe * (dp[i] - d4[e]);
        }
      }
      p0 = pack8(ps); p1 = pack8(ps + 8); s0 = pack8(ds); s1 = pack8(ds + 8);
      const int sw = (r >> 3) & 1;
      *(bf16x8*)(mydst + r * 64 + (sw * 32) + half * 16) = s0;
      *(bf16x8*)(mydst + r * 64 + ((sw ^ 1) * 32) + half * 16) = s1;
    }
    // ---- the one barrier of a tile: every wave's dS tile is visible; everyone is done with tile t - 1 (its dS buffer).
    // (No vmcnt: nothing is loaded inside the loop, and the dq stores of the previous tile may stay in flight.)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int d = 0; d < DT; ++d) {
      dvt[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag(dt_, fa, d, 0), p0, dvt[d], 0, 0, 0);
      dvt[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag(dt_, fa, d, 1), p1, dvt[d], 0, 0, 0);
      dkt[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag(qt_, fa, d, 0), s0, dkt[d], 0, 0, 0);
      dkt[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag(qt_, fa, d, 1), s1, dkt[d], 0, 0, 0);
    }
    // ---- dQ^T = K^T dS^T: blocks wid, wid + W, ... of the ten (d-block bi % 5, query block bi / 5), whole key range each
    const char* dsr = dsb + (t & 1) * (W * BW_DST);
    for (int bi = wid; bi < 10; bi += W) {
      const int db = bi % 5, qb = bi / 5;
      const int lo = blk_lo(db), hi = blk_hi(db), kb = trow * 64 + ((qb ^ (kq & 1)) * 32) + pp * 8;
      f32x4 e = {0, 0, 0, 0};
#pragma unroll
      for (int k = 0; k < NKT; ++k) {
        const char* kp = kimg + k * TILE_B;
        const char* sp = dsr + k * BW_DST + kb;
        e = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr8(kp + lo, kp + hi), tr8(sp, sp + 256), e, 0, 0, 0);
      }
      const int q = 32 * t + 16 * qb + qpos16;
      if (q < T) {
        const int d0 = 16 * db + 4 * kq;
        const bf16x4 o4 = {(bf16)(e[0] * a.scale), (bf16)(e[1] * a.scale), (bf16)(e[2] * a.scale), (bf16)(e[3] * a.scale)};
        hm_store8_at(dqb, q * gT, d0 >> 3, (d0 >> 2) & 1, __builtin_bit_cast(u32x2, o4), hm.q);
      }
    }
  }
  // ---- epilogue: dK, dV rows of the own key tile
  {
    const int ki = wid * 32 + r;
    if (ki < T) {
      store_rows16(a.dk + g_off + (int64_t)ki * a.gT, dkt, a.scale, half, hm.k);
      store_rows16(a.dv + g_off + (int64_t)ki * a.gT, dvt, 1.0f, half, hm.v);
    }
  }
}

}  // namespace a80

// routing override OCTIC_ROUTE_ATTN_BWD_PAIR: 1 = the round-2 dq + dkv pair for every shape

// shapes of the single-pass backward: head_dim 80; 257 tokens (8 key tiles + one extra row), 193 .. 256 tokens (7 - 8 key
// tiles, every token inside one: DINOv2 ViT-H/16's 197) or <= 64 tokens (bwd_small_kernel: the 37-token local crops); 65 .. 192
// tokens would leave most of the eight waves without a key tile (and those instantiations spill) and stay on the dq + dkv
// pair; 32-bit offsets
int attn80_bwd_ok(const AttnBwdArgs& a) {
  using namespace a80;
  return (!route(OCTIC_ROUTE_ATTN_BWD_PAIR) && a.hd == HD && (a.T == BW_T || (a.T > 192 && a.T <= 256) || a.T <= 64) &&
          (int64_t)a.T * a.sT * 2 < 0x7FFFFFF0ll && (int64_t)a.T * a.oT * 2 < 0x7FFFFFF0ll &&
          (int64_t)a.T * a.gT * 2 < 0x7FFFFFF0ll) ? 1 : 0;
}

int attn80_bwd_launch(const AttnBwdArgs& a, int64_t B, hipStream_t s) {
  using namespace a80;
  static DeviceOnce once;
  if (once.first()) {
    (void)hipFuncSetAttribute((const void*)bwd_kernel<8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)bwd_kernel<8, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)bwd_kernel<7, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipGetLastError();
  }
  const int grid = (int)(B * a.H);
  if (a.T <= 32) { bwd_small_kernel<1><<<grid, 64, 5 * TILE_B + 2 * 32 * 4, s>>>(a); return launch_status(); }
  if (a.T <= 64) { bwd_small_kernel<2><<<grid, 128, 10 * TILE_B + 2 * 64 * 4, s>>>(a); return launch_status(); }
  if (a.T == BW_T) bwd_kernel<8, true><<<grid, 512, BW_LDS, s>>>(a);
  else if (a.T > 224) bwd_kernel<8, false><<<grid, 512, BW_LDS, s>>>(a);
  else bwd_kernel<7, false><<<grid, 512, BW_LDS, s>>>(a);
  return launch_status();
}

}  // namespace octic
