// LinearD8 fused with the D8 isotypic GELU (MlpD8: fc1 -> TritonGeluD8, d8_layers.py:215-247 with d8_gelu.py:104-331).
//
//   forward  : h = fc1(x) (+ A1 bias), y = F(gelu(F^-1 h))        -> writes h (kept for the backward) and y
//   backward : g = dy . W2 (input gradient of fc2), dh = F(gelu'(F^-1 h) * F^-1 g)   -> reads h, writes dh
//
// The GELU mixes the 8 isotypic components of one hidden channel j (A1[j], A2[j], B1[j], B2[j], E[0,j], E[1,j],
// E[0,c+j], E[1,c+j]), which the per-irrep GEMM kernels of csrc/gemm.hip produce in five different workgroups.  This
// kernel uses a CHANNEL-SLICED decomposition instead: a wave owns 16 token rows and walks slices of 16 hidden channels;
// for a slice it runs all eight sub-GEMMs (four one-dimensional irreps with K = c_in, the two E rows against the two
// 16-row blocks j and c+j of W_E with K = 2 c_in), so that with the operands swapped (W rows = MFMA "A", tokens = MFMA
// "B") lane (fr, kg) ends up holding, in eight accumulators, all eight components of channels 4 kg .. 4 kg + 3 of token
// fr: the 8-point butterflies, GELU and the inverse butterflies run in registers, nothing is exchanged between lanes.
//   * X is stationary: the MFMA operand fragments of the wave's 16 tokens for ALL irreps (K = 8 c_in per token, 160
//     VGPRs at ViT-H) are loaded once from HBM and reused for every channel slice;
//   * W streams through a 3-stage LDS-DMA ring shared by the 4 waves of a workgroup (stages alternate "the four 1-D
//     irrep blocks of a slice" / "the two E blocks of a slice", 16 rows x 64-element K tiles of 128-byte rows, XOR
//     swizzle on the source side as in csrc/gemm.hip);
//   * work = (64-token group, slice) units dealt to <= 512 workgroups (two per CU) in equal contiguous ranges, so
//     M = 16 448 = 257 x 64 does not leave a straggler round;
//   * pairs of lanes (kg, kg ^ 1) exchange halves with v_permlane16_swap so that every global access moves 16 bytes
//     (8 channels of one component) instead of 8.
// Traffic per call at ViT-H, B = 64: X 42 MB + h 168 MB + y (dh) 168 MB = 380 MB instead of 212 + 337 (505) MB for
// the separate GEMM and GELU passes.  bf16 only (the f32 path keeps the separate exact kernels).
#include <type_traits>
#include "octic_common.hpp"

namespace octic {

#ifndef GG_ABL
#define GG_ABL 0     // developer ablations: 1 = no global stores, 2 = identity instead of GELU, 4 = no MFMAs
#endif
constexpr int GG_STAGE = 24 * 1024;           // max(1-D stage: 4 irreps x 3 k-tiles, E stage: 2 blocks x 5 k-tiles) x 2 KiB
constexpr int GG_NST = 3;

struct GgArgs {
  const bf16* x;      // packed rows [M, 8 cin]   (forward: LN output; backward: cotangent of fc2's output)
  int64_t ldx;
  const bf16* w;      // flat [4 x (cout x cin) | 2 cout x 2 cin]: forward fc1's bf16 weights, backward fc2's transposed (scaled) copies
  const float* bias;  // [cout] A1 bias (forward) or null
  bf16* h;            // packed [M, 8 cout]: forward writes the pre-activation, backward reads it
  bf16* y;            // packed [M, 8 cout]: forward: activation; backward: dh
  int64_t ldh;
  int M, cin, cout;
  int groups;         // ceil(M / 64)
  int nslices;        // cout / 16
};

__device__ inline float gg_erf(float x) {       // Abramowitz-Stegun 7.1.26, |err| <= 1.5e-7 (see csrc/dense_gemm.hip)
  const float ax = fabsf(x);
  const float t = __frcp_rn(1.0f + 0.3275911f * ax);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  return copysignf(1.0f - poly * __expf(-ax * ax), x);
}
__device__ inline float gg_gelu(float r) { return 0.5f * r * (1.0f + gg_erf(r * kSqrt1Over2)); }
__device__ inline float gg_gelu_grad(float r) {
  return 0.5f * (1.0f + gg_erf(r * kSqrt1Over2)) + r * kInvSqrt2Pi * __expf(-0.5f * r * r);
}

__device__ inline unsigned gg_pack2(float a, float b) {
  const bf16 x = (bf16)a, y = (bf16)b;
  return (unsigned)__builtin_bit_cast(unsigned short, x) | ((unsigned)__builtin_bit_cast(unsigned short, y) << 16);
}
__device__ inline float gg_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ inline float gg_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xFFFF0000u); }

__device__ inline void gg_wait_vmcnt(int n) {     // wave-uniform n; a smaller count than needed is always safe
  switch (n) {
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
    case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
    case 63: break;                                    // nothing to wait for
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

// KS = cin / 32 MFMA k-steps of a one-dimensional irrep (the E blocks have 2 KS).  BWD = 0 forward, 1 backward.
template <int KS, int BWD>
__global__ __launch_bounds__(256, 2) void mlp_d8_gelu_kernel(GgArgs a) {
  constexpr int NKT1 = (KS + 1) / 2;          // 64-wide K tiles of a 1-D block
  constexpr int NKTE = KS;                    // of an E block (2 KS k-steps)
  constexpr int DMA1 = 4 * NKT1 * 2, DMAE = 2 * NKTE * 2;   // 1 KiB wave-instructions per stage
  constexpr int NA = DMA1 / 4, NE = (DMAE + 3) / 4;   // DMA instructions per wave and stage
  constexpr int NST = BWD ? 4 : 8;            // 16-byte stores per lane per slice (forward: h and y; backward: dh)
  extern __shared__ __attribute__((aligned(16))) char lds[];   // GG_NST x GG_STAGE

  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int fr = lane & 15, kg = lane >> 4;
  const int cin = a.cin, cout = a.cout;

  // ---- this workgroup's range of (token group, slice) units
  const int64_t U = (int64_t)a.groups * a.nslices;
  int64_t u = U * blockIdx.x / gridDim.x;
  const int64_t u_end = U * (blockIdx.x + 1) / gridDim.x;
  if (u >= u_end) return;

  // ---- W DMA: descriptors and lane constants.  A wave-instruction fills 8 rows of a [16][128 B] tile: lane -> row
  // (lane >> 3), chunk position (lane & 7) holding source chunk (lane & 7) ^ (row & 7).
  // the five matrices are one flat buffer [4 x (cout x cin) | 2 cout x 2 cin] (octic_linear_d8_prep layout): ONE
  // descriptor, the irrep is a scalar offset (a runtime-selected descriptor array would live in scratch memory)
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, (int)((int64_t)8 * cout * cin * 2), 0x27000);
  const int wblk = cout * cin * 2;             // bytes of one 1-D matrix; the E matrix starts at 4 wblk
  const int drow = lane >> 3;
  const int dch = (lane & 7) ^ drow;
  const unsigned vo1 = (unsigned)((drow * cin + dch * 8) * 2);            // 1-D blocks: row stride cin
  const unsigned voE = (unsigned)((drow * 2 * cin + dch * 8) * 2);        // E block: row stride 2 cin
  // one-dimensional stage: tiles (irrep i, k-tile kt), each 2 instructions (rows 0-7, 8-15): instruction id q = (i * NKT1 + kt) * 2 + half
  // E stage: tiles (copy cp, k-tile kt)
  auto issue_stage = [&](int stage_idx, int slice, bool estage) {
    char* base = lds + (stage_idx % GG_NST) * GG_STAGE;
    const int j0 = slice * 16;
    if (!estage) {
#pragma unroll
      for (int qq = 0; qq < DMA1 / 4; ++qq) {
        const int q = wid * (DMA1 / 4) + qq;                                // uniform
        const int tile = q >> 1, half = q & 1;
        const int i = tile / NKT1, kt = tile - i * NKT1;
        const int so = i * wblk + ((j0 + half * 8) * cin + kt * 64) * 2;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (__attribute__((address_space(3))) void*)(base + q * 1024), 16, vo1, so, 0, 0);
      }
    } else {
#pragma unroll
      for (int qq = 0; qq < (DMAE + 3) / 4; ++qq) {
        const int q = wid * ((DMAE + 3) / 4) + qq;
        if (q < DMAE) {
          const int tile = q >> 1, half = q & 1;
          const int cp = tile / NKTE, kt = tile - cp * NKTE;
          const int so = 4 * wblk + ((cp * cout + j0 + half * 8) * 2 * cin + kt * 64) * 2;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (__attribute__((address_space(3))) void*)(base + q * 1024), 16, voE, so,
                                                   0, 0);
        }
      }
    }
  };

  // ---- W fragment reads: lane (fr, kg) reads 16 bytes of row fr, chunk (ks & 1) * 4 + kg of k-tile ks >> 1
  const int sw = fr & 7;
  const int rd0 = fr * 128 + ((kg ^ sw) << 4), rd1 = fr * 128 + (((4 + kg) ^ sw) << 4);

  // ---- output addressing: component c of channel j lives at element offset coff(c) + j of the packed row
  //      c < 4: c * cout ; 4: 4 cout (E row 0, copy 1) ; 5: 6 cout (E row 1, copy 1) ; 6: 5 cout ; 7: 7 cout
  // after the lane-pair exchange a lane with even kg owns components 0-3, odd kg components 4-7, of 8 channels
  const int odd = kg & 1;
  const int ch8 = (kg >> 1) * 8;               // first of the lane's 8 channels inside the slice
  int coff[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c = odd * 4 + q;
    coff[q] = (c < 4 ? c * cout : (c == 4 ? 4 * cout : (c == 5 ? 6 * cout : (c == 6 ? 5 * cout : 7 * cout)))) + ch8;
  }

  bf16x8 xf1[4][KS];                           // [irrep][k-step]    fragments of this wave's 16 tokens
  bf16x8 xfe[2][2 * KS];                       // [E row][k-step]
  int cur_group = -1;
  int64_t row_h = 0;                            // element offset of this lane's token row in h / y
  bool row_ok = false;

  // A1 bias through LDS (an in-loop global load would need vmcnt(0), i.e. drain the DMA ring and the stores)
  float* const lbias = (float*)(lds + GG_NST * GG_STAGE);
  if (!BWD && a.bias) {
    for (int i = threadIdx.x; i < cout; i += 256) lbias[i] = a.bias[i];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  int stage = 0;                                // stage counter of the DMA / compute streams (slice * 2 + estage)
  // prologue: two stages in flight
  {
    const int sl0 = (int)(u % a.nslices);
    issue_stage(0, sl0, false);
    issue_stage(1, sl0, true);
  }

  for (; u < u_end; ++u) {
    const int group = (int)(u / a.nslices), slice = (int)(u - (int64_t)group * a.nslices);
    const bool has_next = u + 1 < u_end;
    const int nslice = has_next ? (int)((u + 1) % a.nslices) : 0;
    if (group != cur_group) {
      cur_group = group;
      const int64_t tok = (int64_t)group * 64 + wid * 16 + fr;
      row_ok = tok < a.M;
      const int64_t tc = row_ok ? tok : a.M - 1;
      const bf16* xr = a.x + tc * a.ldx + kg * 8;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) xf1[i][ks] = *(const bf16x8*)(xr + i * cin + ks * 32);
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int ks = 0; ks < 2 * KS; ++ks) xfe[r][ks] = *(const bf16x8*)(xr + 4 * cin + r * 2 * cin + ks * 32);
      row_h = tc * a.ldh;
      // the fragments are loop-invariant from here on: re-define them so that the compiler's waitcnt pass does not
      // protect every MFMA block of the loop with vmcnt(0) (see linear_d8_xreg_kernel in csrc/gemm.hip)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(xf1[i][ks]));
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int ks = 0; ks < 2 * KS; ++ks) asm volatile("" : "+v"(xfe[r][ks]));
    }

    f32x4 acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = f32x4{0, 0, 0, 0};

    // ---- 1-D stage.  vmcnt bookkeeping (vector-memory operations retire in issue order).  Forward: younger than this
    // stage's DMA are the E-stage DMA of this slice (NE) and the NST stores of the previous epilogue - never wait for the
    // stores themselves (they take ~1k cycles to retire).  Backward: the vmcnt(0) in front of the previous epilogue's
    // math (it needs the h loads) has already retired every DMA issued before it.
    const bool first = stage == 0;
    const int j0 = slice * 16;
    // backward: the saved pre-activation of this slice (16 bytes x 4 components per lane) is requested first, a whole
    // slice of MFMAs ahead of its use
    u32x4 hin[4];
    if (BWD) {
#pragma unroll
      for (int q = 0; q < 4; ++q) hin[q] = *(const u32x4*)(a.h + row_h + coff[q] + j0);
    }
    constexpr int NLD = BWD ? 4 : 0;
    gg_wait_vmcnt(first ? 0 : NE + NST + NLD);
    __builtin_amdgcn_s_barrier();
    if (has_next) issue_stage(stage + 2, nslice, false);
    {
      const char* base = lds + (stage % GG_NST) * GG_STAGE;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const bf16x8 wf = *(const bf16x8*)(base + ((i * NKT1 + (ks >> 1)) * 2048) + ((ks & 1) ? rd1 : rd0));
          if (GG_ABL & 4) { acc[i][0] += (float)wf[0]; continue; }
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf1[i][ks], acc[i], 0, 0, 0);
        }
    }
    ++stage;

    // ---- E stage: younger than its DMA are the previous epilogue's stores, this slice's h loads and the 1-D DMA just
    // issued for the next slice
    gg_wait_vmcnt((first ? 0 : NST + NLD) + (has_next ? NA : 0));
    __builtin_amdgcn_s_barrier();
    if (has_next) issue_stage(stage + 2, nslice, true);
    {
      const char* base = lds + (stage % GG_NST) * GG_STAGE;
#pragma unroll
      for (int cp = 0; cp < 2; ++cp)
#pragma unroll
        for (int ks = 0; ks < 2 * KS; ++ks) {
          const bf16x8 wf = *(const bf16x8*)(base + ((cp * NKTE + (ks >> 1)) * 2048) + ((ks & 1) ? rd1 : rd0));
          if (GG_ABL & 4) { acc[4 + 2 * cp][0] += (float)wf[0]; continue; }
          acc[4 + 2 * cp] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xfe[0][ks], acc[4 + 2 * cp], 0, 0, 0);
          acc[5 + 2 * cp] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xfe[1][ks], acc[5 + 2 * cp], 0, 0, 0);
        }
    }
    ++stage;

    // ---- epilogue.  Lane (fr, kg) holds component c of channels j0 + 4 kg + e (e = 0..3) of token fr in acc[c][e].
    if (!BWD && a.bias) acc[0] += *(const f32x4*)(lbias + j0 + kg * 4);
    // round to the storage dtype first: the backward recomputes from the stored h, the activation must be its function
    unsigned pk[8][2];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      pk[c][0] = gg_pack2(acc[c][0], acc[c][1]);
      pk[c][1] = gg_pack2(acc[c][2], acc[c][3]);
    }
    // lane-pair exchange (lanes l and l ^ 16): even kg keeps components 0-3 and receives the partner's channels,
    // odd kg keeps components 4-7.  v_permlane16_swap swaps the odd rows of its first operand with the even rows of
    // its second:  (A, B) -> A' = [A.r0 B.r0 A.r2 B.r2], B' = [A.r1 B.r1 A.r3 B.r3]  with A = comps 0-3, B = comps 4-7.
    // After it a lane's FIRST register holds channels 4 (kg & ~1) .. +3, the SECOND channels +4 .. +7, of components
    // (kg odd ? 4-7 : 0-3).
    u32x4 own[4];                               // [component q] = 8 channels (bf16) of component odd * 4 + q
    if (BWD) {
      // h loads: younger than them are only the DMAs of the next slice (first slice: everything older was drained)
      gg_wait_vmcnt(has_next ? NA + NE : 0);
    }
    float v[8][4];                               // working values in the MFMA layout: [component][channel e]
    if (!BWD) {
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        v[c][0] = gg_lo(pk[c][0]); v[c][1] = gg_hi(pk[c][0]); v[c][2] = gg_lo(pk[c][1]); v[c][3] = gg_hi(pk[c][1]);
      }
      // h in the exchanged layout
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const auto s0 = __builtin_amdgcn_permlane16_swap(pk[q][0], pk[4 + q][0], false, false);
        const auto s1 = __builtin_amdgcn_permlane16_swap(pk[q][1], pk[4 + q][1], false, false);
        // s*[0] = A' (even-kg lanes: own comp q ch 0-3 regs; odd-kg lanes: partner's comp 4+q), s*[1] = B'
        own[q] = odd ? u32x4{s0[0], s1[0], s0[1], s1[1]} : u32x4{s0[0], s1[0], s0[1], s1[1]};
      }
      if ((GG_ABL & 1) ? (a.M < 0) : row_ok) {
#pragma unroll
        for (int q = 0; q < 4; ++q) *(u32x4*)(a.h + row_h + coff[q] + j0) = own[q];
      }
    } else {
      // backward: v = F^-1-side input h from memory (exchanged layout -> MFMA layout), g = rounded accumulators
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        // inverse of the exchange: registers (r0, r1 | r2, r3) of `hin[q]` are (A'.lo, A'.hi | B'.lo, B'.hi) pairs
        const auto s0 = __builtin_amdgcn_permlane16_swap(hin[q][0], hin[q][2], false, false);
        const auto s1 = __builtin_amdgcn_permlane16_swap(hin[q][1], hin[q][3], false, false);
        // s0[0], s1[0] = this lane's component q channels (2 regs), s0[1], s1[1] = component 4 + q
        v[q][0] = gg_lo(s0[0]); v[q][1] = gg_hi(s0[0]); v[q][2] = gg_lo(s1[0]); v[q][3] = gg_hi(s1[0]);
        v[4 + q][0] = gg_lo(s0[1]); v[4 + q][1] = gg_hi(s0[1]); v[4 + q][2] = gg_lo(s1[1]); v[4 + q][3] = gg_hi(s1[1]);
      }
    }
    // butterflies + GELU per channel
    unsigned ok[8][2];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float xin[8], r[8], o[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) xin[c] = v[c][e];
      iso_to_reg(xin, r);
      if (!BWD) {
#pragma unroll
        for (int c = 0; c < 8; ++c) r[c] = (GG_ABL & 2) ? kSqrt2Over4 * r[c] : gg_gelu(kSqrt2Over4 * r[c]);
      } else {
        float gi[8], gr[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const unsigned w = pk[c][e >> 1];
          gi[c] = (e & 1) ? gg_hi(w) : gg_lo(w);
        }
        iso_to_reg(gi, gr);
#pragma unroll
        for (int c = 0; c < 8; ++c) r[c] = (kSqrt2Over4 * gr[c]) * ((GG_ABL & 2) ? r[c] : gg_gelu_grad(kSqrt2Over4 * r[c]));
      }
      reg_to_iso(r, o);
#pragma unroll
      for (int c = 0; c < 8; ++c) v[c][e] = kSqrt2Over4 * o[c];
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      ok[c][0] = gg_pack2(v[c][0], v[c][1]);
      ok[c][1] = gg_pack2(v[c][2], v[c][3]);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const auto s0 = __builtin_amdgcn_permlane16_swap(ok[q][0], ok[4 + q][0], false, false);
      const auto s1 = __builtin_amdgcn_permlane16_swap(ok[q][1], ok[4 + q][1], false, false);
      own[q] = u32x4{s0[0], s1[0], s0[1], s1[1]};
    }
    if ((GG_ABL & 1) ? (a.M < 0) : row_ok) {
#pragma unroll
      for (int q = 0; q < 4; ++q) *(u32x4*)(a.y + row_h + coff[q] + j0) = own[q];
    }
  }
}

}  // namespace octic

using namespace octic;

extern "C" {

// mode 0: forward (h = x W^T + b on A1, y = gelu_D8(h)); mode 1: backward (y := dh = gelu_D8'(h; x W^T)).
// x packed [M, 8 cin] bf16, h / y packed [M, 8 cout] bf16, w_flat = [4 x (cout x cin) | 2 cout x 2 cin] bf16 (prep layout).
int octic_mlp_d8_gelu(const void* x, const void* w_flat, const float* bias, void* h, void* y, int64_t M, int cin,
                      int cout, int mode, void* stream) {
  if (!x || !w_flat || !h || !y) return OCTIC_ENULL;
  if (M <= 0 || cin <= 0 || (cin % 32) || (cin / 32 != 4 && cin / 32 != 5) || cout <= 0 || (cout % 16)) return OCTIC_ESHAPE;
  if ((((uintptr_t)x) | ((uintptr_t)h) | ((uintptr_t)y)) & 15) return OCTIC_EALIGN;
  GgArgs a = {};
  a.x = (const bf16*)x; a.ldx = 8 * (int64_t)cin;
  a.w = (const bf16*)w_flat;
  a.bias = mode == 0 ? bias : nullptr;
  a.h = (bf16*)h; a.y = (bf16*)y; a.ldh = 8 * (int64_t)cout;
  a.M = (int)M; a.cin = cin; a.cout = cout;
  a.groups = (int)((M + 63) / 64);
  a.nslices = cout / 16;
  const int64_t U = (int64_t)a.groups * a.nslices;
  const int grid = U < 512 ? (int)U : 512;
  const int smem = GG_NST * GG_STAGE + cout * 4;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)mlp_d8_gelu_kernel<4, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    (void)hipFuncSetAttribute((const void*)mlp_d8_gelu_kernel<4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    (void)hipFuncSetAttribute((const void*)mlp_d8_gelu_kernel<5, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    (void)hipFuncSetAttribute((const void*)mlp_d8_gelu_kernel<5, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    (void)hipGetLastError();
    attr_done = true;
  }
  hipStream_t s = (hipStream_t)stream;
  const int ks = cin / 32;
  if (ks == 5 && mode == 0) mlp_d8_gelu_kernel<5, 0><<<grid, 256, smem, s>>>(a);
  else if (ks == 5) mlp_d8_gelu_kernel<5, 1><<<grid, 256, smem, s>>>(a);
  else if (mode == 0) mlp_d8_gelu_kernel<4, 0><<<grid, 256, smem, s>>>(a);
  else mlp_d8_gelu_kernel<4, 1><<<grid, 256, smem, s>>>(a);
  return launch_status();
}

}  // extern "C"
