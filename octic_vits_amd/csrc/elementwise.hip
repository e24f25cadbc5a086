// HBM-bound elementwise / permutation kernels of the octic block engine (gfx950).
// All of them move 16 bytes per lane per access (8 bf16 or 4+4 f32) over the token-row layout;
// arithmetic is f32 in registers.  Roofline: HBM (bytes moved / 8 TB/s); see DESIGN.md.
#include <stdlib.h>
#include "octic_common.hpp"

namespace octic {

// Offsets (in elements, relative to the token row of each tensor) of the 8 isotypic components of
// hidden channel j: x0..x3 in A1..B2 at j; x4 = E[0,j], x5 = E[1,j], x6 = E[0,c+j], x7 = E[1,c+j].
template <typename T>
__device__ inline void comp_ptrs(const View& v, int64_t m, int j, int c, T* p[8]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) p[i] = (T*)v.p[i] + m * v.ld[i] + j;
  T* e = (T*)v.p[4] + m * v.ld[4] + j;
  p[4] = e; p[5] = e + 2 * c; p[6] = e + c; p[7] = e + 3 * c;
}

// CH channels per thread.  CH = 4 halves the live block (8 x 4 values) so the kernel fits four waves per SIMD; the
// loads become 8 bytes per lane for bf16.
template <typename T, int CH> struct LoadCH;
template <> struct LoadCH<float, 4> {
  static __device__ inline void load(const float* p, float v[4]) { f32x4 a = *(const f32x4*)p; v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; }
  static __device__ inline void store(float* p, const float v[4]) { *(f32x4*)p = f32x4{v[0], v[1], v[2], v[3]}; }
};
template <> struct LoadCH<bf16, 4> {
  static __device__ inline void load(const bf16* p, float v[4]) { bf16x4 a = *(const bf16x4*)p; v[0] = (float)a[0]; v[1] = (float)a[1]; v[2] = (float)a[2]; v[3] = (float)a[3]; }
  static __device__ inline void store(bf16* p, const float v[4]) { *(bf16x4*)p = bf16x4{(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]}; }
};

template <typename T>
__global__ __launch_bounds__(256, 4) void gelu_fwd4_kernel(View x, View y, int64_t M, int c) {
  const int c4 = c >> 2;
  const int64_t total = M * c4;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t m = idx / c4;
    const int j = (int)(idx - m * c4) << 2;
    T *px[8], *py[8];
    comp_ptrs<T>(x, m, j, c, px);
    comp_ptrs<T>(y, m, j, c, py);
    float a[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i) LoadCH<T, 4>::load(px[i], a[i]);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v[8], r[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = a[i][e];
      iso_to_reg(v, r);
#pragma unroll
      for (int i = 0; i < 8; ++i) r[i] = gelu_exact(kSqrt2Over4 * r[i]);
      reg_to_iso(r, v);
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i][e] = kSqrt2Over4 * v[i];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) LoadCH<T, 4>::store(py[i], a[i]);
  }
}

template <typename T>
__global__ __launch_bounds__(256, 4) void gelu_bwd4_kernel(View g, View x, View gin, int64_t M, int c) {
  const int c4 = c >> 2;
  const int64_t total = M * c4;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t m = idx / c4;
    const int j = (int)(idx - m * c4) << 2;
    T *px[8], *pg[8], *po[8];
    comp_ptrs<T>(x, m, j, c, px);
    comp_ptrs<T>(g, m, j, c, pg);
    comp_ptrs<T>(gin, m, j, c, po);
    float a[8][4], b[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      LoadCH<T, 4>::load(px[i], a[i]);
      LoadCH<T, 4>::load(pg[i], b[i]);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v[8], r[8], gv[8], gr[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        v[i] = a[i][e];
        gv[i] = b[i][e];
      }
      iso_to_reg(v, r);
      iso_to_reg(gv, gr);
#pragma unroll
      for (int i = 0; i < 8; ++i) gr[i] = (kSqrt2Over4 * gr[i]) * gelu_grad(kSqrt2Over4 * r[i]);
      reg_to_iso(gr, gv);
#pragma unroll
      for (int i = 0; i < 8; ++i) b[i][e] = kSqrt2Over4 * gv[i];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) LoadCH<T, 4>::store(po[i], b[i]);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void gelu_fwd_kernel(View x, View y, int64_t M, int c) {
  const int c8 = c >> 3;
  const int64_t total = M * c8;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t m = idx / c8;
    const int j = (int)(idx - m * c8) << 3;
    T *px[8], *py[8];
    comp_ptrs<T>(x, m, j, c, px);
    comp_ptrs<T>(y, m, j, c, py);
    float a[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i) load8<T>(px[i], a[i]);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float v[8], r[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = a[i][e];
      iso_to_reg(v, r);
#pragma unroll
      for (int i = 0; i < 8; ++i) r[i] = gelu_exact(kSqrt2Over4 * r[i]);
      reg_to_iso(r, v);
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i][e] = kSqrt2Over4 * v[i];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) store8<T>(py[i], a[i]);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void gelu_bwd_kernel(View g, View x, View gin, int64_t M, int c) {
  const int c8 = c >> 3;
  const int64_t total = M * c8;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t m = idx / c8;
    const int j = (int)(idx - m * c8) << 3;
    T *px[8], *pg[8], *po[8];
    comp_ptrs<T>(x, m, j, c, px);
    comp_ptrs<T>(g, m, j, c, pg);
    comp_ptrs<T>(gin, m, j, c, po);
    float a[8][8], b[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      load8<T>(px[i], a[i]);
      load8<T>(pg[i], b[i]);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float v[8], r[8], gv[8], gr[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        v[i] = a[i][e];
        gv[i] = b[i][e];
      }
      iso_to_reg(v, r);     // the transform is orthogonal: its transpose is its inverse, so the
      iso_to_reg(gv, gr);   // cotangent goes through the same iso->regular map (d8_gelu.py:284-294)
#pragma unroll
      for (int i = 0; i < 8; ++i) gr[i] = (kSqrt2Over4 * gr[i]) * gelu_grad(kSqrt2Over4 * r[i]);
      reg_to_iso(gr, gv);
#pragma unroll
      for (int i = 0; i < 8; ++i) b[i][e] = kSqrt2Over4 * gv[i];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) store8<T>(po[i], b[i]);
  }
}

// y = rs * x  (f32 -> TOUT), 8 logical packed columns per thread.
template <typename TOUT>
__global__ __launch_bounds__(256) void cast_rowscale_kernel(View x, View y, const float* rs, int64_t rps, int64_t M,
                                                            int c) {
  const int64_t total = M * c;  // 8c/8 chunks per row
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t m = idx / c;
    const int e = (int)(idx - m * c) << 3;
    float v[8];
    load8<float>(view_ptr<float>(x, m, e, c), v);
    if (rs) {
      const float s = rs[m / rps];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] *= s;
    }
    store8<TOUT>(view_ptr<TOUT>(y, m, e, c), v);
  }
}

// Head packing.  DIR 0: view -> heads, DIR 1: heads -> view.  heads[s] is a [B][H][T][8w] array; the view has
// n_s*8c channels.  A workgroup moves a tile of TT consecutive tokens of one sample through LDS: token rows
// are read/written as whole coalesced rows (16 B per lane), head vectors as runs of TT*8w contiguous
// elements per (s, head) — the per-head vector interleaves six irrep pieces of w / 2w elements (w = 10 at
// ViT-H: 20-byte pieces), which is why this cannot be a plain strided copy.  V = elements per LDS<->global
// access on the head side (V divides w).
struct HeadPtrs {
  char* p[3];
};

// Per-lane constants of one head-side access slot.  A slot is ONE 16-byte access of the heads array = NP pieces of
// V elements of token r starting at head-vector position e; piece p sits at packed column
// lds[p] + mult[p] * (s*c + h*w) of the LDS tile (mult = 2 inside the two-dimensional irrep's rows).
template <int NP>
struct HeadSlot {
  int lds[NP];   // element offset inside the LDS tile for (s,h) = (0,0), or -1 beyond the run
  int mult[NP];
  int glb;       // element offset inside the (s,h) run of the heads array: r*hd + e
};
template <int PB> struct Piece;
template <> struct Piece<2> { typedef unsigned short type; };
template <> struct Piece<4> { typedef unsigned int type; };
template <> struct Piece<8> { typedef uint2 type; };
template <> struct Piece<16> { typedef u32x4 type; };

template <typename T, int V, int DIR, int SLOTS>
__global__ __launch_bounds__(256) void heads_permute_kernel(View v, HeadPtrs heads, int64_t B, int64_t T_, int H, int c,
                                                            int n_s, int TT) {
  extern __shared__ __attribute__((aligned(16))) char hsm[];
  T* tile = (T*)hsm;  // [TT][n_s*8c] token rows in packed order
  const int w = c / H, hd = 8 * w;
  const int row_elems = n_s * 8 * c;
  const int tiles_per_b = (int)((T_ + TT - 1) / TT);
  const int64_t b = blockIdx.x / tiles_per_b;
  const int64_t t0 = (int64_t)(blockIdx.x - b * tiles_per_b) * TT;
  const int nt = (int)((T_ - t0) < TT ? (T_ - t0) : TT);
  constexpr int EPC = 16 / (int)sizeof(T);
  const int row_stride = row_elems;
  const int cpr = row_elems / EPC;  // 16-byte chunks per token row
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;

  auto row_phase = [&](bool load) {
    // whole token rows, 16 B per lane, coalesced; packed column e -> (tensor, offset); a chunk never straddles
    // a segment because c % 8 == 0
    const int cv = n_s * c;
    for (int q = threadIdx.x; q < nt * cpr; q += 256) {
      const int r = q / cpr, e = (q - r * cpr) * EPC;
      const int64_t m = b * T_ + t0 + r;
      T* gp;
      if (e < 4 * cv) {
        const int g = e / cv;
        gp = (T*)v.p[g] + m * v.ld[g] + (e - g * cv);
      } else {
        gp = (T*)v.p[4] + m * v.ld[4] + (e - 4 * cv);
      }
      u32x4* lp = (u32x4*)(tile + (size_t)r * row_stride + e);
      if (load) *lp = *(const u32x4*)gp;
      else *(u32x4*)gp = *lp;
    }
  };

  // head side: for a fixed (s,h) the TT head vectors are ONE contiguous run of TT*hd elements.  Slot k of a lane
  // is the access at run element (lane + 64k)*SE; it is gathered from / scattered to NP pieces of V
  // elements in the LDS tile (V divides w, so a piece never straddles an irrep block; hd = 8w is a multiple of EPC
  // whenever V*NP = EPC, so a slot never straddles a token).  Everything that does not depend on (s,h) is computed
  // once here.
  constexpr int PB = V * (int)sizeof(T);     // piece bytes
  constexpr int NP = DIR == 1 ? 16 / PB : 1; // pieces per head-side access: a 16-byte gather pays off when scattering
                                             // INTO the tile (heads -> view: 105 -> 91 us) but not when reading it
                                             // (view -> heads: 62 us with single pieces, 79 with gathers)
  constexpr int SE = NP * V;                 // elements per head-side access
  typedef typename Piece<NP * PB>::type slot_t;
  typedef typename Piece<PB>::type piece_t;
  HeadSlot<NP> sl[SLOTS];
  const int run = nt * hd;  // valid elements of the run (last tile of a sample may be short)
#pragma unroll
  for (int k = 0; k < SLOTS; ++k) {
    const int idx = (lane + 64 * k) * SE;
    sl[k].glb = idx < run ? idx : -1;
#pragma unroll
    for (int pz = 0; pz < NP; ++pz) {
      const int id = idx + pz * V;
      const int r = id / hd, e = id - r * hd;
      int a, mult;
      if (e < 4 * w) {
        const int g = e / w;
        a = g * (n_s * c) + (e - g * w); mult = 1;
      } else {
        const int rr = (e - 4 * w) / (2 * w);
        a = 4 * n_s * c + rr * (2 * n_s * c) + (e - 4 * w - rr * 2 * w); mult = 2;
      }
      sl[k].lds[pz] = r * row_stride + a;
      sl[k].mult[pz] = mult;
    }
  }
  auto head_phase = [&](bool to_heads) {
#pragma unroll 4
    for (int sh = wid; sh < n_s * H; sh += 4) {      // one wave per (s, head)
      const int s = sh / H, h = sh - s * H;
      T* gbase = (T*)heads.p[s] + ((b * H + h) * T_ + t0) * hd;
      const int col = s * c + h * w;
#pragma unroll
      for (int k = 0; k < SLOTS; ++k) {
        if (sl[k].glb >= 0) {
          union { slot_t v; piece_t pc[NP]; } u;
          if (to_heads) {
#pragma unroll
            for (int pz = 0; pz < NP; ++pz) u.pc[pz] = *(const piece_t*)(tile + sl[k].lds[pz] + sl[k].mult[pz] * col);
            *(slot_t*)(gbase + sl[k].glb) = u.v;
          } else {
            u.v = *(const slot_t*)(gbase + sl[k].glb);
#pragma unroll
            for (int pz = 0; pz < NP; ++pz) *(piece_t*)(tile + sl[k].lds[pz] + sl[k].mult[pz] * col) = u.pc[pz];
          }
        }
      }
    }
  };
  if (DIR == 0) {
    row_phase(true);
    __syncthreads();
    head_phase(true);
  } else {
    head_phase(false);
    __syncthreads();
    row_phase(false);
  }
}

// dense (8-tuple order) <-> view.  DIR 0: view(f32) -> dense(TOUT); DIR 1: dense(f32) -> view(f32).
template <typename TOUT, int DIR>
__global__ __launch_bounds__(256) void handoff_cat_kernel(View x, TOUT* dense, int64_t M, int c) {
  const int c8 = c >> 3;
  const int64_t total = M * c;  // 8c/8 chunks per row
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t m = idx / c;
    const int q = (int)(idx - m * c);
    const int blk = q / c8, off = (q - blk * c8) << 3;
    float* src;
    if (blk < 4) src = (float*)x.p[blk] + m * x.ld[blk] + off;
    else {
      // 8-tuple entries 4..7 = E[0,:c], E[1,:c], E[0,c:], E[1,c:]
      const int row = (blk - 4) & 1, half = (blk - 4) >> 1;
      src = (float*)x.p[4] + m * x.ld[4] + row * 2 * c + half * c + off;
    }
    TOUT* d = dense + m * (int64_t)(8 * c) + (int64_t)q * 8;
    float v[8];
    if (DIR == 0) {
      load8<float>(src, v);
      store8<TOUT>(d, v);
    } else {
      load8<TOUT>(d, v);
      store8<float>(src, v);
    }
  }
}

template <typename TOUT>
__global__ __launch_bounds__(256) void power_spectrum_fwd_kernel(View x, TOUT* dense, int64_t M, int c) {
  const int c8 = c >> 3;
  const int64_t total = M * (6 * c8);
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t m = idx / (6 * c8);
    const int q = (int)(idx - m * (6 * c8));
    float v[8];
    if (q < 4 * c8) {
      const int g = q / c8, off = (q - g * c8) << 3;
      load8<float>((float*)x.p[g] + m * x.ld[g] + off, v);
      if (g > 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = fabsf(v[i]);
      }
    } else {
      const int off = (q - 4 * c8) << 3;
      float a[8], b[8];
      const float* e = (float*)x.p[4] + m * x.ld[4] + off;
      load8<float>(e, a);
      load8<float>(e + 2 * c, b);
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = sqrtf(a[i] * a[i] + b[i] * b[i]);
    }
    store8<TOUT>(dense + m * (int64_t)(6 * c) + (int64_t)q * 8, v);
  }
}

__global__ __launch_bounds__(256) void power_spectrum_bwd_kernel(const float* dd, View x, View dx, int64_t M, int c) {
  const int c8 = c >> 3;
  const int64_t total = M * (6 * c8);
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t m = idx / (6 * c8);
    const int q = (int)(idx - m * (6 * c8));
    float g[8];
    load8<float>(dd + m * (int64_t)(6 * c) + (int64_t)q * 8, g);
    if (q < 4 * c8) {
      const int gi = q / c8, off = (q - gi * c8) << 3;
      if (gi > 0) {
        float v[8];
        load8<float>((float*)x.p[gi] + m * x.ld[gi] + off, v);
#pragma unroll
        for (int i = 0; i < 8; ++i) g[i] = v[i] > 0.f ? g[i] : (v[i] < 0.f ? -g[i] : 0.f);
      }
      store8<float>((float*)dx.p[gi] + m * dx.ld[gi] + off, g);
    } else {
      const int off = (q - 4 * c8) << 3;
      float a[8], b[8], da[8], db[8];
      const float* e = (float*)x.p[4] + m * x.ld[4] + off;
      load8<float>(e, a);
      load8<float>(e + 2 * c, b);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float n = sqrtf(a[i] * a[i] + b[i] * b[i]);
        const float s = n > 0.f ? g[i] / n : 0.f;
        da[i] = a[i] * s;
        db[i] = b[i] * s;
      }
      float* de = (float*)dx.p[4] + m * dx.ld[4] + off;
      store8<float>(de, da);
      store8<float>(de + 2 * c, db);
    }
  }
}

// img [B,Cin,H,W] f32 -> patches [B*G*G, Kpad] T ; column = (ch, py, px); zero padded.
template <typename T>
__global__ __launch_bounds__(256) void im2col_kernel(const float* img, T* patches, int64_t B, int Cin, int Himg,
                                                     int Wimg, int p, int Kpad) {
  const int Gh = Himg / p, Gw = Wimg / p, K = Cin * p * p, k8 = Kpad >> 3;
  const int64_t total = B * Gh * Gw * k8;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t row = idx / k8;
    const int kc = (int)(idx - row * k8) << 3;
    const int64_t b = row / (Gh * Gw);
    const int n = (int)(row - b * (Gh * Gw));
    const int gy = n / Gw, gx = n - gy * Gw;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = kc + i;
      float val = 0.f;
      if (k < K) {
        const int ch = k / (p * p), rem = k - ch * p * p, py = rem / p, px = rem - py * p;
        val = img[((b * Cin + ch) * Himg + (gy * p + py)) * (int64_t)Wimg + gx * p + px];
      }
      v[i] = val;
    }
    store8<T>(patches + row * (int64_t)Kpad + kc, v);
  }
}

constexpr int kColsumRows = 128;  // rows per block

template <typename T>
__device__ inline void load4e(const T* p, float v[4]);
template <>
__device__ inline void load4e<float>(const float* p, float v[4]) {
  f32x4 a = *(const f32x4*)p;
  v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
}
template <>
__device__ inline void load4e<bf16>(const bf16* p, float v[4]) {
  bf16x4 a = *(const bf16x4*)p;
  v[0] = (float)a[0]; v[1] = (float)a[1]; v[2] = (float)a[2]; v[3] = (float)a[3];
}

template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* a, int64_t ld, int64_t M, int c, float* partials) {
  // block b sums rows [b*kColsumRows, ...) of the [M, c] matrix; thread = (row lane, 4-column chunk)
  __shared__ float red[256 * 4];
  const int c4 = c >> 2;
  const int lanes = 256 / c4 > 0 ? 256 / c4 : 1;
  const int rl = threadIdx.x / c4, cc = threadIdx.x - rl * c4;
  float acc[4] = {0, 0, 0, 0};
  const int64_t r0 = (int64_t)blockIdx.x * kColsumRows;
  const int64_t r1 = r0 + kColsumRows < M ? r0 + kColsumRows : M;
  if (rl < lanes) {
    for (int64_t m = r0 + rl; m < r1; m += lanes) {
      float v[4];
      load4e<T>(a + m * ld + cc * 4, v);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] += v[i];
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) red[threadIdx.x * 4 + i] = acc[i];
  __syncthreads();
  for (int col = threadIdx.x; col < c; col += 256) {
    const int ch = col >> 2, i = col & 3;
    float s = 0.f;
    for (int l = 0; l < lanes; ++l) s += red[(l * c4 + ch) * 4 + i];
    partials[(int64_t)blockIdx.x * c + col] = s;
  }
}

// out[col] = sum_b partials[b][col]; 256 threads = 16 columns x 16 block-lanes, combined through LDS in a fixed order
__global__ __launch_bounds__(256) void colsum_finish_kernel(const float* partials, int nblk, int c, float* out) {
  __shared__ float red[16][17];
  const int cl = threadIdx.x & 15, bl = threadIdx.x >> 4;
  const int col = blockIdx.x * 16 + cl;
  float s = 0.f;
  if (col < c)
    for (int b = bl; b < nblk; b += 16) s += partials[(int64_t)b * c + col];
  red[bl][cl] = s;
  __syncthreads();
  if (bl == 0 && col < c) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][cl];
    out[col] = t;
  }
}

// Compute-dtype copies of a LinearD8's f32 master weights, one launch per layer:
//   wb = [W_A1|W_A2|W_B1|W_B2|W_E] as they are (forward GEMM), wt = each matrix transposed with the
//   layer-scale folded in, wt_g[k][n] = cs_g[n] * W_g[n][k]  (input-gradient GEMM dX = dY diag(cs) W).
struct PrepArgs {
  const float* w[5];
  const float* cs[5];
};
template <typename T>
__global__ __launch_bounds__(256) void linear_prep_kernel(PrepArgs a, T* wb, T* wt, int cin, int cout) {
  const int64_t small = (int64_t)cin * cout;
  const int64_t total = 8 * small;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int g = idx < 4 * small ? (int)(idx / small) : 4;
    const int64_t base = g < 4 ? g * small : 4 * small;
    const int K = g < 4 ? cin : 2 * cin, N = g < 4 ? cout : 2 * cout;
    const int64_t loc = idx - base;
    const int n = (int)(loc / K), k = (int)(loc - (int64_t)n * K);
    const float v = a.w[g][loc];
    if (wb) wb[idx] = (T)v;
    if (wt) wt[base + (int64_t)k * N + n] = (T)(a.cs[g] ? a.cs[g][n] * v : v);
  }
}

// All layers' preparations in one launch (after an optimizer step): items sorted by block_begin.  A workgroup owns one
// 64 x 64 tile of one irrep's [N,K] matrix: rows are read and the bf16 copy written along k (coalesced), the
// transposed, layer-scaled copy is written along n out of an LDS tile (the per-layer kernel above scatters 2-byte
// stores instead; fine for one layer, 0.4 ms for all 64).
__host__ __device__ inline int prep_tiles(int cin, int cout) {
  const int t1 = ((cout + 63) / 64) * ((cin + 63) / 64), te = ((2 * cout + 63) / 64) * ((2 * cin + 63) / 64);
  return 4 * t1 + te;
}

template <typename T>
__global__ __launch_bounds__(256) void linear_prep_batch_kernel(const octic_prep_item* __restrict__ items, int n_items) {
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
  const octic_prep_item& I = items[it];
  const int cin = I.cin, cout = I.cout;
  int lt = b - I.block_begin;
  const int t1 = ((cout + 63) / 64) * ((cin + 63) / 64);
  const int g = lt < 4 * t1 ? lt / t1 : 4;
  lt -= g < 4 ? g * t1 : 4 * t1;
  const int K = g < 4 ? cin : 2 * cin, N = g < 4 ? cout : 2 * cout;
  const int kt = (K + 63) / 64;
  const int n0 = (lt / kt) * 64, k0 = (lt % kt) * 64;
  const int64_t base = (int64_t)(g < 4 ? g : 4) * cin * cout;
  const float* w = I.w[g];
  const float* cs = I.cs[g];
  T* wb = (T*)I.wb;
  T* wt = (T*)I.wt;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll 4
  for (int r = ty; r < 64; r += 4) {
    const int n = n0 + r, k = k0 + tx;
    float v = 0.f;
    if (n < N && k < K) {
      v = w[(int64_t)n * K + k];
      if (wb) wb[base + (int64_t)n * K + k] = (T)v;
      if (cs) v *= cs[n];
    }
    tile[r][tx] = v;
  }
  __syncthreads();
  if (wt) {
#pragma unroll 4
    for (int r = ty; r < 64; r += 4) {
      const int k = k0 + r, n = n0 + tx;
      if (k < K && n < N) wt[base + (int64_t)k * N + n] = (T)tile[tx][r];
    }
  }
}

inline int grid_for(int64_t total) {
  int64_t g = (total + 255) / 256;
  const int64_t cap = 256 * 16;  // 16 blocks per CU, grid-stride beyond
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

template <typename T, int V, int DIR>
static int heads_launch(View vv, HeadPtrs hp, int64_t B, int64_t T_, int H, int c, int n_s, hipStream_t s) {
  const int w = c / H, hd = 8 * w;
  const size_t row_bytes = (size_t)n_s * 8 * c * sizeof(T);
  constexpr int tt_env = 4;   // 4 tokens = 30 KiB of LDS at ViT-H: five workgroups per CU (8: 78/90 us, 4: 61/45 us pack/unpack)
  int TT = tt_env;
  while (TT > 1 && TT * row_bytes > 64 * 1024) TT >>= 1;
  if (TT * row_bytes > 160 * 1024) return OCTIC_ESHAPE;
  const size_t smem = TT * row_bytes;
  const int grid = (int)(B * ((T_ + TT - 1) / TT));
  const int se = DIR == 1 ? 16 / (int)sizeof(T) : V;                // elements per head-side access (see the kernel)
  const int slots = (TT * hd / se + 63) / 64;
#define OCTIC_HEADS_LAUNCH(SL)                                                                                          \
  do {                                                                                                                  \
    if (smem > 64 * 1024)                                                                                               \
      (void)hipFuncSetAttribute((const void*)heads_permute_kernel<T, V, DIR, SL>,                                       \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                                \
    heads_permute_kernel<T, V, DIR, SL><<<grid, 256, smem, s>>>(vv, hp, B, T_, H, c, n_s, TT);                          \
  } while (0)
  if (slots <= 1) OCTIC_HEADS_LAUNCH(1);
  else if (slots <= 2) OCTIC_HEADS_LAUNCH(2);
  else if (slots <= 4) OCTIC_HEADS_LAUNCH(4);
  else if (slots <= 5) OCTIC_HEADS_LAUNCH(5);
  else if (slots <= 8) OCTIC_HEADS_LAUNCH(8);
  else if (slots <= 16) OCTIC_HEADS_LAUNCH(16);
  else return OCTIC_ESHAPE;
#undef OCTIC_HEADS_LAUNCH
  return launch_status();
}

template <typename T, int DIR>
static int heads_dispatch(const octic_view* v, void* const heads[3], int64_t B, int64_t T_, int H, int c, int n_s,
                          void* stream) {
  View vv = make_view<void>(v);
  HeadPtrs hp;
  for (int i = 0; i < 3; ++i) hp.p[i] = (char*)(i < n_s ? heads[i] : nullptr);
  const int w = c / H;
  hipStream_t s = (hipStream_t)stream;
  // V elements per head-side access: must divide w (pieces are w / 2w long) and stay <= 16 bytes
  const int vmax = 16 / (int)sizeof(T);
  int V = 1;
  while (V * 2 <= vmax && (w % (V * 2)) == 0) V *= 2;
  if constexpr (sizeof(T) == 2) {
    if (V == 8) return heads_launch<T, 8, DIR>(vv, hp, B, T_, H, c, n_s, s);
  }
  switch (V) {
    case 4: return heads_launch<T, 4, DIR>(vv, hp, B, T_, H, c, n_s, s);
    case 2: return heads_launch<T, 2, DIR>(vv, hp, B, T_, H, c, n_s, s);
    default: return heads_launch<T, 1, DIR>(vv, hp, B, T_, H, c, n_s, s);
  }
}

static int heads_check(const octic_view* v, void* const heads[3], int64_t B, int64_t T_, int H, int c, int n_s, int dtype) {
  int e;
  if ((e = check_c(c))) return e;
  if (B <= 0 || T_ <= 0 || H <= 0 || (c % H) != 0 || (n_s != 1 && n_s != 3)) return OCTIC_ESHAPE;
  if ((e = check_view(v, n_s * c, dtype))) return e;
  if (!heads) return OCTIC_ENULL;
  for (int i = 0; i < n_s; ++i) {
    if (!heads[i]) return OCTIC_ENULL;
    if (((uintptr_t)heads[i]) & 15) return OCTIC_EALIGN;
  }
  if (dtype != OCTIC_F32 && dtype != OCTIC_BF16) return OCTIC_EDTYPE;
  return OCTIC_OK;
}

// Sample blocks between a full batch and a compacted one (d8_layers.COMPACT_DROP_PATH: a branch runs on the samples its
// stochastic-depth mask keeps):  gather  dst[i] = src[idx[i]]  or scatter  dst[idx[i]] = src[i]  for i < n, a block = the
// `block_bytes` (a multiple of 16) of one sample's token rows.  16 bytes per lane, blockIdx.y = sample.
template <int SCATTER>
__global__ __launch_bounds__(256) void sample_blocks_kernel(const char* __restrict__ src, char* __restrict__ dst,
                                                            const int64_t* __restrict__ idx, int64_t block_bytes) {
  const int64_t i = blockIdx.y;
  const int64_t j = idx[i];
  const char* s = src + (SCATTER ? i : j) * block_bytes;
  char* d = dst + (SCATTER ? j : i) * block_bytes;
  const int64_t n16 = block_bytes >> 4;
  for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n16; k += (int64_t)gridDim.x * 256)
    ((u32x4*)d)[k] = ((const u32x4*)s)[k];
}

int g_route[OCTIC_ROUTE_COUNT] = {};

}  // namespace octic

using namespace octic;

extern "C" {

int octic_abi_version(void) { return OCTIC_ABI_VERSION; }

int octic_route_override(int knob, int value) {
  if (knob < 0 || knob >= OCTIC_ROUTE_COUNT) return OCTIC_ESHAPE;
  const int old = octic::g_route[knob];
  octic::g_route[knob] = value;
  return old;
}

const char* octic_strerror(int code) {
  switch (code) {
    case OCTIC_OK: return "ok";
    case OCTIC_ESHAPE: return "octic: bad shape (c must be a positive multiple of 8; sizes positive; heads must divide c)";
    case OCTIC_EALIGN: return "octic: pointer or row stride breaks 16-byte row alignment";
    case OCTIC_EDTYPE: return "octic: unsupported dtype combination";
    case OCTIC_ENULL: return "octic: required pointer is NULL";
    case OCTIC_EWORKSPACE: return "octic: workspace too small";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "octic: unknown error";
  }
}

int octic_gelu_d8_fwd(const octic_view* x, const octic_view* y, int64_t M, int c, int dtype, void* stream) {
  int e;
  if ((e = check_c(c)) || (e = check_view(x, c, dtype)) || (e = check_view(y, c, dtype))) return e;
  if (M <= 0) return OCTIC_ESHAPE;
  View vx = make_view<void>(x), vy = make_view<void>(y);
  // bf16: four channels per thread (111 VGPRs, four waves per SIMD: 81 -> 62 us in-step at ViT-H); f32 keeps eight
  constexpr int ch4 = 1;      // four channels per thread (eight: 207-250 VGPRs, slower)
  if (ch4 && dtype == OCTIC_BF16) {
    gelu_fwd4_kernel<bf16><<<grid_for(M * (c / 4)), 256, 0, (hipStream_t)stream>>>(vx, vy, M, c);
    return launch_status();
  }
  const int grid = grid_for(M * (c / 8));
  if (dtype == OCTIC_F32) gelu_fwd_kernel<float><<<grid, 256, 0, (hipStream_t)stream>>>(vx, vy, M, c);
  else if (dtype == OCTIC_BF16) gelu_fwd_kernel<bf16><<<grid, 256, 0, (hipStream_t)stream>>>(vx, vy, M, c);
  else return OCTIC_EDTYPE;
  return launch_status();
}

int octic_gelu_d8_bwd(const octic_view* g, const octic_view* x, const octic_view* gin, int64_t M, int c, int dtype,
                      void* stream) {
  int e;
  if ((e = check_c(c)) || (e = check_view(g, c, dtype)) || (e = check_view(x, c, dtype)) ||
      (e = check_view(gin, c, dtype)))
    return e;
  if (M <= 0) return OCTIC_ESHAPE;
  View vg = make_view<void>(g), vx = make_view<void>(x), vo = make_view<void>(gin);
  constexpr int ch4 = 1;      // four channels per thread (eight: 207-250 VGPRs, slower)
  if (ch4 && dtype == OCTIC_BF16) {                       // 109 -> 95 us in-step
    gelu_bwd4_kernel<bf16><<<grid_for(M * (c / 4)), 256, 0, (hipStream_t)stream>>>(vg, vx, vo, M, c);
    return launch_status();
  }
  const int grid = grid_for(M * (c / 8));
  if (dtype == OCTIC_F32) gelu_bwd_kernel<float><<<grid, 256, 0, (hipStream_t)stream>>>(vg, vx, vo, M, c);
  else if (dtype == OCTIC_BF16) gelu_bwd_kernel<bf16><<<grid, 256, 0, (hipStream_t)stream>>>(vg, vx, vo, M, c);
  else return OCTIC_EDTYPE;
  return launch_status();
}

int octic_cast_rowscale(const octic_view* x, const octic_view* y, const float* rs, int64_t rows_per_sample, int64_t M,
                        int c, int out_dtype, void* stream) {
  int e;
  if ((e = check_c(c)) || (e = check_view(x, c, OCTIC_F32)) || (e = check_view(y, c, out_dtype))) return e;
  if (M <= 0 || (rs && rows_per_sample <= 0)) return OCTIC_ESHAPE;
  View vx = make_view<void>(x), vy = make_view<void>(y);
  const int grid = grid_for(M * c);
  if (out_dtype == OCTIC_F32)
    cast_rowscale_kernel<float><<<grid, 256, 0, (hipStream_t)stream>>>(vx, vy, rs, rows_per_sample, M, c);
  else if (out_dtype == OCTIC_BF16)
    cast_rowscale_kernel<bf16><<<grid, 256, 0, (hipStream_t)stream>>>(vx, vy, rs, rows_per_sample, M, c);
  else return OCTIC_EDTYPE;
  return launch_status();
}

int octic_attn_pack_heads(const octic_view* qkv, void* const heads[3], int64_t B, int64_t T_, int H, int c, int n_s,
                          int dtype, void* stream) {
  int e = heads_check(qkv, heads, B, T_, H, c, n_s, dtype);
  if (e) return e;
  return dtype == OCTIC_F32 ? heads_dispatch<float, 0>(qkv, heads, B, T_, H, c, n_s, stream)
                            : heads_dispatch<bf16, 0>(qkv, heads, B, T_, H, c, n_s, stream);
}

int octic_attn_unpack_heads(void* const heads[3], const octic_view* y, int64_t B, int64_t T_, int H, int c, int n_s,
                            int dtype, void* stream) {
  int e = heads_check(y, heads, B, T_, H, c, n_s, dtype);
  if (e) return e;
  return dtype == OCTIC_F32 ? heads_dispatch<float, 1>(y, heads, B, T_, H, c, n_s, stream)
                            : heads_dispatch<bf16, 1>(y, heads, B, T_, H, c, n_s, stream);
}

int octic_linear_d8_prep(const float* const w32[5], const float* const cs[5], int cin, int cout, void* wb, void* wt,
                         int dtype, void* stream) {
  if (!w32 || (!wb && !wt)) return OCTIC_ENULL;
  if (cin <= 0 || cout <= 0) return OCTIC_ESHAPE;
  PrepArgs a;
  for (int i = 0; i < 5; ++i) {
    if (!w32[i]) return OCTIC_ENULL;
    a.w[i] = w32[i];
    a.cs[i] = cs ? cs[i] : nullptr;
  }
  const int grid = grid_for((int64_t)8 * cin * cout);
  if (dtype == OCTIC_F32) linear_prep_kernel<float><<<grid, 256, 0, (hipStream_t)stream>>>(a, (float*)wb, (float*)wt, cin, cout);
  else if (dtype == OCTIC_BF16) linear_prep_kernel<bf16><<<grid, 256, 0, (hipStream_t)stream>>>(a, (bf16*)wb, (bf16*)wt, cin, cout);
  else return OCTIC_EDTYPE;
  return launch_status();
}

int octic_linear_d8_prep_batch_blocks(int cin, int cout) { return prep_tiles(cin, cout); }

int octic_linear_d8_prep_batch(const octic_prep_item* items_dev, int n_items, int total_blocks, int dtype, void* stream) {
  if (!items_dev) return OCTIC_ENULL;
  if (n_items <= 0 || total_blocks <= 0) return OCTIC_ESHAPE;
  if (dtype == OCTIC_BF16) linear_prep_batch_kernel<bf16><<<total_blocks, 256, 0, (hipStream_t)stream>>>(items_dev, n_items);
  else if (dtype == OCTIC_F32) linear_prep_batch_kernel<float><<<total_blocks, 256, 0, (hipStream_t)stream>>>(items_dev, n_items);
  else return OCTIC_EDTYPE;
  return launch_status();
}

int octic_handoff_cat_fwd(const octic_view* x, void* dense, int64_t M, int c, int out_dtype, void* stream) {
  int e;
  if ((e = check_c(c)) || (e = check_view(x, c, OCTIC_F32))) return e;
  if (!dense) return OCTIC_ENULL;
  if (M <= 0) return OCTIC_ESHAPE;
  View vx = make_view<void>(x);
  const int grid = grid_for(M * c);
  if (out_dtype == OCTIC_F32) handoff_cat_kernel<float, 0><<<grid, 256, 0, (hipStream_t)stream>>>(vx, (float*)dense, M, c);
  else if (out_dtype == OCTIC_BF16) handoff_cat_kernel<bf16, 0><<<grid, 256, 0, (hipStream_t)stream>>>(vx, (bf16*)dense, M, c);
  else return OCTIC_EDTYPE;
  return launch_status();
}

int octic_handoff_cat_bwd(const float* ddense, const octic_view* dx, int64_t M, int c, void* stream) {
  int e;
  if ((e = check_c(c)) || (e = check_view(dx, c, OCTIC_F32))) return e;
  if (!ddense) return OCTIC_ENULL;
  if (M <= 0) return OCTIC_ESHAPE;
  View vx = make_view<void>(dx);
  handoff_cat_kernel<float, 1><<<grid_for(M * c), 256, 0, (hipStream_t)stream>>>(vx, (float*)ddense, M, c);
  return launch_status();
}

int octic_power_spectrum_fwd(const octic_view* x, void* dense, int64_t M, int c, int out_dtype, void* stream) {
  int e;
  if ((e = check_c(c)) || (e = check_view(x, c, OCTIC_F32))) return e;
  if (!dense) return OCTIC_ENULL;
  if (M <= 0) return OCTIC_ESHAPE;
  View vx = make_view<void>(x);
  const int grid = grid_for(M * 6 * (c / 8));
  if (out_dtype == OCTIC_F32) power_spectrum_fwd_kernel<float><<<grid, 256, 0, (hipStream_t)stream>>>(vx, (float*)dense, M, c);
  else if (out_dtype == OCTIC_BF16) power_spectrum_fwd_kernel<bf16><<<grid, 256, 0, (hipStream_t)stream>>>(vx, (bf16*)dense, M, c);
  else return OCTIC_EDTYPE;
  return launch_status();
}

int octic_power_spectrum_bwd(const float* ddense, const octic_view* x, const octic_view* dx, int64_t M, int c,
                             void* stream) {
  int e;
  if ((e = check_c(c)) || (e = check_view(x, c, OCTIC_F32)) || (e = check_view(dx, c, OCTIC_F32))) return e;
  if (!ddense) return OCTIC_ENULL;
  if (M <= 0) return OCTIC_ESHAPE;
  View vx = make_view<void>(x), vd = make_view<void>(dx);
  power_spectrum_bwd_kernel<<<grid_for(M * 6 * (c / 8)), 256, 0, (hipStream_t)stream>>>(ddense, vx, vd, M, c);
  return launch_status();
}

int octic_im2col_patches(const float* img, void* patches, int64_t B, int Cin, int Himg, int Wimg, int p, int Kpad,
                         int dtype, void* stream) {
  if (!img || !patches) return OCTIC_ENULL;
  if (B <= 0 || Cin <= 0 || p <= 0 || Himg % p || Wimg % p || Kpad % 8 || Kpad < Cin * p * p) return OCTIC_ESHAPE;
  const int64_t total = B * (Himg / p) * (Wimg / p) * (Kpad / 8);
  const int grid = grid_for(total);
  if (dtype == OCTIC_F32) im2col_kernel<float><<<grid, 256, 0, (hipStream_t)stream>>>(img, (float*)patches, B, Cin, Himg, Wimg, p, Kpad);
  else if (dtype == OCTIC_BF16) im2col_kernel<bf16><<<grid, 256, 0, (hipStream_t)stream>>>(img, (bf16*)patches, B, Cin, Himg, Wimg, p, Kpad);
  else return OCTIC_EDTYPE;
  return launch_status();
}

int octic_colsum_blocks(int64_t M) { return (int)((M + kColsumRows - 1) / kColsumRows); }

int octic_colsum_a1(const octic_view* dy, int64_t M, int c, int dtype, float* partials, float* out, void* stream) {
  int e;
  if ((e = check_c_dt(c, dtype)) || (e = check_view(dy, c, dtype))) return e;
  if (!partials || !out) return OCTIC_ENULL;
  if (M <= 0 || c > 1024) return OCTIC_ESHAPE;
  const int nblk = octic_colsum_blocks(M);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == OCTIC_F32) colsum_partial_kernel<float><<<nblk, 256, 0, s>>>((const float*)dy->ptr[0], dy->ld[0], M, c, partials);
  else if (dtype == OCTIC_BF16) colsum_partial_kernel<bf16><<<nblk, 256, 0, s>>>((const bf16*)dy->ptr[0], dy->ld[0], M, c, partials);
  else return OCTIC_EDTYPE;
  colsum_finish_kernel<<<(c + 15) / 16, 256, 0, s>>>(partials, nblk, c, out);
  return launch_status();
}

int octic_sample_blocks(const void* src, void* dst, const int64_t* idx, int64_t n, int64_t block_bytes, int scatter,
                        void* stream) {
  if (!src || !dst || !idx) return OCTIC_ENULL;
  if (n <= 0 || block_bytes <= 0 || (block_bytes & 15) || n > 65535) return OCTIC_ESHAPE;
  if ((((uintptr_t)src) | ((uintptr_t)dst)) & 15) return OCTIC_EALIGN;
  int64_t gx = (block_bytes / 16 + 255) / 256;
  gx = gx > 64 ? 64 : gx;                      // 64 x n workgroups walk a block: enough to fill the chip at n >= 4
  const dim3 grid((unsigned)gx, (unsigned)n);
  if (scatter) sample_blocks_kernel<1><<<grid, 256, 0, (hipStream_t)stream>>>((const char*)src, (char*)dst, idx, block_bytes);
  else sample_blocks_kernel<0><<<grid, 256, 0, (hipStream_t)stream>>>((const char*)src, (char*)dst, idx, block_bytes);
  return launch_status();
}

}  // extern "C"
