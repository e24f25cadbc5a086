// Shared device/host helpers for the gfx950 octic engine.  CDNA4 only (wave64, MFMA, 160 KiB LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/octic_hip.h"

namespace octic {

// routing overrides (octic_route_override, include/octic_hip.h): one table for the whole library, defined in elementwise.hip
extern int g_route[OCTIC_ROUTE_COUNT];
inline int route(int knob) { return g_route[knob]; }

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

constexpr float kSqrt2Over4 = 0.35355339059327376220f;
constexpr float kSqrt1Over2 = 0.70710678118654752440f;
constexpr float kInvSqrt2Pi = 0.39894228040143267794f;

// A 5-tuple view with compile-time-free helpers.  `c` is the width of a one-dimensional irrep.
struct View {
  char* p[5];
  int64_t ld[5];
};

template <typename T>
__host__ __device__ inline View make_view(const octic_view* v) {
  View r;
  for (int i = 0; i < 5; ++i) {
    r.p[i] = (char*)v->ptr[i];
    r.ld[i] = v->ld[i];
  }
  return r;
}

// Element pointer of logical packed column `e` (0 <= e < 8c, order A1|A2|B1|B2|E0|E1) of token m.
template <typename T>
__device__ inline T* view_ptr(const View& v, int64_t m, int e, int c) {
  if (e < 4 * c) {
    int g = e / c;
    return (T*)v.p[g] + m * v.ld[g] + (e - g * c);
  }
  return (T*)v.p[4] + m * v.ld[4] + (e - 4 * c);
}

__device__ inline float bf2f(bf16 x) { return (float)x; }
__device__ inline bf16 f2bf(float x) { return (bf16)x; }  // v_cvt_pk_bf16_f32: RNE, NaN-preserving

// 8 consecutive elements <-> 8 floats (16-byte accesses)
template <typename T>
__device__ inline void load8(const T* p, float v[8]);
template <>
__device__ inline void load8<float>(const float* p, float v[8]) {
  f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
  v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
}
template <>
__device__ inline void load8<bf16>(const bf16* p, float v[8]) {
  bf16x8 a = *(const bf16x8*)p;
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (float)a[i];
}
template <typename T>
__device__ inline void store8(T* p, const float v[8]);
template <>
__device__ inline void store8<float>(float* p, const float v[8]) {
  f32x4 a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
  *(f32x4*)p = a;
  *(f32x4*)(p + 4) = b;
}
template <>
__device__ inline void store8<bf16>(bf16* p, const float v[8]) {
  bf16x8 a;
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = (bf16)v[i];
  *(bf16x8*)p = a;
}

// 8-point D8 Fourier butterflies (reference d8_utils.py:276-344).  Unscaled; callers
// multiply by (sqrt2/4) once.
__device__ inline void iso_to_reg(const float x[8], float r[8]) {
  float a = x[0] + x[1], b = x[0] - x[1], c = x[2] + x[3], d = x[2] - x[3];
  float e = x[4] + x[5], f = x[4] - x[5], g = x[6] + x[7], h = x[6] - x[7];
  float apc = a + c, amc = a - c, bpd = b + d, bmd = b - d;
  float eph = e + h, emh = e - h, fpg = f + g, fmg = f - g;
  r[0] = apc + eph; r[1] = amc + fmg; r[2] = apc - eph; r[3] = amc - fmg;
  r[4] = bpd - fpg; r[5] = bmd - emh; r[6] = bpd + fpg; r[7] = bmd + emh;
}
__device__ inline void reg_to_iso(const float x[8], float r[8]) {
  float a = x[0] + x[1], b = x[0] - x[1], c = x[2] + x[3], d = x[2] - x[3];
  float e = x[4] + x[5], f = x[4] - x[5], g = x[6] + x[7], h = x[6] - x[7];
  float apc = a + c, cma = c - a, bpd = b + d, bmd = b - d;
  float epg = e + g, gme = g - e, fph = f + h, fmh = f - h;
  r[0] = apc + epg; r[1] = apc - epg; r[2] = bpd + fph; r[3] = bpd - fph;
  r[4] = gme - cma; r[5] = bmd + fmh; r[6] = bmd - fmh; r[7] = gme + cma;
}

__device__ inline float gelu_exact(float r) { return 0.5f * r * (1.0f + erff(r * kSqrt1Over2)); }
__device__ inline float gelu_grad(float r) {
  return 0.5f * (1.0f + erff(r * kSqrt1Over2)) + r * kInvSqrt2Pi * __expf(-0.5f * r * r);
}

// DPP reductions (VALU only, no LDS permutes): sums over aligned groups of 8 / 16 lanes and over the whole wave.
template <int CTRL, int RMASK>
__device__ inline float dpp_add(float v) {
  const int r = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, RMASK, 0xF, false);
  return v + __int_as_float(r);
}
__device__ inline float sum8(float v) {          // every lane: sum over its aligned group of 8 lanes
  v = dpp_add<0xB1, 0xF>(v);                     // quad_perm [1,0,3,2]
  v = dpp_add<0x4E, 0xF>(v);                     // quad_perm [2,3,0,1]
  return dpp_add<0x141, 0xF>(v);                 // row_half_mirror
}
__device__ inline float sum16_from8(float v8) { return dpp_add<0x140, 0xF>(v8); }   // row_mirror
__device__ inline float total_from16(float v16) {  // uniform: sum over the wave, given per-row (16 lane) sums
  v16 = dpp_add<0x142, 0xA>(v16);                // row_bcast15 into rows 1,3
  v16 = dpp_add<0x143, 0xC>(v16);                // row_bcast31 into rows 2,3
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v16), 63));
}
__device__ inline float wave_total(float v) { return total_from16(sum16_from8(sum8(v))); }
__device__ inline float wave_sum(float v) { return wave_total(v); }


// LDS-DMA of 16 bytes per lane (global_load_lds_dwordx4) as inline asm: lane L's 16 bytes land at lds_dst + 16 L
// (lds_dst wave-uniform).  Why not __builtin_amdgcn_global_load_lds: hipcc tracks the builtin as a pending LDS write and
// puts `s_waitcnt vmcnt(0)` in front of the next LDS read it cannot prove disjoint (every ds_read_b64_tr_b16 builtin of
// the weight-gradient ring: the whole DMA queue, including the tile issued a few instructions earlier, was drained at
// every step).  The kernels order DMA and reads themselves with counted `s_waitcnt vmcnt(n)` + `s_barrier`.
__device__ __forceinline__ unsigned lds_offset(const void* p) { return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)p; }
__device__ __forceinline__ void dma16_to_lds(unsigned lds_dst, const void* src) {
  lds_dst = __builtin_amdgcn_readfirstlane(lds_dst);
  // M0 is compiler-reserved (movrel indexing, its own *_load_lds builtins) and a clobber entry for it is only a warning:
  // every statement that writes M0 saves and restores it itself (two scalar moves)
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(lds_dst), "v"(src) : "memory");
}

// ---- per-device one-time setup ---------------------------------------------------------------
// hipFuncSetAttribute (the > 64 KiB dynamic-LDS opt-in) applies to the CURRENT device only and the CU count is a
// property of a device: both are keyed on hipGetDevice(), so a process that drives a second GPU gets its own opt-in
// and its own grid sizes (round-3 advisor finding: process-wide `static bool done`).
inline int current_device() {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess) { (void)hipGetLastError(); d = 0; }
  return (d < 0 || d >= 64) ? 0 : d;
}
struct DeviceOnce {               // `static DeviceOnce once; if (once.first()) { ...setup for this device... }`
  unsigned long long mask = 0;
  bool first() {
    const unsigned long long bit = 1ull << current_device();
    if (mask & bit) return false;
    mask |= bit;
    return true;
  }
};
inline int device_cus() {         // CUs of the current device (256 when no device is visible)
  static int cus[64] = {};
  const int d = current_device();
  if (!cus[d]) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || n <= 0) n = 256;
    (void)hipGetLastError();
    cus[d] = n;
  }
  return cus[d];
}

// ---- host-side argument checks -------------------------------------------------------------
inline int elem_size(int dtype) { return dtype == OCTIC_BF16 ? 2 : 4; }

inline int check_view(const octic_view* v, int c, int dtype) {
  if (!v) return OCTIC_ENULL;
  const int es = elem_size(dtype);
  for (int i = 0; i < 5; ++i) {
    if (!v->ptr[i]) return OCTIC_ENULL;
    if (((uintptr_t)v->ptr[i]) & 15) return OCTIC_EALIGN;
    if ((v->ld[i] * es) & 15) return OCTIC_EALIGN;
    if (v->ld[i] < (i < 4 ? c : 4 * c)) return OCTIC_ESHAPE;
  }
  return OCTIC_OK;
}

inline int check_c(int c) { return (c > 0 && (c % 8) == 0) ? OCTIC_OK : OCTIC_ESHAPE; }
// GEMM paths move 16-byte chunks of the compute dtype: f32 only needs c % 4 == 0
inline int check_c_dt(int c, int dtype) {
  if (c <= 0) return OCTIC_ESHAPE;
  return (c % (dtype == OCTIC_F32 ? 4 : 8)) == 0 ? OCTIC_OK : OCTIC_ESHAPE;
}

inline int launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? OCTIC_OK : (int)e;
}

}  // namespace octic
