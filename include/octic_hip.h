/*
 * octic_hip.h — C ABI of the MI355X (gfx950) octic-ViT block engine.
 *
 * The reference (davnords/octic-vits) is pure Python; its replaceable surface for the hot path
 * is the set of torch modules in octic_vits/d8_layers.py and the one custom-op boundary
 * TritonGeluD8Function (octic_vits/d8_gelu.py:456-478).  This header is the boundary a
 * maintainer binds instead (ctypes stub in INTEGRATION.md): plain pointers, sizes and a HIP
 * stream — no torch types.  Every entry point cites the reference code it replaces.
 *
 * Conventions
 *  - All pointers are DEVICE pointers.  Nothing here allocates, frees or synchronises; work is
 *    enqueued on `stream` (a hipStream_t passed as void*), so calls are graph-capturable.
 *  - Return value: 0 on success, a negative OCTIC_E* code on a rejected argument, or the positive
 *    hipError_t of a failed launch.  octic_strerror() renders either.
 *  - An octic feature of M token rows and D = 8c channels is a 5-tuple (A1,A2,B1,B2:[M,c];
 *    E:[M,2,2c]) exactly as in the reference (d8_layers.py:64-81,111-112).  It is passed as an
 *    octic_view: five base pointers plus five row strides (in elements).  Two layouts matter:
 *      tuple  : five separate contiguous tensors            ld = {c,c,c,c,4c}
 *      packed : one [M, 8c] row  [A1|A2|B1|B2|E_row0|E_row1]  ptr[i] = base + i*c (i<4),
 *               ptr[4] = base + 4c, ld = {8c,...}  — the engine's native HBM layout
 *               ("(B, tokens, irrep, channel)"): one token = one contiguous row.
 *    Row r of E for token m starts at ptr[4] + m*ld[4] + r*2c.  E[..,r,0:c] / E[..,r,c:2c] are the
 *    two E copies (8-tuple entries x(4+r) / x(6+r), d8_utils.py:358-385).
 *  - c must be a multiple of 8 (16-byte vectors in bf16); every base pointer and row stride
 *    must keep rows 16-byte aligned.  Violations return OCTIC_EALIGN/OCTIC_ESHAPE — they are never
 *    silently routed to a slower path.
 *  - dtype codes: OCTIC_F32 (exact f32 MFMA path, used for the reference's fp32 equivariance
 *    tolerances) and OCTIC_BF16 (bf16 operands, f32 accumulate — the training path).
 */
#ifndef OCTIC_HIP_H
#define OCTIC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OCTIC_ABI_VERSION 20

enum { OCTIC_F32 = 0, OCTIC_BF16 = 1 };

enum {
  OCTIC_OK = 0,
  OCTIC_ESHAPE = -1,   /* c not a multiple of 8, non-positive sizes, heads not dividing c ... */
  OCTIC_EALIGN = -2,   /* pointer or row stride breaks 16-byte row alignment */
  OCTIC_EDTYPE = -3,   /* unsupported dtype combination */
  OCTIC_ENULL = -4,    /* required pointer is NULL */
  OCTIC_EWORKSPACE = -5 /* workspace too small (see the *_workspace_bytes query) */
};

typedef struct {
  void* ptr[5];   /* A1, A2, B1, B2, E */
  int64_t ld[5];  /* row stride in ELEMENTS (E: stride between tokens, the 2 rows are adjacent) */
} octic_view;

int octic_abi_version(void);
const char* octic_strerror(int code);

/* ---- routing overrides (measurement / tests) --------------------------------------------------
 * Every entry point chooses its kernel and tiling from the shapes alone.  The alternatives it chooses between are all
 * shipped and parity-tested; this ONE call forces a choice so that an A/B or a test can run the other kernel on the
 * same operands.  It is the library's only process-global mutable state: not thread-safe against concurrent launches,
 * value 0 = automatic (the default), returns the previous value (OCTIC_ESHAPE for an unknown knob).              */
enum {
  OCTIC_ROUTE_DENSE_TILE = 0,      /* octic_dense_gemm_nt plain mode: 4 = 256 x 256 tile, 5 = 256 x 320 tile              */
  OCTIC_ROUTE_DENSE_SPLIT = 1,     /* octic_dense_gemm_nt: n = K-split of the last partial round (1 = unsplit, in front)  */
  OCTIC_ROUTE_WGRAD_SLABS = 2,     /* octic_dense_wgrad_tn: number of row slabs                                           */
  OCTIC_ROUTE_WGRAD_TILE = 3,      /* octic_dense_wgrad_tn: 256 | 320 = tile width along K                                */
  OCTIC_ROUTE_LINEAR_RING = 4,     /* octic_linear_d8_fwd: 1 = ring kernel for every shape (no W-stationary kernel)       */
  OCTIC_ROUTE_RING_EVEN = 5,       /* octic_linear_d8_fwd ring kernel: 1 = even item spread instead of the planned order  */
  OCTIC_ROUTE_ATTN_LEGACY = 6,     /* octic_attn_*: 1 = the two-kernel online-softmax family for every shape              */
  OCTIC_ROUTE_ATTN_ONLINE = 7,     /* octic_attn_fwd*: 1 = persistent online-softmax forward instead of the one-shot one  */
  OCTIC_ROUTE_ATTN_BWD_PAIR = 8,   /* octic_attn_bwd*: 1 = the dq + dkv kernel pair instead of the single-pass backward   */
  OCTIC_ROUTE_DENSE_IMAGE = 9,     /* octic_dense_gemm_nt_tokens: 1 = per-image panels wherever legal, 2 = never, 3 = plain only */
  OCTIC_ROUTE_DENSE_CLS2 = 10,     /* class-token rows of per-image launches as two launches (K split): 1 = never, 2 = wherever legal */
  OCTIC_ROUTE_COUNT = 11
};
int octic_route_override(int knob, int value);

/* ---- D8 GELU -------------------------------------------------------------------------------
 * Replaces d8_gelu_fwd / d8_gelu_bwd (Triton, d8_gelu.py:104-196, 210-331, 333-453) and the
 * torch twin GeluD8 (d8_layers.py:98-102): per (row, channel j) gather the 8 isotypic
 * components, iso->regular butterfly, exact-erf GELU, regular->iso.  Arithmetic is f32 in
 * registers for both dtypes (the reference's bf16 path does the butterflies in bf16,
 * d8_gelu.py:11-26; f32 is strictly more accurate).  y may alias x.                       */
int octic_gelu_d8_fwd(const octic_view* x, const octic_view* y, int64_t M, int c, int dtype, void* stream);
/* gin = F( gelu'(F^-1 x) * F^-1 g )   (d8_gelu.py:283-321) */
int octic_gelu_d8_bwd(const octic_view* g, const octic_view* x, const octic_view* gin, int64_t M, int c,
                      int dtype, void* stream);

/* ---- LayerNormD8 (+AffineD8) -----------------------------------------------------------------
 * Replaces LayerNormD8.forward (d8_layers.py:166-186): per-segment mean removal, one shared
 * std = (sqrt2/4)*sqrt(sum_1D var + mean_rows var_E + eps), then alpha (and beta on A1).
 * x is f32 (the residual stream); y is out_dtype.  alpha[i] may be NULL as a group (=> no affine,
 * elementwise_affine=False); beta may be NULL.  stats (optional, [M,8] f32: 6 means, rstd, 0) is
 * what the backward needs.                                                                     */
int octic_layernorm_d8_fwd(const octic_view* x, const octic_view* y, const float* const alpha[5],
                           const float* beta, float* stats, int64_t M, int c, float eps, int out_dtype,
                           void* stream);
/* dx = (dres ? dres : 0) + LN'(g).  g is g_dtype, x/dx/dres are f32.  Column sums for the affine
 * parameters are written as per-block partial slabs into `partials` ([nblk, 2, 8c] f32,
 * nblk = octic_layernorm_d8_bwd_blocks(M)); reduce them with octic_layernorm_d8_bwd_finish.     */
int octic_layernorm_d8_bwd_blocks(int64_t M);
int octic_layernorm_d8_bwd(const octic_view* g, const octic_view* x, const float* stats,
                           const float* const alpha[5], const octic_view* dres, const octic_view* dx,
                           float* partials, int64_t M, int c, int g_dtype, void* stream);
/* octic_layernorm_d8_bwd_cast: octic_layernorm_d8_bwd for a bf16 cotangent on packed rows with c in {32,...,160}
 * (other arguments: OCTIC_ESHAPE, call the two kernels), which also stores gcast[row] = bf16(rs[row / rows_per_sample] *
 * dx[row]) (packed rows of 8c, rs may be NULL): the drop-path-scaled bf16 cotangent that the backward of the
 * residual-fused LinearD8 in front of this norm needs (d8_layers.py:698-707 chained over two branches) - what
 * octic_cast_rowscale makes of dx in a pass of its own.                                                          */
int octic_layernorm_d8_bwd_cast(const octic_view* g, const octic_view* x, const float* stats, const float* const alpha[5],
                                const octic_view* dres, const octic_view* dx, float* partials, int64_t M, int c,
                                const float* rs, int64_t rows_per_sample, void* gcast, void* stream);
int octic_layernorm_d8_bwd_finish(const float* partials, int nblk, int c, float* const dalpha[5], float* dbeta,
                                  void* stream);
/* njobs of the reductions above in ceil(njobs / 48) launches, bit-identical to njobs calls (see octic_dense_finish_batch). */
typedef struct octic_ln_finish_job {
  const float* partials; /* [nblk][2][8c] */
  float* dalpha[5];      /* each may be NULL */
  float* dbeta;          /* may be NULL */
  int nblk;
  int c;
} octic_ln_finish_job;
int octic_layernorm_d8_bwd_finish_batch(const octic_ln_finish_job* jobs, int njobs, void* stream);

/* Sample blocks between a full batch and a compacted one (the opt-in batch compaction of stochastic depth, drop_path_d8 of
 * d8_layers.py:249-270 computed on the kept samples only): scatter == 0: dst[i] = src[idx[i]], scatter != 0: dst[idx[i]] =
 * src[i], i < n; a block = block_bytes (multiple of 16) = one sample's token rows; idx int64 on the device, distinct.        */
int octic_sample_blocks(const void* src, void* dst, const int64_t* idx, int64_t n, int64_t block_bytes, int scatter,
                        void* stream);

/* ---- LinearD8 (irrep-blocked GEMM on MFMA) ----------------------------------------------------
 * Replaces LinearD8.forward (d8_layers.py:124-127) = five nn.Linear calls, as ONE launch:
 *     y_g[m, n] = resid_g[m, n] + rs[token(m)/rows_per_sample] * cs_g[n] * ( sum_k x_g[m,k] W_g[n,k] + bias[n] (g==A1) )
 * W_g: [cout_g, cin_g] row-major in `dtype` (nn.Linear layout), g = A1,A2,B1,B2 (c wide) and E
 * (2c wide, shared by both E rows, d8_layers.py:127).  resid (out_dtype view), rs (f32 per
 * sample: drop-path mask/keep-prob, d8_layers.py:256-270), cs (f32 per channel: AffineD8 /
 * LayerScaleD8 gamma, d8_layers.py:147-158,205-212) and bias (f32 [cout]) are each optional
 * (NULL) — together they fuse `x + drop_path(gamma * linear(h))` (d8_layers.py:704-707) into
 * the GEMM epilogue.  The same entry point computes the input gradient when given the
 * transposed weights (dX = dY W  ==  linear with W^T).  x and W are `dtype`; y/resid are
 * out_dtype.  Supported (dtype,out_dtype): (F32,F32), (BF16,BF16), (BF16,F32).             */
int octic_linear_d8_fwd(const octic_view* x, const void* const w[5], const float* bias, const octic_view* y,
                        const octic_view* resid, const float* rs, int64_t rows_per_sample,
                        const float* const cs[5], int64_t M, int cin, int cout, int dtype, int out_dtype,
                        void* stream);

/* Compute-dtype copies of the f32 master weights in ONE launch per layer: wb = [W_A1|W_A2|W_B1|W_B2|W_E]
 * (forward), wt = each matrix transposed with the layer-scale folded in, wt_g[k][n] = cs_g[n] W_g[n][k]
 * (input gradient dX = dY diag(cs) W).  Either output may be NULL; cs may be NULL.                 */
int octic_linear_d8_prep(const float* const w32[5], const float* const cs[5], int cin, int cout, void* wb, void* wt,
                         int dtype, void* stream);
/* The same for many layers in ONE launch (after an optimizer step).  items_dev: device array sorted by block_begin;
 * item i owns workgroups [block_begin, block_begin + block_count) of the total_blocks launched, with block_count =
 * octic_linear_d8_prep_batch_blocks(cin, cout) (one 64 x 64 tile of one irrep's matrix each); wb / wt / cs[] entries
 * may be NULL as in octic_linear_d8_prep.                                                              */
typedef struct octic_prep_item {
  const float* w[5];
  const float* cs[5];
  void* wb;
  void* wt;
  int32_t cin, cout;
  int32_t block_begin, block_count;
} octic_prep_item;
int octic_linear_d8_prep_batch_blocks(int cin, int cout);
int octic_linear_d8_prep_batch(const octic_prep_item* items_dev, int n_items, int total_blocks, int dtype, void* stream);

/* Output-tile width (32*NT) the launcher picks for this problem; the kernel instantiation that runs is
 * linear_d8_kernel<TIN, TOUT, NT> — exposed so profilers/benchmarks can name it.                 */
int octic_linear_d8_tile_n(int64_t M, int cin, int cout);
/* Host-only query (no device call): the workgroup -> item order of a long-K ("ring") launch.  A launch is `ngroups` (<= 5)
 * item classes, class g with items[g] items of ksteps[g] K-steps each (class 0 = the long one, the E irrep); workgroups are
 * dispatched in blockIdx order round-robin over the 8 XCDs to slots_per_xcd slots each, so the order is a schedule.
 * out_group / out_item (sum(items) entries each) receive, per workgroup, the item it runs.  Returns 1 if the planned order
 * applies, 0 for the even spread, negative on bad arguments.                                                            */
int octic_linear_d8_ring_order(int ngroups, const int* items, const int* ksteps, int slots_per_xcd, int* out_group,
                               int* out_item);

/* Weight gradient  G_g[n,k] = sum_rows dy_g[row,n] x_g[row,k]  (E: both rows).  Reduction over the
 * M (2M) rows is split over `splits` row ranges whose f32 partial slabs go to `workspace`
 * (octic_linear_d8_wgrad_workspace_bytes).  The finish kernel sums the slabs in a fixed order
 * (bitwise reproducible) and applies the layer-scale chain rule when cs != NULL:
 *     dW_g[n,k] = cs_g[n] * G_g[n,k]
 *     dcs_g[n]  = sum_k W_g[n,k] G_g[n,k]  + (g==A1 ? bias[n]*dysum[n] : 0)
 *     dbias[n]  = cs_A1[n] * dysum[n]          with dysum = column sums of dy_A1
 * which is the gradient of  y = resid + rs*cs*(xW^T+b)  w.r.t. W, cs, b when dy = rs*dL/dy.
 * w32 (f32 master weights) and bias are only read when cs != NULL.                            */
/* octic_linear_d8_wgrad_has_colsum: 1 if the wgrad launch for this shape also leaves the column sums of the invariant
 * irrep's dY (the bias gradient) in the workspace; _finish then takes them when called with dysum == NULL and
 * octic_colsum_a1 is not needed.                                                                       */
int octic_linear_d8_wgrad_has_colsum(int cin, int cout, int dtype);
int64_t octic_linear_d8_wgrad_workspace_bytes(int cin, int cout, int splits);
int octic_linear_d8_wgrad_splits(int64_t M, int cin, int cout);
int octic_linear_d8_wgrad_tile(int64_t M, int cin, int cout);   /* tile width 32*TT of wgrad_kernel<TIN, TT> */
int octic_linear_d8_wgrad(const octic_view* x, const octic_view* dy, int64_t M, int cin, int cout, int dtype,
                          float* workspace, int splits, void* stream);
int octic_linear_d8_wgrad_finish(const float* workspace, int splits, int cin, int cout,
                                 const float* const w32[5], const float* const cs[5], const float* bias,
                                 const float* dysum, float* const dw[5], float* const dcs[5], float* dbias,
                                 void* stream);
/* njobs of the finishes above in ceil(njobs / 8) launches, bit-identical to njobs calls (see octic_dense_finish_batch);
 * has_cs = 0 stands for cs == NULL (w32, cs, dcs ignored).                                                          */
typedef struct octic_wgrad_finish_job {
  const float* workspace;
  const float* w32[5];
  const float* cs[5];
  const float* bias;
  const float* dysum;
  float* dw[5];
  float* dcs[5];
  float* dbias;
  int splits, cin, cout, has_cs;
} octic_wgrad_finish_job;
int octic_linear_d8_wgrad_finish_batch(const octic_wgrad_finish_job* jobs, int njobs, void* stream);

/* Column sums of the A1 block of dy (bias gradient, bias exists on A1 only: d8_layers.py:117-122).
 * out[n] = sum_m dy_A1[m,n]; `partials` holds octic_colsum_blocks(M) * c floats.               */
int octic_colsum_blocks(int64_t M);
int octic_colsum_a1(const octic_view* dy, int64_t M, int c, int dtype, float* partials, float* out, void* stream);

/* y = rs[token/rows_per_sample] * x, converted f32 -> out_dtype (cotangent entering a fused
 * residual branch: dL/d(branch) = drop-path mask * dL/dx_out).  rs may be NULL (pure cast).     */
int octic_cast_rowscale(const octic_view* x, const octic_view* y, const float* rs, int64_t rows_per_sample,
                        int64_t M, int c, int out_dtype, void* stream);

/* ---- attention head packing (d8_layers.py:631-643, 650-656) ------------------------------------
 * pack:  qkv view (3*8c channels: per irrep [q|k|v] thirds) -> q,k,v  [B,H,T,8w], w = c/H, per-head
 *        vector [A1 w|A2 w|B1 w|B2 w|E_row0 2w|E_row1 2w].  qkv_out = 3 consecutive [B,H,T,8w] arrays.
 * unpack: o [B,H,T,8w] -> view with 8c channels.  Each is the other's adjoint (n_s = 3 / 1).
 * heads[s] (s < n_s) are n_s separately allocated [B,H,T,8w] arrays (q, k, v — or their gradients, which
 * autograd hands back as three unrelated tensors).                                              */
int octic_attn_pack_heads(const octic_view* qkv, void* const heads[3], int64_t B, int64_t T, int H, int c, int n_s,
                          int dtype, void* stream);
int octic_attn_unpack_heads(void* const heads[3], const octic_view* y, int64_t B, int64_t T, int H, int c, int n_s,
                            int dtype, void* stream);

/* ---- attention core (bf16) ---------------------------------------------------------------------
 * o = softmax(scale * q k^T) v per (batch, head); replaces F.scaled_dot_product_attention in AttentionD8
 * (d8_layers.py:645-648) and the standard blocks (deit/vit.py:41-45).  Element (b,h,t,d) of q/k/v is at
 * base + b*sB + h*sH + t*sT + d (one stride set for the three, so [B,H,T,hd] and the [B,T,3,H,hd] views of a
 * fused qkv tensor both work); o likewise with oB/oH/oT.  lse ([B,H,T] f32, may be NULL) receives the
 * log2-domain log-sum-exp needed by the backward.  T <= 320, hd a multiple of 16 (<= 128); otherwise
 * OCTIC_ESHAPE (callers keep torch SDPA for such shapes).                                            */
int octic_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse, int64_t B, int H, int T, int hd,
                   int64_t sB, int64_t sH, int64_t sT, int64_t oB, int64_t oH, int64_t oT, float scale, void* stream);

/* Backward of octic_attn_fwd (P recomputed from q, k and lse; nothing T x T is stored).  o / dout share the stride
 * set (oB,oH,oT); dq/dk/dv share (gB,gH,gT).  delta: [B,H,T] f32 (row sums of dout*o).  phase bit 0: query-owned
 * kernel (writes delta and dq); bit 1: key-owned kernel (reads delta, writes dk and dv); 3 = both, in that order. */
int octic_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
                   float* delta, void* dq, void* dk, void* dv, int64_t B, int H, int T, int hd, int64_t sB, int64_t sH,
                   int64_t sT, int64_t oB, int64_t oH, int64_t oT, int64_t gB, int64_t gH, int64_t gT, float scale,
                   int phase, void* stream);

/* AttentionD8 straight on packed rows - reference octic_vits/d8_layers.py:631-656 (head split of the five-irrep
 * projection output, F.scaled_dot_product_attention, re-assembly of the irreps) WITHOUT the pack / unpack copies:
 * qkv = LinearD8 output [B, T, 3*8c] (row stride ld_qkv elements), o = packed [B, T, 8c] input of the output
 * projection.  c / H must be 10 (head_dim 80), bf16.  lse [B,H,T] as in octic_attn_fwd. */
int octic_attn_fwd_packed(const void* qkv, void* o, float* lse, int64_t B, int H, int T, int c, int64_t ld_qkv,
                          int64_t ld_o, float scale, void* stream);
/* Backward of octic_attn_fwd_packed (autograd of d8_layers.py:631-656): dqkv packed like qkv (row stride ld_g)
 * receives dq | dk | dv, dout packed like o; phase as in octic_attn_bwd. */
int octic_attn_bwd_packed(const void* qkv, const void* o, const void* dout, const float* lse, float* delta, void* dqkv,
                          int64_t B, int H, int T, int c, int64_t ld_qkv, int64_t ld_o, int64_t ld_g, float scale,
                          int phase, void* stream);

/* ---- octic -> standard hand-off (model.py:196-200) ---------------------------------------------
 * hybrid:    dense[m, :] = cat(A1,A2,B1,B2, E[0,:c], E[1,:c], E[0,c:], E[1,c:])   (8-tuple order,
 *            d8_utils.py:370-385; the following standard blocks' weights depend on it)
 * invariant: dense[m, :] = cat(A1,|A2|,|B1|,|B2|, sqrt(E[0]^2+E[1]^2))  [6c]  (PowerSpectrumInvariant,
 *            d8_invariantization.py:49-64)
 * x is f32 (residual stream); dense is out_dtype.  The *_bwd forms take dense f32 gradients.    */
int octic_handoff_cat_fwd(const octic_view* x, void* dense, int64_t M, int c, int out_dtype, void* stream);
int octic_handoff_cat_bwd(const float* ddense, const octic_view* dx, int64_t M, int c, void* stream);
int octic_power_spectrum_fwd(const octic_view* x, void* dense, int64_t M, int c, int out_dtype, void* stream);
int octic_power_spectrum_bwd(const float* ddense, const octic_view* x, const octic_view* dx, int64_t M, int c,
                             void* stream);

/* ---- lift patch embedding (d8_layers.py:284-486, model.py:172-181) -----------------------------
 * im2col: img [B,Cin,Himg,Wimg] f32 -> patches [B*G*G, Kpad] `dtype`, column = (ch, py, px), zero
 * padded to Kpad (a multiple of 8 >= Cin*p*p).  The conv with stride = kernel is then one GEMM
 * against the symmetry-expanded weights; use octic_lift_gemm: out_packed[b, tok0 + n, :] =
 * patches[b*G*G + n, :] W^T + bias + pos[n, :]  with W:[8c, Kpad] rows in packed channel order and
 * pos:[G*G, 8c] f32 (unfolded positional embedding, may be NULL); rows [0,tok0) of every sample
 * are left untouched (cls token).  out is f32 [B, tok0+G*G, 8c].                               */
int octic_im2col_patches(const float* img, void* patches, int64_t B, int Cin, int Himg, int Wimg, int p,
                         int Kpad, int dtype, void* stream);
int octic_lift_gemm(const void* patches, const void* w, const float* bias, const float* pos, float* out,
                    int64_t B, int64_t n_patches, int tok0, int Kpad, int D, int dtype, void* stream);
/* dW[n,k] = sum_rows dout[row, n] patches[row, k]  with dout the [rows, D] cotangent of the patch
 * tokens in `dtype` (cls rows removed by the caller); f32 result, split-row slabs like wgrad.
 * (bias here is a full [D] vector, zero outside the A1 block.)                                   */
int64_t octic_lift_wgrad_workspace_bytes(int Kpad, int D, int splits);
int octic_lift_wgrad(const void* patches, const void* dout, float* dw, float* workspace, int splits, int64_t rows,
                     int Kpad, int D, int dtype, void* stream);

/* ---- fused multi-tensor LAMB + EMA ----------------------------------------------------------------
 * One optimizer step of the reference recipe (apex FusedLAMB via timm create_optimizer 'fusedlamb',
 * experiments/train_deit.py:42; timm ModelEma, deit/main.py:344-351) over ALL tensors in five launches:
 * global grad-norm clip to max_grad_norm, Adam moments with bias correction, + wd*p, per-tensor trust ratio
 * |p|/|u| (only where wd != 0), p -= lr*ratio*u, ema += (1-decay)(p-ema).  The gradient buffers are
 * overwritten (they hold the update between the two passes).  Tables are device arrays: per-tensor
 * pointers p,g,m,v,ema (ema may be NULL), per-tensor weight decay, and a chunk list
 * (tensor id, element offset, length) with tensor_chunk_begin[ntensors+1] giving each tensor's chunk
 * range.  workspace: octic_lamb_workspace_floats() f32, zero-initialised by the caller once; after the call
 * workspace[1] = global grad norm (the number deit/engine.py:84 logs), workspace[2] = 1 if the step was SKIPPED
 * because that norm is not finite (parameters, moments, EMA and bf16 copies untouched: the reference exits
 * before optimizer.step() on a non-finite loss, deit/engine.py:67-71), workspace[3] = applied steps,
 * workspace[6] = skipped steps so far.  step > 0: the bias-correction step t given by the host; step == 0: t is
 * the device-side counter workspace[3] (+1 per applied step), so a hipGraph capture of the call replays
 * correctly.  bf16_shadow (may be NULL; entries may be NULL): per-tensor bf16
 * buffers that receive a rounded copy of the updated parameter in the same pass - the compute-dtype weights
 * torch.autocast would otherwise re-cast at every use.                                              */
int64_t octic_lamb_workspace_floats(int ntensors, int nchunks);
int octic_lamb_step(void* const* p, void* const* g, void* const* m, void* const* v, void* const* ema, const float* wd,
                    const int* chunk_tensor, const int64_t* chunk_off, const int* chunk_len,
                    const int* tensor_chunk_begin, int ntensors, int nchunks, float* workspace, float lr, float beta1,
                    float beta2, float eps, float max_grad_norm, int step, float ema_decay, void* const* bf16_shadow,
                    void* stream);
/* The same fused step WITHOUT the layer-wise trust ratio: AdamW with decoupled weight decay, p -= lr (mhat / (sqrt(vhat) +
 * eps) + wd p) - torch.optim.AdamW as the DINOv2 recipe builds it (dinov2/train/train.py:60-61), with clip_grad_norm_ of the
 * tensors of this call to max_grad_norm (train.py:274-279: one call per sub-model) and the teacher's EMA (`ema` = the
 * teacher's parameters, ema_decay = the momentum of the step: ssl_meta_arch.py:356-367) in the same two passes.           */
int octic_adamw_step(void* const* p, void* const* g, void* const* m, void* const* v, void* const* ema, const float* wd,
                     const int* chunk_tensor, const int64_t* chunk_off, const int* chunk_len,
                     const int* tensor_chunk_begin, int ntensors, int nchunks, float* workspace, float lr, float beta1,
                     float beta2, float eps, float max_grad_norm, int step, float ema_decay, void* const* bf16_shadow,
                     void* stream);

/* ---- standard (non-equivariant) half of the hybrid: row kernels around the library GEMMs ----------------
 * The reference's standard blocks (deit/models_v2.py Layer_scale_init_Block, used for the second half of the
 * depth by octic_vits/model.py:130-150) are  x = x + drop_path(gamma * f(LayerNorm(x))).  The four projections
 * stay on the BLAS library; these entry points replace the eager chain around them.  Rows are dense
 * [rows, d] with d % 4 == 0, d <= 2048; x / residual stream / statistics are f32, branch tensors `dtype`.
 *
 * octic_dense_layernorm_fwd: y = (x-mean)*rstd*w + b (w, b may be NULL), stats[rows,2] = (mean, rstd).
 * octic_dense_layernorm_bwd: dx = rstd*(g - mean(g) - xhat*mean(g*xhat)) + dres with g = gy*w (dres = cotangent
 *   of the residual path, may be NULL); partials (may be NULL) receives octic_dense_blocks(rows) slabs [2][d]
 *   holding sum gy*xhat and sum gy; octic_dense_finish reduces them into dw / db.
 * octic_scale_residual_fwd: out = x + rs[row / rows_per_scale] * gamma[col] * y  (rs, gamma may be NULL).
 * octic_scale_residual_bwd: gy = rs*gamma*gout in y's dtype; slabs [2][d] of sum rs*gout*y (= d gamma) and
 *   sum rs*gout (times gamma = bias gradient of the projection that produced y).
 * octic_dense_finish: out0[j] = sum_b slab[b][0][j];  out1[j] = scale1[j] * sum_b slab[b][1][j]
 *   (out0, out1, scale1 may each be NULL).                                                              */
int octic_dense_blocks(int64_t rows);
/* octic_dense_resid_layernorm_fwd: xout = x + rs[row / rows_per_scale] * gamma * yb (the tail of one branch of
 * Layer_scale_init_Block, deit/vit.py:131-134) and y = LayerNorm(xout) (the norm that opens the next branch,
 * deit/vit.py:132-133) in one row pass; stats[rows,2] = (mean, rstd) of xout.  Same arithmetic as
 * octic_scale_residual_fwd followed by octic_dense_layernorm_fwd.                                            */
int octic_dense_resid_layernorm_fwd(const float* x, const void* yb, int yb_dtype, const float* gamma, const float* rs,
                                    int64_t rows_per_scale, float* xout, void* y, int y_dtype, const float* w,
                                    const float* b, float* stats, int64_t rows, int d, float eps, void* stream);
int octic_dense_layernorm_fwd(const float* x, void* y, int y_dtype, const float* w, const float* b, float* stats,
                              int64_t rows, int d, float eps, void* stream);
int octic_dense_layernorm_bwd(const void* gy, int g_dtype, const float* x, const float* w, const float* stats,
                              const float* dres, float* dx, float* partials, int64_t rows, int d, void* stream);
int octic_dense_finish(const float* partials, int nblocks, int d, float* out0, float* out1, const float* scale1,
                       void* stream);
/* octic_dense_finish_batch: njobs reductions of the kind above in ceil(njobs / 64) launches, bit-identical to njobs calls of
 * octic_dense_finish (same summation order).  For callers that can postpone the parameter-gradient reductions of a backward
 * pass to its end (nothing reads them before the optimizer): 96 five-microsecond launches per ViT-H step become two.       */
typedef struct octic_finish_job {
  const float* partials; /* [nblocks][2][d] slabs */
  float* out0;           /* [d] or NULL */
  float* out1;           /* [d] or NULL */
  const float* scale1;   /* [d] or NULL */
  int nblocks;
  int d;
} octic_finish_job;
int octic_dense_finish_batch(const octic_finish_job* jobs, int njobs, void* stream);
/* octic_dense_layernorm_bwd_tail: octic_dense_layernorm_bwd (bf16 gy) followed by octic_scale_residual_bwd on its result,
 * one row pass for d = 256, 512, ... 1280 (other d: OCTIC_ESHAPE, call the two): dx as above; gyb = rs*gamma*dx in bf16
 * (cotangent of the bf16 branch output yb whose residual add produced the normalised stream); partials / partials2:
 * octic_dense_blocks(rows) slabs [2][d] each, as the two kernels leave them (both may be NULL).  The autograd of
 * `x = x + drop_path(gamma * f(norm(x)))` chained over two branches (deit/vit.py:131-134).                       */
int octic_dense_layernorm_bwd_tail(const void* gy, const float* x, const float* w, const float* stats, const float* dres,
                                   float* dx, float* partials, const void* yb, const float* gamma, const float* rs,
                                   int64_t rows_per_scale, void* gyb, float* partials2, int64_t rows, int d, void* stream);
/* octic_dense_gelu_bwd: dh = gelu'(h) * g (exact erf GELU, bf16 [rows, d], d % 8 == 0) and, when partials != NULL,
 * octic_dense_gelu_blocks() slabs [d] of column sums of dh (bias gradient of the projection that produced h;
 * reduce with octic_dense_finish(partials, blocks, d/2, out, out + d/2, NULL)).  Replaces GeluBackward + the
 * bias-gradient reduction of the standard MLP (deit/vit.py Mlp).                                          */
int octic_dense_gelu_blocks(void);
int octic_dense_gelu_bwd(const void* h, const void* g, void* dh, float* partials, int64_t rows, int d, void* stream);
/* octic_dense_colsum: octic_dense_gelu_blocks() slabs [d] of f32 column sums of a bf16 [rows, d] tensor (row stride ld
 * elements, d % 8 == 0), in a fixed order; reduce with octic_dense_finish as above.  The bias gradient of the fused-qkv
 * projection (autograd of deit/vit.py:33: grad.sum(0) over the token rows).                                      */
int octic_dense_colsum(const void* g, int64_t rows, int d, int64_t ld, float* partials, void* stream);
int octic_scale_residual_fwd(const float* x, const void* y, int y_dtype, const float* gamma, const float* rs,
                             int64_t rows_per_scale, float* out, int64_t rows, int d, void* stream);
int octic_scale_residual_bwd(const float* gout, const void* y, int y_dtype, const float* gamma, const float* rs,
                             int64_t rows_per_scale, void* gy, float* partials, int64_t rows, int d, void* stream);

/* The same four row kernels with a ROW MAP (int32 [rows]): the compact rows 0..rows-1 of a branch correspond to rows
 * rowmap[r] of a larger f32 tensor - the residual stream (or its cotangent) of which the branch sees only the kept samples:
 * DINOv2's batch-subset stochastic depth (dinov2/layers/block.py:113-140: x[brange] in, index_add back) without the gather
 * and scatter passes.  _fwd_rows reads x[rowmap[r]] (and, xcopy != NULL, leaves the rows as read in a compact copy: what the
 * backward needs once the stream has been edited in place); scale_residual_fwd_rows writes out[rowmap[r]] = x[r] + rs gamma
 * y[r]; scale_residual_bwd_rows reads gout[rowmap[r]]; layernorm_bwd_rows reads dres[rowmap[r]] and writes dx[rowmap[r]]
 * (dres == dx edits the stream's cotangent in place).  rowmap == NULL: the plain kernels.  Rows of a map must be distinct. */
int octic_dense_layernorm_fwd_rows(const float* x, void* y, int y_dtype, const float* w, const float* b, float* stats,
                                   int64_t rows, int d, float eps, const int* rowmap, float* xcopy, void* stream);
int octic_dense_layernorm_bwd_rows(const void* gy, int g_dtype, const float* x, const float* w, const float* stats,
                                   const float* dres, float* dx, float* partials, int64_t rows, int d, const int* rowmap,
                                   void* stream);
int octic_scale_residual_fwd_rows(const float* x, const void* y, int y_dtype, const float* gamma, const float* rs,
                                  int64_t rows_per_scale, float* out, int64_t rows, int d, const int* rowmap, void* stream);
int octic_scale_residual_bwd_rows(const float* gout, const void* y, int y_dtype, const float* gamma, const float* rs,
                                  int64_t rows_per_scale, void* gy, float* partials, int64_t rows, int d, const int* rowmap,
                                  void* stream);

/* bf16 operand copies of nn.Linear weights [N,K] for octic_dense_gemm_nt, all layers in one launch (what autocast's
 * per-use weight casts amount to, deit/engine.py:56): wb = bf16(src) [N,K] (may be NULL) and wt = bf16(src)^T [K,N]
 * (the input-gradient GEMM dX = dY W is then an NT problem too).  src is the f32 master or an existing bf16 copy
 * (src_dtype).  block_begin = running sum of octic_dense_prep_batch_blocks(N, K) over the items.            */
typedef struct octic_dense_prep_item {
  const void* src;
  void* wb;
  void* wt;
  int32_t N, K;
  int32_t block_begin, pad;
} octic_dense_prep_item;
int octic_dense_prep_batch_blocks(int N, int K);
int octic_dense_prep_batch(const octic_dense_prep_item* items_dev, int n_items, int total_blocks, int src_dtype, void* stream);

/* ---- hand-written dense bf16 GEMMs of the standard half -------------------------------------
 * The four projections of the reference's standard block (deit/vit.py:14-56 Attention.qkv / .proj, timm Mlp fc1 / fc2
 * used by Layer_scale_init_Block, deit/vit.py:90-134) and their input gradients are "NT" problems
 *     C[M,N] = A[M,K] . B[N,K]^T          A, B bf16 with K contiguous (lda, ldb = row strides in elements),
 * M token rows, f32 accumulation on v_mfma_f32_16x16x32_bf16, K % 128 == 0, N % 4 == 0.  `mode` selects the fused tail:
 *   0 PLAIN : C = acc + bias                                   (qkv forward; input gradients: bias = NULL)
 *   1 GELU  : C = acc + bias (pre-activation, kept for backward), C2 = gelu(C) exact erf (fc1 + nn.GELU, vit.py:131-134)
 *   2 RESID : C = acc + bias (branch output, kept for d gamma),  OUT = X + rs[row / rows_per_sample] * gamma * C
 *             = x + drop_path(gamma * f(x)) of deit/vit.py:131-134 with f32 residual stream X / OUT [M,N] dense
 *   3 DGELU : C = gelu'(H) * acc   with H the saved pre-activation (fc2 input gradient fused with GELU backward);
 *             colsum (may be NULL): octic_dense_gemm_colsum_rows(M,N,K) slabs [N] f32 whose sum over slabs is the column
 *             sum of C (= fc1's bias gradient), every element written by each launch, fixed order -> octic_dense_finish
 *   4 GELUF : like 1, but C = gelu'(acc + bias) rounded to bf16 - the factor the backward multiplies by - instead of the
 *             pre-activation itself (gelu and gelu' share one erf evaluation in the epilogue);  C2 = gelu(acc + bias)
 *   5 DFACT : like 3 with H = the factor stored by mode 4: C = H * acc, no transcendental in the epilogue (one more bf16
 *             rounding of the factor than mode 1 + 3; same column sums)
 *   6 GELUO : C = gelu(acc + bias) only (the value mode 1 leaves in C2): passes that never run a backward - inference, the
 *             DINOv2 teacher - write half the bytes
 * C / C2 / H are bf16 [M,N] with row stride ldc.  bias, gamma [N] f32 and rs f32 may be NULL.  workspace:
 * octic_dense_gemm_workspace_bytes(M,N,K) bytes (split-K slabs of the last partial round of tiles + counters), ZEROED once
 * by the caller when it is allocated (the kernels re-arm their counters; calls sharing a workspace must be stream-ordered). */
int64_t octic_dense_gemm_workspace_bytes(int M, int N, int K);
int octic_dense_gemm_colsum_rows(int M, int N, int K);
/* Output tile width the kernel will use for this problem and mode: 256 (256 x 256 tile) or 320 (256 x 320 tile: plain mode,
 * N % 320 == 0, chosen where it makes the launch ONE round of workgroups - the N = 1280 problems of ViT-H).  Informational
 * (profilers see two kernel symbols: dense_nt_kernel<mode, 4> and <0, 5>). */
int octic_dense_gemm_tile(int M, int N, int K, int mode);
int octic_dense_gemm_nt(const void* A, const void* B, int M, int N, int K, int64_t lda, int64_t ldb, int mode, void* C,
                        void* C2, int64_t ldc, const float* bias, const float* gamma, const float* rs,
                        int64_t rows_per_sample, const float* X, float* OUT, const void* H, float* colsum,
                        void* workspace, void* stream);
/* The same GEMM for token rows that are whole images: A / C / C2 / H rows are [B, tokens, .] flattened (M = B * tokens), as every
 * projection of deit/vit.py's blocks sees them (x: [B, N, C], vit.py:31-32).  tokens = 257 (ViT-H/14 at 224 x 224: one class
 * token + 256 patches) lets the launch take per-image row panels - panel b = the 256 patch rows of image b, so M = 64 x 257
 * is 64 full panels instead of 64 + a 64-row last panel whose tiles are split along K - and run the B class-token rows
 * (row stride tokens * lda) as a skinny [B, K] x [K, N] launch of their own, wherever the launch model says that is shorter
 * (all modes but 2; K % 128 == 0).  Same results up to the f32 summation order of the class-token rows (bitwise reproducible
 * from launch to launch).  tokens = 0 (or any other value): exactly octic_dense_gemm_nt.  The column-sum slabs of modes 3 / 5
 * then have octic_dense_gemm_plan()'s out[1] rows.  Workspace: octic_dense_gemm_workspace_bytes covers both plans. */
int octic_dense_gemm_nt_tokens(const void* A, const void* B, int M, int N, int K, int64_t lda, int64_t ldb, int mode, void* C,
                               void* C2, int64_t ldc, const float* bias, const float* gamma, const float* rs,
                               int64_t rows_per_sample, const float* X, float* OUT, const void* H, float* colsum,
                               void* workspace, int tokens, void* stream);
/* What octic_dense_gemm_nt_tokens will do for this problem: out[0] = output tile width (256 | 320), out[1] = rows of the
 * column-sum slabs of modes 3 / 5, out[2] = 1 when the launch uses per-image panels + the class-token kernel, out[3] =
 * workgroups of the main launch. */
int octic_dense_gemm_plan(int M, int N, int K, int mode, int tokens, int* out4);

/* Weight gradient of an nn.Linear of the standard half (the autograd of deit/vit.py:33,46 and of timm Mlp.fc1 / fc2):
 *     dW[N,K] = dY[M,N]^T . X[M,K]     f32, nn.Linear layout
 * dY, X bf16 row-major (ldy, ldx row strides in elements), N % 256 == 0, K % 256 == 0 or K % 320 == 0, at most 1024
 * output tiles, M * ld * 2 < 2^31.  The reduction over the M token rows is cut into row slabs (one workgroup per slab and
 * 256 x 256 or 256 x 320 tile, all tiles of a slab walking the same rows in lockstep); the f32 partial tiles are summed in slab order by the
 * last workgroup of a tile (bitwise reproducible).  workspace: octic_dense_wgrad_workspace_bytes(M,N,K) bytes, zeroed
 * once at allocation.                                                                                                 */
int64_t octic_dense_wgrad_workspace_bytes(int M, int N, int K);
/* Tile width along K the launch will use: 256 (256 x 256 tiles, whenever K % 256 == 0) or 320 (256 x 320 tiles: K % 320 == 0
 * only).  Informational (profilers see dense_tn_kernel<4> / <5>). */
int octic_dense_wgrad_tile(int M, int N, int K);
int octic_dense_wgrad_tn(const void* dY, const void* X, int M, int N, int K, int64_t ldy, int64_t ldx, float* dW,
                         void* workspace, void* stream);
/* Two weight gradients that share the token rows M and K as ONE launch (tile list = [problem 0 | problem 1], row slabs chosen
 * for the sum): the qkv and proj weight gradients of a standard block (deit/vit.py:33-45).  Results bit-identical to two
 * octic_dense_wgrad_tn calls whenever the joint launch picks the slab counts those would (each tile sums its slabs in slab
 * order); workspace: octic_dense_wgrad_pair_workspace_bytes.                                                            */
int64_t octic_dense_wgrad_pair_workspace_bytes(int M, int N0, int N1, int K);
int octic_dense_wgrad_tn_pair(const void* dY0, const void* X0, int N0, int64_t ldy0, int64_t ldx0, float* dW0,
                              const void* dY1, const void* X1, int N1, int64_t ldy1, int64_t ldx1, float* dW1, int M, int K,
                              void* workspace, void* stream);

/* ---- row kernels of the DINOv2 objective over the prototype axis (K % 8 == 0, row strides % 8 == 0) -----
 * octic_softmax_center: out[r, :] = softmax((t[r, :] - center) * inv_temp) in f32 - the teacher's centred, sharpened
 * probabilities (dinov2/loss/dino_clstoken_loss.py:43-51, ibot_patch_loss.py:63-77); center may be NULL; t f32 or bf16.
 * octic_soft_ce_fwd: loss[r] = -sum_k t_k log_softmax(s[r, :] * inv_temp)_k with t = tprob[r % t_rows, :] (several student
 * crops against the same teacher rows: dino_clstoken_loss.py:78-92; the masked patch tokens: ibot_patch_loss.py:26-34), plus
 * the row statistics the gradient needs (lse[r], tsum[r] = sum_k t_k).  octic_soft_ce_bwd: ds[r, k] = g[r] inv_temp
 * (softmax(s inv_temp)_k tsum[r] - t_k) in s's dtype.  s is read in its storage dtype (bf16 under autocast), arithmetic f32. */
int octic_softmax_center(const void* t, int t_dtype, int64_t ldt, const float* center, float inv_temp, float* out,
                         int64_t rows, int K, void* stream);
int octic_soft_ce_fwd(const void* s, int s_dtype, int64_t lds, const float* tprob, int64_t t_rows, float inv_temp,
                      float* loss, float* lse, float* tsum, int64_t rows, int K, void* stream);
int octic_soft_ce_bwd(const void* s, int s_dtype, int64_t lds, const float* tprob, int64_t t_rows, float inv_temp,
                      const float* g, const float* lse, const float* tsum, void* ds, int64_t ldd, int64_t rows, int K,
                      void* stream);

#ifdef __cplusplus
}
#endif
#endif /* OCTIC_HIP_H */
