/*
 * rsp_hip.h -- C ABI of librsp_hip.so: the MI355X (gfx950) kernels behind the
 * RSPrompter inference hot path (SAM ViT encoder -> anchor prompter -> SAM mask
 * decoder).
 *
 * Conventions
 *   - every entry point is `extern "C"`, takes raw DEVICE pointers, explicit
 *     sizes/strides and a HIP stream handle (`void*` == hipStream_t), and
 *     returns 0 on success or a negative RSP_E* code (Python maps !=0 to
 *     RuntimeError).  No torch types, no global state.
 *   - all activations are fp32, row-major, channels-last ("NHWC": a feature map
 *     is a [B*H*W, C] matrix).  GEMM weights are passed pre-split into two fp16
 *     planes (hi, lo) produced by rsp_split_f16 (see DESIGN.md "fp16x3").
 *   - "reference" citations are relative to /root/reference, `HF:` is
 *     transformers/models/sam/modeling_sam.py (the reference's pinned
 *     third-party SAM implementation).
 */
#ifndef RSP_HIP_H_
#define RSP_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RSP_OK 0
#define RSP_EINVAL (-1)   /* bad argument / unsupported shape */
#define RSP_ELAUNCH (-2)  /* hipLaunch / runtime error         */

#define RSP_ACT_NONE 0
#define RSP_ACT_RELU 1
#define RSP_ACT_GELU 2    /* exact erf GELU (torch nn.GELU default) */
#define RSP_ACT_SIGMOID 3
#define RSP_ACT_RELU_POST 4   /* C = relu(alpha*acc + bias + res): the ReLU after the shortcut of a ResNet block */

typedef void* rsp_stream_t;

/* Library / build identification (also used by the "symbols load" CPU test). */
int rsp_abi_version(void);
const char* rsp_build_info(void);

/* ------------------------------------------------------------------------ */
/* Plane format word                                                          */
/* ------------------------------------------------------------------------ */
/* Every `*_scale_log2` argument / descriptor field that describes a pair of fp16 planes is a FORMAT WORD:          */
/*   bits 0..7  (signed)  e: the planes hold x * 2^e                                                                  */
/*   bit  8     RSP_PLANE_F8: the second plane is not lo = f16(x*2^e - hi) but the "cat8" plane of the                */
/*              fp8-corrected product (same bytes: 64 per row and 32-wide K block):                                   */
/*                bytes  0..31  e4m3(lo[k] * 2^RSP_F8_LO_EXP)      bytes 32..63  e4m3(hi[k] * 2^-RSP_F8_HI_EXP)       */
/*              A GEMM whose A and B words both carry the bit computes                                                */
/*                a_hi b_hi  (fp16 MFMA)  +  a_lo8 b_hi8 + a_hi8 b_lo8  (ONE K=64 fp8 MFMA per 32 k, MX block scales  */
/*                2^-LO_EXP / 2^+HI_EXP undo the storage scales)                                                      */
/*              -- 2 units of matrix time instead of 3, error class 2^-15 instead of 2^-22.     */
/* Decoding is strict: a word is an F8 word only when its upper bits are exactly RSP_PLANE_F8; a plain (sign-extended)   */
/* negative exponent -- legal under the pre-format-word contract of these fields -- has all upper bits set and is NOT  */
/* mistaken for one; any other upper-bit pattern is invalid (entry points return RSP_EINVAL).                          */
#define RSP_PLANE_F8 0x100
#define RSP_PLANE_EXP(w) ((int)(int8_t)((w) & 0xff))
#define RSP_PLANE_IS_F8(w) ((((int32_t)(w)) & ~0xff) == RSP_PLANE_F8)
#define RSP_PLANE_WORD_VALID(w) \
  (RSP_PLANE_IS_F8(w) || (((int32_t)(w)) & ~0xff) == 0 || (((int32_t)(w)) & ~0xff) == ~0xff)
#define RSP_PLANE_WORD(e, f8) ((((int)(e)) & 0xff) | ((f8) ? RSP_PLANE_F8 : 0))
#define RSP_F8_LO_EXP 5
#define RSP_F8_HI_EXP 7

/* ------------------------------------------------------------------------ */
/* Weight preparation                                                        */
/* ------------------------------------------------------------------------ */
/* hi = f16(w * 2^e), lo = f16(w * 2^e - hi).  n elements.                   */
int rsp_split_f16(const float* w, uint16_t* hi, uint16_t* lo, int64_t n,
                  int scale_log2, rsp_stream_t stream);
/* same for a row-major [rows, K] matrix (K % 32 == 0), written in the KB32     */
/* plane layout [K/32][rows][32] that the DMA GEMM path consumes.               */
int rsp_split_f16_kb32(const float* w, uint16_t* hi, uint16_t* lo, int64_t rows, int32_t K,
                       int scale_log2, rsp_stream_t stream);

/* ------------------------------------------------------------------------ */
/* GEMM with fused prologue/epilogue: the workhorse.                          */
/*   C[crow(m), n] = act(alpha * sum_k A(m,k) * B[n,k] + bias[n]) + res       */
/* replaces: every nn.Linear / 1x1 conv / 3x3 conv / ConvTranspose2d(k2,s2)   */
/* on the path -- HF:116-128 (patch embed), HF:717,806-812 (qkv), HF:830      */
/* (proj), HF:132-143 (MLP), HF:975-992 (neck), models.py:1009-1057           */
/* (aggregator), models.py:1296-1363 (SimpleFPN), rpn_head.py:80-97,          */
/* convfc_bbox_head.py:163-217, models.py:1641-1651, HF:231-270, HF:447-450.  */
/* ------------------------------------------------------------------------ */
typedef struct RspGemmDesc {
  const float* A;          /* [*, lda] fp32 (plain mode) or NHWC image (conv) */
  const uint16_t* Bhi;     /* [N, K] fp16 hi plane of the weight              */
  const uint16_t* Blo;     /* [N, K] fp16 lo plane                            */
  float* C;                /* [*, ldc]                                        */
  const float* bias;       /* [N] or NULL                                     */
  const float* res;        /* residual [*, ldr] or NULL (added after act)     */
  const int32_t* a_rowmap; /* [M] source row of A per GEMM row, -1 => zeros   */
  const int32_t* c_rowmap; /* [M] destination row of C, -1 => drop            */
  int32_t M, N, K;         /* K % 32 == 0                                     */
  int32_t lda, ldc, ldr;
  int32_t res_mod;         /* >0: residual row = crow % res_mod (broadcast)   */
  int32_t act;             /* RSP_ACT_*                                       */
  float alpha;             /* 2^-(a_scale_log2 + weight scale_log2)           */
  int32_t a_scale_log2;    /* plane format word of A (and of W): exponent e of the   */
                           /* fp16 split, RSP_PLANE_F8 selects the fp8-corrected     */
                           /* product (A planes and W planes then both carry cat8)   */
  /* implicit-GEMM convolution over an NHWC input (conv_k == 0: plain GEMM)   */
  int32_t conv_k;          /* kernel size (3) ; K must equal conv_k^2 * conv_C */
  int32_t conv_stride, conv_pad;
  int32_t conv_H, conv_W, conv_C; /* input  spatial size / channels           */
  int32_t conv_Ho, conv_Wo;       /* output spatial size; M = B*Ho*Wo         */
  /* ConvTranspose2d(k=2,s=2) as two GEMMs (one per output row parity dy) with  */
  /* N = 2*Cout (dx, co): GEMM row (y*W + x) -> C row ((y*2 + dy)*W + x) of an   */
  /* output viewed as [B*2H*W, 2*Cout].  ct_W == 0 disables.                     */
  int32_t ct_W, ct_dy;
  /* residual batch gather: residual row = res_bmap[crow / res_brows] * res_brows */
  /* + crow % res_brows  (per-RoI rows adding their image's rows). NULL disables. */
  const int32_t* res_bmap;
  int32_t res_brows;
  /* fp16 "plane" operands/results: A given as two fp16 tensors (hi, lo) of      */
  /* x * 2^a_scale_log2 in the K-BLOCKED layout [K/32][a_rows][32] ("KB32": every 128x32 K tile   */
  /* of a plane is one contiguous 8 KiB run of full cache lines) -> the DMA fast path             */
  /* (global_load_lds, no register staging).  Weights (Bhi/Blo) use the same layout [K/32][N][32] */
  /* when A is given as planes.  Chi/Clo: additionally (or, with C == NULL, only) write the       */
  /* result pre-split, scale 2^c_scale_log2, KB32 layout [N/32][c_rows][32], for the next GEMM.   */
  const uint16_t* Ahi; const uint16_t* Alo;
  uint16_t* Chi; uint16_t* Clo;
  int32_t c_scale_log2;    /* plane format word of Chi / Clo (RSP_PLANE_F8: Clo is written as the cat8 plane) */
  int32_t a_rows, c_rows;
  int32_t b_rows;   /* rows of the Bhi/Blo plane tensors when the weight is a row slice of them (0 = N) */
  /* hyper-network epilogue (HF:523-531 fused into the last ConvTranspose of the SAM upscaler): instead of  */
  /* storing the [rows, N] tile, every 32-column group g of a row is reduced against hyper[row / hd_rows]:   */
  /* hd_out[(row / hd_rows) * hd_ostride + pix(row, g)] = sum_c act(...)[row, 32g + c] * hyper[., c]          */
  /* (pix = the sub-pixel the 32-column group writes); needs A planes, ct_W > 0 and N == 64 (ct_dy >= 0) or   */
  /* N == 128 (ct_dy < 0: columns are (dy, dx, co), all four sub-pixels from one pass over A).                */
  /* grouped-LayerNorm epilogue (HF:519-520 fused into the first ConvTranspose of the SAM upscaler): with  */
  /* ct_W > 0, ct_dy < 0 and N == 256 every 64-column group (one output sub-pixel) is normalised over its  */
  /* 64 channels (LayerNorm2d, eps ln_eps, affine ln_gamma/ln_beta [64]) BEFORE `act`; plane output only.  */
  const float* ln_gamma; const float* ln_beta; float ln_eps;
  /* residual given as fp16 planes (KB32 [N/32][res_rows][32], value * 2^res_scale_log2) instead of fp32 `res`:    */
  /* C = act(...) + (hi + lo) * 2^-e, rows mapped like `res` (res_mod / res_bmap).                                */
  const uint16_t* res_hi; const uint16_t* res_lo; int32_t res_scale_log2, res_rows;
  const float* hd_hyper; float* hd_out;
  int32_t hd_rows;    /* GEMM rows per RoI (input pixels of the ConvTranspose)                                */
  /* column-range outputs (plane path, plain GEMMs): fp32 C is written only for columns < c_ncols (0 = all, same ldc),  */
  /* the planes Chi/Clo only for columns >= pl_col0 (pl_col0 % 32 == 0) as a [c_rows, N - pl_col0] KB32 tensor.          */
  /* The SAM ViT qkv projection uses c_ncols = pl_col0 = D: q leaves as fp32 (rel-pos + attention Q operand), K | V      */
  /* as the fp16 planes the attention kernel DMAs (rsp_vit_attention_planes) -- no fp32 K / V tensor is ever written.    */
  int32_t c_ncols, pl_col0;
  int32_t tile_hint;  /* 0 = auto (cost model in gemm_dma.hip).  Benchmarking only: 1/2/3 = plain 128x128 / 256x128 / */
                      /* 256x256, 11-14 = register-pipelined loops, 17-20 = + DMA spread between the MFMA groups    */
                      /* (17 = 256x256, 18 = 256x128, 20 = 128x128); other values are tuning probes.                */
} RspGemmDesc;

int rsp_gemm(const RspGemmDesc* desc, rsp_stream_t stream);
/* 1 when rsp_gemm serves this descriptor with the two-blocks-per-CU persistent kernel (csrc/gemm_s2.hip), 0 when with   */
/* one of the csrc/gemm_dma.hip / gemm.hip tiles (profiler labels; no device work)                                       */
int rsp_gemm_uses_s2(const RspGemmDesc* desc);
/* -1 when rsp_gemm_uses_s2(desc) == 0, else the epilogue form that kernel runs for the descriptor: bit flags 1 = fp32     */
/* residual, 2 = GELU, 4 = fp32 output, 8 = plane output, 16 = output row map; 64 = the run-time form that serves every    */
/* other mode (tests assert which compile-time specialisation they exercise; no device work)                               */
int rsp_gemm_s2_epilogue(const RspGemmDesc* desc);
/* 256 / 128 when rsp_gemm serves this descriptor with the ping-pong kernel (csrc/gemm_pp.hip: one 512-thread block per   */
/* CU, block tile 256 x 256 or 128 x 256, the two waves of a SIMD alternating matrix and load phases) = the tile's rows;   */
/* 0 otherwise.  tile_hint 200 / 201 force the two tiles for every descriptor the kernel implements (those with one of the */
/* compile-time epilogue forms above and K >= 128).                                                                       */
int rsp_gemm_uses_pp(const RspGemmDesc* desc);

/* ------------------------------------------------------------------------ */
/* LayerNorm over the last dim of a [rows, C] matrix (C % 4 == 0, C <= 2048). */
/* replaces nn.LayerNorm(eps=1e-6) HF:894-896 and every channel LN on NHWC    */
/* data: SamLayerNorm(channels_first) HF:147-170, LN2d models.py:33-50.       */
/* act: RSP_ACT_NONE or RSP_ACT_GELU (fused LN->GELU of models.py:1299-1300,  */
/* HF:519-520).                                                               */
/* ------------------------------------------------------------------------ */
int rsp_layernorm(const float* x, const float* gamma, const float* beta, float* y,
                  int64_t rows, int32_t C, float eps, int32_t act, rsp_stream_t stream);
/* same, optionally writing the result as fp16 planes (hi, lo of y * 2^scale_log2; KB32 layout  */
/* [C/32][rows][32], C % 32 == 0) for a following plane-mode GEMM; y may then be NULL.          */
int rsp_layernorm_ex(const float* x, const float* gamma, const float* beta, float* y,
                     uint16_t* yhi, uint16_t* ylo, int32_t scale_log2, int64_t rows, int32_t C,
                     float eps, int32_t act, rsp_stream_t stream);

/* ------------------------------------------------------------------------ */
/* SAM ViT attention (windowed and global) with decomposed rel-pos bias.      */
/* replaces SamVisionAttention.forward HF:803-831 + get_decomposed_rel_pos    */
/* HF:761-801 (vit_sam.py:117-157, 202-221).                                  */
/* ------------------------------------------------------------------------ */
/* rel[bp*nh + h, t, 0:S]   = q(t) . Rh[qh(t) - kh + S-1]   (unscaled q)       */
/* rel[bp*nh + h, t, S:2S]  = q(t) . Rw[qw(t) - kw + S-1]                      */
/* qkv: [Bp, T, 3, nh, dh] fp32 (T = S*S), rel_pos_{h,w}: [2S-1, dh].          */
int rsp_vit_relpos(const float* qkv, const float* rel_pos_h, const float* rel_pos_w,
                   float* rel, int32_t Bp, int32_t S, int32_t nh, int32_t dh,
                   rsp_stream_t stream);

/* out[Bp, T, nh*dh] = softmax_fp32((q*scale) k^T + rel_h (+) rel_w) v         */
int rsp_vit_attention(const float* qkv, const float* rel, float* out,
                      int32_t Bp, int32_t S, int32_t nh, int32_t dh, float scale,
                      rsp_stream_t stream);
/* same with an optional fp16-plane copy of the output (KB32 layout [D/32][Bp*T][32]; feeds the  */
/* proj GEMM's DMA path)                                                                         */
int rsp_vit_attention_ex(const float* qkv, const float* rel, float* out, uint16_t* out_hi,
                         uint16_t* out_lo, int32_t out_scale_log2, int32_t Bp, int32_t S,
                         int32_t nh, int32_t dh, float scale, rsp_stream_t stream);
/* Global-attention layers (S == 64 or 32): K and V of every (image, head) are split once into fp16 hi/lo planes   */
/* rsp_vit_attention_planes: the same attention with K | V given as the KB32 fp16 planes [2D/32][kv_rows][32] of the  */
/* qkv GEMM (value * 2^kv_scale_log2) and q as fp32 rows of stride q_ld: DMA-fed key tiles, transposing LDS reads for  */
/* V (csrc/attn_stream.hip).  S = 14 (windows), 32, 64; dh = 64 / 80.  Outputs as rsp_vit_attention_ex.                */
int rsp_vit_attention_planes(const float* q, int64_t q_ld, const uint16_t* kv_hi, const uint16_t* kv_lo,
                             int64_t kv_rows, int32_t kv_scale_log2, const float* rel, float* out, uint16_t* out_hi,
                             uint16_t* out_lo, int32_t out_scale_log2, int32_t Bp, int32_t S, int32_t nh, int32_t dh,
                             float scale, rsp_stream_t stream);
/* ... with the window grid of window_partition (HF:900-922) known: the Bp windows are (image, wy, wx) over a            */
/* win_per_side x win_per_side grid whose last row / column holds only win_real_last real rows / columns (64-grid, 14:    */
/* 5 and 8).  Outputs of the padded tokens are NOT written (window_unpartition crops them); their keys / values take     */
/* part as usual.  win_per_side = 0: every query is computed (= rsp_vit_attention_planes).                                */
int rsp_vit_attention_planes_ex(const float* q, int64_t q_ld, const uint16_t* kv_hi, const uint16_t* kv_lo,
                                int64_t kv_rows, int32_t kv_scale_log2, const float* rel, float* out,
                                uint16_t* out_hi, uint16_t* out_lo, int32_t out_scale_log2, int32_t Bp, int32_t S,
                                int32_t nh, int32_t dh, float scale, int32_t win_per_side, int32_t win_real_last,
                                rsp_stream_t stream);
/* Windowed layers, rel-pos terms computed INSIDE the attention kernel (csrc/attn_win.hip; replaces the pair             */
/* rsp_vit_relpos_rows + rsp_vit_attention_planes_ex of a windowed SamVisionAttention.forward, HF:803-831 + 761-801):     */
/* rel_tab = the layer's two tables packed once by rsp_pack_relpos_tables; q / K | V planes / outputs / window grid as    */
/* above with S = 14.  The kernel is persistent (a block walks the windows of one head); variant: 0 = product (768        */
/* blocks), n > 0 = 16 n blocks (tests and measurements).                                                                 */
/* The K | V plane exponent (kv_scale_log2) must lie in [-8, 4] for S = 14 (both entry points): the padded-key mask and   */
/* the rel-pos bias share the score scale 2^(6 + exponent); outside that range the call returns RSP_EINVAL.                */
int rsp_vit_window_attention(const float* q, int64_t q_ld, const uint16_t* kv_hi, const uint16_t* kv_lo,
                             int64_t kv_rows, int32_t kv_scale_log2, const uint16_t* rel_tab, float* out,
                             uint16_t* out_hi, uint16_t* out_lo, int32_t out_scale_log2, int32_t Bp, int32_t nh,
                             int32_t dh, float scale, int32_t win_per_side, int32_t win_real_last, int32_t variant,
                             rsp_stream_t stream);
/* rel_pos_h / rel_pos_w [2S-1, dh] fp32 (2S-1 <= 32, dh % 8 == 0) -> out [2][2][32][dh + 8] fp16: per table the hi and   */
/* lo planes of table * 2^6, rows >= 2S-1 and the 8 pad columns zero (16-byte aligned, 2*2*32*(dh+8) halves)               */
int rsp_pack_relpos_tables(const float* rel_pos_h, const float* rel_pos_w, uint16_t* out, int32_t S, int32_t dh,
                           rsp_stream_t stream);
/* rsp_vit_relpos with an explicit token stride of q (q rows of [Bp*T, q_ld], head h at column h*dh)                   */
int rsp_vit_relpos_q(const float* q, int64_t q_ld, const float* rel_pos_h, const float* rel_pos_w, float* rel,
                     int32_t Bp, int32_t S, int32_t nh, int32_t dh, rsp_stream_t stream);
/* ... for a list of rows only (windowed layers): rows_map[n_rows] = the q / rel rows to compute, e.g. the real tokens  */
/* of padded windows -- the rel rows of padded queries are never read once the attention skips them                     */
/* (rsp_vit_attention_planes_ex).  rows_map = NULL: all Bp*S*S rows.                                                     */
int rsp_vit_relpos_rows(const float* q, int64_t q_ld, const float* rel_pos_h, const float* rel_pos_w, float* rel,
                        int32_t Bp, int32_t S, int32_t nh, int32_t dh, const int32_t* rows_map, int64_t n_rows,
                        rsp_stream_t stream);
/* (K as [key][dh], V transposed) inside `workspace` (rsp_vit_attention_global_ws_bytes) and the attention kernel   */
/* streams them HBM -> LDS with the DMA engine.  Same semantics and outputs as rsp_vit_attention_ex.                */
int64_t rsp_vit_attention_global_ws_bytes(int32_t Bp, int32_t S, int32_t nh, int32_t dh);
int rsp_vit_attention_global(const float* qkv, const float* rel, void* workspace, float* out, uint16_t* out_hi,
                             uint16_t* out_lo, int32_t out_scale_log2, int32_t Bp, int32_t S, int32_t nh, int32_t dh,
                             float scale, rsp_stream_t stream);

/* Generic multi-head attention out = softmax((q*scale) k^T) v with strided    */
/* operands (element strides; all multiples of 4), used by the SAM mask        */
/* decoder: SamAttention.forward HF:231-270 (self-attn dh=32, token<->image    */
/* cross-attn dh=16 with 4096 image keys).  kv_batch_map lets several query    */
/* batches (RoIs) share one K/V batch (their image).  dh in {16, 32, 64}.      */
typedef struct RspAttnDesc {
  const float* q; const float* k; const float* v; float* out;
  const int32_t* kv_batch_map;   /* [B] or NULL */
  const int32_t* q_batch_map;    /* [B] or NULL: batch b reads Q of batch q_batch_map[b] */
  int64_t q_bs, q_ts, q_hs;      /* batch / token / head strides (elements)   */
  int64_t k_bs, k_ts, k_hs;
  int64_t v_bs, v_ts, v_hs;
  int64_t o_bs, o_ts, o_hs;
  int32_t B, nh, dh, Tq, Tk;
  float scale;
  /* optional fp16-plane (KB32) copy of a dense [B*Tq, nh*dh] output; `out` may then be NULL */
  uint16_t* out_hi; uint16_t* out_lo; int32_t out_scale_log2;
  /* optional attention mask [B, Tq, Tk] bytes, non-zero = blocked, shared by all heads               */
  /* (Mask2Former masked cross-attention, mask2former_layers.py:113-121)                               */
  const uint8_t* mask;
} RspAttnDesc;
int rsp_attention(const RspAttnDesc* desc, rsp_stream_t stream);

/* SAM two-way transformer cross attentions, internal width 128 = 8 heads x 16 (HF:243-288 as used by HF:306-348 and */
/* HF:396-404), exact fp32:                                                                                          */
/*  token -> image: q [R,T,128] (T <= 12), kv [Rkv*N, 256] = image rows with K | V side by side, kv_map[r] = image    */
/*  row block of RoI r (NULL: r); out [R,T,128].                                                                      */
int rsp_sam_t2i_attention(const float* q, const float* kv, const int32_t* kv_map, float* out, int32_t R, int32_t T,
                          int32_t N, float scale, rsp_stream_t stream);
/*  image -> token: q [Rq*N,128] image-side queries (q_map[r] = row block, NULL: r), k, v [R,T,128] (T <= 16);        */
/*  result [R*N,128] as fp32 `out` and/or fp16 planes (KB32, value * 2^out_scale_log2).                               */
int rsp_sam_i2t_attention(const float* q, const int32_t* q_map, const float* k, const float* v, float* out,
                          uint16_t* out_hi, uint16_t* out_lo, int32_t out_scale_log2, int32_t R, int32_t T, int32_t N,
                          float scale, rsp_stream_t stream);
/*  image -> token attention FUSED with out_proj, the residual and layer_norm4 (HF:340-348):                          */
/*    y = LayerNorm(residual + out_proj(attention(q, k, v)))   rows [R*N, 256]                                        */
/*  out_proj is folded into the values (exact algebra, fp32), so the [R*N,128] attention output, the K = 128 GEMM over */
/*  it and the separate LayerNorm pass are never materialised.  wo [256,128], bo [256] = out_proj; the residual is      */
/*  either fp32 rows `res` [Rres*N, 256] with res_map[r] = row block of RoI r (NULL: r) -- layer 0, one copy per image   */
/*  -- or fp16 planes res_hi / res_lo of the [R*N, 256] tensor (layer 1); result as fp32 `out` and/or fp16 planes.      */
typedef struct RspI2tFusedDesc {
  const float* q; const int32_t* q_map; const float* k; const float* v;
  const float* wo; const float* bo;
  const float* res; const int32_t* res_map;
  const uint16_t* res_hi; const uint16_t* res_lo; int32_t res_scale_log2;
  const float* gamma; const float* beta; float eps;
  float* out; uint16_t* out_hi; uint16_t* out_lo; int32_t out_scale_log2;
  int32_t R, T, N; float scale;
} RspI2tFusedDesc;
int rsp_sam_i2t_fused(const RspI2tFusedDesc* desc, rsp_stream_t stream);

/* ------------------------------------------------------------------------ */
/* RoI feature extraction (single_level_roi_extractor.py:44-119 + mmcv RoIAlign, */
/* sampling_ratio=0, aligned=True) over <=4 NHWC levels; `pe` adds an          */
/* input-independent positional map per level (models.py:1566-1574) on the fly. */
/* out: [K, P, P, C].                                                          */
/* ------------------------------------------------------------------------ */
typedef struct RspRoiAlignDesc {
  const float* feat[4];      /* [B, H, W, C] per level                         */
  const float* pe[4];        /* [H, W, C] per level or NULL                    */
  int32_t H[4], W[4];
  float spatial_scale[4];    /* 1/stride                                       */
  const float* rois;         /* [K, 5] (batch index, x1, y1, x2, y2)           */
  float* out;
  int32_t K, P, C, num_levels, finest_scale;
} RspRoiAlignDesc;
int rsp_roi_align(const RspRoiAlignDesc* desc, rsp_stream_t stream);

/* ------------------------------------------------------------------------ */
/* Proposal / detection selection (rpn_head.py:134-304, bbox_head.py:476-571, */
/* bbox_nms.py:12-105, delta_xywh_bbox_coder.py:264-361, mmcv batched_nms).    */
/* Deterministic: ties are (score desc, position asc).                         */
/* ------------------------------------------------------------------------ */
/* DeltaXYWHBBoxCoder.decode (delta_xywh_bbox_coder.py:71-131 -> delta2bbox,  */
/* :264-361): denormalise with stds / means, clamp, optionally clip to the    */
/* image.  Every step rounds as the reference's eager fp32 expression does.   */
typedef struct RspBoxCoder {
  float means[4], stds[4];   /* target_means / target_stds                     */
  float max_ratio;           /* |log(wh_ratio_clip)|                           */
  float ctr_clamp;           /* pixels; read when add_ctr_clamp != 0           */
  int32_t clip_border;       /* clamp to [0, w] x [0, h] of img_hw             */
  int32_t add_ctr_clamp;     /* :340-342 (YOLOF form): clamp the centre shift, */
                             /* sizes bounded from above only                  */
} RspBoxCoder;
typedef struct RspRpnDesc {
  const float* head[5];      /* per level [B*H*W, ld]: cols [0,A) objectness,  */
                             /* cols [A, 5A) deltas (anchor-major)             */
  int32_t H[5], W[5];
  float stride[5];
  int32_t ld, A, nms_pre, num_levels;
  const float* base_anchors; /* device [L, A, 4] (anchor_generator.py:161-205) */
  RspBoxCoder coder;         /* rpn_head.bbox_coder                            */
  float min_bbox_size;       /* <0 disables the w/h filter                     */
} RspRpnDesc;
/* sel_idx/sel_score [B, L, nms_pre], sel_cnt [B, L] */
int rsp_rpn_topk(const RspRpnDesc* d, int32_t B, int32_t* sel_idx, float* sel_score,
                 int32_t* sel_cnt, rsp_stream_t stream);
/* decode + min-size filter -> per-image compacted candidates (level-major)    */
int rsp_rpn_decode(const RspRpnDesc* d, int32_t B, const int32_t* sel_idx, const float* sel_score,
                   const int32_t* sel_cnt, const float* img_hw, int32_t cap, float* cand_boxes,
                   float* cand_scores, int32_t* cand_ids, int32_t* cand_src, int32_t* cand_cnt,
                   rsp_stream_t stream);
/* softmax + per-class decode + score threshold -> candidates ((roi, class) order) */
int rsp_bbox_post(const float* head, int32_t ld, const float* rois, const int32_t* roi_start,
                  const float* img_hw, int32_t B, int32_t num_classes, float score_thr,
                  const RspBoxCoder* coder /*host*/, int32_t cap, float* cand_boxes,
                  float* cand_scores, int32_t* cand_ids, int32_t* cand_src, int32_t* cand_cnt,
                  rsp_stream_t stream);
int64_t rsp_nms_workspace_bytes(int32_t B, int32_t cap);
/* greedy NMS per image with per-id coordinate offsets; outputs [B, max_out(,4)] */
int rsp_batched_nms(const float* boxes, const float* scores, const int32_t* ids, const int32_t* src,
                    const int32_t* cnt, int32_t B, int32_t cap, float iou_thr, int32_t max_out,
                    void* workspace, int32_t* keep, int32_t* keep_cnt, float* out_boxes,
                    float* out_scores, int32_t* out_ids, int32_t* out_src, rsp_stream_t stream);

/* Glue of the folded token -> image attention below (round 6: torch index_put / gather before).  expand: tq [R*T, 128]    */
/* (q_proj output, head h at columns 16 h ..; HF:243-262 "_separate_heads") -> fp16 planes of the block-diagonal matrix      */
/* [R*96, 128]: row r*96 + h*T + t = scale * tq[r, t, head h] in columns 16 h .. 16 h + 15, zeros elsewhere; rows of the    */
/* columns >= 8 T zero.  gather: full [R*96, 128] -> ao [R*T, 128], ao[r*T + t, 16 h + d] = full[r*96 + h*T + t, 16 h + d]    */
/* (HF:264-268 "_recombine_heads" of the columns' own heads).  8 T <= 96.                                                   */
int rsp_sam_fold_expand(const float* tq, uint16_t* out_hi, uint16_t* out_lo, int32_t out_scale_log2, int32_t R, int32_t T,
                        float scale, rsp_stream_t stream);
int rsp_sam_fold_gather(const float* full, float* ao, int32_t R, int32_t T, rsp_stream_t stream);

/* Token -> image attention of the SAM two-way transformer with the K | V projections of the PER-RoI keys folded in   */
/* (HF:326-331, 397-400; csrc/t2i_fold.hip): keys = fp16 planes of [k_rows >= R*N, 256]; pek = planes of k_proj(pe) +    */
/* bias [N, 128]; qp = planes of q' [q_rows >= R*96, 256] with q'[r*96 + h*T + t] = Wk_h^T tq[r, t, h] (softmax scale     */
/* inside); tqx = planes of the block-diagonal tq [q_rows, 128]; *_e = plane scale exponents.  u [R*96, 256] fp32 receives */
/* sum_n softmax(score)[n] * keys[n] per column (rows of columns >= ncols = 8 T <= 96 stay untouched); the caller applies  */
/* v_proj to it.  N % 32 == 0.                                                                                           */
int rsp_sam_t2i_fold(const uint16_t* keys_hi, const uint16_t* keys_lo, int64_t k_rows, int32_t keys_e,
                     const uint16_t* pek_hi, const uint16_t* pek_lo, int32_t pek_e, const uint16_t* qp_hi,
                     const uint16_t* qp_lo, int32_t qp_e, const uint16_t* tqx_hi, const uint16_t* tqx_lo,
                     int32_t tqx_e, int64_t q_rows, float* u, int32_t R, int32_t N, int32_t ncols,
                     rsp_stream_t stream);

/* The upscaler tail of the SAM mask decoder in one pass over the per-RoI keys (HF:513-531; csrc/upscale.hip,          */
/* sam_upscale_fused_kernel): ConvTranspose2d(256 -> 64, k2 s2) + LayerNorm2d(64) + GELU + ConvTranspose2d(64 -> 32) + GELU */
/* + <., hyper_in>.  x: planes of [x_rows >= rows, 256] (rows = R*h*w, rows_per_roi = h*w, W = w); w1: planes of the packed  */
/* weight [(dy, dx, co), 256], bias1 [256] (tiled x4); w2: planes of [(dy2, dx2, c2), 64] with its K columns in the order    */
/* k' = 16 s + 8 hh + j <- channel 32 (s >> 1) + 8 ((8 (s & 1) + j) >> 2) + 4 hh + (j & 3), bias2 [128]; out [R, 4h, 4w].   */
int rsp_sam_upscale_fused(const uint16_t* x_hi, const uint16_t* x_lo, int64_t x_rows, int32_t x_scale_log2,
                          const uint16_t* w1_hi, const uint16_t* w1_lo, int32_t w1_scale_log2, const float* bias1,
                          const float* gamma, const float* beta, float eps, const uint16_t* w2_hi,
                          const uint16_t* w2_lo, int32_t w2_scale_log2, const float* bias2, const float* hyper,
                          float* out, int64_t rows, int32_t rows_per_roi, int32_t W, rsp_stream_t stream);

/* ------------------------------------------------------------------------ */
/* SAM decoder tail / mask post-process                                        */
/* ------------------------------------------------------------------------ */
/* out[r, pix] = sum_c up[r, pix, c] * hyper[r, c]     (HF:523-531)             */
/* Evaluation hand-off: COCO RLE counts of k bool masks [k,H,W] (row-major bytes), i.e. the run      */
/* lengths of the column-major pixel stream, first run = zeros (pycocotools.mask.encode as used by                   */
/* encode_mask_results, mmdet/structures/mask/utils.py:38-53).  counts [k, cap] uint32, n_counts[k] = number of runs */
/* or -(needed) when cap is too small; workspace: k*cap uint32.  W <= 8192.                                          */
int rsp_mask_rle(const uint8_t* masks, int32_t k, int32_t H, int32_t W, void* workspace, uint32_t* counts,
                 int32_t* n_counts, int32_t cap, rsp_stream_t stream);
/* ... and their compression to COCO's ASCII strings (cocoapi maskApi.c rleToString: what encode_mask_results returns   */
/* as `counts`, structures/mask/utils.py:38-53; consumed by CocoMetric.process, coco_metric.py:346-391): for the k      */
/* instances of `counts` [k, cap] / n_counts [k] (as written by rsp_mask_rle; n_counts <= 0 gives an empty string)      */
/* lens[m] = bytes of string m, offs [k + 1] = exclusive scan of lens, flat[offs[m] .. offs[m + 1]) = string m.          */
/* Strings that would end beyond flat_cap are not written: the caller compares offs[k] with flat_cap.                    */
int rsp_rle_to_string(const uint32_t* counts, const int32_t* n_counts, int32_t k, int32_t cap, int32_t* lens,
                      int64_t* offs, uint8_t* flat, int64_t flat_cap, rsp_stream_t stream);
/* SAM upscaler tail, streaming form (HF:521-531): rows = R*H2*W2 input pixels as fp16 planes (KB32, K = 64),      */
/* weight planes [2][128][32] with rows (dy, dx, c) of ConvTranspose2d(64 -> 32, k2, s2), bias tiled x4 [128],       */
/* hyper [R, 32]; out [R, 2*H2, 2*W2] = sum_c GELU(convT)[.., c] * hyper[r, c].  rows_per_roi = H2*W2, ct_W = W2.    */
int rsp_sam_upscale2(const uint16_t* a_hi, const uint16_t* a_lo, int64_t rows, int32_t a_scale_log2,
                     const uint16_t* w_hi, const uint16_t* w_lo, int32_t w_scale_log2, const float* bias,
                     const float* hyper, float* out, int32_t rows_per_roi, int32_t ct_W, rsp_stream_t stream);
int rsp_hyper_mask(const float* up, const float* hyper, float* out, int32_t R, int32_t npix,
                   int32_t C, rsp_stream_t stream);
/* models.py:1746-1784: sigmoid, bilinear (h,w)->(Hb,Wb), crop (crop_h,crop_w),  */
/* bilinear -> (out_h,out_w), >= thr.  out_prob optional.  sig_ws: [k*h*w] scratch  */
/* (sigmoid of the logits, computed once instead of per output tap).               */
int rsp_mask_post(const float* low_res, float* sig_ws, int32_t k, int32_t h, int32_t w, int32_t Hb, int32_t Wb,
                  int32_t crop_h, int32_t crop_w, int32_t out_h, int32_t out_w, float thr,
                  uint8_t* out_mask, float* out_prob, rsp_stream_t stream);
/* SAMDet.predict (models.py:1185-1206): the same resize -> crop -> resize chain on the raw low-res logits */
/* [k, h, w] of the SAM decoder, then `> thr` (strict; the reference uses 0).  out_val optional.           */
int rsp_mask_post_logits(const float* low_res, int32_t k, int32_t h, int32_t w, int32_t Hb, int32_t Wb,
                         int32_t crop_h, int32_t crop_w, int32_t out_h, int32_t out_w, float thr,
                         uint8_t* out_mask, float* out_val, rsp_stream_t stream);

/* ------------------------------------------------------------------------ */
/* SAMDet (models.py:1061-1215): FasterRCNN R50-FPN detector + SAM box prompts  */
/* ------------------------------------------------------------------------ */
/* ResNet stem (mmdet/models/backbones/resnet.py:640-647): conv1 7x7 s2 p3 (3 -> 64) + eval-mode bn1 folded   */
/* into w / bias + ReLU.  x [B,3,H,W] fp32 NCHW, w [147][64] (tap = (c*7 + ky)*7 + kx), y [B,Ho,Wo,64] NHWC.  */
int rsp_resnet_stem(const float* x, const float* w, const float* bias, float* y, int32_t B, int32_t H, int32_t W,
                    rsp_stream_t stream);
/* nn.MaxPool2d(k, stride s, padding p) on channels-last data (resnet.py:598: k3 s2 p1); C % 4 == 0          */
int rsp_maxpool_nhwc(const float* x, float* y, int32_t B, int32_t H, int32_t W, int32_t C, int32_t k, int32_t s,
                     int32_t p, rsp_stream_t stream);
/* FPN top-down step (mmdet/models/necks/fpn.py:190-204): dst [B,H,W,C] += nearest-upsampled src [B,h,w,C]   */
int rsp_upsample_nearest_add(const float* src, float* dst, int32_t B, int32_t h, int32_t w, int32_t H, int32_t W,
                             int32_t C, rsp_stream_t stream);
/* HF SamPromptEncoder._embed_boxes (transformers 4.38.1 modeling_sam.py:647-656): boxes [n,4] (x1,y1,x2,y2 in  */
/* input pixels) -> sparse prompt embeddings out [n, 2, 2*num_pos_feats]; gauss [2, num_pos_feats] is the shared */
/* positional_embedding, pe_top_left / pe_bottom_right are point_embed[2] / point_embed[3] [2*num_pos_feats].    */
int rsp_sam_embed_boxes(const float* boxes, const float* gauss, const float* pe_top_left,
                        const float* pe_bottom_right, float* out, int32_t n, int32_t num_pos_feats,
                        int32_t input_h, int32_t input_w, rsp_stream_t stream);

/* ------------------------------------------------------------------------ */
/* Query prompter (RSMask2FormerHead + MSDeformAttnPixelDecoder + fusion head)  */
/* ------------------------------------------------------------------------ */
/* GroupNorm(G) on a channels-last map x [B, HW, C] (+ReLU, + optional `add` after the norm):          */
/* mmcv ConvModule(norm_cfg=GN) of msdeformattn_pixel_decoder.py:73-111.  stats_ws: 2*B*G doubles.       */
int rsp_groupnorm_nhwc(const float* x, const float* gamma, const float* beta, const float* add, float* y,
                       double* stats_ws, int32_t B, int32_t HW, int32_t C, int32_t G, float eps,
                       int32_t relu, rsp_stream_t stream);
/* F.interpolate(bilinear, align_corners=False) on channels-last data                                    */
int rsp_resize_bilinear_nhwc(const float* x, float* y, int32_t B, int32_t H, int32_t W, int32_t Ho, int32_t Wo,
                             int32_t C, rsp_stream_t stream);
/* mmcv MultiScaleDeformableAttention core (8 heads x 16, 3 levels, 4 points): value [B,Ntok,128];         */
/* offs_weights [B,Ntok,ld]: 192 offsets (h,l,p,xy) then 96 attention logits (h,l*p); ref [Ntok,2];         */
/* level_hw: HOST int32 [L,2] (h, w), levels concatenated along Ntok.  out [B,Ntok,128] (pre output_proj).   */
int rsp_msdeform_attn(const float* value, const float* offs_weights, int32_t ld_ow, const float* ref_points,
                      float* out, int32_t B, int32_t Ntok, int32_t num_levels, const int32_t* level_hw,
                      rsp_stream_t stream);
/* the same with head_dim 16 | 32 (value / out rows of 8*head_dim): head_dim 32 is the embed_dims=256 pixel decoder of  */
/* the standard Mask2FormerHead (configs/rsprompter/_base_/samseg-mask2former.py:101-118)                             */
int rsp_msdeform_attn_ex(const float* value, const float* offs_weights, int32_t ld_ow, const float* ref_points,
                         float* out, int32_t B, int32_t Ntok, int32_t num_levels, const int32_t* level_hw,
                         int32_t head_dim, rsp_stream_t stream);
/* mask[row, k] = sigmoid(bilinear(mask_pred_plus[row], (h,w)))[k] < 0.5, fully-blocked rows cleared        */
/* (models.py:386-391, 439-442); rows = B*Nq maps of size Hs x Ws.                                          */
int rsp_query_attn_mask(const float* mask_pred_plus, uint8_t* mask, int64_t rows, int32_t Hs, int32_t Ws,
                        int32_t h, int32_t w, rsp_stream_t stream);
/* SamMaskEmbedding (HF:584-593) of mask_pred_plus [R, 4he, 4we] fused with `image_embeddings + dense`      */
/* (HF:499): out [R, he*we, C] = emb[roi_img[r]] + conv3(gelu(ln(conv2(gelu(ln(conv1(m)))))))               */
typedef struct RspMaskEmbedDesc {
  const float* mask_pred_plus; const float* image_embeddings; const int32_t* roi_img;
  const float *conv1_w, *conv1_b, *ln1_w, *ln1_b, *conv2_w, *conv2_b, *ln2_w, *ln2_b, *conv3_w, *conv3_b;
  float* out;
  int32_t R, he, we, C;
  float eps;
} RspMaskEmbedDesc;
int rsp_sam_mask_embed(const RspMaskEmbedDesc* d, rsp_stream_t stream);   /* C % 256 == 0, else RSP_EINVAL */
/* maskformer_fusion_head.py:149-162: softmax(cls)[:, :-1], top-k over Nq*nc, (score desc, index asc)        */
int rsp_query_topk(const float* cls, int32_t B, int32_t Nq, int32_t nc, int32_t k, float* out_score,
                   int32_t* out_flat, rsp_stream_t stream);
/* models.py:652-656,684-695 + maskformer_fusion_head.py:164-176 + mask2bbox: per selected instance,         */
/* bilinear logits -> (Hb,Wb) -> crop -> (out_h,out_w); mask = logit > 0; det_score = cls_score * mean        */
/* sigmoid over the mask; bbox = mask extents.  stats_ws: k * 32 bytes.                                       */
int rsp_query_mask_post(const float* low_res, const int32_t* qidx, const float* cls_score, int32_t k, int32_t h,
                        int32_t w, int32_t Hb, int32_t Wb, int32_t crop_h, int32_t crop_w, int32_t out_h,
                        int32_t out_w, void* stats_ws, uint8_t* out_mask, float* out_logits, float* det_score,
                        float* bboxes, rsp_stream_t stream);

/* ------------------------------------------------------------------------ */
/* Data movement                                                              */
/* ------------------------------------------------------------------------ */
/* DetDataPreprocessor.forward (data_preprocessor.py:110-149 + mmengine       */
/* ImgDataPreprocessor): BGR->RGB, (x-mean)/std, pad bottom/right.  src is    */
/* one CHW image (uint8 or fp32), dst one [3, Hp, Wp] slice of the batch.     */
/* mean3/std3 are HOST pointers.                                              */
int rsp_preprocess(const void* src, int32_t src_is_u8, float* dst, int32_t H, int32_t W,
                   int32_t Hp, int32_t Wp, const float* mean3, const float* std3,
                   int32_t swap_rb, float pad_value, rsp_stream_t stream);

/* FCNMaskHead._predict_by_feat_single + _do_paste_mask (fcn_mask_head.py:276-480; the mask post-process of the       */
/* SAMSegMaskRCNN sibling model, models.py:1219-1244): logits [k, Hm, Wm, C] NHWC, labels [k] (NULL / C == 1: class   */
/* agnostic), boxes [k, 4] in output-image coordinates -> out bool [k, img_h, img_w] (sigmoid, bilinear paste, >= thr). */
/* thr < 0 (:390-394): out holds the pasted probabilities as uint8, (p * 255) truncated.                               */
int rsp_paste_masks(const float* logits, const int32_t* labels, const float* boxes, int32_t k, int32_t Hm, int32_t Wm,
                    int32_t C, int32_t img_h, int32_t img_w, float thr, uint8_t* out, rsp_stream_t stream);

/* Test-pipeline front end: `Resize(scale, keep_ratio=True)` + `Pad(size, pad_val)`                       */
/* (configs/rsprompter/_base_/rsprompter_anchor.py:231-241; mmdet/datasets/transforms/transforms.py:134-247 */
/* and :704-786 over mmcv.imrescale / cv2.resize INTER_LINEAR and mmcv.impad), optionally fused with the     */
/* DetDataPreprocessor arithmetic (data_preprocessor.py:110-149).  src: one decoded HWC image (uint8 or      */
/* fp32, 3 interleaved channels); dst: [3, Hp, Wp] fp32; (Hn, Wn) = resized size inside the padded canvas.   */
/* pad3 / mean3 / std3 are HOST pointers.                                                                    */
int rsp_resize_pad(const void* src, int32_t src_is_u8, float* dst, int32_t H, int32_t W, int32_t Hn, int32_t Wn,
                   int32_t Hp, int32_t Wp, const float* pad3, int32_t normalise, int32_t swap_rb,
                   const float* mean3, const float* std3, rsp_stream_t stream);

/* im2col of the 16x16/s16 patch-embedding conv (HF:116-128): NCHW image ->   */
/* [B*gh*gw, C*p*p] rows, k = (c, ky, kx) (the conv weight's own flattening). */
int rsp_patchify(const float* img, float* out, int32_t B, int32_t C, int32_t H, int32_t W,
                 int32_t patch, rsp_stream_t stream);
/* NHWC pooling: mode 0 = MaxPool2d(2,2) (models.py:1307), mode 1 = max_pool2d(k=1,s=2) (:1362) */
int rsp_pool2(const float* x, float* y, int32_t B, int32_t H, int32_t W, int32_t C, int32_t mode,
              rsp_stream_t stream);
/* y[r,:] = x[r,:] + v[r % vmod, :] */
int rsp_add_rows(const float* x, const float* v, float* y, int64_t rows, int32_t C, int32_t vmod,
                 rsp_stream_t stream);
/* y[i] = sin(x[2i]) + x[2i+1]  (models.py:1672) */
int rsp_sincos_pairs(const float* x, float* y, int64_t n_out, rsp_stream_t stream);
/* out[i, j] = boxes[i, j] / sf4[j]   (bboxes /= scale_factor, models.py:1763-1764); sf4 is a HOST pointer */
int rsp_div_boxes(const float* boxes, float* out, int64_t n, const float* sf4, rsp_stream_t stream);
/* Rows of a GEMM output that the GEMM does not write because their A row is zero (window padding of window_partition,  */
/* HF:913-915: qkv(0) = bias): C[rows[i], 0:c_ncols) = bias and the planes of columns [pl_col0, N) (KB32, c_rows rows,    */
/* format word c_scale_log2 as in RspGemmDesc) = split(bias * 2^e) -- what the GEMM epilogue writes for alpha*0 + bias.    */
int rsp_fill_bias_rows(const float* bias, const int32_t* rows, int32_t n_rows, int32_t N, float* C, int32_t ldc,
                       int32_t c_ncols, uint16_t* Chi, uint16_t* Clo, int64_t c_rows, int32_t pl_col0,
                       int32_t c_scale_log2, rsp_stream_t stream);
/* scale_boxes (structures/bbox/transforms.py:391-414): out = boxes * (f0, f1, f2, f3); the R-CNN head's rescale   */
/* multiplies by fp32(1 / scale_factor) (bbox_head.py:549-552).  f4: HOST float[4].  In place allowed.               */
int rsp_scale_boxes(const float* boxes, float* out, int64_t n, const float* f4 /*host*/, rsp_stream_t stream);
/* bool bytes -> bits (little-endian in a byte) : the payload of the multi-GPU result all-gather,  */
/* replacing CocoMetric's per-rank RLE + mmengine collect_results (coco_metric.py:356-391).       */
int rsp_pack_bits(const uint8_t* src, uint8_t* dst, int64_t n_bits, rsp_stream_t stream);
/* dst[i,:] = src[idx[i],:] (idx<0 -> 0) */
int rsp_gather_rows(const float* src, const int32_t* idx, float* dst, int64_t rows, int32_t C,
                    rsp_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* RSP_HIP_H_ */
