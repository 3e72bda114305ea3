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

typedef void* rsp_stream_t;

/* Library / build identification (also used by the "symbols load" CPU test). */
int rsp_abi_version(void);
const char* rsp_build_info(void);

/* ------------------------------------------------------------------------ */
/* Weight preparation                                                        */
/* ------------------------------------------------------------------------ */
/* hi = f16(w * 2^e), lo = f16(w * 2^e - hi).  n elements.                   */
int rsp_split_f16(const float* w, uint16_t* hi, uint16_t* lo, int64_t n,
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
  int32_t a_scale_log2;    /* A is multiplied by 2^e before the fp16 split    */
  /* implicit-GEMM convolution over an NHWC input (conv_k == 0: plain GEMM)   */
  int32_t conv_k;          /* kernel size (3) ; K must equal conv_k^2 * conv_C */
  int32_t conv_stride, conv_pad;
  int32_t conv_H, conv_W, conv_C; /* input  spatial size / channels           */
  int32_t conv_Ho, conv_Wo;       /* output spatial size; M = B*Ho*Wo         */
} RspGemmDesc;

int rsp_gemm(const RspGemmDesc* desc, rsp_stream_t stream);

/* ------------------------------------------------------------------------ */
/* LayerNorm over the last dim of a [rows, C] matrix (C % 4 == 0, C <= 2048). */
/* replaces nn.LayerNorm(eps=1e-6) HF:894-896 and every channel LN on NHWC    */
/* data: SamLayerNorm(channels_first) HF:147-170, LN2d models.py:33-50.       */
/* act: RSP_ACT_NONE or RSP_ACT_GELU (fused LN->GELU of models.py:1299-1300,  */
/* HF:519-520).                                                               */
/* ------------------------------------------------------------------------ */
int rsp_layernorm(const float* x, const float* gamma, const float* beta, float* y,
                  int64_t rows, int32_t C, float eps, int32_t act, rsp_stream_t stream);

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

/* im2col of the 16x16/s16 patch-embedding conv (HF:116-128): NCHW image ->   */
/* [B*gh*gw, C*p*p] rows, k = (c, ky, kx) (the conv weight's own flattening). */
int rsp_patchify(const float* img, float* out, int32_t B, int32_t C, int32_t H, int32_t W,
                 int32_t patch, rsp_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* RSP_HIP_H_ */
