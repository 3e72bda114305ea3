#include "rsp_common.h"
extern "C" int rsp_abi_version(void) { return 5; }   // 5: rsp_sam_fold_expand / rsp_sam_fold_gather (round 6); 4: rsp_sam_t2i_fold lost its `variant` argument, rsp_gemm_uses_pp (round 5); 2: RspGemmDesc gained c_ncols / pl_col0 (round 2), plane format words decode strictly; 3: RspBoxCoder in RspRpnDesc / rsp_bbox_post (round 4)
extern "C" const char* rsp_build_info(void) {
  return "librsp_hip gfx950 (CDNA4) fp16x3-MFMA; built " __DATE__ " " __TIME__;
}
