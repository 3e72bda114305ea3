#include "rsp_common.h"
extern "C" int rsp_abi_version(void) { return 2; }   // 2: RspGemmDesc gained c_ncols / pl_col0 (round 2), plane format words decode strictly
extern "C" const char* rsp_build_info(void) {
  return "librsp_hip gfx950 (CDNA4) fp16x3-MFMA; built " __DATE__ " " __TIME__;
}
