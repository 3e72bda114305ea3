#include "rsp_common.h"
extern "C" int rsp_abi_version(void) { return 1; }
extern "C" const char* rsp_build_info(void) {
  return "librsp_hip gfx950 (CDNA4) fp16x3-MFMA; built " __DATE__ " " __TIME__;
}
