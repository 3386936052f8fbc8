// Version / error strings of the C ABI.
#include "common.h"

extern "C" int cvllm_version(void) { return 110; }  // 0.1.1 (round 3: cvllm_select_status, pool_tile)

extern "C" const char* cvllm_error_string(int status) {
  switch (status) {
    case CVLLM_OK: return "ok";
    case CVLLM_ERR_ARG: return "invalid argument (null pointer or non-positive size)";
    case CVLLM_ERR_SHAPE: return "unsupported shape or dtype (head dim, GQA group, page size, dtype code)";
    case CVLLM_ERR_WORKSPACE: return "workspace missing or too small";
    case CVLLM_ERR_LAUNCH: return "HIP kernel launch failed";
    default: return "unknown status";
  }
}
