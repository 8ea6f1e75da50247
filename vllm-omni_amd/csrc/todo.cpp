// Entry points declared in include/vllm_omni_amd.h whose kernels are not built yet.
// They fail loudly (MI_ERR_UNSUPPORTED); nothing falls back to a CPU path.
#include "../../include/vllm_omni_amd.h"
extern "C" {
int mi_dynamic_int8_quant(int8_t*, float*, const void*, int32_t, int32_t, int64_t, int, void*) {
  return MI_ERR_UNSUPPORTED; }
int mi_int8_gemm_dequant(void*, const int8_t*, const float*, const int8_t*, const float*,
                         const void*, int32_t, int32_t, int32_t, int64_t, int, void*) {
  return MI_ERR_UNSUPPORTED; }
}
