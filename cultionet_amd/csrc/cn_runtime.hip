// HIP runtime plumbing of the C ABI that torch does not expose: streams with an explicit priority or a CU mask.
// The weight-gradient side stream of the training step (cultionet_amd/engine.py) must not starve the data-gradient
// chain on the compute stream: it is created with the LOWEST priority the device offers (torch.cuda.Stream can only
// ask for "normal" or higher), or restricted to a subset of the CUs.
#include "cn_common.h"

// out[0] = least (numerically largest) priority, out[1] = greatest.
extern "C" int cn_stream_priority_range(int* out) {
  if (out == nullptr) return CN_ERR_ARG;
  if (hipDeviceGetStreamPriorityRange(&out[0], &out[1]) != hipSuccess) return CN_ERR_LAUNCH;
  return CN_OK;
}

// Non-blocking stream on the current device. cu_mask == nullptr: hipStreamCreateWithPriority(priority) (clamped by the
// runtime to the device's range). cu_mask != nullptr: hipExtStreamCreateWithCUMask over cu_mask_words 32-bit words
// (bit i = CU i may run this stream's kernels; the priority argument is ignored by the runtime API).
extern "C" int cn_stream_create(int priority, const unsigned* cu_mask, int cu_mask_words, void** stream_out) {
  if (stream_out == nullptr) return CN_ERR_ARG;
  hipStream_t s = nullptr;
  hipError_t e;
  if (cu_mask != nullptr) {
    if (cu_mask_words <= 0) return CN_ERR_ARG;
    e = hipExtStreamCreateWithCUMask(&s, (uint32_t)cu_mask_words, cu_mask);
  } else {
    e = hipStreamCreateWithPriority(&s, hipStreamNonBlocking, priority);
  }
  if (e != hipSuccess) return CN_ERR_LAUNCH;
  *stream_out = (void*)s;
  return CN_OK;
}

extern "C" int cn_stream_destroy(void* stream) {
  if (stream == nullptr) return CN_ERR_ARG;
  return hipStreamDestroy((hipStream_t)stream) == hipSuccess ? CN_OK : CN_ERR_LAUNCH;
}
