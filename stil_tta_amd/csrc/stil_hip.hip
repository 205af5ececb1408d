// Single translation unit of libstil_hip.so (gfx950).  C ABI declared in include/stil_hip.h.
#include "common.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";
void stil_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* stil_last_error(void) { return g_err; }
extern "C" int stil_version(void) { return 104; }   // round 4: stil_gemm_nt gained bstats / scale_var / split_ws, stil_weight_layouts, stil_bn_train_bwd_tiles
// number of HIP devices visible (0 = none): lets the host fail loudly before any launch
extern "C" int stil_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

#include "gemm.hip"
#include "bn.hip"
#include "transformer.hip"
#include "loss.hip"
#include "saint.hip"
#include "optim.hip"
#include "augment.hip"
#include "layout.hip"
