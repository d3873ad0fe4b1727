// Error plumbing and ABI version of liblfdm_hip.so (include/lfdm_hip.h).
#include <string.h>

#include "lfdm_device.h"
#include "../../include/lfdm_hip.h"

static thread_local char g_err[512] = "";

extern "C" void lfdm_set_error(const char* msg) {
  strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
  g_err[sizeof(g_err) - 1] = 0;
}

extern "C" const char* lfdm_last_error(void) { return g_err; }

extern "C" int lfdm_abi_version(void) { return 12; }   // 12: defer_reduce / gn_in_* reserved, lfdm_groupnorm_splitk_* removed, slab base rounded up to 128 bytes; 11: BatchNorm segments; 10: LFAE stage-1 training glue (train_lfae.hip); 9: lfdm_wgrad_params.dw_layout / .dbias, lfdm_multi_linear_*; 8: lfdm_conv_params.gn_in_*; 7: lfdm_calib_mfma_f32; 2: lfdm_conv_params.deconv4 / .groups; 3: .pool2; 4: *_lowres_cl_f32, schedule 3; 5: .weight_wino4, schedule 4; 6: .defer_reduce, lfdm_groupnorm_splitk_*

int lfdm_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    char buf[400];
    snprintf(buf, sizeof(buf), "%s: launch failed: %s", what, hipGetErrorString(e));
    lfdm_set_error(buf);
    return LFDM_ELAUNCH;
  }
  return LFDM_OK;
}
