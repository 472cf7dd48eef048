#include "common.h"

static thread_local char g_err[512] = "";

void sp_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char *sp_last_error(void) { return g_err; }
extern "C" int sp_version(void) { return 100; }
extern "C" size_t sp_gemm_desc_size(void) { return sizeof(sp_gemm_desc); }

// Raises a kernel's dynamic-LDS limit once per (kernel, device): the attribute is per device, so a process that
// launches on a second GPU needs it set there too.  `done` is the caller's per-kernel table (one static per template
// instance); the return code is reported through sp_set_error.
int sp_ensure_dyn_lds(const void *kernel, int bytes, bool (&done)[SP_MAX_DEVICES], const char *name) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= SP_MAX_DEVICES) {
    sp_set_error("%s: cannot identify the current HIP device", name);
    return SP_ELAUNCH;
  }
  if (done[dev]) return SP_OK;
  const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) {
    sp_set_error("%s: hipFuncSetAttribute(MaxDynamicSharedMemorySize=%d) failed on device %d: %s", name, bytes, dev,
                 hipGetErrorString(e));
    return SP_ELAUNCH;
  }
  done[dev] = true;
  return SP_OK;
}
