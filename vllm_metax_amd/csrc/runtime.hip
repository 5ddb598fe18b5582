// runtime.hip — error plumbing and device-attribute queries of the C-ABI.
#include <stdarg.h>
#include <stdio.h>

#include "common.cuh"

namespace mi355x {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return MI355X_ELAUNCH;
  }
  return MI355X_OK;
}

}  // namespace mi355x

extern "C" {

int mi355x_abi_version(void) { return MI355X_ABI_VERSION; }

const char* mi355x_last_error(void) { return mi355x::g_err; }

// ref: csrc/cuda_utils_kernels.cu — cudaDeviceGetAttribute(&v, attr, dev)
int64_t mi355x_get_device_attribute(int64_t attribute, int64_t device_id) {
  int dev = static_cast<int>(device_id);
  if (device_id < 0) {
    if (hipGetDevice(&dev) != hipSuccess) {
      mi355x::set_error("hipGetDevice failed");
      return MI355X_ELAUNCH;
    }
  }
  int value = 0;
  hipError_t e =
      hipDeviceGetAttribute(&value, static_cast<hipDeviceAttribute_t>(attribute), dev);
  if (e != hipSuccess) {
    mi355x::set_error("hipDeviceGetAttribute(%ld, %d): %s", (long)attribute, dev,
                      hipGetErrorString(e));
    return MI355X_ELAUNCH;
  }
  return value;
}

// ref: csrc/cuda_utils_kernels.cu — cudaDevAttrMaxSharedMemoryPerBlockOptin.
// On gfx950 one workgroup may own the CU's whole 160 KiB LDS.
int64_t mi355x_get_max_shared_memory_per_block_device_attribute(int64_t device_id) {
  return mi355x_get_device_attribute(
      static_cast<int64_t>(hipDeviceAttributeMaxSharedMemoryPerBlock), device_id);
}

}  // extern "C"
