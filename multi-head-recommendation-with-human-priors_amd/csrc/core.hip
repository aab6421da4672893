// Error plumbing + ABI version.
#include "mhr_common.h"

static thread_local char g_err[512] = "";

void mhr_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* mhr_last_error(void) { return g_err; }
extern "C" int mhr_abi_version(void) { return 1; }

// ---- deterministic mode ---------------------------------------------------------------------
static int g_mhr_det = 0;
int mhr_deterministic() { return g_mhr_det; }
extern "C" int mhr_set_deterministic(int on) {
  g_mhr_det = on ? 1 : 0;
  return MHR_OK;
}
extern "C" int mhr_get_deterministic(void) { return g_mhr_det; }

namespace {
__global__ __launch_bounds__(256) void det_flush_kernel(long long* __restrict__ acc, float* __restrict__ dst, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const long long a = acc[i];
    if (a != 0) {
      dst[i] += (float)((double)a * (1.0 / (double)MHR_DET_SCALE));
      acc[i] = 0;
    }
  }
}

// dst[0] += scale * sum(parts[0 .. n)) with ONE fixed association order: lane l sums parts[l], parts[l + 64], ... in index order,
// then the 64 lane sums are folded by the butterfly of wave_sum
__global__ __launch_bounds__(64) void det_sum_kernel(const float* __restrict__ parts, int64_t n, const float* __restrict__ scale_dev,
                                                      int apply_exp_scale, float* __restrict__ dst) {
  float s = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += 64) s += parts[i];
  s = wave_sum(s);
  if (threadIdx.x == 0) {
    float sc = 1.0f;
    if (scale_dev) sc = apply_exp_scale ? __expf(fminf(fmaxf(scale_dev[0], 0.f), 4.605170185988092f)) : scale_dev[0];
    dst[0] += s * sc;
  }
}
}  // namespace

extern "C" int mhr_det_flush(int64_t* acc, float* dst, int64_t n, void* stream) {
  MHR_REQUIRE(acc && dst && n >= 0, "det_flush: null pointer");
  if (n == 0) return MHR_OK;
  hipLaunchKernelGGL(det_flush_kernel, dim3(mhr_grid_for(n, 256 * 4)), dim3(256), 0, (hipStream_t)stream, (long long*)acc, dst, n);
  MHR_CHECK_LAUNCH("det_flush");
  return MHR_OK;
}

extern "C" int mhr_det_sum_into(const float* parts, int64_t n, const float* scale_dev, int exp_clamped_scale, float* dst,
                                void* stream) {
  MHR_REQUIRE(parts && dst && n >= 0, "det_sum_into: null pointer");
  hipLaunchKernelGGL(det_sum_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, parts, n, scale_dev, exp_clamped_scale, dst);
  MHR_CHECK_LAUNCH("det_sum_into");
  return MHR_OK;
}
