// Shared device/host helpers for the gfx950 kernels.  Wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/mhr.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define MHR_WAVE 64

// ---- host-side error plumbing ------------------------------------------------------------
void mhr_set_error(const char* fmt, ...);

#define MHR_REQUIRE(cond, ...)                 \
  do {                                         \
    if (!(cond)) {                             \
      mhr_set_error(__VA_ARGS__);              \
      return MHR_EINVAL;                       \
    }                                          \
  } while (0)

#define MHR_CHECK_LAUNCH(name)                                                     \
  do {                                                                             \
    hipError_t e_ = hipGetLastError();                                             \
    if (e_ != hipSuccess) {                                                        \
      mhr_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));        \
      return MHR_ELAUNCH;                                                          \
    }                                                                              \
  } while (0)

static inline int mhr_grid_for(int64_t work_items, int per_block, int max_blocks = 2048 * 4) {
  int64_t b = (work_items + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > max_blocks) b = max_blocks;
  return (int)b;
}

// device address of the counter of item ids outside [0, n_rows) that the gather kernels met (csrc/embedding.hip; read by
// mhr_bad_id_count): kernels of other translation units that index the item table count into it too
unsigned int* mhr_bad_id_counter_addr();

// Deterministic mode (mhr_set_deterministic): launchers that would split a reduction over workgroups adding with float atomics
// pick an order-independent form instead (one row range per column block; fixed-point accumulators; ordered partial sums).
int mhr_deterministic();

// ---- device helpers ----------------------------------------------------------------------
// Order-independent accumulation: fixed-point int64 atomics (integer addition is associative, so the sum has one value whatever
// order the workgroups arrive in).  2^-40 resolution, |sum| < 2^23 - gradient magnitudes; mhr_det_flush converts back.
#define MHR_DET_SCALE 1099511627776.0f
__device__ __forceinline__ void det_atomic_add(long long* acc, float v) {
  atomicAdd(reinterpret_cast<unsigned long long*>(acc), (unsigned long long)__float2ll_rn(v * MHR_DET_SCALE));
}

__device__ __forceinline__ float bf2f(bf16_t x) { return (float)x; }
__device__ __forceinline__ bf16_t f2bf(float x) { return (bf16_t)x; }   // v_cvt_pk_bf16_f32: RNE, NaN stays NaN

// sigmoid by one v_exp_f32 and one v_rcp_f32 (1 ulp; an IEEE division costs a Newton step and a fix-up on top)
__device__ __forceinline__ float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float silu_f(float x) { return x * sigmoid_f(x); }
// d/dx silu(x) = sig * (1 + x * (1 - sig))
__device__ __forceinline__ float dsilu_f(float x) {
  const float s = sigmoid_f(x);
  return s * (1.0f + x * (1.0f - s));
}
// silu and its derivative from ONE sigmoid
__device__ __forceinline__ void silu_both(float x, float& y, float& dy) {
  const float s = sigmoid_f(x);
  y = x * s;
  dy = s * (1.0f + x * (1.0f - s));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// 4 consecutive elements, f32 or bf16 storage
template <typename T> struct Vec4IO;
template <> struct Vec4IO<float> {
  static __device__ __forceinline__ f32x4 load(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
  static __device__ __forceinline__ void store(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
};
template <> struct Vec4IO<bf16_t> {
  static __device__ __forceinline__ f32x4 load(const bf16_t* p) {
    bf16x4 r = *reinterpret_cast<const bf16x4*>(p);
    f32x4 v = {(float)r[0], (float)r[1], (float)r[2], (float)r[3]};
    return v;
  }
  static __device__ __forceinline__ void store(bf16_t* p, f32x4 v) {
    bf16x4 r = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
    *reinterpret_cast<bf16x4*>(p) = r;
  }
};

// Dropout seed of a launch.  Host-issued steps pass the whole seed as a kernel argument; a step replayed from a hipGraph
// cannot (arguments are frozen at capture), so it passes the per-layer constant part and a pointer to the step counter in
// device memory.  Both give (step * 1000003 + layer_part) & (2^63 - 1) - the host formula of REC/model/IDNet/hstu.py:_encode.
__device__ __forceinline__ uint64_t mhr_step_seed(uint64_t seed, const int64_t* __restrict__ step_seed) {
  if (!step_seed) return seed;
  return ((uint64_t)step_seed[0] * 1000003ull + seed) & 0x7FFFFFFFFFFFFFFFull;
}

// counter-based uniform in [0,1) for the dropout masks: murmur3's 32-bit finaliser over (index, seed) folded to 32 bits - a
// dozen 32-bit operations per element (the 64-bit splitmix finaliser used before cost three 64-bit multiplies per element:
// about 7 us of a 16 us ln_gate launch at cfg1).  Forward and backward regenerate the same mask from (seed, index).
__device__ __forceinline__ float mhr_uniform(uint64_t seed, uint64_t idx) {
  uint32_t x = (uint32_t)idx ^ ((uint32_t)(idx >> 32) * 0x9E3779B9u) ^ (uint32_t)seed ^ ((uint32_t)(seed >> 32) * 0x7F4A7C15u);
  x ^= x >> 16;
  x *= 0x85EBCA6Bu;
  x ^= x >> 13;
  x *= 0xC2B2AE35u;
  x ^= x >> 16;
  x += (uint32_t)idx * 0x27D4EB2Fu;          // decorrelates the fixed points of the finaliser on consecutive counters
  x ^= x >> 15;
  return (float)(x >> 8) * (1.0f / 16777216.0f);
}

// Any-hit summary of the false-negative bit table (mhr_nce_fix_bits: fix_any[row]): bit k = some negative of the tiles
// [k << shift, (k + 1) << shift) is suppressed for the row.  shift makes the n_tiles 32-negative tiles fit 32 groups.
__host__ __device__ inline int fix_group_shift(int n_tiles) {
  int sh = 0;
  while (((n_tiles + (1 << sh) - 1) >> sh) > 32) ++sh;
  return sh;
}

