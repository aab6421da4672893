// Bare v_mfma_f32_32x32x16_bf16 rate: s_memtime ticks per MFMA and wall-clock TFLOP/s, one workgroup vs a full chip.
// hipcc --offload-arch=gfx950 -O3 -o tools/mfma_rate tools/mfma_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int WAVES_PER_SIMD>
__global__ __launch_bounds__(256 * WAVES_PER_SIMD) void k(float* out, unsigned long long* ticks, int iters) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) {
    if (iters & 1) {   // odd iteration count: pseudo-random N(0,1)-ish operands (high toggle rate)
      unsigned h = (threadIdx.x * 2654435761u) ^ (i * 40503u) ^ (blockIdx.x * 97u);
      h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
      a[i] = (__bf16)(((int)(h & 0xffff) - 32768) / 16384.0f);
      b[i] = (__bf16)(((int)(h >> 16) - 32768) / 16384.0f);
    } else { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x ^ i)); }
  }
  f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float s = 0;
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int W>
void run(int blocks, int iters) {
  float* out; unsigned long long* ticks;
  hipMalloc(&out, sizeof(float) * blocks * 256 * W);
  hipMalloc(&ticks, 8 * blocks);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<W><<<blocks, 256 * W>>>(out, ticks, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<W><<<blocks, 256 * W>>>(out, ticks, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks);
  hipMemcpy(h.data(), ticks, 8 * blocks, hipMemcpyDeviceToHost);
  double n_mfma = 16.0 * iters;                      // per wave
  double flops = n_mfma * 32768.0 * blocks * 4 * W;
  printf("waves/SIMD %d blocks %5d: %.3f ms  %.0f TFLOP/s  ticks/MFMA(wave 0 of block 0) %.1f  ns/MFMA/wave %.2f\n", W, blocks, ms,
         flops / ms / 1e9, h[0] / n_mfma, ms * 1e6 / n_mfma);
  hipFree(out); hipFree(ticks);
}

int main() {
  for (int it : {20000, 20001}) {
    printf("-- %s operands\n", (it & 1) ? "random" : "smooth");
    run<1>(1, it);
    run<1>(256, it);
    run<2>(256, it);
    run<2>(1024, it);
  }
  return 0;
}
