// Dependent-chain cost of v_mfma_f32_32x32x16_bf16 on gfx950: one accumulator vs several, VGPR vs AGPR accumulators.
// hipcc --offload-arch=gfx950 -O3 -o tools/mfma_chain tools/mfma_chain.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define MFMA_V(c) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b))
#define MFMA_A(c) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b))

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* ticks, int iters) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x ^ i)); }
  f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int i = 0; i < iters; ++i) {
    if constexpr (MODE == 0) { MFMA_V(c0); MFMA_V(c0); MFMA_V(c0); MFMA_V(c0); }          // 1 chain, VGPR
    if constexpr (MODE == 1) { MFMA_V(c0); MFMA_V(c1); MFMA_V(c0); MFMA_V(c1); }          // 2 chains, VGPR
    if constexpr (MODE == 2) { MFMA_V(c0); MFMA_V(c1); MFMA_V(c2); MFMA_V(c3); }          // 4 chains, VGPR
    if constexpr (MODE == 3) { MFMA_A(c0); MFMA_A(c0); MFMA_A(c0); MFMA_A(c0); }          // 1 chain, AGPR
    if constexpr (MODE == 4) { MFMA_A(c0); MFMA_A(c1); MFMA_A(c0); MFMA_A(c1); }          // 2 chains, AGPR
    if constexpr (MODE == 5) { MFMA_A(c0); MFMA_A(c0); MFMA_A(c1); MFMA_A(c1); }          // pairs, AGPR
    if constexpr (MODE == 6) { MFMA_A(c0); MFMA_A(c1); MFMA_A(c2); MFMA_A(c3); }          // 4 chains, AGPR
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float s = 0;
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name) {
  const int blocks = 256, iters = 5000;
  float* out; unsigned long long* ticks;
  (void)hipMalloc(&out, sizeof(float) * blocks * 256);
  (void)hipMalloc(&ticks, 8 * blocks);
  k<MODE><<<blocks, 256>>>(out, ticks, iters);
  (void)hipDeviceSynchronize();
  unsigned long long h;
  (void)hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost);
  printf("%-22s %.1f ticks/MFMA\n", name, h / (4.0 * iters));
  (void)hipFree(out); (void)hipFree(ticks);
}

int main() {
  run<0>("1 chain  VGPR acc");
  run<1>("2 chains VGPR acc");
  run<2>("4 chains VGPR acc");
  run<3>("1 chain  AGPR acc");
  run<4>("2 chains AGPR acc");
  run<5>("pairs    AGPR acc");
  run<6>("4 chains AGPR acc");
  return 0;
}
