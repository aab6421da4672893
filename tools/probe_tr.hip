// Probe of ds_read_b64_tr_b16: LDS element e holds the value e; lane l supplies byte address l*8
// (elements 4l..4l+3).  Prints, per lane, the 4 received element ids.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void k(short* out) {
  __shared__ __attribute__((aligned(16))) short sm[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) sm[i] = (short)i;
  __syncthreads();
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sm + threadIdx.x * 4));
  for (int e = 0; e < 4; ++e) out[threadIdx.x * 4 + e] = v[e];
}
int main() {
  short* d; hipMalloc(&d, 64 * 4 * 2);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d:", l);
    for (int e = 0; e < 4; ++e) printf(" (src lane %2d, elem %d)", h[l * 4 + e] / 4, h[l * 4 + e] % 4);
    printf("\n");
  }
  return 0;
}
