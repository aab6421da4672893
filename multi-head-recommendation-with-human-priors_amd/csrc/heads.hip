// Decoding heads, everything after the one concatenated GEMM (reference model/llm_heads.py:5-40 `x + SiLU(Linear(x))` per
// head, stacked by hstu.py:665-667 and permuted to [B, H, L, D] for the loss; one ResBlock per head):
//   out[b, h, l, :] = x[b, l, :] + silu(z[b, l, h, :])        z = x W^T + bias, [B L, H D] bf16 from the GEMM
// written straight in the [B, H, L, D] layout the loss reads (head rows (b H + h) L + l), fp32.  Backward: dz = d_out *
// silu'(z) in bf16 (the operand of the weight / input gradient GEMMs), dx = sum_h d_out (the residual branch) in fp32.
// Replaces F.silu + broadcast add + permute().contiguous() (3 passes over the 105 MB head tensor) and silu_backward + two
// layout copies + a sum over heads in the backward.  One wave per token, 16 bytes per lane.
#include "mhr_common.h"

namespace {

__global__ __launch_bounds__(256) void heads_residual_fwd_kernel(const float* __restrict__ x, const bf16_t* __restrict__ z,
                                                                 float* __restrict__ out, int64_t n_tok, int L, int H, int D) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), n_waves = (int64_t)gridDim.x * (blockDim.x >> 6);
  for (int64_t t = wave; t < n_tok; t += n_waves) {
    const int64_t b = t / L, l = t - b * L;
    for (int c = lane * 4; c < D; c += 256) {
      const f32x4 xv = *reinterpret_cast<const f32x4*>(x + t * D + c);
      for (int h = 0; h < H; ++h) {
        const f32x4 zv = Vec4IO<bf16_t>::load(z + (t * H + h) * D + c);
        f32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = xv[i] + silu_f(zv[i]);
        *reinterpret_cast<f32x4*>(out + (((b * H + h) * L + l) * D + c)) = o;
      }
    }
  }
}

__global__ __launch_bounds__(256) void heads_residual_bwd_kernel(const float* __restrict__ d_out, const bf16_t* __restrict__ z,
                                                                 bf16_t* __restrict__ dz, float* __restrict__ dx, int64_t n_tok,
                                                                 int L, int H, int D) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), n_waves = (int64_t)gridDim.x * (blockDim.x >> 6);
  for (int64_t t = wave; t < n_tok; t += n_waves) {
    const int64_t b = t / L, l = t - b * L;
    for (int c = lane * 4; c < D; c += 256) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      for (int h = 0; h < H; ++h) {
        const f32x4 g = *reinterpret_cast<const f32x4*>(d_out + (((b * H + h) * L + l) * D + c));
        const f32x4 zv = Vec4IO<bf16_t>::load(z + (t * H + h) * D + c);
        f32x4 dzv;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          dzv[i] = g[i] * dsilu_f(zv[i]);
          acc[i] += g[i];
        }
        Vec4IO<bf16_t>::store(dz + (t * H + h) * D + c, dzv);
      }
      *reinterpret_cast<f32x4*>(dx + t * D + c) = acc;
    }
  }
}

}  // namespace

extern "C" int mhr_heads_residual_fwd(const float* x, const void* z_bf16, float* out, int64_t n_tok, int seq_len, int n_heads,
                                      int dim, void* stream) {
  MHR_REQUIRE(x && z_bf16 && out, "heads_residual_fwd: null pointer");
  MHR_REQUIRE(n_tok > 0 && seq_len > 0 && n_tok % seq_len == 0 && n_heads > 0 && dim > 0 && dim % 4 == 0,
              "heads_residual_fwd: bad sizes (n_tok=%lld seq_len=%d heads=%d dim=%d)", (long long)n_tok, seq_len, n_heads, dim);
  hipLaunchKernelGGL(heads_residual_fwd_kernel, dim3(mhr_grid_for(n_tok, 8)), dim3(256), 0, (hipStream_t)stream, x,
                     (const bf16_t*)z_bf16, out, n_tok, seq_len, n_heads, dim);
  MHR_CHECK_LAUNCH("heads_residual_fwd");
  return MHR_OK;
}

extern "C" int mhr_heads_residual_bwd(const float* d_out, const void* z_bf16, void* dz_bf16, float* dx, int64_t n_tok,
                                      int seq_len, int n_heads, int dim, void* stream) {
  MHR_REQUIRE(d_out && z_bf16 && dz_bf16 && dx, "heads_residual_bwd: null pointer");
  MHR_REQUIRE(n_tok > 0 && seq_len > 0 && n_tok % seq_len == 0 && n_heads > 0 && dim > 0 && dim % 4 == 0,
              "heads_residual_bwd: bad sizes (n_tok=%lld seq_len=%d heads=%d dim=%d)", (long long)n_tok, seq_len, n_heads, dim);
  hipLaunchKernelGGL(heads_residual_bwd_kernel, dim3(mhr_grid_for(n_tok, 8)), dim3(256), 0, (hipStream_t)stream, d_out,
                     (const bf16_t*)z_bf16, (bf16_t*)dz_bf16, dx, n_tok, seq_len, n_heads, dim);
  MHR_CHECK_LAUNCH("heads_residual_bwd");
  return MHR_OK;
}
