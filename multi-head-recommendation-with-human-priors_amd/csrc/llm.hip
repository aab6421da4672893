// Elementwise pieces of the LLM decoder blocks that sit between the library GEMMs (HLLM user / item towers):
// the SwiGLU gate of the MLP and the rotary position embedding of q / k.  HBM-bound, 16 B per lane per access.
//
// Reference (file:line under code/REC/model/HLLM/):
//   SwiGLU  modeling_llama.py:484  `down_proj(act_fn(gate_proj(x)) * up_proj(x))`  (xformers `swiglu` in
//           baichuan/modeling_baichuan.py:197-206) - here on the output of ONE [gate | up] GEMM;
//   RoPE    modeling_llama.py:426-441 `rotate_half` convention: out[i] = x[i] cos_i - x[i + hd/2] sin_i,
//           out[i + hd/2] = x[i + hd/2] cos_i + x[i] sin_i, cos/sin of position * theta^(-2i/hd) (332-344), computed in
//           fp32 and rounded back to the activations' bf16 (441).
#include "mhr_common.h"

namespace {

__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + __expf(-x)); }

// a[r, c] = silu(gu[r, c]) * gu[r, F + c]
__global__ __launch_bounds__(256) void swiglu_fwd_kernel(const bf16_t* __restrict__ gu, bf16_t* __restrict__ a, int64_t rows,
                                                         int F) {
  const int64_t n8 = rows * (F / 8);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / (F / 8);
    const int c = (int)(i % (F / 8)) * 8;
    const bf16x8 g = *reinterpret_cast<const bf16x8*>(gu + r * 2 * F + c);
    const bf16x8 u = *reinterpret_cast<const bf16x8*>(gu + r * 2 * F + F + c);
    bf16x8 o;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float x = (float)g[k];
      o[k] = (bf16_t)(x * sigmoid_f(x) * (float)u[k]);
    }
    *reinterpret_cast<bf16x8*>(a + r * F + c) = o;
  }
}

// dgu[r, c] = da * u * silu'(g);  dgu[r, F + c] = da * silu(g)
__global__ __launch_bounds__(256) void swiglu_bwd_kernel(const bf16_t* __restrict__ gu, const bf16_t* __restrict__ da,
                                                         bf16_t* __restrict__ dgu, int64_t rows, int F) {
  const int64_t n8 = rows * (F / 8);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / (F / 8);
    const int c = (int)(i % (F / 8)) * 8;
    const bf16x8 g = *reinterpret_cast<const bf16x8*>(gu + r * 2 * F + c);
    const bf16x8 u = *reinterpret_cast<const bf16x8*>(gu + r * 2 * F + F + c);
    const bf16x8 d = *reinterpret_cast<const bf16x8*>(da + r * F + c);
    bf16x8 dg, du;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float x = (float)g[k], s = sigmoid_f(x), dd = (float)d[k];
      dg[k] = (bf16_t)(dd * (float)u[k] * s * (1.0f + x * (1.0f - s)));
      du[k] = (bf16_t)(dd * x * s);
    }
    *reinterpret_cast<bf16x8*>(dgu + r * 2 * F + c) = dg;
    *reinterpret_cast<bf16x8*>(dgu + r * 2 * F + F + c) = du;
  }
}

// In-place rotation of `n_heads` consecutive heads of width hd starting at column 0 of `x` (row stride `stride`).
// One thread per (token, head, 8 columns of the first half) - it owns the matching 8 columns of the second half.
__global__ __launch_bounds__(256) void rope_kernel(bf16_t* __restrict__ x, int64_t stride, const int32_t* __restrict__ pos,
                                                   const float* __restrict__ cos_t, const float* __restrict__ sin_t,
                                                   int64_t n_tok, int seq_len, int n_heads, int hd, int max_pos, float sign) {
  const int half = hd / 2, per_head = half / 8;
  const int64_t n = n_tok * n_heads * per_head;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int j = (int)(i % per_head) * 8;
    const int h = (int)((i / per_head) % n_heads);
    const int64_t t = i / ((int64_t)per_head * n_heads);
    int p = pos ? pos[t] : (int)(t % seq_len);
    p = p < 0 ? 0 : (p >= max_pos ? max_pos - 1 : p);
    bf16_t* row = x + t * stride + h * hd;
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(row + j);
    const bf16x8 b = *reinterpret_cast<const bf16x8*>(row + half + j);
    const float* cp = cos_t + (int64_t)p * half + j;
    const float* sp = sin_t + (int64_t)p * half + j;
    bf16x8 oa, ob;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float c = cp[k], s = sign * sp[k];
      const float xa = (float)a[k], xb = (float)b[k];
      oa[k] = (bf16_t)(xa * c - xb * s);
      ob[k] = (bf16_t)(xb * c + xa * s);
    }
    *reinterpret_cast<bf16x8*>(row + j) = oa;
    *reinterpret_cast<bf16x8*>(row + half + j) = ob;
  }
}

}  // namespace

extern "C" int mhr_swiglu_fwd(const void* gate_up_bf16, void* act_bf16, int64_t rows, int ffn, void* stream) {
  MHR_REQUIRE(gate_up_bf16 && act_bf16, "swiglu_fwd: null pointer");
  MHR_REQUIRE(ffn > 0 && ffn % 8 == 0, "swiglu_fwd: ffn=%d must be a positive multiple of 8", ffn);
  if (rows <= 0) return MHR_OK;
  const int grid = mhr_grid_for(rows * (ffn / 8), 256);
  hipLaunchKernelGGL(swiglu_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)gate_up_bf16,
                     (bf16_t*)act_bf16, rows, ffn);
  MHR_CHECK_LAUNCH("swiglu_fwd");
  return MHR_OK;
}

extern "C" int mhr_swiglu_bwd(const void* gate_up_bf16, const void* d_act_bf16, void* d_gate_up_bf16, int64_t rows, int ffn,
                              void* stream) {
  MHR_REQUIRE(gate_up_bf16 && d_act_bf16 && d_gate_up_bf16, "swiglu_bwd: null pointer");
  MHR_REQUIRE(ffn > 0 && ffn % 8 == 0, "swiglu_bwd: ffn=%d must be a positive multiple of 8", ffn);
  if (rows <= 0) return MHR_OK;
  const int grid = mhr_grid_for(rows * (ffn / 8), 256);
  hipLaunchKernelGGL(swiglu_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)gate_up_bf16,
                     (const bf16_t*)d_act_bf16, (bf16_t*)d_gate_up_bf16, rows, ffn);
  MHR_CHECK_LAUNCH("swiglu_bwd");
  return MHR_OK;
}

extern "C" int mhr_rope_inplace(void* x_bf16, int64_t row_stride, const int32_t* positions, const float* cos_table,
                                const float* sin_table, int64_t n_tokens, int seq_len, int n_heads, int head_dim, int max_pos,
                                int inverse, void* stream) {
  MHR_REQUIRE(x_bf16 && cos_table && sin_table, "rope_inplace: null pointer");
  MHR_REQUIRE(head_dim > 0 && head_dim % 16 == 0, "rope_inplace: head_dim=%d must be a multiple of 16", head_dim);
  MHR_REQUIRE(row_stride % 8 == 0 && n_heads > 0 && max_pos > 0, "rope_inplace: bad sizes");
  MHR_REQUIRE(positions || seq_len > 0, "rope_inplace: positions or seq_len required");
  if (n_tokens <= 0) return MHR_OK;
  const int grid = mhr_grid_for(n_tokens * n_heads * (head_dim / 16), 256);
  hipLaunchKernelGGL(rope_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (bf16_t*)x_bf16, row_stride, positions,
                     cos_table, sin_table, n_tokens, seq_len > 0 ? seq_len : 1, n_heads, head_dim, max_pos,
                     inverse ? -1.0f : 1.0f);
  MHR_CHECK_LAUNCH("rope_inplace");
  return MHR_OK;
}
