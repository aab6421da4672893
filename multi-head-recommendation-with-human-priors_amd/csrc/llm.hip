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

// (sigmoid_f: mhr_common.h - one v_exp and one v_rcp)

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

// ------------------------------------------------------------------------------------------
// out[c] += sum_r x[r, c]   (x bf16 [rows, cols], out fp32 [cols])
// ------------------------------------------------------------------------------------------
// The reduction of split-K weight-gradient partials ([S, N*K] bf16) and the bias gradient (column sums of dy) straight into
// the optimizer's flat fp32 gradient buffer: torch's generic dim-0 reduce takes 13 us for 8 MB, 26 times per step at cfg1.
// Block = 32 column threads (8 columns each) x 8 row lanes; row ranges are split over blockIdx.y when there are many rows
// (then the partial sums meet in out[] through float atomics; with one row range the update is a plain read-modify-write).
namespace {
template <typename T> struct Row8;
template <> struct Row8<bf16_t> {
  bf16x8 v;
  __device__ __forceinline__ void load(const bf16_t* p) { v = *reinterpret_cast<const bf16x8*>(p); }
  __device__ __forceinline__ float operator[](int e) const { return (float)v[e]; }
};
template <> struct Row8<float> {
  f32x4 a, b;
  __device__ __forceinline__ void load(const float* p) {
    a = reinterpret_cast<const f32x4*>(p)[0];
    b = reinterpret_cast<const f32x4*>(p)[1];
  }
  __device__ __forceinline__ float operator[](int e) const { return e < 4 ? a[e] : b[e - 4]; }
};

template <typename T>
__device__ __forceinline__ void sum_rows_body(const T* __restrict__ x, int64_t rows, int64_t cols, float* __restrict__ out,
                                              int64_t rows_per_block) {
  __shared__ float red[8][32][8];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int64_t c0 = ((int64_t)blockIdx.x * 32 + tx) * 8;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c0 < cols) {
    int64_t r = r0 + ty;
    for (; r + 56 < r1; r += 64) {                       // eight independent 16-byte loads in flight per thread
      Row8<T> v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u].load(x + (r + 8 * u) * cols + c0);
#pragma unroll
      for (int e = 0; e < 8; ++e)
        acc[e] += ((v[0][e] + v[1][e]) + (v[2][e] + v[3][e])) + ((v[4][e] + v[5][e]) + (v[6][e] + v[7][e]));
    }
    for (; r + 24 < r1; r += 32) {
      Row8<T> v0, v1, v2, v3;
      v0.load(x + r * cols + c0);
      v1.load(x + (r + 8) * cols + c0);
      v2.load(x + (r + 16) * cols + c0);
      v3.load(x + (r + 24) * cols + c0);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += (v0[e] + v1[e]) + (v2[e] + v3[e]);
    }
    for (; r < r1; r += 8) {
      Row8<T> v;
      v.load(x + r * cols + c0);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += v[e];
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[ty][tx][e] = acc[e];
  __syncthreads();
  if (ty == 0 && c0 < cols) {
    float s[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      s[e] = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) s[e] += red[j][tx][e];
    }
    if (gridDim.y > 1) {
#pragma unroll
      for (int e = 0; e < 8; ++e) atomicAdd(out + c0 + e, s[e]);
    } else {                                               // 32 contiguous bytes per thread: two 16-byte read-modify-writes
      f32x4* o = reinterpret_cast<f32x4*>(out + c0);
      f32x4 a = o[0], b = o[1];
      a += f32x4{s[0], s[1], s[2], s[3]};
      b += f32x4{s[4], s[5], s[6], s[7]};
      o[0] = a;
      o[1] = b;
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void sum_rows_into_kernel(const T* __restrict__ x, int64_t rows, int64_t cols,
                                                            float* __restrict__ out, int64_t rows_per_block) {
  sum_rows_body<T>(x, rows, cols, out, rows_per_block);
}

// the same reduction for MANY equally shaped matrices in one launch (blockIdx.z picks the matrix): the bias gradients of the
// encoder's output projections - eight [B L, D] column sums per step at cfg1, each too small to fill the chip on its own
__global__ __launch_bounds__(256) void sum_rows_many_kernel(const int64_t* __restrict__ ptrs, int n, int64_t rows, int64_t cols,
                                                            int64_t rows_per_block) {
  const bf16_t* x = reinterpret_cast<const bf16_t*>(ptrs[blockIdx.z]);
  float* out = reinterpret_cast<float*>(ptrs[n + blockIdx.z]);
  sum_rows_body<bf16_t>(x, rows, cols, out, rows_per_block);
}
}  // namespace

static int sum_rows_launch(const void* x, bool f32, int64_t rows, int64_t cols, float* out, void* stream, const char* who) {
  MHR_REQUIRE(x && out, "%s: null pointer", who);
  MHR_REQUIRE(rows >= 0 && cols > 0 && cols % 8 == 0, "%s: cols=%lld must be a positive multiple of 8", who, (long long)cols);
  MHR_REQUIRE((uintptr_t)out % 16 == 0 && (uintptr_t)x % 16 == 0, "%s: buffers must be 16-byte aligned", who);
  if (rows == 0) return MHR_OK;
  const int64_t col_blocks = (cols / 8 + 31) / 32;
  MHR_REQUIRE(col_blocks < (1ll << 31), "%s: too many columns", who);
  // a few row ranges when there are few columns (many would only contend on the same atomics: 256 ranges adding into the
  // 256 floats of a bias gradient doubled the launch's time - measured twice, rounds 1 and 2); the row loop keeps eight loads
  // in flight instead
  int64_t splits = 1;
  // (measured: [25600, 1024] 128 workgroups 15.9 us, 256: 21 us, 512: 33 us; [25600, 256] 64 workgroups 16.7 us, 128: 28 us)
  while (splits < 64 && col_blocks * splits < 128 && rows / (splits * 2) >= 64) splits *= 2;
  if (mhr_deterministic()) splits = 1;             // one row range per column block: a plain read-modify-write, one fixed order
  const int64_t rpb = (rows + splits - 1) / splits;
  const dim3 grid((unsigned)col_blocks, (unsigned)((rows + rpb - 1) / rpb));
  if (f32)
    hipLaunchKernelGGL((sum_rows_into_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, (const float*)x, rows, cols, out, rpb);
  else
    hipLaunchKernelGGL((sum_rows_into_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, rows, cols, out, rpb);
  MHR_CHECK_LAUNCH(who);
  return MHR_OK;
}

extern "C" int mhr_sum_rows_into(const void* x_bf16, int64_t rows, int64_t cols, float* out, void* stream) {
  return sum_rows_launch(x_bf16, false, rows, cols, out, stream, "sum_rows_into");
}

extern "C" int mhr_sum_rows_f32_into(const float* x, int64_t rows, int64_t cols, float* out, void* stream) {
  return sum_rows_launch(x, true, rows, cols, out, stream, "sum_rows_f32_into");
}

extern "C" int mhr_sum_rows_many(const int64_t* ptrs, int n, int64_t rows, int64_t cols, void* stream) {
  MHR_REQUIRE(ptrs && n > 0 && n <= 65535, "sum_rows_many: bad arguments");
  MHR_REQUIRE(rows > 0 && cols > 0 && cols % 8 == 0, "sum_rows_many: cols=%lld must be a positive multiple of 8", (long long)cols);
  const int64_t col_blocks = (cols / 8 + 31) / 32;
  int64_t splits = 1;
  // one workgroup per CU: more row ranges only contend on the float atomics of the n x cols destinations (8 x [25600, 256]:
  // 256 workgroups 22 us, 512: 30 us, 1024: 53 us, 2048: 103 us)
  while (splits < 1024 && col_blocks * splits * n < 256 && rows / (splits * 2) >= 64) splits *= 2;
  if (splits < 2 && rows >= 128) splits = 2;        // (>= 2 row ranges: the atomic path; one range would do a plain read-modify-write)
  if (mhr_deterministic()) splits = 1;              // deterministic mode: the plain read-modify-write, one fixed order per column
  const int64_t rpb = (rows + splits - 1) / splits;
  hipLaunchKernelGGL(sum_rows_many_kernel, dim3((unsigned)col_blocks, (unsigned)((rows + rpb - 1) / rpb), (unsigned)n), dim3(256), 0,
                     (hipStream_t)stream, ptrs, n, rows, cols, rpb);
  MHR_CHECK_LAUNCH("sum_rows_many");
  return MHR_OK;
}
