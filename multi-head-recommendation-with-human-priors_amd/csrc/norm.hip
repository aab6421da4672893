// Row-wise normalisation kernels of the HSTU layer: affine-free LayerNorm fwd/bwd, the fused
// silu(u) * LayerNorm(attn) * dropout gate fwd/bwd, and L2 row normalisation.
// One wave per row, 16 B (f32) / 8 B (bf16) per lane per 256-column chunk, statistics in f32 by
// wave shuffles.  HBM-bound.
//
// Reference ops replaced: F.layer_norm model/IDNet/hstu.py:213-219 (eps 1e-6, no affine),
// u * norm(attn) + F.dropout hstu.py:277-285, x / x.norm() hstu.py:605-606,672,966,975,1021.
#include "mhr_common.h"

template <int NC>
struct RowRegs {
  f32x4 v[NC];
};

template <typename T, int NC>
__device__ __forceinline__ void load_row(const T* p, int dim, int lane, RowRegs<NC>& r) {
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    int c = lane * 4 + i * 256;
    if (c < dim) r.v[i] = Vec4IO<T>::load(p + c);
    else r.v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
}
template <typename T, int NC>
__device__ __forceinline__ void store_row(T* p, int dim, int lane, const RowRegs<NC>& r) {
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    int c = lane * 4 + i * 256;
    if (c < dim) Vec4IO<T>::store(p + c, r.v[i]);
  }
}
template <int NC>
__device__ __forceinline__ float row_sum(const RowRegs<NC>& r) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NC; ++i) s += (r.v[i][0] + r.v[i][1]) + (r.v[i][2] + r.v[i][3]);
  return wave_sum(s);
}
// mean / rstd with masked tail (columns >= dim hold zeros and must not enter the variance)
template <int NC>
__device__ __forceinline__ void row_stats(const RowRegs<NC>& r, int dim, int lane, float eps, float& mean, float& rstd) {
  mean = row_sum(r) / (float)dim;
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    int c = lane * 4 + i * 256;
    if (c < dim) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float d = r.v[i][k] - mean;
        s += d * d;
      }
    }
  }
  float var = wave_sum(s) / (float)dim;
  rstd = rsqrtf(var + eps);
}

// Rows in front of a sequence's first valid key (the loaders pad at the FRONT: trainset.py:111-137, evalset.py:34-41) take part in
// nothing: no valid key reads them, the loss and the decode skip them, and every gradient that reaches them is exactly zero.
// Given `first_row[b]` (mhr_attn_seq_layout) and the sequence length the row-wise kernels of the encoder do not LOAD such rows:
// their operands read as zeros, the arithmetic runs unchanged and writes the zeros it produces (so the weight-gradient products
// over all rows meet finite operands whose partner is zero).  The last row of a sequence always counts as live - the decode
// reads it even from an all-padding sequence.  first_row == NULL: every row is live.
__device__ __forceinline__ bool row_is_live(const int32_t* __restrict__ first_row, int seq_len, int64_t row) {
  if (!first_row) return true;
  const int r32 = (int)row, b = r32 / seq_len, l = r32 - b * seq_len;
  return l >= min(first_row[b], seq_len - 1);
}
template <typename T, int NC>
__device__ __forceinline__ void load_row_if(bool live, const T* p, int dim, int lane, RowRegs<NC>& r) {
  if (live) {
    load_row<T, NC>(p, dim, lane, r);
  } else {
#pragma unroll
    for (int i = 0; i < NC; ++i) r.v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
}

#define WAVE_ROW_LOOP(rows)                                                                                         \
  const int lane = threadIdx.x & 63;                                                                                \
  const int64_t wave0 = (int64_t)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); \
  const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x >> 6);                                                   \
  for (int64_t row = wave0; row < (rows); row += n_waves)

// ------------------------------------------------------------------------------------------
// LayerNorm
// ------------------------------------------------------------------------------------------
template <typename XT, typename YT, int NC>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const XT* __restrict__ x, YT* __restrict__ y, float* __restrict__ mean_o,
                                                     float* __restrict__ rstd_o, int64_t rows, int dim, float eps) {
  WAVE_ROW_LOOP(rows) {
    RowRegs<NC> r;
    load_row<XT, NC>(x + row * dim, dim, lane, r);
    float mean, rstd;
    row_stats<NC>(r, dim, lane, eps, mean, rstd);
#pragma unroll
    for (int i = 0; i < NC; ++i)
#pragma unroll
      for (int k = 0; k < 4; ++k) r.v[i][k] = (r.v[i][k] - mean) * rstd;
    store_row<YT, NC>(y + row * dim, dim, lane, r);
    if (lane == 0) {
      if (mean_o) mean_o[row] = mean;
      if (rstd_o) rstd_o[row] = rstd;
    }
  }
}

template <typename DT, typename XT, typename OT, int NC>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const DT* __restrict__ dy, const XT* __restrict__ x,
                                                     const float* __restrict__ mean_i, const float* __restrict__ rstd_i,
                                                     OT* __restrict__ dx, int accumulate, int64_t rows, int dim) {
  WAVE_ROW_LOOP(rows) {
    RowRegs<NC> g, xv;
    load_row<DT, NC>(dy + row * dim, dim, lane, g);
    load_row<XT, NC>(x + row * dim, dim, lane, xv);
    const float mean = mean_i[row], rstd = rstd_i[row];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      int c = lane * 4 + i * 256;
      if (c < dim) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float xh = (xv.v[i][k] - mean) * rstd;
          xv.v[i][k] = xh;
          s1 += g.v[i][k];
          s2 += g.v[i][k] * xh;
        }
      }
    }
    s1 = wave_sum(s1) / (float)dim;
    s2 = wave_sum(s2) / (float)dim;
    RowRegs<NC> o;
    if (accumulate) load_row<OT, NC>(dx + row * dim, dim, lane, o);
#pragma unroll
    for (int i = 0; i < NC; ++i)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float d = rstd * (g.v[i][k] - s1 - xv.v[i][k] * s2);
        o.v[i][k] = accumulate ? o.v[i][k] + d : d;
      }
    store_row<OT, NC>(dx + row * dim, dim, lane, o);
  }
}

static inline int nc_for(int dim) { return dim <= 256 ? 1 : dim <= 512 ? 2 : dim <= 1024 ? 4 : 8; }

#define DISPATCH_NC(dim, MACRO) \
  switch (nc_for(dim)) {        \
    case 1: MACRO(1); break;    \
    case 2: MACRO(2); break;    \
    case 4: MACRO(4); break;    \
    default: MACRO(8); break;   \
  }

extern "C" int mhr_layernorm_fwd(const void* x, int x_dtype, void* y, int y_dtype, float* mean, float* rstd, int64_t rows,
                                 int dim, float eps, void* stream) {
  MHR_REQUIRE(x && y, "layernorm_fwd: null pointer");
  MHR_REQUIRE(dim > 0 && dim % 4 == 0 && dim <= 2048, "layernorm_fwd: dim=%d unsupported (multiple of 4, <= 2048)", dim);
  if (rows <= 0) return MHR_OK;
  hipStream_t s = (hipStream_t)stream;
  int grid = mhr_grid_for(rows, 4);
  bool xb = x_dtype == MHR_BF16, yb = y_dtype == MHR_BF16;
#define L(NC)                                                                                                           \
  if (xb && yb) hipLaunchKernelGGL((ln_fwd_kernel<bf16_t, bf16_t, NC>), dim3(grid), dim3(256), 0, s, (const bf16_t*)x,  \
                                   (bf16_t*)y, mean, rstd, rows, dim, eps);                                             \
  else if (xb) hipLaunchKernelGGL((ln_fwd_kernel<bf16_t, float, NC>), dim3(grid), dim3(256), 0, s, (const bf16_t*)x,    \
                                  (float*)y, mean, rstd, rows, dim, eps);                                               \
  else if (yb) hipLaunchKernelGGL((ln_fwd_kernel<float, bf16_t, NC>), dim3(grid), dim3(256), 0, s, (const float*)x,     \
                                  (bf16_t*)y, mean, rstd, rows, dim, eps);                                              \
  else hipLaunchKernelGGL((ln_fwd_kernel<float, float, NC>), dim3(grid), dim3(256), 0, s, (const float*)x, (float*)y,   \
                          mean, rstd, rows, dim, eps)
  DISPATCH_NC(dim, L);
#undef L
  MHR_CHECK_LAUNCH("layernorm_fwd");
  return MHR_OK;
}

extern "C" int mhr_layernorm_bwd(const void* dy, int dy_dtype, const void* x, int x_dtype, const float* mean,
                                 const float* rstd, void* dx, int dx_dtype, int accumulate, int64_t rows, int dim,
                                 void* stream) {
  MHR_REQUIRE(dy && x && mean && rstd && dx, "layernorm_bwd: null pointer");
  MHR_REQUIRE(dim > 0 && dim % 4 == 0 && dim <= 2048, "layernorm_bwd: dim=%d unsupported", dim);
  MHR_REQUIRE(!(accumulate && dx_dtype != MHR_F32), "layernorm_bwd: accumulate needs f32 dx");
  if (rows <= 0) return MHR_OK;
  hipStream_t s = (hipStream_t)stream;
  int grid = mhr_grid_for(rows, 4);
  // supported combinations: dy in {bf16,f32}, x in {f32,bf16}, dx in {f32,bf16}
#define LK(DT, XT, OT, NC)                                                                                    \
  hipLaunchKernelGGL((ln_bwd_kernel<DT, XT, OT, NC>), dim3(grid), dim3(256), 0, s, (const DT*)dy, (const XT*)x, mean, \
                     rstd, (OT*)dx, accumulate, rows, dim)
#define L(NC)                                                                   \
  {                                                                             \
    bool db = dy_dtype == MHR_BF16, xb = x_dtype == MHR_BF16, ob = dx_dtype == MHR_BF16; \
    if (db && xb && ob) LK(bf16_t, bf16_t, bf16_t, NC);                         \
    else if (db && xb) LK(bf16_t, bf16_t, float, NC);                           \
    else if (db && ob) LK(bf16_t, float, bf16_t, NC);                           \
    else if (db) LK(bf16_t, float, float, NC);                                  \
    else if (xb && ob) LK(float, bf16_t, bf16_t, NC);                           \
    else if (xb) LK(float, bf16_t, float, NC);                                  \
    else if (ob) LK(float, float, bf16_t, NC);                                  \
    else LK(float, float, float, NC);                                           \
  }
  DISPATCH_NC(dim, L);
#undef L
#undef LK
  MHR_CHECK_LAUNCH("layernorm_bwd");
  return MHR_OK;
}

// ------------------------------------------------------------------------------------------
// Residual add fused with the NEXT layer's LayerNorm (reference hstu.py:286-287 `output + x`, then 241 of the next
// layer): x_out = x + y (fp32 stream + bf16 branch), xn = LN(x_out).  12 B per element instead of 16 for the two
// separate kernels; the backward returns the total gradient of x_out once in fp32 (residual path) and once in bf16
// (the branch's GEMM input), which also removes the autograd add and the cast.
// ------------------------------------------------------------------------------------------
template <int NC>
__global__ __launch_bounds__(256) void add_ln_fwd_kernel(const float* __restrict__ x, const bf16_t* __restrict__ y,
                                                         float* __restrict__ x_out, bf16_t* __restrict__ xn,
                                                         float* __restrict__ mean_o, float* __restrict__ rstd_o,
                                                         int64_t rows, int dim, float eps, const int32_t* __restrict__ first_row,
                                                         int seq_len) {
  WAVE_ROW_LOOP(rows) {
    RowRegs<NC> r, b;
    const bool live = row_is_live(first_row, seq_len, row);
    load_row_if<float, NC>(live, x + row * dim, dim, lane, r);
    load_row_if<bf16_t, NC>(live, y + row * dim, dim, lane, b);
#pragma unroll
    for (int i = 0; i < NC; ++i)
#pragma unroll
      for (int k = 0; k < 4; ++k) r.v[i][k] += b.v[i][k];
    store_row<float, NC>(x_out + row * dim, dim, lane, r);
    float mean, rstd;
    row_stats<NC>(r, dim, lane, eps, mean, rstd);
#pragma unroll
    for (int i = 0; i < NC; ++i)
#pragma unroll
      for (int k = 0; k < 4; ++k) r.v[i][k] = (r.v[i][k] - mean) * rstd;
    store_row<bf16_t, NC>(xn + row * dim, dim, lane, r);
    if (lane == 0) {
      mean_o[row] = mean;
      rstd_o[row] = rstd;
    }
  }
}

template <int NC>
__global__ __launch_bounds__(256) void add_ln_bwd_kernel(const bf16_t* __restrict__ d_xn, const float* __restrict__ x_out,
                                                         const float* __restrict__ mean_i, const float* __restrict__ rstd_i,
                                                         const float* __restrict__ d_xout, float* __restrict__ dx,
                                                         bf16_t* __restrict__ dy, int64_t rows, int dim,
                                                         const int32_t* __restrict__ first_row, int seq_len) {
  WAVE_ROW_LOOP(rows) {
    RowRegs<NC> g, xv, o;
    const bool live = row_is_live(first_row, seq_len, row);
    load_row_if<bf16_t, NC>(live, d_xn + row * dim, dim, lane, g);
    load_row_if<float, NC>(live, x_out + row * dim, dim, lane, xv);
    load_row_if<float, NC>(live, d_xout + row * dim, dim, lane, o);
    const float mean = live ? mean_i[row] : 0.f, rstd = live ? rstd_i[row] : 0.f;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      int c = lane * 4 + i * 256;
      if (c < dim) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float xh = (xv.v[i][k] - mean) * rstd;
          xv.v[i][k] = xh;
          s1 += g.v[i][k];
          s2 += g.v[i][k] * xh;
        }
      }
    }
    s1 = wave_sum(s1) / (float)dim;
    s2 = wave_sum(s2) / (float)dim;
#pragma unroll
    for (int i = 0; i < NC; ++i)
#pragma unroll
      for (int k = 0; k < 4; ++k) o.v[i][k] += rstd * (g.v[i][k] - s1 - xv.v[i][k] * s2);
    store_row<float, NC>(dx + row * dim, dim, lane, o);
    store_row<bf16_t, NC>(dy + row * dim, dim, lane, o);
  }
}

#define MHR_REQUIRE_SEQ(name)                                                                                        \
  MHR_REQUIRE(!first_row || (seq_len > 0 && rows % seq_len == 0 && rows < (1ll << 31)), name ": first_row needs rows = B * seq_len")

extern "C" int mhr_add_layernorm_fwd(const float* x, const void* y_bf16, float* x_out, void* xn_bf16, float* mean, float* rstd,
                                     int64_t rows, int dim, float eps, const int32_t* first_row, int seq_len, void* stream) {
  MHR_REQUIRE_SEQ("add_layernorm_fwd");
  MHR_REQUIRE(x && y_bf16 && x_out && xn_bf16 && mean && rstd, "add_layernorm_fwd: null pointer");
  MHR_REQUIRE(dim > 0 && dim % 4 == 0 && dim <= 2048, "add_layernorm_fwd: dim=%d unsupported", dim);
  if (rows <= 0) return MHR_OK;
  const int grid = mhr_grid_for(rows, 4);
#define L(NC)                                                                                                          \
  hipLaunchKernelGGL((add_ln_fwd_kernel<NC>), dim3(grid), dim3(256), 0, (hipStream_t)stream, x, (const bf16_t*)y_bf16, \
                     x_out, (bf16_t*)xn_bf16, mean, rstd, rows, dim, eps, first_row, seq_len)
  DISPATCH_NC(dim, L);
#undef L
  MHR_CHECK_LAUNCH("add_layernorm_fwd");
  return MHR_OK;
}

extern "C" int mhr_add_layernorm_bwd(const void* d_xn_bf16, const float* x_out, const float* mean, const float* rstd,
                                     const float* d_xout, float* dx, void* dy_bf16, int64_t rows, int dim,
                                     const int32_t* first_row, int seq_len, void* stream) {
  MHR_REQUIRE_SEQ("add_layernorm_bwd");
  MHR_REQUIRE(d_xn_bf16 && x_out && mean && rstd && d_xout && dx && dy_bf16, "add_layernorm_bwd: null pointer");
  MHR_REQUIRE(dim > 0 && dim % 4 == 0 && dim <= 2048, "add_layernorm_bwd: dim=%d unsupported", dim);
  if (rows <= 0) return MHR_OK;
  const int grid = mhr_grid_for(rows, 4);
#define L(NC)                                                                                                              \
  hipLaunchKernelGGL((add_ln_bwd_kernel<NC>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)d_xn_bf16, x_out, \
                     mean, rstd, d_xout, dx, (bf16_t*)dy_bf16, rows, dim, first_row, seq_len)
  DISPATCH_NC(dim, L);
#undef L
  MHR_CHECK_LAUNCH("add_layernorm_bwd");
  return MHR_OK;
}

// ------------------------------------------------------------------------------------------
// o = silu(u) * LN(a) * dropmask
// ------------------------------------------------------------------------------------------
template <typename T, typename OT, int NC>
__global__ __launch_bounds__(256) void ln_gate_fwd_kernel(const T* __restrict__ u, int64_t u_stride, const T* __restrict__ a,
                                                          OT* __restrict__ o, float* __restrict__ mean_o,
                                                          float* __restrict__ rstd_o, int64_t rows, int dim, float eps,
                                                          float p, float keep_scale, uint64_t seed,
                                                          const int64_t* __restrict__ step_seed,
                                                          const int32_t* __restrict__ first_row, int seq_len) {
  seed = mhr_step_seed(seed, step_seed);
  WAVE_ROW_LOOP(rows) {
    RowRegs<NC> av, uv;
    const bool live = row_is_live(first_row, seq_len, row);
    load_row_if<T, NC>(live, a + row * dim, dim, lane, av);
    load_row_if<T, NC>(live, u + row * u_stride, dim, lane, uv);
    float mean, rstd;
    row_stats<NC>(av, dim, lane, eps, mean, rstd);
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      int c = lane * 4 + i * 256;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float val = silu_f(uv.v[i][k]) * ((av.v[i][k] - mean) * rstd);
        if (p > 0.f) {
          float r = mhr_uniform(seed, (uint64_t)row * dim + c + k);
          val = r < p ? 0.f : val * keep_scale;
        }
        av.v[i][k] = val;
      }
    }
    store_row<OT, NC>(o + row * dim, dim, lane, av);
    if (lane == 0) {
      mean_o[row] = mean;
      rstd_o[row] = rstd;
    }
  }
}

template <typename GT, typename T, int NC>
__global__ __launch_bounds__(256) void ln_gate_bwd_kernel(const GT* __restrict__ d_o, const T* __restrict__ u,
                                                          int64_t u_stride, const T* __restrict__ a,
                                                          const float* __restrict__ mean_i, const float* __restrict__ rstd_i,
                                                          T* __restrict__ du, int64_t du_stride, T* __restrict__ da,
                                                          int64_t rows, int dim, float p, float keep_scale, uint64_t seed,
                                                          const int64_t* __restrict__ step_seed,
                                                          const int32_t* __restrict__ first_row, int seq_len) {
  seed = mhr_step_seed(seed, step_seed);
  WAVE_ROW_LOOP(rows) {
    RowRegs<NC> g, uv, av;
    const bool live = row_is_live(first_row, seq_len, row);
    load_row_if<GT, NC>(live, d_o + row * dim, dim, lane, g);
    load_row_if<T, NC>(live, u + row * u_stride, dim, lane, uv);
    load_row_if<T, NC>(live, a + row * dim, dim, lane, av);
    const float mean = live ? mean_i[row] : 0.f, rstd = live ? rstd_i[row] : 0.f;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      int c = lane * 4 + i * 256;
      if (c < dim) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float gk = g.v[i][k];
          if (p > 0.f) {
            float r = mhr_uniform(seed, (uint64_t)row * dim + c + k);
            gk = r < p ? 0.f : gk * keep_scale;
          }
          float xh = (av.v[i][k] - mean) * rstd;
          float up = uv.v[i][k];
          float su, dsu;
          silu_both(up, su, dsu);             // one sigmoid for both
          float gy = gk * su;                 // grad w.r.t. LN(a)
          uv.v[i][k] = gk * xh * dsu;          // grad w.r.t. pre-activation u
          av.v[i][k] = xh;
          g.v[i][k] = gy;
          s1 += gy;
          s2 += gy * xh;
        }
      }
    }
    s1 = wave_sum(s1) / (float)dim;
    s2 = wave_sum(s2) / (float)dim;
#pragma unroll
    for (int i = 0; i < NC; ++i)
#pragma unroll
      for (int k = 0; k < 4; ++k) av.v[i][k] = rstd * (g.v[i][k] - s1 - av.v[i][k] * s2);
    store_row<T, NC>(du + row * du_stride, dim, lane, uv);
    store_row<T, NC>(da + row * dim, dim, lane, av);
  }
}


// dim == 256, bf16 in and out (the cfg1 encoder): a HALF-wave per row, 16 bytes per lane and tensor, RPH rows per half-wave with
// all their loads issued together - four rows per wave in flight instead of one (the one-wave-per-row form above moved 39 MB in
// 14.4 us, 2.7 TB/s: load latency, not bandwidth).  Same arithmetic per element, same dropout index (row * dim + column).
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <int RPH>
__global__ __launch_bounds__(256) void ln_gate_fwd_h256_kernel(const bf16_t* __restrict__ u, int64_t u_stride, const bf16_t* __restrict__ a,
                                                               bf16_t* __restrict__ o, float* __restrict__ mean_o,
                                                               float* __restrict__ rstd_o, int64_t rows, float eps, float p,
                                                               float keep_scale, uint64_t seed, const int64_t* __restrict__ step_seed) {
  constexpr int DIM = 256;
  seed = mhr_step_seed(seed, step_seed);
  const int lane = threadIdx.x & 63, hl = lane & 31, hw = lane >> 5;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t row0 = (wave * 2 + hw) * RPH;
  bf16x8 av[RPH], uv[RPH];
#pragma unroll
  for (int j = 0; j < RPH; ++j) {
    const int64_t rr = min(row0 + j, rows - 1);
    av[j] = *reinterpret_cast<const bf16x8*>(a + rr * DIM + hl * 8);
    uv[j] = *reinterpret_cast<const bf16x8*>(u + rr * u_stride + hl * 8);
  }
#pragma unroll
  for (int j = 0; j < RPH; ++j) {
    const int64_t row = row0 + j;
    float x[8], s = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      x[e] = (float)av[j][e];
      s += x[e];
    }
    const float mean = half_sum(s) / (float)DIM;
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) q += (x[e] - mean) * (x[e] - mean);
    const float rstd = rsqrtf(half_sum(q) / (float)DIM + eps);
    bf16x8 out;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float val = silu_f((float)uv[j][e]) * ((x[e] - mean) * rstd);
      if (p > 0.f) {
        const float r = mhr_uniform(seed, (uint64_t)row * DIM + hl * 8 + e);
        val = r < p ? 0.f : val * keep_scale;
      }
      out[e] = (bf16_t)val;
    }
    if (row < rows) {
      *reinterpret_cast<bf16x8*>(o + row * DIM + hl * 8) = out;
      if (hl == 0) {
        mean_o[row] = mean;
        rstd_o[row] = rstd;
      }
    }
  }
}

template <int RPH>
__global__ __launch_bounds__(256) void ln_gate_bwd_h256_kernel(const bf16_t* __restrict__ d_o, const bf16_t* __restrict__ u, int64_t u_stride,
                                                               const bf16_t* __restrict__ a, const float* __restrict__ mean_i,
                                                               const float* __restrict__ rstd_i, bf16_t* __restrict__ du, int64_t du_stride,
                                                               bf16_t* __restrict__ da, int64_t rows, float p, float keep_scale,
                                                               uint64_t seed, const int64_t* __restrict__ step_seed) {
  constexpr int DIM = 256;
  seed = mhr_step_seed(seed, step_seed);
  const int lane = threadIdx.x & 63, hl = lane & 31, hw = lane >> 5;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t row0 = (wave * 2 + hw) * RPH;
  bf16x8 gv[RPH], av[RPH], uv[RPH];
  float mean[RPH], rstd[RPH];
#pragma unroll
  for (int j = 0; j < RPH; ++j) {
    const int64_t rr = min(row0 + j, rows - 1);
    gv[j] = *reinterpret_cast<const bf16x8*>(d_o + rr * DIM + hl * 8);
    av[j] = *reinterpret_cast<const bf16x8*>(a + rr * DIM + hl * 8);
    uv[j] = *reinterpret_cast<const bf16x8*>(u + rr * u_stride + hl * 8);
    mean[j] = mean_i[rr];
    rstd[j] = rstd_i[rr];
  }
#pragma unroll
  for (int j = 0; j < RPH; ++j) {
    const int64_t row = row0 + j;
    float gy[8], xh[8], s1 = 0.f, s2 = 0.f;
    bf16x8 du_o, da_o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float gk = (float)gv[j][e];
      if (p > 0.f) {
        const float r = mhr_uniform(seed, (uint64_t)row * DIM + hl * 8 + e);
        gk = r < p ? 0.f : gk * keep_scale;
      }
      xh[e] = ((float)av[j][e] - mean[j]) * rstd[j];
      float su, dsu;
      silu_both((float)uv[j][e], su, dsu);             // one sigmoid for both
      gy[e] = gk * su;                                 // grad w.r.t. LN(a)
      du_o[e] = (bf16_t)(gk * xh[e] * dsu);            // grad w.r.t. pre-activation u
      s1 += gy[e];
      s2 += gy[e] * xh[e];
    }
    s1 = half_sum(s1) / (float)DIM;
    s2 = half_sum(s2) / (float)DIM;
#pragma unroll
    for (int e = 0; e < 8; ++e) da_o[e] = (bf16_t)(rstd[j] * (gy[e] - s1 - xh[e] * s2));
    if (row < rows) {
      *reinterpret_cast<bf16x8*>(du + row * du_stride + hl * 8) = du_o;
      *reinterpret_cast<bf16x8*>(da + row * DIM + hl * 8) = da_o;
    }
  }
}

extern "C" int mhr_ln_gate_fwd(const void* u_base, int64_t u_stride, const void* a, int dtype, void* o, int o_dtype,
                               float* mean, float* rstd, int64_t rows, int dim, float eps, float dropout_p, uint64_t seed,
                               const int64_t* step_seed, const int32_t* first_row, int seq_len, void* stream) {
  MHR_REQUIRE(u_base && a && o && mean && rstd, "ln_gate_fwd: null pointer");
  MHR_REQUIRE_SEQ("ln_gate_fwd");
  MHR_REQUIRE(dim > 0 && dim % 4 == 0 && dim <= 2048 && u_stride >= dim, "ln_gate_fwd: dim=%d / stride unsupported", dim);
  MHR_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "ln_gate_fwd: dropout_p out of range");
  if (rows <= 0) return MHR_OK;
  hipStream_t s = (hipStream_t)stream;
  int grid = mhr_grid_for(rows, 4);
  float ks = 1.0f / (1.0f - dropout_p);
  bool tb = dtype == MHR_BF16, ob = o_dtype == MHR_BF16;
  if (tb && ob && dim == 256 && u_stride % 8 == 0 && ((uintptr_t)u_base | (uintptr_t)a | (uintptr_t)o) % 16 == 0) {
    constexpr int RPH = 2;                               // rows per half-wave
    const int64_t waves = (rows + 2 * RPH - 1) / (2 * RPH);
    hipLaunchKernelGGL((ln_gate_fwd_h256_kernel<RPH>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, (const bf16_t*)u_base, u_stride,
                       (const bf16_t*)a, (bf16_t*)o, mean, rstd, rows, eps, dropout_p, ks, seed, step_seed);
    // (first_row is not used by this form: its waves are one-shot and latency-bound - predicated loads, even with the liveness
    //  of a wave's four rows from scalar loads, left it 0.4 us SLOWER than loading the dead rows; 0.9 us in the backward.  In
    //  the encoder the dead rows of `a` are the attention's zeros and those of d_o are zero gradients, so the outputs are the
    //  same zeros either way)
    MHR_CHECK_LAUNCH("ln_gate_fwd");
    return MHR_OK;
  }
#define L(NC)                                                                                                             \
  if (tb && ob) hipLaunchKernelGGL((ln_gate_fwd_kernel<bf16_t, bf16_t, NC>), dim3(grid), dim3(256), 0, s,                 \
                                   (const bf16_t*)u_base, u_stride, (const bf16_t*)a, (bf16_t*)o, mean, rstd, rows, dim,  \
                                   eps, dropout_p, ks, seed, step_seed, first_row, seq_len);                                                            \
  else if (tb) hipLaunchKernelGGL((ln_gate_fwd_kernel<bf16_t, float, NC>), dim3(grid), dim3(256), 0, s,                   \
                                  (const bf16_t*)u_base, u_stride, (const bf16_t*)a, (float*)o, mean, rstd, rows, dim,    \
                                  eps, dropout_p, ks, seed, step_seed, first_row, seq_len);                                                             \
  else if (ob) hipLaunchKernelGGL((ln_gate_fwd_kernel<float, bf16_t, NC>), dim3(grid), dim3(256), 0, s,                   \
                                  (const float*)u_base, u_stride, (const float*)a, (bf16_t*)o, mean, rstd, rows, dim,     \
                                  eps, dropout_p, ks, seed, step_seed, first_row, seq_len);                                                             \
  else hipLaunchKernelGGL((ln_gate_fwd_kernel<float, float, NC>), dim3(grid), dim3(256), 0, s, (const float*)u_base,      \
                          u_stride, (const float*)a, (float*)o, mean, rstd, rows, dim, eps, dropout_p, ks, seed, step_seed, first_row, seq_len)
  DISPATCH_NC(dim, L);
#undef L
  MHR_CHECK_LAUNCH("ln_gate_fwd");
  return MHR_OK;
}

extern "C" int mhr_ln_gate_bwd(const void* d_o, int do_dtype, const void* u_base, int64_t u_stride, const void* a, int dtype,
                               const float* mean, const float* rstd, void* du_base, int64_t du_stride, void* da,
                               int64_t rows, int dim, float dropout_p, uint64_t seed, const int64_t* step_seed,
                               const int32_t* first_row, int seq_len, void* stream) {
  MHR_REQUIRE(d_o && u_base && a && mean && rstd && du_base && da, "ln_gate_bwd: null pointer");
  MHR_REQUIRE_SEQ("ln_gate_bwd");
  MHR_REQUIRE(dim > 0 && dim % 4 == 0 && dim <= 2048 && u_stride >= dim && du_stride >= dim, "ln_gate_bwd: bad dims");
  if (rows <= 0) return MHR_OK;
  hipStream_t s = (hipStream_t)stream;
  int grid = mhr_grid_for(rows, 4);
  float ks = 1.0f / (1.0f - dropout_p);
  bool tb = dtype == MHR_BF16, gb = do_dtype == MHR_BF16;
  if (tb && gb && dim == 256 && u_stride % 8 == 0 && du_stride % 8 == 0 &&
      ((uintptr_t)u_base | (uintptr_t)a | (uintptr_t)d_o | (uintptr_t)du_base | (uintptr_t)da) % 16 == 0) {
    constexpr int RPH = 2;
    const int64_t waves = (rows + 2 * RPH - 1) / (2 * RPH);
    hipLaunchKernelGGL((ln_gate_bwd_h256_kernel<RPH>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, (const bf16_t*)d_o,
                       (const bf16_t*)u_base, u_stride, (const bf16_t*)a, mean, rstd, (bf16_t*)du_base, du_stride, (bf16_t*)da, rows,
                       dropout_p, ks, seed, step_seed);
    MHR_CHECK_LAUNCH("ln_gate_bwd");
    return MHR_OK;
  }
#define LK(GT, T, NC)                                                                                                  \
  hipLaunchKernelGGL((ln_gate_bwd_kernel<GT, T, NC>), dim3(grid), dim3(256), 0, s, (const GT*)d_o, (const T*)u_base,    \
                     u_stride, (const T*)a, mean, rstd, (T*)du_base, du_stride, (T*)da, rows, dim, dropout_p, ks, seed, step_seed, first_row, seq_len)
#define L(NC)                          \
  if (gb && tb) LK(bf16_t, bf16_t, NC); \
  else if (gb) LK(bf16_t, float, NC);   \
  else if (tb) LK(float, bf16_t, NC);   \
  else LK(float, float, NC)
  DISPATCH_NC(dim, L);
#undef L
#undef LK
  MHR_CHECK_LAUNCH("ln_gate_bwd");
  return MHR_OK;
}

// ------------------------------------------------------------------------------------------
// out = x + y (f32 + bf16 -> f32) and its bf16 copy in one pass: the residual add behind the LAST encoder layer, whose
// output feeds the decoding heads twice - in f32 (residual of the heads) and in bf16 (operand of their GEMM).  The backward
// is the same arithmetic on the gradients: dx = d_out + d_out16, dy = bf16(dx).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void add_cast_kernel(const float* __restrict__ x, const bf16_t* __restrict__ y, float* __restrict__ out,
                                                       bf16_t* __restrict__ out16, int64_t n8) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
    const f32x4 a = reinterpret_cast<const f32x4*>(x)[2 * i], b = reinterpret_cast<const f32x4*>(x)[2 * i + 1];
    const bf16x8 v = reinterpret_cast<const bf16x8*>(y)[i];
    f32x4 o0, o1;
    bf16x8 w;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      o0[e] = a[e] + (float)v[e];
      o1[e] = b[e] + (float)v[4 + e];
      w[e] = (bf16_t)o0[e];
      w[4 + e] = (bf16_t)o1[e];
    }
    reinterpret_cast<f32x4*>(out)[2 * i] = o0;
    reinterpret_cast<f32x4*>(out)[2 * i + 1] = o1;
    reinterpret_cast<bf16x8*>(out16)[i] = w;
  }
}

extern "C" int mhr_add_cast(const float* x, const void* y_bf16, float* out, void* out_bf16, int64_t n, void* stream) {
  MHR_REQUIRE(x && y_bf16 && out && out_bf16, "add_cast: null pointer");
  MHR_REQUIRE(n >= 0 && n % 8 == 0, "add_cast: n=%lld must be a multiple of 8", (long long)n);
  MHR_REQUIRE(((uintptr_t)x | (uintptr_t)y_bf16 | (uintptr_t)out | (uintptr_t)out_bf16) % 16 == 0, "add_cast: buffers must be 16-byte aligned");
  if (n == 0) return MHR_OK;
  hipLaunchKernelGGL(add_cast_kernel, dim3(mhr_grid_for(n / 8, 256)), dim3(256), 0, (hipStream_t)stream, x, (const bf16_t*)y_bf16, out,
                     (bf16_t*)out_bf16, n / 8);
  MHR_CHECK_LAUNCH("add_cast");
  return MHR_OK;
}

// ------------------------------------------------------------------------------------------
// L2 row normalisation
// ------------------------------------------------------------------------------------------
template <typename XT, typename YT, int NC>
__global__ __launch_bounds__(256) void l2norm_kernel(const XT* __restrict__ x, YT* __restrict__ y, float* __restrict__ norms,
                                                     int64_t rows, int dim) {
  WAVE_ROW_LOOP(rows) {
    RowRegs<NC> r;
    load_row<XT, NC>(x + row * dim, dim, lane, r);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i)
#pragma unroll
      for (int k = 0; k < 4; ++k) s += r.v[i][k] * r.v[i][k];
    float nrm = sqrtf(wave_sum(s));
    float inv = 1.0f / nrm;
#pragma unroll
    for (int i = 0; i < NC; ++i)
#pragma unroll
      for (int k = 0; k < 4; ++k) r.v[i][k] = r.v[i][k] * inv;
    if (y) store_row<YT, NC>(y + row * dim, dim, lane, r);
    if (norms && lane == 0) norms[row] = nrm;
  }
}

// backward of y = x / |x|:  dx = (dy - n (n . dy)) / |x|,  n = x / |x|  (fp32; one wave per row)
template <int NC>
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                         const float* __restrict__ norms, float* __restrict__ dx, int64_t rows,
                                                         int dim) {
  WAVE_ROW_LOOP(rows) {
    RowRegs<NC> rx, rg;
    load_row<float, NC>(x + row * dim, dim, lane, rx);
    load_row<float, NC>(dy + row * dim, dim, lane, rg);
    const float inv = 1.0f / norms[row];
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        rx.v[i][k] *= inv;
        dot += rx.v[i][k] * rg.v[i][k];
      }
    dot = wave_sum(dot);
#pragma unroll
    for (int i = 0; i < NC; ++i)
#pragma unroll
      for (int k = 0; k < 4; ++k) rg.v[i][k] = (rg.v[i][k] - rx.v[i][k] * dot) * inv;
    store_row<float, NC>(dx + row * dim, dim, lane, rg);
  }
}

// The backward reading its rows THROUGH an id list: x[r] = table[ids[r]] - the negative pools of a training step are
// gathered AND normalised by mhr_embedding_gather_step (csrc/embedding.hip; reference hstu.py:670-672, 752-754) without an
// fp32 copy of the gathered rows, so the backward re-reads them from the table (no optimizer step lies between a step's
// forward and backward).  Same arithmetic in the same order as l2norm_bwd_kernel.  Ids outside [0, n_src) are clamped (the
// forward counted them).
__device__ __forceinline__ int64_t clamp_count_id(int64_t v, int64_t n_src, int lane, unsigned int* __restrict__ bad) {
  if (v < 0 || v >= n_src) {
    if (lane == 0 && bad) atomicAdd(bad, 1u);
    v = v < 0 ? 0 : n_src - 1;
  }
  return v;
}

template <int NC>
__global__ __launch_bounds__(256) void l2norm_indexed_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ table,
                                                                 int64_t n_src, const int64_t* __restrict__ ids,
                                                                 const float* __restrict__ norms, float* __restrict__ dx,
                                                                 int64_t rows, int dim) {
  WAVE_ROW_LOOP(rows) {
    const int64_t src = clamp_count_id(ids[row], n_src, lane, nullptr);
    RowRegs<NC> rx, rg;
    load_row<float, NC>(table + src * dim, dim, lane, rx);
    load_row<float, NC>(dy + row * dim, dim, lane, rg);
    const float inv = 1.0f / norms[row];
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        rx.v[i][k] *= inv;
        dot += rx.v[i][k] * rg.v[i][k];
      }
    dot = wave_sum(dot);
#pragma unroll
    for (int i = 0; i < NC; ++i)
#pragma unroll
      for (int k = 0; k < 4; ++k) rg.v[i][k] = (rg.v[i][k] - rx.v[i][k] * dot) * inv;
    store_row<float, NC>(dx + row * dim, dim, lane, rg);
  }
}

extern "C" int mhr_l2norm_rows_indexed_bwd(const float* dy, const float* table, int64_t n_src_rows, const int64_t* ids,
                                           const float* norms, float* dx, int64_t rows, int dim, void* stream) {
  MHR_REQUIRE(dy && table && ids && norms && dx, "l2norm_rows_indexed_bwd: null pointer");
  MHR_REQUIRE(dim > 0 && dim % 4 == 0 && dim <= 2048 && n_src_rows > 0, "l2norm_rows_indexed_bwd: dim=%d unsupported", dim);
  if (rows == 0) return MHR_OK;
  hipStream_t s = (hipStream_t)stream;
  int grid = mhr_grid_for(rows, 4);
#define LIB(NC) hipLaunchKernelGGL((l2norm_indexed_bwd_kernel<NC>), dim3(grid), dim3(256), 0, s, dy, table, n_src_rows, ids, norms, dx, rows, dim)
  DISPATCH_NC(dim, LIB);
#undef LIB
  MHR_CHECK_LAUNCH("l2norm_rows_indexed_bwd");
  return MHR_OK;
}

extern "C" int mhr_l2norm_rows_bwd(const float* dy, const float* x, const float* norms, float* dx, int64_t rows, int dim,
                                   void* stream) {
  MHR_REQUIRE(dy && x && norms && dx, "l2norm_rows_bwd: null pointer");
  MHR_REQUIRE(dim > 0 && dim % 4 == 0 && dim <= 2048, "l2norm_rows_bwd: dim=%d unsupported", dim);
  if (rows == 0) return MHR_OK;
  hipStream_t s = (hipStream_t)stream;
  int grid = mhr_grid_for(rows, 4);
#define LB(NC) hipLaunchKernelGGL((l2norm_bwd_kernel<NC>), dim3(grid), dim3(256), 0, s, dy, x, norms, dx, rows, dim)
  DISPATCH_NC(dim, LB);
#undef LB
  MHR_CHECK_LAUNCH("l2norm_rows_bwd");
  return MHR_OK;
}

extern "C" int mhr_l2norm_rows(const void* x, int x_dtype, void* y, int y_dtype, float* norms, int64_t rows, int dim,
                               void* stream) {
  MHR_REQUIRE(x && (y || norms), "l2norm_rows: null pointer");
  MHR_REQUIRE(dim > 0 && dim % 4 == 0 && dim <= 2048, "l2norm_rows: dim=%d unsupported", dim);
  if (rows <= 0) return MHR_OK;
  hipStream_t s = (hipStream_t)stream;
  int grid = mhr_grid_for(rows, 4);
  bool xb = x_dtype == MHR_BF16, yb = y_dtype == MHR_BF16;
#define L(NC)                                                                                                          \
  if (xb && yb) hipLaunchKernelGGL((l2norm_kernel<bf16_t, bf16_t, NC>), dim3(grid), dim3(256), 0, s, (const bf16_t*)x, \
                                   (bf16_t*)y, norms, rows, dim);                                                      \
  else if (xb) hipLaunchKernelGGL((l2norm_kernel<bf16_t, float, NC>), dim3(grid), dim3(256), 0, s, (const bf16_t*)x,   \
                                  (float*)y, norms, rows, dim);                                                        \
  else if (yb) hipLaunchKernelGGL((l2norm_kernel<float, bf16_t, NC>), dim3(grid), dim3(256), 0, s, (const float*)x,    \
                                  (bf16_t*)y, norms, rows, dim);                                                       \
  else hipLaunchKernelGGL((l2norm_kernel<float, float, NC>), dim3(grid), dim3(256), 0, s, (const float*)x, (float*)y,  \
                          norms, rows, dim)
  DISPATCH_NC(dim, L);
#undef L
  MHR_CHECK_LAUNCH("l2norm_rows");
  return MHR_OK;
}

// ------------------------------------------------------------------------------------------
// RMSNorm of the LLM decoder blocks (HLLM user / item towers), optionally fused with the residual add that precedes it.
// Reference: LlamaRMSNorm model/HLLM/modeling_llama.py:266-280 (statistics in fp32, `weight * x_hat`), used at
// modeling_llama.py:768, 783, 1108 around `residual + hidden_states` (779, 785); Baichuan's RMSNorm
// (baichuan/modeling_baichuan.py:110-133) is the same arithmetic.
//   fwd: x_out = x (+ res);  y = bf16(w * x_out * rsqrt(mean(x_out^2) + eps))
//   bwd: g = dy * w;  dx = rstd * (g - x_hat * mean(g * x_hat)) (+ d_xout);  dw partial = sum_rows dy * x_hat
// The weight gradient is reduced deterministically: every wave keeps a private partial over the rows it visits, the four
// waves of a workgroup are summed through LDS, and one [grid, dim] block of partials goes to the caller's fp32 sum.
// ------------------------------------------------------------------------------------------
static inline int nc_for_rms(int dim) { return dim <= 2048 ? nc_for(dim) : 16; }
#define DISPATCH_NC_RMS(dim, MACRO) \
  switch (nc_for_rms(dim)) {        \
    case 1: MACRO(1); break;        \
    case 2: MACRO(2); break;        \
    case 4: MACRO(4); break;        \
    case 8: MACRO(8); break;        \
    default: MACRO(16); break;      \
  }

template <int NC>
__global__ __launch_bounds__(256) void rmsnorm_fwd_kernel(const float* __restrict__ x, const bf16_t* __restrict__ res,
                                                          const float* __restrict__ w, float* __restrict__ x_out,
                                                          bf16_t* __restrict__ y, float* __restrict__ rstd_o, int64_t rows,
                                                          int dim, float eps) {
  RowRegs<NC> wv;
  load_row<float, NC>(w, dim, threadIdx.x & 63, wv);
  WAVE_ROW_LOOP(rows) {
    RowRegs<NC> r;
    load_row<float, NC>(x + row * dim, dim, lane, r);
    if (res) {
      RowRegs<NC> b;
      load_row<bf16_t, NC>(res + row * dim, dim, lane, b);
#pragma unroll
      for (int i = 0; i < NC; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) r.v[i][k] += b.v[i][k];
      store_row<float, NC>(x_out + row * dim, dim, lane, r);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i)
#pragma unroll
      for (int k = 0; k < 4; ++k) s += r.v[i][k] * r.v[i][k];       // columns >= dim hold zeros
    const float rstd = rsqrtf(wave_sum(s) / (float)dim + eps);
#pragma unroll
    for (int i = 0; i < NC; ++i)
#pragma unroll
      for (int k = 0; k < 4; ++k) r.v[i][k] = wv.v[i][k] * (r.v[i][k] * rstd);
    store_row<bf16_t, NC>(y + row * dim, dim, lane, r);
    if (lane == 0) rstd_o[row] = rstd;
  }
}

template <int NC>
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(const bf16_t* __restrict__ dy, const float* __restrict__ x_out,
                                                          const float* __restrict__ w, const float* __restrict__ rstd_i,
                                                          const float* __restrict__ d_xout, float* __restrict__ dx,
                                                          bf16_t* __restrict__ dres, float* __restrict__ dw_part, int64_t rows,
                                                          int dim) {
  extern __shared__ float red[];                   // [3][dim] partials of waves 1..3
  RowRegs<NC> wv, acc;
  load_row<float, NC>(w, dim, threadIdx.x & 63, wv);
#pragma unroll
  for (int i = 0; i < NC; ++i) acc.v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  WAVE_ROW_LOOP(rows) {
    RowRegs<NC> g, xv;
    load_row<bf16_t, NC>(dy + row * dim, dim, lane, g);
    load_row<float, NC>(x_out + row * dim, dim, lane, xv);
    const float rstd = rstd_i[row];
    float s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float xh = xv.v[i][k] * rstd;
        acc.v[i][k] += g.v[i][k] * xh;
        g.v[i][k] *= wv.v[i][k];
        xv.v[i][k] = xh;
        s2 += g.v[i][k] * xh;
      }
    s2 = wave_sum(s2) / (float)dim;
    RowRegs<NC> o;
    if (d_xout) load_row<float, NC>(d_xout + row * dim, dim, lane, o);
#pragma unroll
    for (int i = 0; i < NC; ++i)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float d = rstd * (g.v[i][k] - xv.v[i][k] * s2);
        o.v[i][k] = d_xout ? o.v[i][k] + d : d;
      }
    store_row<float, NC>(dx + row * dim, dim, lane, o);
    if (dres) store_row<bf16_t, NC>(dres + row * dim, dim, lane, o);
  }
  const int wv_id = threadIdx.x >> 6, ln = threadIdx.x & 63;
  if (wv_id > 0) store_row<float, NC>(red + (wv_id - 1) * dim, dim, ln, acc);
  __syncthreads();
  if (wv_id == 0) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      RowRegs<NC> t;
      load_row<float, NC>(red + j * dim, dim, ln, t);
#pragma unroll
      for (int i = 0; i < NC; ++i) acc.v[i] += t.v[i];
    }
    store_row<float, NC>(dw_part + (int64_t)blockIdx.x * dim, dim, ln, acc);
  }
}

extern "C" int mhr_rmsnorm_fwd(const float* x, const void* res_bf16, const float* weight, float* x_out, void* y_bf16,
                               float* rstd, int64_t rows, int dim, float eps, void* stream) {
  MHR_REQUIRE(x && weight && y_bf16 && rstd, "rmsnorm_fwd: null pointer");
  MHR_REQUIRE(!res_bf16 || x_out, "rmsnorm_fwd: x_out is required with a residual branch");
  MHR_REQUIRE(dim > 0 && dim % 4 == 0 && dim <= 4096, "rmsnorm_fwd: dim=%d unsupported (multiple of 4, <= 4096)", dim);
  if (rows <= 0) return MHR_OK;
  const int grid = mhr_grid_for(rows, 4);
#define L(NC)                                                                                                          \
  hipLaunchKernelGGL((rmsnorm_fwd_kernel<NC>), dim3(grid), dim3(256), 0, (hipStream_t)stream, x, (const bf16_t*)res_bf16, \
                     weight, x_out, (bf16_t*)y_bf16, rstd, rows, dim, eps)
  DISPATCH_NC_RMS(dim, L);
#undef L
  MHR_CHECK_LAUNCH("rmsnorm_fwd");
  return MHR_OK;
}

extern "C" int mhr_rmsnorm_bwd_parts(int64_t rows) { return mhr_grid_for(rows, 16, 1024); }

extern "C" int mhr_rmsnorm_bwd(const void* dy_bf16, const float* x_out, const float* weight, const float* rstd,
                               const float* d_xout, float* dx, void* dres_bf16, float* dw_part, int64_t rows, int dim,
                               void* stream) {
  MHR_REQUIRE(dy_bf16 && x_out && weight && rstd && dx && dw_part, "rmsnorm_bwd: null pointer");
  MHR_REQUIRE(dim > 0 && dim % 4 == 0 && dim <= 4096, "rmsnorm_bwd: dim=%d unsupported (multiple of 4, <= 4096)", dim);
  MHR_REQUIRE(rows > 0, "rmsnorm_bwd: no rows");
  const int grid = mhr_rmsnorm_bwd_parts(rows);               // dw_part is [grid, dim]
  const size_t lds = (size_t)3 * dim * sizeof(float);
#define L(NC)                                                                                                            \
  hipLaunchKernelGGL((rmsnorm_bwd_kernel<NC>), dim3(grid), dim3(256), lds, (hipStream_t)stream, (const bf16_t*)dy_bf16,  \
                     x_out, weight, rstd, d_xout, dx, (bf16_t*)dres_bf16, dw_part, rows, dim)
  DISPATCH_NC_RMS(dim, L);
#undef L
  MHR_CHECK_LAUNCH("rmsnorm_bwd");
  return MHR_OK;
}
