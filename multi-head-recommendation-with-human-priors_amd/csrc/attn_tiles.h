// Shared pieces of the LDS-resident attention kernels (attention.hip: HSTU pointwise gate; softmax_attn.hip: LLM decoder):
// operand staging as swizzled 32-row tile images, register fragments, key-validity words, head_dim dispatch.
#pragma once
#include "mhr_common.h"
#include "stream_gemm.h"

namespace attn {

using sg::crow;
using sg::zero16;
using sg::zero8;

constexpr float LOG2E_F = 1.4426950408889634f;

// sigmoid with one v_exp and one v_rcp (the division form costs a Newton step per element; every score tile applies it
// to 16 values per lane, which is where these kernels spend their VALU time)
__device__ __forceinline__ float fast_sigmoid(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-x * LOG2E_F));
}

__device__ __forceinline__ bf16x8 silu8(bf16x8 x) {
  bf16x8 y;
#pragma unroll
  for (int i = 0; i < 8; ++i) y[i] = (bf16_t)silu_f((float)x[i]);
  return y;
}

// 8 consecutive elements of row `row` at column `koff` of a [rows, stride] bf16 matrix (zeros outside).
__device__ __forceinline__ bf16x8 load_frag(const bf16_t* base, int64_t stride, int row, int n_rows, int koff, int n_cols) {
  if (row < n_rows && koff < n_cols) return *reinterpret_cast<const bf16x8*>(base + (int64_t)row * stride + koff);
  return zero8();
}

__device__ __forceinline__ void pack_acc(const f32x16& x, bf16x8& f0, bf16x8& f1) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    f0[j] = (bf16_t)x[j];
    f1[j] = (bf16_t)x[8 + j];
  }
}

// Stage a [L, hd] row-major global block as ceil(L/32) swizzled 32-row tile images (stream_gemm.h Tile<NKS>, feature
// dim padded to 16*NKS with zeros, rows >= L zero).  One image serves BOTH access shapes the kernels need: row
// fragments (ds_read_b128: operand rows on the lanes, features on the vector) and transposed fragments
// (ds_read_b64_tr_b16: operand summed over the ROW index) - so no transposed copy is ever built.
template <int NKS>
__device__ __forceinline__ void stage_tiles(unsigned char* dst, const bf16_t* src, int64_t stride, int L, int Lp, int hd,
                                            bool do_silu, bf16_t* act, int64_t act_stride) {
  using T = sg::Tile<NKS>;
  // four 16-byte chunks per thread in flight: all loads of a batch are issued before the first SiLU / LDS store (one chunk per
  // iteration left every staging at 3-4 dependent global round trips, and a backward workgroup stages four operands)
  constexpr int U = 4;
  const int total = Lp * T::CH, nt = blockDim.x;
  for (int c0 = threadIdx.x; c0 < total; c0 += U * nt) {
    bf16x8 val[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int c = c0 + u * nt;
      const int m = c / T::CH, j = c % T::CH;
      val[u] = zero8();
      if (c < total && m < L && j * 8 < hd) val[u] = *reinterpret_cast<const bf16x8*>(src + (int64_t)m * stride + j * 8);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int c = c0 + u * nt;
      if (c >= total) break;
      const int m = c / T::CH, j = c % T::CH;
      const bool in = m < L && j * 8 < hd;
      if (in && do_silu) val[u] = silu8(val[u]);
      if (in && act) *reinterpret_cast<bf16x8*>(act + (int64_t)m * act_stride + j * 8) = val[u];
      *reinterpret_cast<bf16x8*>(dst + (m >> 5) * T::BYTES + T::off(m & 31, j)) = val[u];
    }
  }
}

__device__ __forceinline__ void build_valid_mask(uint32_t* vmask, const uint8_t* kv, int L, int nb) {
  if ((int)threadIdx.x < nb) {
    uint32_t bits = 0;
    for (int i = 0; i < 32; ++i) {
      int m = threadIdx.x * 32 + i;
      if (m < L && (!kv || kv[m])) bits |= 1u << i;      // kv == nullptr: every key of the sequence is valid
    }
    vmask[threadIdx.x] = bits;
  }
}

struct AttnShape {
  int nks, nd;
};
static inline bool attn_shape(int hd, AttnShape& s) {
  if (hd <= 0 || hd % 8 != 0 || hd > 128) return false;
  if (hd <= 16) s = {1, 1};
  else if (hd <= 32) s = {2, 1};
  else if (hd <= 64) s = {4, 2};
  else s = {8, 4};
  return true;
}

}  // namespace attn

#define ATTN_DISPATCH(shape, MACRO)                  \
  if (shape.nks == 1) { MACRO(1, 1); }               \
  else if (shape.nks == 2) { MACRO(2, 1); }          \
  else if (shape.nks == 4) { MACRO(4, 2); }          \
  else { MACRO(8, 4); }
