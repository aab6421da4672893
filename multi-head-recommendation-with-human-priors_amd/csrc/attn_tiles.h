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
                                            bool do_silu, bf16_t* act, int64_t act_stride, int m_begin = 0) {
  using T = sg::Tile<NKS>;   // (m_begin: rows below it are not staged - the leading all-padding blocks of a sequence, see leading_dead_blocks)
  // four 16-byte chunks per thread in flight: all loads of a batch are issued before the first SiLU / LDS store (one chunk per
  // iteration left every staging at 3-4 dependent global round trips, and a backward workgroup stages four operands)
  constexpr int U = 4;
  const int total = Lp * T::CH, nt = blockDim.x;
  for (int c0 = m_begin * T::CH + threadIdx.x; c0 < total; c0 += U * nt) {
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

// Two operands staged together: the loads of BOTH are in flight before the first LDS store (one memory round trip for the
// pair instead of two; a backward workgroup stages two pairs).
template <int NKS>
__device__ __forceinline__ void stage_tiles2(unsigned char* dst_a, const bf16_t* src_a, int64_t stride_a, bool silu_a,
                                             unsigned char* dst_b, const bf16_t* src_b, int64_t stride_b, bool silu_b,
                                             int L, int Lp, int hd, int m_begin = 0) {
  using T = sg::Tile<NKS>;
  constexpr int U = 4;
  const int total = Lp * T::CH, nt = blockDim.x;
  for (int c0 = m_begin * T::CH + threadIdx.x; c0 < total; c0 += U * nt) {
    bf16x8 va[U], vb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int c = c0 + u * nt;
      const int m = c / T::CH, j = c % T::CH;
      va[u] = zero8();
      vb[u] = zero8();
      if (c < total && m < L && j * 8 < hd) {
        va[u] = *reinterpret_cast<const bf16x8*>(src_a + (int64_t)m * stride_a + j * 8);
        vb[u] = *reinterpret_cast<const bf16x8*>(src_b + (int64_t)m * stride_b + j * 8);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int c = c0 + u * nt;
      if (c >= total) break;
      const int m = c / T::CH, j = c % T::CH;
      const bool in = m < L && j * 8 < hd;
      if (in && silu_a) va[u] = silu8(va[u]);
      if (in && silu_b) vb[u] = silu8(vb[u]);
      const int o = (m >> 5) * T::BYTES + T::off(m & 31, j);
      *reinterpret_cast<bf16x8*>(dst_a + o) = va[u];
      *reinterpret_cast<bf16x8*>(dst_b + o) = vb[u];
    }
  }
}

// Four operands at once (the backward's resident form): sixteen 16-byte loads per thread in flight before the first store.
template <int NKS>
__device__ __forceinline__ void stage_tiles4(unsigned char* d0, const bf16_t* s0, int64_t st0, bool silu0, unsigned char* d1,
                                             const bf16_t* s1, int64_t st1, bool silu1, unsigned char* d2, const bf16_t* s2,
                                             int64_t st2, bool silu2, unsigned char* d3, const bf16_t* s3, int64_t st3,
                                             bool silu3, int L, int Lp, int hd, int m_begin = 0) {
  using T = sg::Tile<NKS>;
  constexpr int U = 4;
  const int total = Lp * T::CH, nt = blockDim.x;
  unsigned char* dst[4] = {d0, d1, d2, d3};
  const bf16_t* src[4] = {s0, s1, s2, s3};
  const int64_t st[4] = {st0, st1, st2, st3};
  const bool sl[4] = {silu0, silu1, silu2, silu3};
  for (int c0 = m_begin * T::CH + threadIdx.x; c0 < total; c0 += U * nt) {
    bf16x8 v[4][U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int c = c0 + u * nt;
      const int m = c / T::CH, j = c % T::CH;
      const bool in = c < total && m < L && j * 8 < hd;
#pragma unroll
      for (int o = 0; o < 4; ++o) v[o][u] = in ? *reinterpret_cast<const bf16x8*>(src[o] + (int64_t)m * st[o] + j * 8) : zero8();
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int c = c0 + u * nt;
      if (c >= total) break;
      const int m = c / T::CH, j = c % T::CH;
      const bool in = m < L && j * 8 < hd;
      const int off = (m >> 5) * T::BYTES + T::off(m & 31, j);
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        if (in && sl[o]) v[o][u] = silu8(v[o][u]);
        *reinterpret_cast<bf16x8*>(dst[o] + off) = v[o][u];
      }
    }
  }
}

// Gradient tile of one wave (32 rows x 32 features: accumulator rows on the registers, features on the lanes) -> global
// rows, through a wave-private LDS scratch so that memory sees 16-byte accesses: the accumulator layout gives a lane ONE
// bf16 of 16 different rows - stored directly that is sixteen 2-byte stores and, with the SiLU' chain, sixteen dependent
// 2-byte loads per tile, each behind its own s_waitcnt (measured: 32 k of the 37 k cycles of the dK / dV pass at cfg1).
// Here a lane takes 8 consecutive features of one row: one 16-byte load of the pre-activations (fetched before the tile-pair
// loop, so their latency is hidden), eight SiLU' factors, one 16-byte store.
// scratch: 32 x GS floats, private to the wave (program order suffices between its writes and reads).
constexpr int GS = 36;      // row stride in floats: 16-byte aligned rows, and the 8-float reads of a lane group spread over the banks

// Lane -> (row, 8-feature chunk) map of the row-wise pass: lane (r, half) takes row r, chunks half and 2 + half - the same
// rows and features its MFMA row fragments cover, and conflict-free on the GS-float scratch rows.
struct GradPre {
  bf16x8 v[2];
};
// pre-activation values of a gradient tile's rows, fetched EARLY (before the tile-pair loop): the loop hides their latency
__device__ __forceinline__ GradPre prefetch_pre(const bf16_t* pre, int64_t pre_stride, int row_base, int n_rows, int col_base,
                                                int n_cols, bool chain, int lane) {
  GradPre p;
  const int row = row_base + (lane & 31), half = lane >> 5;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int f0 = col_base + 8 * (2 * u + half);
    p.v[u] = zero8();
    if (chain && row < n_rows && f0 < n_cols) p.v[u] = *reinterpret_cast<const bf16x8*>(pre + (int64_t)row * pre_stride + f0);
  }
  return p;
}

__device__ __forceinline__ void store_grad_tile(float* scratch, const f32x16& acc, float scale, bf16_t* out, int64_t out_stride,
                                                const GradPre& pre, int row_base, int n_rows, int col_base, int n_cols,
                                                bool chain, int lane) {
  const int r = lane & 31, half = lane >> 5;
#pragma unroll
  for (int g = 0; g < 16; ++g) scratch[crow(g, half) * GS + r] = acc[g] * scale;
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int ch = 2 * u + half, f0 = col_base + 8 * ch;
    const f32x4 lo = *reinterpret_cast<const f32x4*>(scratch + r * GS + 8 * ch);
    const f32x4 hi = *reinterpret_cast<const f32x4*>(scratch + r * GS + 8 * ch + 4);
    bf16x8 w;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float gv = j < 4 ? lo[j] : hi[j - 4];
      if (chain) gv *= dsilu_f((float)pre.v[u][j]);
      w[j] = (bf16_t)gv;
    }
    if (row_base + r < n_rows && f0 < n_cols) *reinterpret_cast<bf16x8*>(out + (int64_t)(row_base + r) * out_stride + f0) = w;
  }
  __builtin_amdgcn_wave_barrier();
}

// Leading 32-row blocks of a sequence that hold no valid key (mhr_attn_seq_layout computes them once per batch).  With front
// padding (evalset.py:34-41, trainset.py:111-137: the valid items sit at the END of the window) those blocks take part in nothing:
// as keys they are masked, as queries every key at or before them is masked, so their outputs and gradients are exactly zero - the
// kernels skip staging, tile pairs and the gradient epilogues of them and write the zeros directly (measured before: a backward
// over sequences with 32 valid keys of 200 took 70 % of the time of full sequences).

// zero rows [0, n_rows) of a head's slice (hd columns, a multiple of 8) of a [*, stride] bf16 matrix: 16-byte stores
__device__ __forceinline__ void zero_head_rows(bf16_t* out, int64_t stride, int n_rows, int hd) {
  const int ch = hd >> 3;
  for (int c = threadIdx.x; c < n_rows * ch; c += blockDim.x) {
    const int m = c / ch, j = c - m * ch;
    *reinterpret_cast<bf16x8*>(out + (int64_t)m * stride + j * 8) = zero8();
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
