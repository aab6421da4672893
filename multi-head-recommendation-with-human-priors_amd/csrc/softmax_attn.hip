// Causal softmax attention of the LLM decoder blocks (HLLM user decoder / item towers), forward and backward, gfx950.
//
//   out[t, h, :] = sum_{m <= t, valid[m]} softmax_m(q[t,h].k[m,g(h)] * scale) v[m, g(h), :]      g(h) = h / (n_heads / n_kv_heads)
//
// Reference (file:line under code/REC/model/HLLM/): the eager path modeling_llama.py:648-682 (repeat_kv, QK^T/sqrt(hd) +
// additive causal & key-padding mask, fp32 softmax, PV) and the flash path 683-705 -> flash_self_attn.py:61-130
// (padded batch with a key mask, or ONE packed row of concatenated sequences described by `cu_input_lens`); Baichuan
// baichuan/modeling_baichuan.py:246-338 is the same computation.  Both shapes are served here: a sequence is the row range
// [cu[b], cu[b+1]) of the token matrix (or [b*L, (b+1)*L) without `cu`), optionally with a key-valid byte per token.
//
// Same skeleton as attention.hip: one workgroup per (sequence, query head); K and V of the matching KV head staged once
// in LDS as swizzled 32-row tile images; 32x32 score tiles on v_mfma_f32_32x32x16_bf16, computed TRANSPOSED (keys on the
// accumulator rows, queries on the lanes) so that the online-softmax statistics of a query are per-lane scalars (one
// cross-half exchange per tile) and the probability tile, rounded to bf16 in registers, is directly the B operand of
// O^T += V^T P^T.  The [L, L] score matrix never exists in memory; the forward keeps only lse[t, h].
// The backward recomputes P from lse in two passes (dK/dV per key block with Q and dO resident, then dQ per query block
// with K and V resident): no atomics, bitwise reproducible.  With grouped KV heads every QUERY head writes its own dK/dV
// slab and the caller sums the group (a [T, group, hd] reduction) - keys are shared, gradients are not.
// Query rows with no admissible key (front padding) produce zeros and lse = 0 (the reference's eager path yields an
// arbitrary finite average there, its flash path never computes them; nothing downstream reads those rows).
#include "mhr_common.h"
#include "stream_gemm.h"
#include "attn_tiles.h"

namespace {

using namespace attn;

struct SeqGeom {
  int64_t row0;
  int len;
};
__device__ __forceinline__ SeqGeom seq_geom(const int32_t* cu, int b, int L) {
  if (cu) return {(int64_t)cu[b], min(cu[b + 1] - cu[b], L)};
  return {(int64_t)b * L, L};
}

template <int NKS, int ND>
__global__ __launch_bounds__(256) void softmax_attn_fwd_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                               const bf16_t* __restrict__ v, int64_t stride,
                                                               const int32_t* __restrict__ cu, const uint8_t* __restrict__ key_valid,
                                                               bf16_t* __restrict__ out, int64_t out_stride, float* __restrict__ lse,
                                                               int L_max, int n_heads, int group, int hd, float scale) {
  using T = sg::Tile<NKS>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int nb_max = (L_max + 31) >> 5;
  unsigned char* Kt = smem;
  unsigned char* Vt = smem + nb_max * T::BYTES;
  uint32_t* vmask = reinterpret_cast<uint32_t*>(Vt + nb_max * T::BYTES);

  const int b = blockIdx.x / n_heads, head = blockIdx.x % n_heads, kvh = head / group;
  const SeqGeom sq = seq_geom(cu, b, L_max);
  const int L = sq.len, Lp = (L + 31) & ~31, nb = Lp >> 5;
  if (L <= 0) return;
  const int64_t row0 = sq.row0;
  const bf16_t* qp = q + row0 * stride + head * hd;
  const bf16_t* kp = k + row0 * stride + kvh * hd;
  const bf16_t* vp = v + row0 * stride + kvh * hd;

  stage_tiles<NKS>(Kt, kp, stride, L, Lp, hd, false, nullptr, 0);
  stage_tiles<NKS>(Vt, vp, stride, L, Lp, hd, false, nullptr, 0);
  build_valid_mask(vmask, key_valid ? key_valid + row0 : nullptr, L, nb);
  __syncthreads();

  const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6, lane = threadIdx.x & 63, r = lane & 31, half = lane >> 5;
  const float c2 = scale * LOG2E_F;
  sg::LaneAddr<NKS> la;
  la.init(lane);
  for (int it = 0; it * nw < nb; ++it) {
    const int qb = nb - 1 - ((it & 1) ? it * nw + (nw - 1 - wave) : it * nw + wave);   // snake over the causal triangle from its heavy end
    if (qb < 0) continue;
    const int qrow = qb * 32 + r;
    bf16x8 qf[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) qf[ks] = load_frag(qp, stride, qrow, L, ks * 16 + 8 * half, hd);
    f32x16 o[ND];
#pragma unroll
    for (int dc = 0; dc < ND; ++dc) o[dc] = zero16();
    float m_run = -INFINITY, l_run = 0.f;

    for (int kb = 0; kb <= qb; ++kb) {
      const uint32_t vm = vmask[kb];
      if (vm == 0) continue;
      const unsigned char* kt = Kt + kb * T::BYTES;
      const unsigned char* vt = Vt + kb * T::BYTES;
      f32x16 s = zero16();
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks)
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(la.read_a(kt, ks), qf[ks], s, 0, 0, 0);   // S^T: rows = keys, cols = queries
      float tmax = -INFINITY;
      if (kb < qb && vm == 0xFFFFFFFFu) {
#pragma unroll
        for (int g = 0; g < 16; ++g) tmax = fmaxf(tmax, s[g]);
      } else {
#pragma unroll
        for (int g = 0; g < 16; ++g) {
          const int kl = crow(g, half);
          const bool ok = ((vm >> kl) & 1u) && (kb * 32 + kl <= qrow);
          s[g] = ok ? s[g] : -INFINITY;
          tmax = fmaxf(tmax, s[g]);
        }
      }
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));              // both halves hold rows of the same query
      const float m_new = fmaxf(m_run, tmax);
      const float m_use = m_new == -INFINITY ? 0.f : m_new;
      const float corr = __builtin_amdgcn_exp2f((m_run - m_use) * c2);
      l_run *= corr;
#pragma unroll
      for (int dc = 0; dc < ND; ++dc)
#pragma unroll
        for (int g = 0; g < 16; ++g) o[dc][g] *= corr;
      const float mc = m_use * c2;
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const float p = __builtin_amdgcn_exp2f(s[g] * c2 - mc);
        l_run += p;
        s[g] = p;
      }
      m_run = m_new;
      bf16x8 p0, p1;
      pack_acc(s, p0, p1);
#pragma unroll
      for (int dc = 0; dc < ND; ++dc) {
        o[dc] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(la.read_tr(vt, dc, 0), p0, o[dc], 0, 0, 0);   // O^T += V^T . P^T
        o[dc] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(la.read_tr(vt, dc, 1), p1, o[dc], 0, 0, 0);
      }
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv_l = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    if (qrow < L) {
      bf16_t* orow = out + (row0 + qrow) * out_stride + head * hd;
#pragma unroll
      for (int dc = 0; dc < ND; ++dc)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int d0 = dc * 32 + 8 * g4 + 4 * half;
          if (d0 < hd) {
            bf16x4 w = {(bf16_t)(o[dc][4 * g4] * inv_l), (bf16_t)(o[dc][4 * g4 + 1] * inv_l), (bf16_t)(o[dc][4 * g4 + 2] * inv_l),
                        (bf16_t)(o[dc][4 * g4 + 3] * inv_l)};
            *reinterpret_cast<bf16x4*>(orow + d0) = w;
          }
        }
      if (half == 0) lse[(row0 + qrow) * n_heads + head] = l_tot > 0.f ? m_run * scale + __logf(l_tot) : 0.f;
    }
  }
}

template <int NKS, int ND>
__global__ __launch_bounds__(256) void softmax_attn_bwd_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, int64_t stride,
    const int32_t* __restrict__ cu, const uint8_t* __restrict__ key_valid, const bf16_t* __restrict__ out,
    const bf16_t* __restrict__ d_out, int64_t o_stride, const float* __restrict__ lse, bf16_t* __restrict__ dq, int64_t dq_stride,
    bf16_t* __restrict__ dk, bf16_t* __restrict__ dv, int64_t dkv_stride, int L_max, int n_heads, int group, int hd, float scale) {
  using T = sg::Tile<NKS>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int nb_max = (L_max + 31) >> 5;
  unsigned char* T0 = smem;                          // Q tiles in pass A, K tiles in pass B
  unsigned char* T1 = smem + nb_max * T::BYTES;      // dO tiles in pass A, V tiles in pass B
  uint32_t* vmask = reinterpret_cast<uint32_t*>(T1 + nb_max * T::BYTES);
  float* lse_s = reinterpret_cast<float*>(vmask + nb_max);          // log2-domain lse per query (0 on padded rows)
  float* dlt_s = lse_s + nb_max * 32;                                // delta = rowsum(dO * O)

  const int b = blockIdx.x / n_heads, head = blockIdx.x % n_heads, kvh = head / group;
  const SeqGeom sq = seq_geom(cu, b, L_max);
  const int L = sq.len, Lp = (L + 31) & ~31, nb = Lp >> 5;
  if (L <= 0) return;
  const int64_t row0 = sq.row0;
  const bf16_t* qp = q + row0 * stride + head * hd;
  const bf16_t* kp = k + row0 * stride + kvh * hd;
  const bf16_t* vp = v + row0 * stride + kvh * hd;
  const bf16_t* op = out + row0 * o_stride + head * hd;
  const bf16_t* dop = d_out + row0 * o_stride + head * hd;

  stage_tiles<NKS>(T0, qp, stride, L, Lp, hd, false, nullptr, 0);
  stage_tiles<NKS>(T1, dop, o_stride, L, Lp, hd, false, nullptr, 0);
  build_valid_mask(vmask, key_valid ? key_valid + row0 : nullptr, L, nb);
  for (int t = threadIdx.x; t < Lp; t += blockDim.x) {
    float d = 0.f, l2 = 0.f;
    if (t < L) {
      for (int j = 0; j < hd; j += 8) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(op + (int64_t)t * o_stride + j);
        const bf16x8 g = *reinterpret_cast<const bf16x8*>(dop + (int64_t)t * o_stride + j);
#pragma unroll
        for (int e = 0; e < 8; ++e) d += (float)a[e] * (float)g[e];
      }
      l2 = lse[(row0 + t) * n_heads + head] * LOG2E_F;
    }
    lse_s[t] = l2;
    dlt_s[t] = d;
  }
  __syncthreads();

  const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6, lane = threadIdx.x & 63, r = lane & 31, half = lane >> 5;
  const float c2 = scale * LOG2E_F;
  sg::LaneAddr<NKS> la;
  la.init(lane);

  // ---- pass A: dK, dV of key block kb (keys on the lanes) ------------------------------------------
  for (int it = 0; it * nw < nb; ++it) {
    const int kb = (it & 1) ? it * nw + wave : it * nw + (nw - 1 - wave);     // early key blocks are the heavy ones
    if (kb >= nb) continue;
    const int key = kb * 32 + r;
    const uint32_t vm_kb = vmask[kb];
    const bool kvalid = (vm_kb >> r) & 1u;
    bf16x8 kf[NKS], vf[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      kf[ks] = load_frag(kp, stride, key, L, ks * 16 + 8 * half, hd);
      vf[ks] = load_frag(vp, stride, key, L, ks * 16 + 8 * half, hd);
    }
    f32x16 dvacc[ND], dkacc[ND];
#pragma unroll
    for (int dc = 0; dc < ND; ++dc) {
      dvacc[dc] = zero16();
      dkacc[dc] = zero16();
    }
    for (int qb = kb; qb < nb && vm_kb != 0; ++qb) {
      const unsigned char* qt = T0 + qb * T::BYTES;
      const unsigned char* dot = T1 + qb * T::BYTES;
      f32x16 s = zero16(), dp = zero16();
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(la.read_a(qt, ks), kf[ks], s, 0, 0, 0);      // S: rows = queries, cols = keys
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(la.read_a(dot, ks), vf[ks], dp, 0, 0, 0);   // dP = dO . V^T
      }
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const int qi = qb * 32 + crow(g, half);
        const bool ok = kvalid && key <= qi && qi < L;
        const float p = ok ? __builtin_amdgcn_exp2f(s[g] * c2 - lse_s[qi]) : 0.f;
        s[g] = p;                                         // P
        dp[g] = p * (dp[g] - dlt_s[qi]);                  // dS / scale
      }
      bf16x8 pa0, pa1, da0, da1;
      pack_acc(s, pa0, pa1);
      pack_acc(dp, da0, da1);
#pragma unroll
      for (int dc = 0; dc < ND; ++dc) {
        dvacc[dc] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa0, la.read_tr(dot, dc, 0), dvacc[dc], 0, 0, 0);   // dV += P^T . dO
        dvacc[dc] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa1, la.read_tr(dot, dc, 1), dvacc[dc], 0, 0, 0);
        dkacc[dc] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da0, la.read_tr(qt, dc, 0), dkacc[dc], 0, 0, 0);    // dK += dS^T . Q
        dkacc[dc] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da1, la.read_tr(qt, dc, 1), dkacc[dc], 0, 0, 0);
      }
    }
    // results: rows (regs) = keys, cols (lanes) = feature
#pragma unroll
    for (int dc = 0; dc < ND; ++dc) {
      const int d = dc * 32 + r;
      if (d < hd) {
#pragma unroll
        for (int g = 0; g < 16; ++g) {
          const int m = kb * 32 + crow(g, half);
          if (m < L) {
            dv[(row0 + m) * dkv_stride + head * hd + d] = (bf16_t)dvacc[dc][g];
            dk[(row0 + m) * dkv_stride + head * hd + d] = (bf16_t)(dkacc[dc][g] * scale);
          }
        }
      }
    }
  }

  // ---- pass B: dQ of query block qb (queries on the lanes) -----------------------------------------
  __syncthreads();                      // everyone is done reading the Q / dO tiles
  stage_tiles<NKS>(T0, kp, stride, L, Lp, hd, false, nullptr, 0);
  stage_tiles<NKS>(T1, vp, stride, L, Lp, hd, false, nullptr, 0);
  __syncthreads();
  for (int it = 0; it * nw < nb; ++it) {
    const int qb = nb - 1 - ((it & 1) ? it * nw + (nw - 1 - wave) : it * nw + wave);
    if (qb < 0) continue;
    const int qcol = qb * 32 + r;
    bf16x8 qf[NKS], dof[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      qf[ks] = load_frag(qp, stride, qcol, L, ks * 16 + 8 * half, hd);
      dof[ks] = load_frag(dop, o_stride, qcol, L, ks * 16 + 8 * half, hd);
    }
    const float lse_q = lse_s[qcol], dlt_q = dlt_s[qcol];
    f32x16 dqacc[ND];
#pragma unroll
    for (int dc = 0; dc < ND; ++dc) dqacc[dc] = zero16();
    for (int kb = 0; kb <= qb; ++kb) {
      const uint32_t vm = vmask[kb];
      if (vm == 0) continue;
      const unsigned char* kt = T0 + kb * T::BYTES;
      const unsigned char* vt = T1 + kb * T::BYTES;
      f32x16 s = zero16(), dp = zero16();
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(la.read_a(kt, ks), qf[ks], s, 0, 0, 0);      // S^T: rows = keys, cols = queries
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(la.read_a(vt, ks), dof[ks], dp, 0, 0, 0);   // dP^T = V . dO^T
      }
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const int kl = crow(g, half);
        const bool ok = ((vm >> kl) & 1u) && (kb * 32 + kl <= qcol) && qcol < L;
        const float p = ok ? __builtin_amdgcn_exp2f(s[g] * c2 - lse_q) : 0.f;
        dp[g] = p * (dp[g] - dlt_q);                                                 // dS^T / scale
      }
      bf16x8 a0, a1;
      pack_acc(dp, a0, a1);
#pragma unroll
      for (int dc = 0; dc < ND; ++dc) {
        dqacc[dc] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, la.read_tr(kt, dc, 0), dqacc[dc], 0, 0, 0);   // dQ += dS . K
        dqacc[dc] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, la.read_tr(kt, dc, 1), dqacc[dc], 0, 0, 0);
      }
    }
#pragma unroll
    for (int dc = 0; dc < ND; ++dc) {
      const int d = dc * 32 + r;
      if (d < hd) {
#pragma unroll
        for (int g = 0; g < 16; ++g) {
          const int n = qb * 32 + crow(g, half);
          if (n < L) dq[(row0 + n) * dq_stride + head * hd + d] = (bf16_t)(dqacc[dc][g] * scale);
        }
      }
    }
  }
}

inline size_t softmax_attn_lds(int L, int nks, bool bwd) {
  const int nb = (L + 31) / 32;
  return (size_t)2 * nb * (32 * nks * 32) + (size_t)nb * 4 + (bwd ? (size_t)2 * nb * 32 * 4 : 0) + 16;
}

}  // namespace

extern "C" int mhr_softmax_attn_fwd(const void* q, const void* k, const void* v, int64_t row_stride, const int32_t* cu_seqlens,
                                    const uint8_t* key_valid, void* out, float* lse, int n_seqs, int max_len, int n_heads,
                                    int n_kv_heads, int head_dim, float scale, void* stream) {
  MHR_REQUIRE(q && k && v && out && lse, "softmax_attn_fwd: null pointer");
  AttnShape sh;
  MHR_REQUIRE(attn_shape(head_dim, sh), "softmax_attn_fwd: head_dim=%d unsupported (multiple of 8, <= 128)", head_dim);
  MHR_REQUIRE(n_seqs > 0 && max_len > 0 && n_heads > 0 && n_kv_heads > 0 && n_heads % n_kv_heads == 0,
              "softmax_attn_fwd: bad sizes (n_seqs=%d max_len=%d heads=%d kv_heads=%d)", n_seqs, max_len, n_heads, n_kv_heads);
  MHR_REQUIRE(row_stride % 8 == 0 && (n_heads * head_dim) % 4 == 0, "softmax_attn_fwd: strides must be multiples of 8");
  const size_t lds = softmax_attn_lds(max_len, sh.nks, false);
  MHR_REQUIRE(lds <= 160 * 1024, "softmax_attn_fwd: max_len=%d head_dim=%d needs %zu B of LDS (> 160 KiB)", max_len, head_dim, lds);
  const int nb = (max_len + 31) / 32, threads = 64 * (nb < 4 ? nb : 4);
  hipStream_t s = (hipStream_t)stream;
#define L_(NKS, ND)                                                                                                    \
  {                                                                                                                    \
    auto kern = softmax_attn_fwd_kernel<NKS, ND>;                                                                      \
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL(kern, dim3(n_seqs * n_heads), dim3(threads), lds, s, (const bf16_t*)q, (const bf16_t*)k,        \
                       (const bf16_t*)v, row_stride, cu_seqlens, key_valid, (bf16_t*)out, (int64_t)n_heads * head_dim, lse, \
                       max_len, n_heads, n_heads / n_kv_heads, head_dim, scale);                                        \
  }
  ATTN_DISPATCH(sh, L_);
#undef L_
  MHR_CHECK_LAUNCH("softmax_attn_fwd");
  return MHR_OK;
}

extern "C" int mhr_softmax_attn_bwd(const void* q, const void* k, const void* v, int64_t row_stride, const int32_t* cu_seqlens,
                                    const uint8_t* key_valid, const void* out, const void* d_out, const float* lse, void* dq,
                                    int64_t dq_stride, void* dk, void* dv, int64_t dkv_stride, int n_seqs, int max_len,
                                    int n_heads, int n_kv_heads, int head_dim, float scale, void* stream) {
  MHR_REQUIRE(q && k && v && out && d_out && lse && dq && dk && dv, "softmax_attn_bwd: null pointer");
  AttnShape sh;
  MHR_REQUIRE(attn_shape(head_dim, sh), "softmax_attn_bwd: head_dim=%d unsupported (multiple of 8, <= 128)", head_dim);
  MHR_REQUIRE(n_seqs > 0 && max_len > 0 && n_heads > 0 && n_kv_heads > 0 && n_heads % n_kv_heads == 0,
              "softmax_attn_bwd: bad sizes (n_seqs=%d max_len=%d heads=%d kv_heads=%d)", n_seqs, max_len, n_heads, n_kv_heads);
  MHR_REQUIRE(row_stride % 8 == 0 && dq_stride > 0 && dkv_stride >= (int64_t)n_heads * head_dim,
              "softmax_attn_bwd: dk/dv hold one slab per QUERY head (stride >= n_heads * head_dim)");
  const size_t lds = softmax_attn_lds(max_len, sh.nks, true);
  MHR_REQUIRE(lds <= 160 * 1024, "softmax_attn_bwd: max_len=%d head_dim=%d needs %zu B of LDS (> 160 KiB)", max_len, head_dim, lds);
  const int nb = (max_len + 31) / 32, threads = 64 * (nb < 4 ? nb : 4);
  hipStream_t s = (hipStream_t)stream;
#define L_(NKS, ND)                                                                                                    \
  {                                                                                                                    \
    auto kern = softmax_attn_bwd_kernel<NKS, ND>;                                                                      \
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL(kern, dim3(n_seqs * n_heads), dim3(threads), lds, s, (const bf16_t*)q, (const bf16_t*)k,        \
                       (const bf16_t*)v, row_stride, cu_seqlens, key_valid, (const bf16_t*)out, (const bf16_t*)d_out,   \
                       (int64_t)n_heads * head_dim, lse, (bf16_t*)dq, dq_stride, (bf16_t*)dk, (bf16_t*)dv, dkv_stride,  \
                       max_len, n_heads, n_heads / n_kv_heads, head_dim, scale);                                        \
  }
  ATTN_DISPATCH(sh, L_);
#undef L_
  MHR_CHECK_LAUNCH("softmax_attn_bwd");
  return MHR_OK;
}
