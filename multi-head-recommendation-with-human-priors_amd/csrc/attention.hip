// HSTU pointwise-gated attention, forward and backward, for gfx950.
//
//   out[b,n,h,:] = sum_{m<=n, valid[b,m]} silu(q[b,n,h].k[b,m,h]) / L * v[b,m,h,:]
//
// (reference model/IDNet/hstu.py:137-160; the [B,H,L,L] score tensor the reference materialises three
// times never leaves registers here).  No softmax, hence no running max / rescale: every 32x32 score
// tile is independent.
//
// Mapping: one 256-thread workgroup (4 waves, one per SIMD) per (batch, head).  Tiles are 32 queries x
// 32 keys on v_mfma_f32_32x32x16_bf16.  Scores are computed TRANSPOSED (keys on the accumulator rows,
// queries on the lanes) so that the gated tile, converted to bf16 in registers, is directly the B operand
// of the second product O^T = V^T . P^T (summing over the accumulator's row index needs no lane
// movement, cdna guide section 3).  K and V are staged once per workgroup in LDS as swizzled 32-row tile images
// (stream_gemm.h) with the load-time SiLU of hstu.py:244-245 applied; the second product reads V^T fragments
// from the same image with ds_read_b64_tr_b16.  The backward runs two passes per workgroup (dK/dV per key block
// with Q and dO resident, then dQ per query block with K and V resident) so that no cross-wave reduction or
// atomic is needed and the result is bitwise reproducible; every inner-loop operand comes from LDS.
#include "mhr_common.h"
#include "stream_gemm.h"
#include "attn_tiles.h"

#ifndef ATTN_BWD_WG
#define ATTN_BWD_WG 3
#endif
namespace {

using namespace attn;

#ifdef MHR_STAMP   // in-kernel phase timing, only in the builds tools/stamp_nce.py makes
__device__ unsigned long long g_attn_stamps[16];
#define ASTAMP(k)                                                                       \
  {                                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                  \
    unsigned long long t_;                                                              \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");          \
    __builtin_amdgcn_sched_barrier(0);                                                  \
    if (blockIdx.x == 37 && threadIdx.x == 0) g_attn_stamps[k] = t_;                    \
  }
#else
#define ASTAMP(k)
#endif


// ------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------
// Workgroups are dealt to the 8 XCDs round-robin (workgroup i runs on XCD i % 8) and every XCD has its own L2.  A head's
// q / k / v / dO slice of a row is head_dim * 2 bytes - 64 B at head_dim 32, HALF a 128-byte line whose other half belongs
// to the neighbouring head - so with the natural order (b * n_heads + head, n_heads = 8) the heads of one sequence land on
// eight different XCDs and every line is fetched from HBM once per head that touches it.  This decode keeps all heads of a
// sequence on one XCD (b = 8 * (slot / n_heads) + xcd): the neighbours' halves are L2 hits.
__device__ __forceinline__ void decode_seq_head(int n_heads, int& b, int& head) {
  const int n_wg = gridDim.x, i = blockIdx.x;
  const int full = (n_wg / (8 * n_heads)) * (8 * n_heads);          // whole groups of 8 sequences; the remainder keeps the plain order
  if (i < full) {
    const int xcd = i & 7, slot = i >> 3;
    head = slot % n_heads;
    b = (slot / n_heads) * 8 + xcd;
  } else {
    b = i / n_heads;
    head = i % n_heads;
  }
}

template <int NKS, int ND>
__global__ __launch_bounds__(256) void hstu_attn_fwd_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                            const bf16_t* __restrict__ v, int64_t stride,
                                                            const uint8_t* __restrict__ key_valid, bf16_t* __restrict__ out,
                                                            int64_t out_stride, bf16_t* act_q, bf16_t* act_k, bf16_t* act_v,
                                                            int64_t act_stride, int L, int n_heads, int hd, int apply_silu,
                                                            float inv_n) {
  using T = sg::Tile<NKS>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int Lp = (L + 31) & ~31, nb = Lp >> 5;
  unsigned char* Kt = smem;                       // nb tiles: activated K
  unsigned char* Vt = smem + nb * T::BYTES;       // nb tiles: activated V
  uint32_t* vmask = reinterpret_cast<uint32_t*>(Vt + nb * T::BYTES);

  int b, head;
  decode_seq_head(n_heads, b, head);
  const int64_t row0 = (int64_t)b * L;
  const bf16_t* qp = q + row0 * stride + head * hd;
  const bf16_t* kp = k + row0 * stride + head * hd;
  const bf16_t* vp = v + row0 * stride + head * hd;
  bf16_t* aq = act_q ? act_q + row0 * act_stride + head * hd : nullptr;
  bf16_t* ak = act_k ? act_k + row0 * act_stride + head * hd : nullptr;
  bf16_t* av = act_v ? act_v + row0 * act_stride + head * hd : nullptr;
  const bool do_silu = apply_silu != 0;

  stage_tiles<NKS>(Kt, kp, stride, L, Lp, hd, do_silu, ak, act_stride);
  stage_tiles<NKS>(Vt, vp, stride, L, Lp, hd, do_silu, av, act_stride);
  build_valid_mask(vmask, key_valid + row0, L, nb);
  __syncthreads();

  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 31, half = lane >> 5;
  sg::LaneAddr<NKS> la;
  la.init(lane);
  for (int it = 0; it * 4 < nb; ++it) {
    // snake from the HEAVY end (query block qb costs qb + 1 tile pairs): blocks nb-1 .. nb-4 to waves 0..3, the next four to
    // waves 3..0, ...  Dealt from the light end, nb = 7 (L = 200) came out as 1 / 9 / 9 / 9 pairs per wave instead of 7 each.
    const int qb = nb - 1 - ((it & 1) ? it * 4 + (3 - wave) : it * 4 + wave);
    if (qb < 0) continue;
    const int qrow = qb * 32 + r;
    bf16x8 qf[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const int koff = ks * 16 + 8 * half;
      qf[ks] = load_frag(qp, stride, qrow, L, koff, hd);
      if (do_silu) qf[ks] = silu8(qf[ks]);
      if (aq && qrow < L && koff < hd) *reinterpret_cast<bf16x8*>(aq + (int64_t)qrow * act_stride + koff) = qf[ks];
    }
    f32x16 o[ND];
#pragma unroll
    for (int dc = 0; dc < ND; ++dc) o[dc] = zero16();

    for (int kb = 0; kb <= qb; ++kb) {
      const uint32_t vm = vmask[kb];
      if (vm == 0) continue;                       // a block of padding keys contributes exactly nothing
      const unsigned char* kt = Kt + kb * T::BYTES;
      const unsigned char* vt = Vt + kb * T::BYTES;
      f32x16 s = zero16();
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks)
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(la.read_a(kt, ks), qf[ks], s, 0, 0, 0);   // S^T: rows = keys, cols = queries
      // off-diagonal tiles of fully valid key blocks need no mask at all (1/L is applied once, at the store)
      if (kb < qb && vm == 0xFFFFFFFFu) {
#pragma unroll
        for (int g = 0; g < 16; ++g) s[g] *= fast_sigmoid(s[g]);
      } else {
#pragma unroll
        for (int g = 0; g < 16; ++g) {
          const int kl = crow(g, half);
          const bool ok = ((vm >> kl) & 1u) && (kb * 32 + kl <= qrow);
          s[g] = ok ? s[g] * fast_sigmoid(s[g]) : 0.f;
        }
      }
      bf16x8 p0, p1;
      pack_acc(s, p0, p1);
#pragma unroll
      for (int dc = 0; dc < ND; ++dc) {
        o[dc] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(la.read_tr(vt, dc, 0), p0, o[dc], 0, 0, 0);   // O^T += V^T . P^T
        o[dc] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(la.read_tr(vt, dc, 1), p1, o[dc], 0, 0, 0);
      }
    }
    if (qrow < L) {
      bf16_t* orow = out + (row0 + qrow) * out_stride + head * hd;
#pragma unroll
      for (int dc = 0; dc < ND; ++dc)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int d0 = dc * 32 + 8 * g4 + 4 * half;
          if (d0 < hd) {
            bf16x4 w = {(bf16_t)(o[dc][4 * g4] * inv_n), (bf16_t)(o[dc][4 * g4 + 1] * inv_n), (bf16_t)(o[dc][4 * g4 + 2] * inv_n),
                        (bf16_t)(o[dc][4 * g4 + 3] * inv_n)};
            *reinterpret_cast<bf16x4*>(orow + d0) = w;
          }
        }
    }
  }
}

// ------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------
template <int NKS, int ND>
__global__ __launch_bounds__(256, NKS <= 2 ? ATTN_BWD_WG : 1) void hstu_attn_bwd_kernel(
    const bf16_t* __restrict__ q_pre, const bf16_t* __restrict__ k_pre, const bf16_t* __restrict__ v_pre, int64_t stride,
    const bf16_t* __restrict__ act_q, const bf16_t* __restrict__ act_k, const bf16_t* __restrict__ act_v, int64_t act_stride,
    const uint8_t* __restrict__ key_valid, const bf16_t* __restrict__ d_out, int64_t do_stride, bf16_t* __restrict__ dq,
    bf16_t* __restrict__ dk, bf16_t* __restrict__ dv, int64_t d_stride, int L, int n_heads, int hd, int apply_silu,
    float inv_n) {
  using T = sg::Tile<NKS>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int Lp = (L + 31) & ~31, nb = Lp >> 5;
  unsigned char* T0 = smem;                      // Q tiles in pass A, K tiles in pass B
  unsigned char* T1 = smem + nb * T::BYTES;      // dO tiles in pass A, V tiles in pass B
  uint32_t* vmask = reinterpret_cast<uint32_t*>(T1 + nb * T::BYTES);

  int b, head;
  decode_seq_head(n_heads, b, head);
  const int64_t row0 = (int64_t)b * L;
  const int hoff = head * hd;
  // without saved activations (act_q == nullptr) silu(q), silu(k), silu(v) are recomputed from the pre-activation values
  // while they are staged / loaded: 39 MB less to write in the forward and to read here per layer at cfg1
  const bool redo = act_q == nullptr;
  const int64_t a_stride = redo ? stride : act_stride;
  const bf16_t* aq = redo ? q_pre + row0 * stride + hoff : act_q + row0 * act_stride + hoff;
  const bf16_t* ak = redo ? k_pre + row0 * stride + hoff : act_k + row0 * act_stride + hoff;
  const bf16_t* av = redo ? v_pre + row0 * stride + hoff : act_v + row0 * act_stride + hoff;
  const bf16_t* dop = d_out + row0 * do_stride + hoff;
  const bool chain = apply_silu != 0;

  // Everything the inner loops touch lives in LDS: the per-(query block, key block) work used to fetch its Q / dO (or
  // K / V) fragments straight from global memory, one dependent L2 round trip per 8 MFMAs (386 MB of traffic per
  // launch against 65 MB of operands, 176 us per layer at cfg1).
  ASTAMP(0)
  stage_tiles<NKS>(T0, aq, a_stride, L, Lp, hd, redo, nullptr, 0);
  stage_tiles<NKS>(T1, dop, do_stride, L, Lp, hd, false, nullptr, 0);
  build_valid_mask(vmask, key_valid + row0, L, nb);
  __syncthreads();
  ASTAMP(1)

  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 31, half = lane >> 5;
  sg::LaneAddr<NKS> la;
  la.init(lane);

  // ---- pass A: dK, dV for key block kb (keys on the lanes) -----------------------------------------
  for (int it = 0; it * 4 < nb; ++it) {
    const int kb = (it & 1) ? it * 4 + wave : it * 4 + (3 - wave);   // early key blocks are the heavy ones
    if (kb >= nb) continue;
    const int key = kb * 32 + r;
    const bool kvalid = (vmask[kb] >> r) & 1u;
    bf16x8 kf[NKS], vf[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      kf[ks] = load_frag(ak, a_stride, key, L, ks * 16 + 8 * half, hd);
      vf[ks] = load_frag(av, a_stride, key, L, ks * 16 + 8 * half, hd);
      if (redo) {
        kf[ks] = silu8(kf[ks]);
        vf[ks] = silu8(vf[ks]);
      }
    }
    f32x16 dvacc[ND], dkacc[ND];
#pragma unroll
    for (int dc = 0; dc < ND; ++dc) {
      dvacc[dc] = zero16();
      dkacc[dc] = zero16();
    }
    const uint32_t vm_kb = vmask[kb];
    for (int qb = kb; qb < nb && vm_kb != 0; ++qb) {     // a block of padding keys gets zero gradients
      const unsigned char* qt = T0 + qb * T::BYTES;
      const unsigned char* dot = T1 + qb * T::BYTES;
      f32x16 s = zero16(), dp = zero16();
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(la.read_a(qt, ks), kf[ks], s, 0, 0, 0);      // S: rows = queries, cols = keys
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(la.read_a(dot, ks), vf[ks], dp, 0, 0, 0);   // dP = dO . V^T
      }
      // P = silu(S), dS = dP silu'(S); the 1/L of both is applied once, when dV / dK are stored.  Off-diagonal tiles of
      // fully valid key blocks skip the mask arithmetic.
      if (qb > kb && vm_kb == 0xFFFFFFFFu) {
#pragma unroll
        for (int g = 0; g < 16; ++g) {
          const float x = s[g];
          const float sig = fast_sigmoid(x);
          s[g] = x * sig;
          dp[g] = dp[g] * sig * (1.0f + x * (1.0f - sig));
        }
      } else {
#pragma unroll
        for (int g = 0; g < 16; ++g) {
          const int qi = qb * 32 + crow(g, half);
          const bool ok = kvalid && key <= qi;
          const float x = s[g];
          const float sig = fast_sigmoid(x);
          s[g] = ok ? x * sig : 0.f;                                           // P
          dp[g] = ok ? dp[g] * sig * (1.0f + x * (1.0f - sig)) : 0.f;          // dS
        }
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
            float gv = dvacc[dc][g] * inv_n, gk = dkacc[dc][g] * inv_n;
            if (chain) {
              gv *= dsilu_f((float)v_pre[(row0 + m) * stride + hoff + d]);
              gk *= dsilu_f((float)k_pre[(row0 + m) * stride + hoff + d]);
            }
            dv[(row0 + m) * d_stride + hoff + d] = (bf16_t)gv;
            dk[(row0 + m) * d_stride + hoff + d] = (bf16_t)gk;
          }
        }
      }
    }
  }

  // ---- pass B: dQ for query block qb (queries on the lanes) ----------------------------------------
  ASTAMP(2)
  __syncthreads();                      // everyone is done reading the Q / dO tiles
  ASTAMP(3)
  stage_tiles<NKS>(T0, ak, a_stride, L, Lp, hd, redo, nullptr, 0);
  stage_tiles<NKS>(T1, av, a_stride, L, Lp, hd, redo, nullptr, 0);
  __syncthreads();
  ASTAMP(4)
  for (int it = 0; it * 4 < nb; ++it) {
    const int qb = nb - 1 - ((it & 1) ? it * 4 + (3 - wave) : it * 4 + wave);     // snake from the heavy end (see the forward)
    if (qb < 0) continue;
    const int qcol = qb * 32 + r;
    bf16x8 qf[NKS], dof[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      qf[ks] = load_frag(aq, a_stride, qcol, L, ks * 16 + 8 * half, hd);
      if (redo) qf[ks] = silu8(qf[ks]);
      dof[ks] = load_frag(dop, do_stride, qcol, L, ks * 16 + 8 * half, hd);
    }
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
      if (kb < qb && vm == 0xFFFFFFFFu) {
#pragma unroll
        for (int g = 0; g < 16; ++g) {
          const float x = s[g];
          const float sig = fast_sigmoid(x);
          dp[g] = dp[g] * sig * (1.0f + x * (1.0f - sig));
        }
      } else {
#pragma unroll
        for (int g = 0; g < 16; ++g) {
          const int kl = crow(g, half);
          const bool ok = ((vm >> kl) & 1u) && (kb * 32 + kl <= qcol);
          const float x = s[g];
          const float sig = fast_sigmoid(x);
          dp[g] = ok ? dp[g] * sig * (1.0f + x * (1.0f - sig)) : 0.f;          // dS^T
        }
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
          if (n < L) {
            float gq = dqacc[dc][g] * inv_n;
            if (chain) gq *= dsilu_f((float)q_pre[(row0 + n) * stride + hoff + d]);
            dq[(row0 + n) * d_stride + hoff + d] = (bf16_t)gq;
          }
        }
      }
    }
  }
  ASTAMP(5)
}

}  // namespace


extern "C" int mhr_hstu_attn_fwd(const void* q, const void* k, const void* v, int64_t row_stride, const uint8_t* key_valid,
                                 void* out, void* act_q, void* act_k, void* act_v, int64_t act_stride, int B, int L,
                                 int n_heads, int head_dim, int apply_silu, void* stream) {
  MHR_REQUIRE(q && k && v && key_valid && out, "hstu_attn_fwd: null pointer");
  AttnShape sh;
  MHR_REQUIRE(attn_shape(head_dim, sh), "hstu_attn_fwd: head_dim=%d unsupported (multiple of 8, <= 128)", head_dim);
  MHR_REQUIRE(B > 0 && L > 0 && n_heads > 0, "hstu_attn_fwd: bad sizes");
  MHR_REQUIRE(row_stride % 8 == 0 && (!act_q || act_stride % 8 == 0), "hstu_attn_fwd: strides must be multiples of 8");
  MHR_REQUIRE((act_q != nullptr) == (act_k != nullptr) && (act_k != nullptr) == (act_v != nullptr),
              "hstu_attn_fwd: act_q/act_k/act_v must be all set or all null");
  const int Lp = (L + 31) & ~31, nb = Lp / 32;
  size_t lds = (size_t)2 * nb * (32 * sh.nks * 32) + (size_t)nb * 4 + 16;     // K and V tile images + validity words
  MHR_REQUIRE(lds <= 160 * 1024 && nb <= 256, "hstu_attn_fwd: L=%d head_dim=%d needs %zu B of LDS (> 160 KiB)", L, head_dim, lds);
  const float inv_n = 1.0f / (float)L;
  const int64_t out_stride = (int64_t)n_heads * head_dim;
  hipStream_t s = (hipStream_t)stream;
#define L_(NKS, ND)                                                                                                    \
  {                                                                                                                    \
    auto kern = hstu_attn_fwd_kernel<NKS, ND>;                                                                         \
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL(kern, dim3(B * n_heads), dim3(256), lds, s, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, \
                       row_stride, key_valid, (bf16_t*)out, out_stride, (bf16_t*)act_q, (bf16_t*)act_k, (bf16_t*)act_v, \
                       act_stride, L, n_heads, head_dim, apply_silu, inv_n);                                           \
  }
  ATTN_DISPATCH(sh, L_);
#undef L_
  MHR_CHECK_LAUNCH("hstu_attn_fwd");
  return MHR_OK;
}

extern "C" int mhr_hstu_attn_bwd(const void* q_pre, const void* k_pre, const void* v_pre, int64_t row_stride,
                                 const void* act_q, const void* act_k, const void* act_v, int64_t act_stride,
                                 const uint8_t* key_valid, const void* d_out, void* dq, void* dk, void* dv, int64_t d_stride,
                                 int B, int L, int n_heads, int head_dim, int apply_silu, void* stream) {
  MHR_REQUIRE(key_valid && d_out && dq && dk && dv, "hstu_attn_bwd: null pointer");
  MHR_REQUIRE((act_q != nullptr) == (act_k != nullptr) && (act_k != nullptr) == (act_v != nullptr),
              "hstu_attn_bwd: act_q/act_k/act_v must be all set or all null");
  MHR_REQUIRE(act_q || apply_silu, "hstu_attn_bwd: without saved activations the inputs are recomputed as silu(pre): apply_silu must be set");
  MHR_REQUIRE(!apply_silu || (q_pre && k_pre && v_pre), "hstu_attn_bwd: pre-activation inputs required with apply_silu");
  AttnShape sh;
  MHR_REQUIRE(attn_shape(head_dim, sh), "hstu_attn_bwd: head_dim=%d unsupported (multiple of 8, <= 128)", head_dim);
  MHR_REQUIRE(B > 0 && L > 0 && n_heads > 0, "hstu_attn_bwd: bad sizes");
  MHR_REQUIRE((!act_q || act_stride % 8 == 0) && row_stride % 8 == 0 && ((int64_t)n_heads * head_dim) % 8 == 0,
              "hstu_attn_bwd: strides must be multiples of 8");
  const int Lp = (L + 31) & ~31, nb = Lp / 32;
  size_t lds = (size_t)2 * nb * (32 * sh.nks * 32) + (size_t)nb * 4 + 16;     // two tensors' tile images at a time
  MHR_REQUIRE(lds <= 160 * 1024 && nb <= 256, "hstu_attn_bwd: L=%d head_dim=%d needs %zu B of LDS (> 160 KiB)", L, head_dim, lds);
  const float inv_n = 1.0f / (float)L;
  const int64_t do_stride = (int64_t)n_heads * head_dim;
  hipStream_t s = (hipStream_t)stream;
#define L_(NKS, ND)                                                                                                    \
  {                                                                                                                    \
    auto kern = hstu_attn_bwd_kernel<NKS, ND>;                                                                         \
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL(kern, dim3(B * n_heads), dim3(256), lds, s, (const bf16_t*)q_pre, (const bf16_t*)k_pre,          \
                       (const bf16_t*)v_pre, row_stride, (const bf16_t*)act_q, (const bf16_t*)act_k, (const bf16_t*)act_v, \
                       act_stride, key_valid, (const bf16_t*)d_out, do_stride, (bf16_t*)dq, (bf16_t*)dk, (bf16_t*)dv,   \
                       d_stride, L, n_heads, head_dim, apply_silu, inv_n);                                             \
  }
  ATTN_DISPATCH(sh, L_);
#undef L_
  MHR_CHECK_LAUNCH("hstu_attn_bwd");
  return MHR_OK;
}

#ifdef MHR_STAMP
extern "C" int mhr_debug_read_attn_stamps(unsigned long long* host16) {
  return (int)hipMemcpyFromSymbol(host16, HIP_SYMBOL(g_attn_stamps), 16 * sizeof(unsigned long long));
}
#endif
