// Sampled softmax (InfoNCE) with false-negative suppression, fused forward and backward for gfx950.
//
// Reference (file:line under code/REC/): model/IDNet/hstu.py:600-619 (nce_loss) + F.cross_entropy at
// hstu.py:697/833, called once per prior head.  The reference materialises neg_logits, fix_logits, their
// concatenation and the log-softmax, each [N_tok, N_neg]; here a logit lives only in an MFMA accumulator.
//
//   s_ij = cos(q_i, n_j), f_ij = cos(p_i, n_j), s_i+ = cos(q_i, p_i), scale = exp(clamp(logit_scale, 0, ln 100))
//   keep_ij = !(f_ij > thres);   lse_i = log( exp(scale*s_i+) + sum_j keep_ij exp(scale*s_ij) );   loss_i = lse_i - scale*s_i+
//
// Three kernels on the row-stationary streaming GEMM core (stream_gemm.h), all with S^T orientation
// (streamed rows on accumulator registers, stationary rows on lanes).  One launch serves every prior category
// (grid.z = group): per-group token lists (q_idx/p_idx into the shared head / target row matrices), per-group
// negative pools, per-group live counts read from device memory (no host sync).
//   nce_fwd    : tokens stationary (rows gathered through the index lists, L2-normalised in registers, kept as
//                MFMA B fragments, 32 tokens per wave), 32-negative tiles streamed through a 3-slot LDS ring
//                filled by LDS-DMA (global_load_lds_dwordx4); two MFMAs per LDS fragment read (s and f share
//                the negative operand); running sums are one VGPR per lane because |logit| <= scale bounds the
//                exponent (no running max).  Stores the normalised bf16 token rows, 1/||q||, 1/||p||, s+ and the
//                false-negative suppression BITS (one bit per (token, negative)) for the backward.
//   nce_finalize: per-token lse / loss / rank across the negative splits, per-group means.
//   nce_bwd_q  : tokens stationary; recomputes s only (suppression comes from the saved bits);
//                G_ij = w_i keep_ij exp(scale*s_ij - lse_i) in bf16 is the A operand of dQn += G . N, where the
//                N^T fragments come from the SAME LDS tile via ds_read_b64_tr_b16 (no transposed copy).  The
//                S(t) MFMAs carry the exp/convert epilogue of S(t-1) in their issue gaps (one-tile skew).
//                Finishes with the L2-normalisation chain rule and adds dq/dp rows straight into the head /
//                target row gradients with 256-B-shaped float atomics.
//   nce_bwd_n  : negatives stationary (32 per wave), token tiles streamed (Qn tile + its w/lse/supp words by
//                LDS-DMA); dN += G^T . Qn with Qn^T fragments from ds_read_b64_tr_b16; accumulated in registers
//                across the workgroup's token share, then one float-atomic pass.
#include "mhr_common.h"
#include "stream_gemm.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;

// In-kernel phase timing (cdna guide section 7, "stamp before you restructure"): only in -DMHR_STAMP builds made by
// tools/stamp_nce.py; the product library carries none of it.
#ifdef MHR_STAMP
__device__ unsigned long long g_stamps[16];
#define STAMP_DECL unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tp_ = 0; \
  const bool st_on_ = blockIdx.x == 3 && blockIdx.y == 0 && blockIdx.z == 0;
#ifdef STAMP_COARSE
#define STAMP_SKIP(k) ((k) != 5 && (k) != -1)
#else
#define STAMP_SKIP(k) false
#endif
#define STAMP(k)                                                                             \
  if constexpr (!STAMP_SKIP(k)) {                                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                       \
    unsigned long long t_;                                                                   \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");               \
    __builtin_amdgcn_sched_barrier(0);                                                       \
    if ((k) >= 0) st_[(k) < 0 ? 0 : (k)] += t_ - tp_;                                        \
    tp_ = t_;                                                                                \
  }
#define STAMP_FLUSH                                                            \
  if (st_on_ && threadIdx.x == 0)                                              \
    for (int k_ = 0; k_ < 8; ++k_) g_stamps[k_] = st_[k_];
#else
#define STAMP_DECL
#define STAMP(k)
#define STAMP_FLUSH
#endif

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

__device__ __forceinline__ float clamp_scale(const float* logit_scale_dev) {
  float ls = *logit_scale_dev;
  ls = fminf(fmaxf(ls, 0.0f), 4.605170185988092f);   // [0, ln 100]  (hstu.py:602)
  return __expf(ls);
}

template <typename IT>
__device__ __forceinline__ void load8(const IT* p, float (&out)[8]);
template <>
__device__ __forceinline__ void load8<bf16_t>(const bf16_t* p, float (&out)[8]) {
  bf16x8 v = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
  for (int i = 0; i < 8; ++i) out[i] = (float)v[i];
}
template <>
__device__ __forceinline__ void load8<float>(const float* p, float (&out)[8]) {
  f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    out[i] = a[i];
    out[4 + i] = b[i];
  }
}

// lane (r, half) owns elements k = ks*16 + 8*half + j of row `src`; returns 1/||row||
template <int NKS, typename IT>
__device__ __forceinline__ float row_inv_norm(const IT* src, bool live, int half) {
  float ss = 0.f;
  if (live) {
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      float v[8];
      load8<IT>(src + ks * 16 + 8 * half, v);
#pragma unroll
      for (int i = 0; i < 8; ++i) ss += v[i] * v[i];
    }
  }
  ss += __shfl_xor(ss, 32, 64);
  return live ? 1.0f / sqrtf(ss) : 0.f;
}

template <int NKS, typename IT>
__device__ __forceinline__ void load_norm_frags(const IT* src, bool live, int half, float inv, bf16x8 (&frag)[NKS]) {
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    if (live) {
      float v[8];
      load8<IT>(src + ks * 16 + 8 * half, v);
#pragma unroll
      for (int i = 0; i < 8; ++i) frag[ks][i] = (bf16_t)(v[i] * inv);
    } else {
      frag[ks] = sg::zero8();
    }
  }
}

// exp2 of the gated logit with the per-token weight folded into the exponent, then masked BITWISE (dead rows may hold
// NaN / garbage: x & 0 is 0 where 0.0f * NaN is not).  The bit tests are spelled as v_bfe_i32 + v_and / v_bfi: hipcc turns
// the C form into v_lshlrev + v_cmp + s_nop + v_cndmask, which overflows the 32-cycle MFMA gap the epilogue lives in.
// One asm block per element, with the bit extract BETWEEN v_exp_f32 and its consumer: on gfx940+ a VALU instruction
// reading a transcendental result needs one instruction in between (hipcc inserts that for code it can see, not for
// inline asm - back to back, the first lanes read the register before v_exp has written it).
__device__ __forceinline__ float gate_alive(float s, float c1, float c0, uint32_t alive_bits, int pos) {   // bit = 1: keep
  const float x = s * c1 - c0;
  float g;
  int m;
  asm("v_exp_f32 %0, %2\n\tv_bfe_i32 %1, %3, %4, 1\n\tv_and_b32 %0, %0, %1"
      : "=&v"(g), "=&v"(m) : "v"(x), "v"(alive_bits), "v"(pos));
  return g;
}
// nothing to mask (the plain forms): exp2 only.  The s_nop is the wait state a consumer of a transcendental needs when hipcc
// cannot see it (the running sum is an inline-asm v_add: without the nop its first lanes read the register too early).
__device__ __forceinline__ float gate_plain(float s, float c1, float c0) {
  const float x = s * c1 - c0;
  float g;
  asm("v_exp_f32 %0, %1\n\ts_nop 0" : "=v"(g) : "v"(x));
  return g;
}
__device__ __forceinline__ float gate_dead(float s, float c1, float c0, uint32_t dead_bits, int pos) {     // bit = 1: drop
  const float x = s * c1 - c0;
  float g;
  int m;
  asm("v_exp_f32 %0, %2\n\tv_bfe_i32 %1, %3, %4, 1\n\tv_bfi_b32 %0, %1, 0, %0"                          // (m & 0) | (~m & e)
      : "=&v"(g), "=&v"(m) : "v"(x), "v"(dead_bits), "v"(pos));
  return g;
}

// ------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------
template <int NKS, typename IT, bool LOGS>
__global__ __launch_bounds__(256, 2) void nce_fwd_kernel(const IT* __restrict__ q_rows, const int32_t* q_idx,
                                                         const IT* __restrict__ p_rows, const int32_t* p_idx,
                                                         const bf16_t* negs, int n_neg,
                                                         const int32_t* n_tok_dev, int tok_cap,
                                                         const float* __restrict__ logit_scale_dev, float thres,
                                                         int tiles_per_split, float* sum_out,
                                                         int32_t* n_valid, int32_t* rank,
                                                         bf16_t* qn_out, bf16_t* pn_out,
                                                         uint32_t* supp_out, float* q_inv,
                                                         float* p_inv, float* s_pos_out, int log_group) {
  using T = sg::Tile<NKS>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // blockIdx.z selects the group (negative pool / prior category): every per-token array is [groups, tok_cap(, D)],
  // negatives are [groups, n_neg, D]; query / positive source rows are shared.
  {
    const int64_t grp = blockIdx.z, to = grp * tok_cap;
    q_idx += to; p_idx += to; n_tok_dev += grp; negs += grp * (int64_t)((n_neg + 31) & ~31) * T::DIM; sum_out += to;
    if (n_valid) n_valid += to;
    if (rank) rank += to;
    if (qn_out) qn_out += to * T::DIM;
    if (pn_out) pn_out += to * T::DIM;
    if (supp_out) supp_out += grp * (int64_t)((n_neg + 31) >> 5) * tok_cap;
    if (q_inv) q_inv += to;
    if (p_inv) p_inv += to;
    s_pos_out += to;
  }
  const int n_tok = min(*n_tok_dev, tok_cap);
  const int tok0 = blockIdx.x * 128;
  if (tok0 >= n_tok) return;
  // blockIdx.y selects a contiguous range of negative tiles: (token block, negative range) units keep the
  // 512 workgroup slots busy without a long tail; partial sums meet in sum_out (float atomics) and
  // mhr_nce_finalize turns them into lse / loss.
  const int n_tiles = (n_neg + 31) >> 5;
  const int t_begin = blockIdx.y * tiles_per_split, t_end = min(n_tiles, t_begin + tiles_per_split);
  if (t_begin >= t_end) return;
  const bool first_split = blockIdx.y == 0;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 31, half = lane >> 5;
  const int tok = tok0 + wave * 32 + r;
  const bool live = tok < n_tok;

  bf16x8 frag[2][NKS];   // [0] = normalised query, [1] = normalised positive
  float qi = 0.f, pi = 0.f;
  {
    const IT* qs = q_rows + (live ? (int64_t)q_idx[tok] * T::DIM : 0);
    const IT* ps = p_rows + (live ? (int64_t)p_idx[tok] * T::DIM : 0);
    qi = row_inv_norm<NKS, IT>(qs, live, half);
    pi = row_inv_norm<NKS, IT>(ps, live, half);
    load_norm_frags<NKS, IT>(qs, live, half, qi, frag[0]);
    load_norm_frags<NKS, IT>(ps, live, half, pi, frag[1]);
  }
  float spos = 0.f;
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
    for (int i = 0; i < 8; ++i) spos += (float)frag[0][ks][i] * (float)frag[1][ks][i];
  spos += __shfl_xor(spos, 32, 64);
  // Saved state for the backward.  Lanes past n_tok inside a processed block (tok < tok_cap) are written too - zero
  // rows here and all-ones suppression words below - so that the backward can stream whole 32-token tiles without
  // clamping: a padded token contributes exactly nothing.
  const bool in_cap = tok < tok_cap;
  if (in_cap && first_split) {
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const int k0 = ks * 16 + 8 * half;
      if (qn_out) *reinterpret_cast<bf16x8*>(qn_out + (int64_t)tok * T::DIM + k0) = frag[0][ks];
      if (pn_out) *reinterpret_cast<bf16x8*>(pn_out + (int64_t)tok * T::DIM + k0) = frag[1][ks];
    }
    if (live && half == 0) {
      if (q_inv) q_inv[tok] = qi;
      if (p_inv) p_inv[tok] = pi;
      if (s_pos_out) s_pos_out[tok] = spos;
    }
  }

  const float scale = clamp_scale(logit_scale_dev);
  const float c1 = scale * LOG2E;
  float sum = 0.f;
  int nv = 0, rk = 0;

  // Negative tiles stream through a 3-slot LDS-DMA ring (stream_gemm.h), branch-free: every iteration waits
  // `vmcnt(PW)` (tile i landed, tile i+1 may be in flight) and issues the PW pieces of tile min(i+2, last) with
  // SGPR-base addressing; the pool is padded to whole tiles, rows past n_neg are masked by `rem`.
  using P = sg::DmaPieces<NKS>;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  P dp;
  dp.init(wv, lane);
  auto dma_tile = [&](auto slot_c, int tn) {
    auto f = [&](auto k_c) {
      dp.template piece<decltype(k_c)::value>(smem + decltype(slot_c)::value * T::BYTES,
                                               reinterpret_cast<const char*>(negs) + (int64_t)tn * (32 * T::ROW_BYTES));
    };
    sg::static_for<P::PW>(f);
  };
  sg::LaneAddr<NKS> la;
  la.init(lane);
  const int n_loc = t_end - t_begin, t_last = t_end - 1;
  const bool do_logs = LOGS && (log_group < 0 || (int)blockIdx.z == log_group);
  dma_tile(std::integral_constant<int, 0>{}, t_begin);
  dma_tile(std::integral_constant<int, 1>{}, min(t_begin + 1, t_last));
  sg::ring_loop<3>(n_loc, [&](auto slot_c, int i) {
    constexpr int cur = decltype(slot_c)::value, nxt = (cur + 2) % 3;
    const int t = t_begin + i;
    sg::wait_vmcnt<P::PW>();
    sg::ring_barrier();
    dma_tile(std::integral_constant<int, nxt>{}, min(t + 2, t_last));
    f32x16 acc[2] = {sg::zero16(), sg::zero16()};
    sg::mma_tile<NKS, 2>(smem + cur * T::BYTES, la, frag, acc);
    const int rem = n_neg - t * 32;
    // False negatives (cos(positive, negative) > thres) are rare: one max over the tile's 16 values per lane and a
    // wave-wide vote select the common path, which is exp + add per logit and nothing else.
    float fmax = acc[1][0];
#pragma unroll
    for (int g = 1; g < 16; g += 3) fmax = fmaxf(fmaxf(fmax, acc[1][g]), fmaxf(acc[1][g + 1], acc[1][g + 2]));
    const bool plain = rem >= 32 && __builtin_amdgcn_ballot_w64(fmax > thres) == 0;
    uint32_t sbits = 0;
    if (plain) {
#pragma unroll
      for (int g = 0; g < 16; ++g) sum += fast_exp2(acc[0][g] * c1 - c1);
      if (do_logs) {
        nv += 16;
#pragma unroll
        for (int g = 0; g < 16; ++g) rk += acc[0][g] > spos ? 1 : 0;
      }
    } else {
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const bool supp = acc[1][g] > thres;               // false negative: cos(positive, negative) > thres
        const bool keep = (sg::crow(g, half) < rem) && !supp;
        const float e = fast_exp2(acc[0][g] * c1 - c1);
        sum += keep ? e : 0.f;
        sbits |= supp ? (1u << sg::crow(g, half)) : 0u;
        if (do_logs) {
          nv += keep ? 1 : 0;
          rk += (keep && acc[0][g] > spos) ? 1 : 0;
        }
      }
      sbits |= __shfl_xor(sbits, 32, 64);
    }
    // one word per (negative tile, token): bit j = negative t*32+j suppressed
    if (supp_out && in_cap && half == 0) supp_out[(int64_t)t * tok_cap + tok] = live ? sbits : 0xFFFFFFFFu;
  });
  sg::wait_vmcnt<0>();
  sum += __shfl_xor(sum, 32, 64);
  if (LOGS) {
    nv += __shfl_xor(nv, 32, 64);
    rk += __shfl_xor(rk, 32, 64);
  }
  if (live && half == 0) {
    atomicAdd(sum_out + tok, sum);
    if (LOGS) {
      if (n_valid) atomicAdd(n_valid + tok, nv);
      if (rank) atomicAdd(rank + tok, rk);
    }
  }
}

// ------------------------------------------------------------------------------------------
// forward fused with the token-side backward product (training path)
// ------------------------------------------------------------------------------------------
// dQn_i = sum_j G_ij n_j with G_ij = w_i keep_ij exp(scale s_ij - lse_i) factorises as
//     dQn_i = w_i exp(scale - lse_i) * U_i,     U_i = sum_j keep_ij exp(scale (s_ij - 1)) n_j ,
// and U_i needs neither lse nor w: it is the numerator that goes with the softmax denominator the forward already
// accumulates (the flash-attention output accumulator without the running max - |logit| <= scale bounds the
// exponent).  So the forward keeps a second accumulator U [32 tokens x D] per wave and feeds the gated tile
// E = keep * exp(scale (s - 1)) (bf16, in registers) straight back into the matrix pipe against the SAME LDS tile
// (transposed fragments).  The separate token-stationary backward kernel - a second full pass over the negatives
// that recomputed every logit - disappears; what is left of it is a row-wise kernel (nce_bwd_rows_kernel below).
// One wave per SIMD (U is 128 accumulator registers), 4-slot DMA ring, hand-ordered tile step (sg::tile_step, RF = 2):
// the s and f MFMAs of tile t carry the exp / suppression epilogue of tile t-1 in their gaps.
template <int NKS, typename IT, bool LOGS>
__global__ __launch_bounds__(256, 1) void nce_fwd_u_kernel(const IT* __restrict__ q_rows, const int32_t* q_idx,
                                                           const IT* __restrict__ p_rows, const int32_t* p_idx,
                                                           const bf16_t* negs, int n_neg,
                                                           const int32_t* n_tok_dev, int tok_cap,
                                                           const float* __restrict__ logit_scale_dev, float thres,
                                                           float* sum_out, int32_t* n_valid, int32_t* rank,
                                                           bf16_t* qn_out, bf16_t* pn_out,
                                                           uint32_t* supp_out, float* q_inv,
                                                           float* p_inv, float* s_pos_out, int log_group,
                                                           float* __restrict__ u_out) {
  using T = sg::Tile<NKS>;
  constexpr int ND = (NKS + 1) / 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  {
    const int64_t grp = blockIdx.z, to = grp * tok_cap;
    q_idx += to; p_idx += to; n_tok_dev += grp; negs += grp * (int64_t)((n_neg + 31) & ~31) * T::DIM; sum_out += to;
    if (n_valid) n_valid += to;
    if (rank) rank += to;
    qn_out += to * T::DIM; pn_out += to * T::DIM; u_out += to * T::DIM;
    supp_out += grp * (int64_t)((n_neg + 31) >> 5) * tok_cap;
    q_inv += to; p_inv += to; s_pos_out += to;
  }
  const int n_tok = min(*n_tok_dev, tok_cap);
  const int tok0 = blockIdx.x * 128;
  if (tok0 >= n_tok) return;
  const int n_tiles = (n_neg + 31) >> 5;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 31, half = lane >> 5;
  const int tok = tok0 + wave * 32 + r;
  const bool live = tok < n_tok;
  const bool in_cap = tok < tok_cap;

  bf16x8 frag[2][NKS];   // [0] = normalised query, [1] = normalised positive
  float qi = 0.f, pi = 0.f;
  {
    const IT* qs = q_rows + (live ? (int64_t)q_idx[tok] * T::DIM : 0);
    const IT* ps = p_rows + (live ? (int64_t)p_idx[tok] * T::DIM : 0);
    qi = row_inv_norm<NKS, IT>(qs, live, half);
    pi = row_inv_norm<NKS, IT>(ps, live, half);
    load_norm_frags<NKS, IT>(qs, live, half, qi, frag[0]);
    load_norm_frags<NKS, IT>(ps, live, half, pi, frag[1]);
  }
  float spos = 0.f;
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
    for (int i = 0; i < 8; ++i) spos += (float)frag[0][ks][i] * (float)frag[1][ks][i];
  spos += __shfl_xor(spos, 32, 64);
  // saved state; lanes past n_tok inside the block write zero rows / all-ones suppression words (see nce_fwd_kernel)
  if (in_cap) {
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const int k0 = ks * 16 + 8 * half;
      *reinterpret_cast<bf16x8*>(qn_out + (int64_t)tok * T::DIM + k0) = frag[0][ks];
      *reinterpret_cast<bf16x8*>(pn_out + (int64_t)tok * T::DIM + k0) = frag[1][ks];
    }
    if (live && half == 0) {
      q_inv[tok] = qi;
      p_inv[tok] = pi;
      s_pos_out[tok] = spos;
    }
  }
  const float scale = clamp_scale(logit_scale_dev);
  const float c1 = scale * LOG2E;
  const bool do_logs = LOGS && (log_group < 0 || (int)blockIdx.z == log_group);

  f32x16 u[ND];
#pragma unroll
  for (int dc = 0; dc < ND; ++dc) u[dc] = sg::zero16();
  float sum = 0.f;
  int nv = 0, rk = 0;

  using P = sg::DmaPieces<NKS>;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  P dp;
  dp.init(wv, lane);
  const int t_last = n_tiles - 1;
  auto dma_k = [&](auto k_c, auto slot_c, int tn) {
    dp.template piece<decltype(k_c)::value>(smem + decltype(slot_c)::value * T::BYTES,
                                             reinterpret_cast<const char*>(negs) + (int64_t)tn * (32 * T::ROW_BYTES));
  };
  auto dma_all = [&](auto slot_c, int tn) {
    auto f = [&](auto k_c) { dma_k(k_c, slot_c, tn); };
    sg::static_for<P::PW>(f);
  };
  sg::LaneAddr<NKS> la;
  la.init(lane);
  sg::TrAddr<NKS> ta;
  ta.init(la, smem);
  sg::RowAddr<NKS> ra;
  ra.init(la, smem);
  // slot 3 is the "previous tile" of the first iteration (E = 0 there): make it finite
  for (int o = threadIdx.x * 16; o < T::BYTES; o += 256 * 16) *reinterpret_cast<f32x4*>(smem + 3 * T::BYTES + o) = f32x4{0.f, 0.f, 0.f, 0.f};
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // published by the first ring barrier
  dma_all(std::integral_constant<int, 0>{}, 0);
  dma_all(std::integral_constant<int, 1>{}, min(1, t_last));
  // scalars first used inside the loop: touch them here, or hipcc places the `s_waitcnt lgkmcnt(0)` that covers their
  // kernel-argument load INSIDE the loop body, where it drains the LDS read pipeline once per tile
  asm volatile("" ::"s"(thres), "s"(c1), "s"(tok_cap), "s"(n_neg));
  // S / f accumulators ping-pong by tile parity: tile t's epilogue runs one iteration later, straight from the other set
  f32x16 sf[2][2] = {{sg::zero16(), sg::zero16()}, {sg::zero16(), sg::zero16()}};
  uint32_t alive_prev = 0;               // live-row bits of the previous tile, pre-shifted by 4*half
  uint32_t my_word = 0, other_word = 0;  // suppression bits of the tile before the previous one (store pending)
  // the loop exists twice (with / without the rank + n_valid counting of the logged group): a per-element runtime test
  // would put a branch into every MFMA gap
  STAMP_DECL
  STAMP(-1)
  auto run = [&](auto logs_c) {
    constexpr bool WITH_LOGS = decltype(logs_c)::value;
    sg::ring_loop<4>(n_tiles + 1, [&](auto slot_c, int i) {
      constexpr int cur = decltype(slot_c)::value, nxt = (cur + 2) % 4, prv = (cur + 3) % 4, par = cur & 1;
      STAMP(2)
      sg::wait_vmcnt<P::PW>();
      STAMP(0)
      sg::ring_barrier();
      STAMP(1)
      const int tn = min(i + 2, t_last);
      sf[par][0] = sg::zero16();
      sf[par][1] = sg::zero16();
      const f32x16& s_prev = sf[par ^ 1][0];
      const f32x16& f_cur = sf[par][1];
      uint32_t sbits = 0;                  // false negatives of THIS tile (bit (g&3)+8(g>>2) = accumulator row g of my half)
      sg::tile_step<NKS, ND, cur * T::BYTES, prv * T::BYTES, P::PW, 2>(
          ra, ta, frag, sf[par], u, [](auto) {},
          [&](int g) {                     // gated logit of the previous tile: alive_prev already excludes its false negatives
            const float ek = gate_alive(s_prev[g], c1, c1, alive_prev, (g & 3) + 8 * (g >> 2));
            asm volatile("v_add_f32 %0, %0, %1" : "+v"(sum) : "v"(ek));      // volatile: keep the accumulation in this gap
            if constexpr (WITH_LOGS) {
              const int pos = (g & 3) + 8 * (g >> 2);
              const int km = ((int)(alive_prev << (31 - pos))) >> 31;       // -1: a kept, live logit
              nv -= km;
              rk -= s_prev[g] > spos ? km : 0;
            }
            return ek;
          },
          [&](auto k_c) {                  // gaps of the second product: next tile's DMA + this tile's suppression tests
            constexpr int k = decltype(k_c)::value;
            if constexpr (k < P::PW) dma_k(k_c, std::integral_constant<int, nxt>{}, tn);
            uint32_t& sb = sbits;         // (named here: an asm operand alone does not capture it in a generic lambda)
            // the 16 tests are spread over gaps 2 .. 2 ND - 1: the first two gaps still wait for the f accumulator chain
            constexpr int G = 2 * ND - 2;
            constexpr int g_lo = G > 0 ? (k >= 2 ? (k - 2) * 16 / G : 0) : 0;
            constexpr int g_hi = G > 0 ? (k >= 2 ? (k - 1) * 16 / G : 0) : (k == 2 * ND - 1 ? 16 : 0);
#pragma unroll
            for (int g = g_lo; g < g_hi; ++g) {
              const int sm = f_cur[g] > thres ? -1 : 0;                      // false negative: cos(positive, negative) > thres
              asm volatile("v_and_or_b32 %0, %1, %2, %0" : "+v"(sb) : "v"(sm), "s"(1u << ((g & 3) + 8 * (g >> 2))));
            }
          },
          sg::EpiIdentity{}, [&] { STAMP(3) });
      STAMP(4)
      // suppression word of this tile (bit j = negative 32 i + j suppressed for my token): the two lane halves hold
      // disjoint bits; the cross-half exchange is issued here and consumed (stored) at the top of the next iteration
      if (i > 0 && in_cap && half == 0) supp_out[(int64_t)(i - 1) * tok_cap + tok] = live ? (my_word | other_word) : 0xFFFFFFFFu;
      my_word = sbits << (4 * half);
      other_word = __shfl_xor(my_word, 32, 64);
      const int rem = n_neg - i * 32;
      const uint32_t tail = rem >= 32 ? 0xFFFFFFFFu : (rem > 0 ? ~(0xFFFFFFFFu << rem) : 0u);
      alive_prev = (live && i < n_tiles) ? ((tail >> (4 * half)) & ~sbits) : 0u;
    });
  };
  if (do_logs) run(std::true_type{});
  else run(std::false_type{});
  STAMP(2)
  STAMP_FLUSH
  sg::wait_vmcnt<0>();

  sum += __shfl_xor(sum, 32, 64);
  if (LOGS) {
    nv += __shfl_xor(nv, 32, 64);
    rk += __shfl_xor(rk, 32, 64);
  }
  if (live && half == 0) {
    atomicAdd(sum_out + tok, sum);
    if (do_logs) {
      if (n_valid) atomicAdd(n_valid + tok, nv);
      if (rank) atomicAdd(rank + tok, rk);
    }
  }
  // U: rows (regs) = tokens wave*32 + crow(g, half), columns (lanes) = features dc*32 + r
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    const int tk = tok0 + wave * 32 + sg::crow(g, half);
    if (tk < n_tok) {
#pragma unroll
      for (int dc = 0; dc < ND; ++dc) {
        const int d = dc * 32 + r;
        if (d < T::DIM) u_out[(int64_t)tk * T::DIM + d] = u[dc][g];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// false-negative bits per (group, target row, negative): fixw[g][tile][row] bit j = cos(target row, negative 32 tile + j) > thres
// ------------------------------------------------------------------------------------------
// Same streaming skeleton as the catalog scorer (catalog.hip): 64 target rows per wave as two stationary fragment sets
// (normalised and rounded exactly like the token kernels do: same helper, same bf16 values), the group's negative tiles
// through a 3-slot LDS-DMA ring; the epilogue is 16 compares per fragment and one 128-byte store per (fragment, tile).
template <int NKS, typename IT>
__global__ __launch_bounds__(256, 2) void nce_fix_bits_kernel(const IT* __restrict__ p_rows, int n_rows, const bf16_t* negs,
                                                              int n_neg, float thres, uint32_t* __restrict__ fixw,
                                                              int n_rows_pad, int tiles_per_slice,
                                                              const int32_t* __restrict__ row_list, const int32_t* __restrict__ n_list,
                                                              int32_t* __restrict__ slot_of_row, int32_t* __restrict__ any_out) {
  using T = sg::Tile<NKS>;
  constexpr int RF = 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int n_tiles = (n_neg + 31) >> 5;
  negs += (int64_t)blockIdx.z * ((n_neg + 31) & ~31) * T::DIM;
  fixw += (int64_t)blockIdx.z * n_tiles * n_rows_pad;
  if (any_out) any_out += (int64_t)blockIdx.z * n_rows_pad;
  const int any_shift = fix_group_shift((n_neg + 31) >> 5);
  const int t0 = blockIdx.y * tiles_per_slice, t1 = min(n_tiles, t0 + tiles_per_slice);
  if (t0 >= t1) return;
  // with a row list only the rows some token of this group points at are tested (slot j of the list = column j of the
  // bit table; the inverse map is written here for the token kernel)
  const int n_live = row_list ? min(n_list[blockIdx.z], n_rows_pad) : n_rows;
  if ((int)blockIdx.x * 256 >= n_live) return;
  if (row_list) {
    row_list += (int64_t)blockIdx.z * n_rows_pad;
    slot_of_row += (int64_t)blockIdx.z * n_rows;
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 31, half = lane >> 5;
  bf16x8 frag[RF][NKS];
  int row[RF];
#pragma unroll
  for (int f = 0; f < RF; ++f) {
    row[f] = blockIdx.x * 256 + wave * 64 + f * 32 + r;
    const bool live = row[f] < n_live;
    int src_row = row[f];
    if (row_list) {
      src_row = live ? row_list[row[f]] : 0;
      if (live && half == 0 && blockIdx.y == 0) slot_of_row[src_row] = row[f];
    }
    const IT* src = p_rows + (live ? (int64_t)src_row * T::DIM : 0);
    const float inv = row_inv_norm<NKS, IT>(src, live, half);
    load_norm_frags<NKS, IT>(src, live, half, inv, frag[f]);
  }
  using P = sg::DmaPieces<NKS>;
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  P dp;
  dp.init(wv, lane);
  auto dma_tile = [&](auto slot_c, int tn) {
    auto f = [&](auto k_c) {
      dp.template piece<decltype(k_c)::value>(smem + decltype(slot_c)::value * T::BYTES,
                                               reinterpret_cast<const char*>(negs) + (int64_t)tn * (32 * T::ROW_BYTES));
    };
    sg::static_for<P::PW>(f);
  };
  sg::LaneAddr<NKS> la;
  la.init(lane);
  sg::RowAddr<NKS> ra;
  ra.init(la, smem);
  const int t_last = t1 - 1;
  dma_tile(std::integral_constant<int, 0>{}, t0);
  dma_tile(std::integral_constant<int, 1>{}, min(t0 + 1, t_last));
  sg::ring_loop<3>(t1 - t0, [&](auto slot_c, int i) {
    constexpr int cur = decltype(slot_c)::value, nxt = (cur + 2) % 3;
    const int t = t0 + i;
    sg::wait_vmcnt<P::PW>();
    sg::ring_barrier();
    dma_tile(std::integral_constant<int, nxt>{}, min(t + 2, t_last));
    f32x16 acc[RF];
#pragma unroll
    for (int f = 0; f < RF; ++f) acc[f] = sg::zero16();
    sg::mma_tile_asm<NKS, RF, cur * T::BYTES>(ra, frag, acc);
#pragma unroll
    for (int f = 0; f < RF; ++f) {
      uint32_t b = 0;
#pragma unroll
      for (int g = 0; g < 16; ++g) b |= acc[f][g] > thres ? (1u << ((g & 3) + 8 * (g >> 2))) : 0u;
      uint32_t w = b << (4 * half);
      w |= __shfl_xor(w, 32, 64);
      if (half == 0) fixw[(int64_t)t * n_rows_pad + row[f]] = w;       // rows >= n_rows are zero rows: w = 0, inside the padding
      if (half == 0 && w != 0u && any_out) atomicOr(reinterpret_cast<unsigned int*>(any_out + row[f]), 1u << (t >> any_shift));   // rare: token kernels skip rows without hits, and scan only the flagged tile groups
    }
  });
  sg::wait_vmcnt<0>();
}

// ------------------------------------------------------------------------------------------
// fused forward, false-negative test hoisted out (training path, default)
// ------------------------------------------------------------------------------------------
// cos(target, negative) > thres depends on the TARGET ROW, not on the token: at cfg1 every target row is the positive of up
// to P = 8 (position, offset) tokens per category, so nce_fwd_u_kernel evaluates the same f = p.n product up to 8 times.
// Here nce_fix_bits_kernel computes it once per (group, target row, negative) and leaves one bit per pair; this kernel
// then carries ONE stationary fragment set (the query: 64 VGPRs fewer), 16 instead of 32 MFMAs in the S phase, no compare
// chain - the tile's 32 suppression bits of a token arrive as one LDS-DMA word per lane, gathered by target row, riding the
// same ring (one more piece per tile) - and the saved per-token suppression words for nce_bwd_n are that very word.
template <int NKS, typename IT, bool LOGS, bool SUPP = true>
__global__ __launch_bounds__(256, 1) void nce_fwd_d_kernel(const IT* __restrict__ q_rows, const int32_t* q_idx,
                                                           const IT* __restrict__ p_rows, const int32_t* p_idx,
                                                           const bf16_t* negs, int n_neg,
                                                           const int32_t* n_tok_dev, int tok_cap,
                                                           const float* __restrict__ logit_scale_dev, float thres,
                                                           float* sum_out, int32_t* n_valid, int32_t* rank,
                                                           bf16_t* qn_out, bf16_t* pn_out,
                                                           uint32_t* supp_out, float* q_inv,
                                                           float* p_inv, float* s_pos_out, int log_group,
                                                           float* __restrict__ u_out,
                                                           const uint32_t* __restrict__ fixw, int n_rows_pad,
                                                           const int32_t* __restrict__ slot_of_row, int n_p_rows) {
  using T = sg::Tile<NKS>;
  constexpr int ND = (NKS + 1) / 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  {
    const int64_t grp = blockIdx.z, to = grp * tok_cap;
    q_idx += to; p_idx += to; n_tok_dev += grp; negs += grp * (int64_t)((n_neg + 31) & ~31) * T::DIM; sum_out += to;
    if (n_valid) n_valid += to;
    if (rank) rank += to;
    qn_out += to * T::DIM; u_out += to * T::DIM;
    if (pn_out) pn_out += to * T::DIM;
    q_inv += to; p_inv += to; s_pos_out += to;
    if constexpr (SUPP) {
      supp_out += grp * (int64_t)((n_neg + 31) >> 5) * tok_cap;
      fixw += grp * (int64_t)((n_neg + 31) >> 5) * n_rows_pad;
      if (slot_of_row) slot_of_row += grp * (int64_t)n_p_rows;
    }
  }
  const int n_tok = min(*n_tok_dev, tok_cap);
  const int tok0 = blockIdx.x * 128;
  if (tok0 >= n_tok) return;
  const int n_tiles = (n_neg + 31) >> 5;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 31, half = lane >> 5;
  const int tok = tok0 + wave * 32 + r;
  const bool live = tok < n_tok;
  const bool in_cap = tok < tok_cap;

  bf16x8 frag[1][NKS];   // the normalised query (the positive only feeds s+ and the saved state here)
  float qi = 0.f, pi = 0.f, spos = 0.f;
  int p_row = 0;
  {
    bf16x8 pfrag[NKS];
    p_row = live ? p_idx[tok] : 0;
    const IT* qs = q_rows + (live ? (int64_t)q_idx[tok] * T::DIM : 0);
    const IT* ps = p_rows + (int64_t)p_row * T::DIM;
    qi = row_inv_norm<NKS, IT>(qs, live, half);
    pi = row_inv_norm<NKS, IT>(ps, live, half);
    load_norm_frags<NKS, IT>(qs, live, half, qi, frag[0]);
    load_norm_frags<NKS, IT>(ps, live, half, pi, pfrag);
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
      for (int i = 0; i < 8; ++i) spos += (float)frag[0][ks][i] * (float)pfrag[ks][i];
    spos += __shfl_xor(spos, 32, 64);
    if (in_cap && pn_out) {
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks)
        *reinterpret_cast<bf16x8*>(pn_out + (int64_t)tok * T::DIM + ks * 16 + 8 * half) = pfrag[ks];
    }
  }
  // saved state; lanes past n_tok inside the block write zero rows / all-ones suppression words (see nce_fwd_kernel)
  if (in_cap) {
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const int k0 = ks * 16 + 8 * half;
      *reinterpret_cast<bf16x8*>(qn_out + (int64_t)tok * T::DIM + k0) = frag[0][ks];
    }
    if (live && half == 0) {
      q_inv[tok] = qi;
      p_inv[tok] = pi;
      s_pos_out[tok] = spos;
    }
  }
  const float scale = clamp_scale(logit_scale_dev);
  const float c1 = scale * LOG2E;
  const bool do_logs = LOGS && (log_group < 0 || (int)blockIdx.z == log_group);

  f32x16 u[ND];
#pragma unroll
  for (int dc = 0; dc < ND; ++dc) u[dc] = sg::zero16();
  float sum = 0.f;
  int nv = 0, rk = 0;

  using P = sg::DmaPieces<NKS>;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  P dp;
  dp.init(wv, lane);
  const int t_last = n_tiles - 1;
  auto dma_k = [&](auto k_c, auto slot_c, int tn) {
    dp.template piece<decltype(k_c)::value>(smem + decltype(slot_c)::value * T::BYTES,
                                             reinterpret_cast<const char*>(negs) + (int64_t)tn * (32 * T::ROW_BYTES));
  };
  unsigned char* words = smem + 4 * T::BYTES;                  // 4 slots x [4 waves][64 lanes] suppression words of my token's target
  const uint32_t* my_fix = SUPP ? fixw + (slot_of_row && live ? slot_of_row[p_row] : (slot_of_row ? 0 : p_row)) : nullptr;   // + tile * n_rows_pad
  auto dma_w = [&](auto slot_c, int tn) {
    if constexpr (SUPP) sg::dma_words(my_fix + (int64_t)tn * n_rows_pad, words + decltype(slot_c)::value * 1024 + wv * 256);
  };
  auto dma_all = [&](auto slot_c, int tn) {
    auto f = [&](auto k_c) { dma_k(k_c, slot_c, tn); };
    sg::static_for<P::PW>(f);
    dma_w(slot_c, tn);
  };
  constexpr int NDMA = P::PW + (SUPP ? 1 : 0);               // LDS-DMA instructions per tile and wave
  const uint32_t word_addr = sg::lds_addr(words) + (uint32_t)(wv * 256 + lane * 4);
  sg::LaneAddr<NKS> la;
  la.init(lane);
  sg::TrAddr<NKS> ta;
  ta.init(la, smem);
  sg::RowAddr<NKS> ra;
  ra.init(la, smem);
  // slot 3 is the "previous tile" of the first iteration (E = 0 there): make it finite
  for (int o = threadIdx.x * 16; o < T::BYTES; o += 256 * 16) *reinterpret_cast<f32x4*>(smem + 3 * T::BYTES + o) = f32x4{0.f, 0.f, 0.f, 0.f};
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // published by the prologue's ring barrier
  dma_all(std::integral_constant<int, 0>{}, 0);
  dma_all(std::integral_constant<int, 1>{}, min(1, t_last));
  // scalars first used inside the loop: touch them here, or hipcc places the `s_waitcnt lgkmcnt(0)` that covers their
  // kernel-argument load INSIDE the loop body, where it drains the LDS read pipeline once per tile
  asm volatile("" ::"s"(n_rows_pad), "s"(c1), "s"(tok_cap), "s"(n_neg));
  // The ring barrier sits in the MIDDLE of a tile step (sg::tile_step_pf): the barrier of step i publishes tile i + 1, whose
  // first row fragments (and suppression word) are requested from the last gaps of step i.  Prologue: tile 0 the same way.
  sg::wait_vmcnt<NDMA>();                               // my pieces of tile 0 (tile 1's may still be in flight)
  sg::ring_barrier();
  constexpr int PA = NKS < 4 ? NKS : 4;
  sg::u32x4 a_pf[PA + 1];
  uint32_t word = 0;                     // false negatives of the CURRENT tile for my token: bit j = negative 32 i + j suppressed
  if constexpr (SUPP) word = sg::ds_read_b32_asm<0>(word_addr);
  sg::prefetch_first<NKS, 0>(ra, a_pf);
  // S accumulators ping-pong by tile parity: tile t's epilogue runs one iteration later, straight from the other set
  f32x16 sf[2][1] = {{sg::zero16()}, {sg::zero16()}};
  uint32_t alive_prev = 0;               // live-row bits of the previous tile, pre-shifted by 4*half
  // the loop exists twice (with / without the rank + n_valid counting of the logged group): a per-element runtime test
  // would put a branch into every MFMA gap
  STAMP_DECL
  STAMP(-1)
  auto run = [&](auto logs_c) {
    constexpr bool WITH_LOGS = decltype(logs_c)::value;
    sg::ring_loop<4>(n_tiles + 1, [&](auto slot_c, int i) {
      constexpr int cur = decltype(slot_c)::value, nx1 = (cur + 1) % 4, nxt = (cur + 2) % 4, prv = (cur + 3) % 4, par = cur & 1;
      STAMP(2)
      const int tn = min(i + 2, t_last);
      sf[par][0] = sg::zero16();
      const f32x16& s_prev = sf[par ^ 1][0];
      // (SUPP = false: nothing is suppressed - the row-sharing path, whose per-token kernels take the false negatives back
      //  out - so no words, no bit test; the 'previous tile' of the first iteration is switched off through the exponent)
      const float c0t = (!SUPP && i == 0) ? INFINITY : c1;
      uint32_t word_next = 0;
      sg::tile_step_pf<NKS, ND, cur * T::BYTES, prv * T::BYTES, nx1 * T::BYTES, NDMA, SUPP ? 1 : 0>(
          ra, ta, frag, sf[par], u, a_pf,
          [&](auto n_c) {
            if constexpr (SUPP) sg::wait_lgkm_values<decltype(n_c)::value>(word);
          },
          [&](int g) {                     // gated logit of the previous tile: alive_prev already excludes its false negatives
            float ek;
            if constexpr (SUPP) ek = gate_alive(s_prev[g], c1, c1, alive_prev, (g & 3) + 8 * (g >> 2));
            else ek = gate_plain(s_prev[g], c1, c0t);
            asm volatile("v_add_f32 %0, %0, %1" : "+v"(sum) : "v"(ek));      // volatile: keep the accumulation in this gap
            if constexpr (WITH_LOGS) {
              const int pos = (g & 3) + 8 * (g >> 2);
              const int km = ((int)(alive_prev << (31 - pos))) >> 31;       // -1: a kept, live logit
              nv -= km;
              rk -= s_prev[g] > spos ? km : 0;
            }
            return ek;
          },
          [&](auto k_c) {                  // gaps of the second sweep: tile i + 2's DMA (tile pieces, then the words)
            constexpr int k = decltype(k_c)::value;
            if constexpr (k < P::PW) dma_k(k_c, std::integral_constant<int, nxt>{}, tn);
            else if constexpr (k == P::PW && SUPP) dma_w(std::integral_constant<int, nxt>{}, tn);
          },
          [&] {                            // between the products: tile i + 1 has landed (mine), then everybody's
            STAMP(3)
            sg::wait_vmcnt<0>();
            STAMP(0)
            sg::ring_barrier();
            STAMP(1)
          },
          [&] {                            // my word of tile i + 1, requested right before its first row fragments
            if constexpr (SUPP) word_next = sg::ds_read_b32_asm<nx1 * 1024>(word_addr);
          });
      STAMP(4)
      if constexpr (SUPP) {
        if (i < n_tiles && in_cap && half == 0) supp_out[(int64_t)i * tok_cap + tok] = live ? word : 0xFFFFFFFFu;
      }
      const uint32_t sbits = (word >> (4 * half)) & 0x0F0F0F0Fu;
      const int rem = n_neg - i * 32;
      const uint32_t tail = rem >= 32 ? 0xFFFFFFFFu : (rem > 0 ? ~(0xFFFFFFFFu << rem) : 0u);
      alive_prev = (live && i < n_tiles) ? ((tail >> (4 * half)) & ~sbits) : 0u;
      word = word_next;
    });
  };
  if (do_logs) run(std::true_type{});
  else run(std::false_type{});
  STAMP(2)
  STAMP_FLUSH
  sg::wait_vmcnt<0>();

  sum += __shfl_xor(sum, 32, 64);
  if (LOGS) {
    nv += __shfl_xor(nv, 32, 64);
    rk += __shfl_xor(rk, 32, 64);
  }
  if (live && half == 0) {
    atomicAdd(sum_out + tok, sum);
    if (do_logs) {
      if (n_valid) atomicAdd(n_valid + tok, nv);
      if (rank) atomicAdd(rank + tok, rk);
    }
  }
  // U: rows (regs) = tokens wave*32 + crow(g, half), columns (lanes) = features dc*32 + r
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    const int tk = tok0 + wave * 32 + sg::crow(g, half);
    if (tk < n_tok) {
#pragma unroll
      for (int dc = 0; dc < ND; ++dc) {
        const int d = dc * 32 + r;
        if (d < T::DIM) u_out[(int64_t)tk * T::DIM + d] = u[dc][g];
      }
    }
  }
}

// lse / loss from the partial sums of all negative ranges
constexpr int MAX_BUCKETS = 64;
__global__ __launch_bounds__(256) void nce_finalize_kernel(const float* sum, const float* s_pos,
                                                           const int32_t* __restrict__ n_tok_dev, int tok_cap,
                                                           const float* __restrict__ logit_scale_dev,
                                                           float* loss, float* lse, int32_t* n_valid,
                                                           const int32_t* __restrict__ bucket_idx, int n_buckets,
                                                           float* __restrict__ bucket_sum, float* __restrict__ bucket_cnt) {
  const int64_t to = (int64_t)blockIdx.y * tok_cap;
  sum += to; s_pos += to; loss += to; lse += to;
  if (n_valid) n_valid += to;
  __shared__ float s_sum[MAX_BUCKETS], s_cnt[MAX_BUCKETS];
  if (bucket_idx) {
    for (int b = threadIdx.x; b < n_buckets; b += blockDim.x) s_sum[b] = s_cnt[b] = 0.f;
    __syncthreads();
  }
  const int n_tok = min(n_tok_dev[blockIdx.y], tok_cap);
  const int tok = blockIdx.x * blockDim.x + threadIdx.x;
  if (tok < n_tok) {
    const float scale = clamp_scale(logit_scale_dev);
    const float c1 = scale * LOG2E;
    const float sp = s_pos[tok];
    const float total = sum[tok] + fast_exp2(sp * c1 - c1);
    const float l = scale + __logf(total);
    lse[tok] = l;
    loss[tok] = l - scale * sp;
    if (n_valid) n_valid[tok] += 1;        // the positive itself (hstu.py:622: logits > finfo.min / 100)
    if (bucket_idx) {                      // per-(group, prediction offset) sums: the loss is a mean per offset (hstu.py:697-700)
      const int b = bucket_idx[to + tok];
      if (b >= 0 && b < n_buckets) {
        atomicAdd(&s_sum[b], l - scale * sp);
        atomicAdd(&s_cnt[b], 1.0f);
      }
    }
  }
  if (bucket_idx) {
    __syncthreads();
    for (int b = threadIdx.x; b < n_buckets; b += blockDim.x)
      if (s_cnt[b] != 0.f) {
        atomicAdd(bucket_sum + blockIdx.y * n_buckets + b, s_sum[b]);
        atomicAdd(bucket_cnt + blockIdx.y * n_buckets + b, s_cnt[b]);
      }
  }
}

// ------------------------------------------------------------------------------------------
// backward, token-stationary: dq_rows, dp_rows, d_logit_scale
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void pack_acc(const f32x16& x, bf16x8& f0, bf16x8& f1) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    f0[j] = (bf16_t)x[j];
    f1[j] = (bf16_t)x[8 + j];
  }
}

__device__ __forceinline__ float half_sum(float v) {   // sum over the 32 lanes of one wave half
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Row-wise remainder of the token-side backward: dQn_i = w_i exp(scale - lse_i) U_i (U from nce_fwd_u_kernel), the
// positive-pair term, the L2-normalisation chain rule, d(logit_scale), and the accumulation into the shared source
// rows.  One wave per token, four tokens in flight per wave; lanes cover the feature dim 64 at a time (256-B float
// atomic segments).  Also writes lw = lse log2(e) - log2(w) for nce_bwd_n.
__global__ __launch_bounds__(256) void nce_bwd_rows_kernel(const bf16_t* qn, const bf16_t* pn, const float* u, int dim,
                                                           const int32_t* n_tok_dev, int tok_cap,
                                                           const float* __restrict__ logit_scale_dev,
                                                           const float* lse, const float* w,
                                                           const float* q_inv, const float* p_inv,
                                                           const float* s_pos, const int32_t* q_idx,
                                                           const int32_t* p_idx, float* __restrict__ dq_rows,
                                                           float* __restrict__ dp_rows, float* __restrict__ d_logit_scale,
                                                           float* __restrict__ lw_out, const int32_t* __restrict__ w_bucket,
                                                           int n_buckets, long long* __restrict__ dq_fix,
                                                           long long* __restrict__ dp_fix, float* __restrict__ dls_part) {
  // dq_fix / dp_fix / dls_part (deterministic mode, include/mhr.h): fixed-point accumulators shadowing dq_rows / dp_rows and one
  // d(logit_scale) partial per wave - what would be float atomics in arrival order becomes order-independent
  constexpr int TB = 8;            // consecutive tokens per wave pass (loads in flight; runs sharing a head row are combined)
  constexpr int NC = 4;            // 64-column chunks (dim <= 256)
  {
    const int64_t grp = blockIdx.z, to = grp * tok_cap;
    qn += to * dim; pn += to * dim; u += to * dim;
    n_tok_dev += grp; lse += to; q_inv += to; p_inv += to; s_pos += to; q_idx += to; p_idx += to;
    if (lw_out) lw_out += to;
    if (w_bucket) { w_bucket += to; w += grp * n_buckets; } else { w += to; }
  }
  const int n_tok = min(*n_tok_dev, tok_cap);
  const int lane = threadIdx.x & 63;
  const int wave_g = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), n_waves = gridDim.x * (blockDim.x >> 6);
  const float scale = clamp_scale(logit_scale_dev);
  float dls = 0.f;
  for (int t0 = wave_g * TB; t0 < n_tok; t0 += n_waves * TB) {
    float qv[TB][NC], pv[TB][NC], uv[TB][NC];
    float wi[TB], sp[TB], ls[TB], iq[TB], ip[TB];
    int qi[TB], pi[TB];
#pragma unroll
    for (int b = 0; b < TB; ++b) {
      const int tk = t0 + b;
      const bool tl = tk < n_tok;
      wi[b] = tl ? (w_bucket ? w[w_bucket[tk]] : w[tk]) : 0.f;
      sp[b] = tl ? s_pos[tk] : 0.f;
      ls[b] = tl ? lse[tk] : 0.f;
      iq[b] = tl ? q_inv[tk] : 0.f;
      ip[b] = tl ? p_inv[tk] : 0.f;
      qi[b] = tl ? q_idx[tk] : 0;
      pi[b] = tl ? p_idx[tk] : 0;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int d = c * 64 + lane;
        const bool ok = tl && d < dim;
        qv[b][c] = ok ? (float)qn[(int64_t)tk * dim + d] : 0.f;
        pv[b][c] = ok ? (float)pn[(int64_t)tk * dim + d] : 0.f;
        uv[b][c] = ok ? u[(int64_t)tk * dim + d] : 0.f;
      }
    }
    // With the token lists ordered prediction-offset-fastest (multihead.py) consecutive tokens of a group share their
    // head row (the offsets p of one position): their dQ rows are summed in registers and leave as ONE 256-float atomic
    // set per run instead of one per token (the float-atomic rate bounds this kernel).
    float accq[NC] = {0.f, 0.f, 0.f, 0.f};
    int run_row = -1;
    auto flush = [&]() {
      if (run_row >= 0) {
        float* qdst = dq_rows + (int64_t)run_row * dim;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          const int d = c * 64 + lane;
          if (d < dim) {
            if (dq_fix) det_atomic_add(dq_fix + (int64_t)run_row * dim + d, accq[c]);
            else atomicAdd(qdst + d, accq[c]);
          }
        }
      }
    };
#pragma unroll
    for (int b = 0; b < TB; ++b) {
      const int tk = t0 + b;
      if (tk >= n_tok) break;
      const float a = wi[b] * __expf(scale - ls[b]);                          // G_ij = a * E_ij
      const float coef = wi[b] * (__expf(scale * sp[b] - ls[b]) - 1.0f);     // w (p_pos - 1)
      if (lw_out && lane == 0) lw_out[tk] = ls[b] * LOG2E - __log2f(wi[b]);
      float dqn[NC], dpn[NC];
      float dot_q = 0.f, dot_p = 0.f, dot_raw = 0.f;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const float raw = a * uv[b][c];
        dqn[c] = scale * (raw + coef * pv[b][c]);
        dpn[c] = scale * coef * qv[b][c];
        dot_raw += qv[b][c] * raw;
        dot_q += qv[b][c] * dqn[c];
        dot_p += pv[b][c] * dpn[c];
      }
      dot_q = wave_sum(dot_q);
      dot_p = wave_sum(dot_p);
      dls += wave_sum(dot_raw) + coef * sp[b];      // sum_j g_ij s_ij = qn_i . dQn_i, plus the positive term
      if (qi[b] != run_row) {                        // wave-uniform
        flush();
        run_row = qi[b];
#pragma unroll
        for (int c = 0; c < NC; ++c) accq[c] = 0.f;
      }
#pragma unroll
      for (int c = 0; c < NC; ++c) accq[c] += (dqn[c] - qv[b][c] * dot_q) * iq[b];
      // target rows (l + 1 + p = const) are shared too, but not by neighbours in the list: float atomics per token
      float* pdst = dp_rows + (int64_t)pi[b] * dim;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int d = c * 64 + lane;
        if (d < dim) {
          if (dp_fix) det_atomic_add(dp_fix + (int64_t)pi[b] * dim + d, (dpn[c] - pv[b][c] * dot_p) * ip[b]);
          else atomicAdd(pdst + d, (dpn[c] - pv[b][c] * dot_p) * ip[b]);
        }
      }
    }
    flush();
  }
  if (lane == 0 && d_logit_scale) {
    if (dls_part) dls_part[((int64_t)blockIdx.z * gridDim.x + blockIdx.x) * 4 + (threadIdx.x >> 6)] = dls;   // (x exp(param) at the fold)
    else if (dls != 0.f) atomicAdd(d_logit_scale, dls * scale);   // d/d(param), scale = exp(param)
  }
}

// ------------------------------------------------------------------------------------------
// backward, negative-stationary: d_negs
// ------------------------------------------------------------------------------------------
template <int NKS, bool SUPP = true>
__global__ __launch_bounds__(256, 1) void nce_bwd_n_kernel(const bf16_t* qn, const bf16_t* negs,
                                                           const uint32_t* supp, int n_neg,
                                                           const int32_t* n_tok_dev, int tok_cap,
                                                           const float* __restrict__ logit_scale_dev,
                                                           const float* lw, float* d_negs, long long* dn_fix) {
  using T = sg::Tile<NKS>;
  constexpr int ND = (NKS + 1) / 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // per ring slot: Q tile | [4 waves][64] words: lanes 0-31 = suppression words of that wave's negative tile for the
  // 32 tokens, lanes 32-63 = lw (= lse log2e - log2 w, written by nce_bwd_q) of the 32 tokens
  constexpr int BUF = T::BYTES + 1024;
  using P = sg::DmaPieces<NKS>;
  {
    const int64_t grp = blockIdx.z, to = grp * tok_cap;
    qn += to * T::DIM; negs += grp * (int64_t)((n_neg + 31) & ~31) * T::DIM;
    if constexpr (SUPP) supp += grp * (int64_t)((n_neg + 31) >> 5) * tok_cap;
    n_tok_dev += grp; lw += to; d_negs += grp * (int64_t)n_neg * T::DIM;
    if (dn_fix) dn_fix += grp * (int64_t)n_neg * T::DIM;
  }
  const int n_tok = min(*n_tok_dev, tok_cap);
  const int n_tok_tiles = (n_tok + 31) >> 5;
  const int neg0 = blockIdx.x * 128;
  // token tiles are dealt round-robin over gridDim.y so that every split gets an equal share of the LIVE tiles
  // whatever *n_tok_dev is (the host only knows the capacity)
  const int tt0 = blockIdx.y, tstep = gridDim.y, tt1 = n_tok_tiles;
  if (tt0 >= tt1) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 31, half = lane >> 5;
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  const int neg = neg0 + wave * 32 + r;                 // my stationary negative (lane column)
  const bool nlive = neg < n_neg;                       // dead negatives keep a zero fragment; their dN rows are not written
  const int n_neg_tiles = (n_neg + 31) >> 5;

  bf16x8 frag[1][NKS];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks)
    frag[0][ks] = nlive ? *reinterpret_cast<const bf16x8*>(negs + (int64_t)neg * T::DIM + ks * 16 + 8 * half) : sg::zero8();
  const float scale = clamp_scale(logit_scale_dev);
  const float c1 = scale * LOG2E;

  f32x16 dn[ND];
#pragma unroll
  for (int dc = 0; dc < ND; ++dc) dn[dc] = sg::zero16();

  // Token tiles stream like the negative tiles of nce_bwd_q (same branch-free ring).  tok_cap is a multiple of 32 and
  // the forward pads the last live 32-token tile (zero Qn rows, all-ones suppression words), so no row is ever
  // clamped and a padded token contributes exactly nothing.
  P dp;
  dp.init(wv, lane);
  const int my_nt = min((neg0 >> 5) + wv, n_neg_tiles - 1);
  // my word of a token tile: lanes 0-31 the suppression word of (my negative tile, token r), lanes 32-63 lw of token r
  // (SUPP = false - nothing suppressed, the row-sharing path: both halves fetch lw, and a padded token is switched off by
  //  lw = +inf, which mhr_nce_row_lw writes behind the live rows)
  const char* word_base = (SUPP && half == 0) ? reinterpret_cast<const char*>(supp + (int64_t)my_nt * tok_cap + r)
                                              : reinterpret_cast<const char*>(lw + r);
  const int n_loc = (tt1 - tt0 + tstep - 1) / tstep;
  const int tt_last = tt0 + (n_loc - 1) * tstep;
  auto dma_k = [&](auto k_c, auto slot_c, int tn) {
    constexpr int k = decltype(k_c)::value, slot = decltype(slot_c)::value;
    unsigned char* base = smem + slot * BUF;
    if constexpr (k < P::PW) dp.template piece<k>(base, reinterpret_cast<const char*>(qn) + (int64_t)tn * (32 * T::ROW_BYTES));
    else sg::dma_words(word_base + (int64_t)tn * 128, base + T::BYTES + wv * 256);
  };
  auto dma_all = [&](auto slot_c, int tn) {
    auto f = [&](auto k_c) { dma_k(k_c, slot_c, tn); };
    sg::static_for<P::PW + 1>(f);
  };
  sg::LaneAddr<NKS> la;
  la.init(lane);
  sg::TrAddr<NKS> ta;
  ta.init(la, smem);
  sg::RowAddr<NKS> ra;
  ra.init(la, smem);
  // my 16 accumulator rows are 4 runs of 4 consecutive tokens (8 q4 + 4 half + 0..3): their words come as 16-byte reads
  const uint32_t wd_addr = sg::lds_addr(smem) + T::BYTES + wv * 256 + 16 * half;
  // slot 3 is the "previous tile" of the first iteration: zero rows, all-ones suppression words (G = 0)
  for (int o = threadIdx.x * 16; o < T::BYTES; o += 256 * 16) *reinterpret_cast<f32x4*>(smem + 3 * BUF + o) = f32x4{0.f, 0.f, 0.f, 0.f};
  reinterpret_cast<uint32_t*>(smem + 3 * BUF + T::BYTES)[threadIdx.x] = SUPP ? 0xFFFFFFFFu : 0x7F800000u;   // (+inf as lw)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // published by the prologue's ring barrier
  dma_all(std::integral_constant<int, 0>{}, tt0);
  dma_all(std::integral_constant<int, 1>{}, min(tt0 + tstep, tt_last));
  // ring barrier in the MIDDLE of the tile step (sg::tile_step_pf, see nce_fwd_d_kernel): the barrier of step i publishes tile
  // i + 1; the per-token words of tile i (the epilogue of step i + 1 needs them) and the first row fragments of tile i + 1 are
  // requested from the last gaps of step i
  sg::wait_vmcnt<P::PW + 1>();
  sg::ring_barrier();
  constexpr int PA = NKS < 4 ? NKS : 4;
  sg::u32x4 a_pf[PA + 1];
  sg::u32x4 s4[4], l4[4];                // per-token words of the PREVIOUS tile: suppression words / lw
  auto rd_words = [&](auto slot_c) {
    constexpr int sl = decltype(slot_c)::value;
    auto rd = [&](auto q_c) {
      constexpr int q4 = decltype(q_c)::value;
      if constexpr (SUPP) s4[q4] = sg::ds_read_b128_asm<sl * BUF + 32 * q4>(wd_addr);
      l4[q4] = sg::ds_read_b128_asm<sl * BUF + 128 + 32 * q4>(wd_addr);
    };
    sg::static_for<4>(rd);
  };
  rd_words(std::integral_constant<int, 3>{});
  sg::prefetch_first<NKS, 0>(ra, a_pf);
  f32x16 s_prev = sg::zero16();
  sg::ring_loop<4>(n_loc + 1, [&](auto slot_c, int i) {
    constexpr int cur = decltype(slot_c)::value, nx1 = (cur + 1) % 4, nxt = (cur + 2) % 4, prv = (cur + 3) % 4;
    const int tn = min(tt0 + (i + 2) * tstep, tt_last);
    f32x16 accs[1] = {sg::zero16()};
    sg::tile_step_pf<NKS, ND, cur * BUF, prv * BUF, nx1 * BUF, P::PW + 1, SUPP ? 8 : 4>(
        ra, ta, frag, accs, dn, a_pf,
        [&](auto n_c) {
          if constexpr (SUPP) sg::wait_lgkm_values<decltype(n_c)::value>(s4[0], s4[1], s4[2], s4[3], l4[0], l4[1], l4[2], l4[3]);
          else sg::wait_lgkm_values<decltype(n_c)::value>(l4[0], l4[1], l4[2], l4[3]);
        },
        [&](int g) {   // bit r (my negative) of the token's suppression word
          // (whole-vector bit cast, then element: hipcc 7.2 folds element-extract + scalar bit cast of an asm result to
          //  element 0 - the same bug as the ds_read_tr builtin note in stream_gemm.h)
          const f32x4 lf = __builtin_bit_cast(f32x4, l4[g >> 2]);
          if constexpr (SUPP) return gate_dead(s_prev[g], c1, lf[g & 3], s4[g >> 2][g & 3], r);
          else return gate_plain(s_prev[g], c1, lf[g & 3]);
        },
        [&](auto k_c) {
          if constexpr (decltype(k_c)::value < P::PW + 1) dma_k(k_c, std::integral_constant<int, nxt>{}, tn);
        },
        [&] {                              // between the products: tile i + 1 has landed (mine), then everybody's
          sg::wait_vmcnt<0>();
          sg::ring_barrier();
        },
        [&] { rd_words(std::integral_constant<int, cur>{}); });     // the words of THIS tile, for the next step's epilogue
    s_prev = accs[0];
  });
  sg::wait_vmcnt<0>();
  // dn[dc][g]: row (reg) = negative neg0 + wave*32 + crow(g,half), column (lane) = feature dc*32 + r
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    const int nj = neg0 + wave * 32 + sg::crow(g, half);
    if (nj < n_neg) {
#pragma unroll
      for (int dc = 0; dc < ND; ++dc) {
        const int d = dc * 32 + r;
        if (d < T::DIM) {
          if (dn_fix) det_atomic_add(dn_fix + (int64_t)nj * T::DIM + d, scale * dn[dc][g]);     // order-independent (mhr.h)
          else atomicAdd(d_negs + (int64_t)nj * T::DIM + d, scale * dn[dc][g]);
        }
      }
    }
  }
}

inline bool nks_for(int dim, int& nks) {
  switch (dim) {
    case 16: nks = 1; return true;
    case 32: nks = 2; return true;
    case 64: nks = 4; return true;
    case 128: nks = 8; return true;
    case 256: nks = 16; return true;
    default: return false;
  }
}

}  // namespace

#define NKS_SWITCH(nks, MACRO) \
  switch (nks) {               \
    case 1: MACRO(1); break;   \
    case 2: MACRO(2); break;   \
    case 4: MACRO(4); break;   \
    case 8: MACRO(8); break;   \
    default: MACRO(16); break; \
  }

static inline int nce_splits(int n_tiles, int n_groups, int want, int& tiles_per_split) {
  // (token block, negative range, group) units should outnumber the ~512 workgroup slots several times; groups
  // already multiply the unit count, so fewer negative ranges are needed (each range repeats the prologue/epilogue)
  int splits = want / (n_groups > 0 ? n_groups : 1);
  if (splits < 1) splits = 1;
  while (splits > 1 && n_tiles / splits < 32) splits >>= 1;
  tiles_per_split = (n_tiles + splits - 1) / splits;
  return (n_tiles + tiles_per_split - 1) / tiles_per_split;
}

extern "C" int mhr_nce_fwd(const void* q_rows, const int32_t* q_idx, const void* p_rows, const int32_t* p_idx, int io_dtype,
                           const void* negs, int n_neg, int dim, int n_groups, const int32_t* n_tok_dev, int tok_cap,
                           const float* logit_scale_dev, float thres, float* sum_out, int32_t* n_valid, int32_t* rank,
                           void* qn_out, void* pn_out, uint32_t* supp_out, float* q_inv, float* p_inv, float* s_pos,
                           int log_group, float* u_out, int64_t n_p_rows, uint32_t* fix_words, const int32_t* fix_row_list,
                           const int32_t* fix_n_rows, int32_t* fix_slot_of_row, void* stream) {
  MHR_REQUIRE(q_rows && q_idx && p_rows && p_idx && negs && n_tok_dev && logit_scale_dev && sum_out && s_pos,
              "nce_fwd: null pointer");
  MHR_REQUIRE(!fix_words || (u_out && n_p_rows > 0 && n_p_rows < (1ll << 31) - 256), "nce_fwd: fix_words needs u_out and n_p_rows");
  MHR_REQUIRE((fix_row_list != nullptr) == (fix_n_rows != nullptr) && (fix_row_list != nullptr) == (fix_slot_of_row != nullptr) &&
                  (!fix_row_list || fix_words),
              "nce_fwd: fix_row_list, fix_n_rows and fix_slot_of_row go together (and need fix_words)");
  const bool plain = u_out && !fix_words && !supp_out;     // fused forward with NOTHING suppressed (row-sharing path)
  MHR_REQUIRE(!u_out || (qn_out && q_inv && p_inv && tok_cap % 32 == 0 && (plain || (pn_out && supp_out))),
              "nce_fwd: u_out (fused training path) needs every saved tensor and tok_cap %% 32 == 0");
  MHR_REQUIRE(!plain || n_neg % 32 == 0, "nce_fwd: the no-suppression form (supp_out = fix_words = NULL) needs n_neg %% 32 == 0");
  int nks;
  MHR_REQUIRE(nks_for(dim, nks), "nce_fwd: dim=%d unsupported (16/32/64/128/256)", dim);
  MHR_REQUIRE(n_neg > 0 && tok_cap > 0 && n_groups >= 1 && n_groups <= 65535, "nce_fwd: bad sizes");
  int tps;
  const int splits = nce_splits((n_neg + 31) / 32, n_groups, 4, tps);
  const dim3 grid((tok_cap + 127) / 128, splits, n_groups);
  hipStream_t s = (hipStream_t)stream;
  const bool logs = n_valid != nullptr || rank != nullptr;
#define ARGS(IT)                                                                                                         \
  (const IT*)q_rows, q_idx, (const IT*)p_rows, p_idx, (const bf16_t*)negs, n_neg, n_tok_dev, tok_cap, logit_scale_dev,   \
      thres, tps, sum_out, n_valid, rank, (bf16_t*)qn_out, (bf16_t*)pn_out, supp_out, q_inv, p_inv, s_pos, log_group
#define LU_(NKS)                                                                                                         \
  {                                                                                                                      \
    size_t lds = 4 * sg::Tile<NKS>::BYTES;                                                                               \
    const dim3 gu((tok_cap + 127) / 128, 1, n_groups);                                                                   \
    if (io_dtype == MHR_BF16) {                                                                                          \
      if (logs) hipLaunchKernelGGL((nce_fwd_u_kernel<NKS, bf16_t, true>), gu, dim3(256), lds, s, UARGS(bf16_t));         \
      else hipLaunchKernelGGL((nce_fwd_u_kernel<NKS, bf16_t, false>), gu, dim3(256), lds, s, UARGS(bf16_t));             \
    } else {                                                                                                             \
      if (logs) hipLaunchKernelGGL((nce_fwd_u_kernel<NKS, float, true>), gu, dim3(256), lds, s, UARGS(float));           \
      else hipLaunchKernelGGL((nce_fwd_u_kernel<NKS, float, false>), gu, dim3(256), lds, s, UARGS(float));               \
    }                                                                                                                    \
  }
#define UARGS(IT)                                                                                                        \
  (const IT*)q_rows, q_idx, (const IT*)p_rows, p_idx, (const bf16_t*)negs, n_neg, n_tok_dev, tok_cap, logit_scale_dev,   \
      thres, sum_out, n_valid, rank, (bf16_t*)qn_out, (bf16_t*)pn_out, supp_out, q_inv, p_inv, s_pos, log_group, u_out
  if (u_out && fix_words) {
    // false-negative bits once per (group, target row, negative), then the fused forward with one stationary operand
    const int n_tiles = (n_neg + 31) / 32, n_rows_pad = (int)((n_p_rows + 255) / 256 * 256);
    int slices = 1;                       // fill the 512 workgroup slots about twice over
    while (slices < 8 && (n_rows_pad / 256) * n_groups * slices < 768 && n_tiles / (slices * 2) >= 16) slices *= 2;
    const int tps_f = (n_tiles + slices - 1) / slices;
#define LF_(NKS)                                                                                                         \
  {                                                                                                                      \
    const dim3 gf(n_rows_pad / 256, slices, n_groups);                                                                   \
    const size_t ldsf = 3 * sg::Tile<NKS>::BYTES;                                                                        \
    const size_t ldsd = 4 * sg::Tile<NKS>::BYTES + 4 * 1024;                                                             \
    const dim3 gu((tok_cap + 127) / 128, 1, n_groups);                                                                   \
    if (io_dtype == MHR_BF16) {                                                                                          \
      hipLaunchKernelGGL((nce_fix_bits_kernel<NKS, bf16_t>), gf, dim3(256), ldsf, s, (const bf16_t*)p_rows, (int)n_p_rows, \
                         (const bf16_t*)negs, n_neg, thres, fix_words, n_rows_pad, tps_f, fix_row_list, fix_n_rows,     \
                         fix_slot_of_row, (int32_t*)nullptr);                                                            \
      if (logs) hipLaunchKernelGGL((nce_fwd_d_kernel<NKS, bf16_t, true>), gu, dim3(256), ldsd, s, UARGS(bf16_t), fix_words, n_rows_pad, fix_slot_of_row, (int)n_p_rows); \
      else hipLaunchKernelGGL((nce_fwd_d_kernel<NKS, bf16_t, false>), gu, dim3(256), ldsd, s, UARGS(bf16_t), fix_words, n_rows_pad, fix_slot_of_row, (int)n_p_rows);     \
    } else {                                                                                                             \
      hipLaunchKernelGGL((nce_fix_bits_kernel<NKS, float>), gf, dim3(256), ldsf, s, (const float*)p_rows, (int)n_p_rows, \
                         (const bf16_t*)negs, n_neg, thres, fix_words, n_rows_pad, tps_f, fix_row_list, fix_n_rows,     \
                         fix_slot_of_row, (int32_t*)nullptr);                                                            \
      if (logs) hipLaunchKernelGGL((nce_fwd_d_kernel<NKS, float, true>), gu, dim3(256), ldsd, s, UARGS(float), fix_words, n_rows_pad, fix_slot_of_row, (int)n_p_rows); \
      else hipLaunchKernelGGL((nce_fwd_d_kernel<NKS, float, false>), gu, dim3(256), ldsd, s, UARGS(float), fix_words, n_rows_pad, fix_slot_of_row, (int)n_p_rows);     \
    }                                                                                                                    \
  }
    NKS_SWITCH(nks, LF_);
#undef LF_
    MHR_CHECK_LAUNCH("nce_fwd (fused, hoisted false-negative test)");
    return MHR_OK;
  }
  if (plain) {
#define LP_(NKS)                                                                                                         \
  {                                                                                                                      \
    const size_t ldsd = 4 * sg::Tile<NKS>::BYTES + 4 * 1024;                                                             \
    const dim3 gu((tok_cap + 127) / 128, 1, n_groups);                                                                   \
    if (io_dtype == MHR_BF16) {                                                                                          \
      if (logs) hipLaunchKernelGGL((nce_fwd_d_kernel<NKS, bf16_t, true, false>), gu, dim3(256), ldsd, s, UARGS(bf16_t), (const uint32_t*)nullptr, 0, (const int32_t*)nullptr, 0); \
      else hipLaunchKernelGGL((nce_fwd_d_kernel<NKS, bf16_t, false, false>), gu, dim3(256), ldsd, s, UARGS(bf16_t), (const uint32_t*)nullptr, 0, (const int32_t*)nullptr, 0);     \
    } else {                                                                                                             \
      if (logs) hipLaunchKernelGGL((nce_fwd_d_kernel<NKS, float, true, false>), gu, dim3(256), ldsd, s, UARGS(float), (const uint32_t*)nullptr, 0, (const int32_t*)nullptr, 0); \
      else hipLaunchKernelGGL((nce_fwd_d_kernel<NKS, float, false, false>), gu, dim3(256), ldsd, s, UARGS(float), (const uint32_t*)nullptr, 0, (const int32_t*)nullptr, 0);     \
    }                                                                                                                    \
  }
    NKS_SWITCH(nks, LP_);
#undef LP_
    MHR_CHECK_LAUNCH("nce_fwd (fused, nothing suppressed)");
    return MHR_OK;
  }
  if (u_out) {
    NKS_SWITCH(nks, LU_);
    MHR_CHECK_LAUNCH("nce_fwd (fused)");
    return MHR_OK;
  }
#define L_(NKS)                                                                                                          \
  {                                                                                                                      \
    size_t lds = 3 * sg::Tile<NKS>::BYTES;                                                                               \
    if (io_dtype == MHR_BF16) {                                                                                          \
      if (logs) hipLaunchKernelGGL((nce_fwd_kernel<NKS, bf16_t, true>), grid, dim3(256), lds, s, ARGS(bf16_t));          \
      else hipLaunchKernelGGL((nce_fwd_kernel<NKS, bf16_t, false>), grid, dim3(256), lds, s, ARGS(bf16_t));              \
    } else {                                                                                                             \
      if (logs) hipLaunchKernelGGL((nce_fwd_kernel<NKS, float, true>), grid, dim3(256), lds, s, ARGS(float));            \
      else hipLaunchKernelGGL((nce_fwd_kernel<NKS, float, false>), grid, dim3(256), lds, s, ARGS(float));                \
    }                                                                                                                    \
  }
  NKS_SWITCH(nks, L_);
#undef L_
#undef LU_
#undef ARGS
#undef UARGS
  MHR_CHECK_LAUNCH("nce_fwd");
  return MHR_OK;
}

extern "C" int mhr_nce_fix_bits(const void* p_rows, int io_dtype, int64_t n_p_rows, const void* negs, int n_neg, int dim,
                                int n_groups, float thres, uint32_t* fix_words, const int32_t* fix_row_list,
                                const int32_t* fix_n_rows, int32_t* fix_slot_of_row, int32_t* fix_any, void* stream) {
  MHR_REQUIRE(p_rows && negs && fix_words, "nce_fix_bits: null pointer");
  MHR_REQUIRE((fix_row_list != nullptr) == (fix_n_rows != nullptr) && (fix_n_rows != nullptr) == (fix_slot_of_row != nullptr),
              "nce_fix_bits: fix_row_list, fix_n_rows and fix_slot_of_row go together");
  int nks;
  MHR_REQUIRE(nks_for(dim, nks), "nce_fix_bits: dim=%d unsupported (16/32/64/128/256)", dim);
  MHR_REQUIRE(n_neg > 0 && n_p_rows > 0 && n_groups >= 1 && n_groups <= 65535, "nce_fix_bits: bad sizes");
  hipStream_t s = (hipStream_t)stream;
  const int n_tiles = (n_neg + 31) / 32, n_rows_pad = (int)((n_p_rows + 255) / 256 * 256);
  int slices = 1;                       // fill the 512 workgroup slots about twice over
  while (slices < 8 && (n_rows_pad / 256) * n_groups * slices < 768 && n_tiles / (slices * 2) >= 16) slices *= 2;
  const int tps_f = (n_tiles + slices - 1) / slices;
#define LFB_(NKS)                                                                                                        \
  {                                                                                                                      \
    const dim3 gf(n_rows_pad / 256, slices, n_groups);                                                                   \
    const size_t ldsf = 3 * sg::Tile<NKS>::BYTES;                                                                        \
    if (io_dtype == MHR_BF16)                                                                                            \
      hipLaunchKernelGGL((nce_fix_bits_kernel<NKS, bf16_t>), gf, dim3(256), ldsf, s, (const bf16_t*)p_rows, (int)n_p_rows, \
                         (const bf16_t*)negs, n_neg, thres, fix_words, n_rows_pad, tps_f, fix_row_list, fix_n_rows,     \
                         fix_slot_of_row, fix_any);                                                                      \
    else                                                                                                                 \
      hipLaunchKernelGGL((nce_fix_bits_kernel<NKS, float>), gf, dim3(256), ldsf, s, (const float*)p_rows, (int)n_p_rows, \
                         (const bf16_t*)negs, n_neg, thres, fix_words, n_rows_pad, tps_f, fix_row_list, fix_n_rows,     \
                         fix_slot_of_row, fix_any);                                                                      \
  }
  NKS_SWITCH(nks, LFB_);
#undef LFB_
  MHR_CHECK_LAUNCH("nce_fix_bits");
  return MHR_OK;
}

// Deterministic form of the per-(group, offset) loss sums (mhr_set_deterministic): ONE workgroup per group; thread t adds the losses of
// tokens t, t + 256, ... into its own column of an LDS table (in token order), then every bucket's 256 column sums are folded in
// a fixed order - no atomics.  The regular kernel adds block partials with float atomics (LDS, then global).
constexpr int DET_MAX_BUCKETS = 32;
__global__ __launch_bounds__(256) void nce_bucket_sums_det_kernel(const float* __restrict__ loss, const int32_t* __restrict__ bucket_idx,
                                                                  const int32_t* __restrict__ n_tok_dev, int tok_cap, int n_buckets,
                                                                  float* __restrict__ bucket_sum, float* __restrict__ bucket_cnt) {
  extern __shared__ float sp[];                               // [n_buckets][256] sums, then [n_buckets][256] counts
  float* cp = sp + n_buckets * 256;
  const int g = blockIdx.x, tid = threadIdx.x;
  const int64_t to = (int64_t)g * tok_cap;
  const int n_tok = min(n_tok_dev[g], tok_cap);
  for (int b = 0; b < n_buckets; ++b) sp[b * 256 + tid] = cp[b * 256 + tid] = 0.f;
  for (int tk = tid; tk < n_tok; tk += 256) {
    const int b = bucket_idx[to + tk];
    if (b >= 0 && b < n_buckets) {
      sp[b * 256 + tid] += loss[to + tk];
      cp[b * 256 + tid] += 1.0f;
    }
  }
  __syncthreads();
  const int wave = tid >> 6, lane = tid & 63;
  for (int b = wave; b < n_buckets; b += 4) {
    float s4 = ((sp[b * 256 + lane] + sp[b * 256 + 64 + lane]) + (sp[b * 256 + 128 + lane] + sp[b * 256 + 192 + lane]));
    float c4 = ((cp[b * 256 + lane] + cp[b * 256 + 64 + lane]) + (cp[b * 256 + 128 + lane] + cp[b * 256 + 192 + lane]));
    s4 = wave_sum(s4);
    c4 = wave_sum(c4);
    if (lane == 0 && c4 != 0.f) {
      bucket_sum[g * n_buckets + b] += s4;
      bucket_cnt[g * n_buckets + b] += c4;
    }
  }
}

extern "C" int mhr_nce_finalize(const float* sum, const float* s_pos, int n_groups, const int32_t* n_tok_dev, int tok_cap,
                                const float* logit_scale_dev, float* loss, float* lse, int32_t* n_valid,
                                const int32_t* bucket_idx, int n_buckets, float* bucket_sum, float* bucket_cnt, void* stream) {
  MHR_REQUIRE(sum && s_pos && n_tok_dev && logit_scale_dev && loss && lse, "nce_finalize: null pointer");
  MHR_REQUIRE(tok_cap > 0 && n_groups >= 1 && n_groups <= 65535, "nce_finalize: bad sizes");
  MHR_REQUIRE(!bucket_idx || (bucket_sum && bucket_cnt && n_buckets >= 1 && n_buckets <= MAX_BUCKETS),
              "nce_finalize: bucket sums need bucket_sum, bucket_cnt and 1 <= n_buckets <= %d", MAX_BUCKETS);
  const bool det = bucket_idx && mhr_deterministic();
  MHR_REQUIRE(!det || n_buckets <= DET_MAX_BUCKETS, "nce_finalize: deterministic mode supports at most %d buckets", DET_MAX_BUCKETS);
  hipLaunchKernelGGL(nce_finalize_kernel, dim3((tok_cap + 255) / 256, n_groups), dim3(256), 0, (hipStream_t)stream, sum, s_pos,
                     n_tok_dev, tok_cap, logit_scale_dev, loss, lse, n_valid, det ? nullptr : bucket_idx, n_buckets, bucket_sum,
                     bucket_cnt);
  MHR_CHECK_LAUNCH("nce_finalize");
  if (det) {
    hipLaunchKernelGGL(nce_bucket_sums_det_kernel, dim3(n_groups), dim3(256), (size_t)2 * n_buckets * 256 * sizeof(float),
                       (hipStream_t)stream, loss, bucket_idx, n_tok_dev, tok_cap, n_buckets, bucket_sum, bucket_cnt);
    MHR_CHECK_LAUNCH("nce_finalize (deterministic bucket sums)");
  }
  return MHR_OK;
}

extern "C" int mhr_nce_bwd_tokens(const void* qn, const void* pn, const float* u, int dim, int n_groups,
                                  const int32_t* n_tok_dev, int tok_cap, const float* logit_scale_dev, const float* lse,
                                  const float* w, const float* q_inv, const float* p_inv, const float* s_pos,
                                  const int32_t* q_idx, const int32_t* p_idx, float* dq_rows, float* dp_rows,
                                  float* d_logit_scale, float* lw_out, const int32_t* w_bucket, int n_buckets, int64_t* dq_fix,
                                  int64_t* dp_fix, float* dls_part, void* stream) {
  MHR_REQUIRE(qn && pn && u && n_tok_dev && logit_scale_dev && lse && w && q_inv && p_inv && s_pos,
              "nce_bwd_tokens: null input pointer");
  MHR_REQUIRE(q_idx && p_idx && dq_rows && dp_rows, "nce_bwd_tokens: null index/output pointer");
  MHR_REQUIRE(dim > 0 && dim <= 256, "nce_bwd_tokens: dim=%d unsupported (<= 256)", dim);
  MHR_REQUIRE(tok_cap > 0 && n_groups >= 1 && n_groups <= 65535, "nce_bwd_tokens: bad sizes");
  int blocks = (tok_cap + 31) / 32;                 // 4 waves x 8 tokens per pass
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(nce_bwd_rows_kernel, dim3(blocks, 1, n_groups), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)qn,
                     (const bf16_t*)pn, u, dim, n_tok_dev, tok_cap, logit_scale_dev, lse, w, q_inv, p_inv, s_pos, q_idx, p_idx,
                     dq_rows, dp_rows, d_logit_scale, lw_out, w_bucket, n_buckets, (long long*)dq_fix, (long long*)dp_fix, dls_part);
  MHR_CHECK_LAUNCH("nce_bwd_tokens");
  return MHR_OK;
}

extern "C" int mhr_nce_bwd_negs(const void* qn, const void* negs, const uint32_t* supp, int n_neg, int dim, int n_groups,
                                const int32_t* n_tok_dev, int tok_cap, const float* logit_scale_dev, const float* lw,
                                float* d_negs, int64_t* dn_fix, void* stream) {
  MHR_REQUIRE(qn && negs && n_tok_dev && logit_scale_dev && lw && d_negs, "nce_bwd_negs: null pointer");
  MHR_REQUIRE(tok_cap % 32 == 0, "nce_bwd_negs: tok_cap=%d must be a multiple of 32", tok_cap);
  int nks;
  MHR_REQUIRE(nks_for(dim, nks), "nce_bwd_negs: dim=%d unsupported (16/32/64/128/256)", dim);
  MHR_REQUIRE(n_neg > 0 && tok_cap > 0 && n_groups >= 1 && n_groups <= 65535, "nce_bwd_negs: bad sizes");
  hipStream_t s = (hipStream_t)stream;
  const int n_tok_tiles = (tok_cap + 31) / 32;
  const int neg_groups = (n_neg + 127) / 128;
  int splits = (2048 / n_groups + neg_groups - 1) / neg_groups;   // ~2048 workgroups in all, token tiles dealt round-robin
  if (splits > n_tok_tiles) splits = n_tok_tiles;
  if (splits < 1) splits = 1;
#define L_(NKS)                                                                                                        \
  {                                                                                                                    \
    size_t lds_n = 4 * (sg::Tile<NKS>::BYTES + 1024);                                                                  \
    if (supp)                                                                                                          \
      hipLaunchKernelGGL((nce_bwd_n_kernel<NKS, true>), dim3(neg_groups, splits, n_groups), dim3(256), lds_n, s,       \
                         (const bf16_t*)qn, (const bf16_t*)negs, supp, n_neg, n_tok_dev, tok_cap, logit_scale_dev, lw, d_negs, \
                         (long long*)dn_fix);                                                                          \
    else                                                                                                               \
      hipLaunchKernelGGL((nce_bwd_n_kernel<NKS, false>), dim3(neg_groups, splits, n_groups), dim3(256), lds_n, s,      \
                         (const bf16_t*)qn, (const bf16_t*)negs, supp, n_neg, n_tok_dev, tok_cap, logit_scale_dev, lw, d_negs, \
                         (long long*)dn_fix);                                                                          \
  }
  NKS_SWITCH(nks, L_);
#undef L_
  MHR_CHECK_LAUNCH("nce_bwd_negs");
  return MHR_OK;
}

#ifdef MHR_STAMP
extern "C" int mhr_debug_read_stamps(unsigned long long* host16) {
  return (int)hipMemcpyFromSymbol(host16, HIP_SYMBOL(g_stamps), 16 * sizeof(unsigned long long));
}
#endif
