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
                                                         float* p_inv, float* s_pos_out) {
  using T = sg::Tile<NKS>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // blockIdx.z selects the group (negative pool / prior category): every per-token array is [groups, tok_cap(, D)],
  // negatives are [groups, n_neg, D]; query / positive source rows are shared.
  {
    const int64_t grp = blockIdx.z, to = grp * tok_cap;
    q_idx += to; p_idx += to; n_tok_dev += grp; negs += grp * (int64_t)n_neg * T::DIM; sum_out += to;
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
  if (live && first_split) {
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const int k0 = ks * 16 + 8 * half;
      if (qn_out) *reinterpret_cast<bf16x8*>(qn_out + (int64_t)tok * T::DIM + k0) = frag[0][ks];
      if (pn_out) *reinterpret_cast<bf16x8*>(pn_out + (int64_t)tok * T::DIM + k0) = frag[1][ks];
    }
    if (half == 0) {
      if (q_inv) q_inv[tok] = qi;
      if (p_inv) p_inv[tok] = pi;
      if (s_pos_out) s_pos_out[tok] = spos;
    }
  }

  const float scale = clamp_scale(logit_scale_dev);
  const float c1 = scale * LOG2E;
  float sum = 0.f;
  int nv = 0, rk = 0;

  // negative tiles stream through a 3-deep LDS-DMA ring (stream_gemm.h); rows past n_neg are clamped (masked by `rem`)
  using D = sg::Dma<NKS>;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  auto row_ptr_for = [&](int tile) {
    return [=](int rr) -> const bf16_t* { return negs + (int64_t)min(tile * 32 + rr, n_neg - 1) * T::DIM; };
  };
  sg::LaneAddr<NKS> la;
  la.init(lane);
  const int n_loc = t_end - t_begin;
  D::issue(smem, row_ptr_for(t_begin), wv, lane);
  if (n_loc > 1) D::issue(smem + T::BYTES, row_ptr_for(t_begin + 1), wv, lane);
  sg::ring_loop<3>(n_loc, [&](auto slot_c, int i) {
    constexpr int cur = decltype(slot_c)::value, nxt = (cur + 2) % 3;
    const int t = t_begin + i;
    if (i + 1 < n_loc) sg::wait_vmcnt<D::PW>(); else sg::wait_vmcnt<0>();
    sg::ring_barrier();
    if (i + 2 < n_loc) D::issue(smem + nxt * T::BYTES, row_ptr_for(t + 2), wv, lane);
    f32x16 acc[2] = {sg::zero16(), sg::zero16()};
    sg::mma_tile<NKS, 2>(smem + cur * T::BYTES, la, frag, acc);
    const int rem = n_neg - t * 32;
    uint32_t sbits = 0;
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const bool supp = acc[1][g] > thres;               // false negative: cos(positive, negative) > thres
      const bool keep = (sg::crow(g, half) < rem) && !supp;
      const float e = fast_exp2(acc[0][g] * c1 - c1);
      sum += keep ? e : 0.f;
      sbits |= supp ? (1u << sg::crow(g, half)) : 0u;
      if (LOGS) {
        nv += keep ? 1 : 0;
        rk += (keep && acc[0][g] > spos) ? 1 : 0;
      }
    }
    if (supp_out) {                                       // one word per (negative tile, token): bit j = negative t*32+j suppressed
      sbits |= __shfl_xor(sbits, 32, 64);
      if (live && half == 0) supp_out[(int64_t)t * tok_cap + tok] = sbits;
    }
  });
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

// lse / loss from the partial sums of all negative ranges
__global__ __launch_bounds__(256) void nce_finalize_kernel(const float* sum, const float* s_pos,
                                                           const int32_t* __restrict__ n_tok_dev, int tok_cap,
                                                           const float* __restrict__ logit_scale_dev,
                                                           float* loss, float* lse, int32_t* n_valid) {
  const int64_t to = (int64_t)blockIdx.y * tok_cap;
  sum += to; s_pos += to; loss += to; lse += to;
  if (n_valid) n_valid += to;
  const int n_tok = min(n_tok_dev[blockIdx.y], tok_cap);
  const int tok = blockIdx.x * blockDim.x + threadIdx.x;
  if (tok >= n_tok) return;
  const float scale = clamp_scale(logit_scale_dev);
  const float c1 = scale * LOG2E;
  const float sp = s_pos[tok];
  const float total = sum[tok] + fast_exp2(sp * c1 - c1);
  const float l = scale + __logf(total);
  lse[tok] = l;
  loss[tok] = l - scale * sp;
  if (n_valid) n_valid[tok] += 1;        // the positive itself (hstu.py:622: logits > finfo.min / 100)
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

template <int NKS>
__global__ __launch_bounds__(256, 1) void nce_bwd_q_kernel(const bf16_t* qn, const bf16_t* pn,
                                                           const bf16_t* negs, const uint32_t* supp,
                                                           int n_neg, const int32_t* n_tok_dev, int tok_cap,
                                                           int tiles_per_split, const float* __restrict__ logit_scale_dev,
                                                           const float* lse, const float* w,
                                                           const float* q_inv, const float* p_inv,
                                                           const float* s_pos, const int32_t* q_idx,
                                                           const int32_t* p_idx, float* __restrict__ dq_rows,
                                                           float* __restrict__ dp_rows, float* __restrict__ d_logit_scale) {
  using T = sg::Tile<NKS>;
  constexpr int ND = (NKS + 1) / 2;          // 32-column chunks of the feature dim
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* tiles = smem;               // 4 x T::BYTES, then 4 x 1 KiB of suppression words
  {
    const int64_t grp = blockIdx.z, to = grp * tok_cap;
    qn += to * T::DIM; pn += to * T::DIM; negs += grp * (int64_t)n_neg * T::DIM;
    supp += grp * (int64_t)((n_neg + 31) >> 5) * tok_cap;
    n_tok_dev += grp; lse += to; w += to; q_inv += to; p_inv += to; s_pos += to; q_idx += to; p_idx += to;
  }
  const int n_tok = min(*n_tok_dev, tok_cap);
  const int tok0 = blockIdx.x * 128;
  if (tok0 >= n_tok) return;
  const int n_tiles = (n_neg + 31) >> 5;
  const int t_begin = blockIdx.y * tiles_per_split, t_end = min(n_tiles, t_begin + tiles_per_split);
  if (t_begin >= t_end) return;
  const bool first_split = blockIdx.y == 0;       // owns the positive-pair terms
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 31, half = lane >> 5;
  const int tok = tok0 + wave * 32 + r;
  const bool live = tok < n_tok;

  bf16x8 frag[1][NKS];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks)
    frag[0][ks] = live ? *reinterpret_cast<const bf16x8*>(qn + (int64_t)tok * T::DIM + ks * 16 + 8 * half) : sg::zero8();
  const float scale = clamp_scale(logit_scale_dev);
  const float c1 = scale * LOG2E;
  const float my_w = live ? w[tok] : 0.f;
  const float my_l2 = live ? lse[tok] * LOG2E : 0.f;

  f32x16 dq[ND];
#pragma unroll
  for (int dc = 0; dc < ND; ++dc) dq[dc] = sg::zero16();
  float dsc = 0.f;   // sum_j g_ij * s_ij for my token (my half's rows)

  // negative tiles + this wave's 32 suppression words per tile stream through a 3-deep LDS-DMA ring
  using D = sg::Dma<NKS>;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned char* words = smem + 4 * T::BYTES;                       // 4 x [4 waves][64] uint32
  auto row_ptr_for = [&](int tile) {
    return [=](int rr) -> const bf16_t* { return negs + (int64_t)min(tile * 32 + rr, n_neg - 1) * T::DIM; };
  };
  const int tokc = min(tok, tok_cap - 1);
  auto issue_all = [&](int buf, int tile) {
    D::issue(tiles + buf * T::BYTES, row_ptr_for(tile), wv, lane);
    sg::dma_words(supp + (int64_t)tile * tok_cap + tokc, words + buf * 1024 + wv * 256);
  };
  // Software pipeline, one-tile skew, 4-slot ring: the 16 MFMAs of S(t) are issued with the VALU epilogue of S(t-1)
  // placed element by element in their gaps (mma_tile_epi), then dQ += G(t-1) . N(t-1) runs on tile t-1, which is
  // still resident.  (Letting hipcc interleave with sched_group_barrier hints, or pairing the product with the
  // epilogue of the same iteration, measured 5-15 % slower than no skew at all.)
  sg::LaneAddr<NKS> la;
  la.init(lane);
  const int n_loc = t_end - t_begin;
  issue_all(0, t_begin);
  if (n_loc > 1) issue_all(1, t_begin + 1);
  f32x16 s_prev = sg::zero16();          // S of the previous tile
  uint32_t dead_prev = 0xFFFFFFFFu;      // ... and its dead-row mask (all dead before the first tile: G = 0)
  // one extra iteration drains the pipeline (its MFMAs run on a stale but finite tile and are discarded)
  sg::ring_loop<4>(n_loc + 1, [&](auto slot_c, int i) {
    constexpr int cur = decltype(slot_c)::value, nxt = (cur + 2) % 4, prv = (cur + 3) % 4;
    const int t = t_begin + i;
    const bool has_tile = i < n_loc;
    if (has_tile) {
      if (i + 1 < n_loc) sg::wait_vmcnt<D::PW + 1>(); else sg::wait_vmcnt<0>();
    }
    sg::ring_barrier();
    if (i + 2 < n_loc) issue_all(nxt, t + 2);
    const unsigned char* tile = tiles + cur * T::BYTES;
    const uint32_t sw = (live && has_tile) ? reinterpret_cast<const uint32_t*>(words + cur * 1024 + wv * 256)[lane] : 0u;
    const int rem = n_neg - t * 32;
    // rows past n_neg and suppressed pairs: one 32-bit "dead" mask per lane, tested branch-free
    const uint32_t dead = has_tile ? (sw | (rem >= 32 ? 0u : (0xFFFFFFFFu << (rem > 0 ? rem : 0)))) : 0xFFFFFFFFu;
    f32x16 acc = sg::zero16();
    f32x16 gacc;
    sg::mma_tile_epi<NKS, 4>(tile, la, frag, acc, [&](int g) {
      const float e = my_w * fast_exp2(s_prev[g] * c1 - my_l2);
      const float gij = ((dead_prev >> sg::crow(g, half)) & 1u) ? 0.f : e;
      dsc += gij * s_prev[g];
      gacc[g] = gij;
    });
    bf16x8 g0, g1;
    pack_acc(gacc, g0, g1);   // G^T of tile t-1 (negatives on rows) as the A operand: computes G . N
    // (first iteration: prv holds nothing yet, so the product runs on the current tile with G = 0)
    sg::mma_tile_tr<NKS, ND>(i == 0 ? tile : tiles + prv * T::BYTES, la, g0, g1, dq);
    s_prev = acc;
    dead_prev = dead;
  });

  // ---- finish: positive term, L2-normalisation chain rule, accumulation into the source rows -------
  // dq[dc][g]: row (reg) = token wave*32 + crow(g,half), column (lane) = feature dc*32 + r
  float dls = 0.f;
  dsc += __shfl_xor(dsc, 32, 64);   // both halves hold rows of the same token column -> full sum per token (lane r)
  if (live && half == 0) {
    const float ppos = __expf(scale * s_pos[tok] - lse[tok]);
    dls = dsc + (first_split ? my_w * (ppos - 1.0f) * s_pos[tok] : 0.f);
  }
  dls = wave_sum(dls);
  if (lane == 0 && d_logit_scale) atomicAdd(d_logit_scale, dls * scale);   // d/d(param) with scale = exp(param)

#pragma unroll
  for (int g = 0; g < 16; ++g) {
    const int tk = tok0 + wave * 32 + sg::crow(g, half);
    const bool tl = tk < n_tok;
    const float wi = tl ? w[tk] : 0.f;
    const float sp = tl ? s_pos[tk] : 0.f;
    const float coef = (tl && first_split) ? wi * (__expf(scale * sp - lse[tk]) - 1.0f) : 0.f;   // w (p_pos - 1)
    float qv[ND], pv[ND], dqn[ND], dpn[ND];
    float dot_q = 0.f, dot_p = 0.f;
#pragma unroll
    for (int dc = 0; dc < ND; ++dc) {
      const int d = dc * 32 + r;
      const bool ok = tl && d < T::DIM;
      qv[dc] = ok ? (float)qn[(int64_t)tk * T::DIM + d] : 0.f;
      pv[dc] = ok ? (float)pn[(int64_t)tk * T::DIM + d] : 0.f;
      dqn[dc] = ok ? scale * (dq[dc][g] + coef * pv[dc]) : 0.f;
      dpn[dc] = scale * coef * qv[dc];
      dot_q += qv[dc] * dqn[dc];
      dot_p += pv[dc] * dpn[dc];
    }
    dot_q = half_sum(dot_q);
    dot_p = half_sum(dot_p);
    if (tl) {
      // several tokens share a head row (offsets p of one segment) or a target row (l + 1 + p = const): accumulate
      // with float atomics, one 128-byte segment per wave half per instruction (full-rate shape)
      const float iq = q_inv[tk], ip = p_inv[tk];
      float* qdst = dq_rows + (int64_t)q_idx[tk] * T::DIM;
      float* pdst = dp_rows + (int64_t)p_idx[tk] * T::DIM;
#pragma unroll
      for (int dc = 0; dc < ND; ++dc) {
        const int d = dc * 32 + r;
        if (d < T::DIM) {
          atomicAdd(qdst + d, (dqn[dc] - qv[dc] * dot_q) * iq);
          if (first_split) atomicAdd(pdst + d, (dpn[dc] - pv[dc] * dot_p) * ip);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// backward, negative-stationary: d_negs
// ------------------------------------------------------------------------------------------
template <int NKS>
__global__ __launch_bounds__(256, 1) void nce_bwd_n_kernel(const bf16_t* qn, const bf16_t* negs,
                                                           const uint32_t* supp, int n_neg,
                                                           const int32_t* n_tok_dev, int tok_cap,
                                                           const float* __restrict__ logit_scale_dev,
                                                           const float* lse, const float* w,
                                                           float* d_negs) {
  using T = sg::Tile<NKS>;
  constexpr int ND = (NKS + 1) / 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // per ring slot: Q tile | [4 waves][64] words: lanes 0-31 = suppression words of that wave's negative tile for the
  // 32 tokens, lanes 32-63 = w (wave 0) / lse (wave 1) of the 32 tokens
  constexpr int BUF = T::BYTES + 1024;
  using D = sg::Dma<NKS>;
  {
    const int64_t grp = blockIdx.z, to = grp * tok_cap;
    qn += to * T::DIM; negs += grp * (int64_t)n_neg * T::DIM; supp += grp * (int64_t)((n_neg + 31) >> 5) * tok_cap;
    n_tok_dev += grp; lse += to; w += to; d_negs += grp * (int64_t)n_neg * T::DIM;
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
  const bool nlive = neg < n_neg;
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

  auto qrow_for = [&](int tile) {
    return [=](int rr) -> const bf16_t* { return qn + (int64_t)min(tile * 32 + rr, n_tok - 1) * T::DIM; };
  };
  const int my_nt = min((neg0 >> 5) + wv, n_neg_tiles - 1);
  auto issue_all = [&](int buf, int tile) {
    unsigned char* base = smem + buf * BUF;
    D::issue(base, qrow_for(tile), wv, lane);
    const int tk = min(tile * 32 + r, tok_cap - 1);
    const void* src = supp + (int64_t)my_nt * tok_cap + tk;
    if (half == 1 && wv == 0) src = w + tk;
    if (half == 1 && wv == 1) src = lse + tk;
    sg::dma_words(src, base + T::BYTES + wv * 256);
  };
  // one-tile skew as in nce_bwd_q: S(t) MFMAs with the epilogue of S(t-1) in their gaps, then dN += G(t-1)^T . Q(t-1)
  sg::LaneAddr<NKS> la;
  la.init(lane);
  const int n_loc = (tt1 - tt0 + tstep - 1) / tstep;
  issue_all(0, tt0);
  if (n_loc > 1) issue_all(1, tt0 + tstep);
  f32x16 s_prev = sg::zero16();
  int trem_prev = 0;                     // 0 live tokens before the first tile: G = 0
  sg::ring_loop<4>(n_loc + 1, [&](auto slot_c, int i) {
    constexpr int cur = decltype(slot_c)::value, nxt = (cur + 2) % 4, prv = (cur + 3) % 4;
    const int t = tt0 + i * tstep;
    const bool has_tile = i < n_loc;
    if (has_tile) {
      if (i + 1 < n_loc) sg::wait_vmcnt<D::PW + 1>(); else sg::wait_vmcnt<0>();
    }
    sg::ring_barrier();
    if (i + 2 < n_loc) issue_all(nxt, t + 2 * tstep);
    const unsigned char* base = smem + cur * BUF;
    const unsigned char* pbase = i == 0 ? base : smem + prv * BUF;    // first iteration: G = 0 against the current tile
    // per-token scalars of the PREVIOUS tile: suppression words of my wave's negative tile, w, lse
    const uint32_t* wd = reinterpret_cast<const uint32_t*>(pbase + T::BYTES);
    const uint32_t* swd = wd + wv * 64;
    const float* wsc = reinterpret_cast<const float*>(wd + 32);
    const float* lsc = reinterpret_cast<const float*>(wd + 64 + 32);
    // my 16 accumulator rows are 4 runs of 4 consecutive tokens: their scalars come in as 16-byte LDS reads BEFORE the
    // MFMA batch (scalar-sized LDS reads inside the MFMA gaps put an LDS round trip into every gap: measured +25 %)
    const int tb = 4 * half;
    float wr[16], lr[16];
    uint32_t dm = 0;                                   // bit g set: row g contributes nothing
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const f32x4 w4 = *reinterpret_cast<const f32x4*>(wsc + 8 * q4 + tb);
      const f32x4 l4 = *reinterpret_cast<const f32x4*>(lsc + 8 * q4 + tb);
      const __attribute__((ext_vector_type(4))) uint32_t s4 =
          *reinterpret_cast<const __attribute__((ext_vector_type(4))) uint32_t*>(swd + 8 * q4 + tb);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int g = 4 * q4 + e;
        wr[g] = w4[e];
        lr[g] = l4[e] * LOG2E;
        const bool dead = (!nlive) | (sg::crow(g, half) >= trem_prev) | (((s4[e] >> r) & 1u) != 0u);
        dm |= dead ? (1u << g) : 0u;
      }
    }
    f32x16 acc = sg::zero16();
    f32x16 gacc;
    sg::mma_tile_epi<NKS, 4>(base, la, frag, acc, [&](int g) {
      const float ex = wr[g] * fast_exp2(s_prev[g] * c1 - lr[g]);
      gacc[g] = ((dm >> g) & 1u) ? 0.f : ex;
    });
    bf16x8 g0, g1;
    pack_acc(gacc, g0, g1);   // G of tile t-1 (tokens on rows) as the A operand: computes G^T . Qn
    sg::mma_tile_tr<NKS, ND>(pbase, la, g0, g1, dn);
    s_prev = acc;
    trem_prev = has_tile ? n_tok - t * 32 : 0;
  });
  // dn[dc][g]: row (reg) = negative neg0 + wave*32 + crow(g,half), column (lane) = feature dc*32 + r
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    const int nj = neg0 + wave * 32 + sg::crow(g, half);
    if (nj < n_neg) {
#pragma unroll
      for (int dc = 0; dc < ND; ++dc) {
        const int d = dc * 32 + r;
        if (d < T::DIM) atomicAdd(d_negs + (int64_t)nj * T::DIM + d, scale * dn[dc][g]);
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
                           void* stream) {
  MHR_REQUIRE(q_rows && q_idx && p_rows && p_idx && negs && n_tok_dev && logit_scale_dev && sum_out && s_pos,
              "nce_fwd: null pointer");
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
      thres, tps, sum_out, n_valid, rank, (bf16_t*)qn_out, (bf16_t*)pn_out, supp_out, q_inv, p_inv, s_pos
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
#undef ARGS
  MHR_CHECK_LAUNCH("nce_fwd");
  return MHR_OK;
}

extern "C" int mhr_nce_finalize(const float* sum, const float* s_pos, int n_groups, const int32_t* n_tok_dev, int tok_cap,
                                const float* logit_scale_dev, float* loss, float* lse, int32_t* n_valid, void* stream) {
  MHR_REQUIRE(sum && s_pos && n_tok_dev && logit_scale_dev && loss && lse, "nce_finalize: null pointer");
  MHR_REQUIRE(tok_cap > 0 && n_groups >= 1 && n_groups <= 65535, "nce_finalize: bad sizes");
  hipLaunchKernelGGL(nce_finalize_kernel, dim3((tok_cap + 255) / 256, n_groups), dim3(256), 0, (hipStream_t)stream, sum, s_pos,
                     n_tok_dev, tok_cap, logit_scale_dev, loss, lse, n_valid);
  MHR_CHECK_LAUNCH("nce_finalize");
  return MHR_OK;
}

extern "C" int mhr_nce_bwd_tokens(const void* qn, const void* pn, const void* negs, const uint32_t* supp, int n_neg, int dim,
                                  int n_groups, const int32_t* n_tok_dev, int tok_cap, const float* logit_scale_dev, const float* lse,
                                  const float* w, const float* q_inv, const float* p_inv, const float* s_pos,
                                  const int32_t* q_idx, const int32_t* p_idx, float* dq_rows, float* dp_rows,
                                  float* d_logit_scale, void* stream) {
  MHR_REQUIRE(qn && pn && negs && supp && n_tok_dev && logit_scale_dev && lse && w && q_inv && p_inv && s_pos,
              "nce_bwd_tokens: null input pointer");
  MHR_REQUIRE(q_idx && p_idx && dq_rows && dp_rows, "nce_bwd_tokens: null index/output pointer");
  int nks;
  MHR_REQUIRE(nks_for(dim, nks), "nce_bwd_tokens: dim=%d unsupported (16/32/64/128/256)", dim);
  MHR_REQUIRE(n_neg > 0 && tok_cap > 0 && n_groups >= 1 && n_groups <= 65535, "nce_bwd_tokens: bad sizes");
  hipStream_t s = (hipStream_t)stream;
  int tps;
  const int splits = nce_splits((n_neg + 31) / 32, n_groups, 2, tps);   // every range repeats the atomic epilogue
  const dim3 grid_q((tok_cap + 127) / 128, splits, n_groups);
#define L_(NKS)                                                                                                          \
  {                                                                                                                      \
    size_t lds_q = 4 * sg::Tile<NKS>::BYTES + 4 * 1024;                                                                  \
    hipLaunchKernelGGL((nce_bwd_q_kernel<NKS>), grid_q, dim3(256), lds_q, s, (const bf16_t*)qn, (const bf16_t*)pn,       \
                       (const bf16_t*)negs, supp, n_neg, n_tok_dev, tok_cap, tps, logit_scale_dev, lse, w, q_inv, p_inv, \
                       s_pos, q_idx, p_idx, dq_rows, dp_rows, d_logit_scale);                                            \
  }
  NKS_SWITCH(nks, L_);
#undef L_
  MHR_CHECK_LAUNCH("nce_bwd_tokens");
  return MHR_OK;
}

extern "C" int mhr_nce_bwd_negs(const void* qn, const void* negs, const uint32_t* supp, int n_neg, int dim, int n_groups,
                                const int32_t* n_tok_dev, int tok_cap, const float* logit_scale_dev, const float* lse,
                                const float* w, float* d_negs, void* stream) {
  MHR_REQUIRE(qn && negs && supp && n_tok_dev && logit_scale_dev && lse && w && d_negs, "nce_bwd_negs: null pointer");
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
    hipLaunchKernelGGL((nce_bwd_n_kernel<NKS>), dim3(neg_groups, splits, n_groups), dim3(256), lds_n, s, (const bf16_t*)qn,      \
                       (const bf16_t*)negs, supp, n_neg, n_tok_dev, tok_cap, logit_scale_dev, lse, w, d_negs);          \
  }
  NKS_SWITCH(nks, L_);
#undef L_
  MHR_CHECK_LAUNCH("nce_bwd_negs");
  return MHR_OK;
}
