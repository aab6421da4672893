// Sampled softmax with QUERY-ROW SHARING (reference model/IDNet/hstu.py:600-619, 682-723, 808-870), gfx950.
//
// In the multi-head loss the tokens (b, l, p), p = 0..P-1, of one prior category all use the SAME query row - head c's
// embedding at (b, l) - and differ in their target (position l + 1 + p) only.  The logits s_j = cos(q, n_j) against the
// category's negative pool, the gated values E_j = exp(scale (s_j - 1)), the sum over them and the token-side gradient
// U = sum_j E_j n_j therefore depend on the ROW, not on the token: the reference (and the per-token kernels in nce.hip)
// evaluate them once per token, i.e. up to P times.  What does depend on the token:
//   * the positive pair: s+ = cos(q, target_t)                              (one dot product per token)
//   * the false-negative suppression: negative j is dropped for token t when cos(target_t, n_j) > thres; that is a
//     property of the TARGET row (bit table of nce_fix_bits) and it fires for a few percent of the tokens at most (the
//     sampled pool happens to contain the target or a near-duplicate).
// So the two streaming MFMA kernels of nce.hip (mhr_nce_fwd fused forward, mhr_nce_bwd_negs) run here on the DISTINCT
// ROWS with no suppression at all (cfg1: ~3.4x fewer rows than tokens), and the kernels of this file add the per-token
// terms and take the suppressed pairs back out, pair by pair:
//   sum_t = sum_row - sum_{j in J_t} E_j,   U_t = U_row - sum_{j in J_t} bf16(E_j) n_j,
//   dN_j -= scale w_t exp(scale s_j - lse_t) qn_row   for j in J_t,
//   n_valid_t = n_valid_row - |J_t|,   rank_t = rank_row - #{j in J_t : s_j > s+}
// (bf16(E_j): the row kernel feeds the gated tile to the matrix pipe in bf16, the correction removes what went in).
// Per-row weight of the negative-side backward: sum_t w_t exp(scale s - lse_t) = exp2(c1 s - lw_row),
//   lw_row = -log2 sum_t 2^(-lw_t)  (mhr_nce_row_lw, evaluated max-shifted).
//
// Token lists are ordered prediction-offset-fastest (multihead.py), so the tokens of a row are neighbours: tok2row[t] is
// the row of token t, row_first[r] .. row_first[r + 1] its tokens.
#include "mhr_common.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;
constexpr int NC = 4;   // 64-column chunks (dim <= 256): lane owns columns c * 64 + lane

__device__ __forceinline__ float clamp_scale(const float* logit_scale_dev) {
  float ls = *logit_scale_dev;
  ls = fminf(fmaxf(ls, 0.0f), 4.605170185988092f);   // [0, ln 100]  (hstu.py:602)
  return __expf(ls);
}

// Visits the suppressed negatives of one target row (wave-uniform control flow): f(j) for every set bit of the row's
// column of the bit table.  fixw: [n_tiles, n_rows_pad] words of this group, bit b of word [tile, slot] = negative 32 tile + b.
// `groups` = fix_any[slot] (wave-uniform): only the flagged tile groups are read - typically 8 words of the 256 a cfg1 row
// has (its column of the table is strided by the row count: every word is a cache line of its own).
template <typename F>
__device__ __forceinline__ void for_each_hit(const uint32_t* __restrict__ fixw, int n_tiles, int n_rows_pad, int slot, int n_neg,
                                             int lane, uint32_t groups, F&& f) {
  const int sh = fix_group_shift(n_tiles), per = 1 << sh;
  while (groups) {
    const int k = __builtin_ctz(groups);
    groups &= groups - 1;
    const int g0 = k << sh, g1 = min(g0 + per, n_tiles);
    for (int w0 = g0; w0 < g1; w0 += 64) {
      const int tile = w0 + lane;
      const uint32_t word = tile < g1 ? fixw[(int64_t)tile * n_rows_pad + slot] : 0u;
      uint64_t m = __ballot(word != 0u);
      while (m) {
        const int l = __builtin_ctzll(m);
        m &= m - 1;
        uint32_t wv = (uint32_t)__builtin_amdgcn_readlane((int)word, l);
        while (wv) {
          const int b = __builtin_ctz(wv);
          wv &= wv - 1;
          const int j = (w0 + l) * 32 + b;
          if (j < n_neg) f(j);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// forward, per token: normalised target, s+, the row's sum minus this token's suppressed negatives, log counters
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void shared_tok_fwd_kernel(
    const bf16_t* __restrict__ pn_rows, int n_p_rows, const int32_t* __restrict__ p_idx, const int32_t* __restrict__ tok2row,
    const int32_t* __restrict__ n_tok_dev, int tok_cap, int row_cap, const bf16_t* __restrict__ qn_row,
    const float* __restrict__ sum_row, const int32_t* __restrict__ n_valid_row, const int32_t* __restrict__ rank_row,
    const bf16_t* __restrict__ negs, int n_neg, int dim, const float* __restrict__ logit_scale_dev,
    const uint32_t* __restrict__ fixw, int n_rows_pad, const int32_t* __restrict__ slot_of_row,
    const int32_t* __restrict__ fix_any, float* __restrict__ s_pos,
    float* __restrict__ sum_tok, int32_t* __restrict__ n_valid_tok, int32_t* __restrict__ rank_tok) {
  const int n_tiles = (n_neg + 31) >> 5;
  {
    const int64_t grp = blockIdx.z, to = grp * tok_cap, ro = grp * row_cap;
    p_idx += to; tok2row += to; n_tok_dev += grp; s_pos += to; sum_tok += to;
    qn_row += ro * dim; sum_row += ro;
    if (n_valid_row) { n_valid_row += ro; n_valid_tok += to; }
    if (rank_row) { rank_row += ro; rank_tok += to; }
    negs += grp * (int64_t)((n_neg + 31) & ~31) * dim;
    fixw += grp * (int64_t)n_tiles * n_rows_pad;
    fix_any += grp * (int64_t)n_rows_pad;
    if (slot_of_row) slot_of_row += grp * (int64_t)n_p_rows;
  }
  const int n_tok = min(*n_tok_dev, tok_cap);
  const int lane = threadIdx.x & 63;
  const int wave_g = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), n_waves = gridDim.x * (blockDim.x >> 6);
  const float scale = clamp_scale(logit_scale_dev);
  const float c1 = scale * LOG2E;
  // A HALF-wave per token: lane owns the 8 consecutive columns 8 hl .. 8 hl + 7 of its half's token (one 16-byte load per row
  // and lane; a 512-byte row is one half-wave-instruction), TB token PAIRS in flight per wave, and the dependent index loads
  // (target row -> table slot -> any-hit flag) of all of them are issued before the first row is consumed.  (One wave per token
  // with 8-byte loads: 99 us for the 273 k tokens of a cfg1 step - shuffle reductions and load latency, not bandwidth.)
  // Tokens with suppressed negatives - a few per cent - are then visited by the WHOLE wave, one after the other.
  constexpr int TB = 4;
  const int hl = lane & 31, hw = lane >> 5;
  const int d0 = hl * 8;
  const bool in_dim = d0 < dim;
  const int d0w = lane * 4;                                          // (whole-wave layout of the hit path: 4 columns per lane)
  const bool in_dim_w = d0w < dim;
  auto half_sum = [](float v) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
  };
  for (int t0 = wave_g * 2 * TB; t0 < n_tok; t0 += n_waves * 2 * TB) {
    int rr[TB], prr[TB], slot[TB], any[TB];
    bf16x8 pv[TB], qv[TB];
#pragma unroll
    for (int b = 0; b < TB; ++b) {
      const int tk = min(t0 + 2 * b + hw, n_tok - 1);
      rr[b] = tok2row[tk];
      prr[b] = p_idx[tk];
    }
#pragma unroll
    for (int b = 0; b < TB; ++b) {
      bf16x8 z;
#pragma unroll
      for (int e = 0; e < 8; ++e) z[e] = (bf16_t)0.f;
      pv[b] = in_dim ? *reinterpret_cast<const bf16x8*>(pn_rows + (int64_t)prr[b] * dim + d0) : z;   // the target's normalised row
      qv[b] = in_dim ? *reinterpret_cast<const bf16x8*>(qn_row + (int64_t)rr[b] * dim + d0) : z;
      slot[b] = slot_of_row ? slot_of_row[prr[b]] : prr[b];
    }
#pragma unroll
    for (int b = 0; b < TB; ++b) any[b] = fix_any[slot[b]];
#pragma unroll
    for (int b = 0; b < TB; ++b) {
      if (t0 + 2 * b >= n_tok) break;                                // (wave-uniform)
      const int tk = t0 + 2 * b + hw;
      float sp = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) sp += (float)qv[b][e] * (float)pv[b][e];
      sp = half_sum(sp);
      float corr = 0.f;
      int hits = 0, above = 0;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int any_h = __shfl(any[b], h * 32, 64);
        if (any_h != 0 && t0 + 2 * b + h < n_tok) {                  // wave-uniform; a few per cent of the tokens
          const int slot_h = __shfl(slot[b], h * 32, 64), r_h = __shfl(rr[b], h * 32, 64);
          const float sp_h = __shfl(sp, h * 32, 64);
          float q4[4] = {0.f, 0.f, 0.f, 0.f};
          if (in_dim_w) {
            const bf16x4 qw = *reinterpret_cast<const bf16x4*>(qn_row + (int64_t)r_h * dim + d0w);
#pragma unroll
            for (int e = 0; e < 4; ++e) q4[e] = (float)qw[e];
          }
          float corr_h = 0.f;
          int hits_h = 0, above_h = 0;
          for_each_hit(fixw, n_tiles, n_rows_pad, slot_h, n_neg, lane, (uint32_t)any_h, [&](int j) {
            float sj = 0.f;
            if (in_dim_w) {
              const bf16x4 nv = *reinterpret_cast<const bf16x4*>(negs + (int64_t)j * dim + d0w);
#pragma unroll
              for (int e = 0; e < 4; ++e) sj += q4[e] * (float)nv[e];
            }
            sj = wave_sum(sj);
            corr_h += __builtin_amdgcn_exp2f(sj * c1 - c1);
            hits_h += 1;
            above_h += sj > sp_h ? 1 : 0;
          });
          if (hw == h) {
            corr = corr_h;
            hits = hits_h;
            above = above_h;
          }
        }
      }
      if (hl == 0 && tk < n_tok) {
        const int r = rr[b];
        s_pos[tk] = sp;
        sum_tok[tk] = fmaxf(sum_row[r] - corr, 0.f);
        if (n_valid_row) n_valid_tok[tk] = n_valid_row[r] - hits;
        if (rank_row) rank_tok[tk] = max(rank_row[r] - above, 0);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// backward, per token (the row-wise kernel of nce.hip reading the row's state): dq_rows, dp_rows, d(logit_scale), lw,
// and the suppressed pairs taken back out of U (token side) and of d_negs (negative side)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void shared_tok_bwd_kernel(
    const bf16_t* __restrict__ qn_row, const float* __restrict__ u_row, const float* __restrict__ q_inv_row, int row_cap,
    const int32_t* __restrict__ tok2row, const bf16_t* __restrict__ pn, int dim, const int32_t* __restrict__ n_tok_dev,
    int tok_cap, const float* __restrict__ logit_scale_dev, const float* __restrict__ lse, const float* __restrict__ w,
    const float* __restrict__ p_inv, const float* __restrict__ s_pos, const int32_t* __restrict__ q_idx,
    const int32_t* __restrict__ p_idx, float* __restrict__ dq_rows, float* __restrict__ dp_rows,
    float* __restrict__ d_logit_scale, float* __restrict__ lw_out, const int32_t* __restrict__ w_bucket, int n_buckets,
    const bf16_t* __restrict__ negs, int n_neg, const uint32_t* __restrict__ fixw, int n_rows_pad, int n_p_rows,
    const int32_t* __restrict__ slot_of_row, const int32_t* __restrict__ fix_any, float* __restrict__ d_negs) {
  const int n_tiles = (n_neg + 31) >> 5;
  {
    const int64_t grp = blockIdx.z, to = grp * tok_cap, ro = grp * row_cap;
    qn_row += ro * dim; u_row += ro * dim; q_inv_row += ro;
    tok2row += to; n_tok_dev += grp; lse += to; s_pos += to; q_idx += to; p_idx += to;      // pn / p_inv: per target row, shared by the groups
    lw_out += to;
    if (w_bucket) { w_bucket += to; w += grp * n_buckets; } else { w += to; }
    negs += grp * (int64_t)((n_neg + 31) & ~31) * dim;
    fixw += grp * (int64_t)n_tiles * n_rows_pad;
    fix_any += grp * (int64_t)n_rows_pad;
    if (slot_of_row) slot_of_row += grp * (int64_t)n_p_rows;
    if (d_negs) d_negs += grp * (int64_t)n_neg * dim;
  }
  const int n_tok = min(*n_tok_dev, tok_cap);
  const int lane = threadIdx.x & 63;
  const int wave_g = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), n_waves = gridDim.x * (blockDim.x >> 6);
  const float scale = clamp_scale(logit_scale_dev);
  const float c1 = scale * LOG2E;
  constexpr int TB = 8;      // consecutive tokens per wave pass: the tokens of a row are neighbours, their dq rows leave as one
  float dls = 0.f;
  for (int t0 = wave_g * TB; t0 < n_tok; t0 += n_waves * TB) {
    float accq[NC] = {0.f, 0.f, 0.f, 0.f};
    float qv[NC] = {0.f, 0.f, 0.f, 0.f}, uv[NC] = {0.f, 0.f, 0.f, 0.f};
    int run_row = -1, cur_r = -1;
    float iq = 0.f;
    auto flush = [&]() {
      if (run_row >= 0) {
        float* qdst = dq_rows + (int64_t)run_row * dim;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          const int d = c * 64 + lane;
          if (d < dim) atomicAdd(qdst + d, accq[c]);
        }
      }
    };
    for (int b = 0; b < TB; ++b) {
      const int tk = t0 + b;
      if (tk >= n_tok) break;
      const int r = tok2row[tk];
      if (r != cur_r) {                               // wave-uniform: the row's state is loaded once per run
        cur_r = r;
        iq = q_inv_row[r];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          const int d = c * 64 + lane;
          qv[c] = d < dim ? (float)qn_row[(int64_t)r * dim + d] : 0.f;
          uv[c] = d < dim ? u_row[(int64_t)r * dim + d] : 0.f;
        }
      }
      const float wi = w_bucket ? w[w_bucket[tk]] : w[tk];
      const float sp = s_pos[tk], ls = lse[tk];
      const int qi = q_idx[tk], pi = p_idx[tk];
      const float ip = p_inv[pi];
      float pv[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int d = c * 64 + lane;
        pv[c] = d < dim ? (float)pn[(int64_t)pi * dim + d] : 0.f;
      }
      const float a = wi * __expf(scale - ls);                          // G_ij = a * E_ij
      const float coef = wi * (__expf(scale * sp - ls) - 1.0f);         // w (p_pos - 1)
      if (lane == 0) lw_out[tk] = ls * LOG2E - __log2f(wi);
      // this token's suppressed negatives: out of U, and out of what the row-level negative-side kernel adds to d_negs
      float uc[NC] = {0.f, 0.f, 0.f, 0.f};
      const int slot = slot_of_row ? slot_of_row[pi] : pi;
      const uint32_t hit_groups = (uint32_t)fix_any[slot];
      if (hit_groups != 0) {
        for_each_hit(fixw, n_tiles, n_rows_pad, slot, n_neg, lane, hit_groups, [&](int j) {
          const bf16_t* ns = negs + (int64_t)j * dim;
          float nv[NC];
          float s = 0.f;
#pragma unroll
          for (int c = 0; c < NC; ++c) {
            const int d = c * 64 + lane;
            nv[c] = d < dim ? (float)ns[d] : 0.f;
            s += qv[c] * nv[c];
          }
          s = wave_sum(s);
          const float eb = (float)(bf16_t)__builtin_amdgcn_exp2f(s * c1 - c1);       // what the row kernel put into U
          const float gneg = -scale * wi * __expf(scale * s - ls);                   // what bwd_negs adds for this token
#pragma unroll
          for (int c = 0; c < NC; ++c) {
            const int d = c * 64 + lane;
            uc[c] += eb * nv[c];
            if (d_negs && d < dim && wi != 0.f) atomicAdd(d_negs + (int64_t)j * dim + d, gneg * qv[c]);
          }
        });
      }
      float dqn[NC], dpn[NC];
      float dot_q = 0.f, dot_p = 0.f, dot_raw = 0.f;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const float raw = a * (uv[c] - uc[c]);
        dqn[c] = scale * (raw + coef * pv[c]);
        dpn[c] = scale * coef * qv[c];
        dot_raw += qv[c] * raw;
        dot_q += qv[c] * dqn[c];
        dot_p += pv[c] * dpn[c];
      }
      dot_q = wave_sum(dot_q);
      dot_p = wave_sum(dot_p);
      dls += wave_sum(dot_raw) + coef * sp;           // sum_j g_ij s_ij = qn_i . dQn_i, plus the positive term
      if (qi != run_row) {                            // wave-uniform
        flush();
        run_row = qi;
#pragma unroll
        for (int c = 0; c < NC; ++c) accq[c] = 0.f;
      }
#pragma unroll
      for (int c = 0; c < NC; ++c) accq[c] += (dqn[c] - qv[c] * dot_q) * iq;
      float* pdst = dp_rows + (int64_t)pi * dim;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int d = c * 64 + lane;
        if (d < dim) atomicAdd(pdst + d, (dpn[c] - pv[c] * dot_p) * ip);
      }
    }
    flush();
  }
  if (lane == 0 && d_logit_scale && dls != 0.f) atomicAdd(d_logit_scale, dls * scale);   // d/d(param), scale = exp(param)
}

// ------------------------------------------------------------------------------------------
// backward without per-token atomics (window-structured token lists: slot (b, l, p) -> target row b (L + P) + l + 1 + p)
// ------------------------------------------------------------------------------------------
// Everything the token-side backward adds is linear in the per-token vectors, so it is summed where it lands:
//   * per ROW (one wave per (group, row)): dq - its tokens share the query row, so A = sum a_t, sum coef_t pn_t and the
//     suppressed-pair corrections are formed in registers and the L2-normalisation chain rule runs ONCE per row; one
//     256-byte-shaped float-atomic set per row (heads can be shared between groups); also lw_row, d(logit_scale) and the
//     d_negs corrections;
//   * per TARGET row (one wave per row of p_rows): dp - the tokens that point at target (b, pos) are (l = pos - 1 - p, p) in
//     every group, found through the inverse slot map of mhr_token_compact; their pn is the same vector, so
//     A = sum scale coef_t qn_row(t) is gathered and the chain rule runs once; plain read-modify-write, no atomics,
//     bitwise reproducible.
// (The per-token form, shared_tok_bwd_kernel above, adds 1 KiB of float atomics per token for dp: 275 MB at cfg1.)
__global__ __launch_bounds__(256) void shared_bwd_rows_kernel(
    const bf16_t* __restrict__ qn_row, const float* __restrict__ u_row, const float* __restrict__ q_inv_row,
    const int32_t* __restrict__ row_q, const int32_t* __restrict__ row_first, const int32_t* __restrict__ n_row_dev, int row_cap,
    const bf16_t* __restrict__ pn, int dim, int tok_cap, const float* __restrict__ logit_scale_dev,
    const float* __restrict__ lse, const float* __restrict__ w, const float* __restrict__ s_pos,
    const int32_t* __restrict__ p_idx, float* __restrict__ dq_rows, float* __restrict__ d_logit_scale,
    float* __restrict__ lw_row, const int32_t* __restrict__ w_bucket, int n_buckets, const bf16_t* __restrict__ negs, int n_neg,
    const uint32_t* __restrict__ fixw, int n_rows_pad, int n_p_rows, const int32_t* __restrict__ slot_of_row,
    const int32_t* __restrict__ fix_any, float* __restrict__ d_negs, int exclusive_rows, long long* __restrict__ dn_fix,
    float* __restrict__ dls_part) {
  const int n_tiles = (n_neg + 31) >> 5;
  {
    const int64_t grp = blockIdx.z, to = grp * tok_cap, ro = grp * row_cap;
    qn_row += ro * dim; u_row += ro * dim; q_inv_row += ro; row_q += ro; row_first += ro; lw_row += ro; n_row_dev += grp;
    lse += to; s_pos += to; p_idx += to;
    if (w_bucket) { w_bucket += to; w += grp * n_buckets; } else { w += to; }
    negs += grp * (int64_t)((n_neg + 31) & ~31) * dim;
    fixw += grp * (int64_t)n_tiles * n_rows_pad;
    fix_any += grp * (int64_t)n_rows_pad;
    if (slot_of_row) slot_of_row += grp * (int64_t)n_p_rows;
    if (d_negs) d_negs += grp * (int64_t)n_neg * dim;
    if (dn_fix) dn_fix += grp * (int64_t)n_neg * dim;
  }
  const int n_row = min(*n_row_dev, row_cap - 1);
  // A HALF-wave per (group, row): two rows per wave, lane hl of a half = token hl of ITS row in the scalar round and the 8
  // consecutive columns 8 hl .. 8 hl + 7 of the row vectors (16-byte bf16 / two 16-byte fp32 accesses).  Float atomics want a
  // wave instruction covering consecutive floats, so what is added atomically passes through a half-private 1 KB LDS transpose.
  // Tokens with suppressed negatives - a few per cent - are visited by the WHOLE wave, one after the other, in the 4-columns-
  // per-lane layout of for_each_hit; their corrections reach the row's accumulators through the same transpose buffer.
  const int lane = threadIdx.x & 63, hl = lane & 31, hw = lane >> 5;
  const int wave_g = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), n_waves = gridDim.x * (blockDim.x >> 6);
  const float scale = clamp_scale(logit_scale_dev);
  const float c1 = scale * LOG2E;
  float dls = 0.f;
  __shared__ float xpose[4][2][256];
  float* xp = xpose[threadIdx.x >> 6][hw];
  const int d0 = hl * 8, d0w = lane * 4;
  const bool in_dim = d0 < dim, in_dim_w = d0w < dim;
  auto half_sum = [](float v) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
  };
  auto half_max = [](float v) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
  };
  for (int r0 = wave_g * 2; r0 < n_row; r0 += n_waves * 2) {
    const int r = r0 + hw;
    const bool row_ok = r < n_row;
    const int rr = row_ok ? r : n_row - 1;
    const int t0 = row_ok ? row_first[rr] : 0, t1 = row_ok ? min(row_first[rr + 1], tok_cap) : 0;
    float q8[8], u8[8], ap[8], au[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) q8[e] = u8[e] = ap[e] = au[e] = 0.f;
    if (in_dim) {
      const bf16x8 qb = *reinterpret_cast<const bf16x8*>(qn_row + (int64_t)rr * dim + d0);
      const f32x4 ua = *reinterpret_cast<const f32x4*>(u_row + (int64_t)rr * dim + d0);
      const f32x4 ub = *reinterpret_cast<const f32x4*>(u_row + (int64_t)rr * dim + d0 + 4);
#pragma unroll
      for (int e = 0; e < 8; ++e) q8[e] = (float)qb[e];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        u8[e] = ua[e];
        u8[4 + e] = ub[e];
      }
    }
    float auw[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};   // hit-path corrections of half 0 / 1's row, 4-col layout
    bool any_hit[2] = {false, false};                                  // (wave-uniform)
    float a_sum = 0.f, lw_m = INFINITY, lw_s = 0.f;
    const int len = t1 - t0;
    const int len_max = max(__shfl(len, 0, 64), __shfl(len, 32, 64));
    for (int off = 0; off < len_max; off += 32) {
      // lane = token of the row: every per-token scalar (and the dependent slot / hit-flag look-ups) in ONE round of loads
      const int tk = t0 + off + hl;
      const bool live = tk < t1;
      float wi = 0.f, sp = 0.f, ls = 0.f;
      int slot = 0, pi = 0, hit_groups = 0;
      bool hit = false;
      if (live) {
        wi = w_bucket ? w[w_bucket[tk]] : w[tk];
        sp = s_pos[tk];
        ls = lse[tk];
        pi = p_idx[tk];
        slot = slot_of_row ? slot_of_row[pi] : pi;
        hit_groups = fix_any[slot];
        hit = hit_groups != 0;
      }
      const float a = live ? wi * __expf(scale - ls) : 0.f;
      const float coef = live ? wi * (__expf(scale * sp - ls) - 1.0f) : 0.f;
      const float lw = (live && wi > 0.f) ? ls * LOG2E - __log2f(wi) : INFINITY;
      a_sum += half_sum(a);
      dls += half_sum(coef * sp);
      const float m = -half_max(-lw);                          // min over the chunk
      if (m < INFINITY) {                                      // running max-shifted sum of 2^(-lw), merged chunk by chunk
        const float sc = half_sum(lw < INFINITY ? __builtin_amdgcn_exp2f(m - lw) : 0.f);
        if (m < lw_m) {
          lw_s = (lw_m < INFINITY ? lw_s * __builtin_amdgcn_exp2f(m - lw_m) : 0.f) + sc;
          lw_m = m;
        } else {
          lw_s += sc * __builtin_amdgcn_exp2f(lw_m - m);
        }
      }
      const int cnt = max(0, min(32, len - off));
      const int cnt_max = max(__shfl(cnt, 0, 64), __shfl(cnt, 32, 64));
      for (int i = 0; i < cnt_max; i += 2) {                   // two target rows in flight per half, added in token order
        bf16x8 pb[2];
        float cf[2];
        bool on[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          on[u] = i + u < cnt;
          const int from = hw * 32 + min(i + u, 31);
          cf[u] = __shfl(coef, from, 64);
          const int pj = __shfl(pi, from, 64);
          if (on[u] && in_dim) pb[u] = *reinterpret_cast<const bf16x8*>(pn + (int64_t)pj * dim + d0);   // the token's target row
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
          if (on[u] && in_dim) {
#pragma unroll
            for (int e = 0; e < 8; ++e) ap[e] += cf[u] * (float)pb[u][e];
          }
      }
      uint64_t hm = __ballot(hit);
      while (hm) {                                             // tokens with suppressed negatives: a few percent
        const int i = __builtin_ctzll(hm);
        hm &= hm - 1;
        const int h = i >> 5;                                  // the half (= row) the token belongs to
        const int slot_u = __builtin_amdgcn_readlane(slot, i);
        const uint32_t groups_u = (uint32_t)__builtin_amdgcn_readlane(hit_groups, i);
        const float a_u = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a), i));
        const float wi_u = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wi), i));
        const float ls_u = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ls), i));
        const int r_u = __builtin_amdgcn_readlane(rr, h * 32);
        // the row's query in the whole-wave layouts of the visit: 4 consecutive columns per lane, and column order (d = c 64 + lane)
        float q4[4] = {0.f, 0.f, 0.f, 0.f}, qc[NC] = {0.f, 0.f, 0.f, 0.f};
        if (in_dim_w) {
          const bf16x4 qw = *reinterpret_cast<const bf16x4*>(qn_row + (int64_t)r_u * dim + d0w);
#pragma unroll
          for (int e = 0; e < 4; ++e) q4[e] = (float)qw[e];
        }
        if (d_negs) {
#pragma unroll
          for (int c = 0; c < NC; ++c) {
            const int d = c * 64 + lane;
            if (d < dim) qc[c] = (float)qn_row[(int64_t)r_u * dim + d];
          }
        }
        float acc4[4] = {0.f, 0.f, 0.f, 0.f};
        for_each_hit(fixw, n_tiles, n_rows_pad, slot_u, n_neg, lane, groups_u, [&](int j) {
          float nv[4] = {0.f, 0.f, 0.f, 0.f};
          float s = 0.f;
          if (in_dim_w) {
            const bf16x4 nb = *reinterpret_cast<const bf16x4*>(negs + (int64_t)j * dim + d0w);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              nv[e] = (float)nb[e];
              s += q4[e] * nv[e];
            }
          }
          s = wave_sum(s);
          const float eb = (float)(bf16_t)__builtin_amdgcn_exp2f(s * c1 - c1);
          const float gneg = -scale * wi_u * __expf(scale * s - ls_u);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc4[e] += a_u * eb * nv[e];
          if (d_negs && wi_u != 0.f) {
#pragma unroll
            for (int c = 0; c < NC; ++c) {
              const int d = c * 64 + lane;
              if (d < dim) {
                if (dn_fix) det_atomic_add(dn_fix + (int64_t)j * dim + d, gneg * qc[c]);       // order-independent (deterministic mode)
                else atomicAdd(d_negs + (int64_t)j * dim + d, gneg * qc[c]);
              }
            }
          }
        });
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          auw[0][e] += h == 0 ? acc4[e] : 0.f;
          auw[1][e] += h == 1 ? acc4[e] : 0.f;
        }
        any_hit[h] = true;
      }
    }
    // the hit-path corrections of each half's row: whole-wave 4-col layout -> the half's 8-col layout, through its transpose buffer
#pragma unroll
    for (int h = 0; h < 2; ++h)
      if (any_hit[h]) {                                        // (wave-uniform)
        float* xh = xpose[threadIdx.x >> 6][h];
        if (in_dim_w) *reinterpret_cast<f32x4*>(xh + d0w) = f32x4{auw[h][0], auw[h][1], auw[h][2], auw[h][3]};
        __builtin_amdgcn_wave_barrier();
        if (hw == h && in_dim) {
#pragma unroll
          for (int e = 0; e < 8; ++e) au[e] += xh[d0 + e];
        }
        __builtin_amdgcn_wave_barrier();
      }
    float dqn[8];
    float dot_q = 0.f, dot_raw = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float raw = a_sum * u8[e] - au[e];
      dqn[e] = scale * (raw + ap[e]);
      dot_raw += q8[e] * raw;
      dot_q += q8[e] * dqn[e];
    }
    dot_q = half_sum(dot_q);
    dls += half_sum(dot_raw);
    const float iq = row_ok ? q_inv_row[rr] : 0.f;
    if (exclusive_rows) {
      // no other (group, row) writes this query row (the groups use disjoint heads): a plain 32-byte read-modify-write per lane
      // instead of 256 float atomics per row (15.7 M per step at cfg1 - what this kernel's time went into)
      if (row_ok && in_dim) {
        f32x4* dst = reinterpret_cast<f32x4*>(dq_rows + (int64_t)row_q[rr] * dim + d0);
        f32x4 v0 = dst[0], v1 = dst[1];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v0[e] += (dqn[e] - q8[e] * dot_q) * iq;
          v1[e] += (dqn[4 + e] - q8[4 + e] * dot_q) * iq;
        }
        dst[0] = v0;
        dst[1] = v1;
      }
    } else {
      if (in_dim) {
#pragma unroll
        for (int e = 0; e < 8; ++e) xp[d0 + e] = (dqn[e] - q8[e] * dot_q) * iq;
      }
      __builtin_amdgcn_wave_barrier();
      if (row_ok) {
        float* qdst = dq_rows + (int64_t)row_q[rr] * dim;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const int d = c * 32 + hl;
          if (d < dim) atomicAdd(qdst + d, xp[d]);
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    if (row_ok && hl == 0) lw_row[rr] = lw_m < INFINITY ? lw_m - __log2f(lw_s) : INFINITY;
  }
  // d(logit_scale): ONE atomic per workgroup (every wave adding to the same address serialises: 61 k adds cost 0.6 ms)
  __shared__ float s_dls[4];
  const float dls_w = __shfl(dls, 0, 64) + __shfl(dls, 32, 64);          // the two halves' rows
  if (lane == 0) s_dls[threadIdx.x >> 6] = dls_w;
  __syncthreads();
  if (threadIdx.x == 0 && d_logit_scale) {
    const float t = (s_dls[0] + s_dls[1]) + (s_dls[2] + s_dls[3]);
    // deterministic mode: this workgroup's partial into its own slot; mhr_det_sum_into folds the slots in index order and applies
    // the factor exp(param)
    if (dls_part) dls_part[(int64_t)blockIdx.z * gridDim.x + blockIdx.x] = t;
    else if (t != 0.f) atomicAdd(d_logit_scale, t * scale);       // d/d(param), scale = exp(param)
  }
}


__global__ __launch_bounds__(256) void shared_bwd_targets_kernel(
    const bf16_t* __restrict__ qn_row, int row_cap, const int32_t* __restrict__ tok2row, const int32_t* __restrict__ tok_of_slot,
    const int32_t* __restrict__ n_tok_dev, int n_groups, int n_slots, int tok_cap, int L, int P, const bf16_t* __restrict__ pn,
    const float* __restrict__ p_inv, int dim, const float* __restrict__ logit_scale_dev, const float* __restrict__ lse,
    const float* __restrict__ w, const float* __restrict__ s_pos, const int32_t* __restrict__ w_bucket, int n_buckets,
    int n_p_rows, float* __restrict__ dp_rows) {
  // A HALF-wave per target row (two rows per wave): lane hl of a half = candidate (group, offset) hl, hl + 32, ... of ITS row in
  // the scan, and the 8 consecutive columns 8 hl .. 8 hl + 7 (16-byte bf16 / two 16-byte fp32 accesses) in the accumulation.
  const int lane = threadIdx.x & 63, hl = lane & 31, hw = lane >> 5;
  const int wave_g = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), n_waves = gridDim.x * (blockDim.x >> 6);
  const float scale = clamp_scale(logit_scale_dev);
  const int W = L + P, n_cand = n_groups * P;
  const int d0 = hl * 8;
  const bool in_dim = d0 < dim;
  auto half_sum = [](float v) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
  };
  for (int m0 = wave_g * 2; m0 < n_p_rows; m0 += n_waves * 2) {
    const int m = m0 + hw;
    const bool row_ok = m < n_p_rows;
    const int mm = row_ok ? m : n_p_rows - 1;
    const int b = mm / W, pos = mm - b * W;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    bool t_any = false;
    for (int c0 = 0; c0 < n_cand; c0 += 32) {
      // lane = candidate (group, offset): the token (l = pos - 1 - p, p) of that group, if it is live; its query row index is
      // fetched here too, so the row loads of the loop below do not wait on a dependent index load each
      const int cand = c0 + hl;
      int tk = -1, g = 0, row = 0;
      float coef = 0.f;
      if (row_ok && cand < n_cand) {
        g = cand / P;
        const int p = cand - g * P, l = pos - 1 - p;
        if (l >= 0 && l < L) {
          const int64_t slot = ((int64_t)b * L + l) * P + p;
          if (slot < n_slots) tk = tok_of_slot[(int64_t)g * n_slots + slot];
          if (tk >= min(n_tok_dev[g], tok_cap)) tk = -1;
        }
        if (tk >= 0) {
          const int64_t o = (int64_t)g * tok_cap + tk;
          const float wi = w_bucket ? w[(int64_t)g * n_buckets + w_bucket[o]] : w[o];
          coef = wi * (__expf(scale * s_pos[o] - lse[o]) - 1.0f);
          row = tok2row[o];
        }
      }
      // each half walks ITS live candidates (in candidate order: the sum has one fixed order per row); two row loads in flight
      const uint64_t both = __ballot(tk >= 0);
      uint32_t live = (uint32_t)(both >> (32 * hw));
      const int rounds = max(__popc((uint32_t)both), __popc((uint32_t)(both >> 32)));
      t_any |= live != 0u;
      for (int it = 0; it < rounds; it += 2) {
        int src[2];
        bool on[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          on[u] = live != 0u;
          src[u] = on[u] ? __builtin_ctz(live) : 0;
          live &= live - 1;                                          // (0 stays 0)
        }
        bf16x8 qb[2];
        float cf[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int from = hw * 32 + src[u];
          const int g_u = __shfl(g, from, 64), r = __shfl(row, from, 64);
          cf[u] = __shfl(coef, from, 64);
          if (on[u] && in_dim) qb[u] = *reinterpret_cast<const bf16x8*>(qn_row + ((int64_t)g_u * row_cap + r) * dim + d0);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
          if (on[u] && in_dim) {
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += cf[u] * (float)qb[u][e];
          }
      }
    }
    const float ip = row_ok ? p_inv[mm] : 0.f;
    float pv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, dpn[8];
    float dot = 0.f;
    if (in_dim) {
      const bf16x8 pb = *reinterpret_cast<const bf16x8*>(pn + (int64_t)mm * dim + d0);
#pragma unroll
      for (int e = 0; e < 8; ++e) pv[e] = (float)pb[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      dpn[e] = scale * acc[e];
      dot += pv[e] * dpn[e];
    }
    dot = half_sum(dot);
    if (t_any && row_ok && in_dim) {                         // (no live token points at a row: its gradient stays as it is)
      f32x4* dst = reinterpret_cast<f32x4*>(dp_rows + (int64_t)m * dim + d0);
      f32x4 c0v = dst[0], c1v = dst[1];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        c0v[e] += (dpn[e] - pv[e] * dot) * ip;
        c1v[e] += (dpn[4 + e] - pv[4 + e] * dot) * ip;
      }
      dst[0] = c0v;
      dst[1] = c1v;
    }
  }
}

// lw_row[r] = -log2 sum_{t in row r} 2^(-lw_tok[t]), max-shifted; +inf when every token of the row has zero weight
__global__ __launch_bounds__(256) void row_lw_kernel(const float* __restrict__ lw_tok, const int32_t* __restrict__ row_first,
                                                     const int32_t* __restrict__ n_row_dev, int tok_cap, int row_cap,
                                                     float* __restrict__ lw_row) {
  const int64_t grp = blockIdx.y;
  lw_tok += grp * tok_cap; row_first += grp * row_cap; lw_row += grp * row_cap;
  const int n_row = min(n_row_dev[grp], row_cap - 1);
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_row) {                  // behind the live rows: +inf switches the row off in mhr_nce_bwd_negs (exp2(. - inf) = 0)
    if (r < row_cap) lw_row[r] = INFINITY;
    return;
  }
  const int t0 = row_first[r], t1 = min(row_first[r + 1], tok_cap);
  float m = INFINITY;
  for (int t = t0; t < t1; ++t) m = fminf(m, lw_tok[t]);
  float out = INFINITY;
  if (m < INFINITY) {
    float acc = 0.f;
    for (int t = t0; t < t1; ++t) acc += __builtin_amdgcn_exp2f(m - lw_tok[t]);
    out = m - __log2f(acc);
  }
  lw_row[r] = out;
}

}  // namespace

extern "C" int mhr_nce_shared_fwd_tokens(const void* pn_rows, int64_t n_p_rows, const int32_t* p_idx,
                                         const int32_t* tok2row, int n_groups, const int32_t* n_tok_dev, int tok_cap,
                                         int row_cap, const void* qn_row, const float* sum_row, const int32_t* n_valid_row,
                                         const int32_t* rank_row, const void* negs, int n_neg, int dim,
                                         const float* logit_scale_dev, const uint32_t* fix_words,
                                         const int32_t* fix_slot_of_row, const int32_t* fix_any,
                                         float* s_pos, float* sum_tok, int32_t* n_valid_tok, int32_t* rank_tok, void* stream) {
  MHR_REQUIRE(pn_rows && p_idx && tok2row && n_tok_dev && qn_row && sum_row && negs && logit_scale_dev && fix_words && fix_any,
              "nce_shared_fwd_tokens: null input pointer");
  MHR_REQUIRE(s_pos && sum_tok, "nce_shared_fwd_tokens: null output pointer");
  MHR_REQUIRE((n_valid_row != nullptr) == (n_valid_tok != nullptr) && (rank_row != nullptr) == (rank_tok != nullptr),
              "nce_shared_fwd_tokens: row / token log counters go together");
  MHR_REQUIRE(dim > 0 && dim <= 256 && dim % 8 == 0, "nce_shared_fwd_tokens: dim=%d unsupported (multiple of 8, <= 256)", dim);
  MHR_REQUIRE(tok_cap > 0 && row_cap > 0 && n_neg > 0 && n_p_rows > 0 && n_groups >= 1 && n_groups <= 65535,
              "nce_shared_fwd_tokens: bad sizes");
  const int n_rows_pad = (int)((n_p_rows + 255) / 256 * 256);
  int blocks = (tok_cap + 31) / 32;                  // 4 waves x 8 tokens
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(shared_tok_fwd_kernel, dim3(blocks, 1, n_groups), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)pn_rows, (int)n_p_rows, p_idx, tok2row, n_tok_dev, tok_cap, row_cap, (const bf16_t*)qn_row,
                     sum_row, n_valid_row, rank_row, (const bf16_t*)negs, n_neg, dim, logit_scale_dev, fix_words, n_rows_pad,
                     fix_slot_of_row, fix_any, s_pos, sum_tok, n_valid_tok, rank_tok);
  MHR_CHECK_LAUNCH("nce_shared_fwd_tokens");
  return MHR_OK;
}

extern "C" int mhr_nce_shared_bwd_tokens(const void* qn_row, const float* u_row, const float* q_inv_row, int row_cap,
                                         const int32_t* tok2row, const void* pn, int dim, int n_groups,
                                         const int32_t* n_tok_dev, int tok_cap, const float* logit_scale_dev,
                                         const float* lse, const float* w, const float* p_inv, const float* s_pos,
                                         const int32_t* q_idx, const int32_t* p_idx, float* dq_rows, float* dp_rows,
                                         float* d_logit_scale, float* lw_out, const int32_t* w_bucket, int n_buckets,
                                         const void* negs, int n_neg, const uint32_t* fix_words, int64_t n_p_rows,
                                         const int32_t* fix_slot_of_row, const int32_t* fix_any, float* d_negs, void* stream) {
  MHR_REQUIRE(qn_row && u_row && q_inv_row && tok2row && pn && n_tok_dev && logit_scale_dev && lse && w && p_inv && s_pos,
              "nce_shared_bwd_tokens: null input pointer");
  MHR_REQUIRE(q_idx && p_idx && dq_rows && dp_rows && lw_out && negs && fix_words && fix_any,
              "nce_shared_bwd_tokens: null index/output pointer");
  MHR_REQUIRE(dim > 0 && dim <= 256, "nce_shared_bwd_tokens: dim=%d unsupported (<= 256)", dim);
  MHR_REQUIRE(tok_cap > 0 && row_cap > 0 && n_neg > 0 && n_p_rows > 0 && n_groups >= 1 && n_groups <= 65535,
              "nce_shared_bwd_tokens: bad sizes");
  const int n_rows_pad = (int)((n_p_rows + 255) / 256 * 256);
  int blocks = (tok_cap + 31) / 32;                 // 4 waves x 8 tokens per pass
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(shared_tok_bwd_kernel, dim3(blocks, 1, n_groups), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)qn_row, u_row, q_inv_row, row_cap, tok2row, (const bf16_t*)pn, dim, n_tok_dev, tok_cap,
                     logit_scale_dev, lse, w, p_inv, s_pos, q_idx, p_idx, dq_rows, dp_rows, d_logit_scale, lw_out, w_bucket,
                     n_buckets, (const bf16_t*)negs, n_neg, fix_words, n_rows_pad, (int)n_p_rows, fix_slot_of_row, fix_any,
                     d_negs);
  MHR_CHECK_LAUNCH("nce_shared_bwd_tokens");
  return MHR_OK;
}

extern "C" int mhr_nce_shared_bwd_rows(const void* qn_row, const float* u_row, const float* q_inv_row, const int32_t* row_q,
                                       const int32_t* row_first, const int32_t* n_row_dev, int row_cap, const void* pn, int dim,
                                       int n_groups, int tok_cap, const float* logit_scale_dev, const float* lse,
                                       const float* w, const float* s_pos, const int32_t* p_idx, float* dq_rows,
                                       float* d_logit_scale, float* lw_row, const int32_t* w_bucket, int n_buckets,
                                       const void* negs, int n_neg, const uint32_t* fix_words, int64_t n_p_rows,
                                       const int32_t* fix_slot_of_row, const int32_t* fix_any, float* d_negs, int exclusive_rows,
                                       int64_t* dn_fix, float* dls_part, void* stream) {
  MHR_REQUIRE(qn_row && u_row && q_inv_row && row_q && row_first && n_row_dev && pn && logit_scale_dev && lse && w && s_pos,
              "nce_shared_bwd_rows: null input pointer");
  MHR_REQUIRE(p_idx && dq_rows && lw_row && negs && fix_words && fix_any, "nce_shared_bwd_rows: null index/output pointer");
  MHR_REQUIRE(dim > 0 && dim <= 256 && dim % 8 == 0, "nce_shared_bwd_rows: dim=%d unsupported (multiple of 8, <= 256)", dim);
  MHR_REQUIRE(tok_cap > 0 && row_cap > 1 && n_neg > 0 && n_p_rows > 0 && n_groups >= 1 && n_groups <= 65535,
              "nce_shared_bwd_rows: bad sizes");
  const int n_rows_pad = (int)((n_p_rows + 255) / 256 * 256);
  int blocks = (row_cap + 31) / 32;                 // 4 waves x 2 rows x 4 passes (more, smaller workgroups were slower: 0.126 -> 0.15 ms)
  if (blocks > 1024) blocks = 1024;
  MHR_REQUIRE(!exclusive_rows || ((uintptr_t)dq_rows % 16 == 0), "nce_shared_bwd_rows: exclusive_rows needs a 16-byte aligned dq_rows");
  hipLaunchKernelGGL(shared_bwd_rows_kernel, dim3(blocks, 1, n_groups), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)qn_row, u_row, q_inv_row, row_q, row_first, n_row_dev, row_cap, (const bf16_t*)pn, dim, tok_cap,
                     logit_scale_dev, lse, w, s_pos, p_idx, dq_rows, d_logit_scale, lw_row, w_bucket, n_buckets,
                     (const bf16_t*)negs, n_neg, fix_words, n_rows_pad, (int)n_p_rows, fix_slot_of_row, fix_any, d_negs,
                     exclusive_rows ? 1 : 0, (long long*)dn_fix, dls_part);
  MHR_CHECK_LAUNCH("nce_shared_bwd_rows");
  return MHR_OK;
}

extern "C" int mhr_nce_shared_bwd_targets(const void* qn_row, int row_cap, const int32_t* tok2row, const int32_t* tok_of_slot,
                                          const int32_t* n_tok_dev, int n_groups, int n_slots, int tok_cap, int seq_len,
                                          int pred_len, const void* pn, const float* p_inv, int dim,
                                          const float* logit_scale_dev, const float* lse, const float* w, const float* s_pos,
                                          const int32_t* w_bucket, int n_buckets, int64_t n_p_rows, float* dp_rows,
                                          void* stream) {
  MHR_REQUIRE(qn_row && tok2row && tok_of_slot && n_tok_dev && pn && p_inv && logit_scale_dev && lse && w && s_pos && dp_rows,
              "nce_shared_bwd_targets: null pointer");
  MHR_REQUIRE(dim > 0 && dim <= 256 && dim % 8 == 0, "nce_shared_bwd_targets: dim=%d unsupported (multiple of 8, <= 256)", dim);
  MHR_REQUIRE(seq_len > 0 && pred_len > 0 && n_slots > 0 && n_slots % (seq_len * pred_len) == 0 &&
                  n_p_rows == (int64_t)(n_slots / (seq_len * pred_len)) * (seq_len + pred_len),
              "nce_shared_bwd_targets: slots must be (b, l, p) windows and p_rows the [B, L + P] targets (n_slots=%d L=%d P=%d "
              "n_p_rows=%lld)", n_slots, seq_len, pred_len, (long long)n_p_rows);
  MHR_REQUIRE(tok_cap > 0 && row_cap > 0 && n_groups >= 1 && n_groups <= 65535, "nce_shared_bwd_targets: bad sizes");
  int blocks = (int)((n_p_rows + 7) / 8);             // 4 waves x 2 rows
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(shared_bwd_targets_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)qn_row, row_cap,
                     tok2row, tok_of_slot, n_tok_dev, n_groups, n_slots, tok_cap, seq_len, pred_len, (const bf16_t*)pn, p_inv, dim,
                     logit_scale_dev, lse, w, s_pos, w_bucket, n_buckets, (int)n_p_rows, dp_rows);
  MHR_CHECK_LAUNCH("nce_shared_bwd_targets");
  return MHR_OK;
}

extern "C" int mhr_nce_row_lw(const float* lw_tok, const int32_t* row_first, const int32_t* n_row_dev, int n_groups,
                              int tok_cap, int row_cap, float* lw_row, void* stream) {
  MHR_REQUIRE(lw_tok && row_first && n_row_dev && lw_row, "nce_row_lw: null pointer");
  MHR_REQUIRE(tok_cap > 0 && row_cap > 1 && n_groups >= 1 && n_groups <= 65535, "nce_row_lw: bad sizes");
  hipLaunchKernelGGL(row_lw_kernel, dim3((row_cap + 255) / 256, n_groups), dim3(256), 0, (hipStream_t)stream, lw_tok,
                     row_first, n_row_dev, tok_cap, row_cap, lw_row);
  MHR_CHECK_LAUNCH("nce_row_lw");
  return MHR_OK;
}
