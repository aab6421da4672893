// Epilogues of the wide-feature path (feature dim > 256: HSTU size-4, D = 1024; the HLLM twin, D = 1536 / 2048).
//
// The row-stationary streaming kernels (nce.hip, catalog.hip) keep one operand of the logit product in registers, which
// bounds the feature dim at 256.  Above that the contraction itself is deep enough (K >= 512) for the library GEMM to be
// the right tool: it runs as bf16 x bf16 -> fp32 hipBLASLt GEMMs over token / item chunks, and
// everything the reference does to the logits afterwards is ONE pass over the fp32 chunk here instead of the
// reference's chain of materialised [N_tok, n_neg] / [B, H, N] tensors:
//   nce_dense_fwd   model/IDNet/hstu.py:600-619 + 697 (mask false negatives, temperature, concat, cross entropy,
//                   hstu.py:621-629 top-k logs): per token lse / loss / #kept / rank of the positive;
//   nce_dense_bwd   the softmax gradient tile w * exp(scale*s - lse) * keep as the bf16 operand of the two gradient GEMMs;
//   catalog_mask_dense / catalog_emit_dense   hstu.py:982-999 + trainer.py:724 + collector.py:245: tag / given-prior /
//                   pad masks and the threshold test, emitting (value, item) candidates in the list format of
//                   catalog_score_emit_sliced, so the exact select (topk_select_sliced) is shared with the streaming path.
// All HBM-bound: one wave per logit row, 16 B per lane per access.
#include "mhr_common.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;

#define WAVE_ROWS(rows)                                                                                             \
  const int lane = threadIdx.x & 63;                                                                                \
  const int64_t wave0 = (int64_t)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); \
  const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x >> 6);                                                   \
  for (int64_t row = wave0; row < (rows); row += n_waves)

__global__ __launch_bounds__(256) void nce_dense_fwd_kernel(const float* __restrict__ s, const float* __restrict__ fix, int64_t ld,
                                                            int n_neg, const float* __restrict__ s_pos,
                                                            const float* __restrict__ scale_p, float thres,
                                                            const int32_t* __restrict__ n_live_p, int64_t row_base, int64_t rows,
                                                            float* __restrict__ lse, float* __restrict__ loss,
                                                            int32_t* __restrict__ n_valid, int32_t* __restrict__ rank) {
  const float scale = scale_p[0], c2 = scale * LOG2E;
  const int64_t n_live = n_live_p ? (int64_t)n_live_p[0] : row_base + rows;
  const bool vec = (ld % 4 == 0);
  WAVE_ROWS(rows) {
    const float* sr = s + row * ld;
    const float* fr = fix + row * ld;
    const float sp = s_pos[row];
    float tot = 0.f;
    int nv = 0, rk = 0;
    if (vec) {
      for (int j = lane * 4; j < n_neg; j += 256) {
        if (j + 4 <= n_neg) {
          const f32x4 a = *reinterpret_cast<const f32x4*>(sr + j);
          const f32x4 f = *reinterpret_cast<const f32x4*>(fr + j);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const bool keep = !(f[e] > thres);
            tot += keep ? __builtin_amdgcn_exp2f((a[e] - 1.0f) * c2) : 0.f;
            nv += keep;
            rk += keep && a[e] > sp;
          }
        } else {
          for (int e = j; e < n_neg; ++e) {
            const bool keep = !(fr[e] > thres);
            tot += keep ? __builtin_amdgcn_exp2f((sr[e] - 1.0f) * c2) : 0.f;
            nv += keep;
            rk += keep && sr[e] > sp;
          }
        }
      }
    } else {
      for (int j = lane; j < n_neg; j += 64) {
        const bool keep = !(fr[j] > thres);
        tot += keep ? __builtin_amdgcn_exp2f((sr[j] - 1.0f) * c2) : 0.f;
        nv += keep;
        rk += keep && sr[j] > sp;
      }
    }
    tot = wave_sum(tot);
    nv = wave_sum_i(nv);
    rk = wave_sum_i(rk);
    if (lane == 0) {
      const bool live = row_base + row < n_live;
      const float l = scale + __logf(tot + __builtin_amdgcn_exp2f((sp - 1.0f) * c2));
      lse[row] = l;
      loss[row] = live ? l - scale * sp : 0.f;
      if (n_valid) n_valid[row] = live ? nv + 1 : 0;
      if (rank) rank[row] = live ? rk : 0;
    }
  }
}

__global__ __launch_bounds__(256) void nce_dense_bwd_kernel(const float* __restrict__ s, const float* __restrict__ fix, int64_t ld,
                                                            int n_neg, const float* __restrict__ lse, const float* __restrict__ w,
                                                            const float* __restrict__ scale_p, float thres,
                                                            const int32_t* __restrict__ n_live_p, int64_t row_base, int64_t rows,
                                                            bf16_t* __restrict__ g, int64_t ldg) {
  const float scale = scale_p[0], c2 = scale * LOG2E;
  const int64_t n_live = n_live_p ? (int64_t)n_live_p[0] : row_base + rows;
  const bool vec = (ld % 4 == 0) && (ldg % 4 == 0);
  WAVE_ROWS(rows) {
    const float* sr = s + row * ld;
    const float* fr = fix + row * ld;
    bf16_t* gr = g + row * ldg;
    const bool live = row_base + row < n_live;
    const float wr = live ? w[row] : 0.f, l2 = lse[row] * LOG2E;
    if (vec) {
      for (int j = lane * 4; j < n_neg; j += 256) {
        if (j + 4 <= n_neg) {
          const f32x4 a = *reinterpret_cast<const f32x4*>(sr + j);
          const f32x4 f = *reinterpret_cast<const f32x4*>(fr + j);
          bf16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            o[e] = (bf16_t)((!(f[e] > thres) && live) ? wr * __builtin_amdgcn_exp2f(a[e] * c2 - l2) : 0.f);
          *reinterpret_cast<bf16x4*>(gr + j) = o;
        } else {
          for (int e = j; e < n_neg; ++e)
            gr[e] = (bf16_t)((!(fr[e] > thres) && live) ? wr * __builtin_amdgcn_exp2f(sr[e] * c2 - l2) : 0.f);
        }
      }
    } else {
      for (int j = lane; j < n_neg; j += 64)
        gr[j] = (bf16_t)((!(fr[j] > thres) && live) ? wr * __builtin_amdgcn_exp2f(sr[j] * c2 - l2) : 0.f);
    }
  }
}

// ------------------------------------------------------------------------------------------
// REMI's interest-aware hard-negative loss over a dense logit chunk (reference model/IDNet/remi.py:203-288):
//   l_j = scale s_j, false negatives (fix_j > thres) dropped;  A = logsumexp_j((beta + 1) l_j),  Z = logsumexp_j(beta l_j);
//   log Neg = A - (Z - log n_neg)   (the mean runs over ALL n_neg sampled negatives, dropped ones counted);
//   lse = logaddexp(l+, log Neg),  loss = lse - l+.   Counters as in nce_dense_fwd (standard logits).
// Two passes per row (max, then shifted sums - the exact logsumexp of the reference, whatever the temperature); the row is
// 32 KiB and comes back from L2.  Backward: g_j = w sigma_neg ((beta + 1) exp((beta + 1) l_j - A) - beta exp(beta l_j - Z)),
// sigma_neg = exp(log Neg - lse): the bf16 tile the dQ / dN GEMMs of the wide path consume (same contract as nce_dense_bwd).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ihn_dense_fwd_kernel(const float* __restrict__ s, const float* __restrict__ fix, int64_t ld,
                                                            int n_neg, const float* __restrict__ s_pos,
                                                            const float* __restrict__ scale_p, float thres, float beta,
                                                            const int32_t* __restrict__ n_live_p, int64_t row_base, int64_t rows,
                                                            float* __restrict__ lse, float* __restrict__ log_num,
                                                            float* __restrict__ log_imp, float* __restrict__ loss,
                                                            int32_t* __restrict__ n_valid, int32_t* __restrict__ rank) {
  const float scale = scale_p[0];
  const int64_t n_live = n_live_p ? (int64_t)n_live_p[0] : row_base + rows;
  WAVE_ROWS(rows) {
    const float* sr = s + row * ld;
    const float* fr = fix + row * ld;
    const float sp = s_pos[row];
    float mx = -INFINITY;
    int nv = 0, rk = 0;
    for (int j = lane; j < n_neg; j += 64) {
      const bool keep = !(fr[j] > thres);
      mx = keep ? fmaxf(mx, sr[j]) : mx;
      nv += keep;
      rk += keep && sr[j] > sp;
    }
    mx = wave_max(mx);
    nv = wave_sum_i(nv);
    rk = wave_sum_i(rk);
    float t1 = 0.f, t0 = 0.f;
    if (mx > -INFINITY) {
      const float c1 = (beta + 1.0f) * scale * LOG2E, c0 = beta * scale * LOG2E;
      for (int j = lane; j < n_neg; j += 64) {
        if (!(fr[j] > thres)) {
          const float d = sr[j] - mx;
          t1 += __builtin_amdgcn_exp2f(d * c1);
          t0 += __builtin_amdgcn_exp2f(d * c0);
        }
      }
    }
    t1 = wave_sum(t1);
    t0 = wave_sum(t0);
    if (lane == 0) {
      const bool live = row_base + row < n_live;
      const float lp = scale * sp;
      float a = -INFINITY, z = -INFINITY, ln = -INFINITY;
      if (mx > -INFINITY) {
        a = (beta + 1.0f) * scale * mx + __logf(t1);
        z = beta * scale * mx + __logf(t0);
        ln = a - (z - __logf((float)n_neg));
      }
      const float hi = fmaxf(lp, ln);
      const float l = hi + __logf(__expf(lp - hi) + (ln > -INFINITY ? __expf(ln - hi) : 0.f));
      lse[row] = l;
      log_num[row] = a;
      log_imp[row] = z;
      loss[row] = live ? l - lp : 0.f;
      if (n_valid) n_valid[row] = live ? nv + 1 : 0;
      if (rank) rank[row] = live ? rk : 0;
    }
  }
}

__global__ __launch_bounds__(256) void ihn_dense_bwd_kernel(const float* __restrict__ s, const float* __restrict__ fix, int64_t ld,
                                                            int n_neg, const float* __restrict__ lse,
                                                            const float* __restrict__ log_num, const float* __restrict__ log_imp,
                                                            const float* __restrict__ w, const float* __restrict__ scale_p,
                                                            float thres, float beta, const int32_t* __restrict__ n_live_p,
                                                            int64_t row_base, int64_t rows, bf16_t* __restrict__ g, int64_t ldg) {
  const float scale = scale_p[0];
  const int64_t n_live = n_live_p ? (int64_t)n_live_p[0] : row_base + rows;
  WAVE_ROWS(rows) {
    const float* sr = s + row * ld;
    const float* fr = fix + row * ld;
    bf16_t* gr = g + row * ldg;
    const bool live = row_base + row < n_live;
    const float a = log_num[row], z = log_imp[row];
    const bool any = a > -INFINITY;
    const float ln = any ? a - (z - __logf((float)n_neg)) : -INFINITY;
    const float wn = (live && any) ? w[row] * __expf(ln - lse[row]) : 0.f;         // w sigma_neg
    const float b1 = beta + 1.0f;
    for (int j = lane; j < n_neg; j += 64) {
      float v = 0.f;
      if (wn != 0.f && !(fr[j] > thres)) {
        const float l = scale * sr[j];
        v = wn * (b1 * __expf(b1 * l - a) - beta * __expf(beta * l - z));
      }
      gr[j] = (bf16_t)v;
    }
  }
}

__device__ __forceinline__ bool item_ok(int item, const int32_t* tag_bits, int rb) {
  if (item == 0) return false;                                  // pad id (trainer.py:724)
  const int tb = tag_bits ? tag_bits[item] : (int)0x80000000;
  return (tb & rb) != 0;
}

// scores[r, j] = -inf where item (item_begin + j * item_stride) is not admissible for row r
__global__ __launch_bounds__(256) void catalog_mask_dense_kernel(float* __restrict__ sc, int64_t ld, int n_cols, int item_begin,
                                                                 int item_stride, const int32_t* __restrict__ tag_bits,
                                                                 const int32_t* __restrict__ row_bits, int n_rows) {
  const int row = blockIdx.y;
  const int rb = row_bits[row];
  float* r = sc + (int64_t)row * ld;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n_cols; j += gridDim.x * blockDim.x)
    if (!item_ok(item_begin + j * item_stride, tag_bits, rb)) r[j] = -INFINITY;
}

// One workgroup per (row, segment of the score chunk): candidates with score >= tau[row] that pass the masks go to list
// `list_base + blockIdx.x` of the row (append order arbitrary - the select sorts), the count (not clamped: the select flags
// overflow) to cand_cnt.
__global__ __launch_bounds__(256) void catalog_emit_dense_kernel(const float* __restrict__ sc, int64_t ld, int n_cols, int seg,
                                                                 int item_begin, const int32_t* __restrict__ tag_bits,
                                                                 const int32_t* __restrict__ row_bits, const float* __restrict__ tau,
                                                                 float* __restrict__ cand_val, int32_t* __restrict__ cand_idx,
                                                                 int32_t* __restrict__ cand_cnt, int n_lists, int list_base,
                                                                 int cap_s) {
  __shared__ int s_cnt;
  const int row = blockIdx.y, list = list_base + blockIdx.x;
  if (threadIdx.x == 0) s_cnt = 0;
  __syncthreads();
  const int rb = row_bits[row];
  const float t = tau[row];
  const float* r = sc + (int64_t)row * ld;
  const int j0 = blockIdx.x * seg, j1 = min(n_cols, j0 + seg);
  const int64_t base = ((int64_t)row * n_lists + list) * cap_s;
  if (rb != 0) {
    for (int j = j0 + threadIdx.x; j < j1; j += blockDim.x) {
      const float v = r[j];
      if (v >= t && item_ok(item_begin + j, tag_bits, rb)) {
        const int slot = atomicAdd(&s_cnt, 1);
        if (slot < cap_s) {
          cand_val[base + slot] = v;
          cand_idx[base + slot] = item_begin + j;
        }
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) cand_cnt[(int64_t)row * n_lists + list] = s_cnt;
}

}  // namespace

extern "C" int mhr_nce_dense_fwd(const float* neg_logits, const float* fix_logits, int64_t ld, int n_neg, const float* s_pos,
                                 const float* scale_dev, float thres, const int32_t* n_live_dev, int64_t row_base, int64_t rows,
                                 float* lse, float* loss, int32_t* n_valid, int32_t* rank, void* stream) {
  MHR_REQUIRE(neg_logits && fix_logits && s_pos && scale_dev && lse && loss, "nce_dense_fwd: null pointer");
  MHR_REQUIRE(n_neg > 0 && ld >= n_neg && rows >= 0, "nce_dense_fwd: bad sizes");
  if (rows == 0) return MHR_OK;
  hipLaunchKernelGGL(nce_dense_fwd_kernel, dim3(mhr_grid_for(rows, 4)), dim3(256), 0, (hipStream_t)stream, neg_logits, fix_logits,
                     ld, n_neg, s_pos, scale_dev, thres, n_live_dev, row_base, rows, lse, loss, n_valid, rank);
  MHR_CHECK_LAUNCH("nce_dense_fwd");
  return MHR_OK;
}

extern "C" int mhr_nce_dense_bwd(const float* neg_logits, const float* fix_logits, int64_t ld, int n_neg, const float* lse,
                                 const float* w, const float* scale_dev, float thres, const int32_t* n_live_dev, int64_t row_base,
                                 int64_t rows, void* g_bf16, int64_t ldg, void* stream) {
  MHR_REQUIRE(neg_logits && fix_logits && lse && w && scale_dev && g_bf16, "nce_dense_bwd: null pointer");
  MHR_REQUIRE(n_neg > 0 && ld >= n_neg && ldg >= n_neg && rows >= 0, "nce_dense_bwd: bad sizes");
  if (rows == 0) return MHR_OK;
  hipLaunchKernelGGL(nce_dense_bwd_kernel, dim3(mhr_grid_for(rows, 4)), dim3(256), 0, (hipStream_t)stream, neg_logits, fix_logits,
                     ld, n_neg, lse, w, scale_dev, thres, n_live_dev, row_base, rows, (bf16_t*)g_bf16, ldg);
  MHR_CHECK_LAUNCH("nce_dense_bwd");
  return MHR_OK;
}

extern "C" int mhr_ihn_dense_fwd(const float* neg_logits, const float* fix_logits, int64_t ld, int n_neg, const float* s_pos,
                                 const float* scale_dev, float thres, float beta, const int32_t* n_live_dev, int64_t row_base,
                                 int64_t rows, float* lse, float* log_num, float* log_imp, float* loss, int32_t* n_valid,
                                 int32_t* rank, void* stream) {
  MHR_REQUIRE(neg_logits && fix_logits && s_pos && scale_dev && lse && log_num && log_imp && loss, "ihn_dense_fwd: null pointer");
  MHR_REQUIRE(n_neg > 0 && ld >= n_neg && rows >= 0 && beta > 0.f, "ihn_dense_fwd: bad sizes / beta");
  if (rows == 0) return MHR_OK;
  hipLaunchKernelGGL(ihn_dense_fwd_kernel, dim3(mhr_grid_for(rows, 4)), dim3(256), 0, (hipStream_t)stream, neg_logits, fix_logits,
                     ld, n_neg, s_pos, scale_dev, thres, beta, n_live_dev, row_base, rows, lse, log_num, log_imp, loss, n_valid, rank);
  MHR_CHECK_LAUNCH("ihn_dense_fwd");
  return MHR_OK;
}

extern "C" int mhr_ihn_dense_bwd(const float* neg_logits, const float* fix_logits, int64_t ld, int n_neg, const float* lse,
                                 const float* log_num, const float* log_imp, const float* w, const float* scale_dev, float thres,
                                 float beta, const int32_t* n_live_dev, int64_t row_base, int64_t rows, void* g_bf16, int64_t ldg,
                                 void* stream) {
  MHR_REQUIRE(neg_logits && fix_logits && lse && log_num && log_imp && w && scale_dev && g_bf16, "ihn_dense_bwd: null pointer");
  MHR_REQUIRE(n_neg > 0 && ld >= n_neg && ldg >= n_neg && rows >= 0 && beta > 0.f, "ihn_dense_bwd: bad sizes / beta");
  if (rows == 0) return MHR_OK;
  hipLaunchKernelGGL(ihn_dense_bwd_kernel, dim3(mhr_grid_for(rows, 4)), dim3(256), 0, (hipStream_t)stream, neg_logits, fix_logits,
                     ld, n_neg, lse, log_num, log_imp, w, scale_dev, thres, beta, n_live_dev, row_base, rows, (bf16_t*)g_bf16, ldg);
  MHR_CHECK_LAUNCH("ihn_dense_bwd");
  return MHR_OK;
}

extern "C" int mhr_catalog_mask_dense(float* scores, int64_t ld, int n_cols, int item_begin, int item_stride,
                                      const int32_t* tag_bits, const int32_t* row_bits, int n_rows, void* stream) {
  MHR_REQUIRE(scores && row_bits, "catalog_mask_dense: null pointer");
  MHR_REQUIRE(n_cols > 0 && ld >= n_cols && n_rows > 0 && n_rows <= 65535 && item_stride >= 1, "catalog_mask_dense: bad sizes");
  hipLaunchKernelGGL(catalog_mask_dense_kernel, dim3(mhr_grid_for(n_cols, 1024, 64), n_rows), dim3(256), 0, (hipStream_t)stream,
                     scores, ld, n_cols, item_begin, item_stride, tag_bits, row_bits, n_rows);
  MHR_CHECK_LAUNCH("catalog_mask_dense");
  return MHR_OK;
}

extern "C" int mhr_catalog_emit_dense(const float* scores, int64_t ld, int n_cols, int seg, int item_begin, const int32_t* tag_bits,
                                      const int32_t* row_bits, const float* tau, int n_rows, float* cand_val, int32_t* cand_idx,
                                      int32_t* cand_cnt, int n_lists, int list_base, int cap_s, void* stream) {
  MHR_REQUIRE(scores && row_bits && tau && cand_val && cand_idx && cand_cnt, "catalog_emit_dense: null pointer");
  MHR_REQUIRE(n_cols > 0 && ld >= n_cols && seg > 0 && n_rows > 0 && n_rows <= 65535 && cap_s >= 1, "catalog_emit_dense: bad sizes");
  const int n_seg = (n_cols + seg - 1) / seg;
  MHR_REQUIRE(list_base >= 0 && list_base + n_seg <= n_lists, "catalog_emit_dense: lists %d..%d exceed n_lists=%d", list_base,
              list_base + n_seg, n_lists);
  hipLaunchKernelGGL(catalog_emit_dense_kernel, dim3(n_seg, n_rows), dim3(256), 0, (hipStream_t)stream, scores, ld, n_cols, seg,
                     item_begin, tag_bits, row_bits, tau, cand_val, cand_idx, cand_cnt, n_lists, list_base, cap_s);
  MHR_CHECK_LAUNCH("catalog_emit_dense");
  return MHR_OK;
}
