// Dense scoring of a FEW (user, head) rows against the whole catalog, every score kept - the exact re-run of the rows the
// threshold scorers could not certify (ops.catalog_topk_exact: a margin set of more than k2 near-ties, e.g. duplicated /
// collapsed item rows; wide.py: rows whose candidate lists overflowed) and small catalogs at feature dims the MFMA scorers
// do not cover.
//
// Reference path (file:line under code/REC/): model/IDNet/hstu.py:965-1015 (fp32 normalise + fp32 score matmul, tag /
// given-prior -inf masks), trainer/trainer.py:724-726 (pad id and history suppression), evaluator/collector.py:245
// (torch.topk over the row).  The reference does this for EVERY row on a [B,H,N] tensor; here it is the rare path, so it is
// plain HBM-bound work: the item table streams once per group of RB rows (16 bytes per lane, one item row per wave
// iteration), the RB user rows sit in LDS as fp32, products accumulate in fp32 (operands fp32 or bf16), the masks are
// applied in place and the scores land in the list format of mhr_topk_select (cap = n_items), which ranks them
// (value desc, index asc) like every other path.
#include "mhr_common.h"

namespace {

template <typename T, int RB>
__global__ __launch_bounds__(256) void score_rows_dense_kernel(
    const T* __restrict__ users, const T* __restrict__ items, int dim, int64_t n_items, const int32_t* __restrict__ row_list,
    int n_list, int H, const uint32_t* __restrict__ tag_bits, const uint32_t* __restrict__ row_bits,
    const int32_t* __restrict__ hist_ptr, const int64_t* __restrict__ hist_items, float* __restrict__ out_val,
    int32_t* __restrict__ out_idx, int32_t* __restrict__ out_cnt) {
  extern __shared__ __attribute__((aligned(16))) float u_lds[];          // [RB][dim]
  const int g = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r0 = g * RB;
  for (int i = threadIdx.x * 4; i < RB * dim; i += 256 * 4) {
    const int r = i / dim, c = i - r * dim;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (r0 + r < n_list) v = Vec4IO<T>::load(users + (int64_t)row_list[r0 + r] * dim + c);
    *reinterpret_cast<f32x4*>(u_lds + i) = v;
  }
  __syncthreads();

  // lane r < RB owns the masks and the store of row r0 + r
  const bool owner = lane < RB && r0 + lane < n_list;
  const int my_row = owner ? row_list[r0 + lane] : 0;
  const uint32_t my_bits = owner ? row_bits[my_row] : 0u;
  int hp0 = 0, hp1 = 0;
  if (owner && hist_ptr) {
    hp0 = hist_ptr[my_row / H];
    hp1 = hist_ptr[my_row / H + 1];
  }
  if (owner && blockIdx.x == 0 && wave == 0) out_cnt[r0 + lane] = (int32_t)n_items;

  for (int64_t n = (int64_t)blockIdx.x * 4 + wave; n < n_items; n += (int64_t)gridDim.x * 4) {
    float acc[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) acc[r] = 0.f;
    const T* it = items + n * dim;
    for (int c = lane * 4; c < dim; c += 256) {
      const f32x4 b = Vec4IO<T>::load(it + c);
#pragma unroll
      for (int r = 0; r < RB; ++r) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(u_lds + r * dim + c);
        acc[r] += a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
      }
    }
    float mine = 0.f;
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const float s = wave_sum(acc[r]);
      if (lane == r) mine = s;
    }
    if (owner) {
      const uint32_t tb = tag_bits ? tag_bits[n] : 0x80000000u;
      bool ok = n != 0 && (tb & my_bits) != 0u;                 // n == 0: the pad id (trainer.py:724)
      if (ok && hist_items) {                                   // trainer.py:725-726: the user's own history
        int lo = hp0, hi = hp1;
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if (hist_items[mid] < n) lo = mid + 1;
          else hi = mid;
        }
        ok = !(lo < hp1 && hist_items[lo] == n);
      }
      out_val[(int64_t)(r0 + lane) * n_items + n] = ok ? mine : -INFINITY;
      out_idx[(int64_t)(r0 + lane) * n_items + n] = (int32_t)n;
    }
  }
}

}  // namespace

extern "C" int mhr_catalog_score_rows_dense(const void* users, const void* items, int dtype, int dim, int64_t n_items,
                                            const int32_t* row_list, int n_list, int H, const uint32_t* tag_bits,
                                            const uint32_t* row_bits, const int32_t* hist_ptr, const int64_t* hist_items,
                                            float* out_val, int32_t* out_idx, int32_t* out_cnt, void* stream) {
  MHR_REQUIRE(users && items && row_list && row_bits && out_val && out_idx && out_cnt, "catalog_score_rows_dense: null pointer");
  MHR_REQUIRE(dtype == MHR_F32 || dtype == MHR_BF16, "catalog_score_rows_dense: dtype %d unsupported", dtype);
  MHR_REQUIRE(dim > 0 && dim % 4 == 0 && dim <= 4096, "catalog_score_rows_dense: dim=%d unsupported (multiple of 4, <= 4096)", dim);
  MHR_REQUIRE(n_items > 0 && n_items < (1ll << 31) && n_list >= 0 && H > 0, "catalog_score_rows_dense: bad sizes");
  MHR_REQUIRE((hist_ptr == nullptr) == (hist_items == nullptr), "catalog_score_rows_dense: hist_ptr/hist_items must both be set or null");
  if (n_list == 0) return MHR_OK;
  hipStream_t s = (hipStream_t)stream;
  const int RB = dim <= 2048 ? 8 : 4;
  const int groups = (n_list + RB - 1) / RB;
  int gx = (int)((n_items + 3) / 4);
  const int want = (2048 + groups - 1) / groups;              // ~8 workgroups per CU over all row groups
  if (gx > want) gx = want;
  if (gx < 1) gx = 1;
  const size_t lds = (size_t)RB * dim * sizeof(float);
#define L_(T, RB_)                                                                                                        \
  hipLaunchKernelGGL((score_rows_dense_kernel<T, RB_>), dim3(gx, groups), dim3(256), lds, s, (const T*)users, (const T*)items, \
                     dim, n_items, row_list, n_list, H, tag_bits, row_bits, hist_ptr, hist_items, out_val, out_idx, out_cnt)
  if (dtype == MHR_F32) {
    if (RB == 8) L_(float, 8); else L_(float, 4);
  } else {
    if (RB == 8) L_(bf16_t, 8); else L_(bf16_t, 4);
  }
#undef L_
  MHR_CHECK_LAUNCH("catalog_score_rows_dense");
  return MHR_OK;
}

extern "C" int64_t mhr_catalog_score_rows_dense_workspace_bytes(int n_list, int64_t n_items) {
  if (n_list <= 0 || n_items <= 0) return 0;
  return (int64_t)n_list * n_items * 8 + (int64_t)n_list * 4;   // out_val f32 + out_idx i32 per (row, item), out_cnt i32 per row
}
