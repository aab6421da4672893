// Sampled-softmax logit contractions at feature dims beyond the register-stationary kernels (D = 512 ... 4096: HSTU size-4,
// the HLLM twin's TinyLlama / Baichuan2 widths): the LDS-tiled bf16 MFMA GEMM of the wide catalog scorer
// (csrc/catalog_wide.hip: 256 negatives x 128 tokens per workgroup, 64-feature chunks by LDS-DMA through a three-stage ring,
// four loader + four consumer waves, operands as packed tile images - mhr_pack_tiles) with the loss's arithmetic in its
// epilogue, so the [N_tok, n_neg] logit blocks never exist in memory.  (The library-GEMM form of round 2 wrote and re-read
// them as fp32 twice per direction: 4 x 268 MB per 8192 x 8192 block.)
//
// Reference path replaced (file:line under code/REC/): model/IDNet/hstu.py:600-619 = model/HLLM/hllm.py:378-397 (nce_loss:
// cosine logits against the negative pool, false-negative suppression where cos(target, negative) > 0.99, temperature) and
// F.cross_entropy (hstu.py:697, 833), top-k log counters hstu.py:621-629; backward: their autograd.
//
// Three epilogues on one core (template MODE), negatives on the accumulator registers, tokens on the lanes - a token's state is
// one register per lane:
//   MODE 0  false-negative bits:   rows = normalised TARGETS;  bit (token, negative) = cos > thres, 16 per lane and tile,
//           stored as they sit in the accumulator (the consumers below read them back in the same order);
//   MODE 1  forward:               rows = normalised QUERIES;  per token sum_j keep exp(scale (s_j - 1)), #kept, #{kept s_j > s+},
//           one partial per (item slice, consumer wave, lane half) - plain stores, summed in list order by nce_wide_finalize:
//           no atomics, bitwise reproducible;
//   MODE 2  softmax-gradient tile: rows = normalised QUERIES;  g_tj = keep w_t exp(scale s_tj - lse_t) as bf16 [N_tok, n_neg] -
//           the operand of the two gradient products dQ = G N, dN = G^T Q;
//   MODE 3  plain product:         out[r, i] (+)= alpha sum_k A[i, k] B[r, k] in fp32 - those two gradient products on the same
//           core (operands packed with mhr_pack_tiles / mhr_pack_tiles_t), so no library GEMM is left in the loss.
#include "mhr_common.h"
#include "stream_gemm.h"

namespace {

using WT = sg::Tile<4>;                        // 32 rows x 64 bf16 = 4 KB, XOR-swizzled
constexpr int WK = 64;                         // features per chunk
constexpr int W_ITEMS = 8 * WT::BYTES;         // 256 negative rows: 32 KB
constexpr int W_STAGE = 12 * WT::BYTES;        // + 128 token rows: 48 KB
constexpr int W_NST = 3;
constexpr int W_PW = 12;                       // 1-KiB DMA pieces per loader wave and chunk (48 per stage)
constexpr float LOG2E = 1.4426950408889634f;

template <int N, typename... V>
__device__ __forceinline__ void wait_lgkm_all(V&... v) {
  static_assert(N >= 0 && N <= 15, "lgkmcnt is 4 bits");
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N));
  (sg::redefine(v), ...);
}

struct WideArgs {
  // MODE 0
  float thres;
  uint16_t* bits_out;             // [n_neg_tiles * 2][t_pad]
  // MODE 1 / 2
  const uint16_t* bits;           // as written by MODE 0 (NULL: nothing suppressed)
  const float* s_pos;             // [t_pad] cos(query, target)
  const float* scale_p;           // device scalar: exp(logit_scale)
  float* part_tot;                // MODE 1: [n_lists][t_pad]
  int32_t* part_nv;
  int32_t* part_rk;
  const float* lse;               // MODE 2: [t_pad]
  const float* w;                 // MODE 2: [t_pad] per-token weight
  const int32_t* n_live_p;        // device scalar: live tokens (rows beyond get g = 0)
  bf16_t* g_out;                  // MODE 2: [t_pad, ldg]
  int64_t ldg;
  float* c_out;                   // MODE 3: [n_rows, ldc] fp32
  int64_t ldc;
  const float* alpha_dev;         // MODE 3: device scalar factor (NULL: 1)
  int accumulate;                 // MODE 3: c_out += instead of =
  int n_neg;                      // real negatives / valid "item" rows (the packed image is padded to whole 256-blocks)
  int64_t t_pad;                  // token rows the per-token arrays hold (R * 128)
};

template <int MODE>
__global__ __launch_bounds__(512, 1) void nce_wide_kernel(const unsigned char* __restrict__ rows_p, int n_rows,
                                                          const unsigned char* __restrict__ negs_p, int D, int n_blocks, int R, int U,
                                                          WideArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // XCD-aware decode of csrc/catalog_wide.hip: the 32 workgroups an XCD runs at a time form a U x S grid (U token blocks x S
  // negative slices), so per chunk step the XCD pulls S negative chunks + U token chunks through the fabric
  const int S = 32 / U, n_slices = 8 * S;
  const int w = blockIdx.x, xcd = w & 7, j = w >> 3;
  const int rt = (j >> 5) * U + (j % U), slice = ((j / U) % S) * 8 + xcd;
  if (rt >= R) return;                                        // (whole workgroup: before any barrier)
  const int bps = (n_blocks + n_slices - 1) / n_slices;
  const int b0 = slice * bps, b1 = min(n_blocks, b0 + bps);
  const int wave8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, r = lane & 31, half = lane >> 5;
  const bool loader = wave8 >= 4;                             // waves 4..7: LDS-DMA only
  const int wave = wave8 & 3;
  const int wm = wave >> 1, wn = wave & 1;                    // consumer quadrant: 128-negative strip, 64-token strip
  const int n_lists = 4 * n_slices;

  int tok[2];
  float sp[2] = {0.f, 0.f}, tot[2] = {0.f, 0.f}, l2[2] = {0.f, 0.f}, wt[2] = {0.f, 0.f};
  int nv[2] = {0, 0}, rk[2] = {0, 0};
  float c2 = 0.f;
  if (MODE == 1 || MODE == 2) c2 = a.scale_p[0] * LOG2E;
  if (MODE == 3) c2 = a.alpha_dev ? a.alpha_dev[0] : 1.0f;
#pragma unroll
  for (int jt = 0; jt < 2; ++jt) {
    tok[jt] = rt * 128 + wn * 64 + jt * 32 + r;
    if (MODE == 1) sp[jt] = a.s_pos[tok[jt]];
    if (MODE == 2) {
      const bool live = tok[jt] < a.n_live_p[0];
      l2[jt] = a.lse[tok[jt]] * LOG2E;
      wt[jt] = live ? a.w[tok[jt]] : 0.f;
    }
  }

  if (b0 < b1) {
    const int KC = D / WK, NC = (b1 - b0) * KC;
    const uint32_t lane16 = lane * 16;
    auto piece = [&](auto i_c, int c, int stage) __attribute__((always_inline)) {
      constexpr int i = decltype(i_c)::value;
      const int blk = b0 + c / KC, kc = c - (c / KC) * KC;
      const int pc = wave * W_PW + i;                              // wave-uniform piece index
      unsigned char* dst = smem + stage * W_STAGE + pc * 1024;
      const unsigned char* src = pc < 32 ? negs_p + (((int64_t)blk * KC + kc) * 32 + pc) * 1024
                                         : rows_p + (((int64_t)rt * KC + kc) * 16 + (pc - 32)) * 1024;
      __builtin_amdgcn_global_load_lds((sg::gptr_t)(src + lane16), (sg::lptr_t)dst, 16, 0, 0);
    };
    auto issue = [&](int c, int stage) __attribute__((always_inline)) {
      auto f = [&](auto i_c) __attribute__((always_inline)) { piece(i_c, c, stage); };
      sg::static_for<W_PW>(f);
    };

    sg::LaneAddr<4> la;
    la.init(lane);
    uint32_t ra_i[4], ra_u[4];
    const uint32_t base = sg::lds_addr(smem);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      ra_i[ks] = base + wm * 4 * WT::BYTES + (uint32_t)la.a[ks];
      ra_u[ks] = base + W_ITEMS + wn * 2 * WT::BYTES + (uint32_t)la.a[ks];
    }
    // the per-token state was loaded above: settle it now (see catalog_wide.hip: otherwise an `s_waitcnt vmcnt(0)` lands in
    // the epilogue, or - in the loader waves - in front of the ring's counted waits)
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(sp[0]), "+v"(sp[1]), "+v"(l2[0]), "+v"(l2[1]), "+v"(wt[0]), "+v"(wt[1]), "+v"(c2));

    if (loader) {
      issue(0, 0);
      issue(min(1, NC - 1), 1);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W_PW) : "memory");              // chunk 0 has landed
      for (int c = 0; c < NC; ++c) {
        sg::ring_barrier();                                    // consumers are done with chunk c - 1's stage; chunk c is theirs
        issue(min(c + 2, NC - 1), (c + 2) % W_NST);            // (redundant tail copies keep the DMA count uniform)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W_PW) : "memory");            // chunk c + 1 has landed before the next barrier
      }
      sg::wait_vmcnt<0>();
    } else {
      f32x16 acc[4][2];
      uint16_t mbits[4][2] = {};
      sg::ring_loop<W_NST>(NC, [&](auto stage_c, int c) __attribute__((always_inline)) {
        constexpr int st = decltype(stage_c)::value;
        sg::ring_barrier();                                    // the loaders have landed chunk c
        const int kc = c % KC;
        if (kc == 0) {
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) acc[i][jt] = sg::zero16();
          if ((MODE == 1 || MODE == 2) && a.bits) {            // the block's suppression words: in flight underneath its K loop
            const int blk0 = b0 + c / KC;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
              for (int jt = 0; jt < 2; ++jt)
                mbits[i][jt] = a.bits[((int64_t)(blk0 * 8 + wm * 4 + i) * 2 + half) * a.t_pad + tok[jt]];
          }
        }
        sg::u32x4 fa[2][4], fb[2][2];
        auto rd = [&](auto ks_c) __attribute__((always_inline)) {
          constexpr int ks = decltype(ks_c)::value;
          const uint32_t ai = ra_i[ks] + st * W_STAGE, au = ra_u[ks] + st * W_STAGE;
          fa[ks & 1][0] = sg::ds_read_b128_asm<0 * WT::BYTES>(ai);
          fa[ks & 1][1] = sg::ds_read_b128_asm<1 * WT::BYTES>(ai);
          fa[ks & 1][2] = sg::ds_read_b128_asm<2 * WT::BYTES>(ai);
          fa[ks & 1][3] = sg::ds_read_b128_asm<3 * WT::BYTES>(ai);
          fb[ks & 1][0] = sg::ds_read_b128_asm<0 * WT::BYTES>(au);
          fb[ks & 1][1] = sg::ds_read_b128_asm<1 * WT::BYTES>(au);
        };
        rd(std::integral_constant<int, 0>{});
        auto step = [&](auto ks_c) __attribute__((always_inline)) {
          constexpr int ks = decltype(ks_c)::value, b = ks & 1;
          if constexpr (ks + 1 < 4) {
            rd(std::integral_constant<int, ks + 1>{});
            wait_lgkm_all<6>(fa[b][0], fa[b][1], fa[b][2], fa[b][3], fb[b][0], fb[b][1]);
          } else {
            wait_lgkm_all<0>(fa[b][0], fa[b][1], fa[b][2], fa[b][3], fb[b][0], fb[b][1]);
          }
          auto mm = [&](auto i_c) __attribute__((always_inline)) {
            constexpr int i = decltype(i_c)::value;
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
              acc[i][jt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[b][i]), __builtin_bit_cast(bf16x8, fb[b][jt]),
                                                                   acc[i][jt], 0, 0, 0);
          };
          sg::static_for<4>(mm);
          __builtin_amdgcn_sched_barrier(0);
        };
        sg::static_for<4>(step);

        if (kc == KC - 1) {                                    // the block's cosines are complete
          const int blk = b0 + c / KC;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int tile = blk * 8 + wm * 4 + i;             // 32-negative tile; this lane sees rows crow(g, half) of it
            const int n0 = tile * 32;
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) {
              const int64_t bo = ((int64_t)tile * 2 + half) * a.t_pad + tok[jt];
              if (MODE == 3) {
                float* crow_ = a.c_out + (int64_t)tok[jt] * a.ldc + n0 + 4 * half;
#pragma unroll
                for (int q = 0; q < 4; ++q) {                  // registers 4q .. 4q + 3 = columns n0 + 8q + 4 half + (0..3)
                  if (n0 + 8 * q + 4 * half < a.n_neg && tok[jt] < n_rows) {
                    f32x4 o = {acc[i][jt][4 * q] * c2, acc[i][jt][4 * q + 1] * c2, acc[i][jt][4 * q + 2] * c2, acc[i][jt][4 * q + 3] * c2};
                    f32x4* dst = reinterpret_cast<f32x4*>(crow_ + 8 * q);
                    if (a.accumulate) o += *dst;
                    *dst = o;
                  }
                }
              } else if (MODE == 0) {
                uint32_t m = 0;
#pragma unroll
                for (int g = 0; g < 16; ++g) m |= (acc[i][jt][g] > a.thres ? 1u : 0u) << g;
                a.bits_out[bo] = (uint16_t)m;
              } else {
                const uint32_t m = (uint32_t)mbits[i][jt];
                if (MODE == 1) {
                  if (m == 0u && n0 + 32 <= a.n_neg) {         // nothing suppressed, no padding negative (almost every tile)
#pragma unroll
                    for (int g = 0; g < 16; ++g) {
                      const float s = acc[i][jt][g];
                      tot[jt] += __builtin_amdgcn_exp2f(__builtin_fmaf(s, c2, -c2));
                      rk[jt] += (s > sp[jt]) ? 1 : 0;
                    }
                    nv[jt] += 16;
                  } else {
#pragma unroll
                    for (int g = 0; g < 16; ++g) {
                      const float s = acc[i][jt][g];
                      const bool keep = !((m >> g) & 1u) && (n0 + sg::crow(g, half) < a.n_neg);
                      tot[jt] += keep ? __builtin_amdgcn_exp2f(__builtin_fmaf(s, c2, -c2)) : 0.f;
                      nv[jt] += keep ? 1 : 0;
                      rk[jt] += (keep && s > sp[jt]) ? 1 : 0;
                    }
                  }
                } else {
                  bf16_t* grow = a.g_out + (int64_t)tok[jt] * a.ldg + n0 + 4 * half;
#pragma unroll
                  for (int q = 0; q < 4; ++q) {                // registers 4q .. 4q + 3 = negatives n0 + 8q + 4 half + (0..3)
                    bf16x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                      const int g = 4 * q + e;
                      const bool keep = !((m >> g) & 1u) && (n0 + 8 * q + 4 * half + e < a.n_neg);     // (padding negatives: g = 0)
                      o[e] = (bf16_t)(keep ? wt[jt] * __builtin_amdgcn_exp2f(acc[i][jt][g] * c2 - l2[jt]) : 0.f);
                    }
                    if (n0 + 8 * q + 4 * half < a.ldg && tok[jt] < n_rows) *reinterpret_cast<bf16x4*>(grow + 8 * q) = o;
                  }
                }
              }
            }
          }
        }
      });
    }
  }
  if (MODE == 1 && !loader) {
    // every (slice, negative strip, lane half) owns its own partial of a token: plain stores (zeros where the slice had no block)
    const int list = slice * 4 + wm * 2 + half;
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
      const int64_t o = (int64_t)list * a.t_pad + tok[jt];
      a.part_tot[o] = tot[jt];
      a.part_nv[o] = nv[jt];
      a.part_rk[o] = rk[jt];
    }
  }
  (void)n_rows;
  (void)n_lists;
}

// lse / loss / counters of a token from its partials, summed in list order (reference: logsumexp over [s+, kept negatives] x
// scale, cross entropy with the positive at index 0, hstu.py:612-619 + 697; counters hstu.py:621-629)
__global__ __launch_bounds__(256) void nce_wide_finalize_kernel(const float* __restrict__ part_tot, const int32_t* __restrict__ part_nv,
                                                                const int32_t* __restrict__ part_rk, int n_lists, int64_t t_pad,
                                                                const float* __restrict__ s_pos, const float* __restrict__ scale_p,
                                                                const int32_t* __restrict__ n_live_p, int64_t rows,
                                                                float* __restrict__ lse, float* __restrict__ loss,
                                                                int32_t* __restrict__ n_valid, int32_t* __restrict__ rank) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= rows) return;
  float tot = 0.f;
  int nv = 0, rk = 0;
  for (int l = 0; l < n_lists; ++l) {
    tot += part_tot[(int64_t)l * t_pad + t];
    nv += part_nv[(int64_t)l * t_pad + t];
    rk += part_rk[(int64_t)l * t_pad + t];
  }
  const float scale = scale_p[0], sp = s_pos[t];
  const bool live = t < (int64_t)n_live_p[0];
  const float l = scale + __logf(tot + __builtin_amdgcn_exp2f((sp - 1.0f) * scale * LOG2E));
  lse[t] = l;
  loss[t] = live ? l - scale * sp : 0.f;
  if (n_valid) n_valid[t] = live ? nv + 1 : 0;
  if (rank) rank[t] = live ? rk : 0;
}

int launch_geometry(int n_rows, int& R, int& U) {
  R = (n_rows + 127) / 128;
  U = 1;
  while (U < 8 && 2 * U <= R) U *= 2;
  return 8 * 32 * ((R + U - 1) / U);
}

}  // namespace

extern "C" int mhr_catalog_wide_slices(int n_rows);

// rows_packed: mhr_pack_tiles(x [n_rows, dim] bf16 L2-normalised, tiles_per_block = 4); negs_packed: mhr_pack_tiles(negatives
// [n_neg, dim] bf16 L2-normalised, tiles_per_block = 8).  Per-token arrays hold t_pad = ceil(n_rows / 128) * 128 entries.
extern "C" int mhr_nce_wide_fix_bits(const void* targets_packed, int n_rows, const void* negs_packed, int n_neg, int dim, float thres,
                                     uint16_t* bits, void* stream) {
  MHR_REQUIRE(targets_packed && negs_packed && bits, "nce_wide_fix_bits: null pointer");
  MHR_REQUIRE(dim >= 64 && dim % 64 == 0 && dim <= 8192 && n_rows > 0 && n_neg > 0, "nce_wide_fix_bits: bad sizes (dim=%d)", dim);
  int R, U;
  const int grid = launch_geometry(n_rows, R, U);
  WideArgs a = {};
  a.thres = thres;
  a.bits_out = bits;
  a.n_neg = n_neg;
  a.t_pad = (int64_t)R * 128;
  const size_t lds = (size_t)W_NST * W_STAGE;
  auto kern = nce_wide_kernel<0>;
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, (hipStream_t)stream, (const unsigned char*)targets_packed, n_rows,
                     (const unsigned char*)negs_packed, dim, (n_neg + 255) / 256, R, U, a);
  MHR_CHECK_LAUNCH("nce_wide_fix_bits");
  return MHR_OK;
}

extern "C" int mhr_nce_wide_fwd(const void* queries_packed, int n_rows, const void* negs_packed, int n_neg, int dim,
                                const uint16_t* bits, const float* s_pos, const float* scale_dev, const int32_t* n_live_dev,
                                float* part_tot, int32_t* part_nv, int32_t* part_rk, float* lse, float* loss, int32_t* n_valid,
                                int32_t* rank, void* stream) {
  MHR_REQUIRE(queries_packed && negs_packed && s_pos && scale_dev && n_live_dev && part_tot && part_nv && part_rk && lse && loss,
              "nce_wide_fwd: null pointer");
  MHR_REQUIRE(dim >= 64 && dim % 64 == 0 && dim <= 8192 && n_rows > 0 && n_neg > 0, "nce_wide_fwd: bad sizes (dim=%d)", dim);
  int R, U;
  const int grid = launch_geometry(n_rows, R, U);
  WideArgs a = {};
  a.bits = bits;
  a.s_pos = s_pos;
  a.scale_p = scale_dev;
  a.part_tot = part_tot;
  a.part_nv = part_nv;
  a.part_rk = part_rk;
  a.n_neg = n_neg;
  a.t_pad = (int64_t)R * 128;
  const size_t lds = (size_t)W_NST * W_STAGE;
  auto kern = nce_wide_kernel<1>;
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, s, (const unsigned char*)queries_packed, n_rows,
                     (const unsigned char*)negs_packed, dim, (n_neg + 255) / 256, R, U, a);
  MHR_CHECK_LAUNCH("nce_wide_fwd");
  const int n_lists = 4 * mhr_catalog_wide_slices(n_rows);
  hipLaunchKernelGGL(nce_wide_finalize_kernel, dim3((n_rows + 255) / 256), dim3(256), 0, s, part_tot, part_nv, part_rk, n_lists,
                     a.t_pad, s_pos, scale_dev, n_live_dev, (int64_t)n_rows, lse, loss, n_valid, rank);
  MHR_CHECK_LAUNCH("nce_wide_fwd (finalize)");
  return MHR_OK;
}

extern "C" int mhr_nce_wide_grad_tile(const void* queries_packed, int n_rows, const void* negs_packed, int n_neg, int dim,
                                      const uint16_t* bits, const float* lse, const float* w, const float* scale_dev,
                                      const int32_t* n_live_dev, void* g_bf16, int64_t ldg, void* stream) {
  MHR_REQUIRE(queries_packed && negs_packed && lse && w && scale_dev && n_live_dev && g_bf16, "nce_wide_grad_tile: null pointer");
  MHR_REQUIRE(dim >= 64 && dim % 64 == 0 && dim <= 8192 && n_rows > 0 && n_neg > 0, "nce_wide_grad_tile: bad sizes (dim=%d)", dim);
  MHR_REQUIRE(ldg >= n_neg && ldg % 4 == 0 && (uintptr_t)g_bf16 % 8 == 0, "nce_wide_grad_tile: ldg must be a multiple of 4 >= n_neg");
  int R, U;
  const int grid = launch_geometry(n_rows, R, U);
  WideArgs a = {};
  a.bits = bits;
  a.scale_p = scale_dev;
  a.lse = lse;
  a.w = w;
  a.n_live_p = n_live_dev;
  a.g_out = (bf16_t*)g_bf16;
  a.ldg = ldg;
  a.n_neg = n_neg;
  a.t_pad = (int64_t)R * 128;
  const size_t lds = (size_t)W_NST * W_STAGE;
  auto kern = nce_wide_kernel<2>;
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, (hipStream_t)stream, (const unsigned char*)queries_packed, n_rows,
                     (const unsigned char*)negs_packed, dim, (n_neg + 255) / 256, R, U, a);
  MHR_CHECK_LAUNCH("nce_wide_grad_tile");
  return MHR_OK;
}

// out[r, i] (+)= alpha * sum_k A[i, k] B[r, k], fp32 accumulation of bf16 operands: the plain products of the wide loss backward.
// a_packed: the n_i "item" rows (mhr_pack_tiles / mhr_pack_tiles_t with tiles_per_block = 8); b_packed: the n_r rows
// (tiles_per_block = 4); k_dim: the packed contraction length (a multiple of 64).  n_i % 4 == 0, ldc % 4 == 0, out 16-byte aligned.
extern "C" int mhr_wide_gemm_nt(const void* a_packed, int n_i, const void* b_packed, int n_r, int k_dim, const float* alpha_dev,
                                float* out, int64_t ldc, int accumulate, void* stream) {
  MHR_REQUIRE(a_packed && b_packed && out, "wide_gemm_nt: null pointer");
  MHR_REQUIRE(k_dim >= 64 && k_dim % 64 == 0 && n_i > 0 && n_i % 4 == 0 && n_r > 0 && ldc >= n_i && ldc % 4 == 0 &&
              (uintptr_t)out % 16 == 0, "wide_gemm_nt: bad sizes (k_dim=%d n_i=%d ldc=%lld)", k_dim, n_i, (long long)ldc);
  int R, U;
  const int grid = launch_geometry(n_r, R, U);
  WideArgs a = {};
  a.c_out = out;
  a.ldc = ldc;
  a.alpha_dev = alpha_dev;
  a.accumulate = accumulate ? 1 : 0;
  a.n_neg = n_i;
  a.t_pad = (int64_t)R * 128;
  const size_t lds = (size_t)W_NST * W_STAGE;
  auto kern = nce_wide_kernel<3>;
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, (hipStream_t)stream, (const unsigned char*)b_packed, n_r,
                     (const unsigned char*)a_packed, k_dim, (n_i + 255) / 256, R, U, a);
  MHR_CHECK_LAUNCH("wide_gemm_nt");
  return MHR_OK;
}
