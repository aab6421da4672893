// Full-catalog multi-head scoring at feature dims beyond the register-stationary scorer (D = 512 ... 4096: HSTU size-4,
// the HLLM twin's TinyLlama / Baichuan2 user towers): an LDS-tiled bf16 MFMA GEMM with the threshold emit fused into its
// epilogue, so the [B*H, N] score block never exists in memory (the library-GEMM form wrote and re-read it as fp32:
// 8 B per score, 15 GB per 256-user batch at the TinyLlama shape).
//
// Reference path replaced (file:line under code/REC/): model/IDNet/hstu.py:965-1015 = model/HLLM/hllm.py:838-883 (scores,
// tag / given-prior masks), trainer/trainer.py:724-726 (pad suppression; history in the select kernel),
// evaluator/collector.py:245 (top-k: mhr_topk_select_sliced on the lists written here).
//
// Mapping.  One 512-thread workgroup = 256 items x 128 (user, head) rows per macro tile, in two roles (two waves per
// SIMD): four CONSUMER waves own 128 x 64 quadrants as 4 x 2 accumulator tiles of v_mfma_f32_32x32x16_bf16 (128
// registers) and issue nothing but LDS reads and MFMAs; four LOADER waves do nothing but LDS-DMA.  (Measured: with the
// DMA issued by the MFMA waves themselves - up front or spread over the MFMA gaps - each of the 13 pieces per chunk
// stalled the wave's issue for 100+ cycles: 28 % resp. 18 % of the MFMA peak.)  The contraction runs in chunks of 64
// features: per chunk the 256 item rows and the 128 user rows arrive in LDS as swizzled 32-row tile images
// (sg::Tile<4>) by LDS-DMA (global_load_lds_dwordx4, no registers), three stages of 48 KB, two chunks in flight behind
// the one being consumed; one raw s_barrier per chunk hands a landed stage to the consumers and a drained one back to
// the loaders.  Per k-step a consumer reads 4 item + 2 user fragments (ds_read_b128, inline asm with counted lgkmcnt
// waits, one k-step ahead) for 8 MFMAs.  A workgroup walks a SLICE of the
// catalog (persistent over its item blocks), the user block is re-streamed from L2, and the workgroups that share an item
// slice sit on one XCD so the slice leaves HBM once.  Accumulators hold S[item, row]: items on the registers, rows on the
// lanes - per-row state (threshold, category bit, list fill counts) is one register per lane and column tile, and a hit
// costs two plain stores.
#include "mhr_common.h"
#include "stream_gemm.h"

namespace {

using WT = sg::Tile<4>;                        // 32 rows x 64 bf16 = 4 KB, XOR-swizzled
constexpr int WK = 64;                         // features per chunk
constexpr int W_ITEMS = 8 * WT::BYTES;         // 256 item rows: 32 KB
constexpr int W_STAGE = 12 * WT::BYTES;        // + 128 user rows: 48 KB
constexpr int W_NST = 3;
constexpr int W_TAGS = W_NST * W_STAGE;        // two 1 KB tag-word buffers behind the stages
constexpr int W_PW = 12;                       // 1-KiB DMA pieces per wave and chunk (48 per stage)

// Operand layout of the wide scorer: rows are PACKED once into the LDS tile images themselves - [block][64-feature chunk]
// [tile of 32 rows][4 KB swizzled image] - so that a (block, chunk) is ONE contiguous run (32 KB of items, 16 KB of user
// rows) and an LDS-DMA piece is a straight 1 KB copy.  Streamed from the row-major table the same chunk is 256 separate
// 128-byte segments at a power-of-two pitch (4 KB at D = 2048): measured 25 GB/s per CU from an 81 %-hit L2, a third of
// what contiguous fills reach.  The catalog is packed when the evaluation caches it; the user rows once per batch.
__global__ __launch_bounds__(256) void pack_tiles_kernel(const bf16_t* __restrict__ x, int64_t n_rows, int D, int64_t row_begin,
                                                         int64_t row_stride, int64_t n_sel, int TB, unsigned char* __restrict__ out) {
  const int KC = D / WK;
  const int64_t n_tiles = (n_sel + 31) / 32, n_blocks = (n_tiles + TB - 1) / TB;
  const int64_t total = n_blocks * KC * TB * (WT::BYTES / 16);                 // 16-byte chunks of the packed image
  for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < total; c += (int64_t)gridDim.x * blockDim.x) {
    const int slot = (int)(c & 7), row = (int)((c >> 3) & 31);                // position inside the 4 KB tile image
    int64_t t = c >> 8;                                                        // tile image index: ((block * KC + kc) * TB + tb)
    const int tb = (int)(t % TB);
    t /= TB;
    const int kc = (int)(t % KC);
    const int64_t blk = t / KC;
    const int64_t sel = (blk * TB + tb) * 32 + row;                            // selected-row index
    const int chunk = slot ^ WT::key(row);                                     // the 8-feature chunk this slot holds
    bf16x8 v = sg::zero8();
    if (sel < n_sel) {
      const int64_t src = row_begin + sel * row_stride;
      if (src < n_rows) v = *reinterpret_cast<const bf16x8*>(x + src * D + kc * WK + chunk * 8);
    }
    *reinterpret_cast<bf16x8*>(out + c * 16) = v;
  }
}

// The same packing for an operand that is stored TRANSPOSED: packed row j = column j of x [n_k, ld] (the contraction index runs
// down x's rows).  Consecutive threads take consecutive packed rows, so the eight strided 2-byte reads of a 16-byte chunk are
// coalesced across the threads; contraction indices beyond n_k pack as zeros (K is padded to whole 64-feature chunks).
__global__ __launch_bounds__(256) void pack_tiles_t_kernel(const bf16_t* __restrict__ x, int64_t n_k, int64_t ld, int64_t n_sel, int TB,
                                                           unsigned char* __restrict__ out) {
  const int KC = (int)((n_k + WK - 1) / WK);
  const int64_t n_tiles = (n_sel + 31) / 32, n_blocks = (n_tiles + TB - 1) / TB;
  const int64_t total = n_blocks * KC * TB * (WT::BYTES / 16);
  for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < total; c += (int64_t)gridDim.x * blockDim.x) {
    const int row = (int)(c & 31), slot = (int)((c >> 5) & 7);                 // row fastest: coalesced reads of x[k, sel]
    int64_t t = c >> 8;
    const int tb = (int)(t % TB);
    t /= TB;
    const int kc = (int)(t % KC);
    const int64_t blk = t / KC;
    const int64_t sel = (blk * TB + tb) * 32 + row;
    const int chunk = slot ^ WT::key(row);
    bf16x8 v = sg::zero8();
    if (sel < n_sel) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int64_t k = (int64_t)kc * WK + chunk * 8 + e;
        if (k < n_k) v[e] = x[k * ld + sel];
      }
    }
    *reinterpret_cast<bf16x8*>(out + ((c >> 8) * 256 + (int64_t)row * 8 + slot) * 16) = v;
  }
}

template <int N, typename... V>
__device__ __forceinline__ void wait_lgkm_all(V&... v) {
  static_assert(N >= 0 && N <= 15, "lgkmcnt is 4 bits");
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N));
  (sg::redefine(v), ...);
}

__global__ __launch_bounds__(512, 1) void catalog_emit_wide_kernel(
    const unsigned char* __restrict__ users_p, int n_rows, const unsigned char* __restrict__ items_p, int64_t n_items, int D,
    int64_t item_begin, int64_t item_stride, int n_blocks, int R, int U, const uint32_t* __restrict__ tag_bits,
    const uint32_t* __restrict__ row_bits, const float* __restrict__ tau, float* __restrict__ cand_val,
    int32_t* __restrict__ cand_idx, int32_t* __restrict__ cand_cnt, int cap_s) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  // XCD-aware decode.  Workgroups with equal blockIdx % 8 share an XCD and its 4 MB L2, 32 of them at a time (one per CU).
  // Those 32 form a U x S grid (U row blocks x S item slices, U S = 32): the S workgroups of a row block re-stream the SAME
  // user chunks and the U workgroups of a slice the SAME item chunks, so per chunk step the XCD pulls S item chunks + U user
  // chunks through the fabric instead of 1 + 32 (all row blocks on one slice: measured 14.1 ms at the TinyLlama shape =
  // the 89 GB of LDS fills at the Infinity-Cache rate, i.e. every user chunk missed L2).
  const int S = 32 / U, n_slices = 8 * S;
  const int w = blockIdx.x, xcd = w & 7, j = w >> 3;
  const int rt = (j >> 5) * U + (j % U), slice = ((j / U) % S) * 8 + xcd;
  if (rt >= R) return;                                        // (whole workgroup: before any barrier)
  const int bps = (n_blocks + n_slices - 1) / n_slices;
  const int b0 = slice * bps, b1 = min(n_blocks, b0 + bps);
  const int wave8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, r = lane & 31, half = lane >> 5;
  const bool loader = wave8 >= 4;                             // waves 4..7: LDS-DMA only
  const int wave = wave8 & 3;                                 // index within the role
  const int wm = wave >> 1, wn = wave & 1;                    // consumer quadrant: 128-item strip, 64-row strip
  const int n_lists = 4 * n_slices;

  int urow[2], cnt[2] = {0, 0};
  float my_tau[2];
  uint32_t my_bits[2];
#pragma unroll
  for (int jt = 0; jt < 2; ++jt) {
    urow[jt] = rt * 128 + wn * 64 + jt * 32 + r;
    const bool live = urow[jt] < n_rows;
    my_bits[jt] = live ? row_bits[urow[jt]] : 0u;
    my_tau[jt] = (live && my_bits[jt] != 0u) ? tau[urow[jt]] : INFINITY;     // rows switched off never pass the threshold test
  }

  if (b0 < b1) {
    const int KC = D / WK, NC = (b1 - b0) * KC;
    // ---- LDS-DMA: 48 one-KiB pieces per stage (32 of item rows, 16 of user rows), 12 per wave + the block's tag words ----
    // Both operands are packed tile images: piece pc of chunk (block, kc) is the 1 KB at [(block * KC + kc) * 32 + pc] KB of
    // the item image (pc < 32) or [(rt * KC + kc) * 16 + pc - 32] KB of the user image; lane l copies bytes 16 l .. 16 l + 15.
    const uint32_t* tag_src = tag_bits ? tag_bits : row_bits;      // no tags: any readable word (never tested)
    const uint32_t lane16 = lane * 16;
    auto piece = [&](auto i_c, int c, int stage) __attribute__((always_inline)) {
      constexpr int i = decltype(i_c)::value;
      const int blk = b0 + c / KC, kc = c - (c / KC) * KC;
      const int pc = wave * W_PW + i;                              // wave-uniform piece index
      unsigned char* dst = smem + stage * W_STAGE + pc * 1024;
      const unsigned char* src = pc < 32 ? items_p + (((int64_t)blk * KC + kc) * 32 + pc) * 1024
                                         : users_p + (((int64_t)rt * KC + kc) * 16 + (pc - 32)) * 1024;
      __builtin_amdgcn_global_load_lds((sg::gptr_t)(src + lane16), (sg::lptr_t)dst, 16, 0, 0);
    };
    // the block's 256 tag words, 64 per wave - with EVERY chunk (same words to the same place), so that each chunk is the
    // same number of DMA instructions and one compile-time vmcnt serves the whole loop
    auto tags_dma = [&](int c) __attribute__((always_inline)) {
      const int blk = b0 + c / KC;
      int64_t n = item_begin + ((int64_t)blk * 256 + wave * 64 + lane) * item_stride;
      n = n < n_items ? n : n_items - 1;
      sg::dma_words(tag_src + (tag_bits ? n : 0), smem + W_TAGS + (blk & 1) * 1024 + wave * 256);
    };
    auto issue = [&](int c, int stage) __attribute__((always_inline)) {
      auto f = [&](auto i_c) __attribute__((always_inline)) { piece(i_c, c, stage); };
      sg::static_for<W_PW>(f);
      tags_dma(c);
    };

    sg::LaneAddr<4> la;
    la.init(lane);
    uint32_t ra_i[4], ra_u[4];                                 // per-lane LDS addresses of the row-fragment reads in stage 0
    const uint32_t base = sg::lds_addr(smem);                  // (the 16-bit offset field of ds_read covers one stage, not
#pragma unroll                                                 //  three: the stage offset is added in a register at the read)
    for (int ks = 0; ks < 4; ++ks) {
      ra_i[ks] = base + wm * 4 * WT::BYTES + (uint32_t)la.a[ks];
      ra_u[ks] = base + W_ITEMS + wn * 2 * WT::BYTES + (uint32_t)la.a[ks];
    }
    // the per-row state was loaded long ago: settle it now, or hipcc parks an `s_waitcnt vmcnt(0)` in front of its first
    // use - inside the emit, where it would drain the DMA ring once per block
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(my_tau[0]), "+v"(my_tau[1]), "+v"(my_bits[0]), "+v"(my_bits[1]));

    if (loader) {
      issue(0, 0);
      issue(min(1, NC - 1), 1);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W_PW + 1) : "memory");          // chunk 0 has landed
      for (int c = 0; c < NC; ++c) {
        sg::ring_barrier();                                    // consumers are done with chunk c - 1's stage; chunk c is theirs
        issue(min(c + 2, NC - 1), (c + 2) % W_NST);            // (redundant tail copies keep the DMA count uniform)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W_PW + 1) : "memory");        // chunk c + 1 has landed before the next barrier
      }
      sg::wait_vmcnt<0>();
    } else {
    f32x16 acc[4][2];
    sg::ring_loop<W_NST>(NC, [&](auto stage_c, int c) __attribute__((always_inline)) {
      constexpr int st = decltype(stage_c)::value;
      sg::ring_barrier();                                      // the loaders have landed chunk c
      const int kc = c % KC;
      if (kc == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int jt = 0; jt < 2; ++jt) acc[i][jt] = sg::zero16();
      }
      // 4 k-steps x (4 item + 2 row fragments -> 8 MFMAs); the reads of k-step ks + 1 are in flight under the MFMAs of ks
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

      if (kc == KC - 1) {                                      // the block's scores are complete: threshold emit
        const int blk = b0 + c / KC;
        // tag words by inline-asm LDS reads (an ordinary read would make hipcc drain the DMA ring first: it cannot prove the
        // read does not alias the in-flight LDS-DMA destination)
        const uint32_t taddr = base + W_TAGS + (blk & 1) * 1024 + wm * 512 + 16 * half;
        const int list = slice * 4 + wm * 2 + half;
        auto tile_i = [&](auto i_c) __attribute__((always_inline)) {
          constexpr int i = decltype(i_c)::value;
          uint32_t tg[16];                                     // tag words of the 16 items this lane sees in item tile i
          auto rdt = [&](auto g_c) __attribute__((always_inline)) {
            constexpr int g = decltype(g_c)::value;
            tg[g] = sg::ds_read_b32_asm<i * 128 + 4 * (g & 3) + 32 * (g >> 2)>(taddr);
          };
          sg::static_for<16>(rdt);
          wait_lgkm_all<0>(tg[0], tg[1], tg[2], tg[3], tg[4], tg[5], tg[6], tg[7], tg[8], tg[9], tg[10], tg[11], tg[12], tg[13],
                           tg[14], tg[15]);
#pragma unroll
          for (int jt = 0; jt < 2; ++jt) {
#pragma unroll
            for (int g = 0; g < 16; ++g) {
              const float sc = acc[i][jt][g];
              if (sc >= my_tau[jt]) {                          // a hit: two stores, no round trip
                const int64_t n = item_begin + ((int64_t)blk * 256 + wm * 128 + i * 32 + sg::crow(g, half)) * item_stride;
                const bool tag_ok = !tag_bits || (tg[g] & my_bits[jt]) != 0u;
                if (n < n_items && n != 0 && tag_ok) {         // n == 0: the pad id (trainer.py:724)
                  const int pos = cnt[jt]++;
                  if (pos < cap_s) {
                    const int64_t o = ((int64_t)urow[jt] * n_lists + list) * cap_s + pos;
                    cand_val[o] = sc;
                    cand_idx[o] = (int32_t)n;
                  }
                }
              }
            }
          }
        };
        sg::static_for<4>(tile_i);
      }
    });
    sg::wait_vmcnt<0>();
    }
  }
  // every (row, slice, item strip, lane half) owns its own list: 4 lists per slice
#pragma unroll
  for (int jt = 0; jt < 2; ++jt)
    if (!loader && urow[jt] < n_rows) cand_cnt[(int64_t)urow[jt] * n_lists + slice * 4 + wm * 2 + half] = cnt[jt];
}

}  // namespace

extern "C" int64_t mhr_pack_tiles_bytes(int64_t n_sel, int dim, int tiles_per_block) {
  const int64_t n_tiles = (n_sel + 31) / 32, n_blocks = (n_tiles + tiles_per_block - 1) / tiles_per_block;
  return n_blocks * (dim / WK) * tiles_per_block * (int64_t)WT::BYTES;
}

extern "C" int mhr_pack_tiles(const void* x, int64_t n_rows, int dim, int64_t row_begin, int64_t row_stride, int64_t n_sel,
                              int tiles_per_block, void* out, void* stream) {
  MHR_REQUIRE(x && out, "pack_tiles: null pointer");
  MHR_REQUIRE(dim >= 64 && dim % 64 == 0 && n_rows > 0 && n_sel > 0 && row_begin >= 0 && row_stride >= 1 &&
              (tiles_per_block == 4 || tiles_per_block == 8), "pack_tiles: bad arguments (dim=%d tiles_per_block=%d)", dim, tiles_per_block);
  MHR_REQUIRE((uintptr_t)x % 16 == 0 && (uintptr_t)out % 16 == 0, "pack_tiles: buffers must be 16-byte aligned");
  const int64_t total = mhr_pack_tiles_bytes(n_sel, dim, tiles_per_block) / 16;
  const int grid = mhr_grid_for(total, 256 * 4, 8192);
  hipLaunchKernelGGL(pack_tiles_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, n_rows, dim, row_begin,
                     row_stride, n_sel, tiles_per_block, (unsigned char*)out);
  MHR_CHECK_LAUNCH("pack_tiles");
  return MHR_OK;
}

extern "C" int mhr_pack_tiles_t(const void* x, int64_t n_k, int64_t ld, int64_t n_sel, int tiles_per_block, void* out, void* stream) {
  MHR_REQUIRE(x && out, "pack_tiles_t: null pointer");
  MHR_REQUIRE(n_k > 0 && ld >= n_sel && n_sel > 0 && (tiles_per_block == 4 || tiles_per_block == 8), "pack_tiles_t: bad arguments");
  MHR_REQUIRE((uintptr_t)out % 16 == 0, "pack_tiles_t: out must be 16-byte aligned");
  const int64_t k_pad = (n_k + WK - 1) / WK * WK;
  const int64_t total = mhr_pack_tiles_bytes(n_sel, (int)k_pad, tiles_per_block) / 16;
  MHR_REQUIRE(k_pad < (1ll << 31), "pack_tiles_t: contraction too long");
  const int grid = mhr_grid_for(total, 256 * 4, 8192);
  hipLaunchKernelGGL(pack_tiles_t_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, n_k, ld, n_sel, tiles_per_block,
                     (unsigned char*)out);
  MHR_CHECK_LAUNCH("pack_tiles_t");
  return MHR_OK;
}

extern "C" int mhr_catalog_wide_slices(int n_rows);
extern "C" int64_t mhr_catalog_score_emit_wide_workspace_bytes(int n_rows, int cap_s) {
  if (n_rows <= 0 || cap_s <= 0) return 0;
  const int64_t lists = (int64_t)n_rows * 4 * mhr_catalog_wide_slices(n_rows);
  return lists * cap_s * 8 + lists * 4;
}

extern "C" int mhr_catalog_wide_slices(int n_rows) {
  const int R = (n_rows + 127) / 128;
  int U = 1;
  while (U < 8 && 2 * U <= R) U *= 2;
  return 8 * (32 / U);
}

extern "C" int mhr_catalog_score_emit_wide(const void* users_packed, int n_rows, const void* items_packed, int64_t n_items, int dim,
                                           int64_t item_begin, int64_t item_stride, const uint32_t* tag_bits,
                                           const uint32_t* row_bits, const float* tau, float* cand_val, int32_t* cand_idx,
                                           int32_t* cand_cnt, int n_slices, int cap_s, void* stream) {
  const void *users = users_packed, *items = items_packed;
  MHR_REQUIRE(users && items && row_bits && tau && cand_val && cand_idx && cand_cnt, "catalog_score_emit_wide: null pointer");
  MHR_REQUIRE(dim >= 64 && dim % 64 == 0 && dim <= 8192, "catalog_score_emit_wide: dim=%d must be a multiple of 64 in [64, 8192]", dim);
  MHR_REQUIRE(n_rows > 0 && n_items > 0 && n_items < (1ll << 31) && item_begin >= 0 && item_begin < n_items && item_stride >= 1,
              "catalog_score_emit_wide: bad item range");
  const int R = (n_rows + 127) / 128;
  int U = 1;
  while (U < 8 && 2 * U <= R) U *= 2;                          // row blocks that share an XCD with 32 / U item slices each
  MHR_REQUIRE(cap_s >= 1 && n_slices == 8 * (32 / U), "catalog_score_emit_wide: n_slices must be mhr_catalog_wide_slices(n_rows) = %d",
              8 * (32 / U));
  MHR_REQUIRE(((uintptr_t)users % 16 == 0) && ((uintptr_t)items % 16 == 0), "catalog_score_emit_wide: operands must be 16-byte aligned");
  const int64_t n_sel = (n_items - item_begin + item_stride - 1) / item_stride;
  const int n_blocks = (int)((n_sel + 255) / 256);
  const size_t lds = (size_t)W_TAGS + 2048;
  auto kern = catalog_emit_wide_kernel;
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(kern, dim3(8 * 32 * ((R + U - 1) / U)), dim3(512), lds, (hipStream_t)stream, (const unsigned char*)users, n_rows,
                     (const unsigned char*)items, n_items, dim, item_begin, item_stride, n_blocks, R, U, tag_bits, row_bits, tau, cand_val,
                     cand_idx, cand_cnt, cap_s);
  MHR_CHECK_LAUNCH("catalog_score_emit_wide");
  return MHR_OK;
}
