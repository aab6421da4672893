// Token-rows projection  C[M, N] = A[M, K] . W[N, K]^T (+ bias)  for the encoder's skinny-K layers: K = the model width
// (<= 256), M = every token of the batch (cfg1: 25 600), N = K or 4K.
//
// Reference path (file:line under code/REC/model/IDNet/): hstu.py:236-239 (`torch.mm(normed_x, self._uvqk)`), hstu.py:281-288
// (`self._o(...)`, an nn.Linear) and their autograd input gradients - under bf16-mixed autocast, i.e. bf16 operands, fp32
// accumulation, bf16 result.
//
// Why not the library: these products are HBM-bound (cfg1 uvqk: 13 MB in, 52 MB out, 13.4 GFLOP - a 52 MB fill alone takes 9.7 us
// on this part) and hipBLASLt runs the uvqk shape at 2.1 TB/s (31 us): a 256 x 256 macro tile with eight k-iterations is all
// prologue and epilogue.  Here W is the STATIONARY operand - 64 output columns per wave kept as MFMA fragments in registers for
// the whole kernel - and the token rows stream through the 3-deep LDS-DMA ring of stream_gemm.h in 32-row tiles, each row
// fetched in full 1-KiB pieces.  Measured (tools/rows_gemm_micro.py, M = 25 600): N = 1024: 22-23 us (51 200 rows: 35 us
// against 54), PMC traffic 17 MB read + 51 MB written = the algorithmic bytes; N = 256: 12-14 us against the library's 10-11 us,
// K = 64: 5.0 against 4.7 us - so the callers (hstu_functional._rows_gemm_pays) route only the wide-output K = 256 product
// here.  Where the 22 us go (s_memtime stamps, tools/_exp/rg_stamp.py): 5 us loading the stationary fragments (every resident
// wave holds 32 KiB of W: 64 MB of L2 reads chip-wide, as much as the whole stream), 15 us of stream - the stores and DMA alone
// take 11 us, the products alone 10 us, and they overlap badly whether 4 or 8 waves share a CU.
//
// Epilogue without LDS: W is the MFMA's A operand, so an accumulator holds C^T - a lane owns ONE token row (lane & 31) and, per
// register quad, four consecutive output columns (8 G + 4 (lane >> 5) + 0..3).  After the packed bf16 convert, one
// v_permlane32_swap per dword hands the upper lane half's quad of column group G to the lower half and the lower half's quad of
// group G + 1 to the upper half: every lane then holds eight consecutive columns of its row and stores them as one 16-byte
// piece (the two lane halves write adjacent pieces: 32 contiguous bytes per row and instruction, the row's 128 bytes over the
// wave's four stores).  A first version transposed the tile through LDS for whole-line stores; stamps put that at 1 000 cycles
// per tile next to 1 100 for the products - with every wave already re-reading each streamed tile the LDS pipe was the limit.
//
// Grid: n_cg column groups (256 columns each) x n_streams row streams; the column groups of one stream sit on one XCD (equal
// blockIdx % 8), so a row tile comes from HBM once and the other groups read it from that L2.
#include <algorithm>

#include "stream_gemm.h"

namespace {

typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));

#ifdef RGX_STAMP   // in-kernel phase timing (tools/variant.py build RGX_STAMP; tools/_exp/rg_stamp.py reads it)
__device__ unsigned long long g_rg_stamps[4096 * 4];
#define RG_STAMP(k)                                                                     \
  {                                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                  \
    unsigned long long t_;                                                              \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");          \
    __builtin_amdgcn_sched_barrier(0);                                                  \
    if (threadIdx.x == 0 && blockIdx.x < 4096) g_rg_stamps[blockIdx.x * 4 + (k)] = t_;  \
  }
#define RG_PHASE(k)                                                                     \
  {                                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                  \
    unsigned long long t_;                                                              \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");          \
    __builtin_amdgcn_sched_barrier(0);                                                  \
    pacc_[k] += t_ - ph_;                                                               \
    ph_ = t_;                                                                           \
  }
#else
#define RG_STAMP(k)
#define RG_PHASE(k)
#endif

// sg::mma_tile_asm with the stationary fragment as the A operand (the accumulators hold C^T) and a callback in every k-step's
// gap: the caller spreads the tile's vector-memory instructions - the next tile's LDS-DMA pieces, the previous tile's stores -
// over the product instead of issuing them in bursts between products.  (All workgroups start together and run in lock step:
// with the bursts, the chip's memory pipes idled during everybody's products and the waves stalled in VMEM issue during
// everybody's stores - measured: stores + DMA alone 11 us, products alone 10 us, both 17 us.)
template <int NKS, int RF, int OFF, typename Gap>
__device__ __forceinline__ void mma_tile_gaps(const sg::RowAddr<NKS>& ra, const bf16x8 (&frag)[RF][NKS], f32x16 (&acc)[RF], Gap gap) {
  constexpr int PA = NKS < 4 ? NKS : 4;
  sg::u32x4 a[PA + 1];
  auto issue_a = [&](auto ks_c) {
    constexpr int ks = decltype(ks_c)::value;
    a[ks % (PA + 1)] = sg::ds_read_b128_asm<OFF + 256 * (ks >> 3)>(ra.a[ks & 7]);
  };
  sg::static_for<PA>(issue_a);
  auto step = [&](auto ks_c) {
    constexpr int ks = decltype(ks_c)::value;
    if constexpr (ks + PA < NKS) issue_a(std::integral_constant<int, ks + PA>{});
    constexpr int a_after = (ks + PA < NKS ? ks + PA : NKS - 1) - ks;
    sg::wait_lgkm1<a_after>(a[ks % (PA + 1)]);
    const bf16x8 av = __builtin_bit_cast(bf16x8, a[ks % (PA + 1)]);
#pragma unroll
    for (int f = 0; f < RF; ++f) acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag[f][ks], av, acc[f], 0, 0, 0);
    gap(ks_c);
    __builtin_amdgcn_sched_barrier(0);
  };
  sg::static_for<NKS>(step);
}

template <int NKS, bool STRIDED, bool HAS_BIAS, bool W_KN>
__global__ __launch_bounds__(256, 2) void rows_gemm_kernel(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ W,
                                                           int64_t ldw, const bf16_t* __restrict__ bias, bf16_t* __restrict__ C,
                                                           int64_t ldc, int M, int N, int n_tiles, int n_cg, int n_streams) {
  using T = sg::Tile<NKS>;
  constexpr int RF = 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* tiles = smem;                              // 3 x T::BYTES (LDS-DMA ring); a fourth tile for the W rounds

  const int w = blockIdx.x, xcd = w & 7, j = w >> 3;        // XCD-aware decode
  const int cg = j % n_cg, stream = (j / n_cg) * 8 + xcd;
  const int t0 = (int)((int64_t)stream * n_tiles / n_streams), t1 = (int)((int64_t)(stream + 1) * n_tiles / n_streams);
  if (t0 >= t1) return;
  RG_STAMP(0)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 31, half = lane >> 5;
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  const int n_wave = cg * 256 + wv * 64;                    // first output column of this wave

  // The stationary operand: this wave's 2 x 32 rows of W as MFMA fragments.  Fetched like the streamed tiles - LDS-DMA in whole
  // 1-KiB pieces into four swizzled tile images (one per wave), then 16-byte LDS reads: fragment loads straight from global
  // memory touch 32 rows x 32 bytes per instruction and cost more than the whole stream (measured at cfg1's uvqk shape: 25 us,
  // of which ~11 us were the stream).  Rows past N repeat row N - 1 (never stored).
  bf16x8 frag[RF][NKS];
  sg::LaneAddr<NKS> la;
  la.init(lane);
  // bias of the lane's columns, in accumulator order: the accumulators start from it
  f32x16 acc0[HAS_BIAS ? RF : 1];
  if constexpr (HAS_BIAS) {
#pragma unroll
    for (int f = 0; f < RF; ++f)
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const int n = n_wave + f * 32 + sg::crow(g, half);
        acc0[f][g] = n < N ? (float)bias[n] : 0.f;
      }
  }
  if constexpr (!W_KN) {
#pragma unroll
    for (int f = 0; f < RF; ++f) {
      if (f) __syncthreads();                                // every wave has read its fragments of the previous round
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n0 = cg * 256 + q * 64 + f * 32;
        sg::Dma<NKS>::issue(smem + q * T::BYTES, [=](int rr) { return W + (int64_t)min(n0 + rr, N - 1) * ldw; }, wv, lane);
      }
      sg::wait_vmcnt<0>();
      __syncthreads();
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) frag[f][ks] = la.read_a(smem + wv * T::BYTES, ks);
    }
  } else {
    // W stored [K, N] (the `_uvqk` parameter, hstu.py:236; the input gradient of an nn.Linear): tiles of 32 k-rows x this
    // group's 256 columns, fragments by the transposing LDS read.  ds_read_b64_tr_b16 returns k = 4 half + 0..3 (rows +0) and
    // 8 + 4 half + 0..3 (rows +8) of a k-step; the streamed operand's row reads hold k = 8 half + 0..7, so the lane halves
    // exchange one quad (v_permlane32_swap: the upper half of `lo` against the lower half of `hi`).  N is a multiple of 256 here.
    using TW = sg::Tile<16>;
    typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
    sg::LaneAddr<16> lw;
    lw.init(lane);
    constexpr int KT = NKS / 2, ROUNDS = (KT + 3) / 4;       // 32-row k tiles, four per round
#pragma unroll
    for (int rd = 0; rd < ROUNDS; ++rd) {
      if (rd) __syncthreads();
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (rd * 4 + q < KT) {
          const int k0 = (rd * 4 + q) * 32;
          sg::Dma<16>::issue(smem + q * TW::BYTES, [=](int rr) { return W + (int64_t)(k0 + rr) * ldw + cg * 256; }, wv, lane);
        }
      }
      sg::wait_vmcnt<0>();
      __syncthreads();
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (rd * 4 + q < KT) {
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
            for (int f = 0; f < RF; ++f) {
              const int dc = 2 * wv + f;                     // this wave's 32-column block of the tile
              const unsigned char* base = smem + q * TW::BYTES + 256 * (dc >> 2) + 16 * s2 * TW::ROW_BYTES;
              const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(base + lw.t[0][dc & 3]));
              const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(base + lw.t[1][dc & 3]));
              const u32x2_t l2 = __builtin_bit_cast(u32x2_t, lo), h2 = __builtin_bit_cast(u32x2_t, hi);
              const u32x2_t s0 = __builtin_amdgcn_permlane32_swap(l2.x, h2.x, false, false);
              const u32x2_t s1 = __builtin_amdgcn_permlane32_swap(l2.y, h2.y, false, false);
              frag[f][(rd * 4 + q) * 2 + s2] = __builtin_bit_cast(bf16x8, sg::u32x4{s0.x, s1.x, s0.y, s1.y});
            }
          }
        }
      }
    }
  }
  __syncthreads();

  RG_STAMP(1)
  using P = sg::DmaPieces<NKS>;
  P dp;
  dp.init(wv, lane);
  auto dma_piece = [&](auto slot_c, auto k_c, int tn) {
    constexpr int slot = decltype(slot_c)::value, k = decltype(k_c)::value;
    if constexpr (STRIDED) {
      dp.template piece_rows<k>(tiles + slot * T::BYTES, [=](int rr) {
        const int m = tn * 32 + rr;
        return A + (int64_t)(m < M ? m : M - 1) * lda;
      }, lane);
    } else {
      dp.template piece<k>(tiles + slot * T::BYTES, reinterpret_cast<const char*>(A) + (int64_t)tn * T::BYTES);
    }
  };
  auto dma_tile = [&](auto slot_c, int tn) {
    auto f = [&](auto k_c) { dma_piece(slot_c, k_c, tn); };
    sg::static_for<P::PW>(f);
  };
  sg::RowAddr<NKS> ra;
  ra.init(la, tiles);

  // the lane's row of the tile and its first column: after the exchange below lane (r, half) holds, per fragment f and column
  // group pair gp, the 8 columns n_wave + 32 f + 16 gp + 8 half + 0..7 of row r
  const int n_lane = n_wave + 8 * half;
  const bool full_cols = n_wave + 64 <= N;                 // wave-uniform: every lane of the wave stores (N is a multiple of 8)

  const int n_loc = t1 - t0, t_last = t1 - 1;
  dma_tile(std::integral_constant<int, 0>{}, t0);
  dma_tile(std::integral_constant<int, 1>{}, min(t0 + 1, t_last));
#ifdef RGX_STAMP
  unsigned long long ph_ = 0, pacc_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  RG_PHASE(7)
#endif
  // One store of a finished tile: piece idx = 2 f + gp of tile `ts` (data in `o`)
  sg::u32x4 o[RF][2];
  auto store_piece = [&](auto idx_c, int ts, bool all) {
    constexpr int idx = decltype(idx_c)::value, f = idx >> 1, gp = idx & 1;
    bf16_t* p = C + (int64_t)(ts * 32 + r) * ldc + n_lane + f * 32 + gp * 16;
    if (all || (ts * 32 + r < M && n_lane + f * 32 + gp * 16 < N)) *reinterpret_cast<sg::u32x4*>(p) = o[f][gp];
  };
  constexpr int NOPS = P::PW + 4;                             // vector-memory instructions per tile: DMA pieces, then stores
  sg::ring_loop<3>(n_loc, [&](auto slot_c, int i) {
    constexpr int cur = decltype(slot_c)::value, nxt = (cur + 2) % 3;
    const int t = t0 + i;
    RG_PHASE(0)
    // Vector-memory operations complete in issue order, so the wait counts those YOUNGER than tile t's pieces.  Issue order:
    // [t0, t0+1] | iteration 0: pieces t0+2 | iteration j >= 1: pieces t0+j+2, stores of tile t0+j-1 | ...  (with partial
    // tiles / column blocks a store may not be issued: the count then leaves the stores out - smaller than the truth, never larger)
    if (i > 2 && full_cols) sg::wait_vmcnt<P::PW + 8>();
    else if (i == 2 && full_cols) sg::wait_vmcnt<P::PW + 4>();
    else sg::wait_vmcnt<P::PW>();
    sg::ring_barrier();
    RG_PHASE(1)
    f32x16 acc[RF];
#pragma unroll
    for (int f = 0; f < RF; ++f) {
      if constexpr (HAS_BIAS) acc[f] = acc0[f];
      else acc[f] = sg::zero16();
    }
    const int t_dma = min(t + 2, t_last);
    const bool st_all = full_cols;                            // (tile t-1 is never the partial last tile)
    mma_tile_gaps<NKS, RF, cur * T::BYTES>(ra, frag, acc, [&](auto ks_c) {
      constexpr int ks = decltype(ks_c)::value;
      auto op = [&](auto j_c) {
        constexpr int jj = decltype(j_c)::value;
        if constexpr (jj * NKS / NOPS == ks) {
          if constexpr (jj < P::PW) dma_piece(std::integral_constant<int, nxt>{}, j_c, t_dma);
          else if (i > 0) store_piece(std::integral_constant<int, jj - P::PW>{}, t - 1, st_all);
        }
      };
      sg::static_for<NOPS>(op);
    });
    RG_PHASE(3)
#pragma unroll
    for (int f = 0; f < RF; ++f) {
#pragma unroll
      for (int gp = 0; gp < 2; ++gp) {                       // column groups G = 2 gp (x) and 2 gp + 1 (y)
        const int gx = 8 * gp, gy = 8 * gp + 4;
        const uint32_t x0 = sg::cvt_pk_bf16(acc[f][gx + 0], acc[f][gx + 1]), x1 = sg::cvt_pk_bf16(acc[f][gx + 2], acc[f][gx + 3]);
        const uint32_t y0 = sg::cvt_pk_bf16(acc[f][gy + 0], acc[f][gy + 1]), y1 = sg::cvt_pk_bf16(acc[f][gy + 2], acc[f][gy + 3]);
        // swap: x's upper lane half <-> y's lower lane half
        const u32x2_t s0 = __builtin_amdgcn_permlane32_swap(x0, y0, false, false);
        const u32x2_t s1 = __builtin_amdgcn_permlane32_swap(x1, y1, false, false);
        o[f][gp] = sg::u32x4{s0.x, s1.x, s0.y, s1.y};
      }
    }
    RG_PHASE(4)
  });
  {                                                           // the last tile's stores
    const bool all = full_cols && t_last * 32 + 32 <= M;
    auto st = [&](auto idx_c) { store_piece(idx_c, t_last, all); };
    sg::static_for<4>(st);
  }
  RG_STAMP(2)
#ifdef RGX_STAMP
  if (threadIdx.x == 0 && blockIdx.x == 37)
    for (int k = 0; k < 7; ++k) g_rg_stamps[4096 * 4 - 8 + k] += pacc_[k];
#endif
  sg::wait_vmcnt<0>();
  RG_STAMP(3)
}

}  // namespace

#ifdef RGX_STAMP
extern "C" int mhr_debug_read_rg_stamps(unsigned long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_rg_stamps), (size_t)n * sizeof(unsigned long long));
}
#endif

extern "C" int mhr_rows_gemm_supported(int M, int N, int K, int w_is_kn) {
  return (K == 64 || K == 128 || K == 256) && N > 0 && N % (w_is_kn ? 256 : 8) == 0 && M > 0;
}

extern "C" int mhr_rows_gemm(const void* a, int64_t lda, const void* w, int64_t ldw, int w_is_kn, const void* bias, void* c,
                             int64_t ldc, int M, int N, int K, void* stream) {
  MHR_REQUIRE(a && w && c, "rows_gemm: null pointer");
  MHR_REQUIRE(mhr_rows_gemm_supported(M, N, K, w_is_kn),
              "rows_gemm: M=%d N=%d K=%d unsupported (K in {64, 128, 256}; N a multiple of 8, of 256 with W stored [K, N])", M, N, K);
  MHR_REQUIRE(lda >= K && ldw >= (w_is_kn ? N : K) && ldc >= N && lda % 8 == 0 && ldw % 8 == 0 && ldc % 8 == 0,
              "rows_gemm: leading dimensions must cover the rows and be multiples of 8 elements");
  MHR_REQUIRE(((uintptr_t)a | (uintptr_t)w | (uintptr_t)c) % 16 == 0, "rows_gemm: operands must be 16-byte aligned");
  const int n_tiles = (M + 31) / 32;
  const int n_cg = (N + 255) / 256;
  // two workgroups per CU; every stream needs a few tiles to amortise the stationary operand's load
  int n_streams = (512 / n_cg) & ~7;
  while (n_streams > 8 && n_tiles / n_streams < 3) n_streams -= 8;
  if (n_streams < 8) n_streams = 8;
  const bool strided = lda != K || M % 32 != 0;
  const int nks = K / 16;
  size_t lds = 4 * (size_t)(32 * K * 2);                   // ring (3 tiles) + 1: the four tile slots of a W round
  if (w_is_kn) lds = std::max(lds, (size_t)std::min(K / 32, 4) * 16384);
  hipStream_t s = (hipStream_t)stream;
#define L____(NKS, ST, HB, KN)                                                                                            \
  hipLaunchKernelGGL((rows_gemm_kernel<NKS, ST, HB, KN>), dim3(n_cg * n_streams), dim3(256), lds, s, (const bf16_t*)a, lda, \
                     (const bf16_t*)w, ldw, (const bf16_t*)bias, (bf16_t*)c, ldc, M, N, n_tiles, n_cg, n_streams)
#define L___(NKS, ST, HB)                    \
  if (w_is_kn) { L____(NKS, ST, HB, true); } \
  else { L____(NKS, ST, HB, false); }
#define L__(NKS, ST)                \
  if (bias) { L___(NKS, ST, true) } \
  else { L___(NKS, ST, false) }
#define L_(NKS)                     \
  if (strided) { L__(NKS, true) }   \
  else { L__(NKS, false) }
  if (nks == 16) { L_(16) }
  else if (nks == 8) { L_(8) }
  else { L_(4) }
#undef L_
#undef L__
#undef L___
#undef L____
  MHR_CHECK_LAUNCH("rows_gemm");
  return MHR_OK;
}
