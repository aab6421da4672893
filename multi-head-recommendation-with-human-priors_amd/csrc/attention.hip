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

#ifdef MHR_STAMP   // in-kernel phase timing, only in the builds tools/stamp_nce.py makes (-DASTAMP_FWD: the forward's stamps)
__device__ unsigned long long g_attn_stamps[16];
#define ASTAMP_BODY(k)                                                                  \
  {                                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                  \
    unsigned long long t_;                                                              \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");          \
    __builtin_amdgcn_sched_barrier(0);                                                  \
    if (blockIdx.x == 37 && threadIdx.x == 0 && (k) < 16) g_attn_stamps[k] = t_;        \
  }
#ifdef ASTAMP_FWD
#define ASTAMP(k)
#define FSTAMP(k) ASTAMP_BODY(k)
#else
#define ASTAMP(k) ASTAMP_BODY(k)
#define FSTAMP(k)
#endif
#else
#define ASTAMP(k)
#define FSTAMP(k)
#endif


// ------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------
// Workgroups are dealt to the 8 XCDs round-robin (workgroup i runs on XCD i % 8) and every XCD has its own L2.  A head's
// q / k / v / dO slice of a row is head_dim * 2 bytes - 64 B at head_dim 32, HALF a 128-byte line whose other half belongs
// to the neighbouring head - so with the natural order (b * n_heads + head, n_heads = 8) the heads of one sequence land on
// eight different XCDs and every line is fetched from HBM once per head that touches it.  This decode keeps all heads of a
// sequence on one XCD (b = 8 * (slot / n_heads) + xcd): the neighbours' halves are L2 hits.
// seq_order (optional): the sequence a slot runs - sequences sorted by their live length, longest first, so that where a launch
// needs more than one round of workgroups (the backward: two per CU) the dispatcher hands out the heavy ones first.
__device__ __forceinline__ void decode_seq_head(int n_heads, int& b, int& head, const int32_t* __restrict__ seq_order = nullptr) {
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
  if (seq_order) b = __builtin_amdgcn_readfirstlane(seq_order[b]);       // (keeps b - and every address built on it - scalar)
}

template <int NKS, int ND>
__global__ __launch_bounds__(256) void hstu_attn_fwd_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                            const bf16_t* __restrict__ v, int64_t stride,
                                                            const uint8_t* __restrict__ key_valid, bf16_t* __restrict__ out,
                                                            int64_t out_stride, bf16_t* act_q, bf16_t* act_k, bf16_t* act_v,
                                                            int64_t act_stride, int L_max, int n_heads, int hd, int apply_silu,
                                                            float inv_n, const int32_t* __restrict__ first_block,
                                                            const int32_t* __restrict__ seq_order,
                                                            const int32_t* __restrict__ cu_rows, int n_rows_total) {
  using T = sg::Tile<NKS>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int b, head;
  decode_seq_head(n_heads, b, head, seq_order);
  // cu_rows (optional): PACKED sequences - sequence b owns rows cu_rows[b] .. cu_rows[b + 1] - 1 (no padding rows at all);
  // otherwise B windows of L_max rows.  The LDS budget and 1 / n stay those of L_max.
  const int64_t row0 = cu_rows ? (int64_t)__builtin_amdgcn_readfirstlane(cu_rows[b]) : (int64_t)b * L_max;
  const int L = cu_rows ? __builtin_amdgcn_readfirstlane(cu_rows[b + 1]) - (int)row0 : L_max;
  const int Lp = (L + 31) & ~31, nb = Lp >> 5;
  unsigned char* Kt = smem;                       // nb tiles: activated K
  unsigned char* Vt = smem + nb * T::BYTES;       // nb tiles: activated V
  uint32_t* vmask = reinterpret_cast<uint32_t*>(Vt + nb * T::BYTES);

  const bf16_t* qp = q + row0 * stride + head * hd;
  const bf16_t* kp = k + row0 * stride + head * hd;
  const bf16_t* vp = v + row0 * stride + head * hd;
  bf16_t* aq = act_q ? act_q + row0 * act_stride + head * hd : nullptr;
  bf16_t* ak = act_k ? act_k + row0 * act_stride + head * hd : nullptr;
  bf16_t* av = act_v ? act_v + row0 * act_stride + head * hd : nullptr;
  const bool do_silu = apply_silu != 0;

  FSTAMP(0)
  // leading all-padding blocks: not staged, no tile pairs, zeros written (with saved activations everything is staged: the caller
  // wants silu(q | k | v) of every row)
  const int kb0 = (aq || !first_block) ? 0 : min(nb, __builtin_amdgcn_readfirstlane(first_block[b]));
  stage_tiles<NKS>(Kt, kp, stride, L, Lp, hd, do_silu, ak, act_stride, kb0 * 32);
  stage_tiles<NKS>(Vt, vp, stride, L, Lp, hd, do_silu, av, act_stride, kb0 * 32);
  build_valid_mask(vmask, key_valid + row0, L, nb);
  zero_head_rows(out + row0 * out_stride + head * hd, out_stride, min(L, kb0 * 32), hd);
  // packed batch: the rows behind the last sequence (up to the buffer's capacity) belong to nobody - the workgroups of the last
  // sequence write their zeros, so that whatever multiplies them later meets finite values
  if (cu_rows && b == (int)(gridDim.x / n_heads) - 1)
    zero_head_rows(out + (row0 + L) * out_stride + head * hd, out_stride, n_rows_total - (int)(row0 + L), hd);
  __syncthreads();
  FSTAMP(1)

  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 31, half = lane >> 5;
  sg::LaneAddr<NKS> la;
  la.init(lane);
  for (int it = 0; it * 4 < nb; ++it) {
    // snake from the HEAVY end (query block qb costs qb + 1 tile pairs): blocks nb-1 .. nb-4 to waves 0..3, the next four to
    // waves 3..0, ...  Dealt from the light end, nb = 7 (L = 200) came out as 1 / 9 / 9 / 9 pairs per wave instead of 7 each.
    const int qb = nb - 1 - ((it & 1) ? it * 4 + (3 - wave) : it * 4 + wave);
    if (qb < kb0) continue;                          // (below the sequence, or an all-padding block: zeros already written)
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
    FSTAMP(2 + 3 * it)

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
    FSTAMP(3 + 3 * it)
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
    FSTAMP(4 + 3 * it)
  }
}

// ------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------
// ALL4: the four operand images (Q, dO, K, V) are resident together (short sequences: 4 x 14 KB at cfg1, two workgroups
// per CU): one staging phase with every load in flight, one barrier, and every fragment of both passes is an LDS read.
// Otherwise (long sequences) the K / V images replace the Q / dO images between the passes and the per-block fragments
// come from global memory.  Measured at cfg1 on the two-image form: of 68 k cycles of a workgroup's life only 20 k were
// tile-pair work - the rest was memory latency in front of dependent work (staging twice, 5-7 k cycles per fragment fetch,
// 4.5 k per gradient-tile epilogue).
template <int NKS, int ND, bool ALL4>
__global__ __launch_bounds__(256, ALL4 ? 2 : (NKS <= 2 ? ATTN_BWD_WG : 1)) void hstu_attn_bwd_kernel(
    const bf16_t* __restrict__ q_pre, const bf16_t* __restrict__ k_pre, const bf16_t* __restrict__ v_pre, int64_t stride,
    const bf16_t* __restrict__ act_q, const bf16_t* __restrict__ act_k, const bf16_t* __restrict__ act_v, int64_t act_stride,
    const uint8_t* __restrict__ key_valid, const bf16_t* __restrict__ d_out, int64_t do_stride, bf16_t* __restrict__ dq,
    bf16_t* __restrict__ dk, bf16_t* __restrict__ dv, int64_t d_stride, int L_max, int n_heads, int hd, int apply_silu,
    float inv_n, const int32_t* __restrict__ first_block, const int32_t* __restrict__ seq_order,
    const int32_t* __restrict__ cu_rows, int n_rows_total) {
  using T = sg::Tile<NKS>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int b, head;
  decode_seq_head(n_heads, b, head, seq_order);
  const int64_t row0 = cu_rows ? (int64_t)__builtin_amdgcn_readfirstlane(cu_rows[b]) : (int64_t)b * L_max;   // (packed sequences: see the forward)
  const int L = cu_rows ? __builtin_amdgcn_readfirstlane(cu_rows[b + 1]) - (int)row0 : L_max;
  const int Lp = (L + 31) & ~31, nb = Lp >> 5;
  const int img = nb * T::BYTES;
  unsigned char* Tq = smem;                                  // Q tiles
  unsigned char* Tdo = smem + img;                           // dO tiles
  unsigned char* Tk = ALL4 ? smem + 2 * img : Tq;            // K tiles (two-image form: over the Q tiles, in pass B)
  unsigned char* Tv = ALL4 ? smem + 3 * img : Tdo;           // V tiles
  unsigned char* tail = smem + (ALL4 ? 4 : 2) * img;
  uint32_t* vmask = reinterpret_cast<uint32_t*>(tail);
  // gradient tiles leave through a wave-private scratch (store_grad_tile), behind the validity words (16-byte aligned)
  float* gscratch = reinterpret_cast<float*>(tail + ((nb * 4 + 15) & ~15)) + (threadIdx.x >> 6) * (32 * GS);

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
  const bf16_t* qpre_h = q_pre + (chain ? row0 * stride + hoff : 0);
  const bf16_t* kpre_h = k_pre + (chain ? row0 * stride + hoff : 0);
  const bf16_t* vpre_h = v_pre + (chain ? row0 * stride + hoff : 0);
  bf16_t* dq_h = dq + row0 * d_stride + hoff;
  bf16_t* dk_h = dk + row0 * d_stride + hoff;
  bf16_t* dv_h = dv + row0 * d_stride + hoff;

  ASTAMP(0)
  // leading all-padding blocks (attn_tiles.h: leading_dead_blocks): their dq / dk / dv are zero - not staged, no tile pairs, no
  // SiLU' epilogue, the zeros written directly
  const int kb0 = first_block ? min(nb, __builtin_amdgcn_readfirstlane(first_block[b])) : 0;
  if (ALL4) stage_tiles4<NKS>(Tq, aq, a_stride, redo, Tdo, dop, do_stride, false, Tk, ak, a_stride, redo, Tv, av, a_stride, redo, L, Lp, hd, kb0 * 32);
  else stage_tiles2<NKS>(Tq, aq, a_stride, redo, Tdo, dop, do_stride, false, L, Lp, hd, kb0 * 32);
  build_valid_mask(vmask, key_valid + row0, L, nb);
  {
    const int n_dead = min(L, kb0 * 32);
    zero_head_rows(dq_h, d_stride, n_dead, hd);
    zero_head_rows(dk_h, d_stride, n_dead, hd);
    zero_head_rows(dv_h, d_stride, n_dead, hd);
    if (cu_rows && b == (int)(gridDim.x / n_heads) - 1) {      // packed batch: the rows behind the last sequence (see the forward)
      const int n_tail = n_rows_total - (int)(row0 + L);
      zero_head_rows(dq_h + (int64_t)L * d_stride, d_stride, n_tail, hd);
      zero_head_rows(dk_h + (int64_t)L * d_stride, d_stride, n_tail, hd);
      zero_head_rows(dv_h + (int64_t)L * d_stride, d_stride, n_tail, hd);
    }
  }
  __syncthreads();
  ASTAMP(1)

  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 31, half = lane >> 5;
  sg::LaneAddr<NKS> la;
  la.init(lane);

  // ---- pass A: dK, dV for key block kb (keys on the lanes) -----------------------------------------
  for (int it = 0; it * 4 < nb - kb0; ++it) {
    const int kb = kb0 + ((it & 1) ? it * 4 + wave : it * 4 + (3 - wave));   // early (live) key blocks are the heavy ones
    if (kb >= nb) continue;
    const int key = kb * 32 + r;
    const bool kvalid = (vmask[kb] >> r) & 1u;
    GradPre kpre[ND], vpre[ND];                                      // for the SiLU' chain of the epilogue: in flight under the loop
#pragma unroll
    for (int dc = 0; dc < ND; ++dc) {
      kpre[dc] = prefetch_pre(kpre_h, stride, kb * 32, L, dc * 32, hd, chain, lane);
      vpre[dc] = prefetch_pre(vpre_h, stride, kb * 32, L, dc * 32, hd, chain, lane);
    }
    bf16x8 kf[NKS], vf[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      if (ALL4) {
        kf[ks] = la.read_a(Tk + kb * T::BYTES, ks);
        vf[ks] = la.read_a(Tv + kb * T::BYTES, ks);
      } else {
        kf[ks] = load_frag(ak, a_stride, key, L, ks * 16 + 8 * half, hd);
        vf[ks] = load_frag(av, a_stride, key, L, ks * 16 + 8 * half, hd);
        if (redo) {
          kf[ks] = silu8(kf[ks]);
          vf[ks] = silu8(vf[ks]);
        }
      }
    }
    f32x16 dvacc[ND], dkacc[ND];
#pragma unroll
    for (int dc = 0; dc < ND; ++dc) {
      dvacc[dc] = zero16();
      dkacc[dc] = zero16();
    }
    ASTAMP(2 + 3 * it)
    const uint32_t vm_kb = vmask[kb];
    for (int qb = kb; qb < nb && vm_kb != 0; ++qb) {     // a block of padding keys gets zero gradients
      const unsigned char* qt = Tq + qb * T::BYTES;
      const unsigned char* dot = Tdo + qb * T::BYTES;
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
    ASTAMP(3 + 3 * it)
    // results: rows (regs) = keys, cols (lanes) = feature -> 16-byte rows through the wave's scratch, SiLU' chain applied
#pragma unroll
    for (int dc = 0; dc < ND; ++dc) {
      store_grad_tile(gscratch, dvacc[dc], inv_n, dv_h, d_stride, vpre[dc], kb * 32, L, dc * 32, hd, chain, lane);
      store_grad_tile(gscratch, dkacc[dc], inv_n, dk_h, d_stride, kpre[dc], kb * 32, L, dc * 32, hd, chain, lane);
    }
    ASTAMP(4 + 3 * it)
  }

  // ---- pass B: dQ for query block qb (queries on the lanes) ----------------------------------------
  if (!ALL4) {
    __syncthreads();                      // everyone is done reading the Q / dO tiles
    ASTAMP(8)
    stage_tiles2<NKS>(Tk, ak, a_stride, redo, Tv, av, a_stride, redo, L, Lp, hd, kb0 * 32);
    __syncthreads();
  }
  ASTAMP(9)
  for (int it = 0; it * 4 < nb; ++it) {
    const int qb = nb - 1 - ((it & 1) ? it * 4 + (3 - wave) : it * 4 + wave);     // snake from the heavy end (see the forward)
    if (qb < kb0) continue;                          // (below the sequence, or an all-padding block: zeros already written)
    const int qcol = qb * 32 + r;
    GradPre qpre[ND];
#pragma unroll
    for (int dc = 0; dc < ND; ++dc) qpre[dc] = prefetch_pre(qpre_h, stride, qb * 32, L, dc * 32, hd, chain, lane);
    bf16x8 qf[NKS], dof[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      if (ALL4) {
        qf[ks] = la.read_a(Tq + qb * T::BYTES, ks);
        dof[ks] = la.read_a(Tdo + qb * T::BYTES, ks);
      } else {
        qf[ks] = load_frag(aq, a_stride, qcol, L, ks * 16 + 8 * half, hd);
        if (redo) qf[ks] = silu8(qf[ks]);
        dof[ks] = load_frag(dop, do_stride, qcol, L, ks * 16 + 8 * half, hd);
      }
    }
    f32x16 dqacc[ND];
#pragma unroll
    for (int dc = 0; dc < ND; ++dc) dqacc[dc] = zero16();
    ASTAMP(10 + 3 * it)
    for (int kb = 0; kb <= qb; ++kb) {
      const uint32_t vm = vmask[kb];
      if (vm == 0) continue;
      const unsigned char* kt = Tk + kb * T::BYTES;
      const unsigned char* vt = Tv + kb * T::BYTES;
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
    ASTAMP(11 + 3 * it)
#pragma unroll
    for (int dc = 0; dc < ND; ++dc)
      store_grad_tile(gscratch, dqacc[dc], inv_n, dq_h, d_stride, qpre[dc], qb * 32, L, dc * 32, hd, chain, lane);
    ASTAMP(12 + 3 * it)
  }
}

// the streamed form (attention_stream.hip): any L; chosen when the resident images do not fit the LDS, or when they fill more
// than STREAM_ABOVE bytes of it: a resident workgroup that owns a CU alone runs its staging and its tail with nothing to
// overlap them.  Measured (tools/attn_forms.py, us, resident -> streamed): cfg2 (64 x 512, 16 heads x 64) forward 176 -> 148,
// backward 456 -> 397; 64 x 300, 8 x 64: 59 -> 59 and 145 -> 134; cfg1 (128 x 200, 8 x 32: 28 KB / 67 KB resident) 27 -> 43
// and 62 -> 96, so short sequences stay resident.
// MHR_ATTN_STREAM=0/1 in the environment forces the choice where both forms can run (experiments, tests).
static bool attn_use_stream(size_t resident_lds, size_t stream_above) {
  const char* e = getenv("MHR_ATTN_STREAM");
  const int forced = e ? atoi(e) : -1;
  if (resident_lds > 160 * 1024) return true;
  if (forced >= 0) return forced != 0;
  return resident_lds > stream_above;
}
constexpr size_t ATTN_STREAM_ABOVE_FWD = 64 * 1024;

}  // namespace

int mhr_attn_stream_fwd(const void* q, const void* k, const void* v, int64_t row_stride, const uint8_t* key_valid, void* out,
                        void* act_q, void* act_k, void* act_v, int64_t act_stride, int B, int L, int n_heads, int head_dim,
                        int apply_silu, hipStream_t s);
int mhr_attn_stream_bwd(const void* q_pre, const void* k_pre, const void* v_pre, int64_t row_stride, const void* act_q,
                        const void* act_k, const void* act_v, int64_t act_stride, const uint8_t* key_valid, const void* d_out,
                        void* dq, void* dk, void* dv, int64_t d_stride, int B, int L, int n_heads, int head_dim, int apply_silu,
                        hipStream_t s);

namespace {

// first_block[b] = index of the 32-row block holding the first valid key of sequence b (nb when none); counts[.] feed the
// counting sort below.  One wave per sequence, ballots over the mask bytes.
__global__ __launch_bounds__(64) void seq_first_block_kernel(const uint8_t* __restrict__ key_valid, int L, int nb,
                                                             int32_t* __restrict__ first_block, int32_t* __restrict__ first_row) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const uint8_t* kv = key_valid + (int64_t)b * L;
  int first = L;                                        // index of the first valid key (L: none)
  for (int j0 = 0; j0 < L && first == L; j0 += 64) {
    const int j = j0 + lane;
    const unsigned long long m = __ballot(j < L && kv[j] != 0);
    if (m) first = j0 + __builtin_ctzll(m);
  }
  if (lane == 0) {
    first_block[b] = first < L ? first >> 5 : nb;
    if (first_row) first_row[b] = first;
  }
}

// seq_order: sequences by first_block ascending (most live blocks first), ties in index order - a counting sort in one
// workgroup (B and nb are small: a batch of sequences, L / 32 buckets).
__global__ __launch_bounds__(256) void seq_order_kernel(const int32_t* __restrict__ first_block, int B, int nb,
                                                        int32_t* __restrict__ seq_order) {
  extern __shared__ int32_t cnt[];                      // nb + 1 buckets
  for (int i = threadIdx.x; i <= nb; i += 256) cnt[i] = 0;
  __syncthreads();
  for (int b = threadIdx.x; b < B; b += 256) atomicAdd(&cnt[first_block[b]], 1);
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int i = 0; i <= nb; ++i) {
      const int c = cnt[i];
      cnt[i] = run;
      run += c;
    }
  }
  __syncthreads();
  // stable within a bucket: thread t places the sequences of bucket t in index order
  for (int bucket = threadIdx.x; bucket <= nb; bucket += 256) {
    int at = cnt[bucket];
    for (int b = 0; b < B; ++b)
      if (first_block[b] == bucket) seq_order[at++] = b;
  }
}

}  // namespace

extern "C" int mhr_attn_seq_layout(const uint8_t* key_valid, int B, int L, int32_t* first_block, int32_t* seq_order,
                                   int32_t* first_row, void* stream) {
  MHR_REQUIRE(key_valid && first_block, "attn_seq_layout: null pointer");
  MHR_REQUIRE(B > 0 && L > 0 && L <= 131072, "attn_seq_layout: bad sizes");
  const int nb = (L + 31) / 32;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(seq_first_block_kernel, dim3(B), dim3(64), 0, s, key_valid, L, nb, first_block, first_row);
  if (seq_order)
    hipLaunchKernelGGL(seq_order_kernel, dim3(1), dim3(256), (size_t)(nb + 1) * sizeof(int32_t), s, first_block, B, nb, seq_order);
  MHR_CHECK_LAUNCH("attn_seq_layout");
  return MHR_OK;
}

extern "C" int mhr_hstu_attn_fwd_seq(const void* q, const void* k, const void* v, int64_t row_stride, const uint8_t* key_valid,
                                     void* out, void* act_q, void* act_k, void* act_v, int64_t act_stride, int B, int L,
                                     int n_heads, int head_dim, int apply_silu, const int32_t* first_block,
                                     const int32_t* seq_order, const int32_t* cu_rows, int n_rows_total, void* stream) {
  MHR_REQUIRE(q && k && v && key_valid && out, "hstu_attn_fwd: null pointer");
  AttnShape sh;
  MHR_REQUIRE(attn_shape(head_dim, sh), "hstu_attn_fwd: head_dim=%d unsupported (multiple of 8, <= 128)", head_dim);
  MHR_REQUIRE(B > 0 && L > 0 && n_heads > 0, "hstu_attn_fwd: bad sizes");
  MHR_REQUIRE(row_stride % 8 == 0 && (!act_q || act_stride % 8 == 0), "hstu_attn_fwd: strides must be multiples of 8");
  MHR_REQUIRE((act_q != nullptr) == (act_k != nullptr) && (act_k != nullptr) == (act_v != nullptr),
              "hstu_attn_fwd: act_q/act_k/act_v must be all set or all null");
  const int Lp = (L + 31) & ~31, nb = Lp / 32;
  size_t lds = (size_t)2 * nb * (32 * sh.nks * 32) + (size_t)nb * 4 + 16;     // K and V tile images + validity words
  MHR_REQUIRE(nb <= 4096, "hstu_attn_fwd: L=%d too long (<= 131072)", L);
  const float inv_n = 1.0f / (float)L;
  const int64_t out_stride = (int64_t)n_heads * head_dim;
  hipStream_t s = (hipStream_t)stream;
  if (attn_use_stream(lds, ATTN_STREAM_ABOVE_FWD)) {
    MHR_REQUIRE(!cu_rows, "hstu_attn_fwd: packed sequences (cu_rows) need the resident form (L=%d too long)", L);
    mhr_attn_stream_fwd(q, k, v, row_stride, key_valid, out, act_q, act_k, act_v, act_stride, B, L, n_heads, head_dim, apply_silu, s);
    MHR_CHECK_LAUNCH("hstu_attn_fwd(streamed)");
    return MHR_OK;
  }
#define L_(NKS, ND)                                                                                                    \
  {                                                                                                                    \
    auto kern = hstu_attn_fwd_kernel<NKS, ND>;                                                                         \
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL(kern, dim3(B * n_heads), dim3(256), lds, s, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, \
                       row_stride, key_valid, (bf16_t*)out, out_stride, (bf16_t*)act_q, (bf16_t*)act_k, (bf16_t*)act_v, \
                       act_stride, L, n_heads, head_dim, apply_silu, inv_n, first_block, seq_order, cu_rows, n_rows_total); \
  }
  ATTN_DISPATCH(sh, L_);
#undef L_
  MHR_CHECK_LAUNCH("hstu_attn_fwd");
  return MHR_OK;
}

extern "C" int mhr_hstu_attn_fwd(const void* q, const void* k, const void* v, int64_t row_stride, const uint8_t* key_valid,
                                 void* out, void* act_q, void* act_k, void* act_v, int64_t act_stride, int B, int L,
                                 int n_heads, int head_dim, int apply_silu, void* stream) {
  return mhr_hstu_attn_fwd_seq(q, k, v, row_stride, key_valid, out, act_q, act_k, act_v, act_stride, B, L, n_heads, head_dim,
                               apply_silu, nullptr, nullptr, nullptr, 0, stream);
}

extern "C" int mhr_hstu_attn_bwd_seq(const void* q_pre, const void* k_pre, const void* v_pre, int64_t row_stride,
                                     const void* act_q, const void* act_k, const void* act_v, int64_t act_stride,
                                     const uint8_t* key_valid, const void* d_out, void* dq, void* dk, void* dv,
                                     int64_t d_stride, int B, int L, int n_heads, int head_dim, int apply_silu,
                                     const int32_t* first_block, const int32_t* seq_order, const int32_t* cu_rows, int n_rows_total,
                                     void* stream) {
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
  // operand tile images + validity words + one 32 x 36 float gradient scratch per wave.  Short sequences keep all four
  // operands resident (ALL4, two workgroups per CU); long ones two at a time (K / V replace Q / dO between the passes)
  const size_t img = (size_t)nb * (32 * sh.nks * 32), extra = (((size_t)nb * 4 + 15) & ~(size_t)15) + 4 * 32 * 36 * sizeof(float);
  const bool all4 = 4 * img + extra <= 80 * 1024 && sh.nks <= 4;     // (head_dim 128: the resident form spills)
  const size_t lds = (all4 ? 4 : 2) * img + extra;
  MHR_REQUIRE(nb <= 4096, "hstu_attn_bwd: L=%d too long (<= 131072)", L);
  const float inv_n = 1.0f / (float)L;
  const int64_t do_stride = (int64_t)n_heads * head_dim;
  hipStream_t s = (hipStream_t)stream;
  if (attn_use_stream(lds, all4 ? 160 * 1024 : 0)) {       // streamed whenever the four-image form does not apply
    MHR_REQUIRE(!cu_rows, "hstu_attn_bwd: packed sequences (cu_rows) need the resident form (L=%d too long)", L);
    mhr_attn_stream_bwd(q_pre, k_pre, v_pre, row_stride, act_q, act_k, act_v, act_stride, key_valid, d_out, dq, dk, dv, d_stride, B, L,
                        n_heads, head_dim, apply_silu, s);
    MHR_CHECK_LAUNCH("hstu_attn_bwd(streamed)");
    return MHR_OK;
  }
#define L__(NKS, ND, A4)                                                                                               \
  {                                                                                                                    \
    auto kern = hstu_attn_bwd_kernel<NKS, ND, A4>;                                                                     \
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL(kern, dim3(B * n_heads), dim3(256), lds, s, (const bf16_t*)q_pre, (const bf16_t*)k_pre,          \
                       (const bf16_t*)v_pre, row_stride, (const bf16_t*)act_q, (const bf16_t*)act_k, (const bf16_t*)act_v, \
                       act_stride, key_valid, (const bf16_t*)d_out, do_stride, (bf16_t*)dq, (bf16_t*)dk, (bf16_t*)dv,   \
                       d_stride, L, n_heads, head_dim, apply_silu, inv_n, first_block, seq_order, cu_rows, n_rows_total); \
  }
#define L_(NKS, ND)                \
  if (all4) L__(NKS, ND, true)     \
  else L__(NKS, ND, false)
  ATTN_DISPATCH(sh, L_);
#undef L_
#undef L__
  MHR_CHECK_LAUNCH("hstu_attn_bwd");
  return MHR_OK;
}

extern "C" int mhr_hstu_attn_bwd(const void* q_pre, const void* k_pre, const void* v_pre, int64_t row_stride,
                                 const void* act_q, const void* act_k, const void* act_v, int64_t act_stride,
                                 const uint8_t* key_valid, const void* d_out, void* dq, void* dk, void* dv, int64_t d_stride,
                                 int B, int L, int n_heads, int head_dim, int apply_silu, void* stream) {
  return mhr_hstu_attn_bwd_seq(q_pre, k_pre, v_pre, row_stride, act_q, act_k, act_v, act_stride, key_valid, d_out, dq, dk, dv,
                               d_stride, B, L, n_heads, head_dim, apply_silu, nullptr, nullptr, nullptr, 0, stream);
}

#ifdef MHR_STAMP
extern "C" int mhr_debug_read_attn_stamps(unsigned long long* host16) {
  return (int)hipMemcpyFromSymbol(host16, HIP_SYMBOL(g_attn_stamps), 16 * sizeof(unsigned long long));
}
#endif
