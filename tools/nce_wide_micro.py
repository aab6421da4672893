"""Micro-driver: the wide sampled-softmax contraction kernels (csrc/nce_wide.hip) at BASELINE configs 2 - 4 widths: per launch
time and TFLOP/s of fix_bits / fwd / grad_tile (each one T x n_neg x D product), next to the library-GEMM form of the same
forward (torch.mm into fp32 chunks + the dense epilogue)."""
import math, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mhr_amd  # noqa: F401
from mhr_amd import lib, ops, wide

def time_call(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        fn(); ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(n))
    return ts[n // 2]

st = torch.cuda.current_stream().cuda_stream
for D, T, n_neg in ((1024, 4608, 4096), (1024, 32768, 4096), (2048, 12800, 8192), (4096, 2048, 8192)):
    g = torch.Generator(device="cuda").manual_seed(D + T)
    qn = torch.nn.functional.normalize(torch.randn(T, D, device="cuda", generator=g), dim=-1).bfloat16()
    pn = torch.nn.functional.normalize(torch.randn(T, D, device="cuda", generator=g), dim=-1).bfloat16()
    ng = torch.nn.functional.normalize(torch.randn(n_neg, D, device="cuda", generator=g), dim=-1).bfloat16()
    t_pad, n_tiles = -(-T // 128) * 128, -(-n_neg // 256) * 8
    negs_p, q_p, p_p = ops.pack_tiles(ng, n_sel=n_neg, tiles_per_block=8), ops.pack_tiles(qn, tiles_per_block=4), ops.pack_tiles(pn, tiles_per_block=4)
    bits = torch.empty(n_tiles * 2, t_pad, dtype=torch.int16, device="cuda")
    n_lists = 4 * lib.load().mhr_catalog_wide_slices(T)
    part = torch.empty(3, n_lists, t_pad, device="cuda")
    sp = (qn.float() * pn.float()).sum(-1).contiguous(); sp = torch.nn.functional.pad(sp, (0, t_pad - T))
    scale = torch.tensor([20.0], device="cuda"); nlive = torch.tensor([T], dtype=torch.int32, device="cuda")
    lse, loss, w = torch.empty(t_pad, device="cuda"), torch.empty(t_pad, device="cuda"), torch.rand(t_pad, device="cuda")
    gm = torch.empty(T, n_neg, dtype=torch.bfloat16, device="cuda")
    f_bits = lambda: lib.call("mhr_nce_wide_fix_bits", p_p.data_ptr(), T, negs_p.data_ptr(), n_neg, D, 0.99, bits.data_ptr(), st)
    f_fwd = lambda: lib.call("mhr_nce_wide_fwd", q_p.data_ptr(), T, negs_p.data_ptr(), n_neg, D, bits.data_ptr(), sp.data_ptr(), scale.data_ptr(),
                             nlive.data_ptr(), part[0].data_ptr(), part[1].data_ptr(), part[2].data_ptr(), lse.data_ptr(), loss.data_ptr(), 0, 0, st)
    f_g = lambda: lib.call("mhr_nce_wide_grad_tile", q_p.data_ptr(), T, negs_p.data_ptr(), n_neg, D, bits.data_ptr(), lse.data_ptr(), w.data_ptr(),
                           scale.data_ptr(), nlive.data_ptr(), gm.data_ptr(), n_neg, st)
    f_pack = lambda: ops.pack_tiles(qn, tiles_per_block=4)
    def f_lib():
        s = torch.mm(qn, ng.t(), out_dtype=torch.float32); fx = torch.mm(pn, ng.t(), out_dtype=torch.float32)
        lib.call("mhr_nce_dense_fwd", s.data_ptr(), fx.data_ptr(), s.shape[1], n_neg, sp.data_ptr(), scale.data_ptr(), 0.99, nlive.data_ptr(), 0, T,
                 lse.data_ptr(), loss.data_ptr(), 0, 0, st)
    flop = 2.0 * T * n_neg * D
    tb, tf, tg, tp, tl = (time_call(f) for f in (f_bits, f_fwd, f_g, f_pack, f_lib))
    print(f"D={D} T={T} n_neg={n_neg}: fix_bits {tb*1e3:.0f} us ({flop/tb/1e9:.0f} TF)  fwd+finalize {tf*1e3:.0f} us ({flop/tf/1e9:.0f} TF)  "
          f"grad_tile {tg*1e3:.0f} us ({flop/tg/1e9:.0f} TF)  pack rows {tp*1e3:.0f} us | library form of the forward (2 GEMMs + epilogue) {tl*1e3:.0f} us "
          f"vs hand-written bits+fwd {(tb+tf)*1e3:.0f} us", flush=True)

# the two plain gradient products of the backward: own core (mhr_wide_gemm_nt on packed operands, packing included) vs library GEMM
for D, T, n_neg in ((1024, 4608, 4096), (1024, 32768, 4096), (2048, 12800, 8192)):
    g = torch.Generator(device="cuda").manual_seed(1)
    G = (torch.randn(T, n_neg, device="cuda", generator=g) * 0.01).bfloat16()
    N = torch.nn.functional.normalize(torch.randn(n_neg, D, device="cuda", generator=g), dim=-1).bfloat16()
    Q = torch.nn.functional.normalize(torch.randn(T, D, device="cuda", generator=g), dim=-1).bfloat16()
    dq = torch.empty(T, D, device="cuda"); dn = torch.zeros(n_neg, D, device="cuda")
    def own():
        nt_p, k1 = ops.pack_tiles_t(N, n_sel=D, tiles_per_block=8); g_p = ops.pack_tiles(G, tiles_per_block=4)
        lib.call("mhr_wide_gemm_nt", nt_p.data_ptr(), D, g_p.data_ptr(), T, k1, 0, dq.data_ptr(), D, 0, st)
        qt_p, k2 = ops.pack_tiles_t(Q, n_sel=D, tiles_per_block=8); gt_p, _ = ops.pack_tiles_t(G, n_sel=n_neg, tiles_per_block=4)
        lib.call("mhr_wide_gemm_nt", qt_p.data_ptr(), D, gt_p.data_ptr(), n_neg, k2, 0, dn.data_ptr(), D, 0, st)
    def libf():
        return torch.mm(G, N, out_dtype=torch.float32), torch.mm(G.t(), Q, out_dtype=torch.float32)
    own(); a, b = libf(); torch.cuda.synchronize()
    e1 = float((dq - a).abs().max() / a.abs().max()); e2 = float((dn - b).abs().max() / b.abs().max())
    to, tl = time_call(own), time_call(libf)
    print(f"plain products D={D} T={T} n_neg={n_neg}: own (4 packs + 2 GEMMs) {to*1e3:.0f} us, library {tl*1e3:.0f} us; max rel diff {e1:.1e} {e2:.1e}", flush=True)
