"""Phase stamps of the PLAIN (nothing-suppressed) fused sampled-softmax forward at the cfg1 row-shared shape.
    python tools/stamp_nce.py build ; python tools/stamp_plain.py      (MHR_NCE_ROWS64=0/1 picks the 32- / 64-row kernel)"""
import ctypes, math, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mhr_amd.lib as L
LIB = os.path.join(ROOT, "tools", "_exp", os.environ.get("STAMP", "stamp"), "libmhr_hip.so")
if os.path.exists(LIB):
    L.LIB_PATH = LIB
from mhr_amd import lib, ops
D, G, n_neg, n_row = 256, 4, 8192, int(os.environ.get("NROW", 15700))
cap = (n_row + 255) // 256 * 256 + 256
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
q = torch.randn(cap, D, device=dev, generator=g)
p = torch.randn(cap, D, device=dev, generator=g)
negs = torch.nn.functional.normalize(torch.randn(G, n_neg, D, device=dev, generator=g), dim=-1).bfloat16()
idx = torch.arange(cap, dtype=torch.int32, device=dev)[None].repeat(G, 1).contiguous()
ntd = torch.tensor([n_row] * G, dtype=torch.int32, device=dev)
ls = torch.tensor([math.log(20.0)], device=dev)
ssum = torch.zeros(G, cap, device=dev)
qn = torch.empty(G, cap, D, dtype=torch.bfloat16, device=dev)
qi, pi, sp = (torch.empty(G, cap, device=dev) for _ in range(3))
u = torch.empty(G, cap, D, device=dev)
st = torch.cuda.current_stream().cuda_stream
def run():
    lib.call("mhr_nce_fwd", q.data_ptr(), idx.data_ptr(), p.data_ptr(), idx.data_ptr(), 0, negs.data_ptr(), n_neg, D, G, ntd.data_ptr(),
             cap, ls.data_ptr(), 0.99, ssum.data_ptr(), 0, 0, qn.data_ptr(), 0, 0, qi.data_ptr(), pi.data_ptr(), sp.data_ptr(), -1,
             u.data_ptr(), cap, 0, 0, 0, 0, st)
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
print(f"plain nce_fwd: {e0.elapsed_time(e1):.3f} ms  rows/group {n_row}")
if os.path.exists(LIB):
    dll = ctypes.CDLL(LIB); buf = (ctypes.c_ulonglong * 16)()
    assert dll.mhr_debug_read_stamps(buf) == 0
    names = ["vmcnt wait", "barrier", "loop top / tail", "S MFMAs", "U MFMAs + DMA issue", "epilogue"]
    nt = n_neg // 32
    tot = sum(buf[:6])
    for k_, nm in enumerate(names): print(f"{nm:24s} {buf[k_] / nt:9.0f} cycles/tile {100.0 * buf[k_] / max(1, tot):5.1f} %")
    print(f"total {tot / nt:9.0f} cycles/tile")
