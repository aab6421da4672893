"""Micro-driver: HSTU attention fwd/bwd at the cfg1 shape (B=128, L=200, 8 heads x 32), with phase stamps when the
stamped library (tools/stamp_nce.py build) is loaded via STAMP=1."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mhr_amd.lib as L_
if os.environ.get("STAMP"):
    L_.LIB_PATH = os.path.join(ROOT, "tools", "_exp", os.environ["STAMP"], "libmhr_hip.so")      # STAMP = variant dir, e.g. stamp or stamp_ASTAMP_FWD
from mhr_amd import ops
B, L, H, hd = int(os.environ.get("B", 128)), 200, 8, 32
D = H * hd
g = torch.Generator(device="cuda").manual_seed(0)
h = (torch.randn(B * L, 4 * D, device="cuda", generator=g) * 0.5).bfloat16()
lens = torch.randint(L // 4, L + 1, (B,), device="cuda", generator=g)
if os.environ.get("FULL"): lens[:] = L
if os.environ.get("LEN"): lens[:] = int(os.environ["LEN"])                  # every sequence this long (how the kernels' time depends on the valid length)
kv = (torch.arange(L, device="cuda")[None, :] >= (L - lens)[:, None]).to(torch.uint8).contiguous()   # front padded
d_out = (torch.randn(B * L, D, device="cuda", generator=g) * 0.1).bfloat16()
dh = torch.zeros_like(h)
lay = None
if os.environ.get("LAYOUT", "1") != "0":                                    # LAYOUT=0: plain launch; ORDER=0: dead-block skip without the reordering
    lay = ops.attn_seq_layout(kv, B, L, order=os.environ.get("ORDER", "1") != "0")[:2]
import functools
ops.hstu_attn_fwd = functools.partial(ops.hstu_attn_fwd, layout=lay)
ops.hstu_attn_bwd = functools.partial(ops.hstu_attn_bwd, layout=lay)
for _ in range(3):
    out, act = ops.hstu_attn_fwd(h, kv, B, L, H, hd, save_act=not os.environ.get("NOACT"))
    ops.hstu_attn_bwd(h, act, kv, d_out, dh, B, L, H, hd)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
ev[0].record(); out, act = ops.hstu_attn_fwd(h, kv, B, L, H, hd, save_act=not os.environ.get("NOACT")); ev[1].record()
ops.hstu_attn_bwd(h, act, kv, d_out, dh, B, L, H, hd); ev[2].record(); torch.cuda.synchronize()
print(f"attn fwd {ev[0].elapsed_time(ev[1])*1e3:.1f} us  bwd {ev[1].elapsed_time(ev[2])*1e3:.1f} us  mean len {float(lens.float().mean()):.0f}")
if os.environ.get("STAMP"):
    dll = ctypes.CDLL(L_.LIB_PATH); buf = (ctypes.c_ulonglong * 16)()
    assert dll.mhr_debug_read_attn_stamps(buf) == 0
    names = ["start", "staged Q,dO", "A0 frags", "A0 pairs", "A0 epilogue", "A1 frags", "A1 pairs", "A1 epilogue", "sync", "restaged K,V",
             "B0 frags", "B0 pairs", "B0 epilogue", "B1 frags", "B1 pairs", "B1 epilogue"]
    if "FWD" in os.environ["STAMP"]:
        names = ["start", "staged K,V", "0 q frags", "0 pairs", "0 store", "1 q frags", "1 pairs", "1 store"]
    for i in range(1, len(names)): print(f"  {names[i]:20s} +{buf[i]-buf[i-1]:8d} cycles")
