"""One-rank RCCL rehearsal of the exact torch.distributed calls the data-parallel step makes (a one-GPU box cannot host two
RCCL ranks: 'Duplicate GPU detected').  Checks that the calls are accepted on the nccl backend with the tensor views
the product path passes (sliced outputs, bf16 rows, async handles) and that wait() orders the compute stream."""
import os
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29655")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
W, n_priv, n_sh, D = 1, 1000, 512, 256
ids_all = torch.randint(1, 5000, (n_priv + n_sh,), device=dev)
rows = torch.randn(n_priv + n_sh, D, device=dev)
ids_out = torch.empty(W * n_priv + n_sh, dtype=torch.int64, device=dev)
rows_priv = torch.empty(W * n_priv, D, dtype=torch.bfloat16, device=dev)
w1 = dist.all_gather_into_tensor(ids_out[:W * n_priv], ids_all[:n_priv].contiguous(), async_op=True)
sh = rows[n_priv:].contiguous()
w2 = dist.all_reduce(sh, op=dist.ReduceOp.SUM, async_op=True)
w3 = dist.all_gather_into_tensor(rows_priv, rows[:n_priv].to(torch.bfloat16).contiguous(), async_op=True)
flat = torch.randn(3_000_000, device=dev)
ref = flat.clone()
w4 = dist.all_reduce(flat, async_op=True)
for w in (w1, w2, w3, w4):
    w.wait()
neg = torch.randint(1, 5000, (128, 4, 8), device=dev)
out = torch.empty((W,) + tuple(neg.shape), dtype=neg.dtype, device=dev)
dist.all_gather_into_tensor(out, neg)
dist.barrier()
torch.cuda.synchronize()
assert torch.equal(ids_out[:n_priv], ids_all[:n_priv]) and torch.equal(flat, ref) and torch.equal(out[0], neg)
assert torch.equal(rows_priv.float(), rows[:n_priv].to(torch.bfloat16).float()) and torch.equal(sh, rows[n_priv:])
print("nccl api probe ok", dist.get_backend())
dist.destroy_process_group()
