"""Data-parallel plumbing over torch.distributed (backend "nccl" == RCCL over xGMI on ROCm; "gloo" in CPU tests).

One process per GPU; users (batch rows) are independent, so the only exchanges of a train step are
  (1) an all-gather of negative item *ids* (model side, REC/model/basemodel.py:all_gather_ids),
  (2) one all-reduce of the flat dense-parameter gradient bucket (a few MB), and
  (3) the sparse item-table gradient: the shared-negative block is identically indexed on every rank, so it is a
      plain all-reduce with no indices; the rank-private positive rows travel as one all-gather of (ids, rows).
The table update itself stays dense and local (Adam moments of untouched rows still decay - reference semantics,
trainer.py:292-299), but only touched rows cross xGMI: at cfg1/W=8 about 60 MB per rank instead of the reference's
465 MB dense table gradient.  xGMI is point-to-point (7 links per GPU), so few large collectives are preferred
over many small ones: (1) is one call for all pools, (2) one call, (3) three (ids, shared block, private rows).
Overlap: every collective is issued with async_op on RCCL's own stream; the shared-negative block leaves as soon as the
loss backward has produced it (it crosses xGMI underneath the encoder backward), the dense bucket and the private rows
leave at the end of the backward and travel underneath the id sort and the flat Adam (`optim.FusedAdamW.step`).
"""
import os

import torch
import torch.distributed as dist

# dtype of the rank-private gradient rows on the wire.  bf16 halves the step's largest collective (W x 26 624 rows x D at
# cfg1); the per-id sum over ranks is still accumulated in fp32 by the segment-sum kernel.  The reference's DeepSpeed
# bf16 runs reduce the whole gradient in bf16 (trainer.py:292-299, ds_config bf16); MHR_DP_WIRE=fp32 keeps fp32 rows.
# MHR_DP_OVERLAP=0: the shared-negative block waits for the end of the backward like the other exchanges (A/B knob)
OVERLAP = os.environ.get("MHR_DP_OVERLAP", "1") != "0"
ROWS_WIRE_DTYPE = torch.float32 if os.environ.get("MHR_DP_WIRE", "bf16") == "fp32" else torch.bfloat16


# While the Trainer captures a step as a sequence of hipGraphs (REC/trainer/trainer.py:_GraphSegments) this is that object:
# a collective is then not issued but recorded as a HOST CALL of the replay plan - the running graph segment is closed in front
# of it, the next one opened behind it - and so is every `wait()` on its handle.  A replayed data-parallel step is therefore
# graph, collective, graph, ..., each graph one cheap launch, the collectives issued by the host exactly where the host-issued
# step issues them (same order, same overlap).  The tensors a recorded collective touches live in the capture's memory pool:
# the same addresses at every replay.
CAPTURE = None


class Handle:
    """Work handle of an asynchronous collective.  `wait()` orders the CURRENT stream behind the collective without blocking the
    host (RCCL); under capture both the issue and the wait become host calls of the replay plan."""

    def __init__(self, issue, start=True):
        self._issue, self._work = issue, None
        if not start:
            return
        if CAPTURE is not None:
            CAPTURE.host_call(self._start)
        else:
            self._start()

    @staticmethod
    def group(issues):
        """Handles of several collectives issued back to back: ONE host call of a replay plan starts them all (each keeps its
        own wait)."""
        hs = [Handle(fn, start=False) for fn in issues]

        def start_all():
            for h in hs:
                h._start()
        if CAPTURE is not None:
            CAPTURE.host_call(start_all)
        else:
            start_all()
        return hs

    def _start(self):
        self._work = self._issue()

    def _wait_now(self):
        w, self._work = self._work, None
        if w is not None:
            w.wait()

    def wait(self):
        if CAPTURE is not None:
            CAPTURE.host_call(self._wait_now)
        else:
            self._wait_now()


def run_collective(fn):
    """A blocking-style collective (the current stream is ordered behind it when fn returns): a host call under capture."""
    if CAPTURE is not None:
        CAPTURE.host_call(fn)
    else:
        fn()


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


# MHR_FORCE_DP=1: take the data-parallel code path even with ONE rank (an initialised process group of size 1): every collective
# of the exchange is then a real RCCL call on real buffers - id all-gather, early shared-negative all-reduce, bucket all-reduce,
# row exchange, and their replay as host calls between hipGraph segments - which is as far as a one-GPU box can rehearse the
# multi-GPU step on RCCL itself (two RCCL ranks cannot share a card; the cross-rank arithmetic is covered by the gloo tests).
FORCE_DP = os.environ.get("MHR_FORCE_DP", "0") == "1"


def active():
    """The step takes the data-parallel path (more than one rank, or MHR_FORCE_DP with an initialised group)."""
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or FORCE_DP)


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def allreduce_mean_(flat):
    """In-place mean all-reduce of one flat bucket."""
    if world_size() > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.div_(world_size())
    return flat


def allreduce_sum_begin(t):
    """Start an in-place SUM all-reduce of `t` on the collective stream and return its work handle (None when not data
    parallel).  `handle.wait()` orders the CURRENT stream after the collective without blocking the host (RCCL), so
    kernels launched in between overlap it.  Nobody may touch `t` until then."""
    if not active():
        return None
    return Handle(lambda: dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True))


class RowExchange:
    """In-flight exchange of a step's embedding-gradient rows (see `begin_row_exchange`)."""

    def __init__(self, ids, rows_priv, rows_shared, w_ids=(), w_rows=()):
        self._ids, self._rows_priv, self._rows_shared = ids, rows_priv, rows_shared
        self._w_ids, self._w_rows = list(w_ids), list(w_rows)

    def wait_ids(self):
        """ids [W*n_private + n_shared] (private ids in rank order, then the shared ids)."""
        for w in self._w_ids:
            w.wait()
        self._w_ids = []
        return self._ids

    def wait_rows(self):
        """(private rows [W*n_private, D] in the wire dtype, shared rows [n_shared, D] fp32 summed over ranks)."""
        for w in self._w_rows:
            w.wait()
        self._w_rows = []
        return self._rows_priv, self._rows_shared


def begin_row_exchange(ids_all, d_rows, n_private, wire_dtype=None, shared_pending=None, ids_pre=None):
    """Start combining the per-rank embedding-gradient rows of a step; returns a RowExchange.

    ids_all [R] int64 / d_rows [R, D] fp32: the first `n_private` entries are rank-private (positives: every rank has
    its own users), the rest are the shared negatives (same ids, same order on every rank after the id all-gather).
    `shared_pending` = (rows [n_shared, D] fp32, work): the shared block's all-reduce was already started earlier in
    the backward (the negatives' gradient is complete before the encoder backward runs, so it travels underneath it);
    d_rows[n_private:] is then ignored (d_rows may hold the private rows only).  `ids_pre`: the exchanged id list
    [W*n_private + n_shared] when it was already all-gathered in the forward (HSTU.forward) - no id collective here.
    The collectives are issued asynchronously in the order ids, shared rows,
    private rows: the caller sorts the ids while the rows are still on the wire.  The caller sums duplicates in fp32
    and scales by 1/W (DDP's gradient mean) inside the fused Adam.
    """
    W = world_size()
    wire = ROWS_WIRE_DTYPE if wire_dtype is None else wire_dtype
    if not active():
        return RowExchange(ids_all, d_rows[:n_private], d_rows[n_private:])
    priv_ids, priv_rows = ids_all[:n_private].contiguous(), d_rows[:n_private].to(wire).contiguous()
    shared_ids = ids_all[n_private:]
    n_sh = shared_ids.numel()
    have_ids = ids_pre is not None and ids_pre.numel() == W * n_private + n_sh
    ids_out = ids_pre if have_ids else torch.empty(W * n_private + n_sh, dtype=ids_all.dtype, device=ids_all.device)
    rows_priv = torch.empty(W * n_private, d_rows.shape[1], dtype=wire, device=d_rows.device)
    if not have_ids:
        ids_out[W * n_private:] = shared_ids
    # gathers go straight into the output (no per-rank list + concatenation); gloo implements all_gather_into_tensor too, so
    # the two-rank CPU / one-card tests run exactly the layout and wait-ordering code that ships over RCCL
    ids_dst, rows_dst, rows_src = ids_out[:W * n_private], rows_priv.view(-1), priv_rows.view(-1)
    issues = [] if have_ids else [lambda: dist.all_gather_into_tensor(ids_dst, priv_ids, async_op=True)]
    w_rows = []
    if shared_pending is not None:
        shared_rows, w_sh = shared_pending
        if w_sh is not None:
            w_rows.append(w_sh)
    else:
        shared_rows = d_rows[n_private:].contiguous()
        if n_sh:
            issues.append(lambda: dist.all_reduce(shared_rows, op=dist.ReduceOp.SUM, async_op=True))
    issues.append(lambda: dist.all_gather_into_tensor(rows_dst, rows_src, async_op=True))
    hs = Handle.group(issues)                                  # [ids], [shared rows], private rows: issued in this order
    n_id = 0 if have_ids else 1
    return RowExchange(ids_out, rows_priv, shared_rows, hs[:n_id], w_rows + hs[n_id:])


def exchange_sparse_rows(ids_all, d_rows, n_private, wire_dtype=None, shared_pending=None):
    """Blocking form of `begin_row_exchange`: returns (ids, private rows, shared rows)."""
    ex = begin_row_exchange(ids_all, d_rows, n_private, wire_dtype, shared_pending)
    ids = ex.wait_ids()
    rows_priv, rows_shared = ex.wait_rows()
    return ids, rows_priv, rows_shared


def allreduce_metric_sums(values):
    """One packed all-reduce for all metric sums of an evaluation (the reference issues one per key,
    trainer.py:1060-1066, 1110-1115).  `values`: 1-D float64/float32 tensor on the collective's device."""
    if world_size() > 1:
        dist.all_reduce(values, op=dist.ReduceOp.SUM)
    return values
