"""Fused AdamW for the HSTU / HLLM models: one kernel over the flat dense-parameter buffer, one over the item table (HSTU;
the HLLM twin has no trainable table) whose gradient arrives as a sparse row set (reference: DeepSpeed FusedAdam / torch AdamW over every parameter,
`code/REC/trainer/trainer.py:292-299`; update semantics identical: every table row is updated every step)."""
import os

import torch

from . import distributed as D
from . import ops


GRAD_EPOCH = 0        # bumped by zero_grad(): lets a backward kernel know its first write into p.grad of this step


def fused_views(params):
    """[p0, p1, ...] 2-D parameters with the same trailing dim that FusedAdamW laid out back to back (consecutive in
    `named_parameters()`, e.g. q_proj / k_proj / v_proj or gate_proj / up_proj of a decoder layer): returns
    (bf16 shadow view [sum rows, K], fp32 gradient view [sum rows, K]) of the flat buffers - ONE GEMM operand and ONE
    gradient destination without concatenating or casting anything - or None when they are not adjacent, not laid out,
    or a shadow is stale (somebody edited the parameter since the last optimizer step)."""
    k = params[0].shape[-1]
    rows = 0
    for i, p in enumerate(params):
        sh = getattr(p, "_mhr_bf16", None)
        if sh is None or p.grad is None or p.dim() != 2 or p.shape[-1] != k or p._version != p._mhr_ver or not getattr(p, "_mhr_direct_grad", False):
            return None
        if i and (p.data_ptr() != params[i - 1].data_ptr() + params[i - 1].numel() * 4
                  or sh.data_ptr() != params[i - 1]._mhr_bf16.data_ptr() + params[i - 1].numel() * 2
                  or p.grad.data_ptr() != params[i - 1].grad.data_ptr() + params[i - 1].numel() * 4):
            return None
        rows += p.shape[0]
    return (torch.as_strided(params[0]._mhr_bf16, (rows, k), (k, 1)), torch.as_strided(params[0].grad, (rows, k), (k, 1)))


def fused_flat(params):
    """Like `fused_views` for parameters of any shape (e.g. the H heads' biases): (bf16 shadow, fp32 gradient) as flat views
    over the adjacent parameters, or None."""
    n = 0
    for i, p in enumerate(params):
        sh = getattr(p, "_mhr_bf16", None)
        if sh is None or p.grad is None or p._version != p._mhr_ver or not getattr(p, "_mhr_direct_grad", False):
            return None
        if i and (p.data_ptr() != params[i - 1].data_ptr() + params[i - 1].numel() * 4
                  or sh.data_ptr() != params[i - 1]._mhr_bf16.data_ptr() + params[i - 1].numel() * 2
                  or p.grad.data_ptr() != params[i - 1].grad.data_ptr() + params[i - 1].numel() * 4):
            return None
        n += p.numel()
    return (torch.as_strided(params[0]._mhr_bf16, (n,), (1,)), torch.as_strided(params[0].grad, (n,), (1,)))


GRAPH_SLOTS = 16      # pinned pointer-table slots owned by captured step graphs, per table shape (never recycled)
HOST_RING = 64        # pinned staging slots of the per-step constants upload (the Trainer keeps the host < 48 steps ahead)
LAZY_HIST = 64        # steps a table row may lag behind before everything is flushed (= length of the constants' history)


class FusedAdamW:
    def __init__(self, model, lr=1e-3, weight_decay=0.0, betas=(0.9, 0.999), eps=1e-8, lazy_table=False):
        """lazy_table: update only the rows a step touches and replay the gradient-free steps of a row when it is next read
        (`catch_up`, called by the model's training forward) or flushed (`flush_table`: evaluation, checkpoints, state_dict,
        every LAZY_HIST steps).  Bitwise the same weights as the dense update; the model must read the table through those
        hooks (HSTU does) - leave it off for code that reads `item_embedding.weight` directly while training."""
        self.model = model
        self.lr, self.weight_decay, self.betas, self.eps = lr, weight_decay, betas, eps
        self.step_count = 0
        emb = getattr(model, "item_embedding", None)
        self.table = emb.weight if emb is not None else None
        dev = next(model.parameters()).device
        self.t_m = torch.zeros_like(self.table) if self.table is not None else None
        self.t_v = torch.zeros_like(self.table) if self.table is not None else None
        seen, uniq, names = set(), [], []
        for n, p in model.named_parameters():
            if p.requires_grad and p is not self.table and id(p) not in seen:
                seen.add(id(p))
                uniq.append(p)
                names.append(n)
        # parameters a model wants back to back in the flat buffers (`adjacent_parameters()` -> lists of parameters): their bf16
        # shadows / gradients then form ONE GEMM operand / ONE gradient destination (`fused_views`, `fused_flat`) - e.g. the H
        # decoding heads' weights, which the forward uses as one concatenated [H*D, D] matrix
        order = list(zip(names, uniq))
        for group in (getattr(model, "adjacent_parameters", lambda: [])() or []):
            ids = [id(p) for p in group]
            pos = [i for i, (_, p) in enumerate(order) if id(p) in ids]
            if len(pos) != len(ids):
                continue
            picked = sorted((order[i] for i in pos), key=lambda np_: ids.index(id(np_[1])))
            rest = [np_ for i, np_ in enumerate(order) if i not in set(pos)]
            at = sum(1 for i in range(pos[0]) if i not in set(pos))
            order = rest[:at] + picked + rest[at:]
        names, uniq = [n for n, _ in order], [p for _, p in order]
        self.dense = uniq
        self.layout = [[n, p.numel()] for n, p in zip(names, uniq)]     # order of the flat buffers (checked on load)
        sizes = [(p.numel() + 7) // 8 * 8 for p in self.dense]          # every view 16-byte aligned, fp32 and the bf16 shadow alike
        self.pad = 8                                                     # (recorded with the state: older checkpoints padded to 4)
        total = (sum(sizes) + 7) // 8 * 8                               # (whole 8-column groups for the batched partial sum)
        self.flat_w = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_m = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_v = torch.zeros(total, dtype=torch.float32, device=dev)
        # bf16 shadow of the flat weights, rewritten by the Adam kernel: the GEMM operand of the next step (SplitKLinearFn
        # takes `p._mhr_bf16` while `p._version` still equals `p._mhr_ver`, i.e. nobody edited the parameter in between)
        self.flat_w16 = torch.zeros(total, dtype=torch.bfloat16, device=dev)
        off = 0
        for p, sz in zip(self.dense, sizes):
            n = p.numel()
            self.flat_w[off:off + n].copy_(p.data.reshape(-1))
            p.data = self.flat_w[off:off + n].view(p.shape)
            p.grad = self.flat_g[off:off + n].view(p.shape)
            if os.environ.get("MHR_OPT_SHADOW", "1") != "0":
                p._mhr_bf16 = self.flat_w16[off:off + n].view(p.shape)
                p._mhr_ver = -1                                          # shadow not valid before the first step
            p._mhr_direct_grad = os.environ.get("MHR_OPT_DIRECT", "1") != "0"   # backward kernels may write p.grad in place
            p._mhr_flat_off = off
            p._mhr_opt = None                                            # set by enable_partial_arena()
            off += sz
        self.param_groups = [{"lr": lr}]                                 # scheduler-facing view
        self.lazy = bool(lazy_table) and self.table is not None and os.environ.get("MHR_LAZY_ADAM", "1") != "0"
        # per-step constants (lr decay, bias corrections) of the last LAZY_HIST steps on the device: the lazy table replay
        # reads them, and so does every Adam kernel of a step replayed from a hipGraph (`step_dev`, set by the Trainer's
        # step graph: the step number in device memory, because a replayed launch cannot take it as an argument)
        # Row LAZY_HIST of the same buffer carries the replay control block (two int64: dropout step counter, optimizer step),
        # so ONE small host-to-device copy per step refreshes everything a replayed step reads.  The copy's source is a slot
        # of a pinned ring: an asynchronous copy reads host memory when the DMA runs, not when it is issued.
        self.hist = torch.zeros(LAZY_HIST + 1, 4, dtype=torch.float32, device=dev)
        self.ctrl = self.hist[LAZY_HIST].view(torch.int64)               # device int64[2]
        self._hist_host = torch.zeros(LAZY_HIST + 1, 4, dtype=torch.float32)
        self._hist_ring = torch.zeros(HOST_RING, LAZY_HIST + 1, 4, dtype=torch.float32)
        if dev.type == "cuda":
            self._hist_ring = self._hist_ring.pin_memory()
        self._n_push = 0
        self.step_dev = None
        self.in_graph = False                                            # True while a step is being captured
        if self.lazy:
            self.last_step = torch.zeros(self.table.shape[0], dtype=torch.int32, device=dev)
            model._table_optimizer = self                                # the model's forward / eval hooks find us here
            self._lagging = False                                        # rows behind step_count exist (host-side flag)

    # ---- split-K partials of ALL weight gradients in one arena, reduced by ONE launch at the step -------------------------
    def enable_partial_arena(self, on=True):
        """The projections' weight gradients are split-K batched GEMMs whose S bf16 partials are summed into the flat fp32
        gradient (SplitKLinearFn).  With the arena on, every such GEMM writes its partials straight into its parameter's columns
        of ONE [S, total] buffer and `step()` sums the whole buffer with one launch - 16 small launches per step fewer at cfg1.
        Only valid with exactly one backward per optimizer step (the Trainer turns it on when accumulate_grad == 1): a second
        backward would overwrite the first one's partials."""
        self._arena_on = bool(on) and os.environ.get("MHR_PARTIAL_ARENA", "1") != "0"
        for p in self.dense:
            p._mhr_opt = self if self._arena_on else None
        self._arena = None
        self._arena_used = set()

    def partial_view(self, p, S):
        """[S, *p.shape] bf16 view of the arena for parameter p (batch stride = the arena's row pitch), or None when the arena
        cannot serve this call: a different S than the arena was built for, a parameter already written in this step (tied
        weights: the second use must ADD), or an arena that would be larger than 2 GB (LLM-sized models)."""
        if not getattr(self, "_arena_on", False) or p.dim() != 2:
            return None
        if getattr(self, "_arena", None) is None:
            if S * self.flat_g.numel() * 2 > (2 << 30):
                self._arena_on = False
                return None
            self._arena = torch.zeros(S, self.flat_g.numel(), dtype=torch.bfloat16, device=self.flat_g.device)
            self._arena_used = set()
        off, n = p._mhr_flat_off, p.numel()
        if id(p) in self._arena_used:
            return None
        if self._arena.shape[0] != S:
            if getattr(p, "_mhr_in_arena", False):       # its columns hold an earlier step's partials: never add them again
                self._arena[:, off:off + n].zero_()
                p._mhr_in_arena = False
            return None
        self._arena_used.add(id(p))
        p._mhr_in_arena = True
        return self._arena[:, off:off + n].view(S, *p.shape)

    def finish_grads(self):
        """Fold every deferred reduction into `flat_g` (step() does this itself; anything that reads the flat gradient before
        the step - a test, a norm - calls it first)."""
        self._reduce_arena()

    def defer(self, fn):
        """Run fn() first thing at the step (before the deferred reductions are folded into the flat gradient): work of the
        backward that nothing in the backward waits for (WeightGradStack.flush)."""
        if not hasattr(self, "_deferred"):
            self._deferred = []
        self._deferred.append(fn)
        if len(self._deferred) > 8:               # forwards whose backward never came: their closures do nothing, drop the oldest
            self._deferred.pop(0)

    def _reduce_arena(self):
        todo, self._deferred = getattr(self, "_deferred", []), []
        for fn in todo:
            fn()
        if getattr(self, "_arena", None) is not None and self._arena_used:
            ops.sum_rows_into(self._arena, self.flat_g)
            self._arena_used = set()
        self._reduce_bias_queue()

    # ---- bias gradients of equally shaped projections: ONE column-sum launch for all of them ------------------------------
    def queue_rows_sum(self, src, grad):
        """grad (fp32, contiguous, a parameter's view of the flat gradient) += src.sum(0) for src [rows, grad.numel()] bf16,
        deferred to the step and batched with every other sum of the same shape into one launch."""
        g = grad.view(-1)
        ok = (src.dtype == torch.bfloat16 and src.is_contiguous() and src.dim() == 2 and src.shape[1] == g.numel() and src.shape[1] % 8 == 0
              and g.dtype == torch.float32 and g.is_contiguous() and g.data_ptr() % 16 == 0 and src.data_ptr() % 16 == 0)
        if not ok:
            g.add_(torch.sum(src, 0, dtype=torch.float32).view(-1))
            return
        if not hasattr(self, "_bias_q"):
            self._bias_q = []
        self._bias_q.append((src, g))

    def queue_bias_sum(self, dy, bias_param):
        """Defer `bias.grad += dy.sum(0)` to the step (same one-backward-per-step condition as the arena): the eight output
        projections of the cfg1 encoder each reduce a [B L, D] matrix - alone a launch that cannot fill the chip."""
        if not getattr(self, "_arena_on", False) or dy.dtype != torch.bfloat16 or not dy.is_contiguous() or dy.dim() != 2 \
                or dy.shape[1] % 8 or bias_param.grad is None or not bias_param.grad.is_contiguous():
            return False
        if not hasattr(self, "_bias_q"):
            self._bias_q = []
        self._bias_q.append((dy, bias_param.grad))
        return True

    def _reduce_bias_queue(self):
        q, self._bias_q = getattr(self, "_bias_q", []), []
        groups = {}
        for dy, g in q:
            groups.setdefault(tuple(dy.shape), []).append((dy, g))
        for (rows, cols), items in groups.items():
            if len(items) == 1:
                ops.sum_rows_into(items[0][0], items[0][1])
                continue
            n = len(items)
            key = (n, rows, cols)
            if not hasattr(self, "_bias_tabs"):
                self._bias_tabs = {}
            if key not in self._bias_tabs:       # pinned ring of pointer tables + their device copy (grow-only: captured graphs read them)
                # rows [0, HOST_RING): the ring of host-issued steps; rows [HOST_RING, HOST_RING + GRAPH_SLOTS): one per captured
                # graph, handed out once and never recycled (allocated here, in a host-issued warm-up step: no pinned allocation
                # under capture)
                host = torch.zeros(HOST_RING + GRAPH_SLOTS, 2 * n, dtype=torch.int64)
                if self.flat_g.is_cuda:
                    host = host.pin_memory()
                self._bias_tabs[key] = [host, torch.zeros(2 * n, dtype=torch.int64, device=self.flat_g.device), 0, 0]
            host, devt, k, n_graph = self._bias_tabs[key]
            if self.in_graph:
                # a captured step bakes a copy node whose SOURCE is this pinned slot and re-reads it at every replay: the slot
                # belongs to the graph for good (never a ring slot that host-issued steps of the same shape would recycle)
                if n_graph >= GRAPH_SLOTS:
                    raise RuntimeError(f"more than {GRAPH_SLOTS} captured step graphs share the bias-sum table {key}")
                slot = host[HOST_RING + n_graph]
                self._bias_tabs[key][3] = n_graph + 1
            else:
                slot = host[k % HOST_RING]
                self._bias_tabs[key][2] = k + 1
                self._throttle()
            slot.numpy()[:] = [dy.data_ptr() for dy, _ in items] + [g.data_ptr() for _, g in items]     # (one host write, not 2 n)
            devt.copy_(slot, non_blocking=True)
            ops.sum_rows_many(devt, n, rows, cols, keep=items)

    def shadows_stale(self):
        """Somebody edited a parameter in place since the last optimizer step (load_state_dict, a manual copy_): its bf16 shadow
        - the GEMM operand of the forward, and what a captured step graph reads through baked pointers - no longer matches."""
        return any(hasattr(p, "_mhr_bf16") and p._version != p._mhr_ver for p in self.dense)

    def refresh_shadows(self):
        """Re-cast every bf16 shadow from its fp32 master (one pass over the flat buffer) and mark them valid.  Host-issued
        forwards fall back to casting a stale parameter themselves (SplitKLinearFn checks the version); a REPLAYED step cannot -
        the check is Python and is not replayed - so the Trainer calls this after resume() and in front of a replay whenever
        `shadows_stale()`."""
        if os.environ.get("MHR_OPT_SHADOW", "1") == "0":
            return
        self.flat_w16.copy_(self.flat_w)
        for p in self.dense:
            if hasattr(p, "_mhr_bf16"):
                p._mhr_ver = p._version

    def reset_step_state(self):
        """Drop everything a half-finished step left queued for the optimizer step: deferred closures, split-K partials in the
        arena, queued bias sums, the flat gradient.  (A hipGraph capture that raises partway through the step has run the
        step's Python: its queues reference capture-pool tensors that were never executed.)"""
        self._deferred = []
        self._bias_q = []
        if getattr(self, "_arena", None) is not None:
            for p in self.dense:
                if getattr(p, "_mhr_in_arena", False):
                    p._mhr_in_arena = False
            self._arena.zero_()
        self._arena_used = set()
        self.zero_grad()

    def zero_grad(self):
        global GRAD_EPOCH
        GRAD_EPOCH += 1                                                  # first gradient write of a step overwrites
        self.flat_g.zero_()
        if self.table is not None:
            self.table.grad = None
        self.model.sparse_grad = None

    def graph_capable(self):
        """A step can be replayed from a hipGraph when every per-step scalar can come from device memory: the flat Adam and
        the lazy table Adam can; the dense table pass (`mhr_adam_rows`) takes its constants as arguments."""
        return self.table is None or self.lazy

    def begin_replayed_step(self, lr, seed_counter=0):
        """Host side of a step that is replayed from a hipGraph: advance the counters the captured Python would have advanced
        and upload the step's constants and control block (stream-ordered in front of the replay)."""
        self.step_count += 1
        self.param_groups[0]["lr"] = lr
        self._push_consts(self.step_count, lr, ctrl=(int(seed_counter), self.step_count))
        if self.lazy:
            self._lagging = True

    def end_replayed_step(self):
        if self.lazy and self.step_count % LAZY_HIST == 0:               # nobody lags further than the history reaches
            self.flush_table()

    def step(self):
        self.step_count += 1
        lr = self.param_groups[0]["lr"]
        W = D.world_size()
        self._reduce_arena()                         # the split-K partials of all weight gradients: one launch
        # data parallel: dense bucket and gradient rows go on the wire back to back (RCCL's stream); the flat Adam waits
        # for the bucket only, so it - and the id sort of the row reduction - run underneath the row all-gather
        dense_work = D.allreduce_sum_begin(self.flat_g)
        if self.table is not None and hasattr(self.model, "begin_sparse_exchange"):
            self.model.begin_sparse_exchange()
        if dense_work is not None:
            dense_work.wait()
        sd = self.step_dev if self.in_graph else None
        ops.adam_flat(self.flat_w, self.flat_g, self.flat_m, self.flat_v, self.step_count, lr, 1.0 / W, self.betas, self.eps,
                      self.weight_decay, w_bf16=self.flat_w16, hist=self.hist if sd is not None else None, step_dev=sd)
        for p in self.dense:
            if hasattr(p, "_mhr_bf16"):
                p._mhr_ver = p._version
        if self.table is None:
            return
        sg = self.model.finish_sparse_grad() if hasattr(self.model, "finish_sparse_grad") else self.model.sparse_grad
        if self.lazy and not self.in_graph:          # every step has its constants in the history, with or without a gradient
            self._push_consts(self.step_count, lr)   # (a captured step gets them from begin_replayed_step at every replay)
        if sg is not None and self.lazy:
            self._lazy_call(1, sg.sorted_ids, sg.rows, sg.row_slot, 1.0 / W)
            self._lagging = True
            if self.step_count % LAZY_HIST == 0 and not self.in_graph:   # nobody lags further than the history reaches
                self.flush_table()
        elif sg is not None:
            ops.adam_rows(self.table, self.t_m, self.t_v, sg.rows, sg.row_slot, self.step_count, lr, 1.0 / W, self.betas,
                          self.eps, self.weight_decay)
        elif self.table.grad is not None:                                # dense_embedding_grad mode
            g = self.table.grad
            if self.lazy:                      # a dense step in between lazy ones: everything up to date first, and after it
                self.step_count -= 1
                self.flush_table()
                self.step_count += 1
                self.last_step.fill_(self.step_count)
            D.allreduce_mean_(g)
            ops.adam_rows(self.table, self.t_m, self.t_v, g, None, self.step_count, lr, 1.0, self.betas, self.eps,
                          self.weight_decay)

    # ---- lazy table update -------------------------------------------------------------------
    def _push_consts(self, step, lr, ctrl=None):
        """The step's Adam constants (and, for a replayed step, the control block) into the device-side history: one
        asynchronous copy from a pinned ring slot, no host stall."""
        from . import lib
        row = step % LAZY_HIST
        lib.call("mhr_adam_consts", float(lr), self.betas[0], self.betas[1], self.eps, self.weight_decay, int(step),
                 self._hist_host[row].data_ptr())
        if ctrl is not None:
            c = self._hist_host[LAZY_HIST].view(torch.int64)
            c[0], c[1] = ctrl
        slot = self._hist_ring[self._n_push % HOST_RING]
        self._n_push += 1
        slot.copy_(self._hist_host)
        self.hist.copy_(slot, non_blocking=True)
        self._throttle()

    def _throttle(self):
        """The pinned staging rings have HOST_RING slots and an asynchronous copy reads its slot when the DMA runs: keep the host
        fewer than ~48 uploads ahead of the GPU, whoever pushes (replayed steps of any batch signature, host-issued steps): an
        event every 16 uploads, the third-youngest waited for."""
        if not self.flat_g.is_cuda:
            return
        self._n_upload = getattr(self, "_n_upload", 0) + 1
        if self._n_upload % 16 == 0:
            ev = torch.cuda.Event()
            ev.record()
            evs = self.__dict__.setdefault("_upload_events", [])
            evs.append(ev)
            if len(evs) > 3:
                evs.pop(0).synchronize()

    def _lazy_call(self, mode, ids, grad_rows, row_slot, grad_scale, step=None):
        t = self.table
        sd = self.step_dev if (self.in_graph and mode != 2) else None     # captured launches read the step from the device
        ops._timed_call("mhr_adam_rows_lazy", t.data_ptr(), self.t_m.data_ptr(), self.t_v.data_ptr(), t.shape[0], t.shape[1],
                 ops._ptr(ids), 0 if ids is None else ids.numel(), ops._ptr(grad_rows), ops._ptr(row_slot),
                 self.last_step.data_ptr(), self.hist.data_ptr(), LAZY_HIST, self.step_count if step is None else step,
                 float(grad_scale), self.betas[0], self.betas[1], self.eps, mode, ops._ptr(sd), ops._stream())

    def catch_up(self, ids):
        """Bring the rows of `ids` (int64, duplicates allowed) up to the last optimizer step before they are read."""
        if self.lazy and self.step_count > 0:
            self._lazy_call(0, ids.contiguous(), None, None, 1.0, step=self.step_count + 1)

    def flush_table(self):
        """Every row up to the last optimizer step (no launch when nothing lags)."""
        if self.lazy and self.step_count > 0 and self._lagging:
            self._lazy_call(2, None, None, None, 1.0)
            self._lagging = False

    def state_dict(self):
        self.flush_table()
        sd = {"step": self.step_count, "flat_m": self.flat_m, "flat_v": self.flat_v, "layout": self.layout, "pad": self.pad}
        if self.table is not None:
            sd.update(t_m=self.t_m, t_v=self.t_v)
        return sd

    def load_state_dict(self, sd):
        lay = sd.get("layout")
        if lay is None:
            # a checkpoint from before the layout was recorded: its flat moments follow plain named_parameters() order.  If
            # this optimizer reordered anything (`adjacent_parameters()`), they would land on the wrong parameters - silently.
            plain, seen = [], set()
            for n, p in self.model.named_parameters():
                if p.requires_grad and p is not self.table and id(p) not in seen:
                    seen.add(id(p))
                    plain.append([n, p.numel()])
            if plain != self.layout:
                raise RuntimeError("optimizer state without a 'layout' record (an older checkpoint) and this optimizer's flat "
                                   "buffers are not in plain named_parameters() order: refusing to load the moments onto the "
                                   "wrong parameters")
        elif [list(x) for x in lay] != self.layout:
            raise RuntimeError("optimizer state was saved for a different parameter layout (names / sizes / order of the flat "
                               "moment buffers differ): refusing to load it onto the wrong parameters")
        self.step_count = int(sd["step"])
        pad = int(sd.get("pad", 4))                      # padding of each parameter's slot in the saved flat buffers
        for k in ("flat_m", "flat_v"):
            if pad == self.pad:
                getattr(self, k).copy_(sd[k])
            else:                                        # same parameters in the same order, other slot sizes: slot by slot
                src, dst, so, do = sd[k], getattr(self, k), 0, 0
                for _, n in self.layout:
                    dst[do:do + n].copy_(src[so:so + n])
                    so += (n + pad - 1) // pad * pad
                    do += (n + self.pad - 1) // self.pad * self.pad
        for k in (("t_m", "t_v") if self.table is not None else ()):
            getattr(self, k).copy_(sd[k])
        if self.lazy:                                    # a checkpoint is a flushed state: every row stands at `step`
            self.last_step.fill_(self.step_count)
