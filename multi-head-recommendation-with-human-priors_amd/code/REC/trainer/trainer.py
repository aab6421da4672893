"""Trainer for the fused HSTU path: same `Trainer(config)`, `setup_model`, `fit`, `evaluate` surface and log keys as
the reference (`code/REC/trainer/trainer.py:55-1153`) without DeepSpeed / Lightning / wandb: plain data-parallel
replicas over RCCL, the fused AdamW, a cosine warm-up schedule, and the fused multi-head decode in evaluation.

Per train step (reference trainer.py:494-536): model(batch) -> loss.backward() -> [RCCL: dense bucket all-reduce +
sparse row exchange] -> fused Adam (flat dense params + item table) -> lr schedule.  The reference's two
`losses.item()` per step are replaced by logging every `update_interval` steps.
Per eval batch (trainer.py:698-729, 985-990): predict_topk (scores never materialised) -> merge/dedup -> hit matrix.
"""
import math
import os
import time
from collections import OrderedDict, defaultdict
from logging import getLogger

import numpy as np
import torch

from REC.evaluator import Collector, Evaluator
from REC.utils import early_stopping
from REC.utils.lr_scheduler import cosine_warmup_factor


def _dp_active():
    from mhr_amd import distributed as D
    return D.active()


class _GraphSegments:
    """The captured step as a SEQUENCE of single-stream hipGraphs instead of one graph with branches.

    A graph whose nodes sit on one stream is launched by ROCm with one cheap host call (0.1 ms for the ~175 nodes of a cfg1
    step); a graph with a forked branch cost 2.4 ms of host time per replay (measured: `host_enqueue_ms_per_step` of bench.py).
    The model's second-stream work (`HSTU.forward`: the batch-only half of the loss) is therefore captured as a graph of its own: `side_branch(fn)` closes the running main segment, captures fn() on the side
    stream into its own graph and memory pool, and opens the next main segment; `join()` closes a segment where the main
    stream must have the side graph's results.  Replay: segments in order on the caller's stream, the side graph on the side
    stream behind an event, the join as an event wait - every launch a linear graph.

    Memory: the main segments share one pool (they replay one after the other, in capture order); the side graph has its own,
    because it runs CONCURRENTLY with the main segment behind it - what it reads from the main pool (gathered rows, ids,
    normalised negatives) is alive until the backward of the same step, what the main segments read from its pool (token
    lists, bit table, accumulators) is written before the join and not touched by it again."""

    def __init__(self, device):
        self.main = torch.cuda.Stream(device=device)
        self.side = torch.cuda.Stream(device=device)
        self.plan, self.cur, self.pool = [], None, None
        self.host_s, self.n_replays = 0.0, 0          # host time spent inside the plan's host calls (the collectives), replays
        self.ev_fork, self.ev_side = torch.cuda.Event(), torch.cuda.Event()

    @staticmethod
    def _capture_mode():
        """With a process group alive, RCCL's watchdog THREAD polls the events of finished collectives (`hipEventQuery`) whenever
        it likes - a call the default 'global' capture mode forbids to every thread of the process while a capture is open
        (seen once: `operation not permitted when stream is capturing` thrown in the watchdog, process aborted, one-rank RCCL
        rehearsal).  'thread_local' confines the check to the capturing thread; begin / cut / end of a capture all happen on it
        (data-parallel captures run the backward single-threaded for that reason)."""
        import torch.distributed as dist
        return "thread_local" if (dist.is_available() and dist.is_initialized()) else "global"

    def begin(self):
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        self._ctx = torch.cuda.stream(self.main)
        self._ctx.__enter__()
        self._open()

    def _open(self):
        g = torch.cuda.CUDAGraph()
        if self.pool is None:
            self.pool = torch.cuda.graph_pool_handle()
        g.capture_begin(pool=self.pool, capture_error_mode=self._capture_mode())
        self.cur = g

    def _close(self):
        self.cur.capture_end()
        self.plan.append(("main", self.cur))
        self.cur = None

    def side_branch(self, fn):
        self._close()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(self.side):
            g.capture_begin(pool=torch.cuda.graph_pool_handle(), capture_error_mode=self._capture_mode())
            try:
                fn()
            finally:
                g.capture_end()
        self.plan.append(("side", g))
        self._open()

    def join(self):
        self._close()
        self.plan.append(("join", None))
        self._open()

    def host_call(self, fn):
        """Work the HOST must issue at this point of every replay (a collective of the data-parallel exchange,
        mhr_amd.distributed.Handle): the running segment ends here, fn is recorded, the next segment begins.  fn is NOT run
        now - a capture executes nothing; every rank captures the same step, so nobody waits for a collective that never comes."""
        self._close()
        self.plan.append(("host", fn))
        self._open()

    def end(self):
        try:
            if self.cur is not None:
                self._close()
        finally:
            ctx, self._ctx = getattr(self, "_ctx", None), None
            if ctx is not None:
                ctx.__exit__(None, None, None)

    def replay(self):
        self.n_replays += 1
        cur = torch.cuda.current_stream()
        for kind, g in self.plan:
            if kind == "main":
                g.replay()
            elif kind == "host":
                t0 = time.perf_counter()
                g()
                self.host_s += time.perf_counter() - t0
            elif kind == "side":
                self.ev_fork.record(cur)
                self.side.wait_event(self.ev_fork)
                with torch.cuda.stream(self.side):
                    g.replay()
                self.ev_side.record(self.side)
            else:
                cur.wait_event(self.ev_side)


class _StepGraph:
    """One optimisation step captured as a hipGraph (torch.cuda.CUDAGraph over the C-ABI launches, the library GEMMs and the
    torch glue of the step) and replayed with ONE host call per step.

    What changes from step to step cannot be a kernel argument of a replayed launch, so it lives in device memory:
      * the batch: static input tensors, refreshed by device-to-device copies in front of the replay;
      * `FusedAdamW.ctrl` int64[2] = (dropout step counter, optimizer step): the ln_gate kernels derive their seeds from
        ctrl[0] (`mhr_ln_gate_*`'s step_seed), the Adam kernels read ctrl[1] (`step_dev`);
      * the step's Adam constants (lr schedule, bias corrections): row `step % 64` of the optimizer's device-side history;
        both travel in ONE small upload from a pinned ring in front of the replay (`FusedAdamW.begin_replayed_step`).
    The host may run ahead of the GPU by at most ~48 uploads (`FusedAdamW._throttle`: an event every 16 uploads, counted over all
    step graphs and host-issued steps, the third-youngest waited for), so the 64-slot pinned rings are never overwritten while a
    copy from them is pending."""
    WARM, RING = 3, 64

    def __init__(self, trainer, sig):
        self.tr, self.sig = trainer, sig
        self.warm, self.graph, self.static, self.out = 0, None, None, None
        self.n = 0

    def release(self):
        self.graph = self.static = self.out = None

    def _capture(self, data):
        tr = self.tr
        model, opt, dev = tr.model, tr.optimizer, data[0].device
        self.static = tuple(t.clone() for t in data)
        for st_, t in zip(self.static, data):                 # host-side hints travel with the static copies
            if getattr(t, "_mhr_rows_cap", None) is not None:
                st_._mhr_rows_cap = t._mhr_rows_cap
        keep = (model._step_seed, opt.step_count, tr.train_step, tr._micro_step, opt.param_groups[0]["lr"])
        graph = _GraphSegments(dev)
        model._seed_dev, opt.step_dev, opt.in_graph = opt.ctrl[0:1], opt.ctrl[1:2], True
        # no cyclic garbage collection while the stream is capturing: a collected object may own pinned host memory or an
        # event whose release is a synchronising HIP call - illegal under capture, and it takes the process down (seen with
        # the pinned staging rings of earlier trainers).  torch.cuda.graph collects once on entry; a step's own Python can
        # trigger the generational collector again.
        import gc
        gc_was = gc.isenabled()
        gc.disable()
        try:
            from mhr_amd import distributed as D
            model._graph_segments = graph
            graph.begin()
            D.CAPTURE = graph if D.active() else None      # data parallel: collectives become host calls of the replay plan
            # a host call ends the running capture and begins the next - legal only on the thread that began it, and the
            # autograd engine runs a device's backward nodes on a worker thread of its own: data-parallel captures (whose
            # backward starts the shared-negative all-reduce) keep the backward on this thread
            import contextlib
            same_thread = (torch.autograd.set_multithreading_enabled(False)
                           if (D.active() or _GraphSegments._capture_mode() == "thread_local") else contextlib.nullcontext())
            try:
                with same_thread:
                    out = tr._eager_step(self.static)
            finally:
                D.CAPTURE = None
                graph.end()
        finally:
            model._graph_segments = None
            if gc_was:
                gc.enable()
            model._seed_dev, opt.step_dev, opt.in_graph = None, None, False
            # the capture ran the Python of one step (counters moved) and launched nothing: put the counters back
            model._step_seed, opt.step_count, tr.train_step, tr._micro_step, opt.param_groups[0]["lr"] = keep
        self.graph, self.out = graph, out

    def step(self, data):
        tr = self.tr
        if self.graph is None:
            if self.warm < self.WARM:          # host-issued steps first: caches, bf16 weight shadows, allocator pools
                self.warm += 1
                return tr._eager_step(data)
            try:
                self._capture(data)
            except Exception as e:  # noqa: BLE001 - a step that cannot be captured still trains, launch by launch
                tr._graph_failed = True
                tr.graph_failure = f"{type(e).__name__}: {e}"
                tr.logger.warning(f"hipGraph capture of the train step failed ({tr.graph_failure}); continuing with host-issued launches")
                self.release()
                # the capture ran part of a step's Python and launched nothing: whatever it queued for the optimizer step
                # (deferred weight-gradient products, arena partials, bias sums, pending table rows, the pre-sorted ids) points
                # at capture-pool tensors that never ran - drop it before the host-issued step
                tr.optimizer.reset_step_state()
                if hasattr(tr.model, "reset_step_state"):
                    tr.model.reset_step_state()
                if tr.config.get("hip_graph_required", False) or os.environ.get("MHR_HIP_GRAPH_REQUIRED", "0") == "1":
                    raise RuntimeError(f"hipGraph capture of the train step failed and hip_graph_required is set: {tr.graph_failure}") from e
                return tr._eager_step(data)
        model, opt = tr.model, tr.optimizer
        if opt.shadows_stale():                     # weights edited in place since the last step (load_state_dict, resume):
            opt.refresh_shadows()                   # the graph reads the bf16 shadows through baked pointers
        pairs = [(s_, t) for s_, t in zip(self.static, data) if s_ is not t]
        if pairs:                                  # the batch into the static inputs: one multi-tensor copy launch per dtype
            torch._foreach_copy_([p[0] for p in pairs], [p[1] for p in pairs], non_blocking=True)
        model._step_seed += 1
        lr = tr._lr_at(tr.train_step)
        tr.train_step += 1
        tr._micro_step += 1
        opt.begin_replayed_step(lr, seed_counter=model._step_seed)        # constants + control block: one small upload
        self.n += 1
        self.graph.replay()
        opt.end_replayed_step()
        return self.out


class _EvalGraph:
    """One evaluation step - encoder, heads, full-catalog multi-head scoring, exact per-head top-k up to its verification
    flags - captured as a hipGraph and replayed (reference trainer.py:698-729 + hstu.py:874-1016 + collector.py:245 per batch).
    Issued launch by launch the step costs the host 2.4 ms for 2.0 ms of GPU work and ends in a blocking read, so the two add up;
    replayed, the host's part is one graph launch in front of that read.  Static inputs: the id window, the target tags and the
    batch's history pairs PADDED to a capacity (pad pairs carry user id B: they sort behind every real user and fall outside the
    CSR offsets the kernels read).  The flags are read after the replay; the rare repair (rows re-run exactly) runs eagerly on the
    graph's tensors.  Graphs bake the pointers of the cached item table / bf16 weight casts: `Trainer.compute_item_feature` drops
    them."""
    WARM, HIST_BLOCK = 2, 8192

    def __init__(self, trainer, key):
        self.tr, self.key = trainer, key
        self.warm, self.graph, self.static, self.finish = 0, None, None, None

    @staticmethod
    def hist_capacity(n):
        return max(_EvalGraph.HIST_BLOCK, -(-n // _EvalGraph.HIST_BLOCK) * _EvalGraph.HIST_BLOCK)

    def _pad_hist(self, hu, hi, cap, B):
        n = hu.numel()
        if n == cap:
            return hu, hi
        return (torch.cat([hu, hu.new_full((cap - n,), B)]), torch.cat([hi, hi.new_zeros(cap - n)]))

    def run(self, item_seq, target_tags, hu, hi, k):
        tr = self.tr
        m = tr.model.module
        B = item_seq.shape[0]
        cap = self.key[-1]
        hu, hi = self._pad_hist(hu, hi, cap, B)
        if self.graph is None:
            if self.warm < self.WARM:                # host-issued first: caches (item table casts, row bits, allocator pools)
                self.warm += 1
                return m.predict_topk(item_seq, tr.item_feature, tr.all_item_tags, target_tags, (hu, hi), k=k)
            try:
                self.static = (item_seq.clone(), target_tags.clone(), hu.clone(), hi.clone())
                g = torch.cuda.CUDAGraph()
                m.sync_table()
                with torch.cuda.graph(g):
                    self.finish = m.predict_topk(self.static[0], tr.item_feature, tr.all_item_tags, self.static[1],
                                                 (self.static[2], self.static[3]), k=k, defer=True)
                self.graph = g
            except Exception as e:  # noqa: BLE001 - an evaluation step that cannot be captured still runs, launch by launch
                tr._eval_graph_failed = f"{type(e).__name__}: {e}"
                tr.logger.warning(f"hipGraph capture of the evaluation step failed ({tr._eval_graph_failed}); continuing host-issued")
                self.graph = self.static = self.finish = None
                return m.predict_topk(item_seq, tr.item_feature, tr.all_item_tags, target_tags, (hu, hi), k=k)
        torch._foreach_copy_(list(self.static), [item_seq, target_tags, hu, hi], non_blocking=True)
        self.graph.replay()
        return self.finish()                         # the decode's one host read (+ the rare exact re-run of flagged rows)


class Trainer(object):
    def __init__(self, config):
        self.config = config
        self.logger = getLogger()
        self.eval_pred_len = config["eval_pred_len"]
        self.metrics_pred_len_list = config["metrics_pred_len_list"]
        self.optim_args = config["optim_args"]
        self.stopping_step = config["stopping_step"]
        self.valid_metric = (config["valid_metric"] or "NDCG@10").lower()
        self.valid_metric_bigger = config.get("valid_metric_bigger", True)
        self.device = config["device"]
        self.rank = torch.distributed.get_rank() if torch.distributed.is_initialized() else 0
        self.world = torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1
        self.total_iters = config["total_iters"]
        self.accumulate_grad = config.get("accumulate_grad", 1)
        self.update_interval = config["update_interval"] if config["update_interval"] else 20
        self.eval_interval = config["eval_interval"]
        self.scheduler_config = config["scheduler_args"]
        self.checkpoint_dir = config["checkpoint_dir"]
        self.saved_model_name = f'{config["model"]}-{config["dataset"]}-{config["save_model_note"]}.pth'
        self.train_step = 0
        self.no_improve_times = 0
        self.best_valid_score = -np.inf if self.valid_metric_bigger else np.inf
        self.best_valid_result = None
        self.eval_collector = Collector(config)
        self.evaluator = Evaluator(config)
        self.item_feature = None
        self.all_item_tags = None
        self.eval_by_cat = config.get("eval_by_cat", True) and config["eval_num_cats"] > 1
        self.outlier_user_metrics = config["outlier_user_metrics"]

    # ------------------------------------------------------------------------------------------
    def setup_model(self, model):
        from mhr_amd.optim import FusedAdamW
        self.model = model
        # lazy table update: only the rows a step touches are written, the gradient-free steps of the others are replayed
        # when they are next read (bitwise the dense result; `lazy_table_adam: False` restores the dense pass per step)
        lazy = self.config.get('lazy_table_adam', True) and not getattr(model, "dense_embedding_grad", False)
        self.optimizer = FusedAdamW(model, lr=self.optim_args['learning_rate'], weight_decay=self.optim_args['weight_decay'],
                                    lazy_table=bool(lazy))
        self._micro_step = 0
        if self.accumulate_grad == 1 and hasattr(self.optimizer, "enable_partial_arena"):
            self.optimizer.enable_partial_arena(True)     # one backward per step: all split-K partials summed by one launch
        for st in self.__dict__.pop("_step_graphs", {}).values():      # graphs captured over a previous model / optimizer
            st.release()
        self._step_graph = None
        if self.accumulate_grad > 1:                # the item-table gradient rows of all micro-batches are reduced at the step
            model.accumulate_rows = True

    def _lr_at(self, step):
        base = self.optim_args['learning_rate']
        if not self.scheduler_config:
            return base
        warm = self.total_iters * self.scheduler_config.get("warmup", 0.001)
        return base * cosine_warmup_factor(step, warm, self.total_iters)

    def _check_nan(self, loss):
        if torch.isnan(loss):
            raise ValueError('Training loss is nan')
        if loss.is_cuda:                       # (the loss was just read: the device is idle, one more word costs nothing)
            from mhr_amd import ops
            bad = ops.bad_id_count()
            if bad:
                raise IndexError(f"{bad} item ids outside [0, item_num) reached the embedding gather (the reference's nn.Embedding "
                                 f"raises 'index out of range in self' for them)")

    def train_step_fn(self, data, graph=None):
        """One forward / backward; every `accumulate_grad`-th call also exchanges the gradients and runs the fused Adam
        (reference trainer.py:511-536).  Returns the model_out dict (device tensors).

        `hip_graph` (config, default on): after three host-issued steps on a batch signature the whole step - forward,
        backward, fused Adam, zero_grad: about 300 launches - is captured ONCE as a hipGraph and every later step of that
        signature is one graph launch (`_StepGraph`).  The returned tensors are then the graph's static outputs: they are
        overwritten by the next step (copy what must outlive it).  `graph=False` forces a host-issued step."""
        if graph is not False and self._graph_ok(data):
            return self._graph_step(data)
        return self._eager_step(data)

    def _eager_step(self, data):
        model_out = self.model(data)
        loss = model_out["loss"]
        # d(loss / accumulate_grad): the scale enters as the root gradient (a cached device scalar) - no division kernel in the
        # forward, no ones-fill and no division backward in front of the loss kernels' backward
        gs = self.__dict__.get("_grad_scale")
        if gs is None or gs.device != loss.device or gs.dtype != loss.dtype:
            gs = self._grad_scale = torch.full((), 1.0 / self.accumulate_grad, dtype=loss.dtype, device=loss.device)
        loss.backward(gradient=gs)
        self._micro_step += 1
        if self._micro_step % self.accumulate_grad == 0:
            self.optimizer.param_groups[0]["lr"] = self._lr_at(self.train_step)
            self.train_step += 1
            self.optimizer.step()
            self.optimizer.zero_grad()
        # the backward is done: hand back values, not the autograd graph.  (A caller that keeps the loss of the previous step
        # alive would keep that step's AccumulateGrad nodes - bound to the stream they were created on - alive into the next
        # one; under a hipGraph capture on its side stream that is a cross-stream dependency and breaks the capture.)
        return {k: (v.detach() if torch.is_tensor(v) else v) for k, v in model_out.items()}

    # ---- hipGraph replay of the step -------------------------------------------------------------
    def _graph_ok(self, data):
        m = self.model
        return (self.config.get("hip_graph", True) and os.environ.get("MHR_HIP_GRAPH", "1") != "0"
                and self.accumulate_grad == 1 and (not _dp_active() or os.environ.get("MHR_DP_GRAPH", "1") != "0")
                and getattr(m, "graph_capable", False)
                and self.optimizer.graph_capable() and isinstance(data, (tuple, list))
                and all(torch.is_tensor(t) and t.is_cuda for t in data) and m.training
                and not getattr(m, "dense_embedding_grad", False) and not getattr(self, "_graph_failed", False))

    def _graph_step(self, data):
        # one graph per batch signature (the ragged last batch of an epoch has its own), the four most recent kept
        # (a batch may carry host-side hints the captured step is specialised on: the row capacity of the packed encoder)
        hints = tuple(getattr(t, "_mhr_rows_cap", None) for t in data)
        sig = tuple((tuple(t.shape), t.dtype) for t in data) + (hints,)
        graphs = self.__dict__.setdefault("_step_graphs", OrderedDict())
        st = graphs.get(sig)
        if st is None:
            if len(graphs) >= (8 if any(h is not None for h in hints) else 4):      # (one graph per row capacity of a packed batch)
                graphs.popitem(last=False)[1].release()
            st = graphs[sig] = _StepGraph(self, sig)
        graphs.move_to_end(sig)
        self._step_graph = st
        return st.step(data)

    @property
    def graph_active(self):
        st = getattr(self, "_step_graph", None)
        return st is not None and st.graph is not None

    def fit(self, train_data, valid_data=None, verbose=True, saved=True, show_progress=False, callback_fn=None):
        self.model.train()
        iterator = iter(train_data)
        t0 = time.time()
        running = None
        for it in range(self.train_step * self.accumulate_grad, self.total_iters * self.accumulate_grad):
            try:
                data = next(iterator)
            except StopIteration:
                iterator = iter(train_data)
                data = next(iterator)
            if isinstance(data, dict):                                         # HLLM batches are dicts (hllm.py:482-493)
                data = {k: (v.to(self.device, non_blocking=True) if torch.is_tensor(v) else v) for k, v in data.items()}
            else:
                data = tuple(d.to(self.device, non_blocking=True) for d in data)
            out = self.train_step_fn(data)
            running = out
            if self._micro_step % self.accumulate_grad:
                continue
            if verbose and self.train_step % self.update_interval == 0:
                loss = out["loss"].item()                                   # the only host sync of the train loop
                self._check_nan(out["loss"])
                if self.rank == 0:
                    extra = {k: (float(v) if torch.is_tensor(v) else v) for k, v in out.items() if k != "loss"}
                    self.logger.info(f"step {self.train_step} loss {loss:.4f} lr {self.optimizer.param_groups[0]['lr']:.3e} "
                                     f"{(time.time() - t0) / self.train_step:.3f}s/step {extra}")
            if valid_data is not None and self.eval_interval and self.train_step % self.eval_interval == 0:
                result = self.evaluate(valid_data, load_best_model=False)
                key = f"pred_{self.metrics_pred_len_list[-1]}"
                score = result[key].get(self.valid_metric, 0.0)
                self.best_valid_score, self.no_improve_times, stop, update = early_stopping(
                    score, self.best_valid_score, self.no_improve_times, self.stopping_step, self.valid_metric_bigger)
                if update:
                    self.best_valid_result = result
                    if saved and self.checkpoint_dir:
                        self._save_checkpoint()
                if callback_fn:
                    callback_fn(self.train_step, score)
                self.model.train()
                if stop:
                    break
        return self.best_valid_score, self.best_valid_result

    def _save_checkpoint(self):
        """DP replicas are identical: rank 0 writes model + optimizer state, the step counters, the dropout step counter of the
        model and the RNG states (reference trainer.py:319-340 saves model, optimizer, rng_state / cuda_rng_state and config)."""
        if self.rank == 0:
            os.makedirs(self.checkpoint_dir, exist_ok=True)
            m = self.model.module if hasattr(self.model, "module") else self.model
            ck = {"model": self.model.state_dict(), "optimizer": self.optimizer.state_dict(), "iter_idx": self.train_step,
                  "best_valid_score": self.best_valid_score, "micro_step": int(getattr(self, "_micro_step", 0)),
                  "step_seed": int(getattr(m, "_step_seed", 0)), "rng_state": torch.get_rng_state(),
                  "config": {k: v for k, v in getattr(self.config, "final_config_dict", {}).items()
                             if isinstance(v, (int, float, str, bool, list, tuple, type(None)))}}
            if self.device.type == "cuda":
                ck["cuda_rng_state"] = torch.cuda.get_rng_state(self.device)
            torch.save(ck, os.path.join(self.checkpoint_dir, self.saved_model_name))
        if self.world > 1:
            torch.distributed.barrier()

    def resume(self, path):
        """Continue a run: weights, optimizer moments and step count, lr-schedule position, the model's dropout step counter
        (the masks of step n + 1 follow, they do not restart at step 1) and the RNG streams the batchers draw from."""
        ck = torch.load(path, map_location=self.device, weights_only=True)
        self.model.load_state_dict(ck["model"])
        self.optimizer.load_state_dict(ck["optimizer"])
        self.train_step = int(ck["iter_idx"])
        self.best_valid_score = ck["best_valid_score"]
        self._micro_step = int(ck.get("micro_step", 0))
        m = self.model.module if hasattr(self.model, "module") else self.model
        if hasattr(m, "_step_seed"):
            m._step_seed = int(ck.get("step_seed", self.train_step))
        if "rng_state" in ck:
            torch.set_rng_state(ck["rng_state"].cpu())
        if "cuda_rng_state" in ck and self.device.type == "cuda":
            torch.cuda.set_rng_state(ck["cuda_rng_state"].cpu(), self.device)
        # captured step graphs read the fp32 masters in place and their counters from device memory, but the GEMM operands are
        # the optimizer's bf16 shadows of the weights: re-cast them from the loaded masters (a replay runs no Python that could
        # notice a stale shadow)
        if hasattr(self.optimizer, "refresh_shadows"):
            self.optimizer.refresh_shadows()

    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def compute_item_feature(self, item_tags=None, item_loader=None):
        """Cached item table + [C, N] tags for the evaluation (reference trainer.py:731-824).
        ID models (and HLLM with a frozen, already cached item tower): `compute_item_all()` and the data layer's multi-hot
        item_tags [N, C].  Text models: `item_loader` yields the reference's packed item-text batches (dicts with
        pos_input_ids / pos_position_ids / pos_cu_input_lens / pos_tag_categories / pos_original_tag_categories,
        batchset.py:57-268) in item-id order; every batch goes through the item tower once (`mode='compute_item'`,
        trainer.py:770) and the embeddings / tags are concatenated."""
        self.model.eval()
        if item_loader is not None:
            feats, tags, orig = [], [], []
            for items in item_loader:
                items = {k: (v.to(self.device, non_blocking=True) if torch.is_tensor(v) else v) for k, v in items.items()}
                f, t, o = self.model(items, mode="compute_item")
                feats.append(f)
                tags.append(t)
                orig.append(o)
            self.item_feature = torch.cat(feats)
            item_tags, orig_tags = torch.cat(tags), torch.cat(orig)
        else:
            self.item_feature = self.model.module.compute_item_all()
            orig_tags = item_tags
        self.eval_collector.set_all_tags(orig_tags.long())
        self.all_item_tags = item_tags.long().transpose(0, 1).contiguous()
        self.__dict__.pop("_eval_graphs", None)          # captured evaluation steps read the previous table / weight casts

    @torch.no_grad()
    def cache_item_tower(self, item_loader):
        """freeze_item_llm: run the item tower over the catalog once and hand the table to the model
        (reference trainer.py:466-470 -> hllm.py:889-891)."""
        self.compute_item_feature(item_loader=item_loader)
        self.model.module.set_all_item_embeds(self.item_feature)
        return self.item_feature

    @torch.no_grad()
    def _full_sort_batch_eval(self, batched_data, stats=None, graph=None):
        """-> (FusedTopK, positive_u, item_target, target_tags, outlier_users).  After two host-issued batches of a shape the step
        is replayed from a hipGraph (`_EvalGraph`); its FusedTopK then lives in the graph's static tensors and is overwritten by the
        next batch - consume it (the Collector does) or clone it.  graph=False forces the launch-by-launch form."""
        _, item_seq, item_target, history_index, positive_u, time_seq, target_tags, outlier_users = batched_data
        dev = self.device
        item_seq, item_target, target_tags = item_seq.to(dev), item_target.to(dev), target_tags.to(dev)
        hist = history_index if self.config.get("suppress_history", True) else None
        k = max(self.config["topk"])
        if stats is None and graph is not False and hist is not None and self._eval_graph_ok(item_seq):
            hu, hi = hist[0].to(dev), hist[1].to(dev)
            key = (tuple(item_seq.shape), tuple(target_tags.shape), _EvalGraph.hist_capacity(hu.numel()))
            graphs = self.__dict__.setdefault("_eval_graphs", OrderedDict())
            eg = graphs.get(key)
            if eg is None:
                if len(graphs) >= 4:
                    graphs.popitem(last=False)
                eg = graphs[key] = _EvalGraph(self, key)
            fused = eg.run(item_seq, target_tags, hu, hi, k)
        else:
            fused = self.model.module.predict_topk(item_seq, self.item_feature, self.all_item_tags, target_tags, hist, k=k, stats=stats)
        return fused, positive_u, item_target, target_tags, outlier_users

    def _eval_graph_ok(self, item_seq):
        """The evaluation step can be replayed: streaming decode (feature dim <= 256, a catalog beyond the small-catalog path),
        fp32-ranked top-k with its deferred verification, static head constraints (no prior given at test time, no prior switch)."""
        m = self.model.module
        from mhr_amd import ops
        # OFF by default: measured at cfg1 (B = 256, one MI355X) the replayed step is no faster than the host-issued one (2.36 vs
        # 2.30 ms) - the step is GPU-bound and ends in the decode's verification read either way; it frees 2.3 ms of host time per
        # batch for whoever needs the host (`hip_graph_eval: True` / MHR_HIP_GRAPH_EVAL=1)
        return ((self.config.get("hip_graph_eval", False) or os.environ.get("MHR_HIP_GRAPH_EVAL", "0") == "1")
                and item_seq.is_cuda and type(m).__name__ == "HSTU" and getattr(m, "_hstu_embedding_dim", 0) in ops.STREAM_DIMS
                and getattr(m, "exact_fp32_topk", False) and m.item_num > 4096 and not getattr(m, "prior_given_at_test", False)
                and getattr(m, "prior_switch", None) is None and isinstance(m.item_id_proj_tower, torch.nn.Identity)
                and not getattr(self, "_eval_graph_failed", None))

    @torch.no_grad()
    def evaluate(self, eval_data, load_best_model=False, show_progress=False, init_model=False, item_tags=None):
        self.model.eval()
        if item_tags is None:
            item_tags = eval_data.item_tags
        self.compute_item_feature(item_tags)
        n_local = 0
        for batched in eval_data:
            fused, positive_u, positive_i, target_tags, outlier = self._full_sort_batch_eval(batched)
            self.eval_collector.eval_batch_collect(
                fused, positive_u, positive_i, tag_category=target_tags if self.eval_by_cat else None,
                outlier_users=outlier if self.outlier_user_metrics is not None else None)
            n_local += positive_i.shape[0]
        n_total = getattr(eval_data, "num_total_examples", None)
        # metric sums -> one packed all-reduce (the reference reduces key by key, trainer.py:1109-1123)
        results = OrderedDict()
        struct = self.eval_collector.get_data_struct(-1)
        results['shared'] = self.evaluator.evaluate(struct, pred_len=-1) if 'rec.rec_tags' in struct else OrderedDict()
        self.eval_collector.reset_all_tags()
        for p in self.metrics_pred_len_list:
            results[f"pred_{p}"] = self.evaluator.evaluate(self.eval_collector.get_data_struct(p), pred_len=p)
        flat, layout = [float(n_local)], []
        for name, res in results.items():
            for k in sorted(res.keys()):
                v = res[k]
                if isinstance(v, tuple):
                    flat += [float(v[0]), float(v[1])]
                    layout.append((name, k, True))
                else:
                    flat.append(float(v))
                    layout.append((name, k, False))
        from mhr_amd import distributed as D
        vec = D.allreduce_metric_sums(torch.tensor(flat, dtype=torch.float64, device=self.device)).tolist()
        total = n_total if n_total is not None else vec[0]
        dp = 5 if self.config["metric_decimal_place"] is None else self.config["metric_decimal_place"]
        summary, i = OrderedDict((name, OrderedDict()) for name in results), 1
        for name, k, is_tuple in layout:
            if is_tuple:
                summary[name][k] = round(vec[i] / max(1, vec[i + 1]), dp)
                i += 2
            else:
                summary[name][k] = round(vec[i] / max(1, total), dp)
                i += 1
        self.model.train()
        return summary
