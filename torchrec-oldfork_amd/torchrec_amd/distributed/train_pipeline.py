"""TrainPipelineSparseDist for the MI355X path (torchrec/distributed/train_pipeline.py:422-558).

Same three stages and stream roles as the reference — (1) host->device copy of batch i+2 on a
memcpy stream, (2) input_dist (id all-to-all) of batch i+1 on a data_dist stream, (3)
forward/backward/optimizer of batch i on the default stream — without the reference fork's
debug file write on the hot loop (train_pipeline.py:538-544) and without fx tracing: the model
exposes its sharded modules directly (DistributedModelParallel.sharded_modules()), and the
pipelined forward is injected through `set_precomputed_input_dist`.
"""
from typing import Any, Iterator, List, Optional, Tuple

import torch

from fbgemm_gpu._streams import side_stream

from ..profiling import label
from .model_parallel import DistributedModelParallel


class _PipelinedEBC:
    """The pipelined forward of a ShardedEmbeddingBagCollection: consumes the input_dist result computed earlier on the
    data_dist stream.  Installed the way the reference installs its PipelinedForward — `sharded.forward` is rewritten on
    the INSTANCE (train_pipeline.py:193-243) — so the module tree, parameter names and state_dict keys stay what they were
    before the pipeline touched the model (a wrapper module in the tree renamed every key under it)."""

    def __init__(self, sharded, pipeline: "TrainPipelineSparseDist") -> None:
        self.sharded = sharded
        self.pipeline = pipeline

    def _dist_input(self, features):
        req = self.pipeline._requests.pop(id(self.sharded), None)
        if req is None:
            return self.sharded.input_dist(features).wait()
        cur = torch.cuda.current_stream()
        with label("## wait_sparse_data_dist ##"):  # train_pipeline.py:213
            with torch.cuda.stream(self.pipeline._data_dist_stream):
                dist_in = req.wait()
            cur.wait_stream(self.pipeline._data_dist_stream)
        dist_in.record_stream(cur)
        return dist_in

    def forward(self, features):
        return self.sharded.compute_and_output_dist(self._dist_input(features))

    def compute_explicit(self, features, prefetched: bool = False):
        """The no-autograd step of the wrapped collection (embeddingbag.py ExplicitLookupStep) on this batch's queued
        input dist, or None when the collection cannot run it (nothing is consumed then)."""
        if not hasattr(self.sharded, "compute_explicit") or not self.sharded.explicit_step_supported(features.stride()):
            return None
        return self.sharded.compute_explicit(self._dist_input(features), prefetched=prefetched)


class TrainPipelineSparseDist:
    def __init__(self, model: torch.nn.Module, optimizer: Any, device: torch.device, hip_graphs: bool = False,
                 wgrad_overlap: Optional[bool] = None, prefetch_lookup: Optional[bool] = None) -> None:
        self._model, self._optimizer, self._device = model, optimizer, device
        # wgrad_overlap: the dense layers' weight-gradient GEMMs of an eager step run on a side stream, joined right
        # after backward (modules/mlp.py _WgradOverlap).  Opt-in (argument or TORCHREC_AMD_WGRAD_OVERLAP=1), and only
        # when nothing in the model is wrapped in DistributedDataParallel: measured on MI355X at batch 65 536 the step
        # gets 3 % SLOWER (8.81 vs 8.56 ms) — two MFMA-bound GEMMs sharing the chip lose more than the HBM-bound passes
        # beside them gain, and the device offers no stream priority below the default to confine the side stream to
        # idle CUs.
        self._setup_wgrad_overlap(wgrad_overlap)
        # hip_graphs: capture the model's collective-free dense segments as HIP graphs on the first
        # batch (models that offer `capture_hip_graphs(batch_size)`, distributed/hip_graph.py)
        # Under DistributedDataParallel the capture must happen BEFORE the DDP wrap (capturing a backward
        # graph over DDP-managed parameters crashes in hipStreamEndCapture on this stack): build the
        # DistributedModelParallel with init_data_parallel=False, call capture_hip_graphs(batch), then
        # init_data_parallel() — bench.py does.  The lazy capture below is for un-wrapped models only.
        self._hip_graphs = hip_graphs and device.type == "cuda"
        if isinstance(model, DistributedModelParallel) and model.is_data_parallel_wrapped():
            self._hip_graphs = False
        use_streams = device.type == "cuda"
        # streams on other hardware queues than the compute stream's (fbgemm_gpu/_streams.py)
        self._memcpy_stream = side_stream(device) if use_streams else None
        self._data_dist_stream = side_stream(device) if use_streams else None
        self._requests = {}
        # explicit-step models: enqueue batch i+1's lookup + pooled all-to-all right behind batch i's embedding backward
        # (prefetch_lookup=True / False, or TORCHREC_AMD_PREFETCH_LOOKUP=1 / 0).  Unset: on when the model kept part of its
        # weight gradients for exactly that window (DLRMTrain.wants_lookup_prefetch: the late weight-gradient graph then
        # runs while the prefetched all-to-all is on the links), off otherwise — on its own the prefetch measured 0.5 %
        # SLOWER in the one-rank rehearsal (nothing but the dense SGD is left to overlap: DESIGN.md §3c)
        import os
        env = os.environ.get("TORCHREC_AMD_PREFETCH_LOOKUP")
        self._prefetch = bool(prefetch_lookup) if prefetch_lookup is not None else (env == "1" if env is not None else None)
        self._batch_i = None
        self._batch_ip1 = None
        self._batch_ip2 = None
        self._connected = False
        dmp = model if isinstance(model, DistributedModelParallel) else None
        self._sharded = dmp.sharded_modules() if dmp is not None else []
        self._install()

    def _setup_wgrad_overlap(self, want: Optional[bool]) -> None:
        import os

        from torch.nn.parallel import DistributedDataParallel

        from ..modules.mlp import _WgradOverlap

        root = self._model.module if isinstance(self._model, DistributedModelParallel) else self._model
        ok = (self._device.type == "cuda" and hasattr(root, "finish_dense_grads")
              and not any(isinstance(m, DistributedDataParallel) for m in self._model.modules()))
        if want is None:
            want = os.environ.get("TORCHREC_AMD_WGRAD_OVERLAP", "0") == "1"
        self._wgrad_overlap = bool(want and ok)
        # eager steps: the dense gradients' final reductions in one launch (modules/mlp.py _DeferredFinish): the same
        # preconditions (this pipeline calls finish_dense_grads() right after backward, no DistributedDataParallel hooks).
        # Opt-in (TORCHREC_AMD_DEFERRED_FINISH=1): measured neutral at batch 65 536 (8.549 / 8.573 ms with, 8.504 / 8.560
        # without, same box) — an eager step of that size is bound by bytes, not by its 16 small launches; the graph-mode
        # counterpart (hip_graph.py write_sinks) is what pays, at the per-rank batches of N > 1.
        self._deferred_finish = bool(ok and os.environ.get("TORCHREC_AMD_DEFERRED_FINISH", "0") == "1")

    def _install(self) -> None:
        self._wrappers: List[_PipelinedEBC] = []
        for s in self._sharded:
            w = getattr(s, "_pipelined", None)
            if w is None or w.pipeline is not self:
                w = _PipelinedEBC(s, self)
                s.forward = w.forward  # instance attribute: shadows the class's forward for nn.Module.__call__
                s._pipelined = w       # explicit-step models reach compute_explicit() on the queued input dist through it
            self._wrappers.append(w)

    def _prefetch_next_lookup(self):
        """Lookup + pooled all-to-all of batch i+1, enqueued behind the last backward launch of batch i (see
        DLRMTrain.set_between_forward_and_backward).  Returns (its KeyedJaggedTensor, ExplicitLookupStep) or None."""
        nxt = self._batch_ip1
        if nxt is None or len(self._wrappers) != 1:
            return None
        w = self._wrappers[0]
        if id(w.sharded) not in self._requests:  # its input dist was not queued (should not happen)
            return None
        emb = getattr(w.sharded, "_emb_module", None)
        if emb is None or getattr(emb, "_cache", None) is not None:
            return None  # a row cache allows one outstanding training forward; keep those strictly in order
        with label("## prefetch_next_lookup ##"):
            step = w.compute_explicit(nxt.sparse_features, prefetched=True)
        return (nxt.sparse_features, step) if step is not None else None

    def _run_backward(self, losses) -> None:
        if self._wgrad_overlap:
            from ..modules.mlp import _WgradOverlap

            _WgradOverlap.enable(self._device)  # for this backward only: joined right below
            try:
                torch.sum(losses, dim=0).backward()
            finally:
                _WgradOverlap.disable()
        else:
            torch.sum(losses, dim=0).backward()

    def _to_device(self, batch, non_blocking: bool):
        return batch.to(self._device, non_blocking=non_blocking) if batch is not None else None

    def _start_data_dist(self, batch) -> None:
        """Called with the data_dist stream current: the ids were copied on the memcpy stream, so the caching
        allocator must learn that this stream reads them too (the reference's `_wait_for_batch`,
        train_pipeline.py:58-71, 228) — otherwise dropping the batch lets the next host-to-device copy reuse the
        blocks while the id exchange is still queued."""
        if self._data_dist_stream is not None:
            batch.sparse_features.record_stream(self._data_dist_stream)
        for s in self._sharded:
            self._requests[id(s)] = s.input_dist(batch.sparse_features)

    def _fill(self, it: Iterator) -> None:
        if self._memcpy_stream is None:
            self._batch_i = self._to_device(next(it), False)
            self._batch_ip1 = self._to_device(next(it, None), False)
            self._connected = True
            return
        with torch.cuda.stream(self._memcpy_stream):
            self._batch_i = self._to_device(next(it), True)
            self._batch_ip1 = self._to_device(next(it, None), True)
        if self._hip_graphs:
            root = self._model.module if isinstance(self._model, DistributedModelParallel) else self._model
            if hasattr(root, "capture_hip_graphs"):
                B = int(self._batch_i.dense_features.shape[0])
                have = getattr(root, "_graphs", None)
                if have is not None and have[0] == B:
                    # the owner captured already (the documented N > 1 flow: DistributedModelParallel(init_data_parallel=
                    # False) -> capture_hip_graphs(B, flat_grads=True, process_group=pg) -> init_data_parallel()).  Capturing
                    # again would replace the flat gradient state by a world-1 one — the dense all-reduce would silently
                    # disappear and the replicas diverge — and strand a FlatSGD built on the old buffers.
                    pass
                else:
                    env = self._model._env if isinstance(self._model, DistributedModelParallel) else None
                    if env is not None and env.world_size > 1:
                        raise RuntimeError(
                            "TrainPipelineSparseDist(hip_graphs=True): the lazy capture is for world size 1.  With a process "
                            "group capture BEFORE the data-parallel setup: DistributedModelParallel(init_data_parallel=False), "
                            "model.module.capture_hip_graphs(batch, flat_grads=True, process_group=env.process_group), "
                            "model.init_data_parallel() — bench.py does")
                    self._memcpy_stream.synchronize()
                    # before any collective is in flight; flat-gradient mode, so that the step runs without the autograd
                    # engine (models/dlrm.py explicit step)
                    root.capture_hip_graphs(B, flat_grads=True)
        with torch.cuda.stream(self._data_dist_stream):
            self._data_dist_stream.wait_stream(self._memcpy_stream)
            self._start_data_dist(self._batch_i)
        self._connected = True

    def progress(self, dataloader_iter: Iterator) -> Any:
        if not self._connected:
            self._fill(dataloader_iter)
        if self._batch_i is None:
            raise StopIteration
        if self._memcpy_stream is not None:
            with label("## copy_batch_to_gpu ##"):  # train_pipeline.py:507
                with torch.cuda.stream(self._memcpy_stream):
                    self._batch_ip2 = self._to_device(next(dataloader_iter, None), True)
            with label("## wait_for_batch ##"):  # train_pipeline.py:516
                torch.cuda.current_stream().wait_stream(self._data_dist_stream)
        else:
            self._batch_ip2 = self._to_device(next(dataloader_iter, None), False)
        batch = self._batch_i
        if self._memcpy_stream is not None:
            # dense features, labels (and ids, for un-pipelined modules) are read by forward AND backward on this
            # stream long after `progress` has dropped its reference to the batch
            batch.record_stream(torch.cuda.current_stream())
        if self._model.training:
            with label("## zero_grad ##"):  # train_pipeline.py:504
                self._optimizer.zero_grad()
        fwd_event = torch.cuda.Event() if self._data_dist_stream is not None else None
        if fwd_event is not None:
            fwd_event.record()
        started = [False]

        def start_next_input_dist() -> None:
            # input_dist of batch i+1 on the side stream, overlapping fwd/bwd of batch i
            if started[0]:
                return
            started[0] = True
            if self._batch_ip1 is not None and self._data_dist_stream is not None:
                with label("## sparse_data_dist ##"):  # train_pipeline.py:528
                    with torch.cuda.stream(self._data_dist_stream):
                        self._data_dist_stream.wait_stream(self._memcpy_stream)
                        self._data_dist_stream.wait_event(fwd_event)
                        self._start_data_dist(self._batch_ip1)

        root = self._model.module if isinstance(self._model, DistributedModelParallel) else self._model
        explicit = self._model.training and hasattr(root, "set_between_forward_and_backward")
        if explicit:
            # a model that runs its own backward inside forward (models/dlrm.py explicit step) calls this between the two
            # (and labels its own "## forward ##" / "## backward ##" ranges then)
            prefetch = self._prefetch if self._prefetch is not None else bool(getattr(root, "wants_lookup_prefetch", False))
            root.set_between_forward_and_backward(start_next_input_dist, self._prefetch_next_lookup if prefetch else None)
        with label("## forward ##"):  # train_pipeline.py:520 (an explicit step nests "## backward ##" inside)
            losses, output = self._model(batch)
        start_next_input_dist()
        backward_done = explicit and root.take_backward_done()
        if self._model.training and backward_done:
            with label("## optimizer ##"):  # train_pipeline.py:550
                if hasattr(root, "finish_dense_grads"):
                    root.finish_dense_grads()
                self._optimizer.step()
        elif self._model.training:
            with label("## backward ##"):  # train_pipeline.py:546
                from ..modules.mlp import _DeferredFinish

                deferred = self._deferred_finish
                if deferred:
                    _DeferredFinish.enable()
                try:
                    self._run_backward(losses)
                finally:
                    if deferred:
                        _DeferredFinish.disable()  # flushes
            with label("## optimizer ##"):
                if hasattr(root, "finish_dense_grads"):
                    root.finish_dense_grads()  # flat-buffer gradient all-reduce of graphed segments (models/dlrm.py)
                self._optimizer.step()
        self._batch_i, self._batch_ip1 = self._batch_ip1, self._batch_ip2
        return output  # as the reference (train_pipeline.py:558): the model's second result, not the losses
