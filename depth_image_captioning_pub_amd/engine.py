"""Fused depth-soft / depth-hard training step on one MI355X, data-parallel across the GPUs of a node.

Counterpart of the inner loop of train_Cdepth_soft / train_Cdepth_hard
(Captioning_models/Depth_caption_model/depth_train.py:168-229, 500-560) and, with `use_depth=False`, of train_base_soft /
train_base_hard (Captioning_models/Base_caption_model/base_train.py:136-175, 363-403: no depth branch, the optimiser
holds the decoder only - BASELINE config 1): same call order, same
train/eval-mode semantics (incl. quirk Q1: the frozen ResNet normalises with batch statistics while
training), same loss, same AdamW update - every tensor operation runs in libdic_hip.so.

Data parallelism (not in the reference): one process per GPU, each rank takes a slice of
the batch; the 29 gradient tensors live in one flat fp32 buffer that is all-reduced (sum of gradients that
were pre-scaled by 1/world_size) with RCCL over xGMI - the decoder bucket is launched as soon as BPTT ends and
overlaps the depth-encoder backward; BatchNorm statistics stay per rank (DDP semantics).  The cross-entropy gradient of
rank r is weighted by its share of the packed tokens (N_r / sum_r N_r), so the all-reduced sum is the gradient of the
global token mean also for variable-length captions; the attention regulariser is a mean over [B, L] and is weighted 1/N.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import os

import torch

from . import native, synthetic as syn
from ._lib import DicError

ALIGN = 64   # floats: every parameter slice starts 256-B aligned so the GEMM loaders can use 16-B vectors


class FlatParams:
    """One flat fp32 buffer holding a list of named tensors as aligned views (+ twin buffers for grads / Adam)."""

    def __init__(self, tensors: Dict[str, torch.Tensor], device):
        self.names: List[str] = list(tensors)
        self.offsets: Dict[str, int] = {}
        off = 0
        for k in self.names:
            self.offsets[k] = off
            off += (tensors[k].numel() + ALIGN - 1) // ALIGN * ALIGN
        self.total = off
        self.data = torch.zeros(off, dtype=torch.float32, device=device)
        self.grad = torch.zeros_like(self.data)
        self.exp_avg = torch.zeros_like(self.data)
        self.exp_avg_sq = torch.zeros_like(self.data)
        self.shapes = {k: tuple(tensors[k].shape) for k in self.names}
        for k in self.names:
            self.view(self.data, k).copy_(tensors[k].to(device))

    def view(self, buf: torch.Tensor, name: str) -> torch.Tensor:
        o = self.offsets[name]
        n = 1
        for s in self.shapes[name]:
            n *= s
        return buf[o:o + n].view(self.shapes[name])

    def views(self, buf: torch.Tensor, names: Optional[Sequence[str]] = None) -> Dict[str, torch.Tensor]:
        return {k: self.view(buf, k) for k in (names or self.names)}

    def span(self, names: Sequence[str]):
        """[start, end) of the contiguous run covering `names` (must be adjacent in the flat buffer)."""
        lo = self.offsets[names[0]]
        last = names[-1]
        n = 1
        for s in self.shapes[last]:
            n *= s
        hi = (self.offsets[last] + n + ALIGN - 1) // ALIGN * ALIGN
        return lo, min(hi, self.total)


def shard_rows(n_rows: int, world: int, rank: int) -> slice:
    """Rows [r*B/N, (r+1)*B/N) of the (length-sorted) global batch belong to rank r."""
    if n_rows % world != 0:
        raise DicError(f"global batch {n_rows} is not divisible by world size {world}")
    per = n_rows // world
    return slice(rank * per, (rank + 1) * per)


def gradient_scales(n_local_tokens: int, world: int, global_tokens: Optional[int] = None):
    """(cross-entropy gradient scale, regulariser gradient scale) of one rank such that the SUM of the ranks' gradients is the
    gradient of the single-device loss on the global batch: the cross-entropy is a mean over the packed tokens, so rank r
    carries its share N_r / sum_r N_r of them (= 1 / world for equal-length batches, the default when `global_tokens` is not
    given); the attention regulariser is a mean over [B, L] with equal rows per rank: 1 / world."""
    ce = 1.0 / world if global_tokens is None else n_local_tokens / float(global_tokens)
    return ce, 1.0 / world


def exchange_gradients(flat_grad: torch.Tensor, spans, group, between=None) -> None:
    """Sum-all-reduce the gradient buckets `spans` = [(lo, hi), ...] of the flat buffer over `group`
    (RCCL over xGMI on the GPUs; gloo in the CPU tests).  The first bucket is launched asynchronously,
    then `between()` runs (the depth-encoder backward, which produces the second bucket and overlaps the
    first collective), then the remaining buckets go out; returns when all have completed on the current
    stream.  Gradients are expected pre-scaled by 1/world_size, so the sum is the DDP average."""
    work = []
    first = True
    for lo, hi in spans:
        if not first and between is not None:
            between()
            between = None
        work.append(torch.distributed.all_reduce(flat_grad[lo:hi], op=torch.distributed.ReduceOp.SUM, group=group,
                                                 async_op=True))
        first = False
    if between is not None:
        between()
    for w in work:
        w.wait()


class _PrefetchSlot:
    """One frozen-ResNet forward in flight: own HIP stream, own runner (shared weights, own workspace, BatchNorm
    running-statistic pointers redirected to `delta`), own feature buffer, static input buffer and captured graphs."""

    def __init__(self, tr: "CaptionTrainer"):
        self.tr = tr
        self.stream = torch.cuda.Stream(device=tr.device)
        self.delta = torch.zeros_like(tr.rn_stats)            # momentum * batch statistic of the forward in flight
        views, o = {}, 0
        for k in tr.rn_stat_keys:
            n = tr.rn_w[k].numel()
            views[k] = self.delta[o:o + n]
            o += n
        self.runner = tr.resnet.shadow(views)
        self.feat: Optional[torch.Tensor] = None
        self.rn_in: Optional[torch.Tensor] = None
        self.graphs = {}                                     # (shape, compact, workspace ptr) -> CUDAGraph

    def _forward(self, x, compact):
        self.delta.zero_()       # the finalize kernels then leave (1 - m) * 0 + m * stat = m * stat here
        self.runner.forward(x, train_bn=True, out=self.feat, compact=compact)

    def launch(self, imgs: torch.Tensor, compact: bool):
        tr = self.tr
        B = imgs.shape[0]
        cells = native.L_COMPACT if compact else native.L_CELLS
        if self.feat is None or tuple(self.feat.shape[:2]) != (B, cells):
            self.feat = torch.empty((B, cells, native.D_ENC), dtype=torch.float32, device=tr.device)
            self.graphs = {}                                 # captured with the old buffer
        ready = torch.cuda.Event()
        ready.record()                                       # inputs + previous readers of this slot's buffers are done
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ready)
            if not tr.use_graph:
                self._forward(imgs, compact)
            else:
                if self.rn_in is None or self.rn_in.shape != imgs.shape:
                    self.rn_in = torch.empty_like(imgs, memory_format=torch.contiguous_format)
                    self.graphs = {}
                self.rn_in.copy_(imgs, non_blocking=True)
                ws = self.runner.workspace
                # a graph holds raw pointers: it is only valid for the workspace it was captured with
                key = (tuple(imgs.shape), compact, ws.data_ptr() if ws is not None else 0)
                g = self.graphs.get(key)
                if g is None:
                    # first use: one eager forward (sizes the workspace, and is this batch's forward), then capture the same
                    # calls for the following batches (capturing records the launches, it does not run them)
                    self._forward(self.rn_in, compact)
                    self.stream.synchronize()
                    key = key[:2] + (self.runner.workspace.data_ptr(),)
                    try:
                        g = torch.cuda.CUDAGraph()
                        # thread_local: other threads (e.g. the RCCL watchdog) may keep issuing their own HIP calls
                        with torch.cuda.graph(g, stream=self.stream, capture_error_mode="thread_local"):
                            self._forward(self.rn_in, compact)
                        self.graphs = {key: g}
                    except Exception as exc:      # capture unsupported here: stay on eager launches (same results)
                        tr.use_graph = False
                        self.graphs = {}
                        tr.last["resnet_graph_error"] = repr(exc)
                else:
                    g.replay()
            done = torch.cuda.Event()
            done.record(self.stream)
        return done


class CaptionTrainer:
    """Owns weights, optimiser state and workspaces of one rank and runs fused train steps."""

    def __init__(self, vocab: int, device: str = "cuda:0", seed: int = 123, lr: float = 1e-3, hard: bool = False,
                 resnet_layers: Sequence[int] = (3, 8, 36, 3), dropout: float = 0.5, lam: float = 0.7,
                 decoder_init: Optional[Dict[str, torch.Tensor]] = None,
                 depth_init: Optional[Dict[str, torch.Tensor]] = None,
                 depth_state: Optional[Dict[str, torch.Tensor]] = None,
                 resnet_init: Optional[Dict[str, torch.Tensor]] = None,
                 process_group=None, conv_mode: Optional[str] = None, use_depth: bool = True):
        """conv_mode: arithmetic of the frozen ResNet-152's convolutions - None = native.DEFAULT_CONV_MODE ("f16x2", the mode
        bench.py measures); "bf16x3" / "fp32" are the exact-operand alternatives (no range limit, ~1.4x / ~2.3x the step time)."""
        if not torch.cuda.is_available():
            raise DicError("CaptionTrainer needs a GPU: the product path has no CPU fallback")
        self.device = torch.device(device)
        self.vocab, self.lr, self.hard, self.p_drop, self.lam = vocab, lr, hard, dropout, lam
        self.pg = process_group
        self.world = torch.distributed.get_world_size(process_group) if process_group is not None else 1
        self.rank = torch.distributed.get_rank(process_group) if process_group is not None else 0
        self.use_depth = use_depth     # False: base-soft / base-hard (base_train.py): no depth encoder, decoder-only optimiser
        dec = decoder_init if decoder_init is not None else syn.decoder_weights(vocab, seed=seed)
        if use_depth and depth_init is None:
            depth_init, depth_state = syn.depth_encoder_weights(seed=seed + 1)
        rn = resnet_init if resnet_init is not None else syn.resnet152_weights(seed=seed + 2, layers=resnet_layers)
        self.dec_names = [k for k, _ in native.DECODER_FIELDS]
        self.enc_names = [k for k, _ in native.DEPTH_FIELDS] if use_depth else []
        merged = {("decoder." + k): dec[k] for k in self.dec_names}
        merged.update({("depth_encoder." + k): depth_init[k] for k in self.enc_names})
        self.flat = FlatParams(merged, self.device)
        self.dec_w = {k: self.flat.view(self.flat.data, "decoder." + k) for k in self.dec_names}
        self.enc_w = {k: self.flat.view(self.flat.data, "depth_encoder." + k) for k in self.enc_names}
        self.dec_g = {k: self.flat.view(self.flat.grad, "decoder." + k) for k in self.dec_names}
        self.enc_g = {k: self.flat.view(self.flat.grad, "depth_encoder." + k) for k in self.enc_names}
        self.dec_span = self.flat.span(["decoder." + k for k in self.dec_names])
        self.enc_span = self.flat.span(["depth_encoder." + k for k in self.enc_names]) if use_depth else None
        self.enc_state = {k: v.to(self.device).contiguous() for k, v in depth_state.items()} if use_depth else {}
        self.rn_w = {k: v.to(self.device).contiguous() for k, v in rn.items()}
        # every BatchNorm running statistic of the frozen ResNet lives in ONE flat buffer (the dict holds views), so that
        # the update of a whole batch is one dic_bn_ema_update launch (see _PrefetchSlot)
        self.rn_stat_keys = [k for k in self.rn_w if k.endswith("running_mean") or k.endswith("running_var")]
        self.rn_stats = torch.cat([self.rn_w[k].reshape(-1) for k in self.rn_stat_keys]).contiguous()
        o = 0
        for k in self.rn_stat_keys:
            n = self.rn_w[k].numel()
            self.rn_w[k] = self.rn_stats[o:o + n]
            o += n
        self.conv_mode = conv_mode or native.DEFAULT_CONV_MODE
        self.resnet = native.ResNetRunner(self.rn_w, resnet_layers, conv_mode=self.conv_mode)
        # f16x2 overflow guard (include/dic.h, dic_resnet_fwd): `guard` holds the status word of the forward whose features the
        # current step consumes.  AdamW and the BatchNorm running-statistic update take it as their skip word, so a step whose
        # features overflowed the fp16 operand planes changes nothing - without the host looking at the word first.  The host
        # learns of it asynchronously: after every update the word is copied into a pinned ring (no synchronisation) and polled at
        # the start of the following steps / in check_status(), which raise DicError.
        self.guard = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._guard_host = torch.zeros(16, dtype=torch.int32).pin_memory()
        self._guard_pending: List[tuple] = []      # (ring slot, event, step number) of copies in flight
        self._guard_slot = 0
        self.step_count = 0
        self.depth_fwd_count = 0       # train-mode depth-encoder forwards (= BatchNorm num_batches_tracked)
        self.rng_offset = 0
        self.seed = seed
        # every rank draws its own dropout masks: the rank is mixed into the Philox key (same key on all ranks would
        # apply one mask pattern to every shard of the global batch)
        self.drop_seed = (seed + 0x9E3779B97F4A7C15 * self.rank) & 0xFFFFFFFFFFFFFFFF
        self.dec_ws: Optional[torch.Tensor] = None
        self.enc_ws: Optional[torch.Tensor] = None
        self.last = {}
        # Software pipelining of the frozen RGB encoder: the features of the NEXT batches are computed on side streams while
        # the rest of the current step runs (prefetch_features).  Up to `prefetch_depth` forwards are in flight, each on its own
        # stream with its own workspace (_PrefetchSlot): two concurrent forwards fill each other's dependent-launch gaps and
        # shallow-grid idle CUs - 25.5 ms for two batch-64 forwards against 29.6 ms back to back (scripts/bench_resnet_concurrent.py).
        self.prefetch_depth = int(os.environ.get("DIC_PREFETCH_DEPTH", "3"))
        self.slots: List[_PrefetchSlot] = []
        self.slot_next = 0
        self.queue: List[tuple] = []   # FIFO of (imgs tensor, slot, done-event) in launch (= batch) order
        self.side_done = None          # completion event of the newest side-stream forward (see _resnet_eager)
        self.prefetch_dropped = 0      # prefetched forwards discarded because the caller left the announced batch order
        self.feat_copy: Optional[torch.Tensor] = None
        # each forward is ~620 launches (9 ms of host enqueue); it is captured once per (batch shape, slot) into a hipGraph
        # and replayed (0.2 ms).  DIC_RESNET_GRAPH=0 keeps eager launches.
        self.use_graph = os.environ.get("DIC_RESNET_GRAPH", "1") != "0"
        # compact 49-cell layout (quirk Q3): at 224x224 both encoders end in a 7x7 map that AdaptiveAvgPool2d(14) only
        # replicates 2x2, so the soft-attention decoder runs on the 49 distinct cells (same logits / alphas / gradients,
        # 4x less feature traffic).  DIC_COMPACT_CELLS=0 keeps the reference's 196-cell evaluation everywhere.
        self.compact_ok = os.environ.get("DIC_COMPACT_CELLS", "1") != "0"
        self.keep_outputs = False      # True: keep logits intact (loss gradient not written in place)
        self.timing = False            # True: record stage-boundary events on the current stream
        self.marks = []

    def _mark(self, name: str) -> None:
        if self.timing:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self.marks.append((name, e))

    def stage_ms(self) -> Dict[str, float]:
        """Elapsed ms between consecutive stage marks of the last timed step (call after a synchronize)."""
        out: Dict[str, float] = {}
        for (_, e0), (n1, e1) in zip(self.marks[:-1], self.marks[1:]):
            out[n1] = out.get(n1, 0.0) + e0.elapsed_time(e1)
        return out

    @property
    def prefetched(self):
        """Oldest pending prefetch as (imgs, features, done-event), or None (kept for tests / callers of round 1)."""
        if not self.queue:
            return None
        imgs, slot, done = self.queue[0]
        return imgs, slot.feat, done

    @prefetched.setter
    def prefetched(self, value):
        if value is not None:
            raise DicError("only `prefetched = None` (drop every pending prefetch) is supported")
        self.queue = []

    @property
    def rn_graphs(self):
        return {(i,) + k: g for i, sl in enumerate(self.slots) for k, g in sl.graphs.items()}

    def prefetch_features(self, imgs: torch.Tensor, compact: bool = False) -> None:
        """Launch the frozen ResNet-152 forward of an upcoming batch on a side stream.  Legal because the RGB encoder takes
        no gradient and is not touched by the optimiser (depth_train.py:136): its output for batch i+k does not depend on
        the updates of steps i..i+k-1.  Its BatchNorm running statistics (quirk Q1) are still updated once per batch and
        in batch order: the forward leaves momentum * (batch statistic) in the slot's scratch buffers, and train_step applies
        it (dic_bn_ema_update) when it consumes the features - consumption order is batch order."""
        if len(self.queue) >= self.prefetch_depth:
            raise DicError(f"{len(self.queue)} prefetched batches are pending (prefetch_depth = {self.prefetch_depth}): "
                           "consume one with train_step before prefetching more")
        while len(self.slots) < self.prefetch_depth:
            self.slots.append(_PrefetchSlot(self))
        busy = {id(sl) for _, sl, _ in self.queue}
        slot = next(sl for sl in self.slots[self.slot_next:] + self.slots[:self.slot_next] if id(sl) not in busy)
        self.slot_next = (self.slots.index(slot) + 1) % len(self.slots)
        compact = compact and tuple(imgs.shape[-2:]) == (224, 224)
        done = slot.launch(imgs, compact)
        self.queue.append((imgs, slot, done))
        self.side_done = done

    def _take_prefetched(self, imgs: torch.Tensor):
        """Features of `imgs` if its forward is the oldest one in flight: waits for it on the current stream and applies
        its BatchNorm running-statistic update."""
        if not self.queue:
            return None
        if self.queue[0][0] is not imgs:
            # The caller left the announced order (another tensor object, a skipped or re-ordered batch, an exception mid-epoch).
            # Forwards launched for batches that are not consumed now are DISCARDED - their BatchNorm deltas are never applied,
            # exactly as if those forwards had not run (the reference only runs the encoder on batches it trains on) - so the
            # running statistics stay in consumption order and the slots are free again for the following prefetches.
            pos = next((i for i, (q, _, _) in enumerate(self.queue) if q is imgs), len(self.queue))
            for _, _, done in self.queue[:pos]:
                torch.cuda.current_stream().wait_event(done)      # (its buffers must be idle before the slot is reused)
            self.prefetch_dropped += pos
            import warnings
            warnings.warn(f"CaptionTrainer: {pos} prefetched ResNet forward(s) discarded - train_step received a batch that "
                          "was not the next announced one (next_imgs)", RuntimeWarning, stacklevel=3)
            self.queue = self.queue[pos:]
            if not self.queue:
                return None
        _, slot, done = self.queue.pop(0)
        torch.cuda.current_stream().wait_event(done)
        self._guard_take(slot.runner)
        native.bn_ema_update(self.rn_stats, slot.delta, 0.1, skip_if_raised=self.guard)      # (a flagged forward's statistics are dropped)
        self.resnet.train_forwards += 1
        # the slot is free for the next prefetch from here on (train_step launches it before this step has read the
        # features), so the step works on its own copy: one 26-MB device copy (~10 us), ordered on this stream before the
        # `ready` event the slot's next forward waits for
        if self.feat_copy is None or self.feat_copy.shape != slot.feat.shape:
            self.feat_copy = torch.empty_like(slot.feat)
        self.feat_copy.copy_(slot.feat, non_blocking=True)
        return self.feat_copy

    def _resnet_eager(self, imgs: torch.Tensor, train_bn: bool, compact: bool) -> torch.Tensor:
        """ResNet forward on the CURRENT stream with the trainer's own runner (own workspace; BatchNorm running statistics
        updated / read in place).  Ordered after the newest side-stream forward so that a validation pass or a step on an
        un-prefetched batch never competes with a prefetch for the chip's memory system mid-kernel-chain; the pending
        prefetches' running-statistic updates are applied later, when their batches are consumed."""
        if self.side_done is not None:
            torch.cuda.current_stream().wait_event(self.side_done)
        return self.resnet.forward(imgs, train_bn=train_bn, compact=compact)

    # ---- f16x2 overflow guard -------------------------------------------------------------------
    def _guard_take(self, runner: Optional[native.ResNetRunner]) -> None:
        """`guard` <- the status word of the forward that produced this step's features (None / exact modes: clear)."""
        if runner is not None and runner.mode == 2 and runner.workspace is not None:
            self.guard.copy_(runner.status_word(), non_blocking=True)
        else:
            self.guard.zero_()

    def _guard_publish(self) -> None:
        """Copy the word to the host ring (asynchronous) after the update of this step has been enqueued."""
        if self.conv_mode != "f16x2":
            return
        if os.environ.get("DIC_DEBUG_GUARD"):
            torch.cuda.synchronize()
            print(f"[guard rank {self.rank}] step {self.step_count}: word {int(self.guard.item()) & 0xffffffff:#x}; slot words "
                  f"{[hex(int(sl.runner.status_word().item()) & 0xffffffff) for sl in self.slots if sl.runner.workspace is not None]}; own "
                  f"{hex(int(self.resnet.status_word().item()) & 0xffffffff) if self.resnet.workspace is not None else None}; ws "
                  f"{[(sl.runner.workspace.data_ptr(), sl.runner.workspace.numel(), list(sl.graphs)) for sl in self.slots if sl.runner.workspace is not None]}", flush=True)
        if len(self._guard_pending) >= self._guard_host.numel():      # ring full (16 steps un-polled): settle the oldest first
            self._guard_poll(block=True)
        i = self._guard_slot
        self._guard_slot = (i + 1) % self._guard_host.numel()
        self._guard_host[i:i + 1].copy_(self.guard, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._guard_pending.append((i, ev, self.step_count))

    def _guard_poll(self, block: bool = False) -> None:
        bad = []
        while self._guard_pending and (block or self._guard_pending[0][1].query()):
            i, ev, step = self._guard_pending.pop(0)
            ev.synchronize()
            if int(self._guard_host[i]) != 0:
                bad.append((step, int(self._guard_host[i])))
        if bad:
            self.step_count -= len(bad)        # the device skipped those updates: Adam's bias correction must not count them
            raise DicError(f"f16x2 overflow guard: the ResNet-152 forward of optimiser step(s) {[b[0] for b in bad]} (guard words "
                           f"{[hex(b[1]) for b in bad]}: bits in csrc/common.h) produced an activation beyond the "
                           "fp16 range of the operand planes (|x| > 16376) or a non-finite value; its features were NaN and AdamW was "
                           "skipped on the device for those steps, so parameters, Adam moments and BatchNorm running statistics are "
                           "as they were before them.  Re-create the trainer with conv_mode='bf16x3' (exact operands, no range limit)")

    def check_status(self) -> None:
        """Synchronise with the device and raise DicError if any step since the last check tripped the f16x2 overflow guard.
        Call it wherever the loss is read on the host (the reference reads it every iteration, depth_train.py:224)."""
        torch.cuda.current_stream().synchronize()
        self._guard_poll(block=True)

    # ---- pieces -----------------------------------------------------------------------------
    def _compact(self, imgs, depth_map) -> bool:
        """Use the 49-cell layout for this batch? (soft attention, 224x224 RGB and depth inputs)"""
        if self.use_depth and (depth_map is None or tuple(depth_map.shape[-2:]) != (224, 224)):
            return False
        return self.compact_ok and not self.hard and (imgs is None or tuple(imgs.shape[-2:]) == (224, 224))

    def encode(self, imgs: torch.Tensor, depth_map: Optional[torch.Tensor], train: bool):
        compact = self._compact(imgs, depth_map)
        feats = self._resnet_eager(imgs, train, compact)                                    # depth_train.py:179
        if not self.use_depth:
            return feats, None, None
        fdep, dtape = native.depth_encoder_forward(self.enc_w, self.enc_state, depth_map.detach(), train,
                                                   workspace=self.enc_ws, compact=compact)   # :204-206
        self.enc_ws = dtape.workspace
        self.depth_fwd_count += int(train)
        return feats, fdep, dtape

    def train_step(self, imgs: torch.Tensor, depth_map: Optional[torch.Tensor], captions: torch.Tensor, lengths: Sequence[int],
                   drop_mult: Optional[torch.Tensor] = None, gumbel_u: Optional[torch.Tensor] = None,
                   temp: float = 1.0, precomputed_features: Optional[torch.Tensor] = None,
                   next_imgs=None, global_tokens: Optional[int] = None,
                   apply_update: bool = True, virtual_world: Optional[int] = None) -> torch.Tensor:
        """One iteration of depth_train.py:168-221. Returns the loss as a 1-element device tensor (no host sync).
        next_imgs: images of the following batch, or the list [batch i+1, batch i+2, ...] of the next batches in order; their
          (frozen) ResNet forwards run ahead on side streams, up to `prefetch_depth` (3) at a time.  A prefetched forward is
          matched to a later train_step by OBJECT IDENTITY of the images tensor: pass the very tensor objects announced here
          (not a re-wrapped copy, slice or .to() result of them), in the announced order - otherwise the pending forwards ahead
          of the batch are discarded (RuntimeWarning, counted in `prefetch_dropped`) and the step runs its own forward.
        global_tokens: packed tokens (sum of lengths-1) of the GLOBAL batch when data parallel with variable-length
          captions; default = this rank's count x world size (exact for equal-length batches such as bench.py's).
        apply_update=False leaves the (scaled, all-reduced) gradients in self.flat.grad and skips AdamW;
        virtual_world=N scales the gradients as rank-of-N would without any collective - together they let one process
          reproduce an N-rank step shard by shard (tests)."""
        B = imgs.shape[0] if imgs is not None else precomputed_features.shape[0]
        tmax = max(lengths) - 1
        self._guard_poll()              # (non-blocking) an earlier step tripped the f16x2 overflow guard -> DicError
        self.marks = []
        self._mark("start")
        if precomputed_features is None:
            compact = self._compact(imgs, depth_map)
            feats = self._take_prefetched(imgs)
            if feats is not None:
                compact = feats.shape[1] == native.L_COMPACT
            else:
                feats = self._resnet_eager(imgs, True, compact)                             # depth_train.py:179
                self._guard_take(self.resnet)
            if next_imgs is not None:       # one upcoming batch, or the list of the next `prefetch_depth` batches in order
                upcoming = list(next_imgs) if isinstance(next_imgs, (list, tuple)) else [next_imgs]
                for k, nxt in enumerate(upcoming[:self.prefetch_depth]):
                    if k >= len(self.queue):        # (entry k of the queue is batch i+1+k when the caller keeps this order)
                        self.prefetch_features(nxt, compact=self._compact(nxt, depth_map))
            self._mark("resnet152_fwd")
        else:                                       # decoder/depth-encoder-only step (tests)
            feats = precomputed_features
            self._guard_take(None)
            compact = feats.shape[1] == native.L_COMPACT
        fdep = dtape = None
        if self.use_depth:
            fdep, dtape = native.depth_encoder_forward(self.enc_w, self.enc_state, depth_map.detach(), True,
                                                       workspace=self.enc_ws, compact=compact)   # :204-206
            self.enc_ws = dtape.workspace
            self.depth_fwd_count += 1
            self.guard.bitwise_or_(native.depth_status_word(dtape))      # the depth encoder's own guard word joins the step's (one 4-byte op)
        self._mark("depth_encoder_fwd")
        if drop_mult is None and self.p_drop > 0:
            drop_mult = native.dropout_mask((B, tmax, native.D_HID), self.p_drop, self.drop_seed, self.rng_offset,
                                            self.device)
            self.rng_offset += B * tmax * native.D_HID // 4 + 1
        mode = 1 if self.hard else 0
        logits, alphas, tape = native.decoder_forward(self.dec_w, feats, fdep, captions, lengths, drop_mult, mode=mode,
                                                      gumbel_u=gumbel_u, temp=temp, workspace=self.dec_ws)
        self.dec_ws = tape.workspace
        self._mark("decoder_fwd")
        targets = native.pack_targets(captions, lengths)                                     # :210-213
        nworld = virtual_world if virtual_world is not None else self.world
        ce_scale, reg_scale = gradient_scales(int(targets.shape[0]), nworld, global_tokens)
        loss, dlogits, dalphas = native.caption_loss(logits, targets, None if self.hard else alphas, self.lam,
                                                     grad_scale=ce_scale, reg_grad_scale=reg_scale,
                                                     in_place=not self.keep_outputs)         # :214-216
        self._mark("loss")
        _, dfeat = native.decoder_backward(tape, dlogits, dalphas, grads=self.dec_g)         # :219
        self._mark("decoder_bwd")
        if not self.use_depth:       # base-soft / base-hard: the decoder bucket is everything there is (base_train.py:115)
            if self.world > 1:
                exchange_gradients(self.flat.grad, [self.dec_span], self.pg)
        elif self.world > 1:   # decoder bucket goes out while the depth-encoder backward still runs
            exchange_gradients(self.flat.grad, [self.dec_span, self.enc_span], self.pg,
                               between=lambda: native.depth_encoder_backward(dtape, dfeat, grads=self.enc_g))
        else:
            native.depth_encoder_backward(dtape, dfeat, grads=self.enc_g)
        self._mark("depth_encoder_bwd+allreduce")
        if apply_update:
            self.apply_update()
        self._mark("adamw")
        self.last = {"logits": logits, "alphas": alphas, "features": feats, "depth_features": fdep}
        if self.keep_outputs:
            self.last["decoder_tape"] = tape           # parity tests: native.decoder_attention_relu_mask(tape)
        return loss

    def apply_update(self) -> None:
        """AdamW on the flat buffer with whatever self.flat.grad holds (depth_train.py:221)."""
        self.step_count += 1
        native.adamw_step(self.flat.data, self.flat.grad, self.flat.exp_avg, self.flat.exp_avg_sq, self.step_count,
                          lr=self.lr, skip_if_raised=self.guard)
        self._guard_publish()

    @torch.no_grad()
    def eval_loss(self, imgs, depth_map, captions, lengths, gumbel_u: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Validation forward: eval-mode BN in both encoders, dropout off.
        soft (depth_train.py:248-292): CE + attention regulariser;
        hard (depth_train.py:555-610): decoder.eval_forward = Gumbel-max one-hot attention (mode 2, `gumbel_u`
        [Tmax,B,196] uniform draws; drawn from torch's CPU generator like attention.py:40 when omitted), CE only."""
        feats, fdep, _ = self.encode(imgs, depth_map, train=False)
        if self.hard:
            if gumbel_u is None:
                gumbel_u = torch.rand(max(lengths) - 1, len(lengths), native.L_CELLS).clamp_(1e-6, 1 - 1e-6).to(self.device)
            logits, alphas, tape = native.decoder_forward(self.dec_w, feats, fdep, captions, lengths, None, mode=2,
                                                          gumbel_u=gumbel_u, workspace=self.dec_ws)
        else:
            logits, alphas, tape = native.decoder_forward(self.dec_w, feats, fdep, captions, lengths, None,
                                                          workspace=self.dec_ws)
        self.dec_ws = tape.workspace
        loss, _, _ = native.caption_loss(logits, native.pack_targets(captions, lengths),
                                         None if self.hard else alphas, self.lam)
        self.last = {"logits": logits, "alphas": alphas, "features": feats, "depth_features": fdep}
        return loss

    def state_dicts(self):
        """state_dict contents of the three reference modules, loadable with strict=True (keys pinned by
        tests/golden/state_dict_keys.json): Depth_CNN_endoder registers every layer twice (`conv1.*` and
        `features.0.*`, depth_models.py:19-34 - 42 keys), BatchNorm layers carry `num_batches_tracked` (one increment per
        train-mode forward: the depth encoder's = optimiser steps taken, the frozen ResNet's = train-mode forwards, Q1)."""
        dev = self.device
        dec = {k: v.detach().clone() for k, v in self.dec_w.items()}
        rgb = {k: v.detach().clone() for k, v in self.rn_w.items()}
        for k in list(rgb):
            if k.endswith("running_mean"):
                rgb[k[:-len("running_mean")] + "num_batches_tracked"] = torch.tensor(self.resnet.train_forwards,
                                                                                   dtype=torch.int64, device=dev)
        if not self.use_depth:
            return {"decoder": dec, "encoder": rgb}
        enc = {k: v.detach().clone() for k, v in self.enc_w.items()}
        enc.update({k: v.detach().clone() for k, v in self.enc_state.items()})
        nbt = torch.tensor(self.depth_fwd_count, dtype=torch.int64, device=dev)
        for i, (conv_idx, bn_idx) in zip((1, 2, 3), ((0, 1), (4, 5), (8, 9))):
            enc[f"bn{i}.num_batches_tracked"] = nbt.clone()
            for kind in ("weight", "bias"):
                enc[f"features.{conv_idx}.{kind}"] = enc[f"conv{i}.{kind}"].clone()
                enc[f"features.{bn_idx}.{kind}"] = enc[f"bn{i}.{kind}"].clone()
            for kind in ("running_mean", "running_var", "num_batches_tracked"):
                enc[f"features.{bn_idx}.{kind}"] = enc[f"bn{i}.{kind}"].clone()
        return {"decoder": dec, "depth_encoder": enc, "encoder": rgb}
