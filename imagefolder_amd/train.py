"""Tokenizer train step (T1) and data-parallel gradient exchange (D1) for MI355X nodes.

Mirrors the hot loop of reference tokenizer/tokenizer_image/xqgan_train.py:439-478 (generator step, optimizer,
EMA, optional discriminator step) with an MI355X-first data path instead of torch DDP + per-tensor optimizers:

  * every trainable parameter, its gradient, both AdamW moments and the EMA copy live in five flat fp32 arenas
    (FlatArena): one xq_adamw_ema_step launch (HBM-bound, 36-40 B/param) replaces AdamW.step() + update_ema() +
    zero_grad() (xqgan_train.py:447,459-462; utils/ema.py:5-14);
  * data parallel = one process per GPU; the gradient arena is all-reduced by RCCL over xGMI
    (torch.distributed backend "nccl") in a few large chunks, asynchronously on RCCL's own stream, while the
    discriminator step runs on the compute stream — that step only needs recons.detach() (xqgan_train.py:465-470),
    so the generator's optimizer step is deferred past it.  The 1/world of DDP's mean is folded
    into the optimizer kernel.  No DDP wrapper: no per-step buffer broadcast (C9), no wasted discriminator-head
    reduction during the generator backward (C2);
  * on CPU (gloo) the same class runs with a plain tensor AdamW so the N > 1 logic is testable without GPUs.
"""
import ctypes
import os
from typing import Callable, Iterable, List, Optional

import torch
import torch.distributed as dist

from . import _lib
from ._lib import check, ptr
from ._lib import marker as _marker

# round 6: [in][out] bf16 copies of the Linear weights for the data gradients (FlatArena._init_transposed_shadows); 0 = the NN products on W as stored
TRANSPOSED_SHADOWS = os.environ.get("XQ_DGRAD_NT", "1") == "1"
# round 6: the packed 3x3 conv weights of an arena refreshed by one launch behind the optimizer step; 0 = repacked per weight on the next use
CONV_PACKS_BATCHED = os.environ.get("XQ_CONV_PACKS_BATCHED", "1") == "1"


def get_random_ratio(randomness_anneal_start, randomness_anneal_end, end_ratio, cur_step):
    """xqgan_train.py:62-68 (perturbation schedule)"""
    if cur_step < randomness_anneal_start:
        return 1.0
    elif cur_step > randomness_anneal_end:
        return end_ratio
    return 1.0 - (cur_step - randomness_anneal_start) / (randomness_anneal_end - randomness_anneal_start) * end_ratio


class FlatArena:
    """Re-homes `params` (and their grads / AdamW state / EMA) into flat fp32 buffers; tensors keep their identity
    (`p.data` becomes a view), so modules, state_dict() and checkpoints are unaffected."""

    def __init__(self, params: Iterable[torch.nn.Parameter], with_ema: bool = True):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        assert self.params, "no trainable parameters"
        dev = self.params[0].device
        assert all(p.device == dev and p.dtype == torch.float32 for p in self.params), "fp32 master params on one device"
        self.offsets, n = [], 0
        for p in self.params:
            self.offsets.append(n)
            n += (p.numel() + 3) // 4 * 4  # keep every tensor 16-byte aligned inside the arena
        self.numel = n
        self.p = torch.zeros(n, dtype=torch.float32, device=dev)
        self.g = torch.zeros(n, dtype=torch.float32, device=dev)
        self.m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.v = torch.zeros(n, dtype=torch.float32, device=dev)
        self.ema = torch.zeros(n, dtype=torch.float32, device=dev) if with_ema else None
        # bf16 shadow of the masters, refreshed by the optimizer kernel; the dense ops read it through p._xq_w16
        self.p16 = torch.zeros(n, dtype=torch.bfloat16, device=dev) if dev.type == "cuda" else None
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                self.p[o:o + p.numel()].copy_(p.data.reshape(-1))
                p.data = self.p[o:o + p.numel()].view(p.shape)
                p.grad = self.g[o:o + p.numel()].view(p.shape)
            if with_ema:
                self.ema.copy_(self.p)  # ema = deepcopy(model) (xqgan_train.py:316)
            if self.p16 is not None:
                self.p16.copy_(self.p)
                for p, o in zip(self.params, self.offsets):
                    p._xq_w16 = self.p16[o:o + p.numel()].view(p.shape)
        self._init_transposed_shadows()
        # packed bf16 layouts of the 3x3 conv weights (ops_dense._packed_conv_weight registers them on first use): refreshed in one launch
        # behind the optimizer step instead of one launch per weight and layout on the next forward / backward
        self._conv_packs, self._conv_table, self._conv_blocks = {}, None, 0
        self.step_count = 0
        # `epoch` counts the updates that reach the masters through raw pointers (the optimizer kernel, resync): those never
        # bump torch's per-tensor version counter, so every cache derived from a parameter (packed conv weights, ...) keys
        # on (p._version, p._xq_arena.epoch).  `_xq_w16_version` = p._version at the last refresh of the bf16 shadow:
        # torch in-place updates of a parameter (load_state_dict, manual init) DO bump p._version and are picked up by
        # ops_dense._w16 on the next use.
        self.epoch = 0
        for p in self.params:
            p._xq_arena = self
            p._xq_w16_version = p._version

    def _init_transposed_shadows(self):
        """[in][out] bf16 copies of the Linear-shaped weights (2-D, both widths multiples of 64), next to the [out][in] shadow: the data
        gradients g_x = g_y W read them as the K-major operand of an NT product, 7-10 % faster than the transpose reads of the NN product
        on W as stored (profiles/r06_nn_vs_nt_transposed_weight.txt; ops_dense._w16t).  One xq_transpose_bf16_batched launch after every
        optimizer step keeps them current (refresh_transposed_shadows)."""
        self.p16t, self._t_table, self._t_count, self._t_tiles = None, None, 0, 0
        if self.p16 is None or not TRANSPOSED_SHADOWS:
            return
        rows, dst, tiles = [], 0, 0
        for p, o in zip(self.params, self.offsets):
            if p.dim() == 2 and p.shape[0] % 64 == 0 and p.shape[1] % 64 == 0:
                rows.append((o, dst, p.shape[0], p.shape[1], tiles, p))
                dst += p.numel()
                tiles += (p.shape[0] // 64) * (p.shape[1] // 64)
        if not rows:
            return
        self.p16t = torch.empty(dst, dtype=torch.bfloat16, device=self.p16.device)
        self._t_table = torch.tensor([r[:5] for r in rows], dtype=torch.int64).to(self.p16.device)
        self._t_count, self._t_tiles = len(rows), tiles
        for o, d, r, c, _, p in rows:
            p._xq_w16t = self.p16t[d:d + r * c].view(c, r)
        self.refresh_transposed_shadows()

    def refresh_transposed_shadows(self):
        """p16t = the transposes of the bf16 shadow's Linear weights — after anything that rewrote the shadow (one launch, ~4 B per element)"""
        if self.p16t is None:
            return
        with torch.cuda.device(self.p16.device):
            stream = ctypes.c_void_p(torch.cuda.current_stream(self.p16.device).cuda_stream)
            rc = _lib.lib().xq_transpose_bf16_batched(ptr(self.p16), ptr(self.p16t), ptr(self._t_table), self._t_count, self._t_tiles, stream)
        check(rc, "xq_transpose_bf16_batched")

    def register_conv_pack(self, p, for_data_grad: bool, wp):
        """ops_dense._packed_conv_weight packed `p` on its own (first use, or after a resync): from now on the optimizer step refreshes `wp`"""
        if not CONV_PACKS_BATCHED or p.dim() != 4 or tuple(p.shape[2:]) != (3, 3):
            return
        e = self._conv_packs.setdefault(id(p), [p, None, None])
        if e[1 + int(for_data_grad)] is not wp:
            e[1 + int(for_data_grad)] = wp
            self._conv_table = None

    def conv_pack_buffer(self, p, for_data_grad: bool):
        """the registered pack buffer of `p` (reused in place by a lazy repack), or None"""
        e = self._conv_packs.get(id(p))
        return None if e is None else e[1 + int(for_data_grad)]

    def refresh_conv_packs(self):
        """repack every registered 3x3 weight from the fp32 masters (one xq_conv3x3_pack_weights_batched launch) and stamp the caches current"""
        if not self._conv_packs:
            return
        dev = self.p.device
        if self._conv_table is None:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("FlatArena.refresh_conv_packs: a conv weight was packed for the first time inside a stream capture; run one eager "
                                   "step first (CapturedStep's warm-up does) so that the pack table is built outside the capture")
            rows, blocks = [], 0
            for p, wf, wd in self._conv_packs.values():
                rows.append((p.data_ptr(), 0 if wf is None else wf.data_ptr(), 0 if wd is None else wd.data_ptr(), p.shape[0], p.shape[1], blocks))
                blocks += (p.numel() + 255) // 256
            self._conv_table = torch.tensor(rows, dtype=torch.int64).to(dev)
            self._conv_blocks = blocks
        with torch.cuda.device(dev):
            stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            rc = _lib.lib().xq_conv3x3_pack_weights_batched(ptr(self._conv_table), len(self._conv_packs), self._conv_blocks, stream)
        check(rc, "xq_conv3x3_pack_weights_batched")
        for p, wf, wd in self._conv_packs.values():
            stamp = (p._version, self.epoch)
            if wf is not None:
                p._xq_pack_fwd = (stamp, wf)
            if wd is not None:
                p._xq_pack_dgrad = (stamp, wd)

    @torch.no_grad()
    def resync(self, ema: bool = False):
        """Call after the masters were overwritten (checkpoint load, weight surgery): refreshes the bf16 shadow, invalidates
        the derived caches and — `ema=True`, the reference's update_ema(ema, model, decay=0) after loading
        (xqgan_train.py:384) — re-seeds the EMA copy from the masters."""
        if self.p16 is not None:
            self.p16.copy_(self.p)
            self.refresh_transposed_shadows()
        if ema and self.ema is not None:
            self.ema.copy_(self.p)
        self.epoch += 1
        for p in self.params:
            p._xq_w16_version = p._version

    def rebind_grads(self):
        """autograd keeps accumulating into the arena as long as p.grad stays the view; call after anything that
        may have replaced .grad (e.g. zero_grad(set_to_none=True))."""
        for p, o in zip(self.params, self.offsets):
            if p.grad is None or p.grad.data_ptr() != self.g.data_ptr() + 4 * o:
                p.grad = self.g[o:o + p.numel()].view(p.shape)

    def release_grads(self):
        """Before a backward pass: p.grad = None for every parameter.  autograd's AccumulateGrad then ADOPTS the gradient tensor
        each backward node returns instead of adding it into an existing .grad — one `add_` launch per parameter per step
        (~500 for the ViT-B tokenizer, 2.2 ms of GPU time: profiles/r02_glue_vq8192.txt) becomes the few multi-tensor copies of
        collect()."""
        for p in self.params:
            p.grad = None

    @torch.no_grad()
    def collect(self, lo: int = 0, hi: Optional[int] = None):
        """After (part of) a backward pass: move the gradients of params[lo:hi] that live outside the arena into their arena slots
        (torch._foreach_copy_: multi-tensor kernels) and re-point p.grad at the slots.  The slots are zero beforehand (the
        optimizer kernel clears them), so parameters that received no gradient contribute zeros, as with accumulation."""
        hi = len(self.params) if hi is None else hi
        base = self.g.data_ptr()
        dst, src = [], []
        for i in range(lo, hi):
            p, o = self.params[i], self.offsets[i]
            g = p.grad
            view = self.g[o:o + p.numel()].view(p.shape)
            if g is not None and g.data_ptr() != base + 4 * o:
                dst.append(view)
                src.append(g if g.dtype == torch.float32 else g.float())
            p.grad = view
        if dst:
            torch._foreach_copy_(dst, src)

    def ema_state_dict(self, names: List[str]):
        return {n: self.ema[o:o + p.numel()].view(p.shape) for n, p, o in zip(names, self.params, self.offsets)}


class GradAllReducer:
    """Chunked asynchronous all-reduce (SUM) of a flat gradient arena over RCCL (xGMI).

    Chunks are runs of whole parameters (>= chunk_bytes each).  With `params` given, every parameter carries a
    post-accumulate-grad hook: the moment the backward pass has deposited the last gradient of a chunk, that chunk's
    all-reduce is enqueued (RCCL's stream waits for the compute stream at that point only), so the transfers of the
    decoder's gradients run under the encoder's backward — the job DDP's buckets do in the reference
    (xqgan_train.py:412,455) without its per-bucket copies.  `start()` (after backward) enqueues whatever is left (parameters
    that received no gradient this step), `wait()` makes the compute stream wait for all of them and records how long it
    had to (`exposed_ms`, HIP events).  Anything enqueued between start() and wait() — the discriminator step — overlaps.
    comm_dtype=torch.bfloat16 halves the bytes on the links (cast -> all-reduce -> cast back into the fp32 arena); default
    is the arena's fp32, which is what DDP reduces.  always=True runs the collectives at world size 1 too (tests).

    Launch ORDER (round 5).  Collectives of one communicator must be issued in the same sequence on every rank; the order in which hooks
    complete chunks is a property of each rank's autograd graph (node sequence numbers), which nothing forces to be equal across ranks.
    So chunks leave in ONE fixed order, identical on all ranks by construction: a completed chunk is enqueued only once every chunk ahead
    of it in that order has been (DDP's rule for its buckets).  The order: rank 0's completion order observed during the FIRST armed
    backward pass (that pass launches nothing from the hooks; its chunks go out in start()), broadcast once — the true backward order of
    this model, so later steps lose no overlap to the rule; until then, and for chunks no hook ever completes, descending arena order
    (the backward pass walks the model from its end)."""

    def __init__(self, flat_grad: torch.Tensor, group=None, chunk_bytes: int = 64 << 20, params=None, offsets=None,
                 comm_dtype: Optional[torch.dtype] = None, always: bool = False, collect_fn: Optional[Callable] = None):
        self.g = flat_grad
        self.collect_fn = collect_fn          # collect_fn(lo, hi): bring the gradients of params[lo:hi] into the flat buffer
        self.group = group
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        self.active = self.world > 1 or (always and dist.is_available() and dist.is_initialized())
        self.comm_dtype = comm_dtype if comm_dtype not in (None, flat_grad.dtype) else None
        per = max(1, chunk_bytes // 4)
        n = flat_grad.numel()
        self.chunks, self._chunk_of_param, self._param_range = [], [], []
        if params is not None and offsets is not None and len(params):
            start, cur, first = 0, 0, 0
            for i, (p, o) in enumerate(zip(params, offsets)):
                end = offsets[i + 1] if i + 1 < len(params) else n
                self._chunk_of_param.append(cur)
                if end - start >= per or i + 1 == len(params):
                    self.chunks.append((start, end))
                    self._param_range.append((first, i + 1))
                    start, cur, first = end, cur + 1, i + 1
        else:
            self.chunks = [(s, min(s + per, n)) for s in range(0, n, per)]
        self._need = [0] * len(self.chunks)
        for c in self._chunk_of_param:
            self._need[c] += 1
        self._left = list(self._need)
        self._launched = [False] * len(self.chunks)
        self._ready = [False] * len(self.chunks)
        self._order = list(range(len(self.chunks) - 1, -1, -1))      # fixed launch order (see the class docstring)
        self._pos = 0
        self._learned = not (params is not None and len(self.chunks) > 1)   # nothing to learn without hooks
        self._seen = []                                               # completion order of the learning pass
        self._works = []
        self._armed = False
        self.exposed_ms = []          # one entry per wait(): time the compute stream was blocked by the collectives
        self._pending_ev = []         # (start, end) event pairs of wait()s not yet read
        if self.active and params is not None:
            for i, p in enumerate(params):
                p.register_post_accumulate_grad_hook(self._make_hook(self._chunk_of_param[i]))

    def _make_hook(self, ci):
        def hook(_p):
            if not self._armed:
                return
            self._left[ci] -= 1
            if self._left[ci] == 0 and not self._launched[ci] and not self._ready[ci]:
                self._ready[ci] = True
                if not self._learned:
                    self._seen.append(ci)      # learning pass: record, launch nothing (start() sends everything in the default order)
                else:
                    self._drain()
        return hook

    def _drain(self):
        """enqueue, in the fixed order, every chunk that is complete and has no incomplete chunk ahead of it"""
        while self._pos < len(self._order) and self._ready[self._order[self._pos]]:
            ci = self._order[self._pos]
            self._pos += 1
            if not self._launched[ci]:
                self._launch(ci)

    def _adopt_learned_order(self):
        """after the learning pass: rank 0's completion order (chunks no hook completed appended in descending arena order) becomes
        everyone's launch order — one small broadcast, once per reducer"""
        order = list(self._seen) + [c for c in range(len(self.chunks) - 1, -1, -1) if c not in set(self._seen)]
        t = torch.tensor(order, dtype=torch.int64, device=self.g.device)
        if self.world > 1:
            dist.broadcast(t, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
        got = [int(v) for v in t.tolist()]
        assert sorted(got) == list(range(len(self.chunks))), got
        self._order, self._learned, self._seen = got, True, []

    def _launch(self, ci):
        s, e = self.chunks[ci]
        self._launched[ci] = True
        if self.collect_fn is not None and self._param_range:
            self.collect_fn(*self._param_range[ci])
        if self.comm_dtype is not None:
            buf = self.g[s:e].to(self.comm_dtype)
            w = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self._works.append((w, s, e, buf))
        else:
            w = dist.all_reduce(self.g[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self._works.append((w, s, e, None))

    def arm(self):
        """before the backward pass whose gradients are to be reduced"""
        if not self.active:
            return
        assert not self._works, "previous all-reduce not waited for"
        self._left = list(self._need)
        self._launched = [False] * len(self.chunks)
        self._ready = [False] * len(self.chunks)
        self._pos = 0
        self._armed = True

    def start(self):
        if not self.active:
            return
        was_armed, self._armed = self._armed, False
        for ci in self._order:                 # whatever the hooks have not sent, in the fixed order
            if not self._launched[ci]:
                self._launch(ci)
        self._pos = len(self._order)
        if not self._learned and was_armed:
            self._adopt_pending = True      # exchanged at the END of wait(): a blocking broadcast here would stall the host behind every gradient
                                            # collective just enqueued (the learning step would lose all overlap, and a rank that raised between
                                            # its all-reduces and the broadcast would leave the others hanging in it)

    def wait(self):
        if not self._works:
            return
        timed = self.g.is_cuda
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        for w, s, e, buf in self._works:
            w.wait()
            if buf is not None:
                self.g[s:e].copy_(buf)
        if timed:
            e1.record()
            self._pending_ev.append((e0, e1))
        self._works = []
        self._launched = [False] * len(self.chunks)
        if getattr(self, "_adopt_pending", False):
            self._adopt_pending = False
            self._adopt_learned_order()      # (every rank reaches this point: the step's collectives have completed on all of them)

    def collect_exposed_ms(self):
        """durations of the wait()s whose events have completed (never blocks: an event pair still in flight stays pending and is picked
        up by a later call; after a device synchronisation every pair is complete).  Round 4: the one-pair form read elapsed_time of
        the step just enqueued — "Both events must be completed" whenever the host ran ahead of the device, i.e. on every fast box."""
        still = []
        for e0, e1 in self._pending_ev:
            if e1.query():
                self.exposed_ms.append(e0.elapsed_time(e1))
            else:
                still.append((e0, e1))
        self._pending_ev = still
        return self.exposed_ms


class ArenaOptimizer:
    """AdamW (+ optional EMA) over a FlatArena with an optional data-parallel gradient all-reduce: the reference's
    (optimizer, ema, DDP reducer) trio for one parameter set."""

    def __init__(self, params, lr=1e-4, betas=(0.9, 0.95), weight_decay=5e-2, eps=1e-8, ema_decay=0.9999, use_ema=True,
                 group=None, chunk_bytes: int = 64 << 20, comm_dtype: Optional[torch.dtype] = None, hooks: bool = True,
                 always_reduce: bool = False, max_grad_norm: float = 0.0):
        self.lr, self.betas, self.weight_decay, self.eps = lr, betas, weight_decay, eps
        self.ema_decay = ema_decay
        # xqgan_train.py:456-458,471-473: `if args.max_grad_norm != 0.0: scaler.unscale_(optimizer); clip_grad_norm_(parameters, max_grad_norm)`
        # — after the all-reduce (DDP has averaged by then: the norm is taken of g / world), before the optimizer.  Here: one norm pass over
        # the flat gradient arena, the clip coefficient stays on the device and the fused step multiplies it in (no host read, graph-capturable).
        # `last_grad_norm` (device scalar, what clip_grad_norm_ returns) is valid after step().
        self.max_grad_norm = float(max_grad_norm or 0.0)
        self.last_grad_norm = None
        self._clip = self._clip_ws = None
        params = list(params)
        # torch.optim.AdamW(model.parameters()) — what the reference builds (xqgan_train.py:344-347) — numbers its state by position
        # in the FULL parameter list, frozen ones included (the frozen semantic_model sits between the decoder and sem_linear):
        # the checkpoint layout below uses those positions, not positions among the trainable parameters
        self._n_all = len(params)
        self._torch_index = [i for i, p in enumerate(params) if p.requires_grad]
        self.arena = FlatArena(params, with_ema=use_ema)
        self.reducer = GradAllReducer(self.arena.g, group=group, chunk_bytes=chunk_bytes,
                                      params=self.arena.params if hooks else None, offsets=self.arena.offsets if hooks else None,
                                      comm_dtype=comm_dtype, always=always_reduce, collect_fn=self.arena.collect if hooks else None)
        self.world = self.reducer.world

    # -- checkpointing: the layout of torch.optim.AdamW.state_dict() (what the reference saves as "optimizer" /
    #    "optimizer_disc", xqgan_train.py:580-600): state keyed by the parameter's position in model.parameters() (frozen
    #    parameters count and have no entry), one param group listing every position; plus the EMA copy ----------------------------
    def state_dict(self):
        a = self.arena
        state = {}
        for ti, p, o in zip(self._torch_index, a.params, a.offsets):
            n = p.numel()
            state[ti] = {"step": torch.tensor(float(a.step_count)), "exp_avg": a.m[o:o + n].view(p.shape).clone(),
                         "exp_avg_sq": a.v[o:o + n].view(p.shape).clone()}
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.weight_decay,
                 "amsgrad": False, "maximize": False, "params": list(range(self._n_all))}
        out = {"state": state, "param_groups": [group]}
        if a.ema is not None:
            out["ema"] = [a.ema[o:o + p.numel()].view(p.shape).clone() for p, o in zip(a.params, a.offsets)]
        return out

    @torch.no_grad()
    def load_state_dict(self, sd, allow_partial: bool = False):
        a = self.arena
        st = {int(k): v for k, v in sd["state"].items()}
        groups = sd["param_groups"]
        n_listed = sum(len(g["params"]) for g in groups)
        if n_listed != self._n_all:
            raise ValueError(f"optimizer state lists {n_listed} parameters, the model has {self._n_all} (frozen ones included): "
                             "not a checkpoint of this parameter list")
        unknown = sorted(set(st) - set(self._torch_index))
        if unknown:
            raise ValueError(f"optimizer state for parameter positions {unknown[:8]}... that are not trainable here")
        # torch.optim.AdamW creates state only for parameters that have received a gradient, so a checkpoint may lack entries for trainable
        # parameters (never used upstream, or frozen when it was written).  Those start from zero moments here — but NOT as torch would
        # continue them: the arena keeps ONE step count, so such a parameter is bias-corrected with the loaded step N where torch would
        # restart its own count at 0 (its first updates come out scaled by about (1 - beta1) / sqrt(1 - beta2) relative to torch's).  With
        # step > 0 that is only accepted on request (allow_partial=True), and always reported.
        missing = [ti for ti in self._torch_index if ti not in st]
        loaded_steps = {int(float(e["step"])) for e in st.values()}
        if missing and loaded_steps and max(loaded_steps) > 0:
            msg = (f"optimizer state has no entry for {len(missing)} trainable parameter position(s) {missing[:8]}{'...' if len(missing) > 8 else ''} "
                   f"while the others are at step {max(loaded_steps)}")
            if not allow_partial:
                raise ValueError(msg + ": truncated / mismatched checkpoint? (allow_partial=True loads it with zero moments for those, "
                                 "bias-corrected with the shared step count)")
            import warnings
            warnings.warn("ArenaOptimizer.load_state_dict: " + msg + "; they start from zero moments under the SHARED step count "
                          "(torch would restart their own counts at 0)", RuntimeWarning, stacklevel=2)
        steps = set()
        for ti, p, o in zip(self._torch_index, a.params, a.offsets):
            e = st.get(ti)
            n = p.numel()
            if e is None:
                a.m[o:o + n].zero_()
                a.v[o:o + n].zero_()
                continue
            if tuple(e["exp_avg"].shape) != tuple(p.shape):
                raise ValueError(f"optimizer state of parameter {ti}: shape {tuple(e['exp_avg'].shape)} vs {tuple(p.shape)}")
            a.m[o:o + n].copy_(e["exp_avg"].reshape(-1))
            a.v[o:o + n].copy_(e["exp_avg_sq"].reshape(-1))
            steps.add(int(float(e["step"])))
        if len(steps) > 1:
            raise ValueError("per-parameter step counts differ: not an AdamW state this optimizer can hold")
        a.step_count = steps.pop() if steps else 0
        g = groups[0]
        self.lr, self.betas, self.eps, self.weight_decay = g["lr"], tuple(g["betas"]), g["eps"], g["weight_decay"]
        if "ema" in sd and a.ema is not None:
            for t, p, o in zip(sd["ema"], a.params, a.offsets):
                a.ema[o:o + p.numel()].copy_(t.reshape(-1))
        a.resync(ema=False)   # the masters were (re)loaded by the model's own load_state_dict

    def zero_grad(self):
        self.arena.g.zero_()

    def step(self):
        a = self.arena
        a.step_count += 1
        a.epoch += 1            # masters change through raw pointers below: derived caches must refresh
        if a.p.is_cuda:
            stream = ctypes.c_void_p(torch.cuda.current_stream(a.p.device).cuda_stream)
            with torch.cuda.device(a.p.device):
                clip = None
                if self.max_grad_norm > 0.0:
                    if self._clip is None:
                        self._clip = torch.zeros(2, dtype=torch.float32, device=a.p.device)
                        self._clip_ws = torch.empty(int(_lib.lib().xq_grad_norm_workspace_bytes()) // 8, dtype=torch.float64, device=a.p.device)
                    rc = _lib.lib().xq_grad_norm_clip(ptr(a.g), a.numel, ctypes.c_float(1.0 / self.world), ctypes.c_float(self.max_grad_norm),
                                                      ptr(self._clip_ws), self._clip_ws.numel() * 8, ptr(self._clip), stream)
                    check(rc, "xq_grad_norm_clip")
                    clip = self._clip
                    self.last_grad_norm = self._clip[0]
                if torch.cuda.is_current_stream_capturing():
                    # hipGraph capture (CapturedStep): the step count must advance on every REPLAY, so it lives on the device:
                    # counter += 1 and the two bias-correction factors are (captured) tensor ops in double, the kernel reads them
                    if getattr(self, "_step_dev", None) is None:
                        raise RuntimeError("ArenaOptimizer.prepare_capture() must run before the step is captured")
                    with torch.no_grad():
                        self._step_dev.add_(1.0)
                        bc1 = 1.0 - torch.pow(torch.full_like(self._step_dev, self.betas[0]), self._step_dev)
                        bc2 = 1.0 - torch.pow(torch.full_like(self._step_dev, self.betas[1]), self._step_dev)
                        self._coeffs.copy_(torch.cat([self.lr / bc1, torch.rsqrt(bc2)]).float())
                    rc = _lib.lib().xq_adamw_ema_step_ex(ptr(a.p), ptr(a.g), ptr(a.m), ptr(a.v), ptr(a.ema), ptr(a.p16), a.numel,
                                                         ctypes.c_float(self.lr), ctypes.c_float(self.betas[0]),
                                                         ctypes.c_float(self.betas[1]), ctypes.c_float(self.eps),
                                                         ctypes.c_float(self.weight_decay), 0, ptr(self._coeffs), ptr(clip),
                                                         ctypes.c_float(self.ema_decay), ctypes.c_float(1.0 / self.world), 1, stream)
                    check(rc, "xq_adamw_ema_step_ex")
                    a.refresh_transposed_shadows()
                    a.refresh_conv_packs()
                    return
                rc = _lib.lib().xq_adamw_ema_step_ex(ptr(a.p), ptr(a.g), ptr(a.m), ptr(a.v), ptr(a.ema), ptr(a.p16), a.numel,
                                                     ctypes.c_float(self.lr), ctypes.c_float(self.betas[0]),
                                                     ctypes.c_float(self.betas[1]), ctypes.c_float(self.eps),
                                                     ctypes.c_float(self.weight_decay), a.step_count, None, ptr(clip),
                                                     ctypes.c_float(self.ema_decay), ctypes.c_float(1.0 / self.world), 1, stream)
            check(rc, "xq_adamw_ema_step_ex")
            a.refresh_transposed_shadows()
            a.refresh_conv_packs()
        else:
            self._step_host()

    def prepare_capture(self):
        """device-resident step counter (= the steps taken so far) + coefficient buffer for a captured step"""
        dev = self.arena.p.device
        self._step_dev = torch.full((1,), float(self.arena.step_count), dtype=torch.float64, device=dev)
        self._coeffs = torch.zeros(2, dtype=torch.float32, device=dev)

    @torch.no_grad()
    def _step_host(self):
        """CPU twin of xq_adamw_ema_step (used by the gloo multi-process tests; same formulas, tensor ops)."""
        a, (b1, b2) = self.arena, self.betas
        g = a.g * (1.0 / self.world)
        if self.max_grad_norm > 0.0:      # clip_grad_norm_ on the averaged gradient (xqgan_train.py:456-458)
            total = torch.linalg.vector_norm(g.double()).float()
            self.last_grad_norm = total
            g = g * torch.clamp(self.max_grad_norm / (total + 1e-6), max=1.0)
        a.p.mul_(1 - self.lr * self.weight_decay)
        a.m.lerp_(g, 1 - b1)
        a.v.mul_(b2).addcmul_(g, g, value=1 - b2)
        bc1, bc2 = 1 - b1 ** a.step_count, 1 - b2 ** a.step_count
        denom = (a.v.sqrt() / (bc2 ** 0.5)).add_(self.eps)
        a.p.addcdiv_(a.m, denom, value=-self.lr / bc1)
        if a.ema is not None:
            a.ema.mul_(self.ema_decay).add_(a.p, alpha=1 - self.ema_decay)
        a.g.zero_()


class TokenizerTrainStep:
    """One object = the reference's (vq_model, optimizer, ema, vq_loss, optimizer_disc) bundle for one rank.

    gen_loss_fn(model_out, imgs) -> scalar generator loss (reference: VQLoss(..., optimizer_idx=0))
    disc_step_fn(imgs, recons_detached) -> None (optional): the reference's discriminator step (optimizer_idx=1,
        backward, optimizer_disc.step) — runs between the start and the end of the gradient all-reduce.
    """

    def __init__(self, model: torch.nn.Module, gen_loss_fn: Callable, lr=1e-4, betas=(0.9, 0.95), weight_decay=5e-2,
                 eps=1e-8, ema_decay=0.9999, use_ema=True, amp_dtype: Optional[torch.dtype] = torch.bfloat16,
                 disc_step_fn: Optional[Callable] = None, group=None, chunk_bytes: int = 64 << 20,
                 comm_dtype: Optional[torch.dtype] = None, hooks: bool = True, always_reduce: bool = False, max_grad_norm: float = 0.0):
        self.model = model
        self.gen_loss_fn = gen_loss_fn
        self.disc_step_fn = disc_step_fn
        self.amp_dtype = amp_dtype
        self.opt = ArenaOptimizer(model.parameters(), lr=lr, betas=betas, weight_decay=weight_decay, eps=eps,
                                  ema_decay=ema_decay, use_ema=use_ema, group=group, chunk_bytes=chunk_bytes,
                                  comm_dtype=comm_dtype, hooks=hooks, always_reduce=always_reduce, max_grad_norm=max_grad_norm)
        self.arena = self.opt.arena
        self.reducer = self.opt.reducer
        self.world = self.opt.world
        self.device = self.arena.p.device
        # codebook-usage statistics: no host read inside the step (lazy.py) — upstream's .item() per forward (xqgan_model.py:788,
        # quant.py:140) would stall the stream every step and cannot be recorded into a hipGraph
        for m in model.modules():
            if hasattr(m, "lazy_usages"):
                m.lazy_usages = True

    # -- one train step -----------------------------------------------------------------------------------------
    def step(self, imgs, epoch=0, alpha=0.0, beta=0.0, delta=100):
        self.eager_steps = getattr(self, "eager_steps", 0) + 1      # what CapturedStep.replay checks (not the optimizer's step count)
        self.arena.release_grads()                 # backward nodes' gradient tensors are adopted, not added (FlatArena.collect)
        dev_type = imgs.device.type
        with torch.autocast(device_type=dev_type, dtype=self.amp_dtype, enabled=self.amp_dtype is not None):
            out = self.model(imgs, epoch, alpha, beta, delta)
            loss_gen = self.gen_loss_fn(out, imgs)
        _marker(50)
        self.reducer.arm()                         # chunks leave from the backward hooks as soon as they are complete
        loss_gen.backward()
        _marker(62)
        if not (self.reducer.active and self.reducer.collect_fn is not None):
            self.arena.collect()                   # single process (or hook-less reducer): everything at once
        self.reducer.start()                       # the rest (collected chunk by chunk); RCCL over xGMI, overlapped with ...
        if self.disc_step_fn is not None:
            self.disc_step_fn(imgs, out[0].detach())   # ... the discriminator step (needs only recons.detach())
        self.reducer.wait()
        _marker(70)
        self.opt.step()                            # AdamW + EMA + zero_grad + 1/world in one pass
        _marker(71)
        return loss_gen.detach()

    def capture(self, imgs, epoch=0, alpha=0.0, beta=0.0, delta=100, **kw) -> "CapturedStep":
        """record this step into a hipGraph: see CapturedStep (and its caveat about allocator activity between replays)"""
        import warnings
        warnings.warn("CapturedStep: on ROCm 7.2 / PyTorch 2.10 a hipGraph replay that follows allocator activity (an eager step, a large "
                      "allocation) has ended in illegal memory accesses — with a graph of ATen kernels only, in silently wrong values "
                      "(profiles/r03_replay_after_eager.txt). Keep the replay loop allocation-free and validate what it computes.",
                      RuntimeWarning, stacklevel=2)
        return CapturedStep(self, imgs, epoch, alpha, beta, delta, **kw)


class CapturedStep:
    """The complete train step of a TokenizerTrainStep recorded once into a hipGraph (torch.cuda.CUDAGraph) and replayed.

    Why: at B = 128 the GPU kernels of a step take ~204 ms but the eager step ~212 ms — a few thousand small launches
    (discriminator heads, spectral norm, loss assembly) leave bubbles that the 200+ us GEMMs around them do not cover; the replay
    has none (profiles/r02_hipgraph_probe.txt).  What a replay is: the same kernels on the same buffers — new data is copied into
    `static_imgs` first (replay(imgs) does it), torch's CUDA generator advances its Philox offset per replay (DropPath masks,
    DiffAug draws, perturbation draws differ from step to step as in eager mode), the optimizer's step counter lives on the device.
    What is FROZEN at capture time: every host-side decision — DiffAug's three branch draws (constant anyway at the reference's
    aug_prob = 1.0), epoch / alpha / beta / delta (the quantizer-dropout depths of codebook_drop > 0 configs, host draws upstream,
    move to the device generator for a captured model: VQModel.device_dropout_rng),
    learning rates, and the quantizers' `record_hit` counters, i.e. which coefficient (0.9 during the first 100 updates, then 0.99:
    xqgan_model.py:779-785) the codebook-usage EMA uses — a statistic only, but capture after the first 100 steps if it is logged.
    Single process only (collectives are not recorded): with world > 1 use the eager step.
    Replays must not be interleaved with eager steps of the same TokenizerTrainStep: eager steps AFTER the last replay are fine, a replay
    AFTER an eager step is refused (RuntimeError; capture again).  Measured in round 3 at B = 128: replay -> eager step -> replay ended
    in a GPU memory access fault on one box and in a hang on another (profiles/r03_replay_after_eager.txt); replay-only and
    replays-then-eager runs of the same build are clean.  The eager step replaces host-side objects whose device memory the recorded
    kernels still address (gradient tensors adopted by autograd, per-step caches); the exact object was not located — and the same
    pattern breaks a graph of ATen kernels only (tools/graph_alloc_probe.py: silently wrong values after a 1 GiB allocation between two
    replays), so this is a property of the hipGraph + caching-allocator stack, not of these kernels.  Keep the replay loop allocation-free
    (replay(imgs) copies into the static input buffer; allocate `imgs` from a fixed pool, e.g. two pinned staging buffers)."""

    def __init__(self, ts: "TokenizerTrainStep", imgs: torch.Tensor, epoch=0, alpha=0.0, beta=0.0, delta=100, warmup: int = 2,
                 allow_frozen_host_rng: bool = False):
        if ts.reducer.active:
            raise RuntimeError("CapturedStep: the gradient all-reduce is not recorded; use TokenizerTrainStep.step with world > 1")
        multi_scale = len(getattr(ts.model, "v_patch_nums", [0])) > 1
        # the device-RNG switch below stays on for the life of a SUCCESSFUL capture; a failed one puts the host RNG back (the eager steps
        # that follow must draw the dropout depths as upstream does, xqgan_model.py:274)
        self._prev_device_dropout_rng = getattr(ts.model, "device_dropout_rng", None)
        if multi_scale and float(getattr(ts.model, "codebook_drop", 0.0) or 0.0) > 0:
            # upstream draws the quantizer-dropout depths on the host every step (xqgan_model.py:274): a replay would freeze them.
            # The model draws them from the DEVICE generator instead (same distribution, advances per replay like DropPath's masks)
            if hasattr(ts.model, "device_dropout_rng"):
                ts.model.device_dropout_rng = True
            elif not allow_frozen_host_rng:
                raise RuntimeError("CapturedStep: this multi-scale model draws its dropout depths on the host every step; a replay "
                                   "would freeze them (allow_frozen_host_rng=True to accept that)")
        self.ts = ts
        self.static_imgs = imgs.clone()
        side = torch.cuda.Stream(device=imgs.device)
        side.wait_stream(torch.cuda.current_stream(imgs.device))
        with torch.cuda.stream(side):                 # warm-up on a side stream (lazy initialisations, allocator, workspace caches)
            for _ in range(warmup):
                ts.step(self.static_imgs, epoch, alpha, beta, delta)
        torch.cuda.current_stream(imgs.device).wait_stream(side)
        ts.opt.prepare_capture()
        disc = ts.disc_step_fn if isinstance(ts.disc_step_fn, DiscriminatorStep) else None
        if disc is not None:
            disc.opt.prepare_capture()
        self.graph = torch.cuda.CUDAGraph()
        c0 = ts.arena.step_count
        d0 = (disc.opt.arena.step_count, disc.global_step) if disc is not None else None
        try:
            with torch.cuda.graph(self.graph):
                self.loss = ts.step(self.static_imgs, epoch, alpha, beta, delta)
        except BaseException:
            self.restore_host_rng()
            raise
        finally:
            # the capture pass ran the host bookkeeping once without executing anything (also when it raised half way): the step
            # counters go back, replay() advances them per step.  The arena EPOCHS stay advanced (and move once more): every cache
            # filled during the capture (packed conv weights, ...) is stamped with the capture-time epoch but its producer kernels
            # were only recorded — eager code that runs before the first replay must not hit those entries
            ts.arena.step_count = c0
            ts.arena.epoch += 1
            if disc is not None:
                disc.opt.arena.step_count, disc.global_step = d0
                disc.opt.arena.epoch += 1
        self._disc = disc
        self._eager_seen = ts.eager_steps      # eager steps of this TokenizerTrainStep at capture time (see the class docstring)

    def restore_host_rng(self):
        """Put the model's dropout-depth draws back on the host generator (a discarded or failed capture)."""
        if self._prev_device_dropout_rng is not None and hasattr(self.ts.model, "device_dropout_rng"):
            self.ts.model.device_dropout_rng = self._prev_device_dropout_rng

    def replay(self, imgs: Optional[torch.Tensor] = None):
        if self.ts.eager_steps != self._eager_seen:
            raise RuntimeError(f"CapturedStep.replay: {self.ts.eager_steps - self._eager_seen} eager step(s) were taken since the capture; "
                               "capture the step again")
        if imgs is not None and imgs.data_ptr() != self.static_imgs.data_ptr():
            self.static_imgs.copy_(imgs, non_blocking=True)
        # the device step counters follow the host ones (ArenaOptimizer.load_state_dict between capture and replay moves them): the graph adds 1
        self.ts.opt._step_dev.fill_(float(self.ts.arena.step_count))
        if self._disc is not None:
            self._disc.opt._step_dev.fill_(float(self._disc.opt.arena.step_count))
        self.graph.replay()
        self.ts.arena.step_count += 1
        self.ts.arena.epoch += 1
        if self._disc is not None:
            self._disc.opt.arena.step_count += 1
            self._disc.opt.arena.epoch += 1
            self._disc.global_step += 1
        return self.loss


class DiscriminatorStep:
    """The reference's discriminator half-step (xqgan_train.py:464-475): zero_grad, VQLoss(optimizer_idx=1) under
    autocast, backward, AdamW on vq_loss.discriminator.parameters() — with the head gradients all-reduced once per
    step (DDP reduces them a second, wasted, time during the generator backward: SURVEY §2.3 C2)."""

    def __init__(self, vq_loss, lr=1e-4, betas=(0.9, 0.95), weight_decay=5e-2, amp_dtype=torch.bfloat16, group=None,
                 always_reduce: bool = False, max_grad_norm: float = 0.0):
        """group: give the discriminator its OWN process group (dist.new_group()) — a separate RCCL communicator and stream.
        On the generator's group its small all-reduce would queue behind the 689 MB gradient transfer that is in flight
        while this step runs, and the wait below would stall the compute stream until that transfer is through."""
        self.vq_loss = vq_loss
        self.amp_dtype = amp_dtype
        self.opt = ArenaOptimizer(vq_loss.discriminator.parameters(), lr=lr, betas=betas, weight_decay=weight_decay,
                                  use_ema=False, group=group, hooks=False, always_reduce=always_reduce, max_grad_norm=max_grad_norm)
        self.global_step = 0
        self.fade_blur_schedule = 0

    def state_dict(self):
        return {"optimizer_disc": self.opt.state_dict(), "global_step": self.global_step,
                "fade_blur_schedule": self.fade_blur_schedule}

    def load_state_dict(self, sd, allow_partial: bool = False):
        self.opt.load_state_dict(sd["optimizer_disc"], allow_partial=allow_partial)
        self.global_step = int(sd["global_step"])
        self.fade_blur_schedule = sd.get("fade_blur_schedule", 0)

    def __call__(self, imgs, recons_detached):
        self.opt.zero_grad()  # drops what the generator backward left on the heads (upstream: optimizer_disc.zero_grad())
        self.opt.arena.release_grads()
        with torch.autocast(device_type=imgs.device.type, dtype=self.amp_dtype, enabled=self.amp_dtype is not None):
            loss_disc = self.vq_loss(None, None, None, None, imgs, recons_detached, optimizer_idx=1,
                                     global_step=self.global_step + 1, fade_blur_schedule=self.fade_blur_schedule)
        _marker(43)
        loss_disc.backward()
        _marker(44)
        self.opt.arena.collect()
        self.opt.reducer.start()
        self.opt.reducer.wait()
        self.opt.step()
        _marker(45)
        self.global_step += 1
        return loss_disc.detach()
