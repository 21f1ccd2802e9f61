"""Row N3, backward: what `loss.backward()` leaves behind after `PARSeq.training_step` (strhub/models/parseq/system.py:168-199).

`loss_and_grads` = the whole step's forward and backward in fp32 from the model's master weights (dropout: the decoder's
eight sites per permutation pass when the system is in training mode; the encoder has none — timm's drop rates are 0):
`parseq_train_encoder_forward` (keeps the activations the backward needs) -> `parseq_train_decoder` (loss, gradient of every
`decoder.*` / `head.*` / `text_embed.*` parameter and `pos_queries`, and d loss / d memory) -> `parseq_train_encoder_backward`
(gradient of every `encoder.*` parameter).  `decoder_backward` is the middle stage alone, on any `memory`.
`TrainStep` adds gradient averaging across ranks, clipping and AdamW under the OneCycle schedule.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Dict, Optional, Sequence

import torch
from torch import Tensor

from . import _native
from .system import generate_attn_masks


@dataclass
class DecoderBackward:
    loss: Tensor                    # [] fp32 on the device
    perm_losses: Tensor             # [K]
    grads: Dict[str, Tensor]        # reference state_dict key -> gradient, views into `flat` (encoder.* are zero)
    flat: Tensor                    # [parseq_model_grad_elems] the buffer as the library lays it out
    dmemory: Tensor                 # [B, tokens, E]
    perms: Tensor
    workspace: Tensor               # raw workspace (floats); `intermediate(name)` reads it
    memory: Optional[Tensor] = None # the training forward's encoder output (loss_and_grads only)
    _shape: tuple = ()
    _model: Optional[object] = None

    def intermediate(self, name: str, numel: int) -> Tensor:
        """A named intermediate of the last permutation / an accumulator (parseq_train_decoder_workspace_offset), flat."""
        B, L, K = self._shape
        off = _native.lib().parseq_train_decoder_workspace_offset(self._model, B, L, K, name.encode())
        if off < 0:
            raise KeyError(name)
        return self.workspace[off:off + numel]


def param_views(native_model, flat: Tensor, shapes: Dict[str, Sequence[int]]) -> Dict[str, Tensor]:
    lib = _native.lib()
    out = {}
    for i in range(lib.parseq_model_num_params(native_model)):
        key, numel = C.c_char_p(), C.c_int64()
        _native.check(lib.parseq_model_param_info(native_model, i, C.byref(key), C.byref(numel)))
        off = lib.parseq_model_param_offset(native_model, i)
        name = key.value.decode()
        out[name] = flat[off:off + numel.value].view(*shapes[name])
    return out


def grad_segments(native_model):
    """[(begin, end, event)] of the flat gradient buffer in the order the training step's backward finishes them
    (parseq_train_grad_segment; valid after a parseq_train_encoder_backward on this model)."""
    lib = _native.lib()
    out = []
    for k in range(lib.parseq_train_grad_segments(native_model)):
        b, e, ev = C.c_int64(), C.c_int64(), C.c_void_p()
        _native.check(lib.parseq_train_grad_segment(native_model, k, C.byref(b), C.byref(e), C.byref(ev)))
        out.append((b.value, e.value, ev.value))
    return out


def loss_denominator(labels, num_perms: int) -> int:
    """Sum over the permutation passes of their count of non-<pad> targets (system.py:183-196), from the labels alone: every
    label contributes its characters plus <eos> to the first two passes and its characters only to the later ones."""
    chars = sum(len(s) for s in labels)
    return (chars + len(labels)) * min(num_perms, 2) + chars * max(num_perms - 2, 0)


def _set_train_precision(system, native):
    """`system.train_precision`: 'fp32' (default; exact products, the gradient-parity gate) or 'bf16' (GEMM operands rounded to
    bfloat16, fp32 accumulate and master weights — the reference trains `bf16-mixed`, train.py:62-64)."""
    mode = getattr(system, 'train_precision', 'fp32')
    if mode not in ('fp32', 'bf16'):
        raise ValueError(f"train_precision must be 'fp32' or 'bf16', got {mode!r}")
    _native.check(_native.lib().parseq_model_set_train_precision(native, _native.PARSEQ_BF16 if mode == 'bf16' else _native.PARSEQ_F32))


class _PinnedStaging:
    """Host -> device uploads of a step's small integer inputs without stalling anybody.  A `.to(device)` of a pageable tensor makes the
    host wait until the stream has reached the copy — in the middle of a training step that is the end of the encoder's forward, and
    whatever host work is left behind it then runs with the device idle (measured: 0.5 ms per step).  Here the values are written into a
    pinned buffer and copied with non_blocking=True; a ring of slots, each guarded by the event recorded behind its last copy, keeps a
    buffer from being rewritten before the device has read it (the host runs about one step ahead of the device)."""

    SLOTS = 8        # uploads per step = micro-batches (<= 4): at least two steps' worth

    def __init__(self):
        self.slots = [dict() for _ in range(self.SLOTS)]
        self.events = [None] * self.SLOTS
        self.next = 0

    def upload(self, named: Dict[str, Tensor], device) -> Dict[str, Tensor]:
        k = self.next
        self.next = (k + 1) % self.SLOTS
        if self.events[k] is not None:
            self.events[k].synchronize()          # the copies that last read this slot have run (never waits in practice: four steps ago)
        out = {}
        for name, t in named.items():
            buf = self.slots[k].get(name)
            if buf is None or buf.shape != t.shape or buf.dtype != t.dtype:
                buf = self.slots[k][name] = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            buf.copy_(t)
            out[name] = buf.to(device, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(device))
        self.events[k] = ev
        return out


@dataclass
class _DecoderInputs:
    tokens: Tensor                  # int32 [B, L] on the device
    targets: Tensor                 # int32 [2, B * L]: all targets | <eos> dropped (system.py:191-195)
    padding: Tensor                 # uint8 [B, L]
    masks: Tensor                   # uint8 [K, L, L] query masks of the K permutations
    perms: Tensor
    shape: tuple                    # (B, L, K)
    total: int                      # loss denominator


def prepare_decoder_inputs(system, labels, perms: Optional[Tensor] = None, parts: int = 1):
    """Everything `parseq_train_decoder` reads besides `memory`, computed on the HOST (tokenizer, permutation sampler, masks, padding,
    targets: system.py:171-196) and uploaded asynchronously — no device work is waited for, so the caller may run this before or after
    enqueueing the encoder's forward without stalling either side.

    parts > 1 (micro-batches of one step, `loss_and_grads_micro`): the batch is tokenised and its permutations drawn ONCE — every part keeps
    the whole batch's sequence length, permutations, masks and loss denominator, so that the parts' gradients add up to the batch's — and
    a list of `parts` inputs over consecutive row ranges is returned."""
    dev = system.device
    tgt = system.tokenizer.encode(labels, None)                      # on the CPU
    if perms is None:
        perms = system.gen_tgt_perms(tgt)
    perms = perms.cpu()
    B, K = tgt.shape[0], len(perms)
    if parts < 1 or B % parts:
        raise ValueError(f'a batch of {B} does not split into {parts} equal parts')
    total = loss_denominator(labels, K)
    masks = torch.stack([generate_attn_masks(p)[1] for p in perms]).to(torch.uint8).contiguous()
    staging = None
    if dev.type == 'cuda':
        staging = getattr(system, '_train_staging', None)
        if staging is None:
            staging = system._train_staging = _PinnedStaging()
    out = []
    for j in range(parts):
        rows = tgt[j * (B // parts):(j + 1) * (B // parts)]
        tgt_in, tgt_out = rows[:, :-1], rows[:, 1:]
        late = torch.where(tgt_out == system.eos_id, system.pad_id, tgt_out)
        host = {'targets': torch.stack([tgt_out.reshape(-1), late.reshape(-1)]).to(torch.int32).contiguous(), 'masks': masks,
                'padding': ((tgt_in == system.pad_id) | (tgt_in == system.eos_id)).to(torch.uint8).contiguous(),
                'tokens': tgt_in.to(torch.int32).contiguous()}
        d = staging.upload(host, dev) if staging is not None else {k: v.to(dev) for k, v in host.items()}
        out.append(_DecoderInputs(tokens=d['tokens'], targets=d['targets'], padding=d['padding'], masks=d['masks'], perms=perms,
                                  shape=(tgt_in.shape[0], tgt_in.shape[1], K), total=total))
    return out[0] if parts == 1 else out


def decoder_backward(system, images: Tensor, labels, perms: Optional[Tensor] = None, memory: Optional[Tensor] = None,
                     dropout: Optional[float] = None, seed: Optional[int] = None, inputs: Optional[_DecoderInputs] = None,
                     flat: Optional[Tensor] = None) -> DecoderBackward:
    """One training batch up to and including the decoder's backward.  `perms` defaults to a fresh draw from the system's
    sampler (system.py:175); `memory` defaults to `system.model.encode(images)` in the system's precision; `dropout` defaults
    to the model's rate in training mode (`system.train()`) and to 0 in evaluation mode; `seed` (the step's dropout masks)
    defaults to a draw from the system's numpy generator.  `inputs`: the result of `prepare_decoder_inputs` when the caller made it
    earlier (loss_and_grads does, ahead of the encoder's forward)."""
    lib = _native.lib()
    model = system.model
    dev = system.device
    if inputs is None:
        inputs = prepare_decoder_inputs(system, labels, perms)
    B, L, K = inputs.shape
    if memory is None:
        memory = model.encode(images)
    memory = memory.float().contiguous()
    if dropout is None:
        dropout = float(system.hparams.dropout) if system.training else 0.0
    if seed is None:
        seed = int(system.rng.integers(0, 2 ** 63)) if dropout > 0 else 0
    native = model._sync_native().model
    _set_train_precision(system, native)
    if flat is None:
        flat = torch.zeros(lib.parseq_model_grad_elems(native), dtype=torch.float32, device=dev)
    else:                                          # the caller's buffer (a micro-batch's persistent gradient buffer): zeroed on this stream
        flat.zero_()
    dmemory = torch.empty_like(memory)
    ws_bytes = lib.parseq_train_decoder_workspace_bytes(native, B, L, K)
    workspace = torch.empty(ws_bytes // 4, dtype=torch.float32, device=dev)
    loss = torch.empty(1 + K, dtype=torch.float32, device=dev)
    _native.check(lib.parseq_train_decoder(native, _native.ptr(memory), _native.ptr(inputs.tokens), _native.ptr(inputs.targets), _native.ptr(inputs.padding),
                                           _native.ptr(inputs.masks), B, L, K, inputs.total, float(dropout), int(seed), _native.ptr(loss), _native.ptr(flat),
                                           _native.ptr(dmemory),
                                           _native.ptr(workspace), ws_bytes, _native.stream_ptr(dev)))
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    return DecoderBackward(loss=loss[0], perm_losses=loss[1:], grads=param_views(native, flat, shapes), flat=flat, dmemory=dmemory, perms=inputs.perms,
                           workspace=workspace, _shape=(B, L, K), _model=native)


def loss_and_grads(system, images: Tensor, labels, perms: Optional[Tensor] = None, dropout: Optional[float] = None,
                   seed: Optional[int] = None, inputs: Optional[_DecoderInputs] = None, flat: Optional[Tensor] = None) -> DecoderBackward:
    """Loss and the gradient of EVERY parameter for one batch — the state `loss.backward()` leaves after the reference's
    `training_step` (system.py:168-199), dropout off.  `images`: fp32 [B, 3, H, W] on the device, normalised.  Everything is enqueued on
    the CURRENT stream (`inputs`: prepared ahead by the caller — loss_and_grads_micro)."""
    lib = _native.lib()
    model = system.model
    images = model._check_images(images)
    if images.dtype != torch.float32:
        images = ((images.float() / 255.0) - 0.5) / 0.5 if images.dtype == torch.uint8 else images.float()
    # the decoder's integer inputs first: host work + asynchronous uploads, nothing of it waits for the device (the draws from the system's
    # generators happen in the reference's order: permutations, then the dropout seed inside decoder_backward)
    if inputs is None:
        inputs = prepare_decoder_inputs(system, labels, perms)
    native = model._sync_native().model
    _set_train_precision(system, native)
    B = images.shape[0]
    ws_bytes = lib.parseq_train_encoder_workspace_bytes(native, B)
    ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=images.device)
    memory = torch.empty(B, model.encoder.pos_embed.shape[1], model._cfg['embed_dim'], dtype=torch.float32, device=images.device)
    _native.check(lib.parseq_train_encoder_forward(native, _native.ptr(images), B, _native.ptr(memory), _native.ptr(ws), ws_bytes,
                                                   _native.stream_ptr(images)))
    res = decoder_backward(system, images, labels, perms, memory=memory, dropout=dropout, seed=seed, inputs=inputs, flat=flat)
    _native.check(lib.parseq_train_encoder_backward(native, _native.ptr(res.dmemory), B, _native.ptr(res.flat), _native.ptr(ws), ws_bytes,
                                                    _native.stream_ptr(images)))
    res.memory = memory
    return res


def loss_and_grads_micro(system, images: Tensor, labels, perms: Optional[Tensor] = None, parts: int = 2, streams=None,
                         dropout: Optional[float] = None, flats=None) -> DecoderBackward:
    """The same loss and gradients with the batch cut into `parts` micro-batches that run AT ONCE, each on its own stream with its own
    workspaces and gradient buffer (the library keeps one second stream per caller stream for the backward), summed at the end.  NOT faster
    on one MI355X (batch 384: 27.0 ms in two parts against 24.7 ms in one piece — profiles/r06_train_step.md); kept as the tested guarantee
    that independent training chains may share a model on different streams, and for callers whose batch does not fit one workspace.  What is shared so that the sum IS the
    batch's gradient: one tokenisation (every part has the batch's sequence length), one draw of permutations, the batch's loss
    denominator (`total_targets`), the weights.  What differs from the one-piece schedule: the fp32 summation order of the weight gradients,
    and — with dropout on — which elements are dropped (the masks are counters of (seed, site, element index): every part draws its own seed).
    Returns the first part's record with `flat` / `grads` / `loss` replaced by the batch's.  `flats`: gradient buffers of parts 1 .. n - 1 kept by the
    caller across steps (TrainStep does): a buffer allocated on a part's stream and read on the caller's would have to be handed between the
    caching allocator's per-stream pools every step (deferred frees, fresh hipMallocs — and their synchronisations — in the middle of the step)."""
    model, dev = system.model, system.device
    images = model._check_images(images)
    B = images.shape[0]
    if parts <= 1:
        return loss_and_grads(system, images, labels, perms, dropout=dropout)
    inputs = prepare_decoder_inputs(system, labels, perms, parts=parts)
    cur = torch.cuda.current_stream(dev)
    if streams is None:
        streams = [torch.cuda.Stream(device=dev) for _ in range(parts - 1)]
    Bp = B // parts
    results, counts = [], []
    for j in range(parts):
        st = cur if j == 0 else streams[j - 1]
        if j:
            st.wait_stream(cur)                                  # the images (and whatever produced them) are ordered on the caller's stream
        with torch.cuda.stream(st):
            part_labels = labels[j * Bp:(j + 1) * Bp]
            results.append(loss_and_grads(system, images[j * Bp:(j + 1) * Bp], part_labels, inputs[j].perms, dropout=dropout, inputs=inputs[j],
                                          flat=flats[j - 1] if (flats is not None and j) else None))
            counts.append(loss_denominator(part_labels, inputs[j].shape[2]))
    res = results[0]
    loss = res.loss * (counts[0] / float(sum(counts)))
    for j in range(1, parts):
        cur.wait_stream(streams[j - 1])
        results[j].loss.record_stream(cur)                       # (a 4-byte tensor allocated on the part's stream, read here on the caller's)
        if flats is None:
            results[j].flat.record_stream(cur)
        res.flat.add_(results[j].flat)
        loss = loss + results[j].loss * (counts[j] / float(sum(counts)))
    res.loss = loss
    res.memory = None                                            # (the parts' encoder outputs are not stitched together)
    return res


def one_cycle_lr(step_num: int, total_steps: int, max_lr: float, pct_start: float, div_factor: float = 25.0,
                 final_div_factor: float = 1e4) -> float:
    """Learning rate of torch.optim.lr_scheduler.OneCycleLR(max_lr, total_steps, pct_start, cycle_momentum=False) — what
    strhub/models/base.py:103-106 configures — after `step_num` scheduler steps (cosine annealing, two phases)."""
    import math
    initial, floor = max_lr / div_factor, max_lr / div_factor / final_div_factor
    end1, end2 = float(pct_start * total_steps) - 1.0, float(total_steps) - 1.0
    if step_num > end2:
        raise ValueError(f'step {step_num} beyond the {total_steps} steps of the cycle')
    anneal = lambda a, b, pct: b + (a - b) / 2.0 * (math.cos(math.pi * pct) + 1.0)
    if step_num <= end1:
        return anneal(initial, max_lr, step_num / end1)
    return anneal(max_lr, floor, (step_num - end1) / (end2 - end1))


class TrainStep:
    """The per-batch work of `Trainer.fit` on the reference's configuration (base.py:98-110, configs/main.yaml:33-41): forward and
    backward of `training_step`, gradient averaging across ranks (what DDP does), gradient-norm clipping, one AdamW update under
    the OneCycle schedule — all on the device, no host synchronisation inside a step.  Epoch loops, checkpoints, logging and
    SWA are the framework's business and stay outside."""

    def __init__(self, system, total_steps: int, lr: Optional[float] = None, weight_decay: Optional[float] = None,
                 warmup_pct: Optional[float] = None, clip_val: float = 20.0, betas=(0.9, 0.999), eps: float = 1e-8,
                 num_devices: Optional[int] = None, accumulate_grad_batches: int = 1, process_group=None, micro_batches: Optional[int] = None):
        import math
        self.system = system
        self.total_steps = total_steps
        if num_devices is None:       # base.py:99 uses trainer.num_devices: default to the data-parallel world this step averages over
            num_devices = (torch.distributed.get_world_size(process_group)
                           if torch.distributed.is_available() and torch.distributed.is_initialized() else 1)
        self.num_devices = num_devices
        # base.py:98-101: linear scaling with the batch size, sqrt scaling with the number of devices
        scale = accumulate_grad_batches * math.sqrt(num_devices) * system.batch_size / 256.0
        self.max_lr = scale * (system.lr if lr is None else lr)
        self.weight_decay = system.weight_decay if weight_decay is None else weight_decay
        self.pct_start = system.warmup_pct if warmup_pct is None else warmup_pct
        self.clip_val, self.betas, self.eps = clip_val, betas, eps
        self.process_group = process_group
        # micro-batches of a step that run at once on separate streams (loss_and_grads_micro).  Default ONE: measured at batch 384, two parts at
        # once take 27.0 ms against 24.7 ms in one piece, three 28.7, four 34.0 (profiles/r06_train_step.md) — the parts run the same phase at the
        # same time and compete for what that phase is short of; round 5's probe (two free-running half-steps: 23.6 ms) had overlapped DIFFERENT
        # phases of consecutive iterations, which a real step — joined at the optimiser — cannot.  (PARSEQ_TRAIN_MICRO_BATCHES overrides.)
        self.micro_batches = micro_batches
        self._micro_streams = None
        self._micro_flats = None
        self.overlap_allreduce = True       # segment-wise all-reduce behind the backward's events (False: one pass after the backward)
        self.force_collectives = False      # run the collectives in a one-rank group too (self-test of the overlapped path on one GPU)
        self._comm_stream = None
        self.step_count = 0
        lib = _native.lib()
        model = system.model
        native = model._sync_native().model
        n = lib.parseq_model_grad_elems(native)
        dev = system.device
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        self._norm = torch.zeros(1, dtype=torch.float32, device=dev)
        self._norm_ws = torch.empty(1024, dtype=torch.float32, device=dev)
        # timm param_groups_weight_decay: 1-D tensors and biases are not decayed
        flags = [int(p.ndim > 1 and not k.endswith('.bias')) for k, p in model.state_dict().items()]
        self._decay_flags = (C.c_int32 * len(flags))(*flags)

    @property
    def lr(self) -> float:
        return one_cycle_lr(self.step_count, self.total_steps, self.max_lr, self.pct_start)

    def _parts(self, batch: int, distributed: bool) -> int:
        import os
        n = self.micro_batches
        env = os.environ.get('PARSEQ_TRAIN_MICRO_BATCHES')
        if env:
            n = int(env)
        if n is None:
            n = 1
        return n if n >= 1 and batch % n == 0 else 1

    def __call__(self, images: Tensor, labels, perms: Optional[Tensor] = None) -> Tensor:
        lib = _native.lib()
        system, model = self.system, self.system.model
        distributed = self.process_group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized())
        parts = self._parts(images.shape[0], distributed)
        if parts > 1:
            if self._micro_streams is None or len(self._micro_streams) != parts - 1:
                self._micro_streams = [torch.cuda.Stream(device=system.device) for _ in range(parts - 1)]
                self._micro_flats = [torch.empty(lib.parseq_model_grad_elems(model._sync_native().model), dtype=torch.float32, device=system.device) for _ in range(parts - 1)]
            res = loss_and_grads_micro(system, images, labels, perms, parts=parts, streams=self._micro_streams, flats=self._micro_flats)
        else:
            res = loss_and_grads(system, images, labels, perms)      # enqueued, not waited for: the device is still in the backward here
        native = model._sync_native().model
        if distributed:
            from .parallel import average_gradient_segments, average_gradients
            if parts == 1 and self.overlap_allreduce and res.flat.is_cuda and lib.parseq_train_grad_segments(native) > 0:
                # the all-reduce of each gradient segment starts when the backward has finished writing it (events recorded by
                # parseq_train_encoder_backward), on a side stream, while the earlier blocks' backward is still running
                if self._comm_stream is None:
                    self._comm_stream = torch.cuda.Stream(device=res.flat.device)
                average_gradient_segments(res.flat, grad_segments(native), self.process_group,
                                          wait=lambda st, ev: _native.check(lib.parseq_stream_wait_event(C.c_void_p(st.cuda_stream), C.c_void_p(ev))),
                                          comm_stream=self._comm_stream, force=self.force_collectives)
            else:
                average_gradients(res.flat, self.process_group)
        stream = _native.stream_ptr(res.flat)
        norm = None
        if self.clip_val:
            with _native.guard(res.flat):
                _native.check(lib.parseq_grad_norm(_native.ptr(res.flat), res.flat.numel(), _native.ptr(self._norm), _native.ptr(self._norm_ws), stream))
            norm = self._norm
        lr = self.lr
        self.step_count += 1
        _native.check(lib.parseq_adamw_step(native, _native.ptr(res.flat), _native.ptr(self.exp_avg), _native.ptr(self.exp_avg_sq),
                                            self._decay_flags, lr, self.betas[0], self.betas[1], self.eps, self.weight_decay,
                                            self.step_count, _native.ptr(norm), float(self.clip_val or 0.0), stream))
        model._adopt_native_weights()
        self.last = res
        return res.loss


class _TrainingStepFunction(torch.autograd.Function):
    """`training_step` as one autograd node: the forward launches the whole fused forward + backward on the device and keeps the
    gradients; `loss.backward()` hands them to the parameters (scaled by the incoming gradient), so torch optimisers,
    `clip_grad_norm_`, gradient accumulation and Lightning's automatic optimisation work unchanged on top."""

    @staticmethod
    def forward(ctx, system, images, labels, perms, *params):
        res = loss_and_grads(system, images, labels, perms)
        ctx.grads = [res.grads[k] for k, _ in system.model.named_parameters()]
        return res.loss.clone()

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_out):
        # The gradients were computed by the library in forward(); they stay on the node, so a second backward through it
        # (retain_graph=True, gradient-checking utilities) returns the same values instead of failing on a dropped buffer.
        if ctx.grads is None:
            raise RuntimeError('training_step loss: the stored gradients were released')
        return (None, None, None, None) + tuple(g * grad_out for g in ctx.grads)


def training_step_loss(system, images: Tensor, labels, perms: Optional[Tensor] = None) -> Tensor:
    """The loss of `PARSeq.training_step` (system.py:168-199) as a differentiable scalar w.r.t. every parameter of the model."""
    names = [k for k, _ in system.model.named_parameters()]
    if names != list(system.model.state_dict().keys()):
        raise RuntimeError('parameters and state_dict disagree (buffers?): the gradient buffer is laid out by state_dict order')
    return _TrainingStepFunction.apply(system, images, labels, perms, *system.model.parameters())
