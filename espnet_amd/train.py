"""Training-loop pieces of the hot path, re-designed for one process per MI355X:

* FlatParams   - every parameter / gradient lives in one contiguous fp32 arena (sized for HBM3E:
                 one memset, one Adam launch, a handful of large collectives instead of one per tensor);
                 backward kernels accumulate straight into the gradient arena.
* NoamAdam     - Adam(betas=(0.9,0.98), eps=1e-9) + Noam / WarmupLR schedule + clip_grad_norm_ +
                 non-finite-step skip, all evaluated on the device (no host sync, graph-capturable).
* GradReducer  - data-parallel gradient all-reduce over RCCL: fixed-size buckets of the arena, each
                 launched as soon as its last gradient is final, overlapping the rest of backward.
* train_step / GraphedDataParallelStep - espnet2 Trainer.train_one_epoch semantics for one step.

reference: espnet2/train/trainer.py:118-322,325-495 (DDP wrap, per-step collectives, clip, finite
check, optimizer/scheduler step), espnet2/torch_utils/recursive_op.py:14-53, transformer/optimizer.py,
espnet2/schedulers/warmup_lr.py, espnet2/train/distributed_utils.py:28-107.
"""
import os

import torch

from . import functional as F_
from . import graphs, ops


class FlatParams:
    def __init__(self, model):
        params = self._layout_order(model)
        if not params:
            raise ValueError("model has no parameters")
        dev = params[0].device
        self.params = params
        offs, n = [], 0
        for p in params:
            n = (n + 7) // 8 * 8          # 32-byte (fp32) / 16-byte (bf16 shadow) aligned tensors
            offs.append(n)
            n += p.numel()
        self.numel = (n + 7) // 8 * 8
        self.offsets = offs
        self.data = torch.zeros(self.numel, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(self.numel, device=dev, dtype=torch.float32)
        if dev.type == "cuda":
            ops.register_grad_arena(self.grad)      # weight-gradient GEMMs into this arena may be grouped (ops.wgrad_group_begin)
        # bf16 shadow of every weight for the bf16-operand GEMMs; kept current by the Adam kernel
        self.shadow = torch.zeros(self.numel, device=dev, dtype=torch.bfloat16) if dev.type == "cuda" else None
        for p, o in zip(params, offs):
            view = self.data[o:o + p.numel()].view(p.shape)
            view.copy_(p.data)
            p.data = view
            p._eamd_grad = self.grad[o:o + p.numel()].view(p.shape)
            p._eamd_grad._eamd_arena = True      # persistent: kernels may still add to it after the block's backward returned
        self.model = model
        self.refresh_shadow()

    @staticmethod
    def _layout_order(model):
        """registration order, except that parameters which one GEMM can serve are placed back to back (weights, then
        biases): the q / k / v projections of every self-attention module (MHABlockFn runs them as ONE [3D, D] GEMM),
        linear_k / linear_v of ALL source-attention modules of a decoder stack (one [2 D layers, D] GEMM on the encoder
        memory) and linear_pos of ALL self-attention modules of an encoder stack (F_.SharedProjFn).  Only the arena
        layout changes; parameter names / state_dict are the reference's."""
        import re
        named = list(model.named_parameters())
        by_name = dict(named)
        # stack-wide groups: stack prefix -> ordered member names
        stacks = {}
        for name, _p in named:
            m = re.match(r"(.*)\.(\d+)\.src_attn\.linear_k\.weight$", name)
            if m:
                stacks.setdefault(("kv", m.group(1)), []).append(int(m.group(2)))
            m = re.match(r"(.*)\.(\d+)\.self_attn\.linear_pos\.weight$", name)
            if m:
                stacks.setdefault(("pos", m.group(1)), []).append(int(m.group(2)))
        group_of = {}
        for (kind, base), idx in stacks.items():
            idx = sorted(idx)
            if len(idx) < 2:
                continue
            if kind == "kv":
                names = ["%s.%d.src_attn.linear_%s.weight" % (base, i, kv) for i in idx for kv in "kv"]
                names += ["%s.%d.src_attn.linear_%s.bias" % (base, i, kv) for i in idx for kv in "kv"]
            else:
                names = ["%s.%d.self_attn.linear_pos.weight" % (base, i) for i in idx]
            if all(n in by_name for n in names) and len({id(by_name[n]) for n in names}) == len(names):
                for n in names:
                    group_of[n] = names
                    by_name[n]._eamd_stack_group = kind
        seen, order = set(), []
        for name, p in named:
            if id(p) in seen:
                continue
            if name in group_of:
                for g in group_of[name]:
                    order.append(by_name[g])
                    seen.add(id(by_name[g]))
                continue
            if name.endswith("linear_q.weight"):
                base = name[: -len("linear_q.weight")]
                grp = [base + n for n in ("linear_q.weight", "linear_k.weight", "linear_v.weight", "linear_q.bias",
                                          "linear_k.bias", "linear_v.bias")]
                if all(g in by_name and id(by_name[g]) not in seen and g not in group_of for g in grp):
                    for g in grp:
                        order.append(by_name[g])
                        seen.add(id(by_name[g]))
                    continue
            order.append(p)
            seen.add(id(p))
        return order

    def refresh_shadow(self):
        """re-cast the whole arena (after construction / external weight changes)"""
        if self.shadow is None:
            return
        ops.cast_bf16(self.data, self.shadow)
        for p, o in zip(self.params, self.offsets):
            p._eamd_bf16 = self.shadow[o:o + p.numel()].view(p.shape)
            p._eamd_bf16_ver = p._version

    def zero_grad(self):
        self.grad.zero_()   # one memset node

    def expose_grads(self):
        """make .grad visible to code that expects the autograd convention (tests, checkpoints)"""
        for p in self.params:
            p.grad = p._eamd_grad


class NoamAdam:
    """mode 'noam': lr = factor * d^-0.5 * min(step^-0.5, step*warmup^-1.5)  (transformer/optimizer.py)
    mode 'warmuplr': lr = base * warmup^0.5 * min(step^-0.5, step*warmup^-1.5) (schedulers/warmup_lr.py)
    mode 'const': lr = base"""

    def __init__(self, flat, mode="noam", base_lr=1e-3, factor=1.0, model_size=256, warmup=25000,
                 betas=(0.9, 0.98), eps=1e-9, weight_decay=0.0, max_grad_norm=5.0):
        self.flat = flat
        self.mode = {"const": 0, "noam": 1, "warmuplr": 2}[mode]
        self.base_lr, self.factor, self.model_size, self.warmup = base_lr, factor, float(model_size), float(warmup)
        self.betas, self.eps, self.weight_decay, self.max_grad_norm = betas, eps, weight_decay, max_grad_norm
        dev = flat.data.device
        self.m = torch.zeros_like(flat.data)
        self.v = torch.zeros_like(flat.data)
        self.state = torch.zeros(8, device=dev, dtype=torch.float32)
        self.gnorm = torch.zeros(1, device=dev, dtype=torch.float32)
        self.ws = torch.empty(1024, device=dev, dtype=torch.float32)

    def step(self):
        ops.grad_norm(self.flat.grad, self.ws, self.gnorm)
        ops.sched_step(self.state, self.gnorm, self.mode, self.base_lr, self.factor, self.model_size, self.warmup,
                       self.betas[0], self.betas[1], self.max_grad_norm)
        ops.adam_step(self.flat.data, self.flat.grad, self.m, self.v, self.state, self.betas[0], self.betas[1],
                      self.eps, self.weight_decay, p16=self.flat.shadow)

    def stats(self):
        s = self.state.tolist()
        return dict(step=int(s[0]), lr=s[1], grad_norm=s[4], skipped=int(s[5]), clip_coef=s[6])

    def state_dict(self):
        return dict(m=self.m, v=self.v, state=self.state)

    def load_state_dict(self, sd):
        self.m.copy_(sd["m"])
        self.v.copy_(sd["v"])
        self.state.copy_(sd["state"])


class Adadelta:
    """torch.optim.Adadelta(rho=0.95, eps=args.eps, weight_decay=args.weight_decay) on the flat arenas - the optimizer of the
    RNN recipes (espnet/asr/pytorch_backend/asr.py:505-508; egs/librispeech/asr1/conf/tuning/train_rnn.yaml: opt adadelta)
    - with the espnet1 trainer's gradient clipping / non-finite guard (asr.py:228-240) and its eps decay
    (asr.py:798-830 -> `eps_decay(factor)`).  Same interface as NoamAdam (step / stats / state_dict)."""

    def __init__(self, flat, lr=1.0, rho=0.95, eps=1e-8, weight_decay=0.0, max_grad_norm=5.0):
        self.flat = flat
        self.lr, self.rho, self.weight_decay, self.max_grad_norm = lr, rho, weight_decay, max_grad_norm
        dev = flat.data.device
        self.square_avg = torch.zeros_like(flat.data)
        self.acc_delta = torch.zeros_like(flat.data)
        self.state = torch.zeros(8, device=dev, dtype=torch.float32)
        self.state[7] = eps
        self.gnorm = torch.zeros(1, device=dev, dtype=torch.float32)
        self.ws = torch.empty(1024, device=dev, dtype=torch.float32)

    @property
    def eps(self):
        return float(self.state[7])

    def eps_decay(self, factor):
        """asr.py:798-830 (adadelta_eps_decay): p["eps"] *= eps_decay; a device-side update, seen by captured steps"""
        self.state[7:8].mul_(factor)

    def step(self):
        ops.grad_norm(self.flat.grad, self.ws, self.gnorm)
        ops.sched_step(self.state, self.gnorm, 0, self.lr, 1.0, 1.0, 1.0, 0.0, 0.0, self.max_grad_norm)
        ops.adadelta_step(self.flat.data, self.flat.grad, self.square_avg, self.acc_delta, self.state, self.rho,
                          self.weight_decay, p16=self.flat.shadow)

    def stats(self):
        s = self.state.tolist()
        return dict(step=int(s[0]), lr=s[1], grad_norm=s[4], skipped=int(s[5]), clip_coef=s[6], eps=s[7])

    def state_dict(self):
        return dict(square_avg=self.square_avg, acc_delta=self.acc_delta, state=self.state)

    def load_state_dict(self, sd):
        self.square_avg.copy_(sd["square_avg"])
        self.acc_delta.copy_(sd["acc_delta"])
        self.state.copy_(sd["state"])


class GradReducer:
    """Bucketed, backward-overlapped gradient all-reduce on the flat arena (torch.distributed; the
    'nccl' backend is RCCL over xGMI on ROCm, 'gloo' for CPU rehearsal of the control flow)."""

    def __init__(self, flat, bucket_mb=32.0, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.flat = flat
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        cap = max(1, int(bucket_mb * 1024 * 1024 // 4))
        # buckets are contiguous arena ranges; built from the END (last-registered parameters receive
        # their gradients first in backward)
        self.bucket_of = {}
        self.buckets = []   # [start, end, pending, total]
        end = flat.numel
        cur_start, count = end, 0
        for p, o in reversed(list(zip(flat.params, flat.offsets))):
            cur_start = o
            count += 1
            if end - cur_start >= cap:
                self.buckets.append([cur_start, end, count, count])
                end, count = cur_start, 0
        if count > 0 or end > 0:
            self.buckets.append([0, end, count, count])
        bi = 0
        ranges = [(b[0], b[1]) for b in self.buckets]
        for p, o in zip(flat.params, flat.offsets):
            for i, (s, e) in enumerate(ranges):
                if s <= o < e:
                    self.bucket_of[id(p)] = i
                    break
        self.works = []
        self.uses = {}
        self.enabled = self.world > 1

    def begin(self):
        for b in self.buckets:
            b[2] = b[3]
        self.works = []
        self.uses = {}      # id(param) -> uses recorded by this step's forward that have not reported from backward yet

    def use(self, params):
        """called by block forward functions (GradSink.use): one more consumer of these parameters"""
        if not self.enabled:
            return
        for p in params:
            if p is not None and id(p) in self.bucket_of:
                self.uses[id(p)] = self.uses.get(id(p), 0) + 1

    def notify(self, params):
        """called by block backward functions: ONE use of these parameters has accumulated its gradients.  A
        parameter is final when its last recorded use reports (shared parameters, recurrent cells applied once per
        time step); a bucket starts its all-reduce when all of its parameters are final."""
        if not self.enabled:
            return
        for p in params:
            i = self.bucket_of.get(id(p)) if p is not None else None
            if i is None:
                continue
            left = self.uses.get(id(p), 1) - 1      # a use nobody recorded (direct GradSink call): treat as the only one
            self.uses[id(p)] = left
            if left != 0:
                continue
            b = self.buckets[i]
            b[2] -= 1
            if b[2] == 0:
                self._launch(b)

    def _launch(self, b):
        view = self.flat.grad[b[0]:b[1]]
        self.works.append(self.dist.all_reduce(view, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        if not self.enabled:
            return
        for b in self.buckets:        # parameters that took no part in this step's graph
            if b[2] > 0:
                b[2] = 0
                self._launch(b)
        for w in self.works:
            w.wait()
        self.works = []


def attach_reducer(reducer):
    ops.set_comm_overlap(reducer is not None and getattr(reducer, "world", 2) > 1)    # collectives beside backward: see ops.lstm_seq_ok
    F_.GradSink.on_done = reducer.notify if reducer is not None else None
    F_.GradSink.on_use = reducer.use if reducer is not None else None
    ops.defer_ln_reduce = reducer is None     # overlapped buckets need every block's gradients final when it reports


def train_step(model, flat, opt, batch, reducer=None, loss_scale=1.0):
    """One optimizer step: zero grads, forward, backward (+ overlapped all-reduce), clip, Adam.
    `batch` = (xs_pad, ilens, ys_pad) or a dict prepared by model.prepare()."""
    flat.zero_grad()
    ops.rng_advance(flat.data.device)      # new dropout masks for this step (device counter: graph-replay safe)
    if reducer is not None:
        reducer.begin()
    loss = model.forward_core(batch) if isinstance(batch, dict) else model(*batch)
    scale = loss_scale / (reducer.world if reducer is not None else 1)
    ops.wgrad_group_begin()      # small weight-gradient GEMMs of this backward pass leave as one grouped launch
    try:
        loss.backward(torch.full((), scale, device=loss.device, dtype=loss.dtype) if scale != 1.0 else None)
    finally:
        ops.wgrad_group_end()
    ops.wgrad_join()
    if reducer is not None:
        reducer.finish()
    opt.step()
    return loss


class E2EPhases:
    """Forward + backward of the hybrid CTC/attention model in PHASES separated by gradient cuts (nets/modules.GradCuts: the
    activation behind the encoder and those in front of two encoder layers are replaced by detached leaves; `loss.backward()`
    stops at the last cut, `upstream.backward(leaf.grad)` resumes).  The gradient arena keeps registration order, so each
    phase completes ONE contiguous range of it - which a data-parallel driver all-reduces while the next phase runs.
        phase 0      advance the dropout counter, forward, backward of decoder / CTC        -> ranges[0]
        phase 1 ..   the encoder layers from the top down, the input layer last           -> ranges[1 ..]
    Models without an `encoder.encoders` stack (or with interleaved arena ranges) get the single-phase plan.
    Neither zeroes the gradients nor runs the optimizer: gradient accumulation belongs to the caller.
    reference: what DistributedDataParallel's bucketed backward hooks do in espnet2/train/trainer.py:381-412."""

    def __init__(self, model, flat, phases=True):
        self.model, self.flat = model, flat
        self.ranges = [(0, flat.numel)]
        self.stack, self.cut_layers, self.cuts = None, (), []
        if phases:
            self._plan()

    def _plan(self):
        from .nets.modules import MultiSequential
        enc = getattr(self.model, "encoder", None)
        stack = getattr(enc, "encoders", None)
        if not isinstance(stack, MultiSequential) or len(stack) < 2 or not hasattr(self.model, "forward_core"):
            return
        # cuts low in the stack keep the exposed last range small (input layer + the lowest sixth of the layers:
        # 16 % of the bytes at config 2) while every other range has a whole phase of backward to hide under
        n = len(stack)
        cuts = sorted({max(1, n // 6), max(1, (7 * n) // 12)})
        nph = len(cuts) + 2
        phase_of = {}
        for name, p in self.model.named_parameters():
            if not name.startswith("encoder."):
                ph = 0                                              # decoder, CTC, anything behind the encoder
            elif getattr(p, "_eamd_stack_group", None) == "pos":
                ph = nph - 1        # linear_pos of every layer lives (and is finished) with the lowest layers: F_.SharedProjFn
            elif name.startswith("encoder.encoders."):
                ph = 1 + sum(1 for c in cuts if int(name.split(".")[2]) < c)
            elif name.startswith("encoder.after_norm."):
                ph = 1
            else:
                ph = nph - 1                                        # input layer
            phase_of[id(p)] = ph
        spans = {}
        for p, o in zip(self.flat.params, self.flat.offsets):
            ph = phase_of.get(id(p))
            if ph is None:
                return
            lo, hi = spans.get(ph, (o, o))
            spans[ph] = (min(lo, o), max(hi, o + p.numel()))
        if sorted(spans) != list(range(nph)):
            return
        order = sorted(spans.values())
        if any(a[1] > b[0] for a, b in zip(order, order[1:])):      # interleaved in the arena: keep one phase
            return
        bounds = [0] + [sp[0] for sp in order[1:]] + [self.flat.numel]
        self.ranges = [(bounds[order.index(spans[ph])], bounds[order.index(spans[ph]) + 1]) for ph in range(nph)]
        self.stack, self.cut_layers = stack, tuple(cuts)

    def phase(self, k, batch, scale):
        """scale: python float or 0-dim device tensor (a captured phase 0 reads the tensor at replay time).
        phase 0 -> (loss, {"loss_att", "loss_ctc", "acc"} device tensors the model left); later phases -> None"""
        from .nets.modules import GradCuts
        if k == 0:
            ops.rng_advance(self.flat.data.device)
            if self.stack is not None:
                self.stack.cut_before = self.cut_layers
                GradCuts.active = []
            try:
                loss = self.model.forward_core(batch) if isinstance(batch, dict) else self.model(*batch)
                self.cuts = GradCuts.active or []
            finally:
                GradCuts.active = None
                if self.stack is not None:
                    self.stack.cut_before = ()
            if self.stack is not None and len(self.cuts) != len(self.ranges) - 1:
                raise RuntimeError("phased backward expects %d gradient cuts, found %d" % (len(self.ranges) - 1, len(self.cuts)))
            if torch.is_tensor(scale):
                grad = scale.to(loss.dtype).reshape(loss.shape)
            else:
                grad = torch.full((), float(scale), device=loss.device, dtype=loss.dtype) if scale != 1.0 else None
            ops.wgrad_group_begin()      # one grouped launch of the small weight-gradient GEMMs per phase
            try:
                loss.backward(grad)
            finally:
                ops.wgrad_group_end()
            ops.wgrad_join()
            m = self.model
            stats = {k_: getattr(m, a) for k_, a in (("loss_att", "_loss_att_t"), ("loss_ctc", "_loss_ctc_t"), ("acc", "_acc_t"))
                     if getattr(m, a, None) is not None}
            return loss, stats
        upstream, leaf = self.cuts[len(self.cuts) - k]       # phase 1 resumes behind the encoder, the next ones lower in the stack
        ops.wgrad_group_begin()
        try:
            upstream.backward(leaf.grad)
        finally:
            ops.wgrad_group_end()
        ops.wgrad_join()
        if k == len(self.ranges) - 1:
            self.cuts = []
        return None


class GraphedDataParallelStep:
    """hipGraph replay of a data-parallel training step without capturing any collective:
      graph A = zero the gradient arena, advance the dropout counter, forward, backward (loss pre-scaled by
                1/world, as the reference's DDP averaging) - split by gradient cuts into phases A1 | A2 | A3 whose
                arena ranges are contiguous (see __init__);
      eager   = RCCL all-reduce (SUM) of each phase's range of the flat gradient arena, bucket by bucket,
                asynchronously on the communicator's stream, issued right behind the phase's graph so that it
                overlaps the next phase (torch.distributed orders it after that graph; graph B waits for all);
      graph B = gradient norm, clipping, Noam schedule, Adam, bf16 shadow refresh.
    Compared with the eager step + backward-overlapped buckets (train_step with a GradReducer) this gives up the
    overlap (about 1 ms of all-reduce for 187 MB over xGMI) and wins back the launch overhead of ~1500 kernels.
    reference: espnet2/train/trainer.py:381-467 (forward, backward, clip, step under DistributedDataParallel)."""

    def __init__(self, model, flat, opt, batch, world=1, group=None, bucket_mb=128.0, warmup=2, phases=True):
        import torch.distributed as dist
        self.dist, self.group, self.world = dist, group, world
        ops.set_comm_overlap(world > 1)          # each phase's all-reduce runs under the next phase's kernels
        self.model, self.flat, self.opt, self.batch = model, flat, opt, batch
        self.cap = max(1, int(bucket_mb * 1024 * 1024 // 4))
        # Phased backward (phases=True, models with an `encoder.encoders` layer stack): gradient cuts behind the
        # encoder and in the middle of its layer stack split graph A into A1 (forward + backward of decoder / CTC),
        # A2 (upper encoder layers) and A3 (lower layers + input layer).  The arena keeps registration order, so each
        # phase owns one contiguous range of the gradient arena; its all-reduce is issued as soon as the phase's
        # graph has been enqueued and runs on the communicator's stream under the next phase (one collective per
        # range up to bucket_mb: few, large messages).  Only the last range (about 25 % of the bytes at config 2)
        # is exposed.
        self.ranges = [(0, flat.numel)]
        self.cuts = []
        self._stack = None
        if phases:
            self._plan_phases()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                loss = self._phase(0)
                works = self._reduce(0)
                for k in range(1, len(self.ranges)):
                    self._phase(k)
                    works += self._reduce(k)
                self._wait(works)
                opt.step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        # thread_local: the communicator's watchdog thread may query its events while we capture
        self.graphs = []
        for k in range(len(self.ranges)):
            g = graphs.new_graph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                out = self._phase(k)
            graphs.audit(g, "training phase graph %d" % k)      # no memset nodes (they replay wrongly on this ROCm)
            if k == 0:
                self.loss = out
            self.graphs.append(g)
        self.graph_a = self.graphs[0]
        self.graph_b = graphs.new_graph()
        with torch.cuda.graph(self.graph_b, capture_error_mode="thread_local"):
            opt.step()
        graphs.audit(self.graph_b, "optimizer graph")

    def _plan_phases(self):
        plan = E2EPhases(self.model, self.flat)
        if plan.stack is not None:
            self.ranges, self._stack, self._cut_layers = plan.ranges, plan.stack, plan.cut_layers
        self._plan = plan

    def _phase(self, k):
        plan = getattr(self, "_plan", None)
        if plan is None:
            plan = self._plan = E2EPhases(self.model, self.flat, phases=False)
        if k == 0:
            self.flat.zero_grad()
        out = plan.phase(k, self.batch, 1.0 / self.world)
        return out[0] if k == 0 else None

    def _reduce(self, k):
        if not self.dist.is_initialized():
            return []
        lo, hi = self.ranges[k]
        return [self.dist.all_reduce(self.flat.grad[s:min(hi, s + self.cap)], op=self.dist.ReduceOp.SUM, group=self.group,
                                     async_op=True) for s in range(lo, hi, self.cap)]

    @staticmethod
    def _wait(works):
        for w in works:
            w.wait()      # stream-level wait: the current stream continues after the collective, no host block

    def __call__(self):
        works = []
        for k, g in enumerate(self.graphs):
            g.replay()
            works += self._reduce(k)      # runs on the communicator's stream under the next phase's graph
        self._wait(works)
        self.graph_b.replay()
        return self.loss

    def comm_profile(self, steps=8):
        """Where a data-parallel step's time goes, measured on this rank (every rank must call it: it issues collectives):
          allreduce_ms[k]   the all-reduce of phase k's arena range ALONE (nothing else on the device), stream events
                            around issue + wait, mean of `steps`;
          step_ms           the step as __call__ runs it;
          compute_only_ms   the same graphs with no collective issued (what one rank would need without peers);
          exposed_comm_ms   step_ms - compute_only_ms: the communication the backward phases do not hide.
        Parameters move on (the optimizer graph runs); call it after the timed region."""
        import time
        dist = self.dist
        on = dist.is_initialized()

        def sync():
            if on:
                dist.barrier(group=self.group)
            torch.cuda.synchronize()

        def timed(fn):
            for _ in range(2):
                fn()
            sync()
            t0 = time.perf_counter()
            for _ in range(steps):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / steps * 1e3

        def compute_only():
            for g in self.graphs:
                g.replay()
            self.graph_b.replay()

        ar = []
        for k in range(len(self.ranges)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self._wait(self._reduce(k))
            sync()
            e0.record()
            for _ in range(steps):
                self._wait(self._reduce(k))
            e1.record()
            torch.cuda.synchronize()
            ar.append(e0.elapsed_time(e1) / steps if on else 0.0)
        self.flat.zero_grad()      # the repeated SUMs above grew the arena world^steps-fold
        step_ms = timed(self.__call__)
        comp_ms = timed(compute_only)
        return dict(rccl_ranks=(dist.get_world_size(self.group) if on else 1),
                    backend=(dist.get_backend(self.group) if on else "none"),      # "nccl" = RCCL over xGMI; "gloo" = a CPU rehearsal
                    phase_mbytes=[round((hi - lo) * 4 / 1e6, 1) for lo, hi in self.ranges],
                    allreduce_ms=[round(v, 3) for v in ar], allreduce_ms_total=round(sum(ar), 3),
                    step_ms=round(step_ms, 3), compute_only_ms=round(comp_ms, 3),
                    exposed_comm_ms=round(step_ms - comp_ms, 3))


class BucketedGraphStep:
    """hipGraph replay for batches whose shapes vary (real recipes batch by `batch_bins`: B, T and L all move).
    A batch is padded up to its BUCKET - (B, T rounded up to `t_edge` frames, L rounded up to `l_edge` labels) - and
    each bucket owns one captured graph of the whole training step (zero grads, forward, backward, clip, Adam,
    schedule); a least-recently-used cache keeps `max_graphs` of them, all captured into one shared memory pool
    (they never replay concurrently).  First sight of a bucket runs the step eagerly (that IS the warm-up the capture
    needs, and it is a real training step, so the trajectory does not depend on the cache); the second sight captures
    and replays; later ones copy the new batch into the graph's static inputs and replay.

    Padding semantics: frames are padded with zeros, labels with ignore_id, and the step computes WHAT THE REFERENCE COMPUTES ON
    THE BATCH CROPPED TO ITS OWN LONGEST UTTERANCE: `model.prepare(pad_to=)` hands the kernels the length T' of the batch's own
    encoder time axis as a device scalar (`tbound`, refreshed per replay), the encoder mask is the reference's, and the three
    places of the Conformer that depend on T' follow the bound - the legacy rel_shift (eamd_attn_*'s shift_len), the depthwise
    convolution's zero padding (eamd_mask_time in front of it) and the BatchNorm statistics (eamd_bn_*_bounded).  Tested against
    the CPU restatement of the reference on the exact-shape batch (tests/test_gpu_model.py::test_bucketed_graph_step_conformer_is_reference_exact:
    loss 8e-8, running statistics 1e-7, gradients 1e-3) in all three modes.  With relative positions the padded step needs the
    fused attention kernels (d_k = 64, T' of the bucket <= 2048, ops.attn_fwd_supported): this is checked per bucket BEFORE
    anything runs, and a bucket they decline takes the eager step on the batch's exact shape instead (one warning)."""

    def __init__(self, model, flat, opt, t_edge=64, l_edge=8, max_graphs=8):
        from collections import OrderedDict
        self.model, self.flat, self.opt = model, flat, opt
        self.t_edge, self.l_edge, self.max_graphs = int(t_edge), int(l_edge), int(max_graphs)
        self.cache = OrderedDict()          # bucket -> dict(graph, static, loss)
        self.seen = {}                      # bucket -> number of eager runs so far
        self._ok = {}                       # bucket -> the padded step is supported (padded_step_ok)
        self.pool = None
        self.hits = self.misses = self.captures = self.evictions = 0

    def bucket(self, xs_pad, ilens, ys_pad, olens=None):
        il = [int(v) for v in (ilens.tolist() if torch.is_tensor(ilens) else ilens)]
        T = max(il)
        if olens is not None:      # label lengths known on the host (a data loader has them): no device round trip
            L = max(int(v) for v in (olens.tolist() if torch.is_tensor(olens) else olens))
        else:                      # device labels: this waits for the GPU (the previous step) before the host can go on
            L = int((ys_pad != self.model.ignore_id).sum(1).max())
        up = lambda v, e: (v + e - 1) // e * e  # noqa: E731
        return (int(xs_pad.shape[0]), up(T, self.t_edge), up(max(L, 1), self.l_edge))

    def padded_step_ok(self, key):
        """can the step run on the batch padded to bucket `key`?  (relative-position attention on a padded time axis needs
        the rel_shift over the batch's own length, which only the fused attention kernels take)"""
        ok = self._ok.get(key)
        if ok is None:
            from .nets.modules import RelPositionMultiHeadedAttention, embed_output_lengths
            rel = [m for m in self.model.modules() if isinstance(m, RelPositionMultiHeadedAttention)]
            ok = True
            if rel:
                tp = max(embed_output_lengths(self.model.encoder.embed, [key[1]], key[1]))
                ok = all(ops.attn_fwd_supported(tp, tp, m.d_k, True) for m in rel)
            self._ok[key] = ok
            if not ok:
                import logging
                logging.getLogger(__name__).warning(
                    "BucketedGraphStep: bucket %s runs eagerly on exact shapes (relative-position attention on a padded batch "
                    "needs the fused attention kernels: d_k = 64, T' <= 2048)", key)
        return ok

    def __call__(self, xs_pad, ilens, ys_pad, olens=None):
        key = self.bucket(xs_pad, ilens, ys_pad, olens)
        if not self.padded_step_ok(key):
            self.misses += 1
            return train_step(self.model, self.flat, self.opt, self.model.prepare(xs_pad, ilens, ys_pad))
        batch = self.model.prepare(xs_pad, ilens, ys_pad, pad_to=key[1:])
        entry = self.cache.get(key)
        if entry is not None:
            self.cache.move_to_end(key)
            for k, v in batch.items():
                if torch.is_tensor(v):
                    entry["static"][k].copy_(v, non_blocking=True)
            entry["graph"].replay()
            self.hits += 1
            return entry["loss"]
        self.misses += 1
        if self.seen.get(key, 0) == 0:          # first sight: eager step (doubles as the capture's warm-up)
            self.seen[key] = 1
            return train_step(self.model, self.flat, self.opt, batch)
        # second sight: capture the step on this batch's tensors (they become the graph's static inputs), then replay
        torch.cuda.synchronize()
        g = graphs.new_graph()
        if self.pool is None:
            self.pool = torch.cuda.graph_pool_handle()
        with torch.cuda.graph(g, pool=self.pool):
            loss = train_step(self.model, self.flat, self.opt, batch)
        graphs.audit(g, "bucketed step graph")
        g.replay()
        self.captures += 1
        self.cache[key] = dict(graph=g, static=batch, loss=loss)
        if len(self.cache) > self.max_graphs:
            self.cache.popitem(last=False)
            self.evictions += 1
        return loss

    def stats(self):
        n = self.hits + self.misses
        return dict(steps=n, hits=self.hits, hit_rate=(self.hits / n if n else 0.0), captures=self.captures,
                    evictions=self.evictions, graphs=len(self.cache))


class E2EProgram:
    """what ComposedStep needs to know about the hybrid CTC/attention models of this package: shape buckets, padded
    batches (model.prepare(pad_to=)), the phased forward / backward (E2EPhases).  raw batch = (xs_pad, ilens, ys_pad[, olens])."""

    def __init__(self, model, flat, t_edge=64, l_edge=8, phases=True):
        self.model, self.flat = model, flat
        self.t_edge, self.l_edge = int(t_edge), int(l_edge)
        self.plan = E2EPhases(model, flat, phases=phases)
        self.ranges = self.plan.ranges
        self._ok = {}

    def bucket(self, raw):
        xs_pad, ilens, ys_pad = raw[:3]
        olens = raw[3] if len(raw) > 3 else None
        il = [int(v) for v in (ilens.tolist() if torch.is_tensor(ilens) else ilens)]
        if olens is not None:      # label lengths known on the host (a data loader has them): no device round trip
            L = max(int(v) for v in (olens.tolist() if torch.is_tensor(olens) else olens))
        else:
            L = int((ys_pad != self.model.ignore_id).sum(1).max())
        up = lambda v, e: (v + e - 1) // e * e  # noqa: E731
        return (int(xs_pad.shape[0]), up(max(il), self.t_edge), up(max(L, 1), self.l_edge))

    def supported(self, key):
        """the step on the batch PADDED to its bucket computes what the reference computes on the exact shapes only where the
        kernels take the batch's own time bound (BucketedGraphStep.padded_step_ok)"""
        ok = self._ok.get(key)
        if ok is None:
            from .nets.modules import RelPositionMultiHeadedAttention, embed_output_lengths
            rel = [m for m in self.model.modules() if isinstance(m, RelPositionMultiHeadedAttention)]
            ok = True
            if rel:
                tp = max(embed_output_lengths(self.model.encoder.embed, [key[1]], key[1]))
                ok = all(ops.attn_fwd_supported(tp, tp, m.d_k, True) for m in rel)
            self._ok[key] = ok
        return ok

    def prepare(self, raw, key):
        xs_pad, ilens, ys_pad = raw[:3]
        return self.model.prepare(xs_pad, ilens, ys_pad, pad_to=key[1:]) if key is not None else self.model.prepare(xs_pad, ilens, ys_pad)

    def phase(self, k, batch, scale):
        return self.plan.phase(k, batch, scale)


class ComposedStep:
    """ONE training micro-step driver that is at once shape-bucketed, hipGraph-replayed, phased for data parallelism and
    accumulation-aware (what BucketedGraphStep, GraphedDataParallelStep and EpochRunner's eager step each did alone):

      * a batch goes to its shape bucket (program.bucket); first sight of a bucket runs the phases eagerly, the second sight
        captures one hipGraph per PHASE on the padded batch, later ones copy the batch into the static inputs and replay;
      * the backward scale w_r / (sum_r w_r * accum_grad) of espnet2's weighted data-parallel loss is a device scalar the captured
        phase 0 reads at replay time (trainer.py:385-397);
      * with reduce=True (the last micro-step of an accumulation window) the arena range each phase completed is all-reduced
        (SUM) asynchronously right behind the phase - collective sizes depend on the MODEL only, so ranks may sit in different
        buckets (or one eager, one replaying) in the same step - and runs under the next phase; wait() joins them;
      * gradients accumulate in the flat arena across micro-steps; zeroing and the optimizer belong to the caller
        (EpochRunner, or step() below for the plain one-batch-per-step loop).

    `program` supplies the model-specific parts (E2EProgram for the models of this package; any object with ranges / bucket /
    supported / prepare / phase, e.g. a toy model in the CPU tests).  Without CUDA (or graphs=False) every step is the eager
    phased step.  reference: espnet2/train/trainer.py:325-495 under DistributedDataParallel, abs_task.py:1436-1445."""

    def __init__(self, program, flat, opt, group=None, max_graphs=8, bucket_mb=128.0, use_graphs=None, rehearse=False):
        import torch.distributed as dist
        from collections import OrderedDict
        self.program, self.flat, self.opt = program, flat, opt
        self.dist, self.group = dist, group
        self.distributed = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if self.distributed else 1
        self.cap = max(1, int(bucket_mb * 1024 * 1024 // 4))
        self.use_graphs = flat.data.is_cuda if use_graphs is None else bool(use_graphs)
        self.cache, self.seen, self.pool = OrderedDict(), {}, None
        self.max_graphs = int(max_graphs)
        self.scale = torch.ones((), device=flat.data.device, dtype=torch.float32)      # static input of every captured phase 0
        self.works = []
        self.hits = self.misses = self.captures = self.evictions = self.eager_exact = 0
        self.rehearse = bool(rehearse)        # issue the collectives on a ONE-rank group too (bench.py --rehearse-dp)
        ops.set_comm_overlap(self.world > 1)

    # ---- collectives -----------------------------------------------------------------------------------------------------
    def _reduce(self, k):
        if not self.distributed or (self.world == 1 and not self.rehearse):
            return
        lo, hi = self.program.ranges[k]
        g = self.flat.grad
        self.works += [self.dist.all_reduce(g[s:min(hi, s + self.cap)], op=self.dist.ReduceOp.SUM, group=self.group, async_op=True)
                       for s in range(lo, hi, self.cap)]

    def wait(self):
        """join the range all-reduces issued by the last micro_step(reduce=True) (RCCL: a stream-level wait, no host block)"""
        works, self.works = self.works, []
        for w in works:
            w.wait()

    # ---- one micro-step --------------------------------------------------------------------------------------------------
    def micro_step(self, raw, scale=1.0, reduce=False):
        """forward + backward of one batch, gradients ADDED to the arena; -> (loss, stats) device tensors (of a replayed bucket:
        the graph's static outputs, valid until that bucket replays again)"""
        prog = self.program
        nph = len(prog.ranges)
        key = prog.bucket(raw)
        padded = prog.supported(key)
        self.scale.fill_(float(scale))
        entry = self.cache.get(key) if (self.use_graphs and padded) else None
        if entry is not None:
            self.cache.move_to_end(key)
            batch = prog.prepare(raw, key)
            for name, v in batch.items():
                if torch.is_tensor(v):
                    entry["static"][name].copy_(v, non_blocking=True)
            for k, g in enumerate(entry["graphs"]):
                g.replay()
                if reduce:
                    self._reduce(k)
            self.hits += 1
            return entry["out"]
        self.misses += 1
        if not padded:
            self.eager_exact += 1
        batch = prog.prepare(raw, key if padded else None)
        if not (self.use_graphs and padded) or self.seen.get(key, 0) == 0:
            # first sight of a bucket (doubles as the capture's warm-up; a real training step), CPU, or a bucket the padded
            # step does not cover: the eager phased step
            self.seen[key] = 1
            out = None
            for k in range(nph):
                r = prog.phase(k, batch, self.scale)
                out = r if k == 0 else out
                if reduce:
                    self._reduce(k)
            return out
        # second sight: capture the phases on this batch's tensors (they become the graphs' static inputs), then replay
        torch.cuda.synchronize()
        if self.pool is None:
            self.pool = torch.cuda.graph_pool_handle()
        gs, out = [], None
        for k in range(nph):
            g = graphs.new_graph()
            with torch.cuda.graph(g, pool=self.pool, capture_error_mode="thread_local"):
                r = prog.phase(k, batch, self.scale)
            graphs.audit(g, "composed step, phase %d" % k)
            out = r if k == 0 else out
            gs.append(g)
        for k, g in enumerate(gs):
            g.replay()
            if reduce:
                self._reduce(k)
        self.captures += 1
        self.cache[key] = dict(graphs=gs, static=batch, out=out)
        if len(self.cache) > self.max_graphs:
            self.cache.popitem(last=False)
            self.evictions += 1
        return out

    def step(self, raw):
        """the plain loop (one batch per optimizer step, loss averaged over the ranks as DistributedDataParallel does):
        zero, micro-step with the all-reduces under the backward phases, optimizer"""
        self.flat.zero_grad()
        out = self.micro_step(raw, 1.0 / self.world, reduce=True)
        self.wait()
        self.opt.step()
        return out[0]

    def stats(self):
        n = self.hits + self.misses
        return dict(steps=n, hits=self.hits, hit_rate=(self.hits / n if n else 0.0), captures=self.captures,
                    evictions=self.evictions, graphs=len(self.cache), eager_exact_shape=self.eager_exact,
                    phases=len(self.program.ranges), ranks=self.world)


class EpochRunner:
    """espnet2 Trainer.train_one_epoch / validate_one_epoch semantics for one process per GPU
    (reference: espnet2/train/trainer.py:325-495,497-539; recursive_average, torch_utils/recursive_op.py:14-53).

    Per micro-step the reference issues EIGHT small collectives (iterator-stop flag, five weighted statistics, the
    weight sum) around a blocking forward; here ONE 8-float vector per micro-step carries all of it:

        [ stop flag of THIS rank for the NEXT batch, weight of the NEXT batch,
          weight, weight * loss, weight * loss_att, weight * loss_ctc, weight * acc of the batch just finished, spare ]

    The iterator is read one batch ahead, so the vector of micro-step k is known before its forward starts; it is
    all-reduced (SUM) asynchronously under the forward pass, and what the host needs of it before the NEXT step (the
    stop flag) has long arrived by then.  On the device the all-reduced weight sum W gives the backward scale
    w / (W * accum_grad) without a host sync: with SUM all-reduce of the gradients this is exactly the reference's
    loss * weight / W * world_size / accum_grad under DistributedDataParallel's 1 / world_size averaging.

    Gradients accumulate in the flat arena over `accum_grad` micro-steps and are all-reduced once at the boundary
    (linear: same sum as the reference's per-micro-step DDP reduction), then gradient noise (optional), global-norm
    clipping, the non-finite skip, Adam and the scheduler run on the device (NoamAdam.step), then the arena is zeroed.

    The model-specific part is injected: `forward(batch) -> (loss 0-dim tensor, {name: 0-dim tensor}, weight float)`;
    `backward(loss, scale 0-dim tensor)`.  Defaults drive the espnet1 E2E / espnet2 ESPnetASRModel shells."""

    NSTAT = 8

    def __init__(self, model, flat, opt, accum_grad=1, grad_noise=False, group=None, forward=None, backward=None,
                 reduce_grads=None, pre_step=None, bucket_mb=128.0, composed=None):
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.distributed = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if self.distributed else 1
        self.model, self.flat, self.opt = model, flat, opt
        self.accum_grad, self.grad_noise = int(accum_grad), bool(grad_noise)
        # composed = a ComposedStep: training micro-steps run through it (shape buckets, hipGraph replay, phased backward with the
        # gradient all-reduce of each arena range under the next phase) instead of forward() / backward() / reduce_grads()
        self.composed = composed
        self.forward = forward or self._forward_default
        self.backward = backward or (lambda loss, scale: loss.backward(scale))
        self.reduce_grads = reduce_grads or self._reduce_default
        self.pre_step = pre_step or (lambda: ops.rng_advance(flat.data.device))     # new dropout masks per micro-step
        self.cap = max(1, int(bucket_mb * 1024 * 1024 // 4))
        self.dev = flat.data.device
        # reporter.get_total_count() (espnet2/train/reporter.py:154-165): MICRO-steps registered so far by the "train"
        # sub-reporter, the current one included - incremented before forward, carried across epochs (the caller may
        # restore it from a checkpoint); drives the gradient-noise decay (trainer.py:420-427)
        self.total_count = 0
        self.history = []             # averaged statistics of every finished micro-step (device tensors, read lazily)

    # ---- defaults for the model shells of this package ----
    def _forward_default(self, batch):
        m = self.model
        if isinstance(batch, dict) and "speech" in batch:          # espnet2 ESPnetASRModel
            loss, stats, weight = m(**batch)
            return loss.reshape(()), {k: v.reshape(()) for k, v in stats.items() if v is not None and k != "loss"}, float(weight)
        loss = m.forward_core(batch) if isinstance(batch, dict) else m(*batch[:3])      # (xs_pad, ilens, ys_pad[, olens])
        stats = {}
        if getattr(m, "_loss_att_t", None) is not None:
            stats["loss_att"] = m._loss_att_t
        if getattr(m, "_loss_ctc_t", None) is not None:
            stats["loss_ctc"] = m._loss_ctc_t
        if getattr(m, "_acc_t", None) is not None:
            stats["acc"] = m._acc_t
        B = batch["B"] if isinstance(batch, dict) else batch[0].shape[0]
        return loss, stats, float(B)

    def _reduce_default(self):
        """the step WITHOUT a ComposedStep: the whole arena after backward (stream-ordered, exposed: nothing overlaps it).  The
        overlapped form is `composed=` (phased backward, each arena range reduced under the next phase)."""
        if not self.distributed or self.world == 1:
            return
        g = self.flat.grad
        works = [self.dist.all_reduce(g[s:s + self.cap], op=self.dist.ReduceOp.SUM, group=self.group, async_op=True)
                 for s in range(0, g.numel(), self.cap)]
        for w in works:
            w.wait()

    # ---- the fused control / statistics vector ----
    def _send(self, stop, weight_next, pending):
        """all-reduce (SUM) [stop, weight_next, *pending[2:]] off the critical path: on a side stream (device runs) the
        collective and the 2-float device-to-host copy wait only for the kernels that produced `pending`, not for
        whatever the main stream has queued since; -> a ticket for _recv()"""
        if self.dev.type != "cuda":
            vec = pending.clone()
            vec[0], vec[1] = stop, weight_next
            if self.distributed and self.world > 1:
                self.dist.all_reduce(vec, op=self.dist.ReduceOp.SUM, group=self.group)
            return (vec, None, None)
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(device=self.dev)
        side = self._side
        side.wait_stream(torch.cuda.current_stream(self.dev))      # `pending` was written on the main stream
        pending.record_stream(side)
        with torch.cuda.stream(side):
            vec = pending.clone()
            vec[:2] = torch.tensor([stop, weight_next], dtype=vec.dtype).to(self.dev, non_blocking=True)
            if self.distributed and self.world > 1:
                self.dist.all_reduce(vec, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True).wait()
            host = torch.empty(2, dtype=vec.dtype, pin_memory=True)
            host.copy_(vec[:2], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(side)
        return (vec, host, ev)

    @staticmethod
    def _recv(ticket):
        """-> (stop-flag sum, weight sum) as python floats; the only host wait of a micro-step"""
        vec, host, ev = ticket
        if ev is None:
            return vec[:2].tolist()
        ev.synchronize()
        return host.tolist()

    @staticmethod
    def _next(it):
        try:
            return next(it)
        except StopIteration:
            return None

    @staticmethod
    def _weight_of(batch):
        if batch is None:
            return 0.0
        if isinstance(batch, dict):
            return float(batch["speech"].shape[0] if "speech" in batch else batch["B"])
        return float(batch[0].shape[0])

    def _run(self, iterator, train):
        it = iter(iterator)
        dev = self.dev
        keys = ("loss", "loss_att", "loss_ctc", "acc")
        zero = torch.zeros(self.NSTAT, device=dev)
        cur = self._next(it)
        ticket = self._send(1.0 if cur is None else 0.0, self._weight_of(cur), zero)       # about the first batch
        tickets = []
        iiter = 0
        skipped0 = self.opt.stats()["skipped"] if train else 0
        if train:
            self.flat.zero_grad()
        pending = zero
        while True:
            stop, wsum = self._recv(ticket)
            if stop > 0:                                # some rank ran out of data: every rank stops here (trainer.py:369-372)
                break
            batch, cur = cur, self._next(it)            # read one batch ahead
            # about the NEXT batch + the statistics of the PREVIOUS micro-step; travels under this step's kernels
            ticket = self._send(1.0 if cur is None else 0.0, self._weight_of(cur), pending)
            if iiter > 0:
                tickets.append(ticket)
            iiter += 1
            w = self._weight_of(batch)
            if train and self.composed is not None:
                self.total_count += 1
                loss, stats = self.composed.micro_step(batch, w / (wsum * self.accum_grad), reduce=(iiter % self.accum_grad == 0))
            elif train:
                self.total_count += 1
                self.pre_step()
                loss, stats, _w = self.forward(batch)
                ops.wgrad_group_begin()
                try:
                    self.backward(loss, torch.full((), w / (wsum * self.accum_grad), device=dev, dtype=loss.dtype))
                finally:
                    ops.wgrad_group_end()
                ops.wgrad_join()
            else:
                with torch.no_grad():
                    loss, stats, _w = self.forward(batch)
            parts = [torch.zeros((), device=dev)] * 2 + [torch.full((), w, device=dev), loss.detach().float() * w]
            parts += [(stats[k].detach().float() * w if k in stats else torch.zeros((), device=dev)) for k in keys[1:]]
            pending = torch.stack(parts + [torch.zeros((), device=dev)])
            if train and iiter % self.accum_grad == 0:
                if self.composed is not None:
                    self.composed.wait()             # the range all-reduces were issued under the backward phases
                else:
                    self.reduce_grads()
                if self.grad_noise:
                    # add_gradient_noise.py:4-31 with the reference's call-site constants (trainer.py:420-427)
                    ops.add_gradient_noise(self.flat.grad, 1.0 / ((self.total_count // 100) + 1) ** 0.55)
                self.opt.step()                          # clip, non-finite skip, Adam, scheduler: all on the device
                self.flat.zero_grad()
        if iiter > 0:                                    # statistics of the last micro-step: one more (blocking) exchange
            tickets.append(self._send(0.0, 0.0, pending))
        if dev.type == "cuda":
            torch.cuda.current_stream(dev).wait_stream(self._side)
        self.history = [{k: t[0][3 + i] / t[0][2].clamp_min(1e-30) for i, k in enumerate(keys)} | {"weight": t[0][2]}
                        for t in tickets]
        self._check_persistent_launches()
        if not train:
            return None
        nsteps = iiter // self.accum_grad
        return nsteps == 0 or (self.opt.stats()["skipped"] - skipped0) >= nsteps

    def _check_persistent_launches(self):
        """once per epoch, where the host waits anyway: did a persistent LSTM launch give up on a hand-off (its outputs are NaN,
        the optimizer skipped those steps)?  Say so, and run the per-step kernels from here on - a skipped step must not be
        the only trace (csrc/lstm_seq.hip: residency of one workgroup per CU is the caller's side of the contract)."""
        if self.dev.type != "cuda":
            return
        code = ops.lstm_seq_sticky_status(clear=True)
        self.lstm_seq_failures = getattr(self, "lstm_seq_failures", 0) + (1 if code else 0)
        if code:
            import logging
            logging.getLogger(__name__).error(
                "a persistent LSTM launch gave up waiting for a hand-off (code 0x%x): its steps were skipped as non-finite; "
                "switching to the per-step LSTM kernels (another kernel was holding compute units?)", code)
            ops.LSTM_PERSISTENT = False

    def train_one_epoch(self, iterator):
        """-> all_steps_are_invalid (trainer.py:495)"""
        self.model.train()
        self.history = []
        return self._run(iterator, True)

    def validate_one_epoch(self, iterator):
        """no-grad pass in eval mode; -> list of per-batch averaged statistics (trainer.py:497-539)"""
        self.model.eval()
        self.history = []
        self._run(iterator, False)
        return self.history

    def averaged(self):
        """weighted mean of the recorded statistics over the epoch, as python floats (one host copy)"""
        if not self.history:
            return {}
        keys = [k for k in self.history[0] if k != "weight"]
        tab = torch.stack([torch.stack([h[k] for k in keys] + [h["weight"]]) for h in self.history]).tolist()
        wsum = sum(r[-1] for r in tab)
        return {k: sum(r[i] * r[-1] for r in tab) / wsum for i, k in enumerate(keys)}


def init_distributed():
    """reference: espnet2/train/distributed_utils.py:28-107 (env:// rendezvous, one process per GPU)."""
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal knobs (a multi-rank run on a one-GPU box): EAMD_FORCE_DEVICE pins every rank to one device,
    # EAMD_DIST_BACKEND=gloo replaces RCCL, which refuses two ranks on the same GPU
    if os.environ.get("EAMD_FORCE_DEVICE") is not None:
        local_rank = int(os.environ["EAMD_FORCE_DEVICE"])
    if world > 1 and not dist.is_initialized():
        backend = os.environ.get("EAMD_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, init_method="env://", rank=rank, world_size=world)
    elif torch.cuda.is_available():
        torch.cuda.set_device(local_rank)
    return rank, local_rank, world


def shard_batch(batch_items, rank, world):
    """reference: espnet2/tasks/abs_task.py:1445 (batch[rank::world_size])"""
    return batch_items[rank::world]
