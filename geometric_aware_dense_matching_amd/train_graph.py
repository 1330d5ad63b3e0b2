"""One training iteration (pyramid + forward + losses + backward + Adam) as ONE hipGraph launch.

The reference's training loop (/root/reference/train_lm.py:224-296) enqueues every kernel of every iteration from Python.  On MI355X the
iteration is ~1 900 kernels of 5-100 us each, so the eager loop is bound by the host's launch rate, not by the GPU.  All shapes of an
iteration are fixed by (batch, n_points, n_model) and nothing in it depends on values read back to the host, so the whole iteration
is captured once and replayed: one host call per iteration, inputs copied into the capture's static buffers.

What stays outside the graph (exactly as in the reference's loop): the data loader, the learning-rate and BN-momentum schedulers
and checkpointing.  The learning rate lives in a device scalar that the scheduler fills (`torch.optim.Adam(capturable=True)`); a
BatchNorm momentum is a kernel argument, so a momentum change (every `decay_step` samples, train_lm.py:421-427) re-captures.
"""
import torch

from . import ops, train_lm

_INPUT_KEYS = None          # every tensor entry of the collated batch is an input


def make_capturable(optimizer):
    """Adam with its step counter and learning rate on the device, so `optimizer.step()` can live inside a capture."""
    for g in optimizer.param_groups:
        g["capturable"] = True
        if not torch.is_tensor(g["lr"]):
            dev = g["params"][0].device
            g["lr"] = torch.tensor(float(g["lr"]), dtype=torch.float32, device=dev)
        if "initial_lr" in g and torch.is_tensor(g["initial_lr"]):
            g["initial_lr"] = float(g["initial_lr"])
    for p, st in optimizer.state.items():
        if "step" in st and torch.is_tensor(st["step"]) and st["step"].device != p.device:
            st["step"] = st["step"].to(p.device, torch.float32)
    return optimizer


def _bn_momenta(model):
    return tuple(m.momentum for m in model.modules() if isinstance(m, torch.nn.modules.batchnorm._BatchNorm))


class GraphedTrainStep:
    """step(batch) -> {"loss", "seg_loss", "match_loss"} (device scalars owned by the capture; read or clone before the next step).

    model: the (unwrapped, single-process) training module; optimizer: its Adam.  The first call warms up `warmup` eager iterations
    on a side stream (MIOpen picks its algorithms, the allocator reaches steady state, Adam's state exists), then captures.  Those
    warm-up iterations ARE training iterations -- their updates are kept and their losses returned -- so a run of n calls is n
    optimiser steps whichever path ran them.
    """

    def __init__(self, model, optimizer, device, warmup=3):
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            raise RuntimeError("GraphedTrainStep captures a single-process step; multi-rank training runs the eager loop")
        self.model, self.optimizer, self.device = model, make_capturable(optimizer), device
        self.warmup = warmup
        self.calls = 0
        self.graph = None
        self.static_in = None
        self.static_out = None
        self.momenta = None
        self.captures = 0
        self.stream = torch.cuda.Stream(device)
        self.pool = ops.BufferPool()               # scratch buffers of the captured iteration (ops.buffer_pool)

    # ---- one eager iteration on static buffers (what the capture records) -------------------------------------------------------
    def _iteration(self):
        with ops.buffer_pool(self.pool):
            out, _ = train_lm.model_fn_dec(self.model, dict(self.static_in), self.device)
            out["loss"].backward()
        self.optimizer.step()
        return {k: torch.as_tensor(out[k], device=self.device).detach().float() for k in ("loss", "seg_loss", "match_loss")}

    def _load(self, batch):
        cu = train_lm.to_device(batch, self.device)
        tensors = {k: v for k, v in cu.items() if torch.is_tensor(v)}
        if self.static_in is None:
            self.static_in = {k: v.clone() for k, v in tensors.items()}
            self.extra = {k: v for k, v in cu.items() if not torch.is_tensor(v)}
            return
        if set(tensors) != set(self.static_in):
            raise ValueError("batch keys changed: %s vs %s" % (sorted(tensors), sorted(self.static_in)))
        for k, v in tensors.items():
            s = self.static_in[k]
            if v.shape != s.shape or v.dtype != s.dtype:
                raise ValueError("batch entry %r changed shape/dtype: %s %s vs captured %s %s (use drop_last=True)"
                                 % (k, tuple(v.shape), v.dtype, tuple(s.shape), s.dtype))
            s.copy_(v, non_blocking=True)

    def _capture(self):
        self.graph = torch.cuda.CUDAGraph()
        self.optimizer.zero_grad(set_to_none=True)              # the capture's backward allocates the grads from the graph's pool
        with torch.cuda.graph(self.graph, stream=self.stream):
            self.static_out = self._iteration()
        self.momenta = _bn_momenta(self.model)
        self.captures += 1

    def step(self, batch):
        self.model.train()
        self._load(batch)
        self.calls += 1
        if self.calls <= self.warmup:
            cur = torch.cuda.current_stream(self.device)
            self.stream.wait_stream(cur)
            with torch.cuda.stream(self.stream):
                self.optimizer.zero_grad(set_to_none=True)
                out = self._iteration()
            cur.wait_stream(self.stream)
            return out
        if self.graph is None or self.momenta != _bn_momenta(self.model):
            torch.cuda.synchronize(self.device)
            # capturing an iteration does not run it: capture, then replay below
            self.graph = None
            self._capture()
        self.graph.replay()
        return self.static_out
