"""Flat parameter storage, fused DiffGrad and fused EMA (hg_diffgrad_step / hg_ema_update).

MI355X-first layout: every optimizer owns ONE contiguous fp32 parameter buffer and ONE contiguous
gradient buffer (parameters / .grad are views into them).  One kernel launch updates all parameters,
one (or a few large) collectives reduce all gradients, EMA is one launch -- instead of ~12 aten
launches per parameter tensor (reference: torch_optimizer.DiffGrad over ~150 tensors, and the
per-tensor EMA loop of histoGAN/histoGAN.py:698-707).
"""
import ctypes

import torch

from ._lib import check, lib, on_device, raw_stream
from .conv import weights_changed


def _st(t):
    return raw_stream(t.device)


def conv_first(params):
    """`params` with the convolution weights (4-d tensors) first, order otherwise kept: the flat buffer then starts with one
    contiguous region that holds every convolution weight (FlatParams.n_conv elements) -- the region whose gradients are final
    when the last convolution's backward has been enqueued, long before the mapping networks' (DiffGrad.step_early)."""
    params = list(params)
    return [p for p in params if p.dim() == 4] + [p for p in params if p.dim() != 4]


class FlatParams:
    """Re-home `params` (already on their final device) into one flat buffer; optionally with grads."""

    def __init__(self, params, with_grad=True):
        self.params = []
        seen = set()
        for p in params:
            if id(p) not in seen:
                seen.add(id(p))
                self.params.append(p)
        if not self.params:
            raise ValueError('no parameters')
        self.n_conv = 0       # elements of the leading run of 4-d parameters
        for p in self.params:
            if p.dim() != 4:
                break
            self.n_conv += p.numel()
        dev = self.params[0].device
        self.numel = sum(p.numel() for p in self.params)
        self.data = torch.empty(self.numel, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(self.numel, dtype=torch.float32, device=dev) if with_grad else None
        off = 0
        for p in self.params:
            n = p.numel()
            view = self.data[off:off + n].view(p.shape)
            view.copy_(p.data)
            p.data = view
            if with_grad:
                p.grad = self.grad[off:off + n].view(p.shape)
            off += n
        # persistent views of every parameter's slice of the gradient buffer (gather() runs twice per step over ~150
        # parameters: slicing + viewing them anew cost the host ~1.5 ms per call)
        self._gviews = []
        if with_grad:
            off = 0
            for p in self.params:
                n = p.numel()
                self._gviews.append(self.grad[off:off + n].view(p.shape))
                off += n
        # convolution weights may get their gradient written straight into self.grad (conv._direct_wgrad)
        self.direct_ok = False
        self.direct_written = set()
        if with_grad:
            from .conv import register_grad_slots
            register_grad_slots(self)

    def conv_region_final(self):
        """True when every parameter of the leading convolution-weight region (n_conv elements) has its gradient of this step
        in its flat slot, written directly (conv._direct_wgrad, the demodulation term, gfused's to-RGB write) and by nothing
        else -- the condition under which that region may be reduced / updated before gather()."""
        if self.n_conv <= 0 or not self.direct_ok or self.grad is None:
            return False
        base, off = self.grad.data_ptr(), 0
        for p in self.params:
            if off >= self.n_conv:
                break
            if base + 4 * off not in self.direct_written or p.grad is not None:
                return False
            off += p.numel()
        return True

    def zero_grad(self):
        """Drop the parameter gradients.  With `.grad = None` autograd's AccumulateGrad hands over the incoming gradient
        tensor instead of launching one `grad += new` kernel per parameter into a zeroed buffer; `gather()` then
        copies all of them into the flat gradient buffer with a few multi-tensor launches."""
        for p in self.params:
            p.grad = None
        self._gathered = False
        self.direct_written = set()
        self.direct_ok = True

    def gather(self):
        """Make `self.grad` (and every `p.grad`, re-pointed to its slice) hold the gradients of the last backward
        passes; parameters that received none get zeros.  Idempotent."""
        if getattr(self, '_gathered', True):
            return
        self.direct_ok = False
        if self.direct_written and self.grad.is_cuda:   # weight gradients written on the side stream: order them before any reader
            from .conv import GRAPH_WGRAD_INLINE, side_stream
            if not (GRAPH_WGRAD_INLINE and torch.cuda.is_current_stream_capturing()):   # (captured: written on this branch)
                torch.cuda.current_stream(self.grad.device).wait_stream(side_stream(self.grad.device))
        dst, src, missing, both = [], [], [], []
        off = 0
        base = self.grad.data_ptr()
        for p, v in zip(self.params, self._gviews):
            n = p.numel()
            pg = p.grad
            if base + 4 * off in self.direct_written:
                if pg is not None and pg is not v and pg.data_ptr() != base + 4 * off:
                    both.append((v, pg))           # autograd ALSO delivered a part (a pass that recorded a graph)
            elif pg is None:
                missing.append(v)
            elif pg is not v and pg.data_ptr() != base + 4 * off:
                dst.append(v)
                src.append(pg)
            if pg is not v:
                p.grad = v
            off += n
        if dst:
            torch._foreach_copy_(dst, src)
        if missing:
            torch._foreach_zero_(missing)
        if both:
            torch._foreach_add_([v for v, _ in both], [g for _, g in both])
        self._gathered = True


class DiffGrad:
    """DiffGrad (Dubey et al.; torch_optimizer.DiffGrad as used at histoGAN/histoGAN.py:670-671) over a
    FlatParams: p -= lr*sqrt(1-b2^t)/(1-b1^t) * (m * dfc) / (sqrt(v)+eps), dfc = sigmoid(|g_prev - g|)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.flat = params if isinstance(params, FlatParams) else FlatParams(params)
        self.lr, self.betas, self.eps = lr, betas, eps
        self.step_count = 0
        self.graph_mode = False
        self._step_size_dev = None
        z = lambda: torch.zeros_like(self.flat.data)
        self.exp_avg, self.exp_avg_sq, self.previous_grad = z(), z(), z()
        self.param_groups = [{'params': self.flat.params, 'lr': lr, 'betas': betas, 'eps': eps}]

    def zero_grad(self, set_to_none=False):
        self.flat.zero_grad()

    def prepare_replay(self):
        """Graph mode (the train step captured as a hipGraph): advance the step counter on the host and refresh the
        bias-corrected step size the captured launch reads from device memory.  Call once per replay, before it."""
        self.step_count += 1
        if self._step_size_dev is None:
            self._step_size_dev = torch.zeros((), dtype=torch.float32, device=self.flat.data.device)
        lr = self.param_groups[0]['lr']
        self._step_size_dev.fill_(lib.hg_diffgrad_step_size(float(lr), float(self.betas[0]), float(self.betas[1]),
                                                            self.step_count))

    def step_buckets(self, reducer):
        """The update bucket by bucket behind a bucketed gradient all-reduce (ddp.GradAllReduce, already started): bucket
        i's parameters are updated as soon as ITS collective is done, while the later buckets are still on the wire.
        Elementwise over the flat buffer: the same update as one launch."""
        f = self.flat
        f.gather()
        if not f.data.is_cuda:
            raise RuntimeError('DiffGrad: parameters are not on a GPU; no CPU implementation')
        if self.graph_mode:
            raise RuntimeError('DiffGrad.step_buckets: not available inside a captured graph')
        self.step_count += 1
        lr = self.param_groups[0]['lr']
        with on_device(f.data.device):
            for i, (lo, hi) in enumerate(reducer.ranges):
                reducer.wait(i)
                o = 4 * lo
                check(lib.hg_diffgrad_step(f.data.data_ptr() + o, f.grad.data_ptr() + o, self.exp_avg.data_ptr() + o,
                                           self.exp_avg_sq.data_ptr() + o, self.previous_grad.data_ptr() + o, hi - lo,
                                           float(lr), float(self.betas[0]), float(self.betas[1]), float(self.eps),
                                           self.step_count, _st(f.data)), 'hg_diffgrad_step')
        reducer.finish()
        weights_changed(f.data)

    def step_early(self, stream):
        """The update of the flat buffer's leading convolution-weight region NOW, on `stream` -- legal as soon as every one of
        those weights has its final gradient in its flat slot (all written directly: conv._direct_wgrad, the demodulation
        term, gfused's to-RGB slot write), i.e. when the generator's fused backward node returns, while the mapping
        networks' backward (a latency-bound chain of ~70 small launches) is still to run.  `step()` then updates the rest.
        Returns False (nothing done) when a slot of the region was not written directly."""
        f = self.flat
        hi = f.n_conv
        if hi <= 0 or self.graph_mode or not f.data.is_cuda or getattr(self, '_early', None) is not None \
                or not f.conv_region_final():
            return False
        from .conv import side_stream
        dev = f.data.device
        stream.wait_stream(side_stream(dev))                       # the weight gradients and demodulation terms
        stream.wait_event(torch.cuda.current_stream(dev).record_event())   # slots written on the calling stream (to-RGB)
        lr = self.param_groups[0]['lr']
        with torch.cuda.stream(stream), on_device(dev):
            check(lib.hg_diffgrad_step(f.data.data_ptr(), f.grad.data_ptr(), self.exp_avg.data_ptr(),
                                       self.exp_avg_sq.data_ptr(), self.previous_grad.data_ptr(), hi, float(lr),
                                       float(self.betas[0]), float(self.betas[1]), float(self.eps), self.step_count + 1,
                                       _st(f.data)), 'hg_diffgrad_step')
        self._early = (hi, stream)
        weights_changed(f.data)
        return True

    def step(self):
        f = self.flat
        f.gather()
        if not f.data.is_cuda:
            raise RuntimeError('DiffGrad: parameters are not on a GPU; no CPU implementation')
        early = self.__dict__.pop('_early', None)
        if early is not None:         # the convolution-weight region was updated by step_early(): the rest now, same step number
            lo, stream = early
            self.step_count += 1
            lr = self.param_groups[0]['lr']
            with on_device(f.data.device):
                if f.numel > lo:
                    o = 4 * lo
                    check(lib.hg_diffgrad_step(f.data.data_ptr() + o, f.grad.data_ptr() + o, self.exp_avg.data_ptr() + o,
                                               self.exp_avg_sq.data_ptr() + o, self.previous_grad.data_ptr() + o, f.numel - lo,
                                               float(lr), float(self.betas[0]), float(self.betas[1]), float(self.eps),
                                               self.step_count, _st(f.data)), 'hg_diffgrad_step')
            torch.cuda.current_stream(f.data.device).wait_stream(stream)
            return                    # (weights_changed() ran with the early part: the packed operands may already be rebuilt)
        if self.graph_mode:           # being captured: step size from device memory, counter advanced by prepare_replay()
            with on_device(f.data.device):
                check(lib.hg_diffgrad_step_dev(f.data.data_ptr(), f.grad.data_ptr(), self.exp_avg.data_ptr(),
                                               self.exp_avg_sq.data_ptr(), self.previous_grad.data_ptr(), f.numel,
                                               self._step_size_dev.data_ptr(), float(self.betas[0]),
                                               float(self.betas[1]), float(self.eps), _st(f.data)),
                      'hg_diffgrad_step_dev')
            weights_changed(f.data)
            return
        self.step_count += 1
        lr = self.param_groups[0]['lr']
        with on_device(f.data.device):
            check(lib.hg_diffgrad_step(f.data.data_ptr(), f.grad.data_ptr(), self.exp_avg.data_ptr(),
                                       self.exp_avg_sq.data_ptr(), self.previous_grad.data_ptr(), f.numel,
                                       float(lr), float(self.betas[0]), float(self.betas[1]), float(self.eps),
                                       self.step_count, _st(f.data)), 'hg_diffgrad_step')
        weights_changed(f.data)


def ema_update(ma_flat, cur_flat, beta):
    """ma = beta*ma + (1-beta)*cur over two FlatParams with identical layout (HistoGAN.EMA)."""
    if ma_flat.numel != cur_flat.numel:
        raise ValueError('EMA buffers differ in size')
    with on_device(ma_flat.data.device):
        check(lib.hg_ema_update(ma_flat.data.data_ptr(), cur_flat.data.data_ptr(), ma_flat.numel, float(beta),
                                _st(ma_flat.data)), 'hg_ema_update')
    weights_changed(ma_flat.data)
