"""Data parallelism for the HistoGAN train step: one process per GPU, torch.distributed
(backend 'nccl' == RCCL over xGMI on ROCm; 'gloo' for the CPU tests).

The reference is single-GPU (histoGAN.py:242,268); these semantics are new:
* parameters replicated, broadcast from rank 0 at init / after load;
* each rank draws its own latents / noise / data shard; global batch = sum of rank batches;
* gradients averaged with ONE large all-reduce per optimizer over its flat gradient buffer
  (399 MB for G+S+H, 364 MB for D at 256^2/cap16) -- on the fully connected 8-GPU xGMI mesh a few
  large collectives use all 7 links per GPU, where many small per-tensor ones would be latency-bound;
* the D-gradient all-reduce is launched asynchronously right after the D backward and overlaps the
  generator forward of the G phase (which does not touch D); it is waited for before D_opt.step();
* that generator forward runs on a second stream beside the D phase whenever every rank owns its GPU
  (ranks_share_a_device: two processes on one GPU oversubscribe its hardware queues);
* scalar state that steers control flow (NaN flag, pl_mean) is all-reduced so ranks never diverge.
"""
import os

import torch
import torch.distributed as dist

# HG_DIST_FORCE=1: take the data-parallel code paths (broadcasts, gradient / statistics / Hellinger all-reduces) whenever
# a process group exists, even at world size 1 -- lets a 1-GPU box run the whole step through RCCL (tests/test_bench_gpu.py)
FORCE = os.environ.get('HG_DIST_FORCE', '0') == '1'


def is_dist():
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or FORCE)


def world_size():
    return dist.get_world_size() if is_dist() else 1


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def _device_identity(device):
    """(host, physical device) of this rank: PCI address where torch exposes it (two ranks can both call their GPU 'cuda:0')."""
    import socket
    device = torch.device(device)
    if device.type != 'cuda':
        return (socket.gethostname(), 'cpu', os.getpid())
    pr = torch.cuda.get_device_properties(device)
    index = device.index if device.index is not None else torch.cuda.current_device()
    # The PHYSICAL identity where torch / the driver exposes one: the uuid, or the PCI address.  Visibility masks and the logical
    # index are NOT part of it then -- two ranks can reach one GPU through different masks (HIP_VISIBLE_DEVICES=0 for one rank,
    # unset for the other; masks '0,1' and '0', both at index 0) and must still compare equal.  Only when neither is
    # available does (masks, logical index) stand in.
    uuid = str(getattr(pr, 'uuid', '') or '')
    pci = (getattr(pr, 'pci_domain_id', -1), getattr(pr, 'pci_bus_id', -1), getattr(pr, 'pci_device_id', -1))
    if uuid and set(uuid) - set('0-'):
        ident = ('uuid', uuid)
    elif pci[1] >= 0:
        ident = ('pci',) + pci
    else:
        ident = ('mask', os.environ.get('HIP_VISIBLE_DEVICES', ''), os.environ.get('CUDA_VISIBLE_DEVICES', ''),
                 os.environ.get('ROCR_VISIBLE_DEVICES', ''), index)
    return (socket.gethostname(), 'cuda', str(ident))


def _any_shared(identities):
    return len(set(identities)) < len(identities)


_shared = {}


def ranks_share_a_device(device):
    """True when two ranks of the process group sit on ONE physical GPU (the two-gloo-ranks-on-one-GPU test set-up; never on
    a node run as one process per GPU).  A COLLECTIVE on first use per device (all_gather_object): every rank must reach it
    at the same point -- the trainer asks in its first data-parallel step.  Matters because a shared GPU carries the hardware
    queues of both processes: with more than ~4 in total the step's cross-stream waits cost scheduler time slices
    (3752 ms per step with 2 x 4 queues, 272 ms with 2 x 2)."""
    key = str(device)
    if key not in _shared:
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            _shared[key] = False
        else:
            seen = [None] * dist.get_world_size()
            d = torch.device(device)
            if d.type == 'cuda' and torch.cuda.is_available():   # (RCCL moves the pickled object through the CURRENT device: make it this rank's)
                with torch.cuda.device(d):
                    dist.all_gather_object(seen, _device_identity(d))
            else:
                dist.all_gather_object(seen, _device_identity(d))
            _shared[key] = _any_shared([tuple(x) if isinstance(x, (list, tuple)) else x for x in seen])
    return _shared[key]


def broadcast_flat(flat, src=0):
    if is_dist():
        dist.broadcast(flat.data, src)


def broadcast_buffers(module, src=0):
    """Module buffers (the VectorQuantize codebooks of `fq_layers`) are drawn per rank at construction: replicas start
    from rank `src`'s copy, like the parameters."""
    if is_dist():
        for b in module.buffers():
            dist.broadcast(b.data, src)


def _avg_in_collective():
    """RCCL averages inside the collective (ReduceOp.AVG: no separate pass over the 364 + 399 MB gradient buffers);
    gloo has no AVG, there the buffer is pre-scaled and summed (CPU tests, two-ranks-on-one-GPU tests)."""
    return dist.get_backend() == 'nccl'


# Buckets per gradient buffer under data parallelism: the all-reduce of bucket i+1 runs (on RCCL's stream) while the
# optimizer already updates the parameters of bucket i -- DiffGrad is elementwise over the flat buffer, so applying it
# bucket by bucket is the same update (HG_DDP_BUCKETS=1: one collective, one optimizer launch).
BUCKETS = max(1, int(os.environ.get('HG_DDP_BUCKETS', '4')))


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class GradAllReduce:
    """Averaging all-reduce of a flat gradient buffer in `chunks` contiguous buckets, asynchronous: `start()` launches all
    of them in order, `wait(i)` makes the current stream wait for bucket i only (`ranges[i]` = its element range),
    `finish()` for all."""

    def __init__(self, flat, chunks=1):
        self.flat = flat
        self.chunks = max(1, int(chunks))
        self._work = []
        n, c = flat.numel, self.chunks
        step = -(-n // c)
        step = -(-step // 1024) * 1024                      # bucket boundaries on 4 KB
        self.ranges = [(lo, min(n, lo + step)) for lo in range(0, n, step)]
        self._full_ranges = list(self.ranges)

    def _split(self, lo, hi, c):
        step = -(-(hi - lo) // max(1, c))
        step = max(1024, -(-step // 1024) * 1024)           # bucket boundaries on 4 KB
        return [(a, min(hi, a + step)) for a in range(lo, hi, step)]

    def start_early(self, stream):
        """The all-reduce of the flat buffer's leading convolution-weight region NOW, from `stream`: legal as soon as every one
        of those weights has its final gradient in its slot (FlatParams.conv_region_final: when the generator's fused backward
        node returns), while the rest of the backward -- style projections, mapping networks -- is still to run.  83 of the
        generator side's 99.8 M parameters at C3: the collective that was fully exposed behind the last backward kernel
         now runs under that tail and under the remaining small buckets.  `start()` then launches only the rest.
        Returns False (nothing done) when not distributed or a slot of the region was not written directly."""
        f = self.flat
        if not is_dist() or self._work or not f.conv_region_final():
            return False
        from .conv import side_stream
        dev = f.grad.device
        if dev.type == 'cuda':
            stream.wait_stream(side_stream(dev))                                # the weight gradients and demodulation terms
            stream.wait_event(torch.cuda.current_stream(dev).record_event())    # slots written on the calling stream (to-RGB)
        n, hi = f.numel, f.n_conv
        early = self._split(0, hi, self.chunks)
        self.ranges = early + (self._split(hi, n, max(1, self.chunks // 2)) if n > hi else [])
        g = f.grad
        avg = _avg_in_collective()
        ctx = torch.cuda.stream(stream) if dev.type == 'cuda' else _NullCtx()
        with ctx:
            if not avg:
                g[:hi].mul_(1.0 / world_size())
            for lo, up in early:
                self._work.append(dist.all_reduce(g[lo:up], op=dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM, async_op=True))
        self._early = len(early)
        return True

    def start(self):
        self.flat.gather()
        if not is_dist():
            return
        early = self.__dict__.pop('_early', 0)
        if self._work and not early:   # handles of a step that did not get to its wait (an exception in between): drain them
            self.finish()
        g = self.flat.grad
        if not early:
            self.ranges = list(self._full_ranges)
        lo0 = self.ranges[early][0] if early < len(self.ranges) else self.flat.numel
        if _avg_in_collective():
            op = dist.ReduceOp.AVG
        else:
            op = dist.ReduceOp.SUM
            g[lo0:].mul_(1.0 / world_size())
        for lo, hi in self.ranges[early:]:
            self._work.append(dist.all_reduce(g[lo:hi], op=op, async_op=True))

    def wait(self, i):
        """The current stream waits for bucket i (no-op when not distributed / already waited for)."""
        if i < len(self._work) and self._work[i] is not None:
            self._work[i].wait()
            self._work[i] = None

    def finish(self):
        for i in range(len(self._work)):
            self.wait(i)
        self._work = []

    def __call__(self):
        self.start()
        self.finish()


class _GlobalStd(torch.autograd.Function):
    """Unbiased std over dim 0 of the GLOBAL batch, this rank holding a shard of it.  Forward: one all-reduce of the moments
    sum(x), sum(x^2) (fp64: the variance is a difference of the two) and the count.  Backward: the incoming gradient is
    all-reduced (SUM) first -- every rank's loss depends on the one global statistic, so d(sum of the ranks' losses)/d std
    is the sum of the ranks' gradients -- then d std / d x_i = (x_i - mean) / ((n - 1) std) for this rank's rows.  With the
    rank-AVERAGED parameter gradients of the data-parallel step this reproduces the single-process gradient exactly."""

    @staticmethod
    def forward(ctx, x):
        xd = x.detach().double()
        k = xd[0].numel()
        m = torch.cat([xd.sum(dim=0).reshape(-1), (xd * xd).sum(dim=0).reshape(-1),
                       torch.full((1,), float(x.shape[0]), dtype=torch.float64, device=x.device)])
        dist.all_reduce(m, op=dist.ReduceOp.SUM)
        n = m[-1]
        mean = (m[:k] / n).reshape((1,) + tuple(x.shape[1:]))
        var = ((m[k:2 * k] - m[:k] * m[:k] / n) / (n - 1.0)).reshape(mean.shape)
        std = torch.sqrt(torch.clamp(var, min=0.0))
        ctx.save_for_backward(xd, mean, std, n)
        return std.to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        xd, mean, std, n = ctx.saved_tensors
        G = g.detach().double().clone()
        dist.all_reduce(G, op=dist.ReduceOp.SUM)
        gx = G * (xd - mean) / ((n - 1.0) * torch.clamp(std, min=1e-300))
        return torch.where(std > 0, gx, torch.zeros_like(gx)).to(g.dtype)


def batch_std(x):
    """`x.std(dim=0, keepdim=True)` over the GLOBAL batch: the path-length regulariser's `w_styles.std(dim=0)`
    (histoGAN/histoGAN.py:966-968).  The reference is single-GPU, its statistic spans the whole batch; under data
    parallelism (dim 0 = this rank's shard) it still does here (_GlobalStd), differentiable like the original."""
    if not is_dist():
        return x.std(dim=0, keepdim=True)
    return _GlobalStd.apply(x)


def all_reduce_scalar(value, op='mean', device=None):
    """All-reduce a python float ('mean' | 'max' | 'sum'); identity when not distributed."""
    if not is_dist():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if op == 'max':
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        if op == 'mean':
            t /= world_size()
    return float(t.item())
