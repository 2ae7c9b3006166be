"""world_size-2 gloo tests (CPU) of the data-parallel plumbing used by the train step: flat gradient
all-reduce (sync + async), parameter broadcast, scalar reductions, and the weak-scaling shard logic of
bench.py.  The kernels themselves need a GPU; the collective logic does not."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from histogan_amd import ddp
        from histogan_amd.optim import FlatParams
        torch.manual_seed(100 + rank)                       # replicas start DIFFERENT ...
        net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Linear(7, 3))
        flat = FlatParams(net.parameters())
        ddp.broadcast_flat(flat)                            # ... and are made identical from rank 0
        ref0 = flat.data.clone()
        gathered = [torch.empty_like(ref0) for _ in range(world)]
        dist.all_gather(gathered, ref0)
        assert all(torch.equal(g, gathered[0]) for g in gathered)
        # parameters are views of the flat buffer
        assert net[0].weight.data_ptr() == flat.data.data_ptr()

        # gradient averaging: rank r contributes (r+1) * ones -> mean = (1+2)/2
        red = ddp.GradAllReduce(flat, chunks=3)
        flat.zero_grad()
        x = torch.ones(4, 5)
        (net(x).sum() * (rank + 1)).backward()
        flat.gather()                                       # autograd's tensors -> the flat buffer
        local = flat.grad.clone()
        red.start(); red.finish()
        both = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(both, local)
        assert torch.allclose(flat.grad, sum(both) / world, rtol=1e-6, atol=1e-7)
        assert net[1].bias.grad.data_ptr() == flat.grad.data_ptr() + 4 * (flat.numel - 3)   # still views

        # buckets: contiguous, 4 KB-aligned, covering the buffer; waited for one by one (the optimizer updates bucket i
        # while bucket i+1 is still being reduced: histogan_amd/optim.DiffGrad.step_buckets)
        big = FlatParams([torch.nn.Parameter(torch.full((5000,), float(rank + 1))), torch.nn.Parameter(torch.zeros(3000))])
        big.zero_grad()
        big.params[0].grad = torch.full((5000,), float(rank + 1)); big.params[1].grad = torch.full((3000,), 2.0 * (rank + 1))
        rb = ddp.GradAllReduce(big, chunks=4)
        assert rb.ranges[0][0] == 0 and rb.ranges[-1][1] == 8000 and all(a[1] == b[0] for a, b in zip(rb.ranges, rb.ranges[1:]))
        assert all(lo % 1024 == 0 for lo, _ in rb.ranges) and len(rb.ranges) == 4
        rb.start()
        for i, (lo, hi) in enumerate(rb.ranges):
            rb.wait(i)
            want = torch.cat([torch.full((5000,), 1.5), torch.full((3000,), 3.0)])[lo:hi]
            assert torch.allclose(big.grad[lo:hi], want)
        rb.finish()
        assert ddp.all_reduce_scalar(float(rank), 'mean') == pytest.approx(0.5)
        assert ddp.all_reduce_scalar(float(rank), 'max') == 1.0
        # shared-device detection (a collective): two CPU ranks are two processes -> nothing shared; a faked common identity is
        assert ddp.ranks_share_a_device('cpu') is False
        real = ddp._device_identity
        ddp._device_identity = lambda device: ('host', 'cuda', 'the one GPU')
        try:
            assert ddp.ranks_share_a_device('cuda:0') is True          # (first use for this key: gathers the faked identities)
        finally:
            ddp._device_identity = real
        # early start of the convolution-weight region (GradAllReduce.start_early): the leading 4-d parameters' gradients are
        # written straight into their flat slots (simulated), reduced first; the rest follows in start(); next step: plain start()
        from histogan_amd.optim import conv_first
        ps = [torch.nn.Parameter(torch.zeros(7)), torch.nn.Parameter(torch.zeros(40, 8, 3, 3)), torch.nn.Parameter(torch.zeros(3, 40, 1, 1)),
              torch.nn.Parameter(torch.zeros(2000))]
        fe = FlatParams(conv_first(ps))
        assert fe.n_conv == 40 * 8 * 9 + 120 and fe.params[0] is ps[1]
        re_ = ddp.GradAllReduce(fe, 2)
        for step in range(2):
            fe.zero_grad()
            off = 0
            for p_ in fe.params[:2]:
                fe.grad[off:off + p_.numel()].fill_(float(rank + 1 + step))
                fe.direct_written.add(fe.grad.data_ptr() + 4 * off)
                off += p_.numel()
            assert fe.conv_region_final()
            if step == 0:
                assert re_.start_early(None) and len(re_._work) == 2
            ps[0].grad = torch.full((7,), 10.0 * (rank + 1))
            ps[3].grad = torch.full((2000,), 100.0 * (rank + 1))
            re_.start()
            re_.finish()
            assert torch.allclose(fe.grad[:fe.n_conv], torch.full((fe.n_conv,), 1.5 + step))
            assert torch.allclose(ps[0].grad, torch.full((7,), 15.0)) and torch.allclose(ps[3].grad, torch.full((2000,), 150.0))
            assert re_.ranges[-1][1] == fe.numel and re_.ranges[0][0] == 0
        # global-batch std of the path-length term (ddp.batch_std): value and gradient of two ranks on halves of a batch ==
        # one process on the whole batch (loss = mean over ranks of a function of local rows and the global statistic)
        gen = torch.Generator().manual_seed(3)
        xa = torch.randn(6, 4, 5, generator=gen, dtype=torch.float64)
        cw = torch.randn(6, 4, 5, generator=gen, dtype=torch.float64)
        def loss_of(x, c, std):
            return ((x + c / (0.1 / (std + 1e-8) + 1e-8)) ** 2).mean()
        xs = xa[rank * 3:(rank + 1) * 3].clone().requires_grad_(True)
        lr_ = loss_of(xs, cw[rank * 3:(rank + 1) * 3], ddp.batch_std(xs))
        lr_.backward()
        xw = xa.clone().requires_grad_(True)
        std_w = xw.std(dim=0, keepdim=True)
        lw = 0.5 * (loss_of(xw[:3], cw[:3], std_w) + loss_of(xw[3:], cw[3:], std_w))
        lw.backward()
        assert torch.allclose(ddp.batch_std(xs.detach()), std_w.detach(), rtol=1e-12, atol=1e-14)
        # rank-averaged gradient convention: this rank's rows of d(mean of rank losses) = local gradient / world
        assert torch.allclose(xs.grad / 2, xw.grad[rank * 3:(rank + 1) * 3], rtol=1e-10, atol=1e-12)
        q.put((rank, 'ok'))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_flat_grad_allreduce_world2():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(30)
    assert sorted(res) == [(0, 'ok'), (1, 'ok')], res


def test_single_process_is_identity():
    from histogan_amd import ddp
    from histogan_amd.optim import FlatParams
    net = torch.nn.Linear(3, 2)
    flat = FlatParams(net.parameters())
    flat.grad.fill_(2.0)
    ddp.GradAllReduce(flat)()
    assert torch.all(flat.grad == 2.0)
    assert ddp.world_size() == 1 and ddp.rank() == 0
    assert ddp.all_reduce_scalar(3.5) == 3.5
    assert ddp.ranks_share_a_device('cpu') is False                     # no process group: no collective, nothing shared
    assert ddp._any_shared([('a', 'cuda', '0'), ('a', 'cuda', '1'), ('b', 'cuda', '0')]) is False
    assert ddp._any_shared([('a', 'cuda', '0'), ('a', 'cuda', '0')]) is True


def test_flatparams_gather_with_directly_written_slots():
    """FlatParams.gather(): slices written directly (conv._direct_wgrad, here simulated on CPU) are kept, autograd-delivered
    gradients are copied in, a parameter that got both has them added, parameters without a gradient are zeroed."""
    import torch
    from histogan_amd.optim import FlatParams
    ps = [torch.nn.Parameter(torch.randn(4, 3, 3, 3)), torch.nn.Parameter(torch.randn(5)),
          torch.nn.Parameter(torch.randn(2, 3, 1, 1)), torch.nn.Parameter(torch.randn(7))]
    flat = FlatParams(ps)
    flat.grad.fill_(123.0)                       # stale content of the previous step
    flat.zero_grad()
    assert flat.direct_ok and all(p.grad is None for p in ps)
    base = flat.grad.data_ptr()
    offs = [0, ps[0].numel(), ps[0].numel() + ps[1].numel(), ps[0].numel() + ps[1].numel() + ps[2].numel()]
    d0, d2 = torch.randn(ps[0].shape), torch.randn(ps[2].shape)
    flat.grad[offs[0]:offs[0] + ps[0].numel()].copy_(d0.reshape(-1)); flat.direct_written.add(base + 4 * offs[0])
    flat.grad[offs[2]:offs[2] + ps[2].numel()].copy_(d2.reshape(-1)); flat.direct_written.add(base + 4 * offs[2])
    a1, a2 = torch.randn(5), torch.randn(ps[2].shape)
    ps[1].grad = a1.clone()                      # delivered by autograd only
    ps[2].grad = a2.clone()                      # delivered by autograd AND written directly
    flat.gather()
    assert not flat.direct_ok
    assert torch.equal(ps[0].grad, d0) and torch.equal(ps[1].grad, a1) and torch.allclose(ps[2].grad, d2 + a2)
    assert torch.equal(ps[3].grad, torch.zeros(7))
    for p, o in zip(ps, offs):
        assert p.grad.data_ptr() == base + 4 * o            # every .grad is the slice of the flat buffer again
    flat.gather()                                # idempotent
    assert torch.equal(ps[0].grad, d0)
