"""hipGraph replay of the plain train step (histogan_amd/trainer.py::_graphed_step): the captured step must do exactly
what the same static-input step does when run eagerly -- same kernels, same RNG stream, same parameters afterwards."""
import random

import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(mode, steps, tmp_path):
    from histoGAN import Trainer
    random.seed(3)
    torch.manual_seed(3)
    tr = Trainer(f'g{mode}', str(tmp_path / 'r'), str(tmp_path / 'm'), 32, 2, batch_size=2, hist_bin=16, hist_insz=32,
                 hist_resizing='interpolation')
    tr.graph_mode = mode
    tr.run_evaluate = tr.run_save = False
    tr.set_synthetic_data_src()
    tr.init_GAN()
    graphed = 0
    for _ in range(steps):
        tr.train(alpha=2)
        graphed += int(tr.last_step_graphed)
    torch.cuda.synchronize()
    return tr, graphed


def test_graph_replay_equals_static_eager_step(gpu_device, tmp_path):
    a, ga = _run('2', 14, tmp_path)       # static inputs, no capture
    b, gb = _run('1', 14, tmp_path)       # plain step captured at step 6, gradient-penalty step at step 8; replayed afterwards
    assert ga == 0 and gb == 8 and not getattr(b, '_graph_failed', False)        # steps 6,7, 8 (GP), 9,10,11, 12 (GP), 13
    for name in ('_flat_g', '_flat_d'):
        assert torch.equal(getattr(a.GAN, name).data, getattr(b.GAN, name).data), name
    assert a.GAN.G_opt.step_count == b.GAN.G_opt.step_count == 14
    assert (a.d_loss, a.g_loss, a.h_loss) == (b.d_loss, b.g_loss, b.h_loss)
    assert b.host_enqueue_ms < a.host_enqueue_ms


def test_graph_auto_mode_decides_and_trains(gpu_device, tmp_path):
    """'auto' measures the host share of the first eager plain steps; at this tiny size the step is host-bound, so the
    graph is taken; losses stay finite; an eager gradient-penalty step in between sees the weights the replay left."""
    tr, g = _run('auto', 16, tmp_path)
    assert tr._graph_auto in (True, False)
    import math
    assert all(math.isfinite(v) for v in (tr.d_loss, tr.g_loss, tr.h_loss))
    if tr._graph_auto:
        assert g >= 5


def test_graph_is_dropped_when_the_model_is_rebuilt(gpu_device, tmp_path):
    """load() rebuilds the GAN (load_config -> init_GAN): a graph captured on the old buffers must not be replayed."""
    tr, g = _run('1', 9, tmp_path)
    assert g >= 2 and tr._graphs and {k[0] for k in tr._graphs} == {False, True}
    tr.save(0)
    old_ptr = tr.GAN._flat_g.data.data_ptr()
    tr.load(0)
    assert '_graphs' not in tr.__dict__ and '_graph' not in tr.__dict__     # (the allocator may or may not reuse the address)
    del old_ptr
    tr.steps = 9
    before = tr.GAN._flat_g.data.clone()
    for _ in range(3):
        tr.train(alpha=2)                      # captures anew on the new buffers and trains them
    assert tr.last_step_graphed and not torch.equal(before, tr.GAN._flat_g.data)
    import math
    assert all(math.isfinite(v) for v in (tr.d_loss, tr.g_loss, tr.h_loss))


def test_graph_cache_is_keyed_on_alpha(gpu_device, tmp_path):
    """alpha (the Hellinger weight) is a host scalar baked into the captured launches: another value must capture its own
    graph, not replay the old one."""
    tr, g = _run('1', 8, tmp_path)
    n0 = len(tr._graphs)
    tr.steps = 9
    tr.train(alpha=2)
    assert len(tr._graphs) == n0
    h2 = tr.h_loss
    tr.steps = 9
    tr.train(alpha=4)
    assert len(tr._graphs) == n0 + 1 and (False, 4.0, 2) in tr._graphs
    assert tr.h_loss > 1.5 * h2            # the histogram loss scales with alpha (same weights up to one step)


def test_failed_capture_falls_back_to_a_working_eager_step(gpu_device, tmp_path, monkeypatch):
    """A capture error in the G phase (after D's parameters were frozen) must leave the trainer able to run the step
    eagerly: requires_grad restored, gradient buffers reset."""
    import histogan_amd.trainer as T
    tr, g = _run('1', 6, tmp_path)         # six eager steps; the next plain step would capture
    calls = {'n': 0}
    real = T.hellinger_loss

    def boom(*a, **k):
        calls['n'] += 1
        if calls['n'] == 1:
            raise RuntimeError('injected: not capturable')
        return real(*a, **k)
    monkeypatch.setattr(T, 'hellinger_loss', boom)
    tr.steps = 9
    tr.train(alpha=2)                      # capture fails inside the G phase -> eager retry of the same step
    assert tr._graph_failed and calls['n'] == 2
    assert all(p.requires_grad for p in tr.GAN.D.parameters())
    tr.train(alpha=2)
    import math
    assert all(math.isfinite(v) for v in (tr.d_loss, tr.g_loss, tr.h_loss))
