"""bench.py end to end on the GPU box, including the data-parallel path on RCCL at world size 1 (the
driver's 8-GPU run must not be the first time `init_process_group('nccl', device_id=...)`, the AVG gradient all-reduce,
the statistics all-reduces and the global-batch Hellinger term execute on RCCL)."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
    return p


def _bench(extra_env, *args):
    env = dict(os.environ, **extra_env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), *args], env=env, cwd=ROOT, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith('{'), lines      # the contract: ONE JSON line on stdout, diagnostics on stderr
    return json.loads(lines[0])


def test_bench_train_on_rccl_world1(gpu_device):
    env = dict(RANK='0', LOCAL_RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()),
               HG_DIST_BACKEND='nccl', HG_DIST_FORCE='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    out = _bench(env, '--gpus', '1', '--steps', '5', '--warmup', '1', '--batch', '4', '--no-roofline')
    assert out['ddp']['backend'] == 'nccl' and out['ddp']['grad_allreduce_op'].startswith('AVG')
    assert out['ddp']['ranks'][0]['world_size'] == 1 and out['ddp']['ranks'][0]['device'] == 'cuda:0'
    assert set(out['ddp']['allreduce']) == {'D', 'G+S+H'} and out['value'] > 0


def test_bench_hist_line_has_the_contract_fields(gpu_device):
    out = _bench({}, '--workload', 'hist', '--steps', '5', '--warmup', '2', '--cpu-images', '1', '--cpu-reps', '1')
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
              'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in out, k
    assert out['roofline']['bound'] == 'mfma' and 0 < out['roofline']['frac'] < 1
    assert out['cpu_baseline']['kind'] == 'port' and out['cpu_baseline']['cpu']
    thr = out['roofline']['thresholding']
    assert thr['bound'] == 'hbm' and thr['north_star_hbm_target'] == 0.6
    assert thr['north_star_hbm_target_met'] == (thr['frac'] >= 0.6)      # reported honestly, whatever it is


def test_bench_train_stdout_is_one_json_line_under_graph_replay(gpu_device):
    """Round 3: the 'replaying captured hipGraphs' note went to stdout, a second line in front of the JSON whenever
    HG_GRAPH=auto (or 1) took the graph."""
    out = _bench({'HG_GRAPH': '1'}, '--gpus', '1', '--steps', '12', '--warmup', '2', '--batch', '4', '--no-roofline',
                 '--no-cpu-baseline', '--no-reference-eager')
    assert out['graph_replayed_steps'] > 0 and out['value'] > 0


def test_bench_gpus_2_launches_two_ranks_itself(gpu_device):
    """`python bench.py --gpus 2` with NO launcher environment must start two ranks itself (round 3's
    --gpus was decorative).  Two gloo ranks share GPU 0 here; on a multi-GPU node the same path puts one rank per device
    on RCCL (asserted by the launcher before it starts anything and by the `ddp` self-check)."""
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT')}
    env.update(HG_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '5', '--warmup', '1',
                        '--batch', '4', '--size', '64', '--capacity', '4', '--bins', '16', '--no-roofline'],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['ddp']['backend'] == 'gloo'
    assert [rk['rank'] for rk in out['ddp']['ranks']] == [0, 1] and all(rk['world_size'] == 2 for rk in out['ddp']['ranks'])
    assert out['config']['global_batch'] == 8 and out['value'] > 0
