"""bench.py's own rank launcher (`--gpus N` with no WORLD_SIZE in the environment): environment of each rank, port choice,
refusal when the node has fewer GPUs than ranks.  No GPU needed."""
import importlib.util
import io
import os
import socket

from conftest import ROOT


def _bench_module():
    spec = importlib.util.spec_from_file_location('hg_bench_module', os.path.join(ROOT, 'bench.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_rank_env_is_what_torch_distributed_run_exports():
    b = _bench_module()
    base = {'PATH': '/bin', 'RANK': '7', 'HG_GRAPH': '0'}
    envs = [b.rank_env(base, r, 4, 29511) for r in range(4)]
    for r, e in enumerate(envs):
        assert (e['RANK'], e['LOCAL_RANK'], e['WORLD_SIZE'], e['LOCAL_WORLD_SIZE']) == (str(r), str(r), '4', '4')
        assert e['MASTER_ADDR'] == '127.0.0.1' and e['MASTER_PORT'] == '29511'
        assert e['HSA_ENABLE_IPC_MODE_LEGACY'] == '0' and e['HG_GRAPH'] == '0' and e['PATH'] == '/bin'
    assert base['RANK'] == '7'                         # the caller's environment is not modified


def test_free_port_is_bindable():
    b = _bench_module()
    p = b.free_port()
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(('127.0.0.1', p))


def test_launcher_refuses_more_rccl_ranks_than_gpus(monkeypatch):
    """One process per GPU over RCCL: with fewer devices than ranks nothing is started (this container has no GPU)."""
    b = _bench_module()
    monkeypatch.delenv('HG_DIST_BACKEND', raising=False)
    monkeypatch.setattr(b.torch.cuda, 'device_count', lambda: 1)
    started = []
    import subprocess
    monkeypatch.setattr(subprocess, 'Popen', lambda *a, **k: started.append(a) or (_ for _ in ()).throw(AssertionError('started')))
    assert b.launch_ranks(2, ['--gpus', '2'], io.StringIO()) == 2
    assert not started


def test_launcher_forwards_rank0_line_and_worst_exit_code(monkeypatch, tmp_path):
    """The launcher with a stand-in rank program: rank 0's last stdout line is forwarded, any rank's failure fails the job."""
    b = _bench_module()
    prog = tmp_path / 'rank.py'
    prog.write_text("import os, sys\n"
                    "r = int(os.environ['RANK']); assert os.environ['WORLD_SIZE'] == '3'\n"
                    "print('{\"rank\": %d}' % r)\n"
                    "sys.exit(int(os.environ.get('FAIL_RANK', '-1')) == r)\n")
    monkeypatch.setenv('HG_DIST_BACKEND', 'gloo')
    monkeypatch.setattr(b.os.path, 'abspath', lambda p: str(prog))
    out = io.StringIO()
    assert b.launch_ranks(3, [], out) == 0
    assert out.getvalue().strip() == '{"rank": 0}'
    monkeypatch.setenv('FAIL_RANK', '2')
    out = io.StringIO()
    assert b.launch_ranks(3, [], out) != 0 and out.getvalue() == ''
