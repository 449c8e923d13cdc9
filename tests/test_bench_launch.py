"""bench.py's rank-set resolution (`--gpus N` is authoritative) and its self-launch of N ranks.

CPU: the pure decision function against the environments that occur (bare shell, torchrun, a stale inherited
WORLD_SIZE, a launcher that disagrees with --gpus).  GPU: `python bench.py --gpus 8 --backend gloo` WITHOUT torchrun on
the 1-GPU box starts eight ranks by itself and reports n_gpus == 8."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_resolve_launch_cases():
    import bench
    r = bench.resolve_launch
    assert r(1, {}) == ('single', 1, 0, 0)
    assert r(8, {}) == ('spawn', 8, 0, 0)                                   # `python bench.py --gpus 8`: must launch 8 ranks itself
    # an inherited WORLD_SIZE (no RANK / LOCAL_RANK: not a live launcher) must neither turn --gpus 1 into 8 ranks nor be
    # reported as n_gpus
    assert r(1, {'WORLD_SIZE': '8'}) == ('single', 1, 0, 0)
    assert r(2, {'WORLD_SIZE': '8'}) == ('spawn', 2, 0, 0)
    tr = {'RANK': '3', 'LOCAL_RANK': '3', 'WORLD_SIZE': '8', 'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': '29500'}
    assert r(8, tr) == ('worker', 8, 3, 3)                                  # the driver's torchrun line
    assert r(1, {'RANK': '0', 'LOCAL_RANK': '0', 'WORLD_SIZE': '1'}) == ('single', 1, 0, 0)
    with pytest.raises(SystemExit):                                         # live launcher vs --gpus mismatch: refuse, never mis-report
        r(1, tr)
    with pytest.raises(SystemExit):
        r(4, tr)


def test_spawn_command_is_the_drivers_launch_line(monkeypatch):
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen['cmd'], seen['env'] = cmd, env
        return 0
    monkeypatch.setattr(subprocess, 'call', fake_call)
    monkeypatch.setenv('WORLD_SIZE', '8')                                    # stale: must not reach the children
    assert bench.spawn_ranks(4, ['--gpus', '4', '--steps', '2']) == 0
    cmd = seen['cmd']
    assert cmd[1:4] == ['-m', 'torch.distributed.run', '--nnodes=1']
    assert cmd[cmd.index('--nproc-per-node') + 1] == '4' and cmd[cmd.index('--master-addr') + 1] == '127.0.0.1'
    assert cmd[-4:] == ['--gpus', '4', '--steps', '2'] and os.path.basename(cmd[-5]) == 'bench.py'
    assert 'WORLD_SIZE' not in seen['env'] and seen['env']['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'


@pytest.mark.gpu
def test_bench_gpus8_self_launch_on_one_gpu():
    """`python bench.py --gpus 8 --backend gloo` WITHOUT torchrun on the 1-GPU box: eight ranks share the device (RCCL refuses that;
    gloo carries the collectives), the whole product path runs per rank -- hooks, 4 Segmentor buckets, async handles -- and rank
    0 reports n_gpus == 8 with every rank's exchange record (world = 8 had never run, not even functionally)."""
    env = dict(os.environ)
    env['WORLD_SIZE'] = '4'                                                 # stale value in the caller's environment
    for k in ('RANK', 'LOCAL_RANK'):
        env.pop(k, None)
    env['OMP_NUM_THREADS'] = '1'
    # Eight PROCESSES time-slicing one GPU is not a deployment of this code (one process per GPU) and not one the platform
    # is solid on: in 1 of 4 launches (measured, round 4) one rank dies with "HSA_STATUS_ERROR_ILLEGAL_INSTRUCTION ... Queue
    # aborting" -- a fault of the wave save / restore under preemption, never seen with one or two processes per GPU.  That fault,
    # and only that one, is retried; anything else (a dead-lock, a wrong record, a Python error) fails the test at once.
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--backend', 'gloo', '--steps', '2', '--warmup', '1',
           '--batch', '1', '--bands', '4', '--size', '176', '--no-prof', '--no-alt', '--no-cpu-baseline']
    for attempt in range(4):
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1200)
        if r.returncode == 0 or 'HSA_STATUS_ERROR_ILLEGAL_INSTRUCTION' not in r.stderr:
            break
        print('\n[bench 8 ranks on one GPU] attempt %d: a rank hit HSA_STATUS_ERROR_ILLEGAL_INSTRUCTION under time-slicing; retrying' % (attempt + 1))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    assert [l for l in r.stdout.splitlines() if l.strip()][-1] == lines[0], r.stdout[-600:]      # ... and nothing behind it
    res = json.loads(lines[0])
    assert res['n_gpus'] == 8 and res['config']['world_size'] == 8 and res['config']['backend'] == 'gloo'
    assert res['config']['global_batch'] == 8 and res['value'] > 0
    ex = res['config']['grad_exchange_last_step']
    assert ex['S']['buckets'] == 4 and ex['D']['buckets'] == 1
    assert ex['S']['launched_during_backward_min_over_ranks'] >= 1, ex       # on EVERY rank buckets left during backward
    assert res['host']['host_cpu_ms_per_step'] > 0 and res['host']['host_cpu_ms_per_step_max_over_ranks'] > 0
    print('\n[bench 8 ranks, gloo, one GPU] %s' % json.dumps({'exchange': ex, 'host': res['host']}))
    # --gpus 1 under the same stale WORLD_SIZE: one process, n_gpus 1; with --force-exchange the collectives run over a one-rank
    # nccl (= RCCL) group
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '1', '--warmup', '1',
                        '--batch', '1', '--bands', '4', '--size', '176', '--no-prof', '--no-alt', '--no-cpu-baseline',
                        '--force-exchange'],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out_lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert out_lines[-1].startswith('{'), out_lines[-6:]        # the record is the LAST line: RCCL's banner (C stdio) comes out before it
    res = json.loads(out_lines[-1])
    assert res['n_gpus'] == 1 and res['config']['world_size'] == 1 and res['config']['backend'] == 'nccl'
    assert res['config']['grad_exchange_last_step']['S']['buckets'] == 4
