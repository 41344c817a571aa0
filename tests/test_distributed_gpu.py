"""The RCCL branch of dist.py on hardware.  A gpurun box has ONE GPU and RCCL refuses two ranks on one device, so
the group is created for a single rank (CP360_DIST_FORCE_PG=1) through the real launcher: backend "nccl",
init_process_group, all_gather_into_tensor, all_reduce(MAX) and barrier(device_ids=...) all run on the device.
The multi-rank logic itself (sharding, ragged gathers, ordering) is covered on gloo in test_distributed_cpu.py."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_rccl_single_rank_through_torchrun():
    env = dict(os.environ, CP360_DIST_FORCE_PG='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1',
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()),
           os.path.join(REPO, 'tests', '_torchrun_worker.py'), '5']
    r = subprocess.run(cmd, env=env, cwd=REPO, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith('{')][-1]
    d = json.loads(line)
    assert d['backend'] == 'nccl' and d['device'].startswith('cuda')
    assert d['gathered_equal_single_process'] and d['n_gpus'] == 1 and d['max_elapsed'] == 0.5


def test_bench_py_pipelined_steps_through_torchrun_on_rccl():
    """bench.py ITSELF with the real engine under the launcher, one rank on RCCL (CP360_DIST_FORCE_PG=1): the default, pipelined
    form of the timed steps (SaliencyEngine.stream, two HIP streams) with the all-gather of every batch's maps inside the loop,
    then the same steps one by one - same gathered maps, every key of the line present.  Small geometry (256x512, cube 64)."""
    env = dict(os.environ, CP360_DIST_FORCE_PG='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1',
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()),
           os.path.join(REPO, 'bench.py'), '--gpus', '1', '--steps', '3', '--warmup', '1', '--clips', '2', '--frames', '3',
           '--equi', '256x512', '--cube', '64', '--no-secondary', '--no-cpu-baseline']
    r = subprocess.run(cmd, env=env, cwd=REPO, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    assert d['n_gpus'] == 1 and d['steps'] == 3 and d['value'] > 0
    assert d['config']['pipelining'].startswith('SaliencyEngine.stream')
    assert d['sequential']['same_maps'] is True and d['sequential']['value'] > 0
    assert d['map_shape'] == [2, 4, 8] and d['allgather_bytes_per_rank'] == 2 * 4 * 8 * 4
