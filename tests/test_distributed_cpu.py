"""world_size-2 gloo test (CPU) of the clip sharding + all-gather layer: the gathered
maps equal the single-process result clip for clip, including ragged clip counts."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_saliency(clip_id):
    """Deterministic stand-in for a clip's [14, 28] saliency map (the HIP pipeline
    cannot run on CPU and the oracle is too slow to run per rank here; the layer under
    test is sharding + gather order, which is independent of the map values)."""
    r = np.random.RandomState(1234 + clip_id)
    return r.rand(14, 28).astype(np.float32)


def _fake_steps(clip_id, T=3):
    """Stand-in for a clip's per-step maps [T, 14, 28] (``return_all_steps``)."""
    return np.stack([_fake_saliency(1000 * (t + 1) + clip_id) for t in range(T)])


def _worker(rank, world, port, n_clips, q, all_steps=False):
    sys.path.insert(0, REPO)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from cp_360_weakly_supervised_saliency_amd import dist as d
    r, w, _ = d.init_from_env(backend='gloo')
    mine = d.shard_clips(n_clips, r, w)
    fake = _fake_steps if all_steps else _fake_saliency
    empty = (0, 3, 14, 28) if all_steps else (0, 14, 28)
    local = torch.from_numpy(np.stack([fake(c) for c in mine])) if mine else torch.zeros(empty)
    d.barrier()
    # default: the caller owns the result (a later gather never overwrites it, nothing is cached)
    owned = d.gather_maps(local, n_clips, r, w)
    other = d.gather_maps(local + 2.0, n_clips, r, w)
    assert owned.data_ptr() != other.data_ptr() and len(d._GATHER_BUFS) == 0
    assert np.array_equal(other.numpy(), owned.numpy() + 2.0)
    allmaps = d.gather_maps(local, n_clips, r, w, reuse=True)
    first, ptr1 = allmaps.numpy().copy(), allmaps.data_ptr()
    assert np.array_equal(first, owned.numpy())
    # reuse=True (bench.py's timed loop), a second step with other maps: nothing new is allocated (the same send / recv / result
    # buffers) and the result is right
    again = d.gather_maps(local + 1.0, n_clips, r, w, reuse=True)
    assert again.data_ptr() == ptr1 and len(d._GATHER_BUFS) == 1, "gather_maps(reuse=True) allocated on its second call"
    assert np.array_equal(again.numpy(), first + 1.0)
    d.release_gather_buffers()
    assert len(d._GATHER_BUFS) == 0
    t = d.max_over_ranks(1.0 + r, 'cpu')
    q.put((rank, mine, first, t))
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize('n_clips', [4, 5, 1])
def test_two_rank_gather_equals_single_process(n_clips):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_clips, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = np.stack([_fake_saliency(c) for c in range(n_clips)])
    owned = []
    for rank, mine, allmaps, t in res:
        assert allmaps.shape == want.shape and np.array_equal(allmaps, want)
        assert t == 2.0
        owned += mine
    assert sorted(owned) == list(range(n_clips))


@pytest.mark.parametrize('n_clips', [4, 3])
def test_two_rank_gather_of_per_step_maps(n_clips):
    """``return_all_steps``: [clips_per_rank, T, 2w, 4w] goes through the same single all-gather (ragged too)."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_clips, q, True)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = np.stack([_fake_steps(c) for c in range(n_clips)])
    for rank, mine, allmaps, t in res:
        assert allmaps.shape == want.shape == (n_clips, 3, 14, 28) and np.array_equal(allmaps, want)


def test_shard_is_balanced_partition():
    from cp_360_weakly_supervised_saliency_amd.dist import shard_clips
    for n in (0, 1, 7, 32, 33):
        for world in (1, 2, 4, 8):
            parts = [shard_clips(n, r, world) for r in range(world)]
            assert sum(parts, []) == list(range(n))
            sizes = [len(p) for p in parts]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize('n_clips', [8, 5])
def test_torchrun_launcher_contract(n_clips):
    """The launcher / environment contract bench.py relies on for N > 1 (RANK, LOCAL_RANK, WORLD_SIZE,
    MASTER_* set by ``python -m torch.distributed.run``): two ranks started by the real launcher go
    through init_from_env -> shard_clips -> gather_maps -> max_over_ranks and rank 0 prints the JSON line."""
    import json
    import subprocess
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()),
           os.path.join(REPO, 'tests', '_torchrun_worker.py'), str(n_clips)]
    env = dict(os.environ, OMP_NUM_THREADS='1')
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=REPO)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout                     # exactly one line, from rank 0
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['clips'] == n_clips and out['gathered_equal_single_process'] is True
    assert out['max_elapsed'] == 1.5 and out['backend'] == 'gloo'


@pytest.mark.parametrize('world', [2, 1])
def test_bench_py_main_under_the_launcher_with_the_stub_engine(world):
    """bench.py ITSELF (not a look-alike worker) started the way the driver starts it for N > 1 - ``python -m
    torch.distributed.run --nproc-per-node N bench.py --gpus N --steps K --warmup W`` - with ``--stub-engine`` (CPU tensors,
    gloo, a stand-in for the HIP engine): argument handling, init_from_env, the per-rank clips, the all-gather, the
    max-over-ranks timing and every key of the one JSON line are exercised, so the first real multi-GPU run cannot die on a
    key error (8-GPU hardware is the driver's)."""
    import json
    import subprocess
    clips, frames, steps, warmup = 3, 2, 2, 1
    tail = [os.path.join(REPO, 'bench.py'), '--gpus', str(world), '--steps', str(steps), '--warmup', str(warmup), '--stub-engine',
            '--clips', str(clips), '--frames', str(frames), '--equi', '64x128', '--cube', '64']
    if world > 1:
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world),
               '--master-addr', '127.0.0.1', '--master-port', str(_free_port())] + tail
    else:
        cmd = [sys.executable] + tail
    env = dict(os.environ, OMP_NUM_THREADS='1')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        env.pop(k, None)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=REPO)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout                     # exactly one line, from rank 0
    out = json.loads(lines[0])
    assert out['n_gpus'] == world and out['steps'] == steps and out['warmup'] == warmup
    assert out['stub_engine'] is True and out['data'] == 'stub' and out['scaling'] == 'weak' and out['higher_is_better'] is True
    assert len(out['ms_per_step_per_rank']) == world
    assert out['map_shape'] == [world * clips, 4, 8]      # every rank's clips gathered: [N * clips, 2w, 4w], w = 64 / 32
    assert out['value'] > 0 and abs(out['value'] - world * clips * frames / (out['ms_per_step'] * 1e-3)) <= 0.01 * out['value']
    assert out['allgather_ms'] >= 0 and out['allgather_bytes_per_rank'] == clips * 4 * 8 * 4
    assert out['roofline'] is None and out['cpu_baseline'] is None and 'secondary' not in out
    assert out['held_clock_ghz'] is None and out['stage_ms'] is None
    for key in ('metric', 'unit', 'vs_baseline', 'dtype', 'config'):
        assert key in out
    # the default run issues its K steps through the engine's software pipeline (stream of batches) and reports the same K steps
    # issued one by one beside it, with the same gathered maps
    assert out['config']['pipelining'].startswith('SaliencyEngine.stream')
    assert out['sequential']['value'] > 0 and out['sequential']['same_maps'] is True
