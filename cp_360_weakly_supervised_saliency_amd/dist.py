"""Data-parallel sharding of clips over the GPUs of one node (one process per GPU).

Clips share no state (the ConvLSTM recurrence is confined to a clip), so the path
shards by clip with no data-path collective; the only exchange is one all-gather of the
per-clip saliency maps ([clips_per_rank, 2w, 4w] f32 - a few KB per rank, latency-bound
over xGMI).  ``torch.distributed`` backend "nccl" is RCCL on ROCm; the same code runs
under "gloo" on CPU for the tests.  The reference has no distributed code at all
(SURVEY.md section 2): this layer is new, its contract is "gathered result == the
single-process result, clip for clip".
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the launcher (torchrun).
    Returns (rank, world_size, local_rank).  Single-process when WORLD_SIZE is unset."""
    # the host driver of this pool only supports dmabuf IPC: without this RCCL's buffer registration fails with
    # hipIpcGetMemHandle: invalid argument (it is exported on the boxes already; harmless to repeat, and it must be
    # set before the first HIP call of the process)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    # CP360_DIST_FORCE_PG=1: create the process group even for one rank, so that the RCCL branch (init, all-gather,
    # all-reduce, barrier with device_ids) can be exercised on a one-GPU box (tests/test_distributed_gpu.py)
    force = os.environ.get('CP360_DIST_FORCE_PG') == '1' and 'RANK' in os.environ
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl':
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_clips(n_clips, rank, world):
    """Clip ids owned by ``rank``: contiguous blocks, sizes differ by at most one
    (ragged totals are allowed; the gather pads to the largest block)."""
    base, extra = divmod(n_clips, world)
    lo = rank * base + min(rank, extra)
    return list(range(lo, lo + base + (1 if rank < extra else 0)))


# send / recv / result buffers of gather_maps, one set per (shape, dtype, device, n_clips, world): the all-gather sits inside
# bench.py's timed loop, and a fresh zeros + empty + cat per step is ~5 small launches that a 6 ms step (one clip per GPU) sees
_GATHER_BUFS = {}


def _into_tensor_ok():
    return hasattr(dist, 'all_gather_into_tensor') and dist.get_backend() != 'gloo'


def gather_maps(local_maps, n_clips, rank, world, reuse=False):
    """local_maps: [len(shard_clips(...)), h, w] f32 on this rank's device - or [.., T, h, w] with the per-step
    maps of ``return_all_steps`` (any trailing shape).  Returns [n_clips, ...] on every rank, ordered by clip id
    (one all_gather: 6 KB per rank at BASELINE config C4, 100 KB with all 16 per-step maps).

    Default: the result is a fresh tensor the caller owns.  ``reuse=True`` (bench.py's timed loop) allocates nothing after the
    first call of a shape: an even split gathers straight into a cached result tensor, a ragged one goes through a cached
    padded send / recv pair - the result is then OVERWRITTEN by the next ``reuse=True`` call with the same shape (clone it to
    keep it; a consumer on another stream must have finished reading it).  ``release_gather_buffers()`` drops the cache."""
    if world == 1 and not dist.is_initialized():
        return local_maps
    base, extra = divmod(n_clips, world)
    cap = base + (1 if extra else 0)
    tail = tuple(local_maps.shape[1:])
    key = (tail, local_maps.dtype, local_maps.device, n_clips, world)
    bufs = _GATHER_BUFS.get(key) if reuse else None
    if bufs is None:
        mk = lambda *shape: torch.empty(shape + tail, dtype=local_maps.dtype, device=local_maps.device)
        bufs = {'out': mk(n_clips)}
        if extra:
            bufs['send'] = torch.zeros((cap,) + tail, dtype=local_maps.dtype, device=local_maps.device)
            bufs['recv'] = mk(world, cap)
        if reuse:
            _GATHER_BUFS[key] = bufs
    out = bufs['out']
    if not extra:                                    # even split: rank r's block IS rows [r * base, (r + 1) * base) of the result
        send = local_maps if local_maps.is_contiguous() else local_maps.contiguous()
        if _into_tensor_ok():
            dist.all_gather_into_tensor(out, send)
        else:
            dist.all_gather(list(out.view((world, base) + tail).unbind(0)), send)
        return out
    send, recv = bufs['send'], bufs['recv']
    send[: local_maps.shape[0]].copy_(local_maps)    # (rows past this rank's block stay zero from the allocation)
    if _into_tensor_ok():
        dist.all_gather_into_tensor(recv, send)
    else:
        dist.all_gather(list(recv.unbind(0)), send)
    lo = 0
    for r in range(world):
        n = base + (1 if r < extra else 0)
        out[lo: lo + n].copy_(recv[r, :n])
        lo += n
    return out


def release_gather_buffers():
    """Drop the buffers cached by ``gather_maps(..., reuse=True)`` (call before destroy_process_group)."""
    _GATHER_BUFS.clear()


def max_over_ranks(value, device):
    """Scalar max across ranks (bench timing contract)."""
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def all_ranks(value, device):
    """The scalar of every rank, as a list ordered by rank (diagnosis output of bench.py)."""
    if not dist.is_initialized():
        return [float(value)]
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(v.item()) for v in out]


def barrier():
    if dist.is_initialized():
        if dist.get_backend() == 'nccl':            # RCCL: name the device, or the barrier guesses it from the rank
            dist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()
