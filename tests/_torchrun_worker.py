"""Worker of tests/test_distributed_cpu.py::test_torchrun_launcher_contract - started by
``python -m torch.distributed.run --nproc-per-node 2`` exactly as the driver starts bench.py for N > 1.
It goes through the same layer bench.py uses (dist.init_from_env / shard_clips / gather_maps /
max_over_ranks / barrier) - on gloo with CPU tensors where there is no GPU, on nccl (= RCCL) with tensors on
cuda:LOCAL_RANK where there is one (tests/test_distributed_gpu.py) - and rank 0 prints one JSON line."""
import json
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from cp_360_weakly_supervised_saliency_amd import dist as cpdist     # noqa: E402


def fake_map(clip_id):
    return np.random.RandomState(4321 + clip_id).rand(14, 28).astype(np.float32)


def main():
    n_clips = int(sys.argv[1])
    rank, world, local = cpdist.init_from_env()              # backend chosen as in bench.py (gloo without a GPU)
    assert world == int(os.environ['WORLD_SIZE']) and rank == int(os.environ['RANK'])
    dev = torch.device('cuda', local) if torch.cuda.is_available() else torch.device('cpu')
    mine = cpdist.shard_clips(n_clips, rank, world)
    local_maps = (torch.from_numpy(np.stack([fake_map(c) for c in mine])) if mine else torch.zeros((0, 14, 28))).to(dev)
    cpdist.barrier()
    maps = cpdist.gather_maps(local_maps, n_clips, rank, world)
    cpdist.barrier()
    elapsed = cpdist.max_over_ranks(0.5 + rank, dev)
    ok = bool(np.array_equal(maps.cpu().numpy(), np.stack([fake_map(c) for c in range(n_clips)])))
    if rank == 0:
        print(json.dumps({'n_gpus': world, 'clips': n_clips, 'gathered_equal_single_process': ok,
                          'max_elapsed': elapsed, 'local_rank': local,
                          'backend': torch.distributed.get_backend() if torch.distributed.is_initialized() else None,
                          'device': str(dev)}))
    cpdist.barrier()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == '__main__':
    main()
