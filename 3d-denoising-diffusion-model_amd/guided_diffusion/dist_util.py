"""
Single-node multi-GPU helpers: one process per GPU over RCCL (torch's "nccl"
backend on ROCm), launched by torch.distributed.run.  Replaces the reference's
MPI bootstrap (dist_util.py:22-47: mpi4py rank -> CUDA_VISIBLE_DEVICES =
rank % 2, hostname/port broadcast) and its checkpoint broadcast (:58-78).

The sampling path shards by INDEPENDENT sub-volumes (scripts/test.py:235-246:
patch i goes to rank i mod W); nothing is exchanged inside a sample and the
only collective is the all_gather of finished samples (scripts/test.py:74-78).
Two things the reference gets wrong and this module fixes:
  * an uneven work list makes ranks run different numbers of all_gather rounds
    and hang -> `partition` pads the list so every rank runs the same rounds;
  * noise drawn from a per-rank stream makes results depend on the world size ->
    `volume_generator` keys the noise by the GLOBAL volume index.
"""

import os

import torch
import torch.distributed as dist


def setup_dist(backend=None, init_method=None, share_gpu=False):
    """Initialise the default process group from the torchrun environment
    (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT; or an explicit
    `init_method` such as file://...).  A single process (WORLD_SIZE unset or 1)
    needs no group, like dist_util.py:29-31.
    share_gpu: every rank computes on cuda:0 (rehearsal of the multi-rank flow on a
    one-GPU box; needs backend "gloo" -- RCCL refuses two ranks on one device)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = 0 if share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 or dist.is_initialized():
        if torch.cuda.is_available():
            torch.cuda.set_device(local)
        return
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC (required on this stack)
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    extra = {}
    if init_method is not None:
        extra = dict(init_method=init_method, rank=int(os.environ["RANK"]), world_size=world)
    if backend == "nccl":
        if share_gpu:
            raise RuntimeError("share_gpu needs the gloo backend")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local), **extra)
    else:
        if torch.cuda.is_available():
            torch.cuda.set_device(local)
        dist.init_process_group(backend, **extra)


def rank():
    return dist.get_rank() if dist.is_initialized() else 0


def world_size():
    return dist.get_world_size() if dist.is_initialized() else 1


def dev():
    """The device this rank computes on (dist_util.py:49-55)."""
    if torch.cuda.is_available():
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def load_state_dict(path, **kwargs):
    """Every rank reads the local file (the reference broadcasts it over MPI in
    2^30-byte chunks).  Tensors only: nothing in the file is executed."""
    kwargs.setdefault("map_location", "cpu")
    return torch.load(path, weights_only=True, **kwargs)


def partition(n_items, rank_=None, world=None):
    """Indices this rank processes, in rounds: round r handles item r*W + rank
    (scripts/test.py:243).  Padded with None so every rank has ceil(n/W) rounds."""
    rank_ = rank() if rank_ is None else rank_
    world = world_size() if world is None else world
    rounds = (n_items + world - 1) // world
    return [(r * world + rank_) if (r * world + rank_) < n_items else None for r in range(rounds)]


def volume_generator(global_index, seed=10, device=None):
    """RNG for one volume, keyed by its global index: sampling volume i gives the
    same result on 1, 2, 4 or 8 ranks."""
    g = torch.Generator(device=device if device is not None else dev())
    g.manual_seed(int(seed) * 1000003 + int(global_index))
    return g


def gather_round(sample, index):
    """all_gather one round's samples (scripts/test.py:74-78).  `index` is this
    rank's item index for the round or None (padding).  Returns [(index, tensor)]
    for the real items of the round, ordered by index."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [] if index is None else [(index, sample)]
    world = dist.get_world_size()
    if dist.get_backend() == "gloo":
        sample = sample.cpu()                  # gloo collectives run through host memory
    idx = torch.tensor([-1 if index is None else index], dtype=torch.int64, device=sample.device)
    idxs = [torch.empty_like(idx) for _ in range(world)]
    outs = [torch.empty_like(sample) for _ in range(world)]
    dist.all_gather(idxs, idx)
    dist.all_gather(outs, sample)
    got = [(int(i.item()), o) for i, o in zip(idxs, outs) if int(i.item()) >= 0]
    return sorted(got, key=lambda t: t[0])


def barrier():
    if dist.is_initialized():
        dist.barrier()
