"""
N>1 host logic on CPU: two gloo ranks exercise the partition / padding /
gather / index-keyed-noise rules that the GPU ranks use over RCCL.
"""

import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from guided_diffusion import dist_util


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_denoise(index):
    """Stands in for one volume's sampler run: depends only on the GLOBAL index."""
    g = dist_util.volume_generator(index, seed=10, device="cpu")
    return torch.randn(1, 1, 4, 4, 4, generator=g) + index


def _worker(rank, world, rdv_file, n_items, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    # file rendezvous: no TCP port to race for when the whole suite runs
    dist_util.setup_dist(backend="gloo", init_method="file://" + rdv_file)
    assert dist_util.rank() == rank and dist_util.world_size() == world
    mine = dist_util.partition(n_items)
    collected = []
    for idx in mine:                      # every rank runs the same number of rounds
        sample = _fake_denoise(idx) if idx is not None else torch.zeros(1, 1, 4, 4, 4)
        collected += dist_util.gather_round(sample, idx)
    dist_util.barrier()
    q.put((rank, mine, [(i, t.clone()) for i, t in collected]))
    dist.destroy_process_group()


def test_partition_rule():
    assert dist_util.partition(5, 0, 2) == [0, 2, 4]
    assert dist_util.partition(5, 1, 2) == [1, 3, None]          # padded: same number of rounds
    assert dist_util.partition(18, 3, 8) == [3, 11, None]        # the reference's 18 patches on 8 GPUs
    assert dist_util.partition(3, 0, 1) == [0, 1, 2]
    allv = sorted(i for r in range(8) for i in dist_util.partition(18, r, 8) if i is not None)
    assert allv == list(range(18))


def test_two_ranks_gloo_match_single_process(tmp_path):
    n_items = 5                                    # uneven on purpose
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    rdv = str(tmp_path / "rendezvous")
    procs = [ctx.Process(target=_worker, args=(r, 2, rdv, n_items, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect = {i: _fake_denoise(i) for i in range(n_items)}
    for rank, mine, collected in res:
        assert mine == dist_util.partition(n_items, rank, 2)
        assert [i for i, _ in collected] == list(range(n_items))     # every rank ends with all samples
        for i, t in collected:
            assert torch.equal(t, expect[i])                         # same as a 1-rank run


def test_config3_partition_is_even_on_eight_ranks():
    """BASELINE config 3 (64 volumes of 64^3 over 8 GPUs, scripts/test.py:235-246): eight volumes per rank,
    rank-strided, no padding round, every volume exactly once."""
    seen = []
    for r in range(8):
        mine = dist_util.partition(64, r, 8)
        assert mine == list(range(r, 64, 8)) and len(mine) == 8 and None not in mine
        seen += mine
    assert sorted(seen) == list(range(64))


def test_bench_refuses_a_launcher_of_another_world_size():
    """`bench.py --gpus 8` under a launcher that hands over WORLD_SIZE != 8 exits non-zero BEFORE any GPU or
    process-group call (runs on the CPU-only build box), naming both numbers; stdout stays empty."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="4")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--batch", "8"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert "--gpus 8" in r.stderr and "WORLD_SIZE 4" in r.stderr
    assert r.stdout.strip() == ""
