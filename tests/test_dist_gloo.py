"""The N > 1 host logic on CPU: two gloo ranks exchange per-subunit top-k rows exactly as the
GPU ranks do over RCCL, and the shard-merge reproduces the unsharded top-k order."""
import os
import socket

import numpy as np
import pytest

from mad_amd import dist as mdist


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_topk(item, k):
    rng = np.random.default_rng(100 + item)
    n = [k, k - 7, 0, 3, k][item % 5]
    rows = rng.normal(size=(n, 23))
    rows[:, 1] = np.sort(rng.integers(0, 50, n))[::-1]
    return rows


def _worker(rank, world, port, n_items, k, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = mdist.shard_round_robin(list(range(n_items)), rank, world)
    tops = [_fake_topk(i, k) for i in mine]
    got = mdist.all_gather_topk(tops, k, n_items, rank, world)
    flags = np.zeros(37, np.uint8)
    flags[rank::5] = 1
    red = mdist.or_reduce_flags(flags)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), red=red, **{"item%d" % i: g for i, g in enumerate(got)})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_items", [5, 4, 1])
def test_all_gather_topk_two_ranks(tmp_path, n_items):
    import torch.multiprocessing as mp
    world, k = 2, 12
    mp.spawn(_worker, args=(world, _free_port(), n_items, k, str(tmp_path)), nprocs=world, join=True)
    outs = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    for i in range(n_items):
        ref = _fake_topk(i, k)
        for o in outs:
            np.testing.assert_array_equal(o["item%d" % i], ref)
    want = np.zeros(37, np.uint8)
    want[0::5] = 1
    want[1::5] = 1
    for o in outs:
        np.testing.assert_array_equal(o["red"], want)


def test_shard_round_robin_covers_everything():
    items = list("abcdefghij")
    for world in (1, 2, 3, 8):
        seen = []
        for r in range(world):
            mine = mdist.shard_round_robin(items, r, world)
            assert all(mdist.owner_of(items.index(m), world) == r for m in mine)
            seen += mine
        assert sorted(seen) == items


def test_merge_topk_equals_unsharded_order():
    rng = np.random.default_rng(3)
    n_hi, n_lo, k = 40, 90, 25
    counts = rng.integers(0, 6, n_hi * n_lo)
    keep = rng.random(n_hi * n_lo) < 0.2
    pair_rank = np.flatnonzero(keep)
    cnt = counts[pair_rank]
    rows = rng.normal(size=(len(pair_rank), 23))
    # unsharded: python's stable sort by count, descending, over the row-major list (MaD.py:480)
    full = sorted(range(len(pair_rank)), key=lambda i: cnt[i], reverse=True)[:k]
    for world in (2, 3, 8):
        lo_col = pair_rank % n_lo
        shard_rows, shard_cnt, shard_rank = [], [], []
        for r in range(world):
            sel = np.flatnonzero(lo_col % world == r)              # interleaved column blocks
            loc = sorted(sel, key=lambda i: cnt[i], reverse=True)[:k]      # per-shard top-k
            shard_rows.append(rows[loc]); shard_cnt.append(cnt[loc]); shard_rank.append(pair_rank[loc])
        mr, mc, mp_ = mdist.merge_topk(shard_rows, shard_cnt, shard_rank, k)
        np.testing.assert_array_equal(mp_, pair_rank[full])
        np.testing.assert_array_equal(mr, rows[full])
