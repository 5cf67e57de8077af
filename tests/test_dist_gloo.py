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
    # the status word that travels with the rows: rank 1 reports 7, every rank reads [0, 7] and the rows are what they were
    x = mdist.TopkExchange(tops, k, n_items, rank, world, status=7 * rank)
    again = x.finish()
    assert all(np.array_equal(a, b) for a, b in zip(again, got))
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), red=red, status=np.array(x.status), **{"item%d" % i: g for i, g in enumerate(got)})
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
        np.testing.assert_array_equal(o["status"], [0, 7])


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


class _FakeShardLib(object):
    """Stand-in for Lib in the host logic of sharded_match: a fixed pair list with counts, split by lo-row block."""

    def __init__(self, n_hi, n_lo, seed):
        rng = np.random.default_rng(seed)
        keep = np.flatnonzero(rng.random(n_hi * n_lo) < 0.15)
        self.n_lo = n_lo
        self.rank_all = keep
        self.cnt_all = rng.integers(0, 7, len(keep)).astype(np.int32)
        self.rows_all = rng.normal(size=(len(keep), 23))
        self.n_hi_anchors, self.n_lo_anchors = 9, 13
        self.block = None

    def match_shard_pairs(self, hi, lo, b, e, cc):
        lo_row = self.rank_all % self.n_lo
        self.block = np.flatnonzero((lo_row >= b) & (lo_row < e))
        uh, ul = np.zeros(self.n_hi_anchors, np.uint8), np.zeros(self.n_lo_anchors, np.uint8)
        uh[(self.rank_all[self.block] // self.n_lo) % self.n_hi_anchors] = 1
        ul[lo_row[self.block] % self.n_lo_anchors] = 1
        return uh, ul, len(self.block)

    def match_shard_topk(self, hi, lo, uh, ul, dist, k):
        self.flags_seen = np.concatenate([uh, ul])
        order = np.lexsort((self.rank_all[self.block], -self.cnt_all[self.block].astype(np.int64)))[:k]
        sel = self.block[order]
        return self.rows_all[sel], self.cnt_all[sel], self.rank_all[sel], int(uh.sum())


class _FakeSet(object):
    def __init__(self, n_rows):
        self.n = n_rows

    def size(self):
        return self.n, 0


def _shard_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = _FakeShardLib(30, 101, seed=4)
    rows, cnt, prank = mdist.sharded_match(lib, _FakeSet(30), _FakeSet(101), 0.6, 4.0, 20, rank, world)
    np.savez(os.path.join(out_dir, "shard%d.npz" % rank), rows=rows, cnt=cnt, prank=prank, flags=lib.flags_seen)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_match_two_ranks(tmp_path):
    """Exchange 1 (OR of the cloud flags) and Exchange 2 (all-gather + merge of the per-shard top-k) over gloo."""
    import torch.multiprocessing as mp
    world, k = 2, 20
    mp.spawn(_shard_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    ref = _FakeShardLib(30, 101, seed=4)
    order = np.lexsort((ref.rank_all, -ref.cnt_all.astype(np.int64)))[:k]
    uh = np.zeros(ref.n_hi_anchors, np.uint8)
    ul = np.zeros(ref.n_lo_anchors, np.uint8)
    uh[(ref.rank_all // ref.n_lo) % ref.n_hi_anchors] = 1
    ul[(ref.rank_all % ref.n_lo) % ref.n_lo_anchors] = 1
    for r in range(world):
        o = np.load(os.path.join(str(tmp_path), "shard%d.npz" % r))
        np.testing.assert_array_equal(o["prank"], ref.rank_all[order])
        np.testing.assert_array_equal(o["cnt"], ref.cnt_all[order])
        np.testing.assert_array_equal(o["rows"], ref.rows_all[order])
        np.testing.assert_array_equal(o["flags"], np.concatenate([uh, ul]))


class _FakeBuildLib(object):
    """Stand-in for Lib in the host logic of ShardedSetBuild: a "set" is the list of (anchor position, fan-out) rows its
    anchors produce; a wire image is [n_rows, rows...] as int64 in uint8 clothing."""

    class _Set(object):
        def __init__(self):
            self.rows = np.zeros((0, 2), np.int64)

        def size(self):
            return len(self.rows), 0

        def lane(self):
            return 0

        def close(self):
            pass

    def set_build(self, slots, coords, octave, subv, index, r, lim_main, lim_sec, into=None):
        s = into if into is not None else self._Set()
        rows = [(int(i), j) for i in index for j in range(int(i) % 4)]      # anchor "index" makes index % 4 rows
        s.rows = np.array(rows, np.int64).reshape(-1, 2)
        return s

    def set_wire_bytes(self, cap_rows, D=1024):
        return 8 * (1 + 2 * cap_rows)

    def set_export(self, share, cap_rows, wire=None, device_ptr=None):
        img = np.zeros(1 + 2 * cap_rows, np.int64)
        img[0] = len(share.rows)
        img[1:1 + 2 * len(share.rows)] = share.rows.reshape(-1)
        wire[...] = img.view(np.uint8)
        return wire

    def set_import(self, n_shares, cap_rows, coords, octave, subv, index, wires=None, device_ptr=None, into=None):
        img = np.ascontiguousarray(wires).view(np.int64).reshape(n_shares, 1 + 2 * cap_rows)
        parts = [img[r, 1:1 + 2 * img[r, 0]].reshape(-1, 2) for r in range(n_shares)]
        rows = np.concatenate(parts)
        into.rows = rows[np.lexsort((rows[:, 1], rows[:, 0]))]      # the reference's order: anchor, then row within the anchor
        return into


def _share_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    from mad_amd import _lib
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    fake = _FakeBuildLib()
    real_set = _lib.DeviceSet
    _lib.DeviceSet = lambda lib, lane=None: _FakeBuildLib._Set()      # no GPU here: the builder only needs size() / lane()
    try:
        n = 23
        b = mdist.ShardedSetBuild(fake, [0, 1], np.zeros((n, 3)), np.ones(n), np.zeros((n, 3)), np.arange(n), rank, world)
        outs = [b.enqueue().rows.copy() for _ in range(2)]      # sized by the first call, reused by the second
    finally:
        _lib.DeviceSet = real_set
    np.savez(os.path.join(out_dir, "full%d.npz" % rank), a=outs[0], b=outs[1], cap=b.cap_rows, mine=b.mine)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_set_build_two_ranks(tmp_path):
    """Stage A over gloo: anchors dealt round-robin, the image capacity agreed by a MAX all-reduce, one all-gather of the
    images, every rank ends up with the rows of the whole list in the reference's order."""
    import torch.multiprocessing as mp
    world = 2
    mp.spawn(_share_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    want = np.array([(i, j) for i in range(23) for j in range(i % 4)], np.int64)
    seen = []
    for r in range(world):
        o = np.load(os.path.join(str(tmp_path), "full%d.npz" % r))
        np.testing.assert_array_equal(o["a"], want)
        np.testing.assert_array_equal(o["b"], want)
        np.testing.assert_array_equal(o["mine"], np.arange(r, 23, world))
        seen.append(int(o["cap"]))
    per_rank = [sum(i % 4 for i in range(r, 23, world)) for r in range(world)]
    assert seen[0] == seen[1] == max(per_rank) + max(per_rank) // 8 + 64


class _FakePartLib(object):
    """Stand-in for Lib in the host logic of PartitionedMatch: every "device set" of a subunit carries a fixed pair list
    with match counts (a _FakeShardLib); whole matches return its unsharded top-k, block matches go through the shard calls."""

    def match_topk_many_begin(self, his, lo, cc, dist, k):
        return [(h, k) for h in his]

    def match_topk_many_finish(self, handle):
        out = []
        for h, k in handle:
            f = h.fake
            order = np.lexsort((f.rank_all, -f.cnt_all.astype(np.int64)))[:k]
            out.append((f.rows_all[order], f.rank_all[order], dict(n_pairs=len(f.rank_all), l_hi=0, l_lo=0, n_corr=h.n * lo_rows(h))))
        return out

    def match_shard_pairs(self, hi, lo, b, e, cc):
        self.cur = hi.fake
        return hi.fake.match_shard_pairs(hi, lo, b, e, cc)

    def match_shard_topk(self, hi, lo, uh, ul, dist, k):
        return self.cur.match_shard_topk(hi, lo, uh, ul, dist, k)


def lo_rows(h):
    return h.fake.n_lo


def _part_worker(rank, world, port, n_items, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    k, n_lo = 15, 101
    pm = mdist.PartitionedMatch(n_items, rank, world, make_group=dist.new_group)
    his = []
    for item in pm.items:
        h = _FakeSet(30 + item)
        h.fake = _FakeShardLib(30 + item, n_lo, seed=40 + item)
        his.append(h)
    lib = _FakePartLib()
    corr, tops, stats = pm.finish(lib, pm.begin(lib, his, _FakeSet(n_lo), 0.6, 4.0, k))
    got = mdist.all_gather_topk(tops, k, n_items, rank, world)
    np.savez(os.path.join(out_dir, "part%d.npz" % rank), corr=corr, load=np.array(pm.load()), **{"item%d" % i: g for i, g in enumerate(got)})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_items", [(2, 3), (3, 4)])
def test_partitioned_match_balances_the_leftover_subunits(tmp_path, world, n_items):
    """Subunits that do not divide by the ranks (BASELINE configs[4]: 12 on 8): whole subunits round-robin, the leftover ones
    split by blocks of map rows inside groups of ranks -- every rank carries the same share of the pair grids (within 10 %) and
    every subunit's merged top-k is the unsharded one, on every rank."""
    import torch.multiprocessing as mp
    k, n_lo = 15, 101
    mp.spawn(_part_worker, args=(world, _free_port(), n_items, str(tmp_path)), nprocs=world, join=True)
    outs = [np.load(os.path.join(str(tmp_path), "part%d.npz" % r)) for r in range(world)]
    for item in range(n_items):
        ref = _FakeShardLib(30 + item, n_lo, seed=40 + item)
        order = np.lexsort((ref.rank_all, -ref.cnt_all.astype(np.int64)))[:k]
        for o in outs:
            np.testing.assert_array_equal(o["item%d" % item], ref.rows_all[order])
    load = outs[0]["load"]
    assert max(load) <= 1.1 * load.mean()
    corr = np.array([float(o["corr"]) for o in outs])
    assert corr.sum() >= sum((30 + i) * n_lo for i in range(n_items)) - world * 40      # every pair grid covered once (block edges round)
    assert corr.max() <= 1.1 * corr.mean()


def test_plan_partition_shapes():
    for n, w in ((12, 8), (11, 8), (9, 8), (13, 8), (3, 2), (8, 8), (4, 8), (1, 1), (5, 1)):
        units, groups = mdist.plan_partition(n, w)
        load = mdist.partition_load(units)
        assert abs(sum(load) - n) < 1e-9 and max(load) <= 1.1 * (n / w) + 1e-9
        for j, ranks in enumerate(groups):
            item = (n // w) * w + j
            assert item % w in ranks      # the rank the top-k exchange expects the item from belongs to the group (and reports)
        seen = sorted(u[1] for mine in units for u in mine if u[0] == "whole")
        assert seen == list(range((n // w) * w))
