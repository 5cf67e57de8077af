"""Multi-GPU layer: one process per GPU, `torch.distributed` for the exchanges
("nccl" = RCCL over xGMI on the node, "gloo" in the CPU tests).

Stage A of SURVEY.md section 8(e) -- a structure's orient + describe work split over the ranks -- is
`ShardedSetBuild`: anchors dealt round-robin, every rank builds its share (mad_set_build), ONE all-gather of
fixed-size wire images (int8 rows + norms + bin ids, ~1 KB per row; on RCCL straight out of and into device
memory, ordered against the library's kernels on the set's own stream), and every rank assembles the full
set in the reference's row order (mad_set_import) -- bit for bit the set one GPU builds from the whole list.

The hot path shards by independent units: every subunit is docked
against the same map, so subunits are dealt round-robin to the ranks and each rank runs
orient / describe / correlate / pose / top-k for its own subunits with no data-path
collective.  The one exchange is the all-gather of the per-subunit top-k pose rows
(k x 23 float64, <= 155 KB per rank): latency-bound, so it is a single fused collective
per step rather than one per subunit.

`merge_topk` is the reduction for the other sharding the survey describes (the pair grid
of ONE subunit split into column blocks of map rows): it merges per-shard top-k lists
into the global top-k in the reference's order (count descending, row-major pair rank
ascending, MaD.py:480).
"""
import numpy as np

RESULT_COLS = 23


def shard_round_robin(items, rank, world):
    """The items (subunit keys, candidate ids ...) this rank owns."""
    return [it for i, it in enumerate(items) if i % world == rank]


def owner_of(index, world):
    return index % world


def pack_topk(tops, k):
    """list of (<=k, 23) arrays -> (n, k, 23) float64 block + (n,) valid counts."""
    block = np.zeros((len(tops), k, RESULT_COLS), dtype=np.float64)
    valid = np.zeros(len(tops), dtype=np.int64)
    for i, t in enumerate(tops):
        t = np.asarray(t, dtype=np.float64).reshape(-1, RESULT_COLS)[:k]
        block[i, :len(t)] = t
        valid[i] = len(t)
    return block, valid


class TopkExchange(object):
    """One in-flight all-gather of top-k rows: the constructor enqueues it (asynchronously, on the collective library's
    own stream, so it overlaps the kernels of the next step); `finish` waits and unpacks.

    On RCCL the payload goes host -> device and the gathered block device -> host through PINNED staging tensors with
    non-blocking copies on a side stream, so neither call blocks the host on anything but the collective itself; the
    staging sets are reused round-robin (at most three exchanges are ever in flight)."""

    _pool = {}      # (bytes of one payload, world, device) -> [staging sets]; a set is busy from the constructor to finish()

    @classmethod
    def _staging(cls, n, world, dev):
        """A staging set that no unfinished exchange is using: the first idle one of the pool, or a new one.  (Round 2 handed the
        sets out round-robin from a cursor that did not follow the appends: with three exchanges in flight the fourth took the
        set of the second.)"""
        import torch
        key = (n, world, str(dev))
        sets = cls._pool.setdefault(key, [])
        for st in sets:
            if not st["busy"]:
                st["busy"] = True
                return st
        pin = dev != "cpu"
        st = dict(h_in=torch.empty(n, dtype=torch.float64, pin_memory=pin), h_out=torch.empty(world * n, dtype=torch.float64, pin_memory=pin),
                  d_in=torch.empty(n, dtype=torch.float64, device=dev), d_out=torch.empty(world * n, dtype=torch.float64, device=dev),
                  stream=torch.cuda.Stream() if pin else None, done=torch.cuda.Event() if pin else None, busy=True)
        sets.append(st)
        return st

    def __init__(self, tops, k, n_items_total, rank, world, group=None, device=None, status=0):
        """`status`: one word of this rank's own (0 = its step went through) that travels with the rows; after `finish()`
        `self.status` holds every rank's word -- how the ranks learn, without a collective of its own, that one of them has
        to stop or re-size (bench.py: a map image that outgrew its wire buffer)."""
        import torch
        import torch.distributed as dist
        self.k, self.n_items, self.world = k, n_items_total, world
        self.per_rank = (n_items_total + world - 1) // world
        mine = list(tops) + [np.zeros((0, RESULT_COLS))] * (self.per_rank - len(tops))
        block, valid = pack_topk(mine, k)
        self.work = None
        self.status = [int(status)]
        if world == 1 and not (dist.is_available() and dist.is_initialized()):
            self.local = (block, valid)
            return
        dev = device if device is not None else ("cuda" if dist.get_backend(group) == "nccl" else "cpu")
        n = block.size + valid.size + 1
        st = self.st = self._staging(n, world, dev)
        h = st["h_in"].numpy()
        h[:block.size] = block.reshape(-1)
        h[block.size:n - 1] = valid
        h[n - 1] = float(int(status))
        self.n = n
        if dev == "cpu":      # gloo: flat tensors in host memory
            self.work = dist.all_gather_into_tensor(st["h_out"], st["h_in"], group=group, async_op=True)
            return
        with torch.cuda.stream(st["stream"]):      # copy in, gather, copy out: one chain on a side stream, nothing waits on the host
            st["d_in"].copy_(st["h_in"], non_blocking=True)
            self.work = dist.all_gather_into_tensor(st["d_out"], st["d_in"], group=group, async_op=True)
            self.work.wait()      # orders the copy-out behind the collective on this stream (no host wait)
            st["h_out"].copy_(st["d_out"], non_blocking=True)
            st["done"].record()

    def finish(self):
        """-> list of n_items_total arrays (item order), identical on every rank."""
        k, per_rank = self.k, self.per_rank
        if self.work is None:
            block, valid = self.local
            return [block[i, :valid[i]] for i in range(self.n_items)]
        if self.st["done"] is not None:
            self.st["done"].synchronize()
        else:
            self.work.wait()
        out = self.st["h_out"].numpy().reshape(self.world, self.n)
        self.status = [int(x) for x in out[:, self.n - 1]]
        res = []
        for item in range(self.n_items):
            r, slot = owner_of(item, self.world), item // self.world
            blk = out[r, :per_rank * k * RESULT_COLS].reshape(per_rank, k, RESULT_COLS)
            nv = int(out[r, per_rank * k * RESULT_COLS + slot])
            res.append(blk[slot, :nv].copy())
        self.st["busy"] = False      # everything has been copied out of the staging set
        return res


def all_gather_topk(tops, k, n_items_total, rank, world, group=None, device=None):
    """Every rank contributes the top-k rows of ITS items (round-robin owner) and receives all.

    tops: list of (<=k, 23) arrays for shard_round_robin(range(n_items_total), rank, world), in that order.
    Returns a list of n_items_total arrays (item order), identical on every rank."""
    return TopkExchange(tops, k, n_items_total, rank, world, group, device).finish()


def plan_partition(n_items, world):
    """How the pair grids of n_items subunits are dealt to `world` ranks so that no rank carries much more than the mean
    (SURVEY.md 8(e) stages B-C; BASELINE configs[4] is 12 subunits on 8 GPUs: whole subunits alone leave four ranks with two and
    four with one, a ceiling of 6.0x).

    Whole rounds are dealt round-robin as whole subunits (item i -> rank i % world, no data-path collective).  Each of the
    `left = n_items % world` remaining subunits is split by blocks of map rows over a GROUP of ranks (`sharded_match`: an OR
    all-reduce of the cloud flags and an all-gather of the per-shard top-k inside the group): group j = ranks j, j + left,
    j + 2 left, ... (all ranks when such groups would be too unequal), so that rank `item % world` -- the rank the top-k exchange
    expects the item from -- belongs to it and reports.

    -> (units, groups): units[r] = [("whole", item) | ("block", item, part, parts, group index)], groups = [[ranks]]."""
    whole = (n_items // world) * world
    left = n_items - whole

    def deal(group_of):
        units = [[("whole", it) for it in range(r, whole, world)] for r in range(world)]
        groups = []
        for j in range(left):
            ranks = group_of(j)
            groups.append(ranks)
            for part, r in enumerate(ranks):
                units[r].append(("block", whole + j, part, len(ranks), j))
        return units, groups

    units, groups = deal(lambda j: list(range(j, world, left)))
    load = partition_load(units)
    if left and max(load) > 1.1 * sum(load) / world:      # groups of unequal size (13 on 8): every leftover subunit over ALL ranks instead
        units, groups = deal(lambda j: list(range(world)))
    return units, groups


def partition_load(units, rows=None):
    """Pair-grid share of every rank under `units` in subunit equivalents (rows: optional per-item weights)."""
    load = []
    for mine in units:
        t = 0.0
        for u in mine:
            w = 1.0 if rows is None else float(rows[u[1]])
            t += w if u[0] == "whole" else w / u[3]
        load.append(t)
    return load


class PartitionedMatch(object):
    """The matches of one step on one rank under `plan_partition`: the rank's whole subunits through the asynchronous bracket
    (`match_topk_many_begin` / `_finish`, no collective), then its blocks of leftover subunits through `sharded_match` inside their
    group.  `begin` enqueues the bracket and runs the sharded matches (they are synchronous: two small collectives each, while the
    bracket's kernels run on the other lanes); `finish` collects the bracket.

        pm = PartitionedMatch(n_items, rank, world, make_group=torch.distributed.new_group)      # every rank, same order
        st = pm.begin(lib, his, lo, cc, dist, k)        # his: the rank's device sets in the order of pm.items
        corr, tops, stats = pm.finish(lib, st)          # tops: what this rank reports to the top-k exchange, in slot order
    """

    def __init__(self, n_items, rank, world, make_group=None, stand_ins=None):
        self.n_items, self.rank, self.world = n_items, rank, world
        self.n_lo_seen = None      # rows of the map set at the last collected step: what the next steps cut their blocks from
        self.async_ok = False      # RCCL groups (or the rehearsal of one rank: stand_ins == "local"): ShardedMatchAsync
        self.units, self.groups = plan_partition(n_items, world)
        self.mine = self.units[rank]
        self.items = [u[1] for u in self.mine]
        self.n_whole = sum(1 for u in self.mine if u[0] == "whole")
        self.blocks = [u for u in self.mine if u[0] == "block"]
        self.pg = {}
        # stand_ins = (reduce_flags, gather): single-process stand-ins for the group collectives (rehearsals of one rank)
        self.stand_ins = stand_ins
        if make_group is not None and world > 1:
            for j, ranks in enumerate(self.groups):      # collective: every rank creates every group, in the same order
                g = make_group(ranks) if len(ranks) < world else None      # None = the default group (all ranks)
                if rank in ranks:
                    self.pg[j] = g
            import torch.distributed as dist
            self.async_ok = dist.get_backend() == "nccl"
        if stand_ins == "local":      # the rehearsal of ONE rank on one GPU: nobody to exchange with, everything else as on a node
            self.stand_ins, self.async_ok = None, True
            self.local = True
        else:
            self.local = False

    def load(self):
        return partition_load(self.units)

    def _block_sync(self, lib, hi, lo, cc, dist_thr, k, part, parts, j):
        kw = {}
        if self.stand_ins is not None:
            kw = dict(reduce_flags=self.stand_ins[0], gather=self.stand_ins[1])
        elif self.local:
            kw = dict(reduce_flags=lambda f: np.asarray(f, dtype=np.uint8).copy(), gather=lambda parts_: [parts_])
        return sharded_match(lib, hi, lo, cc, dist_thr, k, part, parts, group=self.pg.get(j), **kw)[0]

    def begin(self, lib, his, lo, cc, dist_thr, k):
        handle = lib.match_topk_many_begin(list(his[:self.n_whole]), lo, cc, dist_thr, k)
        done = []
        for hi, (_, item, part, parts, j) in zip(his[self.n_whole:], self.blocks):
            d = dict(item=item, part=part, parts=parts, j=j, hi=hi, lo=lo, args=(cc, dist_thr, k), rows=None, pending=None)
            if self.async_ok and self.n_lo_seen is not None and self.stand_ins is None:
                # no host round trip: both stages and both exchanges on the lane of the subunit's set, collected in finish()
                d["pending"] = ShardedMatchAsync(lib, hi, lo, cc, dist_thr, k, part, parts, self.n_lo_seen, group=self.pg.get(j), local=self.local)
            else:
                d["rows"] = self._block_sync(lib, hi, lo, cc, dist_thr, k, part, parts, j)
            done.append(d)
        return handle, done

    def finish(self, lib, state):
        """-> (correlations of this rank, the top-k rows it reports (whole items, then the leftover item it reports for), stats)"""
        handle, done = state
        corr, tops, stats = 0, [], []
        for top, idx, st in lib.match_topk_many_finish(handle):
            corr += st["n_corr"]
            tops.append(top)
            stats.append(st)
        for d in done:
            if d["pending"] is not None:
                res = d["pending"].finish()
                # (None: a shard of the group raised a flag -- every rank of the group has seen it and repeats the match the slow way)
                d["rows"] = res[0] if res is not None else self._block_sync(lib, d["hi"], d["lo"], *d["args"], d["part"], d["parts"], d["j"])
            n_lo, _ = d["lo"].size()
            n_hi, _ = d["hi"].size()
            self.n_lo_seen = int(n_lo)
            b, e = lo_row_block(n_lo, d["part"], d["parts"])
            d.update(n_corr=int(n_hi) * int(e - b), n_hi=int(n_hi), n_lo=int(e - b))
            corr += d["n_corr"]
            if owner_of(d["item"], self.world) == self.rank:      # rank item % world reports the merged rows (identical on every rank of the group)
                tops.append(d["rows"])
            stats.append(dict(n_pairs=0, l_hi=0, l_lo=0, n_corr=d["n_corr"], n_hi=d["n_hi"], n_lo=d["n_lo"], block=(d["part"], d["parts"])))
        return corr, tops, stats


def merge_topk(shard_rows, shard_counts, shard_pair_rank, k):
    """Merge per-shard top-k lists of ONE subunit into the global top-k.

    shard_rows[s]: (m_s, 23) rows, shard_counts[s]: (m_s,) integer match counts,
    shard_pair_rank[s]: (m_s,) GLOBAL row-major pair rank (hi_row * N_lo + lo_row).
    Order: count descending, then pair rank ascending -- python's stable sort of the
    row-major pair list by repeatability (MaD.py:480)."""
    rows = np.concatenate([np.asarray(r, dtype=np.float64).reshape(-1, RESULT_COLS) for r in shard_rows])
    cnt = np.concatenate([np.asarray(c, dtype=np.int64).reshape(-1) for c in shard_counts])
    rank = np.concatenate([np.asarray(p, dtype=np.int64).reshape(-1) for p in shard_pair_rank])
    order = np.lexsort((rank, -cnt))[:k]
    return rows[order], cnt[order], rank[order]


def lo_row_block(n_lo_rows, rank, world):
    """Contiguous block of map rows rank `rank` of `world` scores (the GEMM operand must be contiguous)."""
    return n_lo_rows * rank // world, n_lo_rows * (rank + 1) // world


def sharded_match(lib, hi, lo, cc, dist, k, rank, world, reduce_flags=None, gather=None, group=None):
    """ONE subunit against the map with the pair grid split over `world` ranks by blocks of map rows
    (stages B-C): every rank holds both sets, correlates hi against its own lo rows, the "anchor takes
    part in a pair" flags are OR-reduced (Exchange 1: the clouds of MaD.py:427-428 are global), every rank scores its
    own pairs against the global clouds and the per-rank top-k lists are gathered and merged (Exchange 2) into the
    unsharded order of MaD.py:480.  Returns (rows (<= k, 23), counts, global pair ranks), identical on every rank.

    `reduce_flags(uint8 array) -> uint8 array` and `gather(list of arrays) -> list over ranks of such lists` default to
    the torch.distributed collectives (RCCL over xGMI on the node); tests pass single-process stand-ins."""
    n_lo, _ = lo.size()
    b, e = lo_row_block(n_lo, rank, world)
    used_hi, used_lo, _ = lib.match_shard_pairs(hi, lo, b, e, cc)
    if reduce_flags is None:
        def reduce_flags(f):
            return or_reduce_flags(f, group=group)
    flags = reduce_flags(np.concatenate([used_hi, used_lo]))
    used_hi_all, used_lo_all = flags[:len(used_hi)], flags[len(used_hi):]
    rows, cnt, prank, _ = lib.match_shard_topk(hi, lo, used_hi_all, used_lo_all, dist, k)
    if gather is None:
        def gather(parts):
            return all_gather_arrays(parts, k, group=group)
    shards = gather([rows, cnt, prank])
    return merge_topk([s[0] for s in shards], [s[1] for s in shards], [s[2] for s in shards], k)


class ShardedMatchAsync(object):
    """`sharded_match` without a host round trip (round 4): both stages are enqueued on the lane of `hi`
    (`mad_match_shard_begin` / `mad_match_shard_score`), the flags and the shard's record stay in device tensors, and the two
    exchanges of SURVEY.md 8(e) -- OR all-reduce, all-gather -- are issued on that lane's stream between and behind them (the
    ExternalStream pattern of ShardedSetBuild).  `finish()` waits for ONE device-to-host copy and merges, or returns None when some
    shard of the group raised a flag (a capacity hint too small, a map set with another row count than `n_lo`): every rank of
    the group sees the same records, so all of them then repeat the match through the synchronous `sharded_match`.

    group: the torch.distributed group of the shards (RCCL); `local=True`: no collective at all -- the rehearsal of one rank, or a
    group of one."""

    _pool = {}      # device / pinned buffers of finished handles, by size (released before the interpreter tears the runtime down)

    def __init__(self, lib, hi, lo, cc, dist_thr, k, part, parts, n_lo, group=None, local=False):
        import torch
        if not ShardedMatchAsync._pool:
            import atexit
            atexit.register(ShardedMatchAsync._pool.clear)
        self.k, self.parts, self.part = int(k), int(parts), int(part)
        self.rec = lib.match_shard_record_doubles(self.k)
        n_fl = max(hi.n_anchors + lo.n_anchors, 1)
        world = 1 if local else parts
        key = (n_fl, self.rec, world)
        free = ShardedMatchAsync._pool.setdefault(key, [])
        if free:
            self.buf = free.pop()
        else:
            # (torch.empty: no fill kernel on torch's stream to race with the library's writes on the lane's; every byte is written.
            # Nothing is ALLOCATED while the lane's stream is torch's current one: the allocator would tie blocks to a stream that the
            # library destroys before torch shuts down)
            self.buf = dict(flags=torch.empty(n_fl, dtype=torch.uint8, device="cuda"), mine=torch.empty(self.rec, dtype=torch.float64, device="cuda"),
                            all=torch.empty(world * self.rec, dtype=torch.float64, device="cuda"))
        self.key = key
        B = self.buf
        b, e = lo_row_block(n_lo, part, parts)
        stream = torch.cuda.ExternalStream(hi.stream())
        lib.match_shard_begin(hi, lo, b, e, n_lo, cc, B["flags"].data_ptr())
        if not local:
            import torch.distributed as dist
            with torch.cuda.stream(stream):      # behind the shard's pair kernels, in front of its pose kernels
                dist.all_reduce(B["flags"], op=dist.ReduceOp.MAX, group=group, async_op=True).wait()
        lib.match_shard_score(hi, lo, B["flags"].data_ptr(), dist_thr, self.k, B["mine"].data_ptr())
        src = B["mine"]
        if not local:
            with torch.cuda.stream(stream):
                dist.all_gather_into_tensor(B["all"], B["mine"], group=group, async_op=True).wait()
            src = B["all"]
        # (the copy to the host and its event are the library's: torch records nothing on a stream it does not own)
        self.lib, self.n_out = lib, src.numel()
        self.ticket = lib.match_shard_collect(hi, src.data_ptr(), self.n_out)

    def finish(self):
        """-> (rows, counts, global pair ranks) merged over the shards, or None (some shard flagged: repeat synchronously)."""
        B, k = self.buf, self.k
        out = self.lib.match_shard_wait(self.ticket, self.n_out).reshape(-1, self.rec)
        ShardedMatchAsync._pool[self.key].append(B)
        if np.any(out[:, 1] != 0):
            return None
        rows, cnt, prank = [], [], []
        for r in out:
            m = int(r[0])
            rows.append(r[4:4 + m * RESULT_COLS].reshape(m, RESULT_COLS))
            cnt.append(r[4 + k * RESULT_COLS:4 + k * RESULT_COLS + m].astype(np.int64))
            prank.append(r[4 + k * (RESULT_COLS + 1):4 + k * (RESULT_COLS + 1) + m].astype(np.int64))
        return merge_topk(rows, cnt, prank, k)


def all_gather_arrays(parts, k, group=None, device=None):
    """All-gather of one rank's (rows (m, 23), counts (m,), ranks (m,)) with m <= k: one fused collective of a
    fixed-size float64 payload [m, rows..., counts..., ranks...] per rank.  -> list over ranks of [rows, counts, ranks]."""
    import torch
    import torch.distributed as dist
    rows, cnt, prank = parts
    m = len(rows)
    payload = np.zeros(1 + k * (RESULT_COLS + 2))
    payload[0] = m
    payload[1:1 + m * RESULT_COLS] = np.asarray(rows, dtype=np.float64).reshape(-1)
    payload[1 + k * RESULT_COLS:1 + k * RESULT_COLS + m] = cnt
    payload[1 + k * (RESULT_COLS + 1):1 + k * (RESULT_COLS + 1) + m] = prank      # < 2^53: exact in float64
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        out = payload[None]
    else:
        dev = device if device is not None else ("cuda" if dist.get_backend(group) == "nccl" else "cpu")
        t = torch.from_numpy(payload).to(dev)
        o = torch.empty(dist.get_world_size(group) * t.numel(), dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(o, t, group=group)
        out = o.view(-1, t.numel()).cpu().numpy()
    res = []
    for row in out:
        m = int(row[0])
        res.append([row[1:1 + m * RESULT_COLS].reshape(m, RESULT_COLS).copy(), row[1 + k * RESULT_COLS:1 + k * RESULT_COLS + m].astype(np.int64),
                    row[1 + k * (RESULT_COLS + 1):1 + k * (RESULT_COLS + 1) + m].astype(np.int64)])
    return res


def or_reduce_flags(flags, group=None, device=None):
    """Bitwise-OR all-reduce of a uint8 flag vector (the "row takes part in a pair" masks that make the
    hi / lo clouds global when the pair grid of one subunit is sharded, MaD.py:427-428)."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return np.asarray(flags, dtype=np.uint8).copy()
    dev = device if device is not None else ("cuda" if dist.get_backend(group) == "nccl" else "cpu")
    t = torch.from_numpy(np.asarray(flags, dtype=np.int32).copy()).to(dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return t.cpu().numpy().astype(np.uint8)


def share_of(n_anchors, rank, world):
    """Indices (into the structure's anchor list) of the anchors rank `rank` of `world` builds: a, a + world, ..."""
    return np.arange(rank, n_anchors, world)


class ShardedSetBuild(object):
    """The rows of ONE structure (normally the map: MaD.py:143-150 describes it once, every subunit is matched against it)
    built in shares over the ranks of a process group.  Orientation and description are independent per anchor
    (Orientator.py:80-108, Descriptor.py:106-116); anchor a goes to rank a % world.

        b = ShardedSetBuild(lib, slots, coords, octave, subv, index, rank, world)
        full = b.enqueue()        # every step: build the share, export, all-gather, import -> the full device set

    With the "nccl" backend (RCCL) nothing blocks the host: export, collective and import are ordered on the share's
    stream (the full set is bound to the same lane).  The image capacity is sized once (a blocking first build + a MAX
    all-reduce of the shares' row counts) and then carried from call to call; should a later build outgrow it, the
    import reports MAD_ENOSPC at the first use of the set and `resize()` repeats the sizing.
    With "gloo" (CPU tests, rehearsals on a box with fewer GPUs than ranks) the images travel through host memory.

    `emulate=(world, rank)`: rehearsal of ONE rank of a larger job on a single GPU -- the other ranks' images are built
    here once (untimed) and stay in the gather buffer; each call builds, exports and "receives" only this rank's share.
    Whatever it reports is a per-rank cost estimate without link traffic, never a multi-GPU measurement."""

    def __init__(self, lib, slots, coords, octave, subv, index, rank, world, group=None, emulate=None, force=False, r=8, lim_main=6, lim_sec=6):
        from . import _lib
        self.lib, self.slots, self.group = lib, slots, group
        self.coords = np.ascontiguousarray(coords, np.int32).reshape(-1, 3)
        self.octave = np.ascontiguousarray(octave, np.int32)
        self.subv = np.ascontiguousarray(subv, np.float64).reshape(-1, 3)
        self.index = np.ascontiguousarray(index, np.int32)
        self.emulate = emulate
        if emulate is not None:
            world, rank = emulate
        self.rank, self.world = rank, world
        self.r, self.lim_main, self.lim_sec = r, lim_main, lim_sec
        self.full = _lib.DeviceSet(lib)
        self.share = self.wire = self.gathered = self.cap_rows = None
        self.backend = None
        self.host_s = dict(build_share=0.0, export=0.0, collective=0.0, **{"import": 0.0}, calls=0)      # host seconds spent enqueuing, by piece
        self.sharded = world > 1 or bool(force)      # force: the export / all-gather / import path even for a single share
        if self.sharded:
            self.mine = share_of(len(self.octave), rank, world)
            self.share = _lib.DeviceSet(lib, lane=self.full.lane())      # one lane: build -> export -> gather -> import in stream order
            if emulate is None:
                import torch.distributed as dist
                self.backend = dist.get_backend(group)
            else:
                self.backend = "emulate"

    # -- pieces ---------------------------------------------------------------------------------
    def _build_share(self, rank=None, into=None):
        sel = self.mine if rank is None else share_of(len(self.octave), rank, self.world)
        return self.lib.set_build(self.slots, self.coords[sel], self.octave[sel], self.subv[sel], self.index[sel], self.r, self.lim_main,
                                  self.lim_sec, into=self.share if into is None else into)

    def resize(self):
        """Blocking: builds the share, agrees on an image capacity for all ranks, (re)allocates the buffers."""
        import torch
        if not self.sharded:
            return
        self._build_share()
        rows, _ = self.share.size()
        if self.backend == "emulate":
            from . import _lib
            tmp = _lib.DeviceSet(self.lib)
            counts = []
            for rr in range(self.world):
                self._build_share(rr, into=tmp)
                counts.append(tmp.size()[0])
            rows_max = max(counts)
        else:
            import torch.distributed as dist
            dev = "cuda" if self.backend == "nccl" else "cpu"
            t = torch.tensor([rows], dtype=torch.int64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
            rows_max = int(t.item())
        self._allocate(rows_max + rows_max // 8 + 64)
        nbytes = self.wire.numel()
        if self.backend == "emulate":      # the other ranks' images, once
            for rr in range(self.world):
                self._build_share(rr, into=tmp)
                self.lib.set_export(tmp, self.cap_rows, device_ptr=self.gathered.data_ptr() + rr * nbytes)
            self.lib.synchronize()
            tmp.close()

    def _allocate(self, cap_rows):
        import torch
        self.cap_rows = int(cap_rows)
        nbytes = self.lib.set_wire_bytes(self.cap_rows)
        dev = "cpu" if self.backend == "gloo" else "cuda"
        self.wire = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
        self.gathered = torch.zeros(self.world * nbytes, dtype=torch.uint8, device=dev)

    def shrink_for_rehearsal(self, cap_rows):
        """Rehearsal of an overflow (bench.py --rehearse-resize, tests): wire images of `cap_rows` rows, as if an earlier, smaller
        build had sized them.  The next import then reports MAD_ENOSPC at the first use of the full set and the ranks have to
        agree on `resize()`.  Every rank calls it with the same value, nothing may be in flight."""
        if self.sharded and self.backend != "emulate":
            self._allocate(cap_rows)

    def build_job(self):
        """The (slots, coords, octave, subv, index, into) of this rank's part, for `lib.prepare_build_many`: the caller builds
        it in one batch with its other structures and then calls `finish()`."""
        if not self.sharded:
            return (self.slots, self.coords, self.octave, self.subv, self.index, self.full)
        if self.cap_rows is None:
            self.resize()
        sel = self.mine
        return (self.slots, self.coords[sel], self.octave[sel], self.subv[sel], self.index[sel], self.share)

    def enqueue(self):
        """-> the full DeviceSet (asynchronous on RCCL; complete on return with gloo)."""
        lib = self.lib
        if not self.sharded:
            return lib.set_build(self.slots, self.coords, self.octave, self.subv, self.index, self.r, self.lim_main, self.lim_sec, into=self.full)
        if self.cap_rows is None:
            self.resize()
        import time
        tb = time.perf_counter()
        self._build_share()
        self.host_s["build_share"] += time.perf_counter() - tb
        return self.finish()

    def finish(self):
        """What follows the build of the share: export, all-gather, import -> the full DeviceSet."""
        import torch
        lib = self.lib
        if not self.sharded:
            return self.full
        nbytes = self.wire.numel()
        if self.backend == "gloo":
            import torch.distributed as dist
            lib.set_export(self.share, self.cap_rows, wire=self.wire.numpy())
            dist.all_gather_into_tensor(self.gathered, self.wire, group=self.group)
            return lib.set_import(self.world, self.cap_rows, self.coords, self.octave, self.subv, self.index, wires=self.gathered.numpy(), into=self.full)
        import time
        T = self.host_s
        t0 = time.perf_counter()
        lib.set_export(self.share, self.cap_rows, device_ptr=self.wire.data_ptr())
        t1 = time.perf_counter()
        stream = torch.cuda.ExternalStream(self.share.stream())      # the library's stream of this lane (lane 0's when the lanes are serialised)
        with torch.cuda.stream(stream):
            if self.backend == "emulate":
                self.gathered[self.rank * nbytes:(self.rank + 1) * nbytes].copy_(self.wire, non_blocking=True)
            else:
                import torch.distributed as dist
                # the collective starts behind the export kernel and the import behind the collective, all in stream order
                dist.all_gather_into_tensor(self.gathered, self.wire, group=self.group, async_op=True).wait()
        t2 = time.perf_counter()
        out = lib.set_import(self.world, self.cap_rows, self.coords, self.octave, self.subv, self.index, device_ptr=self.gathered.data_ptr(), into=self.full)
        t3 = time.perf_counter()
        T["export"] += t1 - t0; T["collective"] += t2 - t1; T["import"] += t3 - t2; T["calls"] += 1
        return out

    def close(self):
        self.full.close()
        if self.share is not None:
            self.share.close()
