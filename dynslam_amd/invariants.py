"""Structural invariants of an engine's map, from its dumps (numpy only): free-list accounting, unique block pointers, every
block position stored once, bucket entries in the bucket their position hashes to, excess entries chained from their bucket,
the visible list ascending and consistent with the entries' types.  Size-independent properties of the data structure — what the
parity tests check at BASELINE.json's full sizes where the CPU oracle is too slow (tests/test_gpu_fullsize.py), what the
sustained configs[4] run checks every thousand frames (tools/bench_cfg5_sustained.py) and what bench.py's short configs[4] leg
reports as `structural_invariants`.  Raises AssertionError on the first violated property."""
import numpy as np


def np_hash(pos, mask):
    p = pos.astype(np.int64).astype(np.uint32)
    return ((p[:, 0] * np.uint32(73856093)) ^ (p[:, 1] * np.uint32(19349669)) ^ (p[:, 2] * np.uint32(83492791))) & np.uint32(mask)


def check_structure(e, n_blocks, n_buckets):
    st = e.get_stats()
    ht = e.dump_hash_table()
    used = np.nonzero(ht["ptr"] >= 0)[0]
    # free-list accounting (InfiniTamDriver.h:241-244)
    assert n_blocks - 1 - st.last_free_block_id == len(used)
    # block pointers unique, disjoint from the live free list
    ptrs = ht["ptr"][used]
    assert len(np.unique(ptrs)) == len(ptrs)
    val, exl = e.dump_allocation_lists()
    free = val[: st.last_free_block_id + 1]
    assert len(np.unique(free)) == len(free) and not np.intersect1d(free, ptrs).size
    # no block position stored twice
    key = ht["pos"][used].astype(np.int64)
    packed = (key[:, 0] + 32768) | ((key[:, 1] + 32768) << 16) | ((key[:, 2] + 32768) << 32)
    assert len(np.unique(packed)) == len(packed)
    # bucket entries sit in the bucket their position hashes to
    in_bucket = used[used < n_buckets]
    assert np.array_equal(np_hash(ht["pos"][in_bucket], n_buckets - 1), in_bucket.astype(np.uint32))
    # every used excess entry is linked from exactly one entry whose position hashes to the same bucket
    in_excess = used[used >= n_buckets]
    link_src = np.nonzero(ht["offset"] >= 1)[0]
    targets = n_buckets + ht["offset"][link_src] - 1
    assert len(np.unique(targets)) == len(targets)
    assert np.isin(in_excess, targets).all()
    src_of = dict(zip(targets.tolist(), link_src.tolist()))
    sample = in_excess[:: max(1, len(in_excess) // 2000)]
    for t in sample.tolist():
        h = int(np_hash(ht["pos"][t:t + 1], n_buckets - 1)[0])
        cur, ok = t, False
        for _ in range(64):
            cur = src_of.get(cur, -1)
            if cur == h:
                ok = True
                break
            if cur < 0:
                break
        assert ok, f"excess entry {t} not chained from its bucket"
    # excess free list: live part unique and not in use
    xfree = exl[: st.last_free_excess_list_id + 1]
    assert len(np.unique(xfree)) == len(xfree)
    assert not np.isin(n_buckets + xfree, targets).any()
    # visible list: ascending, entries marked visible
    vis = e.dump_visible_list()
    assert (np.diff(vis) > 0).all()
    vt = e.dump_visible_types()
    assert (vt[vis] > 0).all() and int((vt > 0).sum()) == len(vis)
    return st, ht, used
