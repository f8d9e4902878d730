"""Host-side mirror of the reference's separator interface (ClusterSeparator, reference
src/ClusterSeparator.h:15-75) over the C ABI.

`ClusterSeparatorHip` keeps the method names of the reference class: separate(), get_cluster1/2(),
get_raw_dps_cluster1/2(), get_separating_dimension/value(), is_best_separation_legal(),
get_best_sensitivity/specificity().  `binning()` replays the work-list loop of reference
src/abawaca.cpp:95-197 around it.
"""
from __future__ import annotations

import ctypes as C
import time

import numpy as np

from ._lib import Context, DimBest, Split, lib


def scg_csr(scg_scaf, scg_id, S):
    """(scaffold, scg id) pairs -> CSR (off int32[S+1], ids int32[M])."""
    scg_scaf = np.asarray(scg_scaf, dtype=np.int64)
    order = np.argsort(scg_scaf, kind="stable")
    off = np.zeros(S + 1, dtype=np.int64)
    np.add.at(off, scg_scaf + 1, 1)
    return np.cumsum(off).astype(np.int32), np.ascontiguousarray(np.asarray(scg_id, dtype=np.int32)[order])


_PER_DTYPE = np.dtype([("found", "i4"), ("key", "i4"), ("sens", "f8"), ("spec", "f8")])     # = struct abw_dimbest


class ColumnStore:
    """The resident feature table (ClusterData + ScafDpData + SCGdb as the separator sees them)."""

    def __init__(self, ctx: Context, keys, dp2scaf, scaf_len, scg_off=None, scg_ids=None, n_scg=0, device_keys=None):
        """keys: [D, N] int32 on the host -- or None with device_keys = (device address of a [D][N] int32 table, D, N), e.g. a torch
        tensor's data_ptr(): the table is then built from the keys where they are (abw_sep_load_dev)."""
        self.ctx = ctx
        if device_keys is None:
            self.keys = np.ascontiguousarray(keys, dtype=np.int32)          # [D, N]
            self.D, self.N = self.keys.shape
        else:
            self.keys = None
            d_ptr, self.D, self.N = int(device_keys[0]), int(device_keys[1]), int(device_keys[2])
        self.dp2scaf = np.ascontiguousarray(dp2scaf, dtype=np.int32)
        self.scaf_len = np.ascontiguousarray(scaf_len, dtype=np.int64)
        self.S = len(self.scaf_len)
        self.scaf_ndps = np.bincount(self.dp2scaf, minlength=self.S).astype(np.int32)
        self.scaf_first = np.concatenate([[0], np.cumsum(self.scaf_ndps)[:-1]]).astype(np.int64)
        self.n_scg = int(n_scg)
        so = np.ascontiguousarray(scg_off, dtype=np.int32) if self.n_scg else None
        si = np.ascontiguousarray(scg_ids, dtype=np.int32) if self.n_scg else None
        p = lambda a: C.c_void_p(a.ctypes.data) if a is not None else None
        if device_keys is None:
            ctx.check(lib().abw_sep_load(ctx.handle, p(self.keys), self.D, self.N, p(self.dp2scaf), p(self.scaf_len), self.S,
                                         p(so), p(si), self.n_scg))
        else:
            ctx.check(lib().abw_sep_load_dev(ctx.handle, C.c_void_p(d_ptr), self.D, self.N, p(self.dp2scaf), p(self.scaf_len), self.S,
                                             p(so), p(si), self.n_scg))

    def dev_keys_ptr(self):
        return lib().abw_sep_dev_keys(self.ctx.handle)


class ClusterSeparatorHip:
    """One node of the recursion: the drop-in for ClusterSeparatorBySensitivitySpecificity."""
    cluster_ndps_threshold = 100      # reference src/ClusterSeparator.h:72

    def __init__(self, store: ColumnStore, scaf_ids=None, per_dim: bool = True):
        """per_dim = False: only the node's winner is asked for, as the reference's separate() reports it (the C++ adapter's call);
        the library may then skip thresholds that cannot beat what another dimension has reached."""
        self.store = store
        self.want_per_dim = per_dim
        self.scaf_ids = None if scaf_ids is None else np.ascontiguousarray(scaf_ids, dtype=np.int32)
        self._res = None

    def separate(self) -> bool:
        st = self.store
        m = st.S if self.scaf_ids is None else len(self.scaf_ids)
        ids = np.arange(st.S, dtype=np.int32) if self.scaf_ids is None else self.scaf_ids
        n = int(st.scaf_ndps[ids].sum())
        out = Split()
        per = (DimBest * st.D)() if self.want_per_dim else None
        # output buffers are kept on the store and re-used (fresh numpy pages would be faulted in by the copy)
        buf = st.__dict__.setdefault("_outbuf", {})
        if "side" not in buf or len(buf["side"]) < m:
            buf["side"] = np.zeros(max(m, 1), dtype=np.uint8)
        if "raw" not in buf or len(buf["raw"]) < n:
            buf["raw"] = np.zeros(max(n, 1), dtype=np.uint8)
        side, raw = buf["side"][:m], buf["raw"][:n]
        t0 = time.perf_counter()
        st.ctx.check(lib().abw_sep_node(st.ctx.handle,
                                        C.c_void_p(self.scaf_ids.ctypes.data) if self.scaf_ids is not None else None,
                                        m, self.cluster_ndps_threshold, C.byref(out), per,
                                        C.c_void_p(side.ctypes.data), C.c_void_p(raw.ctypes.data)))
        self.last_call_s = time.perf_counter() - t0          # the C-ABI call alone (host buffers in and out)
        self._res, self._side, self._raw, self._ids = out, side.copy(), raw.copy(), ids
        self._per = np.frombuffer(per, dtype=_PER_DTYPE).copy() if per is not None else None
        return bool(out.split)

    # -- accessors named after the reference's (ClusterSeparator.h:45-52, ...Specificity.h:51-52) --
    def get_separating_dimension(self):
        return int(self._res.dimension)

    def get_separating_value(self):
        return int(self._res.key)

    def is_best_separation_legal(self):
        return bool(self._res.legal)

    def get_best_sensitivity(self):
        return float(self._res.sensitivity)

    def get_best_specificity(self):
        return float(self._res.specificity)

    def per_dimension(self):
        return self._per

    def _node_dps(self):
        st = self.store
        return np.concatenate([np.arange(st.scaf_first[s], st.scaf_first[s] + st.scaf_ndps[s]) for s in self._ids]).astype(np.int64)

    def get_cluster1(self):
        """scaffold ids assigned to cluster1 (None when the node is terminal)."""
        return self._ids[self._side == 1] if self._res.split else None

    def get_cluster2(self):
        return self._ids[self._side == 2] if self._res.split else None

    def get_raw_dps_cluster1(self):
        return self._node_dps()[self._raw == 1]

    def get_raw_dps_cluster2(self):
        return self._node_dps()[self._raw == 2]

    def ndps(self):
        return int(self._res.ndps1), int(self._res.ndps2)


def binning(store: ColumnStore, max_nodes: int = 1 << 30):
    """reference src/abawaca.cpp:95-197: split until every cluster is terminal.
    -> (nodes list of dict, scaf2cluster int32[S], dp2cluster int32[N])"""
    work = {1: None}
    next_id = 2
    nodes = []
    scaf2cluster = np.zeros(store.S, dtype=np.int32)
    dp2cluster = np.zeros(store.N, dtype=np.int32)
    while work and len(nodes) < max_nodes:
        nid = min(work)
        ids = work.pop(nid)
        sep = ClusterSeparatorHip(store, ids, per_dim=False)
        split = sep.separate()
        all_ids = np.arange(store.S, dtype=np.int32) if ids is None else ids
        n = int(store.scaf_ndps[all_ids].sum())
        row = dict(id=nid, ndps=n, split=int(split), dim=sep.get_separating_dimension(), key=sep.get_separating_value(),
                   spec=sep.get_best_specificity(), sens=sep.get_best_sensitivity(), c1=0, n1=0, c2=0, n2=0, call_s=sep.last_call_s)
        if split:
            c1, c2 = next_id, next_id + 1
            next_id += 2
            work[c1], work[c2] = np.sort(sep.get_cluster1()), np.sort(sep.get_cluster2())
            n1, n2 = sep.ndps()
            row.update(c1=c1, n1=n1, c2=c2, n2=n2)
        else:
            scaf2cluster[all_ids] = nid
            cnt = store.scaf_ndps[all_ids].astype(np.int64)
            within = np.arange(int(cnt.sum()), dtype=np.int64) - np.repeat(np.cumsum(cnt) - cnt, cnt)
            dp2cluster[np.repeat(store.scaf_first[all_ids], cnt) + within] = nid
        nodes.append(row)
    return nodes, scaf2cluster, dp2cluster


_TREE_DTYPE = np.dtype([("id", "i4"), ("n_scaf", "i4"), ("ndps", "i8"), ("split", "i4"), ("dim", "i4"), ("key", "i4"), ("legal", "i4"),
                        ("sens", "f8"), ("spec", "f8"), ("c1", "i4"), ("c2", "i4"), ("n1", "i8"), ("n2", "i8")])      # = struct abw_tree_node


def binning_tree(store: ColumnStore, max_nodes: int = 0):
    """The same recursion in ONE library call (abw_sep_tree: level by level on the device).
    -> (nodes list of dict as binning(), scaf2cluster int32[S], seconds inside the C ABI)"""
    cap = max(64, 2 * (store.N // max(1, ClusterSeparatorHip.cluster_ndps_threshold)) + 8)
    while True:
        arr = np.zeros(cap, dtype=_TREE_DTYPE)
        s2c = np.zeros(store.S, dtype=np.int32)
        n = C.c_int64(0)
        t0 = time.perf_counter()
        rc = lib().abw_sep_tree(store.ctx.handle, ClusterSeparatorHip.cluster_ndps_threshold, int(max_nodes), C.c_void_p(arr.ctypes.data), cap,
                                C.byref(n), C.c_void_p(s2c.ctypes.data))
        dt = time.perf_counter() - t0
        if rc == -3 and n.value > cap:                          # ABW_ERR_CAPACITY
            cap = int(n.value)
            continue
        store.ctx.check(rc)
        break
    rows = [dict(id=int(r["id"]), ndps=int(r["ndps"]), split=int(r["split"]), dim=int(r["dim"]), key=int(r["key"]), spec=float(r["spec"]),
                 sens=float(r["sens"]), c1=int(r["c1"]), n1=int(r["n1"]), c2=int(r["c2"]), n2=int(r["n2"]), n_scaf=int(r["n_scaf"]))
            for r in arr[:int(n.value)]]
    return rows, s2c, dt


# ---- scaffold-sharded separator (SURVEY §8 row e; abw_sep_shard_*) ------------------------------------

class SepExchange(C.Structure):
    _fields_ = [("d_hist", C.c_void_p), ("hist_count", C.c_uint64), ("d_scg_lo", C.c_void_p), ("d_scg_hi", C.c_void_p),
                ("scg_count", C.c_uint64)]


class ShardedColumnStore(ColumnStore):
    """One shard's scaffolds (local ids 0..S-1) with the bin layout all shards share (multi.global_dictionary)."""

    def __init__(self, ctx: Context, keys, dp2scaf, scaf_len, dict_vals, dict_off, scg_off=None, scg_ids=None, n_scg=0):
        self.ctx = ctx
        self.keys = np.ascontiguousarray(keys, dtype=np.int32)
        self.D, self.N = self.keys.shape
        self.dp2scaf = np.ascontiguousarray(dp2scaf, dtype=np.int32)
        self.scaf_len = np.ascontiguousarray(scaf_len, dtype=np.int64)
        self.S = len(self.scaf_len)
        self.scaf_ndps = np.bincount(self.dp2scaf, minlength=self.S).astype(np.int32)
        self.scaf_first = np.concatenate([[0], np.cumsum(self.scaf_ndps)[:-1]]).astype(np.int64)
        self.n_scg = int(n_scg)
        so = np.ascontiguousarray(scg_off, dtype=np.int32) if self.n_scg else None
        si = np.ascontiguousarray(scg_ids, dtype=np.int32) if self.n_scg else None
        dv = np.ascontiguousarray(dict_vals, dtype=np.int32)
        do = np.ascontiguousarray(dict_off, dtype=np.int64)
        p = lambda a: C.c_void_p(a.ctypes.data) if a is not None and a.size else None
        ctx.check(lib().abw_sep_shard_load(ctx.handle, p(self.keys), self.D, self.N, p(self.dp2scaf), p(self.scaf_len), self.S,
                                           p(so), p(si), self.n_scg, p(dv), C.c_void_p(do.ctypes.data)))


class ShardNode:
    """One node of the recursion on one shard: accum -> (caller combines hist / scg_lo / scg_hi) -> eval -> decide."""
    cluster_ndps_threshold = ClusterSeparatorHip.cluster_ndps_threshold

    def __init__(self, store: ShardedColumnStore, local_ids=None, device=None):
        import torch
        from .kmeans import _DevView
        self.store = store
        self.ids = np.arange(store.S, dtype=np.int32) if local_ids is None else np.ascontiguousarray(local_ids, dtype=np.int32)
        self.m = len(self.ids)
        self.n = int(store.scaf_ndps[self.ids].sum()) if self.m else 0
        ex = SepExchange()
        ptr = C.c_void_p(self.ids.ctypes.data) if self.m else None
        store.ctx.check(lib().abw_sep_shard_accum(store.ctx.handle, ptr, -1 if local_ids is None else self.m, C.byref(ex)))
        dev = device if device is not None else torch.device("cuda", store.ctx.device)
        self.hist = torch.as_tensor(_DevView(ex.d_hist, ex.hist_count, "<i8"), device=dev)
        if ex.scg_count:
            self.scg_lo = torch.as_tensor(_DevView(ex.d_scg_lo, ex.scg_count, "<i4"), device=dev)
            self.scg_hi = torch.as_tensor(_DevView(ex.d_scg_hi, ex.scg_count, "<i4"), device=dev)
        else:
            self.scg_lo = self.scg_hi = None

    def eval(self):
        st = self.store
        self.out = Split()
        per = (DimBest * st.D)()
        self.side = np.zeros(max(self.m, 1), dtype=np.uint8)
        self.raw = np.zeros(max(self.n, 1), dtype=np.uint8)
        st.ctx.check(lib().abw_sep_shard_eval(st.ctx.handle, self.cluster_ndps_threshold, C.byref(self.out), per,
                                              C.c_void_p(self.side.ctypes.data), C.c_void_p(self.raw.ctypes.data)))
        self.side, self.raw = self.side[:self.m], self.raw[:self.n]
        self.per = np.frombuffer(per, dtype=_PER_DTYPE).copy()
        return int(self.out.ndps1), int(self.out.ndps2)

    def decide(self, ndps1_total, ndps2_total) -> bool:
        legal = self.out.legal
        lib().abw_sep_shard_decide(C.byref(self.out), int(ndps1_total), int(ndps2_total), self.cluster_ndps_threshold)
        if legal and not self.out.split:
            self.side[:] = 0
        return bool(self.out.split)


def separate_sharded(store: ShardedColumnStore, local_ids, dist=None, device=None) -> ShardNode:
    """One node over torch.distributed (RCCL on GPUs): every rank passes its own part of the node."""
    import torch
    from . import multi
    node = ShardNode(store, local_ids, device)
    store.ctx.synchronize()
    multi.combine_sep(node.hist, node.scg_lo, node.scg_hi, dist)
    torch.cuda.synchronize(node.hist.device)
    n1, n2 = node.eval()
    n1, n2 = multi.sum_counts([n1, n2], dist, node.hist.device)
    node.decide(n1, n2)
    return node


def separate_shards_local(stores, ids_list, combine: str = "library"):
    """The same node played by several shards of ONE process (stores may share a GPU).  combine = "library": the exchange
    is abw_sep_shard_combine (peer copies + add kernel, what integration/ClusterSeparatorHipSharded uses); "torch": plain
    tensor arithmetic (an independent check of it in the tests)."""
    import torch
    nodes = [ShardNode(st, ids) for st, ids in zip(stores, ids_list)]
    if combine == "library":
        arr = (C.c_void_p * len(nodes))(*[nd.store.ctx.handle for nd in nodes])
        rc = lib().abw_sep_shard_combine(arr, len(nodes))
        for nd in nodes:
            nd.store.ctx.check(rc)
    else:
        for nd in nodes:
            nd.store.ctx.synchronize()
        hist = torch.stack([nd.hist for nd in nodes]).sum(0)
        for nd in nodes:
            nd.hist.copy_(hist)
        if nodes[0].scg_lo is not None:
            lo = torch.stack([nd.scg_lo for nd in nodes]).amin(0)
            hi = torch.stack([nd.scg_hi for nd in nodes]).amax(0)
            for nd in nodes:
                nd.scg_lo.copy_(lo)
                nd.scg_hi.copy_(hi)
    torch.cuda.synchronize()
    parts = [nd.eval() for nd in nodes]
    n1, n2 = sum(p[0] for p in parts), sum(p[1] for p in parts)
    for nd in nodes:
        nd.decide(n1, n2)
    return nodes


def binning_sharded(store: ShardedColumnStore, separate, max_nodes: int = 1 << 30):
    """reference src/abawaca.cpp:95-197 with every node's scaffolds spread over the shards.  `separate(local_ids)`
    -> ShardNode (e.g. lambda ids: separate_sharded(store, ids, dist)).  The split decisions are identical on every
    shard, so all shards walk the same tree.  -> (nodes, scaf2cluster of the LOCAL scaffolds)"""
    work = {1: None}
    next_id = 2
    nodes = []
    scaf2cluster = np.zeros(store.S, dtype=np.int32)
    while work and len(nodes) < max_nodes:
        nid = min(work)
        ids = work.pop(nid)
        nd = separate(ids)
        o = nd.out
        row = dict(id=nid, split=int(o.split), dim=int(o.dimension), key=int(o.key), spec=float(o.specificity),
                   sens=float(o.sensitivity), c1=0, n1=0, c2=0, n2=0)
        if o.split:
            c1, c2 = next_id, next_id + 1
            next_id += 2
            work[c1], work[c2] = np.sort(nd.ids[nd.side == 1]), np.sort(nd.ids[nd.side == 2])
            row.update(c1=c1, n1=int(o.ndps1), c2=c2, n2=int(o.ndps2))
        else:
            scaf2cluster[nd.ids] = nid
        nodes.append(row)
    return nodes, scaf2cluster


def carve_shard(keys, dp2scaf, scaf_len, scaf_ids, scg_scaf=(), scg_id=()):
    """The table of one shard: the given (global) scaffolds in the given order, renumbered 0..len-1.
    -> dict(keys, dp2scaf, scaf_len, scg_scaf, scg_id, dps) with dps = the global dp index of every local dp."""
    keys = np.asarray(keys)
    dp2scaf = np.asarray(dp2scaf)
    scaf_ids = np.asarray(scaf_ids, dtype=np.int64)
    S = len(scaf_len)
    ndps = np.bincount(dp2scaf, minlength=S)
    first = np.concatenate([[0], np.cumsum(ndps)[:-1]])
    cnt = ndps[scaf_ids].astype(np.int64)
    dps = np.repeat(first[scaf_ids], cnt) + (np.arange(int(cnt.sum()), dtype=np.int64) - np.repeat(np.cumsum(cnt) - cnt, cnt))
    local_of = np.full(S, -1, dtype=np.int64)
    local_of[scaf_ids] = np.arange(len(scaf_ids))
    scg_scaf = np.asarray(scg_scaf, dtype=np.int64)
    keep = local_of[scg_scaf] >= 0 if len(scg_scaf) else np.zeros(0, bool)
    return dict(keys=np.ascontiguousarray(keys[:, dps]), dp2scaf=np.repeat(np.arange(len(scaf_ids)), ndps[scaf_ids]).astype(np.int32),
                scaf_len=np.asarray(scaf_len)[scaf_ids], scg_scaf=local_of[scg_scaf[keep]] if len(scg_scaf) else scg_scaf,
                scg_id=np.asarray(scg_id)[keep] if len(scg_scaf) else np.asarray(scg_id), dps=dps)


def binning_shards_local(stores, max_nodes: int = 1 << 30, combine: str = "library"):
    """binning_sharded for several shards driven in lock-step from one process (tests)."""
    k = len(stores)
    work = {1: [None] * k}
    next_id = 2
    rows = []
    s2c = [np.zeros(st.S, dtype=np.int32) for st in stores]
    while work and len(rows) < max_nodes:
        nid = min(work)
        ids = work.pop(nid)
        nodes = separate_shards_local(stores, ids, combine)
        o = nodes[0].out
        for nd in nodes[1:]:                                   # every shard reached the same verdict
            p = nd.out
            assert (p.split, p.dimension, p.key, p.legal, p.sensitivity, p.specificity, p.ndps1, p.ndps2) == \
                   (o.split, o.dimension, o.key, o.legal, o.sensitivity, o.specificity, o.ndps1, o.ndps2)
        row = dict(id=nid, ndps=sum(nd.n for nd in nodes), split=int(o.split), dim=int(o.dimension), key=int(o.key),
                   spec=float(o.specificity), sens=float(o.sensitivity), c1=0, n1=0, c2=0, n2=0, empty_shards=sum(nd.m == 0 for nd in nodes))
        if o.split:
            c1, c2 = next_id, next_id + 1
            next_id += 2
            work[c1] = [np.sort(nd.ids[nd.side == 1]) for nd in nodes]
            work[c2] = [np.sort(nd.ids[nd.side == 2]) for nd in nodes]
            row.update(c1=c1, n1=int(o.ndps1), c2=c2, n2=int(o.ndps2))
        else:
            for nd, a in zip(nodes, s2c):
                a[nd.ids] = nid
        rows.append(row)
    return rows, s2c


# ---- dimension-sharded recursion (abw_sep_tree_begin / eval / assign / advance) ---------------------------
# Every rank loads ALL scaffolds but only its own columns; per LEVEL of the recursion the ranks exchange one
# small record per node (their best thresholds) and one byte per scaffold (the sides the winners' owners decided).

_WIN_DTYPE = np.dtype([("found", "i4"), ("dim", "i4"), ("key", "i4"), ("legal", "i4"), ("sens", "f8"), ("spec", "f8"),
                       ("gdim", "i4"), ("reserved", "i4")])                                       # = struct abw_tree_winner


def dim_partition(keys, world: int):
    """Columns dealt to `world` ranks: the wide ones (coverage: any key >= 1024, they cost most) round-robin, then the
    narrow ones round-robin.  -> list of ascending int arrays of global column indices."""
    keys = np.asarray(keys)
    wide = np.array([bool(keys[d].max() >= 1024 or keys[d].min() < 0) for d in range(keys.shape[0])])
    w, n = np.flatnonzero(wide), np.flatnonzero(~wide)
    return [np.sort(np.concatenate([w[r::world], n[r::world]])).astype(np.int64) for r in range(world)]


class DimShardStore(ColumnStore):
    """All scaffolds, the columns `dims` (global 0-based indices) of the table."""

    def __init__(self, ctx: Context, keys, dims, dp2scaf, scaf_len, scg_off=None, scg_ids=None, n_scg=0):
        self.dims = np.ascontiguousarray(dims, dtype=np.int64)
        super().__init__(ctx, np.ascontiguousarray(np.asarray(keys)[self.dims]), dp2scaf, scaf_len, scg_off, scg_ids, n_scg)


class DimShardRun:
    """The level calls of one context."""

    def __init__(self, store: DimShardStore, max_nodes: int = 0):
        self.store = store
        self.cap = 256
        store.ctx.check(lib().abw_sep_tree_begin(store.ctx.handle, ClusterSeparatorHip.cluster_ndps_threshold, int(max_nodes)))

    def eval(self):
        """-> this context's bests of the level's nodes (global_dimension filled in), or None when the recursion is over"""
        st = self.store
        while True:
            best = np.zeros(self.cap, dtype=_WIN_DTYPE)
            k = C.c_int64(0)
            rc = lib().abw_sep_tree_eval(st.ctx.handle, C.c_void_p(best.ctypes.data), self.cap, C.byref(k))
            if rc == -3 and k.value > self.cap:
                self.cap = int(k.value)
                continue
            st.ctx.check(rc)
            break
        if k.value == 0:
            return None
        best = best[:int(k.value)].copy()
        mine = best["found"] != 0
        best["gdim"][mine] = st.dims[best["dim"][mine]] + 1
        return best

    def assign(self, winners, owner_is_me):
        """winners: merged records; owner_is_me: bool per node.  -> torch uint8 view [S] of the side array (device)"""
        import torch
        from .kmeans import _DevView
        st = self.store
        w = np.ascontiguousarray(winners.copy())
        w["dim"][~np.asarray(owner_is_me, dtype=bool)] = -1
        ptr, cnt = C.c_void_p(), C.c_int64(0)
        st.ctx.check(lib().abw_sep_tree_assign(st.ctx.handle, C.c_void_p(w.ctypes.data), len(w), C.byref(ptr), C.byref(cnt)))
        return torch.as_tensor(_DevView(ptr.value, cnt.value, "|u1"), device=torch.device("cuda", st.ctx.device))

    def advance(self):
        self.store.ctx.check(lib().abw_sep_tree_advance(self.store.ctx.handle))

    def result(self):
        st = self.store
        cap = 1024
        while True:
            arr = np.zeros(cap, dtype=_TREE_DTYPE)
            s2c = np.zeros(st.S, dtype=np.int32)
            n = C.c_int64(0)
            rc = lib().abw_sep_tree_result(st.ctx.handle, C.c_void_p(arr.ctypes.data), cap, C.byref(n), C.c_void_p(s2c.ctypes.data))
            if rc == -3 and n.value > cap:
                cap = int(n.value)
                continue
            st.ctx.check(rc)
            break
        rows = [dict(id=int(r["id"]), ndps=int(r["ndps"]), split=int(r["split"]), dim=int(r["dim"]), key=int(r["key"]), spec=float(r["spec"]),
                     sens=float(r["sens"]), c1=int(r["c1"]), n1=int(r["n1"]), c2=int(r["c2"]), n2=int(r["n2"]), n_scaf=int(r["n_scaf"]))
                for r in arr[:int(n.value)]]
        return rows, s2c


def merge_winners(all_best):
    """all_best: [world][K] records -> (winners [K], owner int32 [K]) by the library's rule (abw_sep_tree_merge)."""
    a = np.ascontiguousarray(np.stack(all_best))
    world, K = a.shape
    out = np.zeros(K, dtype=_WIN_DTYPE)
    owner = np.zeros(K, dtype=np.int32)
    rc = lib().abw_sep_tree_merge(C.c_void_p(a.ctypes.data), world, K, C.c_void_p(out.ctypes.data), C.c_void_p(owner.ctypes.data))
    if rc:
        raise RuntimeError(f"abw_sep_tree_merge: {rc}")
    return out, owner


def binning_tree_dims_local(stores, max_nodes: int = 0):
    """The dimension-sharded recursion with all shards driven from one process (tests; the stores may share a GPU)."""
    import torch
    runs = [DimShardRun(st, max_nodes) for st in stores]
    exchanged = 0
    while True:
        bests = [r.eval() for r in runs]
        if bests[0] is None:
            assert all(b is None for b in bests)
            break
        winners, owner = merge_winners(bests)
        sides = [r.assign(winners, owner == k) for k, r in enumerate(runs)]
        for st in stores:
            st.ctx.synchronize()
        tot = torch.stack(sides).amax(0)
        for t in sides:
            t.copy_(tot)
        torch.cuda.synchronize()
        exchanged += winners.nbytes + int(tot.numel())
        for r in runs:
            r.advance()
    res = [r.result() for r in runs]
    return res, exchanged


def binning_tree_dims(store: DimShardStore, dist, max_nodes: int = 0):
    """One rank of the dimension-sharded recursion over torch.distributed (RCCL on GPUs).
    -> (rows, scaf2cluster, seconds, bytes this rank exchanged per direction)"""
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device("cuda", store.ctx.device) if dist.get_backend() == "nccl" else torch.device("cpu")
    t0 = time.perf_counter()
    run = DimShardRun(store, max_nodes)
    exchanged = 0
    while True:
        best = run.eval()
        k = torch.tensor([0 if best is None else len(best)], dtype=torch.int64, device=dev)
        ks = [torch.zeros_like(k) for _ in range(world)]
        dist.all_gather(ks, k)
        if best is None:
            assert all(int(x) == 0 for x in ks)
            break
        assert all(int(x) == len(best) for x in ks), "the ranks walk different trees"
        mine = torch.from_numpy(best.view(np.uint8).reshape(-1).copy()).to(dev)
        parts = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        winners, owner = merge_winners([p.cpu().numpy().view(_WIN_DTYPE) for p in parts])
        side = run.assign(winners, owner == rank)
        store.ctx.synchronize()
        if dev.type == "cpu":
            h = side.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.MAX)
            side.copy_(h)
        else:
            dist.all_reduce(side, op=dist.ReduceOp.MAX)
        torch.cuda.synchronize()
        exchanged += mine.numel() * world + side.numel()
        run.advance()
    rows, s2c = run.result()
    return rows, s2c, time.perf_counter() - t0, exchanged


# ---- scaffold-sharded recursion (abw_sep_stree_*): the signature columns cut by SCAFFOLD, the coverage columns by DIMENSION ----
# Rank r holds (a) the narrow columns of a contiguous range of scaffolds in one context and (b) ALL scaffolds with its share of
# the wide columns in a second one, which walks the recursion with the level calls.  Per level the ranks exchange the packed
# histogram slices of the nodes that are accumulated (integer SUM), the SCG lists (MIN / MAX), one record per node and one byte
# per scaffold.

_LEVEL_NODE_DTYPE = np.dtype([("id", "i4"), ("n_scaf", "i4"), ("ndps", "i8"), ("parent", "i4"), ("reserved", "i4")])    # = struct abw_tree_level_node


class TreeLevel(C.Structure):
    _fields_ = [("n_nodes", C.c_int64), ("n_scaf", C.c_int64), ("d_scaf_slot", C.c_void_p), ("d_side", C.c_void_p)]


def scaffold_ranges(scaf_ndps, world: int):
    """Contiguous scaffold ranges with about the same number of datapoints each -> world + 1 bounds."""
    cum = np.concatenate([[0], np.cumsum(np.asarray(scaf_ndps, dtype=np.int64))])
    b = [0]
    for k in range(1, world):
        b.append(max(b[-1], int(np.searchsorted(cum, cum[-1] * k / world, side="left"))))
    b.append(len(scaf_ndps))
    return b


def stree_mine(best_n, best_w):
    """A rank's candidate of every node: the better of the node's narrow best (the same on every rank) and this rank's own wide
    best -- the larger product sensitivity x specificity, on equal products the smaller column of the whole table."""
    out = best_w.copy()
    pn, pw = best_n["sens"] * best_n["spec"], best_w["sens"] * best_w["spec"]
    take_n = (best_n["found"] != 0) & ((best_w["found"] == 0) | (pn > pw) | ((pn == pw) & (best_n["gdim"] < best_w["gdim"])))
    out[take_n] = best_n[take_n]
    return out


class ScafShard:
    """One rank of the scaffold-sharded recursion: the narrow columns of its scaffolds [s0, s1) (renumbered from 0) in ctx_n, all
    scaffolds with its wide columns in ctx_w.
      narrow_local : int32 [len(narrow_cols), dps of its scaffolds]   wide_rows : int32 [len(wide_mine), N] (or None: no wide column)
      narrow_cols / wide_mine : the columns' numbers in the whole table
      narrow_bins  : the largest key + 1 of every narrow column over ALL ranks' scaffolds (one MAX over the ranks): the shards'
                     histograms share that layout
    dp2scaf, scaf_len, scg_off, scg_ids describe ALL scaffolds (a few bytes each)."""

    def __init__(self, ctx_n: Context, ctx_w: Context, narrow_local, wide_rows, dp2scaf, scaf_len, scg_off, scg_ids, n_scg, s0: int, s1: int,
                 wide_mine, narrow_cols, narrow_bins):
        dp2scaf = np.ascontiguousarray(dp2scaf, dtype=np.int32)
        scaf_len = np.ascontiguousarray(scaf_len, dtype=np.int64)
        S, N = len(scaf_len), len(dp2scaf)
        self.s0, self.s1 = int(s0), int(s1)
        self.narrow_cols = np.ascontiguousarray(narrow_cols, dtype=np.int64)
        self.wide_mine = np.ascontiguousarray(wide_mine, dtype=np.int64)
        p0, p1 = int(np.searchsorted(dp2scaf, s0, side="left")), int(np.searchsorted(dp2scaf, s1, side="left"))
        # (a) the narrow columns of my scaffolds, renumbered from 0
        so = np.ascontiguousarray(scg_off, dtype=np.int32) if n_scg else None
        if n_scg:
            lo_off = (so[s0:s1 + 1] - so[s0]).astype(np.int32)
            lo_ids = np.ascontiguousarray(np.asarray(scg_ids, dtype=np.int32)[so[s0]:so[s1]])
        else:
            lo_off = lo_ids = None
        nb = np.ascontiguousarray(narrow_bins, dtype=np.int32)
        ctx_n.check(lib().abw_sep_narrow_bins(ctx_n.handle, C.c_void_p(nb.ctypes.data), len(nb)))
        self.store_n = ColumnStore(ctx_n, np.ascontiguousarray(narrow_local), dp2scaf[p0:p1] - s0, scaf_len[s0:s1], lo_off, lo_ids, n_scg)
        # (b) all scaffolds, my wide columns (a rank without one walks a constant column: it never finds a threshold)
        if wide_rows is not None and len(self.wide_mine):
            kw, dims = np.ascontiguousarray(wide_rows), self.wide_mine
        else:
            kw, dims = np.zeros((1, N), dtype=np.int32), np.array([len(self.narrow_cols) + 4096], dtype=np.int64)
        self.store_w = DimShardStore(ctx_w, kw, np.arange(kw.shape[0]), dp2scaf, scaf_len, scg_off, scg_ids, n_scg)
        self.store_w.dims = dims
        self.S = S

    # -- the level calls of the narrow shard
    def begin(self, max_nodes: int = 0):
        self.run_w = DimShardRun(self.store_w, max_nodes)
        self.store_n.ctx.check(lib().abw_sep_stree_begin(self.store_n.ctx.handle, self.s0))

    def level(self):
        info = TreeLevel()
        cap = 256
        while True:
            nodes = np.zeros(cap, dtype=_LEVEL_NODE_DTYPE)
            rc = lib().abw_sep_tree_level(self.store_w.ctx.handle, C.byref(info), C.c_void_p(nodes.ctypes.data), cap)
            if rc == -3 and info.n_nodes > cap:
                cap = int(info.n_nodes)
                continue
            self.store_w.ctx.check(rc)
            break
        self.info, self.nodes = info, np.ascontiguousarray(nodes[:int(info.n_nodes)])
        return self.nodes

    def accum(self):
        """-> (hist int64 tensor, scg_lo, scg_hi int32 tensors or None): this shard's part, to be combined over the shards in place"""
        import torch
        from .kmeans import _DevView
        ex = SepExchange()
        ctx = self.store_n.ctx
        ctx.check(lib().abw_sep_stree_accum(ctx.handle, C.c_void_p(self.nodes.ctypes.data), len(self.nodes), C.c_void_p(self.info.d_scaf_slot), C.byref(ex)))
        dev = torch.device("cuda", ctx.device)
        hist = torch.as_tensor(_DevView(ex.d_hist, ex.hist_count, "<i8"), device=dev) if ex.hist_count else None
        lo = torch.as_tensor(_DevView(ex.d_scg_lo, ex.scg_count, "<i4"), device=dev) if ex.scg_count else None
        hi = torch.as_tensor(_DevView(ex.d_scg_hi, ex.scg_count, "<i4"), device=dev) if ex.scg_count else None
        return hist, lo, hi

    def eval_narrow(self):
        ctx = self.store_n.ctx
        best = np.zeros(len(self.nodes), dtype=_WIN_DTYPE)
        ctx.check(lib().abw_sep_stree_eval(ctx.handle, ClusterSeparatorHip.cluster_ndps_threshold, C.c_void_p(best.ctypes.data), len(best)))
        f = best["found"] != 0
        best["gdim"][f] = self.narrow_cols[best["dim"][f]] + 1
        return best

    def assign_sides(self, winners, owner_is_me):
        """winners: merged records.  -> the walking context's side array (torch uint8 [S], device) with this rank's part filled in"""
        narrow_pos = {int(c) + 1: i for i, c in enumerate(self.narrow_cols)}
        is_narrow = np.array([bool(w["found"]) and int(w["gdim"]) in narrow_pos for w in winners], dtype=bool)
        side = self.run_w.assign(winners, np.asarray(owner_is_me, dtype=bool) & ~is_narrow)
        wn = np.ascontiguousarray(winners.copy())
        wn["dim"] = -1
        for k in np.flatnonzero(is_narrow):
            wn["dim"][k] = narrow_pos[int(winners["gdim"][k])]
        self.store_w.ctx.synchronize()                          # the walking context has cleared the side array
        ctx = self.store_n.ctx
        ctx.check(lib().abw_sep_stree_assign(ctx.handle, C.c_void_p(wn.ctypes.data), len(wn), C.c_void_p(self.info.d_scaf_slot), C.c_void_p(self.info.d_side)))
        ctx.synchronize()
        return side


def make_scaf_shards(ctx_pairs, keys, dp2scaf, scaf_len, scg_off, scg_ids, n_scg):
    """The table [D, N] cut for len(ctx_pairs) ranks: (narrow context, walking context) per rank."""
    keys = np.asarray(keys)
    dp2scaf = np.asarray(dp2scaf)
    world = len(ctx_pairs)
    wide = np.array([bool(keys[d].max() >= 1024 or keys[d].min() < 0) for d in range(keys.shape[0])])
    w_cols, n_cols = np.flatnonzero(wide), np.flatnonzero(~wide)
    S = len(scaf_len)
    b = scaffold_ranges(np.bincount(dp2scaf, minlength=S), world)
    bins = keys[n_cols].max(axis=1) + 1
    out = []
    for r, (cn, cw) in enumerate(ctx_pairs):
        p0, p1 = int(np.searchsorted(dp2scaf, b[r], side="left")), int(np.searchsorted(dp2scaf, b[r + 1], side="left"))
        mine = w_cols[r::world]
        out.append(ScafShard(cn, cw, keys[n_cols][:, p0:p1], keys[mine] if len(mine) else None, dp2scaf, scaf_len, scg_off, scg_ids, n_scg,
                             b[r], b[r + 1], mine, n_cols, bins))
    return out


def binning_stree_local(shards, max_nodes: int = 0):
    """The scaffold-sharded recursion with all ranks driven from one process (tests; the contexts may share a GPU).
    -> ([(rows, scaf2cluster) per rank], bytes one rank exchanges per direction, levels)"""
    import torch
    for sh in shards:
        sh.begin(max_nodes)
    exchanged = levels = 0
    while True:
        bests_w = [sh.run_w.eval() for sh in shards]
        if bests_w[0] is None:
            assert all(b is None for b in bests_w)
            break
        parts = []
        for sh in shards:
            sh.level()
            parts.append(sh.accum())
        for sh in shards:
            sh.store_n.ctx.synchronize()
        if parts[0][0] is not None:
            tot = torch.stack([p[0] for p in parts]).sum(0)
            for p in parts:
                p[0].copy_(tot)
            exchanged += int(tot.numel()) * 8
        if parts[0][1] is not None:
            lo = torch.stack([p[1] for p in parts]).amin(0)
            hi = torch.stack([p[2] for p in parts]).amax(0)
            for p in parts:
                p[1].copy_(lo)
                p[2].copy_(hi)
            exchanged += int(lo.numel()) * 8
        torch.cuda.synchronize()
        mines = [stree_mine(sh.eval_narrow(), bw) for sh, bw in zip(shards, bests_w)]
        winners, owner = merge_winners(mines)
        sides = [sh.assign_sides(winners, owner == k) for k, sh in enumerate(shards)]
        tot = torch.stack(sides).amax(0)
        for t in sides:
            t.copy_(tot)
        torch.cuda.synchronize()
        exchanged += winners.nbytes + int(tot.numel())
        for sh in shards:
            sh.run_w.advance()
        levels += 1
    return [sh.run_w.result() for sh in shards], exchanged, levels


def binning_stree(shard: ScafShard, dist, max_nodes: int = 0):
    """One rank of the scaffold-sharded recursion over torch.distributed (RCCL on GPUs; gloo stages through the host).
    -> (rows, scaf2cluster, seconds, bytes this rank exchanged per direction, levels)"""
    import torch
    from . import multi
    rank, world = dist.get_rank(), dist.get_world_size()
    on_dev = dist.get_backend() == "nccl"
    dev = torch.device("cuda", shard.store_w.ctx.device) if on_dev else torch.device("cpu")
    t0 = time.perf_counter()
    shard.begin(max_nodes)
    exchanged = levels = 0
    while True:
        best_w = shard.run_w.eval()
        k = torch.tensor([0 if best_w is None else len(best_w)], dtype=torch.int64, device=dev)
        ks = [torch.zeros_like(k) for _ in range(world)]
        dist.all_gather(ks, k)
        if best_w is None:
            assert all(int(x) == 0 for x in ks)
            break
        assert all(int(x) == len(best_w) for x in ks), "the ranks walk different trees"
        shard.level()
        hist, lo, hi = shard.accum()
        shard.store_n.ctx.synchronize()
        if hist is not None:                                    # every rank accumulates the same set of nodes: the same shapes everywhere
            multi._all_reduce(hist, dist.ReduceOp.SUM, dist)
        if lo is not None:
            multi._all_reduce(lo, dist.ReduceOp.MIN, dist)
            multi._all_reduce(hi, dist.ReduceOp.MAX, dist)
        torch.cuda.synchronize()
        exchanged += (int(hist.numel()) * 8 if hist is not None else 0) + (int(lo.numel()) * 8 if lo is not None else 0)
        mine = stree_mine(shard.eval_narrow(), best_w)
        mt = torch.from_numpy(mine.view(np.uint8).reshape(-1).copy()).to(dev)
        parts = [torch.zeros_like(mt) for _ in range(world)]
        dist.all_gather(parts, mt)
        winners, owner = merge_winners([p.cpu().numpy().view(_WIN_DTYPE) for p in parts])
        side = shard.assign_sides(winners, owner == rank)
        if on_dev:
            dist.all_reduce(side, op=dist.ReduceOp.MAX)
        else:
            h = side.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.MAX)
            side.copy_(h)
        torch.cuda.synchronize()
        exchanged += mt.numel() * world + side.numel()
        shard.run_w.advance()
        levels += 1
    rows, s2c = shard.run_w.result()
    return rows, s2c, time.perf_counter() - t0, exchanged, levels
