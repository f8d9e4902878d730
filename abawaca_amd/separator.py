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


class ColumnStore:
    """The resident feature table (ClusterData + ScafDpData + SCGdb as the separator sees them)."""

    def __init__(self, ctx: Context, keys, dp2scaf, scaf_len, scg_off=None, scg_ids=None, n_scg=0):
        self.ctx = ctx
        self.keys = np.ascontiguousarray(keys, dtype=np.int32)          # [D, N]
        self.D, self.N = self.keys.shape
        self.dp2scaf = np.ascontiguousarray(dp2scaf, dtype=np.int32)
        self.scaf_len = np.ascontiguousarray(scaf_len, dtype=np.int64)
        self.S = len(self.scaf_len)
        self.scaf_ndps = np.bincount(self.dp2scaf, minlength=self.S).astype(np.int32)
        self.scaf_first = np.concatenate([[0], np.cumsum(self.scaf_ndps)[:-1]]).astype(np.int64)
        self.n_scg = int(n_scg)
        so = np.ascontiguousarray(scg_off, dtype=np.int32) if self.n_scg else None
        si = np.ascontiguousarray(scg_ids, dtype=np.int32) if self.n_scg else None
        p = lambda a: C.c_void_p(a.ctypes.data) if a is not None else None
        ctx.check(lib().abw_sep_load(ctx.handle, p(self.keys), self.D, self.N, p(self.dp2scaf), p(self.scaf_len), self.S,
                                     p(so), p(si), self.n_scg))

    def dev_keys_ptr(self):
        return lib().abw_sep_dev_keys(self.ctx.handle)


class ClusterSeparatorHip:
    """One node of the recursion: the drop-in for ClusterSeparatorBySensitivitySpecificity."""
    cluster_ndps_threshold = 100      # reference src/ClusterSeparator.h:72

    def __init__(self, store: ColumnStore, scaf_ids=None):
        self.store = store
        self.scaf_ids = None if scaf_ids is None else np.ascontiguousarray(scaf_ids, dtype=np.int32)
        self._res = None

    def separate(self) -> bool:
        st = self.store
        m = st.S if self.scaf_ids is None else len(self.scaf_ids)
        ids = np.arange(st.S, dtype=np.int32) if self.scaf_ids is None else self.scaf_ids
        n = int(st.scaf_ndps[ids].sum())
        out = Split()
        per = (DimBest * st.D)()
        side = np.zeros(m, dtype=np.uint8)
        raw = np.zeros(n, dtype=np.uint8)
        t0 = time.perf_counter()
        st.ctx.check(lib().abw_sep_node(st.ctx.handle,
                                        C.c_void_p(self.scaf_ids.ctypes.data) if self.scaf_ids is not None else None,
                                        m, self.cluster_ndps_threshold, C.byref(out), per,
                                        C.c_void_p(side.ctypes.data), C.c_void_p(raw.ctypes.data)))
        self.last_call_s = time.perf_counter() - t0          # the C-ABI call alone (host buffers in and out)
        self._res, self._side, self._raw, self._ids = out, side, raw, ids
        self._per = np.array([(p.found, p.key, p.sensitivity, p.specificity) for p in per],
                             dtype=[("found", "i4"), ("key", "i4"), ("sens", "f8"), ("spec", "f8")])
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
        sep = ClusterSeparatorHip(store, ids)
        split = sep.separate()
        all_ids = np.arange(store.S, dtype=np.int32) if ids is None else ids
        n = int(store.scaf_ndps[all_ids].sum())
        row = dict(id=nid, ndps=n, split=int(split), dim=sep.get_separating_dimension(), key=sep.get_separating_value(),
                   spec=sep.get_best_specificity(), sens=sep.get_best_sensitivity(), c1=0, n1=0, c2=0, n2=0)
        if split:
            c1, c2 = next_id, next_id + 1
            next_id += 2
            work[c1], work[c2] = np.sort(sep.get_cluster1()), np.sort(sep.get_cluster2())
            n1, n2 = sep.ndps()
            row.update(c1=c1, n1=n1, c2=c2, n2=n2)
        else:
            scaf2cluster[all_ids] = nid
            for s in all_ids:
                dp2cluster[store.scaf_first[s]:store.scaf_first[s] + store.scaf_ndps[s]] = nid
        nodes.append(row)
    return nodes, scaf2cluster, dp2cluster
