"""CPU-only checks of the C-ABI boundary: the library loads, exports every symbol the header
declares, and its host-only entry points agree with the oracle / golden vectors."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import _oracle as orc
from abawaca_amd import _lib, signature, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "abawaca_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(abw_[a-z0-9_]+)\s*\(", header)))
    assert declared == sorted(_lib.EXPORTS)
    L = _lib.lib()
    for name in declared:
        assert hasattr(L, name), name
    assert L.abw_abi_version() >= 1


def test_missing_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.AbwError):
        _lib.Context(0)


def test_dim_names_match_reference_header(golden_dir):
    g = np.load(os.path.join(golden_dir, "sig_edge.npz"))
    names = signature.dim_names()
    assert names[0] == "A" and names[1:] == [str(x) for x in g["header"]]


@pytest.mark.parametrize("case", ["sig_edge", "sig_rand"])
def test_segment_host_matches_reference(golden_dir, case):
    g = np.load(os.path.join(golden_dir, case + ".npz"))
    seg_scaf, st, en = signature.segment_host(g["ascii"], g["scaf_off"])
    names = [str(x) for x in g["scaf_names"]]
    order = sorted(range(len(names)), key=lambda i: names[i].encode())
    idx = np.concatenate([np.flatnonzero(seg_scaf == i) for i in order])
    assert np.array_equal(seg_scaf[idx], g["seg_scaf"].astype(np.uint32))
    assert np.array_equal(st[idx], g["seg_start"]) and np.array_equal(en[idx], g["seg_end"])


def test_segment_host_random_vs_oracle():
    rng = np.random.default_rng(5)
    seqs = []
    for _ in range(60):
        L = int(rng.integers(0, 9000))
        s = synth.ACGT[rng.integers(0, 4, size=L)].copy()
        for _k in range(int(rng.integers(0, 4))):
            if L > 10:
                a = int(rng.integers(0, L - 1)); b = min(L, a + int(rng.integers(1, 2500)))
                s[a:b] = ord("N")
        seqs.append(s)
    blob = np.concatenate(seqs)
    off = np.zeros(len(seqs) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(s) for s in seqs])
    seg_scaf, st, en = signature.segment_host(blob, off)
    exp_scaf, exp_st, exp_en = [], [], []
    for i, s in enumerate(seqs):
        a, b = orc.segments(s)
        exp_scaf += [i] * len(a); exp_st += list(a); exp_en += list(b)
    assert np.array_equal(seg_scaf, np.array(exp_scaf, dtype=np.uint32))
    assert np.array_equal(st, np.array(exp_st)) and np.array_equal(en, np.array(exp_en))


def test_segment_host_rejects_lowercase_n():
    blob = np.frombuffer(b"ACGTnACGT", dtype=np.uint8)
    with pytest.raises(_lib.AbwError) as e:
        signature.segment_host(blob, np.array([0, 9], dtype=np.uint64))
    assert e.value.code == _lib.ABW_ERR_LOWER_N


def test_lrn_header_host_only(golden_dir):
    """abw_lrn_header needs no GPU: the four header lines of the reference's own abawaca.lrn and of the tricky file."""
    from abawaca_amd import lrn
    g = np.load(os.path.join(golden_dir, "lrn_w.npz"))
    ndps, ndims, names = lrn.lrn_header(g["text"])
    assert (ndps, ndims) == g["milli"].shape and names[:4] == ["C", "AA", "AC", "AG"] and len(names) == ndims
    assert names[:179] == signature.dim_names()[1:]                       # the reference's column order
    t = np.load(os.path.join(golden_dir, "lrn_a.npz"))["text"]
    assert lrn.lrn_header(t)[:2] == (17, 7)
    for bad in (b"", b"% 3\n% 2\n", b"x 3\n% 3\n% 9\t1\t1\n% Key\ta\tb\n", b"% 3\n% 3\n% 9\t1\n% Key\ta\tb\n", b"% 3\n% 3\n% 9\t1\t1\n% Key\ta\n"):
        with pytest.raises(_lib.AbwError) as e:
            lrn.lrn_header(np.frombuffer(bad, dtype=np.uint8))
        assert e.value.code == _lib.ABW_ERR_FORMAT


def test_shard_decide_is_the_final_size_check():
    """abw_sep_shard_decide (host only): reference src/ClusterSeparator.cpp:125-132 on the summed child sizes."""
    L = _lib.lib()

    def mk(dim=5, key=123, legal=1):
        o = _lib.Split()
        o.dimension, o.key, o.legal, o.sensitivity, o.specificity, o.ndps1, o.ndps2 = dim, key, legal, 0.9, 0.95, 7, 9
        return o
    o = mk()
    assert L.abw_sep_shard_decide(C.byref(o), 150, 4000, 100) == 0
    assert (o.split, o.ndps1, o.ndps2, o.dimension, o.key) == (1, 150, 4000, 5, 123)
    o = mk()
    L.abw_sep_shard_decide(C.byref(o), 99, 4000, 100)                      # a child too small: everything is reset
    assert (o.split, o.ndps1, o.ndps2, o.dimension, o.key, o.sensitivity, o.specificity) == (0, 0, 0, -1, -1, 0.0, 0.0)
    o = mk(legal=0)
    L.abw_sep_shard_decide(C.byref(o), 500, 500, 100)                      # an illegal winner stays reported, no split
    assert (o.split, o.dimension, o.key, o.ndps1, o.ndps2) == (0, 5, 123, 0, 0)
    o = mk(dim=-1, key=-1, legal=0)
    L.abw_sep_shard_decide(C.byref(o), 500, 500, 100)
    assert o.split == 0


def test_km_chunk_rule_needs_no_gpu_and_is_what_the_adapter_passes():
    """abw_km_chunk_for (no context, no GPU call): multiples of 32, at most 64 tiles, enough chunks to fill the device; the
    values integration/ClusterSeparatorKMeans passes for 256 compute units on the bench shapes stay what round 2's runs used
    (the chunk is part of the result's bits)."""
    L = _lib.lib()
    assert L.abw_km_chunk_for(256, 799859, 64) == L.abw_km_chunk_for(256, 799859, 64) > 0
    assert L.abw_km_chunk_for(256, 3999008, 256) == 1984                     # configs[2] on one GPU (bench: "chunk": 1984)
    for n in (1, 31, 32, 33, 5000, 799859, 3999008, 2 ** 31 - 64):
        for R in (1, 8, 32, 33, 64, 96, 256, 1000):
            for cu in (1, 64, 256, 304):
                c = L.abw_km_chunk_for(cu, n, R)
                assert c >= 32 and c % 32 == 0 and c <= 64 * 32
    assert L.abw_km_chunk_for(0, 100, 8) == 0 and L.abw_km_chunk_for(256, 0, 8) == 0 and L.abw_km_chunk_for(256, 100, 0) == 0
