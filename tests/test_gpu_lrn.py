"""GPU .lrn loader (abw_lrn_parse, row N4) against the golden vectors of the reference's own loader, against the
oracle on a large table, and its error behaviour."""
import ctypes as C
import os

import numpy as np
import pytest

import _oracle as orc
from abawaca_amd import _lib, lrn

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = _lib.Context(0)
    yield c
    c.close()


def test_matches_reference_loader(ctx, golden_dir):
    g = np.load(os.path.join(golden_dir, "lrn_a.npz"))
    names, keys, dims = lrn.lrn_parse(ctx, g["text"])
    assert dims == [f"dim{j}" for j in range(7)] and len(names) == 17
    by_name = {int(nm): k for k, nm in enumerate(names)}
    for nm, row in zip(g["names"], g["values"]):
        assert np.array_equal(keys[:, by_name[int(nm)]] / 1000.0, row)     # key/1000.0 is the double atof gives
    assert np.array_equal(keys[:, by_name[777]], np.full(7, 1000))


def _table_text(rng, N, D, wide_cols=3, crlf=False):
    milli = rng.integers(0, 1001, size=(N, D))
    wide_cols = min(wide_cols, D)
    milli[:, D - wide_cols:] = rng.integers(0, 2_000_000_000, size=(N, wide_cols))
    names = rng.permutation(N) + 1
    import io
    f = io.StringIO()
    f.write(f"% {N}\n% {D + 1}\n% 9" + "\t1" * D + "\n% Key" + "".join(f"\tc{j}" for j in range(D)) + "\n")
    eol = "\r\n" if crlf else "\n"
    for p in range(N):
        f.write(str(names[p]) + "".join("\t%d.%03d" % divmod(int(v), 1000) for v in milli[p]) + eol)
    return np.frombuffer(f.getvalue().encode(), dtype=np.uint8), names, milli


@pytest.mark.parametrize("N,D,crlf", [(5000, 189, False), (300, 1100, True), (300, 1100, False), (4000, 1, False), (1, 2, False)])
def test_matches_oracle_on_tables(ctx, N, D, crlf):
    """rows of ~1.5 kB and of ~7.7 kB (several 1 KiB pieces per line), 10-digit coverage keys"""
    text, names, milli = _table_text(np.random.default_rng(N), N, D, crlf=crlf)
    got_names, keys, dims = lrn.lrn_parse(ctx, text)
    assert np.array_equal(got_names, names) and np.array_equal(keys, milli.T)
    n, ndps, ndims, onames, vals = orc.lrn_parse(text, N + 1)
    assert (n, ndps, ndims) == (N, N, D) and np.array_equal(onames, got_names)
    assert np.array_equal(keys / 1000.0, vals)                              # bit-exact doubles


def test_mixed_lines_take_both_kernels(ctx):
    """Lines the block kernel converts (the newline directly behind the last field) among lines it leaves to the
    line-by-line kernel (trailing blanks, tabs, CR: the reference chomps them, ref src/ClusterData.cpp:119-120), comment
    and empty lines in between (ref :123), several short lines inside one lane's 16 bytes, no newline at the end."""
    rng = np.random.default_rng(5)
    N, D = 3000, 3
    milli = rng.integers(0, 100000, size=(N, D))
    names = rng.permutation(N) + 1
    out = [f"% {N}\n% {D + 1}\n% 9" + "\t1" * D + "\n% Key\ta\tb\tc\n"]
    tails = ["\n", "\n", "\n", " \n", "\t\n", "\r\n", " \t \n", "\n%comment\tline\n", "\n\n", "\n   \n"]
    for p in range(N):
        vals = "".join("\t%d.%03d" % divmod(int(v), 1000) if rng.random() < 0.7 else "\t%d" % (int(v) // 1000) for v in milli[p])
        out.append(str(names[p]) + vals + tails[rng.integers(0, len(tails))])
    text = "".join(out)
    text = text.rstrip("\n") if not text.endswith(" \n") else text
    # (a value printed without decimals reads back as its thousands * 1000: the oracle is the judge of both forms)
    got_names, keys, _ = lrn.lrn_parse(ctx, np.frombuffer(text.encode(), np.uint8))
    n, ndps, ndims, onames, vals = orc.lrn_parse(np.frombuffer(text.encode(), np.uint8), N + 1)
    assert (n, ndps, ndims) == (N, N, D)
    assert np.array_equal(got_names, names) and np.array_equal(onames, names)
    assert np.array_equal(keys / 1000.0, vals)


def test_errors(ctx):
    head = b"% 2\n% 3\n% 9\t1\t1\n% Key\ta\tb\n"
    ok = head + b"1\t0.5\t2\n2\t3.25\t0.001\n"
    names, keys, _ = lrn.lrn_parse(ctx, np.frombuffer(ok, np.uint8))
    assert keys.tolist() == [[500, 3250], [2000, 1]]
    for bad, what in [(head + b"1\t0.5\n2\t3.25\t0.001\n", "fields"), (head + b"1\t0.5\t2\n", "datapoint lines"),
                      (head + b"1\t1e-3\t2\n2\t3.25\t0.001\n", "digits"), (head + b"1\t0.1234\t2\n2\t3.25\t0.001\n", "digits"),
                      (head + b"1\t0.5\t2\n2\t3.25\t0.001\n3\t1\t1\n", "datapoint lines"),
                      (head + b"1\t9999999.999\t2\n2\t3.25\t0.001\n", "32 bits"), (b"% 2\n% 3\n% 9\t1\n% Key\ta\tb\n", "header")]:
        with pytest.raises(_lib.AbwError) as e:
            lrn.lrn_parse(ctx, np.frombuffer(bad, np.uint8))
        assert e.value.code == _lib.ABW_ERR_FORMAT and what in str(e.value)


def test_writer_reproduces_the_reference_file(ctx, golden_dir):
    """abw_lrn_emit on the table abawaca-build computed == the bytes abawaca-build wrote (tests/golden/lrn_w.npz: the
    reference's own abawaca.lrn for a 12-scaffold assembly with two coverage samples)."""
    g = np.load(os.path.join(golden_dir, "lrn_w.npz"))
    want = bytes(g["text"])[int(g["n_header_bytes"]):]
    got = bytes(lrn.lrn_emit(ctx, g["milli"], 1))
    assert got == want
    # and the loader reads the reference's file back to the same table
    names, keys, dims = lrn.lrn_parse(ctx, g["text"])
    assert np.array_equal(names, np.arange(1, len(names) + 1)) and np.array_equal(keys.T, g["milli"])
    assert dims[:3] == ["C", "AA", "AC"] and len(dims) == 181


@pytest.mark.parametrize("N,D,first", [(3000, 189, 1), (40, 960, 999_999_990), (1, 1, 7)])
def test_writer_round_trip_and_format(ctx, N, D, first):
    rng = np.random.default_rng(D)
    m = rng.integers(0, 1001, size=(N, D)).astype(np.int64)
    m[:, -1] = rng.integers(-2_000_000_000, 2_000_000_000, size=N)             # wide and negative values
    m[0, 0] = 0
    text = bytes(lrn.lrn_emit(ctx, m.astype(np.int32), first))
    want = "".join(str(first + r) + "".join("\t%s%d.%03d" % ((("-" if v < 0 else ""),) + divmod(abs(int(v)), 1000)) for v in m[r]) + "\n"
                   for r in range(N)).encode()
    assert text == want
    if first + N < 2 ** 31:
        head = (f"% {N}\n% {D + 1}\n% 9" + "\t1" * D + "\n% Key" + "".join(f"\tc{j}" for j in range(D)) + "\n").encode()
        names, keys, _ = lrn.lrn_parse(ctx, np.frombuffer(head + text, np.uint8))
        assert np.array_equal(names, first + np.arange(N)) and np.array_equal(keys.T, m)


def test_writer_short_and_long_fields_side_by_side(ctx):
    """The one-launch writer stores a field's head as 12 bytes whatever its length: rows of the shortest fields (6 bytes),
    the longest (13), every digit-count boundary and both int32 ends, in every neighbourhood; a buffer that is too small
    answers with the size and nothing behind its end is written."""
    rng = np.random.default_rng(23)
    edge = np.array([0, 1, -1, 9, 10, 999, 1000, -999, -1000, 9999, 10_000, 99_999, 100_000, 999_999, 1_000_000, 9_999_999, 10_000_000,
                     99_999_999, 100_000_000, 999_999_999, 1_000_000_000, 2 ** 31 - 1, -2 ** 31, -2 ** 31 + 1, 2_147_483_000], dtype=np.int64)
    m = edge[rng.integers(0, len(edge), size=(700, 131))]
    m[0, :] = 0                                                                # a row of 6-byte fields
    m[1, :] = -2 ** 31                                                         # a row of 13-byte fields
    m[2, ::2], m[2, 1::2] = 0, -2 ** 31
    m[3, :len(edge)] = edge
    text = lrn.lrn_emit(ctx, m.astype(np.int32), 18_446_744_073_709_550_000)    # 20-digit keys
    want = "".join(str(18_446_744_073_709_550_000 + r) + "".join("\t%s%d.%03d" % ((("-" if v < 0 else ""),) + divmod(abs(int(v)), 1000)) for v in m[r]) + "\n"
                   for r in range(len(m))).encode()
    assert bytes(text) == want
    mi = np.ascontiguousarray(m.astype(np.int32))
    buf = np.full(len(want) + 100, 0xEE, dtype=np.uint8)
    n = C.c_uint64(0)
    rc = _lib.lib().abw_lrn_emit(ctx.handle, C.c_void_p(mi.ctypes.data), 700, 131, 18_446_744_073_709_550_000, C.c_void_p(buf.ctypes.data), len(want) // 2, C.byref(n))
    assert rc == _lib.ABW_ERR_CAPACITY and n.value == len(want) and (buf == 0xEE).all()
    rc = _lib.lib().abw_lrn_emit(ctx.handle, C.c_void_p(mi.ctypes.data), 700, 131, 18_446_744_073_709_550_000, None, 0, C.byref(n))
    assert rc == _lib.ABW_ERR_CAPACITY and n.value == len(want)
    ctx.check(_lib.lib().abw_lrn_emit(ctx.handle, C.c_void_p(mi.ctypes.data), 700, 131, 18_446_744_073_709_550_000, C.c_void_p(buf.ctypes.data), len(buf), C.byref(n)))
    assert bytes(buf[:n.value]) == want and (buf[n.value:] == 0xEE).all()


# ---- cluster writer: ClusterEsomWriter::write_clustering_info_file (reference src/ClusterWriter.cpp:92-98) --------
def test_cluster_writer_reproduces_the_reference_file(ctx, golden_dir):
    g = np.load(os.path.join(golden_dir, "lrn_c.npz"))
    want = bytes(g["text"])[int(g["n_header_bytes"]):]
    assert bytes(lrn.lrn_emit_rows(ctx, g["values"], g["dps"])) == want


def _wild_table(rng, D, N):
    v = rng.random((D, N))
    v[1::4] *= 10.0 ** rng.integers(-9, 12, size=v[1::4].shape)
    v[2::4] = -v[2::4] * 1000
    v[3::4] = rng.integers(0, 4_000_001, size=v[3::4].shape) / 2.0 ** rng.integers(0, 30, size=v[3::4].shape)   # dyadic: exact ties
    v[0, :8] = [0.0, -0.0, 5e-324, -1e-9, 0.9999995, 999999.9999995, 9.2e12, 0.0078125][:min(8, N)] if N >= 8 else v[0, :8]
    return v


@pytest.mark.parametrize("N,D,n_rows", [(4000, 189, 2500), (50, 700, 50), (1, 1, 1), (300, 64, 900),
                                        (600, 300, 700), (90, 500, 333),   # eight rows per workgroup up to 442 columns, four above
                                        (20000, 8, 70001)])       # 4 376 workgroups: the look-back over many windows
def test_cluster_writer_matches_oracle(ctx, N, D, n_rows):
    rng = np.random.default_rng(N + D)
    v = _wild_table(rng, D, N)
    dps = rng.integers(1, N + 1, size=n_rows).astype(np.int64)               # any order, repeats allowed
    got = lrn.lrn_emit_rows(ctx, v, dps)
    want = orc.lrn_emit_rows(v, dps)
    assert np.array_equal(got, want)


def test_cluster_writer_resident_table(ctx):
    """abw_lrn_emit_rows_dev: the table stays on the device, two clusters one after the other (the second smaller)."""
    import torch
    rng = np.random.default_rng(3)
    v = _wild_table(rng, 37, 1000)
    t_v = torch.from_numpy(v).cuda()
    for n_rows in (800, 13):
        dps = rng.permutation(1000)[:n_rows].astype(np.int64) + 1
        t_d = torch.from_numpy(dps).cuda()
        need = lrn.lrn_emit_rows_dev(ctx, t_v, t_d, None)
        t_text = torch.zeros(need, dtype=torch.uint8, device="cuda")
        assert lrn.lrn_emit_rows_dev(ctx, t_v, t_d, t_text) == need
        assert np.array_equal(t_text.cpu().numpy(), orc.lrn_emit_rows(v, dps))
    # three clusters in one call: the row offsets cut the text where each cluster starts
    lists = [rng.permutation(1000)[:n].astype(np.int64) + 1 for n in (300, 1, 450)]
    t_d = torch.from_numpy(np.concatenate(lists)).cuda()
    t_off = torch.zeros(len(t_d) + 1, dtype=torch.int64, device="cuda")
    need = lrn.lrn_emit_rows_dev(ctx, t_v, t_d, None)
    t_text = torch.zeros(need, dtype=torch.uint8, device="cuda")
    lrn.lrn_emit_rows_dev(ctx, t_v, t_d, t_text, t_off)
    off, text, first = t_off.cpu().numpy(), t_text.cpu().numpy(), 0
    assert off[0] == 0 and off[-1] == need
    for dl in lists:
        assert np.array_equal(text[off[first]:off[first + len(dl)]], orc.lrn_emit_rows(v, dl))
        first += len(dl)


def test_cluster_writer_buffer_too_small_or_too_large(ctx):
    """One launch prints and sizes: a buffer that is too small gives ABW_ERR_CAPACITY with the size and nothing behind
    its end is touched; one that is larger than needed is filled up to the size."""
    import torch
    rng = np.random.default_rng(17)
    v = _wild_table(rng, 40, 3000)
    dps = rng.integers(1, 3001, size=5000).astype(np.int64)
    want = orc.lrn_emit_rows(v, dps)
    t_v, t_d = torch.from_numpy(v).cuda(), torch.from_numpy(dps).cuda()
    cap = len(want) // 3
    t_text = torch.full((len(want) + 999,), 0xAB, dtype=torch.uint8, device="cuda")
    n = C.c_uint64(0)
    rc = _lib.lib().abw_lrn_emit_rows_dev(ctx.handle, C.c_void_p(t_v.data_ptr()), 3000, 40, C.c_void_p(t_d.data_ptr()), len(dps),
                                          C.c_void_p(t_text.data_ptr()), cap, C.byref(n), None)
    assert rc == _lib.ABW_ERR_CAPACITY and n.value == len(want)
    assert bool((t_text[cap:] == 0xAB).all())
    assert lrn.lrn_emit_rows_dev(ctx, t_v, t_d, t_text) == len(want)
    got = t_text.cpu().numpy()
    assert np.array_equal(got[:len(want)], want) and (got[len(want):] == 0xAB).all()
    # host form, one pass: a buffer with room to spare
    big = np.full(len(want) + 5000, 0xCD, dtype=np.uint8)
    ctx.check(_lib.lib().abw_lrn_emit_rows(ctx.handle, C.c_void_p(v.ctypes.data), 3000, 40, C.c_void_p(dps.ctypes.data), len(dps),
                                           C.c_void_p(big.ctypes.data), len(big), C.byref(n)))
    assert n.value == len(want) and np.array_equal(big[:len(want)], want) and (big[len(want):] == 0xCD).all()


def test_cluster_writer_resident_device_copy(ctx):
    """abw_lrn_table_load_dev + abw_lrn_emit_rows_dev(values = NULL): the transposed resident copy prints the same bytes
    (widths that are no multiple of the transposition's tiles included)."""
    import torch
    for D, N in ((37, 1000), (189, 2001), (1, 5), (300, 33)):
        rng = np.random.default_rng(D)
        v = _wild_table(rng, D, N)
        t_v = torch.from_numpy(v).cuda()
        dps = rng.integers(1, N + 1, size=777).astype(np.int64)
        t_d = torch.from_numpy(dps).cuda()
        with pytest.raises(_lib.AbwError) as e:
            lrn.lrn_emit_rows_dev(ctx, None, t_d, None, resident=(D + 1, N))               # no such copy
        assert e.value.code == _lib.ABW_ERR_STATE
        shape = lrn.lrn_table_load_dev(ctx, t_v)
        want = orc.lrn_emit_rows(v, dps)
        assert lrn.lrn_emit_rows_dev(ctx, None, t_d, None, resident=shape) == len(want)
        t_text = torch.zeros(len(want), dtype=torch.uint8, device="cuda")
        lrn.lrn_emit_rows_dev(ctx, None, t_d, t_text, resident=shape)
        assert np.array_equal(t_text.cpu().numpy(), want)
        assert np.array_equal(lrn.lrn_emit_rows_dev_host_check(ctx, dps, shape), want)
    lrn.lrn_table_load_dev(ctx, None)
    with pytest.raises(_lib.AbwError):
        lrn.lrn_emit_rows_dev(ctx, None, t_d, None, resident=shape)


def test_cluster_writer_errors(ctx):
    v = np.arange(12, dtype=np.float64).reshape(3, 4)
    assert len(lrn.lrn_emit_rows(ctx, v, np.zeros(0, dtype=np.int64))) == 0
    for dps, bad_v, code, what in [([1, 5], None, _lib.ABW_ERR_ARG, "outside"), ([0], None, _lib.ABW_ERR_ARG, "outside"),
                                   ([2], np.inf, _lib.ABW_ERR_FORMAT, "finite"), ([2], np.nan, _lib.ABW_ERR_FORMAT, "finite"),
                                   ([2], 9.3e12, _lib.ABW_ERR_FORMAT, "9.2e12")]:
        w = v.copy()
        if bad_v is not None:
            w[1, 1] = bad_v
        with pytest.raises(_lib.AbwError) as e:
            lrn.lrn_emit_rows(ctx, w, np.array(dps, dtype=np.int64))
        assert e.value.code == code and what in str(e.value)
    with pytest.raises(_lib.AbwError) as e:                                    # more columns than the staging buffer takes
        lrn.lrn_emit_rows(ctx, np.zeros((800, 2)), np.array([1], dtype=np.int64))
    assert e.value.code == _lib.ABW_ERR_ARG


def test_cluster_writer_host_form_with_resident_table(ctx):
    """abw_lrn_table_load + abw_lrn_emit_rows(values = NULL): what integration/ClusterEsomWriterHip calls."""
    import ctypes as C
    rng = np.random.default_rng(9)
    v = np.ascontiguousarray(_wild_table(rng, 21, 500))
    dps = (rng.permutation(500)[:77] + 1).astype(np.int64)
    L = _lib.lib()
    n = C.c_uint64(0)
    args = (21, C.c_void_p(dps.ctypes.data), len(dps))
    rc = L.abw_lrn_emit_rows(ctx.handle, None, 500, *args, None, 0, C.byref(n))              # nothing resident yet
    assert rc == _lib.ABW_ERR_STATE
    ctx.check(L.abw_lrn_table_load(ctx.handle, C.c_void_p(v.ctypes.data), 500, 21))
    assert L.abw_lrn_emit_rows(ctx.handle, None, 500, *args, None, 0, C.byref(n)) == _lib.ABW_ERR_CAPACITY
    text = np.empty(n.value, dtype=np.uint8)
    ctx.check(L.abw_lrn_emit_rows(ctx.handle, None, 500, *args, C.c_void_p(text.ctypes.data), len(text), C.byref(n)))
    assert np.array_equal(text, orc.lrn_emit_rows(v, dps))
    assert L.abw_lrn_emit_rows(ctx.handle, None, 400, *args, None, 0, C.byref(n)) == _lib.ABW_ERR_STATE   # another shape
    ctx.check(L.abw_lrn_table_load(ctx.handle, None, 0, 0))                                    # unload
    assert L.abw_lrn_emit_rows(ctx.handle, None, 500, *args, None, 0, C.byref(n)) == _lib.ABW_ERR_STATE
