"""The oracle's .lrn loader against the golden vectors the reference's own ClusterData constructor produced
(tests/golden/make_golden_lrn.py -> lrn_a.npz): comment / empty / CRLF / blank-padded lines, 0-6 decimals,
".125", "5.", negative values, rows in any order, a datapoint the .names file does not know."""
import os

import numpy as np

import _oracle as orc


def test_oracle_lrn_matches_reference_loader(golden_dir):
    g = np.load(os.path.join(golden_dir, "lrn_a.npz"))
    n, ndps, ndims, names, vals = orc.lrn_parse(g["text"], 64)
    assert ndps == 17 and n == 17 and ndims == int(g["ndims"])             # the unknown datapoint is a row too (ref :151)
    assert int(g["ndps_loaded"]) == 16
    by_name = {int(nm): k for k, nm in enumerate(names)}
    assert 777 in by_name and 777 not in set(g["names"].tolist())
    for nm, row in zip(g["names"], g["values"]):
        assert np.array_equal(vals[:, by_name[int(nm)]], row)              # bit-exact doubles


def test_oracle_lrn_rejects_wrong_field_count():
    text = b"% 1\n% 3\n% 9\t1\t1\n% Key\ta\tb\n1\t0.5\n"
    assert orc.lrn_parse(np.frombuffer(text, np.uint8), 4)[0] == -2
    assert orc.lrn_parse(np.frombuffer(b"% 1\n% 3\n% 9\t1\n% Key\ta\tb\n", np.uint8), 4)[0] == -1


def test_oracle_cluster_rows_match_reference_writer(golden_dir):
    """abw_or_lrn_emit_rows == the rows the reference's ClusterEsomWriter wrote (tests/golden/lrn_c.npz: ties at the
    seventh decimal, carries into a new digit, -0, subnormals, 17-digit values)."""
    g = np.load(os.path.join(golden_dir, "lrn_c.npz"))
    want = g["text"][int(g["n_header_bytes"]):]
    got = orc.lrn_emit_rows(g["values"], g["dps"])
    assert np.array_equal(got, want)
    assert b"\t0.007812\t" in bytes(want) or b"\t0.007812\n" in bytes(want)    # 0.0078125: the tie goes to the even digit
