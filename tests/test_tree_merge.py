"""abw_sep_tree_merge (host only): the winner of a node over several contexts' bests follows the reference's comparison of
dimensions -- the larger product sensitivity x specificity, on equal products the smaller dimension of the whole table
(reference src/ClusterSeparatorBySensitivitySpecificity.h:25-30 applied in ascending dimension order,
src/ClusterSeparator.cpp:11-16)."""
import numpy as np

from abawaca_amd import separator


def _rec(found, dim, key, legal, sens, spec, gdim):
    r = np.zeros(1, dtype=separator._WIN_DTYPE)
    r[0] = (found, dim, key, legal, sens, spec, gdim, 0)
    return r


def test_merge_rules():
    none = _rec(0, -1, -1, 0, 0.0, 0.0, -1)
    a = _rec(1, 3, 17, 1, 0.9, 0.95, 40)
    b = _rec(1, 0, 5, 1, 0.95, 0.9, 12)          # the same product as a (commutative), smaller dimension
    c = _rec(1, 1, 9, 0, 0.99, 0.99, 77)         # larger product, illegal: still the winner (legality is judged afterwards)
    # node 0: only context 1 found something; node 1: tie -> smaller global dimension; node 2: larger product; node 3: nobody
    all_best = [np.concatenate([none, a, a, none]), np.concatenate([a, b, c, none]), np.concatenate([none, none, b, none])]
    win, owner = separator.merge_winners(all_best)
    assert owner.tolist() == [1, 1, 1, -1]
    assert (win["gdim"][0], win["key"][0]) == (40, 17)
    assert (win["gdim"][1], win["key"][1], win["dim"][1]) == (12, 5, 0)
    assert (win["gdim"][2], win["legal"][2]) == (77, 0)
    assert win["found"][3] == 0 and win["dim"][3] == -1
    # the order of the contexts does not matter
    win2, owner2 = separator.merge_winners(all_best[::-1])
    assert np.array_equal(win2, win) and owner2.tolist() == [1, 1, 1, -1]
    # products that differ in the last bit are not ties
    lo = _rec(1, 0, 1, 1, 0.9, 0.9, 5)
    hi = _rec(1, 0, 2, 1, np.nextafter(0.9, 1.0), 0.9, 9)
    win3, owner3 = separator.merge_winners([lo, hi])
    assert owner3.tolist() == [1] and win3["gdim"][0] == 9
