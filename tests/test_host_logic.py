"""Host-side pieces that need no GPU."""
import pickle

from recurrent_fusion_network_amd.fusion_model import _LazyList


def test_lazy_list_never_shows_a_placeholder():
    """sample_beam's per-image lists fill themselves on first use: every read path of `list` must see the
    filled entries, and it must pickle as a plain list."""
    calls = []

    def mk():
        return _LazyList(2, lambda: calls.append(1) or [5, 6])

    assert len(mk()) == 2 and not calls                     # the length is known without filling
    assert mk() + [7] == [5, 6, 7] and [4] + mk() == [4, 5, 6] and mk() * 2 == [5, 6, 5, 6]
    assert mk().copy() == [5, 6] and list(reversed(mk())) == [6, 5] and mk()[::-1] == [6, 5]
    assert mk().index(6) == 1 and mk().count(5) == 1 and 6 in mk() and sorted(mk()) == [5, 6]
    assert pickle.loads(pickle.dumps(mk())) == [5, 6] and type(pickle.loads(pickle.dumps(mk()))) is list
    lst = mk()
    lst.append(9)
    lst += [1]
    assert lst == [5, 6, 9, 1] and repr(mk()) == '[5, 6]' and mk() != [None, None]
    one = mk()
    n = len(calls)
    assert one[0] == 5 and one[1] == 6 and list(one) == [5, 6] and len(calls) == n + 1      # filled exactly once


def test_lazy_list_has_no_c_level_back_door():
    """a `list` SUBCLASS is read by C fast paths (json's C encoder, str.join, PySequence_Fast) through its item
    array, behind every Python-level hook: an unfilled lazy list showed them its placeholders.  The type is a plain
    MutableSequence now: consumers either go through __iter__ / __getitem__ (and see the filled entries) or refuse it."""
    import json

    import pytest
    assert not isinstance(_LazyList(1, lambda: [0]), list)
    lazy = _LazyList(2, lambda: [0.5, 0.25])
    with pytest.raises(TypeError):
        json.dumps(lazy)                                      # loud, never '[null, null]'
    assert json.dumps(list(_LazyList(2, lambda: [0.5, 0.25]))) == '[0.5, 0.25]'
    assert json.dumps(_LazyList(2, lambda: [0.5, 0.25]).materialize()) == '[0.5, 0.25]'
    assert json.dumps(_LazyList(2, lambda: [0.5, 0.25]), default=list) == '[0.5, 0.25]'
    assert ','.join(_LazyList(2, lambda: ['a', 'b'])) == 'a,b'        # PySequence_Fast on a non-list iterates
    assert tuple(_LazyList(2, lambda: [1, 2])) == (1, 2) and [*_LazyList(2, lambda: [1, 2])] == [1, 2]
    import numpy as np
    assert np.asarray(_LazyList(2, lambda: [1.0, 2.0])).tolist() == [1.0, 2.0]
    with pytest.raises(Exception):
        _LazyList(3, lambda: [1]).materialize()               # a producer that breaks its promise is an error, not a short list
