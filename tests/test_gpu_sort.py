"""-m gpu: the hand-written stable radix sort (csrc/sort.hip) that orders a batch's (table row, entry) pairs by row.
Integer work: bit-exact against numpy's stable argsort. Stability is what makes the row passes' sums run in ascending
entry order, so it is checked explicitly (values = original positions must ascend inside every run of equal keys)."""
import numpy as np
import pytest

import cunvsm_amd as ca

pytestmark = pytest.mark.gpu


def _sort(keys, bits, repeats=1):
    keys = np.ascontiguousarray(keys, dtype=np.int32)
    ko, vo = np.empty_like(keys), np.empty_like(keys)
    ca._lib.check(ca.lib().nvsm_debug_sort(keys.size, bits, keys.ctypes.data, ko.ctypes.data, vo.ctypes.data, repeats, None))
    return ko, vo


def _check(keys, bits, repeats=1):
    ko, vo = _sort(keys, bits, repeats)
    order = np.argsort(keys, kind="stable")
    np.testing.assert_array_equal(vo, order.astype(np.int32))
    np.testing.assert_array_equal(ko, np.asarray(keys, np.int32)[order])


def zipf(rs, rows, n):
    p = 1.0 / np.arange(1, rows + 1)
    p /= p.sum()
    return rs.choice(rows, size=n, p=p)


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 511, 512, 513, 4095, 4096, 4097, 40960, 69632, 512000, 870400, 1234567])
@pytest.mark.parametrize("rows", [1, 2, 3, 500, 512, 513, 50000, 100000, 200000, 262144, 262145, 2000000])
def test_sort_matches_stable_argsort(n, rows):
    if n > 100000 and rows in (2, 3, 513, 262145):
        pytest.skip("covered at smaller n")
    rs = np.random.RandomState(n % 9973 + rows)
    bits = max(1, int(np.ceil(np.log2(rows)))) if rows > 1 else 1
    _check(rs.randint(0, rows, n), bits)


@pytest.mark.parametrize("rows,n", [(50000, 512000), (200000, 40960), (500000, 512000)])
def test_sort_zipf_keys(rows, n):
    """Word ids: more than half of the entries share a few hundred keys (groups spanning whole waves and workgroups)."""
    rs = np.random.RandomState(7)
    _check(zipf(rs, rows, n), int(np.ceil(np.log2(rows))))


def test_sort_degenerate_inputs():
    _check(np.zeros(100000, np.int32), 17)                       # one key
    _check(np.arange(100000, dtype=np.int32), 17)                # already sorted, all distinct
    _check(np.arange(100000, dtype=np.int32)[::-1].copy(), 17)   # reversed
    _check(np.repeat(np.arange(1000), 100), 10)                  # long equal runs
    _check(np.tile(np.arange(1000), 100), 10)                    # period = number of keys
    _check(np.full(5, (1 << 31) - 1, np.int64).astype(np.int32), 31)   # all 31 key bits


def test_sort_workspace_reused_across_calls():
    """The arrival counter of the grid-wide meeting point only grows: fifty sorts on one workspace, no re-zeroing."""
    rs = np.random.RandomState(3)
    _check(rs.randint(0, 100000, 870400), 17, repeats=50)
