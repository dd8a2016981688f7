"""CPU: the comparator network of rcq::sort_t8 (rc_quarter.h, the threshold rows of reads with up to 128 k-mers) restated in
numpy -- element g of a 16-lane row in register g % 8 of lane g / 8; in-lane compare-exchanges, "flip" stages that pair
register e of lane l with register 7 - e of lane l ^ X (the lane whose bit BIT is clear keeps the minimum) and lane-stride
stages -- must sort: random permutations, inputs with many ties and with the padding value, and 0/1 vectors (a network
that sorts every 0/1 input sorts everything; 2^128 are too many, a few thousand random ones and all thresholded
permutations stand in).  The device code is exercised by every GPU parity test; this pins the network's wiring."""
import numpy as np


def cx(x, a, b):            # registers a, b of every lane
    lo, hi = np.minimum(x[:, a], x[:, b]), np.maximum(x[:, a], x[:, b])
    x[:, a], x[:, b] = lo, hi


def tail(x):                # strides 4, 2, 1
    for pairs in (((0, 4), (1, 5), (2, 6), (3, 7)), ((0, 2), (1, 3), (4, 6), (5, 7)), ((0, 1), (2, 3), (4, 5), (6, 7))):
        for a, b in pairs:
            cx(x, a, b)


def flip(x, X, bit):        # g pairs with g ^ (8 (X + 1) - 1)
    lanes = np.arange(16)
    keep_min = ((lanes >> bit) & 1) == 0
    for e in range(4):
        ya, yb = x[lanes ^ X, 7 - e].copy(), x[lanes ^ X, e].copy()
        x[:, e] = np.where(keep_min, np.minimum(x[:, e], ya), np.maximum(x[:, e], ya))
        x[:, 7 - e] = np.where(keep_min, np.minimum(x[:, 7 - e], yb), np.maximum(x[:, 7 - e], yb))


def lane(x, X, bit):        # stride 8 X
    lanes = np.arange(16)
    keep_min = ((lanes >> bit) & 1) == 0
    for e in range(8):
        y = x[lanes ^ X, e].copy()
        x[:, e] = np.where(keep_min, np.minimum(x[:, e], y), np.maximum(x[:, e], y))


def sort_t8(v):
    x = np.array(v, dtype=np.int64).reshape(16, 8).copy()   # any input layout: the network sorts the multiset
    for a, b in ((0, 1), (2, 3), (4, 5), (6, 7)):
        cx(x, a, b)
    for a, b in ((0, 3), (1, 2), (4, 7), (5, 6), (0, 1), (2, 3), (4, 5), (6, 7)):
        cx(x, a, b)
    for a, b in ((0, 7), (1, 6), (2, 5), (3, 4), (0, 2), (1, 3), (4, 6), (5, 7), (0, 1), (2, 3), (4, 5), (6, 7)):
        cx(x, a, b)
    flip(x, 1, 0); tail(x)
    flip(x, 3, 1); lane(x, 1, 0); tail(x)
    flip(x, 7, 2); lane(x, 2, 1); lane(x, 1, 0); tail(x)
    flip(x, 15, 3); lane(x, 4, 2); lane(x, 2, 1); lane(x, 1, 0); tail(x)
    return x.reshape(128)   # sorted element g = register g % 8 of lane g // 8


def test_transposed_network_sorts():
    rng = np.random.default_rng(5)
    for _ in range(300):
        v = rng.permutation(128)
        assert np.array_equal(sort_t8(v), np.arange(128))
    for _ in range(300):   # ties, negative values (poly-A masked counts are -1), the padding value behind kcnt
        v = rng.integers(-1, 12, size=128)
        n_pad = int(rng.integers(0, 120))
        v[rng.permutation(128)[:n_pad]] = 2147483647
        assert np.array_equal(sort_t8(v), np.sort(v))
    for _ in range(3000):  # 0/1 inputs of every density
        v = (rng.random(128) < rng.random()).astype(np.int64)
        assert np.array_equal(sort_t8(v), np.sort(v))
    p = rng.permutation(128)   # every threshold of one permutation: 129 0/1 inputs that differ in one position each
    for t in range(129):
        v = (p >= t).astype(np.int64)
        assert np.array_equal(sort_t8(v), np.sort(v))
