"""Counter-based RNG: Random123 known-answer vectors and distribution sanity."""
import numpy as np

from oracle import orc


def test_philox_known_answers():
    # Random123 kat_vectors: philox4x32-10 and philox2x32-10
    assert orc.philox4x32([0, 0], [0, 0, 0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert orc.philox4x32([0xffffffff] * 2, [0xffffffff] * 4) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert orc.philox4x32([0xa4093822, 0x299f31d0], [0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
    assert orc.philox2x32(0, 0, 0) == (0xff1dae59, 0x6cd10df2)
    assert orc.philox2x32(0xffffffff, 0xffffffff, 0xffffffff) == (0x2c3f628b, 0xab4fd7ad)
    assert orc.philox2x32(0x13198a2e, 0x243f6a88, 0x85a308d3) == (0xdd7ce038, 0xf62a4c12)


def _mix32(h):
    h ^= h >> 16
    h = (h * 0x85EBCA6B) & 0xffffffff
    h ^= h >> 13
    h = (h * 0xC2B2AE35) & 0xffffffff
    h ^= h >> 16
    return h


def test_child_word_definition_and_avalanche():
    """the bulk generator of reproduction, restated with Python integers: word w of child c in the stream (key, ctr1) =
    mix32(((c << 8) + w) * 0x9E3779B1 + mix32(key ^ ctr1 * 0x85EBCA77)), mix32 = MurmurHash3's 32-bit finaliser
    (known answers of the finaliser: 0 -> 0, and the published test values below)"""
    assert _mix32(0) == 0 and _mix32(1) == 0x514E28B7 and _mix32(0xffffffff) == 0x81F16F39
    for key, ctr1, c, w in [(0, 0, 0, 0), (1, 2, 3, 4), (0xdeadbeef, (1234 << 4) | 8, 129, 8), (0xffffffff, 0xffffffff, (1 << 24) - 1, 255)]:
        stream = _mix32(key ^ ((ctr1 * 0x85EBCA77) & 0xffffffff))
        want = _mix32(((((c << 8) + w) * 0x9E3779B1) + stream) & 0xffffffff)
        assert orc.child_word(key, ctr1, c, w) == want
    # avalanche over the inputs the solver varies: flipping the child, the word, the generation or the key flips ~16 of 32 bits
    rng = np.random.default_rng(0)
    flips = []
    for _ in range(300):
        key, ctr1, c, w = int(rng.integers(1 << 32)), int(rng.integers(1 << 20)), int(rng.integers(2, 600)), int(rng.integers(0, 64))
        a = orc.child_word(key, ctr1, c, w)
        for b in (orc.child_word(key, ctr1, c + 1, w), orc.child_word(key, ctr1, c, w + 1), orc.child_word(key, ctr1 + 16, c, w), orc.child_word(key ^ 1, ctr1, c, w)):
            flips.append(bin(a ^ b).count("1"))
    flips = np.array(flips)
    assert abs(flips.mean() - 16.0) < 0.5 and flips.min() >= 4
    # every bit of the word is balanced over the children and words of one generation
    words = np.array([orc.child_word(77, 5 << 4, c, w) for c in range(2, 514) for w in range(8)], dtype=np.uint64)
    for bit in range(32):
        assert abs(((words >> np.uint64(bit)) & np.uint64(1)).mean() - 0.5) < 0.04
    assert len(np.unique(words)) == words.size


def test_counter_gauss_and_uniform_definition():
    # integer-only post-processing, reproduced here with Python integers
    for key, c0, c1 in [(1, 2, 3), (0xdeadbeef, (77 << 8) | 5, (1234 << 4) | 8), (42, 0, 0)]:
        x0, x1 = orc.philox2x32(key, c0, c1)
        for x in (x0, x1, 0, 0xffffffff, 0x0000ffff, 0xffff0000):
            k = bin(x & 0xffff).count("1") - 8
            t = (((x >> 16) & 0xff) + (x >> 24)) / 256.0 - 1.0
            assert orc.counter_gauss32(x) == (k + t) * 0.4898979485566356
        assert orc.counter_uniform(key, c0, c1) == (((x0 << 32) | x1) >> 11) * 2.0 ** -53


def test_counter_gauss_moments():
    z = np.array([orc.counter_child_gauss(7, c, g, 5 << 4) for c in range(2000) for g in range(10)])
    assert abs(z.mean()) < 0.03
    assert abs(z.var() - 1.0) < 0.03
    assert abs(np.mean(z ** 3)) < 0.1
    assert abs(np.mean(z ** 4) - 3.0) < 0.25
    u = np.array([orc.counter_uniform(9, c, 3) for c in range(20000)])
    assert abs(u.mean() - 0.5) < 0.01 and abs(u.var() - 1 / 12) < 0.005 and u.min() >= 0 and u.max() < 1
    # neighbouring children / genes are uncorrelated
    zz = z.reshape(2000, 10)
    assert abs(np.corrcoef(zz[:, 0], zz[:, 1])[0, 1]) < 0.08
    assert abs(np.corrcoef(zz[:-1, 0], zz[1:, 0])[0, 1]) < 0.08


def test_counter_gauss_distribution_shape():
    """the one-word Gaussian (Binomial(16, 1/2) lattice + triangular jitter): unit variance by construction, support
    +-9 * 0.4899 = 4.41 sigma, distribution function within 0.5 % of the normal one everywhere (exact enumeration of the
    lattice, no sampling), kurtosis 2.885"""
    from math import comb, erf, sqrt
    scale = 0.4898979485566356
    assert abs(scale - 1.0 / sqrt(4.0 + 1.0 / 6.0)) < 1e-15
    assert orc.counter_gauss32(0x00000000) == (-8 - 1.0) * scale and orc.counter_gauss32(0xffffffff) == (8 + 510 / 256.0 - 1.0) * scale
    # exact moments: k ~ Binomial(16) - 8, t = (a + b) / 256 - 1 with a, b uniform on 0..255
    pk = np.array([comb(16, i) for i in range(17)], dtype=np.float64) / 2.0 ** 16
    k = np.arange(17) - 8.0
    ab = np.add.outer(np.arange(256), np.arange(256)).ravel() / 256.0 - 1.0
    m2 = (pk * k ** 2).sum() + (ab ** 2).mean() + 2 * 0  # independent, E[k] = 0
    m4 = (pk * k ** 4).sum() + 6 * (pk * k ** 2).sum() * (ab ** 2).mean() + (ab ** 4).mean() + 4 * 0
    mean_t = ab.mean()
    assert abs(mean_t + 1.0 / 256.0) < 1e-12  # the jitter is centred up to its own grid (-1/256)
    var = (m2 - mean_t ** 2) * scale ** 2
    assert abs(var - 1.0) < 2e-3
    assert abs(m4 * scale ** 4 / var ** 2 - 2.885) < 0.01
    # distribution function on a grid, exact: P(z <= x) = sum_k pk * P(t <= x / scale - k)
    ts = np.sort(ab)
    for x in np.linspace(-3.5, 3.5, 57):
        cdf = sum(pk[i] * np.searchsorted(ts, x / scale - k[i], side="right") / ts.size for i in range(17))
        assert abs(cdf - 0.5 * (1.0 + erf(x / sqrt(2.0)))) < 5e-3, x
