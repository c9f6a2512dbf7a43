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


def _mix1(h):
    h ^= h >> 16
    h = (h * 0x85EBCA6B) & 0xffffffff
    h ^= h >> 13
    return h


def _child_word(key, ctr1, c, w):
    """the bulk generator of reproduction (round 6's form), restated with Python integers"""
    stream = _mix32(key ^ ((ctr1 * 0x85EBCA77) & 0xffffffff))
    base = _mix32((((c << 8) * 0x9E3779B1) + stream) & 0xffffffff)
    return base if w == 0 else _mix1((base + w * 0x9E3779B1) & 0xffffffff)


def test_child_word_definition_and_avalanche():
    """word 0 of child c in the stream (key, ctr1) = the child's base = mix32((c << 8) * 0x9E3779B1 + mix32(key ^ ctr1 * 0x85EBCA77)), word w >= 1 =
    mix1(base + w * 0x9E3779B1); mix32 = MurmurHash3's 32-bit finaliser (known answers: 0 -> 0, and the published test values below), mix1 its first half"""
    assert _mix32(0) == 0 and _mix32(1) == 0x514E28B7 and _mix32(0xffffffff) == 0x81F16F39
    assert _mix1(0) == 0 and _mix1(1) == ((0x85EBCA6B ^ (0x85EBCA6B >> 13)) & 0xffffffff)
    for key, ctr1, c, w in [(0, 0, 0, 0), (1, 2, 3, 4), (0xdeadbeef, (1234 << 4) | 8, 129, 8), (0xffffffff, 0xffffffff, (1 << 24) - 1, 255), (5, 6, 7, 0), (5, 6, 7, 1)]:
        assert orc.child_word(key, ctr1, c, w) == _child_word(key, ctr1, c, w)
    # avalanche over the inputs the solver varies: flipping the child, the generation or the key flips ~16 of 32 bits of every word (each goes through the
    # child's full mix32); the next word of the SAME child is the one-multiply hash of a Weyl step: 16 on average as well, with a wider spread
    rng = np.random.default_rng(0)
    flips, flips_w = [], []
    for _ in range(300):
        key, ctr1, c, w = int(rng.integers(1 << 32)), int(rng.integers(1 << 20)), int(rng.integers(2, 600)), int(rng.integers(0, 64))
        a = orc.child_word(key, ctr1, c, w)
        for b in (orc.child_word(key, ctr1, c + 1, w), orc.child_word(key, ctr1 + 16, c, w), orc.child_word(key ^ 1, ctr1, c, w)):
            flips.append(bin(a ^ b).count("1"))
        flips_w.append(bin(a ^ orc.child_word(key, ctr1, c, w + 1)).count("1"))
    flips, flips_w = np.array(flips), np.array(flips_w)
    assert abs(flips.mean() - 16.0) < 0.5 and flips.min() >= 4
    assert abs(flips_w.mean() - 16.0) < 0.8 and flips_w.min() >= 3
    # every bit of the word is balanced over the children and words of one generation
    words = np.array([orc.child_word(77, 5 << 4, c, w) for c in range(2, 514) for w in range(8)], dtype=np.uint64)
    for bit in range(32):
        assert abs(((words >> np.uint64(bit)) & np.uint64(1)).mean() - 0.5) < 0.04
    assert len(np.unique(words)) == words.size


def _byte_sum(x):
    return (x & 255) + ((x >> 8) & 255) + ((x >> 16) & 255) + (x >> 24)


def test_counter_gauss_and_uniform_definition():
    # integer-only post-processing, reproduced here with Python integers
    scale = 0.006765875086793228
    assert scale == 1.0 / np.sqrt(21845.0)
    for key, c0, c1 in [(1, 2, 3), (0xdeadbeef, (77 << 8) | 5, (1234 << 4) | 8), (42, 0, 0)]:
        x0, x1 = orc.philox2x32(key, c0, c1)
        for x in (x0, x1, 0, 0xffffffff, 0x0000ffff, 0xffff0000, 0x01020304):
            assert orc.counter_gauss32(x) == (_byte_sum(x) - 510) * scale
        assert orc.counter_uniform(key, c0, c1) == (((x0 << 32) | x1) >> 11) * 2.0 ** -53


def test_counter_gauss_moments():
    z = np.array([orc.counter_child_gauss(7, c, g, 5 << 4) for c in range(2000) for g in range(10)])
    assert abs(z.mean()) < 0.03
    assert abs(z.var() - 1.0) < 0.03
    assert abs(np.mean(z ** 3)) < 0.1
    assert abs(np.mean(z ** 4) - 2.7) < 0.25
    u = np.array([orc.counter_uniform(9, c, 3) for c in range(20000)])
    assert abs(u.mean() - 0.5) < 0.01 and abs(u.var() - 1 / 12) < 0.005 and u.min() >= 0 and u.max() < 1
    # neighbouring children / genes are uncorrelated
    zz = z.reshape(2000, 10)
    assert abs(np.corrcoef(zz[:, 0], zz[:, 1])[0, 1]) < 0.08
    assert abs(np.corrcoef(zz[:-1, 0], zz[1:, 0])[0, 1]) < 0.08


def test_counter_gauss_words_of_one_child_are_uncorrelated():
    """the words of ONE child are one-multiply hashes of a Weyl sequence behind the child's base: the Gaussians of any two genes (all pairs of the first 32),
    and a gene's Gaussian and the child's mutation-rate exponent, are uncorrelated over the children of many generations -- restated with numpy integers"""
    M = np.uint64(0xffffffff)

    def mix32(h):
        h = h & M
        h ^= h >> np.uint64(16)
        h = (h * np.uint64(0x85EBCA6B)) & M
        h ^= h >> np.uint64(13)
        h = (h * np.uint64(0xC2B2AE35)) & M
        h ^= h >> np.uint64(16)
        return h

    def mix1(h):
        h = h & M
        h ^= h >> np.uint64(16)
        h = (h * np.uint64(0x85EBCA6B)) & M
        h ^= h >> np.uint64(13)
        return h

    rng = np.random.default_rng(3)
    zs, es = [], []
    for _ in range(24):
        key, ctr1 = int(rng.integers(1 << 32)), (int(rng.integers(1 << 20)) << 4) | (int(rng.integers(2)) << 3)
        stream = _mix32(key ^ ((ctr1 * 0x85EBCA77) & 0xffffffff))
        c = np.arange(2, 2 + 2048, dtype=np.uint64)
        base = mix32(((c << np.uint64(8)) * np.uint64(0x9E3779B1) + np.uint64(stream)) & M)
        w = mix1((base[:, None] + np.arange(1, 33, dtype=np.uint64)[None, :] * np.uint64(0x9E3779B1)) & M)
        s = sum(((w >> np.uint64(8 * i)) & np.uint64(255)).astype(np.int64) for i in range(4))
        zs.append((s - 510) * 0.006765875086793228)
        es.append((base >> np.uint64(28)).astype(np.int64))
    # (spot check of the vectorised restatement against the oracle)
    assert orc.child_word(key, ctr1, 2, 1) == int(w[0, 0]) and orc.child_word(key, ctr1, 2049, 32) == int(w[-1, -1]) and orc.child_word(key, ctr1, 5, 0) == int(base[3])
    z, e = np.concatenate(zs), np.concatenate(es)
    n = z.shape[0]
    corr = np.corrcoef(z.T)
    off = corr[~np.eye(32, dtype=bool)] * np.sqrt(n)  # z-scores of the 992 pair correlations: standard normal if the words are independent
    assert np.abs(off).max() < 5.0 and 0.85 < off.std() < 1.15
    for g in range(8):
        assert abs(np.corrcoef(e, z[:, g])[0, 1]) * np.sqrt(n) < 4.5
    counts = np.bincount(e, minlength=16)
    assert ((counts - n / 16.0) ** 2 / (n / 16.0)).sum() < 45.0  # chi-square, 15 degrees of freedom


def test_counter_gauss_distribution_shape():
    """the one-word Gaussian (the sum of the word's four bytes: Irwin-Hall, n = 4): unit variance by construction, support +-510 / sqrt(21845) = 3.451 sigma,
    distribution function within 0.84 % of the normal one everywhere (exact enumeration, no sampling), kurtosis 2.70"""
    from math import erf, sqrt
    scale = 1.0 / sqrt(21845.0)
    assert orc.counter_gauss32(0x00000000) == -510 * scale and orc.counter_gauss32(0xffffffff) == 510 * scale
    p = np.ones(256) / 256.0
    d = p
    for _ in range(3):
        d = np.convolve(d, p)
    z = (np.arange(d.size) - 510) * scale
    assert abs((d * z).sum()) < 1e-12 and abs((d * z ** 2).sum() - 1.0) < 1e-9
    assert abs((d * z ** 4).sum() - 2.70) < 0.005
    cdf = np.cumsum(d)
    normal = np.array([0.5 * (1.0 + erf(v / sqrt(2.0))) for v in z])
    assert np.abs(cdf - normal).max() < 8.5e-3
