"""bench.py counts the children a solve with secondary goals has walked from the steps the device reports and the counter RNG restated on the host
(bio_ik_amd.workload.preselected_children).  Here: the host's Philox against the oracle's, and the count against the definition, draw by draw."""
import numpy as np

from bio_ik_amd.workload import philox2x32_10, preselected_children
from oracle import orc


def test_host_philox_equals_the_oracles():
    rng = np.random.default_rng(3)
    k, a, b = (rng.integers(0, 1 << 32, 200, dtype=np.uint64) for _ in range(3))
    o0, o1 = philox2x32_10(k, a, b)
    for i in range(200):
        w = orc.philox2x32(int(k[i]), int(a[i]), int(b[i]))
        assert (int(o0[i]), int(o1[i])) == (int(w[0]), int(w[1]))


def test_walked_children_by_definition():
    seed, first, pop = 0x1234567890, 7, 128
    cum = preselected_children(seed, first, 5, 6, pop)
    assert cum.shape == (5, 7) and np.all(cum[:, 0] == 0)
    for q in range(5):
        key = orc.philox2x32(seed & 0xFFFFFFFF, (first + q) & 0xFFFFFFFF, (seed >> 32) ^ ((first + q) >> 32))[0]
        total = 0
        for step in range(6):
            for gen in range(8):
                for species in range(2):
                    o0 = orc.philox2x32(int(key), 0, ((step * 16 + gen) << 4) | (species << 3) | 1)[0]
                    total += int(o0) % (pop - 2 - 1) + 1
            assert int(cum[q, step + 1]) == total
    # a uniform prefix of 1 ... lambda - 1 children: lambda / 2 on average
    big = preselected_children(1, 0, 512, 16, 128)
    assert abs(float(big[:, -1].mean()) / (16 * 16) - 63.0) < 0.5
