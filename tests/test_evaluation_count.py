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


def test_walked_children_are_the_solvers_own_draws():
    """the host-side count against the ORACLE'S random source asked draw by draw (orc_preselect_children: CounterRandom::preselect_count, what
    orc_evolution.h's generation loop calls), not against a restatement of the formula"""
    seed, first, pop = 0x1234567890, 7, 128
    cum = preselected_children(seed, first, 5, 6, pop)
    assert cum.shape == (5, 7) and np.all(cum[:, 0] == 0)
    for q in range(5):
        total = 0
        for step in range(6):
            for gen in range(8):
                for species in range(2):
                    n = orc.preselect_children(seed, first + q, 0, step, gen, species, pop)
                    assert 1 <= n <= pop - 1
                    total += n
            assert int(cum[q, step + 1]) == total
    # a uniform prefix of 1 ... lambda - 1 children: lambda / 2 on average
    big = preselected_children(1, 0, 512, 16, 128)
    assert abs(float(big[:, -1].mean()) / (16 * 16) - 64.0) < 0.5
