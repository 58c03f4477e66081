"""The oracle's photon map (oracle/rt_oracle.c) against the reference's own known answers: the #[cfg(test)] block of
rust/src/photon_map.rs (lines 470-766), transcribed as data.  CPU only."""
import numpy as np

import oracle_lib as orc


def test_left_children():  # photon_map.rs:703-714
    want = {0: 0, 1: 1, 2: 1, 3: 1, 4: 2, 5: 3, 6: 3, 7: 3, 8: 4, 10: 6}
    assert {n: orc.left_children(n) for n in want} == want


def _diag(n, start=0):
    return orc.make_photons([(i, i, i) for i in range(start, start + n)])


def test_balance2():  # photon_map.rs:553-567
    assert orc.photon_balance(_diag(2), (0, 0, 0), (1, 1, 1)) == [1, 0]


def test_balance5():  # photon_map.rs:570-590
    assert orc.photon_balance(_diag(5), (0, 0, 0), (4, 4, 4)) == [3, 1, 4, 0, 2]


def test_balance10():  # photon_map.rs:593-618
    assert orc.photon_balance(_diag(10), (0, 0, 0), (9, 9, 9)) == [6, 3, 8, 1, 5, 7, 9, 0, 2, 4]


def test_balance_same():  # photon_map.rs:621-642: coincident photons - any permutation
    assert sorted(orc.photon_balance(orc.make_photons([(1, 1, 1)] * 5), (1, 1, 1), (1, 1, 1))) == [0, 1, 2, 3, 4]


def test_nearest_neighbor_lookup():  # photon_map.rs:683-700 (indices into the caller's photons; the reference compares the sets)
    ph = _diag(4, start=1)
    found, msq = orc.find_nearest(ph, 2, (2.5, 2.5, 2.5))
    assert sorted(found) == [1, 2] and abs(msq - 0.75) < 1e-12
    found, msq = orc.find_nearest(ph, 2, (0, 0, 0))
    assert sorted(found) == [0, 1] and abs(msq - 12.0) < 1e-12


def test_no_photon_found_gives_unit_radius():  # photon_map.rs:318: unwrap_or(1.0)
    found, msq = orc.find_nearest(orc.make_photons(np.zeros((0, 3))), 3, (0, 0, 0))
    assert found == [] and msq == 1.0


def test_decompress_color():  # photon_map.rs:717-730
    assert np.allclose(orc.decompress_color(1.0, [0, 0, 0]), [0, 0, 0])
    assert np.allclose(orc.decompress_color(1.0, [255, 255, 255]), [1, 1, 1])
    assert np.allclose(orc.decompress_color(1e3, [255, 127, 0]), [1e3, 1e3 * 127.0 / 255.0, 0])


def test_compress_color():  # photon_map.rs:733-750 (assert_almost_eq!: 1e-5)
    for c, want in (((0, 0, 0), (0, 0, 0)), ((1, 0, 0), (1, 0, 0)), ((1e2, 1e3, 1e-3), (98.03922, 1e3, 0.0))):
        got = orc.decompress_color(*orc.compress_color(c))
        assert np.allclose(got, want, atol=1e-4), (c, got)


def test_search_radius_and_brute_force():
    """find_nearest against a brute-force restatement of what locate_photons can reach: with an unlimited radius the n
    nearest photons; the squared radius is the farthest one's."""
    rng = np.random.default_rng(5)
    pos = rng.uniform(-1, 1, (500, 3))
    ph = orc.make_photons(pos)
    p32 = ph["position"].astype(np.float64)
    for q in rng.uniform(-1.2, 1.2, (40, 3)):
        d2 = ((p32 - q) ** 2).sum(axis=1)
        found, msq = orc.find_nearest(ph, 12, q)
        want = np.argsort(d2)[:12]
        assert sorted(found) == sorted(want.tolist())
        assert abs(msq - d2[want[-1]]) < 1e-12
        # a search radius only cuts off far sides of splitting planes: everything within it is still found
        found_r, msq_r = orc.find_nearest(ph, 12, q, max_search_radius=0.25)
        inside = set(np.nonzero(d2 < 0.25 ** 2)[0].tolist())
        assert set(sorted(inside, key=lambda i: d2[i])[:12]) <= set(found_r) or len(inside) > 12
