import numpy as np

from clonealign_amd.rng import EpsStream, normal_draw, philox4x32

# Random123 known-answer vectors for philox4x32-10 (kat_vectors)
KAT = [
    ([0, 0, 0, 0], (0, 0), "6627e8d5 e169c58d bc57ac4c 9b00dbd8"),
    ([0xFFFFFFFF] * 4, (0xFFFFFFFF, 0xFFFFFFFF), "408f276d 41c83b0e a20bc7c6 6d5451fd"),
    ([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], (0xA4093822, 0x299F31D0), "d16cfe09 94fdcceb 5001e420 24126ea1"),
]


def test_philox_known_answers():
    for ctr, key, want in KAT:
        got = " ".join("%08x" % x for x in philox4x32(np.array(ctr, dtype=np.uint32), key))
        assert got == want


def test_normal_stream_moments_and_determinism():
    z = normal_draw(99, 3, 200001)
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1) < 0.01
    assert np.array_equal(z, normal_draw(99, 3, 200001))
    assert not np.array_equal(z[:100], normal_draw(99, 4, 100))
    s = EpsStream(5, 2, 7)
    a = s.next()
    assert a.shape == (2, 7) and s.draw == 1
    assert np.array_equal(EpsStream(5, 2, 7).block(3)[0], a)


def test_cpp_twin_in_engine_library_is_bit_identical():
    from clonealign_amd.engine import eps_draw
    for seed, draw, n in [(0, 0, 5), (123, 5, 1001), (2**40 + 17, 2**33, 64)]:
        assert np.array_equal(eps_draw(seed, draw, n), normal_draw(seed, draw, n))
