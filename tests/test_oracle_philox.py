"""CPU: the numpy restatement of Philox4x32-10 (oracle/philox_np.py) against the known-answer vectors Random123 publishes for
`philox4x32 10` (its kat_vectors file; Salmon et al., SC 2011), and the properties of the sampler's uniforms and normals."""
import numpy as np

from oracle import philox_np as P


def test_random123_known_answer_vectors():
    kat = [  # counter (4 words), key (2 words) -> output (4 words)
        ((0x00000000, 0x00000000, 0x00000000, 0x00000000), (0x00000000, 0x00000000), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
        ((0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff), (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
        ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
    ]
    for ctr, key, want in kat:
        got = P.philox4x32_10(np.array(ctr, np.uint32), np.array(key, np.uint32))
        assert tuple(int(x) for x in got) == want, (ctr, [hex(int(x)) for x in got])
    # vectorised = one at a time
    ctr = np.array([k[0] for k in kat], np.uint32)
    key = np.array([k[1] for k in kat], np.uint32)
    assert np.array_equal(P.philox4x32_10(ctr, key), np.array([k[2] for k in kat], np.uint32))


def test_bijection_and_counter_sensitivity():
    rng = np.random.Generator(np.random.SFC64(1))
    ctr = rng.integers(0, 2 ** 32, (4096, 4), dtype=np.uint64).astype(np.uint32)
    key = np.array([123, 456], np.uint32)
    out = P.philox4x32_10(ctr, key)
    assert len({tuple(r) for r in out}) == 4096                                     # distinct counters -> distinct blocks
    flip = ctr.copy()
    flip[:, 2] ^= 1                                                                 # one counter bit flips about half of the output bits
    bits = np.unpackbits((out ^ P.philox4x32_10(flip, key)).view(np.uint8)).mean()
    assert 0.49 < bits < 0.51


def test_sampler_uniforms_and_normals():
    ua, ub, uc, ud = P.quad_uniforms(1234, 7, np.arange(8)[:, None, None], np.arange(512)[None, :, None], np.arange(3)[None, None, :])
    for u, lo_open in ((ua, True), (uc, True), (ub, False), (ud, False)):
        assert (u > 0).all() and (u <= 1).all() if lo_open else ((u >= 0).all() and (u < 1).all())
        assert np.array_equal(u, u.astype(np.float32).astype(np.float64))           # 24 bits: exact in float32
        assert abs(u.mean() - 0.5) < 0.01
    z = P.standard_normal_quads(1234, 7, np.arange(8)[:, None, None], np.arange(2048)[None, :, None], np.arange(3)[None, None, :])
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1.0) < 0.01 and abs((z ** 4).mean() - 3.0) < 0.1
    kn = P.knots(1234, 7, 100, 2, 64, 6, 0.2121)
    assert kn.shape == (2, 64, 6) and kn.dtype == np.float32
    # the global env index keys the stream: env_offset 100 + env 1 == env_offset 101 + env 0; steps and seeds give fresh streams
    assert np.array_equal(kn[1], P.knots(1234, 7, 101, 1, 64, 6, 0.2121)[0])
    assert not np.array_equal(kn, P.knots(1234, 8, 100, 2, 64, 6, 0.2121)) and not np.array_equal(kn, P.knots(1235, 7, 100, 2, 64, 6, 0.2121))
    # the high words of seed and step counter enter the key
    assert not np.array_equal(P.knots(1234 + (1 << 32), 7, 0, 1, 8, 6, 1.0), P.knots(1234, 7, 0, 1, 8, 6, 1.0))
    assert not np.array_equal(P.knots(1234, 7 + (1 << 32), 0, 1, 8, 6, 1.0), P.knots(1234, 7, 0, 1, 8, 6, 1.0))
