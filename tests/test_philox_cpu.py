"""The numpy restatement of the library's normal generator against the Random123 known-answer vectors for
philox4x32-10 (Random123 kat_vectors: counter, key -> output), and basic distribution checks of the Box-Muller stage."""
import numpy as np

from oracle import philox_oracle as P

KAT = [
    ((0x00000000, 0x00000000, 0x00000000, 0x00000000), (0x00000000, 0x00000000),
     (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff), (0xffffffff, 0xffffffff),
     (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


def test_philox4x32_10_known_answers():
    for ctr, key, want in KAT:
        got = P.philox4x32_10(np.array([ctr], dtype=np.uint32), np.array([key], dtype=np.uint32))[0]
        assert tuple(int(v) for v in got) == want, [hex(int(v)) for v in got]


def test_randn_is_counter_based_and_normal():
    a = P.randn(1 << 16, seed=7, stream_id=3)
    assert a.dtype == np.float32 and np.isfinite(a).all()
    assert np.array_equal(P.randn(1001, 7, 3), a[:1001])                   # any prefix / launch shape: same numbers
    assert not np.array_equal(P.randn(1001, 7, 4), a[:1001])               # another draw: another stream
    assert not np.array_equal(P.randn(1001, 8, 3), a[:1001])
    assert abs(float(a.mean())) < 0.02 and abs(float(a.std()) - 1.0) < 0.02
    assert abs(float((a ** 3).mean())) < 0.05 and abs(float((a ** 4).mean()) - 3.0) < 0.15
