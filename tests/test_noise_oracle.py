"""Pins oracle/noise_oracle.py to the published Philox4x32-10 known-answer vectors (Random123 kat_vectors)."""
import numpy as np

from oracle import noise_oracle as NO

KAT = [
    ((0x00000000, 0x00000000, 0x00000000, 0x00000000), (0x00000000, 0x00000000),
     (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff), (0xffffffff, 0xffffffff),
     (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


def test_philox_known_answers():
    for ctr, key, want in KAT:
        got = NO.philox4x32_10(np.array([ctr], dtype=np.uint32), key)[0]
        assert tuple(int(v) for v in got) == want, (ctr, key, [hex(int(v)) for v in got])


def test_noise_is_standard_normal_and_step_dependent():
    a = NO.tiebreak_noise(1234, 0, 2, 64, 96)
    b = NO.tiebreak_noise(1234, 1, 2, 64, 96)
    assert a.shape == (2, 1, 64, 96) and abs(a.mean()) < 0.03 and abs(a.std() - 1) < 0.03
    assert abs(np.corrcoef(a.ravel(), b.ravel())[0, 1]) < 0.03
    assert np.isfinite(a).all()
