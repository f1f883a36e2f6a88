"""Philox4x32-10: the published known-answer vectors (Random123, Salmon et al. SC'11)
pin the oracle's generator to the public algorithm; the uniform construction is
Julia's rand(Float64): (u64 >> 11) * 2^-53."""
import ctypes as ct

import numpy as np

from conftest import mcs, orc

KAT = [
    ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


def _block(lib, ctr, key):
    c = (ct.c_uint32 * 4)(*ctr); k = (ct.c_uint32 * 2)(*key); o = (ct.c_uint32 * 4)()
    lib.orc_philox_block(c, k, o)
    return tuple(o)


def test_philox_known_answers():
    lib = orc.load("det", mcs.capi)
    for ctr, key, want in KAT:
        assert _block(lib, ctr, key) == want


def test_uniform_construction():
    lib = orc.load("det", mcs.capi)
    key, stream = 0x123456789ABC, 7
    for j in range(0, 40):
        u = lib.orc_uniform(key, stream, j)
        blk = j >> 1
        o = _block(lib, (blk & 0xffffffff, blk >> 32, stream, 0), (key & 0xffffffff, key >> 32))
        w = (j & 1) * 2
        u64 = (o[w + 1] << 32) | o[w]
        assert u == (u64 >> 11) * 2.0 ** -53
        assert 0.0 <= u < 1.0
        assert (u * 2.0 ** 53) == int(u * 2.0 ** 53)


def test_uniform_statistics():
    lib = orc.load("det", mcs.capi)
    u = np.array([lib.orc_uniform(42, 0, j) for j in range(20000)])
    assert abs(u.mean() - 0.5) < 0.01 and abs(u.var() - 1 / 12) < 0.005
    # distinct keys give distinct streams
    v = np.array([lib.orc_uniform(43, 0, j) for j in range(2000)])
    assert not np.any(u[:2000] == v)
