"""Known-answer tests for the Philox4x32-10 restatement (Random123 kat_vectors)."""
import numpy as np

from oracle.philox import philox4x32_10, randn_block, raw_block

KAT = [
    ((0, 0, 0, 0), (0, 0), (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)),
    ((0xFFFFFFFF,) * 4, (0xFFFFFFFF,) * 2, (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)),
    ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0),
     (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1)),
]


def test_known_answers():
    for ctr, key, want in KAT:
        got = tuple(int(v) for v in philox4x32_10(*ctr, *key))
        assert got == want


def test_stream_layout_is_shard_independent():
    """Any (row range, column range) of the block is reproducible on its own --
    the property that replaces the reference's Omega broadcast
    (activeSubspaceProjector.py:443,551)."""
    full = raw_block(1001, 7, seed=0x1234567890ABCDEF, stream=3)
    assert full.shape == (7, 251, 4)
    sub = philox4x32_10(np.uint64(250), 0, np.uint64(5), 3, 0x90ABCDEF, 0x12345678)
    assert tuple(int(v) for v in sub) == tuple(int(v) for v in full[5, 250])


def test_normal_moments():
    Z = randn_block(200001, 6, seed=11)
    assert Z.shape == (200001, 6) and Z.flags.f_contiguous
    assert abs(Z.mean()) < 5e-3 and abs(Z.std() - 1.0) < 5e-3
    # columns and streams are independent
    assert abs(np.corrcoef(Z[:, 0], Z[:, 1])[0, 1]) < 1e-2
    Z2 = randn_block(200001, 1, seed=11, stream=1)
    assert abs(np.corrcoef(Z[:, 0], Z2[:, 0])[0, 1]) < 1e-2
    # fourth moment of a standard normal is 3
    assert abs((Z ** 4).mean() - 3.0) < 0.05
