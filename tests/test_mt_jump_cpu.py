"""CPU: MT19937 jump-ahead tables (mjmpc_amd/control/mt_jump.py) against the stream itself and numpy."""
import numpy as np

from mjmpc_amd.control import mt_jump


def test_raw_sequence_is_numpys_generator():
    seed = 20231
    x = mt_jump.raw_sequence(seed, 624 + 2000)
    st = np.random.RandomState(seed).get_state()
    np.testing.assert_array_equal(x[:624], st[1])                   # seeded state
    want = np.random.RandomState(seed).randint(0, 2 ** 32, size=2000, dtype=np.uint64).astype(np.uint32)
    np.testing.assert_array_equal(mt_jump.temper(x[624:]), want)    # tempered outputs = random_uint32 stream


def test_characteristic_polynomial_annihilates_the_word_sequence():
    phi = mt_jump.characteristic_polynomial()
    assert phi.bit_length() - 1 == mt_jump.DEG
    bits = np.nonzero(np.unpackbits(np.frombuffer(phi.to_bytes(2496, "little"), np.uint8), bitorder="little"))[0]
    x = mt_jump.raw_sequence(99, 624 + 21000)
    for n in (1, 17, 1000):
        assert np.bitwise_xor.reduce(x[n + bits]) == 0
    # word 0 of the seeded state only contributes its top bit to the stream
    assert np.bitwise_xor.reduce(x[0 + bits]) & 0x80000000 == 0


def test_jump_tables_reproduce_the_state_ahead():
    head, seg, nseg = mt_jump.plan_segments(400000, 5)
    assert (head, nseg) == (mt_jump.HEAD_WORDS, 5) and head + seg * nseg >= 400000
    idx, starts = mt_jump.jump_tables(seg, nseg, head)
    assert starts[0] == starts[1] == 0 and idx.max() < mt_jump.DEG
    x = mt_jump.raw_sequence(4242, head + seg * (nseg - 1) + 700)
    for g in (1, 4):
        J = head + g * seg
        y = mt_jump.jump_words(x, idx[starts[g]:starts[g + 1]])
        np.testing.assert_array_equal(y[1:], x[J + 1:J + 624])
        assert (int(y[0]) ^ int(x[J])) & 0x80000000 == 0            # only the top bit of word 0 is state


def test_short_streams_are_not_segmented():
    assert mt_jump.plan_segments(30000, 32) == (30000, 0, 0)
    assert mt_jump.plan_segments(10 ** 6, 0)[2] == 0
