"""``gym.utils.seeding.np_random`` re-stated (gym 0.13 - 0.21) so that seeded draws that the reference
makes through ``env.np_random`` (target resets, dynamics randomization) use the same stream.

[EXT] gym is not vendored in the reference tree and is not installed in the build container: this
follows gym's published source from memory and is NOT pinned by a golden vector."""
import hashlib
import struct

import numpy as np


def _bigint_from_bytes(data):
    sizeof_int = 4
    padding = sizeof_int - len(data) % sizeof_int
    data += b"\0" * padding
    count = len(data) // sizeof_int
    unpacked = struct.unpack("{}I".format(count), data)
    return sum(2 ** (sizeof_int * 8 * i) * v for i, v in enumerate(unpacked))


def _int_list_from_bigint(bigint):
    if bigint == 0:
        return [0]
    ints = []
    while bigint > 0:
        bigint, mod = divmod(bigint, 2 ** 32)
        ints.append(mod)
    return ints


def hash_seed(seed, max_bytes=8):
    h = hashlib.sha512(str(seed).encode("utf8")).digest()
    return _bigint_from_bytes(h[:max_bytes])


def np_random(seed):
    """-> (RandomState, seed) like gym.utils.seeding.np_random for a non-negative int seed."""
    if not isinstance(seed, (int, np.integer)) or seed < 0:
        raise ValueError("Seed must be a non-negative integer")
    seed = int(seed) % 2 ** 64
    rng = np.random.RandomState()
    rng.seed(_int_list_from_bigint(hash_seed(seed)))
    return rng, seed
