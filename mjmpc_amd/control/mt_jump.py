"""Jump-ahead for MT19937 (host side, pure Python integers as GF(2) polynomials).

The twister's only serial part is its recurrence.  To spread one stream over G workgroups, workgroup g
must start from the generator state J_g = g * L words into the stream.  The state transition T is linear
over GF(2) with a primitive characteristic polynomial phi of degree 19937, so T^J = p(T) with
p = t^J mod phi, and, on the word sequence x[n] of the generator,

    x[J + w] = XOR over the set bits i of p of  x[i + w],        w = 0 .. 623

needs only the first 19937 + 624 words - the same for every workgroup.  This module computes phi once
(Berlekamp-Massey on one output bit) and the bit lists of t^J mod phi; the device kernel (noise_mt.hip)
does the XOR reduction and continues the recurrence from there.
"""
import functools

import numpy as np

N, M = 624, 397
DEG = 19937
UPPER, LOWER, MAG = 0x80000000, 0x7FFFFFFF, 0x9908B0DF


def raw_sequence(seed, n_words):
    """x[0 .. n_words): the seeded state (numpy's mt19937_seed / init_genrand) followed by the untempered
    recurrence x[n] = x[n-227] ^ twist(x[n-624], x[n-623]).  Vectorised in blocks of 227."""
    x = np.zeros(max(n_words, N), dtype=np.uint64)
    s = seed & 0xFFFFFFFF
    for pos in range(N):
        x[pos] = s
        s = (1812433253 * (s ^ (s >> 30)) + pos + 1) & 0xFFFFFFFF
    n = N
    while n < n_words:
        m = min(227, n_words - n)
        a, b = x[n - 624:n - 624 + m], x[n - 623:n - 623 + m]
        y = (a & UPPER) | (b & LOWER)
        x[n:n + m] = x[n - 227:n - 227 + m] ^ (y >> np.uint64(1)) ^ np.where(y & np.uint64(1), np.uint64(MAG), np.uint64(0))
        n += m
    return x[:n_words].astype(np.uint32)


def temper(y):
    y = y.astype(np.uint32).copy()
    y ^= y >> np.uint32(11)
    y ^= (y << np.uint32(7)) & np.uint32(0x9D2C5680)
    y ^= (y << np.uint32(15)) & np.uint32(0xEFC60000)
    y ^= y >> np.uint32(18)
    return y


def _bm_fast(bits):
    """Berlekamp-Massey with the sequence kept REVERSED in a Python int (bit j of r = s_{n-j}), so that the
    discrepancy is the parity of C & r - two big-int operations per step."""
    C, B, L, m = 1, 1, 0, 1
    r = 0
    for n, bit in enumerate(bits):
        r = (r << 1) | int(bit)
        d = bin(C & r).count("1") & 1
        if d == 0:
            m += 1
        elif 2 * L <= n:
            T = C
            C ^= B << m
            L = n + 1 - L
            B = T
            m = 1
        else:
            C ^= B << m
            m += 1
    return C, L


@functools.lru_cache(maxsize=1)
def characteristic_polynomial():
    """phi as an int (bit i = coefficient of t^i), degree 19937, such that
    XOR_{i=0..19937} phi_i x[n+i] = 0 for every bit position of the word sequence (n >= 1)."""
    x = raw_sequence(5489, 1 + 2 * DEG + 64)
    bits = (x[1:] & 1).astype(np.uint8)
    C, L = _bm_fast(bits[:2 * DEG + 32])
    if L != DEG:
        raise RuntimeError("Berlekamp-Massey found degree %d, expected %d" % (L, DEG))
    # s_{n+L} = XOR_{i=1..L} c_i s_{n+L-i}  ->  phi_{L-i} = c_i, phi_L = 1
    phi = 0
    for i in range(L + 1):
        if (C >> i) & 1:
            phi |= 1 << (L - i)
    return phi


def _mulmod(a, b, phi):
    """a * b mod phi in GF(2)[t]."""
    r = 0
    while a:
        low = a & -a
        r ^= b << (low.bit_length() - 1)
        a ^= low
    return _mod(r, phi)


def _mod(r, phi):
    d = phi.bit_length() - 1
    while r.bit_length() - 1 >= d:
        r ^= phi << (r.bit_length() - 1 - d)
    return r


@functools.lru_cache(maxsize=64)
def power_of_t(J):
    """t^J mod phi."""
    phi = characteristic_polynomial()
    result, base, e = 1, 2, int(J)          # 2 == the polynomial t
    while e:
        if e & 1:
            result = _mulmod(result, base, phi)
        base = _mulmod(base, base, phi)
        e >>= 1
    return result


HEAD_WORDS = 19968        # generated words of the serial head: >= DEG - 1 so that x[i + w] exists for all i < DEG, w < 624


@functools.lru_cache(maxsize=16)
def jump_tables(segment_words, n_segments, base=0):
    """Set-bit index lists of t^(base + g * segment_words) mod phi for g = 1 .. n_segments-1, concatenated:
    returns (indices int32[total], starts int32[n_segments + 1]); segment 0 needs no jump (empty list)."""
    phi = characteristic_polynomial()
    step = power_of_t(segment_words)
    lists, p = [np.zeros(0, np.int32)], power_of_t(base) if base else 1
    for _ in range(1, n_segments):
        p = _mulmod(p, step, phi)
        b = np.frombuffer(p.to_bytes((DEG + 7) // 8 + 1, "little"), dtype=np.uint8)
        lists.append(np.nonzero(np.unpackbits(b, bitorder="little"))[0].astype(np.int32))
    starts = np.zeros(n_segments + 1, np.int32)
    starts[1:] = np.cumsum([len(x) for x in lists])
    return np.concatenate(lists).astype(np.int32), starts


def plan_segments(total_words, n_segments):
    """(head_words, segment_words, n_segments) covering total_words generated words: a serial head of
    HEAD_WORDS, then n_segments equal segments (the last may be short).  Streams too short to profit
    return n_segments == 0 (the kernel then runs the whole stream as one segment)."""
    rest = total_words - HEAD_WORDS
    if n_segments < 2 or rest < 4 * N * n_segments:
        return total_words, 0, 0
    seg = -(-rest // n_segments)
    return HEAD_WORDS, seg, n_segments


def jump_words(x_head, bit_indices):
    """Host statement of the device reduction: y[w] = XOR_i x_head[i + w], w = 0..623."""
    y = np.zeros(N, np.uint32)
    for i in bit_indices:
        y ^= x_head[i:i + N]
    return y
