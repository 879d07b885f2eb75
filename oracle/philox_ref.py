"""CPU restatement (numpy) of the engine's device-side sampling -- TEST INFRASTRUCTURE ONLY.

The reference draws WaveGlow's noise and the prenet dropout inside `infer` with the Keras backend's generator
(/root/reference/architectures/waveglow_arch.py:272-274,299-302 `keras.random.normal`; tacotron2_arch.py:197-201 dropout),
whose stream is backend-specific; the engine documents its own: Philox4x32-10 (Salmon et al., "Parallel random numbers: as
easy as 1, 2, 3", SC'11; the Random123 known-answer vectors pin this restatement in tests/test_philox.py), key = seed,
block counter = offset + i // 4, element i = word i % 4 of its block (csrc/engine.hip, include/tts_hip.h).
"""
from __future__ import annotations

import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(counter, key):
    """counter uint32 [..., 4], key uint32 [..., 2] -> uint32 [..., 4]."""
    c = [np.asarray(counter[..., i], dtype=np.uint64) for i in range(4)]
    k0 = np.asarray(key[..., 0], dtype=np.uint32).copy()
    k1 = np.asarray(key[..., 1], dtype=np.uint32).copy()
    for _ in range(10):
        p0 = M0 * c[0]
        p1 = M1 * c[2]
        n0 = (p1 >> np.uint64(32)) ^ c[1] ^ k0.astype(np.uint64)
        n1 = p1 & MASK
        n2 = (p0 >> np.uint64(32)) ^ c[3] ^ k1.astype(np.uint64)
        n3 = p0 & MASK
        c = [n0 & MASK, n1, n2 & MASK, n3]
        with np.errstate(over='ignore'):
            k0 = (k0 + W0).astype(np.uint32)
            k1 = (k1 + W1).astype(np.uint32)
    return np.stack([x.astype(np.uint32) for x in c], axis=-1)


def _blocks(n, seed, offset):
    nb = (int(n) + 3) // 4
    ctr = (np.uint64(int(offset) & 0xFFFFFFFFFFFFFFFF) + np.arange(nb, dtype=np.uint64))
    counter = np.zeros((nb, 4), np.uint32)
    counter[:, 0] = (ctr & MASK).astype(np.uint32)
    counter[:, 1] = (ctr >> np.uint64(32)).astype(np.uint32)
    s = np.uint64(int(seed) & 0xFFFFFFFFFFFFFFFF)
    key = np.zeros((nb, 2), np.uint32)
    key[:, 0] = np.uint32(s & MASK)
    key[:, 1] = np.uint32(s >> np.uint64(32))
    return philox4x32_10(counter, key)


def unit_open(x):
    """uint32 word -> float32 in (0, 1): ((x >> 8) + 0.5) * 2^-24"""
    return ((x >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(5.9604644775390625e-8)


def normal(n, seed, offset=0):
    """n float32 N(0, 1) values: Box-Muller on the word pairs (0, 1) and (2, 3) of every block."""
    w = _blocks(n, seed, offset)
    u = unit_open(w)
    out = np.empty((w.shape[0], 4), np.float32)
    for p in range(2):
        r = np.sqrt(np.float32(-2.0) * np.log(u[:, 2 * p]))
        ang = np.float32(6.283185307179586) * u[:, 2 * p + 1]
        out[:, 2 * p] = r * np.cos(ang)
        out[:, 2 * p + 1] = r * np.sin(ang)
    return out.reshape(-1)[:int(n)]


def prenet_masks(n, seed, offset=0):
    """n float32 values in {0, 2}: 2.0 where the word's top bit is set (keep probability 0.5, scale 1 / (1 - 0.5))."""
    w = _blocks(n, seed, offset)
    return ((w >> np.uint32(31)).astype(np.float32) * np.float32(2.0)).reshape(-1)[:int(n)]
