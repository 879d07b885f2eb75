"""Corrupt TTSW containers for the C loader's tests (tests/test_cabi.py on libtts_hip.so, tests/test_host_sanitizer.py on the
ASan / UBSan build of the same parser)."""
import struct

import numpy as np


def small_ttsw(tmp_path):
    from text_to_speech_amd import weights
    p = tmp_path / 'ok.ttsw'
    weights.save_ttsw(p, {'a/kernel': np.arange(12, dtype=np.float32).reshape(3, 4), 'a/bias': np.ones(4, np.float32)})
    return p


def corrupt_cases(good: bytes):
    """name -> bytes (None: a path that does not exist).  Offsets of the first entry: name length at 12, name 'a/kernel'
    (8 bytes), ndim at 24, dims at 28 (2 x int64), payload offset at 44, byte count at 52."""
    cases = {}
    cases['missing'] = None
    cases['bad magic'] = b'XXXX' + good[4:]
    cases['bad version'] = good[:4] + struct.pack('<I', 9) + good[8:]
    cases['huge entry count'] = good[:8] + struct.pack('<I', 0xFFFFFFFF) + good[12:]      # would be a 100 GB vector
    cases['truncated header'] = good[:20]
    cases['negative dim'] = good[:28] + struct.pack('<q', -3) + good[36:]
    cases['overflowing dims'] = good[:28] + struct.pack('<qq', 1 << 40, 1 << 40) + good[44:]
    cases['size mismatch'] = good[:52] + struct.pack('<Q', 44) + good[60:]
    cases['payload outside file'] = good[:44] + struct.pack('<Q', 1 << 40) + good[52:]
    cases['zero ndim'] = good[:24] + struct.pack('<I', 0) + good[28:]
    cases['name too long'] = good[:12] + struct.pack('<I', 1 << 30) + good[16:]
    return cases
