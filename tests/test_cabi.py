"""CPU: the C-ABI library loads and exports exactly the symbols include/tts_hip.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'tts_hip.h')


def _declared():
    src = open(HEADER).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(tts_hip_\w+)\s*\(', src)))


@pytest.fixture(scope='module')
def lib():
    from text_to_speech_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _lib.load_library()


def test_header_and_binding_agree(lib):
    from text_to_speech_amd import _lib
    declared = _declared()
    assert len(declared) >= 14
    assert declared == sorted(_lib.SIGNATURES), 'ctypes SIGNATURES must list exactly the header\'s functions'


def test_every_declared_symbol_is_exported(lib):
    for name in _declared():
        assert hasattr(lib, name), name


def test_abi_version_and_null_handle_errors(lib):
    assert lib.tts_hip_abi_version() == 9
    assert lib.tts_hip_destroy(None) == -1                       # TTS_HIP_EINVAL, no crash
    assert lib.tts_hip_finalize(None) == -1
    assert lib.tts_hip_has_model(None, b'waveglow') == 0
    assert lib.tts_hip_last_error(None) == b'null engine'


def test_create_without_gpu_reports_error_not_crash(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    h = ctypes.c_void_p()
    rc = lib.tts_hip_create(0, ctypes.byref(h))
    assert rc != 0 and not h.value
    from text_to_speech_amd.engine import HipEngine
    from text_to_speech_amd import HipLibraryError
    with pytest.raises(HipLibraryError):
        HipEngine(0)


# ---- the C weight-file loader must reject corrupt files with an error code, never crash or allocate from untrusted sizes
def _check_file(lib, path):
    buf = ctypes.create_string_buffer(512)
    rc = lib.tts_hip_check_weights_file(str(path).encode(), buf, len(buf))
    return rc, buf.value.decode()


def _small_ttsw(tmp_path):
    import numpy as np
    from text_to_speech_amd import weights
    p = tmp_path / 'ok.ttsw'
    weights.save_ttsw(p, {'a/kernel': np.arange(12, dtype=np.float32).reshape(3, 4), 'a/bias': np.ones(4, np.float32)})
    return p


def test_c_loader_accepts_a_valid_file(lib, tmp_path):
    rc, msg = _check_file(lib, _small_ttsw(tmp_path))
    assert rc == 0 and msg == ''


def test_c_loader_rejects_corrupt_files(lib, tmp_path):
    import struct
    good = _small_ttsw(tmp_path).read_bytes()
    cases = {}
    cases['missing'] = None
    cases['bad magic'] = b'XXXX' + good[4:]
    cases['bad version'] = good[:4] + struct.pack('<I', 9) + good[8:]
    cases['huge entry count'] = good[:8] + struct.pack('<I', 0xFFFFFFFF) + good[12:]      # would be a 100 GB vector
    cases['truncated header'] = good[:20]
    # first entry: name length at 12, name 'a/kernel' (8 bytes), ndim at 24, dims at 28 (2 x int64), off at 44, nbytes at 52
    cases['negative dim'] = good[:28] + struct.pack('<q', -3) + good[36:]
    cases['overflowing dims'] = good[:28] + struct.pack('<qq', 1 << 40, 1 << 40) + good[44:]
    cases['size mismatch'] = good[:52] + struct.pack('<Q', 44) + good[60:]
    cases['payload outside file'] = good[:44] + struct.pack('<Q', 1 << 40) + good[52:]
    cases['zero ndim'] = good[:24] + struct.pack('<I', 0) + good[28:]
    cases['name too long'] = good[:12] + struct.pack('<I', 1 << 30) + good[16:]
    for what, blob in cases.items():
        p = tmp_path / 'bad.ttsw'
        if blob is None:
            p = tmp_path / 'does_not_exist.ttsw'
        else:
            p.write_bytes(blob)
        rc, msg = _check_file(lib, p)
        assert rc in (-4, -5), (what, rc, msg)                   # TTS_HIP_EIO / TTS_HIP_ENOMEM
        assert msg, what
