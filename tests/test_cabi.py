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
    assert lib.tts_hip_abi_version() == 11
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


from ttsw_cases import corrupt_cases, small_ttsw as _small_ttsw     # shared with tests/test_host_sanitizer.py


def test_c_loader_accepts_a_valid_file(lib, tmp_path):
    rc, msg = _check_file(lib, _small_ttsw(tmp_path))
    assert rc == 0 and msg == ''


def test_c_loader_rejects_corrupt_files(lib, tmp_path):
    good = _small_ttsw(tmp_path).read_bytes()
    for what, blob in corrupt_cases(good).items():
        p = tmp_path / 'bad.ttsw'
        if blob is None:
            p = tmp_path / 'does_not_exist.ttsw'
        else:
            p.write_bytes(blob)
        rc, msg = _check_file(lib, p)
        assert rc in (-4, -5), (what, rc, msg)                   # TTS_HIP_EIO / TTS_HIP_ENOMEM
        assert msg, what
