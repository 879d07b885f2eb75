"""CPU: the host-only code that reads untrusted weight files (csrc/ttsw_host.h: the TTSW container parser behind
tts_hip_load_weights / tts_hip_check_weights_file) built with -fsanitize=address,undefined and driven with the corrupt-file
cases of tests/test_cabi.py plus random mutations of a valid file.  A sanitizer report aborts the process (non-zero exit).
GPU AddressSanitizer does not exist on the pool and this code needs no GPU: SURVEY.md section 5's plan."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from ttsw_cases import corrupt_cases, small_ttsw

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'text_to_speech_amd', 'csrc')


@pytest.fixture(scope='module')
def checker():
    if shutil.which('g++') is None:
        pytest.skip('no g++')
    subprocess.run(['bash', os.path.join(CSRC, 'build_host_asan.sh')], check=True, capture_output=True)
    exe = os.path.join(CSRC, 'build_host_asan', 'ttsw_check_asan')
    assert os.path.exists(exe)
    return exe


def _run(exe, paths, load):
    env = dict(os.environ, ASAN_OPTIONS='detect_leaks=1:abort_on_error=0', UBSAN_OPTIONS='print_stacktrace=1')
    r = subprocess.run([exe] + (['--load'] if load else []) + [str(p) for p in paths], capture_output=True, text=True, env=env,
                       timeout=120)
    assert r.returncode == 0, f'sanitizer report or crash (exit {r.returncode}):\n{r.stderr[-4000:]}'
    assert 'runtime error' not in r.stderr and 'AddressSanitizer' not in r.stderr, r.stderr[-4000:]
    lines = r.stdout.strip().splitlines()
    assert len(lines) == len(paths)
    return [(int(l.split(' ', 3)[0]), int(l.split(' ', 3)[1]), int(l.split(' ', 3)[2])) for l in lines]


def test_sanitized_loader_on_valid_and_corrupt_files(checker, tmp_path):
    ok = small_ttsw(tmp_path)
    good = ok.read_bytes()
    assert _run(checker, [ok], load=False) == [(0, 0, 0)]
    assert _run(checker, [ok], load=True) == [(0, 2, 16)]                      # 3 x 4 kernel + 4 biases
    paths = []
    for i, (what, blob) in enumerate(corrupt_cases(good).items()):
        p = tmp_path / f'bad{i}.ttsw'
        if blob is not None:
            p.write_bytes(blob)
        paths.append(p)
    for load in (False, True):
        for (rc, n, _), what in zip(_run(checker, paths, load), corrupt_cases(good)):
            assert rc in (-4, -5) and n == 0, (what, rc)


def test_sanitized_loader_survives_random_mutations(checker, tmp_path):
    """400 mutated copies of a valid file (byte flips in the header region, 32- and 64-bit fields overwritten with extreme
    values, truncations, appended garbage): every one is accepted or refused with an error code, never a sanitizer report."""
    good = bytearray(small_ttsw(tmp_path).read_bytes())
    rng = np.random.default_rng(5)
    header = 12 + (4 + 8 + 4 + 16 + 16) + (4 + 6 + 4 + 8 + 16)                 # entry table of the two tensors
    extremes = [0, 1, 0x7FFFFFFF, 0xFFFFFFFF, 0x80000000, 1 << 40, (1 << 63) - 1, (1 << 64) - 1, 1 << 34]
    paths = []
    for i in range(400):
        b = bytearray(good)
        kind = i % 4
        if kind == 0:
            for _ in range(int(rng.integers(1, 4))):
                b[int(rng.integers(0, header))] = int(rng.integers(0, 256))
        elif kind == 1:
            off = int(rng.integers(8, header - 8))
            v = extremes[int(rng.integers(0, len(extremes)))]
            w = 8 if rng.random() < 0.5 else 4
            b[off:off + w] = int(v & ((1 << (8 * w)) - 1)).to_bytes(w, 'little')
        elif kind == 2:
            b = b[:int(rng.integers(0, len(b)))]
        else:
            b += bytes(rng.integers(0, 256, int(rng.integers(1, 64)), dtype=np.uint8))
        p = tmp_path / f'm{i}.ttsw'
        p.write_bytes(bytes(b))
        paths.append(p)
    seen = set()
    for load in (False, True):
        for rc, _, _ in _run(checker, paths, load):
            assert rc in (0, -4, -5)
            seen.add(rc)
    assert 0 in seen and -4 in seen                                            # some mutations are harmless, most are refused
