"""The pure-Python HDF5 reader against files written by the real HDF5 library (tests/golden/make_h5_fixtures.py)."""
import json
import os
import zlib

import numpy as np
import pytest

from text_to_speech_amd.hdf5_reader import H5Error, H5File, read_all

H5 = os.path.join(os.path.dirname(__file__), 'golden', 'h5')
MANIFEST = json.load(open(os.path.join(H5, 'manifest.json')))


def expected(path, shape, dtype):
    """Same closed form as make_h5_fixtures.expected (restated, not imported: that module needs h5py)."""
    n = int(np.prod(shape)) if len(shape) else 1
    seed = zlib.crc32(path.encode())
    v = (np.arange(n, dtype=np.uint64) * np.uint64(2654435761) + np.uint64(seed)) % np.uint64(65521)
    dt = np.dtype(dtype)
    if dt.kind == 'f':
        a = (v.astype(np.float64) / 65521.0 - 0.5).astype(dt)
    else:
        a = (v % np.uint64(100)).astype(dt)
    return a.reshape(shape)


@pytest.mark.parametrize('name', sorted(MANIFEST))
def test_every_dataset_of_the_fixture_reads_back_exactly(name):
    want = MANIFEST[name]
    with H5File(os.path.join(H5, name)) as f:
        found = f.datasets()
        assert sorted(found) == sorted(want)                      # compound types / soft links skipped, nothing else lost
        for path, spec in want.items():
            shape, dtype = tuple(spec[0]), spec[1]
            ds = found[path]
            assert ds.shape == shape
            got = ds.read()
            if dtype == 'str':                                    # fixed- and variable-length strings -> object array of str
                assert got.dtype == object and got.reshape(-1).tolist() == spec[2]
                continue
            assert got.shape == shape and got.dtype == np.dtype(dtype).newbyteorder('=')
            ref = np.zeros(shape, dtype) if len(spec) > 2 else expected(path, shape, dtype)
            np.testing.assert_array_equal(got, ref.astype(got.dtype))


def test_the_wide_group_needs_a_multi_level_btree():
    """Guards the fixture itself: the 420-entry group must not fit one B-tree node (else the recursion is untested)."""
    with H5File(os.path.join(H5, 'wide_and_typed.h5')) as f:
        wide = [p for p in f.datasets() if p.startswith('/wide/')]
        assert len(wide) == 420
        raw = bytes(f._buf)
    levels = {raw[i + 5] for i in range(len(raw) - 8) if raw[i:i + 4] == b'TREE' and raw[i + 4] == 0}
    assert max(levels) >= 1
    chunk_levels = {raw[i + 5] for i in range(len(raw) - 8) if raw[i:i + 4] == b'TREE' and raw[i + 4] == 1}
    assert max(chunk_levels) >= 1


def test_read_by_path_and_read_all():
    p = os.path.join(H5, 'keras_like.h5')
    with H5File(p) as f:
        a = f.read('layers/conv1d_1/vars/0')
        np.testing.assert_array_equal(a, expected('/layers/conv1d_1/vars/0', (5, 8, 8), '<f4'))
        with pytest.raises(KeyError):
            f.read('layers/nope')
    everything = read_all(p)
    assert len(everything) == len(MANIFEST['keras_like.h5'])
    blob = open(p, 'rb').read()
    assert sorted(read_all(blob)) == sorted(everything)              # bytes in memory work too


def test_dense_link_storage_is_refused_with_advice():
    with pytest.raises(H5Error, match='dense link storage.*h5repack'):
        with H5File(os.path.join(H5, 'latest_dense.h5')) as f:
            f.datasets()


def test_garbage_and_truncation_raise_h5error(tmp_path):
    with pytest.raises(H5Error, match='not an HDF5 file'):
        H5File(b'PK\x03\x04' + bytes(4096))
    with pytest.raises(H5Error):
        H5File(b'')
    blob = open(os.path.join(H5, 'keras_like.h5'), 'rb').read()
    for cut in (20, 90, 600, 2000, len(blob) // 2, len(blob) * 3 // 4):
        with pytest.raises(H5Error):
            read_all(blob[:cut])
    # flipped bytes must never escape as anything but H5Error / a clean result
    rng = np.random.default_rng(0)
    for _ in range(200):
        b = bytearray(blob)
        for pos in rng.integers(0, min(len(b), 12000), 4):
            b[pos] = int(rng.integers(0, 256))
        try:
            read_all(bytes(b))
        except H5Error:
            pass
        except (UnicodeDecodeError, zlib.error, MemoryError, OverflowError) as e:   # pragma: no cover
            pytest.fail(f'{type(e).__name__} escaped: {e}')
    p = tmp_path / 'empty.h5'
    p.write_bytes(b'')
    with pytest.raises(H5Error):
        H5File(str(p))
