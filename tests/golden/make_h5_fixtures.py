"""Writes the HDF5 fixtures of tests/test_hdf5_reader.py with the REAL HDF5 library (h5py), so that the pure-Python reader
(text_to_speech_amd/hdf5_reader.py) is checked against files it did not produce itself.

h5py is not importable by the interpreter that runs the tests; the image carries a second interpreter that has it:
    /opt/conda/bin/python3.9 tests/golden/make_h5_fixtures.py          (h5py 3.3.0, HDF5 1.10.6)
Outputs (committed): tests/golden/h5/*.h5 and tests/golden/h5/manifest.json = {file: {dataset path: [shape, dtype]}}.
Dataset contents are a closed-form function of the dataset path (`expected()` below, restated in the test), so no second
copy of the data is stored.
"""
import json
import os
import zlib

import h5py
import numpy as np

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'h5')


def expected(path, shape, dtype):
    n = int(np.prod(shape)) if len(shape) else 1
    seed = zlib.crc32(path.encode())
    v = (np.arange(n, dtype=np.uint64) * np.uint64(2654435761) + np.uint64(seed)) % np.uint64(65521)
    dt = np.dtype(dtype)
    if dt.kind == 'f':
        a = (v.astype(np.float64) / 65521.0 - 0.5).astype(dt)
    else:
        a = (v % np.uint64(100)).astype(dt)
    return a.reshape(shape)


def put(f, manifest, path, shape, dtype='<f4', **kw):
    f.create_dataset(path, data=expected(path, shape, dtype), dtype=dtype, **kw)
    manifest[path] = [list(shape), np.dtype(dtype).str]


def keras_like(f, m):
    """The group structure Keras 3 `save_weights` produces for a small model (nested groups, `vars/<i>` leaves)."""
    put(f, m, '/layers/custom_embedding/vars/0', (11, 8))
    for i, name in enumerate(('conv1d', 'conv1d_1', 'conv1d_2')):
        put(f, m, f'/layers/{name}/vars/0', (5, 8, 8))
        put(f, m, f'/layers/{name}/vars/1', (8,))
    for name in ('batch_normalization', 'batch_normalization_1', 'batch_normalization_2'):
        for j in range(4):
            put(f, m, f'/layers/{name}/vars/{j}', (8,))
    for d in ('forward_layer', 'backward_layer'):
        put(f, m, f'/layers/bidirectional/{d}/cell/vars/0', (8, 16))
        put(f, m, f'/layers/bidirectional/{d}/cell/vars/1', (4, 16))
        put(f, m, f'/layers/bidirectional/{d}/cell/vars/2', (16,))
    f.create_group('/layers/activation/vars')                                # layers without variables leave empty groups
    f.create_group('/vars')
    f.attrs['keras_version'] = '3.3.3'                                       # attributes must be skipped cleanly


TINY_TACOTRON2 = dict(vocab_size=20, embedding_dim=16, encoder_n_conv=3, prenet_sizes=(8, 8), n_mel_channels=6,
                      attention_rnn_dim=12, decoder_rnn_dim=12, attention_dim=10, attention_filters=4,
                      attention_kernel_size=5, postnet_n_conv=5, postnet_filters=14, postnet_kernel_size=5)
TINY_WAVEGLOW = dict(n_mel_channels=6, n_flows=4, n_group=8, n_early_every=2, n_early_size=2, n_layers=2, n_channels=8,
                     kernel_size=3, upsample_kernel=16, upsample_stride=4)


def keras_tacotron2_paths(walk_model_layers):
    """Hand-written statement of where Keras 3 `save_weights` puts each Tacotron2 variable (manifest name -> H5 dataset),
    for both behaviours of the object-tree walk (see text_to_speech_amd/weights_import.py).  Deliberately NOT generated from
    `keras_h5_layout`: the test compares the two."""
    t = {}
    enc, dec = '/encoder/layers', '/decoder'
    post = '/layers/functional_1/layers' if walk_model_layers else '/postnet/layers'
    t['tacotron2/encoder/embeddings'] = f'{enc}/custom_embedding/vars/0'
    for sec, root, n in (('encoder', enc, 3), ('postnet', post, 5)):
        for i in range(n):
            sfx = '' if i == 0 else f'_{i}'
            t[f'tacotron2/{sec}/conv_{i + 1}/kernel'] = f'{root}/conv1d{sfx}/vars/0'
            t[f'tacotron2/{sec}/conv_{i + 1}/bias'] = f'{root}/conv1d{sfx}/vars/1'
            for j, v in enumerate(('gamma', 'beta', 'moving_mean', 'moving_variance')):
                t[f'tacotron2/{sec}/norm_{i + 1}/{v}'] = f'{root}/batch_normalization{sfx}/vars/{j}'
    for d in ('forward', 'backward'):
        for j, v in enumerate(('kernel', 'recurrent_kernel', 'bias')):
            t[f'tacotron2/encoder/bi_lstm/{d}/{v}'] = f'{enc}/bidirectional/{d}_layer/cell/vars/{j}'
    prenet = f'{dec}/layers/tacotron2_prenet' if walk_model_layers else f'{dec}/prenet'
    t['tacotron2/decoder/prenet/layer_0/kernel'] = f'{prenet}/denses/dense/vars/0'
    t['tacotron2/decoder/prenet/layer_1/kernel'] = f'{prenet}/denses/dense_1/vars/0'
    for j, v in enumerate(('kernel', 'recurrent_kernel', 'bias')):
        t[f'tacotron2/decoder/attention_rnn/{v}'] = f'{dec}/cell/attention_rnn/vars/{j}'
        t[f'tacotron2/decoder/decoder_rnn/cell_0/{v}'] = f'{dec}/cell/decoder_rnn/cells/lstm_cell/vars/{j}'
    att = f'{dec}/cell/attention_layer'
    for name in ('query_layer', 'memory_layer', 'value_layer'):
        t[f'tacotron2/decoder/lsa/{name}/kernel'] = f'{att}/{name}/vars/0'
    t['tacotron2/decoder/lsa/location_conv/kernel'] = f'{att}/location_layer/layers/conv1d/vars/0'
    t['tacotron2/decoder/lsa/location_dense/kernel'] = f'{att}/location_layer/layers/dense/vars/0'
    proj = f'{dec}/layers/dense' if walk_model_layers else f'{dec}/linear_projection'
    t['tacotron2/decoder/linear_projection/kernel'] = f'{proj}/vars/0'
    t['tacotron2/decoder/linear_projection/bias'] = f'{proj}/vars/1'
    t['tacotron2/decoder/gate_output/kernel'] = f'{dec}/gate_layer/vars/0'
    t['tacotron2/decoder/gate_output/bias'] = f'{dec}/gate_layer/vars/1'
    return t


def keras_waveglow_paths(walk_model_layers, n_flows=4, n_layers=2):
    t = {}
    up = '/layers/conv1d_transpose' if walk_model_layers else '/upsample'
    t['waveglow/upsample/kernel'], t['waveglow/upsample/bias'] = f'{up}/vars/0', f'{up}/vars/1'
    for k in range(n_flows):
        sfx = '' if k == 0 else f'_{k}'
        t[f'waveglow/invertible_conv-{k}/conv/kernel'] = f'/convinv/invertible1x1_conv{sfx}/conv/vars/0'
        blk = f'/blocks/waveglow_block{sfx}'

        def conv(name, scope):
            t[f'waveglow/block-{k}/{name}/kernel'], t[f'waveglow/block-{k}/{name}/bias'] = f'{scope}/vars/0', f'{scope}/vars/1'
        conv('start_conv', f'{blk}/layers/conv1d' if walk_model_layers else f'{blk}/start')
        conv('end_conv', f'{blk}/end')
        for i in range(n_layers):
            s = '' if i == 0 else f'_{i}'
            conv(f'in_conv-{i}', f'{blk}/in_layers/conv1d{s}')
            conv(f'cond_layer-{i}', f'{blk}/cond_layers/conv1d{s}')
            conv(f'res_skip_conv-{i}', f'{blk}/layers/conv1d_{4 + 3 * i}' if walk_model_layers else f'{blk}/res_skip_layers/conv1d{s}')
    return t


def keras_checkpoints(manifest):
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
    from text_to_speech_amd.config import Tacotron2Config, WaveGlowConfig
    from text_to_speech_amd.weights import tacotron2_manifest, waveglow_manifest
    shapes = {'tacotron2': tacotron2_manifest(Tacotron2Config(**TINY_TACOTRON2)),
              'waveglow': waveglow_manifest(WaveGlowConfig(**TINY_WAVEGLOW))}
    for model, table_fn in (('tacotron2', keras_tacotron2_paths), ('waveglow', keras_waveglow_paths)):
        for walk in (False, True):
            name = f"keras_{model}_{'walk' if walk else 'attrs'}.weights.h5"
            table = table_fn(walk)
            assert sorted(table) == sorted(shapes[model]), sorted(set(table) ^ set(shapes[model]))
            m = manifest[name] = {}
            with h5py.File(os.path.join(HERE, name), 'w') as f:
                for tensor, path in table.items():
                    put(f, m, path, shapes[model][tensor])
                    m[path].append(tensor)                                   # [shape, dtype, manifest name]
                # what else a real file holds: empty `vars` groups of variable-less objects, the seed generator's state
                f.create_group('/vars')
                if model == 'waveglow':
                    f.create_dataset('/seed_generator/vars/0', data=np.array([7, 0], np.uint32))
                else:
                    f.create_dataset('/decoder/prenet/seed_generator/vars/0', data=np.array([42, 0], np.uint32))
                    f.create_group('/encoder/layers/input_layer/vars')
                    f.create_group('/encoder/layers/activation/vars')


def main():
    os.makedirs(HERE, exist_ok=True)
    manifest = {}
    keras_manifest = {}
    keras_checkpoints(keras_manifest)
    with open(os.path.join(HERE, 'keras_manifest.json'), 'w') as fh:
        json.dump(keras_manifest, fh, indent=0, sort_keys=True)

    # 1. default h5py settings (what Keras uses): superblock 0, v1 object headers, symbol-table groups, contiguous data
    m = manifest['keras_like.h5'] = {}
    with h5py.File(os.path.join(HERE, 'keras_like.h5'), 'w') as f:
        keras_like(f, m)

    # 2. a group large enough for a multi-level group B-tree (> 32 leaf nodes) + every dtype / layout the reader claims
    m = manifest['wide_and_typed.h5'] = {}
    with h5py.File(os.path.join(HERE, 'wide_and_typed.h5'), 'w') as f:
        for i in range(420):
            put(f, m, f'/wide/entry_{i:03d}', (3,))
        put(f, m, '/types/f64', (4, 5), '<f8')
        put(f, m, '/types/f16', (7,), '<f2')
        put(f, m, '/types/f32_be', (3, 4), '>f4')
        put(f, m, '/types/i32', (6,), '<i4')
        put(f, m, '/types/u8', (9,), '|u1')
        put(f, m, '/types/i64_be', (2, 2), '>i8')
        put(f, m, '/types/scalar', (), '<f4')
        put(f, m, '/types/empty', (0, 4), '<f4')
        put(f, m, '/layouts/chunked', (37, 10), '<f4', chunks=(8, 4))
        put(f, m, '/layouts/chunked_gzip', (50, 12), '<f4', chunks=(16, 5), compression='gzip')
        put(f, m, '/layouts/chunked_gzip_shuffle', (33, 7), '<f4', chunks=(10, 7), compression='gzip', shuffle=True)
        put(f, m, '/layouts/chunked_fletcher', (20,), '<f8', chunks=(6,), fletcher32=True)
        put(f, m, '/layouts/chunked_many', (40, 40), '<f4', chunks=(2, 2))    # 400 chunks: multi-level chunk B-tree
        put(f, m, '/layouts/resizable', (5, 3), '<f4', maxshape=(None, 3))
        f.create_dataset('/layouts/unwritten', shape=(4, 2), dtype='<f4')    # allocated late: never written -> zeros
        m['/layouts/unwritten'] = [[4, 2], '<f4', 'zeros']
        f['/types/string'] = 'not numeric'                                   # scalar variable-length string
        m['/types/string'] = [[], 'str', ['not numeric']]
        words = ['alpha', '', 'bêta', 'gamma delta', 'e' * 300]              # empty, non-ASCII and long entries
        f.create_dataset('/types/vlen_strings', data=np.array(words, dtype=object), dtype=h5py.string_dtype())
        m['/types/vlen_strings'] = [[5], 'str', words]
        f.create_dataset('/types/fixed_strings', data=np.array([b'ab', b'', b'wxyz'], dtype='S4'))
        m['/types/fixed_strings'] = [[3], 'str', ['ab', '', 'wxyz']]
        f.create_dataset('/types/compound', data=np.zeros(2, dtype=[('a', '<f4'), ('b', '<i4')]))   # skipped by datasets()
        f['/soft'] = h5py.SoftLink('/types/f64')                             # not followed

    # 3. libver='latest': superblock 3, version-2 object headers, link messages, layout version 4
    m = manifest['latest_compact.h5'] = {}
    with h5py.File(os.path.join(HERE, 'latest_compact.h5'), 'w', libver='latest') as f:
        put(f, m, '/layers/dense/vars/0', (6, 4))
        put(f, m, '/layers/dense/vars/1', (4,))
        put(f, m, '/layers/dense_1/vars/0', (4, 2))
        put(f, m, '/single_chunk', (6, 6), '<f4', chunks=(6, 6))
        put(f, m, '/single_chunk_gzip', (9, 5), '<f4', chunks=(9, 5), compression='gzip')
        put(f, m, '/compact_data', (3,), '<f4')

    # 4. libver='latest' with more links than the compact limit (8): dense storage -> the reader must refuse with a clear message
    with h5py.File(os.path.join(HERE, 'latest_dense.h5'), 'w', libver='latest') as f:
        for i in range(12):
            f.create_dataset(f'/many/d{i}', data=np.zeros(2, np.float32))

    # 4b. a speaker-embedding file the way the reference writes one (utils/file_utils.py:358-397 dump_h5 of a DataFrame:
    #     one dataset per column, strings as variable-length strings, ragged vectors padded with -1)
    m = manifest['embeddings_ref_format.h5'] = {}
    with h5py.File(os.path.join(HERE, 'embeddings_ref_format.h5'), 'w') as f:
        ids = ['siwis', 'siwis', 'bob', 'alice', 'bob']
        files = [f'wavs/{i}_{k}.wav' for k, i in enumerate(ids)]
        put(f, m, '/embedding', (5, 16))
        f.create_dataset('/id', data=np.array(ids, dtype=object), dtype=h5py.string_dtype())
        f.create_dataset('/filename', data=np.array(files, dtype=object), dtype=h5py.string_dtype())
        m['/id'] = [[5], 'str', ids]
        m['/filename'] = [[5], 'str', files]

    # 5. a user block in front of the superblock (superblock at offset 512)
    m = manifest['userblock.h5'] = {}
    with h5py.File(os.path.join(HERE, 'userblock.h5'), 'w', userblock_size=512) as f:
        put(f, m, '/a/b', (5,))

    with open(os.path.join(HERE, 'manifest.json'), 'w') as fh:
        json.dump(manifest, fh, indent=0, sort_keys=True)
    for name in sorted(os.listdir(HERE)):
        print(name, os.path.getsize(os.path.join(HERE, name)))


def write_full_size(out_dir, walk=False, model='tacotron2', vocab=148, speaker_dim=0):
    """Full-size checkpoint (seeded synthetic weights) in the Keras layout + the same tensors as .npz: used by the GPU tests of
    `pretrained.load_model` (tests/test_pretrained_gpu.py), which run this file with the h5py interpreter."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
    from text_to_speech_amd.config import Tacotron2Config, WaveGlowConfig
    from text_to_speech_amd.weights import synth_tacotron2, synth_waveglow
    if model == 'tacotron2':
        w = synth_tacotron2(Tacotron2Config(vocab_size=vocab, speaker_embedding_dim=speaker_dim), seed=4321)
        table = keras_tacotron2_paths(walk)
    else:
        w = synth_waveglow(WaveGlowConfig(), seed=4321)
        table = keras_waveglow_paths(walk, n_flows=12, n_layers=8)
    assert sorted(table) == sorted(w)
    os.makedirs(out_dir, exist_ok=True)
    with h5py.File(os.path.join(out_dir, 'ckpt-0000.weights.h5'), 'w') as f:
        for tensor, path in table.items():
            f.create_dataset(path, data=w[tensor])
    np.savez(os.path.join(out_dir, 'tensors.npz'), **{k.replace('/', '|'): v for k, v in w.items()})


if __name__ == '__main__':
    import sys
    if len(sys.argv) >= 3 and sys.argv[1] in ('--full-tacotron2', '--full-waveglow'):
        rest = sys.argv[3:]
        vocab = int(rest[rest.index('--vocab') + 1]) if '--vocab' in rest else 148
        write_full_size(sys.argv[2], walk='walk' in rest, model=sys.argv[1][len('--full-'):], vocab=vocab)
    else:
        main()
