"""Direct import of Keras 3 `.weights.h5` checkpoints (reference: custom_train_objects/checkpoint_manager.py:193-195).

The fixtures are tiny Tacotron2 / WaveGlow checkpoints laid out the way `keras.saving.saving_lib._save_state` lays out the
reference's classes, written by the real HDF5 library from a HAND-WRITTEN path table (tests/golden/make_h5_fixtures.py:
keras_tacotron2_paths / keras_waveglow_paths), once per behaviour of the object-tree walk.  The importer must find every
tensor at the table's path, whichever of the two layouts the file has."""
import json
import os

import numpy as np
import pytest

from test_hdf5_reader import H5, expected
from text_to_speech_amd.config import Tacotron2Config, WaveGlowConfig
from text_to_speech_amd.weights import load_ttsw, tacotron2_manifest, waveglow_manifest
from text_to_speech_amd.weights_import import from_keras_h5, keras_h5_layout, main

KERAS = json.load(open(os.path.join(H5, 'keras_manifest.json')))
TINY = {'tacotron2': Tacotron2Config(vocab_size=20, embedding_dim=16, encoder_n_conv=3, prenet_sizes=(8, 8), n_mel_channels=6,
                                     attention_rnn_dim=12, decoder_rnn_dim=12, attention_dim=10, attention_filters=4,
                                     attention_kernel_size=5, postnet_n_conv=5, postnet_filters=14, postnet_kernel_size=5),
        'waveglow': WaveGlowConfig(n_mel_channels=6, n_flows=4, n_group=8, n_early_every=2, n_early_size=2, n_layers=2,
                                   n_channels=8, kernel_size=3, upsample_kernel=16, upsample_stride=4)}


@pytest.mark.parametrize('name', sorted(KERAS))
def test_every_tensor_comes_from_the_path_the_hand_written_table_names(name):
    model = 'tacotron2' if 'tacotron2' in name else 'waveglow'
    tensors = from_keras_h5(os.path.join(H5, name), model, TINY[model])
    manifest = (tacotron2_manifest if model == 'tacotron2' else waveglow_manifest)(TINY[model])
    assert list(tensors) == list(manifest)
    by_tensor = {spec[2]: (path, tuple(spec[0])) for path, spec in KERAS[name].items()}
    assert sorted(by_tensor) == sorted(manifest)
    for tensor, (path, shape) in by_tensor.items():
        assert tuple(manifest[tensor]) == shape
        np.testing.assert_array_equal(tensors[tensor], expected(path, shape, '<f4'), err_msg=f'{tensor} <- {path}')


def test_layout_covers_the_full_size_manifests_exactly_once():
    for model, manifest in (('tacotron2', tacotron2_manifest()), ('waveglow', waveglow_manifest()),
                            ('tacotron2', tacotron2_manifest(Tacotron2Config(speaker_embedding_dim=256)))):
        names = [f'{prefix}/{n}' for prefix, vars_, _ in keras_h5_layout(model) for n in vars_]
        assert sorted(names) == sorted(manifest)
        scopes = [c for _, _, cands in keras_h5_layout(model) for c in cands]
        assert len(scopes) == len(set(scopes))                      # no H5 group can serve two layers


def test_missing_ambiguous_or_misshapen_files_are_refused(tmp_path):
    with pytest.raises(KeyError, match='expected exactly one of the H5 groups'):
        from_keras_h5(os.path.join(H5, 'keras_waveglow_attrs.weights.h5'), 'tacotron2', TINY['tacotron2'])
    with pytest.raises(ValueError, match='converted shape'):
        from_keras_h5(os.path.join(H5, 'keras_tacotron2_attrs.weights.h5'), 'tacotron2')     # full-size config, tiny file
    with pytest.raises(ValueError, match='model must be'):
        from_keras_h5(os.path.join(H5, 'keras_tacotron2_attrs.weights.h5'), 'hifigan')


def test_cli_refuses_a_checkpoint_of_the_wrong_size_and_reports_the_tensor(tmp_path, capsys):
    out = tmp_path / 'x.ttsw'
    with pytest.raises(ValueError, match='tacotron2/encoder/embeddings'):
        main(['--keras-h5-tacotron2', os.path.join(H5, 'keras_tacotron2_walk.weights.h5'), '-o', str(out)])
    assert not out.exists()


def test_roundtrip_to_ttsw(tmp_path):
    from text_to_speech_amd.weights import save_ttsw
    t = from_keras_h5(os.path.join(H5, 'keras_waveglow_walk.weights.h5'), 'waveglow', TINY['waveglow'])
    p = tmp_path / 'wg.ttsw'
    save_ttsw(str(p), t)
    back = load_ttsw(str(p))
    assert list(back) == list(t)
    for k in t:
        np.testing.assert_array_equal(back[k], t[k])
