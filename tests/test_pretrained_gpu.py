"""`pretrained.load_model` end to end on the GPU: a FULL-SIZE Tacotron2 checkpoint in the Keras `.weights.h5` layout, written
by the real HDF5 library (the image's second interpreter has h5py), opened as a reference model directory, must synthesize
exactly what an engine loaded with the same tensors through the ordinary path does."""
import json
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

H5PY_PYTHON = '/opt/conda/bin/python3.9'
GEN = os.path.join(os.path.dirname(__file__), 'golden', 'make_h5_fixtures.py')


def test_model_directory_with_a_full_size_keras_checkpoint(tmp_path):
    if not os.path.exists(H5PY_PYTHON):
        pytest.skip('no interpreter with h5py on this box')
    d = tmp_path / 'pretrained_tacotron2'
    save = d / 'saving'
    env = {k: v for k, v in os.environ.items() if k not in ('PYTHONPATH', 'PYTHONHOME')}
    done = subprocess.run([H5PY_PYTHON, GEN, '--full-tacotron2', str(save), 'walk'], env=env, capture_output=True, text=True,
                          cwd=str(tmp_path))
    if done.returncode != 0:
        pytest.skip(f'h5py interpreter could not write the checkpoint: {done.stderr[-300:]}')
    (d / 'config.json').write_text(json.dumps({'class_name': 'Tacotron2', 'config': {'name': d.name, 'lang': 'en'}}))
    (save / 'checkpoint.json').write_text(json.dumps({'counter': 1, 'loaded': -1,
                                                      'checkpoints': [{'epoch': 0, 'step': 0, 'counter': 0}]}))
    from text_to_speech_amd import pretrained
    from text_to_speech_amd.engine import HipEngine
    model = pretrained.load_model(str(d), reload=True)
    assert os.path.exists(save / 'ckpt-0000.ttsw')
    z = np.load(save / 'tensors.npz')
    ref = HipEngine(0)
    try:
        ref.load_state({k.replace('|', '/'): z[k] for k in z.files})
        ref.finalize()
        tok = np.random.default_rng(0).integers(1, 148, (2, 40)).astype(np.int32)
        tok[1, 25:] = 0
        masks = (np.random.default_rng(1).random((2, 30, 2, 256)) >= 0.5).astype(np.float32) * 2.0
        want = ref.tacotron2_infer(tok, max_len=30, early_stopping=False, prenet_masks=masks)
        got = model.compiled_infer.engine.tacotron2_infer(tok, max_len=30, early_stopping=False, prenet_masks=masks)
        np.testing.assert_array_equal(got.mel, want.mel)
        np.testing.assert_array_equal(got.attention_weights, want.attention_weights)
    finally:
        ref.close()
