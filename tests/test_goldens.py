"""Committed golden fixtures (tests/golden/oracle_*_small.npz, made by tests/make_oracle_goldens.py).

CPU: the numpy oracle still reproduces them on freshly synthesized weights (pins the oracle and the seeded weight
generator against drift).  GPU: the HIP path, through the C ABI, matches the committed outputs within the north-star
tolerances (waveform RMS 1e-4, mel 1e-3 abs; integer outputs exact).
"""
import hashlib
import os

import numpy as np
import pytest

from conftest import rms

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _digest(w):
    h = hashlib.sha256()
    for k in sorted(w):
        h.update(k.encode())
        h.update(np.ascontiguousarray(w[k]).tobytes())
    return h.hexdigest()


def test_oracle_waveglow_reproduces_golden(wg_weights, wg_cfg):
    from oracle import waveglow_ref
    g = np.load(os.path.join(GOLD, 'oracle_waveglow_small.npz'))
    assert _digest(wg_weights) == str(g['weights_sha256'])          # same seeded weights as when the fixture was made
    out = waveglow_ref.infer(g['mel'], wg_weights, wg_cfg, z=g['z'], sigma=float(g['sigma']))
    assert out.shape == g['audio'].shape and rms(out - g['audio']) <= 1e-6


def test_oracle_tacotron2_reproduces_golden(taco_weights, taco_cfg):
    from oracle import tacotron2_ref
    g = np.load(os.path.join(GOLD, 'oracle_tacotron2_small.npz'))
    assert _digest(taco_weights) == str(g['weights_sha256'])
    ref = tacotron2_ref.infer(g['tokens'], taco_weights, taco_cfg, max_length=14, early_stopping=False,
                              prenet_masks=g['prenet_masks'])
    assert np.array_equal(ref.lengths, g['lengths'])
    for name in ('mel', 'decoder_output', 'stop_tokens', 'attention_weights'):
        assert np.abs(getattr(ref, name) - g[name]).max() <= 1e-5, name


@pytest.mark.gpu
def test_hip_waveglow_matches_golden(gpu_engine):
    g = np.load(os.path.join(GOLD, 'oracle_waveglow_small.npz'))
    out = gpu_engine.waveglow_infer(g['mel'], z=g['z'], sigma=float(g['sigma']))
    assert out.shape == g['audio'].shape and rms(out - g['audio']) <= 1e-4
    out16 = gpu_engine.waveglow_infer(g['mel'], z=g['z'], sigma=float(g['sigma']), precision='f16')
    assert rms(out16 - g['audio']) <= 1e-3


@pytest.mark.gpu
def test_hip_tacotron2_matches_golden(gpu_engine):
    g = np.load(os.path.join(GOLD, 'oracle_tacotron2_small.npz'))
    out = gpu_engine.tacotron2_infer(g['tokens'], max_len=14, early_stopping=False, prenet_masks=g['prenet_masks'])
    assert np.array_equal(out.lengths, g['lengths'])
    for name in ('mel', 'decoder_output', 'stop_tokens', 'attention_weights'):
        assert np.abs(getattr(out, name) - g[name]).max() <= 1e-3, name
