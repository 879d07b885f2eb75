"""GPU parity: HIP mel-STFT vs the reference's golden fixture and the numpy oracle."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), 'golden', 'stft_tacotron_fixture.npz')


def test_mel_stft_reference_fixture(gpu_engine):
    f = np.load(GOLD)
    mel = gpu_engine.mel_stft(f['audio'])[0]
    err = np.abs(mel[:f['mel'].shape[0]] - f['mel']).max()
    print('max err vs reference fixture', err)
    assert err <= float(f['tolerance'])          # the reference's own tolerance (2e-3)


@pytest.mark.parametrize('B,N', [(1, 1024), (2, 5000), (3, 22050)])
def test_mel_stft_matches_oracle(gpu_engine, B, N):
    from oracle import mel_stft_ref
    from text_to_speech_amd.config import MelSTFTConfig
    audio = np.random.default_rng(N).uniform(-1, 1, (B, N)).astype(np.float32)
    ref = mel_stft_ref.mel_spectrogram(audio, MelSTFTConfig())
    out = gpu_engine.mel_stft(audio)
    assert out.shape == ref.shape
    assert np.abs(out - ref).max() <= 1e-3
