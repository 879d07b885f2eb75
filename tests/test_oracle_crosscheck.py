"""CPU: the numpy oracle against an independent torch.nn.functional restatement (oracle/torch_ref.py: channels-first,
NVIDIA formulation).

The two share no code; agreement < 1e-4 on WaveGlow audio and < 1e-4 on Tacotron2 mels (fp32 both sides).
"""
import numpy as np
import pytest
import torch

from oracle.torch_ref import torch_tacotron2, torch_waveglow


@pytest.mark.parametrize('B,T', [(1, 5), (2, 9)])
def test_waveglow_oracle_vs_torch(B, T):
    from oracle import waveglow_ref
    from text_to_speech_amd import weights
    from text_to_speech_amd.config import WaveGlowConfig
    cfg = WaveGlowConfig(n_channels=64, n_layers=8)         # full depth / dilations, narrow channels (fast on CPU)
    w = weights.synth_waveglow(cfg, seed=5)
    rng = np.random.default_rng(B)
    mel = rng.uniform(-11.5, 1.2, (B, T, 80)).astype(np.float32)
    z = rng.standard_normal((B, T * 32, 8)).astype(np.float32)
    ref = waveglow_ref.infer(mel, w, cfg, z=z, sigma=0.8)
    out = torch_waveglow(mel, w, cfg, z, sigma=0.8)
    assert ref.shape == out.shape == (B, T * 256)
    assert np.abs(ref - out).max() < 1e-4


def test_wn_layer_activations_oracle_vs_torch():
    """The layer-level GPU parity test (tests/test_waveglow_gpu.py) compares one WN layer's gated activations with the numpy
    oracle's `wn_block(collect=...)` intermediates: those intermediates against the independent torch restatement, for every
    layer of the first flow that runs (flow 11), on a model with full depth / dilations and 64 channels."""
    from oracle import waveglow_ref
    from text_to_speech_amd import weights
    from text_to_speech_amd.config import WaveGlowConfig
    cfg = WaveGlowConfig(n_channels=64, n_layers=8)
    w = weights.synth_waveglow(cfg, seed=5)
    rng = np.random.default_rng(3)
    mel = rng.uniform(-11.5, 1.2, (2, 9, 80)).astype(np.float32)
    z = rng.standard_normal((2, 9 * 32, 8)).astype(np.float32)
    got = {}
    torch_waveglow(mel, w, cfg, z, collect=got)
    wf = {k: v for k, v in w.items() if k.startswith('waveglow/')}
    spect = waveglow_ref.regroup(waveglow_ref.upsample(mel, wf['waveglow/upsample/kernel'], wf['waveglow/upsample/bias'],
                                                       cfg.upsample_stride), cfg.n_group)
    acts = []
    waveglow_ref.wn_block(z[:, :, :cfg.n_remaining_channels // 2], spect, wf, 'waveglow/block-11', cfg.n_layers, cfg.n_channels,
                          collect=acts, stop_after=6)
    assert len(acts) == 7                                   # `stop_after=6`: layers 0 .. 6
    for i, a in enumerate(acts):
        assert a.shape == got[(11, i)].shape == (2, 9 * 32, 64)
        assert np.abs(a - got[(11, i)]).max() < 1e-5, i


@pytest.mark.parametrize('spk_dim', [0, 256])
def test_tacotron2_oracle_vs_torch(spk_dim):
    from oracle import tacotron2_ref
    from text_to_speech_amd import weights
    from text_to_speech_amd.config import Tacotron2Config
    cfg = Tacotron2Config(speaker_embedding_dim=spk_dim)
    w = weights.synth_tacotron2(cfg, seed=11)
    rng = np.random.default_rng(spk_dim + 1)
    tok = rng.integers(1, 148, (2, 19)).astype(np.int32)
    tok[1, 12:] = 0
    spk = rng.standard_normal((2, spk_dim)).astype(np.float32) if spk_dim else None
    masks = (rng.random((2, 14, 2, 256)) >= 0.5).astype(np.float32) * 2
    ref = tacotron2_ref.infer(tok, w, cfg, speaker_embedding=spk, max_length=14, early_stopping=False,
                              prenet_masks=masks)
    with torch.no_grad():
        dec, mel, stop, attn, lengths = torch_tacotron2(tok, w, cfg, spk, 14, masks)
    assert np.abs(ref.decoder_output - dec).max() < 1e-4
    assert np.abs(ref.mel - mel).max() < 1e-4
    assert np.abs(ref.stop_tokens - stop).max() < 1e-5
    assert np.abs(ref.attention_weights - attn).max() < 1e-5
    assert np.array_equal(ref.lengths, lengths)
