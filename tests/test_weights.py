"""CPU: weight manifests, synthetic generator determinism and the TTSW file."""
import numpy as np


def test_param_counts(wg_cfg, taco_cfg):
    from text_to_speech_amd import weights
    assert weights.n_params(weights.waveglow_manifest(wg_cfg)) == 267_999_848
    assert weights.n_params(weights.tacotron2_manifest(taco_cfg)) == 28_190_241
    assert wg_cfg.flow_channels() == [(8, 4)] * 4 + [(6, 3)] * 4 + [(4, 2)] * 4
    assert wg_cfg.n_remaining_channels == 4


def test_synth_is_deterministic_and_ttsw_roundtrip(tmp_path, taco_cfg):
    from text_to_speech_amd import weights
    a = weights.synth_tacotron2(taco_cfg, seed=7)
    b = weights.synth_tacotron2(taco_cfg, seed=7)
    assert all(np.array_equal(a[k], b[k]) for k in a)
    u = a['tacotron2/decoder/attention_rnn/bias']
    assert abs(u[1024:2048].mean() - 1.0) < 0.02                 # forget-gate bias +1
    p = tmp_path / 'w.ttsw'
    weights.save_ttsw(p, a)
    c = weights.load_ttsw(p)
    assert list(c) == list(a) and all(np.array_equal(a[k], c[k]) and a[k].shape == c[k].shape for k in a)


def test_inv1x1_kernels_are_orthogonal(wg_cfg):
    from text_to_speech_amd import weights
    from text_to_speech_amd.config import WaveGlowConfig
    w = weights.synth_waveglow(WaveGlowConfig(n_channels=8, n_layers=1), seed=1)
    for k in range(12):
        q = w[f'waveglow/invertible_conv-{k}/conv/kernel'][0]
        np.testing.assert_allclose(q @ q.T, np.eye(q.shape[0]), atol=1e-5)


def test_safetensors_roundtrip(tmp_path, taco_cfg):
    from text_to_speech_amd import weights
    a = weights.synth_tacotron2(taco_cfg, seed=3)
    p = tmp_path / 'w.safetensors'
    weights.save_safetensors(p, a)
    b = weights.load_safetensors(p)
    assert set(a) == set(b) and all(np.array_equal(a[k], b[k]) and b[k].dtype == np.float32 for k in a)
