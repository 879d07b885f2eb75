"""CPU: pins the oracle on the reference's own golden vector and on algebraic known answers."""
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(__file__), 'golden', 'stft_tacotron_fixture.npz')


def test_mel_stft_oracle_matches_reference_fixture():
    """tests/__reproduction/stft-TacotronSTFT.npy (reference test_utils_audio.py:85-112, max_err 2e-3)."""
    from oracle import mel_stft_ref
    from text_to_speech_amd.config import MelSTFTConfig
    f = np.load(GOLD)
    assert f['audio'].shape == (32768,) and f['mel'].shape == (120, 80)
    mel = mel_stft_ref.mel_spectrogram(f['audio'], MelSTFTConfig())[0]
    assert mel.shape == (129, 80)
    err = np.abs(mel[:120] - f['mel']).max()
    assert err <= float(f['tolerance']), err
    assert err <= 1e-3          # north-star tolerance (measured 6.5e-4)


def test_mel_filterbank_properties():
    from oracle import mel_stft_ref
    fb = mel_stft_ref.mel_filterbank()
    assert fb.shape == (80, 513) and fb.dtype == np.float32
    assert (fb >= 0).all() and (fb.sum(1) > 0).all()
    assert fb[:, 372:].sum() == 0          # nothing above fmax = 8 kHz (bin 8000 / 11025 * 512 = 371.5)


def test_inv1x1_reverse_undoes_forward(wg_cfg):
    """Invertible1x1Conv: reverse(forward(x)) == x (invertible_conv.py:41-61)."""
    from oracle import waveglow_ref
    rng = np.random.default_rng(0)
    for c in (4, 6, 8):
        q, _ = np.linalg.qr(rng.standard_normal((c, c)))
        kernel = (q + 0.1 * rng.standard_normal((c, c))).astype(np.float32)[None]
        x = rng.standard_normal((2, 7, c)).astype(np.float32)
        y = x @ kernel[0]                                   # forward Conv1D(k=1), Keras kernel [1, in, out]
        back = y @ waveglow_ref.inv1x1_reverse_matrix(kernel)
        np.testing.assert_allclose(back, x, atol=2e-5)


def test_waveglow_zero_end_conv_reduces_to_inv_chain():
    """With `end` = 0 (the reference's initialisation, waveglow_arch.py:60-64) couplings are identity."""
    from oracle import waveglow_ref
    from text_to_speech_amd import weights
    from text_to_speech_amd.config import WaveGlowConfig
    cfg = WaveGlowConfig(n_channels=16, n_layers=2)
    w = weights.synth_waveglow(cfg, seed=3)
    for k in list(w):
        if 'end_conv' in k:
            w[k] = np.zeros_like(w[k])
    rng = np.random.default_rng(1)
    mel = rng.standard_normal((1, 3, 80)).astype(np.float32)
    z = rng.standard_normal((1, 96, 8)).astype(np.float32)
    out = waveglow_ref.infer(mel, w, cfg, z=z)
    audio = z[:, :, :4]
    zz = z[:, :, 4:]
    for k in reversed(range(12)):
        audio = audio @ waveglow_ref.inv1x1_reverse_matrix(w[f'waveglow/invertible_conv-{k}/conv/kernel'])
        if k % 4 == 0 and k > 0:
            audio = np.concatenate([zz[:, :, :2], audio], axis=2)
            zz = zz[:, :, 2:]
    np.testing.assert_allclose(out, audio.reshape(1, -1), atol=1e-5)


def test_decoder_lengths_rule_on_scripted_stop(taco_cfg):
    """lengths excludes the frame on which stop > 0.5 fires; finished rows keep producing frames (:625-627,:664-665)."""
    from oracle import tacotron2_ref
    from text_to_speech_amd import weights
    w = weights.synth_tacotron2(taco_cfg, seed=1234, gate_bias=-6.55)
    w['tacotron2/decoder/gate_output/kernel'] = w['tacotron2/decoder/gate_output/kernel'] * 10
    rng = np.random.default_rng(2)
    tok = rng.integers(1, 148, (4, 30)).astype(np.int32)
    for b, n in enumerate([30, 25, 18, 12]):
        tok[b, n:] = 0
    o = tacotron2_ref.infer(tok, w, taco_cfg, max_length=100, early_stopping=True)
    assert o.lengths.tolist() == [7, 3, 5, 37]
    for b, n in enumerate(o.lengths):
        assert (o.stop_tokens[b, :n] <= 0.5).all() and o.stop_tokens[b, n] > 0.5
    steps = int(o.lengths.max()) + 1
    assert np.abs(o.decoder_output[1, 3:steps]).max() > 0          # row 1 finished at t=3 but kept decoding
    assert np.all(o.decoder_output[:, steps:] == 0)                # loop ended once all rows had fired
    np.testing.assert_allclose(o.attention_weights[:, :steps].sum(-1), 1.0, atol=1e-5)
    assert np.all(o.attention_weights[1, :, 25:] == 0)             # masked tokens: exactly zero weight


def test_waveglow_infer_inverts_the_published_forward_flow():
    """`infer` must undo the generative flow of the WaveGlow paper step by step: channel order of the early outputs, which
    half the coupling transforms, the direction of the 1x1 inverse, (a1 - b) / exp(s).  The forward direction is written from
    the publication (oracle/waveglow_ref.forward_flow); float64 keeps the round trip tight."""
    from oracle import waveglow_ref
    from text_to_speech_amd import weights
    from text_to_speech_amd.config import WaveGlowConfig
    cfg = WaveGlowConfig(n_channels=32, n_layers=3)
    w = weights.synth_waveglow(cfg, seed=5, end_scale=0.3)
    rng = np.random.default_rng(2)
    mel = rng.uniform(-11.5, 1.2, (2, 3, 80))
    audio = rng.uniform(-1, 1, (2, 3 * 256))
    z = waveglow_ref.forward_flow(audio, mel, w, cfg)
    assert z.shape == (2, 96, 8) and np.abs(z - audio.reshape(2, 96, 8)).max() > 0.1      # the flow does something
    back = waveglow_ref.infer(mel, w, cfg, z=z, sigma=1.0, dtype=np.float64)
    np.testing.assert_allclose(back, audio, atol=1e-9)
