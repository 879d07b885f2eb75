"""CPU: NVIDIA PyTorch checkpoint layouts <-> the engine's Keras-layout manifest.

No checkpoint ships here, so NVIDIA-layout state dicts are built from the synthetic weights; the conversion is checked by
round trip AND numerically: torch modules fed with the NVIDIA-layout tensors must agree with the oracle fed with the
converted ones (layout mistakes such as a missing transpose would not survive that).
"""
import numpy as np
import torch
import torch.nn.functional as F


def test_tacotron2_roundtrip_and_numerics(taco_cfg):
    from oracle import tacotron2_ref
    from text_to_speech_amd import weights, weights_import
    w = weights.synth_tacotron2(taco_cfg, seed=21)
    sd = weights_import.to_nvidia_tacotron2(w, taco_cfg)
    assert sd['decoder.attention_rnn.weight_ih'].shape == (4096, 768)
    assert sd['postnet.convolutions.0.0.conv.weight'].shape == (512, 80, 5)
    back = weights_import.from_nvidia_tacotron2({'state_dict': {f'module.{k}': torch.from_numpy(np.asarray(v)) for k, v in sd.items()}})
    assert list(back) == list(weights.tacotron2_manifest(taco_cfg))
    for k in w:
        np.testing.assert_allclose(back[k], w[k], atol=1e-6, err_msg=k)
    # numerics: a torch LSTMCell / Conv1d / BatchNorm built from the NVIDIA tensors vs the oracle on converted tensors
    rng = np.random.default_rng(0)
    x = rng.standard_normal((3, 768)).astype(np.float32)
    h = rng.standard_normal((3, 1024)).astype(np.float32)
    c = rng.standard_normal((3, 1024)).astype(np.float32)
    t = lambda a: torch.from_numpy(np.asarray(a))
    h2, c2 = torch._VF.lstm_cell(t(x), (t(h), t(c)), t(sd['decoder.attention_rnn.weight_ih']), t(sd['decoder.attention_rnn.weight_hh']),
                                 t(sd['decoder.attention_rnn.bias_ih']), t(sd['decoder.attention_rnn.bias_hh']))
    ho, co = tacotron2_ref.lstm_cell(x, h, c, back['tacotron2/decoder/attention_rnn/kernel'],
                                     back['tacotron2/decoder/attention_rnn/recurrent_kernel'], back['tacotron2/decoder/attention_rnn/bias'])
    assert np.abs(h2.numpy() - ho).max() < 1e-5 and np.abs(c2.numpy() - co).max() < 1e-5
    xin = rng.standard_normal((2, 11, 80)).astype(np.float32)
    y = F.conv1d(t(xin).transpose(1, 2), t(sd['postnet.convolutions.0.0.conv.weight']), t(sd['postnet.convolutions.0.0.conv.bias']), padding=2)
    y = F.batch_norm(y, t(sd['postnet.convolutions.0.1.running_mean']), t(sd['postnet.convolutions.0.1.running_var']),
                     t(sd['postnet.convolutions.0.1.weight']), t(sd['postnet.convolutions.0.1.bias']), training=False, eps=1e-5)
    ref = tacotron2_ref.masked_conv_bn(xin, np.ones((2, 11), bool), back, 'tacotron2/postnet/conv_1', 'tacotron2/postnet/norm_1', 1e-5, None)
    assert np.abs(y.transpose(1, 2).numpy() - ref).max() < 1e-4


def test_waveglow_roundtrip_weightnorm_and_fused_cond():
    from oracle import waveglow_ref
    from text_to_speech_amd import weights, weights_import
    from text_to_speech_amd.config import WaveGlowConfig
    cfg = WaveGlowConfig(n_channels=32, n_layers=8)
    w = weights.synth_waveglow(cfg, seed=4)
    for fused, wn in ((False, False), (True, True)):
        sd = weights_import.to_nvidia_waveglow(w, cfg, fused_cond=fused, weight_norm=wn)
        if fused:
            assert sd['WN.0.cond_layer.bias'].shape == (8 * 64,) and 'WN.0.in_layers.0.weight_g' in sd
        back = weights_import.from_nvidia_waveglow(sd, cfg)
        for k in w:
            np.testing.assert_allclose(back[k], w[k], atol=2e-6, err_msg=k)
    # numerics: NVIDIA-style upsampling with the torch-layout tensor == oracle upsample with the converted one
    sd = weights_import.to_nvidia_waveglow(w, cfg)
    mel = np.random.default_rng(1).standard_normal((2, 5, 80)).astype(np.float32)
    y = F.conv_transpose1d(torch.from_numpy(mel).transpose(1, 2), torch.from_numpy(sd['upsample.weight']),
                           torch.from_numpy(sd['upsample.bias']), stride=256)[:, :, :-768]
    ref = waveglow_ref.upsample(mel, back['waveglow/upsample/kernel'], back['waveglow/upsample/bias'])
    assert np.abs(y.transpose(1, 2).numpy() - ref).max() < 1e-4
    # inverse 1x1: NVIDIA reverse = conv1d with W.inverse(); ours = audio @ inv(kernel[0].T).T
    Wt = torch.from_numpy(sd['convinv.3.conv.weight'])[:, :, 0]
    a = np.random.default_rng(2).standard_normal((1, 7, Wt.shape[0])).astype(np.float32)
    z = F.conv1d(torch.from_numpy(a).transpose(1, 2), torch.linalg.inv(Wt.double()).float()[..., None]).transpose(1, 2).numpy()
    ours = a @ waveglow_ref.inv1x1_reverse_matrix(back['waveglow/invertible_conv-3/conv/kernel'])
    assert np.abs(z - ours).max() < 1e-5


def test_cli_writes_loadable_ttsw(tmp_path, taco_cfg):
    from text_to_speech_amd import weights, weights_import
    w = weights.synth_tacotron2(taco_cfg, seed=2)
    ck = tmp_path / 'taco.pt'
    torch.save({'state_dict': {k: torch.from_numpy(np.asarray(v)) for k, v in weights_import.to_nvidia_tacotron2(w, taco_cfg).items()}}, ck)
    out = tmp_path / 'm.ttsw'
    weights_import.main(['--tacotron2', str(ck), '-o', str(out)])
    back = weights.load_ttsw(out)
    assert set(back) == set(w) and np.allclose(back['tacotron2/decoder/gate_output/bias'], w['tacotron2/decoder/gate_output/bias'])
