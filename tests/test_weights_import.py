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


def _keras_paths(manifest_name):
    """A plausible Keras 3 `variable.path` for a manifest tensor, built from the layer names of the reference source
    (tacotron2_arch.py:80-107,168,248,299,347,359-361,503-508; location_sensitive_attention.py:36-73; waveglow_arch.py:58-87,
    197,213,222; invertible_conv.py:32) with the wrappers Keras adds (Bidirectional -> forward_lstm / backward_lstm +
    lstm_cell, StackedRNNCells, the functional encoder model)."""
    parts = manifest_name.split('/')
    if parts[0] == 'waveglow':
        if parts[1].startswith('invertible_conv'):
            return f'wave_glow/{parts[1]}/conv/kernel'
        return 'wave_glow/' + '/'.join(parts[1:])
    var = parts[-1]
    if manifest_name == 'tacotron2/encoder/embeddings':
        return 'encoder/encoder_embeddings/embeddings'
    if parts[1] == 'encoder' and parts[2] == 'bi_lstm':
        return f'encoder/bidirectional/{parts[3]}_lstm/lstm_cell/{var}'
    if parts[1] in ('encoder', 'postnet'):
        return f'{parts[1]}/{parts[2]}/{var}'
    if parts[2] == 'prenet':
        return f'decoder/prenet/{parts[3]}/kernel'
    if parts[2] == 'lsa':
        inner = 'location_layer/' if parts[3].startswith('location') else ''
        return f'decoder/decoder_cell/location_sensitive_attention/{inner}{parts[3]}/kernel'
    if parts[2] == 'attention_rnn':
        return f'decoder/decoder_cell/attention_rnn/{var}'
    if parts[2] == 'decoder_rnn':
        return f'decoder/decoder_cell/decoder_rnn/cell_0/{var}'
    return f'decoder/{parts[2]}/{var}'


def test_keras_variable_paths_map_onto_the_manifest(taco_cfg, wg_cfg):
    """Step 2 of the Keras checkpoint import: renaming + shape check; extra variables are ignored, clashes and gaps raise."""
    import pytest
    from text_to_speech_amd import weights, weights_import
    from text_to_speech_amd.config import Tacotron2Config, WaveGlowConfig
    cfg = Tacotron2Config(speaker_embedding_dim=256)
    w = weights.synth_tacotron2(cfg, seed=2)
    named = {_keras_paths(k): v for k, v in w.items()}
    assert len(named) == len(w)
    named['decoder/decoder_cell/attention_rnn/kernel:0'] = named.pop('decoder/decoder_cell/attention_rnn/kernel')   # TF-style suffix
    named['adam/decoder_gate_output_kernel_momentum'] = np.zeros((3,), np.float32)       # not an inference variable
    named['encoder/speaker_embedding/embeddings'] = np.zeros((10, 256), np.float32)      # speaker table: not in the manifest
    back = weights_import.from_keras_variables(named, 'tacotron2', cfg)
    assert list(back) == list(w) and all(np.array_equal(back[k], w[k]) for k in w)
    small = WaveGlowConfig(n_channels=8, n_layers=2)
    ww = weights.synth_waveglow(small, seed=3)
    back = weights_import.from_keras_variables({_keras_paths(k): v for k, v in ww.items()}, 'waveglow', small)
    assert list(back) == list(ww) and all(np.array_equal(back[k], ww[k]) for k in ww)
    # a missing tensor and a wrong shape are errors, with the tensor named
    broken = dict(named)
    broken.pop('decoder/gate_output/bias')
    with pytest.raises(KeyError, match='gate_output/bias'):
        weights_import.from_keras_variables(broken, 'tacotron2', cfg)
    broken = dict(named)
    broken['decoder/gate_output/bias'] = np.zeros((2,), np.float32)
    with pytest.raises(ValueError, match='gate_output/bias'):
        weights_import.from_keras_variables(broken, 'tacotron2', cfg)
    clash = dict(named)
    clash['other_scope/decoder/gate_output/bias'] = np.zeros((1,), np.float32)
    with pytest.raises(ValueError, match='both map'):
        weights_import.from_keras_variables(clash, 'tacotron2', cfg)


def test_keras_import_cli_roundtrip(tmp_path, taco_cfg):
    from safetensors.numpy import save_file
    from text_to_speech_amd import weights, weights_import
    w = weights.synth_tacotron2(taco_cfg, seed=4)
    src = tmp_path / 'vars.safetensors'
    save_file({_keras_paths(k): v for k, v in w.items()}, str(src))
    out = tmp_path / 'model.ttsw'
    weights_import.main(['--keras-tacotron2', str(src), '-o', str(out)])
    back = weights.load_ttsw(out)
    assert list(back) == list(w) and all(np.array_equal(back[k], w[k]) for k in w)
