"""Second, independent CPU restatement of the hot path on torch.nn.functional (channels-first, the NVIDIA formulation).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): it shares no code with the numpy oracle and exists (i) to cross-check
that oracle (tests/test_oracle_crosscheck.py: < 1e-4 on WaveGlow audio and Tacotron2 mels) and (ii) as the `torch` CPU leg
of bench.py's `cpu_baseline` -- oneDNN / MKL convolutions and GEMMs are the kernel class Keras-on-TF would dispatch to on
the host.  PARITY UNPINNED like the numpy oracle.

Follows the same reference lines as the numpy oracle: architectures/waveglow_arch.py:244-306, :105-141,
architectures/layers/invertible_conv.py:41-51; architectures/tacotron2_arch.py:188-203, :235-333, :422-486, :609-749,
:866-925 and architectures/layers/location_sensitive_attention.py:96-186.
"""
import numpy as np
import torch
import torch.nn.functional as F


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def torch_waveglow(mel, w, cfg, z, sigma=1.0, collect=None):
    """`collect` (a dict): the gated activations of every WN layer, keyed (flow, layer), as [B, L, C] arrays -- the
    intermediates the layer-level parity tests compare (the numpy oracle's `wn_block(collect=...)`)."""
    g = cfg.n_group
    spect = F.conv_transpose1d(_t(mel).transpose(1, 2), _t(w['waveglow/upsample/kernel']).permute(2, 1, 0),
                               _t(w['waveglow/upsample/bias']), stride=cfg.upsample_stride)
    spect = spect[:, :, :-(cfg.upsample_kernel - cfg.upsample_stride)]
    spect = spect.unfold(2, g, g).permute(0, 2, 1, 3)
    spect = spect.contiguous().view(spect.size(0), spect.size(1), -1).permute(0, 2, 1)      # [B, 640, L]
    zt = _t(z).transpose(1, 2)                                                              # [B, 8, L]
    n_rem = cfg.n_remaining_channels
    audio = sigma * zt[:, :n_rem]
    zt = zt[:, n_rem:]
    C = cfg.n_channels
    for k in reversed(range(cfg.n_flows)):
        n_half = audio.size(1) // 2
        a0, a1 = audio[:, :n_half], audio[:, n_half:]
        p = f'waveglow/block-{k}'
        x = F.conv1d(a0, _t(w[f'{p}/start_conv/kernel']).permute(2, 1, 0), _t(w[f'{p}/start_conv/bias']))
        output = torch.zeros_like(x)
        for i in range(cfg.n_layers):
            d = 2 ** i
            acts_in = F.conv1d(x, _t(w[f'{p}/in_conv-{i}/kernel']).permute(2, 1, 0), _t(w[f'{p}/in_conv-{i}/bias']),
                               dilation=d, padding=d)
            cond = F.conv1d(spect, _t(w[f'{p}/cond_layer-{i}/kernel']).permute(2, 1, 0),
                            _t(w[f'{p}/cond_layer-{i}/bias']))
            s = acts_in + cond
            acts = torch.tanh(s[:, :C]) * torch.sigmoid(s[:, C:])
            if collect is not None:
                collect[(k, i)] = acts.permute(0, 2, 1).contiguous().numpy()
            rs = F.conv1d(acts, _t(w[f'{p}/res_skip_conv-{i}/kernel']).permute(2, 1, 0),
                          _t(w[f'{p}/res_skip_conv-{i}/bias']))
            if i < cfg.n_layers - 1:
                x = x + rs[:, :C]
                output = output + rs[:, C:]
            else:
                output = output + rs
        out = F.conv1d(output, _t(w[f'{p}/end_conv/kernel']).permute(2, 1, 0), _t(w[f'{p}/end_conv/bias']))
        b, s = out[:, :n_half], out[:, n_half:]
        a1 = (a1 - b) / torch.exp(s)
        audio = torch.cat([a0, a1], 1)
        W = _t(w[f'waveglow/invertible_conv-{k}/conv/kernel'])[0].t()                      # torch weight [out, in]
        audio = F.conv1d(audio, torch.linalg.inv(W.double()).float()[..., None])
        if k % cfg.n_early_every == 0 and k > 0:
            audio = torch.cat([sigma * zt[:, :cfg.n_early_size], audio], 1)
            zt = zt[:, cfg.n_early_size:]
    return audio.permute(0, 2, 1).contiguous().view(audio.size(0), -1).numpy()


def torch_tacotron2(tokens, w, cfg, speaker, max_len, prenet_masks):
    p = 'tacotron2'
    tok = torch.from_numpy(tokens.astype(np.int64))
    mask = tok != 0
    B, Tin = tok.shape
    x = F.embedding(tok, _t(w[f'{p}/encoder/embeddings'])).transpose(1, 2)                  # [B, 512, Tin]
    mf = mask[:, None, :].float()

    def conv_bn(x, mf, conv, norm, act):
        y = F.conv1d(x * mf, _t(w[f'{conv}/kernel']).permute(2, 1, 0), _t(w[f'{conv}/bias']), padding=2) * mf
        y = F.batch_norm(y, _t(w[f'{norm}/moving_mean']), _t(w[f'{norm}/moving_variance']), _t(w[f'{norm}/gamma']),
                         _t(w[f'{norm}/beta']), training=False, eps=cfg.bn_epsilon)
        return act(y) if act else y

    for i in range(3):
        x = conv_bn(x, mf, f'{p}/encoder/conv_{i + 1}', f'{p}/encoder/norm_{i + 1}', torch.relu)
    x = x.transpose(1, 2)
    outs = []
    for direction in ('forward', 'backward'):
        wi = _t(w[f'{p}/encoder/bi_lstm/{direction}/kernel']).t().contiguous()
        wh = _t(w[f'{p}/encoder/bi_lstm/{direction}/recurrent_kernel']).t().contiguous()
        bi = _t(w[f'{p}/encoder/bi_lstm/{direction}/bias'])
        h = torch.zeros(B, 256)
        c = torch.zeros(B, 256)
        out = torch.zeros(B, Tin, 256)
        for t in (range(Tin) if direction == 'forward' else reversed(range(Tin))):
            h2, c2 = torch._VF.lstm_cell(x[:, t], (h, c), wi, wh, bi, torch.zeros_like(bi))
            m = mask[:, t, None]
            h, c = torch.where(m, h2, h), torch.where(m, c2, c)
            out[:, t] = torch.where(m, h2, torch.zeros_like(h2))
        outs.append(out)
    memory = torch.cat(outs, -1)
    if speaker is not None:
        memory = torch.cat([memory, _t(speaker)[:, None].expand(B, Tin, -1)], -1)
    memory = memory * mask[..., None]
    d = f'{p}/decoder'
    pm = memory @ _t(w[f'{d}/lsa/memory_layer/kernel'])
    enc = memory.shape[-1]
    h_a = torch.zeros(B, 1024); c_a = torch.zeros(B, 1024); h_d = torch.zeros(B, 1024); c_d = torch.zeros(B, 1024)
    ctx = torch.zeros(B, enc); aw = torch.zeros(B, Tin); cum = torch.zeros(B, Tin); frame = torch.zeros(B, 80)
    lstm = {}
    for name in ('attention_rnn', 'decoder_rnn/cell_0'):
        lstm[name] = (_t(w[f'{d}/{name}/kernel']).t().contiguous(), _t(w[f'{d}/{name}/recurrent_kernel']).t().contiguous(),
                      _t(w[f'{d}/{name}/bias']))
    frames, stops, aligns = [], [], []
    for t in range(max_len):
        x1 = torch.relu(frame @ _t(w[f'{d}/prenet/layer_0/kernel']))
        if prenet_masks is not None:
            x1 = x1 * _t(prenet_masks[:, t, 0])
        x2 = torch.relu(x1 @ _t(w[f'{d}/prenet/layer_1/kernel']))
        if prenet_masks is not None:
            x2 = x2 * _t(prenet_masks[:, t, 1])
        wi, wh, bi = lstm['attention_rnn']
        h_a, c_a = torch._VF.lstm_cell(torch.cat([x2, ctx], -1), (h_a, c_a), wi, wh, bi, torch.zeros_like(bi))
        cat = torch.stack([aw, cum], 1)                                                      # [B, 2, Tin]
        loc = F.conv1d(cat, _t(w[f'{d}/lsa/location_conv/kernel']).permute(2, 1, 0), padding=15).transpose(1, 2)
        loc = loc @ _t(w[f'{d}/lsa/location_dense/kernel'])
        e = torch.tanh((h_a @ _t(w[f'{d}/lsa/query_layer/kernel']))[:, None] + pm + loc) @ _t(w[f'{d}/lsa/value_layer/kernel'])
        e = e[..., 0].masked_fill(~mask, float('-inf'))
        aw = torch.softmax(e, -1)
        cum = cum + aw
        ctx = torch.bmm(aw[:, None], memory)[:, 0]
        wi, wh, bi = lstm['decoder_rnn/cell_0']
        h_d, c_d = torch._VF.lstm_cell(torch.cat([h_a, ctx], -1), (h_d, c_d), wi, wh, bi, torch.zeros_like(bi))
        co = torch.cat([h_d, ctx], -1)
        frame = co @ _t(w[f'{d}/linear_projection/kernel']) + _t(w[f'{d}/linear_projection/bias'])
        stops.append(torch.sigmoid(co @ _t(w[f'{d}/gate_output/kernel']) + _t(w[f'{d}/gate_output/bias']))[:, 0])
        frames.append(frame)
        aligns.append(aw)
    dec = torch.stack(frames, 1)
    stop = torch.stack(stops, 1)
    fired = (stop > 0.5).int()
    lengths = torch.where(fired.any(1), fired.argmax(1), torch.full((B,), max_len))
    dmask = (torch.arange(max_len)[None] <= lengths[:, None])[:, None, :].float()
    y = dec.transpose(1, 2)
    for i in range(5):
        y = conv_bn(y, dmask, f'{p}/postnet/conv_{i + 1}', f'{p}/postnet/norm_{i + 1}', torch.tanh if i < 4 else None)
    mel = dec + y.transpose(1, 2)
    return dec.numpy(), mel.numpy(), stop.numpy(), torch.stack(aligns, 1).numpy(), lengths.numpy()
