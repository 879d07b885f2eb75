"""numpy restatement of the reference WaveGlow inference (TEST INFRASTRUCTURE -- see oracle/__init__.py).

PARITY UNPINNED: no reference test/golden exists for this path.

Follows /root/reference/architectures/waveglow_arch.py and architectures/layers/invertible_conv.py:
  upsample + trim            waveglow_arch.py:245-248   (Keras Conv1DTranspose, kernel [k, out, in], 'valid')
  regroup                    waveglow_arch.py:250-253
  noise / z handling         waveglow_arch.py:264-275
  flow loop                  waveglow_arch.py:277-304
  WN (WaveglowBlock.call)    waveglow_arch.py:105-141 ; gate :19-24
  Invertible1x1Conv reverse  invertible_conv.py:41-51
  final reshape              waveglow_arch.py:306
"""
from __future__ import annotations

import numpy as np


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def upsample(mel, kernel, bias, stride=256):
    """Conv1DTranspose(80 -> 80, k=1024, stride=256, 'valid') then drop the last k - stride rows.

    mel [B, T, Cin]; kernel [k, Cout, Cin] (Keras layout); returns [B, T*stride, Cout].
    waveglow_arch.py:245-248.
    """
    B, T, Cin = mel.shape
    k, Cout, _ = kernel.shape
    q = k // stride
    assert q * stride == k
    # y[b, t, kk, o] = sum_i mel[b, t, i] * kernel[kk, o, i]
    w2 = np.ascontiguousarray(kernel.transpose(2, 0, 1).reshape(Cin, k * Cout))
    y = (mel.reshape(B * T, Cin) @ w2).reshape(B, T, q, stride, Cout)
    out = np.zeros((B, T + q - 1, stride, Cout), dtype=mel.dtype)
    for j in range(q):                     # output position p = (t + j) * stride + r  gets tap kk = j * stride + r
        out[:, j:j + T] += y[:, :, j]
    out = out.reshape(B, (T + q - 1) * stride, Cout) + bias
    return out[:, :T * stride]             # (T-1)*s + k - (k - s) = T*s rows survive the trim


def regroup(spect, n_group=8):
    """[B, L*g, C] -> [B, L, C*g] with channel index c*g + j.  waveglow_arch.py:250-253."""
    B, N, C = spect.shape
    L = N // n_group
    return np.ascontiguousarray(spect[:, :L * n_group].reshape(B, L, n_group, C).transpose(0, 1, 3, 2)).reshape(
        B, L, C * n_group)


def conv1d_dilated_same(x, kernel, bias, dilation):
    """Keras Conv1D 'valid' on x zero-padded by `dilation` each side (kernel_size 3).  waveglow_arch.py:113-118."""
    B, L, C = x.shape
    k = kernel.shape[0]
    pad = (k * dilation - dilation) // 2
    xp = np.zeros((B, L + 2 * pad, C), dtype=x.dtype)
    xp[:, pad:pad + L] = x
    out = np.zeros((B, L, kernel.shape[2]), dtype=x.dtype) + bias
    for j in range(k):
        out += xp[:, j * dilation:j * dilation + L] @ kernel[j]
    return out


def wn_block(a0, spect, w, prefix, n_layers=8, n_channels=512, collect=None, stop_after=None):
    """WaveglowBlock.call (non-fused variant).  waveglow_arch.py:105-141.

    `collect` (a list): the gated activations `acts` of every layer (waveglow_arch.py:19-24) are appended to it -- what the
    layer-level parity tests compare, before the res/skip and `end` convolutions attenuate an error; `stop_after`: return
    None after that layer (the tests only need the first layers of one flow)."""
    x = a0 @ w[f'{prefix}/start_conv/kernel'][0] + w[f'{prefix}/start_conv/bias']
    output = None
    for i in range(n_layers):
        d = 2 ** i
        in_act = conv1d_dilated_same(x, w[f'{prefix}/in_conv-{i}/kernel'], w[f'{prefix}/in_conv-{i}/bias'], d)
        cond = spect @ w[f'{prefix}/cond_layer-{i}/kernel'][0] + w[f'{prefix}/cond_layer-{i}/bias']
        s = in_act + cond
        acts = np.tanh(s[..., :n_channels]) * _sigmoid(s[..., n_channels:])
        if collect is not None:
            collect.append(acts)
        if stop_after is not None and i >= stop_after:
            return None
        rs = acts @ w[f'{prefix}/res_skip_conv-{i}/kernel'][0] + w[f'{prefix}/res_skip_conv-{i}/bias']
        if i < n_layers - 1:
            x = rs[..., :n_channels] + x
            skip = rs[..., n_channels:]
        else:
            skip = rs
        output = skip if output is None else skip + output
    return output @ w[f'{prefix}/end_conv/kernel'][0] + w[f'{prefix}/end_conv/bias']


def inv1x1_reverse_matrix(kernel):
    """`W_inverse` of Invertible1x1Conv.build_inverse as a [c_in, c_out] matrix for `audio @ M`.

    invertible_conv.py:41-47: W = kernel[0].T ; W_inverse = inv(W).T ; reverse conv kernel = W_inverse[None]
    (Keras conv kernel layout [1, in, out])  =>  out = audio @ inv(kernel[0].T).T
    """
    W = kernel[0].T
    return np.linalg.inv(W.astype(np.float64)).T.astype(kernel.dtype)


def infer(mel, w, cfg, z=None, sigma=1.0, dtype=np.float32, return_intermediates=False):
    """WaveGlow.infer.  mel [B, T, 80] -> audio [B, T*256].  waveglow_arch.py:244-306.

    `z` is None (=> zeros, the reference's `deterministic=True`) or [B, L, n_group] noise, consumed in the reference's
    order: first n_remaining_channels, then n_early_size per early output.
    """
    w = {k: v.astype(dtype) for k, v in w.items() if k.startswith('waveglow/')}
    mel = np.asarray(mel, dtype=dtype)
    spect = upsample(mel, w['waveglow/upsample/kernel'], w['waveglow/upsample/bias'], cfg.upsample_stride)
    spect = regroup(spect, cfg.n_group)
    B, L, _ = spect.shape
    n_rem = cfg.n_remaining_channels
    if z is None:
        z = np.zeros((B, L, cfg.n_group), dtype=dtype)
    z = np.asarray(z, dtype=dtype)
    audio = dtype(sigma) * z[:, :, :n_rem]
    z = z[:, :, n_rem:]
    inter = {'spect': spect}
    for k in reversed(range(cfg.n_flows)):
        n_half = audio.shape[2] // 2
        a0, a1 = audio[:, :, :n_half], audio[:, :, n_half:]
        out = wn_block(a0, spect, w, f'waveglow/block-{k}', cfg.n_layers, cfg.n_channels)
        s = out[:, :, n_half:]
        b = out[:, :, :n_half]
        a1 = (a1 - b) / np.exp(s)
        audio = np.concatenate([a0, a1], axis=2)
        audio = audio @ inv1x1_reverse_matrix(w[f'waveglow/invertible_conv-{k}/conv/kernel'])
        if k % cfg.n_early_every == 0 and k > 0:
            zi = z[:, :, :cfg.n_early_size]
            z = z[:, :, cfg.n_early_size:]
            audio = np.concatenate([dtype(sigma) * zi, audio], axis=2)
        if return_intermediates:
            inter[f'audio_after_flow_{k}'] = audio.copy()
    res = audio.reshape(B, -1)
    return (res, inter) if return_intermediates else res


def forward_flow(audio, mel, w, cfg, dtype=np.float64):
    """The generative flow in its FORWARD (audio -> z) direction, which the reference does not contain (its `call` is `infer`,
    waveglow_arch.py:241-242): written from the published model (Prenger et al. 2018, section 2; each flow step = invertible
    1x1 convolution, then the affine coupling a1 <- exp(s) a1 + b with (b, s) = WN(a0, spect); n_early_size channels leave
    every n_early_every flows).  Used only to test that `infer` really is its inverse.  audio [B, T*256] -> z in the layout
    `infer` consumes: [final n_remaining | early outputs, latest first]."""
    w = {k: v.astype(dtype) for k, v in w.items() if k.startswith('waveglow/')}
    spect = regroup(upsample(np.asarray(mel, dtype=dtype), w['waveglow/upsample/kernel'], w['waveglow/upsample/bias'],
                             cfg.upsample_stride), cfg.n_group)
    B, L, _ = spect.shape
    a = np.asarray(audio, dtype=dtype).reshape(B, L, cfg.n_group)
    early = []
    for k in range(cfg.n_flows):
        if k % cfg.n_early_every == 0 and k > 0:
            early.append(a[:, :, :cfg.n_early_size])
            a = a[:, :, cfg.n_early_size:]
        a = a @ w[f'waveglow/invertible_conv-{k}/conv/kernel'][0]            # Conv1D(k = 1), Keras kernel [1, in, out]
        n_half = a.shape[2] // 2
        a0, a1 = a[:, :, :n_half], a[:, :, n_half:]
        out = wn_block(a0, spect, w, f'waveglow/block-{k}', cfg.n_layers, cfg.n_channels)
        a1 = np.exp(out[:, :, n_half:]) * a1 + out[:, :, :n_half]
        a = np.concatenate([a0, a1], axis=2)
    return np.concatenate([a] + early[::-1], axis=2)
