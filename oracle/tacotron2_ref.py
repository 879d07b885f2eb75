"""numpy restatement of the reference Tacotron2 inference (TEST INFRASTRUCTURE -- see oracle/__init__.py).

PARITY UNPINNED: no reference test/golden exists for this path.

Follows /root/reference/architectures/tacotron2_arch.py, architectures/layers/location_sensitive_attention.py,
architectures/layers/masked_1d.py, architectures/simple_models.py:154-293 and current_blocks.py:213-358:
  encoder                    tacotron2_arch.py:235-333   (embedding -> 3x[MaskedConv1D k5 -> BN -> relu] -> BiLSTM -> +speaker)
  prenet                     tacotron2_arch.py:188-203
  decoder cell               tacotron2_arch.py:422-486
  LSA                        location_sensitive_attention.py:96-186
  decoder loop               tacotron2_arch.py:609-749
  postnet + residual         tacotron2_arch.py:214-232, 915-917
  Tacotron2.infer            tacotron2_arch.py:866-925

Semantics fixed here where the reference is ambiguous (SURVEY.md "Parity hazards"):
  * the decoder mask is `t <= lengths[b]` per row (the reference's broadcast at :745 is only shape-correct for B == 1);
  * encoder outputs at padded token positions are returned as zeros -- the reference zeroes them before every use
    (location_sensitive_attention.py:96-102), so they are unobservable;
  * prenet dropout takes explicit masks `[B, max_len, 2, 256]` (already scaled by 1/(1-p)) or is off (deterministic).
"""
from __future__ import annotations

from collections import namedtuple

import numpy as np

Tacotron2InferenceOutput = namedtuple(
    'Tacotron2InferenceOutput', ['decoder_output', 'mel', 'stop_tokens', 'attention_weights', 'lengths'])


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def lstm_cell(x, h, c, kernel, recurrent, bias):
    """keras.layers.LSTMCell: z = x@W + h@U + b, gates i, f, c, o; c' = f*c + i*tanh(g); h' = o*tanh(c')."""
    z = x @ kernel + h @ recurrent + bias
    u = h.shape[-1]
    i, f, g, o = z[:, :u], z[:, u:2 * u], z[:, 2 * u:3 * u], z[:, 3 * u:]
    c2 = _sigmoid(f) * c + _sigmoid(i) * np.tanh(g)
    h2 = _sigmoid(o) * np.tanh(c2)
    return h2, c2


def conv1d_same(x, kernel, bias):
    """Keras Conv1D, stride 1, padding 'same', odd kernel (cross-correlation, kernel [k, in, out])."""
    B, T, C = x.shape
    k = kernel.shape[0]
    p = k // 2
    xp = np.zeros((B, T + 2 * p, C), dtype=x.dtype)
    xp[:, p:p + T] = x
    out = np.zeros((B, T, kernel.shape[2]), dtype=x.dtype)
    for j in range(k):
        out += xp[:, j:j + T] @ kernel[j]
    if bias is not None:
        out = out + bias
    return out


def batch_norm(x, w, prefix, eps):
    g, b = w[f'{prefix}/gamma'], w[f'{prefix}/beta']
    m, v = w[f'{prefix}/moving_mean'], w[f'{prefix}/moving_variance']
    return (x - m) / np.sqrt(v + x.dtype.type(eps)) * g + b


def masked_conv_bn(x, mask, w, conv, norm, eps, activation):
    """MaskedConv1D -> BatchNormalization -> activation.  masked_1d.py:101-116, current_blocks.py:314-343."""
    mf = mask[:, :, None].astype(x.dtype)
    out = conv1d_same(x * mf, w[f'{conv}/kernel'], w[f'{conv}/bias']) * mf
    out = batch_norm(out, w, norm, eps)
    if activation == 'relu':
        out = np.maximum(out, 0)
    elif activation == 'tanh':
        out = np.tanh(out)
    return out


def encoder(tokens, w, cfg, speaker_embedding=None):
    """Tacotron2Encoder.  tokens int32 [B, Tin] (0 = pad) -> (memory [B, Tin, enc], mask [B, Tin])."""
    p = 'tacotron2/encoder'
    mask = tokens != cfg.pad_token
    x = w[f'{p}/embeddings'][tokens]
    for i in range(cfg.encoder_n_conv):
        x = masked_conv_bn(x, mask, w, f'{p}/conv_{i + 1}', f'{p}/norm_{i + 1}', cfg.bn_epsilon, 'relu')
    B, Tin, _ = x.shape
    u = cfg.embedding_dim // 2
    outs = []
    for direction in ('forward', 'backward'):
        K_, U_, b_ = (w[f'{p}/bi_lstm/{direction}/{n}'] for n in ('kernel', 'recurrent_kernel', 'bias'))
        h = np.zeros((B, u), dtype=x.dtype)
        c = np.zeros((B, u), dtype=x.dtype)
        out = np.zeros((B, Tin, u), dtype=x.dtype)
        order = range(Tin) if direction == 'forward' else range(Tin - 1, -1, -1)
        for t in order:
            h2, c2 = lstm_cell(x[:, t], h, c, K_, U_, b_)
            m = mask[:, t][:, None]
            h = np.where(m, h2, h)          # masked steps carry the state through
            c = np.where(m, c2, c)
            out[:, t] = np.where(m, h2, 0)  # padded positions: zero (unobservable, see module docstring)
        outs.append(out)
    memory = np.concatenate(outs, axis=-1)
    if cfg.speaker_embedding_dim:
        # ConcatEmbedding 'concat' at 'end', masked.  concat_embedding.py:36-61, tacotron2_arch.py:326
        spk = np.broadcast_to(speaker_embedding[:, None, :].astype(x.dtype), (B, Tin, cfg.speaker_embedding_dim))
        memory = np.concatenate([memory, spk], axis=-1) * mask[:, :, None]
    return memory, mask


def prenet(frame, w, drop=None):
    """Tacotron2Prenet.call.  frame [B, 80]; drop None or [B, 2, 256] multiplicative masks.  tacotron2_arch.py:188-203."""
    x = frame
    for i in range(2):
        x = np.maximum(x @ w[f'tacotron2/decoder/prenet/layer_{i}/kernel'], 0)
        if drop is not None:
            x = x * drop[:, i]
    return x


def location_conv(cat, kernel):
    """Conv1D(2 -> 32, k=31, 'same', no bias) on [B, Tin, 2].  location_sensitive_attention.py:27-41."""
    return conv1d_same(cat, kernel, None)


def attention(query, memory, processed_memory, prev_w, cum_w, mask, w):
    """LocationSensitiveAttention.call.  location_sensitive_attention.py:104-186."""
    p = 'tacotron2/decoder/lsa'
    pq = (query @ w[f'{p}/query_layer/kernel'])[:, None, :]
    cat = np.stack([prev_w, cum_w], axis=-1)
    loc = location_conv(cat, w[f'{p}/location_conv/kernel']) @ w[f'{p}/location_dense/kernel']
    e = (np.tanh(pq + processed_memory + loc) @ w[f'{p}/value_layer/kernel'])[..., 0]
    e = np.where(mask, e, -np.inf)
    e = e - e.max(axis=-1, keepdims=True)
    ex = np.exp(e)
    aw = ex / ex.sum(axis=-1, keepdims=True)
    ctx = (aw[:, None, :] @ memory)[:, 0]
    return ctx, aw, cum_w + aw


def decode(memory, mask, w, cfg, max_length, early_stopping=True, prenet_masks=None,
           attn_mask_win_len=None, attn_mask_offset=None, trace=None):
    """Tacotron2Decoder.infer.  tacotron2_arch.py:609-749.  `trace` (a dict, tests only) receives 'cell_out'
    [B, steps, 1024 + enc]: the projection / gate input of every step (tests script stop tokens from it)."""
    dt = memory.dtype
    B, Tin, enc = memory.shape
    d = 'tacotron2/decoder'
    # process_memory: zero masked rows, then memory_layer.  location_sensitive_attention.py:96-102
    memory = np.where(mask[:, :, None], memory, 0).astype(dt)
    pm = memory @ w[f'{d}/lsa/memory_layer/kernel']
    encoder_length = mask.sum(axis=1)
    A, D = cfg.attention_rnn_dim, cfg.decoder_rnn_dim
    h_att = np.zeros((B, A), dt); c_att = np.zeros((B, A), dt)
    h_dec = np.zeros((B, D), dt); c_dec = np.zeros((B, D), dt)
    ctx = np.zeros((B, enc), dt)
    prev_w = np.zeros((B, Tin), dt); cum_w = np.zeros((B, Tin), dt)
    frame = np.zeros((B, cfg.n_mel_channels), dt)
    outputs = np.zeros((B, max_length, cfg.n_mel_channels), dt)
    stop_tokens = np.zeros((B, max_length), dt)
    attn = np.zeros((B, max_length, Tin), dt)
    lengths = np.zeros((B,), np.int32)
    finished = np.zeros((B,), bool)
    main_attention = np.zeros((B,), np.int64)
    ar = np.arange(Tin)[None]
    t = 0
    while t < max_length and not (early_stopping and finished.all()):
        if attn_mask_win_len is not None:
            center = np.maximum(main_attention, attn_mask_offset)
            center = np.minimum(center, encoder_length - attn_mask_win_len + attn_mask_offset)[:, None]
            amask = (center - attn_mask_offset <= ar) & (ar <= center - attn_mask_offset + attn_mask_win_len) & mask
        else:
            amask = mask
        p_out = prenet(frame, w, None if prenet_masks is None else prenet_masks[:, t])
        # attention rnn -- tacotron2_arch.py:452-455
        h_att, c_att = lstm_cell(np.concatenate([p_out, ctx], -1), h_att, c_att,
                                 w[f'{d}/attention_rnn/kernel'], w[f'{d}/attention_rnn/recurrent_kernel'],
                                 w[f'{d}/attention_rnn/bias'])
        ctx, prev_w, cum_w = attention(h_att, memory, pm, prev_w, cum_w, amask, w)
        # decoder rnn -- :469-476
        h_dec, c_dec = lstm_cell(np.concatenate([h_att, ctx], -1), h_dec, c_dec,
                                 w[f'{d}/decoder_rnn/cell_0/kernel'], w[f'{d}/decoder_rnn/cell_0/recurrent_kernel'],
                                 w[f'{d}/decoder_rnn/cell_0/bias'])
        cell_out = np.concatenate([h_dec, ctx], -1)
        if trace is not None:
            trace.setdefault('cell_out', []).append(cell_out.copy())
        frame = cell_out @ w[f'{d}/linear_projection/kernel'] + w[f'{d}/linear_projection/bias']
        stop = _sigmoid(cell_out @ w[f'{d}/gate_output/kernel'] + w[f'{d}/gate_output/bias'])[:, 0]
        finished = finished | (stop > 0.5)                      # :664
        lengths = lengths + (~finished).astype(np.int32)        # :665
        outputs[:, t] = frame
        stop_tokens[:, t] = stop
        attn[:, t] = prev_w
        main_attention = prev_w.argmax(axis=1)
        t += 1
    dec_mask = np.arange(max_length)[None] <= lengths[:, None]  # :745, per-row (see docstring)
    if trace is not None and 'cell_out' in trace:
        trace['cell_out'] = np.stack(trace['cell_out'], axis=1)
    return outputs, stop_tokens, dec_mask, attn, lengths, t


def postnet(x, dec_mask, w, cfg):
    """Tacotron2Postnet (simple_cnn of MaskedConv1D+BN, tanh on all but the last).  tacotron2_arch.py:214-232."""
    n = cfg.postnet_n_conv
    for i in range(n):
        x = masked_conv_bn(x, dec_mask, w, f'tacotron2/postnet/conv_{i + 1}', f'tacotron2/postnet/norm_{i + 1}',
                           cfg.bn_epsilon, 'tanh' if i < n - 1 else None)
    return x


def resolve_max_length(mask, max_length):
    """tacotron2_arch.py:886-892: float f -> int(max non-pad length * f)."""
    if isinstance(max_length, float):
        return int(np.float32(mask.sum(axis=1).max()) * np.float32(max_length))
    return int(max_length)


def infer(tokens, w, cfg, speaker_embedding=None, max_length=10.0, early_stopping=True, prenet_masks=None,
          attn_mask_win_len=None, attn_mask_offset=0.5, dtype=np.float32, trace=None):
    """Tacotron2.infer.  tacotron2_arch.py:866-925."""
    w = {k: v.astype(dtype) for k, v in w.items() if k.startswith('tacotron2/')}
    tokens = np.asarray(tokens, dtype=np.int32)
    memory, mask = encoder(tokens, w, cfg, speaker_embedding)
    max_length = resolve_max_length(mask, max_length)
    if attn_mask_win_len is not None and isinstance(attn_mask_offset, float):
        attn_mask_offset = int(np.float32(attn_mask_win_len) * np.float32(attn_mask_offset))
    if prenet_masks is not None:
        prenet_masks = np.asarray(prenet_masks, dtype=dtype)
    dec_out, stop_tokens, dec_mask, attn, lengths, _ = decode(
        memory, mask, w, cfg, max_length, early_stopping, prenet_masks, attn_mask_win_len, attn_mask_offset, trace)
    post = postnet(dec_out, dec_mask, w, cfg)
    mel = dec_out + post
    return Tacotron2InferenceOutput(decoder_output=dec_out, mel=mel, stop_tokens=stop_tokens,
                                    attention_weights=attn, lengths=lengths)
