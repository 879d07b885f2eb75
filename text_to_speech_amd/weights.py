"""Weight manifest, seeded synthetic weights and the flat `TTSW` weight file.

Tensor layouts are the Keras 3 layouts the reference's layers own ("Parity hazards" (5)):
  Dense                kernel [in, out]           bias [out]
  Conv1D               kernel [k, in, out]        bias [out]      (cross-correlation)
  Conv1DTranspose      kernel [k, out, in]        bias [out]
  LSTMCell             kernel [in, 4u]  recurrent_kernel [u, 4u]  bias [4u]   gate order i, f, c, o
  BatchNormalization   gamma, beta, moving_mean, moving_variance [C]
  Embedding            embeddings [vocab, dim]

There are no pretrained checkpoints on disk and Keras is not installed, so every run uses the seeded
synthetic weights below (SURVEY.md section 8d "Synthetic inputs").

`TTSW` file (little endian):  b"TTSW" u32 version=1 u32 n_tensors, then per tensor
  u32 name_len, name bytes, u32 ndim, i64 dims[ndim], u64 byte offset (from file start), u64 nbytes
followed by the 64-byte aligned float32 payloads.  The C-ABI (`tts_hip_load_weights`) reads the same file.
"""
from __future__ import annotations

import struct
from collections import OrderedDict

import numpy as np

from .config import Tacotron2Config, WaveGlowConfig

MAGIC = b"TTSW"
VERSION = 1


# --------------------------------------------------------------------------------------
# manifests
# --------------------------------------------------------------------------------------

def waveglow_manifest(cfg: WaveGlowConfig = WaveGlowConfig()) -> "OrderedDict[str, tuple]":
    """name -> shape for every WaveGlow tensor (reference waveglow_arch.py:58-88,196-223)."""
    m = OrderedDict()
    C, M = cfg.n_channels, cfg.n_cond
    m['waveglow/upsample/kernel'] = (cfg.upsample_kernel, cfg.n_mel_channels, cfg.n_mel_channels)
    m['waveglow/upsample/bias'] = (cfg.n_mel_channels,)
    for k, (n_rem, n_half) in enumerate(cfg.flow_channels()):
        p = f'waveglow/block-{k}'
        m[f'{p}/start_conv/kernel'] = (1, n_half, C)
        m[f'{p}/start_conv/bias'] = (C,)
        for i in range(cfg.n_layers):
            m[f'{p}/in_conv-{i}/kernel'] = (cfg.kernel_size, C, 2 * C)
            m[f'{p}/in_conv-{i}/bias'] = (2 * C,)
            m[f'{p}/cond_layer-{i}/kernel'] = (1, M, 2 * C)
            m[f'{p}/cond_layer-{i}/bias'] = (2 * C,)
            rs = 2 * C if i < cfg.n_layers - 1 else C
            m[f'{p}/res_skip_conv-{i}/kernel'] = (1, C, rs)
            m[f'{p}/res_skip_conv-{i}/bias'] = (rs,)
        m[f'{p}/end_conv/kernel'] = (1, C, 2 * n_half)
        m[f'{p}/end_conv/bias'] = (2 * n_half,)
        m[f'waveglow/invertible_conv-{k}/conv/kernel'] = (1, n_rem, n_rem)
    return m


def tacotron2_manifest(cfg: Tacotron2Config = Tacotron2Config()) -> "OrderedDict[str, tuple]":
    """name -> shape for every Tacotron2 tensor (reference tacotron2_arch.py:143-232,235-333,336-362,492-509)."""
    m = OrderedDict()
    E, enc = cfg.embedding_dim, cfg.encoder_dim
    m['tacotron2/encoder/embeddings'] = (cfg.vocab_size, E)
    for i in range(cfg.encoder_n_conv):
        m[f'tacotron2/encoder/conv_{i + 1}/kernel'] = (cfg.encoder_kernel_size, E, E)
        m[f'tacotron2/encoder/conv_{i + 1}/bias'] = (E,)
        for s in ('gamma', 'beta', 'moving_mean', 'moving_variance'):
            m[f'tacotron2/encoder/norm_{i + 1}/{s}'] = (E,)
    for d in ('forward', 'backward'):
        m[f'tacotron2/encoder/bi_lstm/{d}/kernel'] = (E, 4 * (E // 2))
        m[f'tacotron2/encoder/bi_lstm/{d}/recurrent_kernel'] = (E // 2, 4 * (E // 2))
        m[f'tacotron2/encoder/bi_lstm/{d}/bias'] = (4 * (E // 2),)
    prev = cfg.n_mel_channels
    for i, s in enumerate(cfg.prenet_sizes):
        m[f'tacotron2/decoder/prenet/layer_{i}/kernel'] = (prev, s)
        prev = s
    A, D = cfg.attention_rnn_dim, cfg.decoder_rnn_dim
    m['tacotron2/decoder/attention_rnn/kernel'] = (prev + enc, 4 * A)
    m['tacotron2/decoder/attention_rnn/recurrent_kernel'] = (A, 4 * A)
    m['tacotron2/decoder/attention_rnn/bias'] = (4 * A,)
    m['tacotron2/decoder/lsa/query_layer/kernel'] = (A, cfg.attention_dim)
    m['tacotron2/decoder/lsa/memory_layer/kernel'] = (enc, cfg.attention_dim)
    m['tacotron2/decoder/lsa/value_layer/kernel'] = (cfg.attention_dim, 1)
    m['tacotron2/decoder/lsa/location_conv/kernel'] = (cfg.attention_kernel_size, 2, cfg.attention_filters)
    m['tacotron2/decoder/lsa/location_dense/kernel'] = (cfg.attention_filters, cfg.attention_dim)
    m['tacotron2/decoder/decoder_rnn/cell_0/kernel'] = (A + enc, 4 * D)
    m['tacotron2/decoder/decoder_rnn/cell_0/recurrent_kernel'] = (D, 4 * D)
    m['tacotron2/decoder/decoder_rnn/cell_0/bias'] = (4 * D,)
    m['tacotron2/decoder/linear_projection/kernel'] = (D + enc, cfg.n_mel_channels)
    m['tacotron2/decoder/linear_projection/bias'] = (cfg.n_mel_channels,)
    m['tacotron2/decoder/gate_output/kernel'] = (D + enc, 1)
    m['tacotron2/decoder/gate_output/bias'] = (1,)
    cin = cfg.n_mel_channels
    for i in range(cfg.postnet_n_conv):
        cout = cfg.postnet_filters if i < cfg.postnet_n_conv - 1 else cfg.n_mel_channels
        m[f'tacotron2/postnet/conv_{i + 1}/kernel'] = (cfg.postnet_kernel_size, cin, cout)
        m[f'tacotron2/postnet/conv_{i + 1}/bias'] = (cout,)
        for s in ('gamma', 'beta', 'moving_mean', 'moving_variance'):
            m[f'tacotron2/postnet/norm_{i + 1}/{s}'] = (cout,)
        cin = cout
    return m


def n_params(manifest) -> int:
    return int(sum(int(np.prod(s)) for s in manifest.values()))


# --------------------------------------------------------------------------------------
# seeded synthetic weights  
# --------------------------------------------------------------------------------------

def _fan_in(name, shape):
    if name.endswith('upsample/kernel'):
        # every output sample sums 4 taps x 80 input channels of the [k, out, in] kernel
        return 4 * shape[2]
    if len(shape) == 3:
        return shape[0] * shape[1]
    return shape[0]


def synth_waveglow(cfg: WaveGlowConfig = WaveGlowConfig(), seed: int = 1234, end_scale: float = 0.05):
    """Seeded synthetic WaveGlow weights.

    dense/conv kernels ~ N(0, 1/fan_in); biases ~ N(0, 0.01); `end_conv` scaled by `end_scale` so that
    exp(-s) stays near 1; invertible 1x1 kernels are QR-orthogonal (the NVIDIA initialisation).
    """
    rng = np.random.default_rng(seed)
    out = OrderedDict()
    for name, shape in waveglow_manifest(cfg).items():
        if 'invertible_conv' in name:
            c = shape[1]
            q, _ = np.linalg.qr(rng.standard_normal((c, c)))
            if np.linalg.det(q) < 0:
                q[:, 0] = -q[:, 0]
            w = q.astype(np.float32)[None]
        elif name.endswith('/bias'):
            w = (0.1 * rng.standard_normal(shape)).astype(np.float32)
            if 'end_conv' in name:
                w *= end_scale
        else:
            w = (rng.standard_normal(shape) / np.sqrt(_fan_in(name, shape))).astype(np.float32)
            if 'end_conv' in name:
                w *= end_scale
        out[name] = np.ascontiguousarray(w, dtype=np.float32)
    return out


def synth_tacotron2(cfg: Tacotron2Config = Tacotron2Config(), seed: int = 1234, gate_bias: float = -3.0,
                    trivial_bn: bool = False):
    """Seeded synthetic Tacotron2 weights.

    dense/conv ~ N(0, 1/fan_in); LSTM recurrent ~ N(0, 1/units); forget-gate bias +1; the gate (stop token) bias
    is a parameter so that tests can script where decoding stops.  Batch-norm statistics are non-trivial unless
    `trivial_bn` (gamma=1, beta=0, mean=0, var=1) so that the load-time BN folding is exercised.
    """
    rng = np.random.default_rng(seed)
    out = OrderedDict()
    for name, shape in tacotron2_manifest(cfg).items():
        leaf = name.rsplit('/', 1)[1]
        if leaf == 'gamma':
            w = np.ones(shape) if trivial_bn else rng.uniform(0.5, 1.5, shape)
        elif leaf == 'beta':
            w = np.zeros(shape) if trivial_bn else 0.1 * rng.standard_normal(shape)
        elif leaf == 'moving_mean':
            w = np.zeros(shape) if trivial_bn else 0.1 * rng.standard_normal(shape)
        elif leaf == 'moving_variance':
            w = np.ones(shape) if trivial_bn else rng.uniform(0.5, 1.5, shape)
        elif leaf == 'embeddings':
            w = rng.standard_normal(shape)
        elif leaf == 'recurrent_kernel':
            w = rng.standard_normal(shape) / np.sqrt(shape[0])
        elif leaf == 'bias':
            if 'gate_output' in name:
                w = np.full(shape, gate_bias)
            elif 'rnn' in name or 'lstm' in name:
                u = shape[0] // 4
                w = 0.05 * rng.standard_normal(shape)
                w[u:2 * u] += 1.0           # forget gate
            else:
                w = 0.05 * rng.standard_normal(shape)
        else:
            w = rng.standard_normal(shape) / np.sqrt(_fan_in(name, shape))
        out[name] = np.ascontiguousarray(w, dtype=np.float32)
    return out


# --------------------------------------------------------------------------------------
# TTSW file
# --------------------------------------------------------------------------------------

def save_ttsw(path, tensors) -> None:
    names = list(tensors.keys())
    header = bytearray()
    header += MAGIC + struct.pack('<II', VERSION, len(names))
    entries = []
    size = len(header)
    for n in names:
        a = tensors[n]
        nb = n.encode('utf-8')
        size += 4 + len(nb) + 4 + 8 * a.ndim + 16
    off = (size + 63) // 64 * 64
    for n in names:
        a = np.ascontiguousarray(tensors[n], dtype=np.float32)
        entries.append((n, a, off))
        off = (off + a.nbytes + 63) // 64 * 64
    for n, a, o in entries:
        nb = n.encode('utf-8')
        header += struct.pack('<I', len(nb)) + nb + struct.pack('<I', a.ndim)
        header += struct.pack(f'<{a.ndim}q', *a.shape) + struct.pack('<QQ', o, a.nbytes)
    with open(path, 'wb') as f:
        f.write(header)
        for n, a, o in entries:
            f.seek(o)
            f.write(a.tobytes())
        f.truncate(off)


def load_ttsw(path) -> "OrderedDict[str, np.ndarray]":
    out = OrderedDict()
    with open(path, 'rb') as f:
        if f.read(4) != MAGIC:
            raise ValueError(f'{path}: not a TTSW file')
        version, n = struct.unpack('<II', f.read(8))
        if version != VERSION:
            raise ValueError(f'{path}: unsupported TTSW version {version}')
        entries = []
        for _ in range(n):
            (ln,) = struct.unpack('<I', f.read(4))
            name = f.read(ln).decode('utf-8')
            (nd,) = struct.unpack('<I', f.read(4))
            dims = struct.unpack(f'<{nd}q', f.read(8 * nd))
            o, nb = struct.unpack('<QQ', f.read(16))
            entries.append((name, dims, o, nb))
        for name, dims, o, nb in entries:
            f.seek(o)
            out[name] = np.frombuffer(f.read(nb), dtype=np.float32).reshape(dims).copy()
    return out


# ---- safetensors interchange (same tensor names and Keras layouts as the TTSW file) -------------------------------------
def save_safetensors(path, tensors) -> None:
    """Writes the manifest tensors as a `.safetensors` file (float32, contiguous), for exchange with other tooling."""
    from safetensors.numpy import save_file
    save_file({k: np.ascontiguousarray(v, dtype=np.float32) for k, v in tensors.items()}, str(path))


def load_safetensors(path) -> "OrderedDict[str, np.ndarray]":
    """Reads a `.safetensors` file whose tensors use the manifest names (`waveglow/...`, `tacotron2/...`)."""
    from safetensors.numpy import load_file
    loaded = load_file(str(path))
    return OrderedDict((k, np.ascontiguousarray(v, dtype=np.float32)) for k, v in sorted(loaded.items()))
