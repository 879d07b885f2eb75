"""Import NVIDIA Tacotron2 / WaveGlow PyTorch checkpoints into the engine's tensor manifest (and a TTSW file).

The reference's shipped models are NVIDIA's checkpoints converted to Keras (`architectures/tacotron2_arch.py:934-941`,
`waveglow_arch.py:327-335` load them through torch.hub; `models/weights_converter.py:252-322` re-lays them out).  This
module restates those layout rules so a `.pt` state dict can be used directly:

  Linear            torch [out, in]            -> Dense kernel [in, out]
  Conv1d            torch [out, in, k]         -> Conv1D kernel [k, in, out]
  ConvTranspose1d   torch [in, out, k]         -> Conv1DTranspose kernel [k, out, in]
  LSTM / LSTMCell   weight_ih [4u, in], weight_hh [4u, u], bias_ih + bias_hh   (gate order i, f, g, o = Keras i, f, c, o)
                    -> kernel [in, 4u], recurrent_kernel [u, 4u], bias [4u]
  BatchNorm1d       weight, bias, running_mean, running_var -> gamma, beta, moving_mean, moving_variance
  weight_norm       weight_g, weight_v -> weight = g * v / ||v||   (norm over every dim but 0)
  WaveGlow `cond_layer` (fused [8 * 1024, 640, 1]) is split per WN layer.

No checkpoint ships with this repository (no network); `to_nvidia_*` builds NVIDIA-layout state dicts from the synthetic
weights so that the mapping is exercised end to end by tests/test_weights_import.py.

CLI:  python -m text_to_speech_amd.weights_import --tacotron2 tacotron2.pt --waveglow waveglow.pt -o model.ttsw
"""
from __future__ import annotations

import argparse
import os
from collections import OrderedDict

import numpy as np

from .config import Tacotron2Config, WaveGlowConfig
from .weights import save_ttsw, tacotron2_manifest, waveglow_manifest


def _np(x):
    if hasattr(x, 'detach'):
        x = x.detach().cpu().numpy()
    return np.asarray(x, dtype=np.float32)


def _strip(sd):
    """Accepts a raw checkpoint ({'state_dict': ...}), DataParallel prefixes, and resolves weight_norm pairs."""
    if 'state_dict' in sd and not any(k.endswith('.weight') for k in sd):
        sd = sd['state_dict']
    out = {}
    for k, v in sd.items():
        k = k[7:] if k.startswith('module.') else k
        if k.endswith('num_batches_tracked'):
            continue
        out[k] = _np(v)
    for k in [k for k in out if k.endswith('.weight_g')]:
        base = k[:-len('.weight_g')]
        g, v = out.pop(k), out.pop(base + '.weight_v')
        norm = np.sqrt((v.astype(np.float64) ** 2).sum(axis=tuple(range(1, v.ndim)), keepdims=True))
        out[base + '.weight'] = (g.astype(np.float64) * v / norm).astype(np.float32)
    return out


def _lstm(sd, prefix, suffix=''):
    w_ih, w_hh = sd[f'{prefix}weight_ih{suffix}'], sd[f'{prefix}weight_hh{suffix}']
    b = sd[f'{prefix}bias_ih{suffix}'] + sd[f'{prefix}bias_hh{suffix}']
    return w_ih.T.copy(), w_hh.T.copy(), b


def from_nvidia_tacotron2(state_dict, cfg: Tacotron2Config = Tacotron2Config()):
    sd = _strip(state_dict)
    o = OrderedDict()
    p = 'tacotron2'
    o[f'{p}/encoder/embeddings'] = sd['embedding.weight']
    for i in range(cfg.encoder_n_conv):
        o[f'{p}/encoder/conv_{i + 1}/kernel'] = sd[f'encoder.convolutions.{i}.0.conv.weight'].transpose(2, 1, 0).copy()
        o[f'{p}/encoder/conv_{i + 1}/bias'] = sd[f'encoder.convolutions.{i}.0.conv.bias']
        for src, dst in (('weight', 'gamma'), ('bias', 'beta'), ('running_mean', 'moving_mean'),
                         ('running_var', 'moving_variance')):
            o[f'{p}/encoder/norm_{i + 1}/{dst}'] = sd[f'encoder.convolutions.{i}.1.{src}']
    for d, suf in (('forward', '_l0'), ('backward', '_l0_reverse')):
        k, r, b = _lstm(sd, 'encoder.lstm.', suf)
        o[f'{p}/encoder/bi_lstm/{d}/kernel'], o[f'{p}/encoder/bi_lstm/{d}/recurrent_kernel'] = k, r
        o[f'{p}/encoder/bi_lstm/{d}/bias'] = b
    for i in range(len(cfg.prenet_sizes)):
        o[f'{p}/decoder/prenet/layer_{i}/kernel'] = sd[f'decoder.prenet.layers.{i}.linear_layer.weight'].T.copy()
    for src, dst in (('decoder.attention_rnn.', 'attention_rnn'), ('decoder.decoder_rnn.', 'decoder_rnn/cell_0')):
        k, r, b = _lstm(sd, src)
        o[f'{p}/decoder/{dst}/kernel'], o[f'{p}/decoder/{dst}/recurrent_kernel'], o[f'{p}/decoder/{dst}/bias'] = k, r, b
    a = 'decoder.attention_layer.'
    o[f'{p}/decoder/lsa/query_layer/kernel'] = sd[a + 'query_layer.linear_layer.weight'].T.copy()
    o[f'{p}/decoder/lsa/memory_layer/kernel'] = sd[a + 'memory_layer.linear_layer.weight'].T.copy()
    o[f'{p}/decoder/lsa/value_layer/kernel'] = sd[a + 'v.linear_layer.weight'].T.copy()
    o[f'{p}/decoder/lsa/location_conv/kernel'] = sd[a + 'location_layer.location_conv.conv.weight'].transpose(2, 1, 0).copy()
    o[f'{p}/decoder/lsa/location_dense/kernel'] = sd[a + 'location_layer.location_dense.linear_layer.weight'].T.copy()
    o[f'{p}/decoder/linear_projection/kernel'] = sd['decoder.linear_projection.linear_layer.weight'].T.copy()
    o[f'{p}/decoder/linear_projection/bias'] = sd['decoder.linear_projection.linear_layer.bias']
    o[f'{p}/decoder/gate_output/kernel'] = sd['decoder.gate_layer.linear_layer.weight'].T.copy()
    o[f'{p}/decoder/gate_output/bias'] = sd['decoder.gate_layer.linear_layer.bias']
    for i in range(cfg.postnet_n_conv):
        o[f'{p}/postnet/conv_{i + 1}/kernel'] = sd[f'postnet.convolutions.{i}.0.conv.weight'].transpose(2, 1, 0).copy()
        o[f'{p}/postnet/conv_{i + 1}/bias'] = sd[f'postnet.convolutions.{i}.0.conv.bias']
        for src, dst in (('weight', 'gamma'), ('bias', 'beta'), ('running_mean', 'moving_mean'),
                         ('running_var', 'moving_variance')):
            o[f'{p}/postnet/norm_{i + 1}/{dst}'] = sd[f'postnet.convolutions.{i}.1.{src}']
    return _checked(o, tacotron2_manifest(cfg))


def from_nvidia_waveglow(state_dict, cfg: WaveGlowConfig = WaveGlowConfig()):
    sd = _strip(state_dict)
    o = OrderedDict()
    o['waveglow/upsample/kernel'] = sd['upsample.weight'].transpose(2, 1, 0).copy()       # [in, out, k] -> [k, out, in]
    o['waveglow/upsample/bias'] = sd['upsample.bias']
    C2 = 2 * cfg.n_channels
    for k in range(cfg.n_flows):
        p, q = f'waveglow/block-{k}', f'WN.{k}.'
        o[f'{p}/start_conv/kernel'] = sd[q + 'start.weight'].transpose(2, 1, 0).copy()
        o[f'{p}/start_conv/bias'] = sd[q + 'start.bias']
        fused = q + 'cond_layer.weight' in sd
        for i in range(cfg.n_layers):
            o[f'{p}/in_conv-{i}/kernel'] = sd[q + f'in_layers.{i}.weight'].transpose(2, 1, 0).copy()
            o[f'{p}/in_conv-{i}/bias'] = sd[q + f'in_layers.{i}.bias']
            if fused:
                w = sd[q + 'cond_layer.weight'][i * C2:(i + 1) * C2]
                b = sd[q + 'cond_layer.bias'][i * C2:(i + 1) * C2]
            else:
                w, b = sd[q + f'cond_layers.{i}.weight'], sd[q + f'cond_layers.{i}.bias']
            o[f'{p}/cond_layer-{i}/kernel'] = w.transpose(2, 1, 0).copy()
            o[f'{p}/cond_layer-{i}/bias'] = b.copy()
            o[f'{p}/res_skip_conv-{i}/kernel'] = sd[q + f'res_skip_layers.{i}.weight'].transpose(2, 1, 0).copy()
            o[f'{p}/res_skip_conv-{i}/bias'] = sd[q + f'res_skip_layers.{i}.bias']
        o[f'{p}/end_conv/kernel'] = sd[q + 'end.weight'].transpose(2, 1, 0).copy()
        o[f'{p}/end_conv/bias'] = sd[q + 'end.bias']
        o[f'waveglow/invertible_conv-{k}/conv/kernel'] = sd[f'convinv.{k}.conv.weight'].transpose(2, 1, 0).copy()
    return _checked(o, waveglow_manifest(cfg))


# ---- Keras checkpoints (the reference's own `.weights.h5` files) ------------------------------------------------------------
# The reference stores its models as Keras 3 `ckpt-XXXX.weights.h5` (custom_train_objects/checkpoint_manager.py:148-215).
# Two routes.  The direct one (`from_keras_h5`, further down) reads the H5 file itself.  The name-based one, which lets
# Keras itself resolve the file, has two steps:  (1) where the reference runs (Keras + h5py installed), `scripts/export_keras_weights.py <model name>`
# restores the model with the reference's own code and writes `{variable.path: value}` to a `.safetensors` file;
# (2) here, `from_keras_variables` maps those variable paths onto the engine's manifest.  Keras layouts ARE the manifest's
# layouts (Dense [in, out], Conv1D [k, in, out], Conv1DTranspose [k, out, in], LSTM kernel / recurrent_kernel / bias with
# gates i, f, c, o -- models/weights_converter.py:252-322 is only needed for torch checkpoints), so this is renaming plus a
# shape check.  Layer names come from the reference source: architectures/tacotron2_arch.py:80-107,168,248,299,347,359-361,
# 503-508, architectures/layers/location_sensitive_attention.py:36-59, architectures/waveglow_arch.py:58-87,197,213,222,
# architectures/layers/invertible_conv.py:32.
_KERAS_VAR = {'kernel', 'recurrent_kernel', 'bias', 'gamma', 'beta', 'moving_mean', 'moving_variance', 'embeddings'}


def _keras_target(path: str, model: str):
    """Manifest name for one Keras variable path, or None if the variable is not part of the inference path."""
    import re
    parts = [p for p in path.replace(':0', '').split('/') if p]
    if not parts or parts[-1] not in _KERAS_VAR:
        return None
    var, scope = parts[-1], parts[:-1]
    has = lambda name: any(p == name for p in scope)
    find = lambda pat: next((m for m in (re.fullmatch(pat, p) for p in scope) if m), None)
    if model == 'waveglow':
        if has('upsample'):
            return f'waveglow/upsample/{var}'
        inv = find(r'invertible_conv-(\d+)')
        if inv and var == 'kernel':
            return f'waveglow/invertible_conv-{inv.group(1)}/conv/kernel'
        blk = find(r'block-(\d+)')
        if blk:
            for pat in (r'start_conv', r'end_conv', r'in_conv-\d+', r'cond_layer-\d+', r'res_skip_conv-\d+'):
                hit = find(pat)
                if hit:
                    return f'waveglow/block-{blk.group(1)}/{hit.group(0)}/{var}'
        return None
    # tacotron2
    if var == 'embeddings':
        return 'tacotron2/encoder/embeddings' if not has('speaker_embedding') else None
    conv, norm = find(r'conv_(\d+)'), find(r'norm_(\d+)')
    section = 'postnet' if has('postnet') else 'encoder' if has('encoder') else None
    if section and conv and var in ('kernel', 'bias'):
        return f'tacotron2/{section}/conv_{conv.group(1)}/{var}'
    if section and norm:
        return f'tacotron2/{section}/norm_{norm.group(1)}/{var}'
    if var in ('kernel', 'recurrent_kernel', 'bias'):
        direction = next((d for d in ('forward', 'backward') if any(p.startswith(d) for p in scope)), None)
        if direction:
            return f'tacotron2/encoder/bi_lstm/{direction}/{var}'
        if has('attention_rnn'):
            return f'tacotron2/decoder/attention_rnn/{var}'
        cell = find(r'cell_(\d+)')
        if cell and cell.group(1) == '0':
            return f'tacotron2/decoder/decoder_rnn/cell_0/{var}'
    layer = find(r'layer_(\d+)')
    if has('prenet') and layer and var == 'kernel':
        return f'tacotron2/decoder/prenet/layer_{layer.group(1)}/kernel'
    for name in ('query_layer', 'memory_layer', 'value_layer', 'location_conv', 'location_dense'):
        if has(name) and var == 'kernel':
            return f'tacotron2/decoder/lsa/{name}/kernel'
    for name in ('linear_projection', 'gate_output'):
        if has(name):
            return f'tacotron2/decoder/{name}/{var}'
    return None


def from_keras_variables(named, model: str, cfg=None):
    """{Keras variable path: array} (as written by scripts/export_keras_weights.py) -> manifest tensors of `model`
    ('tacotron2' | 'waveglow').  Every manifest tensor must be matched exactly once with the right shape; variables that
    do not belong to inference (optimizer slots, a speaker-embedding table, ...) are ignored."""
    if model == 'tacotron2':
        manifest = tacotron2_manifest(cfg or Tacotron2Config())
    elif model == 'waveglow':
        manifest = waveglow_manifest(cfg or WaveGlowConfig())
    else:
        raise ValueError(f"model must be 'tacotron2' or 'waveglow', got {model!r}")
    out, origin = {}, {}
    for path, value in named.items():
        target = _keras_target(path, model)
        if target is None or target not in manifest:
            continue
        if target in out:
            raise ValueError(f'{origin[target]!r} and {path!r} both map to {target}')
        out[target], origin[target] = _np(value), path
    return _checked(out, manifest)


# ---- Keras `.weights.h5` read directly (pure-Python HDF5 reader; no Keras, no h5py) --------------------------------------------
# Keras 3 `save_weights` does not store variable names: `saving_lib._save_state` walks the OBJECT TREE and writes each
# layer's variables as `<path>/vars/<i>` (i = position in `trainable + non-trainable` order), where <path> is made of
# attribute names (`decoder/cell/attention_rnn`) and, inside lists and Functional / Sequential models, of the snake-cased
# class name with a per-container counter (`layers/conv1d`, `layers/conv1d_1`, `denses/dense_1`).  The tree below is read off
# the reference's classes (architectures/tacotron2_arch.py:143-171 Prenet, :214-333 Postnet / Encoder (Functional models
# built by simple_cnn), :336-362 DecoderCell, :492-509 Decoder, :752-795 Tacotron2; layers/location_sensitive_attention.py:
# 27-75; waveglow_arch.py:27-90 WaveglowBlock, :159-225 WaveGlow; layers/invertible_conv.py:16-37).  One point depends on
# the Keras release: whether the walk also descends through `Model.layers` (alphabetically between a subclassed model's own
# attributes); where that could move a layer, BOTH paths are accepted and the file decides.  Every tensor is shape-checked
# against the manifest.  Unverified against a real checkpoint of the reference (none is available here); the H5 parsing
# itself is verified against libhdf5-written files and the tree against hand-written layouts (tests/test_keras_h5.py).
_BN_VARS = ('gamma', 'beta', 'moving_mean', 'moving_variance')
_LSTM_VARS = ('kernel', 'recurrent_kernel', 'bias')


def _nth(name, k):
    return name if k == 0 else f'{name}_{k}'


def keras_h5_layout(model: str, cfg=None):
    """[(manifest prefix, variable names in `vars/<i>` order, candidate H5 scopes)] for `model`."""
    out = []
    if model == 'tacotron2':
        cfg = cfg or Tacotron2Config()
        convs = ('conv1d', 'masked_conv1d')
        out.append(('tacotron2/encoder', ('embeddings',), ['encoder/layers/custom_embedding']))
        for section, n, roots in (('encoder', cfg.encoder_n_conv, ['encoder/layers']),
                                  ('postnet', cfg.postnet_n_conv, ['postnet/layers', 'layers/functional_1/layers'])):
            for i in range(n):
                out.append((f'tacotron2/{section}/conv_{i + 1}', ('kernel', 'bias'),
                            [f'{r}/{_nth(c, i)}' for r in roots for c in convs]))
                out.append((f'tacotron2/{section}/norm_{i + 1}', _BN_VARS,
                            [f'{r}/{_nth("batch_normalization", i)}' for r in roots]))
        for d in ('forward', 'backward'):
            out.append((f'tacotron2/encoder/bi_lstm/{d}', _LSTM_VARS, [f'encoder/layers/bidirectional/{d}_layer/cell']))
        for i in range(len(cfg.prenet_sizes)):
            out.append((f'tacotron2/decoder/prenet/layer_{i}', ('kernel',),
                        [f'{r}/denses/{_nth("dense", i)}' for r in ('decoder/prenet', 'decoder/layers/tacotron2_prenet')]))
        cell = 'decoder/cell'
        out.append(('tacotron2/decoder/attention_rnn', _LSTM_VARS, [f'{cell}/attention_rnn']))
        for name in ('query_layer', 'memory_layer', 'value_layer'):
            out.append((f'tacotron2/decoder/lsa/{name}', ('kernel',), [f'{cell}/attention_layer/{name}']))
        out.append(('tacotron2/decoder/lsa/location_conv', ('kernel',), [f'{cell}/attention_layer/location_layer/layers/conv1d']))
        out.append(('tacotron2/decoder/lsa/location_dense', ('kernel',), [f'{cell}/attention_layer/location_layer/layers/dense']))
        out.append(('tacotron2/decoder/decoder_rnn/cell_0', _LSTM_VARS, [f'{cell}/decoder_rnn/cells/lstm_cell']))
        out.append(('tacotron2/decoder/linear_projection', ('kernel', 'bias'), ['decoder/linear_projection', 'decoder/layers/dense']))
        out.append(('tacotron2/decoder/gate_output', ('kernel', 'bias'), ['decoder/gate_layer']))
    elif model == 'waveglow':
        cfg = cfg or WaveGlowConfig()
        out.append(('waveglow/upsample', ('kernel', 'bias'), ['upsample', 'layers/conv1d_transpose']))
        for k in range(cfg.n_flows):
            out.append((f'waveglow/invertible_conv-{k}/conv', ('kernel',), [f'convinv/{_nth("invertible1x1_conv", k)}/conv']))
            blk = f'blocks/{_nth("waveglow_block", k)}'
            # order of WaveglowBlock.layers (waveglow_arch.py:58-90): start, end, then in / cond / res_skip per layer
            out.append((f'waveglow/block-{k}/start_conv', ('kernel', 'bias'), [f'{blk}/start', f'{blk}/layers/conv1d']))
            out.append((f'waveglow/block-{k}/end_conv', ('kernel', 'bias'), [f'{blk}/end']))
            for i in range(cfg.n_layers):
                out.append((f'waveglow/block-{k}/in_conv-{i}', ('kernel', 'bias'), [f'{blk}/in_layers/{_nth("conv1d", i)}']))
                out.append((f'waveglow/block-{k}/cond_layer-{i}', ('kernel', 'bias'), [f'{blk}/cond_layers/{_nth("conv1d", i)}']))
                out.append((f'waveglow/block-{k}/res_skip_conv-{i}', ('kernel', 'bias'),
                            [f'{blk}/res_skip_layers/{_nth("conv1d", i)}', f'{blk}/layers/{_nth("conv1d", 4 + 3 * i)}']))
    else:
        raise ValueError(f"model must be 'tacotron2' or 'waveglow', got {model!r}")
    return out


def from_keras_h5(path, model: str, cfg=None):
    """A Keras 3 `.weights.h5` file of the reference's Tacotron2 / WaveGlow -> manifest tensors (see the note above)."""
    from .hdf5_reader import H5File
    if model == 'tacotron2':
        manifest = tacotron2_manifest(cfg or Tacotron2Config())
    elif model == 'waveglow':
        manifest = waveglow_manifest(cfg or WaveGlowConfig())
    else:
        raise ValueError(f"model must be 'tacotron2' or 'waveglow', got {model!r}")
    with H5File(path) as f:
        scopes = {}
        for p, ds in f.datasets().items():
            parts = p.strip('/').split('/')
            if len(parts) >= 2 and parts[-2] == 'vars' and parts[-1].isdigit():
                scopes.setdefault('/'.join(parts[:-2]), {})[int(parts[-1])] = ds
        out = {}
        for prefix, names, candidates in keras_h5_layout(model, cfg):
            wanted = [n for n in names if f'{prefix}/{n}' in manifest]
            if not wanted:
                continue
            present = [c for c in candidates if c in scopes]
            if len(present) != 1:
                near = sorted(s for s in scopes if s.split('/')[0] == candidates[0].split('/')[0])[:12]
                raise KeyError(f'{prefix}: expected exactly one of the H5 groups {candidates} (+ /vars), found {present}; '
                               f'groups of the file under the same root: {near}')
            found = scopes[present[0]]
            if sorted(found) != list(range(len(names))) and sorted(found) != list(range(len(wanted))):
                raise ValueError(f'{prefix}: H5 group {present[0]}/vars holds variables {sorted(found)}, expected {len(names)} ({names})')
            order = names if len(found) == len(names) else wanted
            for i, n in enumerate(order):
                if f'{prefix}/{n}' in manifest:
                    out[f'{prefix}/{n}'] = found[i].read()
    return _checked(out, manifest)


def from_keras_archive(path, model: str, cfg=None):
    """A Keras 3 `.keras` archive (`model.save(...)`, /root/reference/custom_train_objects/checkpoint_manager.py:155,196: the
    manager writes and restores them beside `.weights.h5` files) -> manifest tensors.  The archive is a zip holding
    `config.json`, `metadata.json` and `model.weights.h5`; that member is written by the same object-tree walk as
    `save_weights` (keras.saving.saving_lib: `_save_state` behind both), so it goes through `from_keras_h5` unchanged once it
    is a file: it is extracted to a temporary file beside nothing else (the pure-Python HDF5 reader seeks).  Only the weights
    member is read -- the pickled / JSON model definition is never evaluated."""
    import tempfile
    import zipfile
    if not zipfile.is_zipfile(path):
        raise ValueError(f'{path}: not a zip archive (a `.keras` file is a zip with model.weights.h5 inside)')
    with zipfile.ZipFile(path) as z:
        members = [n for n in z.namelist() if n.rsplit('/', 1)[-1] == 'model.weights.h5']
        if len(members) != 1:
            raise ValueError(f'{path}: expected one model.weights.h5 member, found {members or "none"} in {z.namelist()[:8]}')
        info = z.getinfo(members[0])
        if info.file_size > (64 << 30):
            raise ValueError(f'{path}: {members[0]} claims {info.file_size} bytes')
        with tempfile.TemporaryDirectory(prefix='tts_keras_') as tmp:
            out = os.path.join(tmp, 'model.weights.h5')
            with z.open(info) as src, open(out, 'wb') as dst:          # (never extract by member name: no path traversal)
                while True:
                    block = src.read(1 << 24)
                    if not block:
                        break
                    dst.write(block)
            return from_keras_h5(out, model, cfg)


def from_keras_file(path, model: str, cfg=None):
    """`.weights.h5` or `.keras`, by extension (the two formats CheckpointManager.load accepts, checkpoint_manager.py:196)."""
    return from_keras_archive(path, model, cfg) if str(path).endswith('.keras') else from_keras_h5(path, model, cfg)


def _checked(tensors, manifest):
    out = OrderedDict()
    for name, shape in manifest.items():
        if name not in tensors:
            raise KeyError(f'checkpoint has no tensor for {name}')
        a = np.ascontiguousarray(tensors[name], dtype=np.float32)
        if tuple(a.shape) != tuple(shape):
            raise ValueError(f'{name}: converted shape {a.shape} != expected {shape}')
        out[name] = a
    return out


# ---------------------------------------------------------------------------------------------------------------------
# inverse mapping (tests / exporting synthetic weights to the NVIDIA layout)
# ---------------------------------------------------------------------------------------------------------------------
def to_nvidia_tacotron2(w, cfg: Tacotron2Config = Tacotron2Config(), rng=None):
    """Keras-layout tensors -> NVIDIA state-dict layout.  bias_ih / bias_hh get a random split of the summed bias."""
    rng = rng or np.random.default_rng(0)
    p, sd = 'tacotron2', OrderedDict()
    sd['embedding.weight'] = w[f'{p}/encoder/embeddings']

    def bn(dst, src):
        for a, b in (('weight', 'gamma'), ('bias', 'beta'), ('running_mean', 'moving_mean'), ('running_var', 'moving_variance')):
            sd[f'{dst}.{a}'] = w[f'{src}/{b}']
        sd[f'{dst}.num_batches_tracked'] = np.zeros((), np.float32)

    def lstm(dst, src, suf=''):
        b = w[f'{src}/bias']
        b_ih = rng.standard_normal(b.shape).astype(np.float32)
        sd[f'{dst}weight_ih{suf}'] = w[f'{src}/kernel'].T.copy()
        sd[f'{dst}weight_hh{suf}'] = w[f'{src}/recurrent_kernel'].T.copy()
        sd[f'{dst}bias_ih{suf}'], sd[f'{dst}bias_hh{suf}'] = b_ih, b - b_ih

    for i in range(cfg.encoder_n_conv):
        sd[f'encoder.convolutions.{i}.0.conv.weight'] = w[f'{p}/encoder/conv_{i + 1}/kernel'].transpose(2, 1, 0).copy()
        sd[f'encoder.convolutions.{i}.0.conv.bias'] = w[f'{p}/encoder/conv_{i + 1}/bias']
        bn(f'encoder.convolutions.{i}.1', f'{p}/encoder/norm_{i + 1}')
    lstm('encoder.lstm.', f'{p}/encoder/bi_lstm/forward', '_l0')
    lstm('encoder.lstm.', f'{p}/encoder/bi_lstm/backward', '_l0_reverse')
    for i in range(len(cfg.prenet_sizes)):
        sd[f'decoder.prenet.layers.{i}.linear_layer.weight'] = w[f'{p}/decoder/prenet/layer_{i}/kernel'].T.copy()
    lstm('decoder.attention_rnn.', f'{p}/decoder/attention_rnn')
    lstm('decoder.decoder_rnn.', f'{p}/decoder/decoder_rnn/cell_0')
    a = 'decoder.attention_layer.'
    sd[a + 'query_layer.linear_layer.weight'] = w[f'{p}/decoder/lsa/query_layer/kernel'].T.copy()
    sd[a + 'memory_layer.linear_layer.weight'] = w[f'{p}/decoder/lsa/memory_layer/kernel'].T.copy()
    sd[a + 'v.linear_layer.weight'] = w[f'{p}/decoder/lsa/value_layer/kernel'].T.copy()
    sd[a + 'location_layer.location_conv.conv.weight'] = w[f'{p}/decoder/lsa/location_conv/kernel'].transpose(2, 1, 0).copy()
    sd[a + 'location_layer.location_dense.linear_layer.weight'] = w[f'{p}/decoder/lsa/location_dense/kernel'].T.copy()
    sd['decoder.linear_projection.linear_layer.weight'] = w[f'{p}/decoder/linear_projection/kernel'].T.copy()
    sd['decoder.linear_projection.linear_layer.bias'] = w[f'{p}/decoder/linear_projection/bias']
    sd['decoder.gate_layer.linear_layer.weight'] = w[f'{p}/decoder/gate_output/kernel'].T.copy()
    sd['decoder.gate_layer.linear_layer.bias'] = w[f'{p}/decoder/gate_output/bias']
    for i in range(cfg.postnet_n_conv):
        sd[f'postnet.convolutions.{i}.0.conv.weight'] = w[f'{p}/postnet/conv_{i + 1}/kernel'].transpose(2, 1, 0).copy()
        sd[f'postnet.convolutions.{i}.0.conv.bias'] = w[f'{p}/postnet/conv_{i + 1}/bias']
        bn(f'postnet.convolutions.{i}.1', f'{p}/postnet/norm_{i + 1}')
    return sd


def to_nvidia_waveglow(w, cfg: WaveGlowConfig = WaveGlowConfig(), fused_cond=False, weight_norm=False, rng=None):
    rng = rng or np.random.default_rng(0)
    sd = OrderedDict()

    def put(name, weight):
        if weight_norm:                                         # g * v / ||v|| with a random rescaling of v
            scale = rng.uniform(0.5, 2.0, (weight.shape[0],) + (1,) * (weight.ndim - 1)).astype(np.float32)
            v = weight * scale
            sd[name + '_v'] = v
            sd[name + '_g'] = np.sqrt((weight.astype(np.float64) ** 2).sum(axis=tuple(range(1, weight.ndim)),
                                                                            keepdims=True)).astype(np.float32)
        else:
            sd[name] = weight

    put('upsample.weight', w['waveglow/upsample/kernel'].transpose(2, 1, 0).copy())
    sd['upsample.bias'] = w['waveglow/upsample/bias']
    for k in range(cfg.n_flows):
        p, q = f'waveglow/block-{k}', f'WN.{k}.'
        put(q + 'start.weight', w[f'{p}/start_conv/kernel'].transpose(2, 1, 0).copy())
        sd[q + 'start.bias'] = w[f'{p}/start_conv/bias']
        cw, cb = [], []
        for i in range(cfg.n_layers):
            put(q + f'in_layers.{i}.weight', w[f'{p}/in_conv-{i}/kernel'].transpose(2, 1, 0).copy())
            sd[q + f'in_layers.{i}.bias'] = w[f'{p}/in_conv-{i}/bias']
            cwi = w[f'{p}/cond_layer-{i}/kernel'].transpose(2, 1, 0).copy()
            if fused_cond:
                cw.append(cwi)
                cb.append(w[f'{p}/cond_layer-{i}/bias'])
            else:
                put(q + f'cond_layers.{i}.weight', cwi)
                sd[q + f'cond_layers.{i}.bias'] = w[f'{p}/cond_layer-{i}/bias']
            put(q + f'res_skip_layers.{i}.weight', w[f'{p}/res_skip_conv-{i}/kernel'].transpose(2, 1, 0).copy())
            sd[q + f'res_skip_layers.{i}.bias'] = w[f'{p}/res_skip_conv-{i}/bias']
        if fused_cond:
            put(q + 'cond_layer.weight', np.concatenate(cw, 0))
            sd[q + 'cond_layer.bias'] = np.concatenate(cb, 0)
        sd[q + 'end.weight'] = w[f'{p}/end_conv/kernel'].transpose(2, 1, 0).copy()
        sd[q + 'end.bias'] = w[f'{p}/end_conv/bias']
        sd[f'convinv.{k}.conv.weight'] = w[f'waveglow/invertible_conv-{k}/conv/kernel'].transpose(2, 1, 0).copy()
    return sd


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split('\n')[0])
    ap.add_argument('--tacotron2', help='NVIDIA Tacotron2 checkpoint (.pt)')
    ap.add_argument('--waveglow', help='NVIDIA WaveGlow checkpoint (.pt, weight norm allowed)')
    ap.add_argument('--keras-tacotron2', help='Keras variables of the Tacotron2 model (scripts/export_keras_weights.py output)')
    ap.add_argument('--keras-waveglow', help='Keras variables of the WaveGlow model (scripts/export_keras_weights.py output)')
    ap.add_argument('--keras-h5-tacotron2', help="the reference's Tacotron2 `.weights.h5` checkpoint (or `.keras` archive), read directly")
    ap.add_argument('--keras-h5-waveglow', help="the reference's WaveGlow `.weights.h5` checkpoint (or `.keras` archive), read directly")
    ap.add_argument('--speaker-embedding-dim', type=int, default=0, help='256 for the SV2TTS Tacotron2')
    ap.add_argument('-o', '--output', required=True, help='TTSW file to write')
    args = ap.parse_args(argv)
    tensors = OrderedDict()
    if args.keras_tacotron2 or args.keras_waveglow:
        from safetensors.numpy import load_file
        if args.keras_tacotron2:
            tensors.update(from_keras_variables(load_file(args.keras_tacotron2), 'tacotron2',
                                                Tacotron2Config(speaker_embedding_dim=args.speaker_embedding_dim)))
        if args.keras_waveglow:
            tensors.update(from_keras_variables(load_file(args.keras_waveglow), 'waveglow'))
    if args.keras_h5_tacotron2:
        tensors.update(from_keras_file(args.keras_h5_tacotron2, 'tacotron2',
                                     Tacotron2Config(speaker_embedding_dim=args.speaker_embedding_dim)))
    if args.keras_h5_waveglow:
        tensors.update(from_keras_file(args.keras_h5_waveglow, 'waveglow'))
    if args.tacotron2 or args.waveglow:
        import torch
    if args.tacotron2:
        ck = torch.load(args.tacotron2, map_location='cpu')
        tensors.update(from_nvidia_tacotron2(ck.get('state_dict', ck)))
    if args.waveglow:
        ck = torch.load(args.waveglow, map_location='cpu')
        ck = ck.get('state_dict', ck) if isinstance(ck, dict) else ck
        if hasattr(ck, 'state_dict'):
            ck = ck.state_dict()
        tensors.update(from_nvidia_waveglow(ck))
    if not tensors:
        ap.error('nothing to convert')
    save_ttsw(args.output, tensors)
    print(f'wrote {args.output}: {len(tensors)} tensors')


if __name__ == '__main__':
    main()
