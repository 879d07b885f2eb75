"""Opening a model directory of the reference (`pretrained_models/<name>/`) with this engine.

Layout written by the reference (models/interfaces/base_model.py:127-137,713-758, base_text_model.py:38-96,
custom_train_objects/checkpoint_manager.py:23-26,70-98,137-143,237-244, models/tts/sv2tts_tacotron2.py:25,53-67):

    <name>/config.json                  {"class_name": "Tacotron2" | "SV2TTSTacotron2" | "WaveGlow", "config": {...}}
    <name>/saving/tokenizer.json        Tokenizer.get_config()                               (text models)
    <name>/saving/checkpoint.json       {"counter", "loaded", "checkpoints": [{"epoch", "step", "counter"}, ...], ...}
    <name>/saving/ckpt-0003.weights.h5  Keras 3 weights ('ckpt.weights.h5' format + "-{counter:04d}"), or best.weights.h5
    <name>/embeddings/embeddings.h5     default speaker embeddings                           (SV2TTS)

`load_model(dir)` converts the checkpoint once (weights_import.from_keras_h5 -> a TTSW file cached next to it), builds the
HIP runtime on it and returns the same wrapper `get_models` returns.  Everything up to the engine is CPU code and is
tested on hand-made model directories; the engine step needs a GPU and a real checkpoint.
"""
from __future__ import annotations

import glob
import json
import logging
import os

from .config import Tacotron2Config, WaveGlowConfig

logger = logging.getLogger(__name__)

_SYNTHESIZERS = ('Tacotron2', 'SV2TTSTacotron2')


def _load_json(path, default=None):
    try:
        with open(path, encoding='utf-8') as fh:
            return json.load(fh)
    except (OSError, ValueError):
        return default


def find_checkpoint(save_dir):
    """The weight file `CheckpointManager.load()` would restore (checkpoint_manager.py:90-98,169-193): `best.weights.h5` when
    the state says 'best', else the entry the state marks as loaded (the last one by default), named
    `ckpt-{counter:04d}.weights.h5`; without a usable state file, the newest `*.weights.h5` of the directory."""
    state = _load_json(os.path.join(save_dir, 'checkpoint.json'), {})
    entries = state.get('checkpoints') or []
    loaded = state.get('loaded', -1)
    candidates = []
    if loaded == 'best':
        candidates.append(os.path.join(save_dir, 'best.weights.h5'))
    if entries:
        idx = loaded if isinstance(loaded, int) and not isinstance(loaded, bool) and -len(entries) <= loaded < len(entries) else -1
        info = entries[idx]
        for pattern in ('ckpt-{counter:04d}.weights.h5', 'ckpt-{counter:04d}.keras'):
            try:
                candidates.append(os.path.join(save_dir, pattern.format(**info)))
            except (KeyError, ValueError, TypeError):
                pass
    for c in candidates:
        if os.path.exists(c):
            return c
    found = sorted(glob.glob(os.path.join(save_dir, '*.weights.h5')), key=os.path.getmtime)
    if found:
        return found[-1]
    raise FileNotFoundError(f'no Keras checkpoint (*.weights.h5) in {save_dir}')


def read_model_dir(model_dir):
    """{'class_name', 'config', 'model' ('tacotron2' | 'waveglow'), 'checkpoint', 'tokenizer_file', 'embeddings_dir', 'lang',
    'speaker_embedding_dim'} of a reference model directory."""
    model_dir = os.path.abspath(model_dir)
    top = _load_json(os.path.join(model_dir, 'config.json'))
    if not top or 'class_name' not in top:
        raise FileNotFoundError(f'{model_dir} has no readable config.json with a class_name')
    name, cfg = top['class_name'], top.get('config', {}) or {}
    if name in _SYNTHESIZERS:
        model = 'tacotron2'
    elif name == 'WaveGlow':
        model = 'waveglow'
    else:
        raise ValueError(f'{name} is not a model of the TTS path (Tacotron2, SV2TTSTacotron2, WaveGlow)')
    save_dir = os.path.join(model_dir, 'saving')
    tok = cfg.get('tokenizer')
    tok_file = None
    if model == 'tacotron2':
        # the config stores the path as the reference saw it (relative to ITS working directory): fall back to the layout
        for cand in ([tok] if isinstance(tok, str) else []) + [os.path.join(save_dir, 'tokenizer.json')]:
            if cand and os.path.exists(cand):
                tok_file = cand
                break
    emb_dir = os.path.join(model_dir, 'embeddings')
    spk = int(cfg.get('embedding_dim', 256 if name == 'SV2TTSTacotron2' else 0) or 0) if name == 'SV2TTSTacotron2' else 0
    return {'class_name': name, 'config': cfg, 'model': model, 'checkpoint': find_checkpoint(save_dir),
            'tokenizer_file': tok_file, 'embeddings_dir': emb_dir if os.path.isdir(emb_dir) else None,
            'lang': cfg.get('lang', 'en'), 'speaker_embedding_dim': spk}


def convert_model_dir(model_dir, out=None, cfg=None, force=False):
    """Keras checkpoint of the directory -> TTSW file (default `<checkpoint>.ttsw`, reused while it is newer than the
    checkpoint).  Returns (path, info)."""
    from .weights import save_ttsw
    from .weights_import import from_keras_h5
    info = read_model_dir(model_dir)
    ckpt = info['checkpoint']
    if not ckpt.endswith('.weights.h5'):
        raise ValueError(f'{ckpt}: only `.weights.h5` checkpoints can be read directly (export `.keras` files with '
                         'scripts/export_keras_weights.py where Keras runs)')
    out = out or ckpt[:-len('.weights.h5')] + '.ttsw'
    if force or not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(ckpt):
        if cfg is None:
            cfg = Tacotron2Config(speaker_embedding_dim=info['speaker_embedding_dim']) if info['model'] == 'tacotron2' \
                else WaveGlowConfig()
        tensors = from_keras_h5(ckpt, info['model'], cfg)
        tmp = out + '.tmp'
        save_ttsw(tmp, tensors)
        os.replace(tmp, out)
        logger.info('converted %s -> %s (%d tensors)', ckpt, out, len(tensors))
    return out, info


def load_model(model_dir, device=0, **runtime_kwargs):
    """The wrapper object (`Tacotron2` / `SV2TTSTacotron2` / `WaveGlow`) for a reference model directory, on the HIP runtime."""
    from .runtime import build_runtime
    path, info = convert_model_dir(model_dir)
    rt = build_runtime('hip', path, model=info['model'], device=device,
                       speaker_embedding_dim=info['speaker_embedding_dim'], **runtime_kwargs)
    if info['model'] == 'waveglow':
        from .waveglow import WaveGlow
        return WaveGlow(rt)
    from .tacotron2 import SV2TTSTacotron2, Tacotron2
    if info['class_name'] == 'SV2TTSTacotron2':
        return SV2TTSTacotron2(rt, lang=info['lang'], tokenizer=info['tokenizer_file'],
                               embeddings_dir=info['embeddings_dir'], embedding_dim=info['speaker_embedding_dim'],
                               use_label_embedding=bool(info['config'].get('use_label_embedding', False)),
                               encoder_name=info['config'].get('encoder_name'))
    return Tacotron2(rt, lang=info['lang'], tokenizer=info['tokenizer_file'])
