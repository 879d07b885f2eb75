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
    found = sorted(glob.glob(os.path.join(save_dir, '*.weights.h5')) + glob.glob(os.path.join(save_dir, '*.keras')),
                   key=os.path.getmtime)
    if found:
        return found[-1]
    raise FileNotFoundError(f'no Keras checkpoint (*.weights.h5 or *.keras) in {save_dir}')


# What the engine implements of the reference's hyper-parameters (architectures/tacotron2_arch.py:59-135,
# layers/location_sensitive_attention.py:17-24, waveglow_arch.py:164-178, utils/audio/stft.py:27-60,286-305).  A checkpoint
# trained with anything else has tensors of other shapes or another graph: it is refused here, by name, instead of being
# caught (or not) where a tensor shape happens to differ.
_TACOTRON2_REQUIRED = {
    'n_mel_channels': 80, 'n_frames_per_step': 1, 'pred_stop_on_mel': False, 'with_logits': True,
    'attention_rnn_dim': 1024, 'decoder_rnn_dim': 1024, 'decoder_n_lstm': 1,
    'prenet_sizes': [256, 256], 'prenet_use_bias': False, 'prenet_activation': 'relu', 'prenet_concat_speaker': False,
    'lsa_attention_dim': 128, 'lsa_attention_filters': 32, 'lsa_attention_kernel_size': 31,
    'lsa_probability_function': 'softmax', 'lsa_cumulative': True,
    'encoder_embedding_dim': 512, 'encoder_n_conv': 3, 'encoder_kernel_size': 5, 'encoder_bnorm': 'after',
    'encoder_activation': 'relu', 'encoder_concat_mode': 'concat', 'encoder_linear_projection': False, 'encoder_pad_token': 0,
    'postnet_n_conv': 5, 'postnet_filters': 512, 'postnet_kernel_size': 5, 'postnet_bnorm': 'after',
    'postnet_activation': 'tanh', 'postnet_final_activation': None, 'postnet_linear_projection': False,
    'speaker_concat_pos': 'end',
}
_TACOTRON2_EPS = ('encoder_epsilon', 'postnet_epsilon')          # batch-norm epsilon 1e-5 is folded into the conv weights
_WAVEGLOW_REQUIRED = {'n_mel_channels': 80, 'n_flows': 12, 'n_group': 8, 'n_early_every': 4, 'n_early_size': 2,
                      'n_layers': 8, 'n_channels': 512, 'kernel_size': 3}
_MEL_FN_REQUIRED = {'class_name': 'TacotronSTFT', 'sampling_rate': 22050, 'n_mel_channels': 80, 'filter_length': 1024,
                    'hop_length': 256, 'win_length': 1024, 'mel_fmin': 0.0, 'mel_fmax': 8000.0}


def _same(a, b):
    if isinstance(b, float) or isinstance(a, float):
        try:
            return abs(float(a) - float(b)) <= 1e-9 * max(1.0, abs(float(b)))
        except (TypeError, ValueError):
            return False
    if isinstance(b, list):
        return isinstance(a, (list, tuple)) and list(a) == b
    return a == b


def check_hparams(model, hparams, where='config_models.json'):
    """Raises ValueError naming every hyper-parameter of `hparams` (the architecture's `get_config()`) the engine does not
    implement; absent keys mean the reference's defaults, which are the supported values."""
    required = _TACOTRON2_REQUIRED if model == 'tacotron2' else _WAVEGLOW_REQUIRED
    bad = [f'{k} = {hparams[k]!r} (supported: {v!r})' for k, v in required.items() if k in hparams and not _same(hparams[k], v)]
    if model == 'tacotron2':
        bad += [f'{k} = {hparams[k]!r} (supported: 1e-05)' for k in _TACOTRON2_EPS if k in hparams and not _same(hparams[k], 1e-5)]
        n_spk = hparams.get('encoder_n_speaker', 1)
        if n_spk not in (None, 1):
            bad.append(f'encoder_n_speaker = {n_spk!r} (supported: 1; speaker identity comes in as an embedding)')
    if bad:
        raise ValueError(f'{where}: this checkpoint was built with hyper-parameters the HIP engine does not implement:\n  '
                         + '\n  '.join(bad))


def read_hparams(save_dir, model):
    """The architecture's hyper-parameters from `saving/config_models.json` (base_model.py:739-749: `{'model':
    keras.saving.serialize_keras_object(self.model)}`, whose 'config' is `Tacotron2.get_config()` = the HParams,
    tacotron2_arch.py:927-928 / `WaveGlow.get_config()`, waveglow_arch.py:312-324); {} when the file is absent."""
    top = _load_json(os.path.join(save_dir, 'config_models.json'))
    if not top:
        return {}
    node = top.get('model', top)
    hp = node.get('config', {}) if isinstance(node, dict) else {}
    hp = hp if isinstance(hp, dict) else {}
    check_hparams(model, hp, os.path.join(save_dir, 'config_models.json'))
    return hp


def check_mel_fn(path):
    """`saving/mel_fn.json` (base_audio_model.py:99,208-217: `MelSTFT.get_config()`): the analysis the model was trained on
    must be the TacotronSTFT the engine's mel-STFT implements."""
    cfg = _load_json(path)
    if not cfg:
        return None
    bad = [f'{k} = {cfg[k]!r} (supported: {v!r})' for k, v in _MEL_FN_REQUIRED.items() if k in cfg and not _same(cfg[k], v)]
    if cfg.get('pre_emph') not in (None, 0, 0.0, False):
        bad.append(f"pre_emph = {cfg['pre_emph']!r} (supported: 0)")
    if cfg.get('window', 'hann') != 'hann':
        bad.append(f"window = {cfg['window']!r} (supported: 'hann')")
    if bad:
        raise ValueError(f'{path}: mel front-end the HIP engine does not implement:\n  ' + '\n  '.join(bad))
    return cfg


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
    hparams = read_hparams(save_dir, model)
    mel_fn = check_mel_fn(os.path.join(save_dir, 'mel_fn.json'))
    vocab_size = None
    if model == 'tacotron2':
        # the embedding table has one row per symbol of the model's own tokenizer: the architecture's hyper-parameters say how
        # many (`vocab_size`), else the wrapper's config, else the tokenizer file (vocabulary + the special tokens in use)
        for src in (hparams.get('vocab_size'), cfg.get('vocab_size')):
            if isinstance(src, int) and src > 1:
                vocab_size = src
                break
        if tok_file is not None:
            from .text import CharTokenizer
            tokenizer = CharTokenizer.load_from_file(tok_file)
            if tokenizer.blank_token_idx != 0:
                raise ValueError(f'{tok_file}: the padding token has id {tokenizer.blank_token_idx}; the engine masks on token '
                                 f'!= 0 (encoder pad_token = 0, tacotron2_arch.py:62)')
            if vocab_size is None:
                vocab_size = tokenizer.vocab_size
            elif tokenizer.vocab_size > vocab_size:
                raise ValueError(f'{tok_file}: {tokenizer.vocab_size} symbols but the model has {vocab_size} embedding rows')
        if hparams.get('encoder_speaker_embedding_dim') not in (None, 0) and spk and int(hparams['encoder_speaker_embedding_dim']) != spk:
            raise ValueError(f"speaker embedding width: config.json says {spk}, config_models.json says "
                             f"{hparams['encoder_speaker_embedding_dim']}")
    return {'class_name': name, 'config': cfg, 'model': model, 'checkpoint': find_checkpoint(save_dir),
            'tokenizer_file': tok_file, 'embeddings_dir': emb_dir if os.path.isdir(emb_dir) else None,
            'lang': cfg.get('lang', 'en'), 'speaker_embedding_dim': spk, 'hparams': hparams, 'mel_fn': mel_fn,
            'vocab_size': vocab_size if vocab_size is not None else (148 if model == 'tacotron2' else None)}


def convert_model_dir(model_dir, out=None, cfg=None, force=False):
    """Keras checkpoint of the directory (`.weights.h5`, or a `.keras` archive: its model.weights.h5 member) -> TTSW file
    (default `<checkpoint>.ttsw`, reused while it is newer than the checkpoint).  Returns (path, info)."""
    from .weights import save_ttsw
    from .weights_import import from_keras_file
    info = read_model_dir(model_dir)
    ckpt = info['checkpoint']
    ext = next((x for x in ('.weights.h5', '.keras') if ckpt.endswith(x)), None)      # what CheckpointManager.load accepts (:196)
    if ext is None:
        raise ValueError(f'{ckpt}: only `.weights.h5` checkpoints and `.keras` archives can be read (TF `.index` checkpoints: '
                         'export them with scripts/export_keras_weights.py where TensorFlow runs)')
    out = out or ckpt[:-len(ext)] + '.ttsw'
    if force or not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(ckpt):
        if cfg is None:
            cfg = Tacotron2Config(speaker_embedding_dim=info['speaker_embedding_dim'], vocab_size=info['vocab_size']) \
                if info['model'] == 'tacotron2' else WaveGlowConfig()
        tensors = from_keras_file(ckpt, info['model'], cfg)
        # several ranks / processes may open a fresh directory at once: each writes its own temporary file and publishes it
        # atomically (a shared name let one writer truncate the file another had just finished)
        import tempfile
        fd, tmp = tempfile.mkstemp(prefix=os.path.basename(out) + '.', suffix='.tmp', dir=os.path.dirname(out) or '.')
        os.close(fd)
        try:
            save_ttsw(tmp, tensors)
            os.replace(tmp, out)
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)
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
