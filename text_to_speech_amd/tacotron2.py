"""Host-side Tacotron2 wrapper and the `tts()` / `stream()` facade.

Restates the host logic of /root/reference/models/tts/tacotron2.py:104-241 (`Tacotron2.infer`: split / clean / encode,
per-part batch-1 call, retry while the frame/token ratio is outside (min_fpt_ratio, max_fpt_ratio) up to `max_trial`
times, slice to `lengths`, vocoder call, concatenation, result dict with keys text, cleaned, splitted, mel, attention,
audio, rate, time), :276-352 (`get_inference_callbacks`: the `predicted` map / `map.json` cache and the savers),
:354-367 (`precompile_for_stream`, `stream`), models/interfaces/base_model.py:676-711 (`predict`) and
models/tts/__init__.py:62-101 (`tts`, `stream`).  Callbacks live in text_to_speech_amd/callbacks.py; audio players /
notebook displayers are out of scope.
"""
from __future__ import annotations

import logging
import queue as _queue
import time

import os

import numpy as np

from .callbacks import (AudioSaver, Callback, FunctionCallback, JSONSaver, QueueCallback, SpectrogramSaver,
                        apply_callbacks, load_json)
from .text import CharTokenizer, split_sentences, split_text

logger = logging.getLogger(__name__)


def _to_numpy(x):
    return x.detach().cpu().numpy() if hasattr(x, 'detach') else np.asarray(x)


class Tacotron2:
    rate = 22050

    def __init__(self, compiled_infer, lang='en', tokenizer=None, pred_dir=None):
        self.compiled_infer = compiled_infer
        self.tokenizer = tokenizer or CharTokenizer(lang)
        self.pred_dir = pred_dir or os.path.join('pretrained_models', 'tacotron2_hip', 'outputs')

    def clean_text(self, text, **kwargs):
        return self.tokenizer.clean_text(text, **kwargs)

    def encode_text(self, text, cleaned=False):
        return self.tokenizer.encode(text, cleaned=cleaned)

    def infer(self, text, *, embeddings=None, callbacks=None, predicted=None, overwrite=False, return_output=True,
              max_length=10., max_text_length=-1, max_trial=5, min_fpt_ratio=2., max_fpt_ratio=10., vocoder=None,
              silence_time=0.15, vocoder_config={}, **kwargs):
        if isinstance(text, dict):                                   # get_text_from_paragraph (tacotron2.py:369-370)
            text = text['text' if 'text' in text else 'content']
        callbacks = _as_callbacks(callbacks)
        if predicted and not overwrite and text in predicted:        # cached entry: replay it, nothing is re-saved
            if callbacks:
                apply_callbacks(callbacks, predicted[text], {}, save=False)
            return predicted[text]

        if max_text_length == -1:
            splitted = [text]
        elif max_text_length == -2:
            splitted = split_sentences(text)
        else:
            splitted = split_text(text, max_text_length)
        splitted = [self.clean_text(sent, **kwargs) for sent in splitted]
        splitted = [s for s in splitted if any(c.isalnum() for c in s)]
        if not splitted:
            splitted = ['']
        cleaned = '\n\n'.join(splitted) if len(splitted) > 1 else splitted[0]
        encoded = [self.encode_text(t, cleaned=True) for t in splitted]
        splitted = [splitted[i] for i in range(len(splitted)) if len(encoded[i]) > 0]
        encoded = [enc for enc in encoded if len(enc) > 0]

        synth_time, vocoder_time = 0., 0.
        mels, attention_weights, audios = [], [], []
        for inp in encoded:
            t0 = time.time()
            length = len(inp)
            success = False
            inputs = inp[None] if embeddings is None else (inp[None], np.asarray(embeddings)[None])
            for trial in range(max_trial):
                outputs = self.compiled_infer(inputs, max_length=max_length, **kwargs)
                n_frames = int(_to_numpy(outputs.lengths)[0])
                ratio = n_frames / length
                if min_fpt_ratio < ratio < max_fpt_ratio:
                    success = True
                    break
                logger.info('Inference failed (lengths : %s, frame/token ratio : %.2f) !', outputs.lengths, ratio)
            synth_time += time.time() - t0
            if not success:
                logger.warning('Inference failed too much time ! Result is probably not perfect')
            mels.append(outputs.mel[0, :n_frames])
            attention_weights.append(outputs.attention_weights[0, :n_frames])
            if vocoder is not None:
                t1 = time.time()
                if n_frames > 0:
                    audio = vocoder(mels[-1], **{**kwargs, **vocoder_config})
                    if len(audio.shape) == 2:
                        audio = audio[0]
                    audios.append(_to_numpy(audio))
                vocoder_time += time.time() - t1

        audio_infos = {}
        if vocoder is not None:
            if len(audios) > 0:
                audios = audios[0] if len(audios) == 1 else np.concatenate(audios, axis=0)
                audio_infos = {'audio': audios, 'rate': self.rate, 'time': len(audios) / self.rate}
                logger.info('%.2f s generated in %.3f s (%.3f synthesizer + %.3f vocoder)', audio_infos['time'],
                            synth_time + vocoder_time, synth_time, vocoder_time)
            else:
                audio_infos = {'audio': np.zeros((int(silence_time * self.rate),), dtype='float32'),
                               'rate': self.rate, 'time': silence_time}
        output = {'text': text, 'cleaned': cleaned, 'splitted': splitted, 'mel': mels,
                  'attention': attention_weights, **audio_infos}
        if callbacks:
            if predicted is None:
                predicted = {}
            if text not in predicted:
                predicted[text] = {k: v for k, v in output.items() if k not in ('mel', 'attention', 'audio')}
            apply_callbacks(callbacks, predicted[text], output, save=True)
        if return_output:
            return output
        if vocoder is None or 'audio' in (predicted or {}).get(text, {}):
            return (predicted or {}).get(text, {})
        return {k: v for k, v in output.items() if k not in ('mel', 'attention')}

    def get_inference_callbacks(self, *, vocoder=None, save=None, save_mel=None, save_audio=None, directory=None,
                                mel_dir=None, audio_dir=None, mel_filename='mel-{}.npy',
                                audio_filename='audio-{}.wav', post_processing=None, **_):
        """(predicted, callbacks) with the reference's flag resolution (tacotron2.py:276-352): results are saved when a
        `directory` is given or there is no vocoder; mels only without a vocoder; `map.json` in `directory` is both the
        cache that `infer` consults and the index the JSON saver rewrites."""
        if vocoder is None:
            save_audio = False
        elif save_audio is None:
            save_audio = save is not False
        if save is None:
            save = bool(directory) or vocoder is None
        if save_mel is None:
            save_mel = save and vocoder is None
        save = bool(save_mel or save_audio)                        # (sic: with a vocoder, audio is saved unless save=False)
        if vocoder is not None and save:
            save_audio = True
        predicted, callbacks = {}, []
        if save:
            if directory is None:
                directory = self.pred_dir
            os.makedirs(directory, exist_ok=True)
            map_file = os.path.join(directory, 'map.json')
            predicted = load_json(map_file, {})
            if save_mel:
                callbacks.append(SpectrogramSaver(file_format=os.path.join(mel_dir or os.path.join(directory, 'mels'),
                                                                           mel_filename)))
            if save_audio:
                callbacks.append(AudioSaver(file_format=os.path.join(audio_dir or os.path.join(directory, 'audios'),
                                                                     audio_filename)))
            callbacks.append(JSONSaver(data=predicted, filename=map_file, primary_key='text'))
        if post_processing is not None:
            for fn in (post_processing if isinstance(post_processing, list) else [post_processing]):
                if callable(fn):
                    callbacks.append(FunctionCallback(fn))
                elif hasattr(fn, 'put'):
                    callbacks.append(QueueCallback(fn))
        return predicted, callbacks

    _callback_kwargs = ('save', 'save_mel', 'save_audio', 'directory', 'mel_dir', 'audio_dir', 'mel_filename',
                        'audio_filename', 'post_processing')

    def predict(self, inputs, *, predicted=None, callbacks=None, return_results=True, return_output=None, **kwargs):
        """BaseModel.predict (base_model.py:676-711): builds the callbacks unless the caller brings its own `predicted`
        map, then runs `infer` sequentially; returns the result dicts (or the `predicted` entries when a JSON saver is
        active and `return_output` was not forced)."""
        if isinstance(inputs, (str, dict)):
            inputs = [inputs]
        join_callbacks = predicted is None
        if predicted is None:
            predicted, built = self.get_inference_callbacks(**kwargs)
            callbacks = built + _as_callbacks(callbacks)
        else:
            callbacks = _as_callbacks(callbacks)
        if return_output is None:
            return_output = not any(isinstance(cb, JSONSaver) for cb in callbacks)
        kwargs = {k: v for k, v in kwargs.items() if k not in self._callback_kwargs}
        results = []
        for inp in inputs:
            text = inp['text' if 'text' in inp else 'content'] if isinstance(inp, dict) else inp
            output = self.infer(inp, predicted=predicted, callbacks=callbacks, return_output=return_output, **kwargs)
            if return_results:
                results.append(output if return_output else predicted[text])
        if join_callbacks:
            for cb in callbacks:
                cb.join()
        return results

    def precompile_for_stream(self, **kwargs):
        for m in (64, 128):                                    # tacotron2.py:354-356 (warm-up of both shape buckets)
            self.infer('hello {}'.format(m), max_trial=1, padding_multiple=m, **kwargs)

    def stream(self, stream, *, vocoder, **kwargs):
        """`predict(return_output=False, return_results=False)` over an iterable or a `queue.Queue` (None ends it);
        results leave through the callbacks (tacotron2.py:363-367, base_model.py:713)."""
        self.precompile_for_stream(vocoder=vocoder, **{k: v for k, v in kwargs.items()
                                                       if k not in self._callback_kwargs + ('callbacks', 'predicted')})
        kwargs.setdefault('return_output', False)
        kwargs.setdefault('return_results', False)
        return self.predict(_iterate(stream), vocoder=vocoder, **kwargs)


def _as_callbacks(callbacks):
    """Accepts Callback instances, plain callables (called with the merged entry + result as keyword arguments, like the
    reference's `post_processing` functions) and queues."""
    if not callbacks:
        return []
    out = []
    for cb in callbacks:
        if isinstance(cb, Callback):
            out.append(cb)
        elif hasattr(cb, 'put'):
            out.append(QueueCallback(cb))
        elif callable(cb):
            out.append(FunctionCallback(cb))
        else:
            raise TypeError(f'unsupported callback {cb!r}')
    return out


def _iterate(stream):
    if isinstance(stream, _queue.Queue):
        while True:
            item = stream.get()
            if item is None:
                return
            yield item
    else:
        yield from stream


_models = {}


def get_models(path='synthetic', device=0, lang='en', **kwargs):
    """(Tacotron2, WaveGlow) pair sharing one engine -- the analogue of models/tts/__init__.py:get_models."""
    from .runtime import build_runtime
    from .waveglow import WaveGlow
    key = (path, device, lang)
    if key not in _models:
        synth = build_runtime('hip', path, model='tacotron2', device=device, **kwargs)
        voc = build_runtime('hip', path, model='waveglow', engine=synth.engine, device=device)
        _models[key] = (Tacotron2(synth, lang=lang), WaveGlow(voc))
    return _models[key]


def tts(text, *, lang='en', model=None, vocoder=None, path='synthetic', device=0, **kwargs):
    """models.tts.tts (models/tts/__init__.py:62-77): one text -> its result dict; a list of texts -> list of dicts."""
    if model is None or vocoder is None:
        m, v = get_models(path, device, lang)
        model, vocoder = model or m, vocoder or v
    res = model.predict(text, vocoder=vocoder, **kwargs)
    return res[0] if isinstance(text, (str, dict)) else res


def stream(stream, *, lang='en', model=None, vocoder=None, path='synthetic', device=0, **kwargs):
    """models.tts.stream (models/tts/__init__.py:80-101)."""
    if model is None or vocoder is None:
        m, v = get_models(path, device, lang)
        model, vocoder = model or m, vocoder or v
    return model.stream(stream, vocoder=vocoder, **kwargs)
