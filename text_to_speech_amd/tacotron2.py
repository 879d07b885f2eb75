"""Host-side Tacotron2 wrapper and the `tts()` / `stream()` facade.

Restates the host logic of /root/reference/models/tts/tacotron2.py:104-241 (`Tacotron2.infer`: split / clean / encode,
per-part batch-1 call, retry while the frame/token ratio is outside (min_fpt_ratio, max_fpt_ratio) up to `max_trial`
times, slice to `lengths`, vocoder call, concatenation, result dict with keys text, cleaned, splitted, mel, attention,
audio, rate, time), :354-367 (`precompile_for_stream`, `stream`) and models/tts/__init__.py:62-101 (`tts`, `stream`).
Callbacks / file savers / players (utils/callbacks) are out of scope: `callbacks` are plain callables here.
"""
from __future__ import annotations

import logging
import queue as _queue
import time

import numpy as np

from .text import CharTokenizer, split_sentences, split_text

logger = logging.getLogger(__name__)


def _to_numpy(x):
    return x.detach().cpu().numpy() if hasattr(x, 'detach') else np.asarray(x)


class Tacotron2:
    rate = 22050

    def __init__(self, compiled_infer, lang='en', tokenizer=None):
        self.compiled_infer = compiled_infer
        self.tokenizer = tokenizer or CharTokenizer(lang)

    def clean_text(self, text, **kwargs):
        return self.tokenizer.clean_text(text, **kwargs)

    def encode_text(self, text, cleaned=False):
        return self.tokenizer.encode(text, cleaned=cleaned)

    def infer(self, text, *, embeddings=None, callbacks=None, predicted=None, overwrite=False, return_output=True,
              max_length=10., max_text_length=-1, max_trial=5, min_fpt_ratio=2., max_fpt_ratio=10., vocoder=None,
              silence_time=0.15, vocoder_config={}, **kwargs):
        if predicted and not overwrite and text in predicted:
            if callbacks:
                for cb in callbacks:
                    cb(predicted[text])
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
        if predicted is not None and text not in predicted:
            predicted[text] = {k: v for k, v in output.items() if k not in ('mel', 'attention', 'audio')}
        if callbacks:
            for cb in callbacks:
                cb(output)
        if return_output:
            return output
        return {k: v for k, v in output.items() if k not in ('mel', 'attention')}

    def predict(self, inputs, **kwargs):
        """Sequential `for inp in inputs: infer(inp)` (BaseModel.predict with Stream(max_workers=0), base_model.py:676-711)."""
        if isinstance(inputs, (str, dict)):
            inputs = [inputs]
        return [(text, self.infer(text, **kwargs)) for text in inputs]

    def precompile_for_stream(self, **kwargs):
        for m in (64, 128):                                    # tacotron2.py:354-356 (warm-up of both shape buckets)
            self.infer('hello {}'.format(m), max_trial=1, padding_multiple=m, **kwargs)

    def stream(self, stream, *, vocoder, **kwargs):
        """Consumes an iterable or a `queue.Queue` (None ends it); results leave through `callbacks`."""
        self.precompile_for_stream(vocoder=vocoder, **{k: v for k, v in kwargs.items() if k != 'callbacks'})
        kwargs.setdefault('return_output', False)
        for text in _iterate(stream):
            self.infer(text, vocoder=vocoder, **kwargs)


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
    """models.tts.tts (models/tts/__init__.py:62-77): text (or list of texts) -> list of (text, result dict)."""
    if model is None or vocoder is None:
        m, v = get_models(path, device, lang)
        model, vocoder = model or m, vocoder or v
    return model.predict(text, vocoder=vocoder, **kwargs)


def stream(stream, *, lang='en', model=None, vocoder=None, path='synthetic', device=0, **kwargs):
    """models.tts.stream (models/tts/__init__.py:80-101)."""
    if model is None or vocoder is None:
        m, v = get_models(path, device, lang)
        model, vocoder = model or m, vocoder or v
    return model.stream(stream, vocoder=vocoder, **kwargs)
