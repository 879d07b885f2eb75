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

from .callbacks import (AudioSaver, default_audio_format, Callback, FunctionCallback, JSONSaver, QueueCallback, SpectrogramSaver,
                        apply_callbacks, load_json)
from .text import CharTokenizer, split_sentences, split_text

logger = logging.getLogger(__name__)


def _to_numpy(x):
    return x.detach().cpu().numpy() if hasattr(x, 'detach') else np.asarray(x)


class Tacotron2:
    rate = 22050

    def __init__(self, compiled_infer, lang='en', tokenizer=None, pred_dir=None):
        self.compiled_infer = compiled_infer
        if isinstance(tokenizer, str):                           # a shipped model's `saving/tokenizer.json`
            tokenizer = CharTokenizer.load_from_file(tokenizer, lang=lang)
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
        part = self._synthesize(text, embeddings=embeddings, max_length=max_length, max_text_length=max_text_length,
                                max_trial=max_trial, min_fpt_ratio=min_fpt_ratio, max_fpt_ratio=max_fpt_ratio, **kwargs)
        return self._vocode_and_finish(part, callbacks=callbacks, predicted=predicted, return_output=return_output,
                                       vocoder=vocoder, silence_time=silence_time, vocoder_config=vocoder_config,
                                       **kwargs)

    # `infer` = `_synthesize` (text -> mels; the autoregressive, latency-bound half) followed by `_vocode_and_finish`
    # (mels -> audio, callbacks; the throughput-bound half).  They are separate so that `stream(overlap=True)` can run
    # the first half of sentence n + 1 while the second half of sentence n is still on the GPU.
    def _synthesize(self, text, *, embeddings=None, max_length=10., max_text_length=-1, max_trial=5, min_fpt_ratio=2.,
                    max_fpt_ratio=10., **kwargs):
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

        t0 = time.time()
        mels, attention_weights = [], []
        for inp in encoded:
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
            if not success:
                logger.warning('Inference failed too much time ! Result is probably not perfect')
            mels.append(outputs.mel[0, :n_frames])
            attention_weights.append(outputs.attention_weights[0, :n_frames])
        return {'text': text, 'cleaned': cleaned, 'splitted': splitted, 'mel': mels, 'attention': attention_weights,
                'synth_time': time.time() - t0}

    def _vocode_and_finish(self, part, *, callbacks=None, predicted=None, return_output=True, vocoder=None,
                           silence_time=0.15, vocoder_config={}, **kwargs):
        text, synth_time = part['text'], part.get('synth_time', 0.)
        audio_infos = {}
        if vocoder is not None:
            t1 = time.time()
            audios = []
            for mel in part['mel']:
                if mel.shape[0] > 0:
                    audio = vocoder(mel, **{**kwargs, **vocoder_config})
                    if len(audio.shape) == 2:
                        audio = audio[0]
                    audios.append(_to_numpy(audio))
            vocoder_time = time.time() - t1
            if len(audios) > 0:
                audios = audios[0] if len(audios) == 1 else np.concatenate(audios, axis=0)
                audio_infos = {'audio': audios, 'rate': self.rate, 'time': len(audios) / self.rate}
                logger.info('%.2f s generated in %.3f s (%.3f synthesizer + %.3f vocoder)', audio_infos['time'],
                            synth_time + vocoder_time, synth_time, vocoder_time)
            else:
                audio_infos = {'audio': np.zeros((int(silence_time * self.rate),), dtype='float32'),
                               'rate': self.rate, 'time': silence_time}
        output = {k: part[k] for k in ('text', 'cleaned', 'splitted', 'mel', 'attention')}
        output.update(audio_infos)
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
                                audio_filename=None, post_processing=None, **_):
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
                                                                     audio_filename or default_audio_format())))
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

    def predict(self, inputs, *, predicted=None, callbacks=None, return_results=True, return_output=None,
                overlap=False, **kwargs):
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
        if overlap and kwargs.get('vocoder') is not None:
            outputs = self._infer_overlapped(inputs, predicted=predicted, callbacks=callbacks,
                                             return_output=return_output, **kwargs)
        else:
            outputs = ((inp, self.infer(inp, predicted=predicted, callbacks=callbacks, return_output=return_output,
                                        **kwargs)) for inp in inputs)
        for inp, output in outputs:
            text = inp['text' if 'text' in inp else 'content'] if isinstance(inp, dict) else inp
            if return_results:
                results.append(output if return_output else predicted[text])
        if join_callbacks:
            for cb in callbacks:
                cb.join()
        return results

    _synth_kwargs = ('embeddings', 'max_length', 'max_text_length', 'max_trial', 'min_fpt_ratio', 'max_fpt_ratio')

    def _infer_overlapped(self, inputs, *, predicted, callbacks, return_output, overwrite=False, **kwargs):
        """Sentence-level software pipeline: a worker thread runs `_synthesize` for the next input while this thread
        vocodes the previous one.  The synthesizer and the vocoder must sit on different engine handles (two HIP streams;
        calls on one handle are serialised) -- `get_models(..., overlap=True)` builds such a pair.  Measured on MI355X
        (scripts/overlap_probe.py, 600-frame sentences): 39.2 -> 34.6 ms per sentence with the fp16 vocoder, 87.8 -> 78.5 ms
        in fp32; the decoder's launches only get CU slots as WN GEMM blocks retire, so it runs 2.4x / 5.6x slower while a
        vocoder call is in flight, but that time was idle before.  Results keep the input order."""
        import threading
        synth_kw = {k: kwargs.pop(k) for k in list(kwargs) if k in self._synth_kwargs}
        voc_kw = dict(kwargs)
        q = _queue.Queue(maxsize=2)
        DONE = object()

        def producer():
            try:
                for inp in inputs:
                    text = inp['text' if 'text' in inp else 'content'] if isinstance(inp, dict) else inp
                    if predicted and not overwrite and text in predicted:
                        q.put((inp, text, None, None))
                        continue
                    extra = {k: v for k, v in voc_kw.items() if k not in ('vocoder', 'silence_time', 'vocoder_config')}
                    q.put((inp, text, self._synthesize(text, **synth_kw, **extra), None))
            except BaseException as exc:                              # noqa: BLE001 -- re-raised in the consumer
                q.put((None, None, None, exc))
            finally:
                q.put(DONE)

        th = threading.Thread(target=producer, name='tacotron2-synth', daemon=True)
        th.start()
        try:
            while True:
                item = q.get()
                if item is DONE:
                    break
                inp, text, part, exc = item
                if exc is not None:
                    raise exc
                if part is None:                                      # cache hit
                    if callbacks:
                        apply_callbacks(callbacks, predicted[text], {}, save=False)
                    yield inp, predicted[text]
                else:
                    yield inp, self._vocode_and_finish(part, callbacks=callbacks, predicted=predicted,
                                                       return_output=return_output, **voc_kw)
        finally:
            while th.is_alive():                                      # drain so that the producer can finish
                try:
                    q.get(timeout=0.05)
                except _queue.Empty:
                    pass
            th.join()

    def precompile_for_stream(self, **kwargs):
        for m in (64, 128):                                    # tacotron2.py:354-356 (warm-up of both shape buckets)
            self.infer('hello {}'.format(m), max_trial=1, padding_multiple=m, **kwargs)

    def stream(self, stream, *, vocoder, **kwargs):
        """`predict(return_output=False, return_results=False)` over an iterable or a `queue.Queue` (None ends it);
        results leave through the callbacks (tacotron2.py:363-367, base_model.py:713)."""
        self.precompile_for_stream(vocoder=vocoder, **{k: v for k, v in kwargs.items()
                                                       if k not in self._callback_kwargs + ('callbacks', 'predicted', 'overlap')})
        kwargs.setdefault('return_output', False)
        kwargs.setdefault('return_results', False)
        return self.predict(_iterate(stream), vocoder=vocoder, **kwargs)


# ---- multi-speaker wrapper (models/tts/sv2tts_tacotron2.py:18-128, utils/embeddings.py:249-286) --------------------------
def select_embedding(embeddings, mode='random', **filters):
    """One speaker embedding (1-D) out of a collection: a 2-D array, a 1-D array (a collection of one) or a pandas
    DataFrame with an 'embedding' column (then `filters` on other columns narrow the choice; no match = no filter, with a
    warning).  mode: an int (row), 'mean' / 'avg' / 'average', 'random' (Python's `random`, like the reference) or a
    callable taking the [n, E] array."""
    import random
    if hasattr(embeddings, 'columns'):
        rows = embeddings
        used = {k: v for k, v in filters.items() if k in embeddings.columns}
        if used:
            keep = np.ones(len(embeddings), dtype=bool)
            for col, want in used.items():
                vals = embeddings[col]
                keep &= np.asarray(vals.isin(list(want)) if isinstance(want, (list, tuple, set)) else vals == want)
            if keep.any():
                rows = embeddings[keep]
            else:
                logger.warning('No embedding respect filters %s', filters)
        pool = np.stack([np.asarray(e, dtype=np.float32) for e in rows['embedding'].values])
    else:
        pool = _to_numpy(embeddings)
        if pool.ndim == 1:
            pool = pool[None]
    if isinstance(mode, (int, np.integer)) and not isinstance(mode, bool):
        return pool[mode]
    if callable(mode):
        return mode(pool)
    if mode in ('mean', 'avg', 'average'):
        return pool.mean(axis=0)
    if mode == 'random':
        return pool[random.randrange(len(pool))]
    raise ValueError("Unknown embedding selection mode !\n  Accepted : {}\n  Got : {}".format(
        "(int, callable, 'mean', 'random')", mode))


class SV2TTSTacotron2(Tacotron2):
    """Tacotron2 conditioned on a speaker embedding (encoder output 512 + `embedding_dim`).  `infer(text, embeddings=...)`
    takes the vector itself, or a selector resolved against the model's collection (`self.embeddings`): None -> the default
    mode ('mean' when `use_label_embedding`, else 'random'), an int -> that row, a str -> that mode, a dict ->
    `select_embedding(**dict)`; the reference's default is `embeddings=0`, the first row."""

    def __init__(self, compiled_infer, lang='fr', *, embeddings=None, embeddings_dir=None, embedding_dim=256,
                 use_label_embedding=False, encoder_name=None, **kwargs):
        super().__init__(compiled_infer, lang=lang, **kwargs)
        # `embeddings`: the collection itself (matrix / DataFrame) or a file of one (csv / npy / pkl / the reference's h5);
        # `embeddings_dir`: the model's `<name>/embeddings` directory, searched like sv2tts_tacotron2.py:53-67 (the only
        # file in it, else `embeddings.<ext>`)
        if embeddings is None and embeddings_dir is not None and os.path.isdir(embeddings_dir):
            found = sorted(os.listdir(embeddings_dir))
            embeddings = os.path.join(embeddings_dir, found[0] if len(found) == 1 else 'embeddings')
        if isinstance(embeddings, str):
            from .embeddings import load_embeddings
            embeddings = load_embeddings(embeddings)
        self.embeddings = embeddings
        self.embedding_dim = embedding_dim
        self.use_label_embedding = use_label_embedding
        self.encoder_name = encoder_name

    def select_embedding(self, embeddings=None, mode=None):
        if not hasattr(embeddings, 'shape'):                     # a selector, not data
            if mode is None:
                mode = embeddings
            embeddings = self.embeddings
        if embeddings is None:
            raise ValueError('this model has no speaker embeddings: pass `embeddings=<vector>` or set `model.embeddings`')
        if mode is None:
            mode = {'mode': 'mean' if self.use_label_embedding else 'random'}
        elif not isinstance(mode, dict):
            mode = {'mode': mode}
        vec = np.asarray(select_embedding(embeddings, **mode), dtype=np.float32)
        if vec.shape != (self.embedding_dim,):
            raise ValueError(f'speaker embedding must have shape ({self.embedding_dim},), got {vec.shape}')
        return vec

    def infer(self, text, *, embeddings=0, **kwargs):
        if embeddings is None or isinstance(embeddings, (int, str, dict)):
            embeddings = self.select_embedding(embeddings)
        return super().infer(text, embeddings=embeddings, **kwargs)

    def _infer_overlapped(self, inputs, *, embeddings=0, **kwargs):
        if embeddings is None or isinstance(embeddings, (int, str, dict)):
            embeddings = self.select_embedding(embeddings)
        return super()._infer_overlapped(inputs, embeddings=embeddings, **kwargs)


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


def get_models(path='synthetic', device=0, lang='en', overlap=False, **kwargs):
    """(Tacotron2, WaveGlow) pair -- the analogue of models/tts/__init__.py:get_models.  By default both share one engine
    handle; `overlap=True` gives the vocoder its own handle (second HIP stream, second weight copy) so that
    `stream(..., overlap=True)` can run the two models concurrently."""
    from .runtime import HipRuntime, build_runtime
    from .waveglow import WaveGlow
    spk_dim = int(kwargs.get('speaker_embedding_dim', 0) or 0)
    embeddings = kwargs.pop('embeddings', None)
    key = (path, device, lang, bool(overlap), spk_dim)
    if key not in _models:
        synth = build_runtime('hip', path, model='tacotron2', device=device, **kwargs)
        if overlap:
            eng = HipRuntime.load_engine(path, device=device, **kwargs)
            voc = build_runtime('hip', path, model='waveglow', engine=eng, device=device, **kwargs)
        else:
            voc = build_runtime('hip', path, model='waveglow', engine=synth.engine, device=device, **kwargs)
        model = (SV2TTSTacotron2(synth, lang=lang, embedding_dim=spk_dim, embeddings=embeddings) if spk_dim
                 else Tacotron2(synth, lang=lang))
        _models[key] = (model, WaveGlow(voc))
    elif embeddings is not None and spk_dim:
        _models[key][0].embeddings = embeddings
    return _models[key]


def tts(text, *, lang='en', model=None, vocoder=None, path='synthetic', device=0, **kwargs):
    """models.tts.tts (models/tts/__init__.py:62-77): one text -> its result dict; a list of texts -> list of dicts."""
    if model is None or vocoder is None:
        m, v = get_models(path, device, lang)
        model, vocoder = model or m, vocoder or v
    res = model.predict(text, vocoder=vocoder, **kwargs)
    return res[0] if isinstance(text, (str, dict)) else res


def stream(stream, *, lang='en', model=None, vocoder=None, path='synthetic', device=0, **kwargs):
    """models.tts.stream (models/tts/__init__.py:80-101); `overlap=True` pipelines Tacotron2(n + 1) with WaveGlow(n)."""
    if model is None or vocoder is None:
        m, v = get_models(path, device, lang, overlap=bool(kwargs.get('overlap', False)))
        model, vocoder = model or m, vocoder or v
    return model.stream(stream, vocoder=vocoder, **kwargs)
