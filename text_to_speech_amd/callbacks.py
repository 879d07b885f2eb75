"""Inference callbacks: result savers (`audio-{i}.wav`, `mel-{i}.npy`, `map.json`), function / queue hand-off.

Host-side counterpart of /root/reference/utils/callbacks, with the behaviour
`Tacotron2.infer` / `predict` rely on:
  * a callback is called as `cb(infos, output)`: `infos` is the JSON-able entry of the `predicted` map (mutated in place:
    savers record the file they wrote under their key), `output` the full result dict (callback.py:32-44);
  * file savers number their files by counting what already matches the pattern on first use, then increment
    (file_saver.py:63-76); an entry that already holds a file name is not written again (:104-110);
  * `JSONSaver` stores `infos` under `infos[primary_key]` in the shared map and rewrites the json file (:167-193);
  * `apply_callbacks(..., save=False)` skips the file savers (used when a cached entry is replayed, __init__.py:31-47)
    and never lets a failing callback break inference: the error is logged.
Audio files: `write_audio` prepares samples like the reference (mean removed, peak-normalised 16-bit) and writes `.wav`
itself and every other extension through the `ffmpeg` executable, like the reference's pydub writer; the default file name
is the reference's `audio-{}.mp3` when ffmpeg is on PATH and `audio-{}.wav` otherwise (this image has no encoder).
Players / displayers (displayer.py) are out of scope.
"""
from __future__ import annotations

import glob
import json
import logging
import os
import re

import numpy as np

logger = logging.getLogger(__name__)

_INDEX_RE = re.compile(r'\{i?(:\d{2}d)?\}')


def _to_numpy(x):
    return x.detach().cpu().numpy() if hasattr(x, 'detach') else np.asarray(x)


def to_json(x):
    """JSON-able copy of an `infos` entry (numpy scalars / arrays -> python)."""
    if isinstance(x, dict):
        return {str(k): to_json(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [to_json(v) for v in x]
    if isinstance(x, np.generic):
        return x.item()
    if isinstance(x, np.ndarray) or hasattr(x, 'detach'):
        return _to_numpy(x).tolist()
    return x


def write_wav(filename, audio, rate=22050):
    """float waveform in [-1, 1] (clipped) -> 16-bit PCM wav."""
    from scipy.io import wavfile
    a = np.clip(_to_numpy(audio).astype(np.float32).reshape(-1), -1.0, 1.0)
    wavfile.write(filename, int(rate), np.round(a * 32767.0).astype(np.int16))


def to_pcm16(audio, normalize=True):
    """float waveform -> int16 samples the way the reference's `write_audio` prepares them (utils/audio/audio_io.py:359-361,
    audio_processing.py:51-62): mean removed, peak scaled to 32767, truncated; `normalize=False` keeps the scale ([-1, 1]
    clipped, rounded)."""
    a = _to_numpy(audio).astype(np.float32).reshape(-1)
    if not normalize:
        return np.round(np.clip(a, -1.0, 1.0) * 32767.0).astype(np.int16)
    if a.size == 0:
        return a.astype(np.int16)
    a = a - np.mean(a)
    peak = float(np.max(np.abs(a)))
    if peak <= 1e-9:
        return a.astype(np.int16)
    return (a * (32767 / peak)).astype(np.int16)


def find_ffmpeg():
    import shutil
    return shutil.which('ffmpeg')


def write_audio(filename, audio, rate=22050, normalize=True):
    """Writes `audio` in the format the extension names (audio_io.py:347-364): `.wav` with scipy, anything else (mp3, ogg,
    flac, m4a, ...) through the `ffmpeg` executable fed with raw 16-bit PCM -- what the reference's pydub writer does
    (:371-380).  Without ffmpeg on PATH a non-wav extension is an error, not a silent change of format."""
    ext = os.path.splitext(filename)[1].lower()
    pcm = to_pcm16(audio, normalize)
    if ext == '.wav':
        from scipy.io import wavfile
        wavfile.write(filename, int(rate), pcm)
        return filename
    exe = find_ffmpeg()
    if exe is None:
        raise ValueError(f"cannot write {filename!r}: encoding '{ext}' needs the ffmpeg executable on PATH "
                         "(use a .wav file name, or pass save_fn=...)")
    import subprocess
    cmd = [exe, '-y', '-loglevel', 'error', '-f', 's16le', '-ar', str(int(rate)), '-ac', '1', '-i', 'pipe:0', filename]
    done = subprocess.run(cmd, input=pcm.tobytes(), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    if done.returncode != 0:
        raise RuntimeError(f"ffmpeg failed writing {filename!r}: {done.stderr.decode(errors='replace')[-500:]}")
    return filename


def default_audio_format():
    """`audio-{}.mp3` like the reference (models/tts/tacotron2.py:308) when an encoder is available, else `audio-{}.wav`."""
    return 'audio-{}.mp3' if find_ffmpeg() else 'audio-{}.wav'


class Callback:
    def __init__(self, name=None, cond=None, initializer=None, **_):
        self.name = name or self.__class__.__name__
        self.cond = cond
        self.initializer = initializer
        self.built = False

    def __repr__(self):
        return '<{}>'.format(self.__class__.__name__)

    def build(self):
        self.built = True

    def __call__(self, infos, output, **kwargs):
        if self.cond is not None and not self.cond(**output):
            return None
        if self.initializer:
            for k, fn in self.initializer.items():
                if k not in output:
                    output[k] = fn(**output)
        if not self.built:
            self.build()
        return self.apply(infos=infos, output=output, **kwargs)

    def apply(self, infos, output, **kwargs):
        raise NotImplementedError()

    def join(self):
        pass


class NumberedPath:
    """File-name pattern with an optional running number: `audio-{}.wav`, `mel-{i}.npy`, `part-{:02d}.wav`, and named
    fields taken from the result (`{basename}` = the stem of the entry's `filename`).  The first number is the count of
    files already matching the pattern on disk (so numbering continues across runs), unless the result itself carries
    the number under `index_key`."""

    def __init__(self, pattern, first=-1, index_key=None):
        self.pattern = pattern
        self.numbered = _INDEX_RE.search(pattern) is not None
        self.next_number = first
        self.index_key = index_key

    @property
    def directory(self):
        return os.path.dirname(self.pattern)

    def _take_number(self, output):
        if not self.numbered:
            return -1
        if self.index_key in output:
            return output[self.index_key]
        if self.next_number == -1:
            self.next_number = len(glob.glob(_INDEX_RE.sub('*', self.pattern)))
        self.next_number += 1
        return self.next_number - 1

    def make(self, infos, output):
        number = self._take_number(output)
        fields = {k: v for k, v in output.items() if k != 'i' and isinstance(v, (str, int, float))}
        if '{basename}' in self.pattern and 'basename' not in fields:
            fields['basename'] = os.path.basename(infos['filename']).rpartition('.')[0]
        return self.pattern.format(number, i=number, **fields)


class FileSaver(Callback):
    """Writes `output[data_key]` to a file and records the file name in the entry under `key`.  An entry that already has a
    file name keeps it (a re-synthesized text overwrites its own file); a result whose value already is a file name is only
    recorded.  `save_fn(filename, data, **{k: output[k] for k in additional_keys})` does the writing."""

    def __init__(self, key, file_format, *, data_key=None, additional_keys=None, index=-1, index_key=None,
                 save_fn=None, name=None, **kwargs):
        super().__init__(name=name or 'saving {}'.format(key), **kwargs)
        self.key = key
        self.data_key = data_key or key
        self.path = NumberedPath(file_format, first=index, index_key=index_key)
        self.additional_keys = list(additional_keys or [])
        self.save_fn = save_fn

    @property
    def file_format(self):
        return self.path.pattern

    def build(self):
        super().build()
        if self.path.directory:
            os.makedirs(self.path.directory, exist_ok=True)

    def apply(self, infos, output, **_):
        value = output.get(self.key, None)
        if isinstance(value, str):
            infos.setdefault(self.key, value)
            return None
        if infos.get(self.key, None) is None:
            infos[self.key] = self.path.make(infos, output)
        self.save(infos[self.key], output[self.data_key], **{k: output[k] for k in self.additional_keys})
        return infos[self.key]

    def save(self, filename, data, **kwargs):
        self.save_fn(filename, data, **kwargs)


class AudioSaver(FileSaver):
    def __init__(self, key='audio', file_format=None, **kwargs):
        file_format = file_format or default_audio_format()
        kwargs.setdefault('save_fn', write_audio)
        kwargs['additional_keys'] = ['rate']
        super().__init__(key, file_format, **kwargs)


class SpectrogramSaver(FileSaver):
    def __init__(self, key='mel', file_format='mel-{}.npy', **kwargs):
        kwargs.setdefault('save_fn', lambda filename, data: np.save(filename, data))
        super().__init__(key, file_format, **kwargs)

    def save(self, filename, data):
        if isinstance(data, list):                                  # one mel per sentence part -> one [T, 80] file
            data = np.concatenate([_to_numpy(d) for d in data], axis=0)
        else:
            data = _to_numpy(data)
        return super().save(filename, data)


class JSONSaver(FileSaver):
    def __init__(self, data, filename, primary_key, *, name='saving json', **kwargs):
        super().__init__(None, filename, name=name, **kwargs)
        self.data = data
        self.primary_key = primary_key

    def __repr__(self):
        return '<{} file={}>'.format(self.__class__.__name__, self.file_format)

    def apply(self, infos, output, **_):
        if self.primary_key not in infos:
            return None
        key = infos[self.primary_key]
        if not isinstance(key, str):
            return None
        self.data[key] = to_json(infos)
        self.save()
        return key

    def save(self):
        tmp = self.file_format + '.tmp'
        with open(tmp, 'w', encoding='utf-8') as f:
            json.dump(self.data, f, indent=4)
        os.replace(tmp, self.file_format)


class FunctionCallback(Callback):
    def __init__(self, fn, name=None, include_outputs=True, **kwargs):
        super().__init__(name=name or getattr(fn, '__name__', fn.__class__.__name__), **kwargs)
        self.fn = fn
        self.include_outputs = include_outputs

    def apply(self, infos, output, **kwargs):
        kwargs.update(infos)
        if self.include_outputs:
            kwargs.update(output)
        return self.fn(**kwargs)


class QueueCallback(Callback):
    def __init__(self, queue, name='queue', **kwargs):
        super().__init__(name=name, **kwargs)
        self.queue = queue

    def apply(self, infos, output, **kwargs):
        kwargs.update(infos)
        return self.queue.put(kwargs)


def load_json(filename, default=None):
    if not os.path.exists(filename):
        return default
    with open(filename, 'r', encoding='utf-8') as f:
        return json.load(f)


def apply_callbacks(callbacks, infos, output, save=True, **kwargs):
    """Runs every callback; returns the key `JSONSaver` stored the entry under (or None)."""
    if not callbacks:
        return None
    entry = None
    for callback in callbacks:
        if isinstance(callback, FileSaver) and not save:
            continue
        try:
            res = callback(infos, output, **kwargs)
            if isinstance(callback, JSONSaver):
                entry = res
        except Exception as exc:                                     # noqa: BLE001 -- mirrors the reference: log and go on
            logger.error('- An exception occured while calling %s : %s', callback, exc)
    return entry
