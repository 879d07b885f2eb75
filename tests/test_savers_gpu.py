"""Result savers and the `map.json` cache on the real path: `tts(text, directory=...)` through the HIP
engine writes what the reference writes (models/tts/tacotron2.py:227-241,276-352; utils/callbacks/file_saver.py:118-133) and a
second call for the same text is served from the cache without touching the decoder."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_tts_saves_through_the_engine_and_the_second_call_comes_from_map_json(gpu_engine, tmp_path):
    from scipy.io import wavfile
    from text_to_speech_amd.callbacks import to_pcm16
    from text_to_speech_amd.runtime import HipRuntime
    from text_to_speech_amd.tacotron2 import Tacotron2, tts
    from text_to_speech_amd.waveglow import WaveGlow

    calls = {'decoder': 0, 'vocoder': 0}

    class Counting:
        """the runtime object `compiled_infer`, with a launch counter in front"""
        def __init__(self, rt, key):
            self.rt, self.key, self.engine = rt, key, rt.engine

        def __call__(self, *args, **kwargs):
            calls[self.key] += 1
            return self.rt(*args, **kwargs)

    model = Tacotron2(Counting(HipRuntime('unused', engine=gpu_engine, model='tacotron2', seed=1), 'decoder'), lang='en')
    vocoder = WaveGlow(Counting(HipRuntime('unused', engine=gpu_engine, model='waveglow', seed=2), 'vocoder'))
    text = 'The quick brown fox jumps over the lazy dog.'
    out_dir = tmp_path / 'outputs'
    kw = dict(model=model, vocoder=vocoder, directory=str(out_dir), audio_filename='audio-{}.wav', max_length=48,
              max_trial=1, early_stopping=False, return_output=True)
    first = tts(text, **kw)
    assert calls == {'decoder': 1, 'vocoder': 1}
    audio = first['audio']
    assert audio.shape == (48 * 256,) and first['rate'] == 22050 and np.isfinite(audio).all()
    # the file on disk is the returned waveform after the reference's preparation (mean removed, peak 32767: audio_io.py:359-361)
    wav_path = out_dir / 'audios' / 'audio-0.wav'
    assert wav_path.exists()
    rate, pcm = wavfile.read(wav_path)
    assert rate == 22050 and pcm.dtype == np.int16 and np.array_equal(pcm, to_pcm16(audio))
    # map.json: keyed by the raw text, points at the file, carries the text fields but no tensors (file_saver.py:118-133)
    index = json.loads((out_dir / 'map.json').read_text())
    assert list(index) == [text]
    entry = index[text]
    assert entry['audio'] == str(wav_path) and entry['text'] == text and entry['cleaned'] == model.clean_text(text)
    assert 'mel' not in entry and 'attention' not in entry
    # same text again: served from the cache -- no decoder launch, no vocoder launch, nothing rewritten
    mtime = os.path.getmtime(wav_path)
    again = tts(text, **{**kw, 'return_output': None})
    assert calls == {'decoder': 1, 'vocoder': 1}
    assert again['audio'] == str(wav_path) and os.path.getmtime(wav_path) == mtime
    # another sentence gets the next index; overwrite=True re-synthesizes the first one into its own file
    second = 'Another sentence.'
    tts(second, **kw)
    assert calls == {'decoder': 2, 'vocoder': 2} and (out_dir / 'audios' / 'audio-1.wav').exists()
    tts(text, overwrite=True, **kw)
    assert calls == {'decoder': 3, 'vocoder': 3}
    index = json.loads((out_dir / 'map.json').read_text())
    assert list(index) == [text, second] and index[text]['audio'] == str(wav_path)
