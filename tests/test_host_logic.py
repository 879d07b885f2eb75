"""CPU: host-side logic that mirrors the reference's Python callers of the hot path."""
import queue
import re

import numpy as np
import pytest

from text_to_speech_amd.engine import Tacotron2InferenceOutput


# ------------------------------------------------------------------ H1: WaveGlow wrapper
def test_get_steps_known_answers():
    from text_to_speech_amd.waveglow import _get_steps
    # derived by hand from models/tts/waveglow.py:156-164
    assert list(_get_steps(1000, 256, 192)) == [0, 186, 372, 558, 744]
    assert _get_steps(256, 256, 192) == [0]
    assert list(_get_steps(300, 256, 192)) == [0, 44]
    s = _get_steps(777, 128, 64)
    assert s[0] == 0 and s[-1] == 777 - 128 and np.all(np.diff(s) <= 64)


class FakeVocoder:
    """compiled_infer stand-in: audio sample k of frame t = t + k/256 (so stitching errors are visible)."""
    def __init__(self):
        self.calls = []

    def __call__(self, mel, **kwargs):
        mel = np.asarray(mel)
        self.calls.append((mel.shape, kwargs))
        B, T, _ = mel.shape
        base = mel[:, :, 0]                                    # channel 0 carries the global frame index
        return (base[:, :, None] + np.arange(256)[None, None] / 256.).reshape(B, T * 256).astype(np.float32)


def _indexed_mel(T):
    mel = np.zeros((T, 80), np.float32)
    mel[:, 0] = np.arange(T)
    return mel


def test_waveglow_direct_and_padded():
    from text_to_speech_amd.waveglow import WaveGlow
    f = FakeVocoder()
    wg = WaveGlow(f)
    out = wg.infer(_indexed_mel(10))
    assert out.shape == (1, 2560) and f.calls[0][0] == (1, 10, 80)
    # seq_len <= win_len with the non-keras runtime: no padding, direct call (waveglow.py:94-96)
    out = wg.infer(_indexed_mel(10), win_len=64)
    assert out.shape == (1, 2560) and f.calls[-1][0] == (1, 10, 80)
    out = wg.infer(_indexed_mel(10), win_len=64, force_pad=True)
    assert out.shape == (1, 2560) and f.calls[-1][0] == (1, 64, 80)
    assert f.calls[-1][1]['padding_multiple'] == 64


@pytest.mark.parametrize('batch', [False, True])
def test_waveglow_windowed_stitching_is_seamless(batch):
    from text_to_speech_amd.waveglow import WaveGlow
    f = FakeVocoder()
    T = 1000
    out = WaveGlow(f).infer(_indexed_mel(T), win_len=256, hop_len=-64, batch=batch)
    assert out.shape == (T * 256,)
    expect = (np.arange(T)[:, None] + np.arange(256)[None] / 256.).reshape(-1)
    np.testing.assert_allclose(out, expect, atol=1e-3)          # centre-half stitching keeps every sample once
    assert len(f.calls) == (1 if batch else 5)


def test_waveglow_float_win_len():
    from text_to_speech_amd.waveglow import WaveGlow
    f = FakeVocoder()
    out = WaveGlow(f).infer(_indexed_mel(100), win_len=64., hop_len=-16)   # ceil(100/64)*64 = 128 >= T -> direct
    assert out.shape == (1, 25600)


# ------------------------------------------------------------------ H2: Tacotron2 wrapper
class FakeSynth:
    def __init__(self, lengths_seq, default=50):
        self.lengths_seq = list(lengths_seq)
        self.default = default
        self.calls = []

    def __call__(self, inputs, max_length=None, **kwargs):
        tok = inputs[0] if isinstance(inputs, tuple) else inputs
        self.calls.append((np.asarray(tok).copy(), max_length, kwargs))
        n = self.lengths_seq.pop(0) if self.lengths_seq else self.default
        T = 400
        mel = np.zeros((1, T, 80), np.float32)
        mel[0, :, 0] = np.arange(T)
        return Tacotron2InferenceOutput(decoder_output=mel, mel=mel, stop_tokens=np.zeros((1, T), np.float32),
                                        attention_weights=np.zeros((1, T, tok.shape[1]), np.float32),
                                        lengths=np.array([n], np.int32))


def test_tacotron2_infer_retry_rule_and_result_keys():
    from text_to_speech_amd.tacotron2 import Tacotron2
    from text_to_speech_amd.waveglow import WaveGlow
    text = 'Hello world, this is a test.'                       # 28 chars -> ratio bounds (2, 10) => 56 < n < 280
    synth = FakeSynth([20, 300, 100])                           # two rejected trials, third accepted
    voc = FakeVocoder()
    out = Tacotron2(synth).infer(text, vocoder=WaveGlow(voc))
    assert len(synth.calls) == 3 and synth.calls[0][1] == 10.
    assert set(out) == {'text', 'cleaned', 'splitted', 'mel', 'attention', 'audio', 'rate', 'time'}
    assert out['mel'][0].shape == (100, 80) and out['attention'][0].shape[0] == 100
    assert out['audio'].shape == (100 * 256,) and out['rate'] == 22050
    assert abs(out['time'] - 100 * 256 / 22050) < 1e-9
    assert out['cleaned'] == 'hello world, this is a test.'


def test_tacotron2_infer_gives_up_after_max_trial_and_splits():
    from text_to_speech_amd.tacotron2 import Tacotron2
    synth = FakeSynth([5, 5, 5, 5, 5, 5, 5])
    out = Tacotron2(synth).infer('Short one. Another short sentence here.', max_text_length=-2, max_trial=2)
    assert len(synth.calls) == 4                                # 2 sentences x max_trial
    # sentences keep the space behind their terminator (split_sentences) and the cleaners collapse but do not strip it
    assert out['splitted'] == ['short one. ', 'another short sentence here.']
    assert 'audio' not in out and len(out['mel']) == 2


def test_tacotron2_empty_text_gives_silence():
    from text_to_speech_amd.tacotron2 import Tacotron2
    from text_to_speech_amd.waveglow import WaveGlow
    out = Tacotron2(FakeSynth([])).infer('...', vocoder=WaveGlow(FakeVocoder()))
    assert out['audio'].shape == (int(0.15 * 22050),) and out['time'] == 0.15


def test_stream_consumes_queue_in_order():
    from text_to_speech_amd.tacotron2 import Tacotron2
    from text_to_speech_amd.waveglow import WaveGlow
    synth = FakeSynth([])
    q = queue.Queue()
    for s in ['first sentence to say.', 'second sentence to say.', None]:
        q.put(s)
    got = []
    res = Tacotron2(synth).stream(q, vocoder=WaveGlow(FakeVocoder()), save=False,
                                  callbacks=[lambda text, **_: got.append(text)])
    assert res == []                                            # stream = predict(return_results=False)
    # two warm-up calls ('hello 64', 'hello 128': tacotron2.py:354-356) precede the stream
    assert [c[2].get('padding_multiple') for c in synth.calls[:2]] == [64, 128]
    assert got == ['first sentence to say.', 'second sentence to say.']


def test_predict_saves_audio_and_map_json_and_reuses_the_cache(tmp_path):
    """get_inference_callbacks + predicted-map semantics (tacotron2.py:130-132,227-241,276-352; file_saver.py)."""
    import json
    from scipy.io import wavfile
    from text_to_speech_amd.tacotron2 import Tacotron2, tts
    from text_to_speech_amd.waveglow import WaveGlow
    d = str(tmp_path / 'out')
    synth = FakeSynth([], default=100)
    model, voc = Tacotron2(synth), WaveGlow(FakeVocoder())
    t1, t2 = 'Hello world, this is a test.', 'Another sentence to synthesize.'
    res = model.predict([t1, {'text': t2}], vocoder=voc, directory=d)
    # a JSON saver is active -> predict returns the `predicted` entries (audio = file name), not the raw outputs
    assert [r['text'] for r in res] == [t1, t2]
    assert res[0]['audio'] == str(tmp_path / 'out' / 'audios' / 'audio-0.wav')
    assert res[1]['audio'].endswith('audio-1.wav') and set(res[0]) == {'text', 'cleaned', 'splitted', 'rate', 'time', 'audio'}
    rate, wav = wavfile.read(res[0]['audio'])
    assert rate == 22050 and wav.dtype == np.int16 and wav.shape == (100 * 256,)
    m = json.load(open(tmp_path / 'out' / 'map.json'))
    assert list(m) == [t1, t2] and m[t1]['audio'] == res[0]['audio'] and m[t2]['splitted'] == ['another sentence to synthesize.']
    # second run on a fresh model object: entries come from map.json, the synthesizer is not called, nothing is rewritten
    synth2 = FakeSynth([], default=100)
    seen = []
    res2 = Tacotron2(synth2).predict(t1, vocoder=voc, directory=d, post_processing=lambda text, **kw: seen.append((text, sorted(kw))))
    assert synth2.calls == [] and res2 == [m[t1]] and seen[0][0] == t1 and 'audio' in seen[0][1]
    assert sorted(p.name for p in (tmp_path / 'out' / 'audios').iterdir()) == ['audio-0.wav', 'audio-1.wav']
    # overwrite=True re-synthesizes but keeps the entry's file name (file_saver.py:107-110)
    res3 = Tacotron2(synth2).predict(t1, vocoder=voc, directory=d, overwrite=True)
    assert len(synth2.calls) == 1 and res3[0]['audio'] == res[0]['audio']
    # a new text continues the numbering from the files on disk (file_saver.py:70-72)
    res4 = Tacotron2(synth2).predict('A third one for the index.', vocoder=voc, directory=d)
    assert res4[0]['audio'].endswith('audio-2.wav')
    # no vocoder: mels are saved instead (save_mel defaults to save and vocoder is None)
    res5 = Tacotron2(FakeSynth([], default=100)).predict('Only the spectrogram.', directory=str(tmp_path / 'mels_only'))
    mel = np.load(res5[0]['mel'])
    assert mel.shape == (100, 80) and 'audio' not in res5[0]
    # save=False: nothing on disk, raw outputs come back; tts() unwraps a single text (models/tts/__init__.py:76-77)
    out = tts(t1, model=model, vocoder=voc, save=False)
    assert isinstance(out, dict) and out['audio'].shape == (100 * 256,) and out['mel'][0].shape == (100, 80)
    outs = tts([t1, t2], model=model, vocoder=voc, save=False)
    assert isinstance(outs, list) and len(outs) == 2


def test_overlapped_stream_keeps_order_overlaps_work_and_propagates_errors():
    """stream(overlap=True): Tacotron2(n + 1) in a worker thread while WaveGlow(n) runs in the caller's thread."""
    import time as _time
    from text_to_speech_amd.tacotron2 import Tacotron2
    from text_to_speech_amd.waveglow import WaveGlow

    spans = {'synth': [], 'voc': []}                                  # (start, end) of every call, by stage

    class SlowSynth(FakeSynth):
        def __call__(self, inputs, **kw):
            t = _time.monotonic()
            _time.sleep(0.05)
            out = super().__call__(inputs, **kw)
            spans['synth'].append((t, _time.monotonic()))
            return out

    class SlowVocoder(FakeVocoder):
        def __call__(self, mel, **kw):
            t = _time.monotonic()
            _time.sleep(0.05)
            out = super().__call__(mel, **kw)
            spans['voc'].append((t, _time.monotonic()))
            return out

    texts = [f'This is sentence number {i} of the stream.' for i in range(8)]
    synth, voc = SlowSynth([], default=120), SlowVocoder()
    model = Tacotron2(synth)
    model.precompile_for_stream = lambda **kw: None                  # keep the timing clean
    got = []
    model.stream(iter(texts), vocoder=WaveGlow(voc), save=False, overlap=True,
                 callbacks=[lambda text, audio, **_: got.append((text, len(audio)))])
    assert [g[0] for g in got] == texts and all(n == 120 * 256 for _, n in got)
    # the work overlaps: Tacotron2(n + 1) starts before WaveGlow(n) has ended (judged from the recorded spans, not from a
    # wall-clock budget, which a loaded machine does not keep)
    assert len(spans['synth']) == len(spans['voc']) == 8
    overlapped = sum(spans['synth'][n + 1][0] < spans['voc'][n][1] for n in range(7))
    assert overlapped >= 6, spans
    # same results as the sequential path
    seq = Tacotron2(SlowSynth([], default=120)).predict(texts[:2], vocoder=WaveGlow(SlowVocoder()), save=False)
    ovl = Tacotron2(SlowSynth([], default=120)).predict(texts[:2], vocoder=WaveGlow(SlowVocoder()), save=False, overlap=True)
    assert [r['text'] for r in ovl] == texts[:2] and all(np.array_equal(a['audio'], b['audio']) for a, b in zip(seq, ovl))

    class Boom(FakeSynth):
        def __call__(self, inputs, **kw):
            if len(self.calls) == 1:
                raise RuntimeError('synth exploded')
            return super().__call__(inputs, **kw)
    with pytest.raises(RuntimeError, match='synth exploded'):
        Tacotron2(Boom([], default=120)).predict(texts[:3], vocoder=WaveGlow(FakeVocoder()), save=False, overlap=True)


def test_callbacks_failures_are_logged_not_raised(caplog):
    from text_to_speech_amd.callbacks import FunctionCallback, QueueCallback, apply_callbacks
    q = queue.Queue()

    def boom(**_):
        raise RuntimeError('nope')
    apply_callbacks([FunctionCallback(boom), QueueCallback(q)], {'text': 'a'}, {'audio': 1})
    assert q.get_nowait() == {'text': 'a'} and 'nope' in caplog.text


# ------------------------------------------------------------------ text front-end
def test_symbol_table_and_encoding():
    from text_to_speech_amd.text import CharTokenizer, en_symbols, fr_symbols, number_to_words
    assert len(en_symbols) == 148 and en_symbols[0] == '_' and en_symbols[1] == '-'      # vocab_size default
    assert en_symbols[2:12] == list("!'(),.:;? ") and en_symbols[12] == 'A' and en_symbols[38] == 'a'
    assert len(fr_symbols) == 70
    tk = CharTokenizer('en')
    ids = tk.encode('Dr. Smith has 42 cats!')
    assert ids.dtype == np.int32 and ids.min() >= 1 and ids.max() < 148
    assert tk.clean_text('Dr. Smith has 42 cats!') == 'doctor smith has forty-two cats!'
    assert number_to_words(1905) == 'one thousand, nine hundred and five'
    assert ''.join(en_symbols[i] for i in tk.encode('abc, xyz')) == 'abc, xyz'


# ------------------------------------------------------------------ runtime seam
def test_text_normalisation_known_answers():
    """Known answers derived from the reference source (utils/text/numbers.py:249-271 order of substitutions, cleaners.py
    complete_cleaners :296-342, abreviations/en.json); num2words conventions for the number words."""
    from text_to_speech_amd.text import (english_cleaners, french_cleaners, number_to_words, number_to_words_fr,
                                         ordinal_to_words, ordinal_to_words_fr)
    en = english_cleaners
    assert en('Dr. Smith paid $3.50 on the 21st.') == 'doctor smith paid three dollars, fifty cents on the twenty-first.'
    assert en('St Mary') == 'saint mary'                               # abbreviations match without the dot too (:176-186)
    assert en('3h 20min') == 'three hours and twenty minutes' and en('1h') == 'one hour'
    assert en('12:30:05') == 'twelve hours and thirty minutes and five seconds'
    assert en('50 km/h') == 'fifty kilometers per hour'
    assert en('1,234,567') == 'one million, two hundred and thirty-four thousand, five hundred and sixty-seven'
    assert en('3.14') == 'three punt fourteen' and en('0.05') == 'zero punt zero five'      # (sic) 'punt', numbers.py:18-20
    assert en('100% & more') == 'one hundred percent and more'
    assert en('-5 + 3 = -2') == ' minus five plus three equal minus two'      # leading space kept (test_utils_text.py:52)
    # (sic) English text is not ASCII-folded by the reference (cleaners.py:336 assigns the result to `lang`)
    assert en('**bold** £20 naïve café') == 'bold twenty pounds naïve café'
    assert en('**bold** £20 naïve café', convert_to_ascii=True) == 'bold twenty pounds naive cafe'
    assert en('$1') == 'one dollar' and en('$0.01') == 'one cent' and en('10-5') == 'ten - five'
    assert [ordinal_to_words(n) for n in (1, 2, 3, 5, 12, 20, 21, 100, 101)] == [
        'first', 'second', 'third', 'fifth', 'twelfth', 'twentieth', 'twenty-first', 'one hundredth',
        'one hundred and first']
    assert number_to_words(1001) == 'one thousand and one' and number_to_words(110) == 'one hundred and ten'
    assert en('The FBI and NASA, I think', to_expand_acronyms=True) == 'the af be eye and an ae as ae, i think'
    assert en('sooooo goood', max_repetition=2) == 'soo good' and en('hello tf', replacements={'hello': 'bye'}) == 'bye tensorflow'
    fr = french_cleaners
    assert fr('Il y a 91 chats et 1 200 oiseaux le 1er mai.') == \
        'il y a quatre-vingt-onze chats et mille deux cents oiseaux le premier mai.'
    assert fr('3,14') == 'trois virgule quatorze'                      # single comma = decimal in French (:196-203)
    assert fr('Noël naïf à Liège') == 'noel nahif a liège'            # tremas rule; only âéèêîç survive (cleaners.py:52)
    assert [number_to_words_fr(n) for n in (21, 70, 71, 80, 81, 99, 100, 200, 201, 1000, 2000, 80000, 1000000)] == [
        'vingt et un', 'soixante-dix', 'soixante et onze', 'quatre-vingts', 'quatre-vingt-un', 'quatre-vingt-dix-neuf',
        'cent', 'deux cents', 'deux cent un', 'mille', 'deux mille', 'quatre-vingt mille', 'un million']
    assert [ordinal_to_words_fr(n) for n in (1, 2, 4, 5, 9, 21)] == ['premier', 'deuxième', 'quatrième', 'cinquième',
                                                                      'neuvième', 'vingt et unième']


def test_build_runtime_registry_errors():
    from text_to_speech_amd.runtime import Runtime, HipRuntime, build_runtime, _runtimes
    assert _runtimes == {'hip': HipRuntime} and issubclass(HipRuntime, Runtime)
    with pytest.raises(ValueError, match='Unsupported runtime'):
        build_runtime('onnx', 'x')
    with pytest.raises(TypeError):
        Runtime('p')                                             # abstract, like the reference's ABC


def test_runtime_argument_resolution_with_fake_engine():
    """max_length float -> int(max token count * f) (tacotron2_arch.py:886-892); unknown kwargs ignored; masks sampled."""
    from text_to_speech_amd.runtime import HipRuntime

    class Eng:
        def tacotron2_infer(self, tokens, **kw):
            self.kw = kw
            self.tokens = tokens
            return 'ok'

        def waveglow_infer(self, mel, z=None, sigma=1.0, precision='f32', seed=None, offset=0):
            self.z, self.sigma, self.precision, self.seed, self.offset = z, sigma, precision, seed, offset
            return np.zeros((mel.shape[0], mel.shape[1] * 256), np.float32)

    eng = Eng()
    rt = HipRuntime('fake', engine=eng, seed=0)
    tok = np.zeros((1, 64), np.int32)
    tok[0, :37] = 5
    assert rt(tok, max_length=10., padding_multiple=64, some_unknown_kwarg=1) == 'ok'
    assert eng.kw['max_len'] == 370 and eng.kw['early_stopping'] is True
    m = eng.kw['prenet_masks']
    assert m.shape == (1, 370, 2, 256) and set(np.unique(m)) == {0.0, 2.0} and 0.45 < (m > 0).mean() < 0.55
    rt(tok, max_length=25, deterministic=True, attn_mask_win_len=12, attn_mask_offset=0.5)
    assert eng.kw['max_len'] == 25 and eng.kw['prenet_masks'] is None and eng.kw['attn_mask_offset'] == 6
    # noise: drawn on the device from the runtime's (seed, running block offset); an explicit seed restarts at offset 0
    out = rt(np.zeros((2, 3, 80), np.float32), sigma=0.7)
    from text_to_speech_amd.runtime import MASK_STREAM, NOISE_STREAM
    assert out.shape == (2, 768) and eng.z is None and eng.seed == 0 ^ NOISE_STREAM and eng.sigma == 0.7
    first = eng.offset
    rt(np.zeros((2, 3, 80), np.float32))
    assert eng.seed == 0 ^ NOISE_STREAM and eng.offset == first + 2 * 3 * 256 // 4
    rt(np.zeros((2, 3, 80), np.float32), seed=9)
    assert (eng.seed, eng.offset) == (9 ^ NOISE_STREAM, 0) and MASK_STREAM != NOISE_STREAM      # dropout and noise never share blocks
    z = np.ones((1, 96, 8), np.float32)
    rt(np.zeros((3, 80), np.float32), z=z)
    assert eng.z is z and eng.seed is None
    rt(np.zeros((3, 80), np.float32), deterministic=True)
    assert eng.z is None and eng.seed is None and eng.precision == 'f32'
    # vocoder precision: runtime-wide default, per-call override, validation
    rt16 = HipRuntime('fake', engine=eng, vocoder_precision='f16')
    rt16(np.zeros((1, 2, 80), np.float32), deterministic=True)
    assert eng.precision == 'f16'
    rt16(np.zeros((1, 2, 80), np.float32), deterministic=True, precision='f32')
    assert eng.precision == 'f32'
    with pytest.raises(ValueError, match='vocoder_precision'):
        HipRuntime('fake', engine=eng, vocoder_precision='int8')


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    from text_to_speech_amd import _lib
    with pytest.raises(_lib.HipLibraryError, match='no CPU fallback'):
        _lib.load_library(str(tmp_path / 'nope.so'))


# ------------------------------------------------------------------ SV2TTS wrapper (models/tts/sv2tts_tacotron2.py:103-128)
def test_select_embedding_modes():
    import pandas as pd
    from text_to_speech_amd.tacotron2 import select_embedding
    pool = np.arange(12, dtype=np.float32).reshape(3, 4)
    assert np.array_equal(select_embedding(pool, 1), pool[1])
    for m in ('mean', 'avg', 'average'):
        np.testing.assert_allclose(select_embedding(pool, m), pool.mean(0))
    drawn = select_embedding(pool, 'random')
    assert any(np.array_equal(drawn, r) for r in pool)
    assert np.array_equal(select_embedding(pool, lambda e: e[-1] * 2), pool[-1] * 2)
    assert np.array_equal(select_embedding(pool[0], 0), pool[0])             # a 1-D array is a collection of one
    with pytest.raises(ValueError, match='Unknown embedding selection mode'):
        select_embedding(pool, 'median')
    df = pd.DataFrame({'id': ['a', 'b', 'a'], 'embedding': list(pool)})
    np.testing.assert_allclose(select_embedding(df, 'mean', id='a'), pool[[0, 2]].mean(0))
    np.testing.assert_allclose(select_embedding(df, 'mean', id='zzz'), pool.mean(0))     # no match: filter dropped
    assert np.array_equal(select_embedding(df, 1), pool[1])


def test_sv2tts_infer_resolves_the_speaker_embedding():
    from text_to_speech_amd.tacotron2 import SV2TTSTacotron2
    pool = np.random.default_rng(0).standard_normal((3, 256)).astype(np.float32)
    synth = FakeSynth([], default=100)
    model = SV2TTSTacotron2(synth, lang='fr', embeddings=pool, use_label_embedding=True)
    text = 'Bonjour à tous, ceci est un test.'

    # the runtime receives (tokens [1, Tin], embedding [1, 256]); FakeSynth records only the tokens, so spy on the call
    seen = []

    class Spy(FakeSynth):
        def __call__(self, inputs, **kw):
            seen.append(inputs)
            return FakeSynth.__call__(self, inputs, **kw)
    spy = Spy([], default=100)
    model.compiled_infer = spy
    model.infer(text)                                                # default selector 0 -> first row
    assert isinstance(seen[-1], tuple) and seen[-1][0].shape[0] == 1 and np.array_equal(seen[-1][1], pool[:1])
    model.infer(text, embeddings=2)
    assert np.array_equal(seen[-1][1][0], pool[2])
    model.infer(text, embeddings=None)                               # use_label_embedding -> mean
    np.testing.assert_allclose(seen[-1][1][0], pool.mean(0), rtol=1e-6)
    model.infer(text, embeddings='mean')
    np.testing.assert_allclose(seen[-1][1][0], pool.mean(0), rtol=1e-6)
    model.infer(text, embeddings={'mode': 1})
    assert np.array_equal(seen[-1][1][0], pool[1])
    vec = np.ones(256, np.float32)
    model.infer(text, embeddings=vec)                                # an explicit vector passes through
    assert np.array_equal(seen[-1][1][0], vec)
    with pytest.raises(ValueError, match='shape'):
        model.infer(text, embeddings={'mode': lambda e: e[0][:8]})      # a selector that yields the wrong width
    with pytest.raises(ValueError, match='no speaker embeddings'):
        SV2TTSTacotron2(spy, lang='fr').infer(text)


# ---- audio writer (reference: utils/audio/audio_io.py:347-380, audio_processing.py:51-62) ---------------------------------
def test_pcm16_preparation_follows_the_reference_normalisation():
    from text_to_speech_amd.callbacks import to_pcm16
    a = np.array([0.1, 0.3, -0.2, 0.0], np.float32)
    centred = a - a.mean()
    want = (centred * (32767 / np.abs(centred).max())).astype(np.int16)           # mean removed, peak -> 32767, truncation
    np.testing.assert_array_equal(to_pcm16(a), want)
    assert int(np.abs(to_pcm16(a)).max()) == 32767
    np.testing.assert_array_equal(to_pcm16(np.zeros(5, np.float32)), np.zeros(5, np.int16))      # silence stays silence
    assert to_pcm16(np.zeros(0, np.float32)).shape == (0,)
    np.testing.assert_array_equal(to_pcm16(np.array([2.0, -0.5], np.float32), normalize=False), [32767, -16384])


def test_write_audio_dispatches_on_the_extension(tmp_path, monkeypatch):
    import os
    import stat
    from scipy.io import wavfile
    from text_to_speech_amd import callbacks
    audio = np.sin(np.arange(2205) * 0.05).astype(np.float32) * 0.3
    wav = tmp_path / 'a.wav'
    callbacks.write_audio(str(wav), audio, 22050)
    rate, data = wavfile.read(str(wav))
    assert rate == 22050 and data.dtype == np.int16
    np.testing.assert_array_equal(data, callbacks.to_pcm16(audio))

    monkeypatch.setenv('PATH', str(tmp_path / 'nothing'))
    assert callbacks.default_audio_format() == 'audio-{}.wav'
    with pytest.raises(ValueError, match='needs the ffmpeg executable'):
        callbacks.write_audio(str(tmp_path / 'a.mp3'), audio, 22050)
    assert not (tmp_path / 'a.mp3').exists()

    # a stand-in encoder: records its arguments, copies the PCM it is fed to the output file
    bindir = tmp_path / 'bin'
    bindir.mkdir()
    fake = bindir / 'ffmpeg'
    fake.write_text('#!/bin/sh\nfor last; do :; done\necho "$@" > "$last.args"\ncat > "$last"\n')
    fake.chmod(fake.stat().st_mode | stat.S_IEXEC)
    monkeypatch.setenv('PATH', f'{bindir}:/usr/bin:/bin')
    assert callbacks.default_audio_format() == 'audio-{}.mp3'                       # the reference's default name
    mp3 = tmp_path / 'b.mp3'
    callbacks.write_audio(str(mp3), audio, 16000)
    np.testing.assert_array_equal(np.frombuffer(mp3.read_bytes(), np.int16), callbacks.to_pcm16(audio))
    args = (tmp_path / 'b.mp3.args').read_text().split()
    assert args[args.index('-f') + 1] == 's16le' and args[args.index('-ar') + 1] == '16000' and args[args.index('-ac') + 1] == '1'
    saver = callbacks.AudioSaver(file_format=str(tmp_path / 'out' / callbacks.default_audio_format()))
    infos = {}
    saver(infos, {'audio': audio, 'rate': 22050})
    assert infos['audio'].endswith('audio-0.mp3') and os.path.exists(infos['audio'])

    fake.write_text('#!/bin/sh\necho "no such codec" >&2\nexit 3\n')
    with pytest.raises(RuntimeError, match='no such codec'):
        callbacks.write_audio(str(tmp_path / 'c.ogg'), audio, 22050)


# ---- speaker-embedding files (reference: utils/embeddings.py:30-212, utils/file_utils.py:252-292,358-397) ----------------------
def test_embeddings_to_np_parses_the_reference_string_forms():
    from text_to_speech_amd.embeddings import embeddings_to_np
    np.testing.assert_allclose(embeddings_to_np('[0.5, -1.25, 3]'), [0.5, -1.25, 3.0])
    np.testing.assert_allclose(embeddings_to_np('[0.5\t-1.25\t3]'), [0.5, -1.25, 3.0])
    np.testing.assert_allclose(embeddings_to_np('[ 1.  2.5\n  3. ]'), [1.0, 2.5, 3.0])            # numpy's own repr
    np.testing.assert_allclose(embeddings_to_np('[[1, 2] [3, 4]]'), [[1, 2], [3, 4]])
    np.testing.assert_allclose(embeddings_to_np('[[1, 2, 3], [4, 5]]'), [[1, 2, 3], [4, 5, 0]])  # ragged rows are padded
    m = np.arange(6, dtype=np.float32).reshape(2, 3)
    assert embeddings_to_np(m) is m
    np.testing.assert_allclose(embeddings_to_np({'embedding': ['[1, 2]', '[3, 4]']}), [[1, 2], [3, 4]])
    with pytest.raises(ValueError, match='does not exist'):
        embeddings_to_np('no/such/file.npy')


def test_load_embeddings_reads_the_reference_h5_table_without_h5py():
    import os
    from test_hdf5_reader import H5, expected
    from text_to_speech_amd.embeddings import load_embeddings
    from text_to_speech_amd.tacotron2 import select_embedding
    table = load_embeddings(os.path.join(H5, 'embeddings_ref_format'))          # extension resolved like the reference does
    assert list(table['id']) == ['siwis', 'siwis', 'bob', 'alice', 'bob']
    assert table['filename'][3] == 'wavs/alice_3.wav'
    want = expected('/embedding', (5, 16), '<f4')
    np.testing.assert_array_equal(np.stack(table['embedding'].values), want)
    # aggregate_on = 'id', mode 0: every row gets the first embedding of its speaker
    np.testing.assert_array_equal(table['speaker_embedding'][4], want[2])
    np.testing.assert_array_equal(table['speaker_embedding'][1], want[0])
    mean = load_embeddings(os.path.join(H5, 'embeddings_ref_format.h5'), aggregate_mode='mean')
    np.testing.assert_allclose(mean['speaker_embedding'][2], (want[2] + want[4]) / 2, rtol=1e-6)
    # and feeds select_embedding with its filters
    np.testing.assert_array_equal(select_embedding(table, mode=0, id='alice'), want[3])
    np.testing.assert_allclose(select_embedding(table, mode='mean', id='bob'), (want[2] + want[4]) / 2, rtol=1e-6)
    assert load_embeddings(os.path.join(H5, 'nothing_here')) is None


def test_embeddings_csv_npy_pkl_roundtrip_and_model_directory(tmp_path):
    import pandas as pd
    from text_to_speech_amd.embeddings import load_embeddings, save_embeddings
    from text_to_speech_amd.tacotron2 import SV2TTSTacotron2
    rng = np.random.default_rng(0)
    vecs = rng.standard_normal((3, 8)).astype(np.float32)
    table = pd.DataFrame({'id': ['a', 'b', 'a'], 'embedding': list(vecs)})
    for ext in ('.csv', '.pkl'):
        f = save_embeddings('voices' + ext, table, directory=str(tmp_path / ext[1:]))
        back = load_embeddings(f)
        assert list(back['id']) == ['a', 'b', 'a']
        np.testing.assert_allclose(np.stack(back['embedding'].values), vecs, rtol=1e-6)
    f = save_embeddings('m', vecs, directory=str(tmp_path / 'npy'))
    assert f.endswith('.npy')
    np.testing.assert_array_equal(load_embeddings(f[:-4]), vecs)                   # found without the extension
    # a model directory with a single embeddings file, as the reference keeps it
    d = tmp_path / 'sv2tts' / 'embeddings'
    save_embeddings('embeddings.csv', table, directory=str(d))
    model = SV2TTSTacotron2(lambda *a, **k: None, lang='en', embeddings_dir=str(d), embedding_dim=8)
    np.testing.assert_allclose(model.select_embedding(1), vecs[1], rtol=1e-6)
    np.testing.assert_allclose(model.select_embedding({'mode': 'mean', 'id': 'a'}), (vecs[0] + vecs[2]) / 2, rtol=1e-5)


# ---- tokenizer.json of a shipped model (reference: utils/text/tokenizer.py:54-213,394-452,661-702) ------------------------------
def test_tokenizer_config_file_roundtrip_and_reference_index_rules(tmp_path):
    import json
    from text_to_speech_amd.text import CharTokenizer, en_symbols
    base = CharTokenizer('en')
    f = base.save(str(tmp_path / 'tokenizer.json'))
    again = CharTokenizer.load_from_file(f)
    text = 'Dr. Smith paid $3.50 for 2 apples, i.e. too much!'
    assert again.clean_text(text) == base.clean_text(text)
    np.testing.assert_array_equal(again.encode(text), base.encode(text))
    assert again.vocab_size == base.vocab_size == len(en_symbols) and again.lang == 'en'

    # a config as the reference writes it: cleaners with keyword arguments, an unknown-token, sos / eos in use, specials first
    cfg = {'name': 'Tokenizer', 'vocab': list('_abc '), 'level': 0, 'template': None, 'lstrip': True, 'rstrip': True,
           'cleaners': [{'name': 'french_cleaners', 'to_lowercase': False}], 'split_pattern': None, 'bpe_pairs': None,
           'byte_encoder': None, 'bpe_end_of_word': None, 'pad_token': '_', 'sos_token': '<s>', 'eos_token': '</s>',
           'sep_token': None, 'ukn_token': '?', 'mask_token': None, 'additional_tokens': {}, 'sub_word_prefix': '',
           'use_sos_and_eos': True, 'add_special_tokens_at_end': False}
    p = tmp_path / 'fr.json'
    p.write_text(json.dumps(cfg), encoding='utf-8')
    tok = CharTokenizer.load_from_file(str(p))
    # __build_indexes: specials (ukn, sos, eos) first, then the vocabulary
    assert tok.symbols == ['?', '<s>', '</s>', '_', 'a', 'b', 'c', ' '] and tok.lang == 'fr'
    ids = tok.encode('  ab Zc ').tolist()                           # stripped; 'Z' is not lower-cased, so it is unknown
    assert ids == [1, 4, 5, 7, 0, 6, 2]
    cfg['use_sos_and_eos'], cfg['ukn_token'], cfg['add_special_tokens_at_end'] = False, None, True
    tok = CharTokenizer.from_config(cfg)
    assert tok.symbols == list('_abc ') and tok.encode('ab Zc').tolist() == [1, 2, 4, 3]       # unknown dropped, no sos / eos

    for bad in ({**cfg, 'level': 1}, {**cfg, 'bpe_pairs': [['a', 'b']]}, {**cfg, 'cleaners': ['no_such_cleaner']}):
        with pytest.raises(ValueError):
            CharTokenizer.from_config(bad)
