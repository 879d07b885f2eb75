"""Host-side code against outputs of the REFERENCE's own functions (tests/golden/host_vectors.json, generated in the build
container by tests/golden/make_host_vectors.py from /root/reference/utils/text/numbers.py and models/tts/waveglow.py)."""
import json
import os

import numpy as np
import pytest

VEC = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'host_vectors.json'), encoding='utf-8'))


def test_normalize_numbers_matches_the_reference_pipeline():
    """133 en / fr / be inputs through the reference's `normalize_numbers` (with the absent num2words library replaced by this
    repository's speller -- so the regular expressions, their order, the currency / time grammar and the Belgian rewriting
    are the reference's, the base spelling is ours)."""
    from text_to_speech_amd.text import normalize_numbers
    assert len(VEC['normalize_numbers']) >= 120
    bad = [(lang, text, want, normalize_numbers(text, lang)) for lang, text, want in VEC['normalize_numbers']
           if normalize_numbers(text, lang) != want]
    assert not bad, bad[:10]


def test_window_starts_match_the_reference_get_steps():
    from text_to_speech_amd.waveglow import window_starts
    for length, win, hop, want in VEC['get_steps']:
        assert [int(v) for v in np.asarray(window_starts(length, win, hop)).reshape(-1)] == want, (length, win, hop)


@pytest.mark.parametrize('batch', [False, True])
def test_window_stitching_keeps_exactly_the_samples_the_reference_keeps(batch):
    """`WaveGlow.infer(win_len=..., hop_len=...)` over a vocoder stand-in whose samples are their own absolute indices: the
    stitched result must be the reference's selection, sample for sample -- including the reference's behaviour when two
    windows do not overlap at all (`part[start : -0 // 2]` is empty: only the last window survives)."""
    from text_to_speech_amd.waveglow import WaveGlow

    for case in VEC['stitch']:
        seq_len, win_len = case['seq_len'], case['win_len']
        offsets = {}

        def fake(mel, **_):
            # a window is recognised by its first frame's value (= its absolute frame index)
            mel = np.asarray(mel)
            rows = []
            for m in mel:
                start = int(round(float(m[0, 0])))
                rows.append(np.arange(start * 256, (start + m.shape[0]) * 256, dtype=np.float64))
            return np.stack(rows)

        mel = np.repeat(np.arange(seq_len, dtype=np.float64)[None, :, None], 80, axis=2)
        out = WaveGlow(fake).infer(mel, win_len=win_len, hop_len=case['hop_len'], batch=batch)
        want = np.concatenate([np.arange(a, b, dtype=np.float64) for a, b in
                               [r for r in case['kept_sample_ranges'] if r is not None]] or [np.zeros((0,))])
        assert out.shape[-1] == case['n_samples'], case
        assert np.array_equal(np.asarray(out).reshape(-1), want), case
