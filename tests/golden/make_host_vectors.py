#!/usr/bin/env python3
"""Generates tests/golden/host_vectors.json: outputs of the REFERENCE's own host-side code on chosen inputs -- DATA only
(inputs and expected outputs); this script only runs in the build container (it reads /root/reference), its output travels.

  "normalize_numbers"  `utils/text/numbers.py::normalize_numbers` (units, math symbols, durations, clocks, digit grouping,
                       currencies, decimals, ordinals, the Belgian rewriting of :101-131).  The module is stdlib-only EXCEPT
                       for its lazy `from num2words import num2words` (third-party, not installed here, no network): that
                       one import is satisfied with this repository's own restatement of num2words' en / fr conventions
                       (text_to_speech_amd.text.number_to_words & co).  So these vectors pin everything numbers.py itself
                       does -- the regular expressions, their order, the Belgian rewriting, currency / time grammar -- but
                       NOT the spelling conventions of the absent library (those rest on the reference's own unit-test
                       vectors in text_vectors.json).  Inputs that would hand num2words a non-integer are excluded.
  "get_steps"          `models/tts/waveglow.py::_get_steps` (window starts).  The module cannot be imported (keras): the
                       function's source is taken from the file with `ast` and executed with `math` / `numpy` only.
  "stitch"             the window / overlap / stitch arithmetic of `WaveGlow.infer` (models/tts/waveglow.py:114-142): the
                       statements that resolve `hop_len`, compute `starts` / `overlaps` and assemble `audio` are taken from the
                       method with `ast` and run on parts whose samples are their own absolute sample indices, so the
                       result shows exactly which sample of which window the reference keeps.

Usage: python tests/golden/make_host_vectors.py   (rewrites tests/golden/host_vectors.json)
"""
import ast
import importlib.util
import json
import math
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF_NUMBERS = '/root/reference/utils/text/numbers.py'
REF_WAVEGLOW = '/root/reference/models/tts/waveglow.py'

NUMBER_INPUTS = {
    'en': ['1', '12', '101', '1234', '2000000', '1,234', '3,000,000', '1 000', '12 345 678', '007', '3-4', '10-2 people',
           '1+1', '2 * 3 = 6', '-5', '+7 or -7', '2^10', '9/3', '1.5', '3.14', '0.05', '10.007', '2.50 dollars', '$5', '$1',
           '$2.50', '$0.99', '$1.01', '$.5', '£20', '£1,000', '1st', '2nd', '3rd', '4th', '11th', '21st', '100th', '5 sec',
           '1 s', '3min', '2 h', '1h 1min 1sec', '12:30:15', '01:01:01', '5 km/h', '10 mg/l', '3 l/min', '1 g/s',
           'room 101 has 3 beds, 2 desks and 1 door.', 'in 1999 there were 12,000 of them', 'call 555-1234 now',
           'version 2.0.1', 'a1b2', '1er janvier', '80', '90', '71', '99'],
    'fr': ['1', '12', '21', '71', '80', '81', '91', '99', '100', '101', '200', '1000', '2001', '1 000 000', '3,14', '1,5',
           '1,234,567', '0,05', '10,007', '$5', '$1', '$2.50', '£20', '1er', '2ème', '3ieme', '21ème', '80ème', '100ème',
           '5 sec', '1 s', '1min', '2 h', '1h 1min 1sec', '12:30:15', '01:01:01', '5 km/h', '1 t', '2 t', '10 mg/l',
           '1+1', '2 * 3 = 6', '-5', '2^10', 'il y a 3 chats et 12 chiens.', 'le 14 juillet 1789', 'chambre 101',
           '3-4 fois', '70', '90'],
    'be': ['70', '71', '72', '79', '80', '90', '91', '95', '99', '170', '1990', '70ème', '71ème', '90ème', '91ème', '99ème',
           '3,14', '1 t', '1h 1min', '12:30:15', '$1', 'il y a 93 chats', '5 km/h', '1+1', '2001'],
}

GET_STEPS = [(1000, 256, 192), (256, 256, 192), (300, 256, 192), (777, 128, 64), (257, 256, 192), (448, 256, 192),
             (449, 256, 192), (512, 256, 256), (513, 256, 256), (2000, 512, 448), (1025, 512, 1), (600, 100, 37),
             (101, 100, 99), (5000, 1000, 936), (64, 64, 1), (65, 64, 1), (130, 64, 63)]
# (seq_len, win_len, hop_len as passed to infer: negative = win_len + hop_len, float = fraction of win_len)
STITCH = [(1000, 256, -64), (300, 256, -64), (777, 128, -64), (2000, 512, -64), (2000, 512, 0.75), (2000, 512, 0.5),
          (601, 200, 150), (1025, 512, -1), (900, 300, 299), (512, 256, 1.0), (768, 256, 256), (640, 256, 0.9)]


def load_numbers():
    sys.path.insert(0, ROOT)
    from text_to_speech_amd import text as mine
    fake = types.ModuleType('num2words')

    def num2words(number, ordinal=False, lang='en'):
        s = str(number)
        if not s.isdigit():
            raise ValueError(f'stand-in for the absent num2words library: integers only, got {number!r}')
        n = int(s)
        if lang == 'en':
            return mine.ordinal_to_words(n) if ordinal else mine.number_to_words(n)
        if lang == 'fr':
            return mine.ordinal_to_words_fr(n) if ordinal else mine.number_to_words_fr(n)
        raise NotImplementedError(lang)
    fake.num2words = num2words
    sys.modules['num2words'] = fake
    spec = importlib.util.spec_from_file_location('ref_numbers', REF_NUMBERS)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    return ref


def load_waveglow_pieces():
    tree = ast.parse(open(REF_WAVEGLOW, encoding='utf-8').read())
    steps_fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == '_get_steps')
    ns = {'math': math, 'np': np}
    exec(compile(ast.Module(body=[steps_fn], type_ignores=[]), REF_WAVEGLOW, 'exec'), ns)
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == 'WaveGlow')
    infer = next(n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == 'infer')

    def targets(node):
        return [t.id for t in getattr(node, 'targets', []) if isinstance(t, ast.Name)]

    def mentions(node, name):
        return any(isinstance(x, ast.Name) and x.id == name for x in ast.walk(node))

    picked = []
    for node in infer.body:
        if isinstance(node, ast.If) and mentions(node.test, 'hop_len'):
            picked.append(node)                                   # hop_len resolution
        elif isinstance(node, ast.Assign) and targets(node) and targets(node)[0] in ('starts', 'overlaps'):
            picked.append(node)
        elif isinstance(node, ast.Assign) and targets(node) == ['audio']:
            picked.append(node)
        elif isinstance(node, ast.For) and mentions(node.iter, 'audio_parts'):
            picked.append(node)
    kinds = [type(n).__name__ for n in picked]
    assert kinds == ['If', 'If', 'Assign', 'Assign', 'Assign', 'For'], kinds
    resolve = compile(ast.Module(body=picked[:4], type_ignores=[]), REF_WAVEGLOW, 'exec')     # hop_len, starts, overlaps
    assemble = compile(ast.Module(body=picked[4:], type_ignores=[]), REF_WAVEGLOW, 'exec')    # audio = [] ; for ...
    return ns['_get_steps'], resolve, assemble


def main():
    out = {'normalize_numbers': [], 'normalize_numbers_skipped': [], 'get_steps': [], 'stitch': []}
    ref = load_numbers()
    for lang, texts in NUMBER_INPUTS.items():
        for t in texts:
            try:
                out['normalize_numbers'].append([lang, t, ref.normalize_numbers(t, lang=lang)])
            except Exception as exc:                              # a non-integer reached the absent library: not pinnable here
                out['normalize_numbers_skipped'].append([lang, t, f'{type(exc).__name__}: {exc}'])
    get_steps, resolve, assemble = load_waveglow_pieces()
    for length, win, hop in GET_STEPS:
        out['get_steps'].append([length, win, hop, [int(v) for v in np.asarray(get_steps(length, win, hop)).reshape(-1)]])
    for seq_len, win_len, hop_len in STITCH:
        ns = {'seq_len': seq_len, 'win_len': win_len, 'hop_len': hop_len, '_get_steps': get_steps, 'np': np, 'math': math}
        exec(resolve, ns)
        starts = np.asarray(ns['starts']).reshape(-1)
        # every window's "audio" = the absolute indices of its samples
        ns['audio_parts'] = [np.arange(int(s) * 256, min(seq_len, int(s) + win_len) * 256) for s in starts]
        exec(assemble, ns)
        kept = [[int(a[0]), int(a[-1]) + 1] if len(a) else None for a in ns['audio']]
        flat = np.concatenate(ns['audio']) if len(ns['audio']) else np.zeros((0,), np.int64)
        out['stitch'].append({'seq_len': seq_len, 'win_len': win_len, 'hop_len': hop_len, 'resolved_hop_len': int(ns['hop_len']),
                              'starts': [int(v) for v in starts],
                              'overlaps': [int(v) for v in np.asarray(ns['overlaps']).reshape(-1)],
                              'kept_sample_ranges': kept, 'n_samples': int(len(flat)),
                              'seamless': bool(len(flat) == seq_len * 256 and np.array_equal(flat, np.arange(seq_len * 256)))})
    with open(os.path.join(HERE, 'host_vectors.json'), 'w', encoding='utf-8') as f:
        json.dump(out, f, ensure_ascii=False, indent=1)
    print('wrote', os.path.join(HERE, 'host_vectors.json'), {k: len(v) for k, v in out.items()})
    for s in out['stitch']:
        if not s['seamless']:
            print('  not seamless:', {k: s[k] for k in ('seq_len', 'win_len', 'hop_len', 'overlaps', 'n_samples')})
    for s in out['normalize_numbers_skipped']:
        print('  skipped:', s)


if __name__ == '__main__':
    main()
