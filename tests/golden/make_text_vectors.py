#!/usr/bin/env python3
"""Generates tests/golden/text_vectors.json: known-answer vectors for the text front-end.

Two kinds of vectors, both DATA (inputs and expected outputs), no reference source:

  "reference_tests"   the live cases the reference's own unit tests hold for this front-end
                      (/root/reference/tests/test_utils_text.py:52-123 number / time / money / ordinal / others /
                      abbreviation cases, :163-189 the 22 `split_sentences` cases, :191-207 `merge_texts`), transcribed as
                      (input, expected) pairs.  The unit cases at :36-51 are dead in the reference (a second `test_math` at
                      :52 shadows the first one) and contradict utils/text/numbers.py:75-79, so they are NOT included.
  "reference_outputs" outputs of the reference's own `split_sentences` / `split_text` / `merge_texts`
                      (utils/text/text_processing.py -- stdlib-only, so that one module CAN be imported in the build
                      container, unlike the rest of `utils.text` which needs unidecode / num2words) on additional inputs.
                      This script loads that module by path from /root/reference; it therefore only runs in the build
                      container, and only its OUTPUT (this JSON) is committed and travels.

Usage: python tests/golden/make_text_vectors.py   (rewrites tests/golden/text_vectors.json)
"""
import importlib.util
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference/utils/text/text_processing.py'

REFERENCE_TESTS = {
    # normalize_numbers(text) with the default language 'en'  (test_utils_text.py:52-63)
    'math': [['-1', ' minus one'], ['+1', ' plus one'], ['1+1', 'one plus one'], ['1 + 1', 'one plus one'],
             ['1-1', 'one - one'], ['1 - 1', 'one minus one'], ['-1 - -1', ' minus one minus minus one'],
             ['-1 * -1', ' minus one times minus one'],
             ['-1.5 / - 2.5', ' minus one punt five divide by minus two punt five']],
    # (lang, text, expected)  (:66-68)
    'money': [['en', '$10', 'ten dollars'], ['fr', '$1', 'un dollar']],
    # (:70-91)
    'time': [['en', '1 sec', 'one second'], ['en', '10sec', 'ten seconds'], ['en', '1min', 'one minute'],
             ['en', '2 min 1sec', 'two minutes and one second'], ['en', '1h', 'one hour'],
             ['en', '2 h 2min', 'two hours and two minutes'], ['en', '10 h 10 sec', 'ten hours and ten seconds'],
             ['en', '23h 59min 59sec', 'twenty-three hours and fifty-nine minutes and fifty-nine seconds'],
             ['fr', '1 sec', 'une seconde'], ['fr', '10sec', 'dix secondes'], ['fr', '1min', 'une minute'],
             ['fr', '2 min 1sec', 'deux minutes et une seconde'], ['fr', '1h', 'une heure'],
             ['fr', '2 h 2min', 'deux heures et deux minutes'], ['fr', '10 h 10 sec', 'dix heures et dix secondes'],
             ['fr', '23h 59min 59sec', 'vingt-trois heures et cinquante-neuf minutes et cinquante-neuf secondes']],
    # (:93-108)
    'ordinal': [['en', '3rd', 'third'], ['en', '2nd', 'second'], ['en', '10ème', 'tenth'], ['fr', '2nd', 'deuxième'],
                ['fr', '3rd', 'troisième'], ['fr', '10ième', 'dixième'], ['be', '1er', 'premier'],
                ['be', '3rd', 'troisième'], ['be', '70ème', 'septantième'], ['be', '91ème', 'nonante et unième']],
    # (:110-118)
    'others': [['1, 2, 3, 4 and 5 !', 'one, two, three, four and five !'], ['1 000', 'one thousand'],
               ['1 000 000', 'one million'], ['1.5', 'one punt five'],
               ['put during 3-4 min', 'put during three - four minutes']],
    # expand_abreviations(text, lang='en')  (:143-144)
    'abbreviations': [['Mr test', 'mister test'], ['Mr. test', 'mister test']],
    # (text, number of sentences)  (:163-189)
    'split_sentences': [
        ['Hello World !', 1], ['Hello World ! This is a test', 2], ['Hello World ? This is a test', 2],
        ['Hello World. This is a test', 2], ['Hello World... This is a test.', 2],
        ['This is an url : http://example.example.com', 1], ['This is an email : example.example@example.com', 1],
        ['1. First item.\n2. Second item.\n3. 3rd item.', 3],
        ['Examples :\n1. First item.\n2. Second item.\n3. 3rd item.', 4],
        ['Examples : \n1. First item.\n2. Second item.\n3. 3rd item.', 4],
        ['Example :\n1. First item\n    1.1 First item A\n    1.2 First item B\n2. Second item', 5],
        ['Items are : 1) First item 2) Second item 3) Third item', 1],
        ['List of items :\n- First item\n- Second item\n- Third item', 4],
        ['Equations :\n- 1 + 1 = 2\n- 1 - 1 = 0\n- -1 * 2 = -2', 4],
        ['Equation : 1.2 + 1.8 = 3.0', 1], ['Equation 1 : 1.2 + 1.8 = 3. \nEquation 2 : 1.8 - 1.8 = 0.\nend', 3],
        ['1.2 + 1.3 = 2.5. 1.3 + 1.2 = 2.5. Addition is commutative', 3],
        ['She said "Hello World !"', 1], ['E.g., "Hello World !"', 1], ['E.g. "Hello World !"', 1],
        ['M.H.C.P. stands for "Mental Health Counsuling Program"', 1]],
    # merge_texts(texts, max_length) -> merged indices, character tokens  (:191-198)
    'merge_texts_chars': [[['a', 'b', 'c', 'd'], 2, [[0, 1], [2, 3]]], [['a', 'b', 'c', 'd'], 3, [[0, 1, 2], [3]]],
                          [['ab', 'c', 'def', 'g'], 3, [[0, 1], [2], [3]]]],
}

# additional inputs for the reference's own splitter (outputs are generated below)
EXTRA_SPLIT_INPUTS = [
    'Hello there. General test!',
    'First sentence of the stream. A second, slightly longer sentence follows. Third. And the last one.',
    'Dr. Smith went to Washington. He arrived at 3 p.m. and left.',
    'Wait... what ? Really !? Yes.',
    'A paragraph.\n\nAnother paragraph without final dot',
    'Section 1.2.3. Title of the section. Content follows here.',
    'He said (quietly). "Nothing at all." Then he left.',
    "It's 'quoted.' Then more. `code.` End",
    'No terminal punctuation',
    '   Leading and trailing spaces .   ',
    'a.b.c d.e. F.G. done. next',
    'i.e. this is fine. e.g. that too. End.',
    'Line one\nline two lowercase\nLine Three\n* star item\n4 digit item',
    'Multiple   spaces.   After   dot.',
    'Ends with ellipsis...',
    '? Starts with punctuation. ok',
    '',
    '1. Fact questions (e.g., "What did Albert Einstein win the Nobel Prize for ?")\n',
]
EXTRA_SPLIT_TEXT = [
    # (text, max_length) -> split_text(text, max_length) with the default character tokenizer
    ['Hello World ! This is a test. And another one, with a comma: and a colon (and a parenthesis) to finish.', 30],
    ['Hello World ! This is a test. And another one, with a comma: and a colon (and a parenthesis) to finish.', 50],
    ['Short text.', 100],
    ['One two three four five six seven eight nine ten eleven twelve thirteen fourteen fifteen sixteen.', 20],
    ['First sentence of the stream. A second, slightly longer sentence follows. Third. And the last one.', 40],
    ['Lorem ipsum dolor sit amet, consectetur adipiscing elit, sed do eiusmod tempor incididunt ut labore et dolore '
     'magna aliqua. Ut enim ad minim veniam, quis nostrud exercitation ullamco laboris nisi ut aliquip ex ea commodo '
     'consequat. Duis aute irure dolor in reprehenderit in voluptate velit esse cillum dolore eu fugiat nulla '
     'pariatur.', 150],
]


def main():
    spec = importlib.util.spec_from_file_location('ref_text_processing', REF)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    out = {'reference_tests': REFERENCE_TESTS, 'reference_outputs': {}}
    inputs = [c[0] for c in REFERENCE_TESTS['split_sentences']] + EXTRA_SPLIT_INPUTS
    out['reference_outputs']['split_sentences'] = [[t, ref.split_sentences(t)] for t in inputs]
    out['reference_outputs']['split_sentences_strip'] = [[t, ref.split_sentences(t, strip=True)] for t in inputs]
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        out['reference_outputs']['split_text'] = [[t, n, ref.split_text(t, n)] for t, n in EXTRA_SPLIT_TEXT]
    merges = [c[:2] for c in REFERENCE_TESTS['merge_texts_chars']] + [
        [['Hello World', '!', ' This is', 'a test.'], 12], [['abc', 'defgh', 'i', 'jk', 'lmnopq'], 6]]
    out['reference_outputs']['merge_texts'] = []
    for texts, n in merges:
        merged, _, idx = ref.merge_texts(texts, n)
        out['reference_outputs']['merge_texts'].append([texts, n, merged, idx])
    # (texts, max_length, max_overlap, max_overlap_len) -> chunks, indices
    out['reference_outputs']['merge_texts_overlap'] = []
    parts = ['One.', 'Two two.', 'Three three three.', 'Four.', 'Five five.', 'Six six six six.', 'Seven.']
    for n, ov, ovl in ((24, 1, 0.5), (30, 2, 0.5), (30, 2, 6), (40, 3, 0.2), (18, 1, 0.9)):
        merged, _, idx = ref.merge_texts(parts, n, ov, ovl)
        out['reference_outputs']['merge_texts_overlap'].append([parts, n, ov, ovl, merged, idx])
    # the reference's own expectations must hold for its own function (sanity of the transcription)
    for text, n in REFERENCE_TESTS['split_sentences']:
        assert len(ref.split_sentences(text)) == n, (text, ref.split_sentences(text))
    with open(os.path.join(HERE, 'text_vectors.json'), 'w', encoding='utf-8') as f:
        json.dump(out, f, ensure_ascii=False, indent=1)
    print('wrote', os.path.join(HERE, 'text_vectors.json'))


if __name__ == '__main__':
    main()
