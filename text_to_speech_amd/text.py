"""Character-level text front-end (cleaners -> ids) for the English / French Tacotron2 models.

Restates the parts of /root/reference/utils/text that `Tacotron2.infer` needs (models/tts/tacotron2.py:135-149):
  symbol tables            utils/text/__init__.py:28-55   (en: '_' '-' "!'(),.:;? " A-Z a-z + 84 ARPAbet = 148 ids)
  english / french cleaners utils/text/cleaners.py:296-345 (lowercase, abbreviations, numbers -> words, whitespace)
  sentence splitting        utils/text/text_processing.py:34,228
`num2words` / `unidecode` are not installed here, so numbers are spelled by the small English speller below and
accents are folded with unicodedata; the character->id mapping itself is exactly the reference table.
"""
from __future__ import annotations

import re
import unicodedata

import numpy as np

_pad = '_'
_punctuation = '!\'(),.:;? '
_special = '-'
_letters = 'ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz'
_accents = 'éèêîçô'
_cmudict_symbols = [
    'AA', 'AA0', 'AA1', 'AA2', 'AE', 'AE0', 'AE1', 'AE2', 'AH', 'AH0', 'AH1', 'AH2', 'AO', 'AO0', 'AO1', 'AO2', 'AW',
    'AW0', 'AW1', 'AW2', 'AY', 'AY0', 'AY1', 'AY2', 'B', 'CH', 'D', 'DH', 'EH', 'EH0', 'EH1', 'EH2', 'ER', 'ER0', 'ER1',
    'ER2', 'EY', 'EY0', 'EY1', 'EY2', 'F', 'G', 'HH', 'IH', 'IH0', 'IH1', 'IH2', 'IY', 'IY0', 'IY1', 'IY2', 'JH', 'K',
    'L', 'M', 'N', 'NG', 'OW', 'OW0', 'OW1', 'OW2', 'OY', 'OY0', 'OY1', 'OY2', 'P', 'R', 'S', 'SH', 'T', 'TH', 'UH',
    'UH0', 'UH1', 'UH2', 'UW', 'UW0', 'UW1', 'UW2', 'V', 'W', 'Y', 'Z', 'ZH']
en_symbols = [_pad] + list(_special) + list(_punctuation) + list(_letters) + ['@' + s for s in _cmudict_symbols]
fr_symbols = [_pad] + list(_special) + list(_punctuation) + list(_letters) + list(_accents)

_abbreviations = [(re.compile(r'\b%s\.' % a, re.IGNORECASE), b) for a, b in [
    ('mrs', 'misess'), ('mr', 'mister'), ('dr', 'doctor'), ('st', 'saint'), ('co', 'company'), ('jr', 'junior'),
    ('maj', 'major'), ('gen', 'general'), ('drs', 'doctors'), ('rev', 'reverend'), ('lt', 'lieutenant'),
    ('hon', 'honorable'), ('sgt', 'sergeant'), ('capt', 'captain'), ('esq', 'esquire'), ('ltd', 'limited'),
    ('col', 'colonel'), ('ft', 'fort')]]
_ones = ['zero', 'one', 'two', 'three', 'four', 'five', 'six', 'seven', 'eight', 'nine', 'ten', 'eleven', 'twelve',
         'thirteen', 'fourteen', 'fifteen', 'sixteen', 'seventeen', 'eighteen', 'nineteen']
_tens = ['', '', 'twenty', 'thirty', 'forty', 'fifty', 'sixty', 'seventy', 'eighty', 'ninety']


def number_to_words(n: int) -> str:
    if n < 0:
        return 'minus ' + number_to_words(-n)
    if n < 20:
        return _ones[n]
    if n < 100:
        return _tens[n // 10] + ('-' + _ones[n % 10] if n % 10 else '')
    if n < 1000:
        return _ones[n // 100] + ' hundred' + (' and ' + number_to_words(n % 100) if n % 100 else '')
    for value, name in ((10 ** 9, 'billion'), (10 ** 6, 'million'), (1000, 'thousand')):
        if n >= value:
            rest = n % value
            return number_to_words(n // value) + ' ' + name + ((', ' if rest >= 100 else ' and ') +
                                                               number_to_words(rest) if rest else '')
    return str(n)


def _expand_numbers(text: str) -> str:
    text = re.sub(r'(\d),(\d{3})', r'\1\2', text)
    text = re.sub(r'(\d+)\.(\d+)', lambda m: number_to_words(int(m.group(1))) + ' point ' +
                  ' '.join(_ones[int(c)] for c in m.group(2)), text)
    return re.sub(r'\d+', lambda m: number_to_words(int(m.group(0))), text)


def english_cleaners(text: str) -> str:
    text = unicodedata.normalize('NFKD', text).encode('ascii', 'ignore').decode('ascii')
    text = text.lower()
    text = _expand_numbers(text)
    for regex, repl in _abbreviations:
        text = regex.sub(repl, text)
    return re.sub(r'\s+', ' ', text).strip()


def french_cleaners(text: str) -> str:
    text = text.lower()
    keep = set(_accents)
    text = ''.join(c if c in keep else unicodedata.normalize('NFKD', c).encode('ascii', 'ignore').decode('ascii')
                   for c in text)
    return re.sub(r'\s+', ' ', text).strip()


def split_sentences(text: str):
    parts = re.split(r'(?<=[.!?])\s+', text.strip())
    return [p for p in parts if p]


def split_text(text: str, max_length: int):
    """Greedy sentence packing up to `max_length` characters (text_processing.py:34)."""
    out, cur = [], ''
    for sent in split_sentences(text):
        while len(sent) > max_length:                       # overlong sentence: cut at the last space
            cut = sent.rfind(' ', 0, max_length)
            cut = cut if cut > 0 else max_length
            if cur:
                out.append(cur)
                cur = ''
            out.append(sent[:cut].strip())
            sent = sent[cut:].strip()
        if cur and len(cur) + 1 + len(sent) > max_length:
            out.append(cur)
            cur = sent
        else:
            cur = (cur + ' ' + sent).strip()
    if cur:
        out.append(cur)
    return out


class CharTokenizer:
    def __init__(self, lang='en'):
        self.lang = lang
        self.symbols = en_symbols if lang == 'en' else fr_symbols
        self.index = {s: i for i, s in enumerate(self.symbols)}
        self.cleaner = english_cleaners if lang == 'en' else french_cleaners

    @property
    def vocab_size(self):
        return len(self.symbols)

    def clean_text(self, text, **_):
        return self.cleaner(text)

    def encode(self, text, cleaned=False):
        if not cleaned:
            text = self.clean_text(text)
        ids = [self.index[c] for c in text if c in self.index and c != _pad]
        return np.asarray(ids, dtype=np.int32)
