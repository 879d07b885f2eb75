"""Character-level text front-end (cleaners -> ids) for the English / French Tacotron2 models.

Restates the parts of /root/reference/utils/text that `Tacotron2.infer` needs (models/tts/tacotron2.py:135-149):
  symbol tables            utils/text/__init__.py:28-55   (en: '_' '-' "!'(),.:;? " A-Z a-z + 84 ARPAbet = 148 ids)
  english / french cleaners utils/text/cleaners.py:296-345 (lowercase, abbreviations, numbers -> words, whitespace)
  sentence splitting        utils/text/text_processing.py:34,228
  number normalisation     utils/text/numbers.py:249-271  (units, math symbols, durations, clocks, currencies, decimals,
                                                           ordinals, cardinals -- same order of substitutions)
`num2words` / `unidecode` are not installed here: cardinals / ordinals are spelled by the English and French spellers
below in num2words' conventions, and ASCII folding uses unicodedata plus a small ligature table; the character->id
mapping itself is exactly the reference table.
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

# ---- abbreviations (utils/text/abreviations/en.json; replaced with or without the dot: cleaners.py:176-186) -------------
_abbreviations_en = {
    'mrs': 'misess', 'mr': 'mister', 'dr': 'doctor', 'st': 'saint', 'co': 'company', 'jr': 'junior', 'maj': 'major',
    'gen': 'general', 'drs': 'doctors', 'rev': 'reverend', 'lt': 'lieutenant', 'hon': 'honorable', 'sgt': 'sergeant',
    'capt': 'captain', 'esq': 'esquire', 'ltd': 'limited', 'col': 'colonel', 'ft': 'fort', 'tf': 'tensorflow'}
_abbrev_re = {'en': re.compile(r'\b(%s)(\.|\b)' % '|'.join(sorted(_abbreviations_en, key=len, reverse=True)),
                               re.IGNORECASE)}


def expand_abbreviations(text: str, lang: str = 'en') -> str:
    regex = _abbrev_re.get(lang)
    if regex is None:
        return text
    return regex.sub(lambda m: _abbreviations_en[m.group(1).lower()], text)


# ---- numbers -> words (the reference delegates to `num2words`, numbers.py:96-132; same conventions restated here) ------
_ones = ['zero', 'one', 'two', 'three', 'four', 'five', 'six', 'seven', 'eight', 'nine', 'ten', 'eleven', 'twelve',
         'thirteen', 'fourteen', 'fifteen', 'sixteen', 'seventeen', 'eighteen', 'nineteen']
_tens = ['', '', 'twenty', 'thirty', 'forty', 'fifty', 'sixty', 'seventy', 'eighty', 'ninety']
_ord_irregular = {'one': 'first', 'two': 'second', 'three': 'third', 'five': 'fifth', 'eight': 'eighth', 'nine': 'ninth',
                  'twelve': 'twelfth'}


def number_to_words(n: int) -> str:
    """English cardinal, num2words style: 'one thousand, two hundred and thirty-four'."""
    if n < 0:
        return 'minus ' + number_to_words(-n)
    if n < 20:
        return _ones[n]
    if n < 100:
        return _tens[n // 10] + ('-' + _ones[n % 10] if n % 10 else '')
    if n < 1000:
        return _ones[n // 100] + ' hundred' + (' and ' + number_to_words(n % 100) if n % 100 else '')
    for value, name in ((10 ** 12, 'trillion'), (10 ** 9, 'billion'), (10 ** 6, 'million'), (1000, 'thousand')):
        if n >= value:
            rest = n % value
            return number_to_words(n // value) + ' ' + name + ((', ' if rest >= 100 else ' and ') +
                                                               number_to_words(rest) if rest else '')
    return str(n)


def ordinal_to_words(n: int) -> str:
    words = number_to_words(n)
    head, sep, last = words.rpartition('-') if '-' in words.split(' ')[-1] else words.rpartition(' ')
    if last in _ord_irregular:
        last = _ord_irregular[last]
    elif last.endswith('y'):
        last = last[:-1] + 'ieth'
    else:
        last += 'th'
    return head + sep + last


_fr_units = ['zéro', 'un', 'deux', 'trois', 'quatre', 'cinq', 'six', 'sept', 'huit', 'neuf', 'dix', 'onze', 'douze',
             'treize', 'quatorze', 'quinze', 'seize', 'dix-sept', 'dix-huit', 'dix-neuf']
_fr_tens = {20: 'vingt', 30: 'trente', 40: 'quarante', 50: 'cinquante', 60: 'soixante', 80: 'quatre-vingt'}


def number_to_words_fr(n: int) -> str:
    """French cardinal, num2words style: 'quatre-vingt-onze', 'deux cents', 'mille deux cent un'."""
    if n < 0:
        return 'moins ' + number_to_words_fr(-n)
    if n < 20:
        return _fr_units[n]
    if n < 100:
        ten = n // 10 * 10
        if ten in (70, 90):
            ten -= 10
        unit = n - ten
        base = _fr_tens[ten]
        if unit == 0:
            return base + ('s' if ten == 80 else '')
        if unit in (1, 11) and ten != 80:
            return base + ' et ' + _fr_units[unit]
        return base + '-' + _fr_units[unit]
    if n < 1000:
        h, rest = divmod(n, 100)
        head = 'cent' if h == 1 else _fr_units[h] + ' cent'
        if rest == 0:
            return head + ('s' if h > 1 else '')
        return head + ' ' + number_to_words_fr(rest)
    if n < 10 ** 6:
        k, rest = divmod(n, 1000)
        head = 'mille' if k == 1 else number_to_words_fr(k).removesuffix('s') + ' mille' if k % 100 == 80 or k % 100 == 0 and k >= 200 \
            else number_to_words_fr(k) + ' mille'
        return head + (' ' + number_to_words_fr(rest) if rest else '')
    for value, name in ((10 ** 9, 'milliard'), (10 ** 6, 'million')):
        if n >= value:
            k, rest = divmod(n, value)
            head = number_to_words_fr(k) + ' ' + name + ('s' if k > 1 else '')
            return head + (' ' + number_to_words_fr(rest) if rest else '')
    return str(n)


def ordinal_to_words_fr(n: int) -> str:
    if n == 1:
        return 'premier'
    w = number_to_words_fr(n)
    if w.endswith('e'):
        w = w[:-1]
    elif w.endswith('f'):
        w = w[:-1] + 'v'
    elif w.endswith('q'):
        w += 'u'
    elif w.endswith('s') and not w.endswith('trois'):
        w = w[:-1]
    return w + 'ième'


def _cardinal(n, lang):
    return number_to_words(int(n)) if lang == 'en' else number_to_words_fr(int(n))


def _ordinal(n, lang):
    return ordinal_to_words(int(n)) if lang == 'en' else ordinal_to_words_fr(int(n))


# tables of numbers.py:18-75
_comma_word = {'fr': 'virgule', 'en': 'punt'}                      # (sic) the reference says 'punt' for the decimal point
_math_words = {'=': {'fr': 'égal', 'en': 'equal'}, '+': {'fr': 'plus', 'en': 'plus'}, '-': {'fr': 'moins', 'en': 'minus'},
               '*': {'fr': 'fois', 'en': 'times'}, '/': {'fr': 'divisé par', 'en': 'divide by'},
               '^': {'fr': 'exposant', 'en': 'exponent'}}
_time_words = {'h': {'fr': 'heure', 'en': 'hour'}, 'min': {'fr': 'minute', 'en': 'minute'},
               'sec': {'fr': 'seconde', 'en': 'second'}, 's': {'fr': 'seconde', 'en': 'second'}}
_time_sep = {'fr': ' et ', 'en': ' and '}
_units = {'l': {'fr': 'litre', 'en': 'litre'}, 'g': {'fr': 'gramme', 'en': 'gram'}, 't': {'fr': 'tonne', 'en': 'tonne'},
          'm': {'fr': 'mètre', 'en': 'meter'}, 'mi': {'fr': 'mile', 'en': 'mile'}, 'o': {'fr': 'octet', 'en': 'bytes'},
          'b': {'fr': 'bar', 'en': 'bar'}, 'V': {'fr': 'volt', 'en': 'volt'}, 'W': {'fr': 'watt', 'en': 'watt'},
          'A': {'fr': 'ampère', 'en': 'ampere'}, 'Hz': {'fr': 'hertz', 'en': 'hertz'}, 'J': {'fr': 'joule', 'en': 'joul'},
          'N': {'fr': 'newton', 'en': 'newton'}}
_unit_prefix = {'n': {'fr': 'nano', 'en': 'nano'}, 'm': {'fr': 'mili', 'en': 'mili'}, 'c': {'fr': 'centi', 'en': 'centi'},
                'd': {'fr': 'déci', 'en': 'deci'}, 'k': {'fr': 'kilo', 'en': 'kilo'}, 'M': {'fr': 'méga', 'en': 'mega'},
                'G': {'fr': 'giga', 'en': 'giga'}, 'T': {'fr': 'tera', 'en': 'tera'}}
_units_sep = {'fr': 'par', 'en': 'per'}
_units_re = re.compile(r'(\d+)\s*(%s)?(%s)(?:\/(%s))\b' % ('|'.join(_unit_prefix), '|'.join(_units), '|'.join(_time_words)))
_math_symbol_re = re.compile(r'(?:(?<=\d)(\s*[\+\*\/\^\=]\s*(\+|\-\s*)?)(?=\d)|((?:^|\s+)(\-|\+)\s*(\+|\-\s*)?)(?=\d))')
_sec_pat = r'(\d+)\s*(?:sec|s)\b'
_min_pat = r'(\d+)\s*min(?:\s*%s)?' % _sec_pat
_hours_pat = r'(\d+)\s*h\s*(?:%s|%s)?' % (_min_pat, _sec_pat)
_time_re = re.compile(r'\b(?:%s|%s|%s)\b' % (_hours_pat, _min_pat, _sec_pat))
_clock_re = re.compile(r'(\d{1,2}):(\d{1,2}):(\d{1,2})')
_comma_number_re = re.compile(r'([0-9][0-9\,]+[0-9])')
_space_number_re = re.compile(r'[0-9]+( [0-9]{3,3})+(?!\d)')
_tiret_number_re = re.compile(r'([0-9]+-[0-9])')
_pounds_re = re.compile(r'£([0-9\,]*[0-9]+)')
_dollars_re = re.compile(r'\$([0-9\.\,]*[0-9]+)')
_decimal_number_re = re.compile(r'([0-9]+\.[0-9]+)')
_number_re = re.compile(r'[0-9]+')
_ordinal_re = re.compile(r'([0-9]+)(st|nd|rd|th|er|ère|ème|eme|ième|ieme)')


def _expand_units(m, lang):
    n, prefix, unit, per_time = m.groups()
    if n == '1' and lang == 'fr' and unit == 't':
        n = 'une'
    text = n + ' ' + (_unit_prefix[prefix][lang] if prefix else '') + _units[unit][lang]
    if n != 'une' and n > '1':
        text += 's'
    if per_time:
        text += ' ' + _units_sep[lang] + ' ' + _time_words[per_time][lang]
    return text


def _expand_hms(parts, lang):
    out = []
    for t, unit in parts:
        if t is None:
            continue
        word = _time_words[unit][lang]
        if int(t) > 1:
            word += 's'
        elif lang == 'fr' and int(t) == 1:
            t = 'une'
        out.append('{} {}'.format(t, word))
    return _time_sep[lang].join(out)


def _expand_time(m, lang):
    g = m.groups()
    return _expand_hms(((g[0], 'h'), (g[1] or g[4], 'min'), (g[2] or g[3] or g[5] or g[6], 'sec')), lang)


def _expand_dollars(m):
    parts = m.group(1).split('.')
    if len(parts) > 2:
        return m.group(1) + ' dollars'
    dollars = int(parts[0].replace(',', '')) if parts[0] else 0
    cents = int(parts[1]) if len(parts) > 1 and parts[1] else 0
    if dollars and cents:
        return '{} dollar{}, {} cent{}'.format(dollars, 's' if dollars != 1 else '', cents, 's' if cents != 1 else '')
    if dollars:
        return '{} dollar{}'.format(dollars, 's' if dollars != 1 else '')
    if cents:
        return '{} cent{}'.format(cents, 's' if cents != 1 else '')
    return 'zero dollars'


def _extend_with_zeros(text, lang):
    n = len(text) - len(text.lstrip('0'))
    words = _cardinal(text, lang)
    if n == 0:
        return words
    if n < 4:
        return ' '.join([_cardinal(0, lang)] * n + [words])
    return '{} {} {} {}'.format(_cardinal(n, lang), _math_words['*'][lang], _cardinal(0, lang), words)


def _expand_number(m, lang):
    num = m.group(0)
    if '.' not in num:
        return _cardinal(num, lang)
    ent, dec = num.split('.')
    if dec.count('0') == len(dec):
        return _cardinal(ent, lang)
    return '{} {} {}'.format(_cardinal(ent, lang), _comma_word[lang], _extend_with_zeros(dec, lang))


def normalize_numbers(text: str, lang: str = 'en', expand_symbols: bool = True) -> str:
    """Restatement of numbers.py:249-271, same order of substitutions."""
    if expand_symbols:
        text = _units_re.sub(lambda m: _expand_units(m, lang), text)
        text = _math_symbol_re.sub(lambda m: ' ' + ' '.join(_math_words[s][lang] for s in m.group(0).split()) + ' ', text)
    text = _time_re.sub(lambda m: _expand_time(m, lang), text)
    text = _clock_re.sub(lambda m: _expand_hms(zip(m.groups(), ('h', 'min', 'sec')), lang), text)
    text = _comma_number_re.sub(lambda m: m.group(1).replace(',', '.') if lang == 'fr' and m.group(1).count(',') == 1
                                else m.group(1).replace(',', ''), text)
    text = _tiret_number_re.sub(lambda m: m.group(1).replace('-', ' - '), text)
    text = _space_number_re.sub(lambda m: m.group(0).replace(' ', ''), text)
    text = _pounds_re.sub(r'\1 pounds', text)
    text = _dollars_re.sub(_expand_dollars, text)
    text = _decimal_number_re.sub(lambda m: _expand_number(m, lang), text)
    text = _ordinal_re.sub(lambda m: _ordinal(m.group(1), lang), text)
    return _number_re.sub(lambda m: _expand_number(m, lang), text)


# ---- special symbols, tremas, ascii folding (cleaners.py:188-201,263-277; numbers.py:273-284) -----------------------------
_special_symbols = {'=': {'fr': 'égal', 'en': 'equal'}, '+': {'fr': 'plus', 'en': 'plus'}, '/': {'fr': 'slash', 'en': 'slash'},
                    '^': {'fr': 'chapeau', 'en': 'hat'}, '%': {'fr': 'pourcent', 'en': 'percent'},
                    '§': {'fr': 'paragraphe', 'en': 'paragraph'}, '&': {'fr': 'et', 'en': 'and'},
                    '°C': {'fr': 'degrés', 'en': 'degrees'}, '°': {'fr': 'degrés', 'en': 'degrees'}}


def expand_special_symbols(text: str, lang: str) -> str:
    for symbol, words in _special_symbols.items():
        text = text.replace(symbol, ' ' + words[lang] + ' ')
    return text


def _to_ascii(text: str) -> str:
    """`unidecode` stand-in (not installed here): canonical decomposition, combining marks dropped, a few ligatures and
    typographic quotes / dashes mapped by hand."""
    table = {'œ': 'oe', 'Œ': 'OE', 'æ': 'ae', 'Æ': 'AE', 'ß': 'ss', '’': "'", '‘': "'", '“': '"', '”': '"', '–': '-',
             '—': '-', '…': '...', '«': '<<', '»': '>>'}
    text = ''.join(table.get(c, c) for c in text)
    return unicodedata.normalize('NFKD', text).encode('ascii', 'ignore').decode('ascii')


# letter names used to spell acronyms (cleaners.py:59-86)
_letter_names = {
    'a': ('ha', 'ae'), 'b': ('bé', 'be'), 'c': ('cé', 'ce'), 'd': ('dé', 'de'), 'e': ('euh', 'e'), 'f': ('effe', 'af'),
    'g': ('gé', 'ge'), 'h': ('hache', 'aich'), 'i': ('ih', 'eye'), 'j': ('ji', 'jay'), 'k': ('ka', 'kay'),
    'l': ('elle', 'el'), 'm': ('aime', 'am'), 'n': ('aine', 'an'), 'o': ('eau', 'oo'), 'p': ('pé', 'pe'), 'q': ('cu', 'qu'),
    'r': ('air', 'ar'), 's': ('aisse', 'as'), 't': ('thé', 'tea'), 'u': ('eu', 'yu'), 'v': ('vé', 've'),
    'w': ('double vé', 'double yu'), 'x': ('ix', 'ex'), 'y': ('i grec', 'way'), 'z': ('zed', 'ze')}
_acronym_re = re.compile(r"\b[A-Z]+(?!')\b")


def expand_acronyms(text: str, lang: str) -> str:
    """Words in capitals of at most 4 letters are spelled letter by letter (cleaners.py:211-218); 'I' stays in English."""
    col = 0 if lang in ('fr', 'be') else 1

    def spell(m):
        w = m.group(0)
        if len(w) > 4 or (w == 'I' and col == 1):
            return w
        return ' '.join(_letter_names.get(c.lower(), (c, c))[col] for c in w)
    return _acronym_re.sub(spell, text)


def collapse_repetitions(text: str, max_repetition: int) -> str:
    """Keeps at most `max_repetition` consecutive copies of a character (cleaners.py:254-261)."""
    if not text:
        return text
    keep, count = [text[0]], 1
    for c in text[1:]:
        count = 1 if c != keep[-1] else count + 1
        if count <= max_repetition:
            keep.append(c)
    return ''.join(keep)


def complete_cleaners(text: str, lang: str, *, to_lowercase=True, to_expand=True, to_expand_abrev=True,
                      to_expand_symbols=True, to_expand_acronyms=False, replacements=None, patterns=None,
                      max_repetition=-1, **_) -> str:
    """Restatement of cleaners.py:296-342, same order of steps."""
    if patterns:
        for pattern, repl in patterns.items():
            text = re.sub(pattern, repl, text)
    if replacements:
        low = {k.lower(): v for k, v in replacements.items()}
        regex = re.compile(r'\b(%s)\b' % '|'.join(re.escape(k) for k in replacements), re.IGNORECASE)
        text = regex.sub(lambda m: low[m.group(0).lower()], text)
    if to_expand_acronyms:
        text = expand_acronyms(text, lang)                       # (before lower-casing, or nothing would be left to spell)
    if to_lowercase:
        text = text.lower()
    if to_expand:
        text = re.sub(r'\*\*(.*)\*\*', r'\1', text)                           # remove_markdown
        if to_expand_abrev:
            text = expand_abbreviations(text, lang)
        text = normalize_numbers(text, lang if lang != 'be' else 'fr', expand_symbols=to_expand_symbols)
        if to_expand_symbols:
            text = expand_special_symbols(text, lang if lang != 'be' else 'fr')
    if lang in ('fr', 'be'):
        text = re.sub(r'(ï)', 'hi', re.sub('(aï)\b', 'aille', text))               # expand_tremas (sic: '\b' is a backspace there)
        keep = set(_accents_kept)
        text = ''.join(c if c in keep else _to_ascii(c) for c in text)
    else:
        text = _to_ascii(text)
    if max_repetition > 1:
        text = collapse_repetitions(text, max_repetition)
    return re.sub(r'\s+', ' ', text).strip()


_accents_kept = 'âéèêîç'                                            # cleaners.py:52


def english_cleaners(text: str, **kwargs) -> str:
    return complete_cleaners(text, 'en', **kwargs)


def french_cleaners(text: str, **kwargs) -> str:
    return complete_cleaners(text, 'fr', **kwargs)


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

    def clean_text(self, text, **kwargs):
        allowed = ('to_lowercase', 'to_expand', 'to_expand_abrev', 'to_expand_symbols', 'to_expand_acronyms',
                   'replacements', 'patterns', 'max_repetition')
        return self.cleaner(text, **{k: v for k, v in kwargs.items() if k in allowed})

    def encode(self, text, cleaned=False):
        if not cleaned:
            text = self.clean_text(text)
        ids = [self.index[c] for c in text if c in self.index and c != _pad]
        return np.asarray(ids, dtype=np.int32)
