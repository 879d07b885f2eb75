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
    regex = _abbrev_re.get(_tables_lang(lang) if lang == 'be' else lang)
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


# Belgian French (lang 'be'): the reference rewrites num2words' French output (numbers.py:101-131): "soixante-dix…" becomes
# "septante …" and "quatre-vingt-dix…" "nonante …" -- with a SPACE before the unit ("septante deux", "nonante et un").
# Same rewriting here on the output of the French speller: (French teen stem, Belgian unit stem), longest stems first.
_BE_TENS = (('soixante-', 'septante'), ('quatre-vingt-', 'nonante'))
_BE_TEENS_CARDINAL = (('onze', ' et un'), ('douze', ' deux'), ('treize', ' trois'), ('quatorze', ' quatre'),
                      ('quinze', ' cinq'), ('seize', ' six'), ('dix-sept', ' sept'), ('dix-huit', ' huit'),
                      ('dix-neuf', ' neuf'), ('dix', ''))
_BE_TEENS_ORDINAL = (('onz', ' et un'), ('douz', ' deux'), ('treiz', ' trois'), ('quatorz', ' quatre'),
                     ('quinz', ' cinqu'), ('seiz', ' six'), ('dix-sept', ' sept'), ('dix-huit', ' huit'),
                     ('dix-neuv', ' neuv'))


def _to_belgian(words: str, ordinal: bool) -> str:
    for prefix, new in _BE_TENS:
        if prefix not in words:
            continue
        for teen, unit in (_BE_TEENS_ORDINAL if ordinal else _BE_TEENS_CARDINAL):
            words = words.replace(prefix + teen, new + unit)
        if ordinal:
            words = words.replace(prefix + 'dix', new[:-1])       # "soixante-dixième" -> "septantième"
    return words


def _cardinal(n, lang):
    if lang == 'en':
        return number_to_words(int(n))
    words = number_to_words_fr(int(n))
    return _to_belgian(words, False) if lang == 'be' else words


def _ordinal(n, lang):
    if lang == 'en':
        return ordinal_to_words(int(n))
    words = ordinal_to_words_fr(int(n))
    return _to_belgian(words, True) if lang == 'be' else words


def _tables_lang(lang):
    """'be' shares every word table with 'fr' (numbers.py:22-35 repeats the French entries under 'be')."""
    return 'fr' if lang == 'be' else lang


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
    lang = _tables_lang(lang)                                   # numbers.py:137 maps 'be' to 'fr' here
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
        word = _time_words[unit][_tables_lang(lang)]
        if int(t) > 1:
            word += 's'
        elif lang == 'fr' and int(t) == 1:                      # (sic) not for 'be': "1 sec" -> "un seconde" there
            t = 'une'
        out.append('{} {}'.format(t, word))
    return _time_sep[_tables_lang(lang)].join(out)


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
    return '{} {} {} {}'.format(_cardinal(n, lang), _math_words['*'][_tables_lang(lang)], _cardinal(0, lang), words)


def _expand_number(m, lang):
    num = m.group(0)
    if '.' not in num:
        return _cardinal(num, lang)
    ent, dec = num.split('.')
    if dec.count('0') == len(dec):
        return _cardinal(ent, lang)
    return '{} {} {}'.format(_cardinal(ent, lang), _comma_word.get(lang, ''), _extend_with_zeros(dec, lang))   # no 'be' entry


def normalize_numbers(text: str, lang: str = 'en', expand_symbols: bool = True) -> str:
    """Restatement of numbers.py:249-271, same order of substitutions."""
    if expand_symbols:
        text = _units_re.sub(lambda m: _expand_units(m, lang), text)
        text = _math_symbol_re.sub(lambda m: ' ' + ' '.join(_math_words[s][_tables_lang(lang)]
                                                            for s in m.group(0).split()) + ' ', text)
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
    lang = _tables_lang(lang)
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
                      max_repetition=-1, convert_to_ascii=None, **_) -> str:
    """Restatement of cleaners.py:296-342, same order of steps and the same result, including two quirks of that code:
    the English branch assigns the ASCII-folded text to `lang` instead of `text` (cleaners.py:336), so English text is NOT
    folded (an 'é' later falls out of the vocabulary instead of becoming 'e'), and white space is collapsed but not
    stripped (the tokenizer strips, or not, per its `lstrip` / `rstrip`).  `convert_to_ascii=True` opts into the folding
    the reference evidently intended; `to_expand_acronyms` (accepted but unused by the reference) spells acronyms here."""
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
        text = normalize_numbers(text, lang, expand_symbols=to_expand_symbols)
        if to_expand_symbols:
            text = expand_special_symbols(text, lang)
    if lang in ('fr', 'be'):
        text = re.sub(r'(ï)', 'hi', re.sub('(aï)\b', 'aille', text))               # expand_tremas (sic: '\b' is a backspace there)
        keep = set(_accents_kept)
        text = ''.join(c if c in keep else _to_ascii(c) for c in text)
    elif convert_to_ascii:
        text = _to_ascii(text)
    if max_repetition > 1:
        text = collapse_repetitions(text, max_repetition)
    return re.sub(r'\s+', ' ', text)


_accents_kept = 'âéèêîç'                                            # cleaners.py:52


def english_cleaners(text: str, **kwargs) -> str:
    return complete_cleaners(text, 'en', **kwargs)


def french_cleaners(text: str, **kwargs) -> str:
    return complete_cleaners(text, 'fr', **kwargs)


def belgian_cleaners(text: str, **kwargs) -> str:
    return complete_cleaners(text, 'be', **kwargs)


# ---- sentence / chunk splitting (behaviour of utils/text/text_processing.py:20-31, :34-143, :145-226, :228-279, :388-391;
#      pinned by tests/golden/text_vectors.json: the reference's own test cases plus outputs of the reference functions) ----
# A sentence ends at: a blank line; an ellipsis, '?' or '!' (with the spaces that follow); a dot followed by white space
# unless the dot closes a one-letter abbreviation ("e.g.", "M.H.C.P."); a line break in front of a list marker, a digit or
# a capital.  Terminators stay attached to the sentence they end.
EOS_RULES = (
    '\n\n',
    r'\.\.\.\s*', r'\?\s*', r'\!\s*',
    r'(?<!\.[a-zA-Z]{1})\.\s+',
    r'\n(?=\s*[-\*\dA-Z])',
)
# second level (inside an overlong sentence): a comma not inside a number, a colon, a parenthesised group
SUBSENTENCE_RULES = (r',(?!\d)', ': ', r'\(.*\)')
_OPENER_OF = {')': '(', ']': '[', '}': '{', '"': '"', "'": "'", '`': '`'}


def _compile_rules(rules):
    if isinstance(rules, str):
        rules = (rules,)
    return re.compile('|'.join(r if '\\' in r else re.escape(r) for r in rules))     # plain strings are literals


def _bodies_and_separators(text, regex):
    """[(body, separator)]: the text between two matches and the match that follows it (None after the last body)."""
    out, pos = [], 0
    for m in regex.finditer(text):
        if m.end() == m.start():
            continue                                            # (no rule can match empty; guard against user patterns)
        out.append((text[pos:m.start()], m.group(0)))
        pos = m.end()
    out.append((text[pos:], None))
    return out


def _only_closes_what_was_opened(previous, body):
    """True if the first word of `body` consists of closing quotes / brackets whose openers occur in `previous`: the
    terminator sat inside a quotation (`She said "Hello !"`), so `body` still belongs to the previous sentence."""
    words = body.split()
    return bool(words) and all(c in _OPENER_OF and _OPENER_OF[c] in previous for c in words[0])


def split_sentences(text: str, eos_pattern=EOS_RULES, strip: bool = False):
    """`text` -> list of sentences, each keeping its terminator and the white space behind it (`strip` removes the
    spaces, not the newlines).  Numbered headings ("1. First item", "1.2.3. Title") stay in one piece."""
    segments = _bodies_and_separators(text.strip(), _compile_rules(eos_pattern))
    sentences, k = [], 0
    while k < len(segments):
        body = segments[k][0]
        if sentences and _only_closes_what_was_opened(sentences[-1], body):
            sentences[-1] += body
        elif body.strip():
            # an all-digit body in front of a dot is an enumeration marker: glue what follows (repeatedly: "1. 2. text")
            while k + 1 < len(segments) and segments[k][0].isdigit() and segments[k][1].strip() == '.':
                body += segments[k][1] + segments[k + 1][0]
                k += 1
            sentences.append(body)
        separator = segments[k][1]
        if separator is not None and sentences:
            sentences[-1] += separator
        k += 1
    return [s.strip(' ') for s in sentences] if strip else sentences


def merge_texts(texts, max_length, max_overlap=0, max_overlap_len=0.2, *, tokens=None, tokenizer=None, **_):
    """Greedy packing of consecutive parts into chunks of at most `max_length` tokens (characters by default); parts are
    stripped of spaces and joined by one space.  `max_overlap` > 0 repeats up to that many trailing parts of the previous
    chunk (at most `max_overlap_len` tokens) at the start of the next.  Returns (chunks, chunk tokens, part indices)."""
    if isinstance(max_overlap_len, float):
        max_overlap_len = int(max_overlap_len * max_length)
    if tokenizer is None:
        tokenizer = list
    elif hasattr(tokenizer, 'tokenize'):
        tokenizer = tokenizer.tokenize
    if tokens is None:
        tokens = [tokenizer(t) for t in texts]
    texts = [t.strip(' ') for t in texts]
    groups, size = [], 0                                          # group = list of part indices
    for i, tok in enumerate(tokens):
        if groups and size + len(tok) <= max_length:
            groups[-1].append(i)
            size += len(tok)
            continue
        group, size = [i], len(tok)
        if groups and max_overlap > 0 and len(tok) < max_length:
            budget, used = min(max_overlap_len, max_length - len(tok)), 0
            for j in reversed(groups[-1][-max_overlap:]):
                if used + len(tokens[j]) > budget:
                    break
                group.insert(0, j)
                used += len(tokens[j])
            size += used
        groups.append(group)
    chunks = [' '.join(texts[i] for i in g) for g in groups]
    chunk_tokens = [[tk for i in g for tk in tokens[i]] for g in groups]
    return chunks, chunk_tokens, groups


def split_text(text: str, max_length: int, *, tokens=None, tokenizer=None, eos_pattern=EOS_RULES,
               sent_pattern=SUBSENTENCE_RULES, tolerance=0, sent_tolerance=0, merge=True, err_mode='skip',
               return_tokens=False, **kwargs):
    """Chunks of at most `max_length` tokens (characters by default), cut at the coarsest boundary that fits: sentences
    first, then sub-sentences (commas, colons, parentheses), then words; consecutive pieces are merged back up to the
    limit.  Like the reference, the first sentence is kept whole whatever its length, and a single word that is still
    too long is dropped with a warning (`err_mode`: 'skip' | 'ignore' | 'keep' | 'error')."""
    if tokenizer is None:
        tokenizer = list
    elif hasattr(tokenizer, 'tokenize'):
        tokenizer = tokenizer.tokenize
    if isinstance(tolerance, float):
        tolerance = int(tolerance * max_length)
    if isinstance(sent_tolerance, float):
        sent_tolerance = int(sent_tolerance * max_length)
    text_limit, sent_limit = max_length + tolerance, max_length + sent_tolerance
    if tokens is None:
        tokens = tokenizer(text)
    if len(tokens) <= text_limit:
        return ([text], [tokens]) if return_tokens else [text]

    pieces = split_sentences(text, eos_pattern, strip=False)
    piece_tokens = [tokenizer(p) for p in pieces]
    out_text, out_tokens = pieces[:1], piece_tokens[:1]
    for piece, tok in zip(pieces[1:], piece_tokens[1:]):
        if len(tok) <= sent_limit:
            out_text.append(piece)
            out_tokens.append(tok)
        elif sent_pattern:
            finer = ' ' if sent_pattern != ' ' else None            # sub-sentences -> words -> give up
            sub_text, sub_tokens = split_text(piece, sent_limit, tokens=tok, tokenizer=tokenizer, eos_pattern=sent_pattern,
                                              sent_pattern=finer, err_mode=err_mode, return_tokens=True)
            out_text.extend(sub_text)
            out_tokens.extend(sub_tokens)
        elif err_mode == 'error':
            raise RuntimeError('It was not possible to split `{}`'.format(piece))
        elif err_mode == 'keep':
            out_text.append(piece)
            out_tokens.append(tok)
        elif err_mode == 'skip':
            import warnings
            warnings.warn('The text `{}` is skipped as it is too long'.format(piece))
    if merge:
        out_text, out_tokens, _ = merge_texts(out_text, text_limit, tokens=out_tokens, tokenizer=tokenizer, **kwargs)
    return (out_text, out_tokens) if return_tokens else out_text


class CharTokenizer:
    """Character tokenizer of the TTS models (utils/text/tokenizer.py:53, :345-352, :394): cleaners, then one id per
    character of the symbol table (unknown characters are dropped).  `lstrip` / `rstrip` default to False like the
    reference's `Tokenizer`, so a sentence keeps the space that `split_sentences` leaves behind its terminator."""

    def __init__(self, lang='en', lstrip=False, rstrip=False):
        self.lang = lang
        self.lstrip, self.rstrip = lstrip, rstrip
        self.symbols = en_symbols if lang == 'en' else fr_symbols
        self.index = {s: i for i, s in enumerate(self.symbols)}
        self.cleaner = {'en': english_cleaners, 'fr': french_cleaners, 'be': belgian_cleaners}.get(lang, french_cleaners)
        self.cleaners = [self.cleaner.__name__]
        self.pad_token = _pad
        self.sos_token = self.eos_token = self.ukn_token = None
        self.use_sos_and_eos = False

    # ---- the shipped models carry their own vocabulary: `<model>/saving/tokenizer.json` = Tokenizer.get_config()
    #      (utils/text/tokenizer.py:661-702); a fresh model takes get_symbols(lang) instead (utils/text/__init__.py:93-96)
    @classmethod
    def from_config(cls, config, lang=None):
        """Builds the tokenizer from a reference `Tokenizer` config (character level only).  Ids follow
        `Tokenizer.__build_indexes` (tokenizer.py:183-213): the vocabulary in order, the special tokens that are in use
        (sep / ukn, sos / eos when `use_sos_and_eos`, additional tokens) before or after it; cleaners by name, with their
        keyword arguments (`{'name': 'french_cleaners', 'to_lowercase': False}` or `(name, kwargs)`)."""
        level = config.get('level', 0)
        if str(level).lower() not in ('0', 'char', 'tokenizerlevel.char'):
            raise ValueError(f'only character-level tokenizers are supported (the TTS models), got level {level!r}')
        if config.get('bpe_pairs') or config.get('split_pattern'):
            raise ValueError('BPE / pattern-split tokenizers are not part of the TTS path')
        self = cls.__new__(cls)
        self.lstrip, self.rstrip = bool(config.get('lstrip', False)), bool(config.get('rstrip', False))
        self.pad_token = config.get('pad_token', '')
        self.sos_token, self.eos_token = config.get('sos_token'), config.get('eos_token')
        self.ukn_token = config.get('ukn_token')
        self.use_sos_and_eos = bool(config.get('use_sos_and_eos', False))
        specials = [config.get(k) for k in ('sep_token', 'ukn_token') if config.get(k) is not None]
        if self.use_sos_and_eos:
            specials += [t for t in (self.sos_token, self.eos_token) if t is not None]
        extra = config.get('additional_tokens') or []
        specials += list(extra.values()) if isinstance(extra, dict) else ([extra] if isinstance(extra, str) else list(extra))
        vocab = list(config['vocab'])
        order = (vocab + specials) if config.get('add_special_tokens_at_end', True) else (specials + vocab)
        self.index = {}
        for sym in order:
            self.index.setdefault(sym, len(self.index))
        self.symbols = sorted(self.index, key=self.index.get)
        known = {'english_cleaners': english_cleaners, 'french_cleaners': french_cleaners, 'belgian_cleaners': belgian_cleaners,
                 'complete_cleaners': complete_cleaners}
        self.cleaners, fns = list(config.get('cleaners') or []), []
        for c in self.cleaners if isinstance(self.cleaners, (list, tuple)) else [self.cleaners]:
            name, kw = (c, {}) if isinstance(c, str) else (c[0], dict(c[1] or {})) if isinstance(c, (list, tuple)) \
                else (c['name'], {k: v for k, v in c.items() if k != 'name'})
            if name not in known:
                raise ValueError('Unknown cleaner : {}'.format(name))
            fns.append((known[name], kw))
        names = ' '.join(n.__name__ for n, _ in fns)
        self.lang = lang or ('en' if 'english' in names else 'be' if 'belgian' in names else 'fr' if 'french' in names
                             else fns[0][1].get('lang', 'en') if fns else 'en')

        def run(text, **kwargs):
            for fn, kw in fns:
                args = {**kw, **kwargs}
                if fn is complete_cleaners:
                    args.setdefault('lang', self.lang)
                    text = fn(text, args.pop('lang'), **args)
                else:
                    text = fn(text, **args)
            return text
        self.cleaner = run
        return self

    @classmethod
    def load_from_file(cls, filename, lang=None):
        import json
        with open(filename, encoding='utf-8') as fh:
            return cls.from_config(json.load(fh), lang=lang)

    def get_config(self):
        return {'name': 'Tokenizer', 'vocab': [s for s in self.symbols], 'level': 0, 'lstrip': self.lstrip, 'rstrip': self.rstrip,
                'cleaners': self.cleaners, 'pad_token': self.pad_token, 'sos_token': self.sos_token, 'eos_token': self.eos_token,
                'ukn_token': self.ukn_token, 'use_sos_and_eos': self.use_sos_and_eos, 'add_special_tokens_at_end': True}

    def save(self, filename):
        import json
        with open(filename, 'w', encoding='utf-8') as fh:
            json.dump(self.get_config(), fh, indent=4, ensure_ascii=False)
        return filename

    @property
    def vocab_size(self):
        return len(self.symbols)

    @property
    def blank_token_idx(self):
        """Id of the padding symbol (utils/text/tokenizer.py `blank_token_idx`); the encoder masks on this id (0 in every
        symbol table of the TTS models: '_' comes first)."""
        return self.index.get(self.pad_token, 0)

    def clean_text(self, text, **kwargs):
        allowed = ('to_lowercase', 'to_expand', 'to_expand_abrev', 'to_expand_symbols', 'to_expand_acronyms',
                   'replacements', 'patterns', 'max_repetition', 'convert_to_ascii')
        text = self.cleaner(text, **{k: v for k, v in kwargs.items() if k in allowed})
        if self.lstrip:
            text = text.lstrip()
        if self.rstrip:
            text = text.rstrip()
        return text

    def encode(self, text, cleaned=False):
        if not cleaned:
            text = self.clean_text(text)
        ukn = self.index.get(self.ukn_token, -1) if self.ukn_token is not None else -1
        ids = [self.index.get(c, ukn) for c in text if c != self.pad_token]
        ids = [i for i in ids if i != -1]                        # unknown characters are dropped unless there is an ukn token
        if self.use_sos_and_eos:                                 # tokenizer.py:447-450
            if self.sos_token in self.index and (not ids or ids[0] != self.index[self.sos_token]):
                ids.insert(0, self.index[self.sos_token])
            if self.eos_token in self.index and (not ids or ids[-1] != self.index[self.eos_token]):
                ids.append(self.index[self.eos_token])
        return np.asarray(ids, dtype=np.int32)
