"""Speaker-embedding files of the SV2TTS model: the inference-side part of /root/reference/utils/embeddings.py.

`SV2TTSTacotron2` keeps its default voice(s) in `<model dir>/embeddings/embeddings.<ext>` and loads them with
`load_embeddings` (models/tts/sv2tts_tacotron2.py:53-67).  That file is a table -- columns `id`, `filename`, ..., `embedding` --
stored as `.csv` (vectors as their string representation), `.npy` (the bare matrix), `.pkl` / `.pdpkl` (pickled
DataFrame) or, by default, `.h5` (utils/file_utils.py:358-397: one HDF5 dataset per column, strings as variable-length
strings, ragged vectors padded with -1).  This module reads all of them without h5py (`hdf5_reader`); what it leaves out is
the training side of the reference function (merging with a dataset, rewriting file-name prefixes).
"""
from __future__ import annotations

import logging
import os

import numpy as np

logger = logging.getLogger(__name__)

EMBEDDINGS_FILE_EXT = ('.csv', '.npy', '.pkl', '.pdpkl', '.embeddings.h5', '.h5')     # utils/embeddings.py:27, tried in turn


def embeddings_to_np(embeddings, col='embedding', dtype=float):
    """A matrix (or vector) of float32 from: an array; the string form of a vector `'[0.1, 0.2]'` / `'[0.1\\t0.2]'` or of a
    matrix `'[[..] [..]]'` (utils/embeddings.py:30-75); a DataFrame / dict of columns (column `col`); a file name."""
    if isinstance(embeddings, str):
        text = embeddings.strip()
        if text.startswith('[['):
            rows = [embeddings_to_np(r.strip(' ,\n') + ']', dtype=dtype) for r in text[1:-1].split(']') if r.strip(' ,\n')]
            return _pad_rows(rows)
        if text.startswith('['):
            body = text[1:-1].replace('\n', ' ')
            sep = ',' if ',' in body else None                      # None: any whitespace (tabs, or numpy's space-separated repr)
            return np.array([dtype(x) for x in body.split(sep) if x.strip()], dtype=np.float32)
        if os.path.isfile(text):
            return embeddings_to_np(load_embeddings(text), col=col, dtype=dtype)
        raise ValueError('The file {} does not exist !'.format(embeddings))
    if isinstance(embeddings, np.ndarray) and embeddings.dtype != object:
        return embeddings
    if hasattr(embeddings, 'columns') or isinstance(embeddings, dict):
        values = embeddings[col]
        values = values.values if hasattr(values, 'values') else values
        rows = [embeddings_to_np(v, dtype=dtype) for v in values]
        return np.array(rows) if rows and rows[0].ndim == 1 and len({r.shape for r in rows}) == 1 else _pad_rows(rows)
    if hasattr(embeddings, 'detach'):
        return embeddings.detach().cpu().numpy()
    if isinstance(embeddings, (list, tuple, np.ndarray)):
        return np.asarray(embeddings, dtype=np.float32) if not len(embeddings) or not isinstance(embeddings[0], str) \
            else _pad_rows([embeddings_to_np(e, dtype=dtype) for e in embeddings])
    raise ValueError('Invalid type of embeddings : {}\n{}'.format(type(embeddings), embeddings))


def _pad_rows(rows):
    """Stack arrays of equal rank, padding every axis to the longest (utils/sequence_utils.py:16-60, pad value 0)."""
    rows = [np.asarray(r, dtype=np.float32) for r in rows]
    if not rows:
        return np.zeros((0, 0), np.float32)
    shape = np.max(np.array([r.shape for r in rows], dtype=np.int64), axis=0)
    out = np.zeros((len(rows), *shape), np.float32)
    for i, r in enumerate(rows):
        out[(i, *[slice(0, n) for n in r.shape])] = r
    return out


def _resolve(filename):
    if os.path.exists(filename):
        return filename
    for ext in EMBEDDINGS_FILE_EXT:
        if os.path.exists(filename + ext):
            return filename + ext
    return None


def _table(columns):
    """dict of equally long columns -> pandas DataFrame when pandas is there (what the reference returns), else the dict."""
    try:
        import pandas as pd
    except ImportError:                                             # pragma: no cover
        return columns
    return pd.DataFrame({k: list(v) if getattr(v, 'ndim', 1) > 1 else v for k, v in columns.items()})


def _load_h5(path):
    from .hdf5_reader import read_all
    cols = {}
    for name, arr in read_all(path).items():
        key = name.strip('/').replace('\\', '/')
        if '/' in key:                                              # nested groups: not a table the reference would write
            continue
        if arr.dtype == object:
            cols[key] = arr.tolist() if arr.ndim else arr.item()
        else:
            cols[key] = arr
    return cols


def load_embeddings(filename, *, aggregate_on='id', aggregate_mode=0, aggregate_name='speaker_embedding', **_):
    """Loads a table of embeddings (utils/embeddings.py:119-212).  Returns the matrix for `.npy`, else a DataFrame with the
    `embedding` column as float32 vectors and -- when the `aggregate_on` column exists -- one more column `aggregate_name`
    holding, for every row, the aggregate of its group (`aggregate_mode`: an int picks that member, 'mean' averages)."""
    path = _resolve(filename)
    if path is None:
        logger.warning('Embeddings file %s does not exist !', filename)
        return None
    low = path.lower()
    if low.endswith('.npy'):
        return np.load(path)
    if low.endswith(('.h5', '.hdf5')):
        data = _load_h5(path)
    elif low.endswith(('.csv', '.tsv')):
        import pandas as pd
        data = pd.read_csv(path, sep='\t' if low.endswith('.tsv') else ',')
    elif low.endswith(('.pkl', '.pdpkl')):
        import pickle
        with open(path, 'rb') as fh:
            data = pickle.load(fh)
    else:
        raise ValueError('Unsupported embeddings extension !\n  Accepted : {}\n  Got : {}'.format(EMBEDDINGS_FILE_EXT, path))
    if isinstance(data, np.ndarray):
        return data
    if isinstance(data, dict):
        data = _table(data)
    if not hasattr(data, 'columns'):
        return data
    data = data.drop(columns=[c for c in data.columns if 'Unnamed:' in str(c)])
    for c in data.columns:
        if 'embedding' in str(c):
            vectors = [embeddings_to_np(v) for v in data[c].values]
            if low.endswith(('.h5', '.hdf5')):                      # dump_h5 pads ragged vectors with -1: cut the padding off
                vectors = [v[:len(v) - int(np.argmax(v[::-1] != -1))] if v.size and v[-1] == -1 and (v != -1).any() else v
                           for v in vectors]
            data[c] = vectors
    if aggregate_on and aggregate_on in data.columns and 'embedding' in data.columns:
        groups = {}
        for key, vec in zip(data[aggregate_on].values, data['embedding'].values):
            groups.setdefault(key, []).append(vec)
        pick = (lambda vs: np.mean(np.stack(vs), axis=0)) if aggregate_mode in ('mean', 'avg', 'average') \
            else (lambda vs: vs[min(int(aggregate_mode), len(vs) - 1)])
        agg = {k: pick(v) for k, v in groups.items()}
        data[aggregate_name] = [agg[k] for k in data[aggregate_on].values]
    return data


def save_embeddings(filename, embeddings, *, directory=None):
    """`.npy` (matrix), `.csv` or `.pkl` (table); the reference's default `.h5` needs an HDF5 writer and is not offered."""
    if directory:
        filename = os.path.join(directory, filename)
    if not os.path.splitext(filename)[1]:
        filename += '.npy' if isinstance(embeddings, np.ndarray) else '.pkl'
    os.makedirs(os.path.dirname(os.path.abspath(filename)), exist_ok=True)
    if filename.endswith('.npy'):
        np.save(filename, embeddings_to_np(embeddings))
    elif filename.endswith('.csv'):
        table = embeddings.copy()
        for c in table.columns:
            if 'embedding' in str(c):
                table[c] = [np.array2string(np.asarray(v, np.float32), separator=', ', threshold=1 << 20, max_line_width=1 << 20)
                            for v in table[c].values]
        table.to_csv(filename, index=False)
    elif filename.endswith(('.pkl', '.pdpkl')):
        import pickle
        with open(filename, 'wb') as fh:
            pickle.dump(embeddings, fh)
    else:
        raise ValueError('Unsupported embeddings extension !\n  Accepted : {}\n  Got : {}'.format(('.npy', '.csv', '.pkl'), filename))
    return filename
