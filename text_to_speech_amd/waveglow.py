"""Host-side WaveGlow wrapper: the argument / return contract of the reference's `models.tts.WaveGlow.infer`.

Same behaviour as /root/reference/models/tts/waveglow.py:61-142 (`infer`: path / 2-D / 3-D mel input, `audio_len = T * 256`,
optional pad-to-`win_len` with -11, window / hop chunking with centre-half stitching) and :156-164 (`_get_steps`),
pinned by outputs of the reference's own `_get_steps` and stitch statements (tests/golden/host_vectors.json,
tests/test_host_vectors.py) and the known-answer tests in tests/test_host_logic.py.
The compute call (`self.compiled_infer`) is a `HipRuntime`; nothing here touches a CPU fallback.
"""
from __future__ import annotations

import math

import numpy as np

from .engine import _is_torch_cuda


def window_starts(length, win_len, hop_len):
    """First frames of the analysis windows: the fewest windows of `win_len` frames, at most `hop_len` apart, that cover
    `length` frames, re-spaced evenly so that the last one ends exactly at `length` (the rule of
    models/tts/waveglow.py:156-164).  E.g. (1000, 256, 192) -> 0, 186, 372, 558, 744."""
    n = 1 + max(0, int(math.ceil((length - win_len) / hop_len)))
    if n == 1:
        return [0]
    spacing = (length - win_len) / (n - 1)
    return np.round(spacing * np.arange(n)).astype(np.int32)


_get_steps = window_starts          # the reference's name for it


def _to_numpy(x):
    return x.detach().cpu().numpy() if hasattr(x, 'detach') else np.asarray(x)


# ---- exact time tiling ------------------------------------------------------------------------------------------------
# WaveGlow inference is feed-forward with a finite receptive field: per flow the 8 dilated k=3 convs reach
# 1 + 2 + ... + 128 = 255 groups (of 8 samples) to each side, 12 flows -> 3060 groups = 95.6 mel frames, plus the 4-frame
# window of the upsampling transposed conv.  A tile computed from its frames plus HALO_FRAMES of
# real context on each side therefore reproduces the full-sequence samples of its core region exactly; only samples whose
# receptive field crosses an artificial tile edge differ, and those all lie inside the discarded halo.  (The reference's
# own windowing, `infer(win_len=...)` below, is approximate: it discards half of a 64-frame overlap.)
HALO_FRAMES = 100


def tile_plan(n_frames, tile_frames, halo=HALO_FRAMES):
    """[(start, stop, lo, hi)]: core frames [start, stop) are taken from a run over mel[lo:hi]."""
    if tile_frames <= 0:
        raise ValueError('tile_frames must be positive')
    plan = []
    for start in range(0, n_frames, tile_frames):
        stop = min(n_frames, start + tile_frames)
        plan.append((start, stop, max(0, start - halo), min(n_frames, stop + halo)))
    return plan


def infer_tiled(compiled_infer, mel, z=None, tile_frames=4096, halo=HALO_FRAMES, tiles=None, **kwargs):
    """Exact vocoding of a long mel [B, T, 80] in time tiles; `tiles` (indices into tile_plan) restricts the work to a
    subset (multi-GPU sharding of one utterance) -- the result then is a list of (start, stop, audio [B, (stop-start)*256])."""
    T = mel.shape[1]
    plan = tile_plan(T, tile_frames, halo)
    chosen = range(len(plan)) if tiles is None else tiles
    out = []
    for i in chosen:
        start, stop, lo, hi = plan[i]
        zz = None if z is None else z[:, lo * 32:hi * 32]
        if _is_torch_cuda(mel):
            sub = mel[:, lo:hi].contiguous()
            zz = None if zz is None else zz.contiguous()
        else:
            sub = np.ascontiguousarray(mel[:, lo:hi])
        a = compiled_infer(sub, z=zz, **kwargs)
        out.append((start, stop, a[:, (start - lo) * 256:(stop - lo) * 256]))
    if tiles is not None:
        return out
    parts = [_to_numpy(a) for _, _, a in out]
    return np.concatenate(parts, axis=1)


class WaveGlow:
    """`vocoder(mel, **kwargs)` object accepted by `Tacotron2.infer(..., vocoder=...)`: the argument / return contract of
    the reference's `models.tts.WaveGlow.infer` (models/tts/waveglow.py:61-142).

    mel: a `.npy` path, [T, 80] or [B, T, 80] -> audio [B, T * 256] (windowed single-utterance mode: [T * 256]).
    Without `win_len` the whole mel is vocoded in one call.  With it (frames; a float means "a multiple of": rounded up,
    or down with `use_slice`; capped by `max_win_len`):
      * a mel that fits one window is vocoded directly -- padded to the window with -11 first only if `force_pad`
        (default: only for the keras runtime, i.e. never here), the result cut back to T * 256 samples;
      * a batch is vocoded directly;
      * a longer utterance is cut into overlapping windows (`hop_len` frames apart; negative = window minus that many,
        float = fraction of the window), each vocoded alone or all as one batch (`batch=True`), and stitched by dropping
        half of every overlap from each side.  This is the reference's approximation (the vocoder's receptive field is
        longer than the half-overlap); `infer_exact` is the exact alternative."""
    rate = 22050
    pad_mel_value = -11.
    runtime = 'hip'

    def __init__(self, compiled_infer):
        self.compiled_infer = compiled_infer

    def _resolve_window(self, n_frames, win_len, use_slice, max_win_len):
        if isinstance(win_len, float):
            count = max(1, n_frames // win_len) if use_slice else math.ceil(n_frames / win_len)
            win_len = int(count * win_len) if not use_slice else int(count) * int(win_len)
        return int(win_len if max_win_len is None else min(max_win_len, win_len))

    def _pad_frames(self, mel, n_frames):
        extra = n_frames - mel.shape[1]
        if _is_torch_cuda(mel):
            import torch
            return torch.nn.functional.pad(mel, (0, 0, 0, extra), value=self.pad_mel_value)
        return np.pad(np.asarray(mel), [(0, 0), (0, extra), (0, 0)], constant_values=self.pad_mel_value)

    def infer(self, mel, *, win_len=None, hop_len=-64, force_pad=None, batch=False, use_slice=False,
              max_win_len=None, **kwargs):
        if isinstance(mel, str):
            mel = np.load(mel)
        if len(mel.shape) == 2:
            mel = mel[None]
        n_frames = mel.shape[1]
        n_samples = n_frames * 256
        if win_len is None:
            return self.compiled_infer(mel, **kwargs)[:, :n_samples]

        win_len = self._resolve_window(n_frames, win_len, use_slice, max_win_len)
        kwargs['padding_multiple'] = win_len
        if n_frames <= win_len:
            if force_pad is None:
                force_pad = self.runtime == 'keras'               # waveglow.py:95 -- False for this runtime
            if not force_pad:
                return self.compiled_infer(mel)
            return self.compiled_infer(self._pad_frames(mel, win_len), **kwargs)[:, :n_samples]
        if mel.shape[0] > 1:
            return self.compiled_infer(mel, **kwargs)

        if isinstance(hop_len, float):
            hop_len = int(win_len * hop_len)
        if hop_len < 0:
            hop_len += win_len
        starts = np.asarray(window_starts(n_frames, win_len, hop_len))
        windows = [mel[:, s0:s0 + win_len] for s0 in starts]
        if batch:
            if _is_torch_cuda(mel):
                import torch
                stacked = torch.cat(windows, dim=0)
            else:
                stacked = np.concatenate([np.asarray(w) for w in windows], axis=0)
            pieces = list(_to_numpy(self.compiled_infer(stacked, **kwargs)))
        else:
            pieces = [_to_numpy(self.compiled_infer(w, **kwargs)[0]) for w in windows]
        # consecutive windows share (end of k) - (start of k + 1) frames: each gives up half of those samples
        shared = (starts[:-1] + win_len - starts[1:]) * 256
        head = np.concatenate([[0], shared // 2])
        tail = np.concatenate([-(-shared // 2), [0]])            # the reference slices `[: -overlap // 2]` = ceil half
        last = len(pieces) - 1
        # (sic) a window that shares NO frame with its successor (hop_len == win_len) is sliced `part[start : -0 // 2]` =
        # `part[start : 0]` by the reference (models/tts/waveglow.py:136-139): empty.  Kept, because the contract here is the
        # reference's output on the same arguments (pinned by tests/golden/host_vectors.json); overlapping windows -- every
        # default -- are seamless.
        ends = [len(p) - t if (i == last or t > 0) else 0 for i, (p, t) in enumerate(zip(pieces, tail))]
        return np.concatenate([p[h:e] for p, h, e in zip(pieces, head, ends)], axis=-1)

    __call__ = infer

    def infer_exact(self, mel, *, tile_frames=4096, z=None, **kwargs):
        """Long-form vocoding without the approximation of `infer(win_len=...)`: halo tiling, bit-for-bit the samples of
        one run over the whole mel (which one C-ABI call only accepts up to ~31.7 k frames)."""
        if isinstance(mel, str):
            mel = np.load(mel)
        if len(mel.shape) == 2:
            mel = mel[None]
        return infer_tiled(self.compiled_infer, mel, z=z, tile_frames=tile_frames, **kwargs)
