"""Host-side WaveGlow wrapper: the argument / return contract of the reference's `models.tts.WaveGlow.infer`.

Restates /root/reference/models/tts/waveglow.py:61-142 (`infer`: path / 2-D / 3-D mel input, `audio_len = T * 256`,
optional pad-to-`win_len` with -11, window / hop chunking with centre-half stitching) and :156-164 (`_get_steps`).
The compute call (`self.compiled_infer`) is a `HipRuntime`; nothing here touches a CPU fallback.
"""
from __future__ import annotations

import math

import numpy as np

from .engine import _is_torch_cuda


def _get_steps(length, win_len, hop_len):
    """Window starts, evenly re-spaced so the last window ends at `length` (waveglow.py:156-164)."""
    num_steps = int(math.ceil((length - win_len) / hop_len)) + 1
    if num_steps == 1:
        return [0]
    max_step = length - win_len
    actual_step_size = max_step / (num_steps - 1)
    return np.round(np.arange(num_steps) * actual_step_size).astype(np.int32)


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
    """`vocoder(mel, **kwargs)` object accepted by `Tacotron2.infer(..., vocoder=...)`."""
    rate = 22050
    pad_mel_value = -11.
    runtime = 'hip'

    def __init__(self, compiled_infer):
        self.compiled_infer = compiled_infer

    def infer(self, mel, *, win_len=None, hop_len=-64, force_pad=None, batch=False, use_slice=False,
              max_win_len=None, **kwargs):
        if isinstance(mel, str):
            mel = np.load(mel)
        if len(mel.shape) == 2:
            mel = mel[None]
        seq_len = mel.shape[1]
        audio_len = seq_len * 256
        if win_len is None:
            return self.compiled_infer(mel, **kwargs)[:, :audio_len]

        if isinstance(win_len, float):
            if not use_slice:
                win_len = int(math.ceil(seq_len / win_len) * win_len)
            else:
                win_len = max(1, seq_len // win_len) * int(win_len)
        if max_win_len is not None:
            win_len = min(max_win_len, win_len)
        kwargs['padding_multiple'] = win_len

        if seq_len <= win_len:
            if force_pad is None:
                force_pad = self.runtime == 'keras'         # False for this runtime (waveglow.py:95)
            if not force_pad:
                return self.compiled_infer(mel)
            win_len = max(win_len, seq_len)
            pad = [(0, 0), (0, win_len - seq_len), (0, 0)]
            if _is_torch_cuda(mel):
                import torch
                padded = torch.nn.functional.pad(mel, (0, 0, 0, win_len - seq_len), value=self.pad_mel_value)
            else:
                padded = np.pad(np.asarray(mel), pad, constant_values=self.pad_mel_value)
            return self.compiled_infer(padded, **kwargs)[:, :audio_len]
        elif mel.shape[0] > 1:
            return self.compiled_infer(mel, **kwargs)

        if isinstance(hop_len, float):
            hop_len = int(win_len * hop_len)
        if hop_len < 0:
            hop_len = win_len + hop_len

        starts = _get_steps(seq_len, win_len, hop_len)
        parts = [mel[:, start:start + win_len] for start in starts]
        starts = np.asarray(starts)
        overlaps = ((starts[:-1] + win_len) - starts[1:]) * 256

        if batch:
            if _is_torch_cuda(mel):
                import torch
                stacked = torch.cat(parts, dim=0)
            else:
                stacked = np.concatenate([np.asarray(p) for p in parts], axis=0)
            audio_parts = list(_to_numpy(self.compiled_infer(stacked, **kwargs)))
        else:
            audio_parts = [_to_numpy(self.compiled_infer(p, **kwargs)[0]) for p in parts]

        audio = []
        for i, part in enumerate(audio_parts):
            start = 0 if i == 0 else overlaps[i - 1] // 2
            end = None if i == len(audio_parts) - 1 else -overlaps[i] // 2
            audio.append(part[start:end])
        return np.concatenate(audio, axis=-1)

    __call__ = infer

    def infer_exact(self, mel, *, tile_frames=4096, z=None, **kwargs):
        """Long-form vocoding without the approximation of `infer(win_len=...)`: halo tiling, bit-for-bit the samples of
        one run over the whole mel (which one C-ABI call only accepts up to ~31.7 k frames)."""
        if isinstance(mel, str):
            mel = np.load(mel)
        if len(mel.shape) == 2:
            mel = mel[None]
        return infer_tiled(self.compiled_infer, mel, z=z, tile_frames=tile_frames, **kwargs)
