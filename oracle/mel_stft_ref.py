"""numpy restatement of the reference TacotronSTFT mel spectrogram (TEST INFRASTRUCTURE -- see oracle/__init__.py).

PINNED by the reference's golden pair (tests/golden/stft_tacotron_fixture.npz, cut from the reference's
tests/__reproduction/audio_resample.npy -> stft-TacotronSTFT.npy, tolerance 2e-3 at test_utils_audio.py:110-112).

Follows /root/reference/utils/audio/stft.py:
  STFT basis (windowed DFT rows)       stft.py:194-236
  STFT.transform (reflect pad, conv)   stft.py:242-274
  TacotronSTFT.mel_spectrogram         stft.py:306-314   log(max(mag @ mel_basis.T, 1e-5))
  MelSTFT.__call__ (short-audio pad)   stft.py:101-124
The mel basis is `librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax)` (stft.py:65-72; librosa is a third-party
dependency absent here, unpinned in requirements.txt) -- restated from its published algorithm: Slaney mel scale
(htk=False), triangular filters, Slaney area normalisation.
"""
from __future__ import annotations

import numpy as np


def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, mels)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    freqs = f_sp * m
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), freqs)


def mel_filterbank(sr=22050, n_fft=1024, n_mels=80, fmin=0.0, fmax=8000.0):
    """Slaney-normalised mel filterbank [n_mels, 1 + n_fft//2] (float32), as librosa.filters.mel defaults."""
    fftfreqs = np.linspace(0, sr / 2.0, 1 + n_fft // 2)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    weights = np.zeros((n_mels, 1 + n_fft // 2))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    return (weights * enorm[:, None]).astype(np.float32)


def hann_periodic(n):
    """scipy.signal.get_window('hann', n, fftbins=True)."""
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n))


def forward_basis(filter_length=1024, win_length=1024):
    """Windowed real/imag DFT rows: [2 * (filter_length//2 + 1), filter_length] float32.  stft.py:211-236."""
    cutoff = filter_length // 2 + 1
    fb = np.fft.fft(np.eye(filter_length))
    fb = np.vstack([np.real(fb[:cutoff]), np.imag(fb[:cutoff])]).astype(np.float32)
    win = hann_periodic(win_length)
    if win_length < filter_length:                  # librosa.util.pad_center
        lpad = (filter_length - win_length) // 2
        win = np.pad(win, (lpad, filter_length - win_length - lpad))
    return (fb * win[None, :]).astype(np.float32)    # float32 basis *= float64 window, kept float32 (in-place op)


def mel_spectrogram(audio, cfg, dtype=np.float32):
    """TacotronSTFT()(audio): audio [N] or [B, N] -> [B, N//hop + 1, n_mels]."""
    audio = np.asarray(audio, dtype=dtype)
    if audio.ndim == 1:
        audio = audio[None]
    if audio.shape[1] < cfg.win_length:
        audio = np.pad(audio, [(0, 0), (0, cfg.win_length - audio.shape[1])])
    fl, hop = cfg.filter_length, cfg.hop_length
    x = np.pad(audio, [(0, 0), (fl // 2, fl // 2)], mode='reflect')
    n_frames = (x.shape[1] - fl) // hop + 1
    idx = np.arange(n_frames)[:, None] * hop + np.arange(fl)[None, :]
    frames = x[:, idx]                                              # [B, F, fl]
    basis = forward_basis(fl, cfg.win_length).astype(dtype)         # [2*cutoff, fl]
    ft = frames @ basis.T
    cutoff = fl // 2 + 1
    mag = np.sqrt(ft[..., :cutoff] ** 2 + ft[..., cutoff:] ** 2)
    mb = mel_filterbank(cfg.sampling_rate, fl, cfg.n_mel_channels, cfg.mel_fmin, cfg.mel_fmax).astype(dtype)
    mel = mag @ mb.T
    return np.log(np.maximum(mel, dtype(cfg.clip_val)))
