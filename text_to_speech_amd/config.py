"""Frozen hyper-parameter sets for the Tacotron2 + WaveGlow inference path.

The values are the reference's defaults (= the shapes of the NVIDIA checkpoints):
  * WaveGlow ctor defaults          -- reference architectures/waveglow_arch.py:164-178
  * HParamsTacotron2{Encoder,Prenet,Postnet,Decoder} -- architectures/tacotron2_arch.py:59-135
  * HParamsLSA                      -- architectures/layers/location_sensitive_attention.py:17-24
  * TacotronSTFT / MelSTFT defaults -- utils/audio/stft.py:27-60,286-305
"""
from __future__ import annotations

from dataclasses import dataclass, asdict


@dataclass(frozen=True)
class WaveGlowConfig:
    n_mel_channels: int = 80
    n_flows: int = 12
    n_group: int = 8
    n_early_every: int = 4
    n_early_size: int = 2
    n_layers: int = 8
    n_channels: int = 512
    kernel_size: int = 3
    upsample_kernel: int = 1024
    upsample_stride: int = 256

    @property
    def n_cond(self) -> int:            # channels of the regrouped spectrogram
        return self.n_mel_channels * self.n_group

    def flow_channels(self):
        """(n_remaining_channels, n_half) for flow k = 0 .. n_flows-1.

        Follows the constructor loop at waveglow_arch.py:202-223.
        """
        out = []
        n_half = self.n_group // 2
        n_rem = self.n_group
        for k in range(self.n_flows):
            if k % self.n_early_every == 0 and k > 0:
                n_half -= self.n_early_size // 2
                n_rem -= self.n_early_size
            out.append((n_rem, n_half))
        return out

    @property
    def n_remaining_channels(self) -> int:
        return self.flow_channels()[-1][0]

    def to_dict(self):
        return asdict(self)


@dataclass(frozen=True)
class Tacotron2Config:
    vocab_size: int = 148
    pad_token: int = 0
    # encoder
    embedding_dim: int = 512
    encoder_n_conv: int = 3
    encoder_kernel_size: int = 5
    bn_epsilon: float = 1e-5
    speaker_embedding_dim: int = 0          # 0: single speaker; 256: SV2TTS ('concat' at 'end')
    # prenet
    prenet_sizes: tuple = (256, 256)
    prenet_drop_rate: float = 0.5
    # decoder
    n_mel_channels: int = 80
    attention_rnn_dim: int = 1024
    decoder_rnn_dim: int = 1024
    # location sensitive attention
    attention_dim: int = 128
    attention_filters: int = 32
    attention_kernel_size: int = 31
    # postnet
    postnet_n_conv: int = 5
    postnet_filters: int = 512
    postnet_kernel_size: int = 5

    @property
    def encoder_dim(self) -> int:
        """Width of the encoder output (`enc` in SURVEY.md section 8)."""
        return self.embedding_dim + self.speaker_embedding_dim

    def to_dict(self):
        d = asdict(self)
        d['prenet_sizes'] = list(self.prenet_sizes)
        return d


@dataclass(frozen=True)
class MelSTFTConfig:
    sampling_rate: int = 22050
    n_mel_channels: int = 80
    filter_length: int = 1024
    hop_length: int = 256
    win_length: int = 1024
    mel_fmin: float = 0.0
    mel_fmax: float = 8000.0
    clip_val: float = 1e-5

    def to_dict(self):
        return asdict(self)
