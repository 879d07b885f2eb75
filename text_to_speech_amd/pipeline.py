"""Batched text -> audio pipeline that keeps the mel spectrogram on the GPU between Tacotron2 and WaveGlow.

The reference bounces every sentence through the host (`outputs.mel[0, :len]` -> numpy -> vocoder,
/root/reference/models/tts/tacotron2.py:181-189) and always runs batch 1.  Here a batch of utterances is decoded
together, the padded mel batch (pad value -11, the reference's `pad_mel_value`, models/tts/waveglow.py:27) is vocoded in
one call -- the reference's own batched path does the same (`mel.shape[0] > 1` -> direct inference, waveglow.py:108-112)
-- and only the final waveforms are copied back.  This is BASELINE.json config 3's shape (batch 8, mixed lengths).
"""
from __future__ import annotations

import numpy as np

PAD_MEL_VALUE = -11.0


class TTSPipeline:
    def __init__(self, engine, seed=None, vocoder_precision='f32', synthesizer_precision='f32', rank=None):
        from .runtime import rank_stream
        self.engine = engine
        self.vocoder_precision = vocoder_precision      # 'f16': BASELINE.json configs 3 / 5
        self.synthesizer_precision = synthesizer_precision
        self._rng = np.random.default_rng(seed)
        self._seed = int(seed) if seed is not None else int(np.random.SeedSequence().generate_state(2, np.uint32).view(np.uint64)[0])
        # one job-wide seed on every rank: each rank draws from its own key (ranks synthesize different shards; with one key
        # they would all start at offset 0 and give different utterances the same noise)
        if rank is None:
            import torch.distributed as dist
            rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
        self._seed = rank_stream(self._seed, rank)
        self._offset = 0                                # running block offset in the engine's device-side Philox stream

    def synthesize_tokens(self, tokens, speaker=None, max_length=10.0, deterministic=False, prenet_masks=None, z=None,
                          sigma=1.0, early_stopping=True, round_frames_to=8, on_device=False):
        """tokens int32 [B, Tin] (0 = pad) -> (list of B float32 waveforms, lengths [B] in frames, steps run).
        `on_device`: nothing is copied to the host -- returns (audio [B, S] float32 device tensor, zero beyond each row's
        samples, sample counts [B] int64 device tensor, steps run)."""
        import torch
        eng = self.engine
        dev = torch.device('cuda', eng.device)
        as_dev = lambda x, dt: (x.to(device=dev, dtype=dt) if torch.is_tensor(x)
                                else torch.as_tensor(np.asarray(x), dtype=dt).to(dev))
        tok = as_dev(tokens, torch.int32)
        B = int(tok.shape[0])
        n_tok = int((tok != 0).sum(dim=1).max())
        max_len = int(np.float32(n_tok) * np.float32(max_length)) if isinstance(max_length, float) else int(max_length)
        max_len = max(1, max_len)
        if prenet_masks is None and not deterministic:          # drawn on the device (engine's Philox stream)
            prenet_masks = eng.random_prenet_masks(B, max_len, self._seed, self._offset)
            self._offset += (B * max_len * 512 + 3) // 4
        elif prenet_masks is not None:
            prenet_masks = as_dev(prenet_masks, torch.float32)
        if speaker is not None:
            speaker = as_dev(speaker, torch.float32)
        out = eng.tacotron2_infer(tok, speaker=speaker, max_len=max_len, early_stopping=early_stopping,
                                  prenet_masks=prenet_masks, want_attention=False,
                                  precision=self.synthesizer_precision)
        lengths = out.lengths.clamp(min=0)
        steps = eng.last_steps
        T = int(lengths.max())
        if T <= 0:
            if on_device:
                return torch.zeros((B, 1), dtype=torch.float32, device=dev), torch.zeros(B, dtype=torch.int64, device=dev), steps
            return [np.zeros((0,), np.float32) for _ in range(B)], lengths.cpu().numpy(), steps
        if round_frames_to > 1:                         # keeps the WaveGlow workspace sizes (and M tiles) stable
            T = min(max_len, (T + round_frames_to - 1) // round_frames_to * round_frames_to)
        mel = out.mel[:, :T].clone()
        valid = torch.arange(T, device=dev)[None, :] < lengths[:, None]
        mel = torch.where(valid[:, :, None], mel, torch.full_like(mel, PAD_MEL_VALUE))
        if z is None and not deterministic:
            audio = eng.waveglow_infer(mel.contiguous(), sigma=sigma, precision=self.vocoder_precision, seed=self._seed,
                                       offset=self._offset)
            self._offset += (B * T * 256 + 3) // 4
        else:
            if z is not None:
                z = as_dev(z, torch.float32)[:, :T * 32]
            audio = eng.waveglow_infer(mel.contiguous(), z=z, sigma=sigma, precision=self.vocoder_precision)
        if on_device:
            counts = lengths.to(torch.int64) * 256
            keep = torch.arange(T * 256, device=dev)[None, :] < counts[:, None]
            return torch.where(keep, audio, torch.zeros_like(audio)), counts, steps
        audio_h = audio.cpu().numpy()
        n = lengths.cpu().numpy()
        return [audio_h[b, :int(n[b]) * 256].copy() for b in range(B)], n, steps

    def shard_fn(self, **kwargs):
        """`synth_fn(local_tokens, local_speaker) -> (audio [n, S] zero padded, sample counts [n])` for
        `distributed.synthesize_sharded`: this rank's share of the utterances through `synthesize_tokens(**kwargs)`.  Both
        results are DEVICE tensors (the waveforms go from WaveGlow's output buffer straight into the RCCL gather; the only
        device-to-host copy of the job is rank 0's, after the gather)."""
        def synth(local_tokens, local_speaker):
            audio, counts, _ = self.synthesize_tokens(local_tokens, speaker=local_speaker, on_device=True, **kwargs)
            return audio, counts
        return synth
