"""Runtime plug-in for the reference's execution seam.

Mirrors /root/reference/utils/keras/runtimes/runtime.py:19-41 (`Runtime` ABC: per-path engine cache, abstract
`__call__` and `load_engine`) and utils/keras/runtimes/__init__.py:23-45 (`build_runtime` + `_runtimes` registry), so a
maintainer registers the backend with one line (`_runtimes['hip'] = HipRuntime`) and
`BaseModel(runtime='hip', ...)` (models/interfaces/base_model.py:139-209) hands every `compiled_infer` call
(base_model.py:367-375) to `HipRuntime.__call__`.

Call contracts honoured:
  Tacotron2  models/tts/tacotron2.py:162  compiled_infer(int32[1, Tin] | (tokens, float32[1, E]), max_length=10., **kw)
             -> object with .mel [B, Tmax, 80], .lengths [B], .attention_weights [B, Tmax, Tin] (+ the other fields of
             Tacotron2InferenceOutput, tacotron2_arch.py:52-56); unknown kwargs are ignored.
  WaveGlow   models/tts/waveglow.py:82-132 compiled_infer(float32[B, T, 80], **kw) -> float32[B, T*256];
             honours z / sigma / deterministic (waveglow_arch.py:244).
Randomness: like the reference (prenet dropout and z are sampled inside the graph, on the device:
tacotron2_arch.py:197-201, waveglow_arch.py:272-274,299-302) the default path draws both ON THE GPU -- the engine's
documented Philox4x32-10 stream (include/tts_hip.h, oracle/philox_ref.py), keyed by the runtime's seed and a running block
offset -- so no host-made tensor crosses PCIe.  The dropout bits and the noise are separate streams of one seed (key = seed
XOR a purpose constant in the high word: MASK_STREAM / NOISE_STREAM).  Explicit control stays: pass `prenet_masks` / `z`, or
`deterministic=True`, or `seed=` (that call then starts at offset 0 of that seed's streams and is reproducible).
"""
from __future__ import annotations

import os
import threading
from abc import ABCMeta, abstractmethod

import numpy as np

from .engine import HipEngine, Tacotron2InferenceOutput, _is_torch_cuda


# An explicit `seed=` names one stream per PURPOSE: the prenet dropout bits and the WaveGlow noise of the same seed must not be
# the same Philox blocks (both would start at offset 0 of one key and be correlated), so the purpose is XORed into the key's
# high word.  The engine-level calls (HipEngine.waveglow_infer(seed=...), tts_hip_random_fill) take the key as given.
MASK_STREAM = 0x4D41534B << 32          # "MASK"
NOISE_STREAM = 0x5A4E5345 << 32         # "ZNSE"
_U64 = (1 << 64) - 1


def rank_stream(seed: int, rank: int) -> int:
    """Key of rank `rank`'s stream under a job-wide seed: ranks synthesize different shards and must not draw identical noise."""
    return (int(seed) ^ (((int(rank) + 1) * 0x9E3779B97F4A7C15) & _U64)) & _U64 if rank else int(seed) & _U64


def sample_prenet_masks(rng, B, max_len):
    """Multiplicative prenet dropout masks [B, max_len, 2, 256] in {0, 2} (Bernoulli(0.5), scale 1 / (1 - 0.5):
    tacotron2_arch.py:188-203), drawn from the bits of `rng.integers(0, 256, uint8)` -- 6x cheaper on the host than
    thresholding `rng.random` floats (0.7 ms instead of 4.6 ms for a 1280-step budget), which matters because the reference
    keeps this dropout ON at inference, i.e. every `tts()` call samples them."""
    n = int(B) * int(max_len) * 2 * 256
    bits = rng.integers(0, 256, size=(n + 7) // 8, dtype=np.uint8)
    return (np.unpackbits(bits)[:n].reshape(B, max_len, 2, 256).astype(np.float32)) * np.float32(2.0)


class Runtime(metaclass=ABCMeta):
    """Plug-in interface of the reference's execution seam (utils/keras/runtimes/runtime.py:19-41): constructed as
    `cls(path, **kwargs)` by `build_runtime`, called in place of `model.infer`, with the loaded engine shared between
    instances.  Two deliberate differences from the reference's class: the shared-engine table is keyed by everything that
    changes what gets loaded -- (class, path, device, speaker_embedding_dim), not the path alone, so `device=1` or another
    speaker width never silently returns the first engine -- and it is guarded by a lock (the reference's dict is not)."""
    _engines = {}
    _engines_lock = threading.Lock()

    @classmethod
    def _engine_key(cls, path, kwargs):
        return (cls.__name__, str(path), int(kwargs.get('device', 0) or 0), int(kwargs.get('speaker_embedding_dim', 0) or 0))

    def __init__(self, path, *, engine=None, reload=False, **kwargs):
        self.path = path
        if engine is not None:
            self.engine = engine
            return
        key = self._engine_key(path, kwargs)
        with Runtime._engines_lock:
            if reload or key not in Runtime._engines:
                Runtime._engines[key] = self.load_engine(path, **kwargs)
            self.engine = Runtime._engines[key]

    def __repr__(self):
        return f'<{type(self).__name__} path={self.path}>'

    @abstractmethod
    def __call__(self, *args, **kwargs):
        """Runs inference on the engine."""

    @staticmethod
    @abstractmethod
    def load_engine(path, **kwargs):
        """Builds the engine for `path`."""


class HipRuntime(Runtime):
    """MI355X engine behind the reference's `compiled_infer`.

    path   : a TTSW weight file (text_to_speech_amd.weights.save_ttsw) or a `.safetensors` file with the same tensor
             names, or 'synthetic' / 'synthetic:<seed>' for the
             seeded synthetic weights of SURVEY.md section 8d.
    model  : 'tacotron2' | 'waveglow' | None (None: dispatch on the input dtype -- integer tokens vs float mels).
    vocoder_precision : 'f32' (exact fp32 MFMA, default) or 'f16' (fp16 GEMM operands with fp32 accumulation: the
             counterpart of the reference's `mixed_float16` policy, utils/keras/gpu.py).
    synthesizer_precision : 'f32' (default) or 'f16' (decoder LSTM weights in fp16, everything else fp32).
    """

    def __init__(self, path, *, model=None, engine=None, reload=False, device=0, seed=None, **kwargs):
        super().__init__(path, engine=engine, reload=reload, device=device, **kwargs)
        self.model = model
        self._rng = np.random.default_rng(seed)       # host generator (only `sample_prenet_masks` callers use it)
        # device stream: key + running block offset (a block = 4 values); an unseeded runtime takes its key from the OS
        self._seed = int(seed) if seed is not None else int(np.random.SeedSequence().generate_state(2, np.uint32).view(np.uint64)[0])
        self._offset = 0
        self.max_decoder_steps = int(kwargs.get('max_decoder_steps', 2000))
        self.vocoder_precision = kwargs.get('vocoder_precision', 'f32')
        if self.vocoder_precision not in ('f32', 'f16', 'f16x3'):
            raise ValueError(f"vocoder_precision must be 'f32', 'f16x3' or 'f16', got {self.vocoder_precision!r}")
        self.synthesizer_precision = kwargs.get('synthesizer_precision', 'f32')
        if self.synthesizer_precision not in ('f32', 'f16'):
            raise ValueError(f"synthesizer_precision must be 'f32' or 'f16', got {self.synthesizer_precision!r}")
        # The reference retries a sentence up to `max_trial` times with fresh prenet dropout (models/tts/tacotron2.py:160-179):
        # the encoder output of the last token batch is kept so that a retry only re-runs the decoder loop.
        self._encoded = None                     # (key, EncodedBatch)
        self.encoder_reuses = 0

    @staticmethod
    def load_engine(path, device=0, speaker_embedding_dim=0, **kwargs):
        eng = HipEngine(device)
        if isinstance(path, str) and path.startswith('synthetic'):
            from . import weights
            from .config import Tacotron2Config, WaveGlowConfig
            seed = int(path.split(':', 1)[1]) if ':' in path else 1234
            eng.load_state(weights.synth_waveglow(WaveGlowConfig(), seed=seed))
            eng.load_state(weights.synth_tacotron2(Tacotron2Config(speaker_embedding_dim=speaker_embedding_dim),
                                                   seed=seed))
        else:
            if not os.path.exists(path):
                raise FileNotFoundError(path)
            if str(path).endswith('.safetensors'):
                from . import weights
                eng.load_state(weights.load_safetensors(path))
            else:
                eng.load_weights(path)
        eng.finalize()
        return eng

    # ------------------------------------------------------------------ dispatch
    def __call__(self, inputs, *args, **kwargs):
        model = self.model
        if model is None:
            first = inputs[0] if isinstance(inputs, (tuple, list)) else inputs
            kind = str(getattr(first, 'dtype', ''))
            model = 'tacotron2' if 'int' in kind else 'waveglow'
        if model == 'tacotron2':
            return self.tacotron2_infer(inputs, **kwargs)
        if model == 'waveglow':
            return self.waveglow_infer(inputs, *args, **kwargs)
        raise ValueError(f'unknown model {model!r}')

    # ------------------------------------------------------------------ Tacotron2.infer (tacotron2_arch.py:866-925)
    def tacotron2_infer(self, inputs, *, max_length=None, early_stopping=True, attn_mask_offset=0.5,
                        attn_mask_win_len=None, prenet_masks=None, deterministic=False, seed=None, precision=None,
                        **_ignored):
        if isinstance(inputs, (tuple, list)):
            tokens, speaker = inputs[0], (inputs[1] if len(inputs) > 1 else None)
        else:
            tokens, speaker = inputs, None
        dev = _is_torch_cuda(tokens)
        if not dev:
            tokens = np.asarray(tokens)
        if tokens.ndim == 1:
            tokens = tokens[None]
        B = int(tokens.shape[0])
        if max_length is None:                      # :886-887 (hparam max_decoder_steps)
            max_len = self.max_decoder_steps
        elif isinstance(max_length, float):         # :888-892
            n_tok = int((tokens != 0).sum(1).max())
            max_len = int(np.float32(n_tok) * np.float32(max_length))
        else:
            max_len = int(max_length)
        max_len = max(1, max_len)
        if attn_mask_win_len is not None and isinstance(attn_mask_offset, float):   # :894-897
            attn_mask_offset = int(np.float32(attn_mask_win_len) * np.float32(attn_mask_offset))
        mask_seed = None
        if prenet_masks is None and not deterministic:
            if seed is not None:
                mask_seed = ((int(seed) ^ MASK_STREAM) & _U64, 0)
            else:
                mask_seed = ((self._seed ^ MASK_STREAM) & _U64, self._offset)
                self._offset += (B * max_len * 512 + 3) // 4
        # encoder reuse (the reference retries a sentence with fresh dropout, models/tts/tacotron2.py:160-179): inputs are
        # recognised by CONTENT -- host inputs by their bytes, device inputs by comparing them on the device with the runtime's
        # own copy of the batch that was encoded (an address / version key is not enough: the caching allocator hands the
        # address of a freed token tensor to the next sentence's tensor of the same shape)
        if dev:
            key = self._device_key(tokens, speaker)
        else:
            spk_np = None if speaker is None else (speaker.detach().cpu().numpy() if _is_torch_cuda(speaker) else np.asarray(speaker))
            key = ('host', tokens.shape, tokens.astype(np.int32).tobytes(),
                   None if spk_np is None else spk_np.astype(np.float32).tobytes())
        if not hasattr(self.engine, 'tacotron2_encode'):        # an engine object without the split entry points
            if mask_seed is not None and prenet_masks is None:
                prenet_masks = sample_prenet_masks(np.random.default_rng(mask_seed[0] + mask_seed[1]), B, max_len)
            return self.engine.tacotron2_infer(
                tokens, speaker=speaker, max_len=max_len, early_stopping=bool(early_stopping),
                prenet_masks=prenet_masks, attn_mask_win_len=attn_mask_win_len, attn_mask_offset=int(attn_mask_offset or 0),
                precision=precision or self.synthesizer_precision)
        if self._encoded is not None and (self._encoded[0] is key if dev else self._encoded[0] == key):
            self.encoder_reuses += 1
        elif self._encoded is not None:
            # one encoded-batch handle per runtime, overwritten sentence after sentence: no device allocation per sentence,
            # and the decoder's cached step graphs (keyed by that buffer) are replayed instead of re-captured
            handle = self._encoded[1]
            self._encoded = None
            try:
                self._encoded = (key, self.engine.tacotron2_encode(tokens, speaker=speaker, into=handle))
            except Exception:
                handle.close()
                raise
        else:
            self._encoded = (key, self.engine.tacotron2_encode(tokens, speaker=speaker))
        try:
            return self.engine.tacotron2_decode(
                self._encoded[1], max_len=max_len, early_stopping=bool(early_stopping), prenet_masks=prenet_masks,
                attn_mask_win_len=attn_mask_win_len, attn_mask_offset=int(attn_mask_offset or 0),
                precision=precision or self.synthesizer_precision, mask_seed=mask_seed)
        except Exception:
            # a failed decode may have been caused by the encoded batch itself (e.g. the encoder's block exchange timed out and
            # left its status in the buffer): a retry must run the encoder again, not reuse it
            self._drop_encoded()
            raise

    def _device_key(self, tokens, speaker):
        """Key of a device token batch: the cached key itself when the contents equal the batch it was made from (one tiny
        comparison kernel + a host read of its verdict), else a new key holding private copies of the inputs."""
        import torch
        spk_dev = speaker is not None and _is_torch_cuda(speaker)
        if self._encoded is not None and self._encoded[0][0] == 'dev':
            _, kept_tok, kept_spk = self._encoded[0]
            same = (kept_tok.device == tokens.device and tuple(kept_tok.shape) == tuple(tokens.shape)
                    and (kept_spk is None) == (speaker is None))
            if same and speaker is not None:
                spk = speaker if spk_dev else torch.as_tensor(np.asarray(speaker), dtype=torch.float32)
                same = tuple(kept_spk.shape) == tuple(spk.shape) and bool(torch.equal(kept_spk, spk.to(device=kept_spk.device, dtype=torch.float32)))
            if same and bool(torch.equal(kept_tok, tokens.to(torch.int32))):
                return self._encoded[0]
        spk_copy = None
        if speaker is not None:
            spk_copy = (speaker if spk_dev else torch.as_tensor(np.asarray(speaker), dtype=torch.float32)).to(
                device=tokens.device, dtype=torch.float32).clone()
        return ('dev', tokens.to(torch.int32).clone(), spk_copy)

    def _drop_encoded(self):
        if self._encoded is not None:
            try:
                self._encoded[1].close()
            finally:
                self._encoded = None

    # ------------------------------------------------------------------ WaveGlow.infer (waveglow_arch.py:244-306)
    def waveglow_infer(self, mel, z=None, sigma=1.0, deterministic=False, seed=None, precision=None, **_ignored):
        dev = _is_torch_cuda(mel)
        if not dev:
            mel = np.asarray(mel, dtype=np.float32)
        if mel.ndim == 2:
            mel = mel[None]
        B, T = int(mel.shape[0]), int(mel.shape[1])
        if z is None and not deterministic:
            if seed is not None:
                zs, zo = (int(seed) ^ NOISE_STREAM) & _U64, 0
            else:
                zs, zo = (self._seed ^ NOISE_STREAM) & _U64, self._offset
                self._offset += (B * T * 256 + 3) // 4
            return self.engine.waveglow_infer(mel, sigma=float(sigma), precision=precision or self.vocoder_precision,
                                              seed=zs, offset=zo)
        return self.engine.waveglow_infer(mel, z=z, sigma=float(sigma), precision=precision or self.vocoder_precision)


_runtimes = {'hip': HipRuntime}


def build_runtime(runtime, path, *args, **kwargs):
    """Same signature and error behaviour as the reference's `build_runtime` (runtimes/__init__.py:23-37)."""
    if runtime not in _runtimes:
        raise ValueError('Unsupported runtime !\n  Accepted : {}\n  Got : {}'.format(tuple(_runtimes.keys()), runtime))
    return _runtimes[runtime](path, *args, **kwargs)


__all__ = ['Runtime', 'HipRuntime', 'build_runtime', 'Tacotron2InferenceOutput']
