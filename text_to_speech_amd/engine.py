"""`HipEngine`: Python owner of one `tts_hip_engine` handle (one GPU, one HIP stream).

Inputs may be numpy arrays (host buffers: the library stages them over PCIe) or torch CUDA tensors (device buffers:
passed by pointer, results returned as torch tensors on the same device).  PyTorch is only used to host device memory.
"""
from __future__ import annotations

import ctypes
from collections import namedtuple

import numpy as np

from . import _lib
from ._lib import HipLibraryError, MEM_DEVICE, MEM_HOST

# field order of the reference's namedtuple (architectures/tacotron2_arch.py:52-56)
Tacotron2InferenceOutput = namedtuple(
    'Tacotron2InferenceOutput', ['decoder_output', 'mel', 'stop_tokens', 'attention_weights', 'lengths'])

KERNEL_WN_IN, KERNEL_WN_RES_SKIP, KERNEL_DECODER_STEP = 0, 1, 2


def _is_torch_cuda(x) -> bool:
    return hasattr(x, 'data_ptr') and hasattr(x, 'is_cuda') and bool(x.is_cuda)


class EncodedBatch:
    """Encoder output of one token batch, resident on the GPU (tts_hip_encoded).  Freed by `close()`, the garbage collector
    or the engine's own teardown order (an engine must outlive its encoded batches)."""

    def __init__(self, engine, handle, B, Tin, keep=(), on_device=False):
        self.engine, self.handle, self.B, self.Tin, self.on_device = engine, handle, B, Tin, on_device
        self._keep = keep                       # inputs of the still-running asynchronous encoder

    def close(self):
        if self.handle is not None and getattr(self.engine, '_h', None):
            self.engine._lib.tts_hip_encoded_free(self.engine._h, self.handle)
        self.handle = None
        self._keep = ()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HipEngine:
    def __init__(self, device: int = 0):
        self._lib = _lib.load_library()
        self._h = ctypes.c_void_p()
        rc = self._lib.tts_hip_create(int(device), ctypes.byref(self._h))
        if rc != 0:
            self._h = None
            raise HipLibraryError(f'tts_hip_create(device={device}) failed with code {rc} (no usable MI355X GPU?)')
        self.device = int(device)

    # ------------------------------------------------------------------ plumbing
    def close(self):
        if getattr(self, '_h', None):
            self._lib.tts_hip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            msg = self._lib.tts_hip_last_error(self._h)
            raise HipLibraryError(f'{what} failed ({rc}): {msg.decode("utf-8", "replace") if msg else ""}')

    def _torch(self):
        import torch
        return torch

    def _sync_torch(self):
        torch = self._torch()
        torch.cuda.current_stream(self.device).synchronize()

    def _check_device(self, *tensors):
        """Device tensors are passed by pointer: they must live on this engine's GPU (a pointer into another GPU's memory
        would be read as garbage or fault)."""
        for t in tensors:
            if t is not None and _is_torch_cuda(t) and t.device.index != self.device:
                raise ValueError(f'tensor on {t.device} passed to an engine on cuda:{self.device}')

    def _order_after_torch(self, stream=None):
        """Makes the work this call is about to enqueue wait for what torch has queued so far.  stream=None: the engine's
        own (non-blocking) stream is used and the call returns after it drained, so torch's current stream on this device
        is synchronized first (inputs may still be in flight there).  With a torch stream the engine call is enqueued on
        that very stream: ordering is implicit and nothing is synchronized."""
        if stream is None:
            self._sync_torch()
            return None
        return ctypes.c_void_p(int(stream.cuda_stream))

    def _enter_stream(self, stream):
        """Context for the `stream=` paths: `stream` first waits for what torch's current stream has queued (the caller's
        inputs may still be in flight there), then becomes torch's current stream, so that every conversion, padding and
        output allocation of the call is ordered on the stream the engine kernels run on."""
        torch = self._torch()
        stream.wait_stream(torch.cuda.current_stream(self.device))
        return torch.cuda.stream(stream)

    @staticmethod
    def _used_on(stream, *tensors):
        """Tensors passed to the engine by pointer and read asynchronously on `stream`: tell the caching allocator, so that
        a temporary freed when the call returns is not handed out again while the kernels still read it."""
        for t in tensors:
            if t is not None:
                t.record_stream(stream)

    # ------------------------------------------------------------------ device-side sampling
    def random_normal(self, shape, seed: int, offset: int = 0, stream=None):
        """N(0, 1) float32 tensor of `shape` on this engine's GPU, drawn on the device (Philox4x32-10 + Box-Muller,
        tts_hip_random_fill); the same (seed, offset) always gives the same values."""
        return self._random(0, shape, seed, offset, stream)

    def random_prenet_masks(self, B: int, max_len: int, seed: int, offset: int = 0, stream=None):
        """Prenet dropout masks [B, max_len, 2, 256] (2.0 with probability 0.5, else 0.0) drawn on the device."""
        return self._random(1, (int(B), int(max_len), 2, 256), seed, offset, stream)

    def _random(self, kind, shape, seed, offset, stream):
        torch = self._torch()
        dev = torch.device('cuda', self.device)
        u64 = lambda v: ctypes.c_uint64(int(v) & 0xFFFFFFFFFFFFFFFF)
        if stream is not None:
            with self._enter_stream(stream):
                out = torch.empty(tuple(shape), dtype=torch.float32, device=dev)
            sp = ctypes.c_void_p(int(stream.cuda_stream))
        else:
            out = torch.empty(tuple(shape), dtype=torch.float32, device=dev)
            sp = None
            # the block may have pending readers / writers on torch's current stream (the caching allocator reuses it in that
            # stream's order only); the fill runs on the engine's own stream
            self._order_after_torch()
        self._check(self._lib.tts_hip_random_fill(self._h, kind, u64(seed), u64(offset), ctypes.c_void_p(out.data_ptr()),
                                                  out.numel(), sp), 'random_fill')
        if stream is None:
            self.synchronize()
        return out

    # ------------------------------------------------------------------ weights
    def set_tensor(self, name: str, array) -> None:
        a = np.ascontiguousarray(array, dtype=np.float32)
        dims = (ctypes.c_int64 * a.ndim)(*a.shape)
        self._check(self._lib.tts_hip_set_tensor(self._h, name.encode(), a.ctypes.data_as(ctypes.c_void_p), dims,
                                                 a.ndim), f'set_tensor({name})')

    def load_state(self, tensors) -> None:
        for k, v in tensors.items():
            self.set_tensor(k, v)

    def load_weights(self, path: str) -> None:
        self._check(self._lib.tts_hip_load_weights(self._h, str(path).encode()), f'load_weights({path})')

    def finalize(self) -> None:
        self._check(self._lib.tts_hip_finalize(self._h), 'finalize')

    def has_model(self, model: str) -> bool:
        return bool(self._lib.tts_hip_has_model(self._h, model.encode()))

    # ------------------------------------------------------------------ WaveGlow
    def waveglow_infer(self, mel, z=None, sigma: float = 1.0, precision: str = 'f32', stream=None, seed=None, offset: int = 0):
        """mel [B, T, 80] (+ optional z [B, T*32, 8]) -> audio [B, T*256].  precision: 'f32' (exact fp32 MFMA), 'f16x3'
        (split fp16: fp32-class accuracy, ~3x faster) or 'f16' (fp16 operands).  `seed` (with z=None): the noise is drawn
        on the device from (seed, offset) -- the reference's default `z=None, deterministic=False`
        (waveglow_arch.py:272-274,299-302) without a host-made tensor crossing PCIe; z=None and seed=None: zeros
        (`deterministic=True`).  `stream` (a torch.cuda.Stream, device tensors only): enqueue on that stream and return
        without waiting (tts_hip_waveglow_infer_async)."""
        fns = {'f32': self._lib.tts_hip_waveglow_infer, 'f16': self._lib.tts_hip_waveglow_infer_f16,
               'f16x3': self._lib.tts_hip_waveglow_infer_f16x3}
        if precision not in fns:
            raise ValueError(f"precision must be one of {tuple(fns)}, got {precision!r}")
        if z is not None and seed is not None:
            raise ValueError('pass either z or seed, not both')
        fn = fns[precision]
        pcode = {'f32': 0, 'f16': 1, 'f16x3': 2}[precision]
        u64 = lambda v: ctypes.c_uint64(int(v) & 0xFFFFFFFFFFFFFFFF)
        if _is_torch_cuda(mel):
            torch = self._torch()
            if mel.dim() != 3 or mel.shape[2] != 80:
                raise ValueError(f'mel must be [B, T, 80], got {tuple(mel.shape)}')
            B, T = int(mel.shape[0]), int(mel.shape[1])
            if z is not None and tuple(z.shape) != (B, T * 32, 8):
                raise ValueError(f'z must be [B, T*32, 8] = {(B, T * 32, 8)}, got {tuple(z.shape)}')
            self._check_device(mel, z)

            def prepared():
                m = mel.to(torch.float32).contiguous()
                zz = None if z is None else z.to(device=mel.device, dtype=torch.float32).contiguous()
                return m, zz, torch.empty((B, T * 256), dtype=torch.float32, device=mel.device)

            if stream is not None:
                with self._enter_stream(stream):
                    m, zz, out = prepared()
                    if zz is None and seed is not None:
                        zz = torch.empty((B, T * 32, 8), dtype=torch.float32, device=mel.device)
                        self._check(self._lib.tts_hip_random_fill(self._h, 0, u64(seed), u64(offset),
                                                                  ctypes.c_void_p(zz.data_ptr()), zz.numel(),
                                                                  ctypes.c_void_p(int(stream.cuda_stream))), 'random_fill')
                self._used_on(stream, mel, z, m, zz, out)
                self._check(self._lib.tts_hip_waveglow_infer_async(
                    self._h, ctypes.c_void_p(m.data_ptr()), B, T, None if zz is None else ctypes.c_void_p(zz.data_ptr()),
                    float(sigma), ctypes.c_void_p(out.data_ptr()), pcode, ctypes.c_void_p(int(stream.cuda_stream))),
                    'waveglow_infer_async')
                return out
            m, zz, out = prepared()
            self._order_after_torch()
            if zz is None and seed is not None:
                self._check(self._lib.tts_hip_waveglow_infer_seeded(
                    self._h, ctypes.c_void_p(m.data_ptr()), B, T, u64(seed), u64(offset), float(sigma),
                    ctypes.c_void_p(out.data_ptr()), pcode, MEM_DEVICE), 'waveglow_infer_seeded')
                return out
            self._check(fn(self._h, ctypes.c_void_p(m.data_ptr()), B, T, None if zz is None else ctypes.c_void_p(zz.data_ptr()),
                           float(sigma), ctypes.c_void_p(out.data_ptr()), MEM_DEVICE), 'waveglow_infer')
            return out
        if stream is not None:
            raise ValueError('stream= needs device tensors')
        mel = np.ascontiguousarray(mel, dtype=np.float32)
        if mel.ndim != 3 or mel.shape[2] != 80:
            raise ValueError(f'mel must be [B, T, 80], got {mel.shape}')
        B, T = mel.shape[:2]
        zp = None
        if z is not None:
            z = np.ascontiguousarray(z, dtype=np.float32)
            if z.shape != (B, T * 32, 8):
                raise ValueError(f'z must be [B, T*32, 8] = {(B, T * 32, 8)}, got {z.shape}')
            zp = z.ctypes.data_as(ctypes.c_void_p)
        out = np.empty((B, T * 256), dtype=np.float32)
        if zp is None and seed is not None:
            self._check(self._lib.tts_hip_waveglow_infer_seeded(
                self._h, mel.ctypes.data_as(ctypes.c_void_p), B, T, u64(seed), u64(offset), float(sigma),
                out.ctypes.data_as(ctypes.c_void_p), pcode, MEM_HOST), 'waveglow_infer_seeded')
            return out
        self._check(fn(self._h, mel.ctypes.data_as(ctypes.c_void_p), B, T, zp, float(sigma),
                       out.ctypes.data_as(ctypes.c_void_p), MEM_HOST), 'waveglow_infer')
        return out

    # ------------------------------------------------------------------ Tacotron2
    def tacotron2_infer(self, tokens, speaker=None, max_len: int = 1000, early_stopping: bool = True,
                        prenet_masks=None, attn_mask_win_len=None, attn_mask_offset: int = 0, want_attention=True,
                        precision: str = 'f32'):
        """tokens int32 [B, Tin] -> Tacotron2InferenceOutput of numpy arrays (or torch tensors for CUDA tokens).
        precision 'f16': decoder LSTM weights in fp16 (fp32 accumulate and state)."""
        if precision not in ('f32', 'f16'):
            raise ValueError(f"precision must be 'f32' or 'f16', got {precision!r}")
        fn = self._lib.tts_hip_tacotron2_infer if precision == 'f32' else self._lib.tts_hip_tacotron2_infer_f16
        dev = _is_torch_cuda(tokens)
        if dev:
            torch = self._torch()
            tok = tokens.to(torch.int32).contiguous()
            B, Tin = int(tok.shape[0]), int(tok.shape[1])
            device = tok.device
            mk = lambda shape, dt=None: torch.zeros(shape, dtype=dt or torch.float32, device=device)
            ptr = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
            if speaker is not None:
                speaker = speaker.to(device=device, dtype=torch.float32).contiguous()
            if prenet_masks is not None:
                prenet_masks = prenet_masks.to(device=device, dtype=torch.float32).contiguous()
            i32 = torch.int32
        else:
            tok = np.ascontiguousarray(tokens, dtype=np.int32)
            if tok.ndim != 2:
                raise ValueError(f'tokens must be [B, Tin], got {tok.shape}')
            B, Tin = tok.shape
            mk = lambda shape, dt=None: np.zeros(shape, dtype=dt or np.float32)
            ptr = lambda a: None if a is None else a.ctypes.data_as(ctypes.c_void_p)
            if speaker is not None:
                speaker = np.ascontiguousarray(speaker, dtype=np.float32)
            if prenet_masks is not None:
                prenet_masks = np.ascontiguousarray(prenet_masks, dtype=np.float32)
            i32 = np.int32
        max_len = int(max_len)
        if max_len <= 0:
            raise ValueError('max_len must be positive')
        if prenet_masks is not None and tuple(prenet_masks.shape) != (B, max_len, 2, 256):
            raise ValueError(f'prenet_masks must be [B, max_len, 2, 256], got {tuple(prenet_masks.shape)}')
        mel = mk((B, max_len, 80))
        dec = mk((B, max_len, 80))
        stop = mk((B, max_len))
        attn = mk((B, max_len, Tin)) if want_attention else None
        lengths = mk((B,), i32)
        steps = ctypes.c_int32(0)
        if dev:
            self._check_device(tok, speaker, prenet_masks)
            self._order_after_torch()
        win = int(attn_mask_win_len) if attn_mask_win_len is not None else 0
        self._check(fn(
            self._h, ptr(tok), B, Tin, ptr(speaker), max_len, 1 if early_stopping else 0, ptr(prenet_masks),
            win, int(attn_mask_offset), ptr(mel), ptr(dec), ptr(stop), ptr(attn), ptr(lengths),
            ctypes.cast(ctypes.byref(steps), ctypes.c_void_p), MEM_DEVICE if dev else MEM_HOST), 'tacotron2_infer')
        out = Tacotron2InferenceOutput(decoder_output=dec, mel=mel, stop_tokens=stop, attention_weights=attn,
                                       lengths=lengths)
        self.last_steps = int(steps.value)
        return out

    # ------------------------------------------------------------------ Tacotron2 in two calls
    def tacotron2_encode(self, tokens, speaker=None, stream=None, into=None):
        """tokens int32 [B, Tin] (+ speaker [B, E]) -> `EncodedBatch` (the encoder's output, kept on the GPU).  Asynchronous:
        the encoder is only enqueued (on `stream`, a torch.cuda.Stream, or on the engine's own stream).  `into`: an
        `EncodedBatch` of this engine to overwrite (tts_hip_tacotron2_reencode: no allocation, the decoder's cached graphs
        stay valid); it is returned."""
        dev = _is_torch_cuda(tokens)
        if dev:
            torch = self._torch()
            self._check_device(tokens, speaker)

            def convert():
                t_ = tokens.to(torch.int32).contiguous()
                s_ = None if speaker is None else speaker.to(device=tokens.device, dtype=torch.float32).contiguous()
                return t_, s_

            if stream is not None:
                with self._enter_stream(stream):
                    tok, spk = convert()
                self._used_on(stream, tokens, speaker, tok, spk)
            else:
                tok, spk = convert()
            ptr = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
            sp = self._order_after_torch(stream)
        else:
            if stream is not None:
                raise ValueError('stream= needs device tensors')
            tok = np.ascontiguousarray(tokens, dtype=np.int32)
            spk = None if speaker is None else np.ascontiguousarray(speaker, dtype=np.float32)
            ptr = lambda a: None if a is None else a.ctypes.data_as(ctypes.c_void_p)
            sp = None
        if tok.ndim != 2:
            raise ValueError(f'tokens must be [B, Tin], got {tuple(tok.shape)}')
        if into is not None:
            if into.engine is not self or into.handle is None:
                raise ValueError('`into` belongs to another engine or was freed')
            self._check(self._lib.tts_hip_tacotron2_reencode(self._h, into.handle, ptr(tok), int(tok.shape[0]), int(tok.shape[1]),
                                                             ptr(spk), MEM_DEVICE if dev else MEM_HOST, sp), 'tacotron2_reencode')
            into.B, into.Tin, into.on_device, into._keep = int(tok.shape[0]), int(tok.shape[1]), dev, (tok, spk)
            return into
        h = ctypes.c_void_p()
        self._check(self._lib.tts_hip_tacotron2_encode(self._h, ptr(tok), int(tok.shape[0]), int(tok.shape[1]), ptr(spk),
                                                       MEM_DEVICE if dev else MEM_HOST, sp, ctypes.byref(h)),
                    'tacotron2_encode')
        return EncodedBatch(self, h, int(tok.shape[0]), int(tok.shape[1]), keep=(tok, spk), on_device=dev)

    def tacotron2_decode(self, encoded, max_len: int = 1000, early_stopping: bool = True, prenet_masks=None,
                         attn_mask_win_len=None, attn_mask_offset: int = 0, want_attention=True, precision: str = 'f32',
                         stream=None, mask_seed=None):
        """Decoder loop + postnet on an `EncodedBatch`; may be called repeatedly (new dropout masks, other `max_len`).
        `mask_seed` = (seed, offset): the prenet dropout masks are drawn on the device (tts_hip_tacotron2_decode_seeded)
        instead of being passed in.  Returns after `stream` (or the engine's stream) has drained: the loop's length is
        decided on the GPU."""
        if mask_seed is not None and prenet_masks is not None:
            raise ValueError('pass either prenet_masks or mask_seed, not both')
        if precision not in ('f32', 'f16'):
            raise ValueError(f"precision must be 'f32' or 'f16', got {precision!r}")
        if encoded.engine is not self or encoded.handle is None:
            raise ValueError('this EncodedBatch belongs to another engine or was freed')
        B, Tin, dev = encoded.B, encoded.Tin, encoded.on_device
        max_len = int(max_len)
        if max_len <= 0:
            raise ValueError('max_len must be positive')
        scope = None
        if dev:
            torch = self._torch()
            device = torch.device('cuda', self.device)
            mk = lambda shape, dt=None: torch.zeros(shape, dtype=dt or torch.float32, device=device)
            ptr = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
            if prenet_masks is not None:
                self._check_device(prenet_masks)
            if stream is not None:                   # conversions and zero-filled outputs are ordered on `stream` itself
                scope = self._enter_stream(stream)
            i32 = torch.int32
        else:
            mk = lambda shape, dt=None: np.zeros(shape, dtype=dt or np.float32)
            ptr = lambda a: None if a is None else a.ctypes.data_as(ctypes.c_void_p)
            i32 = np.int32

        def prepare():
            masks = prenet_masks
            if masks is not None and dev:
                masks = masks.to(device=device, dtype=torch.float32).contiguous()
                if stream is not None:
                    self._used_on(stream, prenet_masks, masks)
            elif masks is not None:
                masks = np.ascontiguousarray(masks, dtype=np.float32)
            if masks is not None and tuple(masks.shape) != (B, max_len, 2, 256):
                raise ValueError(f'prenet_masks must be [B, max_len, 2, 256], got {tuple(masks.shape)}')
            return (masks, mk((B, max_len, 80)), mk((B, max_len, 80)), mk((B, max_len)),
                    mk((B, max_len, Tin)) if want_attention else None, mk((B,), i32))

        if scope is not None:
            with scope:                              # (a failing conversion must not leave the caller's stream switched)
                prenet_masks, mel, dec, stop, attn, lengths = prepare()
            self._used_on(stream, mel, dec, stop, attn, lengths)
        else:
            prenet_masks, mel, dec, stop, attn, lengths = prepare()
        steps = ctypes.c_int32(0)
        sp = self._order_after_torch(stream) if dev else None
        win = int(attn_mask_win_len) if attn_mask_win_len is not None else 0
        if mask_seed is not None:
            u64 = lambda v: ctypes.c_uint64(int(v) & 0xFFFFFFFFFFFFFFFF)
            self._check(self._lib.tts_hip_tacotron2_decode_seeded(
                self._h, encoded.handle, max_len, 1 if early_stopping else 0, u64(mask_seed[0]), u64(mask_seed[1]), win,
                int(attn_mask_offset), 1 if precision == 'f16' else 0, ptr(mel), ptr(dec), ptr(stop), ptr(attn), ptr(lengths),
                ctypes.cast(ctypes.byref(steps), ctypes.c_void_p), MEM_DEVICE if dev else MEM_HOST, sp), 'tacotron2_decode_seeded')
        else:
            self._check(self._lib.tts_hip_tacotron2_decode(
                self._h, encoded.handle, max_len, 1 if early_stopping else 0, ptr(prenet_masks), win, int(attn_mask_offset),
                1 if precision == 'f16' else 0, ptr(mel), ptr(dec), ptr(stop), ptr(attn), ptr(lengths),
                ctypes.cast(ctypes.byref(steps), ctypes.c_void_p), MEM_DEVICE if dev else MEM_HOST, sp), 'tacotron2_decode')
        self.last_steps = int(steps.value)
        return Tacotron2InferenceOutput(decoder_output=dec, mel=mel, stop_tokens=stop, attention_weights=attn,
                                        lengths=lengths)

    def set_decoder_mode(self, mode: str) -> None:
        """How the autoregressive decoder loop runs: 'auto' (default: the persistent weight-stationary kernel for 1 - 2 rows,
        the fused two-kernel step for 3 - 8 rows, whichever applies otherwise), 'persistent' / 'fused' (that machine when the
        call shape allows it) or 'graph' (always one hipGraph of 7 kernels per step, the fallback of the other two)."""
        modes = {'graph': 0, 'persistent': 1, 'fused': 2, 'auto': 3}
        if mode not in modes:
            raise ValueError(f'mode must be one of {tuple(modes)}, got {mode!r}')
        self._check(self._lib.tts_hip_set_decoder_mode(self._h, modes[mode]), 'set_decoder_mode')

    @property
    def last_decoder_mode(self) -> str:
        """How the last `tacotron2_infer` call ran its loop: 'fused', 'persistent', 'graph', or 'none' before the first call."""
        return {2: 'fused', 1: 'persistent', 0: 'graph'}.get(self._lib.tts_hip_last_decoder_mode(self._h), 'none')

    def set_waveglow_form(self, form: str) -> None:
        """How the fp32 vocoder evaluates the dilated convolutions of WN layers 1 - 7: 'winograd' (default: minimal filtering
        along the tap axis, F(4,3), for calls of 144 frames or more) or 'direct' (always three taps).  Both are fp32; they differ by rounding only."""
        # 'winograd-3pass' / 'winograd-prepass': the two earlier stages of the Winograd form (pre-pass + per-product GEMM + combine
        # pass; fused GEMM behind the pre-pass), kept for measurement and as bit-identical cross-checks of the default kernel
        forms = {'direct': 0, 'winograd': 1, 'winograd-3pass': 2, 'winograd-prepass': 3}
        if form not in forms:
            raise ValueError(f'form must be one of {tuple(forms)}, got {form!r}')
        self._check(self._lib.tts_hip_set_waveglow_form(self._h, forms[form]), 'set_waveglow_form')

    @property
    def last_waveglow_form(self) -> str:
        """'winograd' or 'direct' for the last `waveglow_infer` call on this handle ('none' before the first)."""
        return {1: 'winograd', 0: 'direct'}.get(self._lib.tts_hip_last_waveglow_form(self._h), 'none')

    def waveglow_probe_acts(self, mel, z=None, sigma: float = 1.0, flow: int = 11, layer: int = 1):
        """Test hook (tts_hip_waveglow_probe_acts): the gated activations [B, T*32, 512] of WN layer `layer` of flow `flow` on
        the fp32 path, in the form `set_waveglow_form` selects -- the values before the res/skip and `end` convolutions."""
        mel = np.ascontiguousarray(mel, dtype=np.float32)
        if mel.ndim != 3 or mel.shape[2] != 80:
            raise ValueError(f'mel must be [B, T, 80], got {mel.shape}')
        B, T = mel.shape[:2]
        zp = None
        if z is not None:
            z = np.ascontiguousarray(z, dtype=np.float32)
            if z.shape != (B, T * 32, 8):
                raise ValueError(f'z must be [B, T*32, 8] = {(B, T * 32, 8)}, got {z.shape}')
            zp = z.ctypes.data_as(ctypes.c_void_p)
        out = np.empty((B, T * 32, 512), dtype=np.float32)
        self._check(self._lib.tts_hip_waveglow_probe_acts(
            self._h, mel.ctypes.data_as(ctypes.c_void_p), B, T, zp, float(sigma), int(flow), int(layer),
            out.ctypes.data_as(ctypes.c_void_p), MEM_HOST), 'waveglow_probe_acts')
        return out

    # ------------------------------------------------------------------ mel-STFT
    def mel_stft(self, audio, stream=None):
        """audio [N] or [B, N] -> mel [B, N // 256 + 1, 80] (the reference's TacotronSTFT()(audio)).  `stream` (torch.cuda.Stream,
        device tensors only): enqueue there and return without waiting."""
        if _is_torch_cuda(audio):
            torch = self._torch()
            self._check_device(audio)

            def prepared():
                a_ = audio.to(torch.float32)
                if a_.dim() == 1:
                    a_ = a_[None]
                if a_.shape[1] < 1024:
                    a_ = torch.nn.functional.pad(a_, (0, 1024 - a_.shape[1]))
                a_ = a_.contiguous()
                return a_, torch.empty((int(a_.shape[0]), int(a_.shape[1]) // 256 + 1, 80), dtype=torch.float32, device=a_.device)

            if stream is not None:
                with self._enter_stream(stream):
                    a, out = prepared()
                self._used_on(stream, audio, a, out)
                B, N = int(a.shape[0]), int(a.shape[1])
                self._check(self._lib.tts_hip_mel_stft_async(self._h, ctypes.c_void_p(a.data_ptr()), B, N,
                                                             ctypes.c_void_p(out.data_ptr()), self._order_after_torch(stream)),
                            'mel_stft_async')
                return out
            a, out = prepared()
            B, N = int(a.shape[0]), int(a.shape[1])
            self._order_after_torch()
            self._check(self._lib.tts_hip_mel_stft(self._h, ctypes.c_void_p(a.data_ptr()), B, N,
                                                   ctypes.c_void_p(out.data_ptr()), MEM_DEVICE), 'mel_stft')
            return out
        if stream is not None:
            raise ValueError('stream= needs device tensors')
        a = np.asarray(audio, dtype=np.float32)
        if a.ndim == 1:
            a = a[None]
        if a.shape[1] < 1024:                     # MelSTFT.__call__ pads short audio (utils/audio/stft.py:113-115)
            a = np.pad(a, [(0, 0), (0, 1024 - a.shape[1])])
        a = np.ascontiguousarray(a)
        B, N = a.shape
        out = np.empty((B, N // 256 + 1, 80), dtype=np.float32)
        self._check(self._lib.tts_hip_mel_stft(self._h, a.ctypes.data_as(ctypes.c_void_p), B, N,
                                               out.ctypes.data_as(ctypes.c_void_p), MEM_HOST), 'mel_stft')
        return out

    # ------------------------------------------------------------------ measurement hooks
    def kernel_timing(self, enable: bool) -> None:
        self._check(self._lib.tts_hip_kernel_timing(self._h, 1 if enable else 0), 'kernel_timing')

    def kernel_time_us(self, kind: int):
        avg = ctypes.c_double(0)
        n = ctypes.c_int64(0)
        self._check(self._lib.tts_hip_kernel_time_us(self._h, int(kind), ctypes.byref(avg), ctypes.byref(n)),
                    'kernel_time_us')
        return avg.value, n.value

    def probe_mfma_f32(self):
        """(TFLOP/s, shader clock in GHz) of a bare fp32 MFMA loop on this device right now (tts_hip_probe_mfma_f32, ~60 ms)."""
        tf, ghz = ctypes.c_double(0), ctypes.c_double(0)
        self._check(self._lib.tts_hip_probe_mfma_f32(self._h, ctypes.byref(tf), ctypes.byref(ghz)), 'probe_mfma_f32')
        return tf.value, ghz.value

    def synchronize(self) -> None:
        self._check(self._lib.tts_hip_synchronize(self._h), 'synchronize')
