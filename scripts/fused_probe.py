"""Bring-up / timing probe for the fused two-kernel Tacotron2 decoder step: parity against the oracle on small cases, then
per-step time of the fused path and the per-step graph at BASELINE shapes.  usage: python scripts/fused_probe.py [quick]"""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from oracle import tacotron2_ref
from text_to_speech_amd import weights
from text_to_speech_amd.config import Tacotron2Config
from text_to_speech_amd.engine import HipEngine

cfg = Tacotron2Config()
tw = weights.synth_tacotron2(cfg, seed=1234)
eng = HipEngine(0)
eng.load_state(tw)
eng.finalize()


def tokens(B, Tin, lens, seed=0):
    rng = np.random.default_rng(seed)
    tok = rng.integers(1, 148, (B, Tin)).astype(np.int32)
    for b, n in enumerate(lens):
        tok[b, n:] = 0
    return tok


for B, Tin, lens, T in ((1, 24, [24], 40), (3, 40, [40, 31, 9], 20), (5, 37, [37, 20, 30, 11, 37], 35),
                        (8, 130, [130, 100, 64, 128, 90, 77, 130, 5], 34), (8, 64, [64] * 8, 70)):
    tok = tokens(B, Tin, lens, seed=B)
    masks = (np.random.default_rng(3).random((B, T, 2, 256)) >= 0.5).astype(np.float32) * 2.0
    ref = tacotron2_ref.infer(tok, tw, cfg, max_length=T, early_stopping=False, prenet_masks=masks)
    for mode in ('fused', 'graph'):
        eng.set_decoder_mode(mode)
        t0 = time.perf_counter()
        out = eng.tacotron2_infer(tok, max_len=T, early_stopping=False, prenet_masks=masks)
        dt = time.perf_counter() - t0
        errs = {k: float(np.abs(getattr(out, k) - getattr(ref, k)).max()) for k in ('decoder_output', 'mel', 'stop_tokens', 'attention_weights')}
        print(f'B={B} Tin={Tin} T={T} {mode:10s} ran={eng.last_decoder_mode:10s} steps={eng.last_steps} lengths={out.lengths.tolist()} '
              + ' '.join(f'{k}={v:.2e}' for k, v in errs.items()) + f' ({dt*1e3:.1f} ms)', flush=True)

if len(sys.argv) > 1 and sys.argv[1] == 'quick':
    sys.exit(0)
import torch
for B in (3, 4, 8):
    tok = np.zeros((B, 128), np.int32)
    tok[:, :100] = np.random.default_rng(5).integers(1, 148, (B, 100))
    tok_d = torch.from_numpy(tok).cuda()
    for prec in ('f32', 'f16'):
        for mode in ('fused', 'graph') + (('persistent',) if B <= 4 else ()):
            eng.set_decoder_mode(mode)
            eng.tacotron2_infer(tok_d, max_len=64, early_stopping=False, want_attention=False, precision=prec)
            ts = {}
            for n in (400, 800):
                t0 = time.perf_counter()
                for _ in range(3):
                    eng.tacotron2_infer(tok_d, max_len=n, early_stopping=False, want_attention=False, precision=prec)
                ts[n] = (time.perf_counter() - t0) / 3
            print(f'B={B} {prec} {mode:10s} ran={eng.last_decoder_mode:10s}: {1e6 * ts[800] / 800:.2f} us/step whole call, '
                  f'{1e6 * (ts[800] - ts[400]) / 400:.2f} us/step marginal', flush=True)
