"""Marginal time per decoder step of one decoder path.  usage: python scripts/fused_time.py [mode] [B ...]
(TTS_HIP_LIBRARY selects a build variant; FUSED_TIME_TIN = padded token count, default 128; 78 % of it are real tokens)"""
import os, sys, time
import numpy as np
sys.path.insert(0, '.')
import torch
from text_to_speech_amd import weights
from text_to_speech_amd.config import Tacotron2Config
from text_to_speech_amd.engine import HipEngine

mode = sys.argv[1] if len(sys.argv) > 1 else 'fused'
Bs = [int(v) for v in sys.argv[2:]] or [8]
eng = HipEngine(0)
eng.load_state(weights.synth_tacotron2(Tacotron2Config(), seed=1234))
eng.finalize()
eng.set_decoder_mode(mode)
for B in Bs:
    Tin = int(os.environ.get('FUSED_TIME_TIN', '128'))
    n_tok = max(2, Tin * 100 // 128)
    tok = np.zeros((B, Tin), np.int32)
    tok[:, :n_tok] = np.random.default_rng(5).integers(1, 148, (B, n_tok))
    tok_d = torch.from_numpy(tok).cuda()
    for prec in ('f32', 'f16'):
        eng.tacotron2_infer(tok_d, max_len=64, early_stopping=False, want_attention=False, precision=prec)
        ts = {}
        for n in (400, 800):
            best = 1e9
            for _ in range(4):
                t0 = time.perf_counter()
                eng.tacotron2_infer(tok_d, max_len=n, early_stopping=False, want_attention=False, precision=prec)
                best = min(best, time.perf_counter() - t0)
            ts[n] = best
        print(f'B={B} Tin={Tin} {prec} {mode} ran={eng.last_decoder_mode}: {1e6 * ts[800] / 800:.2f} us/step whole call, '
              f'{1e6 * (ts[800] - ts[400]) / 400:.2f} us/step marginal', flush=True)
