"""Marginal us per decoder step of the persistent path for batch 1, 2, 4 (quick A/B of library builds via TTS_HIP_LIBRARY)."""
import os, sys, time
import numpy as np
sys.path.insert(0, '.')
import torch
from text_to_speech_amd import weights
from text_to_speech_amd.config import Tacotron2Config
from text_to_speech_amd.engine import HipEngine
eng = HipEngine(0)
eng.load_state(weights.synth_tacotron2(Tacotron2Config(), seed=1234))
eng.finalize()
eng.set_decoder_mode('persistent')
out = []
for B in (1, 2, 3, 4):
    tok = np.zeros((B, 128), np.int32)
    tok[:, :100] = np.random.default_rng(5).integers(1, 148, (B, 100))
    tok_d = torch.from_numpy(tok).cuda()
    eng.tacotron2_infer(tok_d, max_len=64, early_stopping=False, want_attention=False)
    ts = {}
    for n in (400, 800):
        t0 = time.perf_counter()
        for _ in range(3):
            eng.tacotron2_infer(tok_d, max_len=n, early_stopping=False, want_attention=False)
        ts[n] = (time.perf_counter() - t0) / 3
    out.append(f'B={B}: {1e6 * (ts[800] - ts[400]) / 400:.2f}')
print(os.path.basename(os.environ.get('TTS_HIP_LIBRARY', 'default')), eng.last_decoder_mode, ' '.join(out), 'us/step', flush=True)
