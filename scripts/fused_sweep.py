"""Coordinate descent over the first-look delays of the fused decoder's five hops (debug build: TTS_FUSED_DELAYS).
usage: TTS_HIP_LIBRARY=text_to_speech_amd/libtts_hip_dbg.so python scripts/fused_sweep.py [B] [prec]"""
import os, sys, time
import numpy as np
sys.path.insert(0, '.')
import torch
from text_to_speech_amd import weights
from text_to_speech_amd.config import Tacotron2Config
from text_to_speech_amd.engine import HipEngine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
prec = sys.argv[2] if len(sys.argv) > 2 else 'f32'
eng = HipEngine(0)
eng.load_state(weights.synth_tacotron2(Tacotron2Config(), seed=1234))
eng.finalize()
eng.set_decoder_mode('fused')
Tin = int(os.environ.get('FUSED_TIME_TIN', '128'))
n_tok = max(2, Tin * 100 // 128)
tok = np.zeros((B, Tin), np.int32)
tok[:, :n_tok] = np.random.default_rng(5).integers(1, 148, (B, n_tok))
tok_d = torch.from_numpy(tok).cuda()


def step_us(delays):
    os.environ['TTS_FUSED_DELAYS'] = ','.join(str(int(d)) for d in delays)
    ts = {}
    for n in (300, 600):
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            eng.tacotron2_infer(tok_d, max_len=n, early_stopping=False, want_attention=False, precision=prec)
            best = min(best, time.perf_counter() - t0)
        ts[n] = best
    return 1e6 * (ts[600] - ts[300]) / 300


cur = [int(v) for v in os.environ.get('SWEEP_START', '40,90,115,115,100').split(',')]
eng.tacotron2_infer(tok_d, max_len=64, early_stopping=False, want_attention=False, precision=prec)
best = step_us(cur)
print(f'B={B} {prec} start {cur}: {best:.2f} us/step', flush=True)
for rnd in range(2):
    for h in range(5):
        for cand in (0, 20, 40, 60, 80, 100, 120, 150, 200):
            if cand == cur[h]:
                continue
            trial = list(cur)
            trial[h] = cand
            v = step_us(trial)
            print(f'   hop {h} = {cand}: {v:.2f}', flush=True)
            if v < best - 0.05:
                best, cur = v, trial
        print(f'round {rnd} hop {h}: {cur} -> {best:.2f} us/step', flush=True)
print('best', cur, f'{best:.2f}')
