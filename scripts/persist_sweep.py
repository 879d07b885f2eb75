"""Finds the first-poll delays of the persistent decoder's timed optimistic polls (debug build only: the delays come from
TTS_PERSIST_DELAYS, which a -DTTS_DEBUG_HOOKS library reads per call).  Coordinate descent over the six hops.
usage: TTS_HIP_LIBRARY=text_to_speech_amd/libtts_hip_dbg.so python scripts/persist_sweep.py [B]"""
import os, sys, time
import numpy as np
sys.path.insert(0, '.')
import torch
from text_to_speech_amd import weights
from text_to_speech_amd.config import Tacotron2Config
from text_to_speech_amd.engine import HipEngine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
eng = HipEngine(0)
eng.load_state(weights.synth_tacotron2(Tacotron2Config(), seed=1234))
eng.finalize()
tok = np.zeros((B, 128), np.int32)
tok[:, :100] = np.random.default_rng(5).integers(1, 148, (B, 100))
tok_d = torch.from_numpy(tok).cuda()


def step_us(delays):
    os.environ['TTS_PERSIST_DELAYS'] = ','.join(str(int(d)) for d in delays)
    eng.tacotron2_infer(tok_d, max_len=64, early_stopping=False, want_attention=False)
    ts = {}
    for n in (400, 800):
        t0 = time.perf_counter()
        for _ in range(3):
            eng.tacotron2_infer(tok_d, max_len=n, early_stopping=False, want_attention=False)
        ts[n] = (time.perf_counter() - t0) / 3
    return 1e6 * (ts[800] - ts[400]) / 400


best = [int(x) for x in os.environ.get('SWEEP_START', '0,0,0,0,0,0').split(',')]
base = step_us(best)
print(f'B={B}: all sentinel polls: {base:.2f} us/step', flush=True)
# candidate delays in 10-ns ticks, per hop (A, B, C, D, E, F), around the anchors' expected arrival
cands = {0: (55, 65, 75), 1: (80, 90, 100, 110, 120), 2: (150, 165, 180, 195, 210, 225),
         3: (70, 85), 4: (50, 58, 66, 74, 82), 5: (120, 135, 150, 165, 180, 195)}
if os.environ.get('SWEEP_CANDS'):                  # e.g. '{"1": [180, 240, 300], "4": [120, 170, 220]}': only these hops are swept
    import json
    cands = {h: () for h in range(6)}
    cands.update({int(k): tuple(v) for k, v in json.loads(os.environ['SWEEP_CANDS']).items()})
cur = base
for sweep in range(2):
    for hop in (4, 1, 5, 2, 0, 3):
        for d in cands[hop]:
            trial = list(best)
            trial[hop] = d
            us = step_us(trial)
            mark = ''
            if us < cur - 0.02:
                cur, best, mark = us, trial, '  <- kept'
            print(f'  hop {"ABCDEF"[hop]} delay {d * 10:5d} ns: {us:.2f} us/step{mark}', flush=True)
    print(f'sweep {sweep}: best {best} -> {cur:.2f} us/step', flush=True)
