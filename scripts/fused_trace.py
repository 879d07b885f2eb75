"""Phase breakdown of the fused two-kernel decoder step from the debug build's timestamps (TTS_FUSED_TRACE_FILE, 100 MHz clock).
usage: TTS_HIP_LIBRARY=text_to_speech_amd/libtts_hip_dbg.so TTS_FUSED_TRACE_FILE=/tmp/tr.bin python scripts/fused_trace.py [B] [prec]"""
import os, sys
import numpy as np
sys.path.insert(0, '.')
import torch
from text_to_speech_amd import weights
from text_to_speech_amd.config import Tacotron2Config
from text_to_speech_amd.engine import HipEngine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
prec = sys.argv[2] if len(sys.argv) > 2 else 'f32'
path = os.environ['TTS_FUSED_TRACE_FILE']
eng = HipEngine(0)
eng.load_state(weights.synth_tacotron2(Tacotron2Config(), seed=1234))
eng.finalize()
eng.set_decoder_mode('fused')
Tin = int(os.environ.get('FUSED_TIME_TIN', '128'))
n_tok = max(2, Tin * 100 // 128)
tok = np.zeros((B, Tin), np.int32)
tok[:, :n_tok] = np.random.default_rng(5).integers(1, 148, (B, n_tok))
tok_d = torch.from_numpy(tok).cuda()
for _ in range(2):
    eng.tacotron2_infer(tok_d, max_len=160, early_stopping=False, want_attention=False, precision=prec)
assert eng.last_decoder_mode == 'fused'
tr = np.fromfile(path, dtype=np.int64).reshape(128, 2, 4, 16).astype(np.float64) * 0.01      # us
xn = {0: 'role entry', 1: 'role DMA staged', 2: 'role past #1', 3: 'p1 out', 4: 'p1 polled, p2 computed', 5: 'p2 out', 6: 'p2 quarter polled',
      7: 'role past #2', 8: 'lstm entry', 9: 'lstm past #1', 10: 'lstm early FMAs done', 15: 'p1 dots done',
      11: 'lstm past #2', 12: 'lstm reduced', 13: 'kernel first instruction (role wave 0, block 0)', 14: 'lstm end'}
yn = {0: 'role entry', 1: 'role DMA staged', 2: 'role past #1', 3: 'q out', 4: 'q polled', 5: 'e out', 6: 'e row polled', 7: 'ctx out',
      8: 'ctx quarter polled', 9: 'role past #2', 10: 'lstm entry', 11: 'lstm past #1', 12: 'lstm early FMAs done',
      15: 'q dots done', 13: 'lstm past #2', 14: 'lstm end'}
steps = slice(40, 120)
base = tr[steps, 0, 0, 0][:, None]              # block 0, kernel X, role entry
print('step period (X entry to next X entry, block 0): %.2f us' % np.diff(tr[steps, 0, 0, 0]).mean())
for kind, names in ((0, xn), (1, yn)):
    print('kernel', 'XY'[kind], ': event times relative to block 0 X entry, traced blocks 0 / 80 / 200 / 255')
    for k in sorted(names, key=lambda k: np.nanmean(np.where(tr[steps, kind, 0, k] > 0, tr[steps, kind, 0, k] - base[:, 0], np.nan)) if (tr[steps, kind, 0, k] > 0).any() else 1e9):
        row = []
        for bi in range(4):
            v = tr[steps, kind, bi, k]
            row.append(float(np.mean(v[v > 0] - base[v > 0, 0])) if (v > 0).any() else float('nan'))
        print(f'   {names[k]:32s} ' + ' '.join(f'{v:7.2f}' for v in row))
