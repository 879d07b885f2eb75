"""Phase breakdown of the persistent decoder from the debug build's timestamps (TTS_PERSIST_TRACE_FILE, 100 MHz clock).
usage: TTS_HIP_LIBRARY=text_to_speech_amd/libtts_hip_dbg.so TTS_PERSIST_TRACE_FILE=/tmp/tr.bin python scripts/persist_trace.py [B]"""
import os, sys
import numpy as np
sys.path.insert(0, '.')
import torch
from text_to_speech_amd import weights
from text_to_speech_amd.config import Tacotron2Config
from text_to_speech_amd.engine import HipEngine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
path = os.environ['TTS_PERSIST_TRACE_FILE']
eng = HipEngine(0)
eng.load_state(weights.synth_tacotron2(Tacotron2Config(), seed=1234))
eng.finalize()
tok = np.zeros((B, 128), np.int32)
tok[:, :100] = np.random.default_rng(5).integers(1, 148, (B, 100))
tok_d = torch.from_numpy(tok).cuda()
for _ in range(2):
    eng.tacotron2_infer(tok_d, max_len=300, early_stopping=False, want_attention=False)
assert eng.last_decoder_mode == 'persistent'
tr = np.fromfile(path, dtype=np.int64).reshape(4, 256, 20).astype(np.float64) * 0.01      # us
names = {0: 'top', 1: 'A synced', 2: 'A done (p1 out)', 3: 'recurrent gemv done', 4: 'C synced', 5: 'h_att out', 6: 'D synced',
         7: 'D gemv done', 8: 'F synced', 9: 'h_dec out', 10: 'w2 B arrive', 11: 'w2 p1 polled', 12: 'w2 p2 out',
         13: 'w3 E arrive', 14: 'w3 q polled', 15: 'w3 e out', 16: 'w3 loc done'}
steps = slice(50, 250)
# absolute picture: every event relative to block 0's loop top of the same step (the clock is global)
print('event times relative to block 0 top, per traced block (0, 80, 200, 255):')
base0 = tr[0, steps, 0:1]
for k in sorted(names, key=lambda k: (tr[0, steps, k:k + 1] - base0).mean()):
    row = [float((tr[bi, steps, k:k + 1] - base0).mean()) for bi in range(4)]
    print(f'   {names[k]:22s} ' + ' '.join(f'{v:7.2f}' for v in row))
for bi, blk in enumerate((0,)):
    t = tr[bi, steps]
    base = t[:, 0:1]
    rel = t - base
    period = np.diff(tr[bi, steps, 0]).mean()
    print(f'block {blk}: step period {period:.2f} us')
    order = sorted(names, key=lambda k: rel[:, k].mean())
    prev = 0.0
    for k in order:
        m = rel[:, k].mean()
        print(f'   {names[k]:22s} at {m:6.2f} us  (+{m - prev:5.2f})')
        prev = m
