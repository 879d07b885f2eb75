import sys, threading, time
import numpy as np
sys.path.insert(0, '.')
from text_to_speech_amd import weights
from text_to_speech_amd.config import Tacotron2Config
from text_to_speech_amd.engine import HipEngine
tw = weights.synth_tacotron2(Tacotron2Config(), seed=1234)
engs = []
for _ in range(2):
    e = HipEngine(0); e.load_state(tw); e.finalize(); e.set_decoder_mode('fused'); engs.append(e)
rng = np.random.default_rng(0)
toks = [rng.integers(1, 148, (4, 60)).astype(np.int32), rng.integers(1, 148, (6, 45)).astype(np.int32)]
alone = [e.tacotron2_infer(t, max_len=300, early_stopping=False, want_attention=False) for e, t in zip(engs, toks)]
for rep in range(4):
    res, modes, times = [None, None], [None, None], [0, 0]
    def run(i):
        t0 = time.perf_counter()
        res[i] = engs[i].tacotron2_infer(toks[i], max_len=300, early_stopping=False, want_attention=False)
        times[i] = time.perf_counter() - t0
        modes[i] = engs[i].last_decoder_mode
    th = [threading.Thread(target=run, args=(i,)) for i in range(2)]
    [t.start() for t in th]; [t.join() for t in th]
    d = [float(np.abs(res[i].mel - alone[i].mel).max()) for i in range(2)]
    print('rep', rep, modes, ['%.1f ms' % (1e3 * t) for t in times], d, flush=True)
