"""Does vocoding two single sentences at once (two engine handles = two HIP streams, two threads) beat doing them one after
the other?  Batch-1 WaveGlow leaves CUs idle in the last wave of blocks of every kernel; a second stream can use them."""
import os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from text_to_speech_amd import config, weights
from text_to_speech_amd.engine import HipEngine
w = weights.synth_waveglow(config.WaveGlowConfig())
engs = []
for _ in range(3):
    e = HipEngine(0)
    e.load_state(w)
    e.finalize()
    engs.append(e)
for prec in ('f16', 'f16x3', 'f32'):
    for T in (300, 513, 800):
        mel = torch.from_numpy(np.random.default_rng(1).uniform(-11.5, 1.2, (1, T, 80)).astype(np.float32)).cuda()
        for e in engs:
            e.waveglow_infer(mel, precision=prec)
        n = 6
        res = {}
        for lanes in (1, 2, 3):
            def work(e, k):
                for _ in range(k):
                    e.waveglow_infer(mel, precision=prec)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ths = [threading.Thread(target=work, args=(engs[i], n // lanes)) for i in range(lanes)]
            for t in ths: t.start()
            for t in ths: t.join()
            torch.cuda.synchronize()
            res[lanes] = (time.perf_counter() - t0) / n
        print(f'{prec} T={T}: per sentence ' + ', '.join(f'{k} lane(s) {v * 1e3:.2f} ms' for k, v in res.items()), flush=True)
