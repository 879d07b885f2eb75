"""Config 2 (8 x 800 frames, fp32) as ONE call on one handle against TWO half-batch calls (4 x 800 each) on two handles / two
HIP streams / two threads: do the HBM-bound passes of one half (Winograd pre-pass, combine + gate, end fold) run under the
MFMA-bound GEMMs of the other?"""
import os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from text_to_speech_amd import config, weights
from text_to_speech_amd.engine import HipEngine
w = weights.synth_waveglow(config.WaveGlowConfig())
engs = []
for _ in range(2):
    e = HipEngine(0)
    e.load_state(w)
    e.finalize()
    engs.append(e)
mel = torch.from_numpy(np.random.default_rng(1).uniform(-11.5, 1.2, (8, 800, 80)).astype(np.float32)).cuda()
z = torch.randn(8, 800 * 32, 8, device='cuda')
halves = [(mel[:4].contiguous(), z[:4].contiguous()), (mel[4:].contiguous(), z[4:].contiguous())]
for form in ('winograd', 'direct'):
    for e in engs:
        e.set_waveglow_form(form)
    engs[0].waveglow_infer(mel, z=z)
    for e, (m, zz) in zip(engs, halves):
        e.waveglow_infer(m, z=zz)
    torch.cuda.synchronize()
    n = 3
    t0 = time.perf_counter()
    for _ in range(n):
        engs[0].waveglow_infer(mel, z=z)
    torch.cuda.synchronize()
    one = (time.perf_counter() - t0) / n

    def work(e, m, zz):
        for _ in range(n):
            e.waveglow_infer(m, z=zz)
    t0 = time.perf_counter()
    ths = [threading.Thread(target=work, args=(e, m, zz)) for e, (m, zz) in zip(engs, halves)]
    for t in ths: t.start()
    for t in ths: t.join()
    torch.cuda.synchronize()
    two = (time.perf_counter() - t0) / n
    print(f'{form}: one call of 8 rows {one * 1e3:.1f} ms; two concurrent calls of 4 rows {two * 1e3:.1f} ms', flush=True)
