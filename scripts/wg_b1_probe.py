"""Single-sentence WaveGlow (batch 1): step time and in-layer / residual GEMM time per launch for a few lengths and every
precision mode, against the batch-8 rate (how much of a single sentence's cost is tile padding and block-count rounding)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from text_to_speech_amd import config, weights
from text_to_speech_amd.engine import HipEngine
e = HipEngine(0)
e.load_state(weights.synth_waveglow(config.WaveGlowConfig()))
e.finalize()
Ts = [int(t) for t in sys.argv[1:]] or [200, 300, 400, 513, 640, 800]
for prec in ('f16', 'f16x3', 'f32'):
    for T in Ts:
        mel = torch.from_numpy(np.random.default_rng(1).uniform(-11.5, 1.2, (1, T, 80)).astype(np.float32)).cuda()
        for _ in range(2):
            e.waveglow_infer(mel, precision=prec)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 5
        for _ in range(n):
            e.waveglow_infer(mel, precision=prec)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        e.kernel_timing(True)
        e.waveglow_infer(mel, precision=prec)
        us0, n0 = e.kernel_time_us(0)
        us1, n1 = e.kernel_time_us(1)
        e.kernel_timing(False)
        print(f'{prec} T={T}: {dt * 1e3:.2f} ms = {dt * 1e6 / T:.1f} us/frame; in-layer {us0:.0f} us x {n0}, residual {us1:.0f} us x {n1}; '
              f'GEMMs = {(us0 * n0 + us1 * n1) / 1e3:.2f} ms', flush=True)
