"""fp32 WaveGlow call time in the Winograd and in the direct form for a list of shapes (argv: B T pairs)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from text_to_speech_amd import weights
from text_to_speech_amd.config import WaveGlowConfig
from text_to_speech_amd.engine import HipEngine
eng = HipEngine(0)
eng.load_state(weights.synth_waveglow(WaveGlowConfig(), seed=1234)); eng.finalize()
shapes = [(int(sys.argv[i]), int(sys.argv[i + 1])) for i in range(1, len(sys.argv) - 1, 2)] or [(1, 800), (1, 513), (1, 300), (2, 400)]
for B, T in shapes:
    mel = torch.from_numpy(np.random.default_rng(7).uniform(-11.5, 1.2, (B, T, 80)).astype(np.float32)).cuda()
    z = torch.from_numpy(np.random.default_rng(11).standard_normal((B, T * 32, 8)).astype(np.float32)).cuda()
    res = {}
    outs = {}
    for form in ('winograd', 'direct'):
        eng.set_waveglow_form(form)
        for _ in range(2): outs[form] = eng.waveglow_infer(mel, z=z)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): eng.waveglow_infer(mel, z=z)
        torch.cuda.synchronize(); res[form] = ((time.perf_counter() - t0) / 3, eng.last_waveglow_form)
    diff = float(torch.sqrt(torch.mean((outs['winograd'] - outs['direct']) ** 2)))
    print(f'B {B} T {T}: ' + ', '.join(f'{k} -> ran {v[1]} {v[0] * 1e3:.1f} ms' for k, v in res.items()) + f'; rms diff {diff:.2e}', flush=True)
