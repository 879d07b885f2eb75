"""One WaveGlow call at batch 1 (argv: T, precision) under rocprofv3 --kernel-trace: used to compare the sum of kernel
durations with the wall time of the call (launch gaps)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from text_to_speech_amd import config, weights
from text_to_speech_amd.engine import HipEngine
T = int(sys.argv[1]) if len(sys.argv) > 1 else 170
prec = sys.argv[2] if len(sys.argv) > 2 else 'f16'
e = HipEngine(0)
e.load_state(weights.synth_waveglow(config.WaveGlowConfig()))
e.finalize()
mel = torch.rand((1, T, 80), device='cuda') * 12.7 - 11.5
z = torch.randn((1, T * 32, 8), device='cuda')
for _ in range(3):
    e.waveglow_infer(mel, z=z, precision=prec)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    e.waveglow_infer(mel, z=z, precision=prec)
print(f'T={T} {prec}: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms per call', flush=True)
