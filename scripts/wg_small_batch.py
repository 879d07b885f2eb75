"""WaveGlow throughput vs utterance length at batch 1 (configs 1 / 5 run sentence by sentence)."""
import sys, time
import numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import torch
from text_to_speech_amd import config, weights
from text_to_speech_amd.engine import HipEngine
e = HipEngine(0)
e.load_state(weights.synth_waveglow(config.WaveGlowConfig()))
e.finalize()
for prec in ('f32', 'f16x3', 'f16'):
    for B, T in [(1, 50), (1, 100), (1, 170), (1, 200), (1, 300), (1, 400), (1, 600), (1, 800)]:
        mel = torch.rand((B, T, 80), device='cuda') * 12.7 - 11.5
        z = torch.randn((B, T * 32, 8), device='cuda')
        for _ in range(2):
            e.waveglow_infer(mel, z=z, precision=prec)
        n = 5 if B * T < 3000 else 2
        t0 = time.perf_counter()
        for _ in range(n):
            e.waveglow_infer(mel, z=z, precision=prec)
        dt = (time.perf_counter() - t0) / n
        print(f'{prec} B={B} T={T}: {dt*1e3:8.2f} ms  {B*T*256/dt/1e6:6.2f} M samples/s  ({B*T*256/22050/dt:6.1f}x RT)', flush=True)
