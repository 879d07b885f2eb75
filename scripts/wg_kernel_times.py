"""Prints the average duration of the WaveGlow GEMM kinds (HIP events) for the bench workload."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from text_to_speech_amd import weights
from text_to_speech_amd.config import WaveGlowConfig
from text_to_speech_amd.engine import HipEngine
B, T = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (8, 800)
eng = HipEngine(0)
eng.load_state(weights.synth_waveglow(WaveGlowConfig(), seed=1234)); eng.finalize()
mel = torch.from_numpy(np.random.default_rng(7).uniform(-11.5, 1.2, (B, T, 80)).astype(np.float32)).cuda()
z = torch.from_numpy(np.random.default_rng(11).standard_normal((B, T * 32, 8)).astype(np.float32)).cuda()
eng.waveglow_infer(mel, z=z)
eng.kernel_timing(True)
import time; t0 = time.perf_counter()
for _ in range(2): eng.waveglow_infer(mel, z=z)
dt = (time.perf_counter() - t0) / 2
for kind, name, flops in ((0, 'in-layer K=1856', 2.0*B*T*32*1856*1024), (3, 'in-layer-0 K=368', 2.0*B*T*32*368*1024), (1, 'residual N=512', 2.0*B*T*32*512*512)):
    us, n = eng.kernel_time_us(kind)
    print(f'{name:18s} {n:4d} launches  avg {us:8.1f} us  {flops/us/1e6:6.1f} TFLOP/s')
print(f'B {B} T {T}: step {dt*1e3:.1f} ms')
