"""Probe: fp16-operand WaveGlow vs the exact fp32 HIP path (error + speed).  Parity against the oracle lives in tests/."""
import sys, time
import numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from text_to_speech_amd import config, weights
from text_to_speech_amd.engine import HipEngine

cfg = config.WaveGlowConfig()
w = weights.synth_waveglow(cfg)
e = HipEngine(0)
e.load_state(w)
e.finalize()
rms = lambda a: float(np.sqrt(np.mean(np.square(a, dtype=np.float64))))
for B, T in [(1, 8), (2, 13)]:
    mel = np.random.default_rng(7).uniform(-11.5, 1.2, (B, T, 80)).astype(np.float32)
    z = np.random.default_rng(11).standard_normal((B, T * 32, 8)).astype(np.float32)
    o32 = e.waveglow_infer(mel, z=z)
    ref = o32
    o16 = e.waveglow_infer(mel, z=z, precision='f16')
    print(f'B={B} T={T} ref_rms={rms(ref):.4f} f16_vs_f32={rms(o16-ref):.3e} '
          f'f16_max={np.abs(o16-ref).max():.3e} finite={np.isfinite(o16).all()}', flush=True)
import torch
mel = torch.from_numpy(np.random.default_rng(1).uniform(-11.5, 1.2, (8, 800, 80)).astype(np.float32)).cuda()
z = torch.randn(8, 800 * 32, 8, device='cuda')
for prec in ('f32', 'f16'):
    for _ in range(2):
        o = e.waveglow_infer(mel, z=z, precision=prec)
    t0 = time.time()
    for _ in range(3):
        o = e.waveglow_infer(mel, z=z, precision=prec)
    dt = (time.time() - t0) / 3
    print(f'{prec}: {dt*1e3:.1f} ms/step  {8*800*256/dt/1e6:.2f} M samples/s', flush=True)
    if prec == 'f32':
        o32 = o.clone()
    else:
        print('f16 vs f32 at config 2: rms diff', float((o - o32).double().pow(2).mean().sqrt()), 'rms', float(o32.double().pow(2).mean().sqrt()))
e.kernel_timing(True)
o = e.waveglow_infer(mel, z=z, precision='f16')
print('f16 in-layer us', e.kernel_time_us(0), 'res us', e.kernel_time_us(1), 'layer0 us', e.kernel_time_us(3))
