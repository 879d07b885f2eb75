"""Probe: split-fp16 (f16x3) WaveGlow vs the exact fp32 HIP path: error and speed at small sizes and at config 2."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from text_to_speech_amd import config, weights
from text_to_speech_amd.engine import HipEngine
cfg = config.WaveGlowConfig()
e = HipEngine(0)
e.load_state(weights.synth_waveglow(cfg))
e.finalize()
rms = lambda a: float(np.sqrt(np.mean(np.square(np.asarray(a, dtype=np.float64)))))
for B, T in [(1, 8), (2, 13), (1, 100), (2, 128)]:
    mel = np.random.default_rng(7).uniform(-11.5, 1.2, (B, T, 80)).astype(np.float32)
    z = np.random.default_rng(11).standard_normal((B, T * 32, 8)).astype(np.float32)
    o32 = e.waveglow_infer(mel, z=z)
    o3 = e.waveglow_infer(mel, z=z, precision='f16x3')
    print(f'B={B} T={T} rms={rms(o32):.4f} x3_vs_f32 rms={rms(o3-o32):.3e} max={np.abs(o3-o32).max():.3e} finite={np.isfinite(o3).all()}', flush=True)
mel = torch.from_numpy(np.random.default_rng(1).uniform(-11.5, 1.2, (8, 800, 80)).astype(np.float32)).cuda()
z = torch.randn(8, 800 * 32, 8, device='cuda')
outs = {}
for prec in ('f32', 'f16x3', 'f16'):
    for _ in range(2):
        o = e.waveglow_infer(mel, z=z, precision=prec)
    t0 = time.time()
    for _ in range(3):
        o = e.waveglow_infer(mel, z=z, precision=prec)
    dt = (time.time() - t0) / 3
    outs[prec] = o.clone()
    print(f'{prec}: {dt*1e3:.1f} ms/step  {8*800*256/dt/1e6:.2f} M samples/s', flush=True)
for prec in ('f16x3', 'f16'):
    d = (outs[prec] - outs['f32']).double()
    print(f'{prec} vs f32 at config 2: rms diff {float(d.pow(2).mean().sqrt()):.3e}')
e.kernel_timing(True)
e.waveglow_infer(mel, z=z, precision='f16x3')
print('f16x3 in-layer us %.1f' % e.kernel_time_us(0)[0], 'res us %.1f' % e.kernel_time_us(1)[0], 'layer0 us %.1f' % e.kernel_time_us(3)[0])
