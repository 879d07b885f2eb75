"""Step time and in-layer GEMM time of the fp16 / split-fp16 WaveGlow modes at config 2 (for comparing library builds:
TTS_HIP_LIBRARY=... python scripts/abl_time.py)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from text_to_speech_amd import config, weights
from text_to_speech_amd.engine import HipEngine
e = HipEngine(0)
e.load_state(weights.synth_waveglow(config.WaveGlowConfig()))
e.finalize()
mel = torch.from_numpy(np.random.default_rng(1).uniform(-11.5, 1.2, (8, 800, 80)).astype(np.float32)).cuda()
z = torch.randn(8, 800 * 32, 8, device='cuda')
for prec in sys.argv[1:] or ('f16', 'f16x3'):
    for _ in range(2):
        e.waveglow_infer(mel, z=z, precision=prec)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        e.waveglow_infer(mel, z=z, precision=prec)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    e.kernel_timing(True)
    e.waveglow_infer(mel, z=z, precision=prec)
    us, n = e.kernel_time_us(0)
    us1, n1 = e.kernel_time_us(1)        # residual GEMM
    us2, n2 = e.kernel_time_us(2)        # debug builds with TTS_TIME_SPLIT_DIL=1: the layers with dilation >= 32
    e.kernel_timing(False)
    print(f"{os.environ.get('TTS_HIP_LIBRARY', 'default')} {prec}: {dt * 1e3:.1f} ms/step, in-layer {us:.0f} us x {n}, residual {us1:.0f} us x {n1}"
          + (f", dilation >= 32: {us2:.0f} us x {n2}" if n2 else ''), flush=True)
