"""Kernel timings of the fp16 WaveGlow path at config 2 (uses TTS_HIP_LIBRARY if set)."""
import sys
import numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import torch
from text_to_speech_amd import config, weights
from text_to_speech_amd.engine import HipEngine
cfg = config.WaveGlowConfig()
e = HipEngine(0)
e.load_state(weights.synth_waveglow(cfg))
e.finalize()
mel = torch.from_numpy(np.random.default_rng(1).uniform(-11.5, 1.2, (8, 800, 80)).astype(np.float32)).cuda()
z = torch.randn(8, 800 * 32, 8, device='cuda')
prec = sys.argv[1] if len(sys.argv) > 1 else 'f16'
e.waveglow_infer(mel, z=z, precision=prec)
e.kernel_timing(True)
e.waveglow_infer(mel, z=z, precision=prec)
print(prec, 'in-layer us %.1f' % e.kernel_time_us(0)[0], 'res us %.1f' % e.kernel_time_us(1)[0],
      'layer0 us %.1f' % e.kernel_time_us(3)[0], flush=True)
