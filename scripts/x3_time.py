"""One split-fp16 (or other precision: argv[1]) WaveGlow call at config 2, for profiling."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from text_to_speech_amd import config, weights
from text_to_speech_amd.engine import HipEngine
prec = sys.argv[1] if len(sys.argv) > 1 else 'f16x3'
e = HipEngine(0)
e.load_state(weights.synth_waveglow(config.WaveGlowConfig()))
e.finalize()
mel = torch.from_numpy(np.random.default_rng(1).uniform(-11.5, 1.2, (8, 800, 80)).astype(np.float32)).cuda()
z = torch.randn(8, 800 * 32, 8, device='cuda')
e.waveglow_infer(mel, z=z, precision=prec)
e.waveglow_infer(mel, z=z, precision=prec)
print('done', prec)
