import sys, time
import numpy as np
sys.path.insert(0, '.')
import torch
from text_to_speech_amd import config, weights
from text_to_speech_amd.engine import HipEngine
e = HipEngine(0)
e.load_state(weights.synth_waveglow(config.WaveGlowConfig()))
e.finalize()
mel = np.random.default_rng(1).uniform(-11.5, 1.2, (8, 800, 80)).astype(np.float32)
z = np.random.default_rng(2).standard_normal((8, 25600, 8)).astype(np.float32)
md, zd = torch.from_numpy(mel).cuda(), torch.from_numpy(z).cuda()
for prec in ('f32', 'f16x3', 'f16'):
    for name, a, b in (('device', md, zd), ('host', mel, z)):
        for _ in range(2): e.waveglow_infer(a, z=b, precision=prec)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): e.waveglow_infer(a, z=b, precision=prec)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        print(f'{prec} {name}: {dt*1e3:.2f} ms/step = {8*800*256/dt/1e6:.3f} M samples/s', flush=True)
