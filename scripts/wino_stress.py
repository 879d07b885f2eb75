"""Random call shapes through the fp32 vocoder in both forms of the WN convolutions (Winograd F(4,3) vs direct) on the same
inputs: batch 1 - 12, 30 - 900 frames per utterance (partial frame groups, padded phase blocks, 128- and 256-row tiles), noise
given or not.  HIP against HIP: the two forms must agree within fp32 rounding on every shape, and the one-kernel Winograd form must EQUAL the
three-pass form of round 3; which form ran is reported."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from text_to_speech_amd import weights
from text_to_speech_amd.config import WaveGlowConfig
from text_to_speech_amd.engine import HipEngine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
eng = HipEngine(0)
eng.load_state(weights.synth_waveglow(WaveGlowConfig(), seed=1234)); eng.finalize()
rng = np.random.default_rng(2024)
worst, ran = 0.0, {'winograd': 0, 'direct': 0}
for it in range(n):
    B = int(rng.integers(1, 13))
    T = int(rng.integers(30, 901))
    if B * T > 9000:
        T = max(30, 9000 // B)
    mel = torch.from_numpy(rng.uniform(-11.5, 1.2, (B, T, 80)).astype(np.float32)).cuda()
    z = torch.from_numpy(rng.standard_normal((B, T * 32, 8)).astype(np.float32)).cuda() if it % 4 else None
    outs = {}
    for form in ('winograd', 'direct', 'winograd-3pass'):
        eng.set_waveglow_form(form)
        outs[form] = eng.waveglow_infer(mel, z=z)
        if form == 'winograd':
            ran[eng.last_waveglow_form] += 1
            took = eng.last_waveglow_form
    # round 4: the one-kernel Winograd form against round 3's three passes -- the same products in the same order: bit for bit
    # (up to 512 frames per call the two run on different phase-block heights -- 64 / 128 rows -- i.e. with first-layer and residual
    #  kernels of different tile shapes: reported, not asserted)
    same = torch.equal(outs['winograd'], outs['winograd-3pass'])
    if B * T > 512:
        assert same, (B, T, 'fused kernel differs from the three passes')
    elif not same:
        print(f'     (B*T = {B * T}: fused vs three-pass rms diff {float(torch.sqrt(torch.mean((outs["winograd"] - outs["winograd-3pass"]) ** 2))):.2e})')
    assert bool(torch.isfinite(outs['winograd']).all()) and outs['winograd'].shape == (B, T * 256)
    d = float(torch.sqrt(torch.mean((outs['winograd'] - outs['direct']) ** 2)))
    m = float((outs['winograd'] - outs['direct']).abs().max())
    worst = max(worst, d)
    print(f'{it:3d}: B {B:2d} T {T:3d} ({B * T:4d} frames, z {"given" if z is not None else "none "}) -> {took:8s} rms diff {d:.2e} max {m:.2e}', flush=True)
    assert d <= 5e-6, (B, T, d)
print(f'{n} shapes: worst rms difference {worst:.2e}; default form ran {ran}')
