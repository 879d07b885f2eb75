"""How the parity numbers depend on the `end_conv` attenuation of the synthetic WaveGlow weights:
waveform RMS error of every precision mode against the numpy oracle for end_scale in {0.05, 0.2, 0.5, 1.0}."""
import sys
import numpy as np
sys.path.insert(0, '.')
from oracle import waveglow_ref
from text_to_speech_amd import weights
from text_to_speech_amd.config import WaveGlowConfig
from text_to_speech_amd.engine import HipEngine

cfg = WaveGlowConfig()
B, T = 2, 40
mel = np.random.default_rng(7).uniform(-11.5, 1.2, (B, T, 80)).astype(np.float32)
z = np.random.default_rng(11).standard_normal((B, T * 32, 8)).astype(np.float32)
for scale in (0.05, 0.2, 0.5, 1.0):
    w = weights.synth_waveglow(cfg, seed=1234, end_scale=scale)
    eng = HipEngine(0)
    eng.load_state(w)
    eng.finalize()
    ref = waveglow_ref.infer(mel, w, cfg, z=z, sigma=1.0)
    sig = float(np.sqrt(np.mean(ref.astype(np.float64) ** 2)))
    line = f'end_scale={scale}: signal rms {sig:.3g}, max |x| {np.abs(ref).max():.3g};'
    for prec in ('f32', 'f16x3', 'f16'):
        out = eng.waveglow_infer(mel, z=z, sigma=1.0, precision=prec)
        err = float(np.sqrt(np.mean((out.astype(np.float64) - ref) ** 2)))
        line += f' {prec}: rms err {err:.2e} (rel {err / sig:.2e})'
    print(line, flush=True)
    del eng
