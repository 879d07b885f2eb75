"""The Winograd forms of the fp32 in-layer GEMM on one shape: 2 = three passes (round 3), 3 = fused GEMM behind the pre-pass,
1 = fused GEMM with the input transform in its operand reads (default), 0 = direct form: call time, average in-layer launch
time (HIP events) and the waveform difference from the first form listed.  argv: B T [forms ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from text_to_speech_amd import weights
from text_to_speech_amd.config import WaveGlowConfig
from text_to_speech_amd.engine import HipEngine
B, T = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (8, 800)
forms = [int(a) for a in sys.argv[3:]] or [2, 3, 1, 0]
eng = HipEngine(0)
eng.load_state(weights.synth_waveglow(WaveGlowConfig(), seed=1234)); eng.finalize()
mel = torch.from_numpy(np.random.default_rng(7).uniform(-11.5, 1.2, (B, T, 80)).astype(np.float32)).cuda()
z = torch.from_numpy(np.random.default_rng(11).standard_normal((B, T * 32, 8)).astype(np.float32)).cuda()
base = None
for form in forms:
    eng._check(eng._lib.tts_hip_set_waveglow_form(eng._h, form), 'set_waveglow_form')
    out = eng.waveglow_infer(mel, z=z)
    eng.kernel_timing(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(2): out = eng.waveglow_infer(mel, z=z)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 2
    us, n = eng.kernel_time_us(0)
    us1, n1 = eng.kernel_time_us(1)
    eng.kernel_timing(False)
    if base is None: base = out
    d = out - base
    print(f'form {form}: step {dt * 1e3:7.1f} ms  in-layer {n:4d} launches avg {us:8.1f} us  residual avg {us1:6.1f} us  '
          f'vs first form: rms {float(torch.sqrt(torch.mean(d * d))):.2e} max {float(d.abs().max()):.2e}  finite {bool(torch.isfinite(out).all())}', flush=True)
free, total = torch.cuda.mem_get_info()
print(f'device memory in use {(total - free) / 2**30:.1f} GiB')
