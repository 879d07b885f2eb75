"""Small driver used under rocprofv3: one Tacotron2 decode (B utterances, 100 tokens padded to 128, N steps)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from text_to_speech_amd import weights
from text_to_speech_amd.config import Tacotron2Config
from text_to_speech_amd.engine import HipEngine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 256
prec = sys.argv[3] if len(sys.argv) > 3 else 'f32'
mode = sys.argv[4] if len(sys.argv) > 4 else 'persistent'
eng = HipEngine(0)
eng.load_state(weights.synth_tacotron2(Tacotron2Config(), seed=1234))
eng.finalize()
eng.set_decoder_mode(mode)
tok = np.zeros((B, 128), np.int32)
tok[:, :100] = np.random.default_rng(5).integers(1, 148, (B, 100))
eng.tacotron2_infer(tok, max_len=32, early_stopping=False, precision=prec)
t0 = time.perf_counter()
out = eng.tacotron2_infer(tok, max_len=steps, early_stopping=False, precision=prec)
dt = time.perf_counter() - t0
print(f'[{prec}, {eng.last_decoder_mode}] B={B} steps={steps}: {1e6 * dt / steps:.1f} us/step  {B * steps / dt:.0f} frames/s', flush=True)
eng.close()
