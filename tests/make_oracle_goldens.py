"""Writes tests/golden/oracle_{waveglow,tacotron2}_small.npz: seed-recorded inputs and outputs of the numpy oracle on the
seeded synthetic weights.  They pin the ORACLE against drift (CPU test) and give the HIP path
a committed fixture to match (GPU test); they are not reference outputs -- WaveGlow / Tacotron2 parity stays "unpinned"
with respect to the reference, which cannot run here.

usage: python tests/make_oracle_goldens.py
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # tests/ -> repo root
sys.path.insert(0, ROOT)
from oracle import tacotron2_ref, waveglow_ref                      # noqa: E402
from text_to_speech_amd import weights                              # noqa: E402
from text_to_speech_amd.config import Tacotron2Config, WaveGlowConfig   # noqa: E402

out_dir = os.path.join(ROOT, 'tests', 'golden')


def digest(w):
    h = hashlib.sha256()
    for k in sorted(w):
        h.update(k.encode())
        h.update(np.ascontiguousarray(w[k]).tobytes())
    return h.hexdigest()


# ---- WaveGlow: B = 2, T = 5 frames, seeded mel / z, sigma 0.8
cfg = WaveGlowConfig()
w = weights.synth_waveglow(cfg, seed=1234)
mel = np.random.default_rng(101).uniform(-11.5, 1.2, (2, 5, 80)).astype(np.float32)
z = np.random.default_rng(102).standard_normal((2, 5 * 32, 8)).astype(np.float32)
audio = waveglow_ref.infer(mel, w, cfg, z=z, sigma=0.8)
np.savez_compressed(os.path.join(out_dir, 'oracle_waveglow_small.npz'), mel=mel, z=z, sigma=np.float32(0.8),
                    audio=audio.astype(np.float32), weights_seed=1234, weights_sha256=digest(w))
print('waveglow', audio.shape, float(np.sqrt((audio ** 2).mean())))

# ---- Tacotron2: B = 2 ragged tokens, 14 decoder steps, explicit prenet dropout masks
tcfg = Tacotron2Config()
tw = weights.synth_tacotron2(tcfg, seed=1234)
rng = np.random.default_rng(103)
tok = rng.integers(1, 148, (2, 19)).astype(np.int32)
tok[1, 11:] = 0
masks = (np.random.default_rng(104).random((2, 14, 2, 256)) >= 0.5).astype(np.float32) * 2.0
ref = tacotron2_ref.infer(tok, tw, tcfg, max_length=14, early_stopping=False, prenet_masks=masks)
np.savez_compressed(os.path.join(out_dir, 'oracle_tacotron2_small.npz'), tokens=tok, prenet_masks=masks,
                    mel=ref.mel.astype(np.float32), decoder_output=ref.decoder_output.astype(np.float32),
                    stop_tokens=ref.stop_tokens.astype(np.float32),
                    attention_weights=ref.attention_weights.astype(np.float32), lengths=ref.lengths.astype(np.int32),
                    weights_seed=1234, weights_sha256=digest(tw))
print('tacotron2', ref.mel.shape, ref.lengths)
