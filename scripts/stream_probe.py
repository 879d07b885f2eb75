"""BASELINE.json configs[4] shape: 64 sentences streamed one by one (token counts cycling 50..200), Tacotron2 decode in
hipGraph chunks, WaveGlow in the fp16-operand mode; sequential vs sentence-pipelined (overlap=True, two engine handles).
Synthetic weights never fire the stop token: every sentence decodes max_length = 4 frames per token."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from text_to_speech_amd import weights
from text_to_speech_amd.config import Tacotron2Config, WaveGlowConfig
from text_to_speech_amd.engine import HipEngine
from text_to_speech_amd.runtime import HipRuntime
from text_to_speech_amd.tacotron2 import Tacotron2
from text_to_speech_amd.waveglow import WaveGlow

prec = sys.argv[1] if len(sys.argv) > 1 else 'f16'
n_sent = int(sys.argv[2]) if len(sys.argv) > 2 else 64
sampled = len(sys.argv) > 3 and sys.argv[3] == 'sampled'      # prenet dropout masks + WaveGlow noise sampled per call
dec_mode = sys.argv[4] if len(sys.argv) > 4 else 'auto'       # decoder machine: auto (persistent at batch 1) | graph | ...
e1, e2 = HipEngine(0), HipEngine(0)
tw, ww = weights.synth_tacotron2(Tacotron2Config(), seed=1234), weights.synth_waveglow(WaveGlowConfig())
for e in (e1, e2):
    e.load_state(tw); e.load_state(ww); e.finalize()
e1.set_decoder_mode(dec_mode)
model = Tacotron2(HipRuntime('t', model='tacotron2', engine=e1, seed=0, synthesizer_precision='f16' if prec == 'f16' else 'f32'))
voc_same = WaveGlow(HipRuntime('w', model='waveglow', engine=e1, seed=0, vocoder_precision=prec))
voc_own = WaveGlow(HipRuntime('w2', model='waveglow', engine=e2, seed=0, vocoder_precision=prec))
rng = np.random.default_rng(0)
letters = np.array(list('abcdefghijklmnopqrstuvwxyz     '))
lens = [50, 70, 90, 110, 130, 150, 170, 200]
texts = [''.join(rng.choice(letters, lens[i % 8])).strip() + f' {i}.' for i in range(n_sent)]
kw = dict(max_length=4., deterministic=not sampled, save=False, return_results=False)
model.predict(texts[:2], vocoder=voc_same, **kw); model.predict(texts[:2], vocoder=voc_own, overlap=True, **kw)
for name, v, ov in (('sequential', voc_same, False), ('overlapped', voc_own, True)):
    secs = []
    t0 = time.perf_counter()
    model.predict(texts, vocoder=v, overlap=ov, callbacks=[lambda time, **_: secs.append(time)], **kw)
    dt = time.perf_counter() - t0
    print(f'{name} [{prec}{", sampled" if sampled else ""}, decoder {dec_mode} -> {e1.last_decoder_mode}]: {n_sent} sentences, {sum(secs):.1f} s of audio in {dt*1e3:.0f} ms = {sum(secs)/dt:.0f}x real time', flush=True)
