"""Probe: Tacotron2 decode and WaveGlow vocoding running concurrently from two engine handles (two HIP streams) on one
GPU -- how much does each slow down?  (Decides whether sentence-level pipelining of stream() is worth building.)"""
import os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from text_to_speech_amd import weights
from text_to_speech_amd.config import Tacotron2Config, WaveGlowConfig
from text_to_speech_amd.engine import HipEngine

prec = sys.argv[1] if len(sys.argv) > 1 else 'f16'
T = 600
et, ew = HipEngine(0), HipEngine(0)
et.load_state(weights.synth_tacotron2(Tacotron2Config(), seed=1234)); et.finalize()
ew.load_state(weights.synth_waveglow(WaveGlowConfig())); ew.finalize()
tok = np.zeros((1, 128), np.int32); tok[:, :100] = np.random.default_rng(5).integers(1, 148, (1, 100))
tok_d = torch.from_numpy(tok).cuda()
mel = torch.rand((1, T, 80), device='cuda') * 12.7 - 11.5
z = torch.randn((1, T * 32, 8), device='cuda')

def taco(n):
    for _ in range(n):
        et.tacotron2_infer(tok_d, max_len=T, early_stopping=False, want_attention=False)

def wg(n):
    for _ in range(n):
        ew.waveglow_infer(mel, z=z, precision=prec)

taco(1); wg(1)
N = 6
t0 = time.perf_counter(); taco(N); t_t = (time.perf_counter() - t0) / N
t0 = time.perf_counter(); wg(N); t_w = (time.perf_counter() - t0) / N
print(f'alone: tacotron2 {t_t*1e3:.1f} ms, waveglow[{prec}] {t_w*1e3:.1f} ms, sequential {1e3*(t_t+t_w):.1f} ms per sentence', flush=True)
res = {}
def timed(name, fn):
    t0 = time.perf_counter(); fn(N); res[name] = (time.perf_counter() - t0) / N
a = threading.Thread(target=timed, args=('t', taco)); b = threading.Thread(target=timed, args=('w', wg))
t0 = time.perf_counter(); a.start(); b.start(); a.join(); b.join(); tot = (time.perf_counter() - t0) / N
print(f'concurrent: tacotron2 {res["t"]*1e3:.1f} ms, waveglow {res["w"]*1e3:.1f} ms, wall per sentence pair {tot*1e3:.1f} ms', flush=True)
