"""Stress run of the fused two-kernel Tacotron2 decoder step: random call shapes, with and without a second engine keeping the GPU busy
with WaveGlow launches from another thread (the stream(overlap=True) situation).  Every call is compared with the per-step
graph path of the same engine on the same inputs (two independent HIP implementations of the loop).
usage: python scripts/fused_stress.py [iterations] [seed]"""
import sys, threading, time
import numpy as np
sys.path.insert(0, '.')
import torch
from text_to_speech_amd import weights
from text_to_speech_amd.config import Tacotron2Config, WaveGlowConfig
from text_to_speech_amd.engine import HipEngine

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
tw = weights.synth_tacotron2(Tacotron2Config(), seed=1234, gate_bias=-1.0)     # some rows do fire their stop token
eng = HipEngine(0)
eng.load_state(tw)
eng.finalize()
busy = HipEngine(0)
busy.load_state(weights.synth_waveglow(WaveGlowConfig()))
busy.finalize()
mel = torch.from_numpy(np.random.default_rng(1).uniform(-11.5, 1.2, (2, 120, 80)).astype(np.float32)).cuda()

stop = threading.Event()
launches = [0]


def hammer():
    while not stop.is_set():
        busy.waveglow_infer(mel, precision='f16')
        launches[0] += 1


worst, ran = 0.0, {'fused': 0, 'graph': 0, 'persistent': 0}
free0 = torch.cuda.mem_get_info()[0]
t_start = time.time()
for phase, contended in (('alone', False), ('contended', True), ('contended again', True), ('alone again', False)):
    th = None
    if contended:
        stop.clear()
        th = threading.Thread(target=hammer, daemon=True)
        th.start()
    for it in range(iters):
        B = int(rng.integers(1, 9))
        Tin = int(rng.integers(3, 260))
        T = int(rng.integers(2, 160))
        lens = rng.integers(1, Tin + 1, B)
        lens[int(rng.integers(0, B))] = Tin
        tok = rng.integers(1, 148, (B, Tin)).astype(np.int32)
        for b in range(B):
            tok[b, lens[b]:] = 0
        masks = (rng.random((B, T, 2, 256)) >= 0.5).astype(np.float32) * 2.0 if rng.random() < 0.7 else None
        early = bool(rng.random() < 0.5)
        prec = 'f16' if rng.random() < 0.3 else 'f32'
        win = int(rng.integers(5, 40)) if rng.random() < 0.3 else None
        kw = dict(max_len=T, early_stopping=early, prenet_masks=masks, precision=prec, attn_mask_win_len=win)
        outs = {}
        for mode in ('fused', 'graph'):
            eng.set_decoder_mode(mode)
            outs[mode] = eng.tacotron2_infer(tok, **kw)
            if mode == 'fused':
                ran[eng.last_decoder_mode] += 1
        p, g = outs['fused'], outs['graph']
        assert p.lengths.tolist() == g.lengths.tolist(), (it, B, Tin, T, p.lengths, g.lengths)
        for k in ('decoder_output', 'mel', 'stop_tokens', 'attention_weights'):
            d = float(np.abs(getattr(p, k) - getattr(g, k)).max())
            # fp16 mode: both paths stream the same fp16 rows; the gate non-linearities differ (v_exp / v_rcp against libm):
            # within the fp16 tolerance of the oracle (5e-3)
            assert np.isfinite(d) and d < (2e-4 if prec == 'f32' else 3e-3), (phase, it, B, Tin, T, early, prec, win, k, d)
            if prec == 'f32':
                worst = max(worst, d)
        if it % 50 == 49:
            print(f'{phase}: {it + 1} calls ok, worst diff {worst:.2e}, fused requests served by {ran}, '
                  f'waveglow calls alongside {launches[0]}, {time.time() - t_start:.0f} s', flush=True)
    if th is not None:
        stop.set()
        th.join()
    print(f'{phase}: device memory in use +{(free0 - torch.cuda.mem_get_info()[0]) / 2**20:.0f} MiB since start', flush=True)
print(f'done: worst diff {worst:.2e}; fused requests served by {ran}; device memory in use grew by '
      f'{(free0 - torch.cuda.mem_get_info()[0]) / 2**20:.0f} MiB over the run (workspaces are high-water-mark arenas, graphs an LRU of 16)')
