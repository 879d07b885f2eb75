"""Edge shapes of the persistent decoder against the per-step graph path: 1-2 tokens, 1 step, every row stopping on the
first frames (gate bias +3), a row that never stops among rows that do, window attention on very short inputs."""
import sys
import numpy as np
sys.path.insert(0, '.')
from text_to_speech_amd import weights
from text_to_speech_amd.config import Tacotron2Config
from text_to_speech_amd.engine import HipEngine

rng = np.random.default_rng(0)
n = 0
for gate_bias in (3.0, 0.0, -1.0):
    eng = HipEngine(0)
    eng.load_state(weights.synth_tacotron2(Tacotron2Config(), seed=1234, gate_bias=gate_bias))
    eng.finalize()
    for B in (1, 2, 3, 4):
        for Tin in (1, 2, 3, 31, 64, 65, 128):
            for T in (1, 2, 3, 33):
                for early in (True, False):
                    for win in (None, 1, 5):
                        tok = rng.integers(1, 148, (B, Tin)).astype(np.int32)
                        lens = rng.integers(1, Tin + 1, B)
                        lens[0] = Tin
                        for b in range(B):
                            tok[b, lens[b]:] = 0
                        masks = (rng.random((B, T, 2, 256)) >= 0.5).astype(np.float32) * 2.0
                        kw = dict(max_len=T, early_stopping=early, prenet_masks=masks, attn_mask_win_len=win)
                        outs = {}
                        for mode in ('persistent', 'graph'):
                            eng.set_decoder_mode(mode)
                            outs[mode] = eng.tacotron2_infer(tok, **kw)
                            if mode == 'persistent':
                                assert eng.last_decoder_mode == 'persistent', (B, Tin, T)
                        p, g = outs['persistent'], outs['graph']
                        assert p.lengths.tolist() == g.lengths.tolist(), (gate_bias, B, Tin, T, early, win, p.lengths, g.lengths)
                        for k in ('decoder_output', 'mel', 'stop_tokens', 'attention_weights'):
                            a, b_ = getattr(p, k), getattr(g, k)
                            assert a.shape == b_.shape, (k, a.shape, b_.shape)
                            d = float(np.abs(a - b_).max()) if a.size else 0.0
                            assert np.isfinite(d) and d < 2e-4, (gate_bias, B, Tin, T, early, win, k, d)
                        n += 1
    eng.close()
    print(f'gate_bias {gate_bias}: {n} cases ok so far', flush=True)
print('done', n)
