"""Wall time of one 800-frame vocoding call through `HipRuntime` with the noise drawn on the device (default since round 3)
against the round-2 behaviour (z built by numpy on the host and shipped over PCIe), and the same for a Tacotron2 decode with
prenet dropout masks.  usage: python scripts/noise_probe.py"""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import torch
from text_to_speech_amd.runtime import HipRuntime, sample_prenet_masks

rt = HipRuntime('synthetic', seed=3)
eng = rt.engine
mel = np.random.default_rng(0).uniform(-11.5, 1.2, (1, 800, 80)).astype(np.float32)
mel_d = torch.from_numpy(mel).cuda()
rng = np.random.default_rng(1)


def timed(fn, n=5):
    fn()
    best = 1e9
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


for prec in ('f32', 'f16'):
    host_z = lambda: eng.waveglow_infer(mel, z=rng.standard_normal((1, 800 * 32, 8)).astype(np.float32), precision=prec)
    dev_z = lambda: rt.waveglow_infer(mel, precision=prec)
    host_z_dev_mel = lambda: eng.waveglow_infer(mel_d, z=torch.from_numpy(rng.standard_normal((1, 800 * 32, 8)).astype(np.float32)).cuda(), precision=prec)
    dev_z_dev_mel = lambda: rt.waveglow_infer(mel_d, precision=prec)
    print(f'WaveGlow 1 x 800 frames {prec}: host mel: numpy z {timed(host_z):.2f} ms -> device z {timed(dev_z):.2f} ms; '
          f'device mel: numpy z {timed(host_z_dev_mel):.2f} ms -> device z {timed(dev_z_dev_mel):.2f} ms', flush=True)

tok = np.zeros((1, 128), np.int32)
tok[0, :100] = np.random.default_rng(5).integers(1, 148, 100)
host_masks = lambda: eng.tacotron2_infer(tok, max_len=800, early_stopping=False, want_attention=False,
                                         prenet_masks=sample_prenet_masks(rng, 1, 800))
dev_masks = lambda: rt.tacotron2_infer(tok, max_length=800, early_stopping=False)
print(f'Tacotron2 1 x 800 steps: numpy masks {timed(host_masks):.2f} ms -> device masks {timed(dev_masks):.2f} ms', flush=True)
