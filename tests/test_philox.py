"""The device-side sampling stream: the numpy restatement against the published Philox4x32-10 known-answer vectors (CPU), and
the HIP generator against the restatement (GPU)."""
import numpy as np
import pytest

from oracle import philox_ref


def test_philox4x32_10_known_answers():
    # Random123 kat_vectors (philox4x32, 10 rounds)
    kats = [
        ((0x00000000, 0x00000000, 0x00000000, 0x00000000), (0x00000000, 0x00000000),
         (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
        ((0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff), (0xffffffff, 0xffffffff),
         (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
        ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
         (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
    ]
    for ctr, key, want in kats:
        got = philox_ref.philox4x32_10(np.array([ctr], np.uint32), np.array([key], np.uint32))[0]
        assert [int(v) for v in got] == list(want), (ctr, key, [hex(int(v)) for v in got])


def test_restated_stream_properties():
    z = philox_ref.normal(400_000, seed=7)
    assert abs(float(z.mean())) < 6e-3 and abs(float(z.std()) - 1.0) < 6e-3
    assert abs(float((z ** 3).mean())) < 0.02 and abs(float((z ** 4).mean()) - 3.0) < 0.05
    assert np.isfinite(z).all() and float(np.abs(z).max()) < 6.0            # u >= 2^-25: |z| <= sqrt(2 * 25 ln 2) = 5.9
    m = philox_ref.prenet_masks(400_000, seed=7)
    assert set(np.unique(m).tolist()) == {0.0, 2.0} and abs(float(m.mean()) - 1.0) < 6e-3
    # element i depends on (seed, offset + i // 4, i % 4) only: a later offset continues the same stream
    a = philox_ref.normal(64, seed=3, offset=10)
    b = philox_ref.normal(24, seed=3, offset=20)
    assert np.array_equal(a[40:], b)
    assert not np.array_equal(philox_ref.normal(64, seed=4, offset=10), a)


@pytest.mark.gpu
def test_hip_generator_matches_the_restatement(gpu_engine):
    for n, seed, offset in ((1, 1, 0), (7, 2 ** 40 + 5, 3), (4096, 12345, 2 ** 33), (100_003, 0xDEADBEEFCAFE, 17)):
        got = gpu_engine.random_normal((n,), seed, offset).cpu().numpy()
        ref = philox_ref.normal(n, seed, offset)
        assert np.abs(got - ref).max() <= 2e-5, (n, seed, offset, float(np.abs(got - ref).max()))
        masks = gpu_engine._random(1, (n,), seed, offset, None).cpu().numpy()
        assert np.array_equal(masks, philox_ref.prenet_masks(n, seed, offset))
    m = gpu_engine.random_prenet_masks(3, 50, 9, 4)
    assert tuple(m.shape) == (3, 50, 2, 256)
    assert np.array_equal(m.cpu().numpy().reshape(-1), philox_ref.prenet_masks(3 * 50 * 512, 9, 4))
    # determinism per (seed, offset); another seed gives another stream
    a = gpu_engine.random_normal((2, 96, 8), 5, 0)
    assert np.array_equal(a.cpu().numpy(), gpu_engine.random_normal((2, 96, 8), 5, 0).cpu().numpy())
    assert not np.array_equal(a.cpu().numpy(), gpu_engine.random_normal((2, 96, 8), 6, 0).cpu().numpy())
    z = gpu_engine.random_normal((1_000_000,), 11, 0).cpu().numpy()
    assert abs(float(z.mean())) < 4e-3 and abs(float(z.std()) - 1.0) < 4e-3


@pytest.mark.gpu
def test_seeded_waveglow_equals_explicit_noise_from_the_same_stream(gpu_engine, wg_weights, wg_cfg):
    """`waveglow_infer(mel, seed=...)` draws z on the device; it must equal the explicit-z call with the restated stream (and
    so the oracle on that z), on device and host buffers, and `HipRuntime`'s default path must use it."""
    import torch
    from oracle import waveglow_ref
    from text_to_speech_amd.runtime import HipRuntime
    rng = np.random.default_rng(0)
    B, T = 2, 6
    mel = rng.uniform(-11.5, 1.2, (B, T, 80)).astype(np.float32)
    z = philox_ref.normal(B * T * 32 * 8, seed=77, offset=5).reshape(B, T * 32, 8)
    ref = waveglow_ref.infer(mel, wg_weights, wg_cfg, z=z)
    host = gpu_engine.waveglow_infer(mel, seed=77, offset=5)
    dev = gpu_engine.waveglow_infer(torch.from_numpy(mel).cuda(), seed=77, offset=5).cpu().numpy()
    for got in (host, dev):
        assert float(np.sqrt(np.mean((got - ref) ** 2))) <= 1e-4
    assert np.array_equal(host, dev)
    rt = HipRuntime('unused', engine=gpu_engine, model='waveglow', seed=77)
    first = rt(mel)                                    # offset 0 of seed 77's noise stream
    from text_to_speech_amd.runtime import NOISE_STREAM                    # the runtime's noise key: seed ^ purpose constant
    z0 = philox_ref.normal(B * T * 32 * 8, seed=77 ^ NOISE_STREAM, offset=0).reshape(B, T * 32, 8)
    assert float(np.sqrt(np.mean((first - waveglow_ref.infer(mel, wg_weights, wg_cfg, z=z0)) ** 2))) <= 1e-4
    second = rt(mel)                                   # the stream has advanced: other noise
    assert not np.array_equal(first, second)
    assert np.array_equal(rt(mel, seed=77), first)     # an explicit seed restarts at offset 0
    assert np.array_equal(rt(mel, deterministic=True), gpu_engine.waveglow_infer(mel))
