"""GPU parity: HIP WaveGlow (through the C ABI) vs the numpy oracle on the same seeded inputs.

Tolerance (BASELINE.json north_star): waveform RMS error <= 1e-4 (fp32).
"""
import numpy as np
import pytest

from conftest import rms

pytestmark = pytest.mark.gpu

RMS_TOL = 1e-4


def _inputs(B, T, seed=7):
    mel = np.random.default_rng(seed).uniform(-11.5, 1.2, (B, T, 80)).astype(np.float32)
    z = np.random.default_rng(seed + 4).standard_normal((B, T * 32, 8)).astype(np.float32)
    return mel, z


@pytest.mark.parametrize('B,T', [(1, 8), (2, 13), (3, 5)])
def test_waveglow_matches_oracle(gpu_engine, wg_weights, wg_cfg, B, T):
    from oracle import waveglow_ref
    mel, z = _inputs(B, T)
    ref = waveglow_ref.infer(mel, wg_weights, wg_cfg, z=z, sigma=1.0)
    out = gpu_engine.waveglow_infer(mel, z=z, sigma=1.0)
    assert out.shape == ref.shape == (B, T * 256)
    assert np.isfinite(out).all()
    err = rms(out - ref)
    print(f'B={B} T={T} rms_err={err:.3e} max_err={np.abs(out - ref).max():.3e} ref_rms={rms(ref):.3f}')
    assert err <= RMS_TOL


def test_waveglow_parity_with_four_times_less_end_attenuation(wg_cfg):
    """The session weights scale every `end` conv by 0.05, which damps WN errors ~20x on their way to the waveform
    (DESIGN.md section 2).  Same check with end_scale = 0.2: signal RMS 2.6, peaks ~46, the largest scale at which a
    random-weight flow is still a well-conditioned map (at 0.5 the two CPU restatements of the oracle already differ by
    2.5 %, scripts/end_scale_probe.py).  The absolute 1e-4 tolerance still holds; measured 4.6e-6."""
    from oracle import waveglow_ref
    from text_to_speech_amd import weights
    from text_to_speech_amd.engine import HipEngine
    w = weights.synth_waveglow(wg_cfg, seed=1234, end_scale=0.2)
    mel, z = _inputs(2, 24, seed=9)
    ref = waveglow_ref.infer(mel, w, wg_cfg, z=z, sigma=1.0)
    eng = HipEngine(0)
    try:
        eng.load_state(w)
        eng.finalize()
        for prec, tol in (('f32', RMS_TOL), ('f16x3', RMS_TOL)):
            out = eng.waveglow_infer(mel, z=z, sigma=1.0, precision=prec)
            err = rms(out - ref)
            print(f'end_scale=0.2 {prec}: rms_err={err:.3e} ref_rms={rms(ref):.3f} max|ref|={np.abs(ref).max():.1f}')
            assert rms(ref) > 1.5 and err <= tol
    finally:
        eng.close()


def test_waveglow_hip_inverts_the_published_forward_flow(gpu_engine, wg_weights, wg_cfg):
    """Round trip through the model itself: z = forward flow of a waveform (written from the WaveGlow paper, float64, CPU;
    oracle/waveglow_ref.forward_flow), then the HIP `infer` must give the waveform back.  Independent of the oracle's `infer`."""
    from oracle import waveglow_ref
    rng = np.random.default_rng(21)
    mel = rng.uniform(-11.5, 1.2, (2, 5, 80)).astype(np.float32)
    audio = rng.uniform(-0.9, 0.9, (2, 5 * 256)).astype(np.float32)
    z = waveglow_ref.forward_flow(audio, mel, wg_weights, wg_cfg).astype(np.float32)
    for prec in ('f32', 'f16x3'):
        out = gpu_engine.waveglow_infer(mel, z=z, sigma=1.0, precision=prec)
        err = rms(out - audio)
        print(f'round trip {prec}: rms_err={err:.3e} max_err={np.abs(out - audio).max():.3e}')
        assert err <= RMS_TOL


def test_waveglow_deterministic_zero_noise(gpu_engine, wg_weights, wg_cfg):
    """z=None is the reference's deterministic=True path (zeros), waveglow_arch.py:264-267,293-296."""
    from oracle import waveglow_ref
    mel, _ = _inputs(1, 6, seed=3)
    ref = waveglow_ref.infer(mel, wg_weights, wg_cfg, z=None)
    out = gpu_engine.waveglow_infer(mel, z=None)
    assert rms(out - ref) <= RMS_TOL


def test_waveglow_sigma(gpu_engine, wg_weights, wg_cfg):
    from oracle import waveglow_ref
    mel, z = _inputs(1, 4, seed=5)
    ref = waveglow_ref.infer(mel, wg_weights, wg_cfg, z=z, sigma=0.6)
    out = gpu_engine.waveglow_infer(mel, z=z, sigma=0.6)
    assert rms(out - ref) <= RMS_TOL


# ---- fp16-operand mode (BASELINE.json configs 3 / 5: the reference's mixed_float16 policy) -------------------------
# The HIP path keeps fp32 accumulators, fp32 epilogue math and fp32 master copies of the residual stream and of the
# flow state; only the GEMM operands (activations, mel, weights) are rounded to fp16.  Measured on MI355X against the
# fp32 oracle: waveform RMS error 2.0e-4 at signal RMS 1.06 (scripts/f16_probe.py).  The fp32 tolerance (1e-4) applies
# to the fp32 path; the fp16 bound below is 5x the measured error.
F16_RMS_TOL = 1e-3


@pytest.mark.parametrize('B,T', [(1, 8), (2, 13)])
def test_waveglow_f16_close_to_fp32_oracle(gpu_engine, wg_weights, wg_cfg, B, T):
    from oracle import waveglow_ref
    mel, z = _inputs(B, T)
    ref = waveglow_ref.infer(mel, wg_weights, wg_cfg, z=z, sigma=1.0)
    out = gpu_engine.waveglow_infer(mel, z=z, sigma=1.0, precision='f16')
    assert out.shape == ref.shape and np.isfinite(out).all()
    err = rms(out - ref)
    print(f'f16 B={B} T={T} rms_err={err:.3e} max_err={np.abs(out - ref).max():.3e}')
    assert err <= F16_RMS_TOL
    # and it is a different arithmetic from the exact path (guards against the flag being ignored)
    exact = gpu_engine.waveglow_infer(mel, z=z, sigma=1.0)
    assert rms(exact - ref) <= RMS_TOL < 1e3 * rms(out - exact)


def test_waveglow_f16_then_f32_same_engine(gpu_engine, wg_weights, wg_cfg):
    """The two precisions share the engine's flow-state buffers: interleaving them must not change the fp32 result."""
    mel, z = _inputs(2, 7, seed=21)
    a = gpu_engine.waveglow_infer(mel, z=z)
    gpu_engine.waveglow_infer(mel, z=z, precision='f16')
    b = gpu_engine.waveglow_infer(mel, z=z)
    assert np.array_equal(a, b)


def test_waveglow_bad_precision(gpu_engine):
    with pytest.raises(ValueError):
        gpu_engine.waveglow_infer(np.zeros((1, 4, 80), np.float32), precision='bf16')


# ---- the 256-row-tile kernels (the ones the headline config runs) ---------------------------------------------------
# B*T = 256 frames pads to the same number of rows with 128- and 256-row tiles, so waveglow_run picks the 256-row kernels
# (fp32: 4 waves, 256 x 128 blocks; fp16: 8 waves, 256 x 256 blocks) -- the small cases above all take the 128-row ones.
def test_waveglow_128x64_tiles_match_oracle(gpu_engine, wg_weights, wg_cfg):
    """B*T = 100 frames: 64- and 128-row padding coincide (128 rows per phase), so the run takes the 128 x 64 tile kernels
    (short utterances: twice the blocks); the tiny cases above take the 64-row-tile kernels, config 2 the 256-row ones."""
    from oracle import waveglow_ref
    mel, z = _inputs(1, 100, seed=51)
    ref = waveglow_ref.infer(mel, wg_weights, wg_cfg, z=z, sigma=1.0)
    err = rms(gpu_engine.waveglow_infer(mel, z=z, sigma=1.0) - ref)
    err16 = rms(gpu_engine.waveglow_infer(mel, z=z, sigma=1.0, precision='f16') - ref)
    print(f'128x64 tiles: f32 rms_err={err:.3e}  f16 rms_err={err16:.3e}')
    assert err <= RMS_TOL and err16 <= F16_RMS_TOL


def test_waveglow_256_row_tiles_match_oracle(gpu_engine, wg_weights, wg_cfg):
    from oracle import waveglow_ref
    mel, z = _inputs(2, 128, seed=31)
    ref = waveglow_ref.infer(mel, wg_weights, wg_cfg, z=z, sigma=1.0)
    out = gpu_engine.waveglow_infer(mel, z=z, sigma=1.0)
    err = rms(out - ref)
    out16 = gpu_engine.waveglow_infer(mel, z=z, sigma=1.0, precision='f16')
    err16 = rms(out16 - ref)
    print(f'256-row tiles: f32 rms_err={err:.3e}  f16 rms_err={err16:.3e}')
    assert err <= RMS_TOL and err16 <= F16_RMS_TOL


def test_waveglow_winograd_and_direct_forms_against_the_oracle(gpu_engine, wg_weights, wg_cfg):
    """The fp32 path evaluates WN layers 1 - 7 in their Winograd F(4,3) form from 144 frames per call (csrc/wn_wino.hip): both
    forms against the oracle on the same inputs, the switch, the report of which one ran, utterance lengths that leave partial
    frame groups, and the small shapes that keep the direct form."""
    from oracle import waveglow_ref
    mel, z = _inputs(2, 192, seed=33)                              # 384 frames (round 3's smallest Winograd call; the form starts at 144 now)
    ref = waveglow_ref.infer(mel, wg_weights, wg_cfg, z=z, sigma=1.0)
    try:
        wino = gpu_engine.waveglow_infer(mel, z=z, sigma=1.0)
        assert gpu_engine.last_waveglow_form == 'winograd'
        gpu_engine.set_waveglow_form('direct')
        direct = gpu_engine.waveglow_infer(mel, z=z, sigma=1.0)
        assert gpu_engine.last_waveglow_form == 'direct'
    finally:
        gpu_engine.set_waveglow_form('winograd')
    e_w, e_d, diff = rms(wino - ref), rms(direct - ref), rms(wino - direct)
    print(f'winograd rms_err={e_w:.3e}  direct rms_err={e_d:.3e}  winograd vs direct {diff:.3e}')
    assert e_w <= RMS_TOL and e_d <= RMS_TOL and diff <= 5e-6
    # utterance lengths that are not a multiple of the group sizes (2 x 195 and 3 x 131 frames): the frame groups are cut per
    # utterance (the last group of an utterance is partial), so these run the Winograd form as well
    for B2, T2, seed in ((2, 195, 34), (3, 131, 36)):
        mel2, z2 = _inputs(B2, T2, seed=seed)
        out2 = gpu_engine.waveglow_infer(mel2, z=z2, sigma=1.0)
        assert gpu_engine.last_waveglow_form == 'winograd'
        e2 = rms(out2 - waveglow_ref.infer(mel2, wg_weights, wg_cfg, z=z2, sigma=1.0))
        print(f'{B2} x {T2} frames: winograd rms_err={e2:.3e}')
        assert e2 <= RMS_TOL
    # calls below 144 frames keep the direct form (too few blocks for the fused kernel's 64 x 128 tiles: measured break-even
    # between 100 and 150 frames); the smallest Winograd call, 144 frames on 64-row phase blocks, against the oracle
    m4, z4 = _inputs(1, 144, seed=37)
    o4 = gpu_engine.waveglow_infer(m4, z=z4)
    assert gpu_engine.last_waveglow_form == 'winograd'
    e4 = rms(o4 - waveglow_ref.infer(m4, wg_weights, wg_cfg, z=z4, sigma=1.0))
    print(f'1 x 144 frames: winograd rms_err={e4:.3e}')
    assert e4 <= RMS_TOL
    for B3, T3 in ((1, 16), (2, 64)):
        m3, z3 = _inputs(B3, T3, seed=35)
        gpu_engine.waveglow_infer(m3, z=z3)
        assert gpu_engine.last_waveglow_form == 'direct'
    with pytest.raises(ValueError):
        gpu_engine.set_waveglow_form('fft')


def test_winograd_kernel_is_bit_identical_to_its_three_pass_form(gpu_engine):
    """The default Winograd form is ONE kernel per layer (input transform in the operand reads, six accumulator sets, output
    transform + gate in the epilogue; csrc/wn_wino.hip); round 3's three passes (pre-pass, per-product GEMM, combine) and the
    intermediate stage (fused GEMM behind the pre-pass) are kept as measurement forms.  All three form the same products in
    the same k order and write the transforms identically, so their waveforms are EQUAL bit for bit -- which makes this a
    sharp test of the fused kernel's LDS-DMA pipeline (a tile read before it landed shows up as a difference, and as a
    difference between two runs): two runs of each shape, shapes with partial frame groups and padded group rows."""
    try:
        # (more than 512 frames per call: below that the fused forms run on 64-row phase blocks and the three-pass form on 128-row
        #  ones, i.e. the first-layer and residual kernels of the forms differ in tile shape)
        for B, T, seed in ((3, 200, 81), (5, 131, 82), (1, 513, 83)):
            mel, z = _inputs(B, T, seed=seed)
            outs = {}
            for form in ('winograd-3pass', 'winograd-prepass', 'winograd', 'winograd'):
                gpu_engine.set_waveglow_form(form)
                out = gpu_engine.waveglow_infer(mel, z=z)
                assert gpu_engine.last_waveglow_form == 'winograd' and np.isfinite(out).all()
                if form in outs:
                    assert np.array_equal(out, outs[form]), f'{B} x {T}: two runs of the fused kernel differ'
                outs[form] = out
            assert np.array_equal(outs['winograd'], outs['winograd-3pass']), f'{B} x {T}: fused kernel vs three passes'
            assert np.array_equal(outs['winograd-prepass'], outs['winograd-3pass']), f'{B} x {T}: fused GEMM behind the pre-pass'
    finally:
        gpu_engine.set_waveglow_form('winograd')


def test_waveglow_winograd_form_with_four_times_less_end_attenuation(wg_cfg):
    """The Winograd form where the session weights' 0.05 `end` scaling does not damp it 20x: `end_scale` = 0.2 (signal RMS
    ~2.6, the largest scale at which a random-weight flow is a well-conditioned map) on 2 x 200 frames
    = 400 frames per call, which takes the Winograd form by default; the direct form's error is printed beside it.
    Reference maths: /root/reference/architectures/waveglow_arch.py:105-141.  Absolute tolerance unchanged: 1e-4."""
    from oracle import waveglow_ref
    from text_to_speech_amd import weights
    from text_to_speech_amd.engine import HipEngine
    w = weights.synth_waveglow(wg_cfg, seed=1234, end_scale=0.2)
    mel, z = _inputs(2, 200, seed=19)
    ref = waveglow_ref.infer(mel, w, wg_cfg, z=z, sigma=1.0)
    eng = HipEngine(0)
    try:
        eng.load_state(w)
        eng.finalize()
        wino = eng.waveglow_infer(mel, z=z, sigma=1.0)
        assert eng.last_waveglow_form == 'winograd'               # the default form of a 400-frame fp32 call
        eng.set_waveglow_form('direct')
        direct = eng.waveglow_infer(mel, z=z, sigma=1.0)
        assert eng.last_waveglow_form == 'direct'
    finally:
        eng.close()
    e_w, e_d = rms(wino - ref), rms(direct - ref)
    print(f'end_scale=0.2, 2 x 200 frames: winograd rms_err={e_w:.3e}  direct rms_err={e_d:.3e}  ref_rms={rms(ref):.3f} '
          f'max|ref|={np.abs(ref).max():.1f}  winograd vs direct {rms(wino - direct):.3e}')
    assert rms(ref) > 1.5 and e_w <= RMS_TOL and e_d <= RMS_TOL


# relative RMS error bound of one WN layer's gated activations (|acts| < 1, RMS ~0.3): fp32 rounding of a K = 1856 dot
# product is ~1e-6 relative; F(4,3)'s transform constants cost a factor ~3 (csrc/wn_wino.hip).  Measured: see the printout.
ACTS_REL_TOL = 2e-5


def test_wn_layer_activations_against_the_oracle_in_both_forms(gpu_engine, wg_weights, wg_cfg):
    """Layer-level parity, before the res/skip and `end` convolutions attenuate anything: the gated activations
    tanh(.) * sigmoid(.) of one WN layer per Winograd group kind -- dilation 4 (four phases of a frame), 16 (two phases x two
    frames), 64 (four frames of a phase) -- of the first flow that runs (flow 11), against `oracle/waveglow_ref.wn_block`
    intermediates on the same x / mel, in the Winograd and in the direct form.  2 x 200 frames: the Winograd form's domain.
    Reference: /root/reference/architectures/waveglow_arch.py:19-24,105-127."""
    from oracle import waveglow_ref
    mel, z = _inputs(2, 200, seed=23)
    w = {k: v for k, v in wg_weights.items() if k.startswith('waveglow/')}
    spect = waveglow_ref.regroup(waveglow_ref.upsample(mel, w['waveglow/upsample/kernel'], w['waveglow/upsample/bias'],
                                                       wg_cfg.upsample_stride), wg_cfg.n_group)
    n_half = wg_cfg.n_remaining_channels // 2
    ref_acts = []
    waveglow_ref.wn_block(z[:, :, :n_half], spect, w, 'waveglow/block-11', wg_cfg.n_layers, wg_cfg.n_channels,
                          collect=ref_acts, stop_after=6)
    try:
        for layer in (2, 4, 6):
            ref = ref_acts[layer]
            errs = {}
            for form in ('winograd', 'direct'):
                gpu_engine.set_waveglow_form(form)
                acts = gpu_engine.waveglow_probe_acts(mel, z=z, flow=11, layer=layer)
                assert gpu_engine.last_waveglow_form == form and acts.shape == ref.shape
                errs[form] = rms(acts - ref) / rms(ref)
            print(f'WN layer {layer} (dilation {1 << layer}): acts rel. RMS error winograd {errs["winograd"]:.3e}, '
                  f'direct {errs["direct"]:.3e} (acts RMS {rms(ref):.3f})')
            assert errs['winograd'] <= ACTS_REL_TOL and errs['direct'] <= ACTS_REL_TOL
    finally:
        gpu_engine.set_waveglow_form('winograd')


def test_waveglow_config2_rows_equal_batch1_runs(gpu_engine):
    """Full BASELINE.json config 2 (8 x 800 frames, 256-row tiles, Winograd form) against batch-1 runs of single rows in both
    forms (Winograd on 256-row tiles, direct on 128-row tiles): a size-independent property (utterances are independent) that
    also cross-checks the two forms and tile configurations at the full size."""
    mel, z = _inputs(8, 800, seed=41)
    full = gpu_engine.waveglow_infer(mel, z=z)
    assert full.shape == (8, 800 * 256) and np.isfinite(full).all()
    full16 = gpu_engine.waveglow_infer(mel, z=z, precision='f16')
    assert np.isfinite(full16).all() and rms(full16 - full) <= F16_RMS_TOL
    assert gpu_engine.last_waveglow_form == 'direct'              # (the fp16 modes have no Winograd form)
    for b in (0, 5):
        single = gpu_engine.waveglow_infer(mel[b:b + 1], z=z[b:b + 1])         # Winograd form on 256-row tiles, like `full`
        assert gpu_engine.last_waveglow_form == 'winograd' and rms(single[0] - full[b]) <= 5e-6
        try:                                                                   # ... and the direct form on 128-row tiles
            gpu_engine.set_waveglow_form('direct')
            direct = gpu_engine.waveglow_infer(mel[b:b + 1], z=z[b:b + 1])
        finally:
            gpu_engine.set_waveglow_form('winograd')
        assert rms(direct[0] - full[b]) <= 5e-6
        single16 = gpu_engine.waveglow_infer(mel[b:b + 1], z=z[b:b + 1], precision='f16')
        assert rms(single16[0] - full16[b]) <= 5e-5      # fp16 rounding of differently-ordered fp32 sums


# ---- split-fp16 mode (tts_hip_waveglow_infer_f16x3): three fp16 MFMAs per product on (hi, lo) operand planes --------------
# ~22 operand bits and fp32 accumulation: the SAME tolerance as the exact fp32 path applies (measured 4.4e-7 RMS against the
# exact-fp32 kernels, i.e. the level of fp32 rounding itself).
@pytest.mark.parametrize('B,T', [(1, 8), (2, 13), (1, 100), (2, 128)])
def test_waveglow_f16x3_matches_oracle_at_fp32_tolerance(gpu_engine, wg_weights, wg_cfg, B, T):
    from oracle import waveglow_ref
    mel, z = _inputs(B, T, seed=61)
    ref = waveglow_ref.infer(mel, wg_weights, wg_cfg, z=z, sigma=1.0)
    out = gpu_engine.waveglow_infer(mel, z=z, sigma=1.0, precision='f16x3')
    exact = gpu_engine.waveglow_infer(mel, z=z, sigma=1.0)
    err = rms(out - ref)
    print(f'f16x3 B={B} T={T} rms_err={err:.3e} (exact fp32 path: {rms(exact - ref):.3e}; x3 vs exact {rms(out - exact):.3e})')
    assert out.shape == ref.shape and np.isfinite(out).all() and err <= RMS_TOL
    assert not np.array_equal(out, exact)                              # a different arithmetic, not the flag ignored


def test_waveglow_f16x3_config2_close_to_exact_path(gpu_engine):
    """Full BASELINE config 2 shape (256 x 256 tiles, two LDS buffers): against the exact-fp32 kernels and batch-1 rows."""
    mel, z = _inputs(8, 800, seed=71)
    exact = gpu_engine.waveglow_infer(mel, z=z)
    out = gpu_engine.waveglow_infer(mel, z=z, precision='f16x3')
    assert np.isfinite(out).all() and rms(out - exact) <= 5e-6
    single = gpu_engine.waveglow_infer(mel[3:4], z=z[3:4], precision='f16x3')
    assert rms(single[0] - out[3]) <= 5e-6
