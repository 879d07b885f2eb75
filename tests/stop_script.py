"""Test helper: gate weights that make every row's stop token fire at a chosen step ("scripted stops").

The stop token never feeds back into the decoder state (tacotron2_arch.py:655-665), so the gate input of every step --
`cell_out = [h_dec | ctx]` -- does not depend on the gate weights.  Run the oracle once without early stopping, record
`cell_out` (oracle `trace`), then fit a gate kernel / bias whose logit ramps through zero half a step before row b's target
step: logit[b, t] = slope * (t - T_b + 0.5), i.e. -slope/2 on the last quiet frame and +slope/2 on the firing one.  A ramp
(not a step) keeps the kernel small -- consecutive decoder states are close -- and a little ridge shrinks it further.
The fit is checked: every decision must be on the right side with at least 0.4 * slope of logit to spare.  Returned
`sensitivity` = ||kernel||_2: the logit change a unit perturbation of cell_out can cause at worst.  Measured on the CPU
with the oracle itself: rounding the two LSTM matrices to fp16 moves these logits by < 0.04 (slope 2, targets <= 40).
"""
import numpy as np


def script_stop_tokens(weights, cfg, tokens, target_lengths, speaker_embedding=None, prenet_masks=None, slope=2.0,
                       ridge=1e-5, **infer_kwargs):
    """Returns (new weight dict, sensitivity, margin).  `target_lengths[b]` = the value `lengths[b]` must take (= the
    index of the frame on which the stop token fires, tacotron2_arch.py:664-665)."""
    from oracle import tacotron2_ref
    target_lengths = [int(t) for t in target_lengths]
    steps = max(target_lengths) + 1
    trace = {}
    tacotron2_ref.infer(tokens, weights, cfg, speaker_embedding=speaker_embedding, max_length=steps, early_stopping=False,
                        prenet_masks=None if prenet_masks is None else prenet_masks[:, :steps], trace=trace,
                        **infer_kwargs)
    cell = trace['cell_out'].astype(np.float64)                     # [B, steps, D]
    X = np.concatenate([cell[b, :T + 1] for b, T in enumerate(target_lengths)], 0)
    X = np.concatenate([X, np.ones((X.shape[0], 1))], 1)
    y = np.concatenate([slope * (np.arange(T + 1) - T + 0.5) for T in target_lengths])
    sol = X.T @ np.linalg.solve(X @ X.T + ridge * np.eye(len(X)), y)   # ridge regression, dual form (rows << columns)
    fit = X @ sol
    margin = float(np.abs(fit).min())
    assert np.all(np.sign(fit) == np.sign(y)) and margin >= 0.4 * slope, 'scripted stops are not realisable with margin'
    out = dict(weights)
    out['tacotron2/decoder/gate_output/kernel'] = sol[:-1, None].astype(np.float32)
    out['tacotron2/decoder/gate_output/bias'] = sol[-1:].astype(np.float32)
    return out, float(np.linalg.norm(sol[:-1])), margin
