"""CPU oracle for the Tacotron2 + WaveGlow + mel-STFT hot path.

TEST INFRASTRUCTURE ONLY.  Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import
this package, and only as the checker.  The shipped path (`text_to_speech_amd/`) never imports it and has no CPU
fallback: without the HIP library it raises.

Pinning status:
  * `mel_stft_ref`  -- PINNED by the reference's own golden pair tests/__reproduction/{audio_resample,stft-TacotronSTFT}.npy
                       (committed, truncated, under tests/golden/; reference tolerance 2e-3, test_utils_audio.py:110-112).
  * `waveglow_ref`, `tacotron2_ref` -- PARITY UNPINNED: the reference has no test, golden vector or checkpoint for these
                       and cannot be imported here (Keras 3 is not installed: ordinary ModuleNotFoundError, nothing was
                       denied).  They restate the reference source line by line (file:line cited per function) and are
                       cross-checked by an independent torch.nn.functional restatement (oracle/torch_ref.py,
                       tests/test_oracle_crosscheck.py).
"""
