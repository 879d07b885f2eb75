"""Cut the reference's mel-STFT golden pair into a small committed fixture.

Run HERE (where /root/reference exists):  python scripts/make_stft_fixture.py
Inputs (data files held by the reference's own tests, test_utils_audio.py:85-112):
  /root/reference/tests/__reproduction/audio_resample.npy      float32 (89412,)
  /root/reference/tests/__reproduction/stft-TacotronSTFT.npy   float32 (350, 80)  = TacotronSTFT()(audio)[0]
Output: tests/golden/stft_tacotron_fixture.npz with audio[:32768] and mel rows 0..119.  Frame t reads original samples
[t*256 - 512, t*256 + 512); rows with t*256 + 512 <= 32768 (t <= 126) do not see the end-of-signal reflect padding,
so the truncated audio reproduces them exactly.
"""
import hashlib
import os
import numpy as np

REF = '/root/reference/tests/__reproduction'
N_AUDIO, N_ROWS = 32768, 120

def sha(p):
    return hashlib.sha256(open(p, 'rb').read()).hexdigest()

a_path, m_path = os.path.join(REF, 'audio_resample.npy'), os.path.join(REF, 'stft-TacotronSTFT.npy')
audio, mel = np.load(a_path), np.load(m_path)
assert audio.dtype == np.float32 and mel.shape == (350, 80), (audio.dtype, audio.shape, mel.shape)
out = os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden', 'stft_tacotron_fixture.npz')
np.savez_compressed(out, audio=audio[:N_AUDIO], mel=mel[:N_ROWS],
                    audio_sha256=sha(a_path), mel_sha256=sha(m_path),
                    audio_full_len=np.int64(audio.shape[0]), mel_full_rows=np.int64(mel.shape[0]),
                    tolerance=np.float32(2e-3))
print('wrote', os.path.abspath(out), os.path.getsize(out), 'bytes')
