import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def wg_cfg():
    from text_to_speech_amd.config import WaveGlowConfig
    return WaveGlowConfig()


@pytest.fixture(scope='session')
def taco_cfg():
    from text_to_speech_amd.config import Tacotron2Config
    return Tacotron2Config()


@pytest.fixture(scope='session')
def wg_weights(wg_cfg):
    from text_to_speech_amd import weights
    return weights.synth_waveglow(wg_cfg, seed=1234)


@pytest.fixture(scope='session')
def taco_weights(taco_cfg):
    from text_to_speech_amd import weights
    return weights.synth_tacotron2(taco_cfg, seed=1234)


@pytest.fixture(scope='session')
def gpu_engine(wg_weights, taco_weights):
    """One engine for the whole GPU session, loaded through the C ABI (libtts_hip.so); fails loudly if missing."""
    from text_to_speech_amd.engine import HipEngine
    eng = HipEngine(0)
    eng.load_state(wg_weights)
    eng.load_state(taco_weights)
    eng.finalize()
    yield eng
    eng.close()


def rms(x):
    return float(np.sqrt(np.mean(np.square(np.asarray(x, dtype=np.float64)))))
