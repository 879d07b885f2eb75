#!/usr/bin/env python3
"""Offline step 1 of the Keras checkpoint import (run where the REFERENCE runs: keras 3 + a backend + h5py installed).

Restores one of the reference's models with the reference's own code -- so the `.weights.h5` layout of whatever Keras
version wrote it is resolved by Keras itself -- and writes every variable as `{variable.path: value}` to a .safetensors
file.  Step 2 runs anywhere (no Keras, no h5py):

    python -m text_to_speech_amd.weights_import --keras-tacotron2 tacotron2_vars.safetensors \
                                                 --keras-waveglow waveglow_vars.safetensors -o model.ttsw

usage (from the reference's repository root):
    python /path/to/export_keras_weights.py pretrained_tacotron2 tacotron2_vars.safetensors
    python /path/to/export_keras_weights.py WaveGlow waveglow_vars.safetensors

This script is NOT run by the tests or on the GPU box (neither has Keras); what is tested is step 2's name mapping
(tests/test_weights_import.py) against the layer names of the reference source.
"""
import sys


def main():
    if len(sys.argv) != 3:
        raise SystemExit(__doc__)
    name, out = sys.argv[1], sys.argv[2]
    import numpy as np
    from models import get_pretrained                     # the reference's registry (models/__init__.py:22)
    model = get_pretrained(name)
    net = getattr(model, 'model', model)                    # BaseModel keeps the keras.Model in `.model`
    tensors = {}
    for v in net.variables:
        path = getattr(v, 'path', None) or v.name
        tensors[path] = np.ascontiguousarray(np.asarray(v), dtype=np.float32)
    try:
        from safetensors.numpy import save_file
        save_file(tensors, out)
    except ImportError:
        np.savez(out if out.endswith('.npz') else out + '.npz', **{k.replace('/', '|'): a for k, a in tensors.items()})
    print(f'wrote {len(tensors)} variables of {name} to {out}')


if __name__ == '__main__':
    main()
