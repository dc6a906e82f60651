"""Rebuild BASELINE configs[0]'s run (engine.train_MTD_GAN_Ours at the reference's default init, seeds 2024, 16 patches,
SURVEY 8c) from tests/golden/config1_replay.json without the reference: the construction's random draws are replayed
from the recorded tape (oracle/rng_tape.py), the result is checked tensor by tensor against the recorded float64
checksums, and the two input batches continue the same generator stream."""
import json
import os

import numpy as np
import torch

import rng_tape

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load():
    z = json.load(open(os.path.join(GOLD, "config1_replay.json")))
    torch.manual_seed(z["seed"])
    state = rng_tape.replay(z["tape"], z["derivation"])
    for k, (s, a) in z["checksums"].items():
        got = rng_tape.checksum(state[k])
        assert abs(got[0] - s) <= 1e-9 * max(1.0, abs(a)) and abs(got[1] - a) <= 1e-9 * max(1.0, abs(a)), \
            f"replayed default-init state differs from the reference's at {k}: this torch build draws differently"
    x, y = torch.rand(16, 1, 64, 64), torch.rand(16, 1, 64, 64)
    assert np.allclose(rng_tape.checksum(x), z["x_checksum"], rtol=1e-12) and np.allclose(rng_tape.checksum(y), z["y_checksum"], rtol=1e-12)
    bits = np.load(os.path.join(GOLD, "config1_replay_masks.npz"))["keep_bits"]
    keep = np.unpackbits(bits)[:5 * 16 * 512].reshape(5, 16, 512)
    masks = [torch.from_numpy(k.astype(np.float32)) / 0.7 for k in keep]
    return z, state, x, y, masks
