"""Fixture hygiene: the committed golden vectors, their manifest and the generator cannot drift apart.

tests/golden/MANIFEST.json lists every array of every committed .npz (shape, dtype, CRC-32 of its bytes); it is written by
`python tests/golden/make_golden.py --manifest`.  (1) always: the committed files equal the manifest; (2) in the build
container, where the read-only reference is present: the generator is re-run for the groups that take seconds (the conv
blocks + tiny model, geometry, schedules / EMA / L1) into a scratch directory and must reproduce the committed files bit for
bit - a changed generator line (round 1: `out_eval[:, ::3]`) fails here instead of at the next regeneration."""
import json
import os
import subprocess
import sys
import zlib

import numpy as np
import pytest

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def describe(path):
    z = np.load(path, allow_pickle=False)
    return {k: [list(z[k].shape), str(z[k].dtype), zlib.crc32(np.ascontiguousarray(z[k]).tobytes())] for k in sorted(z.files)}


def test_committed_fixtures_match_manifest():
    man = json.load(open(os.path.join(HERE, "MANIFEST.json")))
    files = sorted(f for f in os.listdir(HERE) if f.endswith(".npz"))
    assert files == sorted(man), (set(files) ^ set(man))
    for f in files:
        assert describe(os.path.join(HERE, f)) == man[f], f


@pytest.mark.skipif(not os.path.isdir("/root/reference/yolox_24p"), reason="the reference tree exists in the build container only")
def test_generator_reproduces_committed_fixtures(tmp_path):
    env = dict(os.environ, EP24_GOLDEN_OUT=str(tmp_path), PYTHONDONTWRITEBYTECODE="1")
    p = subprocess.run([sys.executable, os.path.join(HERE, "make_golden.py"), "model", "geometry", "n2"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:]
    made = sorted(f for f in os.listdir(tmp_path) if f.endswith(".npz"))
    assert len(made) >= 15, made
    for f in made:
        assert describe(os.path.join(tmp_path, f)) == describe(os.path.join(HERE, f)), f
