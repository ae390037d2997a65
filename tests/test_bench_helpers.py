"""bench.py's host-side helpers that need no GPU: the cd-hit probe against a stand-in
executable that writes a known .clstr, and the launcher rules of --gpus."""
import os
import stat
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

FAKE = """#!/bin/sh
# stand-in for cd-hit: clusters {s0, s2} (representative s0) and {s1}
while [ $# -gt 0 ]; do case "$1" in -o) out="$2"; shift;; esac; shift; done
printf '>Cluster 0\\n0\\t30aa, >s0... *\\n1\\t28aa, >s2... at 90.00%%\\n>Cluster 1\\n0\\t29aa, >s1... *\\n' > "$out.clstr"
: > "$out"
"""


def test_cdhit_probe_compares_membership(tmp_path, monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    res = np.frombuffer(b'A' * 30 + b'C' * 29 + b'A' * 28, dtype=np.uint8)
    off = np.array([0, 30, 59, 87], dtype=np.uint64)
    monkeypatch.setenv('PATH', str(tmp_path / 'nothing'))
    assert bench.cdhit_reference(res, off, np.array([0, 1, 0]), np.array([0, 0, 1]), str(tmp_path)) is None
    exe = tmp_path / 'bin' / 'cd-hit'
    exe.parent.mkdir()
    exe.write_text(FAKE)
    exe.chmod(exe.stat().st_mode | stat.S_IEXEC)
    monkeypatch.setenv('PATH', str(exe.parent) + os.pathsep + '/usr/bin:/bin')
    same = bench.cdhit_reference(res, off, np.array([0, 1, 0]), np.array([0, 0, 1]), str(tmp_path))
    assert same['as_reference']['membership_equal'] and same['as_reference']['clusters'] == 2
    other = bench.cdhit_reference(res, off, np.array([0, 1, 1]), np.array([0, 0, 1]), str(tmp_path))
    assert not other['all_cores_unchunked']['membership_equal']
    assert other['all_cores_unchunked']['sequences_in_a_different_cluster'] == 1


def test_gpus_must_agree_with_the_launcher():
    env = dict(os.environ, WORLD_SIZE='2', RANK='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1'], env=env, capture_output=True, text=True)
    assert r.returncode != 0 and 'must agree' in r.stderr
