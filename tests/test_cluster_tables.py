"""The constant tables compiled into libpgx (pangenomix_amd/csrc/cluster_tables.h) against
the oracle's independently typed copy (oracle/cluster_ref.c), and
against the properties a substitution matrix must have. Reads the header as text: no GPU."""
import os
import re

import numpy as np

import oracle

HDR = os.path.join(os.path.dirname(__file__), '..', 'pangenomix_amd', 'csrc', 'cluster_tables.h')


def _ints(body):
    return [int(x) for x in re.findall(r'-?\d+', body)]


def _header_tables():
    text = open(HDR).read()
    aa = re.search(r'kAa2Idx\[26\]\s*=\s*\{([^}]*)\}', text, re.S).group(1)
    init = re.search(r'#define PGXC_BLOSUM62_INIT \{(.*?)\n\}', text, re.S).group(1)
    flat = re.search(r'#define PGXC_BLOSUM62_FLAT \{(.*?)\}', text, re.S).group(1)
    return (np.array(_ints(aa)), np.array(_ints(init)).reshape(21, 21),
            np.array(_ints(flat)).reshape(21, 21))


def test_header_tables_equal_oracle_tables():
    aa, init, flat = _header_tables()
    o_aa, o_bl = oracle.protein_tables()
    assert np.array_equal(aa, np.asarray(o_aa).ravel())
    assert np.array_equal(init, np.asarray(o_bl)[:21, :21])   # the oracle keeps the full 23-row table
    assert np.array_equal(flat, init)


def test_matrix_properties():
    aa, init, _ = _header_tables()
    assert np.array_equal(init, init.T)                       # symmetric
    assert all(init[i, i] == init[i].max() for i in range(20))  # identity scores dominate
    # the 20 standard residues map onto 0..19 exactly once; B/Z fold onto D/E's neighbours
    letters = 'ARNDCQEGHILKMFPSTWYV'
    assert [aa[ord(c) - 65] for c in letters] == list(range(20))
    assert aa[ord('B') - 65] == aa[ord('N') - 65] and aa[ord('Z') - 65] == aa[ord('E') - 65]
    assert all(aa[ord(c) - 65] == 20 for c in 'JOUX')


def test_scalars():
    text = open(HDR).read()
    assert re.search(r'kScoreScale\s*=\s*655360', text)
    assert re.search(r'kGapOpen\s*=\s*-11,\s*kGapExt\s*=\s*-1', text)
