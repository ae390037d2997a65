"""Host side of the clustering boundary (pangenomix_amd/cluster.py): FASTA clean-up as cd-hit
reads it, cd-hit argument translation, filter cut-offs, .clstr writer. No GPU needed."""
import numpy as np
import pytest

from pangenomix_amd import cluster


def test_read_fasta_applies_cdhit_sequence_rules(tmp_path):
    p = tmp_path / 'x.faa'
    p.write_text('>h1 some description\nMKV\nLLA*\n>h2\nmkvlla \n\n>h3|x\tdesc\nMK-VL\n>h4\nACD EF\n>\nAAAA\n>h6\n'
                 '>h7\nMK*VL1\n2la*\n')
    headers, res, off, records = cluster.read_fasta_for_clustering(str(p))
    assert headers == ['h1', 'h2', 'h3|x', 'h4', '', 'h6', 'h7']
    seqs = [bytes(res[off[i]:off[i + 1]]).decode() for i in range(len(headers))]
    assert seqs == ['MKVLLA',        # trailing '*' stripped
                    'MKVLLA',        # upper-cased, trailing blanks dropped
                    'MKVL',          # '-' is dropped, the sequence is kept (oracle/cluster_ref.c:452-458)
                    'ACDEF',         # inner blank removed
                    'AAAA', '',
                    'MKVLLA']        # inner '*', digits: dropped; lower case folded
    assert records[0] == '>h1 some description\nMKV\nLLA*\n'
    assert off.dtype == np.uint64 and res.dtype == np.uint8


def test_params_follow_the_reference_call():
    p = cluster.params_from_cdhit_args({'-n': 5, '-c': 0.8})          # pangenome.py:45 default
    assert (p.alphabet, p.word_len, p.band_width, p.min_length) == (0, 5, 20, 10)
    assert p.identity == 0.8
    assert (p.aan_cutoff, p.aas_cutoff) == cluster.filter_cutoffs(0.8, 5) == (0.23, 0.61)
    q = cluster.params_from_cdhit_args({'-n': 5, '-c': 0.8}, 'nt')    # the .fna branch (:444)
    assert (q.alphabet, q.both_strands) == (1, 1)
    assert 0 <= q.aan_cutoff < 1e-12 and abs(q.aas_cutoff - 0.2) < 1e-12   # analytic bounds only: 1-(1-c)n, 1-(1-c)4
    assert cluster.params_from_cdhit_args({'-c': 0.8, '-M': 0, '-T': 1}).identity == 0.8   # the unchunked rule itself
    assert cluster.params_from_cdhit_args({'-c': 0.8}, 'nt').word_len == 10               # cd-hit-est's own default
    assert cluster.params_from_cdhit_args({'-c': 0.8}).word_len == 5


def test_filter_cutoffs_never_fall_below_the_analytic_bound():
    for pct in range(40, 101):
        c = pct / 100.0
        for n in (2, 3, 4, 5):
            aan, aas = cluster.filter_cutoffs(c, n)
            assert aan >= 1 - (1 - c) * n - 1e-12 and aas >= 1 - (1 - c) * 2 - 1e-12
            assert 0 <= aan <= 1 and 0 < aas <= 1
    with pytest.raises(ValueError):
        cluster.filter_cutoffs(0.8, 5, tolerance=3)


@pytest.mark.parametrize('bad', [{'-c': 0.3}, {'-n': 6}, {'-n': 1}, {'-c': 0.8, '-l': 2}, {'-c': 0.8, '-G': 0},
                                 {'-c': 0.8, '-d': 20}, {'-c': 0.8, '-sc': 1}, {'-c': 0.8, '-M': 800},
                                 {'-c': 0.8, '-M': 16000}])
def test_params_reject_what_is_not_implemented(bad):
    with pytest.raises(ValueError):
        cluster.params_from_cdhit_args(bad)


def test_cluster_with_cdhit_fails_loudly_without_gpu(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip('a GPU is present')
    from pangenomix_amd import _native, pangenome
    _native._default_ctx = None
    f = tmp_path / 'a.faa'
    f.write_text('>a\nMKVLLAMKVLLAMKVLLA\n')
    with pytest.raises(_native.PgxError):
        pangenome.cluster_with_cdhit(str(f), str(f) + '.cdhit', {'-n': 5, '-c': 0.8})
    assert not (tmp_path / 'a.faa.cdhit.clstr').exists()
