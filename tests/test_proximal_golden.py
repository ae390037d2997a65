"""the 5'/3' UTR (proximal) pangenome builders against what the reference itself produced
(tests/golden/proximal, written by tests/golden/make_golden_next.py running pangenome.py:743-1184):
per-genome extracts, the non-redundant FASTA, labels and .npz members, and what is printed."""
import json
import os
import shutil

import pytest

from pangenomix_amd import pangenome as pg
from test_host_golden import assert_same_npz, same_file

GENOMES = ['p1', 'p2', 'p10']
RUNS = {'upstream': (pg.build_upstream_pangenome, {}, 'Test', 'upstream', ''),
        'downstream': (pg.build_downstream_pangenome, {}, 'Test', 'downstream', ''),
        'upstream_ov5': (pg.build_upstream_pangenome,
                         dict(max_overlap=5, include_fragments=True, fna_output_footer='_ov5', name='O'), 'O', 'upstream', '_ov5')}


@pytest.mark.parametrize('tag', sorted(RUNS))
def test_proximal_pangenome_matches_reference(tag, tmp_path, golden_dir, capsys):
    fn, kw, name, side, footer = RUNS[tag]
    din = tmp_path / 'in'
    shutil.copytree(os.path.join(golden_dir, 'proximal', 'in'), din)
    exp = os.path.join(golden_dir, 'proximal', 'expected', tag)
    out = tmp_path / 'out'
    out.mkdir()
    pairs = [(str(din / (g + '.gff')), str(din / (g + '.fna'))) for g in GENOMES]
    capsys.readouterr()
    df = fn(pairs, str(din / 'T_allele_names.tsv'), str(out), **kw)
    printed = capsys.readouterr().out.replace(str(din), '<in>').replace(str(out), '<out>')
    assert printed == json.load(open(os.path.join(golden_dir, 'proximal', 'expected', 'stdout.json')))[tag]
    for g in GENOMES:
        f = '%s_%s%s.fna' % (g, side, footer)
        same_file(str(din / 'derived' / f), os.path.join(exp, f))
    same_file(str(out / ('%s_nr_%s.fna' % (name, side))), os.path.join(exp, '%s_nr_%s.fna' % (name, side)))
    npz = '%s_strain_by_%s.npz' % (name, side)
    same_file(str(out / (npz + '.labels.txt')), os.path.join(exp, npz + '.labels.txt'))
    assert_same_npz(str(out / npz), os.path.join(exp, npz))
    assert list(df.columns) == ['p10', 'p1', 'p2']      # sorted FILE names ('p10_up...' < 'p1_up...': '0' < '_'), not sorted genome names
    # a second call re-uses the extracts (reference :861) and gives the same table
    df2 = fn(pairs, str(din / 'T_allele_names.tsv'), str(out), **kw)
    assert 'Using pre-existing' in capsys.readouterr().out
    assert (df2.data != df.data).nnz == 0 and list(df2.index) == list(df.index)


def test_feature_map_and_errors(tmp_path, golden_dir):
    din = os.path.join(golden_dir, 'proximal', 'in')
    fmap = pg.__load_feature_to_allele__(os.path.join(din, 'T_allele_names.tsv'))
    assert fmap['fig|p1.peg.2'] == 'T_C10A0' and fmap['fig|p10.peg.2'] == 'T_C10A1' and 'fig|p10.peg.5' not in fmap
    # without a mapping the reference's membership test fails on None (pangenome.py:1166): same here
    with pytest.raises(TypeError):
        pg.extract_upstream_sequences(os.path.join(din, 'p1.gff'), os.path.join(din, 'p1.fna'), str(tmp_path / 'u.fna'))
    # an empty extract has no last record to name: KeyError('') as in the reference (:975-978)
    (tmp_path / 'g_upstream.fna').write_text('')
    with pytest.raises(KeyError):
        pg.consolidate_proximal([str(tmp_path / 'g_upstream.fna')], str(tmp_path / 'nr.fna'), fmap, 'upstream')
