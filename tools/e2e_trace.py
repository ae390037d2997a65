"""Stage times of build_cds_pangenome() on the benchmark's 400-genome set (development aid): PGX_TRACE laps of the native
pipeline + the time of the .npz writes.  usage: PGX_TRACE=1 python tools/e2e_trace.py [genomes]"""
import os
import shutil
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pangenomix_amd import pangenome, synth   # noqa: E402

os.environ.setdefault('PGX_TRACE', '1')
pset = synth.protein_set('cfg-3s')
if len(sys.argv) > 1:
    pset = synth.ProteinSet(int(sys.argv[1]), pset.cds, pset.F, pset.C, pset.seed)
tmp = tempfile.mkdtemp(prefix='pgx_e2e_')
try:
    paths = pset.write_faa(os.path.join(tmp, 'genomes'))
    for rep in range(3):
        out = os.path.join(tmp, 'out%d' % rep)
        os.mkdir(out)
        so = sys.stdout
        with open(os.devnull, 'w') as null:
            sys.stdout = null
            try:
                t = time.perf_counter()
                pangenome.build_cds_pangenome(paths, out, name='Bench')
                e = time.perf_counter() - t
            finally:
                sys.stdout = so
        print('run %d: %.3f s' % (rep, e), file=sys.stderr)
finally:
    shutil.rmtree(tmp, ignore_errors=True)
