"""
Pan-genome construction on MI355X: host side of the hot path.

Drop-in mirror of the reference entry points in pangenomix/pangenome.py
(file:line cited per function) for the path SURVEY.md §8 scopes:

  build_cds_pangenome / build_noncoding_pangenome
     consolidate_seqs              exact dedupe by sha256          (H1)
   * cluster_with_cdhit            greedy clustering -> HIP (K1/K2) instead of
                                   the `cd-hit` / `cd-hit-est` shell-out
     rename_genes_and_alleles      .clstr -> <name>_C#A# names      (H2)
     build_genetic_feature_tables  allele x genome, gene x genome   (H4)
     LightSparseDataFrame.to_npz   .npz + .labels.txt               (H5)
     extract_noncoding             GFF+FNA -> feature FASTA         (H6)
  build_upstream/downstream/proximal_pangenome   5'/3' UTR pangenomes

Same names, positional order, defaults, intermediate files and return values.
The only behavioural differences are deliberate: a failing
clustering call raises instead of surfacing later as FileNotFoundError, and
unknown `cdhit_args` keys are rejected instead of silently passed through.
"""

from __future__ import print_function
import os
import hashlib
import subprocess as sp

import numpy as np
import scipy.sparse

from . import sparse_utils

LOG_RATE = 10  # reference pangenome.py:32
CLUSTER_TYPES = {'cds': 'C', 'noncoding': 'T'}
VARIANT_TYPES = {'allele': 'A', 'upstream': 'U', 'downstream': 'D'}
_COMPLEMENT = str.maketrans('ACGTWSRYMKNacgtwsrymkn', 'TGCAWSYRKMNtgcawsyrkmn')
_COMPLEMENT_KEYS = frozenset('ACGTWSRYMKNacgtwsrymkn')


# ---------------------------------------------------------------------------
# small helpers (reference pangenome.py:1938-1969, :2040-2059)
# ---------------------------------------------------------------------------
def __get_gene_from_allele__(allele):
    """<name>_C#A# -> <name>_C# (reference :2040-2044: split on the last 'A')."""
    return allele[:allele.rindex('A')] if 'A' in allele else ''


def __get_genome_from_filename__(filepath):
    """Basename without extension (reference :2046-2051)."""
    return os.path.splitext(os.path.split(filepath)[1])[0]


def __get_header_from_fasta_line__(line):
    """First whitespace token minus '>' (reference :2053-2055)."""
    return line.split()[0][1:].strip()


def __hash_sequence__(seq):
    return hashlib.sha256(seq.encode('utf-8')).digest()


def reverse_complement(seq):
    """Reference :1938-1941 (table :37-41); unknown bases raise KeyError there."""
    for base in seq:
        if base not in _COMPLEMENT_KEYS:
            raise KeyError(base)
    return seq.translate(_COMPLEMENT)[::-1]


def create_feature_name(name, cluster_type, cluster_num, variant_type=None, variant_num=-1):
    """<name>_<C|T><cluster>[<A|U|D><variant>] (reference :1944-1969)."""
    short = name + '_' + CLUSTER_TYPES[cluster_type] + str(cluster_num)
    if variant_type is not None and int(variant_num) >= 0:
        short += VARIANT_TYPES[variant_type] + str(variant_num)
    return short


def list_faa_files(directory_path):
    """*.faa in os.listdir order, unsorted (reference :407-423)."""
    return [os.path.join(directory_path, f) for f in os.listdir(directory_path)
            if f.endswith('.faa')]


def find_matching_genome_files(gff_dir, fna_dir):
    """(gff, fna) pairs sharing a basename, in FNA listing order (reference :318-334)."""
    gff = {os.path.splitext(f)[0]: os.path.join(gff_dir, f)
           for f in os.listdir(gff_dir) if f.endswith('.gff')}
    pairs = []
    for f in os.listdir(fna_dir):
        if f.endswith('.fna') and os.path.splitext(f)[0] in gff:
            pairs.append((gff[os.path.splitext(f)[0]], os.path.join(fna_dir, f)))
    return pairs


def _iter_fasta_records(path):
    """Yields (header_token, [stripped sequence lines]) as the reference's line
    scanners see them (:381-391, :638-656): a record starts at a line whose first
    character is '>', sequence lines are .strip()-ed, lines before the first
    header belong to an empty header."""
    header, blocks = '', []
    with open(path, 'r') as f:
        for line in f:
            if line[0] == '>':
                yield header, blocks
                header, blocks = __get_header_from_fasta_line__(line), []
            else:
                blocks.append(line.strip())
    yield header, blocks


def load_sequences_from_fasta(fasta, header_fxn=None, seq_fxn=None, filter_fxn=None):
    """Header -> sequence dict (reference :1892-1916); full header line by default."""
    out = {}
    header, blocks = '', []

    def flush():
        seq = ''.join(blocks)
        if header and seq and (filter_fxn is None or filter_fxn(header)):
            out[header] = seq_fxn(seq) if seq_fxn else seq
    with open(fasta, 'r') as f:
        for line in f:
            if line[0] == '>':
                flush()
                header = line.strip()[1:]
                header = header_fxn(header) if header_fxn else header
                blocks = []
            else:
                blocks.append(line.strip())
    flush()
    return out


# ---------------------------------------------------------------------------
# H1: exact-duplicate consolidation (reference :336-405)
# ---------------------------------------------------------------------------
def consolidate_seqs(genome_paths, nr_out, shared_headers_out, missing_headers_out=None):
    """Merge genome FASTA files into one non-redundant FASTA.

    Files are read in the order given; the first header seen for a sequence is
    its representative and the record is written with its original line
    wrapping. Returns ({sha256 digest: [headers]}, [headers without sequence]).
    """
    groups = {}     # digest -> headers in order observed (insertion order = encounter order)
    missing = []
    with open(nr_out, 'w+') as f_nr:
        for path in genome_paths:
            for header, blocks in _iter_fasta_records(path):
                if not header:
                    continue
                seq = ''.join(blocks)
                if not seq:
                    missing.append(header)
                    continue
                digest = __hash_sequence__(seq)
                known = groups.get(digest)
                if known is None:
                    groups[digest] = [header]
                    f_nr.write('>' + header + '\n' + '\n'.join(blocks) + '\n')
                else:
                    known.append(header)
    with open(shared_headers_out, 'w+') as f:
        for headers in groups.values():
            if len(headers) > 1:
                f.write('\t'.join(headers) + '\n')
    if missing_headers_out:
        print('Headers without sequences:', len(missing))
        with open(missing_headers_out, 'w+') as f:
            for header in missing:
                f.write(header + '\n')
    return groups, missing


# ---------------------------------------------------------------------------
# K1/K2: greedy incremental clustering on the GPU
# ---------------------------------------------------------------------------
def cluster_with_cdhit(fasta_file, cdhit_out, cdhit_args={'-n': 5, '-c': 0.8}):
    """Cluster a FASTA file; writes `cdhit_out` (representatives) and
    `cdhit_out + '.clstr'` exactly where the reference's cd-hit call would
    (reference :425-450). `.fna` input selects the nucleotide (cd-hit-est) rules.

    The clustering itself runs in the HIP library (pangenomix_amd.cluster);
    there is no CPU fallback: a missing library or GPU raises.
    """
    from . import cluster
    cluster.cluster_fasta_to_clstr(fasta_file, cdhit_out, cdhit_args)


# ---------------------------------------------------------------------------
# H2: .clstr -> allele names (reference :453-560)
# ---------------------------------------------------------------------------
def _parse_clstr(clstr_file):
    """Yields (cluster_num_str, member_num_str, header) per member line, using
    exactly the tokens the reference consumes (:505-513, :716-724)."""
    cluster_num = None
    with open(clstr_file, 'r') as f:
        for line in f:
            if line[0] == '>':
                cluster_num = line.split()[-1].strip()
            else:
                data = line.split()
                yield cluster_num, data[0], data[2][1:-3]


def rename_genes_and_alleles(clstr_file, nr_fasta_in, nr_fasta_out,
                             feature_names_out, name='Test', cluster_type='cds',
                             shared_headers_file=None, fastasort_path=None):
    """Name every clustered sequence <name>_C#A# / <name>_T#A#, write the
    name table, and rewrite the non-redundant FASTA with the new headers.
    Returns {original header (incl. synonyms): allele name}."""
    synonyms = {}
    if shared_headers_file:
        with open(shared_headers_file, 'r') as f:
            for line in f:
                headers = line.strip().split('\t')
                synonyms[headers[0]] = headers[1:]

    header_to_allele = {}
    with open(feature_names_out, 'w+') as f_names:
        for cluster_num, member_num, header in _parse_clstr(clstr_file):
            allele = create_feature_name(name, cluster_type, cluster_num, 'allele', member_num)
            header_to_allele[header] = allele
            mapped = [header]
            for syn in synonyms.get(header, ()):
                header_to_allele[syn] = allele
                mapped.append(syn)
            f_names.write(allele + '\t' + '\t'.join(mapped).strip() + '\n')

    tmp = nr_fasta_out + '.tmp'
    with open(nr_fasta_in, 'r') as f_old, open(tmp, 'w+') as f_new:
        keep = False
        for line in f_old:
            if line[0] == '>':
                header = line[1:].strip()
                allele = header_to_allele.get(header)
                keep = allele is not None
                if keep:
                    f_new.write('>' + allele + '\n')
                else:
                    print('MISSING:', header)
            elif keep:
                f_new.write(line)
    if nr_fasta_out == nr_fasta_in:
        os.remove(nr_fasta_in)
    os.rename(tmp, nr_fasta_out)

    if fastasort_path:  # optional external Exonerate fastasort (reference :547-559)
        print('Sorting sequences by header...')
        with open(tmp, 'w+') as f_sort:
            p = sp.Popen(['./' + fastasort_path, nr_fasta_out], stdout=f_sort, stderr=sp.PIPE)
            _, stderr = p.communicate()
            print(stderr)
        if p.returncode == 1:
            print('Aborting sort, exitcode', p.returncode)
            os.remove(tmp)
        else:
            os.rename(tmp, nr_fasta_out)
    return header_to_allele


# ---------------------------------------------------------------------------
# H3: header -> allele map (reference :683-740)
# ---------------------------------------------------------------------------
def load_header_to_allele(clstr_file=None, shared_header_file=None,
                          header_to_allele=None, name='Test', cluster_type='cds'):
    if header_to_allele is None:
        full = {}
        for cluster_num, member_num, header in _parse_clstr(clstr_file):
            full[header] = create_feature_name(name, cluster_type, cluster_num, 'allele', member_num)
    else:
        full = dict(header_to_allele)
    if shared_header_file:
        with open(shared_header_file, 'r') as f:
            for line in f:
                headers = [x.strip() for x in line.split('\t')]
                for alt in headers[1:]:
                    full[alt] = full[headers[0]]
    return full


# ---------------------------------------------------------------------------
# H4: allele x genome and gene x genome tables (reference :563-680)
# ---------------------------------------------------------------------------
def _first_occurrence_coo(rows, cols, n_rows, n_cols):
    """COO with ones, duplicates dropped, triples in order of first insertion:
    the order scipy's dok_matrix.tocoo() yields for the reference's loop
    (:649-650), pinned by tests/golden/cds."""
    from . import _native
    rows = np.asarray(rows, dtype=np.int64)
    cols = np.asarray(cols, dtype=np.int64)
    first = _native.first_insertions(rows, cols, n_cols)            # hash-based: linear, keeps the order
    data = np.ones(first.size, dtype=np.int64)
    return scipy.sparse.coo_matrix(
        (data, (rows[first].astype(np.int32), cols[first].astype(np.int32))),
        shape=(n_rows, n_cols))


def build_genetic_feature_tables(clstr_file, genome_fasta_paths, name='Test', cluster_type='cds',
                                 output_format='lsdf', shared_header_file=None,
                                 header_to_allele=None, log_rate=LOG_RATE):
    """Binary allele x genome and gene x genome tables as LSDFs.

    Row labels are the lexicographically sorted allele names (so C100 < C10 <
    C11 < C1 < C2, reference :615) and their consecutive-deduped gene prefixes
    (:618-623); columns are the sorted genome basenames (:612). Genome files are
    scanned in sorted(path) order (:635) and a record counts only if it has a
    non-empty sequence (:643).
    """
    print('Loadings header-allele mappings...')
    # NB the reference passes cluster_type into the `name` slot here (:608-609);
    # harmless because the dict is always supplied on the hot path.
    header_to_allele = load_header_to_allele(clstr_file, shared_header_file,
                                             header_to_allele, cluster_type)
    genome_order = sorted(__get_genome_from_filename__(p) for p in genome_fasta_paths)
    print('Sorting alleles...')
    allele_order = sorted(set(header_to_allele.values()))
    print('Sorting clusters...')
    gene_of_allele = np.empty(len(allele_order), dtype=np.int64)
    gene_order, last = [], None
    for i, allele in enumerate(allele_order):
        gene = __get_gene_from_allele__(allele)
        if gene != last:
            gene_order.append(gene)
            last = gene
        gene_of_allele[i] = len(gene_order) - 1
    print('Genomes:', len(genome_order))
    print('Clusters:', len(gene_order))
    print('Alleles:', len(allele_order))

    allele_index = {a: i for i, a in enumerate(allele_order)}
    header_index = {h: allele_index[a] for h, a in header_to_allele.items()}
    genome_index = {}
    for i, g in enumerate(genome_order):
        genome_index.setdefault(g, i)      # list.index() semantics: first match

    rec_allele, rec_genome = [], []
    for i, path in enumerate(sorted(genome_fasta_paths)):
        genome = __get_genome_from_filename__(path)
        genome_i = genome_index[genome]
        for header, blocks in _iter_fasta_records(path):
            if not any(blocks):
                continue
            allele_i = header_index.get(header)
            if allele_i is None:
                print('MISSING:', header)
                continue
            rec_allele.append(allele_i)
            rec_genome.append(genome_i)
        if (i + 1) % log_rate == 0:
            print('Updating genome', i + 1, ':', genome)

    print('Building binary matrix...')
    rec_allele = np.asarray(rec_allele, dtype=np.int64)
    rec_genome = np.asarray(rec_genome, dtype=np.int64)
    sp_alleles = _first_occurrence_coo(rec_allele, rec_genome, len(allele_order), len(genome_order))
    sp_genes = _first_occurrence_coo(gene_of_allele[rec_allele] if rec_allele.size else rec_allele,
                                     rec_genome, len(gene_order), len(genome_order))
    df_alleles = sparse_utils.LightSparseDataFrame(allele_order, genome_order, sp_alleles)
    df_genes = sparse_utils.LightSparseDataFrame(gene_order, genome_order, sp_genes)
    if output_format == 'sparr':
        raise NotImplementedError(
            "output_format='sparr' is the legacy pandas-SparseArray format that the "
            "reference itself no longer supports on pandas 2 (SURVEY App. B.8); use 'lsdf'")
    return df_alleles, df_genes


# ---------------------------------------------------------------------------
# H6: non-coding feature extraction (reference :1187-1243)
# ---------------------------------------------------------------------------
def extract_noncoding(genome_gff, genome_fna, noncoding_out, flanking=(0, 0),
                      allowed_features=['transcript', 'tRNA', 'rRNA', 'misc_binding']):
    """Write the nucleotide sequence of every allowed GFF feature (PATRIC
    flavour: contig column is 'accn|<contig>', 5 characters trimmed, :1221)."""
    contigs = load_sequences_from_fasta(genome_fna, header_fxn=lambda x: x.split()[0])
    with open(noncoding_out, 'w+') as f_out, open(genome_gff, 'r') as f_gff:
        for line in f_gff:
            if line[0] == '#' or len(line.strip()) == 0:
                continue
            contig, _src, ftype, start, stop, _score, strand, _phase, meta = line.split('\t')
            contig = contig[5:]
            start, stop = int(start), int(stop)
            if ftype not in allowed_features or contig not in contigs:
                continue
            lo = max(0, start - 1 - flanking[0])
            seq = contigs[contig][lo:stop + flanking[1]]
            if strand == '-':
                seq = reverse_complement(seq)
            fields = [kv.split('=') for kv in meta.split(';')]
            feature_id = {kv[0]: kv[1] for kv in fields}['ID']
            wrapped = '\n'.join(seq[i:i + 70] for i in range(0, len(seq), 70))
            f_out.write('>' + feature_id + '\n' + wrapped + '\n')


# ---------------------------------------------------------------------------
# the same pipeline on libpgx's host side: consolidate -> cluster -> name -> tables
# ---------------------------------------------------------------------------
def _lex_key(values, shorter_first):
    """Sort key under which non-negative integers order like their decimal strings do inside an
    allele name: digit by digit, and where one string ends the other continues with a digit, the
    shorter one sorts first (`shorter_first`: the name ends there) or last (the cluster number is
    followed by the variant letter 'A', which sorts after every digit: C100 < C10 < C1, reference
    :615)."""
    v = np.asarray(values, dtype=np.int64)
    pow10 = 10 ** np.arange(1, 19, dtype=np.int64)
    nd = np.searchsorted(pow10, v, side='right') + 1             # number of decimal digits
    width = int(nd.max()) if nd.size else 1
    # digits left-aligned to the longest number, padded with zeros where the name ends after the number (a shorter
    # string sorts first: 1 < 10 < 100) and with nines where the letter follows (it sorts after every digit:
    # 19A < 1A, and among equals the longer first: 100A < 10A < 1A); the digit count breaks the ties that
    # padding creates (all lengths are at most 18: 5 bits)
    scale = np.concatenate(([1], pow10))[width - nd]
    if shorter_first:
        return v * scale * 32 + nd
    return ((v + 1) * scale - 1) * 32 + (31 - nd)


class _Beside(object):
    """A step of the pipeline that runs beside the caller's next steps on a thread of its own (the library's file
    work releases the GIL). wait() returns when it is done and raises what it raised."""

    def __init__(self, fn):
        import threading
        self._exc = None

        def run():
            try:
                fn()
            except BaseException as exc:          # handed to the thread that waits
                self._exc = exc
        self._thread = threading.Thread(target=run)
        self._thread.start()

    def wait(self):
        self._thread.join()
        exc, self._exc = self._exc, None
        if exc is not None:
            raise exc


def _native_pipeline(genome_paths, nr_fasta, shared, missing, names_tsv, name, cluster_type, cdhit_args,
                     fastasort_path, cluster_fn=None, after_tables=None):
    """consolidate_seqs -> cluster_with_cdhit -> rename_genes_and_alleles -> build_genetic_feature_tables
    with the file work in libpgx's multi-threaded host code (csrc/ingest.cpp) and the tables from
    arrays instead of per-record dictionaries: the same files, byte for byte, and the same LSDFs as the
    step-by-step functions above (tests/test_host_golden.py pins both against the reference's output).
    Returns None when the fast path does not apply -- inputs the reference's line-by-line semantics
    treat specially (see pgx.h), duplicate genome names, an external fastasort, a multi-process group:
    the caller then runs the step-by-step functions. `cluster_fn(residues, offsets, params)` replaces the
    GPU clustering call in tests (the CPU oracle, or a reader of a given .clstr). `after_tables(tables)` is called once
    the tables exist, while the text outputs may still be being written (the caller's .npz files go there)."""
    import sys
    import time
    from . import _native, cluster
    genomes = [__get_genome_from_filename__(p) for p in genome_paths]
    if fastasort_path or cluster._group is not None or len(set(genomes)) != len(genomes):
        return None
    t_mark = [time.perf_counter()]

    def lap(what):   # PGX_TRACE: wall time of the pipeline's stages
        if os.environ.get('PGX_TRACE'):
            now = time.perf_counter()
            print('[pgx] pipeline: %-28s %8.1f ms' % (what, (now - t_mark[0]) * 1e3), file=sys.stderr)
            t_mark[0] = now
    try:
        fs = _native.FastaSet(genome_paths)
    except _native.PgxError as exc:
        if getattr(exc, 'status', 0) != _native.ERR_NOMEM:
            raise
        print('Note: taking the step-by-step path (%s)' % exc)   # the native ingest holds 2-3x the input in memory
        return None
    lap('ingest (parse, sha256, dedupe)')
    beside = []           # file work in flight: it reads the set's memory, so it is waited for before the set is closed
    try:
        if not fs.simple:
            print('Note: taking the step-by-step path (%s)' % fs.why)
            return None
        # H1 (:336-405); the nr FASTA itself is written once, below, with the allele names (:524-544 rewrites it)
        # (the header files do not depend on the clustering and are written beside it)
        headers_written = _Beside(lambda: fs.write_consolidated(None, shared, missing))
        beside.append(headers_written)
        print('Headers without sequences:', fs.n_missing)
        nucleotide = nr_fasta[-4:].lower() == '.fna'                           # K1/K2 (:425-450)
        params = cluster.params_from_cdhit_args(cdhit_args, 'nt' if nucleotide else 'aa')
        print('Running: libpgx greedy clustering (%s rules) -i %s -o %s -c %g -n %d' % (
            'cd-hit-est' if nucleotide else 'cd-hit', nr_fasta, nr_fasta + '.cdhit', params.identity, params.word_len))
        if cluster_fn is None:      # (nobody reads the work counters here: the library leaves out what only they need)
            cl, mem, iden, strand, n_clusters = cluster.cluster_sequences(fs.residues, fs.offsets, params, want_stats=False)[:5]
        else:
            cl, mem, iden, strand, n_clusters = cluster_fn(fs.residues, fs.offsets, params)[:5]
        print('%9d  finished  %9d  clusters' % (int((cl >= 0).sum()), n_clusters))
        lap('clustering (H2D included)')
        headers_written.wait()
        lap('redundant / missing headers (the rest)')
        prefix = name + '_' + CLUSTER_TYPES[cluster_type]                      # H2 (:453-560)

        def write_outputs():
            fs.write_clustered(cl, mem, iden, strand, nucleotide, prefix, VARIANT_TYPES['allele'],
                               clstr_path=nr_fasta + '.cdhit.clstr', names_path=names_tsv, nr_out_path=nr_fasta + '.tmp')
            os.replace(nr_fasta + '.tmp', nr_fasta)
        unclustered = np.flatnonzero(cl < 0)
        if unclustered.size:
            for h in fs.headers(fs.rep_of_group[unclustered]):
                print('MISSING:', h)

        print('Loadings header-allele mappings...')                            # H4 (:563-680)
        genome_order = sorted(genomes)
        print('Sorting alleles...')
        # rows of the allele table = the clustered sequences in the order of their names, genes = runs of one cluster;
        # records in sorted(path) order, file order inside (:635); a record counts if it has a sequence (:643); a pair
        # (row, genome) is kept where it is inserted first (:649-650) -- one pass over the parsed files in the library
        genome_of_file = np.array([genome_order.index(g) for g in genomes], dtype=np.int64)
        file_order = np.argsort(np.array(genome_paths, dtype=object), kind='stable')
        t = fs.feature_coo(cl, mem, file_order, genome_of_file)
        allele_groups = t['allele_groups']
        lap('tables: coordinates')
        # (the three text files and the rest of the tables need nothing of each other: the files are written meanwhile;
        # started behind the coordinates, which use all cores themselves)
        outputs_written = _Beside(write_outputs)
        beside.append(outputs_written)
        print('Sorting clusters...')
        c_sorted = cl[allele_groups]
        new_gene = np.ones(c_sorted.size, dtype=bool)
        new_gene[1:] = c_sorted[1:] != c_sorted[:-1]
        allele_order = _native.format_labels(prefix, c_sorted, mem[allele_groups], VARIANT_TYPES['allele'])
        gene_order = _native.format_labels(prefix, c_sorted[new_gene])
        lap('tables: names')
        print('Genomes:', len(genome_order))
        print('Clusters:', len(gene_order))
        print('Alleles:', len(allele_order))
        if t['lost_records'].size:
            for h in fs.headers(t['lost_records']):
                print('MISSING:', h)
        print('Building binary matrix...')

        def ones_coo(row, col, shape):
            return scipy.sparse.coo_matrix((np.ones(row.size, dtype=np.int64), (row, col)), shape=shape)
        sp_alleles = ones_coo(t['a_row'], t['a_col'], (len(allele_order), len(genome_order)))
        sp_genes = ones_coo(t['g_row'], t['g_col'], (len(gene_order), len(genome_order)))
        out = (sparse_utils.LightSparseDataFrame(allele_order, genome_order, sp_alleles),
               sparse_utils.LightSparseDataFrame(gene_order, genome_order, sp_genes))
        lap('tables: matrices')
        if cluster_fn is None:
            # Device-resident hand-off: the gene x genome bitmap is built on the GPU straight from the clustering
            # result (rows = cluster numbers; the curves do not depend on the row order) and stays in the context;
            # estimate_pan_core_size(df_genes) on the returned table uses it instead of uploading the table's
            # coordinates, as long as the table is the object returned here and the bitmap is still resident.
            try:
                ctx = _native.default_context()
                token = ctx.bitmap_from_clusters(cl, fs.group_of_record, fs.file_of_record, genome_of_file,
                                                 len(gene_order), len(genome_order))
                out[1]._pgx_resident = {'ctx': ctx, 'token': token, 'shape': out[1].shape, 'data': out[1].data,
                                        'nnz': int(out[1].data.nnz)}
            except _native.PgxError as exc:       # (the tables are complete without it)
                print('Note: no device-resident bitmap (%s)' % exc)
            lap('device-resident bitmap')
        if after_tables is not None:
            after_tables(out)
        outputs_written.wait()
        lap('.clstr, names, nr FASTA (the rest)')
        return out
    finally:
        for b in beside:
            try:
                b.wait()
            except BaseException:     # (reported by the wait() above unless something else failed first)
                pass
        # (giving back the set's memory, a GB for the benchmark's input, takes 50 ms: nobody waits for it)
        import threading
        threading.Thread(target=fs.close).start()


# ---------------------------------------------------------------------------
# entry points (reference :44-156, :159-316)
# ---------------------------------------------------------------------------
def _save_both(df_alleles, allele_npz, df_genes, gene_npz):
    """The two tables of a pangenome, written side by side (the deflate of the .npz members releases the GIL);
    the messages come in the reference's order."""
    import sys
    import threading
    import time
    t0 = time.perf_counter()
    print('Saving', allele_npz, '...')
    errors = []

    def save_genes():
        try:
            df_genes.to_npz(gene_npz)
        except BaseException as exc:      # re-raised by the caller's thread
            errors.append(exc)
    t = threading.Thread(target=save_genes)
    t.start()
    try:
        df_alleles.to_npz(allele_npz)
    finally:
        print('Saving', gene_npz, '...')
        t.join()
    if errors:
        raise errors[0]
    if os.environ.get('PGX_TRACE'):
        print('[pgx] pipeline: %-28s %8.1f ms' % ('.npz + labels', (time.perf_counter() - t0) * 1e3), file=sys.stderr)


def _check_format(output_format):
    if output_format not in {'lsdf', 'sparr'}:
        print('Unrecognized output format, switching to lsdf')
        return 'lsdf'
    return output_format


def _p(output_dir, name, suffix):
    return (output_dir + '/' + name + suffix).replace('//', '/')


def build_cds_pangenome(genome_faa_paths, output_dir, name='Test',
                        cdhit_args={'-n': 5, '-c': 0.8}, fastasort_path=None,
                        output_format='lsdf'):
    """Protein pan-genome: dedupe -> cluster (HIP) -> name -> tables -> npz.
    Output files and return value as the reference (:44-156)."""
    output_format = _check_format(output_format)
    print('Identifying non-redundant CDS sequences...')
    nr_faa = _p(output_dir, name, '_nr.faa')
    shared = _p(output_dir, name, '_redundant_headers.tsv')
    missing = _p(output_dir, name, '_missing_headers.txt')
    allele_npz = _p(output_dir, name, '_strain_by_allele') + '.npz'
    gene_npz = _p(output_dir, name, '_strain_by_gene') + '.npz'
    fast = _native_pipeline(genome_faa_paths, nr_faa, shared, missing, _p(output_dir, name, '_allele_names.tsv'),
                            name, 'cds', cdhit_args, fastasort_path,
                            after_tables=lambda t: _save_both(t[0], allele_npz, t[1], gene_npz)) \
        if output_format == 'lsdf' else None
    if fast is not None:
        return fast                 # (the tables were saved beside the last of the text outputs)
    else:
        consolidate_seqs(genome_faa_paths, nr_faa, shared, missing)

        cluster_with_cdhit(nr_faa, nr_faa + '.cdhit', cdhit_args)
        os.remove(nr_faa + '.cdhit')
        clstr = nr_faa + '.cdhit.clstr'

        header_to_allele = rename_genes_and_alleles(
            clstr, nr_faa, nr_faa, _p(output_dir, name, '_allele_names.tsv'), name=name,
            cluster_type='cds', shared_headers_file=shared, fastasort_path=fastasort_path)
        df_alleles, df_genes = build_genetic_feature_tables(
            clstr, genome_faa_paths, name, cluster_type='cds',
            output_format=output_format, header_to_allele=header_to_allele)

    allele_npz = _p(output_dir, name, '_strain_by_allele') + '.npz'
    gene_npz = _p(output_dir, name, '_strain_by_gene') + '.npz'
    _save_both(df_alleles, allele_npz, df_genes, gene_npz)
    return df_alleles, df_genes


def build_noncoding_pangenome(genome_data, output_dir, name='Test', flanking=(0, 0),
                              allowed_features=['transcript', 'tRNA', 'rRNA', 'misc_binding'],
                              cdhit_args={'-n': 5, '-c': 0.8}, fastasort_path=None,
                              output_format='lsdf', fna_output_footer='', overwrite_extract=False):
    """Non-coding pan-genome from (gff, fna) pairs; nucleotide clustering
    (cd-hit-est rules) because the merged file ends in .fna (reference :159-316)."""
    output_format = _check_format(output_format)
    print('Extracting non-coding sequences...')
    nc_paths = []
    for i, (gff, fna) in enumerate(genome_data):
        genome = __get_genome_from_filename__(gff)
        gdir = '/'.join(gff.split('/')[:-1]) + '/' if '/' in gff else ''
        nc_dir = gdir + 'derived/'
        if not os.path.exists(nc_dir):
            os.mkdir(nc_dir)
        nc = nc_dir + genome + '_noncoding' + fna_output_footer + '.fna'
        nc_paths.append(nc)
        if os.path.exists(nc) and not overwrite_extract:
            print(i + 1, 'Using pre-existing noncoding sequences for', genome)
        else:
            print(i + 1, 'Extracting noncoding regions for', genome)
            extract_noncoding(gff, fna, nc, flanking=flanking, allowed_features=allowed_features)

    print('Identifying non-redundant non-coding sequences...')
    nr_fna = _p(output_dir, name, '_noncoding_nr.fna')
    shared = _p(output_dir, name, '_noncoding_redundant_headers.tsv')
    missing = _p(output_dir, name, '_noncoding_missing_headers.txt')
    fast = _native_pipeline(nc_paths, nr_fna, shared, missing, _p(output_dir, name, '_noncoding_allele_names.tsv'),
                            name, 'noncoding', cdhit_args, fastasort_path) if output_format == 'lsdf' else None
    if fast is not None:
        df_alleles, df_genes = fast
    else:
        consolidate_seqs(nc_paths, nr_fna, shared, missing)

        cluster_with_cdhit(nr_fna, nr_fna + '.cdhit', cdhit_args)
        os.remove(nr_fna + '.cdhit')
        clstr = nr_fna + '.cdhit.clstr'

        header_to_allele = rename_genes_and_alleles(
            clstr, nr_fna, nr_fna, _p(output_dir, name, '_noncoding_allele_names.tsv'), name=name,
            cluster_type='noncoding', shared_headers_file=shared, fastasort_path=fastasort_path)
        df_alleles, df_genes = build_genetic_feature_tables(
            clstr, nc_paths, name, cluster_type='noncoding',
            output_format=output_format, header_to_allele=header_to_allele)
    # plain lists on purpose; the maps are not refreshed (reference :292-293, App. B.5)
    strip = '_noncoding' + fna_output_footer
    df_alleles.columns = [x.replace(strip, '') for x in df_alleles.columns]
    df_genes.columns = [x.replace(strip, '') for x in df_genes.columns]

    allele_npz = _p(output_dir, name, '_strain_by_noncoding_allele') + '.npz'
    gene_npz = _p(output_dir, name, '_strain_by_noncoding_gene') + '.npz'
    _save_both(df_alleles, allele_npz, df_genes, gene_npz)
    return df_alleles, df_genes

# ---------------------------------------------------------------------------
# proximal (5'/3' UTR) pangenomes (reference :743-1184, :2027-2038)
# ---------------------------------------------------------------------------
# Host-only rows: exact-match grouping per gene instead of clustering, the same table machinery. Mirrored
# statement by statement where Python's own semantics decide the outcome (negative slice starts at contig
# ends, dictionary insertion order, the unconditional last record); tests/golden/proximal holds what the
# reference produced.
def __load_feature_to_allele__(allele_names):
    """{'fig|<genome>.peg.#': allele} from <name>_allele_names.tsv: every synonym of a line, cut to its
    first two '|' fields (reference :2027-2038)."""
    feat_to_allele = {}
    with open(allele_names, 'r') as f:
        for line in f:
            data = line.strip().split('\t')
            for synonym in data[1:]:
                feat_to_allele['|'.join(synonym.split('|')[:2])] = data[0]
    return feat_to_allele


def extract_proximal_sequences(genome_gff, genome_fna, proximal_out, limits, max_overlap, side,
                               feature_to_allele=None, allele_names=None, include_fragments=False):
    """Nucleotides upstream / downstream of every mapped GFF feature (PATRIC flavour: contig column
    'accn|<contig>', ID attribute) written as '<ID>_<side>(<limits>[,<max_overlap>])' records
    (reference :1038-1184). limits = (-X, Y): upstream X bases before the start codon + the first Y coding
    bases; downstream the last X coding bases + Y bases after the stop. max_overlap >= 0 truncates a region
    that runs into the neighbouring CDS on the same strand (GFF order). Regions cut off by a contig end are
    dropped unless include_fragments."""
    neighbours = {}      # contig -> strand -> (start, stop) -> (end of the CDS before, start of the CDS after)
    if max_overlap >= 0:
        placed = {}
        with open(genome_gff, 'r') as f_gff:
            for line in f_gff:
                line = line.strip()
                if len(line) > 0 and line[0] != '#':
                    contig, _src, ftype, start, stop, _score, strand, _phase, _attr = line.split('\t')
                    if ftype == 'CDS':
                        contig = contig.split('|')[-1]
                        placed.setdefault(contig, {'+': [], '-': []})[strand].append((int(start) - 1, int(stop)))
        for contig, by_strand in placed.items():
            neighbours[contig] = {'+': {}, '-': {}}
            for strand, feats in by_strand.items():
                for i, feat in enumerate(feats):
                    left = -np.inf if i == 0 else feats[i - 1][1]
                    right = np.inf if i == len(feats) - 1 else feats[i + 1][0]
                    neighbours[contig][strand][feat] = (left, right)

    contigs = load_sequences_from_fasta(genome_fna, header_fxn=lambda x: x.split()[0])
    if feature_to_allele:
        feat_to_allele = feature_to_allele
    elif allele_names:
        feat_to_allele = __load_feature_to_allele__(allele_names)
    else:
        feat_to_allele = None

    params = (limits[0], limits[1], max_overlap) if max_overlap >= 0 else limits
    footer = '_' + side + str(params).replace(' ', '')
    coding_length = limits[1] if side == 'upstream' else -limits[0]
    count = 0
    with open(proximal_out, 'w+') as f_prox, open(genome_gff, 'r') as f_gff:
        for line in f_gff:
            line = line.strip()
            if len(line) == 0 or line[0] == '#':
                continue
            contig, _src, _ftype, start, stop, _score, strand, _phase, attr_raw = line.split('\t')
            contig = contig.split('|')[-1]
            start, stop = int(start) - 1, int(stop)
            attrs = {}
            for entry in attr_raw.split(';'):
                k, v = entry.split('=')
                attrs[k] = v
            gffid = attrs['ID']
            if contig not in contigs:
                continue
            # (evaluated in the reference's order: without a mapping the membership test itself fails, :1166)
            if not (gffid in feat_to_allele or feat_to_allele is None):
                continue
            seq = contigs[contig]
            anchor = start if (side, strand) in (('upstream', '+'), ('downstream', '-')) else stop
            lo, hi = limits if strand == '+' else (-limits[1], -limits[0])
            utr_start, utr_stop = anchor + lo, anchor + hi
            if max_overlap >= 0:
                left, right = neighbours[contig][strand][(start, stop)]
                if utr_start < left - max_overlap:
                    utr_start = left - max_overlap
                if utr_stop > right + max_overlap:
                    utr_stop = right + max_overlap
            proximal = seq[utr_start:utr_stop].strip()      # (a negative start counts from the contig's end, as there)
            if strand == '-':
                proximal = reverse_complement(proximal)
            is_fragment = utr_start < 0 or utr_stop > len(seq)
            if len(proximal) > coding_length and (not is_fragment or include_fragments):
                f_prox.write('>' + gffid + footer + '\n' + proximal + '\n')
                count += 1
    print('Loaded', side, 'sequences:', count)


def extract_upstream_sequences(genome_gff, genome_fna, upstream_out, limits=(-50, 3), max_overlap=-1,
                               feature_to_allele=None, allele_names=None, include_fragments=False):
    """Reference :1021-1029."""
    extract_proximal_sequences(genome_gff, genome_fna, proximal_out=upstream_out, limits=limits, max_overlap=max_overlap,
                               side='upstream', feature_to_allele=feature_to_allele, allele_names=allele_names,
                               include_fragments=include_fragments)


def extract_downstream_sequences(genome_gff, genome_fna, downstream_out, limits=(-3, 50), max_overlap=-1,
                                 feature_to_allele=None, allele_names=None, include_fragments=False):
    """Reference :1032-1040."""
    extract_proximal_sequences(genome_gff, genome_fna, proximal_out=downstream_out, limits=limits, max_overlap=max_overlap,
                               side='downstream', feature_to_allele=feature_to_allele, allele_names=allele_names,
                               include_fragments=include_fragments)


def consolidate_proximal(genome_proximals, nr_proximal_out, feature_to_allele, side, output_format='lsdf'):
    """Non-redundant proximal sequences PER GENE (<name>_C#U# upstream, <name>_C#D# downstream; the number
    is the order of first appearance within the gene) and the proximal x genome table (reference :900-1018).
    Files are taken in sorted order; the genome is the file name up to '_<side>'."""
    letter = VARIANT_TYPES[side]
    variants = {}        # gene -> {sequence: number}
    per_genome = {}      # genome -> {proximal id: 1}, insertion order = first appearance
    genome_order = []
    ids = set()
    with open(nr_proximal_out, 'w+') as f_nr:
        def record(header, seq, genome):
            feature = header.split('_' + side + '(')[0]
            gene = __get_gene_from_allele__(feature_to_allele[feature])
            known = variants.setdefault(gene, {})
            is_new = seq not in known
            if is_new:
                known[seq] = len(known)
            prox_id = gene + letter + str(known[seq])
            ids.add(prox_id)
            per_genome[genome][prox_id] = 1
            if is_new:
                f_nr.write('>' + prox_id + '\n' + seq + '\n')
        for path in sorted(genome_proximals):
            genome = path.split('/')[-1].split('_' + side)[0]
            per_genome[genome] = {}
            genome_order.append(genome)
            header, seq = '', ''
            with open(path, 'r') as f:
                for line in f.readlines():
                    if line[0] == '>':
                        if len(seq) > 0:
                            record(header, seq, genome)
                        header, seq = line[1:].strip(), ''
                    else:
                        seq += line.strip()
            record(header, seq, genome)          # the last record, unconditionally (:975-990)
    print('Sparsifying', side, 'table...')
    prox_order = sorted(ids)
    index_of = {p: i for i, p in enumerate(prox_order)}
    rows, cols = [], []
    for genome_i, genome in enumerate(genome_order):
        for prox_id in per_genome[genome]:
            rows.append(index_of[prox_id])
            cols.append(genome_i)
    print('Building binary matrix...')
    data = _first_occurrence_coo(rows, cols, len(prox_order), len(genome_order))
    if output_format == 'sparr':
        raise NotImplementedError("output_format='sparr' is not supported (SURVEY App. B.8); use 'lsdf'")
    return sparse_utils.LightSparseDataFrame(prox_order, genome_order, data)


def build_proximal_pangenome(genome_data, allele_names, output_dir, limits, side, name='Test',
                             include_fragments=False, max_overlap=-1, fastasort_path=None,
                             output_format='lsdf', fna_output_footer='', overwrite_extract=False):
    """Proximal-region pangenome relative to the gene clusters of build_cds_pangenome(): per genome
    `<gffdir>/derived/<genome>_<side><footer>.fna`, then `<name>_nr_<side>.fna` and
    `<name>_strain_by_<side>.npz` (reference :777-897)."""
    output_format = _check_format(output_format)
    print('Loading header-allele mapping...')
    feature_to_allele = __load_feature_to_allele__(allele_names)
    print('Extracting', side, 'sequences...')
    genome_proximals = []
    for i, (gff, fna) in enumerate(genome_data):
        genome = __get_genome_from_filename__(gff)
        gdir = '/'.join(gff.split('/')[:-1]) + '/' if '/' in gff else ''
        prox_dir = gdir + 'derived/'
        if not os.path.exists(prox_dir):
            os.mkdir(prox_dir)
        prox = prox_dir + genome + '_' + side + fna_output_footer + '.fna'
        genome_proximals.append(prox)
        if os.path.exists(prox) and not overwrite_extract:
            print(i + 1, 'Using pre-existing', side, 'regions for', genome)
        else:
            print(i + 1, 'Extracting', side, 'regions for', genome)
            extract_proximal_sequences(gff, fna, prox, limits=limits, side=side, feature_to_allele=feature_to_allele,
                                       include_fragments=include_fragments, max_overlap=max_overlap)
    print('Identifying non-redundant', side, 'sequences per gene...')
    nr_out = _p(output_dir, name, '_nr_' + side + '.fna')
    df_proximal = consolidate_proximal(genome_proximals, nr_out, feature_to_allele, side, output_format=output_format)
    if fastasort_path:
        print('Sorting sequences by header...')
        with open(nr_out + '.tmp', 'w+') as f_sort:
            sp.call(['./' + fastasort_path, nr_out], stdout=f_sort)
        os.rename(nr_out + '.tmp', nr_out)
    npz = _p(output_dir, name, '_strain_by_' + side) + '.npz'
    print('Saving', npz, '...')
    df_proximal.to_npz(npz)
    return df_proximal


def build_upstream_pangenome(genome_data, allele_names, output_dir, limits=(-50, 3), name='Test',
                             include_fragments=False, max_overlap=-1, fastasort_path=None, output_format='lsdf',
                             fna_output_footer='', overwrite_extract=False):
    """5'UTR pangenome (reference :743-757)."""
    return build_proximal_pangenome(genome_data, allele_names, output_dir, limits, side='upstream', name=name,
                                    include_fragments=include_fragments, max_overlap=max_overlap,
                                    fastasort_path=fastasort_path, output_format=output_format,
                                    fna_output_footer=fna_output_footer, overwrite_extract=overwrite_extract)


def build_downstream_pangenome(genome_data, allele_names, output_dir, limits=(-3, 50), name='Test',
                               include_fragments=False, max_overlap=-1, fastasort_path=None, output_format='lsdf',
                               fna_output_footer='', overwrite_extract=False):
    """3'UTR pangenome (reference :761-775)."""
    return build_proximal_pangenome(genome_data, allele_names, output_dir, limits, side='downstream', name=name,
                                    include_fragments=include_fragments, max_overlap=max_overlap,
                                    fastasort_path=fastasort_path, output_format=output_format,
                                    fna_output_footer=fna_output_footer, overwrite_extract=overwrite_extract)
