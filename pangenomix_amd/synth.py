"""Deterministic synthetic inputs for tests and bench.py (SURVEY.md §8d recipe; the
Bacteroides .faa sets the reference was used on are not available offline).

Protein set: F family ancestors with lengths clip(gamma(2.2, 150), 40, 2500) over the 20
standard letters; each genome holds C core families plus accessory families drawn without
replacement; an instance copies its ancestor, with prob. 0.6 substitutes 2 % of its sites,
a 5 % tier substitutes 15-25 % (straddles the 0.8 threshold), and 1 % of the records are
<= 10 residues long (exercises cd-hit's discard length).
"""

import os
import sys

import numpy as np

AA20 = np.frombuffer(b'ACDEFGHIKLMNPQRSTVWY', dtype=np.uint8)

CONFIGS = {
    # name: (genomes, cds/genome, families, core families, seed)
    'tiny': (6, 60, 150, 30, 6),
    'small': (20, 400, 1500, 200, 20),
    'cfg-2s': (50, 4500, 30000, 2000, 50),
    'cfg-3s': (400, 4500, 30000, 2000, 400),
    'cfg-4': (4000, 3000, 60000, 1500, 4000),
}


class ProteinSet(object):
    def __init__(self, n_genomes, cds_per_genome, n_families, n_core, seed):
        self.n_genomes, self.cds, self.F, self.C, self.seed = n_genomes, cds_per_genome, n_families, n_core, seed
        rng = np.random.default_rng(seed)
        lens = np.clip(rng.gamma(2.2, 150.0, n_families), 40, 2500).astype(np.int64)
        self.fam_off = np.zeros(n_families + 1, dtype=np.int64)
        np.cumsum(lens, out=self.fam_off[1:])
        self.fam_res = AA20[rng.integers(0, 20, int(self.fam_off[-1]))]
        self.fam_len = lens

    def genome(self, g):
        """(family ids, [bytes sequence per record]) of genome g; deterministic in (seed, g)."""
        rng = np.random.default_rng([self.seed, g])
        n_acc = self.cds - self.C
        acc = self.C + rng.choice(self.F - self.C, size=n_acc, replace=False)
        fams = np.concatenate([np.arange(self.C), acc])
        rng.shuffle(fams)
        tier = rng.random(fams.size)
        seqs = []
        for k, f in enumerate(fams):
            s = self.fam_res[self.fam_off[f]:self.fam_off[f + 1]].copy()
            t = tier[k]
            if t < 0.01:                       # too short for cd-hit
                s = s[:int(rng.integers(3, 11))]
            elif t < 0.06:                     # 15-25 % substituted
                self._mutate(rng, s, rng.uniform(0.15, 0.25))
            elif t < 0.06 + 0.94 * 0.6:        # 2 % substituted
                self._mutate(rng, s, 0.02)
            seqs.append(s.tobytes())
        return fams, seqs

    @staticmethod
    def _mutate(rng, s, frac):
        n = max(1, int(round(frac * s.size)))
        pos = rng.choice(s.size, size=n, replace=False)
        s[pos] = AA20[rng.integers(0, 20, n)]

    def header(self, g, k, fam):
        return 'fig|%d.1.peg.%d|F%d' % (g, k, fam)

    def write_faa(self, directory, wrap=60):
        """One <directory>/genome_<g>.faa per genome; returns the paths."""
        os.makedirs(directory, exist_ok=True)
        paths = []
        for g in range(self.n_genomes):
            fams, seqs = self.genome(g)
            path = os.path.join(directory, 'genome_%04d.faa' % g)
            with open(path, 'w') as f:
                for k, (fam, s) in enumerate(zip(fams, seqs)):
                    s = s.decode('ascii')
                    f.write('>%s   hypothetical protein\n' % self.header(g, k, fam))
                    f.write('\n'.join(s[i:i + wrap] for i in range(0, len(s), wrap)) + '\n')
            paths.append(path)
        return paths

    def nr_arrays(self, progress=None):
        """Exact-deduplicated records in genome order (what consolidate_seqs hands to the
        clustering call): (residues uint8 ASCII, offsets uint64, n_raw)."""
        seen, chunks, lens, n_raw = set(), [], [], 0
        for g in range(self.n_genomes):
            _, seqs = self.genome(g)
            n_raw += len(seqs)
            for s in seqs:
                if s not in seen:
                    seen.add(s)
                    chunks.append(s)
                    lens.append(len(s))
            if progress and (g + 1) % progress == 0:
                print('  synth: genome %d / %d, %d non-redundant' % (g + 1, self.n_genomes, len(lens)), file=sys.stderr, flush=True)
        offsets = np.zeros(len(lens) + 1, dtype=np.uint64)
        np.cumsum(np.asarray(lens, dtype=np.uint64), out=offsets[1:])
        residues = np.frombuffer(b''.join(chunks), dtype=np.uint8)
        return residues, offsets, n_raw


def protein_set(name):
    return ProteinSet(*CONFIGS[name])


def pancore_matrix(n_genes=150000, n_genomes=400, seed=1):
    """Synthetic gene x genome presence matrix of SURVEY §8d: 2,000 genes at p=0.99, the rest
    Beta(0.08, 3); rows that come out empty are dropped and drawing continues until exactly
    `n_genes` non-empty rows exist. Returns COO (row int32, col int32, n_genes)."""
    rng = np.random.default_rng(seed)
    rows, cols, have = [], [], 0
    first = True
    while have < n_genes:
        chunk = 20000
        p = rng.beta(0.08, 3.0, chunk)
        if first:
            p[:min(2000, chunk)] = 0.99
            first = False
        m = rng.random((chunk, n_genomes)) < p[:, None]
        m = m[m.any(axis=1)][:n_genes - have]
        r, c = np.nonzero(m)
        rows.append(r + have)
        cols.append(c)
        have += m.shape[0]
    return (np.concatenate(rows).astype(np.int32), np.concatenate(cols).astype(np.int32), int(n_genes))


def _revcomp(b):
    return b[::-1].translate(bytes.maketrans(b'ACGT', b'TGCA'))


def noncoding_set(n_genomes=400, seed=5, n_trna=60, rrna_lengths=(120, 1500, 2900), copies_rrna=3):
    """Synthetic non-coding feature set of SURVEY §8d (config 5): per genome ~90 features --
    tRNA-like families (74-90 nt) and rRNA-like ones (120 / 1,500 / 2,900 nt) -- each copy
    0-10 % diverged from its ancestor, about half of them given on the reverse strand, a few
    with N. Returns the exact-deduplicated (residues uint8 ASCII, offsets uint64, n_raw)."""
    rng = np.random.default_rng(seed)
    nt = np.frombuffer(b'ACGT', dtype=np.uint8)
    fams = [nt[rng.integers(0, 4, int(rng.integers(74, 91)))] for _ in range(n_trna)]
    fams += [nt[rng.integers(0, 4, L)] for L in rrna_lengths for _ in range(copies_rrna)]
    seen, chunks, lens, n_raw = set(), [], [], 0
    for g in range(n_genomes):
        for f, anc in enumerate(fams):
            copies = 1 if f < n_trna else int(rng.integers(1, 4))
            for _ in range(copies):
                s = anc.copy()
                div = rng.uniform(0.0, 0.10) if rng.random() < 0.7 else 0.0
                k = int(div * s.size)
                if k:
                    pos = rng.choice(s.size, size=k, replace=False)
                    s[pos] = nt[rng.integers(0, 4, k)]
                if rng.random() < 0.02:
                    s[int(rng.integers(0, s.size))] = ord('N')
                b = s.tobytes()
                if rng.random() < 0.5:
                    b = _revcomp(b)
                n_raw += 1
                if b not in seen:
                    seen.add(b)
                    chunks.append(b)
                    lens.append(len(b))
    offsets = np.zeros(len(lens) + 1, dtype=np.uint64)
    np.cumsum(np.asarray(lens, dtype=np.uint64), out=offsets[1:])
    return np.frombuffer(b''.join(chunks), dtype=np.uint8), offsets, n_raw
