"""CPU oracle for the pangenomix hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes loader for oracle/libpgx_oracle.so (built from pancore_ref.c and cluster_ref.c by
oracle/Makefile). Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this package; nothing under pangenomix_amd/ does.

  pan_core(row, col, val, n_genes, n_genomes, perms)    pangenome_analysis.py:72-98 restated;
                                                        pinned by tests/golden/pancore
  cluster_greedy(residues, offsets, params)             cd-hit greedy clustering restated
                                                        (SURVEY.md App. A); PARITY UNPINNED
"""
import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_DIR, 'libpgx_oracle.so')
_lib = None


def build(force=False):
    if force or not os.path.exists(_LIB) or any(
            os.path.getmtime(os.path.join(_DIR, s)) > os.path.getmtime(_LIB)
            for s in ('pancore_ref.c', 'cluster_ref.c', 'Makefile')):
        subprocess.check_call(['make', '-s', '-C', _DIR] + (['-B'] if force else []))
    return _LIB


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB)
        _lib.pgxo_pan_core.restype = C.c_int
        _lib.pgxo_cluster_greedy.restype = C.c_int
        _lib.pgxo_protein_tables.restype = None
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def pan_core(row, col, val, n_genes, n_genomes, perms):
    """int64 (pan, core) tables [n_iter, n_genomes] from COO triples and permutations."""
    row = np.ascontiguousarray(row, dtype=np.int32)
    col = np.ascontiguousarray(col, dtype=np.int32)
    val = None if val is None else np.ascontiguousarray(val, dtype=np.int64)
    perms = np.ascontiguousarray(perms, dtype=np.int32)
    n_iter = perms.shape[0]
    pan = np.zeros((n_iter, n_genomes), dtype=np.int64)
    core = np.zeros((n_iter, n_genomes), dtype=np.int64)
    rc = lib().pgxo_pan_core(_p(row), _p(col), _p(val), C.c_uint64(row.size), C.c_uint32(n_genes),
                             C.c_uint32(n_genomes), _p(perms), C.c_uint32(n_iter), _p(pan), _p(core))
    if rc != 0:
        raise RuntimeError('pgxo_pan_core failed (%d)' % rc)
    return pan, core


def cluster_greedy(residues, offsets, params):
    """Same outputs as pangenomix_amd._native.Context.cluster_greedy. `params` is a
    pangenomix_amd._native.ClusterParams (the struct layout is shared via include/pgx.h)."""
    from pangenomix_amd._native import ClusterStats
    residues = np.ascontiguousarray(residues, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    n = offsets.size - 1
    cluster = np.empty(n, dtype=np.int32)
    member = np.empty(n, dtype=np.int32)
    identity = np.empty(n, dtype=np.float32)
    strand = np.zeros(n, dtype=np.uint8)
    n_clusters = C.c_uint32(0)
    stats = ClusterStats()
    rc = lib().pgxo_cluster_greedy(_p(residues), _p(offsets), C.c_uint32(n), C.byref(params),
                                   _p(cluster), _p(member), _p(identity), _p(strand),
                                   C.byref(n_clusters), C.byref(stats))
    if rc != 0:
        raise RuntimeError('pgxo_cluster_greedy failed (%d)' % rc)
    return cluster, member, identity, strand, int(n_clusters.value), stats.as_dict()


def protein_tables():
    aa2idx = np.zeros(26, dtype=np.int8)
    blosum = np.zeros((23, 23), dtype=np.int8)
    lib().pgxo_protein_tables(_p(aa2idx), _p(blosum))
    return aa2idx, blosum
