"""ctypes binding of libpgx.so (include/pgx.h). Thin on purpose: argument marshalling
and error translation only. There is no CPU fallback -- if the library or a GPU is
missing, the first compute call raises PgxError."""

import ctypes as C
import importlib.util
import os
import threading

import numpy as np

_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libpgx.so')


class PgxError(RuntimeError):
    pass


class DeviceInfo(C.Structure):
    _fields_ = [('name', C.c_char * 64), ('arch', C.c_char * 32), ('device_id', C.c_int32),
                ('compute_units', C.c_int32), ('wavefront_size', C.c_int32),
                ('lds_bytes_per_block', C.c_int32), ('hbm_bytes', C.c_uint64),
                ('clock_khz', C.c_int32), ('reserved', C.c_int32)]


# int (*exchange)(void *user, void *stream, int slot): enqueue the all-gather of a window's best keys (pgx.h)
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int)
EXCHANGE_KEYS = 65536                  # PGX_EXCHANGE_KEYS
EXCHANGE_WORDS = EXCHANGE_KEYS + 8     # PGX_EXCHANGE_WORDS: the keys + the error word (+ padding)
EXCHANGE_SLOTS = 2                     # PGX_EXCHANGE_SLOTS: windows in flight


class ClusterParams(C.Structure):
    _fields_ = [('alphabet', C.c_int32), ('word_len', C.c_int32), ('band_width', C.c_int32),
                ('min_length', C.c_int32), ('both_strands', C.c_int32), ('batch_size', C.c_int32),
                ('identity', C.c_double), ('aan_cutoff', C.c_double), ('aas_cutoff', C.c_double),
                # record-sharded multi-GPU mode (all zero / NULL = single GPU)
                ('shard_index', C.c_int32), ('shard_count', C.c_int32),
                ('exchange', EXCHANGE_FN), ('exchange_user', C.c_void_p), ('exchange_send', C.c_void_p),
                ('exchange_recv', C.c_void_p),
                # cd-hit's memory-chunked rule: flush positions in the sorted list (NULL = unchunked)
                ('chunk_boundaries', C.POINTER(C.c_uint32)), ('n_chunk_boundaries', C.c_uint32), ('reserved0', C.c_uint32)]


STAT_FIELDS = ('n_input', 'n_clustered', 'n_clusters', 'sum_len_queries', 'sum_len_reps', 'rep_words',
               'posting_visits', 'filter_pairs', 'aligned_pairs', 'aligned_rep_len', 'dp_cells', 'sweeps')


class ClusterStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in STAT_FIELDS] + [('reserved', C.c_uint64 * 4)]

    def as_dict(self):
        d = {n: int(getattr(self, n)) for n in STAT_FIELDS}
        d['gpu'] = {'pairs': int(self.reserved[0]), 'aligned': int(self.reserved[1]),
                    'aligned_bytes': int(self.reserved[2]), 'filter_walk_words': int(self.reserved[3])}
        return d


class FastaInfo(C.Structure):
    _fields_ = [('n_records', C.c_uint64), ('n_missing', C.c_uint64), ('n_groups', C.c_uint64),
                ('n_residue_bytes', C.c_uint64), ('n_header_bytes', C.c_uint64), ('simple', C.c_uint32),
                ('reserved', C.c_uint32), ('why', C.c_char * 256)]


# every symbol include/pgx.h declares: (restype, argtypes)
_P = C.c_void_p
_S = C.c_char_p
SIGNATURES = {
    'pgx_fasta_open': (C.c_int, [C.POINTER(C.c_char_p), C.c_uint32, C.c_int, C.POINTER(_P)]),
    'pgx_fasta_close': (None, [_P]),
    'pgx_fasta_info': (C.c_int, [_P, C.POINTER(FastaInfo)]),
    'pgx_fasta_group_of_record': (_P, [_P]),
    'pgx_fasta_file_of_record': (_P, [_P]),
    'pgx_fasta_rep_of_group': (_P, [_P]),
    'pgx_fasta_residues': (_P, [_P]),
    'pgx_fasta_offsets': (_P, [_P]),
    'pgx_fasta_letters': (_P, [_P]),
    'pgx_fasta_digests': (_P, [_P]),
    'pgx_fasta_header_blob': (_P, [_P]),
    'pgx_fasta_header_offsets': (_P, [_P]),
    'pgx_fasta_write_consolidated': (C.c_int, [_P, _S, _S, _S]),
    'pgx_legacy_shuffles': (C.c_int, [_P, C.POINTER(C.c_int32), C.c_uint32, C.c_uint32, _P]),
    'pgx_pan_core_coo_rng': (C.c_int, [_P, _P, _P, C.c_uint64, C.c_uint32, C.c_uint32, _P, C.POINTER(C.c_int32), C.c_uint32,
                                      _P, _P, _P, C.POINTER(C.c_uint64)]),
    'pgx_pan_core_table': (C.c_int, [_P, _P, _P, _P, C.c_uint64, C.c_uint32, C.c_uint32, _P, C.POINTER(C.c_int32), C.c_uint32,
                                    _P, _P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    'pgx_bitmap_from_clusters': (C.c_int, [_P, _P, C.c_uint64, _P, _P, C.c_uint64, _P, C.c_uint32, C.c_uint32, C.c_uint32,
                                          C.POINTER(C.c_uint64)]),
    'pgx_bitmap_resident_read': (C.c_int, [_P, C.c_uint64, _P]),
    'pgx_pan_core_table_resident': (C.c_int, [_P, C.c_uint64, C.c_uint32, C.c_uint32, _P, C.POINTER(C.c_int32), C.c_uint32, _P, _P]),
    'pgx_allele_order': (C.c_int, [_P, _P, C.c_uint64, _P]),
    'pgx_fasta_feature_coo': (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), _P, _P,
                                       C.POINTER(C.c_uint64), _P, _P, C.POINTER(C.c_uint64), _P, C.POINTER(C.c_uint64)]),
    'pgx_first_insertions': (C.c_int, [_P, _P, C.c_uint64, C.c_uint64, _P, C.POINTER(C.c_uint64)]),
    'pgx_format_labels': (C.c_int, [_S, _S, _P, _P, C.c_uint64, C.c_uint32, _P]),
    'pgx_format_labels_ucs4': (C.c_int, [_S, _S, _P, _P, C.c_uint64, C.c_uint32, _P]),
    'pgx_fasta_write_clustered': (C.c_int, [_P, _P, _P, _P, _P, C.c_int, _S, _S, _S, _S, _S]),
    'pgx_version': (C.c_int, []),
    'pgx_last_error': (C.c_char_p, []),
    'pgx_ctx_create': (C.c_int, [C.c_int, C.POINTER(_P)]),
    'pgx_ctx_create_on': (C.c_int, [C.POINTER(C.c_int), C.c_int, C.POINTER(_P)]),
    'pgx_ctx_destroy': (None, [_P]),
    'pgx_device_info': (C.c_int, [_P, C.POINTER(DeviceInfo)]),
    'pgx_profile_enable': (C.c_int, [_P, C.c_int]),
    'pgx_profile_reset': (C.c_int, [_P]),
    'pgx_profile_count': (C.c_int, [_P]),
    'pgx_profile_read': (C.c_int, [_P, C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_double),
                                   C.POINTER(C.c_uint64)]),
    'pgx_bitmap_stride_words': (C.c_uint32, [C.c_uint32]),
    'pgx_presence_bitmap': (C.c_int, [_P, _P, _P, C.c_uint64, C.c_uint32, C.c_uint32, _P, C.POINTER(C.c_uint64)]),
    'pgx_presence_bitmap_dev': (C.c_int, [_P, _P, _P, C.c_uint64, C.c_uint32, C.c_uint32, _P, _P, _P]),
    'pgx_pan_core_coo': (C.c_int, [_P, _P, _P, C.c_uint64, C.c_uint32, C.c_uint32, _P, C.c_uint32, _P, _P,
                                   C.POINTER(C.c_uint64)]),
    'pgx_pan_core': (C.c_int, [_P, _P, C.c_uint32, C.c_uint32, _P, C.c_uint32, _P, _P]),
    'pgx_row_counts': (C.c_int, [_P, _P, _P, C.c_uint64, C.c_uint32, C.c_uint32, _P, C.POINTER(C.c_uint64)]),
    'pgx_row_counts_dev': (C.c_int, [_P, _P, C.c_uint32, C.c_uint32, _P, _P]),
    'pgx_heaps_fit': (C.c_int, [_P, _P, C.c_uint32, C.c_uint32, _P, _P]),
    'pgx_heaps_fit_dev': (C.c_int, [_P, _P, C.c_uint32, C.c_uint32, _P, _P, _P]),
    'pgx_pan_core_workspace_bytes': (C.c_size_t, [C.c_uint32, C.c_uint32, C.c_uint32]),
    'pgx_pan_core_dev': (C.c_int, [_P, _P, C.c_uint32, C.c_uint32, _P, C.c_uint32, _P, _P, _P,
                                   C.c_size_t, _P]),
    'pgx_cluster_greedy': (C.c_int, [_P, _P, _P, C.c_uint32, C.POINTER(ClusterParams), _P, _P, _P, _P,
                                     C.POINTER(C.c_uint32), C.POINTER(ClusterStats)]),
    'pgx_cluster_window_cap': (C.c_uint32, [C.POINTER(ClusterParams)]),
    'pgx_rccl_load': (C.c_int, [_S]),
    'pgx_rccl_unique_id': (C.c_int, [_P]),
    'pgx_rccl_comm_create': (C.c_int, [_P, _P, C.c_int, C.c_int]),
    'pgx_rccl_comm_destroy': (C.c_int, [_P]),
    'pgx_cluster_greedy_dev': (C.c_int, [_P, _P, _P, C.c_uint32, C.c_uint64, C.POINTER(ClusterParams), _P, _P,
                                         _P, _P, C.POINTER(C.c_uint32), C.POINTER(ClusterStats), _P]),
}

_lib = None
_lock = threading.Lock()


def _one_hip_runtime():
    """One HIP runtime per process, whatever the import order. PyTorch-ROCm wheels bundle their own
    libamdhip64 under the SAME soname (libamdhip64.so.7) libpgx links to, and the dynamic loader
    binds every later user of that soname to whichever copy was mapped first. With the system copy
    first, a later `import torch` would run on a runtime it was not built for (seen to fail with
    "No HIP GPUs are available"). So: if no libamdhip64 is mapped yet and a torch installation
    exists, its copy is mapped first -- by path, WITHOUT importing torch -- and libpgx, and torch
    whenever it comes, share it. Without torch the system runtime (libpgx's RUNPATH) serves."""
    try:
        with open('/proc/self/maps') as f:
            if 'libamdhip64' in f.read():
                return
    except OSError:
        pass
    if os.environ.get('PGX_HIP_RUNTIME') == 'system':
        return
    try:
        spec = importlib.util.find_spec('torch')     # locates the package, does not import it
    except (ImportError, ValueError):
        spec = None
    if spec is not None and spec.origin:
        cand = os.path.join(os.path.dirname(spec.origin), 'lib', 'libamdhip64.so')
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)


def lib():
    """Load libpgx.so once; raises PgxError if it has not been built."""
    global _lib
    with _lock:
        if _lib is None:
            _one_hip_runtime()
            if not os.path.exists(_LIB_PATH):
                raise PgxError('%s not found: build it with `python -c "import __graft_entry__ as g; '
                               'g.build()"` or `make -C pangenomix_amd/csrc` (there is no CPU fallback)'
                               % _LIB_PATH)
            handle = C.CDLL(_LIB_PATH)
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(handle, name)
                fn.restype, fn.argtypes = res, args
            _lib = handle
    return _lib


def check(rc):
    if rc != 0:
        err = PgxError('libpgx error %d: %s' % (rc, lib().pgx_last_error().decode('utf-8', 'replace')))
        err.status = rc
        raise err


ERR_NOMEM = -4   # PGX_ERR_NOMEM


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def rccl_path():
    """The librccl.so this process should share: PyTorch-ROCm's bundled copy when there is one (found by path, without
    importing torch), else whatever the loader finds. PGX_RCCL_LIB overrides."""
    env = os.environ.get('PGX_RCCL_LIB')
    if env:
        return env
    import importlib.util
    spec = importlib.util.find_spec('torch')
    if spec and spec.origin:
        cand = os.path.join(os.path.dirname(spec.origin), 'lib', 'librccl.so')
        if os.path.exists(cand):
            return cand
    return 'librccl.so'


def rccl_load(path=None):
    check(lib().pgx_rccl_load((path or rccl_path()).encode()))


def rccl_unique_id():
    rccl_load()
    buf = (C.c_uint8 * 128)()
    check(lib().pgx_rccl_unique_id(C.cast(buf, _P)))
    return bytes(buf)


class Context(object):
    """Owns a pgx_ctx (device state). One per thread; not re-entrant."""

    def __init__(self, device_id=0):
        self._h = C.c_void_p()
        check(lib().pgx_ctx_create(int(device_id), C.byref(self._h)))

    def close(self):
        if self._h:
            lib().pgx_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def handle(self):
        return self._h

    def device_info(self):
        info = DeviceInfo()
        check(lib().pgx_device_info(self._h, C.byref(info)))
        return {'name': info.name.decode(), 'arch': info.arch.decode(), 'device_id': info.device_id,
                'compute_units': info.compute_units, 'wavefront_size': info.wavefront_size,
                'lds_bytes_per_block': info.lds_bytes_per_block, 'hbm_bytes': info.hbm_bytes,
                'clock_khz': info.clock_khz}

    # -- the library's own RCCL communicator (record-sharded mode without a callback) ----------
    def comm_create(self, unique_id, rank, world):
        """Collective: every process calls it with the same 128-byte id (rccl_unique_id() of one of them)."""
        uid = bytes(unique_id)
        if len(uid) != 128:
            raise ValueError('the RCCL unique id is 128 bytes')
        rccl_load()
        check(lib().pgx_rccl_comm_create(self._h, C.cast(C.c_char_p(uid), _P), int(rank), int(world)))
        self.comm = (int(rank), int(world))

    def comm_destroy(self):
        check(lib().pgx_rccl_comm_destroy(self._h))
        self.comm = None

    comm = None

    # -- per-kernel timing ---------------------------------------------------
    def profile(self, on=True):
        check(lib().pgx_profile_enable(self._h, 1 if on else 0))

    def profile_reset(self):
        check(lib().pgx_profile_reset(self._h))

    def profile_read(self):
        """{kernel name: (total_ms, launches)}; waits for pending events."""
        out = {}
        for slot in range(lib().pgx_profile_count(self._h)):
            name = C.create_string_buffer(64)
            ms, n = C.c_double(0), C.c_uint64(0)
            check(lib().pgx_profile_read(self._h, slot, name, 64, C.byref(ms), C.byref(n)))
            out[name.value.decode()] = (ms.value, int(n.value))
        return out

    # -- device-resident variants (pointers are raw device addresses, e.g. torch data_ptr) --
    def presence_bitmap_dev(self, d_rows, d_genomes, n_records, n_rows, n_genomes, d_bits, stream=0, d_counters=None):
        check(lib().pgx_presence_bitmap_dev(self._h, d_rows, d_genomes, int(n_records), int(n_rows),
                                            int(n_genomes), d_bits, d_counters, stream))

    def pan_core_dev(self, d_bits, n_genes, n_genomes, d_perms, n_iter, d_pan, d_core, d_ws, ws_bytes, stream=0):
        check(lib().pgx_pan_core_dev(self._h, d_bits, int(n_genes), int(n_genomes), d_perms, int(n_iter),
                                     d_pan, d_core, d_ws, int(ws_bytes), stream))

    # -- K3 ----------------------------------------------------------------
    def presence_bitmap(self, rows, genomes, n_rows, n_genomes, return_duplicates=False):
        rows = np.ascontiguousarray(rows, dtype=np.int32)
        genomes = np.ascontiguousarray(genomes, dtype=np.int32)
        if rows.shape != genomes.shape or rows.ndim != 1:
            raise ValueError('rows and genomes must be 1-D arrays of equal length')
        stride = lib().pgx_bitmap_stride_words(int(n_rows))
        bits = np.empty((int(n_genomes), stride), dtype=np.uint64)
        dup = C.c_uint64(0)
        check(lib().pgx_presence_bitmap(self._h, _ptr(rows), _ptr(genomes), rows.size,
                                        int(n_rows), int(n_genomes), _ptr(bits), C.byref(dup)))
        return (bits, int(dup.value)) if return_duplicates else bits

    def heaps_fit(self, pan):
        """(alpha, kappa) float64 per row of the pan table [n_iter, n_genomes]."""
        pan = np.ascontiguousarray(pan, dtype=np.float64)
        if pan.ndim != 2:
            raise ValueError('pan must be [n_iter, n_genomes]')
        alpha = np.empty(pan.shape[0], dtype=np.float64)
        kappa = np.empty(pan.shape[0], dtype=np.float64)
        check(lib().pgx_heaps_fit(self._h, _ptr(pan), pan.shape[0], pan.shape[1], _ptr(alpha), _ptr(kappa)))
        return alpha, kappa

    def row_counts(self, rows, genomes, n_rows, n_genomes):
        """(counts int32[n_rows], duplicates): genomes per row, from the device bitmap."""
        rows = np.ascontiguousarray(rows, dtype=np.int32)
        genomes = np.ascontiguousarray(genomes, dtype=np.int32)
        if rows.shape != genomes.shape or rows.ndim != 1:
            raise ValueError('rows and genomes must be 1-D arrays of equal length')
        counts = np.empty(int(n_rows), dtype=np.int32)
        dup = C.c_uint64(0)
        check(lib().pgx_row_counts(self._h, _ptr(rows), _ptr(genomes), rows.size, int(n_rows), int(n_genomes),
                                   _ptr(counts), C.byref(dup)))
        return counts, int(dup.value)

    def pan_core_coo(self, rows, genomes, n_genes, n_genomes, perms):
        """(pan, core, duplicates): bitmap built and consumed on the device in one call."""
        rows = np.ascontiguousarray(rows, dtype=np.int32)
        genomes = np.ascontiguousarray(genomes, dtype=np.int32)
        perms = np.ascontiguousarray(perms, dtype=np.int32)
        if rows.shape != genomes.shape or rows.ndim != 1:
            raise ValueError('rows and genomes must be 1-D arrays of equal length')
        n_iter = perms.shape[0]
        if perms.ndim != 2 or perms.shape[1] != int(n_genomes):
            raise ValueError('perms must be [n_iter, n_genomes]')
        pan = np.empty((n_iter, int(n_genomes)), dtype=np.int32)
        core = np.empty((n_iter, int(n_genomes)), dtype=np.int32)
        dup = C.c_uint64(0)
        check(lib().pgx_pan_core_coo(self._h, _ptr(rows), _ptr(genomes), rows.size, int(n_genes), int(n_genomes),
                                     _ptr(perms), n_iter, _ptr(pan), _ptr(core), C.byref(dup)))
        return pan, core, int(dup.value)

    def pan_core_coo_rng(self, rows, genomes, n_genes, n_genomes, n_iter, mt_key, mt_pos):
        """(pan, core, duplicates, perms, new_pos): as pan_core_coo, with the permutations drawn by the library from the
        legacy generator's state (`mt_key` uint32[624], advanced in place; `mt_pos`) beside the upload."""
        rows = np.ascontiguousarray(rows, dtype=np.int32)
        genomes = np.ascontiguousarray(genomes, dtype=np.int32)
        if rows.shape != genomes.shape or rows.ndim != 1:
            raise ValueError('rows and genomes must be 1-D arrays of equal length')
        perms = np.empty((int(n_iter), int(n_genomes)), dtype=np.int32)
        pan = np.empty((int(n_iter), int(n_genomes)), dtype=np.int32)
        core = np.empty((int(n_iter), int(n_genomes)), dtype=np.int32)
        dup, pos = C.c_uint64(0), C.c_int32(int(mt_pos))
        check(lib().pgx_pan_core_coo_rng(self._h, _ptr(rows), _ptr(genomes), rows.size, int(n_genes), int(n_genomes),
                                         _ptr(mt_key), C.byref(pos), int(n_iter), _ptr(perms), _ptr(pan), _ptr(core),
                                         C.byref(dup)))
        return pan, core, int(dup.value), perms, int(pos.value)

    def pan_core_table(self, rows, genomes, values, n_genes, n_genomes, n_iter, mt_key, mt_pos):
        """(table float64 [n_iter, 2 n_genomes], duplicates, values that are not 1, perms, new_pos): the whole of
        estimate_pan_core_size() in one library call (pgx.h: pgx_pan_core_table). `values`: the table's stored
        values as int64, or None."""
        rows = np.ascontiguousarray(rows, dtype=np.int32)
        genomes = np.ascontiguousarray(genomes, dtype=np.int32)
        if rows.shape != genomes.shape or rows.ndim != 1:
            raise ValueError('rows and genomes must be 1-D arrays of equal length')
        if values is not None:
            values = np.ascontiguousarray(values, dtype=np.int64)
            if values.shape != rows.shape:
                raise ValueError('values must match the coordinates')
        perms = np.empty((int(n_iter), int(n_genomes)), dtype=np.int32)
        table = np.empty((int(n_iter), 2 * int(n_genomes)), dtype=np.float64)
        dup, bad, pos = C.c_uint64(0), C.c_uint64(0), C.c_int32(int(mt_pos))
        check(lib().pgx_pan_core_table(self._h, _ptr(rows), _ptr(genomes), _ptr(values), rows.size, int(n_genes),
                                       int(n_genomes), _ptr(mt_key), C.byref(pos), int(n_iter), _ptr(perms), _ptr(table),
                                       C.byref(dup), C.byref(bad)))
        return table, int(dup.value), int(bad.value), perms, int(pos.value)

    # -- device-resident hand-off: clustering result -> bitmap kept in the context -> pan/core curves --------------
    def bitmap_from_clusters(self, cluster_of_group, group_of_record, file_of_record, genome_of_file, n_genes, n_genomes):
        """Build the gene x genome bitmap on the device from a clustering result and keep it there; returns its token
        (pgx.h: pgx_bitmap_from_clusters)."""
        cl = np.ascontiguousarray(cluster_of_group, dtype=np.int32)
        grp = np.ascontiguousarray(group_of_record, dtype=np.int32)
        fil = np.ascontiguousarray(file_of_record, dtype=np.uint32)
        gof = np.ascontiguousarray(genome_of_file, dtype=np.int32)
        if grp.shape != fil.shape:
            raise ValueError('group_of_record and file_of_record must have one entry per record')
        token = C.c_uint64(0)
        check(lib().pgx_bitmap_from_clusters(self._h, _ptr(cl), cl.size, _ptr(grp), _ptr(fil), grp.size, _ptr(gof), gof.size,
                                             int(n_genes), int(n_genomes), C.byref(token)))
        return int(token.value)

    def bitmap_resident_read(self, token, n_genes, n_genomes):
        bits = np.empty((int(n_genomes), lib().pgx_bitmap_stride_words(int(n_genes))), dtype=np.uint64)
        check(lib().pgx_bitmap_resident_read(self._h, int(token), _ptr(bits)))
        return bits

    def pan_core_table_resident(self, token, n_genes, n_genomes, n_iter, mt_key, mt_pos):
        """(table float64 [n_iter, 2 n_genomes], perms, new_pos) from the bitmap that is resident under `token`."""
        perms = np.empty((int(n_iter), int(n_genomes)), dtype=np.int32)
        table = np.empty((int(n_iter), 2 * int(n_genomes)), dtype=np.float64)
        pos = C.c_int32(int(mt_pos))
        check(lib().pgx_pan_core_table_resident(self._h, int(token), int(n_genes), int(n_genomes), _ptr(mt_key), C.byref(pos),
                                                int(n_iter), _ptr(perms), _ptr(table)))
        return table, perms, int(pos.value)

    def pan_core(self, bits, n_genes, perms):
        perms = np.ascontiguousarray(perms, dtype=np.int32)
        n_iter, n_genomes = perms.shape
        bits = np.ascontiguousarray(bits, dtype=np.uint64)
        stride = lib().pgx_bitmap_stride_words(int(n_genes))
        if bits.shape != (n_genomes, stride):
            raise ValueError('bitmap shape %r does not match (%d, %d)' % (bits.shape, n_genomes, stride))
        pan = np.empty((n_iter, n_genomes), dtype=np.int32)
        core = np.empty((n_iter, n_genomes), dtype=np.int32)
        check(lib().pgx_pan_core(self._h, _ptr(bits), int(n_genes), int(n_genomes), _ptr(perms),
                                 int(n_iter), _ptr(pan), _ptr(core)))
        return pan, core

    # -- K1/K2 ---------------------------------------------------------------
    def cluster_greedy(self, residues, offsets, params, want_stats=True):
        """want_stats=False: no work counters (stats is None in the result) and the library leaves out the look-ups
        that only the counters need (pgx.h); the clustering is the same."""
        residues = np.ascontiguousarray(residues, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = offsets.size - 1
        out_cluster = np.empty(n, dtype=np.int32)
        out_member = np.empty(n, dtype=np.int32)
        out_identity = np.empty(n, dtype=np.float32)
        out_strand = np.zeros(n, dtype=np.uint8)
        n_clusters = C.c_uint32(0)
        stats = ClusterStats()
        check(lib().pgx_cluster_greedy(self._h, _ptr(residues), _ptr(offsets), n, C.byref(params),
                                       _ptr(out_cluster), _ptr(out_member), _ptr(out_identity),
                                       _ptr(out_strand), C.byref(n_clusters), C.byref(stats) if want_stats else None))
        return out_cluster, out_member, out_identity, out_strand, int(n_clusters.value), stats.as_dict() if want_stats else None


    def cluster_greedy_dev(self, d_residues, d_offsets, n, total_bytes, params, stream=0, want_stats=True):
        """Sequences resident in HBM (raw device addresses); outputs are host arrays."""
        out_cluster = np.empty(n, dtype=np.int32)
        out_member = np.empty(n, dtype=np.int32)
        out_identity = np.empty(n, dtype=np.float32)
        out_strand = np.zeros(n, dtype=np.uint8)
        n_clusters = C.c_uint32(0)
        stats = ClusterStats()
        check(lib().pgx_cluster_greedy_dev(self._h, d_residues, d_offsets, int(n), int(total_bytes),
                                           C.byref(params), _ptr(out_cluster), _ptr(out_member),
                                           _ptr(out_identity), _ptr(out_strand), C.byref(n_clusters),
                                           C.byref(stats) if want_stats else None, stream))
        return out_cluster, out_member, out_identity, out_strand, int(n_clusters.value), stats.as_dict() if want_stats else None


class FastaSet(object):
    """Genome FASTA files parsed, hashed and de-duplicated by libpgx's host side (csrc/ingest.cpp):
    what consolidate_seqs() (reference pangenome.py:336-405) computes, as arrays. `simple` is False
    when the files hold something the reference's line-by-line semantics treat specially (see
    pgx.h); the arrays are then unavailable and the caller takes the Python path."""

    def __init__(self, paths, threads=0):
        self._h = C.c_void_p()
        arr = (C.c_char_p * len(paths))(*[os.fsencode(p) for p in paths])
        check(lib().pgx_fasta_open(arr, len(paths), int(threads), C.byref(self._h)))
        info = FastaInfo()
        check(lib().pgx_fasta_info(self._h, C.byref(info)))
        self.n_records, self.n_missing, self.n_groups = int(info.n_records), int(info.n_missing), int(info.n_groups)
        self.simple, self.why = bool(info.simple), info.why.decode('utf-8', 'replace')
        self._n_res, self._n_hdr = int(info.n_residue_bytes), int(info.n_header_bytes)

    def close(self):
        h, self._h = self._h, C.c_void_p()      # (taken first: a second caller, e.g. __del__ on another thread, finds nothing)
        if h:
            lib().pgx_fasta_close(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _view(self, fn, dtype, n):
        """numpy view (no copy) of one of the library's arrays; valid until close()."""
        ptr = getattr(lib(), fn)(self._h)
        if not ptr or n == 0:
            return np.zeros(0, dtype=dtype)
        buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
        return np.frombuffer(buf, dtype=dtype, count=n)

    group_of_record = property(lambda self: self._view('pgx_fasta_group_of_record', np.int32, self.n_records))
    file_of_record = property(lambda self: self._view('pgx_fasta_file_of_record', np.uint32, self.n_records))
    rep_of_group = property(lambda self: self._view('pgx_fasta_rep_of_group', np.uint64, self.n_groups))
    residues = property(lambda self: self._view('pgx_fasta_residues', np.uint8, self._n_res))
    offsets = property(lambda self: self._view('pgx_fasta_offsets', np.uint64, self.n_groups + 1))
    letters = property(lambda self: self._view('pgx_fasta_letters', np.uint32, self.n_groups))
    digests = property(lambda self: self._view('pgx_fasta_digests', np.uint8, self.n_groups * 32))
    header_offsets = property(lambda self: self._view('pgx_fasta_header_offsets', np.uint64, self.n_records + 1))

    def headers(self, records=None):
        """Header strings of all records (or of the given record indices)."""
        blob = bytes(self._view('pgx_fasta_header_blob', np.uint8, self._n_hdr))
        off = self.header_offsets
        idx = range(self.n_records) if records is None else records
        return [blob[off[i]:off[i + 1]].decode('ascii') for i in idx]

    def write_consolidated(self, nr_path, shared_path, missing_path=None):
        check(lib().pgx_fasta_write_consolidated(self._h, os.fsencode(nr_path) if nr_path else None, os.fsencode(shared_path),
                                                 os.fsencode(missing_path) if missing_path else None))

    def feature_coo(self, cluster, member, file_order, genome_of_file):
        """Coordinates of the allele and the gene table (pgx_fasta_feature_coo). Returns a dict: allele_groups,
        gene_of_allele, n_genes, a_row, a_col, g_row, g_col, lost_records."""
        cluster = np.ascontiguousarray(cluster, dtype=np.int32)
        member = np.ascontiguousarray(member, dtype=np.int32)
        file_order = np.ascontiguousarray(file_order, dtype=np.int32)
        genome_of_file = np.ascontiguousarray(genome_of_file, dtype=np.int32)
        if not (cluster.size == member.size == self.n_groups) or file_order.size != genome_of_file.size:
            raise ValueError('one entry per non-redundant sequence / per file expected')
        groups = np.empty(self.n_groups, dtype=np.int64)
        gene_of = np.empty(self.n_groups, dtype=np.int32)
        coo = [np.empty(self.n_records, dtype=np.int32) for _ in range(4)]
        lost = np.empty(self.n_records, dtype=np.int64)
        n = [C.c_uint64(0) for _ in range(5)]    # alleles, genes, allele triples, gene triples, lost records
        check(lib().pgx_fasta_feature_coo(self._h, _ptr(cluster), _ptr(member), _ptr(file_order), _ptr(genome_of_file),
                                          _ptr(groups), _ptr(gene_of), C.byref(n[0]), C.byref(n[1]), _ptr(coo[0]), _ptr(coo[1]),
                                          C.byref(n[2]), _ptr(coo[2]), _ptr(coo[3]), C.byref(n[3]), _ptr(lost), C.byref(n[4])))
        na, ng, ta, tg, nl = (int(x.value) for x in n)
        return {'allele_groups': groups[:na], 'gene_of_allele': gene_of[:na], 'n_genes': ng, 'a_row': coo[0][:ta],
                'a_col': coo[1][:ta], 'g_row': coo[2][:tg], 'g_col': coo[3][:tg], 'lost_records': lost[:nl]}

    def write_clustered(self, cluster, member, identity, strand, nucleotide, prefix, variant,
                        clstr_path=None, names_path=None, nr_out_path=None):
        cluster = np.ascontiguousarray(cluster, dtype=np.int32)
        member = np.ascontiguousarray(member, dtype=np.int32)
        identity = np.ascontiguousarray(identity, dtype=np.float32)
        strand = None if strand is None else np.ascontiguousarray(strand, dtype=np.uint8)
        if not (cluster.size == member.size == identity.size == self.n_groups):
            raise ValueError('one entry per non-redundant sequence expected')
        enc = lambda x: os.fsencode(x) if x else None   # noqa: E731
        check(lib().pgx_fasta_write_clustered(self._h, _ptr(cluster), _ptr(member), _ptr(identity), _ptr(strand),
                                              1 if nucleotide else 0, prefix.encode(), variant.encode(),
                                              enc(clstr_path), enc(names_path), enc(nr_out_path)))


def format_labels(prefix, cluster, member=None, variant=None):
    """numpy unicode array of feature names <prefix><cluster>[<variant><member>] (reference
    pangenome.py:1944-1969), formatted by the library."""
    cluster = np.ascontiguousarray(cluster, dtype=np.int32)
    if variant is not None:
        member = np.ascontiguousarray(member, dtype=np.int32)
    digits = lambda a: len(str(int(a.max()))) if a.size else 1   # noqa: E731
    if prefix.isascii() and (variant is None or variant.isascii()) and (cluster.size == 0 or int(cluster.min()) >= 0) \
            and (variant is None or member.size == 0 or int(member.min()) >= 0):
        # straight into numpy's own 'U' layout (UCS-4), exact width, several threads
        width = len(prefix) + digits(cluster) + (len(variant) + digits(member) if variant is not None else 0)
        if width <= 64:
            out = np.zeros(cluster.size, dtype='U%d' % width)
            check(lib().pgx_format_labels_ucs4(prefix.encode(), variant.encode() if variant is not None else None,
                                               _ptr(cluster), _ptr(member) if variant is not None else None,
                                               cluster.size, width, _ptr(out)))
            return out
    width = len(prefix.encode()) + 11 + (len(variant) + 11 if variant is not None else 0)
    out = np.zeros(cluster.size, dtype='S%d' % width)
    check(lib().pgx_format_labels(prefix.encode(), variant.encode() if variant is not None else None, _ptr(cluster),
                                  _ptr(member) if variant is not None else None, cluster.size, width, _ptr(out)))
    return np.char.decode(out, 'utf-8')


def allele_order(cluster, member):
    """Positions of the (cluster, member) pairs in the order of their allele names sorted as strings (stable)."""
    cluster = np.ascontiguousarray(cluster, dtype=np.int32)
    member = np.ascontiguousarray(member, dtype=np.int32)
    if cluster.shape != member.shape or cluster.ndim != 1:
        raise ValueError('cluster and member must be 1-D arrays of one length')
    out = np.empty(cluster.size, dtype=np.int64)
    check(lib().pgx_allele_order(_ptr(cluster), _ptr(member), cluster.size, _ptr(out)))
    return out


def first_insertions(rows, cols, n_cols):
    """Ascending positions i at which the pair (rows[i], cols[i]) occurs for the first time."""
    rows = np.ascontiguousarray(rows, dtype=np.int64)
    cols = np.ascontiguousarray(cols, dtype=np.int64)
    if rows.shape != cols.shape or rows.ndim != 1:
        raise ValueError('rows and cols must be 1-D arrays of one length')
    out = np.empty(rows.size, dtype=np.int64)
    m = C.c_uint64(0)
    check(lib().pgx_first_insertions(_ptr(rows), _ptr(cols), rows.size, int(max(n_cols, 1)), _ptr(out), C.byref(m)))
    return out[:m.value]


_default_ctx = None


def default_context():
    """Process-wide context on device LOCAL_RANK (one process per GPU) or 0."""
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(int(os.environ.get('LOCAL_RANK', '0')))
    return _default_ctx
