"""
LightSparseDataFrame (LSDF): labelled scipy-sparse container and its on-disk
format, the output contract of the pangenome hot path.

Mirrors the reference interface of pangenomix/sparse_utils.py
(`LightSparseDataFrame` :182-364, `read_lsdf` :18-42) for the members that sit
on the hot path: constructor, `.index/.columns/.data/.shape`,
`to_npz`, `read_lsdf`, plus the small dense helpers downstream consumers use.

On-disk contract (pinned by tests/golden/cds/expected/*.npz):
  <f>.npz            scipy.sparse.save_npz of the COO matrix
                     members: row int32, col int32, data int64, shape int64[2],
                     format b'coo'
  <f>.npz.labels.txt index labels then column labels, one per line
"""

import numpy as np
import scipy.sparse


_deflate_pool = None


def _pool():
    global _deflate_pool
    if _deflate_pool is None:
        import os
        from concurrent.futures import ThreadPoolExecutor
        _deflate_pool = ThreadPoolExecutor(max(1, min(16, os.cpu_count() or 1)))
    return _deflate_pool


def _deflate_chunk(args):
    import zlib
    part, last = args
    co = zlib.compressobj(1, zlib.DEFLATED, -15)            # raw deflate, as a zip member holds it
    return co.compress(part) + co.flush(zlib.Z_FINISH if last else zlib.Z_SYNC_FLUSH)


def write_npz_parallel(path, members, chunk=1 << 20):
    """A .npz (zip of deflated .npy members, as numpy.savez_compressed / scipy.sparse.save_npz write) whose members are
    deflated in 1 MB pieces by a pool of threads (zlib releases the GIL): a piece that is not the last ends with a
    sync flush -- a byte-aligned empty block -- so the pieces put end to end are one deflate stream (what pigz does).
    `members`: (name, array) pairs. Members of 4 GiB or more are left to the caller (zip64)."""
    import io
    import struct
    import time
    import zlib
    entries, jobs = [], []
    for name, arr in members:
        arr = np.asarray(arr)
        if not arr.flags.c_contiguous:
            arr = np.ascontiguousarray(arr)                   # (never for a 0-d array, which it would make 1-d)
        head = io.BytesIO()
        np.lib.format.write_array_header_1_0(head, np.lib.format.header_data_from_array_1_0(arr))
        try:
            body = arr.tobytes() if arr.ndim == 0 or arr.nbytes < chunk else memoryview(arr.reshape(-1)).cast('B')
        except (ValueError, TypeError):       # (a dtype the buffer protocol does not export, e.g. 'U': one copy)
            body = arr.tobytes()
        raw = [head.getvalue() + bytes(body)] if len(body) < chunk else None
        if raw is None:     # the header travels with the first piece
            raw = [head.getvalue() + bytes(body[:chunk])] + [body[o:o + chunk] for o in range(chunk, len(body), chunk)]
        usize = sum(len(r) for r in raw)
        if usize >= 0xFFFFFFFF:
            raise OverflowError('member of 4 GiB or more')
        crc = 0
        for r in raw:
            crc = zlib.crc32(r, crc)
        entries.append([name + '.npy', crc, usize, len(jobs), len(raw)])
        jobs.extend((r, i == len(raw) - 1) for i, r in enumerate(raw))
    packed = list(_pool().map(_deflate_chunk, jobs))
    t = time.localtime()
    dos_time = (t.tm_hour << 11) | (t.tm_min << 5) | (t.tm_sec // 2)
    dos_date = (max(t.tm_year, 1980) - 1980) << 9 | (t.tm_mon << 5) | t.tm_mday
    central, offset = [], 0
    with open(path, 'wb') as f:
        for name, crc, usize, j0, nj in entries:
            fname = name.encode('ascii')
            csize = sum(len(packed[j]) for j in range(j0, j0 + nj))
            if csize >= 0xFFFFFFFF:
                raise OverflowError('member of 4 GiB or more')
            fixed = (20, 0, 8, dos_time, dos_date, crc & 0xFFFFFFFF, csize, usize, len(fname), 0)
            f.write(struct.pack('<4s5H3L2H', b'PK\x03\x04', *fixed))
            f.write(fname)
            for j in range(j0, j0 + nj):
                f.write(packed[j])
            central.append(struct.pack('<4s6H3L5H2L', b'PK\x01\x02', (3 << 8) | 20, *fixed, 0, 0, 0, 0o600 << 16, offset) + fname)
            offset += 30 + len(fname) + csize
        cd = b''.join(central)
        f.write(cd)
        f.write(struct.pack('<4s4H2LH', b'PK\x05\x06', 0, 0, len(central), len(central), len(cd), offset, 0))


def read_lsdf(npz_file, label_file=None):
    """Load an LSDF from `<npz_file>` + `<npz_file>.labels.txt`.

    Reference: sparse_utils.py:18-42 (labels are split by the matrix' row count).
    """
    data = scipy.sparse.load_npz(npz_file)
    label_path = npz_file + '.labels.txt' if label_file is None else label_file
    with open(label_path, 'r') as f:
        labels = [line.strip() for line in f]
    n_rows = data.shape[0]
    return LightSparseDataFrame(labels[:n_rows], labels[n_rows:], data)


class LightSparseDataFrame(object):
    """scipy.sparse matrix (kept as COO) with string row / column labels."""

    def __init__(self, index, columns, data):
        # Reference: sparse_utils.py:184-208. A conversion failure is reported
        # and leaves .data = nan there; here it is an error (the
        # build raises where the reference prints and continues).
        self.data = data.tocoo()
        self.index = np.asarray(index)       # (label arrays of a million names are not copied again)
        self.columns = np.asarray(columns)
        self.shape = self.data.shape
        # label -> position (:199-200), built on first use FROM THE CONSTRUCTION-TIME LABELS: the reference fills
        # both dicts in __init__ and never refreshes them, so re-assigning .index / .columns afterwards (as
        # build_noncoding_pangenome does, pangenome.py:292-293) leaves labelslice() resolving the old names
        # (SURVEY App. B.5). The label arrays are kept by reference, not copied.
        self._index0, self._columns0 = index, columns
        self._index_map = self._column_map = None
        if len(index) != self.shape[0]:
            print('ERROR: Index length does not match data')
        if len(columns) != self.shape[1]:
            print('ERROR: Column length does no match data')

    @staticmethod
    def _label_map(labels):
        return {label: i for i, label in enumerate(labels.tolist() if hasattr(labels, 'tolist') else labels)}

    @property
    def index_map(self):
        if self._index_map is None:
            self._index_map = self._label_map(self._index0)
        return self._index_map

    @index_map.setter
    def index_map(self, value):
        self._index_map = value

    @property
    def column_map(self):
        if self._column_map is None:
            self._column_map = self._label_map(self._columns0)
        return self._column_map

    @column_map.setter
    def column_map(self, value):
        self._column_map = value

    # -- output contract ---------------------------------------------------
    def to_npz(self, npz_file, label_file=None):
        """Write `<npz_file>` and its label file (reference :295-314). The .npz holds what
        scipy.sparse.save_npz writes for the COO matrix -- members row, col, format, shape, data with the
        same dtypes, deflated -- at the fastest deflate level (the default level spends 1.7 s on the
        400-genome allele table, this 0.3 s; readers do not see the difference) and in pieces by several threads
        (write_npz_parallel)."""
        import zipfile
        label_path = npz_file + '.labels.txt' if label_file is None else label_file
        with open(label_path, 'w+') as f:
            for labels in (self.index, self.columns):
                is_str = getattr(getattr(labels, 'dtype', None), 'kind', None) == 'U'
                labels = labels.tolist() if hasattr(labels, 'tolist') else list(labels)
                if labels:
                    f.write('\n'.join(labels if is_str else [str(x) for x in labels]) + '\n')
        m = self.data.tocoo()
        members = (('row', m.row), ('col', m.col), ('format', np.array(m.format.encode('ascii'))),
                   ('shape', np.array(m.shape, dtype=np.int64)), ('data', m.data))
        if not npz_file.endswith('.npz'):
            npz_file += '.npz'      # (as numpy.savez does)
        try:
            write_npz_parallel(npz_file, members)
        except OverflowError:       # a member of 4 GiB or more: the zip64 writer of the standard library
            with zipfile.ZipFile(npz_file, 'w', zipfile.ZIP_DEFLATED, compresslevel=1) as z:
                for name, arr in members:
                    with z.open(name + '.npy', 'w', force_zip64=True) as f:
                        np.lib.format.write_array(f, np.asanyarray(arr), allow_pickle=False)

    # -- small helpers used by downstream consumers ------------------------
    def transpose(self):
        return LightSparseDataFrame(self.columns, self.index, self.data.transpose())

    def islice(self, i_indices=None, i_columns=None):
        """Positional row / column selection (reference :238-269)."""
        if i_indices is None and i_columns is None:
            print('No indices or columns selected')
            return None
        new_index, new_columns, new_data = self.index, self.columns, self.data
        if i_columns is not None:
            new_columns = self.columns[i_columns]
            new_data = new_data.tocsc()[:, i_columns]
        if i_indices is not None:
            new_index = self.index[i_indices]
            new_data = new_data.tocsr()[i_indices, :]
        return LightSparseDataFrame(new_index, new_columns, new_data)

    def labelslice(self, indices=None, columns=None):
        i_idx = None if indices is None else [self.index_map[x] for x in indices]
        i_col = None if columns is None else [self.column_map[x] for x in columns]
        return self.islice(i_idx, i_col)

    def sum(self, axis='index'):
        if axis in ('index', 0):
            return np.asarray(self.data.sum(axis=1))[:, 0]
        return np.asarray(self.data.sum(axis=0))[0, :]

    def drop_empty(self, axis='index'):
        if axis in ('index', 0):
            return self.islice(i_indices=np.where(self.sum(0) > 0)[0])
        return self.islice(i_columns=np.where(self.sum(1) > 0)[0])

    @property
    def values(self):
        return self.data.toarray()
