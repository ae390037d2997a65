"""TEST INFRASTRUCTURE: a host-level model of the record-sharded clustering protocol (SURVEY 8e;
pangenomix_amd/csrc/cluster.hip, "Record-sharded mode"), small enough to run on CPU ranks under gloo.

What it shares with the product: the partition rule (cluster.owner_of), the exchange
(cluster.group_all_gather over a real torch.distributed group) and the fold of the gathered keys
(cluster.fold_best_keys). What stands in for the HIP kernels: the per-pair evaluator is the CPU
oracle (a pair is accepted iff the oracle clusters the two sequences together), word counts come
from Python sets. The window protocol itself -- phase A, discovery of certain representatives,
in-order block resolution, every rank evaluating only its own members -- is restated step by step,
so that running it at world size 2 and 3 and comparing with the oracle's sequential result checks
that the partition is sound: no step needs anything but the all-gathered best keys.
"""
import numpy as np

NONE = np.uint64(2 ** 64 - 1)
AA = 'ARNDCQEGHILKMFPSTWYV'
_IDX = {c: i for i, c in enumerate(AA)}


def words_of(seq, k=5):
    """{code: multiplicity} of the distinct k-mers (base 21, first residue most significant)."""
    out = {}
    for i in range(len(seq) - k + 1):
        code = 0
        for ch in seq[i:i + k]:
            code = code * 21 + _IDX.get(ch, 20)
        out[code] = out.get(code, 0) + 1
    return out


class Model(object):
    def __init__(self, seqs, params, oracle, pack):
        self.oracle, self.pack, self.params = oracle, pack, params
        order = sorted(range(len(seqs)), key=lambda i: -len(seqs[i]))            # stable: ties in input order
        self.order = [i for i in order if len(seqs[i]) > params.min_length]
        self.seqs = [seqs[i] for i in self.order]
        self.words = [words_of(s, params.word_len) for s in self.seqs]
        self.thr = [max(int(params.aan_cutoff * len(s)), 1) for s in self.seqs]
        self.evaluated = 0

    def count(self, q, r):
        """(shared word count, smallest shared code) of sorted sequences q and r."""
        wq, wr = self.words[q], self.words[r]
        shared = [c for c in wq if c in wr]
        return sum(min(wq[c], wr[c]) for c in shared), (min(shared) if shared else None)

    def accepted(self, q, r):
        """The oracle's verdict on the pair (filter, diagonal test and alignment)."""
        self.evaluated += 1
        res, off = self.pack([self.seqs[r], self.seqs[q]])
        return self.oracle.cluster_greedy(res, off, self.params)[0][1] == 0

    def key(self, q, r):
        cnt, minc = self.count(q, r)
        return None if cnt < self.thr[q] else np.uint64((minc << 32) | r)


def run(seqs, params, oracle, pack, rank, world, all_gather, fold, owner_of, window=16, keys_cap=65536):
    """Cluster `seqs` as process `rank` of `world`. all_gather(recv, send, stream) / fold(rows) /
    owner_of(member, world) are the product's helpers. Returns cluster numbers in input order
    (-1 = discarded) and the number of pairs THIS rank evaluated."""
    import torch
    M = Model(seqs, params, oracle, pack)
    n = len(M.seqs)
    reps, cluster_of = [], {}
    send = torch.empty(keys_cap, dtype=torch.int64)
    recv = torch.empty((world, keys_cap), dtype=torch.int64)

    for b0 in range(0, n, window):
        members = list(range(b0, min(n, b0 + window)))
        nb = len(members)
        best = np.full(nb, NONE, dtype=np.uint64)
        is_rep = np.zeros(nb, dtype=bool)
        mine = [ql for ql in range(nb) if owner_of(ql, world) == rank]

        def exchange():
            send.numpy().view(np.uint64)[:nb] = best
            send.numpy().view(np.uint64)[nb:] = NONE
            all_gather(recv, send, 0)
            best[:] = fold(recv.numpy())[:nb]

        def compare_with(new_reps):                      # own members against representatives r < q
            for ql in mine:
                q = b0 + ql
                if is_rep[ql]:
                    continue
                cands = sorted(k for k in (M.key(q, r) for r in new_reps if r < q) if k is not None)
                for k in cands:
                    if k > best[ql]:
                        break                            # cannot beat the current winner any more
                    if M.accepted(q, int(k) & 0xFFFFFFFF):
                        best[ql] = k
                        break
            exchange()

        compare_with(reps)                               # phase A
        rounds = 0
        while True:
            open_ = [ql for ql in range(nb) if best[ql] == NONE and not is_rep[ql]]
            if not open_:
                break
            if rounds < 2:                               # discovery: members no earlier open member can claim
                rounds += 1
                seen, new = {}, []
                for ql in open_:                         # in order: words of the earlier open members
                    q = b0 + ql
                    if sum(m for c, m in M.words[q].items() if c in seen) < M.thr[q]:
                        new.append(q)
                    for c in M.words[q]:
                        seen.setdefault(c, ql)
            else:                                        # block: exact, in order, replicated on every rank
                status, new = {}, []
                for ql in open_:
                    q = b0 + ql
                    cands = sorted(k for k in (M.key(q, b0 + e) for e in open_ if e < ql and status[e]) if k is not None)
                    status[ql] = not any(M.accepted(q, int(k) & 0xFFFFFFFF) for k in cands)
                    if status[ql]:
                        new.append(q)
                    else:                                # the winner is found again, by the owner, in compare_with
                        pass
            for q in new:
                is_rep[q - b0] = True
            compare_with(new)
            reps.extend(sorted(new))
            reps.sort()
        for ql in range(nb):                             # close: clusters are numbered by representative creation
            q = b0 + ql
            if is_rep[ql]:
                cluster_of[q] = None
        for q in sorted(k for k in cluster_of if cluster_of[k] is None):
            cluster_of[q] = sum(1 for r in reps if r < q)
        for ql in range(nb):
            if not is_rep[ql]:
                cluster_of[b0 + ql] = cluster_of[int(best[ql]) & 0xFFFFFFFF]
    out = np.full(len(seqs), -1, dtype=np.int32)
    for k, i in enumerate(M.order):
        out[i] = cluster_of[k]
    return out, M.evaluated
