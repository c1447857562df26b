"""Synthetic PacBio-shaped read sets and the caller-side k-mer dictionary (host, numpy).

This is the part of the pipeline *above* the hot path: what the reference does in
``include/kmercount.hpp:467-677`` (SplitCount: multiplicity of the canonical k-mer, reliable iff
``lower <= m <= upper``) and ``src/main.cpp:339-423`` (one ``(kmer_id, read_id, pos)`` tuple per read
position whose canonical k-mer is reliable, positions ascending inside a read).  k-mer ids are the
rank of the canonical k-mer in sorted order (the reference's ids are libcuckoo iteration order, i.e.
arbitrary -- SURVEY.md section 0.4; any id assignment is a valid run, parity is defined on the tuples).

Read-set parameters follow SURVEY.md section 8(d): uniform random genome, reads of a fixed template
length at uniformly random positions/strands, error model ``u<p_del`` drop, ``u<p_del+p_sub``
substitute by one of the 3 other bases, ``u<p_del+p_sub+p_ins`` keep and append one random base with
``(p_sub,p_ins,p_del) = e*(0.10,0.60,0.30)``.
"""
from __future__ import annotations

import dataclasses
import numpy as np

BASES = np.frombuffer(b"ACGT", dtype=np.uint8)
_CODE = np.full(256, 255, dtype=np.uint8)
for _i, _c in enumerate(b"ACGT"):
    _CODE[_c] = _i


@dataclasses.dataclass
class ReadSet:
    codes: np.ndarray      # uint8, 0..3 (A,C,G,T), all reads concatenated
    offsets: np.ndarray    # int64, nreads+1
    names: list

    @property
    def nreads(self) -> int:
        return len(self.offsets) - 1

    @property
    def lengths(self) -> np.ndarray:
        return np.diff(self.offsets).astype(np.uint32)

    def seq(self, r: int) -> bytes:
        return BASES[self.codes[self.offsets[r]:self.offsets[r + 1]]].tobytes()

    def seqs(self) -> list:
        asc = BASES[self.codes]
        return [asc[self.offsets[r]:self.offsets[r + 1]].tobytes() for r in range(self.nreads)]

    @staticmethod
    def from_strings(seqs, names=None) -> "ReadSet":
        raw = [np.frombuffer(x.encode() if isinstance(x, str) else bytes(x), dtype=np.uint8) for x in seqs]
        offs = np.zeros(len(raw) + 1, dtype=np.int64)
        np.cumsum([len(x) for x in raw], out=offs[1:])
        codes = _CODE[np.concatenate(raw)] if raw else np.zeros(0, np.uint8)
        if codes.size and codes.max() > 3:
            raise ValueError("reads must be upper-case ACGT")
        return ReadSet(codes.astype(np.uint8), offs, list(names) if names is not None else ["r%d" % i for i in range(len(raw))])

    def subset(self, n: int) -> "ReadSet":
        return ReadSet(self.codes[: self.offsets[n]], self.offsets[: n + 1].copy(), self.names[:n])


def make_reads(nreads: int, read_len: int = 10000, coverage: float = 30.0, err: float = 0.15, seed: int = 1,
               mix=(0.10, 0.60, 0.30), genome_len: int | None = None, len_jitter: float = 0.0) -> ReadSet:
    """SURVEY 8(d) generator.  ``mix`` = (sub, ins, del) shares of the error rate."""
    rng = np.random.default_rng(seed)
    G = int(genome_len if genome_len is not None else max(read_len + 1, round(nreads * read_len / coverage)))
    genome = rng.integers(0, 4, size=G, dtype=np.uint8)
    p_sub, p_ins, p_del = (err * m for m in mix)
    chunks, offsets, names = [], [0], []
    step = max(1, (1 << 24) // max(read_len, 1))
    for lo in range(0, nreads, step):
        hi = min(nreads, lo + step)
        n = hi - lo
        if len_jitter > 0:
            L = np.clip(np.rint(read_len * np.exp(rng.normal(0, len_jitter, size=n))), 200, min(60000, G)).astype(np.int64)
        else:
            L = np.full(n, min(read_len, G), dtype=np.int64)
        start = (rng.random(n) * (G - L + 1)).astype(np.int64)
        strand = rng.integers(0, 2, size=n)
        toff = np.concatenate([[0], np.cumsum(L)])
        rid = np.repeat(np.arange(n), L)
        within = np.arange(toff[-1]) - toff[rid]
        # reverse strand: template = revcomp(genome[start:start+L])
        gpos = np.where(strand[rid] == 1, start[rid] + L[rid] - 1 - within, start[rid] + within)
        tmpl = genome[gpos]
        tmpl = np.where(strand[rid] == 1, 3 - tmpl, tmpl).astype(np.uint8)
        u = rng.random(tmpl.shape[0], dtype=np.float32)
        sub = (u >= p_del) & (u < p_del + p_sub)
        ins = (u >= p_del + p_sub) & (u < p_del + p_sub + p_ins)
        keep = u >= p_del
        base = np.where(sub, (tmpl + rng.integers(1, 4, size=tmpl.shape[0], dtype=np.uint8)) & 3, tmpl).astype(np.uint8)
        emit = keep.astype(np.int64) + ins.astype(np.int64)          # bases produced by each template base
        out_off = np.concatenate([[0], np.cumsum(emit)])
        out = np.empty(out_off[-1], dtype=np.uint8)
        kidx = np.nonzero(keep)[0]
        out[out_off[kidx]] = base[kidx]
        iidx = np.nonzero(ins)[0]
        out[out_off[iidx] + 1] = rng.integers(0, 4, size=iidx.shape[0], dtype=np.uint8)
        rl = np.add.reduceat(emit, toff[:-1]) if n else np.zeros(0, dtype=np.int64)
        chunks.append(out)
        for j in range(n):
            offsets.append(offsets[-1] + int(rl[j]))
            names.append("r%d_%d_%d_%d" % (lo + j, start[j], L[j], strand[j]))
    codes = np.concatenate(chunks) if chunks else np.zeros(0, dtype=np.uint8)
    return ReadSet(codes, np.asarray(offsets, dtype=np.int64), names)


def make_reads_torch(nreads: int, read_len: int = 10000, coverage: float = 30.0, err: float = 0.15, seed: int = 1,
                     mix=(0.10, 0.60, 0.30), genome_len: int | None = None, device: str = "cuda:0",
                     chunk_bases: int = 1 << 27) -> ReadSet:
    """The SURVEY 8(d) generator of make_reads on a torch device (the GPU box: 100k x 10 kb reads in about a second instead of
    25 s of numpy, 1M x 15 kb HiFi reads in seconds instead of minutes).  Same parameters, same error model (per template base
    u < p_del drop, u < p_del + p_sub substitute, u < p_del + p_sub + p_ins keep and append one random base; reverse strand =
    reverse complement of the template), same read names; torch's generator instead of numpy's -- the parameters are the contract,
    the PRNG is not.  Fixed template length (no jitter).  The bases come back to the host once (ReadSet holds numpy arrays)."""
    import torch
    dev = torch.device(device)
    gen = torch.Generator(device=dev)
    gen.manual_seed(int(seed))
    G = int(genome_len if genome_len is not None else max(read_len + 1, round(nreads * read_len / coverage)))
    L = int(min(read_len, G))
    genome = torch.randint(0, 4, (G,), dtype=torch.uint8, device=dev, generator=gen)
    p_sub, p_ins, p_del = (err * m for m in mix)
    step = max(1, int(chunk_bases) // max(L, 1))
    chunks, lens, names = [], [], []
    within = torch.arange(L, device=dev, dtype=torch.int64)[None, :]
    for lo in range(0, nreads, step):
        n = min(nreads, lo + step) - lo
        start = (torch.rand(n, device=dev, generator=gen, dtype=torch.float64) * (G - L + 1)).to(torch.int64).clamp_(max=G - L)
        strand = torch.randint(0, 2, (n,), device=dev, generator=gen, dtype=torch.int64)
        rev = (strand == 1)[:, None]
        gpos = torch.where(rev, start[:, None] + (L - 1) - within, start[:, None] + within)
        tmpl = genome[gpos.reshape(-1)].reshape(n, L)
        del gpos
        tmpl = torch.where(rev, 3 - tmpl, tmpl)
        u = torch.rand((n, L), device=dev, generator=gen, dtype=torch.float32)
        keep = u >= p_del
        sub = keep & (u < p_del + p_sub)
        ins = (u >= p_del + p_sub) & (u < p_del + p_sub + p_ins)
        del u
        shift = torch.randint(1, 4, (n, L), device=dev, generator=gen, dtype=torch.uint8)
        base = torch.where(sub, (tmpl + shift) & 3, tmpl)
        del shift, tmpl, sub
        emit = keep.to(torch.int64) + ins.to(torch.int64)
        rl = emit.sum(dim=1)
        off = torch.cumsum(emit.reshape(-1), 0) - emit.reshape(-1)
        del emit
        total = int(rl.sum().item())
        out = torch.empty(total, dtype=torch.uint8, device=dev)
        kf, inf = keep.reshape(-1), ins.reshape(-1)
        out[off[kf]] = base.reshape(-1)[kf]
        ip = off[inf] + 1
        out[ip] = torch.randint(0, 4, (int(ip.shape[0]),), device=dev, generator=gen, dtype=torch.uint8)
        del off, kf, inf, ip, base, keep, ins
        chunks.append(out.cpu().numpy())
        lens.append(rl.cpu().numpy())
        st, sd = start.cpu().numpy(), strand.cpu().numpy()
        names.extend("r%d_%d_%d_%d" % (lo + j, st[j], L, sd[j]) for j in range(n))
        del out
    codes = np.concatenate(chunks) if chunks else np.zeros(0, dtype=np.uint8)
    offsets = np.zeros(nreads + 1, dtype=np.int64)
    if lens:
        np.cumsum(np.concatenate(lens), out=offsets[1:])
    if dev.type == "cuda":
        torch.cuda.empty_cache()
    return ReadSet(codes, offsets, names)


def make_reads_fast(nreads: int, **kw) -> ReadSet:
    """make_reads_torch on the GPU when torch sees one, make_reads (numpy) otherwise: the bench and the large GPU tests"""
    try:
        import torch
        if torch.cuda.is_available():
            return make_reads_torch(nreads, **kw)
    except ImportError:
        pass
    kw.pop("device", None); kw.pop("chunk_bases", None)
    return make_reads(nreads, **kw)


def make_reads_from_intervals(start, end, genome_len: int = 4641652, err: float = 0.15, seed: int = 1,
                              mix=(0.10, 0.60, 0.30)) -> ReadSet:
    """one read per interval [start,end) (clipped to the genome) of a uniform-random genome,
    uniform strand, the same error model as make_reads."""
    rng = np.random.default_rng(seed)
    genome = rng.integers(0, 4, size=genome_len, dtype=np.uint8)
    p_sub, p_ins, p_del = (err * m for m in mix)
    st = np.clip(np.asarray(start, np.int64), 0, genome_len - 1)
    en = np.clip(np.asarray(end, np.int64), st + 1, genome_len)
    seqs, names = [], []
    for r in range(len(st)):
        t = genome[st[r]:en[r]]
        strand = int(rng.integers(0, 2))
        if strand:
            t = (3 - t)[::-1]
        u = rng.random(t.shape[0], dtype=np.float32)
        keep = u >= p_del
        sub = keep & (u < p_del + p_sub)
        ins = (u >= p_del + p_sub) & (u < p_del + p_sub + p_ins)
        base = np.where(sub, (t + rng.integers(1, 4, size=t.shape[0], dtype=np.uint8)) & 3, t).astype(np.uint8)
        emit = keep.astype(np.int64) + ins.astype(np.int64)
        off = np.concatenate([[0], np.cumsum(emit)])
        out = np.empty(off[-1], dtype=np.uint8)
        k_ = np.nonzero(keep)[0]
        out[off[k_]] = base[k_]
        i_ = np.nonzero(ins)[0]
        out[off[i_] + 1] = rng.integers(0, 4, size=i_.shape[0], dtype=np.uint8)
        seqs.append(out)
        names.append("e%d_%d_%d_%d" % (r, st[r], en[r] - st[r], strand))
    offs = np.zeros(len(seqs) + 1, dtype=np.int64)
    np.cumsum([len(x) for x in seqs], out=offs[1:])
    return ReadSet(np.concatenate(seqs) if seqs else np.zeros(0, np.uint8), offs, names)


def readset_from_seqs(seqs, names=None) -> ReadSet:
    offs = np.zeros(len(seqs) + 1, dtype=np.int64)
    for i, s in enumerate(seqs):
        offs[i + 1] = offs[i] + len(s)
    raw = np.frombuffer(b"".join(bytes(s) if not isinstance(s, str) else s.encode() for s in seqs), dtype=np.uint8)
    codes = _CODE[raw]
    if (codes == 255).any():
        raise ValueError("reads must be upper-case ACGT only (align.hpp:40-55 complementbase asserts otherwise)")
    return ReadSet(codes, offs, list(names) if names is not None else ["read%d" % i for i in range(len(seqs))])


def write_fastq(path: str, rs: ReadSet, qual: str = "5") -> None:
    asc = BASES[rs.codes]
    with open(path, "wb") as f:
        for r in range(rs.nreads):
            s = asc[rs.offsets[r]:rs.offsets[r + 1]].tobytes()
            f.write(b"@" + rs.names[r].encode() + b"\n" + s + b"\n+\n" + qual.encode() * len(s) + b"\n")


def read_fastq(path: str) -> ReadSet:
    """Names are cut at the first blank and lose the '@' (kmercode/fq_reader.c:90-130, main.cpp:357)."""
    import gzip
    op = gzip.open if str(path).endswith(".gz") else open
    names, seqs = [], []
    with op(path, "rb") as f:
        lines = f.read().split(b"\n")
    i = 0
    while i + 3 < len(lines) + 1 and i < len(lines) and lines[i].startswith(b"@"):
        names.append(lines[i][1:].split()[0].decode())
        seqs.append(lines[i + 1].strip())
        i += 4
    return readset_from_seqs(seqs, names)


def kmer_words(rs: ReadSet, k: int):
    """Forward and reverse-complement 2k-bit words of the k-mer starting at every base of the
    concatenated read array, plus the validity mask (k-mer does not cross a read end).
    Word order = kmercode/Kmer.cpp:160-169 (numeric order of the left-aligned packed word)."""
    assert 1 <= k <= 32
    n = rs.codes.shape[0]
    m = max(0, n - k + 1)
    c = rs.codes.astype(np.uint64)
    fw = np.zeros(m, dtype=np.uint64)
    rc = np.zeros(m, dtype=np.uint64)
    for t in range(k):
        sl = c[t:t + m]
        fw = (fw << np.uint64(2)) | sl
        rc = rc | ((np.uint64(3) - sl) << np.uint64(2 * t))
    lens = np.diff(rs.offsets)
    rid = np.repeat(np.arange(rs.nreads, dtype=np.int64), lens)[:m]
    pos = (np.arange(m, dtype=np.int64) - rs.offsets[rid])
    valid = pos <= (lens[rid] - k)
    return fw, rc, rid, pos, valid


@dataclasses.dataclass
class Tuples:
    kmer: np.ndarray   # uint32
    read: np.ndarray   # uint32 (non-decreasing)
    pos: np.ndarray    # uint16 (ascending inside a read)
    nkmers: int


def _count_and_tuples_torch(rs: ReadSet, k: int, lower: int, upper: int, device: str) -> Tuples:
    """same result as the numpy path, with the sort/unique of the k-mer words done by torch on the GPU
    (input preparation for benchmarks -- not part of the overlap hot path)"""
    import torch
    dev = torch.device(device)
    n = rs.codes.shape[0]
    m = max(0, n - k + 1)
    c = torch.from_numpy(rs.codes).to(dev).to(torch.int64)
    fw = torch.zeros(m, dtype=torch.int64, device=dev)
    rc = torch.zeros(m, dtype=torch.int64, device=dev)
    for t in range(k):
        sl = c[t:t + m]
        fw = (fw << 2) | sl
        rc = rc | ((3 - sl) << (2 * t))
    canon = torch.minimum(fw, rc)
    del fw, rc, c
    lens = torch.from_numpy(np.diff(rs.offsets)).to(dev)
    offs = torch.from_numpy(rs.offsets).to(dev)
    rid = torch.repeat_interleave(torch.arange(rs.nreads, device=dev), lens)[:m]
    pos = torch.arange(m, device=dev) - offs[rid]
    valid = pos <= (lens[rid] - k)
    canon, rid, pos = canon[valid], rid[valid], pos[valid]
    uniq, cnt = torch.unique(canon, return_counts=True)
    rel = uniq[(cnt >= lower) & (cnt <= upper)]
    del uniq, cnt
    idx = torch.searchsorted(rel, canon)
    idx[idx >= rel.shape[0]] = 0
    hit = rel[idx] == canon
    out = Tuples(idx[hit].to(torch.int32).cpu().numpy().astype(np.uint32), rid[hit].to(torch.int32).cpu().numpy().astype(np.uint32),
                 pos[hit].to(torch.int32).cpu().numpy().astype(np.uint16), int(rel.shape[0]))
    torch.cuda.empty_cache()
    return out


def count_and_tuples(rs: ReadSet, k: int = 17, lower: int = 2, upper: int = 8, device: str | None = None) -> Tuples:
    """kmercount.hpp:467-677 (SplitCount, exact 1-thread semantics) + main.cpp:391-416."""
    if int(np.diff(rs.offsets).max(initial=0)) >= 65536:
        raise ValueError("reads must be shorter than 65,536 bases (u16 positions, common.h:122-126)")
    if device is not None and k <= 31:
        try:
            import torch
            if torch.cuda.is_available():
                return _count_and_tuples_torch(rs, k, lower, upper, device)
        except ImportError:
            pass
    fw, rc, rid, pos, valid = kmer_words(rs, k)
    canon = np.minimum(fw, rc)[valid]
    rid = rid[valid]
    pos = pos[valid]
    uniq, cnt = np.unique(canon, return_counts=True)
    rel = uniq[(cnt >= lower) & (cnt <= upper)]
    idx = np.searchsorted(rel, canon)
    idx[idx >= rel.shape[0]] = 0
    hit = rel[idx] == canon if rel.shape[0] else np.zeros(canon.shape[0], dtype=bool)
    return Tuples(idx[hit].astype(np.uint32), rid[hit].astype(np.uint32), pos[hit].astype(np.uint16), int(rel.shape[0]))


def read_mtx_tuples(path: str) -> Tuples:
    """``readbykmers.mtx`` written by the reference's -DWRITEDATAMATRIX build
    (include/common/bellaio.h:2-47): header ``nreads nkmers ntuples`` then ``read+1 kmer+1 pos``."""
    import gzip
    op = gzip.open if str(path).endswith(".gz") else open
    with op(path, "rb") as f:
        hdr = f.readline().split()
        arr = np.loadtxt(f, dtype=np.int64, ndmin=2)
    nk = int(hdr[1])
    if arr.size == 0:
        return Tuples(np.zeros(0, np.uint32), np.zeros(0, np.uint32), np.zeros(0, np.uint16), nk)
    return Tuples((arr[:, 1] - 1).astype(np.uint32), (arr[:, 0] - 1).astype(np.uint32), arr[:, 2].astype(np.uint16), nk)
