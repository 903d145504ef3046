"""Test infrastructure: a minimal BAM (BGZF) writer and reader-side helpers, straight from the SAM specification (sections
4.1, 4.2), so the native BAM loader can be tested against files made here.  Nothing in the product imports this."""
import struct
import zlib

import numpy as np

_NT16 = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}
_CIG = {c: i for i, c in enumerate("MIDNSHP=X")}


def bgzf_block(data: bytes) -> bytes:
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    comp = co.compress(data) + co.flush()
    bsize = len(comp) + 25
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", bsize) + comp +
            struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data)))


def bgzf_bytes(data: bytes, block=0xff00) -> bytes:
    out = [bgzf_block(data[i:i + block]) for i in range(0, len(data), block)]
    out.append(bgzf_block(b""))   # end-of-file marker
    return b"".join(out)


def bgzf_decompress(raw: bytes) -> bytes:
    """Every BGZF block of `raw`, checked (header, BSIZE, CRC, ISIZE) and concatenated."""
    out, i = [], 0
    while i < len(raw):
        assert raw[i:i + 4] == b"\x1f\x8b\x08\x04", "not a BGZF block"
        xlen = struct.unpack_from("<H", raw, i + 10)[0]
        assert raw[i + 12:i + 14] == b"BC"
        bsize = struct.unpack_from("<H", raw, i + 16)[0] + 1
        comp = raw[i + 12 + xlen:i + bsize - 8]
        crc, isize = struct.unpack_from("<II", raw, i + bsize - 8)
        data = zlib.decompress(comp, -15)
        assert len(data) == isize and (zlib.crc32(data) & 0xffffffff) == crc
        out.append(data)
        i += bsize
    return b"".join(out)


def _record(ref_id, pos0, name, mapq, flag, cigar, seq, qual, next_ref=-1, next_pos=-1, tlen=0):
    name_b = name.encode() + b"\0"
    cig = b"".join(struct.pack("<I", (n << 4) | _CIG[op]) for n, op in cigar)
    l_seq = len(seq)
    packed = bytearray((l_seq + 1) // 2)
    for i, c in enumerate(seq):
        packed[i >> 1] |= _NT16[c] << (0 if i & 1 else 4)
    body = (struct.pack("<iiBBHHHIiii", ref_id, pos0, len(name_b), mapq, 4680, len(cigar), flag, l_seq, next_ref, next_pos,
                        tlen) + name_b + cig + bytes(packed) + bytes(qual))
    return struct.pack("<i", len(body)) + body


def write_bam(path, refs, alignments, sorted_header=True, index=False, block=0xff00):
    """refs: [(name, length)]; alignments: dicts with ref_id, pos (1-based), name, mapq, flag, cigar [(n, op)], seq, qual
    (list of ints) and optionally tlen.  index: also write <path>.bai holding the linear index (SAM spec 5.2; no bins -- the
    loader only uses the linear index)."""
    text = ("@HD\tVN:1.6\tSO:%s\n" % ("coordinate" if sorted_header else "unsorted") +
            "".join(f"@SQ\tSN:{n}\tLN:{l}\n" for n, l in refs)).encode()
    data = b"BAM\1" + struct.pack("<i", len(text)) + text + struct.pack("<i", len(refs))
    for n, l in refs:
        nb = n.encode() + b"\0"
        data += struct.pack("<i", len(nb)) + nb + struct.pack("<i", l)
    recs = [_record(a["ref_id"], a["pos"] - 1, a["name"], a["mapq"], a["flag"], a["cigar"], a["seq"], a["qual"],
                    tlen=a.get("tlen", 0)) for a in alignments]
    starts = np.cumsum([len(data)] + [len(r) for r in recs])[:-1]   # uncompressed offset of every record
    raw = data + b"".join(recs)
    blocks = [bgzf_block(raw[i:i + block]) for i in range(0, len(raw), block)]
    with open(path, "wb") as f:
        f.write(b"".join(blocks) + bgzf_block(b""))
    if index:
        coff = np.cumsum([0] + [len(b) for b in blocks])
        lin = [dict() for _ in refs]
        for a, p in zip(alignments, starts):
            voff = (int(coff[p // block]) << 16) | int(p % block)
            ref_len = sum(n for n, op in a["cigar"] if op in "MDN=X")
            for w in range((a["pos"] - 1) >> 14, ((a["pos"] - 1 + max(ref_len, 1) - 1) >> 14) + 1):
                if w not in lin[a["ref_id"]] or voff < lin[a["ref_id"]][w]:
                    lin[a["ref_id"]][w] = voff
        out = b"BAI\1" + struct.pack("<i", len(refs))
        for d in lin:
            n_intv = (max(d) + 1) if d else 0
            out += struct.pack("<i", 0) + struct.pack("<i", n_intv) + b"".join(struct.pack("<Q", d.get(w, 0)) for w in range(n_intv))
        with open(path + ".bai", "wb") as f:
            f.write(out)


def sample_to_alignments(sample, L, ref, alt, rng, read_len_pad=5, mapq=60, chrom_id=0, genome=None):
    """Alignments (one per read, plain M CIGAR) that pile up to exactly `sample` (flattened sampleReads with 0-based sites
    `u` and signed qualities `bq`): every read spans its first to last site, shows the ref / alt allele with |bq| at its
    sites, and -- so that no unintended site is hit -- a base that is neither allele at every other site it crosses."""
    alns = []
    other = {("A", "C"): "G", ("A", "G"): "C", ("A", "T"): "C", ("C", "G"): "A", ("C", "T"): "A", ("G", "T"): "A"}
    L = np.asarray(L)
    for r in range(sample.nReads):
        a, b = sample.read_ptr[r], sample.read_ptr[r + 1]
        us, bqs = sample.u[a:b], sample.bq[a:b]
        start = int(L[us[0]]) - int(rng.integers(0, read_len_pad + 1))
        start = max(start, 1)
        end = int(L[us[-1]]) + int(rng.integers(0, read_len_pad + 1))
        n = end - start + 1
        seq = list(rng.choice(list("ACGT"), size=n))
        qual = list(rng.integers(20, 41, size=n))
        lo, hi = np.searchsorted(L, start), np.searchsorted(L, end, side="right")
        for t in range(lo, hi):   # neutralise every site the read crosses ...
            seq[int(L[t]) - start] = other[tuple(sorted((ref[t], alt[t])))]
        for t, q in zip(us, bqs):   # ... then set the ones the read reports
            seq[int(L[t]) - start] = alt[t] if q > 0 else ref[t]
            qual[int(L[t]) - start] = abs(int(q))
        alns.append(dict(ref_id=chrom_id, pos=start, name=f"r{r}", mapq=mapq, flag=0, cigar=[(n, "M")], seq="".join(seq),
                         qual=qual))
    alns.sort(key=lambda x: x["pos"])
    return alns
