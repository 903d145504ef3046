"""Writes tests/golden/sam_spec_example.bam (+ .bai) and sam_spec_example.json: the example alignment of the SAM specification
(section 1.1: r001/1, r002, r003, r004, r003's supplementary line, r001/2 on a 45 bp reference), encoded byte by byte from the
spec's tables (4.2 BAM record layout, 4.1 BGZF, 5.2 BAI, 5.3 reg2bin) by THIS script -- not by tests/bamutil.py, the writer
the other loader tests use, and with what that writer never produces: records that straddle BGZF blocks (blocks of 96
bytes), an insertion, a deletion, a padding and a skipped region in the CIGARs, soft and hard clips, a supplementary
alignment, auxiliary fields (SA:Z, NM:i), a bin index with several bins beside the linear index.  The spec's example has
no base qualities ('*'); here every base has quality 40 so that the loader keeps them (mapping qualities as in the spec).

The expected pile-up in the JSON was worked out BY HAND from the spec's picture of the example (see the comments below), not
by running any code.

Run:  python tests/golden/make_sam_spec_bam.py
"""
import json
import os
import struct
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))

REF = "AGCATGTTAGATAAGATAGCTGTGCTAGTAGGCAGTCAGCGCCAT"   # 45 bp: the spec's `ref` without the two pad columns
# qname, flag, pos (1-based), mapq, cigar, rnext ("=": same), pnext, tlen, seq, aux
ALN = [
    ("r001", 99, 7, 30, "8M2I4M1D3M", "=", 37, 39, "TTAGATAAAGGATACTG", b""),
    ("r002", 0, 9, 30, "3S6M1P1I4M", "*", 0, 0, "AAAAGATAAGGATA", b""),
    ("r003", 0, 9, 30, "5S6M", "*", 0, 0, "GCCTAAGCTAA", b"SAZref,29,-,6H5M,17,0;\0"),
    ("r004", 0, 16, 30, "6M14N5M", "*", 0, 0, "ATAGCTTCAGC", b""),
    ("r003", 2064, 29, 17, "6H5M", "*", 0, 0, "TAGGC", b"SAZref,9,+,5S6M,30,1;\0"),
    ("r001", 147, 37, 30, "9M", "=", 7, -39, "CAGCGGCAT", b"NMC\x01"),
]
# a second, long reference ("big", 200 000 bp of A) so that the index has several bins and 16 kb intervals: 30-base reads of
# C's at these 1-based positions (the one at 16 370 crosses the 16 384 boundary: bin 585, the others bins 4681 + interval)
BIG = [1000, 16370, 40000, 40010, 150000]
OPS = "MIDNSHP=X"
NT16 = "=ACMGRSVTWYHKDBN"


def reg2bin(beg, end):   # SAM spec 5.3 (0-based, half open)
    end -= 1
    if beg >> 14 == end >> 14: return ((1 << 15) - 1) // 7 + (beg >> 14)
    if beg >> 17 == end >> 17: return ((1 << 12) - 1) // 7 + (beg >> 17)
    if beg >> 20 == end >> 20: return ((1 << 9) - 1) // 7 + (beg >> 20)
    if beg >> 23 == end >> 23: return ((1 << 6) - 1) // 7 + (beg >> 23)
    if beg >> 26 == end >> 26: return ((1 << 3) - 1) // 7 + (beg >> 26)
    return 0


def parse_cigar(c):
    out, n = [], ""
    for ch in c:
        if ch.isdigit():
            n += ch
        else:
            out.append((int(n), OPS.index(ch)))
            n = ""
    return out


def record(qname, flag, pos, mapq, cigar, rnext, pnext, tlen, seq, aux, ref_id=0):
    cg = parse_cigar(cigar)
    ref_len = sum(n for n, op in cg if op in (0, 2, 3, 7, 8))
    name = qname.encode() + b"\0"
    packed = bytearray((len(seq) + 1) // 2)
    for i, ch in enumerate(seq):
        packed[i >> 1] |= NT16.index(ch) << (4 if i % 2 == 0 else 0)
    body = struct.pack("<iiBBHHHiiii", ref_id, pos - 1, len(name), mapq, reg2bin(pos - 1, pos - 1 + max(ref_len, 1)), len(cg), flag,
                       len(seq), 0 if rnext == "=" else -1, pnext - 1, tlen)
    body += name + b"".join(struct.pack("<I", n << 4 | op) for n, op in cg) + bytes(packed) + bytes([40] * len(seq)) + aux
    return struct.pack("<i", len(body)) + body, ref_len


def bgzf_member(data):
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    comp = co.compress(data) + co.flush()
    bsize = len(comp) + 25
    return (bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 66, 67, 2, 0]) + struct.pack("<H", bsize) + comp +
            struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data)))


def main(block=96):
    text = b"@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:ref\tLN:45\n@SQ\tSN:big\tLN:200000\n"
    stream = (b"BAM\1" + struct.pack("<i", len(text)) + text + struct.pack("<i", 2) + struct.pack("<i", 4) + b"ref\0" +
              struct.pack("<i", 45) + struct.pack("<i", 4) + b"big\0" + struct.pack("<i", 200000))
    spans = []   # (uncompressed begin, end, 0-based ref begin, end, reference) of every record
    for a in ALN:
        rec, ref_len = record(*a)
        spans.append((len(stream), len(stream) + len(rec), a[2] - 1, a[2] - 1 + max(ref_len, 1), 0))
        stream += rec
    for i, pos in enumerate(BIG):
        rec, ref_len = record("b%d" % i, 0, pos, 60, "30M", "*", 0, 0, "C" * 30, b"", ref_id=1)
        spans.append((len(stream), len(stream) + len(rec), pos - 1, pos - 1 + 30, 1))
        stream += rec
    # BGZF: fixed-size blocks of `block` uncompressed bytes (records straddle them), then the empty end-of-file block
    out, starts = b"", []
    for off in range(0, len(stream), block):
        starts.append(len(out))
        out += bgzf_member(stream[off:off + block])
    eof_at = len(out)
    out += bgzf_member(b"")
    open(os.path.join(HERE, "sam_spec_example.bam"), "wb").write(out)

    def voff(u):   # virtual offset of uncompressed position u
        if u == len(stream):
            return eof_at << 16
        return starts[u // block] << 16 | (u % block)
    # BAI: per reference its bins (one chunk per record) and the linear index over 16 kb intervals: the smallest offset of an
    # alignment overlapping each interval, 0 where none does
    bai = b"BAI\1" + struct.pack("<i", 2)
    n_bins_total = 0
    for rid, length in ((0, 45), (1, 200000)):
        bins, n_intv = {}, (length + 16383) >> 14
        lin = [0] * n_intv
        for (ub, ue, rb, re_, r) in spans:
            if r != rid:
                continue
            bins.setdefault(reg2bin(rb, re_), []).append((voff(ub), voff(ue)))
            for iv in range(rb >> 14, ((re_ - 1) >> 14) + 1):
                if lin[iv] == 0 or voff(ub) < lin[iv]:
                    lin[iv] = voff(ub)
        n_bins_total += len(bins)
        bai += struct.pack("<i", len(bins))
        for b in sorted(bins):
            bai += struct.pack("<Ii", b, len(bins[b])) + b"".join(struct.pack("<QQ", s, e) for s, e in bins[b])
        bai += struct.pack("<i", n_intv) + b"".join(struct.pack("<Q", v) for v in lin)
    open(os.path.join(HERE, "sam_spec_example.bam.bai"), "wb").write(bai)

    # ---- the expected pile-up, by hand from the spec's picture --------------------------------------------------------
    #   Coor     12345678901234  5678901234567890123456789012345
    #   ref      AGCATGTTAGATAA**GATAGCTGTGCTAGTAGGCAGTCAGCGCCAT
    #   +r001/1        TTAGATAAAGGATA*CTG                       8M 7-14, 2I, 4M 15-18, 1D 19, 3M 20-22
    #   +r002         aaaAGATAA*GGATA                           3S (would lie on 6-8), 6M 9-14, 1P, 1I, 4M 15-18
    #   +r003       gcctaAGCTAA                                 5S (would lie on 4-8), 6M 9-14 with C at 11
    #   +r004                     ATAGCT..............TCAGC     6M 16-21, 14N 22-35, 5M 36-40
    #   -r003                            ttagctTAGGC            supplementary (flag 2064): not used
    #   -r001/2                                        CAGCGGCAT  9M 37-45 with G at 42
    sites = dict(L=[8, 11, 13, 17, 19, 20, 30, 38, 42], ref="TAATGCAAC", alt="ACGCATGTG")
    q = 30   # min(base quality 40, mapping quality 30)
    expect = {
        "default": [   # bqFilter 17, mates merged, soft clips not used; reads in file order of their first alignment
            dict(name="r001", u=[0, 1, 2, 3, 5, 7, 8], bq=[-q, -q, -q, -q, -q, -q, q]),   # 8 T, 11 A, 13 A, 17 T, 20 C | 38 A, 42 G (alt)
            dict(name="r002", u=[1, 2, 3], bq=[-q, -q, -q]),                              # 11 A, 13 A, 17 T
            dict(name="r003", u=[1, 2], bq=[q, -q]),                                      # 11 C (alt), 13 A
            dict(name="r004", u=[3, 4, 5, 7], bq=[-q, -q, -q, -q]),                       # 17 T, 19 G, 20 C | 38 A
        ],
        "soft_clips": [   # useSoftClippedBases: r002's aaa lies on 6-8 (site 8: A = alt), r003's gccta on 4-8 (site 8: A = alt)
            dict(name="r001", u=[0, 1, 2, 3, 5, 7, 8], bq=[-q, -q, -q, -q, -q, -q, q]),
            dict(name="r002", u=[0, 1, 2, 3], bq=[q, -q, -q, -q]),
            dict(name="r003", u=[0, 1, 2], bq=[q, q, -q]),
            dict(name="r004", u=[3, 4, 5, 7], bq=[-q, -q, -q, -q]),
        ],
        "window_30_45": [   # alignments overlapping 30..45 only: r004 (16-40) whole, r001/2 alone (its mate ends at 22)
            dict(name="r004", u=[3, 4, 5, 7], bq=[-q, -q, -q, -q]),
            dict(name="r001", u=[7, 8], bq=[-q, q]),
        ],
        "stats_default": dict(seen=6, used=5, by_flags=1, mates_merged=1),
    }
    # the long reference: sites at 1010, 16390 (inside the read that crosses 16 384), 40005, 40020, 150010, all ref A / alt C
    expect["big"] = dict(L=[1010, 16390, 40005, 40020, 150010], ref="AAAAA", alt="CCCCC",
                         whole=[dict(u=[0], bq=[40]), dict(u=[1], bq=[40]), dict(u=[2, 3], bq=[40, 40]), dict(u=[3], bq=[40]),
                                dict(u=[4], bq=[40])],          # (mapping quality 60: q = base quality 40)
                         window_39000_41000=[dict(u=[2, 3], bq=[40, 40]), dict(u=[3], bq=[40])],
                         window_16384_17000=[dict(u=[1], bq=[40])])
    json.dump(dict(sites=sites, expect=expect, n_bgzf_blocks=len(starts) + 1, stream_bytes=len(stream)),
              open(os.path.join(HERE, "sam_spec_example.json"), "w"), indent=1)
    print(len(out), "bytes,", len(starts) + 1, "BGZF blocks,", n_bins_total, "bins")


if __name__ == "__main__":
    main()
