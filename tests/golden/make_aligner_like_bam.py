"""Writes tests/golden/aligner_like.bam (+ .bai) and aligner_like.json: a stand-in for BASELINE configs[0]'s input -- what
`bwa mem | samtools sort | samtools markdup` (short reads) and minimap2 (a long read) actually emit, which neither the SAM
specification's toy example nor tests/bamutil.py's writer contains -- assembled byte by byte from the SAM specification (4.2
record layout, 4.2.4 auxiliary data, 4.1 BGZF, 5.2 BAI) by THIS script, reusing only the field encoders of
make_sam_spec_bam.py (its own reading of the same tables):

  * a whole-genome header: @HD, 25 @SQ (chr1 .. chr22, chrX, chrY, chrM: the target chr20 is reference 19), @RG, two @PG lines;
    alignments on chr1 before and chrX after the target, and unplaced unmapped reads (reference -1) at the end;
  * flags: proper pairs (99 / 147), a placed UNMAPPED mate (flag 133, the mate's reference and position, CIGAR `*`), secondary
    (256), supplementary (2048), duplicate (1024), QC-fail (512); MAPQ 0; a reverse-strand read with soft clips;
  * auxiliary fields as the aligners write them: NM:i (as C), MD:Z, AS:i / XS:i (as C / c), RG:Z, MC:Z, SA:Z, and `B` ARRAYS
    (ZB:B:s, ZC:B:C, a float array ZF:B:f) ahead of the fields behind them;
  * a long read whose CIGAR has 70 000 operations: n_cigar_op cannot hold it, so the record carries the placeholder
    `<l_seq>S<ref_len>N` and the real CIGAR in CG:B,I (SAM spec 4.2.2); the 332 kB record straddles five 64 KiB BGZF blocks;
  * short records straddling 200-byte BGZF blocks; an N base and a low-quality base on a SNP.

The expected pile-up in the JSON is derived BY HAND below from the alignments' coordinates -- not by running any code.

Run:  python tests/golden/make_aligner_like_bam.py
"""
import json
import os
import struct

from make_sam_spec_bam import NT16, OPS, bgzf_member, parse_cigar, reg2bin

HERE = os.path.dirname(os.path.abspath(__file__))

CHROMS = [("chr%d" % i, 1000000 + 1000 * i) for i in range(1, 23)] + [("chrX", 900000), ("chrY", 500000), ("chrM", 16569)]
TARGET = 19   # chr20


def aux_Z(tag, s): return tag.encode() + b"Z" + s.encode() + b"\0"
def aux_C(tag, v): return tag.encode() + b"C" + bytes([v])
def aux_c(tag, v): return tag.encode() + b"c" + struct.pack("<b", v)
def aux_i(tag, v): return tag.encode() + b"i" + struct.pack("<i", v)
def aux_B(tag, sub, vals):
    fmt = {"c": "b", "C": "B", "s": "h", "S": "H", "i": "i", "I": "I", "f": "f"}[sub]
    return tag.encode() + b"B" + sub.encode() + struct.pack("<I", len(vals)) + b"".join(struct.pack("<" + fmt, v) for v in vals)


def record(ref_id, pos, qname, flag, mapq, cigar_ops, seq, quals, aux=b"", next_ref=-1, pnext=0, tlen=0, n_cigar_field=None,
           ref_len=None):
    """One BAM alignment record (SAM spec 4.2).  cigar_ops: list of (len, op index); n_cigar_field / ref_len: what the fixed
    fields say when the operations themselves travel in CG:B,I."""
    if ref_len is None:
        ref_len = sum(n for n, op in cigar_ops if op in (0, 2, 3, 7, 8))
    name = qname.encode() + b"\0"
    packed = bytearray((len(seq) + 1) // 2)
    for i, ch in enumerate(seq):
        packed[i >> 1] |= NT16.index(ch) << (4 if i % 2 == 0 else 0)
    n_cig = len(cigar_ops) if n_cigar_field is None else n_cigar_field
    end = pos - 1 + max(ref_len, 1)
    body = struct.pack("<iiBBHHHiiii", ref_id, pos - 1, len(name), mapq, reg2bin(pos - 1, end) if ref_id >= 0 else 4680, n_cig, flag,
                       len(seq), next_ref, pnext - 1, tlen)
    body += name + b"".join(struct.pack("<I", n << 4 | op) for n, op in cigar_ops) + bytes(packed) + bytes(quals) + aux
    return struct.pack("<i", len(body)) + body, ref_len


def seq_with(n, fill, subs):
    s = [fill] * n
    for i, ch in subs.items():
        s[i] = ch
    return "".join(s)


def main():
    text = ("@HD\tVN:1.6\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % c for c in CHROMS) +
            "@RG\tID:grp1\tSM:NA12878\tPL:ILLUMINA\tLB:lib1\n"
            "@PG\tID:bwa\tPN:bwa\tVN:0.7.17-r1188\tCL:bwa mem -R @RG\\tID:grp1 ref.fa r1.fq r2.fq\n"
            "@PG\tID:samtools\tPN:samtools\tPP:bwa\tVN:1.17\tCL:samtools markdup - out.bam\n").encode()
    stream = b"BAM\1" + struct.pack("<i", len(text)) + text + struct.pack("<i", len(CHROMS))
    for name, ln in CHROMS:
        stream += struct.pack("<i", len(name) + 1) + name.encode() + b"\0" + struct.pack("<i", ln)
    rg = aux_Z("RG", "grp1")
    recs = []   # (record bytes, ref id, 0-based begin, end)

    def add(ref_id, pos, *a, **k):
        r, ref_len = record(ref_id, pos, *a, **k)
        recs.append((r, ref_id, pos - 1, pos - 1 + max(ref_len, 1)))

    M = lambda n: [(n, 0)]
    # ---- chr1 (reference 0): before the target
    add(0, 500, "c1a", 0, 60, M(30), "A" * 30, [30] * 30, rg + aux_C("NM", 0))
    add(0, 900000, "c1b", 16, 60, M(30), "C" * 30, [30] * 30, rg)
    # ---- chr20 (reference 19).  Sites (1-based) and alleles:
    #        L    = 1000 1010 1020 1050 1100 3000 3001 60000
    #        ref  =  A    A    C    G    C    A    A    A
    #        alt  =  T    C    G    T    T    C    C    G
    # r1/1  flag 99, pos 990, 30M: covers 990..1019 -> site 1000 is read index 10 (T = alt), site 1010 index 20 (A = ref); quality 35,
    #       MAPQ 60 -> q = 35
    add(TARGET, 990, "r1", 99, 60, M(30), seq_with(30, "A", {10: "T"}), [35] * 30,
        aux_C("NM", 1) + aux_Z("MD", "10A19") + aux_C("AS", 25) + aux_c("XS", 0) + rg + aux_Z("MC", "30M"),
        next_ref=TARGET, pnext=1040, tlen=80)
    # dup   flag 1024 (PCR duplicate of r1/1's position): covers 1000, 1010 -- dropped by its flag
    add(TARGET, 991, "dup", 1024, 60, M(30), "A" * 30, [35] * 30, rg)
    # sec   flag 256, supp flag 2048 (with SA:Z), qcf flag 512: dropped by their flags
    add(TARGET, 992, "sec", 256, 60, M(30), "A" * 30, [35] * 30, rg)
    add(TARGET, 993, "sup", 2048, 60, [(10, 5)] + M(20), "A" * 20, [35] * 20, aux_Z("SA", "chr1,500,+,10M20S,60,0;") + rg)
    add(TARGET, 994, "qcf", 512, 60, M(30), "A" * 30, [35] * 30, rg)
    # mq0   MAPQ 0 (a repeat): covers 1000, 1010, 1020 -- dropped (mapping quality below bqFilter)
    add(TARGET, 995, "mq0", 0, 0, M(30), seq_with(30, "A", {5: "T", 15: "C"}), [35] * 30, aux_c("XS", 25) + rg)
    # um    flag 133: the UNMAPPED mate of a pair, placed at its mate's coordinates, no CIGAR -- dropped by its flag
    add(TARGET, 1005, "um", 133, 0, [], "ACGTACGTAC", [20] * 10, rg, next_ref=TARGET, pnext=1005)
    # r2    flag 16, pos 1015, 5S20M5S: aligned part 1015..1034; site 1020 is aligned offset 5 = read index 10 (C = ref);
    #       quality 30, MAPQ 40 -> q = 30.  B arrays ahead of the other fields.
    add(TARGET, 1015, "r2", 16, 40, [(5, 4)] + M(20) + [(5, 4)], seq_with(30, "C", {}), [30] * 30,
        aux_B("ZB", "s", [1, -2, 3]) + aux_B("ZC", "C", list(range(7))) + aux_B("ZF", "f", [0.5, 1.5]) + aux_C("NM", 0) + rg)
    # r1/2  flag 147, pos 1040, 30M: covers 1040..1069 -> site 1050 is index 10 (G = ref); merged into r1
    add(TARGET, 1040, "r1", 147, 60, M(30), seq_with(30, "G", {}), [35] * 30, aux_C("NM", 0) + rg + aux_Z("MC", "30M"),
        next_ref=TARGET, pnext=990, tlen=-80)
    # r4    pos 1090, 20M: site 1100 is index 10, base T (alt) but base quality 10 < bqFilter -> no usable base
    add(TARGET, 1090, "r4", 0, 60, M(20), seq_with(20, "C", {10: "T"}), [10 if i == 10 else 35 for i in range(20)], rg)
    # r3    pos 1095, 10M: site 1100 is index 5, base N -> neither allele -> no usable base
    add(TARGET, 1095, "r3", 0, 60, M(10), seq_with(10, "C", {5: "N"}), [35] * 10, rg)
    # r5    pos 1098, 5M: site 1100 is index 2, base T (alt); base quality 40, MAPQ 25 -> q = 25
    add(TARGET, 1098, "r5", 0, 25, M(5), seq_with(5, "C", {2: "T"}), [40] * 5, rg)
    # long  pos 2000, CIGAR (1M1D) x 35 000 = 70 000 operations: read index i sits on 2000 + 2 i, odd offsets are deleted.
    #       site 3000 = offset 1000 -> index 500 (C = alt); site 3001 = offset 1001 -> deleted (no base); site 60000 = offset 58 000
    #       -> index 29 000 (G = alt); quality 20, MAPQ 60 -> q = 20.  Fixed fields: n_cigar_op = 2 (35000S70000N), CIGAR in CG:B,I
    n_long = 35000
    ops = [(1, 0), (1, 2)] * n_long
    cg = aux_B("CG", "I", [n << 4 | op for n, op in ops])
    add(TARGET, 2000, "long", 0, 60, [(n_long, 4), (2 * n_long, 3)], seq_with(n_long, "A", {500: "C", 29000: "G"}), [20] * n_long,
        rg + aux_B("ZB", "C", [9, 9]) + cg + aux_i("NM", n_long), ref_len=2 * n_long)
    # ---- chrX (reference 22) after the target, and unplaced unmapped reads
    add(22, 1000, "x1", 0, 60, M(30), "G" * 30, [30] * 30, rg)
    add(-1, 0, "u1", 77, 0, [], "ACGT", [2] * 4, rg, next_ref=-1, pnext=0)
    add(-1, 0, "u1", 141, 0, [], "TTTT", [2] * 4, rg, next_ref=-1, pnext=0)

    spans = []
    for r, rid, rb, re_ in recs:
        spans.append((len(stream), len(stream) + len(r), rb, re_, rid))
        stream += r
    # BGZF: small blocks up to the long record (short records straddle them), 64 KiB blocks from there on
    long_at = next(ub for (ub, ue, rb, re_, rid) in spans if ue - ub > 100000)
    cuts, at = [], 0
    while at < len(stream):
        step = 200 if at < long_at else 0xff00
        if at < long_at: step = min(step, long_at - at) if long_at - at < 200 else 200
        cuts.append((at, min(at + step, len(stream))))
        at = cuts[-1][1]
    out, starts = b"", []
    for a, b in cuts:
        starts.append((a, len(out)))
        out += bgzf_member(stream[a:b])
    eof_at = len(out)
    out += bgzf_member(b"")
    open(os.path.join(HERE, "aligner_like.bam"), "wb").write(out)

    def voff(u):
        if u == len(stream):
            return eof_at << 16
        for (a, o), (a2, _) in zip(starts, starts[1:] + [(len(stream) + 1, 0)]):
            if a <= u < a2:
                return o << 16 | (u - a)
        raise AssertionError

    bai = b"BAI\1" + struct.pack("<i", len(CHROMS))
    for rid, (_, length) in enumerate(CHROMS):
        bins, n_intv = {}, (length + 16383) >> 14
        lin = [0] * n_intv
        for (ub, ue, rb, re_, r) in spans:
            if r != rid:
                continue
            bins.setdefault(reg2bin(rb, re_), []).append((voff(ub), voff(ue)))
            for iv in range(rb >> 14, min(((re_ - 1) >> 14) + 1, n_intv)):
                if lin[iv] == 0 or voff(ub) < lin[iv]:
                    lin[iv] = voff(ub)
        bai += struct.pack("<i", len(bins))
        for b in sorted(bins):
            bai += struct.pack("<Ii", b, len(bins[b])) + b"".join(struct.pack("<QQ", s, e) for s, e in bins[b])
        bai += struct.pack("<i", n_intv) + b"".join(struct.pack("<Q", v) for v in lin)
    open(os.path.join(HERE, "aligner_like.bam.bai"), "wb").write(bai)

    sites = dict(L=[1000, 1010, 1020, 1050, 1100, 3000, 3001, 60000], ref="AACGCAAA", alt="TCGTTCCG")
    expect = {
        # bqFilter 17, mates merged, soft clips not used; reads in file order of their first alignment
        "default": [
            dict(name="r1", u=[0, 1, 3], bq=[35, -35, -35]),   # 1000 T (alt), 1010 A (ref) | mate: 1050 G (ref)
            dict(name="r2", u=[2], bq=[-30]),                  # 1020 C (ref)
            dict(name="r5", u=[4], bq=[25]),                   # 1100 T (alt), q = min(40, MAPQ 25)
            dict(name="long", u=[5, 7], bq=[20, 20]),          # 3000 C (alt), 60000 G (alt); 3001 is deleted
        ],
        # alignments on chr20: 13 (r1, dup, sec, sup, qcf, mq0, um, r2, r1/2, r4, r3, r5, long); dropped by flag: dup, sec, sup, qcf, um = 5; by mapping quality: mq0 = 1; in the window with no
        # usable base: r3, r4 = 2; mate pairs merged: 1
        "stats_default": dict(seen=13, by_flags=5, low_mapq=1, no_base=2, mates_merged=1),
        "window_2500_70000": [dict(name="long", u=[5, 7], bq=[20, 20])],
    }
    json.dump(dict(chr="chr20", sites=sites, expect=expect, n_bgzf_blocks=len(starts) + 1, stream_bytes=len(stream),
                   long_record_bytes=max(ue - ub for (ub, ue, _, _, _) in spans), n_cigar_ops_long=len(ops)),
              open(os.path.join(HERE, "aligner_like.json"), "w"), indent=1)
    print(len(out), "bytes,", len(starts) + 1, "BGZF blocks; long record", max(ue - ub for (ub, ue, _, _, _) in spans), "bytes")


if __name__ == "__main__":
    main()
