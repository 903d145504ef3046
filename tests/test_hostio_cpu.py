"""Host-side formats either side of the hot path (include/quilt_amd_io.h): BAM -> sampleReads and
the VCF writer.  No device work: these run on CPU in the build container and on the GPU box alike.

  * pile-up round trip: sampleReads -> alignments (tests/bamutil.py writes the BAM from the SAM specification) -> loader
    -> the same sampleReads, bit for bit (functions.R:243-298)
  * the loader's filters: mapping quality, base quality capped by mapping quality, flags, insert size, soft clips,
    insertions / deletions / skips, mate merging, coverage cap, window, unsorted files, truncated files
  * VCF column strings against a line-by-line restatement of the R expressions (functions.R:1420-1459), the unimputed
    entry, INFO strings (writers.R:73-80), the BGZF framing (every block checked) and the body (writers.R:81-117)
"""
import gzip
import os

import numpy as np
import pytest

from tests import bamutil


@pytest.fixture(scope="module")
def sites():
    rng = np.random.default_rng(11)
    T = 400
    L = np.cumsum(rng.integers(3, 60, size=T)).astype(np.int32) + 1000
    alleles = [tuple(rng.choice(list("ACGT"), size=2, replace=False)) for _ in range(T)]
    ref = [a for a, _ in alleles]
    alt = [b for _, b in alleles]
    return L, ref, alt


def _random_sample(rng, T, n_reads, max_span=6):
    from tests.util import sample_from_arrays
    reads = []
    for _ in range(n_reads):
        n = int(rng.integers(1, max_span + 1))
        s = int(rng.integers(0, T - n))
        us = np.arange(s, s + n)
        keep = rng.random(n) < 0.8
        keep[0] = keep[-1] = True       # the alignment spans first .. last site; inner sites may be "other base"
        us = us[keep]
        bq = rng.integers(17, 41, size=len(us)) * rng.choice([-1, 1], size=len(us))
        reads.append((us, bq))
    cen = [u[(len(u) - 1) // 2] // 32 for u, _ in reads]
    order = np.argsort(cen, kind="stable")
    reads = [reads[i] for i in order]
    ptr = np.r_[0, np.cumsum([len(u) for u, _ in reads])]
    return sample_from_arrays(ptr, np.concatenate([u for u, _ in reads]), np.concatenate([b for _, b in reads]),
                              [u[(len(u) - 1) // 2] // 32 for u, _ in reads])


def _same(a, b):
    assert a.nReads == b.nReads
    assert np.array_equal(a.read_ptr, b.read_ptr) and np.array_equal(a.u, b.u)
    assert np.array_equal(a.bq, b.bq) and np.array_equal(a.wif, b.wif)


def test_pileup_round_trip(tmp_path, sites):
    from quilt_amd.io import loadBamAndConvert
    L, ref, alt = sites
    rng = np.random.default_rng(5)
    s = _random_sample(rng, len(L), 300)
    alns = bamutil.sample_to_alignments(s, L, ref, alt, rng)
    # the loader returns reads in file order within a grid; build the expectation in that order
    order = sorted(range(s.nReads), key=lambda r: (s.wif[r], [a["name"] for a in alns].index(f"r{r}")))
    from tests.util import sample_from_arrays
    exp = sample_from_arrays(np.r_[0, np.cumsum(np.diff(s.read_ptr)[order])],
                             np.concatenate([s.u[s.read_ptr[r]:s.read_ptr[r + 1]] for r in order]),
                             np.concatenate([s.bq[s.read_ptr[r]:s.read_ptr[r + 1]] for r in order]), s.wif[order])
    path = str(tmp_path / "a.bam")
    bamutil.write_bam(path, [("chr20", 10 ** 6), ("chr21", 10 ** 6)], alns)
    got, st = loadBamAndConvert(path, "chr20", L, ref, alt, downsampleToCov=0, return_stats=True)
    _same(got, exp)
    assert st["alignments_on_chr"] == 300 and st["used"] == 300 and st["removed_by_coverage_cap"] == 0
    # other chromosome: no reads; unknown chromosome: an error
    assert loadBamAndConvert(path, "chr21", L, ref, alt).nReads == 0
    from quilt_amd.native import QuiltAmdError
    with pytest.raises(QuiltAmdError):
        loadBamAndConvert(path, "chrX", L, ref, alt)
    with pytest.raises(QuiltAmdError):
        loadBamAndConvert(str(tmp_path / "missing.bam"), "chr20", L, ref, alt)
    # a truncated file is an error, not a shorter sample
    raw = open(path, "rb").read()
    open(str(tmp_path / "cut.bam"), "wb").write(raw[:len(raw) // 2])
    with pytest.raises(QuiltAmdError):
        loadBamAndConvert(str(tmp_path / "cut.bam"), "chr20", L, ref, alt)


def _aln(pos, seq, qual, cigar=None, name="x", mapq=60, flag=0, tlen=0):
    return dict(ref_id=0, pos=pos, name=name, mapq=mapq, flag=flag, cigar=cigar or [(len(seq), "M")], seq=seq,
                qual=qual if isinstance(qual, list) else [qual] * len(seq), tlen=tlen)


def test_loader_filters_and_cigar(tmp_path):
    from quilt_amd.io import loadBamAndConvert
    L = np.array([100, 105, 110, 120, 200], dtype=np.int32)
    ref, alt = list("AAAAA"), list("CCCCC")
    path = str(tmp_path / "f.bam")

    def load(alns, **kw):
        bamutil.write_bam(path, [("1", 10000)], alns, sorted_header=kw.pop("sorted_header", True))
        return loadBamAndConvert(path, "1", L, ref, alt, **{"downsampleToCov": 0, **kw})

    # read 98..112: sites 100 (alt C), 105 (ref A), 110 (G: neither allele, skipped)
    seq = list("T" * 15)
    seq[2], seq[7], seq[12] = "C", "A", "G"
    s = load([_aln(98, "".join(seq), 30)])
    assert s.nReads == 1 and s.u.tolist() == [0, 1] and s.bq.tolist() == [30, -30] and s.wif.tolist() == [0]
    # base quality below bqFilter drops the base; capped by mapping quality: mapq 20 makes q = 20
    q = [30] * 15
    q[7] = 10
    s = load([_aln(98, "".join(seq), q, mapq=20)])
    assert s.u.tolist() == [0] and s.bq.tolist() == [20]
    # mapping quality below bqFilter: read not used; flags: unmapped, secondary, QC fail, duplicate, supplementary
    assert load([_aln(98, "".join(seq), 30, mapq=16)]).nReads == 0
    for flag in (0x4, 0x100, 0x200, 0x400, 0x800):
        assert load([_aln(98, "".join(seq), 30, flag=flag)]).nReads == 0
    assert load([_aln(98, "".join(seq), 30, flag=0x10)]).nReads == 1           # reverse strand is fine
    # insert size
    assert load([_aln(98, "".join(seq), 30, tlen=-700)], iSizeUpperLimit=600).nReads == 0
    assert load([_aln(98, "".join(seq), 30, tlen=-700)]).nReads == 1
    # deletion: 5M 3D 7M starting at 98: reference 98..102, skip 103..105, then 106..112 -> site 105 is deleted
    cig = [(5, "M"), (3, "D"), (7, "M")]
    sq = list("T" * 12)
    sq[2] = "C"                    # ref 100
    sq[5 + (110 - 106)] = "C"      # ref 110
    s = load([_aln(98, "".join(sq), 30, cigar=cig)])
    assert s.u.tolist() == [0, 2] and s.bq.tolist() == [30, 30]
    # insertion: 3M 2I 10M at 98: query index of ref 105 is 3 + 2 + (105 - 101)
    cig = [(3, "M"), (2, "I"), (10, "M")]
    sq = list("T" * 15)
    sq[2] = "A"
    sq[3 + 2 + 4] = "C"
    s = load([_aln(98, "".join(sq), 30, cigar=cig)])
    assert s.u.tolist() == [0, 1] and s.bq.tolist() == [-30, 30]
    # spliced (N) and hard clip; soft clip: aligned part starts at 103 (3S), site 100 only with useSoftClippedBases
    cig = [(3, "S"), (10, "M"), (2, "H")]
    sq = list("T" * 13)
    sq[0] = "C"                    # soft-clipped base over ref 100
    sq[3 + 2] = "A"                # ref 105
    s = load([_aln(103, "".join(sq), 30, cigar=cig)])
    assert s.u.tolist() == [1] and s.bq.tolist() == [-30]
    s = load([_aln(103, "".join(sq), 30, cigar=cig)], useSoftClippedBases=True)
    assert s.u.tolist() == [0, 1] and s.bq.tolist() == [30, -30]
    cig = [(4, "M"), (90, "N"), (10, "M")]   # 98..101, skip 102..191, then 192..201
    sq = list("T" * 14)
    sq[2] = "C"
    sq[4 + (200 - 192)] = "A"
    s = load([_aln(98, "".join(sq), 30, cigar=cig)])
    assert s.u.tolist() == [0, 4] and s.bq.tolist() == [30, -30] and s.wif.tolist() == [0]
    # mates: two alignments of one template become one read, sites in order; without merging, two reads
    m1 = _aln(98, "TTCTT", 30, name="pair", flag=0x1 | 0x40, tlen=110)
    m2 = _aln(198, "TTATT", 25, name="pair", flag=0x1 | 0x80, tlen=-110)
    s = load([m1, m2])
    assert s.nReads == 1 and s.u.tolist() == [0, 4] and s.bq.tolist() == [30, -25]
    s, st = load([m1, m2], merge_mates=False, return_stats=True)
    assert s.nReads == 2 and st["mates_merged"] == 0
    # OVERLAPPING mates (short inserts: cfDNA): a site both mates cover is one observation -- agreeing calls keep the better
    # quality, disagreeing calls drop the site; a pair left without any site is no read
    o1 = _aln(98, "TTCTTTTATTTTC", [30] * 13, name="ov", flag=0x1 | 0x40, tlen=20)      # 100 alt q30, 105 ref q30, 110 alt q30
    sq, q = list("TTTTTTTATTTTA"), [25] * 13                                              # 105 ref q25, 110 REF q25 (conflict)
    o2 = _aln(98, "".join(sq), q, name="ov", flag=0x1 | 0x80, tlen=-20)
    s, st = load([o1, o2], return_stats=True)
    assert s.nReads == 1 and s.u.tolist() == [0, 1] and s.bq.tolist() == [30, -30] and st["mates_merged"] == 1
    o3 = _aln(98, "TTATT", 40, name="cf", flag=0x1 | 0x40, tlen=5)
    o4 = _aln(98, "TTCTT", 20, name="cf", flag=0x1 | 0x80, tlen=-5)
    assert load([o3, o4]).nReads == 0
    s = load([_aln(98, "TTCTT", 20, name="hq", flag=0x1 | 0x40, tlen=5), _aln(98, "TTCTT", 35, name="hq", flag=0x1 | 0x80, tlen=-5)])
    assert s.nReads == 1 and s.u.tolist() == [0] and s.bq.tolist() == [35]
    # a hard clip in front of the leading soft clip (2H3S10M)
    cig = [(2, "H"), (3, "S"), (10, "M")]
    sq = list("T" * 13)
    sq[0] = "C"
    sq[3 + 2] = "A"
    s = load([_aln(103, "".join(sq), 30, cigar=cig)], useSoftClippedBases=True)
    assert s.u.tolist() == [0, 1] and s.bq.tolist() == [30, -30]
    # window: alignments must overlap chrStart..chrEnd
    assert load([m1, m2], merge_mates=False, chrStart=150, chrEnd=400).u.tolist() == [4]
    assert load([m1, m2], merge_mates=False, chrStart=1, chrEnd=99).u.tolist() == [0]
    # an unsorted file is read to the end
    s = load([m2, m1], merge_mates=False, sorted_header=False)
    assert sorted(s.u.tolist()) == [0, 4]


def test_coverage_cap(tmp_path):
    from quilt_amd.io import loadBamAndConvert
    L = np.array([100, 200], dtype=np.int32)
    ref, alt = list("AA"), list("CC")
    alns = [_aln(98, "TTCTT", 30, name=f"a{i}") for i in range(50)] + [_aln(198, "TTATT", 30, name=f"b{i}") for i in range(10)]
    path = str(tmp_path / "c.bam")
    bamutil.write_bam(path, [("1", 10000)], alns)
    s, st = loadBamAndConvert(path, "1", L, ref, alt, downsampleToCov=30, return_stats=True)
    assert np.bincount(s.u, minlength=2).tolist() == [30, 10] and st["removed_by_coverage_cap"] == 20
    s2 = loadBamAndConvert(path, "1", L, ref, alt, downsampleToCov=30)
    assert np.array_equal(s.u, s2.u)                                 # same seed, same choice
    assert loadBamAndConvert(path, "1", L, ref, alt, downsampleToCov=0).nReads == 60


# ---------------------------------------------------------------------------------------------------------------------------
def _r_round(x, d):
    """R's round(x, d) for the values used here (sprintf rounds the exact binary value, as R >= 4.0 does)."""
    return float(f"{x:.{d}f}")


def _r_num(x):
    """R's as.character / paste0 of a number already rounded to a few decimals."""
    if x == int(x):
        return str(int(x))
    s = repr(float(x))
    if "e" in s:
        m, e = s.split("e")
        return f"{m.rstrip('0').rstrip('.')}e{int(e):+03d}"
    return s


def test_vcf_columns_follow_the_r_expressions():
    from quilt_amd.io import make_per_sample_vcf_col, make_per_sample_vcf_col_nipt, missing_entry
    rng = np.random.default_rng(3)
    T = 500
    gp = rng.dirichlet([0.3, 0.3, 0.3], size=T).T
    gp[:, :6] = np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [0.9, 0.1, 0], [0.05, 0.9, 0.05], [0.5, 0.5, 0]]).T
    hd = rng.random((T, 2))
    hd[:4] = [[0.5, 1.5], [0.49999, 0.5001], [0, 1], [2.5, 0.5]]   # round half to even: 0, 2 / 0, 1 / 0, 1 / 2, 0
    col = make_per_sample_vcf_col(gp, hd, output_gt_phased_genotypes=False).tolist()
    colp = make_per_sample_vcf_col(gp, hd, output_gt_phased_genotypes=True).tolist()
    assert len(col) == T
    for t in range(T):
        gt = "./."
        for i, name in enumerate(("0/0", "0/1", "1/1")):
            if gp[i, t] >= 0.9:
                gt = name
                break
        tail = (f":{gp[0, t]:.3f},{gp[1, t]:.3f},{gp[2, t]:.3f}:{gp[1, t] + 2 * gp[2, t]:.3f}:{hd[t, 0]:.3f},{hd[t, 1]:.3f}")
        assert col[t] == gt + tail
        # functions.R:1435-1439: paste0(round(hd1), "|", round(hd2), substring(col, 4, 100))
        assert colp[t] == f"{int(np.round(hd[t, 0]))}|{int(np.round(hd[t, 1]))}" + col[t][3:]
    assert col[0].startswith("0/0:1.000,0.000,0.000:0.000:") and col[3].startswith("0/0:") and col[5].startswith("./.:")
    assert colp[0].startswith("0|2:") and colp[3].startswith("2|0:")
    assert missing_entry() == "./.:.,.,.:.:.,."
    # NIPT: paste0(round(x, 3)) everywhere (functions.R:1446-1459)
    m = rng.dirichlet([0.3, 0.3, 0.3], size=T).T
    f = rng.dirichlet([0.3, 0.3, 0.3], size=T).T
    m[:, 0], f[:, 0] = [1, 0, 0], [0.5, 0.25, 0.25]
    m[:, 1] = [0.0004, 0.9991, 0.0005]
    h3 = rng.random((T, 3))
    mds, fds = m[1] + 2 * m[2], f[1] + 2 * f[2]
    got = make_per_sample_vcf_col_nipt(m, f, h3, mds, fds).tolist()
    for t in range(T):
        n = lambda x: _r_num(_r_round(x, 3))
        exp = (f"{int(np.round(h3[t, 0]))}|{int(np.round(h3[t, 1]))}|{int(np.round(h3[t, 2]))}:"
               f"{n(m[0, t])},{n(m[1, t])},{n(m[2, t])}:{n(mds[t])}:{n(f[0, t])},{n(f[1, t])},{n(f[2, t])}:{n(fds[t])}")
        assert got[t] == exp, (t, got[t], exp)
    assert got[0].split(":")[1:] == ["1,0,0", "0", "0.5,0.25,0.25", "0.75"]
    assert got[1].split(":")[1] in ("0,0.999,0", "0,0.999,0.001")


def test_hwe_exact_known_values():
    """Wigginton et al. 2005's worked example: 100 individuals, 21 rare alleles, 5 / 11 / 84 genotypes is far from
    equilibrium; equilibrium-like counts give p near 1; p is symmetric in the homozygote labels."""
    import ctypes as C
    from quilt_amd.io import _io_lib
    from quilt_amd.native import ptr
    counts = np.asfortranarray(np.array([[84, 11, 5], [5, 11, 84], [81, 18, 1], [0, 0, 0], [10, 0, 0], [25, 50, 25],
                                         [50, 0, 50]], dtype=np.float64))
    p = np.zeros(len(counts))
    assert _io_lib().qa_hwe_exact(C.c_int32(len(counts)), ptr(counts), ptr(p)) == 0
    assert p[0] == p[1] and p[0] < 1e-3
    assert p[2] > 0.99 and p[3] == 1 and p[4] == 1 and p[5] > 0.9 and p[6] < 1e-20
    # brute force: P(n_ab | n, n_a) = 2^n_ab n! n_a! n_b! / (n_aa! n_ab! n_bb! (2n)!)
    from math import lgamma, exp, log

    def brute(aa, ab, bb):
        n, na = aa + ab + bb, 2 * aa + ab
        nb = 2 * n - na
        pr = {}
        for het in range(na % 2, min(na, nb) + 1, 2):
            a, b = (na - het) // 2, (nb - het) // 2
            pr[het] = exp(het * log(2) + lgamma(n + 1) + lgamma(na + 1) + lgamma(nb + 1) - lgamma(a + 1) - lgamma(het + 1)
                          - lgamma(b + 1) - lgamma(2 * n + 1))
        return sum(v for v in pr.values() if v <= pr[ab] * (1 + 1e-9))
    for i, (aa, ab, bb) in enumerate(counts.astype(int)):
        if aa + ab + bb:
            assert abs(p[i] - min(1.0, brute(aa, ab, bb))) < 1e-9


def test_write_vcf_file(tmp_path):
    from quilt_amd.io import (SummaryCounts, make_and_write_output_file, make_per_sample_vcf_col, per_sample_counts)
    from tests.util import sample_from_arrays
    rng = np.random.default_rng(9)
    T, N = 3000, 5
    pos_bp = (np.cumsum(rng.integers(1, 300, size=T)) + 5_000_000).astype(np.int32)
    ref, alt = list(rng.choice(list("AC"), size=T)), list(rng.choice(list("GT"), size=T))
    counts = SummaryCounts(T)
    cols = []
    for i in range(N):
        gp = rng.dirichlet([0.2, 0.2, 0.2], size=T).T
        hd = rng.random((T, 2))
        n = 2000
        s = sample_from_arrays(np.arange(n + 1), rng.integers(0, T, size=n), rng.integers(17, 40, size=n) * rng.choice([-1, 1], size=n),
                               np.zeros(n))
        if i == 3:
            cols.append(None)      # fewer than minimum_number_of_sample_reads reads: not imputed, not counted
            continue
        counts.add_sample(*per_sample_counts(gp, s, T))
        cols.append(make_per_sample_vcf_col(gp, hd))
    keep = np.ones(T, dtype=bool)
    keep[:100] = keep[-50:] = False
    names = [f"s{i}" for i in range(N)]
    out = str(tmp_path / "quilt.chr20.vcf.gz")
    fin = make_and_write_output_file(out, names, "chr20", pos_bp, ref, alt, cols, counts, inRegion2=keep)
    raw = open(out, "rb").read()
    text = bamutil.bgzf_decompress(raw).decode()       # every block: header, BSIZE, CRC, ISIZE
    assert raw.endswith(bamutil.bgzf_block(b"")) and gzip.decompress(raw).decode() == text
    lines = text.rstrip("\n").split("\n")
    meta = [l for l in lines if l.startswith("##")]
    assert meta[0] == "##fileformat=VCFv4.0" and sum(l.startswith("##INFO=") for l in meta) == 6
    assert sum(l.startswith("##FORMAT=") for l in meta) == 4
    head = lines[len(meta)].split("\t")
    assert head[:9] == ["#CHROM", "POS", "ID", "REF", "ALT", "QUAL", "FILTER", "INFO", "FORMAT"] and head[9:] == names
    body = [l.split("\t") for l in lines[len(meta) + 1:]]
    assert len(body) == int(keep.sum())
    idx = np.nonzero(keep)[0]
    for row, t in list(zip(body, idx))[::97]:
        assert row[0] == "chr20" and int(row[1]) == pos_bp[t] and row[2] == "." and row[3] == ref[t] and row[4] == alt[t]
        assert row[5] == "." and row[6] == "PASS" and row[8] == "GT:GP:DS:HD" and len(row) == 9 + N
        assert row[9 + 3] == "./.:.,.,.:.:.,."
        assert row[9] == cols[0][int(t)]
        info = dict(kv.split("=") for kv in row[7].split(";"))
        assert list(info) == ["EAF", "INFO_SCORE", "HWE", "ERC", "EAC", "PAF"]
        assert abs(float(info["EAF"]) - fin["estimatedAlleleFrequency"][t]) <= 5.1e-6
        assert abs(float(info["INFO_SCORE"]) - fin["info"][t]) <= 5.1e-6
        assert info["HWE"] == "%.2e" % fin["hwe"][t]
        assert abs(float(info["ERC"]) - fin["alleleCount"][t, 0]) <= 5.1e-6
        assert abs(float(info["EAC"]) - (fin["alleleCount"][t, 1] - fin["alleleCount"][t, 0])) <= 5.1e-6
    # plain-text output when the name does not end in .gz
    out2 = str(tmp_path / "quilt.vcf")
    make_and_write_output_file(out2, names, "chr20", pos_bp, ref, alt, cols, counts, inRegion2=keep)
    assert open(out2).read() == text
    # counts survive the flat-vector form used for the cross-rank sum
    c2 = SummaryCounts(T).from_vector(counts.as_vector())
    assert np.array_equal(c2.hweCount, counts.hweCount) and np.array_equal(c2.alleleCount, counts.alleleCount)


def test_info_number_format():
    """round(x, 5) pasted by R: shortest form, scientific below 1e-4 (3e-05), NaN for 0 / 0 pileups."""
    import ctypes as C
    from quilt_amd.io import _io_lib, _two_pass
    from quilt_amd.native import ptr
    eaf = np.array([0.5, 0.123456789, 0.00003, 0.0, 1.0, 0.1])
    info = np.array([1.0, 0.999996, 0.0, 0.25, 0.5, 0.75])
    hwe = np.array([1.0, 0.05, 1e-12, 0.5, 0.123, 3.3e-5])
    ac = np.asfortranarray(np.array([[1.5, 3.0, 0.5], [0, 0, np.nan], [12.123456, 20.5, 0.59139], [1, 2, 0.5], [0, 4, 0], [2, 2, 1]]))
    col = _two_pass(_io_lib().qa_vcf_info_column, 6, ptr(eaf), ptr(info), ptr(hwe), ptr(ac)).tolist()
    assert col[0] == "EAF=0.5;INFO_SCORE=1;HWE=1.00e+00;ERC=1.5;EAC=1.5;PAF=0.5"
    assert col[1] == "EAF=0.12346;INFO_SCORE=1;HWE=5.00e-02;ERC=0;EAC=0;PAF=NaN"
    assert col[2] == "EAF=3e-05;INFO_SCORE=0;HWE=1.00e-12;ERC=12.12346;EAC=8.37654;PAF=0.59139"
    assert col[5] == "EAF=0.1;INFO_SCORE=0.75;HWE=3.30e-05;ERC=2;EAC=0;PAF=1"


def test_accumulate_dosage_equals_the_r_expressions():
    """functions.R:999-1020 in one native pass: bit-identical to the expressions written out with numpy, chain by chain."""
    from quilt_amd.io import accumulate_dosage
    rng = np.random.default_rng(2)
    for n_label in (2, 3):
        n_chain, n_sample, T = 9, 3, 1000
        hap = rng.random((n_chain, n_label, T))
        cs = np.repeat(np.arange(n_sample), 3)
        d, g = rng.random((n_sample, T)), rng.random((n_sample, 3, T))
        fd, fg = (rng.random((n_sample, T)), rng.random((n_sample, 3, T))) if n_label == 3 else (None, None)
        exp = [x.copy() if x is not None else None for x in (d, g, fd, fg)]
        for c in range(n_chain):
            h1, h2 = hap[c, 0], hap[c, 1]
            exp[0][cs[c]] += h1 + h2
            exp[1][cs[c]] += np.stack([(1 - h1) * (1 - h2), (1 - h1) * h2 + h1 * (1 - h2), h1 * h2])
            if n_label == 3:
                h3 = hap[c, 2]
                exp[2][cs[c]] += h1 + h3
                exp[3][cs[c]] += np.stack([(1 - h1) * (1 - h3), (1 - h1) * h3 + h1 * (1 - h3), h1 * h3])
        accumulate_dosage(hap, cs, d, g, fd, fg)
        assert np.array_equal(d, exp[0]) and np.array_equal(g, exp[1])
        if n_label == 3:
            assert np.array_equal(fd, exp[2]) and np.array_equal(fg, exp[3])


def test_bai_linear_index_is_used(tmp_path):
    """A coordinate-sorted BAM with a .bai next to it: the loader starts at the linear index's offset for the window (SAM spec
    5.1.3) instead of scanning from the top -- the same reads, fewer alignments looked at -- including an alignment that
    starts before the window and reaches into it."""
    from quilt_amd.io import loadBamAndConvert
    rng = np.random.default_rng(13)
    L = (np.arange(400) * 500 + 1000).astype(np.int32)          # sites every 500 bp up to 200 kb
    ref, alt = ["A"] * 400, ["C"] * 400
    alns = [dict(ref_id=0, pos=int(p), name=f"o{i}", mapq=60, flag=0, cigar=[(50, "M")], seq="T" * 50, qual=[30] * 50)
            for i, p in enumerate(np.sort(rng.integers(1, 150000, size=300)))]          # another chromosome first
    pos = np.sort(rng.integers(1, 199000, size=800))
    for i, p in enumerate(pos):
        n = 700 if i % 97 == 0 else 120                            # a few long alignments spanning a 16 kb boundary region
        seq = "".join(rng.choice(list("AC"), size=n))
        alns.append(dict(ref_id=1, pos=int(p), name=f"r{i}", mapq=60, flag=0, cigar=[(n, "M")], seq=seq, qual=[30] * n))
    long_one = dict(ref_id=1, pos=98000, name="span", mapq=60, flag=0, cigar=[(3000, "M")], seq="C" * 3000, qual=[30] * 3000)
    alns.append(long_one)
    alns.sort(key=lambda a: (a["ref_id"], a["pos"]))
    plain, indexed = str(tmp_path / "plain.bam"), str(tmp_path / "indexed.bam")
    bamutil.write_bam(plain, [("chr19", 10 ** 6), ("chr20", 10 ** 6)], alns, block=4096)
    bamutil.write_bam(indexed, [("chr19", 10 ** 6), ("chr20", 10 ** 6)], alns, index=True, block=4096)
    assert os.path.exists(indexed + ".bai")
    for start, end in ((100000, 140000), (1, 30000), (163841, 200000), (0, 0)):
        a, sa = loadBamAndConvert(plain, "chr20", L, ref, alt, chrStart=start, chrEnd=end, downsampleToCov=0, return_stats=True)
        b, sb = loadBamAndConvert(indexed, "chr20", L, ref, alt, chrStart=start, chrEnd=end, downsampleToCov=0, return_stats=True)
        _same(a, b)
        assert a.nReads > 0
        if start > 40000:
            assert sb["alignments_on_chr"] < sa["alignments_on_chr"]        # the index skipped the alignments before the window
    # the alignment that starts at 98 000 and reaches 101 000 is found for a window starting at 100 000
    a, _ = loadBamAndConvert(indexed, "chr20", L, ref, alt, chrStart=100000, chrEnd=100600, downsampleToCov=0, return_stats=True)
    assert any((a.bq[a.read_ptr[r]:a.read_ptr[r + 1]] > 0).all() and a.read_ptr[r + 1] - a.read_ptr[r] >= 5 for r in range(a.nReads))


def test_synthetic_bam_generator_round_trip(tmp_path):
    """quilt_amd/synth.py::write_synthetic_bam (the input-side generator behind `bench.py --bam`) against the native loader:
    the pile-up gives back the sample's reads (as a multiset: the file is coordinate-sorted), and the file is well-formed BGZF."""
    from quilt_amd.io import loadBamAndConvert
    from quilt_amd.synth import make_synthetic_panel, make_synthetic_sample, synthetic_alleles, write_synthetic_bam
    panel = make_synthetic_panel(K=300, nSNPs=3200, seed=3)
    ref, alt = synthetic_alleles(panel.nSNPs, 1)
    assert all(a != b for a, b in zip(ref, alt))
    for mode, n in (("short", 1500), ("ont", 40)):
        s = make_synthetic_sample(panel, seed=5, n_reads=n, mode=mode)
        path = str(tmp_path / f"{mode}.bam")
        write_synthetic_bam(path, s, panel.L, ref, alt, seed=2)
        assert bamutil.bgzf_decompress(open(path, "rb").read())[:4] == b"BAM\x01"
        # (the ONT-style synthetic reads carry base qualities 5-15: below the default bqFilter = 17 the loader drops them all)
        assert mode != "ont" or loadBamAndConvert(path, "chr20", panel.L, ref, alt, panel.grid).nReads == 0
        g = loadBamAndConvert(path, "chr20", panel.L, ref, alt, panel.grid, downsampleToCov=0, bqFilter=1)
        key = lambda x: sorted((tuple(x.u[x.read_ptr[r]:x.read_ptr[r + 1]]), tuple(x.bq[x.read_ptr[r]:x.read_ptr[r + 1]]))
                               for r in range(x.nReads))
        assert g.nReads == s.nReads and key(g) == key(s) and np.array_equal(g.wif, np.sort(s.wif))


def _reads_of(s):
    return [dict(u=s.u[s.read_ptr[r]:s.read_ptr[r + 1]].tolist(), bq=s.bq[s.read_ptr[r]:s.read_ptr[r + 1]].tolist())
            for r in range(s.nReads)]


def test_sam_spec_example_bam_from_an_independent_encoder():
    """tests/golden/sam_spec_example.bam: the example alignment of the SAM specification (section 1.1), encoded byte by byte
    from the spec's tables by tests/golden/make_sam_spec_bam.py -- not by tests/bamutil.py -- with records straddling 96-byte
    BGZF blocks, I / D / N / P / S / H operations, a supplementary line, auxiliary fields and a multi-bin BAI.  The expected
    pile-up was worked out by hand from the spec's picture (comments in the generator)."""
    import gzip
    import json
    from quilt_amd.io import loadBamAndConvert
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    bam = os.path.join(gold, "sam_spec_example.bam")
    z = json.load(open(os.path.join(gold, "sam_spec_example.json")))
    raw = gzip.open(bam, "rb").read()        # python's gzip reads the BGZF members as one multi-member stream
    assert raw[:4] == b"BAM\x01" and len(raw) == z["stream_bytes"] and z["n_bgzf_blocks"] > 6
    L, ref, alt = np.array(z["sites"]["L"], dtype=np.int32), list(z["sites"]["ref"]), list(z["sites"]["alt"])
    want = lambda key: [dict(u=e["u"], bq=e["bq"]) for e in z["expect"][key]]
    s, st = loadBamAndConvert(bam, "ref", L, ref, alt, downsampleToCov=0, return_stats=True)
    assert _reads_of(s) == want("default")
    e = z["expect"]["stats_default"]
    assert (st["alignments_on_chr"], st["used"], st["flagged"], st["mates_merged"]) == (e["seen"], e["used"], e["by_flags"], e["mates_merged"])
    assert _reads_of(loadBamAndConvert(bam, "ref", L, ref, alt, downsampleToCov=0, useSoftClippedBases=True)) == want("soft_clips")
    assert _reads_of(loadBamAndConvert(bam, "ref", L, ref, alt, downsampleToCov=0, chrStart=30, chrEnd=45)) == want("window_30_45")
    # the long reference: whole, and windows entered through the index (bins + linear index) -- the same reads as a scan
    b = z["expect"]["big"]
    Lb = np.array(b["L"], dtype=np.int32)
    assert _reads_of(loadBamAndConvert(bam, "big", Lb, list(b["ref"]), list(b["alt"]), downsampleToCov=0)) == b["whole"]
    for lo, hi, key in ((39000, 41000, "window_39000_41000"), (16384, 17000, "window_16384_17000")):
        got = loadBamAndConvert(bam, "big", Lb, list(b["ref"]), list(b["alt"]), downsampleToCov=0, chrStart=lo, chrEnd=hi)
        assert _reads_of(got) == b[key], key


def test_cram_is_refused_with_a_recipe(tmp_path):
    """cramlist (quilt.R:106-108): CRAM is not decoded; the loader says so and how to convert (no silent empty sample)."""
    from quilt_amd.io import loadBamAndConvert
    from quilt_amd.native import QuiltAmdError
    p = str(tmp_path / "x.cram")
    open(p, "wb").write(b"CRAM\x03\x00" + b"\0" * 64)
    with pytest.raises(QuiltAmdError) as ei:
        loadBamAndConvert(p, "chr20", np.array([10, 20], dtype=np.int32), list("AA"), list("CC"))
    assert "samtools view -b" in str(ei.value) and "status -3" in str(ei.value)


def test_vcf_columns_reject_values_they_cannot_print():
    """The column writers format into fixed entries: a non-finite dosage or posterior (or one beyond 1e9) is refused with the SNP
    named instead of being formatted past the entry."""
    import ctypes as C
    from quilt_amd.native import ptr
    from quilt_amd.io import _io_lib
    lib = _io_lib()
    lib.qa_last_error.restype = C.c_char_p
    T = 4
    gp = np.tile(np.array([0.25, 0.5, 0.25]), T)
    hd = np.full(2 * T, 0.5)
    buf = np.zeros(4096, dtype=np.uint8)
    off = np.zeros(T + 1, dtype=np.int64)
    need = C.c_int64()
    def call(gp_, hd_):
        return lib.qa_vcf_column_diploid(C.c_int32(T), ptr(gp_), ptr(hd_), C.c_int32(1), ptr(buf), C.c_int64(len(buf)), ptr(off),
                                         C.byref(need))
    assert call(gp, hd) == 0
    for bad in (np.nan, np.inf, 1e300):
        h = hd.copy(); h[2] = bad
        assert call(gp, h) == -2 and b"SNP 2" in lib.qa_last_error()
        g = gp.copy(); g[3 * 1 + 1] = bad
        assert call(g, hd) == -2 and b"SNP 1" in lib.qa_last_error()


def test_aligner_like_bam_fixture():
    """tests/golden/aligner_like.bam (make_aligner_like_bam.py, byte by byte from the SAM specification): what bwa mem + samtools
    markdup and a long-read aligner emit and the other fixtures lack -- a 25-reference header with @RG / @PG, the target as
    reference 19 between alignments on other chromosomes and unplaced unmapped reads, a placed unmapped mate, secondary /
    supplementary / duplicate / QC-fail flags, MAPQ 0, `B`-array auxiliary fields ahead of others, a 70 000-operation CIGAR in
    CG:B,I behind the <l_seq>S<ref_len>N placeholder (a 332 kB record across five BGZF blocks), an N and a low-quality base on
    SNPs.  The expected pile-up is hand-derived in the generator.  BASELINE configs[0]'s real BAM cannot be had here (no data in
    the image): this is its stand-in for the FORMAT; the numbers of configs[0] stay untested (README)."""
    import gzip
    import json
    from quilt_amd.io import loadBamAndConvert
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    bam = os.path.join(gold, "aligner_like.bam")
    z = json.load(open(os.path.join(gold, "aligner_like.json")))
    raw = gzip.open(bam, "rb").read()
    assert raw[:4] == b"BAM\x01" and len(raw) == z["stream_bytes"] and z["n_cigar_ops_long"] > 65535 and z["long_record_bytes"] > 4 * 65280
    L, ref, alt = np.array(z["sites"]["L"], dtype=np.int32), list(z["sites"]["ref"]), list(z["sites"]["alt"])
    want = lambda key: [dict(u=e["u"], bq=e["bq"]) for e in z["expect"][key]]
    s, st = loadBamAndConvert(bam, z["chr"], L, ref, alt, downsampleToCov=0, return_stats=True)
    assert _reads_of(s) == want("default")
    e = z["expect"]["stats_default"]
    assert (st["alignments_on_chr"], st["flagged"], st["low_mapq"], st["no_site"], st["mates_merged"]) == \
        (e["seen"], e["by_flags"], e["low_mapq"], e["no_base"], e["mates_merged"])
    # through the index (bins + linear index of reference 19) into a window that only the long read overlaps
    got = loadBamAndConvert(bam, z["chr"], L, ref, alt, downsampleToCov=0, chrStart=2500, chrEnd=70000)
    assert _reads_of(got) == want("window_2500_70000")
    # and without the index (a copy without its .bai): the sequential scan gives the same reads
    import shutil, tempfile
    with tempfile.TemporaryDirectory() as d:
        shutil.copy(bam, os.path.join(d, "x.bam"))
        assert _reads_of(loadBamAndConvert(os.path.join(d, "x.bam"), z["chr"], L, ref, alt, downsampleToCov=0)) == want("default")


def test_fixed_three_decimals_equal_printf_on_ties_and_their_neighbours():
    """The diploid column's six numbers per SNP are written by an integer formatter (csrc/hostio.cpp fmt_fixed3: the decimal
    expansion of a double is exact, so "%.3f" is the mantissa times 1000 rounded on its exact remainder, ties to even) instead of
    snprintf.  It must print what printf prints -- on random values, on every multiple of 1 / 8000 (exact ties at the fourth
    decimal), on the doubles either side of every odd multiple of 0.0005 (decimal ties that are NOT ties in binary), on
    subnormals and on large values."""
    import ctypes as C
    from quilt_amd.io import _io_lib
    from quilt_amd.native import ptr
    lib = _io_lib()
    rng = np.random.default_rng(0)
    t = np.arange(1, 4000, 2) / 2000.0
    vals = np.concatenate([rng.random(30000), rng.random(15000) * 2, np.arange(0, 4001) / 8000.0, np.arange(0, 2001) / 2000.0,
                           [0.0, 1.0, 2.0, 0.0005, 0.0015, 0.0025, 0.00049999999999999, 0.9995, 1.9995, 1e-320, 0.5, 0.0625, 0.1875,
                            1.0005, 1e8, 123456.7895, 0.0004999999999999999], np.nextafter(t, 0), np.nextafter(t, 10), t])
    n = len(vals) // 3
    gp = np.ascontiguousarray(vals[:3 * n])           # 3 x n column-major: entry i holds gp[3 i .. 3 i + 2]
    hd = np.concatenate([vals[:n], vals[n:2 * n]])
    buf, off, need = np.zeros(200 * n, dtype=np.uint8), np.zeros(n + 1, dtype=np.int64), C.c_int64()
    assert lib.qa_vcf_column_diploid(C.c_int32(n), ptr(gp), ptr(hd), C.c_int32(0), ptr(buf), C.c_int64(len(buf)), ptr(off), C.byref(need)) == 0
    txt = bytes(buf[:off[-1]]).decode().split("\0")[:-1]
    for i in range(n):
        g0, g1, g2 = gp[3 * i:3 * i + 3]
        assert txt[i][3:] == ":%.3f,%.3f,%.3f:%.3f:%.3f,%.3f" % (g0, g1, g2, g1 + 2 * g2, hd[i], hd[n + i]), i
