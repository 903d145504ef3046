// hostio.cpp -- host-side data formats either side of the hot path (include/quilt_amd_io.h):
// BGZF/BAM -> flattened sampleReads, and sampleReads' way out: per-sample VCF columns, INFO strings, BGZF-framed body.
// Plain C++ on the feeding thread; nothing here touches the device.
#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/quilt_amd.h"
#include "../../include/quilt_amd_io.h"

namespace qa { void set_error(const char *fmt, ...); }   // panel.hip: the library's last-error text (qa_last_error)

namespace {

// ---------------------------------------------------------------------------------------------------------------------
// BGZF: a series of gzip members, each <= 64 KiB, with the compressed size in a 'BC' extra field (SAM spec 4.1)
// ---------------------------------------------------------------------------------------------------------------------
struct BgzfReader {
    FILE *f = nullptr;
    std::vector<uint8_t> in, out;
    size_t pos = 0;      // read cursor in out
    bool eof = false, bad = false;

    explicit BgzfReader(const char *path) : f(fopen(path, "rb")) { if (!f) bad = true; }
    ~BgzfReader() { if (f) fclose(f); }

    bool next_block() {
        uint8_t h[18];
        size_t n = fread(h, 1, 12, f);
        if (n == 0) { eof = true; return false; }
        if (n != 12 || h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) { bad = true; return false; }
        int xlen = h[10] | (h[11] << 8);
        std::vector<uint8_t> extra(xlen);
        if (fread(extra.data(), 1, xlen, f) != (size_t)xlen) { bad = true; return false; }
        int bsize = -1;
        for (int i = 0; i + 4 <= xlen;) {
            int slen = extra[i + 2] | (extra[i + 3] << 8);
            if (extra[i] == 'B' && extra[i + 1] == 'C' && slen == 2 && i + 6 <= xlen) bsize = extra[i + 4] | (extra[i + 5] << 8);
            i += 4 + slen;
        }
        if (bsize < 0) { bad = true; return false; }
        int clen = bsize - xlen - 19;   // deflate bytes; then CRC32 and ISIZE
        if (clen < 0) { bad = true; return false; }
        in.resize((size_t)clen + 8);
        if (fread(in.data(), 1, in.size(), f) != in.size()) { bad = true; return false; }
        uint32_t crc, isize;
        memcpy(&crc, in.data() + clen, 4);
        memcpy(&isize, in.data() + clen + 4, 4);
        if (isize > (1u << 16)) { bad = true; return false; }
        out.resize(isize);
        pos = 0;
        if (isize == 0) return true;   // the end-of-file marker (or an empty block): keep going
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        if (inflateInit2(&zs, -15) != Z_OK) { bad = true; return false; }
        zs.next_in = in.data();
        zs.avail_in = (uInt)clen;
        zs.next_out = out.data();
        zs.avail_out = isize;
        int rc = inflate(&zs, Z_FINISH);
        inflateEnd(&zs);
        if (rc != Z_STREAM_END || zs.total_out != isize || crc32(crc32(0L, Z_NULL, 0), out.data(), isize) != crc) {
            bad = true;
            return false;
        }
        return true;
    }
    // position at a BGZF virtual file offset (compressed block start << 16 | offset inside the inflated block)
    bool seek_virtual(uint64_t voff) {
        if (fseeko(f, (off_t)(voff >> 16), SEEK_SET) != 0) { bad = true; return false; }
        out.clear();
        pos = 0;
        eof = false;
        if (!next_block()) return false;
        const size_t u = (size_t)(voff & 0xffff);
        if (u > out.size()) { bad = true; return false; }
        pos = u;
        return true;
    }
    // exactly n bytes, or false at a clean end of file before the first byte (eof) / on a truncated stream (bad)
    bool read(void *dst, size_t n) {
        uint8_t *d = static_cast<uint8_t *>(dst);
        size_t got = 0;
        while (got < n) {
            if (pos == out.size()) {
                if (!next_block()) { if (got > 0) bad = true; return false; }
                continue;
            }
            size_t m = std::min(n - got, out.size() - pos);
            memcpy(d + got, out.data() + pos, m);
            pos += m;
            got += m;
        }
        return true;
    }
};

struct BgzfWriter {
    FILE *f = nullptr;
    bool framed;
    std::vector<uint8_t> buf;
    static constexpr size_t kBlock = 0xff00;   // bgzip's uncompressed block size
    BgzfWriter(const char *path, bool framed_, bool truncate) : f(fopen(path, truncate ? "wb" : "ab")), framed(framed_) {}
    ~BgzfWriter() { if (f) fclose(f); }
    bool ok() const { return f != nullptr; }
    bool flush_block(const uint8_t *p, size_t n) {
        uint8_t comp[1 << 16];
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        if (deflateInit2(&zs, 6, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
        zs.next_in = const_cast<uint8_t *>(p);
        zs.avail_in = (uInt)n;
        zs.next_out = comp;
        zs.avail_out = sizeof comp;
        int rc = deflate(&zs, Z_FINISH);
        size_t clen = zs.total_out;
        deflateEnd(&zs);
        if (rc != Z_STREAM_END) return false;
        uint32_t bsize = (uint32_t)(clen + 25);   // header 18 + data + 8, stored minus one
        uint8_t h[18] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0,
                         (uint8_t)(bsize & 0xff), (uint8_t)(bsize >> 8)};
        uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), p, (uInt)n), isize = (uint32_t)n;
        return fwrite(h, 1, 18, f) == 18 && fwrite(comp, 1, clen, f) == clen && fwrite(&crc, 4, 1, f) == 1 &&
               fwrite(&isize, 4, 1, f) == 1;
    }
    bool write(const char *p, size_t n) {
        if (!framed) return fwrite(p, 1, n, f) == n;
        buf.insert(buf.end(), p, p + n);
        size_t done = 0;
        while (buf.size() - done >= kBlock) {
            if (!flush_block(buf.data() + done, kBlock)) return false;
            done += kBlock;
        }
        buf.erase(buf.begin(), buf.begin() + done);
        return true;
    }
    bool finish(bool eof_marker) {
        if (!framed) return fflush(f) == 0;
        if (!buf.empty() && !flush_block(buf.data(), buf.size())) return false;
        buf.clear();
        if (eof_marker && !flush_block(nullptr, 0)) return false;
        return fflush(f) == 0;
    }
};

inline uint64_t stream_key(uint64_t seed, uint64_t i) {   // the library's counter stream (gibbs_dev.hpp stream_uniform)
    uint64_t z = seed + (i + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// Where to start reading for alignments that overlap [beg0, end0) (0-based, half open) of reference `ref`, from the BAI
// index beside the file (SAM spec 5.2): 0 when there is no usable index (scan from the top).
//   * linear index: ioffset[beg0 >> 14] = the smallest offset of an alignment overlapping that 16 kb interval -- a lower
//     bound of everything the window needs (an interval no alignment overlaps holds 0 in some writers: the next one that
//     does is taken);
//   * bin index: the chunks of the bins that can hold an overlapping alignment (reg2bins, spec 5.3).  Every alignment the
//     window needs lies in one of those chunks, so nothing before the first such chunk that ends after the linear bound is
//     needed: in a whole-genome file whose window starts in the tail of a 16 kb interval this skips the alignments of
//     larger bins' unrelated chunks the linear bound would start in.
// The scan is sequential from the returned offset and filters by position itself, so any valid lower bound is correct.
uint64_t bai_start_offset(const char *bam_path, int32_t ref, int32_t beg0, int32_t end0) {
    std::string p1 = std::string(bam_path) + ".bai", p2 = bam_path;
    if (p2.size() > 4 && p2.compare(p2.size() - 4, 4, ".bam") == 0) p2 = p2.substr(0, p2.size() - 4) + ".bai"; else p2.clear();
    FILE *f = fopen(p1.c_str(), "rb");
    if (!f && !p2.empty()) f = fopen(p2.c_str(), "rb");
    if (!f) return 0;
    uint64_t linear = 0;
    std::vector<std::pair<uint64_t, uint64_t>> chunks;   // (begin, end) of the chunks of overlapping bins
    auto rd = [&](void *d, size_t n) { return fread(d, 1, n, f) == n; };
    auto overlaps = [&](uint32_t bin) {   // does bin cover part of [beg0, end0)?  (bins 0; 1-8; 9-72; 73-584; 585-4680; 4681-37448)
        static const uint32_t first[6] = {0, 1, 9, 73, 585, 4681};
        static const int shift[6] = {29, 26, 23, 20, 17, 14};
        for (int l = 5; l >= 0; l--) {
            if (bin < first[l]) continue;
            const int64_t lo = (int64_t)(bin - first[l]) << shift[l], hi = lo + ((int64_t)1 << shift[l]);
            return lo < (int64_t)end0 && hi > (int64_t)beg0;
        }
        return false;
    };
    char magic[4];
    int32_t n_ref = 0;
    bool ok = rd(magic, 4) && memcmp(magic, "BAI\1", 4) == 0 && rd(&n_ref, 4) && ref < n_ref;
    for (int32_t r = 0; ok && r <= ref; r++) {
        int32_t n_bin = 0, n_intv = 0;
        ok = rd(&n_bin, 4) && n_bin >= 0;
        for (int32_t b = 0; ok && b < n_bin; b++) {
            uint32_t bin;
            int32_t n_chunk = 0;
            ok = rd(&bin, 4) && rd(&n_chunk, 4) && n_chunk >= 0;
            if (!ok) break;
            if (r == ref && bin < 37449 && overlaps(bin)) {   // (bin 37450 is samtools' pseudo-bin of counts)
                for (int32_t c = 0; ok && c < n_chunk; c++) {
                    uint64_t v[2];
                    ok = rd(v, 16);
                    if (ok) chunks.emplace_back(v[0], v[1]);
                }
            } else {
                ok = fseeko(f, (off_t)n_chunk * 16, SEEK_CUR) == 0;
            }
        }
        ok = ok && rd(&n_intv, 4) && n_intv >= 0;
        if (!ok) break;
        if (r < ref) { ok = fseeko(f, (off_t)n_intv * 8, SEEK_CUR) == 0; continue; }
        const int32_t first = std::min<int32_t>(beg0 >> 14, n_intv);
        if (fseeko(f, (off_t)first * 8, SEEK_CUR) != 0) { ok = false; break; }
        for (int32_t i = first; i < n_intv; i++) {
            uint64_t v = 0;
            if (!rd(&v, 8)) break;
            if (v != 0) { linear = v; break; }
        }
    }
    fclose(f);
    if (!ok) return 0;
    uint64_t best = 0;
    bool have = false;
    for (auto &c : chunks) {
        if (c.second <= linear) continue;            // ends before anything the window can need
        const uint64_t at = std::max(c.first, linear);
        if (!have || at < best) { best = at; have = true; }
    }
    return have ? best : linear;
}

struct Base { int32_t u, bq; };
struct Read { std::vector<Base> b; bool alive = true; };

}  // namespace

struct qa_sample_reads {
    std::vector<int32_t> read_ptr, u, bq, wif, central;
    int64_t stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};

extern "C" {

void qa_bam_opts_default(qa_bam_opts_t *o) {
    if (!o) return;
    o->bqFilter = 17;
    o->iSizeUpperLimit = 1000000;
    o->useSoftClippedBases = 0;
    o->downsampleToCov = 30;
    o->chrStart = o->chrEnd = 0;
    o->merge_mates = 1;
    o->seed = 1;
}

int qa_bam_load_sample_reads(const char *bam_path, const char *chr, int32_t nSNPs, const int32_t *L, const char *ref,
                             const char *alt, const int32_t *grid, const qa_bam_opts_t *opts, qa_sample_reads_t **out) {
    if (out) *out = nullptr;
    if (!bam_path || !chr || nSNPs < 1 || !L || !ref || !alt || !grid || !out) {
        qa::set_error("qa_bam_load_sample_reads: missing argument");
        return QA_ERR_INVALID;
    }
    for (int32_t t = 1; t < nSNPs; t++)
        if (L[t] <= L[t - 1]) { qa::set_error("qa_bam_load_sample_reads: the site positions must ascend (L[%d] <= L[%d])", t, t - 1); return QA_ERR_INVALID; }
    // every refusal says which file and what about it (qa_impute_bam_range reports it as "cannot load <file>: <this>")
    auto refuse = [&](const char *what) {
        qa::set_error("%s: %s", bam_path, what);
        return (int)QA_ERR_INVALID;
    };
    qa_bam_opts_t o;
    if (opts) o = *opts; else qa_bam_opts_default(&o);
    {   // CRAM (cramlist + reference, quilt.R:106-108): not decoded here -- say so, with the way round it
        FILE *fc = fopen(bam_path, "rb");
        char m4[4] = {0, 0, 0, 0};
        const bool cram = fc && fread(m4, 1, 4, fc) == 4 && memcmp(m4, "CRAM", 4) == 0;
        if (fc) fclose(fc);
        if (cram) {
            qa::set_error("%s is a CRAM file: qa_bam_load_sample_reads reads BAM only.  Convert it first -- "
                          "`samtools view -b -T <reference.fa> -o sample.bam sample.cram && samtools index sample.bam` -- "
                          "(the reference reads CRAM through STITCH / htslib given `reference`; test-acceptance-cram.R makes its "
                          "CRAMs with the inverse command)", bam_path);
            return QA_ERR_UNSUPPORTED;
        }
    }
    BgzfReader bz(bam_path);
    if (bz.bad) return refuse("cannot be opened");
    char magic[4];
    int32_t l_text, n_ref;
    if (!bz.read(magic, 4) || memcmp(magic, "BAM\1", 4) != 0 || !bz.read(&l_text, 4) || l_text < 0)
        return refuse("not a BAM file (no BGZF block with the BAM magic at its start)");
    std::string text((size_t)l_text, '\0');
    if (l_text && !bz.read(&text[0], (size_t)l_text)) return refuse("the header text is cut short");
    const bool sorted = text.find("SO:coordinate") != std::string::npos;
    if (!bz.read(&n_ref, 4) || n_ref < 0) return refuse("the reference dictionary is cut short");
    int32_t target = -1;
    for (int32_t i = 0; i < n_ref; i++) {
        int32_t l_name, l_ref;
        if (!bz.read(&l_name, 4) || l_name < 1 || l_name > 65536) return refuse("the reference dictionary is damaged");
        std::string name((size_t)l_name, '\0');
        if (!bz.read(&name[0], (size_t)l_name) || !bz.read(&l_ref, 4)) return refuse("the reference dictionary is cut short");
        if (strcmp(name.c_str(), chr) == 0) target = i;
    }
    if (target < 0) { qa::set_error("%s: no reference sequence named %s", bam_path, chr); return QA_ERR_INVALID; }
    // A coordinate-sorted file with a BAI index next to it (<file>.bai or <file without .bam>.bai): start at the linear
    // index's offset for the window's first 16 kb interval -- the first alignment overlapping it (SAM spec 5.1.3) --, tightened
    // by the bin index's chunks (bai_start_offset), instead of scanning from the top of a whole-genome file.  Any problem with the index just means the sequential scan.
    if (sorted) {
        const uint64_t voff = bai_start_offset(bam_path, target, o.chrStart > 0 ? o.chrStart - 1 : 0,
                                               o.chrEnd > 0 ? o.chrEnd : INT32_MAX);
        if (voff != 0 && !bz.seek_virtual(voff)) return refuse("the index points outside the file");
    }

    auto *S = new qa_sample_reads;
    std::vector<Read> reads;
    std::unordered_map<std::string, size_t> by_name;
    const int32_t lo_bp = o.chrStart > 0 ? o.chrStart : 1, hi_bp = o.chrEnd > 0 ? o.chrEnd : INT32_MAX;
    std::vector<uint8_t> rec;
    static const char kNt16[] = "=ACMGRSVTWYHKDBN";
    for (;;) {
        int32_t block_size;
        if (!bz.read(&block_size, 4)) break;
        if (block_size < 32 || block_size > (1 << 28)) { bz.bad = true; break; }   // (no alignment record is a quarter of a GB: a damaged length)
        rec.resize((size_t)block_size);
        if (!bz.read(rec.data(), rec.size())) { bz.bad = true; break; }
        int32_t refID, pos0, next_ref, next_pos, tlen, l_seq;
        memcpy(&refID, &rec[0], 4);
        memcpy(&pos0, &rec[4], 4);
        const int l_read_name = rec[8], mapq = rec[9];
        uint16_t n_cigar, flag;
        memcpy(&n_cigar, &rec[12], 2);
        memcpy(&flag, &rec[14], 2);
        memcpy(&l_seq, &rec[16], 4);
        memcpy(&next_ref, &rec[20], 4);
        memcpy(&next_pos, &rec[24], 4);
        memcpy(&tlen, &rec[28], 4);
        (void)next_ref; (void)next_pos;
        if (refID != target) {
            if (sorted && refID > target) break;
            continue;
        }
        size_t need = 32 + (size_t)l_read_name + 4 * (size_t)n_cigar + (size_t)((l_seq + 1) / 2) + (size_t)l_seq;
        if (l_seq < 0 || need > rec.size()) { bz.bad = true; break; }
        if (sorted && pos0 + 1 > hi_bp) break;
        S->stats[0]++;
        if (flag & (0x4 | 0x100 | 0x200 | 0x400 | 0x800)) { S->stats[4]++; continue; }
        if (mapq < o.bqFilter) { S->stats[2]++; continue; }
        if (std::abs((int64_t)tlen) > (int64_t)o.iSizeUpperLimit) { S->stats[3]++; continue; }
        const uint8_t *cig = &rec[32 + l_read_name], *seq = cig + 4 * (size_t)n_cigar, *qual = seq + (l_seq + 1) / 2;
        // A CIGAR of more than 65 535 operations (long reads) does not fit n_cigar_op: the record then carries the placeholder
        // <l_seq>S<reference length>N and the real CIGAR as the auxiliary array CG:B,I (SAM spec 4.2.2).
        int64_t n_ops = n_cigar;
        if (n_cigar == 2) {
            uint32_t c0, c1;
            memcpy(&c0, cig, 4);
            memcpy(&c1, cig + 4, 4);
            if ((c0 & 15) == 4 && (int64_t)(c0 >> 4) == l_seq && (c1 & 15) == 3) {
                const uint8_t *a = qual + l_seq, *end = rec.data() + rec.size();
                while (a + 3 <= end) {   // walk the auxiliary fields: tag[2] type value
                    const char t0 = (char)a[0], t1 = (char)a[1], ty = (char)a[2];
                    a += 3;
                    size_t len = 0;
                    if (ty == 'A' || ty == 'c' || ty == 'C') len = 1;
                    else if (ty == 's' || ty == 'S') len = 2;
                    else if (ty == 'i' || ty == 'I' || ty == 'f') len = 4;
                    else if (ty == 'Z' || ty == 'H') { while (a + len < end && a[len]) len++; len++; }
                    else if (ty == 'B') {
                        if (a + 5 > end) break;
                        const char sub = (char)a[0];
                        uint32_t cnt;
                        memcpy(&cnt, a + 1, 4);
                        const size_t es = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
                        if (t0 == 'C' && t1 == 'G' && sub == 'I' && a + 5 + 4 * (size_t)cnt <= end) {
                            cig = a + 5;
                            n_ops = cnt;
                            break;
                        }
                        len = 5 + es * (size_t)cnt;
                    } else break;   // unknown type: stop looking (the placeholder then stands: no base is used)
                    if (a + len > end) break;
                    a += len;
                }
            }
        }
        // walk the CIGAR; a soft clip is laid out left of / right of the aligned part when its bases are to be used
        int64_t rpos = (int64_t)pos0 + 1;   // 1-based reference coordinate of the next reference-consuming base
        if (o.useSoftClippedBases) {   // the leading soft clip: the first operation that is not a hard clip (2H3S4M)
            for (int64_t ci = 0; ci < n_ops; ci++) {
                uint32_t c0;
                memcpy(&c0, cig + 4 * (size_t)ci, 4);
                if ((c0 & 15) == 5) continue;
                if ((c0 & 15) == 4) rpos -= (c0 >> 4);
                break;
            }
        }
        const int64_t aln_start = rpos;
        int32_t q = 0;
        std::vector<Base> bases;
        // first site at or after the alignment start
        int32_t t = (int32_t)(std::lower_bound(L, L + nSNPs, (int32_t)std::max<int64_t>(rpos, INT32_MIN)) - L);
        for (int64_t ci = 0; ci < n_ops; ci++) {
            uint32_t c;
            memcpy(&c, cig + 4 * (size_t)ci, 4);
            const int op = c & 15;
            const int32_t len = (int32_t)(c >> 4);
            const bool clip_used = op == 4 && o.useSoftClippedBases;
            if (op == 0 || op == 7 || op == 8 || clip_used) {   // M, =, X (and S when used): query and reference advance
                while (t < nSNPs && L[t] < rpos) t++;
                while (t < nSNPs && L[t] < rpos + len) {
                    const int32_t qi = q + (int32_t)(L[t] - rpos);
                    if (qi < l_seq) {
                        const char base = kNt16[(seq[qi >> 1] >> ((qi & 1) ? 0 : 4)) & 15];
                        int32_t bqv = qual[qi] == 0xff ? 0 : qual[qi];
                        if (bqv > mapq) bqv = mapq;
                        if (bqv >= o.bqFilter) {
                            if (base == ref[t]) bases.push_back({t, -bqv});
                            else if (base == alt[t]) bases.push_back({t, bqv});
                        }
                    }
                    t++;
                }
                q += len;
                rpos += len;
            } else if (op == 1 || op == 4) {   // I, S (unused): query only
                q += len;
            } else if (op == 2 || op == 3) {   // D, N: reference only
                rpos += len;
            }                                  // H, P: neither
        }
        if (rpos - 1 < lo_bp || aln_start > hi_bp) continue;   // outside the window
        S->stats[1]++;
        if (bases.empty()) { S->stats[7]++; continue; }
        if (o.merge_mates && (flag & 0x1)) {
            std::string name(reinterpret_cast<const char *>(&rec[32]), (size_t)std::max(0, l_read_name - 1));
            auto it = by_name.find(name);
            if (it != by_name.end()) {
                auto &dst = reads[it->second].b;
                dst.insert(dst.end(), bases.begin(), bases.end());
                std::stable_sort(dst.begin(), dst.end(), [](const Base &a, const Base &b) { return a.u < b.u; });
                // A site both mates cover is ONE observation of the molecule, not two: mates that agree keep the call with the
                // higher quality, mates that disagree drop the site (rule stated in include/quilt_amd_io.h; unpinned vs STITCH)
                size_t w = 0;
                for (size_t i = 0; i < dst.size();) {
                    size_t j = i + 1;
                    Base keep = dst[i];
                    bool conflict = false;
                    for (; j < dst.size() && dst[j].u == dst[i].u; j++) {
                        if ((dst[j].bq < 0) != (keep.bq < 0)) conflict = true;
                        else if (std::abs(dst[j].bq) > std::abs(keep.bq)) keep = dst[j];
                    }
                    if (!conflict) dst[w++] = keep;
                    i = j;
                }
                dst.resize(w);
                if (dst.empty()) { reads[it->second].alive = false; S->stats[7]++; }
                by_name.erase(it);
                S->stats[6]++;
                continue;
            }
            by_name.emplace(std::move(name), reads.size());
        }
        reads.emplace_back();
        reads.back().b = std::move(bases);
    }
    if (bz.bad) { delete S; return refuse("damaged or cut short (a BGZF block or an alignment record does not decode)"); }

    // coverage cap (quilt.R:54): sites in ascending order; at a site above the cap the covering reads with the smallest
    // stream keys go, until the site is at the cap
    if (o.downsampleToCov > 0) {
        std::vector<int32_t> depth((size_t)nSNPs, 0);
        for (auto &r : reads) for (auto &b : r.b) depth[b.u]++;
        bool any = false;
        for (int32_t t = 0; t < nSNPs; t++) any |= depth[t] > o.downsampleToCov;
        if (any) {
            std::vector<std::vector<uint32_t>> cover((size_t)nSNPs);
            for (size_t r = 0; r < reads.size(); r++)
                for (auto &b : reads[r].b) if (depth[b.u] > o.downsampleToCov) cover[b.u].push_back((uint32_t)r);
            for (int32_t t = 0; t < nSNPs; t++) {
                if (depth[t] <= o.downsampleToCov) continue;
                std::vector<std::pair<uint64_t, uint32_t>> cand;
                for (uint32_t r : cover[t]) if (reads[r].alive) cand.push_back({stream_key(o.seed, r), r});
                std::sort(cand.begin(), cand.end());
                for (size_t i = 0; i < cand.size() && depth[t] > o.downsampleToCov; i++) {
                    Read &rd = reads[cand[i].second];
                    if (!rd.alive) continue;
                    rd.alive = false;
                    S->stats[5]++;
                    for (auto &b : rd.b) depth[b.u]--;
                }
            }
        }
    }

    // order by the grid of the central site (stable), flatten
    std::vector<uint32_t> order;
    std::vector<int32_t> cen(reads.size());
    for (size_t r = 0; r < reads.size(); r++) {
        if (!reads[r].alive) continue;
        cen[r] = reads[r].b[(reads[r].b.size() - 1) / 2].u;
        order.push_back((uint32_t)r);
    }
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return grid[cen[a]] < grid[cen[b]]; });
    S->read_ptr.push_back(0);
    for (uint32_t r : order) {
        for (auto &b : reads[r].b) { S->u.push_back(b.u); S->bq.push_back(b.bq); }
        S->read_ptr.push_back((int32_t)S->u.size());
        S->wif.push_back(grid[cen[r]]);
        S->central.push_back(cen[r]);
    }
    *out = S;
    return QA_OK;
}

int32_t qa_sample_reads_n_reads(const qa_sample_reads_t *s) { return s ? (int32_t)s->wif.size() : 0; }
int64_t qa_sample_reads_n_bases(const qa_sample_reads_t *s) { return s ? (int64_t)s->u.size() : 0; }
void qa_sample_reads_stats(const qa_sample_reads_t *s, int64_t stats[8]) {
    if (s && stats) memcpy(stats, s->stats, sizeof s->stats);
}
int qa_sample_reads_export(const qa_sample_reads_t *s, int32_t *read_ptr, int32_t *u, int32_t *bq, int32_t *wif,
                           int32_t *central) {
    if (!s) return QA_ERR_INVALID;
    auto cp = [](int32_t *d, const std::vector<int32_t> &v) { if (d && !v.empty()) memcpy(d, v.data(), 4 * v.size()); };
    cp(read_ptr, s->read_ptr);
    cp(u, s->u);
    cp(bq, s->bq);
    cp(wif, s->wif);
    cp(central, s->central);
    return QA_OK;
}
void qa_sample_reads_destroy(qa_sample_reads_t *s) { delete s; }

// ---------------------------------------------------------------------------------------------------------------------
// f4
// ---------------------------------------------------------------------------------------------------------------------
}  // extern "C"

namespace {

// printf("%.3f", x) for |x| <= 1e9 without printf: the decimal expansion of a double is exact and finite, so "three decimals,
// correctly rounded, ties to even" is integer arithmetic on its mantissa -- x = m * 2^-k (m < 2^53), x * 1000 = (1000 m) / 2^k, the
// quotient rounded on the exact remainder.  What glibc prints (round-to-nearest mode), at a fifth of the cost: a VCF column is
// six such numbers for each of 64 000 SNPs, 50 ms per sample through snprintf.
inline int fmt_fixed3(char *p, double x) {
    uint64_t bits;
    memcpy(&bits, &x, 8);
    int n = 0;
    if (bits >> 63) p[n++] = '-';
    const int E = (int)((bits >> 52) & 0x7ff);
    uint64_t N = 0;   // round_half_even(|x| * 1000)
    if (E != 0) {     // (subnormals print as 0.000)
        const uint64_t m = (bits & 0x000fffffffffffffull) | 0x0010000000000000ull;
        const int k = 1075 - E;   // |x| = m * 2^-k
        if (k <= 0) {
            N = (m * 1000ull) << (-k);   // (|x| <= 1e9 < 2^30: m << -k < 2^30 only when -k <= 0 ... kept for completeness: k <= 0 means |x| >= 2^52)
        } else {
            const unsigned __int128 P = (unsigned __int128)m * 1000u;   // < 2^63
            if (k < 127) {
                const unsigned __int128 q = P >> k, r = P & ((((unsigned __int128)1) << k) - 1), half = ((unsigned __int128)1) << (k - 1);
                N = (uint64_t)q + ((r > half || (r == half && ((uint64_t)q & 1))) ? 1 : 0);
            }
        }
    }
    uint64_t ip = N / 1000;
    const unsigned fp = (unsigned)(N % 1000);
    char tmp[24];
    int d = 0;
    do { tmp[d++] = (char)('0' + ip % 10); ip /= 10; } while (ip);
    while (d) p[n++] = tmp[--d];
    p[n++] = '.';
    p[n++] = (char)('0' + fp / 100);
    p[n++] = (char)('0' + fp / 10 % 10);
    p[n++] = (char)('0' + fp % 10);
    p[n] = 0;
    return n;
}

// R's paste0(round(x, 3)) for the magnitudes a posterior / dosage takes: three decimals, then the shortest form
int fmt_round3(char *p, double x) {
    if (std::isnan(x)) { memcpy(p, "NA", 2); return 2; }
    int n = (std::isfinite(x) && std::fabs(x) <= 1e9) ? fmt_fixed3(p, x) : snprintf(p, 32, "%.3f", x);
    if (strcmp(p, "-0.000") == 0) { memcpy(p, "0", 2); return 1; }
    while (n > 0 && p[n - 1] == '0') n--;
    if (n > 0 && p[n - 1] == '.') n--;
    p[n] = 0;
    return n;
}
// R's as.character(round(x, 5))
int fmt_round5(char *p, double x) {
    if (std::isnan(x)) { memcpy(p, "NaN", 3); return 3; }
    if (std::isinf(x)) { int n = snprintf(p, 32, x > 0 ? "Inf" : "-Inf"); return n; }
    int n = snprintf(p, 32, "%.5f", x);
    while (n > 0 && p[n - 1] == '0') n--;
    if (n > 0 && p[n - 1] == '.') n--;
    p[n] = 0;
    if (strcmp(p, "-0") == 0) { memcpy(p, "0", 2); return 1; }
    // R switches to scientific notation when that is no wider: below 1e-4 after rounding to 5 decimals only 1e-05 .. 9e-05
    if (x != 0 && std::fabs(x) < 1e-4 && n > 1) {
        double r = std::round(std::fabs(x) * 1e5);
        if (r >= 1 && r <= 9) n = snprintf(p, 32, "%s%de-05", x < 0 ? "-" : "", (int)r);
    }
    return n;
}
inline int r_round_int(double x) { return (int)std::nearbyint(x); }   // half to even, as R's round(x)
// what the column writers accept: posteriors, dosages and counts -- finite and of a magnitude whose fixed-point text has a known
// width (so that no entry can outgrow its buffer); anything else is the caller's error, reported with the SNP
inline bool printable(double x) { return std::isfinite(x) && std::fabs(x) <= 1e9; }
inline bool bounded(double x) { return !std::isfinite(x) || std::fabs(x) <= 1e9; }   // NaN / Inf have their own short texts
inline int bad_value(const char *fn, int32_t t, const char *what) {
    qa::set_error("%s: %s at SNP %d is not finite (or beyond 1e9)", fn, what, (int)t);
    return QA_ERR_INVALID;
}

struct Sink {
    char *buf;
    int64_t cap, used = 0;
    int64_t *off;
    void put(int32_t t, const char *s, int n) {
        if (used + n + 1 <= cap) {
            if (off) off[t] = used;
            memcpy(buf + used, s, (size_t)n);
            buf[used + n] = 0;
        }
        used += n + 1;
    }
    int done(int32_t T, int64_t *needed) {
        if (needed) *needed = used;
        if (used > cap) return QA_ERR_CAPACITY;
        if (off) off[T] = used;
        return QA_OK;
    }
};

}  // namespace

extern "C" {

const char *qa_vcf_missing_entry(void) { return "./.:.,.,.:.:.,."; }

int qa_vcf_column_diploid(int32_t T, const double *gp_t, const double *hd, int32_t phased_gt, char *buf, int64_t cap,
                          int64_t *off, int64_t *needed) {
    if (T < 0 || !gp_t || !hd || (!buf && cap > 0)) return QA_ERR_INVALID;
    Sink s{buf, cap, 0, off};
    char e[160];
    for (int32_t t = 0; t < T; t++) {
        const double g0 = gp_t[3 * (size_t)t], g1 = gp_t[3 * (size_t)t + 1], g2 = gp_t[3 * (size_t)t + 2];
        const double h1 = hd[t], h2 = hd[(size_t)T + t];
        if (!printable(g0) || !printable(g1) || !printable(g2)) return bad_value("qa_vcf_column_diploid", t, "a genotype posterior");
        if (!printable(h1) || !printable(h2)) return bad_value("qa_vcf_column_diploid", t, "a haploid dosage");
        int n;
        if (phased_gt) {
            const int a1 = r_round_int(h1), a2 = r_round_int(h2);
            if (a1 >= 0 && a1 <= 9 && a2 >= 0 && a2 <= 9) { e[0] = (char)('0' + a1); e[1] = '|'; e[2] = (char)('0' + a2); n = 3; }
            else n = snprintf(e, sizeof e, "%d|%d", a1, a2);
        } else {
            const char *gt = g0 >= 0.9 ? "0/0" : g1 >= 0.9 ? "0/1" : g2 >= 0.9 ? "1/1" : "./.";
            memcpy(e, gt, 3);
            n = 3;
        }
        // ":%.3f,%.3f,%.3f:%.3f:%.3f,%.3f" (six numbers of at most 15 characters each: well inside e[160])
        const double ds = g1 + 2 * g2;
        if (!printable(ds)) return bad_value("qa_vcf_column_diploid", t, "the dosage");
        e[n++] = ':'; n += fmt_fixed3(e + n, g0);
        e[n++] = ','; n += fmt_fixed3(e + n, g1);
        e[n++] = ','; n += fmt_fixed3(e + n, g2);
        e[n++] = ':'; n += fmt_fixed3(e + n, ds);
        e[n++] = ':'; n += fmt_fixed3(e + n, h1);
        e[n++] = ','; n += fmt_fixed3(e + n, h2);
        e[n] = 0;
        s.put(t, e, n);
    }
    return s.done(T, needed);
}

int qa_vcf_column_nipt(int32_t T, const double *m, const double *f, const double *hd, const double *mds, const double *fds,
                       char *buf, int64_t cap, int64_t *off, int64_t *needed) {
    if (T < 0 || !m || !f || !hd || !mds || !fds || (!buf && cap > 0)) return QA_ERR_INVALID;
    Sink s{buf, cap, 0, off};
    char e[320];
    for (int32_t t = 0; t < T; t++) {
        const double v[8] = {m[3 * (size_t)t], m[3 * (size_t)t + 1], m[3 * (size_t)t + 2], mds[t],
                             f[3 * (size_t)t], f[3 * (size_t)t + 1], f[3 * (size_t)t + 2], fds[t]};
        for (int i = 0; i < 3; i++)
            if (!printable(hd[i * (size_t)T + t])) return bad_value("qa_vcf_column_nipt", t, "a haploid dosage");
        for (int i = 0; i < 8; i++)
            if (!bounded(v[i])) return bad_value("qa_vcf_column_nipt", t, "a posterior or dosage");
        // (3 integers, then 8 numbers of at most 14 characters each: well inside e[320])
        int n = snprintf(e, sizeof e, "%d|%d|%d:", r_round_int(hd[t]), r_round_int(hd[(size_t)T + t]),
                         r_round_int(hd[2 * (size_t)T + t]));
        static const char sep[8] = {',', ',', ':', ':', ',', ',', ':', 0};
        for (int i = 0; i < 8; i++) {
            n += fmt_round3(e + n, v[i]);
            if (sep[i]) e[n++] = sep[i];
        }
        e[n] = 0;
        s.put(t, e, n);
    }
    return s.done(T, needed);
}

int qa_vcf_info_column(int32_t T, const double *eaf, const double *info, const double *hwe, const double *ac, char *buf,
                       int64_t cap, int64_t *off, int64_t *needed) {
    if (T < 0 || !eaf || !info || !hwe || !ac || (!buf && cap > 0)) return QA_ERR_INVALID;
    Sink s{buf, cap, 0, off};
    char e[320];
    for (int32_t t = 0; t < T; t++) {
        int n = 0;
        // (six fields of at most 16 characters each plus their keys: well inside e[320]; HWE may be any finite value)
        const double chk[6] = {eaf[t], info[t], ac[t], ac[(size_t)T + t], ac[2 * (size_t)T + t], 0.0};
        for (double x : chk)
            if (!bounded(x)) return bad_value("qa_vcf_info_column", t, "EAF, INFO_SCORE or an allele count");
        memcpy(e + n, "EAF=", 4); n += 4; n += fmt_round5(e + n, eaf[t]);
        memcpy(e + n, ";INFO_SCORE=", 12); n += 12; n += fmt_round5(e + n, info[t]);
        // formatC(hwe, format = "e", digits = 2): C's %.2e (two-digit exponent at least)
        n += snprintf(e + n, sizeof e - n, ";HWE=%.2e", hwe[t]);
        memcpy(e + n, ";ERC=", 5); n += 5; n += fmt_round5(e + n, ac[t]);
        memcpy(e + n, ";EAC=", 5); n += 5; n += fmt_round5(e + n, ac[(size_t)T + t] - ac[t]);
        memcpy(e + n, ";PAF=", 5); n += 5; n += fmt_round5(e + n, ac[2 * (size_t)T + t]);
        e[n] = 0;
        s.put(t, e, n);
    }
    return s.done(T, needed);
}

// Exact Hardy-Weinberg test on genotype counts (Wigginton, Cutler & Abecasis 2005: sum of the probabilities of all
// heterozygote counts no more likely than the observed one, given the allele counts); replaces
// STITCH::generate_hwe_on_counts as called at writers.R:58.  counts nSNPs x 3 column-major (hom-ref, het, hom-alt).
int qa_hwe_exact(int32_t T, const double *counts, double *p_out) {
    if (T < 0 || !counts || !p_out) return QA_ERR_INVALID;
    std::vector<double> pr;
    for (int32_t t = 0; t < T; t++) {
        const long n_aa = std::lround(counts[t]), n_ab = std::lround(counts[(size_t)T + t]),
                   n_bb = std::lround(counts[2 * (size_t)T + t]);
        if (n_aa < 0 || n_ab < 0 || n_bb < 0) return QA_ERR_INVALID;
        const long n = n_aa + n_ab + n_bb;
        if (n == 0) { p_out[t] = 1; continue; }
        const long rare = 2 * std::min(n_aa, n_bb) + n_ab;
        pr.assign((size_t)rare + 1, 0.0);
        long mid = (long)((double)rare * (double)(2 * n - rare) / (double)(2 * n));
        if ((mid & 1) != (rare & 1)) mid++;
        long het = mid, hr = (rare - mid) / 2, hc = n - het - hr;
        pr[mid] = 1;
        double sum = 1;
        for (het = mid; het > 1; het -= 2) {
            pr[het - 2] = pr[het] * (double)het * (double)(het - 1) / (4.0 * (double)(hr + 1) * (double)(hc + 1));
            sum += pr[het - 2];
            hr++;
            hc++;
        }
        hr = (rare - mid) / 2;
        hc = n - mid - hr;
        for (het = mid; het <= rare - 2; het += 2) {
            pr[het + 2] = pr[het] * 4.0 * (double)hr * (double)hc / ((double)(het + 2) * (double)(het + 1));
            sum += pr[het + 2];
            hr--;
            hc--;
        }
        double p = 0;
        const double obs = pr[n_ab] / sum;
        for (long i = 0; i <= rare; i++) if (pr[i] / sum <= obs) p += pr[i] / sum;
        p_out[t] = p > 1 ? 1 : p;
    }
    return QA_OK;
}

// functions.R:999-1020 for a whole round: every chain's haploid dosages h1, h2 (h3: NIPT) added to its sample's dosage and
// genotype-posterior sums -- dosage += h1 + h2; gp_t += rbind((1-h1)(1-h2), (1-h1)h2 + h1(1-h2), h1 h2); fetal:
// the same with (h1, h3) -- chain by chain in order, each expression rounded as R / numpy round it (built without
// contraction), one pass over the chain's rows instead of a dozen temporaries.
int qa_accumulate_dosage(int32_t n_chain, int32_t n_label, int32_t T, const double *hap, const int32_t *chain_sample,
                         int32_t n_sample, double *dosage, double *gp_t, double *fet_dosage, double *fet_gp_t) {
    if (n_chain < 0 || n_label < 2 || n_label > 3 || T < 0 || !hap || !chain_sample || !dosage || !gp_t) return QA_ERR_INVALID;
    if ((fet_dosage || fet_gp_t) && (n_label != 3 || !fet_dosage || !fet_gp_t)) return QA_ERR_INVALID;
    for (int32_t c = 0; c < n_chain; c++) {
        const int32_t s = chain_sample[c];
        if (s < 0 || s >= n_sample) return QA_ERR_INVALID;
        const double *h1 = hap + ((size_t)c * n_label) * T, *h2 = h1 + T, *h3 = h1 + 2 * (size_t)T;
        double *d = dosage + (size_t)s * T, *g0 = gp_t + (size_t)s * 3 * T, *g1 = g0 + T, *g2 = g1 + T;
        for (int32_t t = 0; t < T; t++) {
            const double a = h1[t], b = h2[t], na = 1 - a, nb = 1 - b;
            d[t] += a + b;
            g0[t] += na * nb;
            g1[t] += na * b + a * nb;
            g2[t] += a * b;
        }
        if (fet_dosage) {
            double *fd = fet_dosage + (size_t)s * T, *f0 = fet_gp_t + (size_t)s * 3 * T, *f1 = f0 + T, *f2 = f1 + T;
            for (int32_t t = 0; t < T; t++) {
                const double a = h1[t], b = h3[t], na = 1 - a, nb = 1 - b;
                fd[t] += a + b;
                f0[t] += na * nb;
                f1[t] += na * b + a * nb;
                f2[t] += a * b;
            }
        }
    }
    return QA_OK;
}

// The match weights of select_new_haps_mspbwt_v3 (QUILT/R/mspbwt.R:418-427): matches in the given order, each weighted
// by its length over the coverage accumulated by the matches before it: weight = (e - s + 1) / sum(cur_sum[s:e]), then
// cur_sum[s:e] += 1 (cur_sum starts at 1).  start1 / end1 are 1-based inclusive.
int qa_mspbwt_weights(int32_t n, const int64_t *start1, const int64_t *end1, double *weight) {
    if (n < 0 || (n > 0 && (!start1 || !end1 || !weight))) return QA_ERR_INVALID;
    int64_t top = 0;
    for (int32_t i = 0; i < n; i++) {
        if (start1[i] < 1 || end1[i] < start1[i]) return QA_ERR_INVALID;
        top = std::max(top, end1[i]);
    }
    std::vector<double> cur((size_t)top + 1, 1.0);
    for (int32_t i = 0; i < n; i++) {
        double sum = 0;
        for (int64_t p = start1[i]; p <= end1[i]; p++) sum += cur[p];
        weight[i] = (double)(end1[i] - start1[i] + 1) * 1 / sum;
        for (int64_t p = start1[i]; p <= end1[i]; p++) cur[p] += 1;
    }
    return QA_OK;
}

// select_new_haps_mspbwt_v3 (QUILT/R/mspbwt.R:303-474, heuristic_approach "A") for a batch of chains, from the match tables
// qa_find_good_matches returns.  Restated line by line (quilt_amd/mspbwt.py is the same text in numpy, and the test of this
// function); R's sample() draws are keyed draws of the chain's selection stream (smallest keys at offset 2^21, in key order).
//   match      n_chain x n_label x nindices x max_matches x 3 (index0, start0, len1), n_match the rows in use
//   out        n_chain x Knew 1-based haplotypes
int qa_select_new_haps_mspbwt(int32_t n_chain, int32_t n_label, int32_t nindices, int32_t max_matches, const int32_t *match,
                              const int32_t *n_match, int32_t Knew, int32_t Kfull, int32_t nGrids, const uint64_t *seed,
                              int32_t *out) {
    if (n_chain < 0 || n_label < 1 || n_label > 3 || nindices < 1 || max_matches < 1 || !match || !n_match || Knew < 1 ||
        Knew > Kfull || !seed || !out)
        return QA_ERR_INVALID;
    struct Row { int64_t index1, start1, end1, len1; };
    constexpr uint64_t kPoolOffset = 1ull << 21;
    auto keyed_subset = [](uint64_t sd, int64_t n, int64_t m, uint64_t off, std::vector<int64_t> &pick) {
        std::vector<std::pair<uint64_t, int64_t>> k((size_t)n);
        for (int64_t i = 0; i < n; i++) k[(size_t)i] = {stream_key(sd, off + (uint64_t)i), i};
        m = std::min(m, n);
        std::partial_sort(k.begin(), k.begin() + m, k.end());
        pick.resize((size_t)m);
        for (int64_t i = 0; i < m; i++) pick[(size_t)i] = k[(size_t)i].second;
    };
    std::vector<uint8_t> seen((size_t)Kfull + 1);
    for (int32_t c = 0; c < n_chain; c++) {
        std::vector<std::vector<Row>> outm((size_t)n_label);
        for (int32_t h = 0; h < n_label; h++) {
            std::vector<Row> &all = outm[(size_t)h];
            for (int32_t i = 0; i < nindices; i++) {
                const size_t q = ((size_t)c * n_label + h) * nindices + i;
                const int32_t n = n_match[q];
                if (n < 0 || n > max_matches) return QA_ERR_INVALID;
                const int32_t *m = match + q * (size_t)max_matches * 3;
                std::vector<Row> rows((size_t)n);
                for (int32_t r = 0; r < n; r++) {
                    if (m[3 * r] < 0 || m[3 * r] >= Kfull || m[3 * r + 2] < 1) return QA_ERR_INVALID;
                    rows[(size_t)r] = {(int64_t)m[3 * r] + 1, (int64_t)m[3 * r + 1] + 1, (int64_t)m[3 * r + 1] + m[3 * r + 2],
                                       (int64_t)m[3 * r + 2]};
                }
                if (n > 1) {   // same haplotype and start: the longest first, the others dropped (mspbwt.R:349-358)
                    std::stable_sort(rows.begin(), rows.end(), [](const Row &a, const Row &b) {
                        if (a.index1 != b.index1) return a.index1 < b.index1;
                        if (a.end1 != b.end1) return a.end1 > b.end1;
                        return a.start1 > b.start1;
                    });
                    std::vector<Row> keep;
                    for (size_t r = 0; r < rows.size(); r++)
                        if (r == 0 || !(rows[r].index1 == rows[r - 1].index1 && rows[r].start1 == rows[r - 1].start1))
                            keep.push_back(rows[r]);
                    rows.swap(keep);
                }
                all.insert(all.end(), rows.begin(), rows.end());
            }
            std::stable_sort(all.begin(), all.end(), [](const Row &a, const Row &b) { return a.len1 > b.len1; });
        }
        std::fill(seen.begin(), seen.end(), 0);
        std::vector<int64_t> unique_haps;
        for (auto &all : outm)
            for (auto &r : all)
                if (!seen[(size_t)r.index1]) { seen[(size_t)r.index1] = 1; unique_haps.push_back(r.index1); }
        int32_t *o = out + (size_t)c * Knew;
        std::vector<int64_t> pick;
        if (unique_haps.empty()) {                              // sample(1:Kfull, Knew)
            keyed_subset(seed[c], Kfull, Knew, kPoolOffset, pick);
            for (int32_t i = 0; i < Knew; i++) o[i] = (int32_t)pick[(size_t)i] + 1;
            continue;
        }
        if ((int64_t)unique_haps.size() <= Knew) {              // the identified ones, topped up from the rest of the panel
            std::vector<int64_t> pool;
            for (int64_t k = 1; k <= Kfull; k++) if (!seen[(size_t)k]) pool.push_back(k);
            const int64_t n_u = (int64_t)unique_haps.size();
            keyed_subset(seed[c], (int64_t)pool.size(), Knew - n_u, kPoolOffset, pick);
            for (int64_t i = 0; i < n_u; i++) o[i] = (int32_t)unique_haps[(size_t)i];
            for (int64_t i = 0; i < Knew - n_u; i++) o[n_u + i] = (int32_t)pool[(size_t)pick[(size_t)i]];
            continue;
        }
        // prioritise by length and new-ness per haplotype, then interleave (mspbwt.R:416-466)
        std::vector<std::vector<int64_t>> results((size_t)n_label);
        size_t a = 0;
        for (int32_t h = 0; h < n_label; h++) {
            const std::vector<Row> &all = outm[(size_t)h];
            const int32_t n = (int32_t)all.size();
            std::vector<int64_t> s1((size_t)n), e1((size_t)n);
            std::vector<double> w((size_t)n);
            for (int32_t r = 0; r < n; r++) { s1[(size_t)r] = all[(size_t)r].start1; e1[(size_t)r] = all[(size_t)r].end1; }
            if (qa_mspbwt_weights(n, s1.data(), e1.data(), w.data()) != QA_OK) return QA_ERR_INVALID;
            std::vector<int32_t> ord((size_t)n);
            for (int32_t r = 0; r < n; r++) ord[(size_t)r] = r;
            std::stable_sort(ord.begin(), ord.end(), [&](int32_t x, int32_t y) { return w[(size_t)x] > w[(size_t)y]; });
            for (int32_t r : ord) results[(size_t)h].push_back(all[(size_t)r].index1);
            a = std::max(a, results[(size_t)h].size());
        }
        std::fill(seen.begin(), seen.end(), 0);
        std::vector<int64_t> ordered;
        for (size_t r = 0; r < a; r++)
            for (int32_t h = 0; h < n_label; h++)
                if (r < results[(size_t)h].size()) {
                    const int64_t k = results[(size_t)h][r];
                    if (!seen[(size_t)k]) { seen[(size_t)k] = 1; ordered.push_back(k); }
                }
        if ((int64_t)ordered.size() >= Knew) {
            for (int32_t i = 0; i < Knew; i++) o[i] = (int32_t)ordered[(size_t)i];
        } else {   // c(setdiff(unique_ordered_haps, unique_haps), unique_haps)[1:Knew] (the setdiff is empty by construction)
            for (int32_t i = 0; i < Knew; i++) o[i] = (int32_t)unique_haps[(size_t)i];
        }
    }
    return QA_OK;
}

// Read confidence and consensus labels of one sample (QUILT/R/functions.R:1615-1660 assess_ability_of_reads_to_be_confident,
// :1680-1784 determine_best_read_label_so_far, :1788-1829 its NIPT wrapper): from the Gibbs samples' read labels and the
// reads' likelihoods against each sample's haplotypes, the labels the phasing pass starts from.  The suffix rewrites of
// the reference's working matrix are kept as flip parities (quilt_amd/driver.py has the numpy text and the line-by-line
// form this is tested against), including that the final flip starts at the LAST change point counted in the filtered rows.
//   labels  n x nReads (chain-major) read labels of the n Gibbs samples; p  n x K x nReads likelihoods (K = 2, or 3: NIPT)
int qa_consensus_read_labels(int32_t nReads, int32_t n, const int32_t *labels, const double *p, int32_t K, double minrp,
                             int32_t can_hap, int32_t *out) {
    if (nReads < 0 || n < 1 || !labels || !p || (K != 2 && K != 3) || can_hap < 1 || can_hap > n || !out) return QA_ERR_INVALID;
    const bool nipt = K == 3;
    const size_t R = (size_t)nReads;
    std::vector<int32_t> rl((size_t)n * R);
    std::vector<uint8_t> conf((size_t)n * R);
    for (int32_t c = 0; c < n; c++) {
        const double *pc = p + (size_t)c * K * R;
        for (size_t r = 0; r < R; r++) {
            double mp;
            if (!nipt) {
                mp = pc[r] / (pc[r] + pc[R + r]);
                if (std::isnan(mp)) mp = 0.5;
                if (mp < 0.5) mp = 1 - mp;
            } else {
                const double sum = pc[r] + pc[R + r] + pc[2 * R + r];
                const double q0 = pc[r] / sum, q1 = pc[R + r] / sum, q2 = pc[2 * R + r] / sum;
                mp = q0;
                if (q1 > q0) mp = q1;
                if (q2 > mp) mp = q2;
                if (std::isnan(mp)) mp = 1.0 / 3;
            }
            int32_t l = labels[(size_t)c * R + r];
            bool cf = mp > minrp;
            if (nipt && l == 3) { cf = false; l = 2; }   // label 3 folded into 2 and called not confident (:1800-1806)
            rl[(size_t)c * R + r] = l;
            conf[(size_t)c * R + r] = cf;
        }
    }
    const int32_t can = can_hap - 1;
    for (size_t r = 0; r < R; r++) out[r] = rl[(size_t)can * R + r];
    auto finish = [&]() {
        if (nipt) for (size_t r = 0; r < R; r++) if (labels[(size_t)can * R + r] == 3) out[r] = 3;
        return (int)QA_OK;
    };
    std::vector<size_t> keep;
    for (size_t r = 0; r < R; r++) {
        bool all = true;
        for (int32_t c = 0; c < n && all; c++) all = conf[(size_t)c * R + r] != 0;
        if (all) keep.push_back(r);
    }
    if (keep.size() < 10) return finish();
    auto L0 = [&](size_t i, int32_t c) { return (int64_t)rl[(size_t)c * R + keep[i]]; };
    auto rowsum = [&](size_t i) { int64_t s = 0; for (int32_t c = 0; c < n; c++) s += std::llabs(L0(i, c) - L0(i, can)); return s; };
    std::vector<size_t> change;   // filtered rows at which the pattern of disagreements changes (0-based)
    {
        int64_t prev = rowsum(0);
        for (size_t i = 1; i < keep.size(); i++) {
            const int64_t cur = rowsum(i);
            if (cur != prev) change.push_back(i);
            prev = cur;
        }
    }
    if (change.empty()) return finish();
    std::vector<uint8_t> fc((size_t)n, 0), flipped((size_t)n, 0);
    bool fcan = false;
    const double half = n / 2.0;
    std::vector<int32_t> changed;
    for (size_t i : change) {
        const int64_t canv = fcan ? 3 - L0(i, can) : L0(i, can);
        changed.clear();
        for (int32_t c = 0; c < n; c++) {
            const int64_t lab = fc[(size_t)c] ? 3 - L0(i, c) : L0(i, c);
            if (lab - canv != 0) changed.push_back(c);
        }
        if (!changed.empty()) {
            if ((double)changed.size() > half) {   // the majority moved: it is the canonical haplotype that flips
                std::vector<int32_t> others;
                for (int32_t c = 0; c < n; c++) {
                    const int64_t lab = fc[(size_t)c] ? 3 - L0(i, c) : L0(i, c);
                    if (lab - canv == 0) others.push_back(c);
                }
                changed.swap(others);
                fcan = !fcan;
            }
            for (int32_t c : changed) fc[(size_t)c] ^= 1;
        }
        for (int32_t c : changed) flipped[(size_t)c] = 1;
    }
    if (flipped[(size_t)can]) {
        const size_t w0 = change.back();   // s1[i] - 1 with the loop's last i: an index into the UNFILTERED reads
        for (size_t r = w0; r < R; r++) out[r] = 3 - out[r];
    }
    return finish();
}

int qa_vcf_write_text(const char *path, int32_t bgzf, int32_t truncate, const char *text, int64_t n) {
    if (!path || (!text && n > 0) || n < 0) return QA_ERR_INVALID;
    BgzfWriter w(path, bgzf != 0, truncate != 0);
    if (!w.ok()) return QA_ERR_INVALID;
    if (n && !w.write(text, (size_t)n)) return QA_ERR_INVALID;
    return w.finish(false) ? QA_OK : QA_ERR_INVALID;
}

int qa_vcf_write_body(const char *path, int32_t bgzf, int32_t finish, const char *chr, int32_t T, const int32_t *pos_bp,
                      const char *ref, const char *alt, const uint8_t *keep, const char *info, const int64_t *info_off,
                      const char *format, int32_t N, const char *const *cols, const int64_t *const *offs) {
    if (!path || !chr || T < 0 || !pos_bp || !ref || !alt || !info || !info_off || !format || N < 0 || (N && (!cols || !offs)))
        return QA_ERR_INVALID;
    BgzfWriter w(path, bgzf != 0, false);
    if (!w.ok()) return QA_ERR_INVALID;
    std::string line;
    const char *miss = qa_vcf_missing_entry();
    char num[16];
    for (int32_t t = 0; t < T; t++) {
        if (keep && !keep[t]) continue;
        line.clear();
        line += chr;
        line += '\t';
        snprintf(num, sizeof num, "%d", pos_bp[t]);
        line += num;
        line += "\t.\t";
        line += ref[t];
        line += '\t';
        line += alt[t];
        line += "\t.\tPASS\t";
        line += info + info_off[t];
        line += '\t';
        line += format;
        for (int32_t i = 0; i < N; i++) {
            line += '\t';
            line += cols[i] ? cols[i] + offs[i][t] : miss;
        }
        line += '\n';
        if (!w.write(line.data(), line.size())) return QA_ERR_INVALID;
    }
    return w.finish(finish != 0) ? QA_OK : QA_ERR_INVALID;
}

}  // extern "C"
