"""naive_pileup.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A deliberately naive, independent second statement of the pileup/snp count semantics
("Normative count semantics"), written as a per-read loop into a dict
with no ring buffer, no windows and no sharding, so that the C oracle (pileup_oracle.c) is
not self-certifying.  Pure Python: small inputs only.

Semantics follow pileup/snp/pileup.go:223-231 (addBase), :275-363 (alignRelevantBases),
:366-465 (addReadPair), :691-777 (filters), pileup/snp/firstread.go:176-204 (mate pairing),
pileup/snp/pileup_support.go:47-69 (clipQuals), pileup/common.go:66,94-114.
"""
import math

SEQ8_TO_ENUM = [4, 0, 1, 4, 2, 4, 4, 4, 3, 4, 4, 4, 4, 4, 4, 4]
REF_CONSUMING = {0, 2, 3, 7, 8}


def get_strand(ref, mate_ref, flags):
    if ref != mate_ref:
        return 0
    f = flags & 0xF0
    if f in (0x60, 0x90):
        return 1
    if f in (0x50, 0xA0):
        return 2
    if flags & 0x8:
        f &= 0x30
        if f == 0:
            return 1
        if f == 0x30:
            return 2
    return 0


def qual_sum_table():
    ep = [math.exp(i * (-0.1 * math.log(10.0))) for i in range(96)]
    t = [[0] * 96 for _ in range(96)]
    for i in range(96):
        for j in range(i + 1):
            prod = ep[i] * ep[j]
            p = prod / ((1.0 - ep[i]) * (1.0 - ep[j]) + prod)
            x = -10.0 * math.log10(p)
            q = int(math.floor(x + 0.5))
            q = min(max(q, 0), 95)
            t[i][j] = t[j][i] = q
    return t


class PileupError(Exception):
    pass


def _in_bed(bed_sets, ref, p):
    return p in bed_sets.get(ref, ())


def pileup(opts, bed, batch, region_filter=None, per_read=None):
    """opts: object with clip, flag_exclude, mapq, max_read_len, max_read_span, min_base_qual,
    stitch, ignore_strand, strand_req.  bed: {ref: [[s,e],...]}.  region_filter: optional
    (ref, start, end) = the unpadded single shard of -region mode (only reads whose start lies
    in it are ever seen).  Returns {(ref,pos): [depth, A+,A-,C+,C-,G+,G-,T+,T-,X+,X-]} for every
    BED position.  per_read: optional dict that receives {(ref,pos): [[(dist5p, fraglen, qual,
    strand), ...] x 4]} (appendBase, pileup.go:235-250; unstitched runs only)."""
    if per_read is not None and opts.stitch:
        raise PileupError("stitched per-read features not yet supported")
    bed_sets = {r: set(p for s, e in iv for p in range(s, e)) for r, iv in bed.items()}
    out = {(r, p): [0] * 11 for r, ps in bed_sets.items() for p in ps}
    qst = qual_sum_table()
    n = batch.n

    def cigar(i):
        o = int(batch.cigar_off[i])
        return [(int(c) & 15, int(c) >> 4) for c in batch.cigar[o:o + int(batch.n_cigar[i])]]

    def nib(i, k):
        b = int(batch.seq[int(batch.seq_off[i]) + (k >> 1)])
        return (b >> 4) if (k & 1) == 0 else (b & 15)

    def qual(i, k):
        l = int(batch.l_seq[i])
        c = opts.clip
        if c > 0 and (2 * c >= l or k < c or k >= l - c):
            return 2
        return int(batch.qual[int(batch.qual_off[i]) + k])

    def name(i):
        return bytes(batch.name[int(batch.name_off[i]):int(batch.name_off[i + 1])])

    def aligned(i):
        """[(refpos, readidx)] of M bases on BED positions; raises on a bad op that the
        reference's walk would reach."""
        ref = int(batch.ref_id[i])
        rp, qp = int(batch.pos[i]), 0
        span_end = rp + sum(l for op, l in cigar(i) if op in REF_CONSUMING)
        inbed = [p for p in range(int(batch.pos[i]), span_end) if _in_bed(bed_sets, ref, p)]
        # end of the last BED interval overlapping the span
        last_end = None
        if inbed:
            last_end = inbed[-1] + 1
            while _in_bed(bed_sets, ref, last_end):
                last_end += 1
        res = []
        for op, l in cigar(i):
            if last_end is None or rp >= last_end:
                break  # the reference's walk has returned: later ops are never looked at
            if op == 0:
                for t in range(l):
                    if _in_bed(bed_sets, ref, rp + t):
                        res.append((rp + t, qp + t))
                rp += l
                qp += l
            elif op in (1, 4):
                qp += l
            elif op in (2, 3):
                rp += l
            elif op == 5:
                pass
            else:
                raise PileupError("alignRelevantBases: unexpected CIGAR code")
        return res

    def solo(i, slot):
        for rp, k in aligned(i):
            row = out[(int(batch.ref_id[i]), rp)]
            row[0] += 1
            e = SEQ8_TO_ENUM[nib(i, k)]
            if qual(i, k) >= opts.min_base_qual or e == 4:
                row[1 + 2 * e + slot] += 1
                if per_read is not None and e != 4:
                    lists = per_read.setdefault((int(batch.ref_id[i]), rp), [[], [], [], []])
                    lists[e].append((k, int(batch.l_seq[i]), qual(i, k),
                                     get_strand(int(batch.ref_id[i]), int(batch.mate_ref_id[i]),
                                                int(batch.flags[i]))))

    def pair(b, a, slot):
        """b = later record (found a in the table); strand slot is b's."""
        ref = int(batch.ref_id[b])
        ab, aa = dict(aligned(b)), dict(aligned(a))
        if not aa or not ab:
            # one side has no relevant bases: the other is processed unstitched
            for i, d in ((b, ab), (a, aa)):
                for rp, k in d.items():
                    row = out[(ref, rp)]
                    row[0] += 1
                    e = SEQ8_TO_ENUM[nib(i, k)]
                    if qual(i, k) >= opts.min_base_qual or e == 4:
                        row[1 + 2 * e + slot] += 1
            return
        for rp in sorted(set(ab) | set(aa)):
            row = out[(ref, rp)]
            row[0] += 1
            if rp in ab and rp in aa:
                nb, na = nib(b, ab[rp]), nib(a, aa[rp])
                if nb == na:
                    e = SEQ8_TO_ENUM[nb]
                    qb, qa = qual(b, ab[rp]), qual(a, aa[rp])
                    if qb >= 96 or qa >= 96:
                        raise PileupError("qualPassTable.lookup2: index out of range")
                    if qst[qb][qa] >= opts.min_base_qual or e == 4:
                        row[1 + 2 * e + slot] += 1
                else:
                    row[1 + 8 + slot] += 1
            else:
                i, k = (b, ab[rp]) if rp in ab else (a, aa[rp])
                e = SEQ8_TO_ENUM[nib(i, k)]
                if qual(i, k) >= opts.min_base_qual or e == 4:
                    row[1 + 2 * e + slot] += 1

    stored = {}  # (ref, name, pos, mate_pos) -> FIFO list of record indices
    order = []   # stored records in arrival order (for orphan processing)
    for i in range(n):
        ref, pos = int(batch.ref_id[i]), int(batch.pos[i])
        if region_filter is not None:
            rr, rs, re_ = region_filter
            if ref != rr or pos < rs or pos >= re_:
                continue
        flags = int(batch.flags[i])
        if (opts.flag_exclude & flags) or opts.mapq > int(batch.mapq[i]) or int(batch.n_cigar[i]) == 0:
            continue
        strand = get_strand(ref, int(batch.mate_ref_id[i]), flags)
        if opts.strand_req:
            if strand != opts.strand_req:
                continue
        elif strand == 0 and not opts.ignore_strand:
            continue
        if int(batch.l_seq[i]) > opts.max_read_len:
            raise PileupError("maxReadLen")
        span = sum(l for op, l in cigar(i) if op in REF_CONSUMING)
        if span > opts.max_read_span:
            raise PileupError("maxReadSpan")
        map_end = pos + span
        if not any(_in_bed(bed_sets, ref, p) for p in range(pos, map_end)):
            continue
        slot = 0 if opts.ignore_strand else strand - 1
        if not opts.stitch:
            solo(i, slot)
            continue
        mp = int(batch.mate_pos[i])
        if strand == 0 or mp >= map_end or mp + opts.max_read_span <= pos:
            solo(i, slot)
            continue
        if pos >= mp:
            key = (ref, name(i), mp, pos)
            lst = stored.get(key)
            if lst:
                a = lst.pop(0)
                order.remove(a)
                pair(i, a, slot)
                continue
            if pos != mp:
                solo(i, slot)
                continue
        stored.setdefault((ref, name(i), pos, mp), []).append(i)
        order.append(i)
    for i in order:  # orphans: own strand slot
        s = get_strand(int(batch.ref_id[i]), int(batch.mate_ref_id[i]), int(batch.flags[i]))
        solo(i, 0 if opts.ignore_strand else s - 1)
    return out
