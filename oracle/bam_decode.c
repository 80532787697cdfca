/* bam_decode.c -- TEST INFRASTRUCTURE (CPU checker / baseline), never linked into the product.
 *
 * Plain single-threaded statement of what stands in front of processShard on the reference's
 * CPU path: the BGZF reader (github.com/biogo/hts/bgzf, a go.mod dependency that is not in
 * /root/reference; format = SAM spec section 4.1: gzip members with a BC extra subfield, each
 * inflating to at most 64 KiB, CRC-32 and size in the footer) and the record decode of
 * gbam.Unmarshal (encoding/bam/unmarshal.go:126-224): fixed 32-byte block, name, CIGAR,
 * packed bases, quals.  Output is the column layout of include/bio_pileup.h's
 * bp_record_batch, which is what oracle/pileup_oracle.c takes, so that
 * "BAM file -> rows" can be timed on the CPU next to the product's end-to-end run and the
 * product's own reader (host/bam.cpp) can be checked against an independent decoder.
 * zlib does the inflating (the reference uses a deflate library too); nothing here is shared
 * with bio_amd/csrc.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

typedef struct {
  int64_t n;
  int32_t n_ref;
  int32_t *ref_id, *pos, *mate_ref_id, *mate_pos;
  uint16_t* flags;
  uint8_t* mapq;
  uint32_t *n_cigar, *l_seq;
  uint64_t *cigar_off, *seq_off, *qual_off, *name_off; /* name_off has n+1 entries */
  uint32_t* cigar;
  uint8_t *seq, *qual, *name;
  uint64_t n_cigar_ops, n_seq_bytes, n_qual_bytes, n_name_bytes;
} orc_bam;

static int fail(char* err, int errlen, const char* msg) {
  if (err && errlen > 0) snprintf(err, (size_t)errlen, "%s", msg);
  return -1;
}
static uint32_t le32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static uint32_t le16(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }

void orc_bam_free(orc_bam* b) {
  if (!b) return;
  free(b->ref_id); free(b->pos); free(b->mate_ref_id); free(b->mate_pos); free(b->flags); free(b->mapq);
  free(b->n_cigar); free(b->l_seq); free(b->cigar_off); free(b->seq_off); free(b->qual_off); free(b->name_off);
  free(b->cigar); free(b->seq); free(b->qual); free(b->name);
  memset(b, 0, sizeof(*b));
}

/* Whole file -> inflated bytes (caller frees). */
static int bgzf_inflate_all(const char* path, uint8_t** out, size_t* out_len, char* err, int errlen) {
  FILE* f = fopen(path, "rb");
  if (!f) return fail(err, errlen, "open: no such file");
  fseek(f, 0, SEEK_END);
  long flen = ftell(f);
  fseek(f, 0, SEEK_SET);
  uint8_t* file = (uint8_t*)malloc((size_t)flen + 1);
  if (!file || fread(file, 1, (size_t)flen, f) != (size_t)flen) { fclose(f); free(file); return fail(err, errlen, "read failed"); }
  fclose(f);
  /* pass 1: block boundaries and the inflated size */
  size_t total = 0;
  for (size_t p = 0; p < (size_t)flen;) {
    if ((size_t)flen - p < 18 || file[p] != 31 || file[p + 1] != 139 || file[p + 2] != 8 || !(file[p + 3] & 4)) { free(file); return fail(err, errlen, "bgzf: invalid block header"); }
    uint32_t xlen = le16(file + p + 10);
    int bsize = -1;
    for (uint32_t q = 0; q + 4 <= xlen && p + 12 + q + 4 <= (size_t)flen;) {
      uint32_t slen = le16(file + p + 12 + q + 2);
      if (file[p + 12 + q] == 'B' && file[p + 12 + q + 1] == 'C' && slen == 2) bsize = (int)le16(file + p + 12 + q + 4);
      q += 4 + slen;
    }
    if (bsize < 0 || p + (size_t)bsize + 1 > (size_t)flen) { free(file); return fail(err, errlen, "bgzf: bad block size"); }
    total += le32(file + p + (size_t)bsize + 1 - 4);
    p += (size_t)bsize + 1;
  }
  uint8_t* data = (uint8_t*)malloc(total + 1);
  if (!data) { free(file); return fail(err, errlen, "out of memory"); }
  size_t o = 0;
  for (size_t p = 0; p < (size_t)flen;) {
    uint32_t xlen = le16(file + p + 10);
    int bsize = -1;
    for (uint32_t q = 0; q + 4 <= xlen;) {
      uint32_t slen = le16(file + p + 12 + q + 2);
      if (file[p + 12 + q] == 'B' && file[p + 12 + q + 1] == 'C' && slen == 2) bsize = (int)le16(file + p + 12 + q + 4);
      q += 4 + slen;
    }
    size_t tot = (size_t)bsize + 1, hdr = 12 + (size_t)xlen;
    uint32_t isize = le32(file + p + tot - 4), crc = le32(file + p + tot - 8);
    if (isize) {
      z_stream zs;
      memset(&zs, 0, sizeof zs);
      if (inflateInit2(&zs, -15) != Z_OK) { free(file); free(data); return fail(err, errlen, "inflateInit2"); }
      zs.next_in = file + p + hdr;
      zs.avail_in = (uInt)(tot - hdr - 8);
      zs.next_out = data + o;
      zs.avail_out = isize;
      int rc = inflate(&zs, Z_FINISH);
      inflateEnd(&zs);
      if (rc != Z_STREAM_END || zs.avail_out != 0 || (uint32_t)crc32(crc32(0, NULL, 0), data + o, isize) != crc) {
        free(file); free(data);
        return fail(err, errlen, "bgzf: corrupt block");
      }
      o += isize;
    }
    p += tot;
  }
  free(file);
  *out = data;
  *out_len = o;
  return 0;
}

int orc_bam_decode(const char* path, orc_bam* b, char* err, int errlen) {
  memset(b, 0, sizeof(*b));
  uint8_t* d = NULL;
  size_t len = 0;
  if (bgzf_inflate_all(path, &d, &len, err, errlen)) return -1;
  if (len < 12 || memcmp(d, "BAM\1", 4) != 0) { free(d); return fail(err, errlen, "bam: magic number mismatch"); }
  size_t p = 8 + (size_t)le32(d + 4);
  if (p + 4 > len) { free(d); return fail(err, errlen, "bam: truncated header"); }
  b->n_ref = (int32_t)le32(d + p);
  p += 4;
  for (int32_t r = 0; r < b->n_ref; r++) {
    if (p + 4 > len) { free(d); return fail(err, errlen, "bam: truncated header"); }
    p += 8 + (size_t)le32(d + p);
  }
  if (p > len) { free(d); return fail(err, errlen, "bam: truncated header"); }
  /* pass 1: sizes */
  const size_t first = p;
  int64_t n = 0;
  uint64_t nc = 0, ns = 0, nq = 0, nn = 0;
  while (p < len) {
    if (len - p < 36) { free(d); return fail(err, errlen, "bam: truncated record"); }
    uint32_t bs = le32(d + p), l_name = d[p + 12], ncig = le16(d + p + 16), ls = le32(d + p + 20);
    uint64_t need = 32 + (uint64_t)l_name + 4 * (uint64_t)ncig + ((uint64_t)ls + 1) / 2 + ls;
    if (bs < 32 || len - p - 4 < bs) { free(d); return fail(err, errlen, "bam: truncated record"); }
    if (need > bs) { free(d); return fail(err, errlen, "bam: record fields exceed block size"); }
    n++;
    nc += ncig;
    ns += ((uint64_t)ls + 1) / 2;
    nq += ls;
    nn += l_name ? l_name - 1 : 0;
    p += 4 + (size_t)bs;
  }
  b->n = n;
#define ALLOC(field, count) b->field = malloc(((size_t)(count) + 1) * sizeof(*b->field))
  ALLOC(ref_id, n); ALLOC(pos, n); ALLOC(mate_ref_id, n); ALLOC(mate_pos, n); ALLOC(flags, n); ALLOC(mapq, n);
  ALLOC(n_cigar, n); ALLOC(l_seq, n); ALLOC(cigar_off, n); ALLOC(seq_off, n); ALLOC(qual_off, n); ALLOC(name_off, n + 1);
  ALLOC(cigar, nc); ALLOC(seq, ns); ALLOC(qual, nq); ALLOC(name, nn);
#undef ALLOC
  if (!b->ref_id || !b->pos || !b->mate_ref_id || !b->mate_pos || !b->flags || !b->mapq || !b->n_cigar || !b->l_seq ||
      !b->cigar_off || !b->seq_off || !b->qual_off || !b->name_off || !b->cigar || !b->seq || !b->qual || !b->name) {
    free(d);
    orc_bam_free(b);
    return fail(err, errlen, "out of memory");
  }
  /* pass 2: the fields */
  uint64_t co = 0, so = 0, qo = 0, no = 0;
  p = first;
  for (int64_t i = 0; i < n; i++) {
    const uint8_t* r = d + p + 4;
    uint32_t bs = le32(d + p), l_name = r[8], ncig = le16(r + 12), ls = le32(r + 16);
    b->ref_id[i] = (int32_t)le32(r);
    b->pos[i] = (int32_t)le32(r + 4);
    b->mapq[i] = r[9];
    b->n_cigar[i] = ncig;
    b->flags[i] = (uint16_t)le16(r + 14);
    b->l_seq[i] = ls;
    b->mate_ref_id[i] = (int32_t)le32(r + 20);
    b->mate_pos[i] = (int32_t)le32(r + 24);
    b->name_off[i] = no;
    if (l_name) memcpy(b->name + no, r + 32, l_name - 1);
    no += l_name ? l_name - 1 : 0;
    b->cigar_off[i] = co;
    for (uint32_t k = 0; k < ncig; k++) b->cigar[co + k] = le32(r + 32 + l_name + 4 * k);
    co += ncig;
    b->seq_off[i] = so;
    memcpy(b->seq + so, r + 32 + l_name + 4 * (size_t)ncig, ((size_t)ls + 1) / 2);
    so += ((uint64_t)ls + 1) / 2;
    b->qual_off[i] = qo;
    memcpy(b->qual + qo, r + 32 + l_name + 4 * (size_t)ncig + ((size_t)ls + 1) / 2, ls);
    qo += ls;
    p += 4 + (size_t)bs;
  }
  b->name_off[n] = no;
  b->n_cigar_ops = nc;
  b->n_seq_bytes = ns;
  b->n_qual_bytes = nq;
  b->n_name_bytes = nn;
  free(d);
  return 0;
}
