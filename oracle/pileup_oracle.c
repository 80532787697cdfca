/*
 * pileup_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C, single-threaded CPU restatement of the grailbio/bio pileup/snp hot path, used
 * only as the checker for the HIP implementation (tests/, __graft_entry__.smoke(), and the
 * cpu_baseline leg of bench.py).  Nothing under bio_amd/ may link, load or call it.
 *
 * It follows the reference's algorithm step by step -- per-job ring buffer, firstread mate
 * table, UnionScanner flush -- NOT the GPU design, so that the two are independent:
 *   processShard / pileupSNPMain job loop   pileup/snp/pileup.go:661-810, :862-939
 *   addBase / addUnstitchedSegment          pileup/snp/pileup.go:223-267
 *   alignRelevantBases                      pileup/snp/pileup.go:275-363
 *   addReadPair (stitch merge)              pileup/snp/pileup.go:366-465
 *   addOrphanReads                          pileup/snp/pileup.go:469-500
 *   flushToInternal / flushTo               pileup/snp/pileup.go:513-585
 *   finishRef / nextRef                     pileup/snp/pileup.go:624-651
 *   firstreadSNPTable                       pileup/snp/firstread.go:79-211
 *   qualSumTable / qualPassTable            pileup/snp/qual.go:35-81
 *   clipQuals, writeEmptyEntries, flushEmptyContigs, intersectionIsEmpty
 *                                           pileup/snp/pileup_support.go:47-123
 *   GetStrand, Seq8ToEnumTable              pileup/common.go:66, :94-114
 *   UnpackSeq (scalar definition)           biosimd/biosimd_generic.go:90-106
 *   BEDUnion.{Intersects,IntersectsByID,OverlapByID,Subset}, UnionScanner
 *                                           interval/bedunion.go:182-300, :753-852,
 *                                           interval/endpoint_index.go:60-196
 *   circular.NextExp2 / Bitmap              circular/circular.go, circular/bitmap.go:52-194
 *   Shard padding / GetPositionBasedShards / iterator range
 *                                           encoding/bam/shard.go:91-120, :320-352,
 *                                           encoding/bamprovider/bamprovider.go:286-292, :417-435
 * Third-party pieces restated from their published definitions: murmur3 x86_32 seed 0
 * (github.com/spaolacci/murmur3 Sum32, used at firstread.go:102) and sam.Cigar.Lengths()
 * (github.com/grailbio/hts v1.0.2-0.20200303061016-030e6b9f995e; reference-consuming ops
 * are M, D, N, =, X).
 *
 * Pinned by: tests/golden/pileup_kat.json (pileup_test.go:56-315), align_kat.json
 * (pileup_internal_test.go:69-142), qual KAT (qual_internal_test.go:20-55), shard KATs
 * (encoding/bam/shard_test.go:12-57).  See tests/test_oracle_golden.py.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/bio_pileup.h"

#define POS_MAX 2147483647
#define NQUAL 96
#define HTABLE 1024
#define HWORDS (HTABLE / 64)

/* ---------------------------------------------------------------- error plumbing */
typedef struct {
  char* buf;
  int len;
  int set;
} errbuf;

#define FAIL(e, ...)                                   \
  do {                                                 \
    if (!(e)->set) {                                   \
      snprintf((e)->buf, (size_t)(e)->len, __VA_ARGS__); \
      (e)->set = 1;                                    \
    }                                                  \
    return -1;                                         \
  } while (0)

/* ---------------------------------------------------------------- interval package */
/* SearchPosTypes: sort.Search(len(a), a[i] >= x)  (endpoint_index.go:60-62) */
static uint32_t search_pos(const int32_t* a, uint32_t n, int32_t x) {
  uint32_t lo = 0, hi = n;
  while (lo < hi) {
    uint32_t mid = (lo + hi) >> 1;
    if (a[mid] >= x)
      hi = mid;
    else
      lo = mid + 1;
  }
  return lo;
}

typedef struct {
  int n_ref;
  int32_t** ep; /* idMap: NULL = nil */
  uint32_t* n;
  uint8_t* owned;
} bedunion;

static void bed_free(bedunion* b) {
  if (!b->ep) return;
  for (int i = 0; i < b->n_ref; i++)
    if (b->owned[i]) free(b->ep[i]);
  free(b->ep);
  free(b->n);
  free(b->owned);
  memset(b, 0, sizeof(*b));
}

static void bed_alloc(bedunion* b, int n_ref) {
  b->n_ref = n_ref;
  b->ep = (int32_t**)calloc((size_t)n_ref + 1, sizeof(int32_t*));
  b->n = (uint32_t*)calloc((size_t)n_ref + 1, sizeof(uint32_t));
  b->owned = (uint8_t*)calloc((size_t)n_ref + 1, 1);
}

/* IntersectsByID (bedunion.go:219-254); the sequential-search cache there only changes
 * speed, the answer is SearchPosTypes(endpoints, startPos+1). */
static int bed_intersects_by_id(const bedunion* b, int ref, int32_t start, int32_t limit) {
  const int32_t* e = b->ep[ref];
  if (!e) return 0;
  uint32_t n = b->n[ref];
  uint32_t idx = search_pos(e, n, start + 1);
  if (idx & 1) return 1;
  return idx < n && e[idx] < limit;
}

/* OverlapByID (bedunion.go:266-300): endpoints of all intervals overlapping [start,limit). */
static void bed_overlap_by_id(const bedunion* b, int ref, int32_t start, int32_t limit,
                              const int32_t** out, uint32_t* n_out) {
  const int32_t* e = b->ep[ref];
  if (!e) {
    *out = NULL;
    *n_out = 0;
    return;
  }
  uint32_t n = b->n[ref];
  uint32_t rs = search_pos(e, n, start + 1) & ~1u;
  uint32_t re = (search_pos(e, n, limit) + 1) & ~1u;
  *out = e + rs;
  *n_out = re > rs ? re - rs : 0;
}

/* Intersects (bedunion.go:182-214), multi-reference region. */
static int bed_intersects(const bedunion* b, int sref, int32_t spos, int lref, int32_t lpos) {
  const int32_t* e = b->ep[sref];
  if (e) {
    uint32_t n = b->n[sref];
    uint32_t idx = search_pos(e, n, spos + 1);
    if (sref < lref) {
      if (idx < n) return 1;
    } else {
      if (idx & 1) return 1;
      return idx < n && lpos > e[idx];
    }
  }
  if (sref == lref) return 0;
  for (int r = sref + 1; r < lref; r++)
    if (b->ep[r]) return 1;
  if (b->ep[lref]) return b->ep[lref][0] < lpos;
  return 0;
}

static void bed_set_copy(bedunion* d, int ref, const int32_t* src, uint32_t n) {
  d->ep[ref] = (int32_t*)malloc(sizeof(int32_t) * (n ? n : 1));
  memcpy(d->ep[ref], src, sizeof(int32_t) * n);
  d->n[ref] = n;
  d->owned[ref] = 1;
}

/* Subset (bedunion.go:753-852).  Restated case by case. */
static void bed_subset(const bedunion* u, int sref, int32_t spos, int lref, int32_t lpos,
                       bedunion* out) {
  bed_alloc(out, u->n_ref);
  const int32_t* se = u->ep[sref];
  if (se) {
    uint32_t sn = u->n[sref];
    uint32_t idx_start = search_pos(se, sn, spos + 1);
    int new_alloc = (idx_start & 1) && (se[idx_start - 1] != spos);
    if (sref < lref) {
      if (idx_start < sn) {
        if (new_alloc) {
          uint32_t n_elem = sn - idx_start + 1;
          int32_t* t = (int32_t*)malloc(sizeof(int32_t) * n_elem);
          t[0] = spos;
          memcpy(t + 1, se + idx_start, sizeof(int32_t) * (sn - idx_start));
          out->ep[sref] = t;
          out->n[sref] = n_elem;
          out->owned[sref] = 1;
        } else {
          uint32_t b0 = idx_start & ~1u;
          bed_set_copy(out, sref, se + b0, sn - b0);
        }
      }
    } else {
      uint32_t idx_end = search_pos(se, sn, lpos + 1);
      if ((idx_end & 1) && (se[idx_end - 1] != lpos)) new_alloc = 1;
      if (new_alloc) {
        uint32_t sc = idx_start | 1, ec = (idx_end - 1) | 1;
        uint32_t n_elem = ec - sc + 2;
        int32_t* t = (int32_t*)malloc(sizeof(int32_t) * n_elem);
        t[0] = (idx_start & 1) ? spos : se[idx_start];
        memcpy(t + 1, se + sc, sizeof(int32_t) * (ec - sc));
        t[n_elem - 1] = (idx_end & 1) ? lpos : se[ec];
        out->ep[sref] = t;
        out->n[sref] = n_elem;
        out->owned[sref] = 1;
      } else {
        uint32_t sf = idx_start & ~1u, ef = idx_end & ~1u;
        if (sf == ef) return;
        bed_set_copy(out, sref, se + sf, ef - sf);
      }
    }
  }
  if (sref == lref) return;
  for (int r = sref + 1; r < lref; r++)
    if (u->ep[r]) bed_set_copy(out, r, u->ep[r], u->n[r]);
  const int32_t* le = u->ep[lref];
  if (le) {
    uint32_t ln = u->n[lref];
    uint32_t idx_end = search_pos(le, ln, lpos + 1);
    if (idx_end == 0) return;
    if ((idx_end & 1) && (le[idx_end - 1] != lpos)) {
      int32_t* t = (int32_t*)malloc(sizeof(int32_t) * (idx_end + 1));
      memcpy(t, le, sizeof(int32_t) * idx_end);
      t[idx_end] = lpos;
      out->ep[lref] = t;
      out->n[lref] = idx_end + 1;
      out->owned[lref] = 1;
    } else {
      bed_set_copy(out, lref, le, idx_end & ~1u);
    }
  }
}

/* UnionScanner (endpoint_index.go:141-196) */
typedef struct {
  const int32_t* ep;
  uint32_t n;
  int32_t pos;
  uint32_t idx;
} uscanner;

static uscanner us_new(const int32_t* ep, uint32_t n) {
  uscanner s = {ep, n, POS_MAX, 0};
  if (n >= 1) {
    s.pos = ep[0];
    s.idx = 1;
  }
  return s;
}

static int us_scan(uscanner* s, int32_t* start, int32_t* end, int32_t limit) {
  if (s->pos >= limit) return 0;
  *start = s->pos;
  int32_t iend = s->ep[s->idx];
  if (iend > limit) {
    s->pos = limit;
    *end = limit;
    return 1;
  }
  *end = iend;
  s->idx++;
  if (s->idx >= s->n) {
    s->pos = POS_MAX;
  } else {
    s->pos = s->ep[s->idx];
    s->idx++;
  }
  return 1;
}

/* ---------------------------------------------------------------- common.go tables */
static const uint8_t kSeq8ToEnum[16] = {4, 0, 1, 4, 2, 4, 4, 4, 3, 4, 4, 4, 4, 4, 4, 4};

/* GetStrand (pileup/common.go:94-114).  sam flag bits: Reverse 0x10, MateReverse 0x20,
 * Read1 0x40, Read2 0x80, MateUnmapped 0x8. */
int orc_get_strand(int32_t ref, int32_t mate_ref, uint16_t flags) {
  if (ref != mate_ref) return 0;
  uint16_t f = flags & (0x10 | 0x20 | 0x40 | 0x80);
  if (f == (0x20 | 0x40) || f == (0x10 | 0x80)) return 1;
  if (f == (0x10 | 0x40) || f == (0x20 | 0x80)) return 2;
  if ((flags & 0x8) == 0x8) {
    f &= 0x10 | 0x20;
    if (f == 0) return 1;
    if (f == (0x10 | 0x20)) return 2;
  }
  return 0;
}

/* circular.NextExp2: smallest power of 2 >= x (circular/circular.go:7-10) */
int32_t orc_next_exp2(int32_t x) {
  int32_t r = 1;
  while (r < x) r <<= 1;
  return r;
}

/* biosimd.UnpackSeq scalar definition (biosimd_generic.go:90-106; biosimd_test.go:16-29) */
void orc_unpack_seq(uint8_t* dst, const uint8_t* src, uint32_t l) {
  uint32_t full = l >> 1;
  for (uint32_t i = 0; i < full; i++) {
    dst[2 * i] = src[i] >> 4;
    dst[2 * i + 1] = src[i] & 15;
  }
  if (l & 1) dst[l - 1] = src[full] >> 4;
}

/* ---------------------------------------------------------------- qual.go */
static uint8_t g_qual_sum[NQUAL][NQUAL];
static int g_qual_sum_init = 0;
static double g_qual_min_margin = 1.0;

static void qual_sum_init(void) {
  if (g_qual_sum_init) return;
  double ep[NQUAL];
  const double ln10 = 2.30258509299404568401799145468436421;   /* math.Ln10 */
  const double log10e = 0.43429448190325182765112891891660508; /* math.Log10E */
  for (int i = 0; i < NQUAL; i++) ep[i] = exp((double)i * (-0.1 * ln10));
  for (int i = 0; i < NQUAL; i++) {
    double e1 = ep[i];
    for (int j = 0; j <= i; j++) {
      double e2 = ep[j];
      double prod = e1 * e2;
      double np = prod / ((1.0 - e1) * (1.0 - e2) + prod);
      double real = log(np) * (-10.0 * log10e);
      double r = round(real); /* math.Round: half away from zero, same as C round() */
      double frac = fabs(real - floor(real) - 0.5);
      if (frac < g_qual_min_margin) g_qual_min_margin = frac;
      int q = (int)r; /* byte(...) of a value in [0,191) */
      if (q < 0) q = 0;
      if (q >= NQUAL) q = NQUAL - 1;
      g_qual_sum[i][j] = (uint8_t)q;
      g_qual_sum[j][i] = (uint8_t)q;
    }
  }
  g_qual_sum_init = 1;
}

/* Exposed for tests: qualSumTable[i][j] and the smallest distance of any pre-rounding
 * value from a .5 boundary (libm-vs-Go-math parity margin). */
int orc_qual_sum(int i, int j) {
  qual_sum_init();
  return g_qual_sum[i][j];
}
double orc_qual_round_margin(void) {
  qual_sum_init();
  return g_qual_min_margin;
}
/* newQualPassTable(minBaseQual).lookup2(q1,q2)  (qual.go:62-81); -1 if minBaseQual>=nQual */
int orc_qual_pass(int min_base_qual, int q1, int q2) {
  if (min_base_qual >= NQUAL) return -1;
  qual_sum_init();
  return g_qual_sum[q1][q2] >= min_base_qual;
}

/* ---------------------------------------------------------------- murmur3 x86_32, seed 0 */
static uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
uint32_t orc_murmur3_32(const uint8_t* data, size_t len) {
  const uint32_t c1 = 0xcc9e2d51u, c2 = 0x1b873593u;
  uint32_t h = 0;
  size_t nb = len / 4;
  for (size_t i = 0; i < nb; i++) {
    uint32_t k = (uint32_t)data[4 * i] | ((uint32_t)data[4 * i + 1] << 8) |
                 ((uint32_t)data[4 * i + 2] << 16) | ((uint32_t)data[4 * i + 3] << 24);
    k *= c1;
    k = rotl32(k, 15);
    k *= c2;
    h ^= k;
    h = rotl32(h, 13);
    h = h * 5 + 0xe6546b64u;
  }
  const uint8_t* tail = data + nb * 4;
  uint32_t k = 0;
  switch (len & 3) {
    case 3: k ^= (uint32_t)tail[2] << 16; /* fallthrough */
    case 2: k ^= (uint32_t)tail[1] << 8;  /* fallthrough */
    case 1:
      k ^= tail[0];
      k *= c1;
      k = rotl32(k, 15);
      k *= c2;
      h ^= k;
  }
  h ^= (uint32_t)len;
  h ^= h >> 16;
  h *= 0x85ebca6bu;
  h ^= h >> 13;
  h *= 0xc2b2ae35u;
  h ^= h >> 16;
  return h;
}

/* ---------------------------------------------------------------- circular.Bitmap */
typedef struct {
  uint64_t* bits; /* nCirc * HWORDS */
  uint8_t* word_pops;
  int32_t n_circ;
  int32_t first_pos, last_pos;
} cbitmap;

static void cb_init(cbitmap* b, int32_t n_circ) {
  b->bits = (uint64_t*)calloc((size_t)n_circ * HWORDS, sizeof(uint64_t));
  b->word_pops = (uint8_t*)calloc((size_t)n_circ, 1);
  b->n_circ = n_circ;
  b->first_pos = POS_MAX;
  b->last_pos = -1;
}
static void cb_free(cbitmap* b) {
  free(b->bits);
  free(b->word_pops);
}
static void cb_set(cbitmap* b, int32_t pos, int32_t circ, uint32_t col) {
  uint64_t* row = b->bits + (size_t)circ * HWORDS;
  uint64_t w = row[col / 64];
  if (w == 0) b->word_pops[circ]++;
  row[col / 64] = w | ((uint64_t)1 << (col % 64));
  if (b->first_pos > pos) b->first_pos = pos;
  if (b->last_pos < pos) b->last_pos = pos;
}
/* first nonempty row position in [pos, stop), or stop (bitmap.go:131-147) */
static int32_t cb_first_nonempty(const cbitmap* b, int32_t pos, int32_t stop) {
  for (int32_t p = pos; p < stop; p++)
    if (b->word_pops[p & (b->n_circ - 1)]) return p;
  return stop;
}
/* indexNonzeroRev(arr, start, stop): last nonempty row position in [stop, start], else
 * stop (bitmap.go:17-43) */
static int32_t cb_last_nonempty(const cbitmap* b, int32_t start, int32_t stop) {
  for (int32_t p = start; p > stop; p--)
    if (b->word_pops[p & (b->n_circ - 1)]) return p;
  return stop;
}
static void cb_clear(cbitmap* b, int32_t pos, int32_t circ, uint32_t col) {
  uint64_t* row = b->bits + (size_t)circ * HWORDS;
  uint64_t w = row[col / 64] & ~((uint64_t)1 << (col % 64));
  row[col / 64] = w;
  if (w == 0) {
    uint8_t wp = (uint8_t)(b->word_pops[circ] - 1);
    b->word_pops[circ] = wp;
    if (wp == 0) {
      if (pos == b->first_pos) {
        if (pos == b->last_pos) {
          b->first_pos = POS_MAX;
          b->last_pos = -1;
        } else {
          b->first_pos = cb_first_nonempty(b, pos + 1, b->last_pos);
        }
      } else if (pos == b->last_pos) {
        b->last_pos = cb_last_nonempty(b, pos - 1, b->first_pos);
      }
    }
  }
}
/* NewRowScanner (bitmap.go:178-194): detaches the first nonempty row; returns its circPos. */
static int32_t cb_take_first_row(cbitmap* b) {
  int32_t pos = b->first_pos;
  int32_t circ = pos & (b->n_circ - 1);
  if (pos < b->last_pos) {
    b->first_pos = cb_first_nonempty(b, pos + 1, b->last_pos);
  } else {
    b->first_pos = POS_MAX;
    b->last_pos = -1;
  }
  b->word_pops[circ] = 0;
  return circ;
}

/* ---------------------------------------------------------------- firstread table */
typedef struct {
  int64_t rec;
  int32_t map_end;
} aug_rec;
typedef struct {
  aug_rec* v;
  uint32_t n, cap;
} bucket;
typedef struct {
  bucket* buckets; /* nCirc * HTABLE */
  cbitmap nonempty;
} firstreads;

static void fr_init(firstreads* f, int32_t n_circ) {
  f->buckets = (bucket*)calloc((size_t)n_circ * HTABLE, sizeof(bucket));
  cb_init(&f->nonempty, n_circ);
}
static void fr_free(firstreads* f) {
  if (!f->buckets) return;
  size_t nb = (size_t)f->nonempty.n_circ * HTABLE;
  for (size_t i = 0; i < nb; i++) free(f->buckets[i].v);
  free(f->buckets);
  cb_free(&f->nonempty);
  f->buckets = NULL;
}

/* ---------------------------------------------------------------- job state */
/* PerReadFeatures (row.go:22-31) */
typedef struct {
  uint16_t dist5p, fraglen;
  uint8_t qual, strand;
} orc_feat;

typedef struct {
  uint32_t depth;
  uint32_t counts[5][2];
  /* PerRead[4][]PerReadFeatures (row.go:48-52); only used when perReadNeeded */
  orc_feat* pr[4];
  uint32_t npr[4], cpr[4];
} payload;

typedef struct {
  int64_t rec;
  int32_t map_end;
  uint8_t* seq8;
  uint8_t* qual; /* private mutable copy (clipQuals mutates the record) */
} read_snp;

typedef struct {
  int32_t pos_in_ref, pos_in_read;
} aligned_pos;

typedef struct {
  /* outputs */
  int64_t cap, n;
  int32_t *out_ref, *out_pos;
  uint32_t* out_vals;
  /* per-read lists, in row order: list b of row i = feat[off[4i+b] .. off[4i+b+1]) */
  uint64_t* out_off;
  orc_feat* out_feat;
  int64_t feat_cap, n_feat;
} rowsink;

typedef struct {
  const bp_opts* o;
  const bp_record_batch* recs;
  bedunion bed_part;
  uint8_t qpt[NQUAL][NQUAL];
  /* pileupMutable */
  payload* ring;
  int32_t n_circ;
  aligned_pos* ab[2];
  uint32_t nab[2];
  firstreads first;
  int32_t end_max;
  uscanner wscan;
  /* refContext */
  int ref_id;
  /* pileupShardContext */
  int prev_limit_id;
  int64_t prev_limit_pos;
  int shard_overlap;
  read_snp pair[2];
  read_snp orphan;
  rowsink* sink;
  errbuf* err;
  int per_read; /* pCtx.perReadNeeded (pileup.go:888) */
} job;

static int sink_append(job* j, int32_t ref, int32_t pos, const payload* p) {
  rowsink* s = j->sink;
  if (s->n >= s->cap) FAIL(j->err, "oracle: row capacity %lld exceeded", (long long)s->cap);
  s->out_ref[s->n] = ref;
  s->out_pos[s->n] = pos;
  uint32_t* v = s->out_vals + s->n * BP_ROW_U32;
  if (p) {
    v[0] = p->depth;
    for (int b = 0; b < 5; b++) {
      v[1 + 2 * b] = p->counts[b][0];
      v[2 + 2 * b] = p->counts[b][1];
    }
  } else {
    memset(v, 0, sizeof(uint32_t) * BP_ROW_U32);
  }
  if (s->out_off) {
    for (int b = 0; b < 4; b++) {
      s->out_off[4 * s->n + b] = (uint64_t)s->n_feat;
      uint32_t m = p ? p->npr[b] : 0;
      if (s->n_feat + m > s->feat_cap) FAIL(j->err, "oracle: feature capacity %lld exceeded", (long long)s->feat_cap);
      if (m) memcpy(s->out_feat + s->n_feat, p->pr[b], sizeof(orc_feat) * m);
      s->n_feat += m;
    }
    s->out_off[4 * s->n + 4] = (uint64_t)s->n_feat;
  }
  s->n++;
  return 0;
}

static const uint8_t* rec_name(const bp_record_batch* r, int64_t i, size_t* len) {
  if (!r->name || !r->name_off) {
    *len = 0;
    return (const uint8_t*)"";
  }
  *len = (size_t)(r->name_off[i + 1] - r->name_off[i]);
  return r->name + r->name_off[i];
}

/* sam.Cigar.Lengths() reference span: ops consuming the reference are M,D,N,=,X */
static int32_t cigar_span(const bp_record_batch* r, int64_t i) {
  const uint32_t* c = r->cigar + r->cigar_off[i];
  int64_t span = 0;
  for (uint32_t k = 0; k < r->n_cigar[i]; k++) {
    uint32_t op = c[k] & 15, len = c[k] >> 4;
    if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) span += len;
  }
  return span > POS_MAX ? POS_MAX : (int32_t)span;
}

/* convertSamr (firstread.go:60-67) + private qual copy */
static void convert_samr(job* j, read_snp* rd, int64_t i) {
  const bp_record_batch* r = j->recs;
  uint32_t l = r->l_seq[i];
  if (l) {
    orc_unpack_seq(rd->seq8, r->seq + r->seq_off[i], l);
    memcpy(rd->qual, r->qual + r->qual_off[i], l);
  }
  rd->rec = i;
}

/* clipQuals (pileup_support.go:51-69) */
static void clip_quals(uint8_t* q, int32_t len, int32_t n) {
  if (n == 0) return;
  if (n * 2 >= len) {
    for (int32_t i = 0; i < len; i++) q[i] = 2;
  } else {
    for (int32_t i = 0; i < n; i++) q[i] = 2;
    for (int32_t i = len - n; i < len; i++) q[i] = 2;
  }
}

/* alignRelevantBases (pileup.go:275-363) */
static int align_relevant_bases(job* j, int which, const read_snp* rd) {
  const bp_record_batch* r = j->recs;
  aligned_pos* out = j->ab[which];
  uint32_t n = 0;
  j->nab[which] = 0;
  int64_t i = rd->rec;
  int32_t pos_in_ref = r->pos[i];
  const int32_t* ri;
  uint32_t nri;
  bed_overlap_by_id(&j->bed_part, r->ref_id[i], pos_in_ref, rd->map_end, &ri, &nri);
  if (nri == 0) return 0; /* cannot happen for reads that passed IntersectsByID */
  uint32_t ri_idx = 0;
  int32_t nis = ri[0], nie = ri[1];
  const uint32_t* cg = r->cigar + r->cigar_off[i];
  int32_t pos_in_read = 0;
  for (uint32_t k = 0; k < r->n_cigar[i]; k++) {
    int32_t clen = (int32_t)(cg[k] >> 4);
    uint32_t op = cg[k] & 15;
    switch (op) {
      case 0: { /* M */
        int32_t next_ref = pos_in_ref + clen;
        if (next_ref > nis) {
          int32_t off = 0;
          for (;;) {
            if (pos_in_ref + off < nis) off = nis - pos_in_ref;
            if (nie > next_ref) {
              for (; off < clen; off++) {
                out[n].pos_in_ref = pos_in_ref + off;
                out[n].pos_in_read = pos_in_read + off;
                n++;
              }
              break;
            }
            int32_t stop = nie - pos_in_ref;
            for (; off < stop; off++) {
              out[n].pos_in_ref = pos_in_ref + off;
              out[n].pos_in_read = pos_in_read + off;
              n++;
            }
            ri_idx += 2;
            if (ri_idx == nri) {
              j->nab[which] = n;
              return 0;
            }
            nis = ri[ri_idx];
            nie = ri[ri_idx + 1];
          }
        }
        pos_in_ref = next_ref;
        pos_in_read += clen;
        break;
      }
      case 1: /* I */
        pos_in_read += clen;
        break;
      case 3: /* N */
      case 2: /* D */
        pos_in_ref += clen;
        while (pos_in_ref >= nie) {
          ri_idx += 2;
          if (ri_idx == nri) {
            j->nab[which] = n;
            return 0;
          }
          nie = ri[ri_idx + 1];
        }
        nis = ri[ri_idx];
        break;
      case 4: /* S */
        pos_in_read += clen;
        break;
      case 5: /* H */
        break;
      default:
        j->nab[which] = n;
        FAIL(j->err, "alignRelevantBases: unexpected CIGAR code %u (record %lld)", op,
             (long long)i);
    }
  }
  j->nab[which] = n;
  return 0;
}

/* Exposed for the alignRelevantBases KAT (pileup_internal_test.go:69-160): one contig, one
 * read.  Returns number of pairs, or -1 on the unexpected-CIGAR error. */
int orc_align_relevant_bases(const int32_t* endpoints, int32_t n_endpoints, int32_t pos,
                             const uint32_t* cigar, int32_t n_cigar, int32_t cap,
                             int32_t* out_ref, int32_t* out_read) {
  char ebuf[256];
  errbuf e = {ebuf, sizeof(ebuf), 0};
  job j;
  memset(&j, 0, sizeof(j));
  j.err = &e;
  bed_alloc(&j.bed_part, 1);
  bed_set_copy(&j.bed_part, 0, endpoints, (uint32_t)n_endpoints);
  int32_t rid = 0, p = pos;
  uint32_t nc = (uint32_t)n_cigar;
  uint64_t zero = 0;
  bp_record_batch rb;
  memset(&rb, 0, sizeof(rb));
  rb.n = 1;
  rb.ref_id = &rid;
  rb.pos = &p;
  rb.n_cigar = &nc;
  rb.cigar_off = &zero;
  rb.cigar = cigar;
  j.recs = &rb;
  read_snp rd;
  memset(&rd, 0, sizeof(rd));
  rd.rec = 0;
  rd.map_end = pos + cigar_span(&rb, 0);
  /* the reference appends without bound; size by the span */
  int32_t span = rd.map_end - pos;
  j.ab[0] = (aligned_pos*)malloc(sizeof(aligned_pos) * (size_t)(span + 1));
  int rc = align_relevant_bases(&j, 0, &rd);
  int32_t n = (int32_t)j.nab[0];
  if (rc == 0) {
    for (int32_t k = 0; k < n && k < cap; k++) {
      out_ref[k] = j.ab[0][k].pos_in_ref;
      out_read[k] = j.ab[0][k].pos_in_read;
    }
  }
  free(j.ab[0]);
  bed_free(&j.bed_part);
  return rc == 0 ? n : -1;
}

/* addBase (pileup.go:223-231) */
static inline void add_base(job* j, int32_t circ, int32_t pos_in_read, int32_t is_minus,
                            const uint8_t* seq, const uint8_t* qual) {
  payload* row = &j->ring[circ];
  row->depth++;
  uint8_t base = kSeq8ToEnum[seq[pos_in_read]];
  if (qual[pos_in_read] >= (uint8_t)j->o->min_base_qual || base == 4) row->counts[base][is_minus]++;
}

/* appendBase (pileup.go:235-250) */
static inline void append_base(job* j, int32_t circ, int32_t pos_in_read, int32_t is_minus,
                               const uint8_t* seq, const uint8_t* qual, uint32_t qual_len,
                               uint8_t strand_byte) {
  payload* row = &j->ring[circ];
  row->depth++;
  uint8_t base = kSeq8ToEnum[seq[pos_in_read]];
  if (base == 4) {
    row->counts[base][is_minus]++;
  } else if (qual[pos_in_read] >= (uint8_t)j->o->min_base_qual) {
    row->counts[base][is_minus]++;
    if (row->npr[base] == row->cpr[base]) {
      row->cpr[base] = row->cpr[base] ? 2 * row->cpr[base] : 8;
      row->pr[base] = (orc_feat*)realloc(row->pr[base], sizeof(orc_feat) * row->cpr[base]);
    }
    orc_feat* f = &row->pr[base][row->npr[base]++];
    f->dist5p = (uint16_t)pos_in_read;
    f->fraglen = (uint16_t)qual_len;
    f->qual = qual[pos_in_read];
    f->strand = strand_byte;
  }
}

/* addUnstitchedSegment (pileup.go:254-267) */
static void add_unstitched(job* j, const read_snp* rd, int32_t is_minus, const aligned_pos* ab,
                           uint32_t n) {
  int32_t mask = j->n_circ - 1;
  if (!j->per_read) {
    for (uint32_t k = 0; k < n; k++)
      add_base(j, ab[k].pos_in_ref & mask, ab[k].pos_in_read, is_minus, rd->seq8, rd->qual);
  } else {
    const bp_record_batch* r = j->recs;
    uint8_t sb = (uint8_t)orc_get_strand(r->ref_id[rd->rec], r->mate_ref_id[rd->rec], r->flags[rd->rec]);
    for (uint32_t k = 0; k < n; k++)
      append_base(j, ab[k].pos_in_ref & mask, ab[k].pos_in_read, is_minus, rd->seq8, rd->qual,
                  r->l_seq[rd->rec], sb);
  }
}

/* addReadPair (pileup.go:366-465) */
static int add_read_pair(job* j, read_snp* reads, int n_read, int32_t is_minus) {
  const bp_record_batch* r = j->recs;
  j->nab[1] = 0;
  for (int i = 0; i < n_read; i++) {
    if (align_relevant_bases(j, i, &reads[i])) return -1;
    clip_quals(reads[i].qual, (int32_t)r->l_seq[reads[i].rec], j->o->clip);
  }
  aligned_pos *a0 = j->ab[0], *a1 = j->ab[1];
  uint32_t n0 = j->nab[0], n1 = j->nab[1];
  if (n_read == 1 || n1 == 0) {
    if (n0) {
      add_unstitched(j, &reads[0], is_minus, a0, n0);
      int32_t em = a0[n0 - 1].pos_in_ref + 1;
      if (j->end_max < em) j->end_max = em;
    }
    return 0;
  }
  if (n0 == 0) {
    add_unstitched(j, &reads[1], is_minus, a1, n1);
    int32_t em = a1[n1 - 1].pos_in_ref + 1;
    if (j->end_max < em) j->end_max = em;
    return 0;
  }
  int32_t mask = j->n_circ - 1;
  const uint8_t *s0 = reads[0].seq8, *s1 = reads[1].seq8;
  const uint8_t *q0 = reads[0].qual, *q1 = reads[1].qual;
  uint32_t i0 = 0, i1 = 0;
  while (i0 != n0 && i1 != n1) {
    int32_t p0 = a0[i0].pos_in_ref, p1 = a1[i1].pos_in_ref;
    if (p0 == p1) {
      payload* row = &j->ring[p0 & mask];
      row->depth++;
      int32_t r0 = a0[i0].pos_in_read, r1 = a1[i1].pos_in_read;
      uint8_t c0 = s0[r0], c1 = s1[r1];
      if (c0 == c1) {
        if (j->per_read) /* a Go panic (pileup.go:422-425) */
          FAIL(j->err, "stitched per-read features not yet supported");
        uint8_t base = kSeq8ToEnum[c0];
        uint8_t qa = q0[r0], qb = q1[r1];
        if (qa >= NQUAL || qb >= NQUAL)
          FAIL(j->err, "qualPassTable.lookup2: index out of range [%u,%u] (record %lld)", qa, qb,
               (long long)reads[0].rec); /* Go: runtime panic, index out of range */
        if (j->qpt[qa][qb] || base == 4) row->counts[base][is_minus]++;
      } else {
        row->counts[4][is_minus]++;
      }
      i0++;
      i1++;
    } else if (p0 < p1) {
      if (j->per_read) FAIL(j->err, "stitched per-read features not yet supported"); /* pileup.go:435 */
      add_base(j, p0 & mask, a0[i0].pos_in_read, is_minus, s0, q0);
      i0++;
    } else {
      if (j->per_read) FAIL(j->err, "stitched per-read features not yet supported"); /* pileup.go:442 */
      add_base(j, p1 & mask, a1[i1].pos_in_read, is_minus, s1, q1);
      i1++;
    }
  }
  int32_t em;
  if (i0 != n0) {
    add_unstitched(j, &reads[0], is_minus, a0 + i0, n0 - i0);
    em = a0[n0 - 1].pos_in_ref + 1;
  } else if (i1 != n1) {
    add_unstitched(j, &reads[1], is_minus, a1 + i1, n1 - i1);
    em = a1[n1 - 1].pos_in_ref + 1;
  } else {
    em = a0[n0 - 1].pos_in_ref + 1;
  }
  if (j->end_max < em) j->end_max = em;
  return 0;
}

static uint32_t hash_name(const bp_record_batch* r, int64_t i) {
  size_t len;
  const uint8_t* nm = rec_name(r, i, &len);
  return orc_murmur3_32(nm, len) % HTABLE;
}

/* firstreadSNPTable.add (firstread.go:105-115) */
static void fr_add(job* j, int64_t rec, int32_t map_end) {
  firstreads* f = &j->first;
  int32_t pos = j->recs->pos[rec];
  int32_t circ = pos & (f->nonempty.n_circ - 1);
  uint32_t h = hash_name(j->recs, rec);
  bucket* b = &f->buckets[(size_t)circ * HTABLE + h];
  if (b->n == 0) cb_set(&f->nonempty, pos, circ, h);
  if (b->n == b->cap) {
    b->cap = b->cap ? b->cap * 2 : 4;
    b->v = (aug_rec*)realloc(b->v, sizeof(aug_rec) * b->cap);
  }
  b->v[b->n].rec = rec;
  b->v[b->n].map_end = map_end;
  b->n++;
}

/* tryRemove + remove (firstread.go:118-152) */
static int64_t fr_try_remove(job* j, int64_t rec, int32_t* mate_map_end) {
  firstreads* f = &j->first;
  const bp_record_batch* r = j->recs;
  int32_t expected_mate_pos = r->pos[rec];
  int32_t pos = r->mate_pos[rec];
  int32_t circ = pos & (f->nonempty.n_circ - 1);
  size_t nlen;
  const uint8_t* nm = rec_name(r, rec, &nlen);
  uint32_t h = orc_murmur3_32(nm, nlen) % HTABLE;
  bucket* b = &f->buckets[(size_t)circ * HTABLE + h];
  for (uint32_t i = 0; i < b->n; i++) {
    int64_t cand = b->v[i].rec;
    size_t clen;
    const uint8_t* cn = rec_name(r, cand, &clen);
    if (r->mate_pos[cand] == expected_mate_pos && clen == nlen && memcmp(cn, nm, nlen) == 0) {
      *mate_map_end = b->v[i].map_end;
      uint32_t last = b->n - 1;
      if (i != last) b->v[i] = b->v[last];
      b->n = last;
      if (last == 0) cb_clear(&f->nonempty, pos, circ, h);
      return cand;
    }
  }
  return -1;
}

/* addOrRemove (firstread.go:176-204) */
static int fr_add_or_remove(job* j, int64_t rec, int strand) {
  const bp_record_batch* r = j->recs;
  int32_t mate_pos = r->mate_pos[rec], pos = r->pos[rec];
  int32_t map_end = j->pair[0].map_end;
  if (strand == 0 || mate_pos >= map_end ||
      (int64_t)mate_pos + j->o->max_read_span <= (int64_t)pos) {
    convert_samr(j, &j->pair[0], rec);
    return 1;
  }
  if (pos >= mate_pos) {
    int32_t mme = 0;
    int64_t mate = fr_try_remove(j, rec, &mme);
    if (mate >= 0) {
      convert_samr(j, &j->pair[0], rec);
      j->pair[1].map_end = mme;
      convert_samr(j, &j->pair[1], mate);
      return 2;
    }
    if (pos != mate_pos) {
      convert_samr(j, &j->pair[0], rec);
      return 1;
    }
  }
  fr_add(j, rec, map_end);
  return 0;
}

/* addOrphanReads (pileup.go:469-500) */
static int add_orphan_reads(job* j, int32_t stop_pos) {
  if (!j->o->stitch) return 0;
  firstreads* f = &j->first;
  const bp_record_batch* r = j->recs;
  int32_t is_minus = 0;
  for (int32_t pos = f->nonempty.first_pos; pos < stop_pos; pos = f->nonempty.first_pos) {
    int32_t circ = cb_take_first_row(&f->nonempty);
    uint64_t* row = f->nonempty.bits + (size_t)circ * HWORDS;
    for (uint32_t h = 0; h < HTABLE; h++) {
      if (!(row[h / 64] >> (h % 64) & 1)) continue;
      bucket* b = &f->buckets[(size_t)circ * HTABLE + h];
      for (uint32_t k = 0; k < b->n; k++) {
        int64_t rec = b->v[k].rec;
        convert_samr(j, &j->orphan, rec);
        j->orphan.map_end = b->v[k].map_end;
        if (!j->o->ignore_strand)
          is_minus = orc_get_strand(r->ref_id[rec], r->mate_ref_id[rec], r->flags[rec]) - 1;
        if (add_read_pair(j, &j->orphan, 1, is_minus)) return -1;
      }
      b->n = 0;
    }
    memset(row, 0, sizeof(uint64_t) * HWORDS);
  }
  return 0;
}

/* flushToInternal (pileup.go:513-572) */
static int flush_to_internal(job* j, int32_t write_end) {
  int32_t mask = j->n_circ - 1, start, end;
  while (us_scan(&j->wscan, &start, &end, write_end)) {
    for (int32_t pos = start; pos != end; pos++) {
      payload* row = &j->ring[pos & mask];
      if (row->depth == 0) {
        if (sink_append(j, j->ref_id, pos, NULL)) return -1;
      } else {
        if (sink_append(j, j->ref_id, pos, row)) return -1;
        row->depth = 0;
        memset(row->counts, 0, sizeof(row->counts));
        memset(row->npr, 0, sizeof(row->npr));
      }
    }
  }
  return 0;
}

/* writeEmptyEntries (pileup_support.go:89-102) */
static int write_empty_entries(job* j, int32_t flush_end) {
  int32_t start, end;
  while (us_scan(&j->wscan, &start, &end, flush_end))
    for (int32_t pos = start; pos != end; pos++)
      if (sink_append(j, j->ref_id, pos, NULL)) return -1;
  return 0;
}

/* flushTo (pileup.go:576-585) */
static int flush_to(job* j, int32_t flush_end) {
  if (j->end_max > j->wscan.pos) {
    int32_t write_end = flush_end < j->end_max ? flush_end : j->end_max;
    if (flush_to_internal(j, write_end)) return -1;
    if (j->end_max >= flush_end) return 0;
  }
  return write_empty_entries(j, flush_end);
}

/* flushEmptyContigs (pileup_support.go:112-123) */
static int flush_empty_contigs(job* j, int ref_idx, int ref_idx_end) {
  for (; ref_idx < ref_idx_end; ref_idx++) {
    j->end_max = 0;
    j->wscan = us_new(j->bed_part.ep[ref_idx], j->bed_part.n[ref_idx]);
    j->ref_id = ref_idx;
    if (flush_to(j, POS_MAX)) return -1;
  }
  return 0;
}

/* finishRef (pileup.go:624-639) */
static int finish_ref(job* j, int ref_idx_end) {
  if (j->ref_id != -1) {
    if (add_orphan_reads(j, POS_MAX)) return -1;
    if (flush_to(j, POS_MAX)) return -1;
  }
  return flush_empty_contigs(j, j->ref_id + 1, ref_idx_end);
}

/* nextRef (pileup.go:641-651) */
static int next_ref(job* j, int new_ref) {
  if (finish_ref(j, new_ref)) return -1;
  j->end_max = 0;
  j->wscan = us_new(j->bed_part.ep[new_ref], j->bed_part.n[new_ref]);
  j->ref_id = new_ref;
  return 0;
}

typedef struct orc_shard {
  int32_t start_ref, end_ref; /* reference IDs (StartRef/EndRef) */
  int32_t start, end;
  int32_t padding;
} orc_shard;

static int32_t shard_padded_start(const orc_shard* s) {
  int32_t v = s->start - s->padding;
  return v > 0 ? v : 0;
}
static int32_t shard_padded_end(const orc_shard* s, const int32_t* ref_len) {
  if (s->end == 0) return 0;
  int64_t v = (int64_t)s->end + s->padding;
  int32_t rl = ref_len[s->end_ref];
  return v < rl ? (int32_t)v : rl;
}

/* Exposed for the TestShard KAT (encoding/bam/shard_test.go:12-31): PadStart/PadEnd. */
int32_t orc_shard_pad_start(int32_t start, int32_t padding) {
  orc_shard s = {0, 0, start, 0, padding};
  return shard_padded_start(&s);
}
int32_t orc_shard_pad_end(int32_t end, int32_t padding, int32_t ref_len) {
  orc_shard s = {0, 0, 0, end, padding};
  return shard_padded_end(&s, &ref_len);
}

/* GetPositionBasedShards (encoding/bam/shard.go:320-352), includeUnmapped=false.
 * Returns the number of shards (writes at most cap). */
int32_t orc_position_shards(int32_t n_ref, const int32_t* ref_len, int32_t shard_size,
                            int32_t padding, orc_shard* out, int32_t cap) {
  int32_t n = 0;
  for (int32_t r = 0; r < n_ref; r++) {
    int64_t start = 0;
    while (start < ref_len[r]) {
      int64_t end = start + shard_size;
      if (end > ref_len[r]) end = ref_len[r];
      if (n < cap) {
        out[n].start_ref = r;
        out[n].end_ref = r;
        out[n].start = (int32_t)start;
        out[n].end = (int32_t)end;
        out[n].padding = padding;
      }
      n++;
      start += shard_size;
    }
  }
  return n;
}

static int coord_lt(int32_t r1, int32_t p1, int32_t r2, int32_t p2) {
  /* biopb.Coord order: unmapped (ref -1) sorts after every mapped reference */
  uint32_t a = (uint32_t)r1, b = (uint32_t)r2; /* -1 -> 0xffffffff */
  if (a != b) return a < b;
  return p1 < p2;
}

/* processShard (pileup.go:661-810) */
static int process_shard(job* j, const orc_shard* sh, const int32_t* ref_len) {
  const bp_record_batch* r = j->recs;
  const bp_opts* o = j->o;
  int32_t s_ref = sh->start_ref, s_pos = shard_padded_start(sh);
  int32_t l_ref = sh->end_ref, l_pos = shard_padded_end(sh, ref_len);
  /* iterator: first record with coord >= start (bamprovider.go:429-433) */
  int64_t lo = 0, hi = r->n;
  while (lo < hi) {
    int64_t mid = (lo + hi) >> 1;
    if (coord_lt(r->ref_id[mid], r->pos[mid], s_ref, s_pos))
      lo = mid + 1;
    else
      hi = mid;
  }
  int32_t is_minus = 0;
  for (int64_t i = lo; i < r->n; i++) {
    if (!coord_lt(r->ref_id[i], r->pos[i], l_ref, l_pos)) break;
    if (j->shard_overlap) {
      if (r->ref_id[i] == j->prev_limit_id && (int64_t)r->pos[i] < j->prev_limit_pos) continue;
      j->shard_overlap = 0;
    }
    if ((o->flag_exclude & (int32_t)r->flags[i]) != 0 || o->mapq > (int32_t)r->mapq[i] ||
        r->n_cigar[i] == 0)
      continue;
    int strand = orc_get_strand(r->ref_id[i], r->mate_ref_id[i], r->flags[i]);
    if (o->strand_req != BP_STRAND_NONE) {
      if (strand != o->strand_req) continue;
    } else if (strand == 0 && !o->ignore_strand) {
      continue;
    }
    if (j->ref_id != r->ref_id[i])
      if (next_ref(j, r->ref_id[i])) return -1;
    int32_t flush_end = r->pos[i];
    if (o->stitch) {
      int32_t first_orphan = j->first.nonempty.first_pos;
      if (first_orphan != POS_MAX) {
        if ((int64_t)first_orphan + o->max_read_span < (int64_t)flush_end)
          flush_end -= o->max_read_span;
        else
          flush_end = first_orphan;
      }
    }
    if (j->wscan.pos < flush_end) {
      if (add_orphan_reads(j, flush_end)) return -1;
      if (flush_to(j, flush_end)) return -1;
    }
    if ((int64_t)r->l_seq[i] > (int64_t)o->max_read_len) {
      size_t nl;
      const uint8_t* nm = rec_name(r, i, &nl);
      FAIL(j->err,
           "pileupMutable.processShard: maxReadLen is %d, but read %.*s at ref %d:%d has length %u",
           o->max_read_len, (int)nl, (const char*)nm, r->ref_id[i], r->pos[i], r->l_seq[i]);
    }
    int32_t span = cigar_span(r, i);
    if (span > o->max_read_span) {
      size_t nl;
      const uint8_t* nm = rec_name(r, i, &nl);
      FAIL(j->err,
           "pileupMutable.processShard: maxReadSpan is %d, but read %.*s at ref %d:%d has span %d",
           o->max_read_span, (int)nl, (const char*)nm, r->ref_id[i], r->pos[i], span);
    }
    int32_t map_end = r->pos[i] + span;
    if (!bed_intersects_by_id(&j->bed_part, j->ref_id, r->pos[i], map_end)) continue;
    j->pair[0].map_end = map_end;
    int n_read = 1;
    if (o->stitch) {
      n_read = fr_add_or_remove(j, i, strand);
      if (n_read == 0) continue;
    } else {
      convert_samr(j, &j->pair[0], i);
    }
    if (!o->ignore_strand) is_minus = strand - 1;
    if (add_read_pair(j, j->pair, n_read, is_minus)) return -1;
  }
  return 0;
}

static void read_snp_alloc(read_snp* r, int32_t max_len) {
  r->seq8 = (uint8_t*)malloc((size_t)max_len + 8);
  r->qual = (uint8_t*)malloc((size_t)max_len + 8);
}
static void read_snp_free(read_snp* r) {
  free(r->seq8);
  free(r->qual);
}

/*
 * One pileupSNPMain pass (pileup.go:812-943) over in-memory records.
 *   bed_off[n_ref+1], bed_ep: the full BEDUnion; reference i owns endpoints
 *     bed_ep[bed_off[i] .. bed_off[i+1]); an empty range = reference not in the BED (nil).
 *   shards/n_shards: opts.shards; parallelism: number of jobs (clamped to n_shards).
 * Rows are appended in job order, like the converters read tmpFiles in order
 * (output.go:136).  Returns 0, or -1 with err filled.
 */
static int pileup_impl(const bp_opts* o, int32_t n_ref, const int32_t* ref_len, const uint64_t* bed_off,
                       const int32_t* bed_ep, const bp_record_batch* recs, int32_t n_shards,
                       const orc_shard* shards, int32_t parallelism, int64_t cap_rows, int32_t* out_ref,
                       int32_t* out_pos, uint32_t* out_vals, int64_t* n_rows, uint64_t* out_off,
                       orc_feat* out_feat, int64_t feat_cap, int64_t* n_feat, char* err, int32_t errlen) {
  errbuf e = {err, errlen, 0};
  if (errlen > 0) err[0] = 0;
  *n_rows = 0;
  /* Pileup() validation (pileup.go:975-988) */
  if (o->clip < 0 || o->clip * 2 >= o->max_read_len) FAIL(&e, "Pileup: invalid clip= argument");
  if (o->max_read_len > o->max_read_span)
    FAIL(&e, "Pileup: max-read-len= argument cannot be larger than max-read-span= argument");
  if (o->min_base_qual >= NQUAL) FAIL(&e, "newQualPassTable: minBaseQual too large");
  if (n_shards <= 0) FAIL(&e, "oracle: no shards");
  qual_sum_init();

  bedunion full;
  bed_alloc(&full, n_ref);
  for (int32_t i = 0; i < n_ref; i++) {
    uint64_t n = bed_off[i + 1] - bed_off[i];
    if (n) bed_set_copy(&full, i, bed_ep + bed_off[i], (uint32_t)n);
  }
  rowsink sink = {cap_rows, 0, out_ref, out_pos, out_vals, out_off, out_feat, feat_cap, 0};
  if (parallelism <= 0 || parallelism > n_shards) parallelism = n_shards;

  int32_t n_circ = orc_next_exp2(o->max_read_span);
  if (o->stitch) n_circ *= 2;
  int rc = 0;
  for (int32_t job_idx = 0; job_idx < parallelism && rc == 0; job_idx++) {
    int32_t s0 = (int32_t)(((int64_t)job_idx * n_shards) / parallelism);
    int32_t s1 = (int32_t)((((int64_t)job_idx + 1) * n_shards) / parallelism);
    job j;
    memset(&j, 0, sizeof(j));
    j.o = o;
    j.recs = recs;
    j.sink = &sink;
    j.err = &e;
    j.per_read = out_off != NULL;
    j.ref_id = -1;
    j.prev_limit_id = -1;
    j.n_circ = n_circ;
    j.ring = (payload*)calloc((size_t)n_circ, sizeof(payload));
    /* the reference sizes these by maxReadLen and lets append grow them; a read passing the
     * span check emits at most maxReadSpan pairs */
    j.ab[0] = (aligned_pos*)malloc(sizeof(aligned_pos) * ((size_t)o->max_read_span + 1));
    j.ab[1] = (aligned_pos*)malloc(sizeof(aligned_pos) * ((size_t)o->max_read_span + 1));
    read_snp_alloc(&j.pair[0], o->max_read_len);
    read_snp_alloc(&j.pair[1], o->max_read_len);
    read_snp_alloc(&j.orphan, o->max_read_len);
    if (o->stitch) fr_init(&j.first, n_circ);
    for (int a = 0; a < NQUAL; a++)
      for (int b = 0; b < NQUAL; b++) j.qpt[a][b] = g_qual_sum[a][b] >= o->min_base_qual;
    {
      /* exact BED split for this job (pileup.go:896-909) */
      const orc_shard* f = &shards[s0];
      const orc_shard* l = &shards[s1 - 1];
      int lim_ref = l->end_ref;
      int32_t lim_pos = l->end;
      if (lim_ref < 0) {
        lim_ref = n_ref - 1;
        lim_pos = ref_len[lim_ref];
      }
      bed_subset(&full, f->start_ref, f->start, lim_ref, lim_pos, &j.bed_part);
    }
    for (int32_t s = s0; s < s1 && rc == 0; s++) {
      const orc_shard* sh = &shards[s];
      int lim_ref = sh->end_ref;
      int32_t lim_pos = sh->end;
      if (lim_ref < 0) {
        lim_ref = n_ref - 1;
        lim_pos = ref_len[lim_ref];
      }
      /* intersectionIsEmpty (pileup_support.go:71-85) */
      if (!bed_intersects(&j.bed_part, sh->start_ref, sh->start, lim_ref, lim_pos)) continue;
      rc = process_shard(&j, sh, ref_len);
      j.shard_overlap = 1;
      j.prev_limit_id = sh->end_ref;
      j.prev_limit_pos = (int64_t)sh->end + sh->padding;
    }
    if (rc == 0) rc = finish_ref(&j, n_ref);
    for (int32_t c = 0; c < n_circ; c++)
      for (int b = 0; b < 4; b++) free(j.ring[c].pr[b]);
    free(j.ring);
    free(j.ab[0]);
    free(j.ab[1]);
    read_snp_free(&j.pair[0]);
    read_snp_free(&j.pair[1]);
    read_snp_free(&j.orphan);
    if (o->stitch) fr_free(&j.first);
    bed_free(&j.bed_part);
  }
  bed_free(&full);
  *n_rows = sink.n;
  if (n_feat) *n_feat = sink.n_feat;
  return rc;
}

int orc_pileup(const bp_opts* o, int32_t n_ref, const int32_t* ref_len, const uint64_t* bed_off,
               const int32_t* bed_ep, const bp_record_batch* recs, int32_t n_shards,
               const orc_shard* shards, int32_t parallelism, int64_t cap_rows, int32_t* out_ref,
               int32_t* out_pos, uint32_t* out_vals, int64_t* n_rows, char* err, int32_t errlen) {
  return pileup_impl(o, n_ref, ref_len, bed_off, bed_ep, recs, n_shards, shards, parallelism, cap_rows,
                     out_ref, out_pos, out_vals, n_rows, NULL, NULL, 0, NULL, err, errlen);
}

/* Same run with pCtx.perReadNeeded set (pileup.go:888): additionally returns the ordered
 * per-read feature lists of every row (appendBase, pileup.go:235-250).  out_off has
 * 4*cap_rows+1 entries; out_feat has feat_cap 6-byte entries {u16 dist5p, u16 fraglen, u8 qual,
 * u8 strand}. */
int orc_pileup_per_read(const bp_opts* o, int32_t n_ref, const int32_t* ref_len, const uint64_t* bed_off,
                        const int32_t* bed_ep, const bp_record_batch* recs, int32_t n_shards,
                        const orc_shard* shards, int32_t parallelism, int64_t cap_rows, int32_t* out_ref,
                        int32_t* out_pos, uint32_t* out_vals, int64_t* n_rows, uint64_t* out_off,
                        void* out_feat, int64_t feat_cap, int64_t* n_feat, char* err, int32_t errlen) {
  if (!out_off || !out_feat) return -1;
  out_off[0] = 0;
  return pileup_impl(o, n_ref, ref_len, bed_off, bed_ep, recs, n_shards, shards, parallelism, cap_rows,
                     out_ref, out_pos, out_vals, n_rows, out_off, (orc_feat*)out_feat, feat_cap, n_feat, err,
                     errlen);
}

/* Exposed for BED KATs (interval/bedunion_internal_test.go): Subset of a one-reference
 * union; returns the number of endpoints written. */
int32_t orc_bed_subset1(const int32_t* ep, int32_t n, int32_t start, int32_t limit, int32_t* out,
                        int32_t cap) {
  bedunion u, s;
  bed_alloc(&u, 1);
  bed_set_copy(&u, 0, ep, (uint32_t)n);
  bed_subset(&u, 0, start, 0, limit, &s);
  int32_t m = (int32_t)s.n[0];
  for (int32_t i = 0; i < m && i < cap; i++) out[i] = s.ep[0][i];
  bed_free(&u);
  bed_free(&s);
  return m;
}
int orc_bed_intersects1(const int32_t* ep, int32_t n, int32_t start, int32_t limit) {
  bedunion u;
  bed_alloc(&u, 1);
  bed_set_copy(&u, 0, ep, (uint32_t)n);
  int r = bed_intersects_by_id(&u, 0, start, limit);
  bed_free(&u);
  return r;
}
