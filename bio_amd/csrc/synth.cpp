// synth.cpp -- seeded synthetic coordinate-sorted read generator (libbpsynth.so).
//
// Bench/test support, not part of the pileup path: every BAM/FASTA fixture in the reference
// is a git-LFS pointer, so inputs of BASELINE.json's shapes are produced
// here.  Output is a host bp_record_batch (include/bio_pileup.h), i.e. what a
// bamprovider.Iterator would have decoded.  Deterministic for a given parameter block,
// independent of the number of worker threads.
//
// Model: FR pairs, insert ~ N(mean, sd) clipped to [min, max], pair
// starts stratified-uniform over each region, CIGAR mix of plain M / soft clips / one D /
// one I / hard clip, substitutions against a hash-defined reference, quals from a 4-level
// distribution, a share of low MAPQ, duplicate / secondary / supplementary flags and
// mates on another contig.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/bio_pileup.h"

extern "C" {

typedef struct bps_region {
  int32_t ref_id;
  int32_t start, end;  // pairs start in [start - insert_max, end)
  double coverage;     // mean depth over [start, end)
} bps_region;

typedef struct bps_params {
  uint64_t seed;
  int32_t read_len;
  double insert_mean, insert_sd;
  int32_t insert_min, insert_max;
  double frac_softclip, frac_del, frac_ins, frac_hardclip;  // CIGAR mix; rest is plain M
  double sub_rate;                                          // per-base substitution rate
  double frac_low_mapq, frac_dup, frac_secondary, frac_mate_other_ref;
  int32_t with_names;
  int32_t threads;
  int32_t n_regions;
  const bps_region* regions;
} bps_params;

typedef struct bps_batch bps_batch;

}  // extern "C"

namespace {

inline uint64_t mix64(uint64_t z) {  // splitmix64 finalizer
  z += 0x9e3779b97f4a7c15ULL;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
  return z ^ (z >> 31);
}

struct Rng {
  uint64_t s;
  explicit Rng(uint64_t seed) : s(seed) {}
  uint64_t next() {
    s += 0x9e3779b97f4a7c15ULL;
    uint64_t z = s;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
  }
  double uni() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
  uint32_t below(uint32_t n) { return (uint32_t)(((next() >> 32) * (uint64_t)n) >> 32); }
};

// Reference base (BAM nibble code) at (ref, pos): A/C/G/T uniform, 0.1% N.
inline uint8_t ref_nibble(uint64_t seed, int32_t ref, int64_t pos) {
  uint64_t h = mix64(seed ^ 0x5bd1e995u ^ ((uint64_t)(uint32_t)ref << 40) ^ (uint64_t)pos);
  if ((h >> 20) % 1000 == 0) return 15;
  return (uint8_t)(1u << (h & 3));
}

struct PairSpec {  // everything about a pair that is decided up front
  int32_t ref_id;
  int32_t pos[2];
  uint32_t l_seq[2];
  uint32_t n_cigar[2];
  uint32_t cigar[2][3];
  uint16_t flags[2];
  uint8_t mapq[2];
  int32_t mate_ref[2];
  int32_t mate_pos[2];
};

struct Key {
  int32_t ref;
  int32_t pos;
  uint64_t pair;  // pair id << 1 | which
  bool operator<(const Key& o) const {
    if (ref != o.ref) return ref < o.ref;
    if (pos != o.pos) return pos < o.pos;
    return pair < o.pair;
  }
};

struct RegionPlan {
  bps_region r;
  int64_t first_pair;  // global pair id of the region's first pair
  int64_t n_pairs;
  int64_t span_start;  // pairs start in [span_start, r.end)
};

inline int64_t pair_start(const RegionPlan& rp, int64_t k, uint64_t seed) {
  // stratified uniform: k-th of n_pairs slots over [span_start, end)
  double u = (double)(mix64(seed ^ (uint64_t)(rp.first_pair + k) * 0xd1342543de82ef95ULL) >> 11) *
             (1.0 / 9007199254740992.0);
  double L = (double)(rp.r.end - rp.span_start);
  int64_t s = rp.span_start + (int64_t)(((double)k + u) * L / (double)rp.n_pairs);
  if (s < 0) s = 0;
  return s;
}

void make_cigar(const bps_params& p, Rng& g, int32_t L, uint32_t* cig, uint32_t* n_cig,
                uint32_t* l_seq, int32_t* span) {
  double u = g.uni();
  double a = p.frac_softclip, b = a + p.frac_del, c = b + p.frac_ins, d = c + p.frac_hardclip;
  if (L < 40) u = 1.0;  // too short for the structured variants
  if (u < a) {
    int which = (int)g.below(3);  // leading, trailing, both
    int32_t s1 = 1 + (int32_t)g.below(30 < L / 4 ? 30 : (uint32_t)(L / 4));
    int32_t s2 = 1 + (int32_t)g.below(30 < L / 4 ? 30 : (uint32_t)(L / 4));
    uint32_t n = 0;
    int32_t m = L;
    if (which == 0 || which == 2) { cig[n++] = ((uint32_t)s1 << 4) | 4; m -= s1; }
    uint32_t mi = n++;
    if (which == 1 || which == 2) { cig[n++] = ((uint32_t)s2 << 4) | 4; m -= s2; }
    cig[mi] = ((uint32_t)m << 4) | 0;
    *n_cig = n; *l_seq = (uint32_t)L; *span = m;
  } else if (u < b) {
    int32_t dl = 1 + (int32_t)g.below(10);
    int32_t m1 = 10 + (int32_t)g.below((uint32_t)(L - 20));
    cig[0] = ((uint32_t)m1 << 4) | 0;
    cig[1] = ((uint32_t)dl << 4) | 2;
    cig[2] = ((uint32_t)(L - m1) << 4) | 0;
    *n_cig = 3; *l_seq = (uint32_t)L; *span = L + dl;
  } else if (u < c) {
    int32_t il = 1 + (int32_t)g.below(10);
    int32_t m1 = 10 + (int32_t)g.below((uint32_t)(L - 20 - il));
    cig[0] = ((uint32_t)m1 << 4) | 0;
    cig[1] = ((uint32_t)il << 4) | 1;
    cig[2] = ((uint32_t)(L - m1 - il) << 4) | 0;
    *n_cig = 3; *l_seq = (uint32_t)L; *span = L - il;
  } else if (u < d) {
    int32_t h = 1 + (int32_t)g.below(20);
    cig[0] = ((uint32_t)h << 4) | 5;
    cig[1] = ((uint32_t)(L - h) << 4) | 0;
    *n_cig = 2; *l_seq = (uint32_t)(L - h); *span = L - h;
  } else {
    cig[0] = ((uint32_t)L << 4) | 0;
    *n_cig = 1; *l_seq = (uint32_t)L; *span = L;
  }
}

void make_pair(const bps_params& p, const RegionPlan& rp, int64_t k, PairSpec* ps) {
  uint64_t pid = (uint64_t)(rp.first_pair + k);
  Rng g(mix64(p.seed ^ (pid * 0x9e3779b97f4a7c15ULL + 0x1234567)));
  int64_t s = pair_start(rp, k, p.seed);
  // Box-Muller insert size
  double u1 = g.uni(), u2 = g.uni();
  if (u1 < 1e-300) u1 = 1e-300;
  double z = std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
  int32_t ins = (int32_t)std::lround(p.insert_mean + p.insert_sd * z);
  if (ins < p.insert_min) ins = p.insert_min;
  if (ins > p.insert_max) ins = p.insert_max;
  int32_t L = p.read_len;
  ps->ref_id = rp.r.ref_id;
  int32_t span[2];
  for (int w = 0; w < 2; w++)
    make_cigar(p, g, L, ps->cigar[w], &ps->n_cigar[w], &ps->l_seq[w], &span[w]);
  ps->pos[0] = (int32_t)s;
  int64_t p1 = s + ins - span[1];  // second read ends at the fragment end
  if (p1 < s) p1 = s;
  ps->pos[1] = (int32_t)p1;
  bool first_is_r1 = (g.next() & 1) != 0;
  ps->flags[0] = (uint16_t)(0x1 | 0x2 | 0x20 | (first_is_r1 ? 0x40 : 0x80));
  ps->flags[1] = (uint16_t)(0x1 | 0x2 | 0x10 | (first_is_r1 ? 0x80 : 0x40));
  for (int w = 0; w < 2; w++) {
    ps->mapq[w] = g.uni() < p.frac_low_mapq ? (uint8_t)g.below(60) : 60;
    double u = g.uni();
    if (u < p.frac_dup) ps->flags[w] |= 0x400;
    else if (u < p.frac_dup + p.frac_secondary) ps->flags[w] |= (g.next() & 1) ? 0x100 : 0x800;
    ps->mate_ref[w] = ps->ref_id;
    ps->mate_pos[w] = ps->pos[1 - w];
  }
  if (g.uni() < p.frac_mate_other_ref) {
    // chimeric: each read claims a mate on another contig (GetStrand -> StrandNone)
    for (int w = 0; w < 2; w++) {
      ps->mate_ref[w] = ps->ref_id + 1;
      ps->mate_pos[w] = (int32_t)g.below(1000000);
    }
  }
}

}  // namespace

struct bps_batch {
  int64_t n = 0;
  int64_t aligned_bases = 0;  // sum of M lengths over all reads (no filters)
  std::vector<int32_t> ref_id, pos, mate_ref_id, mate_pos;
  std::vector<uint16_t> flags;
  std::vector<uint8_t> mapq;
  std::vector<uint32_t> n_cigar, l_seq;
  std::vector<uint64_t> cigar_off, seq_off, qual_off, name_off;
  std::vector<uint32_t> cigar;
  std::vector<uint8_t> seq, qual, name;
};

namespace {

template <class F>
void parallel_for(int64_t n_items, int threads, F f) {
  if (threads < 1) threads = 1;
  std::atomic<int64_t> next(0);
  std::vector<std::thread> ts;
  for (int t = 0; t < threads; t++)
    ts.emplace_back([&] {
      for (;;) {
        int64_t i = next.fetch_add(1);
        if (i >= n_items) break;
        f(i);
      }
    });
  for (auto& t : ts) t.join();
}

}  // namespace

extern "C" {

int bps_generate(const bps_params* pp, bps_batch** out) {
  const bps_params& p = *pp;
  if (p.n_regions <= 0 || p.read_len <= 0 || p.insert_max < p.insert_min) return -1;
  std::vector<RegionPlan> plans((size_t)p.n_regions);
  int64_t total_pairs = 0;
  for (int i = 0; i < p.n_regions; i++) {
    RegionPlan& rp = plans[(size_t)i];
    rp.r = p.regions[i];
    rp.first_pair = total_pairs;
    rp.span_start = (int64_t)rp.r.start - p.insert_max;
    if (rp.span_start < 0) rp.span_start = 0;
    double bases = rp.r.coverage * (double)(rp.r.end - rp.span_start);
    rp.n_pairs = (int64_t)(bases / (2.0 * p.read_len));
    if (rp.n_pairs < 1) rp.n_pairs = 1;
    total_pairs += rp.n_pairs;
  }
  // Pass 1: keys, in chunks of pairs (chunk-local sort, then a k-way-free global order
  // because chunk c only overlaps chunk c+1 near its end: handled by a final merge pass).
  const int64_t CH = 1 << 16;
  struct Chunk { int region; int64_t k0, k1; };
  std::vector<Chunk> chunks;
  for (int i = 0; i < p.n_regions; i++)
    for (int64_t k = 0; k < plans[(size_t)i].n_pairs; k += CH)
      chunks.push_back({i, k, std::min(k + CH, plans[(size_t)i].n_pairs)});
  std::vector<Key> keys((size_t)total_pairs * 2);
  parallel_for((int64_t)chunks.size(), p.threads, [&](int64_t ci) {
    const Chunk& c = chunks[(size_t)ci];
    const RegionPlan& rp = plans[(size_t)c.region];
    PairSpec ps;
    for (int64_t k = c.k0; k < c.k1; k++) {
      make_pair(p, rp, k, &ps);
      uint64_t pid = (uint64_t)(rp.first_pair + k);
      keys[(size_t)pid * 2] = {ps.ref_id, ps.pos[0], pid << 1};
      keys[(size_t)pid * 2 + 1] = {ps.ref_id, ps.pos[1], (pid << 1) | 1};
    }
  });
  // Global order.  Keys are already nearly sorted (displacement bounded by insert_max), so
  // sort fixed-size blocks in parallel and then merge adjacent blocks pairwise.
  {
    const int64_t n = (int64_t)keys.size();
    int64_t B = 1 << 20;
    int64_t nb = (n + B - 1) / B;
    parallel_for(nb, p.threads, [&](int64_t b) {
      std::sort(keys.begin() + b * B, keys.begin() + std::min(n, (b + 1) * B));
    });
    for (int64_t w = B; w < n; w *= 2) {
      int64_t nm = (n + 2 * w - 1) / (2 * w);
      parallel_for(nm, p.threads, [&](int64_t m) {
        int64_t lo = m * 2 * w, mid = std::min(n, lo + w), hi = std::min(n, lo + 2 * w);
        if (mid < hi && keys[(size_t)mid] < keys[(size_t)mid - 1])
          std::inplace_merge(keys.begin() + lo, keys.begin() + mid, keys.begin() + hi);
      });
    }
  }
  bps_batch* b = new bps_batch();
  const int64_t n = (int64_t)keys.size();
  b->n = n;
  b->ref_id.resize((size_t)n); b->pos.resize((size_t)n);
  b->mate_ref_id.resize((size_t)n); b->mate_pos.resize((size_t)n);
  b->flags.resize((size_t)n); b->mapq.resize((size_t)n);
  b->n_cigar.resize((size_t)n); b->l_seq.resize((size_t)n);
  b->cigar_off.resize((size_t)n); b->seq_off.resize((size_t)n); b->qual_off.resize((size_t)n);
  if (p.with_names) b->name_off.resize((size_t)n + 1);
  // region lookup for a pair id
  auto region_of = [&](uint64_t pid) -> const RegionPlan& {
    size_t lo = 0, hi = plans.size();
    while (hi - lo > 1) {
      size_t mid = (lo + hi) / 2;
      if ((uint64_t)plans[mid].first_pair <= pid) lo = mid; else hi = mid;
    }
    return plans[lo];
  };
  // Pass 2: fixed fields + sizes
  const int64_t RB = 1 << 15;
  const int64_t nrb = (n + RB - 1) / RB;
  std::vector<int64_t> blk_cig((size_t)nrb + 1, 0), blk_seq((size_t)nrb + 1, 0),
      blk_qual((size_t)nrb + 1, 0), blk_name((size_t)nrb + 1, 0), blk_m((size_t)nrb + 1, 0);
  parallel_for(nrb, p.threads, [&](int64_t rb) {
    PairSpec ps;
    int64_t c = 0, s = 0, q = 0, nm = 0, mb = 0;
    for (int64_t i = rb * RB; i < std::min(n, (rb + 1) * RB); i++) {
      uint64_t pid = keys[(size_t)i].pair >> 1;
      int w = (int)(keys[(size_t)i].pair & 1);
      const RegionPlan& rp = region_of(pid);
      make_pair(p, rp, (int64_t)pid - rp.first_pair, &ps);
      b->ref_id[(size_t)i] = ps.ref_id;
      b->pos[(size_t)i] = ps.pos[w];
      b->mate_ref_id[(size_t)i] = ps.mate_ref[w];
      b->mate_pos[(size_t)i] = ps.mate_pos[w];
      b->flags[(size_t)i] = ps.flags[w];
      b->mapq[(size_t)i] = ps.mapq[w];
      b->n_cigar[(size_t)i] = ps.n_cigar[w];
      b->l_seq[(size_t)i] = ps.l_seq[w];
      b->cigar_off[(size_t)i] = (uint64_t)c;  // block-relative for now
      b->seq_off[(size_t)i] = (uint64_t)s;
      b->qual_off[(size_t)i] = (uint64_t)q;
      c += ps.n_cigar[w];
      s += (ps.l_seq[w] + 1) / 2;
      q += ps.l_seq[w];
      for (uint32_t k = 0; k < ps.n_cigar[w]; k++)
        if ((ps.cigar[w][k] & 15) == 0) mb += ps.cigar[w][k] >> 4;
      if (p.with_names) {
        char tmp[32];
        int len = snprintf(tmp, sizeof(tmp), "p%llu", (unsigned long long)pid);
        b->name_off[(size_t)i] = (uint64_t)nm;
        nm += len;
      }
    }
    blk_cig[(size_t)rb + 1] = c; blk_seq[(size_t)rb + 1] = s; blk_qual[(size_t)rb + 1] = q;
    blk_name[(size_t)rb + 1] = nm; blk_m[(size_t)rb + 1] = mb;
  });
  for (int64_t rb = 0; rb < nrb; rb++) {
    blk_cig[(size_t)rb + 1] += blk_cig[(size_t)rb];
    blk_seq[(size_t)rb + 1] += blk_seq[(size_t)rb];
    blk_qual[(size_t)rb + 1] += blk_qual[(size_t)rb];
    blk_name[(size_t)rb + 1] += blk_name[(size_t)rb];
    blk_m[(size_t)rb + 1] += blk_m[(size_t)rb];
  }
  b->aligned_bases = blk_m[(size_t)nrb];
  b->cigar.resize((size_t)blk_cig[(size_t)nrb]);
  b->seq.resize((size_t)blk_seq[(size_t)nrb]);
  b->qual.resize((size_t)blk_qual[(size_t)nrb]);
  if (p.with_names) {
    b->name.resize((size_t)blk_name[(size_t)nrb]);
    b->name_off[(size_t)n] = (uint64_t)blk_name[(size_t)nrb];
  }
  // Pass 3: payloads
  parallel_for(nrb, p.threads, [&](int64_t rb) {
    PairSpec ps;
    for (int64_t i = rb * RB; i < std::min(n, (rb + 1) * RB); i++) {
      uint64_t pid = keys[(size_t)i].pair >> 1;
      int w = (int)(keys[(size_t)i].pair & 1);
      const RegionPlan& rp = region_of(pid);
      make_pair(p, rp, (int64_t)pid - rp.first_pair, &ps);
      b->cigar_off[(size_t)i] += (uint64_t)blk_cig[(size_t)rb];
      b->seq_off[(size_t)i] += (uint64_t)blk_seq[(size_t)rb];
      b->qual_off[(size_t)i] += (uint64_t)blk_qual[(size_t)rb];
      uint32_t* cg = b->cigar.data() + b->cigar_off[(size_t)i];
      for (uint32_t k = 0; k < ps.n_cigar[w]; k++) cg[k] = ps.cigar[w][k];
      uint8_t* sq = b->seq.data() + b->seq_off[(size_t)i];
      uint8_t* ql = b->qual.data() + b->qual_off[(size_t)i];
      Rng g(mix64(p.seed ^ (keys[(size_t)i].pair * 0xc2b2ae3d27d4eb4fULL + 77)));
      const uint32_t sub_thr = (uint32_t)(p.sub_rate * 65536.0);
      int64_t rpos = ps.pos[w];
      uint32_t bi = 0;  // base index in read
      uint64_t bits = 0;
      int nbits = 0;
      auto emit = [&](uint8_t nib) {
        if (nbits == 0) { bits = g.next(); nbits = 4; }
        uint32_t r16 = (uint32_t)(bits & 0xffff);
        bits >>= 16; nbits--;
        uint32_t qsel = r16 & 127;
        uint8_t qv = qsel < 3 ? 2 : qsel < 9 ? 12 : qsel < 26 ? 23 : 37;
        (void)r16;
        if (bi & 1) sq[bi >> 1] |= nib; else sq[bi >> 1] = (uint8_t)(nib << 4);
        ql[bi] = qv;
        bi++;
      };
      for (uint32_t k = 0; k < ps.n_cigar[w]; k++) {
        uint32_t op = ps.cigar[w][k] & 15, len = ps.cigar[w][k] >> 4;
        if (op == 0) {
          for (uint32_t t = 0; t < len; t++, rpos++) {
            uint8_t nib = ref_nibble(p.seed, ps.ref_id, rpos);
            uint64_t h = mix64((keys[(size_t)i].pair << 20) ^ (uint64_t)bi ^ (p.seed * 31));
            if ((uint32_t)(h & 0xffff) < sub_thr) nib = (uint8_t)(1u << ((h >> 16) & 3));
            emit(nib);
          }
        } else if (op == 1 || op == 4) {
          for (uint32_t t = 0; t < len; t++) {
            uint64_t h = mix64((keys[(size_t)i].pair << 20) ^ (uint64_t)bi ^ (p.seed * 17));
            emit((uint8_t)(1u << (h & 3)));
          }
        } else if (op == 2 || op == 3) {
          rpos += len;
        }
      }
      if (p.with_names) {
        b->name_off[(size_t)i] += (uint64_t)blk_name[(size_t)rb];
        char tmp[32];
        int len = snprintf(tmp, sizeof(tmp), "p%llu", (unsigned long long)pid);
        memcpy(b->name.data() + b->name_off[(size_t)i], tmp, (size_t)len);
      }
    }
  });
  *out = b;
  return 0;
}

void bps_view(const bps_batch* b, bp_record_batch* v) {
  memset(v, 0, sizeof(*v));
  v->n = b->n;
  v->ref_id = b->ref_id.data(); v->pos = b->pos.data();
  v->mate_ref_id = b->mate_ref_id.data(); v->mate_pos = b->mate_pos.data();
  v->flags = b->flags.data(); v->mapq = b->mapq.data();
  v->n_cigar = b->n_cigar.data(); v->l_seq = b->l_seq.data();
  v->cigar_off = b->cigar_off.data(); v->cigar = b->cigar.data();
  v->seq_off = b->seq_off.data(); v->seq = b->seq.data();
  v->qual_off = b->qual_off.data(); v->qual = b->qual.data();
  if (!b->name_off.empty()) { v->name_off = b->name_off.data(); v->name = b->name.data(); }
}

int64_t bps_aligned_bases(const bps_batch* b) { return b->aligned_bases; }
int64_t bps_payload_sizes(const bps_batch* b, int64_t* n_cigar, int64_t* n_seq, int64_t* n_qual,
                          int64_t* n_name) {
  *n_cigar = (int64_t)b->cigar.size(); *n_seq = (int64_t)b->seq.size();
  *n_qual = (int64_t)b->qual.size(); *n_name = (int64_t)b->name.size();
  return b->n;
}
uint8_t bps_ref_nibble(uint64_t seed, int32_t ref, int64_t pos) { return ref_nibble(seed, ref, pos); }
void bps_free(bps_batch* b) { delete b; }

}  // extern "C"
