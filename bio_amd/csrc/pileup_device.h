// pileup_device.h -- data layout shared by the host code and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bp {

// One alignment record's fixed part, 32 bytes (the size of the BAM fixed block decoded at
// encoding/bam/unmarshal.go:126-224).  Fields the kernels never read (mate position, bin,
// tlen, the reference id -- records of one reference are a contiguous index range) make
// room for the first three CIGAR ops, so that a record with <= 3 ops (virtually all) needs
// no dependent load before its payload.  `span` is the reference span sam.Cigar.Lengths()
// returns (pileup.go:769), clamped to 0xffff.
struct ReadCore {
  int32_t pos;
  uint16_t flags;
  uint8_t mapq;
  uint8_t aux;       // AUX_* bits
  uint16_t n_cigar;
  uint16_t span;
  uint16_t l_seq;    // clamped to 0xffff (longer records never pass max-read-len)
  uint16_t qdelta;   // the quals start this many 8-byte units after `blob`
  uint32_t blob;     // offset of the record's CIGAR + bases in 8-byte units
  uint32_t cig[3];   // first three CIGAR ops (the payload holds all of them)
};
static_assert(sizeof(ReadCore) == 32, "ReadCore must be 32 bytes");

enum : uint8_t {
  AUX_SIMPLE = 1,     // CIGAR is a single M op
  AUX_MATE_SAME = 2,  // Ref == MateRef (first test of pileup.GetStrand, common.go:95)
  // what the record-level tests of processShard come to under the job's options, worked out
  // once by the packer (the per-read gather reads these; the pileup kernel does the tests):
  AUX_KEEP = 4,       // passes flag-exclude, mapq, empty CIGAR and the strand rule (pileup.go:691, :717-729)
  AUX_STRAND_SHIFT = 4,  // bits 4-5: pileup.GetStrand (0 none, 1 fwd, 2 rev)
  // stitch: set by stitch_resolve_kernel on the read of a pair that was stored first - the pair is
  // counted on the strand of the read that found its mate (pileup.go:799-802), bit 7 = that is "-"
  AUX_PAIR_FIRST = 64,
  AUX_PAIR_REV = 128
};

// Payload of a record, two 8-byte aligned parts (U = ceil(l_seq/8)):
//   base part   uint32 cigar[n_cigar rounded up to even]
//               uint8  seq4[4*U]   packed 4-bit bases (BAM codes), first base of a byte in the
//                                  LOW nibble (BAM has it in the high one); pad nibbles 0
//               uint8  qpass[U]    in every job but min_base_qual 0 without stitching: bit k of
//                                  byte j = base 8j+k passes it, i.e. its quality, after clipQuals
//                                  (pileup_support.go:51-69), is >= min_base_qual (addBase,
//                                  pileup.go:227).  Both are job options, so the packer can say
//                                  per base what the counting kernel would work out from the qual:
//                                  that kernel then loads 4 bytes per unit of 32 bases instead of
//                                  36, and the clip variant is the same kernel.  Padded to a dword.
//                                  Stitched jobs (qbits = 2): every dword of pass bits is followed by
//                                  a dword of DROP bits for the same 32 bases, zero from the packer.
//                                  stitch_resolve_kernel (once per job, before the first compute)
//                                  decides the positions both reads of a pair cover (addReadPair,
//                                  pileup.go:406-446) and leaves the outcome in the payloads: the
//                                  read stored first gets its drop bits set there (not counted), the
//                                  read that found its mate gets base X where the reads disagree and
//                                  its pass bit = the qualSumTable rule where they agree.  Pass and
//                                  drop both set: a quality >= 96 at an agreeing position, where the
//                                  reference panics; the counting kernel reports it when it gets there.
//               pad to 8 bytes
//   qual part   uint8  qual[8*U]   pad bytes 0
// Records are packed in groups of up to kGroupRecs: the base parts of a group back to back,
// then its qual parts.  A run that never looks at quals (min_base_qual 0 without stitching:
// every base passes, pileup.go:227) then never pulls them through HBM: they are whole
// kilobytes apart from anything it loads, not interleaved with it.
constexpr int kGroupRecs = 128;
constexpr uint64_t kGroupBytes = 400 << 10;  // keeps qdelta (16 bits of 8-byte units) in range
inline uint32_t blob_units(uint32_t l_seq) { return (l_seq + 7) >> 3; }
// qpass: the base part ends with one bit per base, "this base passes the job's quality rule"
// (bit k of byte j = base 8j + k; U bytes padded to a dword), see KernelParams.qpass.
// qbits: dwords of per-base bits per 32 bases (0 none, 1 pass bits, 2 pass + drop bits)
inline uint64_t base_part_bytes(uint32_t n_cigar, uint32_t l_seq, int qbits = 0) {
  const uint64_t U = blob_units(l_seq);
  uint64_t b = (uint64_t)((n_cigar + 1) & ~1u) * 4 + 4 * U + (uint64_t)qbits * 4 * ((U + 3) >> 2);
  return (b + 7) & ~(uint64_t)7;
}
inline uint64_t qual_part_bytes(uint32_t l_seq) { return 8 * (uint64_t)blob_units(l_seq); }
inline uint64_t blob_bytes(uint32_t n_cigar, uint32_t l_seq, int qbits = 0) {
  return base_part_bytes(n_cigar, l_seq, qbits) + qual_part_bytes(l_seq);
}

// A genomic window: at most W consecutive positions of one reference, covering BED
// positions only at its ends; one workgroup accumulates it in LDS and flushes it.
struct Window {
  int32_t ref_id;
  int32_t g0, g1;        // [g0, g1), g1 - g0 <= W
  uint32_t iv_lo, iv_hi; // BED intervals (global index) intersecting the window
  int32_t full;          // 1: [g0,g1) lies inside interval iv_lo
  int32_t n_rows;        // BED positions in the window
  int32_t pad;
  int64_t row_base;      // output row of the first BED position in the window
};
static_assert(sizeof(Window) == 40, "Window must be 40 bytes");

// Mate table entry for stitch mode, one per record, all zero = processed alone.  It carries what
// stitch_resolve_kernel needs of the record's mate (firstread.go:176-204 decides the pairs, on the
// host), so that no second record has to be fetched there:
//   x     0..16 mate start - own start (17 bits two's complement; a pair lies within one span)
//         17    this record found its mate in the table (readPair[0], firstread.go:187)
//         18    strand slot the pair is counted on is "-" (the finder's strand, pileup.go:799-802)
//         19    the mate's CIGAR is one M op          20    has a mate
//   blob  the mate's payload offset (ReadCore::blob)
//   span_lseq    mate span | mate l_seq << 16
//   ncig_qdelta  mate n_cigar | mate qdelta << 16
struct MateEntry {
  uint32_t x, blob, span_lseq, ncig_qdelta;
};
static_assert(sizeof(MateEntry) == 16, "MateEntry must be 16 bytes");
enum : uint32_t { MATE_SECOND = 1u << 17, MATE_PAIR_REV = 1u << 18, MATE_SIMPLE = 1u << 19, MATE_HAS = 1u << 20 };

constexpr int kCoarse = 1024;  // stride of the coarse position index the plan kernel searches first

struct DeviceError {
  int32_t read;  // smallest record index with a data error, or 0x7fffffff
  int32_t kind;
};
enum { ERR_NONE = 0, ERR_QUAL_RANGE = 1, ERR_ITEM_OVERFLOW = 2 };

struct KernelParams {
  const ReadCore* cores;
  const uint8_t* blob;
  const MateEntry* mate;      // NULL unless stitch (read by stitch_resolve_kernel only)
  const Window* windows;
  const int32_t* win_r0;      // per window: first record that may overlap it
  const int32_t* win_r1;      // per window: one past the last record starting before g1
  const int64_t* ref_first;   // [n_ref+1] index of the first record of each reference
  const int32_t* coarse_pos;  // pos of every kCoarse-th record (index k <-> record k*kCoarse)
  const int32_t* iv_start;
  const int32_t* iv_end;
  const int64_t* iv_row;
  uint32_t* rows;             // n_rows * 11
  DeviceError* err;
  const uint8_t* qthr;        // [96] stitch: qthr[q0] = smallest q1 with qualSum[q0][q1] >= minBaseQual (255: none)
  // Deep windows (more than slice_recs records) are cut into even slices of at most
  // slice_recs records: the plan kernel appends them here as (window, first record, end
  // record, 0), the first launch writes the window's rows as zeros, and a second launch adds
  // the slices to them.
  uint4* extra_items;
  uint32_t* n_extra;          // number of appended items (device counter, reset per compute)
  int32_t extra_cap;
  int32_t slice_recs;
  unsigned long long* dbg;    // diagnostic builds only (BP_STAMPS); NULL otherwise
  int64_t n_reads;
  int32_t n_windows;
  int32_t max_span;           // largest (clamped) span among submitted records
  // options (bp_opts)
  int32_t clip, flag_exclude, mapq, min_base_qual, stitch, ignore_strand, strand_req;
  int32_t qpass;              // dwords of per-base bits per 32 bases in the payloads: 0, 1 pass bits, 2 pass + drop bits (see the payload layout)
  int32_t wins_per_block;     // consecutive windows a workgroup of pileup_kernel takes (set by launch_pileup)
};

// One entry of a per-read feature list = PerReadFeatures (pileup/snp/row.go:22-31), 6 bytes.
struct PerRead {
  uint16_t dist5p, fraglen;
  uint8_t qual, strand;
};
static_assert(sizeof(PerRead) == 6, "PerRead must be 6 bytes");

// The per-read gather: a wave owns a tile of kPrT consecutive positions of one BED interval
// and keeps in LDS a list of up to kPrPieces (record, M op) pieces that overlap it and the
// cursors of the tile's lists.
constexpr int kPrT = 32;          // positions per tile
constexpr int kPrBlock = 256;     // threads per workgroup, one tile per wave
constexpr int kPrPieces = 256;    // piece list per wave (16 bytes each); a record has at most kPrT
constexpr int kPrWaveBytes = kPrPieces * 16 + kPrT * 16 + kPrT * 4;
struct PerReadParams {
  const int32_t* iv_ref;          // [n_iv] reference of each flattened BED interval
  const int64_t* iv_tile0;        // [n_iv + 1] first tile of each interval
  // list offsets = exclusive scan of the counts: list i starts at tile_base[i >> 10] + off32[i]
  const uint32_t* off32;          // [n_lists = 4 * n_rows]
  const unsigned long long* tile_base;  // [ceil(n_lists / 1024)]
  unsigned long long total;       // entries in all lists
  int64_t n_lists;
  const int32_t* tile_lo;         // [n_tiles] first record that can reach the tile (per_read_plan_kernel)
  PerRead* out;
  int32_t* bad_row;               // smallest row whose lists disagree with its counts
  int64_t n_tiles;
  int32_t n_iv;
};

// Output conversion on the device: rows -> basestrand-tsv text (ConvertPileupRowsToBasestrandTSV,
// pileup/snp/output.go:376-453, the column set without per-read lists).
struct TextParams {
  const uint32_t* rows;
  const int64_t* iv_row;     // [n_iv + 1]
  const int32_t* iv_start;
  const int32_t* iv_ref;
  const uint8_t* seq8;       // Seq8 bases of reference seq_ref, positions [seq_start, seq_start + seq_len)
  const char* name;          // that reference's name
  uint32_t* len;             // [n] text length of each row, then its offset inside its tile of 1024 rows
  unsigned long long* tile;  // [ceil(n / 1024) + 1] tile totals, then tile offsets and the total
  char* out;
  int32_t* bad;              // set when a row has no reference base
  int64_t first, n;
  int64_t seq_len;
  int32_t n_iv, seq_ref, seq_start, name_len;
  int32_t kind, cols;        // BP_TEXT_* and the -cols bit set (bio_pileup.h)
  int32_t strand;            // 0: the rows as they are; 1 / 2: the forward / reverse strand's share of split-strand rows
};

}  // namespace bp
