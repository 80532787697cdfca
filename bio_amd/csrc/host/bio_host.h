// bio_host.h -- host-side layers around the device pileup: the C++ counterparts of the
// reference packages that sit on either side of the hot path (SURVEY.md section 8 rows a2,
// a16-a20).  Everything here is plain host code; counts are only ever produced by the HIP
// path behind include/bio_pileup.h.
#pragma once
#include <cstdint>
#include <cstdio>
#include <functional>
#include <memory>
#include <string>
#include <vector>

#include "../../../include/bio_pileup.h"

namespace bio {

using PosType = int32_t;                      // interval.PosType (endpoint_index.go:52)
constexpr PosType kPosTypeMax = 2147483647;   // interval.PosTypeMax

// ---------------------------------------------------------------- interval (bedunion.go)
struct Entry {  // interval.Entry (bedunion.go:581-585)
  std::string ref_name;
  PosType start0 = 0, end = 0;
};

// interval.ParseRegionString (bedunion.go:593-651).  Returns "" or the error text.
std::string ParseRegionString(const std::string& region, Entry* out);

// interval.BEDUnion restricted to what pileup uses: per-reference sorted endpoint arrays,
// looked up by reference ID (bedunion.go:72-99).  `present[i]` distinguishes a reference the
// BED never mentions (nil in Go) from one it mentions.
struct BEDUnion {
  std::vector<std::string> ref_names;
  std::vector<std::vector<PosType>> id_map;
  std::vector<uint8_t> present;

  // NewBEDUnionFromPath (bedunion.go:563-578) + scanBEDUnion (:411-540); gzip transparent.
  static std::string FromPath(const std::string& path, const std::vector<std::string>& ref_names,
                              BEDUnion* out);
  // NewBEDUnionFromEntries (bedunion.go:655-735)
  static std::string FromEntries(const std::vector<Entry>& entries,
                                 const std::vector<std::string>& ref_names, BEDUnion* out);
  bool IntersectsByID(int ref, PosType start, PosType limit) const;  // bedunion.go:219-254
  // Subset (bedunion.go:753-852) for a single-reference or multi-reference range.
  BEDUnion Subset(int start_ref, PosType start_pos, int limit_ref, PosType limit_pos) const;
  int64_t NumPositions() const;
};

// ---------------------------------------------------------------- encoding/bam (reader)
struct BamHeader {
  std::string text;
  std::vector<std::string> ref_names;
  std::vector<int32_t> ref_lens;
};

// Growable structure-of-arrays record batch; view() exposes it as a bp_record_batch.
// std::allocator that leaves trivially constructible elements uninitialised on resize(): the
// payload vectors below are sized first and filled by several threads afterwards, and zeroing
// 13 GB (a chr1-size BAM) first was a second of a single core.
template <class T>
struct NoInitAlloc : std::allocator<T> {
  template <class U> struct rebind { using other = NoInitAlloc<U>; };
  NoInitAlloc() = default;
  template <class U> NoInitAlloc(const NoInitAlloc<U>&) {}
  template <class U> void construct(U* p) noexcept { ::new ((void*)p) U; }
  template <class U, class... A> void construct(U* p, A&&... a) { ::new ((void*)p) U(std::forward<A>(a)...); }
};

// Sequential BGZF/BAM reader: header + alignment records in file order, decoded per
// encoding/bam/unmarshal.go:126-224 / SAM spec 4.2.
class BamReader {
 public:
  BamReader();
  ~BamReader();
  std::string Open(const std::string& path);
  const BamHeader& header() const { return header_; }
  // Decodes the next record into raw fields; returns 1 on a record, 0 at EOF, -1 on error
  // (message in err()).
  struct Raw {
    int32_t ref_id, pos, mate_ref_id, mate_pos;
    uint16_t flags, n_cigar;
    uint8_t mapq;
    uint32_t l_seq;
    const char* name; uint32_t l_name;  // without the NUL
    const uint32_t* cigar;
    const uint8_t* seq; const uint8_t* qual;
    const uint8_t* aux; uint32_t l_aux;
  };
  int Next(Raw* r);
  // The fields of the record whose block_size field is at `rec` (all of the record's bytes
  // must be there); false when they do not fit the block size.
  static bool Parse(const uint8_t* rec, Raw* r);
  // All whole records that are buffered, as they lie in the inflated stream: record k starts
  // (block_size field) at data + off[k], and all of them end at or before data + bytes.  The
  // memory stays valid for as long as `hold` is kept, whatever the reader does next.  Same
  // return values as Next(); fields are only surveyed (Parse checks one record's).
  struct Chunk {
    std::shared_ptr<const void> hold;
    const uint8_t* data = nullptr;
    size_t bytes = 0;
    std::vector<uint32_t> off;
    // what the walk saw in the fixed fields of these records:
    bool fields_ok = true;  // every record's fields fit its block
    bool sorted = true;     // (ref, pos) never descends inside the chunk (ref -1 sorts last)
    bool one_ref = true;    // every record lies on first_ref
    int32_t first_ref = -1, first_pos = 0, last_ref = -1, last_pos = 0;
    int rewalked = 0;       // stretches whose guessed start was wrong (walked again from the right place)
    // Resident mode (StartResident): the stream stays on the device.  `dev` is its device address
    // (record k at dev + off[k]), `cols` the device scan's columns (entry k = record k), and `data`
    // is NULL until FetchHost has brought the bytes over (a batch whose device scan failed comes
    // with host bytes and without columns).
    const uint8_t* dev = nullptr;
    const bp_bam_columns* cols = nullptr;
  };
  int NextChunk(Chunk* c);
  // Inflate into pinned host memory (hipHostMalloc), which bp_submit_bam copies from without
  // staging.  Before Open().
  void SetPinned(bool on);
  // Inflate BGZF blocks on HIP device `device` (bp_inflate_blocks: one wavefront per block, CRC-32
  // checked there) instead of on the host threads; the inflated bytes come back into the same
  // (pinned) buffers.  Batches are larger then (4,096 blocks instead of 256: the device wants
  // thousands of blocks in flight).  -1: host.  Any time: batches already read ahead stay as they are.
  void SetDeviceInflate(int device);
  // From here on the inflated stream stays on `device`: blocks are inflated there, records found
  // and their fixed fields read there (bp_bam_scan), and NextChunk hands out device chunks (see
  // Chunk::dev).  After Open(); only NextChunk / FetchHost may follow (no Next, no Seek).
  std::string StartResident(int device, bool want_names = false);
  // The bytes of a device chunk copied to host memory (c->data set; c->hold keeps them too).
  std::string FetchHost(Chunk* c);
  // Batches whose device scan failed and were walked on the host instead.
  int64_t ResidentFallbacks() const;
  // Seconds the device spent in the inflate + CRC kernels, and bytes they produced.
  double DeviceInflateSeconds() const;
  // Wall seconds of the read-ahead stages: file reads + header parse, inflate (either kind), and of that
  // the time spent getting output buffers.
  void StageSeconds(double* read_raw, double* inflate, double* take, double* scan = nullptr) const;
  // True when the next Next() has to bring in more of the file first: the buffers move then,
  // and the pointers of Raw records handed out so far stop being valid.
  bool NextRefills() const;
  // Seconds Next() spent waiting for the inflater (BIO_PILEUP_STATS).
  double WaitSeconds() const;
  const std::string& err() const { return err_; }
  // Continues reading at a BGZF virtual offset (compressed block offset << 16 | offset inside
  // the inflated block), e.g. one taken from an index.
  std::string Seek(uint64_t voffset);
  // File offset of the BGZF block the next record starts in.
  uint64_t BlockOffset() const;

 struct Impl;

 private:
  std::unique_ptr<Impl> impl_;
  BamHeader header_;
  std::string err_;
};

// .bai (SAM spec section 5.2; only the linear index is used) or .gbai
// (encoding/bam/gindex.go:35-73) index of a coordinate-sorted BAM.
class BamIndex {
 public:
  std::string Load(const std::string& path);
  bool loaded() const { return loaded_; }
  // A virtual offset from which reading forward meets every record at or after (ref, pos)
  // (conservative, like bamIterator.reset, bamprovider.go:286-348); false when the index shows
  // that no mapped record lies at or after it.
  bool Offset(int32_t ref, int32_t pos, uint64_t* voffset) const;

 private:
  struct GEntry { int32_t ref, pos; uint32_t seq; uint64_t voff; };
  bool loaded_ = false, gbai_ = false;
  std::vector<std::vector<uint64_t>> lin_;  // .bai: per reference, ioffset of each 16 kb window
  std::vector<GEntry> g_;                   // .gbai entries, ascending (mapped first)
};

// Host worker threads for inflate, rendering and output compression: -parallelism when
// given (the reference's default is every CPU, pileup.go:56-68), else the CPUs of this
// process capped at 16.
int HostThreads();
void SetHostThreads(int n);

// Writes `b` as a BAM file (tools and tests; the driver only reads BAMs).
std::string WriteBam(const std::string& path, const std::vector<std::string>& ref_names,
                     const std::vector<int32_t>& ref_lens, const bp_record_batch& b, int level, int threads);

// Integer value of a 2-character aux tag (A/c/C/s/S/i/I types); returns false if absent.
bool AuxInt(const uint8_t* aux, uint32_t l_aux, const char tag[2], int64_t* out);

// ---------------------------------------------------------------- encoding/fasta + pileup.LoadFa
// fasta.New eager loader with Seq8 encoding (fasta.go:120-175, biosimd_amd64.go:471-511):
// A/a 1, C/c 2, G/g 4, T/t 8, everything else 15.
struct Fasta {
  std::vector<std::string> names;
  std::vector<std::string> seqs;  // Seq8 bytes
  int Find(const std::string& name) const;
};
std::string LoadFaSeq8(const std::string& path, Fasta* out);
// pileup.FaToStringSlice (pileup/common.go:194-225)
std::string FaToStringSlice(const Fasta& fa, const BamHeader& h, std::vector<const std::string*>* out,
                            std::string* warnings);

// ---------------------------------------------------------------- pileup/snp
struct Opts {  // snp.Opts / snp.DefaultOpts (pileup/snp/pileup.go:36-68)
  std::string BedPath, Region, BamIndexPath;
  int Clip = 0;
  std::string Cols;
  int FlagExclude = 0xf00;
  int Mapq = 60;
  int MaxReadLen = 500;
  int MaxReadSpan = 511;
  int MinBagDepth = 0;
  int MinBaseQual = 0;
  int Parallelism = 0;
  bool PerStrand = false;
  bool RemoveSq = false;
  bool Stitch = false;
  std::string TempDir;
  int Device = 0;  // HIP device ordinal (no counterpart in the reference)
  // Jobs are spread over this many logical devices, mapped round-robin onto the physical ones
  // (-1: one per physical device; 0: everything on Device).  The reference's -parallelism with
  // devices for goroutines (pileup.go:862-932).
  int Devices = 0;
};

// Column-set bits (pileup.go:134-157) and pileup.ParseCols (pileup/common.go:118-166).
enum {
  kColDpRef = 1, kColDpAlt = 2, kColEndDists = 4, kColQuals = 8, kColFraglens = 16,
  kColStrands = 32, kColHighQ = 64, kColLowQ = 128
};
std::string ParseCols(const std::string& cols, int default_bitset, int* out);

// snp.Pileup (pileup.go:970-1130).  Returns "" or the error text (the CLI turns it into a
// panic-style exit like main.go:94-96).
std::string Pileup(const std::string& xampath, const std::string& fapath, const std::string& format,
                   const std::string& out_prefix, const Opts& opts);

// cmd/bio-pileup main (main.go:60-98): flag parsing + Pileup.  Returns the process exit code.
int BioPileupMain(int argc, char** argv);

// ---------------------------------------------------------------- output (output.go)
// Row sink used by Pileup: rows arrive in ascending (ref,pos) order.
// Text that the device has rendered comes to the host in strings whose large buffers are pinned
// memory (hipHostMalloc): the copy from the device then runs at the bus rate instead of through
// the runtime's staging buffers, which was half of a closer thread's time.  Small strings use malloc.
void* TextAlloc(size_t n);
void TextFree(void* p, size_t n);
template <class T>
struct TextAllocator {
  using value_type = T;
  TextAllocator() = default;
  template <class U> TextAllocator(const TextAllocator<U>&) {}
  T* allocate(size_t n) { return static_cast<T*>(TextAlloc(n * sizeof(T))); }
  void deallocate(T* p, size_t n) { TextFree(p, n * sizeof(T)); }
  template <class U> bool operator==(const TextAllocator<U>&) const { return true; }
  template <class U> bool operator!=(const TextAllocator<U>&) const { return false; }
};
using TextString = std::basic_string<char, std::char_traits<char>, TextAllocator<char>>;

class RowWriter {
 public:
  virtual ~RowWriter() {}
  // off/feats: per-read feature lists of the n rows (bp_per_read_offsets / _features; feats
  // points at entry off[0]); both NULL unless a per-read column set was requested.
  virtual std::string Write(const int32_t* ref_id, const int32_t* pos, const uint32_t* vals, int64_t n,
                            const uint64_t* off, const bp_per_read* feats) = 0;
  // Lines already rendered (on the device, bp_render_text), appended as they are to output file
  // `which` of the format (basestrand-tsv: 0; tsv: 0 = .ref.tsv, 1 = .alt.tsv).
  // SpareText: a string to render into, one the sink has finished with (capacity kept, no fresh
  // pages) if there is one.
  virtual TextString SpareText(int which) { (void)which; return TextString(); }
  virtual std::string WriteText(int which, TextString&& text) {
    (void)which; (void)text;
    return "pileup output: this format takes rows, not text";
  }
  // The same lines as complete BGZF members (compressed on the device, bp_render_fetch_bgzf): only
  // for the -bgz formats, whose files are runs of such members.
  virtual std::string WriteBgzf(int which, TextString&& members) {
    (void)which; (void)members;
    return "pileup output: this format takes rows, not BGZF members";
  }
  virtual std::string Close() = 0;
};
// ConvertPileupRowsToTSV (output.go:42-300): <prefix>.ref.tsv[.gz] + <prefix>.alt.tsv[.gz]
std::unique_ptr<RowWriter> NewTsvWriter(const std::string& main_path, int col_bitset, bool bgzip,
                                        const std::vector<std::string>& ref_names,
                                        const std::vector<const std::string*>& ref_seqs, std::string* err);
// ConvertPileupRowsToBasestrandTSV (output.go:376-575): <prefix>.basestrand.tsv[.gz]
std::unique_ptr<RowWriter> NewBasestrandTsvWriter(const std::string& main_path, int col_bitset, bool bgzip,
                                                  const std::vector<std::string>& ref_names,
                                                  const std::vector<const std::string*>& ref_seqs,
                                                  std::string* err);

}  // namespace bio
