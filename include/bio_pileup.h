/*
 * bio_pileup.h -- C ABI of libbiopileup.so, the MI355X (gfx950) implementation of the
 * grailbio/bio pileup/snp hot path.
 *
 * The reference has no FFI for this path: the seam is the Go method
 *   pileupMutable.processShard + flushTo*      (pileup/snp/pileup.go:661-810, :513-585)
 * driven per job by pileupSNPMain              (pileup/snp/pileup.go:862-939).
 * One bp_handle replaces one job's {pileupMutable, refContext, pileupContext,
 * pileupShardContext} (pileup.go:179-187, :161-172, :212-220, :653-659).  A cgo shim calls
 * these entry points instead of running the ring-buffer loop; see INTEGRATION.md.
 *
 * All signatures are plain C: pointers, sizes, fixed-width ints.  Every call returns 0 on
 * success and a negative BP_E* code on failure; bp_last_error() returns the message (for
 * data errors the wording follows the reference's fmt.Errorf strings).
 *
 * Threading: handles are independent and may be driven from different OS threads
 * concurrently (cgo pins a goroutine to a thread for the duration of a call).  One handle
 * must not be used from two threads at once, like the reference's goroutine-private
 * pileupMutable (pileup.go:867-871).
 */
#ifndef BIO_PILEUP_H_
#define BIO_PILEUP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* One output row = what pm.w.Append(&PileupRow{...}) receives (pileup.go:524-557,
 * row.go:48-66): Depth, then Counts[base][strand] for base in A,C,G,T,X and strand in +,-. */
#define BP_ROW_U32 11

enum {
  BP_OK = 0,
  BP_EINVAL = -1,     /* bad argument / call order */
  BP_EDEVICE = -2,    /* HIP runtime failure (message has the HIP error string) */
  BP_ENOMEM = -3,
  BP_EDATA = -4,      /* a record violates max-read-len / max-read-span / CIGAR / qual range
                         (pileup.go:766-772, :355-359, qual.go:26) */
  BP_EUNSORTED = -5   /* records not in (ref_id,pos) order */
};

/* strand_req values = pileup.StrandType (pileup/common.go:77-88) */
enum { BP_STRAND_NONE = 0, BP_STRAND_FWD = 1, BP_STRAND_REV = 2 };

/* Options that reach the hot loop: pileupSNPOpts (pileup.go:597-622) fields read by
 * processShard plus pileupContext (pileup.go:212-220) plus pileupShardContext.strandReq. */
typedef struct bp_opts {
  int32_t clip;          /* -clip            pileup_support.go:47-69 */
  int32_t flag_exclude;  /* -flag-exclude    pileup.go:691 */
  int32_t mapq;          /* -mapq            pileup.go:691 */
  int32_t max_read_len;  /* -max-read-len    pileup.go:766 */
  int32_t max_read_span; /* -max-read-span   pileup.go:769 */
  int32_t min_base_qual; /* -min-base-qual   pileup.go:228, qual.go:62 */
  int32_t stitch;        /* -stitch          pileup.go:788 */
  int32_t ignore_strand; /* format is tsv/tsv-bgz (pileup.go:887): all counts in slot 0,
                            StrandNone reads kept */
  int32_t strand_req;    /* -per-strand pass (pileup.go:718-723); BP_STRAND_NONE otherwise */
  int32_t device;        /* HIP device ordinal */
  int32_t reserved[6];   /* must be 0 */
} bp_opts;

/* A run of decoded alignment records in coordinate order, as the reference's
 * bamprovider.Iterator yields them (encoding/bamprovider/provider.go:141-166).  Field
 * meaning = BAM fixed block as decoded by gbam.Unmarshal (encoding/bam/unmarshal.go:126-224).
 * Structure-of-arrays; every pointer is caller-owned host memory and is copied before the
 * call returns (the Go side recycles records with sam.PutInFreePool, pileup.go:805-807).
 * Record i owns cigar[cigar_off[i] .. cigar_off[i]+n_cigar[i]), ceil(l_seq[i]/2) packed
 * 4-bit bases at seq+seq_off[i] (BAM layout: first base in the high nibble), l_seq[i]
 * quality bytes at qual+qual_off[i], and name bytes [name_off[i], name_off[i+1]).
 * name/name_off may be NULL unless opts.stitch is set (names key the mate table,
 * firstread.go:135-152). */
typedef struct bp_record_batch {
  int64_t n;
  const int32_t* ref_id;
  const int32_t* pos;
  const int32_t* mate_ref_id;
  const int32_t* mate_pos;
  const uint16_t* flags;
  const uint8_t* mapq;
  const uint32_t* n_cigar;
  const uint32_t* l_seq;
  const uint64_t* cigar_off; /* in uint32 ops */
  const uint32_t* cigar;     /* len<<4 | op, op codes MIDNSHP=X = 0..8 */
  const uint64_t* seq_off;   /* bytes */
  const uint8_t* seq;
  const uint64_t* qual_off;  /* bytes */
  const uint8_t* qual;
  const uint64_t* name_off;  /* n+1 entries, bytes; may be NULL */
  const uint8_t* name;       /* may be NULL */
} bp_record_batch;

typedef struct bp_handle bp_handle;

/* Timing of the last bp_finish()/bp_compute(), from HIP events on the handle's stream. */
typedef struct bp_timing {
  float plan_ms;     /* window planning kernels */
  float pileup_ms;   /* the pileup (count + flush) kernel */
  float total_ms;    /* first launch to last completion */
  int64_t n_windows;
  int64_t n_items;   /* work items launched (windows x slices) */
  float per_read_ms; /* the gather kernel of the last bp_per_read_build (0 before one) */
  float per_read_scan_ms; /* ... and the scan of the counts before it (kernels + one host round trip) */
  /* host time inside all bp_submit calls of the job so far: the serial pass in record order (order
   * check, data errors, the firstread table of stitch mode) and the passes that run on all packing
   * threads (spans, filters, name hashes, packing into the pinned staging buffers) */
  float submit_scan_ms;
  float submit_pack_ms;
  /* stitch mode: the kernel that works out the pair rule (addReadPair, pileup.go:406-446) for every
   * pair of the job and leaves the outcome in the packed records, summed over the computes that
   * ran it (normally the first one after the last submit; 0 otherwise) */
  float stitch_resolve_ms;
} bp_timing;

/* Library/runtime probe.  Fails (BP_EDEVICE) when no HIP device is usable: there is no
 * CPU fallback in this library. */
int bp_init(int32_t device);

/* Number of HIP devices the library can run jobs on (bp_opts.device < this), or a negative BP_E*
 * code.  For callers that spread their jobs over the GPUs of a node as the reference spreads them
 * over goroutines (pileup.go:862-932). */
int bp_device_count(void);

/* newPileupMutable + pileupContext set-up for one job (pileup.go:189-209, :885-892).
 * ref_len[n_ref] = header reference lengths (sam.Header.Refs()). */
int bp_create(const bp_opts* opts, int32_t n_ref, const int32_t* ref_len,
              const char* const* ref_name /* may be NULL; used in error messages only */,
              bp_handle** out);

/* Optional capacity hint (records, payload bytes) so device buffers are sized once. */
int bp_reserve(bp_handle* h, int64_t n_records, int64_t payload_bytes);

/* pCtx.bedPart for this job (pileup.go:896-909): the sorted endpoint array of one
 * reference as returned by BEDUnion.Subset(...).EndpointsByID(ref_id)
 * (interval/bedunion.go:326-328, :753-852): 2N ints, interval k = [e[2k], e[2k+1]).
 * Must be called before the first bp_submit; references never set have no positions. */
int bp_set_bed(bp_handle* h, int32_t ref_id, const int32_t* endpoints, int32_t n_endpoints);

/* The body of the iter.Scan() loop of processShard (pileup.go:671-808) for a run of
 * records.  Records must continue the (ref_id,pos) order of earlier submits; the caller is
 * responsible for the shard iterator semantics (padding, de-duplication of the overlap,
 * pileup.go:673-689).  Copies the batch into pinned staging and starts the H2D copy. */
int bp_submit(bp_handle* h, const bp_record_batch* batch);

/* The same for records as they lie in an inflated BAM stream: what gbam.Unmarshal
 * (encoding/bam/unmarshal.go:126-224) decodes for the iterator of bamprovider.go:417-435, done
 * here on the device.  Record k starts - with its block_size field - at bam + rec_off[k];
 * rec_off ascends, records do not overlap and end at or before n_bytes (at most 4 GiB a call).
 * The host reads only the fixed fields, CIGARs and (with stitch) names; the bytes are copied to
 * the GPU as they are - straight from `bam` when that is pinned host memory and at least three
 * quarters of [first record, last record] are wanted, else gathered into pinned staging first -
 * and a kernel packs them.  `bam` is the caller's again when the call returns.  Calls may be mixed
 * with bp_submit within a job.  A record that does not fit its block_size or the bytes given fails
 * the call (BP_EINVAL) before anything of it is taken. */
int bp_submit_bam(bp_handle* h, const uint8_t* bam, size_t n_bytes, const uint32_t* rec_off, int64_t n);

/* finishRef + w.Finish (pileup.go:624-639, :934-938): no more records; computes every
 * row.  Blocks until the rows are resident on the device. */
int bp_finish(bp_handle* h);

/* Re-run the device computation on the already-resident records (what bp_finish runs),
 * for measurement and for callers that change nothing but want fresh rows. */
int bp_compute(bp_handle* h);

/* bp_compute for callers that cannot allocate (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc).  NO COUNTERPART IN THE REFERENCE: it exists for those
 * callers only.  Runs the same kernels as bp_compute over the same resident records and BED and
 * writes the same rows, but keeps the kernels' scratch (the plan's record range per window, the
 * work-item queue of deep windows and its counter, the device error flags) in a static device
 * array of 8 MiB + 512 B that every GPU the library runs on holds until the process ends, and
 * calls no hipMalloc / hipFree / allocating routine.  A bp_finish must have run on the handle
 * (it uploads what the kernels read).  Limits, refused with BP_EINVAL: 524,288 windows of 1,024
 * positions (536,870,912 window positions) and 262,144 work items of deep windows (a job can
 * have up to records / 512 + 18 of them with spans below 1,024: about 134 M records).  All calls on one GPU share that
 * GPU's array: calls that overlap on one GPU (two handles, two threads) race; calls one after
 * the other are independent, whatever the earlier one left behind. */
int bp_compute_static(bp_handle* h);

/* Number of rows = number of BED positions of the job (one PileupRow per position,
 * zero-depth ones included: pileup.go:518-570, pileup_support.go:89-123). */
int64_t bp_num_rows(const bp_handle* h);

/* Copy rows [first, first+n) to host, in ascending (ref_id,pos) order.  Any of ref_id,
 * pos, vals may be NULL.  vals receives n*BP_ROW_U32 uint32. */
int bp_rows(bp_handle* h, int64_t first, int64_t n, int32_t* ref_id, int32_t* pos, uint32_t* vals);

/* Device pointer to the row values (num_rows*BP_ROW_U32 uint32, valid until bp_destroy or
 * the next bp_compute) for consumers that keep working on the GPU. */
int bp_rows_device(bp_handle* h, const uint32_t** dptr);

/* Boundary exchange between the devices of a read-disjoint partition (SURVEY 8e scheme B):
 * when every read is given to exactly one device, the device that owns positions
 * [start, end) also accumulates, in the rows of [end, end + T), what its reads contribute
 * past its end; those rows belong to its right neighbour.  bp_rows_export copies the values
 * of rows [first, first+n) into device memory `dst` (n*BP_ROW_U32 uint32, e.g. a send
 * buffer); bp_rows_accumulate adds n*BP_ROW_U32 uint32 at device address `src` (what the
 * left neighbour exported) onto rows [first, first+n).  Both run on the job's stream and
 * return when done.  Counts are commutative u32 sums (pileup.go:223-231), so the merged
 * rows equal those of one job over the whole range; the reference has no such step (its
 * jobs re-read a halo of records instead, pileup.go:896-909). */
int bp_rows_export(bp_handle* h, int64_t first, int64_t n, uint32_t* dst_device);
int bp_rows_accumulate(bp_handle* h, int64_t first, int64_t n, const uint32_t* src_device);
/* The same two, only enqueued on the job's stream (no wait): for an exchange that is ordered
 * against the collective with stream events instead of host synchronisations.  bp_stream
 * hands out that stream (a hipStream_t; kernels, copies and these calls of one handle all run
 * on it, in call order) so that the caller can make its own stream wait for it and back. */
int bp_rows_export_async(bp_handle* h, int64_t first, int64_t n, uint32_t* dst_device);
int bp_rows_accumulate_async(bp_handle* h, int64_t first, int64_t n, const uint32_t* src_device);
int bp_stream(bp_handle* h, void** stream);

/* Sum over all rows of each of the BP_ROW_U32 columns (uint64 each), reduced on device.
 * Size-independent parity check: sums[0] = total depth. */
int bp_column_sums(bp_handle* h, uint64_t sums[BP_ROW_U32]);

/* Per-read feature lists: what appendBase adds to PileupPayload.PerRead when a per-read
 * column set is requested (pCtx.perReadNeeded, pileup.go:235-250, :888; PerReadFeatures
 * row.go:22-31).  List b (A,C,G,T = 0..3) of a row holds one entry per passing non-X base of
 * that position, in read-arrival order; its length is Counts[b][0]+Counts[b][1]. */
typedef struct bp_per_read {
  uint16_t dist5p;  /* index of the base in its read (pileup.go:244) */
  uint16_t fraglen; /* len(Qual) of the read (pileup.go:245) */
  uint8_t qual;     /* after clipQuals */
  uint8_t strand;   /* pileup.GetStrand of the read: 0 none, 1 fwd, 2 rev */
} bp_per_read;

/* Builds the lists of the computed rows on the device (after bp_finish / bp_compute).  With
 * opts.stitch it fails with the reference's panic text (pileup.go:422-443). */
int bp_per_read_build(bp_handle* h);
/* Total number of entries over all rows, or -1 before bp_per_read_build. */
int64_t bp_per_read_total(const bp_handle* h);
/* off[4*i + b] .. off[4*i + b + 1] = entries of list b of row first_row+i; writes
 * 4*n_rows + 1 absolute entry indices. */
int bp_per_read_offsets(bp_handle* h, int64_t first_row, int64_t n_rows, uint64_t* off);
/* Copies entries [first, first+n) to host. */
int bp_per_read_features(bp_handle* h, uint64_t first, uint64_t n, bp_per_read* out);

/* Output conversion on the device: ConvertPileupRowsToBasestrandTSV (pileup/snp/output.go:376-453)
 * and ConvertPileupRowsToTSV (output.go:42-300, writeChromPosRef :36-40) for column sets without
 * per-read lists.  Rows [first, first+n) become the lines of one output file:
 *   BP_TEXT_BASESTRAND  "<chrom>\t<pos+1>\t<REF>\tA+\tA-\tC+\tC-\tG+\tG-\tT+\tT-\n"
 *   BP_TEXT_TSV_REF     "<chrom>\t<pos+1>\t<REF>[\tDP][\tref_depth_tier1][\t0]\n"          (.ref.tsv)
 *   BP_TEXT_TSV_ALT     one "<chrom>\t<pos+1>\t<REF>\t<ALT>[\tDP][\talt_depth_tier1][\t0]\n" per base
 *                       A,C,G,T,N other than REF with a nonzero count                         (.alt.tsv)
 * `cols` = the -cols bit set (pileup.ParseCols, pileup/common.go:118-166) restricted to the BP_COL_*
 * below; it is ignored for BP_TEXT_BASESTRAND.  The lines are rendered on the device and copied
 * to `out` (host memory of at least bp_render_text_bound(h, kind, n) bytes); *len = bytes
 * written.  With out = NULL the text stays on the device, *len says how long it is, and
 * bp_render_fetch copies it to a buffer of that size.  REF comes from the Seq8 bases
 * (encoding/fasta/fasta.go:50-62) given by bp_set_ref_seq8 for ONE reference, positions
 * [start, start+n): every rendered row must lie there (the reference indexes refSeqs[refID][pos],
 * output.go:158).  <chrom> is the ref_name given to bp_create.
 * BP_TEXT_STRAND_FWD / _REV, or-ed into `kind`: the lines of ONE strand's -per-strand pass
 * (pileup.go:718-723, :1118-1125) from the split-strand rows of a single pass: that strand's counts
 * as the row's counts and their sum as DP.  Only for jobs without a quality rule (min_base_qual
 * 0, no stitching), where every base is in a count slot; refused otherwise. */
enum { BP_TEXT_BASESTRAND = 0, BP_TEXT_TSV_REF = 1, BP_TEXT_TSV_ALT = 2, BP_TEXT_STRAND_FWD = 0x100, BP_TEXT_STRAND_REV = 0x200 };
enum { BP_COL_DPREF = 1, BP_COL_DPALT = 2, BP_COL_HIGHQ = 64, BP_COL_LOWQ = 128 };
int bp_set_ref_seq8(bp_handle* h, int32_t ref_id, int32_t start, const uint8_t* seq8, int64_t n);
int64_t bp_render_text_bound(const bp_handle* h, int32_t kind, int64_t n_rows);
int bp_render_text(bp_handle* h, int32_t kind, int32_t cols, int64_t first, int64_t n, char* out, size_t cap,
                   size_t* len);
int bp_render_fetch(bp_handle* h, char* out, size_t cap);
/* The same text as BGZF: what the `-bgz` formats write (bgzf.NewWriter around the tsv writer,
 * pileup/snp/output.go:66-82, :388-392; the reference compresses on `parallelism` host threads).  The
 * text the last bp_render_text(out = NULL) left on the device is compressed there and copied to `out`
 * (at least bp_render_bgzf_bound(h) bytes) as a run of complete BGZF members (SAM spec 4.1; up to
 * 0xff00 bytes of text each, DEFLATE with Huffman codes built for this text, stored when that is
 * shorter; no EOF member: the file's writer appends it once).  *len = bytes written.  With out = NULL
 * the members stay on the device, *len says how long they are, and a second call copies them to a
 * buffer of that size (until the next bp_render_text).  The bytes are
 * this compressor's own; what is pinned is that they inflate to exactly the text bp_render_fetch
 * returns. */
int64_t bp_render_bgzf_bound(const bp_handle* h);
int bp_render_fetch_bgzf(bp_handle* h, char* out, size_t cap, size_t* len);

int bp_get_timing(const bp_handle* h, bp_timing* out);

/* Algorithmic bytes of the resident job per SURVEY.md section 8(d):
 * sum_reads(32 + 4*n_cigar + ceil(l_seq/2) + l_seq) + 44*rows. */
int64_t bp_algorithmic_bytes(const bp_handle* h);
/* Records submitted so far and the device bytes their payloads take: what a caller that opens one
 * job after another of similar size passes to the next job's bp_reserve (the C++ driver does), so
 * that its buffers are taken at their final size instead of grown step by step.  No counterpart
 * in the reference (Go slices grow by themselves). */
int bp_job_size(const bp_handle* h, int64_t* n_records, int64_t* payload_bytes);

const char* bp_last_error(const bp_handle* h); /* h may be NULL: last create/init error */
void bp_destroy(bp_handle* h);

/* ---- BGZF inflate on the device ---------------------------------------
 * Replaces what runs in front of processShard in the reference: bam.Reader over bgzf.Reader
 * behind bamprovider's iterator (encoding/bamprovider/bamprovider.go:417-435; the reader itself is
 * biogo/hts bgzf, a go.mod dependency absent from the tree: BGZF per SAM spec 4.1, DEFLATE per
 * RFC 1951).  The caller parses the BGZF block headers (18 bytes: offsets and BSIZE) and
 * trailers (CRC32, ISIZE) - byte shuffling on a few bytes per 64 KiB - and hands over the
 * compressed bytes as they lie in the file with one bp_bgzf_block per block; the device
 * inflates every block (one wavefront per block), checks ISIZE and, when asked, the CRC-32. */
typedef struct bp_bgzf_block {
  uint32_t in_off;  /* the block's deflate data (behind its header) in `comp` */
  uint32_t in_len;  /* BSIZE + 1 - header - 8 */
  uint32_t out_off; /* where its bytes go in the output: normally the running sum of ISIZE */
  uint32_t out_len; /* ISIZE (at most 65536); blocks of ISIZE 0 need not be passed */
  uint32_t crc32;   /* the trailer's CRC32 */
} bp_bgzf_block;
typedef struct bp_inflater bp_inflater;

/* One inflater = one HIP stream with its device buffers; calls on one inflater are serial,
 * inflaters are independent (as handles are). */
int bp_inflate_create(int32_t device, bp_inflater** out);
void bp_inflate_destroy(bp_inflater* f);
/* Inflates n_blocks blocks.  comp/comp_bytes: host memory (pinned memory is copied from
 * directly); out_host (may be NULL) receives out_bytes of output, which also stays on the device
 * until the inflater's next call (bp_inflate_device_output).  Returns BP_EDATA for a block that
 * does not inflate to ISIZE bytes with the trailer's checksum - a corrupt or truncated block,
 * which the reference's reader refuses too - with its index in *bad_block (may be NULL), BP_EINVAL
 * for a block table that does not fit the bytes given. */
int bp_inflate_blocks(bp_inflater* f, const uint8_t* comp, size_t comp_bytes, const bp_bgzf_block* blocks,
                      int64_t n_blocks, uint8_t* out_host, size_t out_bytes, int32_t check_crc, int64_t* bad_block);
/* The same, with the output written to device memory of the caller's (out_device, out_bytes of it,
 * on the inflater's device) and not copied anywhere: for callers that keep the inflated stream
 * on the device (bp_bam_scan, bp_submit_bam_device). */
int bp_inflate_blocks_to(bp_inflater* f, const uint8_t* comp, size_t comp_bytes, const bp_bgzf_block* blocks,
                         int64_t n_blocks, uint8_t* out_device, size_t out_bytes, int32_t check_crc, int64_t* bad_block);
/* The last call's output on the device (out_bytes of it), valid until the next call or destroy;
 * bp_inflate_stream: the stream it was written on (for callers that order other work behind it). */
const uint8_t* bp_inflate_device_output(const bp_inflater* f);
int bp_inflate_stream(bp_inflater* f, void** stream);
/* HIP-event times of the last call's inflate and CRC kernels */
int bp_inflate_timing(const bp_inflater* f, float* inflate_ms, float* crc_ms);
/* The writer's side (the counterpart of bp_inflate; bgzf.Writer in the reference's output path,
 * output.go:70-71, :391): n bytes of host memory compressed on the inflater's device into complete
 * BGZF members of up to 0xff00 input bytes each, copied to `out` (at least bp_deflate_bgzf_bound(n)
 * bytes; no EOF member).  Synchronous.  The inflater keeps about 6 bytes of device memory per input
 * byte of its largest call (tokens, member slots, output). */
size_t bp_deflate_bgzf_bound(size_t n);
/* The host half of that writer by itself (no device call; exported so that it can be tested without
 * one): from symbol counts hist[0..286) (literals, end of block, lengths; RFC 1951 3.2.5) and
 * hist[288..318) (distances) the canonical Huffman codes, lengths limited to 15 bits, and the dynamic
 * block header that announces them (RFC 1951 3.2.7, BFINAL set).  out: 288 + 32 codes (bit-reversed,
 * length in bits 16-19), the header's length in bits, then its bits in 128 dwords. */
int bp_deflate_codes(const uint32_t* hist, uint64_t n_members, uint32_t* out);
int bp_deflate_bgzf(bp_inflater* f, const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* len);
const char* bp_inflate_last_error(const bp_inflater* f);

/* ---- BAM record scan on the device (second half) -----------------------
 * What bam.Reader.Read does behind bamprovider's iterator (bamprovider.go:417-435; record layout
 * per encoding/bam/unmarshal.go:126-224): find the records of an inflated stream and read their
 * fixed fields - here with the stream resident on the device (bp_inflate_blocks left it there),
 * so that a record's ~270 bytes never visit the host: 45 bytes of columns per record do.
 * One column entry per record, in stream order; host memory owned by the caller. */
typedef struct bp_bam_columns {
  int64_t cap;      /* entries each array has room for */
  uint32_t* off;    /* the record's block_size field, as an offset into the stream */
  int32_t* ref_id;
  int32_t* pos;
  int32_t* mate_ref_id;
  int32_t* mate_pos;
  uint16_t* flags;
  uint8_t* mapq;
  uint8_t* l_name;  /* l_read_name, with the NUL */
  uint32_t* n_cigar;
  uint32_t* l_seq;
  uint32_t* bytes;  /* 4 + block_size */
  int32_t* span;    /* reference bases the CIGAR consumes (sam.Cigar.Lengths) */
  uint8_t* facts;   /* bit 0: one M op; bit 1: an op outside MIDNSH; bit 2: the fields do not fit the block */
  /* Optional (names NULL: not wanted): the read names, one after the other without their NULs,
   * record i's at names[name_off[i], name_off[i + 1]) - what stitching pairs reads by
   * (firstread.go:135-152), so that a stitched job can take its records from the device stream too.
   * name_off has room for cap + 1 entries, names for names_cap bytes. */
  uint32_t* name_off;
  uint8_t* names;
  int64_t names_cap;
} bp_bam_columns;
/* Scans stream bytes [first, end) on the device (device pointer `stream`, on the inflater's
 * device; `first` = the start of a record) for whole records: *n of them, columns filled, *exit
 * = the offset behind the last whole record, *stopped = 1 when a record cut off by `end` (or an
 * invalid block size) follows there.  The chain of block sizes is walked in parallel stretches
 * from guessed starts, every guess verified in order against the walk before it and a stretch
 * whose guess was wrong or missing walked again from the right place, so the records found are
 * those of a serial walk.  Returns 1 (nothing filled) when a stretch holds more records than the
 * scan has room for (records of a few bytes: not BAM reads) and the caller should walk this
 * stream on the host, a negative BP_E* code on errors (BP_ENOMEM: more records than cols->cap, or
 * more name bytes than cols->names_cap). */
int bp_bam_scan(bp_inflater* f, const uint8_t* stream, uint64_t first, uint64_t end, int32_t n_ref,
                bp_bam_columns* cols, int64_t* n, uint64_t* exit, int32_t* stopped);
/* bp_submit_bam for records [first_record, first_record + n) of such a scan: `stream` is the
 * DEVICE address of the stream the columns' offsets refer to (on the job's device); nothing of
 * the records is read on the host.  A job with stitching needs the scan's names (cols->names: the
 * mate table is keyed by read names).  Not for records whose facts carry bit 1 or bit 2 (the
 * caller hands such records over with bp_submit_bam from a host copy).  The pack kernel
 * loads whole aligned 4-byte words: the word that holds the stream's last byte must be readable
 * (any hipMalloc'ed buffer is). */
int bp_submit_bam_device(bp_handle* h, const uint8_t* stream, size_t n_bytes, const bp_bam_columns* cols,
                         int64_t first_record, int64_t n);

#ifdef __cplusplus
}
#endif
#endif /* BIO_PILEUP_H_ */
