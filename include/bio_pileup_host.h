/*
 * bio_pileup_host.h -- C ABI of the host layers of libbiopileup.so that sit on either side of
 * the device hot path (include/bio_pileup.h): the snp.Pileup driver, the bio-pileup command
 * line, and the small parsers the driver is built from.  Plain C signatures; every function
 * that can fail returns 0 on success or -1 with a message in `err` (NUL-terminated, cut to
 * errlen).  Counts are only ever computed on the GPU: bio_pileup_run fails without one.
 */
#ifndef BIO_PILEUP_HOST_H_
#define BIO_PILEUP_HOST_H_

#include <stddef.h>
#include <stdint.h>

#include "bio_pileup.h"

#ifdef __cplusplus
extern "C" {
#endif

/* snp.Opts (pileup/snp/pileup.go:36-54); defaults = snp.DefaultOpts (:56-68). Strings may be
 * NULL (= ""). */
typedef struct bio_pileup_opts {
  const char* bed_path;
  const char* region;
  const char* bam_index_path;
  int32_t clip;
  const char* cols;
  int32_t flag_exclude;
  int32_t mapq;
  int32_t max_read_len;
  int32_t max_read_span;
  int32_t min_bag_depth;
  int32_t min_base_qual;
  int32_t parallelism;
  int32_t per_strand;
  int32_t remove_sq;
  int32_t stitch;
  const char* temp_dir;
  int32_t device; /* HIP device ordinal; no counterpart in the reference */
  /* spread the jobs over this many logical devices, mapped round-robin onto the GPUs present
   * (-1: one per GPU; 0: everything on `device`): the reference's -parallelism
   * (pileup.go:862-932) with devices for goroutines */
  int32_t devices;
} bio_pileup_opts;

/* snp.DefaultOpts (pileup.go:56-68) */
void bio_pileup_default_opts(bio_pileup_opts* o);

/* snp.Pileup(ctx, xampath, fapath, format, outPrefix, opts, nil) (pileup.go:970-1130) */
int bio_pileup_run(const char* xampath, const char* fapath, const char* format, const char* out_prefix,
                   const bio_pileup_opts* opts, char* err, size_t errlen);

/* cmd/bio-pileup main() (main.go:60-98); returns the process exit status */
int bio_pileup_main(int argc, char** argv);

/* interval.ParseRegionString (interval/bedunion.go:593-651) */
int bio_parse_region(const char* region, char* ref_name, size_t ref_name_len, int32_t* start0, int32_t* end,
                     char* err, size_t errlen);

/* pileup.ParseCols with snp's column names (pileup/common.go:118-166, pileup.go:134-157) */
int bio_parse_cols(const char* cols, int32_t default_bitset, int32_t* bitset, char* err, size_t errlen);

/* interval.NewBEDUnionFromPath with a SAM header (bedunion.go:563-578, :343-364), then
 * optionally Subset (bedunion.go:753-852) when subset[] = {startRef, startPos, limitRef,
 * limitPos} is non-NULL.  Returns a handle; endpoints are read per reference ID. */
typedef struct bio_bed bio_bed;
int bio_bed_load(const char* path, int32_t n_ref, const char* const* ref_names, const int32_t* subset,
                 bio_bed** out, char* err, size_t errlen);
/* EndpointsByID (bedunion.go:326-328): returns the count; copies up to cap endpoints */
int32_t bio_bed_endpoints(const bio_bed* b, int32_t ref_id, int32_t* endpoints, int32_t cap);
/* IntersectsByID (bedunion.go:219-254) */
int bio_bed_intersects(const bio_bed* b, int32_t ref_id, int32_t start, int32_t limit);
void bio_bed_free(bio_bed* b);

/* fasta.New(..., OptEncoding(Seq8)) + pileup.FaToStringSlice for one contig
 * (encoding/fasta/fasta.go:132-175, pileup/common.go:194-225): copies up to cap Seq8 bytes
 * of sequence `name`, returns its length or -1. */
int64_t bio_fasta_seq8(const char* path, const char* name, uint8_t* seq8, int64_t cap, char* err,
                       size_t errlen);

/* BAM reader used by the driver (record layout per encoding/bam/unmarshal.go:126-224):
 * counts records and sums l_seq and n_cigar over the file; for tests of the decoder. */
int bio_bam_scan(const char* path, int64_t* n_records, int64_t* sum_l_seq, int64_t* sum_n_cigar,
                 int32_t* n_ref, char* err, size_t errlen);

/* The same file read both ways the driver's reader offers - record by record, and a chunk of
 * whole records at a time (the chain of block sizes walked in parallel stretches from guessed
 * starts, every guess checked against the walk before it) - with an order-sensitive hash of
 * the records' fixed fields each way; for tests: the hashes must agree.
 * n_rewalked: stretches whose guessed start was wrong. */
int bio_bam_scan_chunks(const char* path, int64_t* n_records, int64_t* n_chunks, uint64_t* hash_records,
                        uint64_t* hash_chunks, int32_t* sorted, int32_t* fields_ok, int64_t* n_rewalked, char* err,
                        size_t errlen);

/* Index lookup + seek as the driver does them (bamIterator.reset, bamprovider.go:286-348;
 * .gbai: encoding/bam/gindex.go:35-121; .bai: SAM specification 5.2): loads index_path
 * (.bai or .gbai by suffix), asks for (ref_id, pos), seeks there and reads to the end of the
 * file.  *found = 0 when the index shows no record at or after the target.  *first_ref/pos =
 * the first record met, *n_skipped = records met that lie before the target,
 * *n_at_or_after = the rest (unmapped included); for tests of the index readers. */
int bio_bam_seek_scan(const char* bam_path, const char* index_path, int32_t ref_id, int32_t pos,
                      int32_t* found, int32_t* first_ref, int32_t* first_pos, int64_t* n_skipped,
                      int64_t* n_at_or_after, char* err, size_t errlen);

/* Writes the records of `batch` (coordinate-sorted; names optional) as a BAM file: BGZF at
 * zlib `level`, compressed on `threads` threads.  No counterpart on the reference's pileup
 * path (which only reads BAMs): input generator for tools and tests. */
int bio_bam_write(const char* path, int32_t n_ref, const char* const* ref_names, const int32_t* ref_lens,
                  const bp_record_batch* batch, int32_t level, int32_t threads, char* err, size_t errlen);

/* ConvertPileupRowsToTSV (pileup/snp/output.go:42-300) and ConvertPileupRowsToBasestrandTSV
 * (output.go:376-575) with the rows handed over in memory instead of in the per-job recordio
 * temp files: writes <out_prefix>.ref.tsv + .alt.tsv (format "tsv"/"tsv-bgz") or
 * <out_prefix>.basestrand.tsv ("basestrand-tsv"/"-bgz"), header included, with the product's
 * writers (the ones bio_pileup_run uses).  cols = the -cols string ("" = dpref,highq,lowq,
 * pileup.go:1015).  ref_seq8[r] = Seq8 bases of reference r (ref_seq_len[r] of them; may be NULL
 * for references no row lies on).  vals = n_rows x BP_ROW_U32; off/feats = per-read lists as
 * bp_per_read_offsets / bp_per_read_features give them (off[4*n_rows+1] absolute entry indices,
 * feats pointing at entry off[0]); both NULL without a per-read column. */
int bio_pileup_write_rows(const char* format, const char* out_prefix, const char* cols, int32_t n_ref,
                          const char* const* ref_names, const uint8_t* const* ref_seq8, const int64_t* ref_seq_len,
                          int64_t n_rows, const int32_t* ref_id, const int32_t* pos, const uint32_t* vals,
                          const uint64_t* off, const bp_per_read* feats, char* err, size_t errlen);

/* ---- record-level wire formats (the recordio container around them is grailbio/base's and is
 * not built) */

/* PileupRow.FieldsPresent as flushToInternal sets it (pileup/snp/pileup.go:518-570, bits
 * row.go:33-40): 0 for a zero-depth row, else FieldCounts | FieldPerRead{A,C,G,T} for the
 * non-empty lists when per_read_needed. */
uint32_t bio_pileup_row_fields(const uint32_t* vals, const uint32_t list_len[4], int32_t per_read_needed);
/* MarshalPileupRow (pileup/snp/row.go:113-175).  vals = BP_ROW_U32 uint32 (Depth, Counts);
 * lists[b] = list_len[b] entries of base b.  Returns the record's size; writes it when out has
 * room (out NULL or cap too small: size query), -1 on a bad argument. */
int64_t bio_marshal_pileup_row(uint32_t fields_present, uint32_t ref_id, uint32_t pos, const uint32_t* vals,
                               const uint32_t list_len[4], const bp_per_read* const lists[4], uint8_t* out,
                               size_t cap);
/* unmarshalPileupRow (row.go:179-227).  Returns the number of list entries (feats receives the
 * first feats_cap of them, lists A,C,G,T back to back), -1 on a truncated record. */
int64_t bio_unmarshal_pileup_row(const uint8_t* in, size_t len, uint32_t* fields_present, uint32_t* ref_id,
                                 uint32_t* pos, uint32_t* vals, uint32_t list_len[4], bp_per_read* feats,
                                 size_t feats_cap);
/* marshalBaseStrand (pileup/snp/basestrand.go:103-135) for n rows: 40 bytes each, refID, pos,
 * A+ A- C+ C- G+ G- T+ T- (vals = n x BP_ROW_U32 as bp_rows returns them). */
void bio_marshal_basestrand(int64_t n, const int32_t* ref_id, const int32_t* pos, const uint32_t* vals, uint8_t* out);
/* BaseStrandUnmarshaller.UnmarshalBaseStrand (basestrand.go:183-201); counts8 = n x 8. */
void bio_unmarshal_basestrand(int64_t n, const uint8_t* in, int32_t* ref_id, int32_t* pos, uint32_t* counts8);
/* baseStrandsRioTrailer / parseBaseStrandsTrailer (basestrand.go:77-101) */
void bio_basestrand_rio_trailer(int64_t num_piles, uint8_t out[16]);
int bio_parse_basestrand_rio_trailer(const uint8_t* trailer, size_t len, int64_t* num_piles, char* err, size_t errlen);
/* WriteBaseStrandToTSV (basestrand.go:204-231; ref_bases NULL: no REF column, no '#', POS as
 * stored) or WriteBaseStrandTsv (basestrand.go:282-290; ref_bases[i] = REF of row i).  Returns
 * the text's length; writes it when out has room. */
int64_t bio_basestrand_piles_to_tsv(int64_t n, const int32_t* ref_id, const int32_t* pos, const uint32_t* counts8,
                                    int32_t n_ref, const char* const* ref_names, const char* ref_bases, char* out,
                                    size_t cap);


/* ---- readers of the output (the exported consumers of basestrand files) */

/* BaseStrandTsvRow (pileup/snp/basestrand.go:247-259).  counts = FwdA RevA FwdC RevC FwdG RevG
 * FwdT RevT, the file's column order; chr and ref are NUL-terminated. */
typedef struct bio_basestrand_tsv_row {
  char chr[64];
  int64_t pos;
  char ref[8];
  int64_t counts[8];
} bio_basestrand_tsv_row;

/* ReadBaseStrandTsv (basestrand.go:262-279) over a file's text in memory: lines starting with
 * '#' are comments (that hides the header), columns map to the row's fields by position.
 * Returns the number of rows (the first cap of them are written; rows NULL: count only) or -1
 * with strconv's wording in err ("... invalid syntax") for a column that is no integer. */
int64_t bio_read_basestrand_tsv(const char* text, size_t len, bio_basestrand_tsv_row* rows, int64_t cap, char* err,
                                size_t errlen);
/* ReadSingleStrandBaseStrandTsv (basestrand.go:309-367): the forward- and reverse-strand files
 * (header row CHROM POS DEPTH REF A C G T N INS DEL, columns found by name) read in lockstep
 * into one row per position: Fwd* from the first, Rev* from the second; CHROM/POS and REF must
 * agree (the reference's two error texts). */
int64_t bio_read_single_strand_basestrand_tsv(const char* fwd, size_t fwd_len, const char* rev, size_t rev_len,
                                              bio_basestrand_tsv_row* rows, int64_t cap, char* err, size_t errlen);
/* ChrId (basestrand.go:401-422): "chr1".."chr22" -> 1..22, chrX 23, chrY 24, chrM 25. */
int bio_chr_id(const char* chr, int32_t* id, char* err, size_t errlen);

#ifdef __cplusplus
}
#endif
#endif /* BIO_PILEUP_HOST_H_ */
