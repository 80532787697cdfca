"""ctypes mirror of include/bio_pileup.h.  Keep field order identical to the header."""
import ctypes as C
import os

BP_ROW_U32 = 11
BP_OK, BP_EINVAL, BP_EDEVICE, BP_ENOMEM, BP_EDATA, BP_EUNSORTED = 0, -1, -2, -3, -4, -5
STRAND_NONE, STRAND_FWD, STRAND_REV = 0, 1, 2

_HERE = os.path.dirname(os.path.abspath(__file__))
# BIO_AMD_LIB selects another build of the same library (diagnostic / tuning variants)
LIB_PATH = os.environ.get("BIO_AMD_LIB") or os.path.join(_HERE, "libbiopileup.so")
SYNTH_LIB_PATH = os.path.join(_HERE, "libbpsynth.so")


class bp_opts(C.Structure):
    _fields_ = [
        ("clip", C.c_int32),
        ("flag_exclude", C.c_int32),
        ("mapq", C.c_int32),
        ("max_read_len", C.c_int32),
        ("max_read_span", C.c_int32),
        ("min_base_qual", C.c_int32),
        ("stitch", C.c_int32),
        ("ignore_strand", C.c_int32),
        ("strand_req", C.c_int32),
        ("device", C.c_int32),
        ("reserved", C.c_int32 * 6),
    ]


class bp_record_batch(C.Structure):
    _fields_ = [
        ("n", C.c_int64),
        ("ref_id", C.POINTER(C.c_int32)),
        ("pos", C.POINTER(C.c_int32)),
        ("mate_ref_id", C.POINTER(C.c_int32)),
        ("mate_pos", C.POINTER(C.c_int32)),
        ("flags", C.POINTER(C.c_uint16)),
        ("mapq", C.POINTER(C.c_uint8)),
        ("n_cigar", C.POINTER(C.c_uint32)),
        ("l_seq", C.POINTER(C.c_uint32)),
        ("cigar_off", C.POINTER(C.c_uint64)),
        ("cigar", C.POINTER(C.c_uint32)),
        ("seq_off", C.POINTER(C.c_uint64)),
        ("seq", C.POINTER(C.c_uint8)),
        ("qual_off", C.POINTER(C.c_uint64)),
        ("qual", C.POINTER(C.c_uint8)),
        ("name_off", C.POINTER(C.c_uint64)),
        ("name", C.POINTER(C.c_uint8)),
    ]


class bp_timing(C.Structure):
    _fields_ = [
        ("plan_ms", C.c_float),
        ("pileup_ms", C.c_float),
        ("total_ms", C.c_float),
        ("n_windows", C.c_int64),
        ("n_items", C.c_int64),
        ("per_read_ms", C.c_float),
        ("per_read_scan_ms", C.c_float),
        ("submit_scan_ms", C.c_float),
        ("submit_pack_ms", C.c_float),
        ("stitch_resolve_ms", C.c_float),
    ]


EXPORTS = [
    "bp_init", "bp_device_count", "bp_create", "bp_reserve", "bp_set_bed", "bp_submit", "bp_submit_bam", "bp_finish", "bp_compute", "bp_compute_static",
    "bp_num_rows", "bp_rows", "bp_rows_device", "bp_rows_export", "bp_rows_accumulate", "bp_rows_export_async", "bp_rows_accumulate_async", "bp_stream", "bp_column_sums", "bp_get_timing",
    "bp_algorithmic_bytes",
    "bp_job_size", "bp_last_error", "bp_destroy",
    "bp_per_read_build", "bp_per_read_total", "bp_per_read_offsets", "bp_per_read_features",
    "bp_set_ref_seq8", "bp_render_text_bound", "bp_render_text", "bp_render_fetch", "bp_render_bgzf_bound", "bp_render_fetch_bgzf", "bp_deflate_bgzf_bound", "bp_deflate_bgzf", "bp_deflate_codes",
    "bp_inflate_create", "bp_inflate_destroy", "bp_inflate_blocks", "bp_inflate_blocks_to", "bp_inflate_device_output", "bp_inflate_stream",
    "bp_inflate_timing", "bp_inflate_last_error", "bp_bam_scan", "bp_submit_bam_device",
]

HOST_EXPORTS = [
    "bio_pileup_default_opts", "bio_pileup_run", "bio_pileup_main", "bio_parse_region",
    "bio_parse_cols", "bio_bed_load", "bio_bed_endpoints", "bio_bed_intersects", "bio_bed_free",
    "bio_fasta_seq8", "bio_bam_scan", "bio_bam_scan_chunks", "bio_bam_seek_scan", "bio_bam_write",
    "bio_pileup_write_rows", "bio_pileup_row_fields", "bio_marshal_pileup_row", "bio_unmarshal_pileup_row",
    "bio_marshal_basestrand", "bio_unmarshal_basestrand", "bio_basestrand_rio_trailer",
    "bio_parse_basestrand_rio_trailer", "bio_basestrand_piles_to_tsv",
    "bio_read_basestrand_tsv", "bio_read_single_strand_basestrand_tsv", "bio_chr_id",
]


class bio_pileup_opts(C.Structure):
    """include/bio_pileup_host.h: snp.Opts"""
    _fields_ = [
        ("bed_path", C.c_char_p),
        ("region", C.c_char_p),
        ("bam_index_path", C.c_char_p),
        ("clip", C.c_int32),
        ("cols", C.c_char_p),
        ("flag_exclude", C.c_int32),
        ("mapq", C.c_int32),
        ("max_read_len", C.c_int32),
        ("max_read_span", C.c_int32),
        ("min_bag_depth", C.c_int32),
        ("min_base_qual", C.c_int32),
        ("parallelism", C.c_int32),
        ("per_strand", C.c_int32),
        ("remove_sq", C.c_int32),
        ("stitch", C.c_int32),
        ("temp_dir", C.c_char_p),
        ("device", C.c_int32),
        ("devices", C.c_int32),
    ]


class bp_bgzf_block(C.Structure):
    _fields_ = [("in_off", C.c_uint32), ("in_len", C.c_uint32), ("out_off", C.c_uint32), ("out_len", C.c_uint32),
                ("crc32", C.c_uint32)]


class bp_bam_columns(C.Structure):
    _fields_ = [("cap", C.c_int64), ("off", C.c_void_p), ("ref_id", C.c_void_p), ("pos", C.c_void_p),
                ("mate_ref_id", C.c_void_p), ("mate_pos", C.c_void_p), ("flags", C.c_void_p), ("mapq", C.c_void_p),
                ("l_name", C.c_void_p), ("n_cigar", C.c_void_p), ("l_seq", C.c_void_p), ("bytes", C.c_void_p),
                ("span", C.c_void_p), ("facts", C.c_void_p),
                ("name_off", C.c_void_p), ("names", C.c_void_p), ("names_cap", C.c_int64)]


class bio_basestrand_tsv_row(C.Structure):
    """include/bio_pileup_host.h: BaseStrandTsvRow"""
    _fields_ = [("chr", C.c_char * 64), ("pos", C.c_int64), ("ref", C.c_char * 8), ("counts", C.c_int64 * 8)]


_lib = None


def load():
    """Load libbiopileup.so.  Raises if it is missing: there is no fallback path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  bio_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    H = C.c_void_p
    lib.bp_init.argtypes = [C.c_int32]
    lib.bp_init.restype = C.c_int
    lib.bp_create.argtypes = [C.POINTER(bp_opts), C.c_int32, C.POINTER(C.c_int32),
                              C.POINTER(C.c_char_p), C.POINTER(H)]
    lib.bp_create.restype = C.c_int
    lib.bp_reserve.argtypes = [H, C.c_int64, C.c_int64]
    lib.bp_reserve.restype = C.c_int
    lib.bp_set_bed.argtypes = [H, C.c_int32, C.POINTER(C.c_int32), C.c_int32]
    lib.bp_set_bed.restype = C.c_int
    lib.bp_submit.argtypes = [H, C.POINTER(bp_record_batch)]
    lib.bp_submit.restype = C.c_int
    lib.bp_submit_bam.argtypes = [H, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32), C.c_int64]
    lib.bp_submit_bam.restype = C.c_int
    lib.bp_finish.argtypes = [H]
    lib.bp_finish.restype = C.c_int
    lib.bp_compute.argtypes = [H]
    lib.bp_compute.restype = C.c_int
    lib.bp_num_rows.argtypes = [H]
    lib.bp_num_rows.restype = C.c_int64
    lib.bp_rows.argtypes = [H, C.c_int64, C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                            C.POINTER(C.c_uint32)]
    lib.bp_rows.restype = C.c_int
    lib.bp_rows_device.argtypes = [H, C.POINTER(C.c_void_p)]
    lib.bp_rows_device.restype = C.c_int
    lib.bp_rows_export.argtypes = [H, C.c_int64, C.c_int64, C.c_void_p]
    lib.bp_rows_export.restype = C.c_int
    lib.bp_compute_static.argtypes = [H]
    lib.bp_compute_static.restype = C.c_int
    lib.bp_rows_accumulate.argtypes = [H, C.c_int64, C.c_int64, C.c_void_p]
    lib.bp_rows_accumulate.restype = C.c_int
    lib.bp_rows_export_async.argtypes = [H, C.c_int64, C.c_int64, C.c_void_p]
    lib.bp_rows_export_async.restype = C.c_int
    lib.bp_rows_accumulate_async.argtypes = [H, C.c_int64, C.c_int64, C.c_void_p]
    lib.bp_rows_accumulate_async.restype = C.c_int
    lib.bp_stream.argtypes = [H, C.POINTER(C.c_void_p)]
    lib.bp_stream.restype = C.c_int
    lib.bp_column_sums.argtypes = [H, C.POINTER(C.c_uint64)]
    lib.bp_column_sums.restype = C.c_int
    lib.bp_get_timing.argtypes = [H, C.POINTER(bp_timing)]
    lib.bp_get_timing.restype = C.c_int
    lib.bp_algorithmic_bytes.argtypes = [H]
    lib.bp_algorithmic_bytes.restype = C.c_int64
    lib.bp_last_error.argtypes = [H]
    lib.bp_last_error.restype = C.c_char_p
    lib.bp_destroy.argtypes = [H]
    lib.bp_destroy.restype = None
    lib.bp_set_ref_seq8.argtypes = [H, C.c_int32, C.c_int32, C.c_void_p, C.c_int64]
    lib.bp_set_ref_seq8.restype = C.c_int
    lib.bp_render_text_bound.argtypes = [H, C.c_int32, C.c_int64]
    lib.bp_render_text_bound.restype = C.c_int64
    lib.bp_render_text.argtypes = [H, C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_void_p, C.c_size_t,
                                   C.POINTER(C.c_size_t)]
    lib.bp_render_text.restype = C.c_int
    lib.bp_render_fetch.argtypes = [H, C.c_void_p, C.c_size_t]
    lib.bp_render_fetch.restype = C.c_int
    lib.bp_render_bgzf_bound.argtypes = [H]
    lib.bp_render_bgzf_bound.restype = C.c_int64
    lib.bp_render_fetch_bgzf.argtypes = [H, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    lib.bp_render_fetch_bgzf.restype = C.c_int
    lib.bp_deflate_bgzf_bound.argtypes = [C.c_size_t]
    lib.bp_deflate_bgzf_bound.restype = C.c_size_t
    lib.bp_deflate_bgzf.argtypes = [H, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    lib.bp_deflate_bgzf.restype = C.c_int
    lib.bp_deflate_codes.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
    lib.bp_deflate_codes.restype = C.c_int
    lib.bp_per_read_build.argtypes = [H]
    lib.bp_per_read_build.restype = C.c_int
    lib.bp_per_read_total.argtypes = [H]
    lib.bp_per_read_total.restype = C.c_int64
    lib.bp_per_read_offsets.argtypes = [H, C.c_int64, C.c_int64, C.POINTER(C.c_uint64)]
    lib.bp_per_read_offsets.restype = C.c_int
    lib.bp_per_read_features.argtypes = [H, C.c_uint64, C.c_uint64, C.c_void_p]
    lib.bp_per_read_features.restype = C.c_int
    lib.bp_inflate_create.argtypes = [C.c_int32, C.POINTER(H)]
    lib.bp_inflate_create.restype = C.c_int
    lib.bp_inflate_destroy.argtypes = [H]
    lib.bp_inflate_destroy.restype = None
    lib.bp_inflate_blocks.argtypes = [H, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int64, C.c_void_p, C.c_size_t, C.c_int32,
                                      C.POINTER(C.c_int64)]
    lib.bp_inflate_blocks.restype = C.c_int
    lib.bp_inflate_device_output.argtypes = [H]
    lib.bp_inflate_device_output.restype = C.c_void_p
    lib.bp_inflate_stream.argtypes = [H, C.POINTER(C.c_void_p)]
    lib.bp_inflate_stream.restype = C.c_int
    lib.bp_inflate_timing.argtypes = [H, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.bp_inflate_timing.restype = C.c_int
    lib.bp_inflate_last_error.argtypes = [H]
    lib.bp_inflate_last_error.restype = C.c_char_p
    lib.bp_bam_scan.argtypes = [H, C.c_void_p, C.c_uint64, C.c_uint64, C.c_int32, C.POINTER(bp_bam_columns),
                                C.POINTER(C.c_int64), C.POINTER(C.c_uint64), C.POINTER(C.c_int32)]
    lib.bp_bam_scan.restype = C.c_int
    lib.bp_submit_bam_device.argtypes = [H, C.c_void_p, C.c_size_t, C.POINTER(bp_bam_columns), C.c_int64, C.c_int64]
    lib.bp_submit_bam_device.restype = C.c_int
    # include/bio_pileup_host.h
    E = [C.c_char_p, C.c_size_t]
    lib.bio_pileup_default_opts.argtypes = [C.POINTER(bio_pileup_opts)]
    lib.bio_pileup_default_opts.restype = None
    lib.bio_pileup_run.argtypes = [C.c_char_p] * 4 + [C.POINTER(bio_pileup_opts)] + E
    lib.bio_pileup_run.restype = C.c_int
    lib.bio_pileup_main.argtypes = [C.c_int, C.POINTER(C.c_char_p)]
    lib.bio_pileup_main.restype = C.c_int
    lib.bio_parse_region.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_int32),
                                     C.POINTER(C.c_int32)] + E
    lib.bio_parse_region.restype = C.c_int
    lib.bio_parse_cols.argtypes = [C.c_char_p, C.c_int32, C.POINTER(C.c_int32)] + E
    lib.bio_parse_cols.restype = C.c_int
    lib.bio_bed_load.argtypes = [C.c_char_p, C.c_int32, C.POINTER(C.c_char_p), C.POINTER(C.c_int32),
                                 C.POINTER(H)] + E
    lib.bio_bed_load.restype = C.c_int
    lib.bio_bed_endpoints.argtypes = [H, C.c_int32, C.POINTER(C.c_int32), C.c_int32]
    lib.bio_bed_endpoints.restype = C.c_int32
    lib.bio_bed_intersects.argtypes = [H, C.c_int32, C.c_int32, C.c_int32]
    lib.bio_bed_intersects.restype = C.c_int
    lib.bio_bed_free.argtypes = [H]
    lib.bio_bed_free.restype = None
    lib.bio_fasta_seq8.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_uint8), C.c_int64] + E
    lib.bio_fasta_seq8.restype = C.c_int64
    lib.bio_bam_scan.argtypes = [C.c_char_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                 C.POINTER(C.c_int64), C.POINTER(C.c_int32)] + E
    lib.bio_bam_scan.restype = C.c_int
    lib.bio_bam_scan_chunks.argtypes = [C.c_char_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_uint64),
                                        C.POINTER(C.c_uint64), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                        C.POINTER(C.c_int64)] + E
    lib.bio_bam_scan_chunks.restype = C.c_int
    lib.bio_bam_seek_scan.argtypes = [C.c_char_p, C.c_char_p, C.c_int32, C.c_int32, C.POINTER(C.c_int32),
                                      C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int64),
                                      C.POINTER(C.c_int64), C.c_char_p, C.c_size_t]
    lib.bio_bam_seek_scan.restype = C.c_int
    lib.bio_bam_write.argtypes = [C.c_char_p, C.c_int32, C.POINTER(C.c_char_p), C.POINTER(C.c_int32),
                                  C.c_void_p, C.c_int32, C.c_int32, C.c_char_p, C.c_size_t]
    lib.bio_bam_write.restype = C.c_int
    u32p, i32p = C.POINTER(C.c_uint32), C.POINTER(C.c_int32)
    lib.bio_pileup_write_rows.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int32, C.POINTER(C.c_char_p),
                                          C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_int64, i32p, i32p, u32p,
                                          C.POINTER(C.c_uint64), C.c_void_p] + E
    lib.bio_pileup_write_rows.restype = C.c_int
    lib.bio_pileup_row_fields.argtypes = [u32p, u32p, C.c_int32]
    lib.bio_pileup_row_fields.restype = C.c_uint32
    lib.bio_marshal_pileup_row.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, u32p, u32p, C.POINTER(C.c_void_p),
                                           C.c_void_p, C.c_size_t]
    lib.bio_marshal_pileup_row.restype = C.c_int64
    lib.bio_unmarshal_pileup_row.argtypes = [C.c_void_p, C.c_size_t, u32p, u32p, u32p, u32p, u32p, C.c_void_p,
                                             C.c_size_t]
    lib.bio_unmarshal_pileup_row.restype = C.c_int64
    lib.bio_marshal_basestrand.argtypes = [C.c_int64, i32p, i32p, u32p, C.c_void_p]
    lib.bio_marshal_basestrand.restype = None
    lib.bio_unmarshal_basestrand.argtypes = [C.c_int64, C.c_void_p, i32p, i32p, u32p]
    lib.bio_unmarshal_basestrand.restype = None
    lib.bio_basestrand_rio_trailer.argtypes = [C.c_int64, C.c_void_p]
    lib.bio_basestrand_rio_trailer.restype = None
    lib.bio_parse_basestrand_rio_trailer.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_int64)] + E
    lib.bio_parse_basestrand_rio_trailer.restype = C.c_int
    lib.bio_basestrand_piles_to_tsv.argtypes = [C.c_int64, i32p, i32p, u32p, C.c_int32, C.POINTER(C.c_char_p),
                                                C.c_char_p, C.c_void_p, C.c_size_t]
    lib.bio_basestrand_piles_to_tsv.restype = C.c_int64
    rowp = C.POINTER(bio_basestrand_tsv_row)
    lib.bio_read_basestrand_tsv.argtypes = [C.c_char_p, C.c_size_t, rowp, C.c_int64] + E
    lib.bio_read_basestrand_tsv.restype = C.c_int64
    lib.bio_read_single_strand_basestrand_tsv.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, rowp, C.c_int64] + E
    lib.bio_read_single_strand_basestrand_tsv.restype = C.c_int64
    lib.bio_chr_id.argtypes = [C.c_char_p, C.POINTER(C.c_int32)] + E
    lib.bio_chr_id.restype = C.c_int
    _lib = lib
    return lib
