// inflate_device.h -- what the host tells the inflate kernels about one BGZF block, and what they
// say back (inflate_kernels.hip, inflate_api.hip).
#pragma once
#include <stdint.h>

namespace bp {

struct InflateBlock {
  uint32_t in_off;   // the block's deflate data in the compressed bytes shipped
  uint32_t in_len;
  uint32_t out_off;  // where its ISIZE bytes go in the output
  uint32_t out_len;
  uint32_t crc32;    // of those bytes, from the block's trailer
};
static_assert(sizeof(InflateBlock) == 20, "InflateBlock mirrors bp_bgzf_block");

enum InflateStatus : uint32_t {
  INF_OK = 0,
  INF_BAD_BLOCK_TYPE = 1,    // BTYPE 3
  INF_BAD_STORED_LEN = 2,    // LEN != ~NLEN
  INF_BAD_CODE_LENGTHS = 3,  // over-subscribed or malformed code-length data, no end-of-block code
  INF_BAD_CODE = 4,          // a bit pattern no symbol owns, or symbols 286/287, 30/31
  INF_DISTANCE_TOO_FAR = 5,  // a match reaching in front of the block's first byte
  INF_OUTPUT_OVERFLOW = 6,   // more bytes than ISIZE
  INF_INPUT_OVERRUN = 7,     // the stream runs past the block's compressed bytes
  INF_OUTPUT_SHORT = 8,      // fewer bytes than ISIZE
  INF_BAD_CRC = 9,
};

}  // namespace bp
