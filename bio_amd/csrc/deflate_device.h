// deflate_device.h -- shared between deflate_kernels.hip and deflate_api.hip.
#pragma once
#include <stdint.h>

namespace bp {

constexpr int kDflMember = 0xff00;               // text bytes per BGZF member (what bgzip and hts' bgzf.Writer use)
constexpr int kDflSegs = 32;                     // a member is tokenised in this many segments, a wave each
constexpr int kDflSeg = kDflMember / kDflSegs;   // 2040 bytes
constexpr int kDflSlot = 65536 + 512;            // bytes a member has to itself before the members are packed (a member is given
                                                 // up for the stored form only after the step that made it too long: 384 bytes more)
constexpr int kDflHist = 320;                    // [0, 286) literal/length symbols, [288, 318) distance symbols
constexpr int kDflHdrWords = 128;                // the dynamic block header, at most 4096 bits

// The Huffman codes of one run of members (built on the host from the run's symbol counts) and
// the block header that announces them.  A code is stored bit-reversed (DEFLATE packs Huffman
// codes most significant bit first into a stream that is otherwise filled from bit 0 up), with
// its length in bits 16-19.
struct DeflateCodes {
  uint32_t ll[288];
  uint32_t dist[32];
  uint32_t hdr_bits;
  uint32_t hdr[kDflHdrWords];
};

// A token of the tokeniser: a literal byte (bits 16-24 zero), or a match of `len` = bits 16-24
// (4..258) bytes at distance (bits 0-14) + 1.
}  // namespace bp
