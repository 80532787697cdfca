// pack_device.h -- what the host tells the record-pack kernel about one record of an inflated
// BAM stream (bp_submit_bam): where its bytes lie in the shipped run, where its payload goes,
// and the few values the host pass has worked out anyway.
#pragma once
#include <stdint.h>

struct PackItem {
  uint32_t src;     // byte offset of the record's block_size field in the shipped bytes
  uint32_t blob;    // ReadCore.blob: the base part's place, in 8-byte units
  uint16_t qdelta;  // ReadCore.qdelta
  uint16_t span;    // ReadCore.span (reference bases the CIGAR consumes, clamped)
  uint8_t aux;      // ReadCore.aux
  uint8_t l_name;   // l_read_name (with the NUL)
  uint16_t pad;
};
static_assert(sizeof(PackItem) == 16, "PackItem is shipped as 16-byte entries");
