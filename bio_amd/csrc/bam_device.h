// bam_device.h -- shared by the device record scan (bam_kernels.hip) and its host side.
#pragma once
#include <stdint.h>

namespace bp {

constexpr uint32_t kStretch = 65536;       // bytes of inflated stream one wave walks
constexpr int kMaxPerStretch = 1824;       // records that can start in a stretch (a record is >= 36 bytes)

enum : uint32_t {
  WALK_STOPPED = 1,   // the walk met a record cut off by the end of the stream (or an invalid block size)
  WALK_NO_START = 2,  // no plausible record start behind the cut
  WALK_OVERFLOW = 4,
  WALK_SKIPPED = 8,   // (set by the verifier) the chain passed over this stretch, or stopped before it
  WALK_REWALKED = 16, // (set by the verifier) the guess was wrong: walked again from the right place
};

struct WalkStretch {
  uint32_t start;  // where this stretch's walk began (0xffffffff: nowhere)
  uint32_t exit;   // the first record start it did not take
  uint32_t n;      // records it took
  uint32_t flags;
  uint32_t base;   // (verifier) index of its first record among the stream's
};

// Device pointers, one entry per record.
struct BamColumns {
  uint32_t* off;
  int32_t *ref, *pos, *mref, *mpos;
  uint16_t* flags;
  uint8_t *mapq, *lname;
  uint32_t *ncig, *lseq, *bytes;
  int32_t* span;
  uint8_t* facts;  // bit 0 one M op, bit 1 an op outside MIDNSH, bit 2 the fields do not fit the block
};

}  // namespace bp
