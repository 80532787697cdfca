// deflate_api.h -- the device's BGZF writer as the two C-ABI modules use it (deflate_api.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <string>

namespace bp {

struct DeflateScratch;  // device and pinned buffers of one caller; grows, never shrinks
DeflateScratch* deflate_scratch_new();
void deflate_scratch_free(DeflateScratch* sc);
// the same through a pool per device (for callers that come and go, like a job's handle); the device is current
DeflateScratch* deflate_scratch_acquire(int device);
void deflate_scratch_release(int device, DeflateScratch* sc);

// bytes that n bytes of text take at most as BGZF members (every member can fall back to stored)
size_t deflate_bgzf_bound(size_t n);

// d_text: n bytes on the current device with at least 8 readable bytes behind them.  On BP_OK
// *d_out points at *total bytes of complete BGZF members (no EOF member) in the scratch's device
// memory, valid until the next call; the stream is idle.
int deflate_bgzf_device(DeflateScratch* sc, const uint8_t* d_text, size_t n, hipStream_t stream,
                        const uint8_t** d_out, size_t* total, std::string* err);

void crc_tables(uint32_t* t /*[1280]*/);

}  // namespace bp
