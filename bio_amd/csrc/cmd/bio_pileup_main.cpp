// bio-pileup executable: the work-alike of cmd/bio-pileup/main.go over libbiopileup.so.
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#include "../host/bio_host.h"

static double epoch_s() {
  struct timespec t;
  clock_gettime(CLOCK_REALTIME, &t);
  return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

int main(int argc, char** argv) {
  const double t_in = epoch_s();
  const int rc = bio::BioPileupMain(argc, argv);
  if (const char* e = getenv("BIO_PILEUP_STATS"); e && *e && *e != '0')
    fprintf(stderr, "bio-pileup main: entered at %.3f, left at %.3f (epoch seconds)\n", t_in, epoch_s());
  // (Leaving with _exit here, to skip the piecewise teardown of pinned buffers, cached device
  // buffers and the HIP runtime, was tried: no faster - what takes 0.2 s after main is the kernel
  // releasing the process's device memory either way - and it robs tools that flush at exit,
  // rocprofv3 among them, of their output.)
  return rc;
}
