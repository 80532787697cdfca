// bio-pileup executable: the work-alike of cmd/bio-pileup/main.go over libbiopileup.so.
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include <unistd.h>

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
  // Every output file has been closed (and its errors reported) by now.  What a normal return
  // would still do is tear the process down piece by piece - unpin and free gigabytes of staging
  // buffers, release the device buffers kept for reuse, unload the HIP runtime: 0.1-0.2 s of a
  // 1.8 s run that the kernel does at once when the process simply ends (as a Go program's
  // os.Exit does, cmd/bio-pileup/main.go).
  fflush(stdout);
  fflush(stderr);
  _exit(rc);
}
