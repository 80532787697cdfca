// tools/micro/pwrite_bench.c -- how fast can gigabytes of text get into a file on the GPU box's
// local disk?  N threads pwrite disjoint slices of 32 MiB chunks of one file (what output.cpp's
// Sink does), or of one file per thread; buffered.  usage: pwrite_bench <dir> [GB=4]
#define _GNU_SOURCE
#include <fcntl.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
static char* buf;
static size_t chunk = 32u << 20, total;
static int fd, nt;
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
struct arg { int t; size_t base; };
static void* one(void* p) {
  struct arg* a = p;
  size_t lo = chunk * a->t / nt, hi = chunk * (a->t + 1) / nt;
  if (pwrite(fd, buf + lo, hi - lo, (off_t)(a->base + lo)) != (ssize_t)(hi - lo)) perror("pwrite");
  return 0;
}
int main(int argc, char** argv) {
  const char* dir = argc > 1 ? argv[1] : "/tmp";
  total = (size_t)((argc > 2 ? atof(argv[2]) : 4.0) * (1u << 30));
  buf = malloc(chunk);
  memset(buf, 'x', chunk);
  int counts[] = {1, 2, 4, 8, 16};
  for (int prealloc = 0; prealloc < 2; prealloc++)
    for (int ci = 0; ci < 5; ci++) {
      nt = counts[ci];
      char path[512];
      snprintf(path, sizeof path, "%s/pwb_%d.tmp", dir, getpid());
      fd = open(path, O_CREAT | O_TRUNC | O_WRONLY, 0644);
      if (prealloc && ftruncate(fd, (off_t)total) != 0) perror("ftruncate");
      double t0 = now();
      for (size_t base = 0; base < total; base += chunk) {
        pthread_t th[16];
        struct arg a[16];
        for (int t = 0; t < nt; t++) { a[t].t = t; a[t].base = base; pthread_create(&th[t], 0, one, &a[t]); }
        for (int t = 0; t < nt; t++) pthread_join(th[t], 0);
      }
      double t1 = now();
      close(fd);
      unlink(path);
      printf("%s %2d threads: %.2f GB/s\n", prealloc ? "ftruncate first," : "extending,     ", nt, total / (t1 - t0) / 1e9);
      fflush(stdout);
    }
  return 0;
}
