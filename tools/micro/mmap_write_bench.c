// tools/micro/mmap_write_bench.c -- the other ways of getting gigabytes into ONE file on the GPU
// box's local disk, beside pwrite_bench.c's buffered pwrite (which the inode lock holds to one
// thread's rate): N threads copying into a shared mapping of the file (page faults take no inode
// lock), the same with MADV_POPULATE_WRITE per chunk first, and O_DIRECT pwrite of aligned chunks.
// usage: mmap_write_bench <dir> [GB=4]
#define _GNU_SOURCE
#include <fcntl.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
static char* buf;
static char* map;
static size_t chunk = 32u << 20, total;
static int fd, nt, mode;  // 0 mmap+memcpy, 1 populate+memcpy, 2 O_DIRECT pwrite, 3 buffered pwrite
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
struct arg { int t; };
// Each thread takes whole chunks round-robin (what N closer threads with a job each would do).
static void* one(void* p) {
  struct arg* a = p;
  for (size_t base = (size_t)a->t * chunk; base < total; base += (size_t)nt * chunk) {
    size_t n = total - base < chunk ? total - base : chunk;
    if (mode == 0 || mode == 1) {
      if (mode == 1 && madvise(map + base, n, MADV_POPULATE_WRITE) != 0) perror("madvise");
      memcpy(map + base, buf, n);
    } else if (pwrite(fd, buf, n, (off_t)base) != (ssize_t)n) {
      perror("pwrite");
    }
  }
  return 0;
}
int main(int argc, char** argv) {
  const char* dir = argc > 1 ? argv[1] : "/tmp";
  total = (size_t)((argc > 2 ? atof(argv[2]) : 4.0) * (1u << 30));
  total &= ~(chunk - 1);
  if (posix_memalign((void**)&buf, 1 << 21, chunk) != 0) return 1;
  memset(buf, 'x', chunk);
  const char* names[] = {"mmap + memcpy        ", "mmap + populate + cpy", "O_DIRECT pwrite      ", "buffered pwrite      "};
  int counts[] = {1, 2, 4, 8, 16};
  for (mode = 0; mode < 4; mode++)
    for (int ci = 0; ci < 5; ci++) {
      nt = counts[ci];
      char path[512];
      snprintf(path, sizeof path, "%s/mwb_%d.tmp", dir, getpid());
      fd = open(path, O_CREAT | O_TRUNC | O_RDWR | (mode == 2 ? O_DIRECT : 0), 0644);
      if (fd < 0) { perror("open"); continue; }
      double t0 = now();
      if (mode < 2) {
        if (ftruncate(fd, (off_t)total) != 0) perror("ftruncate");
        map = mmap(0, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        if (map == MAP_FAILED) { perror("mmap"); close(fd); unlink(path); continue; }
      }
      pthread_t th[16];
      struct arg a[16];
      for (int t = 0; t < nt; t++) { a[t].t = t; pthread_create(&th[t], 0, one, &a[t]); }
      for (int t = 0; t < nt; t++) pthread_join(th[t], 0);
      double t1 = now();
      if (mode < 2) munmap(map, total);
      close(fd);
      double t2 = now();
      unlink(path);
      printf("%s %2d threads: %.2f GB/s (+ %.3f s to unmap and close)\n", names[mode], nt, total / (t1 - t0) / 1e9, t2 - t1);
      fflush(stdout);
    }
  return 0;
}
