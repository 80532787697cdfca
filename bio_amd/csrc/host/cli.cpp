// cli.cpp -- bio-pileup command line (reference: cmd/bio-pileup/main.go:33-98).  Flag syntax
// follows Go's flag package: -name value, -name=value, --name; booleans take no value
// unless written -name=false; parsing stops at the first non-flag argument or at "--".
#include <cerrno>
#include <cstdlib>
#include <cstring>

#include "bio_host.h"

namespace bio {

namespace {

struct Flag {
  const char* name;
  enum { kString, kInt, kBool } type;
  void* dst;
  const char* usage;
};

bool ParseGoInt(const std::string& s, int* out) {
  // strconv.ParseInt(s, 0, 64): optional sign, 0x / 0o / 0b / leading-0 octal prefixes
  if (s.empty()) return false;
  char* end = nullptr;
  errno = 0;
  size_t i = (s[0] == '+' || s[0] == '-') ? 1 : 0;
  long long v;
  if (s.size() > i + 2 && s[i] == '0' && (s[i + 1] == 'b' || s[i + 1] == 'B')) {
    v = strtoll(s.c_str() + i + 2, &end, 2);
    if (s[0] == '-') v = -v;
  } else if (s.size() > i + 2 && s[i] == '0' && (s[i + 1] == 'o' || s[i + 1] == 'O')) {
    v = strtoll(s.c_str() + i + 2, &end, 8);
    if (s[0] == '-') v = -v;
  } else {
    v = strtoll(s.c_str(), &end, 0);
  }
  if (errno || !end || *end || v > 2147483647ll || v < -2147483648ll) return false;
  *out = (int)v;
  return true;
}

bool ParseGoBool(const std::string& s, bool* out) {
  static const char* t[] = {"1", "t", "T", "TRUE", "true", "True"};
  static const char* f[] = {"0", "f", "F", "FALSE", "false", "False"};
  for (const char* x : t) if (s == x) { *out = true; return true; }
  for (const char* x : f) if (s == x) { *out = false; return true; }
  return false;
}

}  // namespace

int BioPileupMain(int argc, char** argv) {
  Opts o;
  std::string format = "tsv", out_prefix = "bio-pileup", devices;
  const Flag flags[] = {
      {"bed", Flag::kString, &o.BedPath, "Input BED path; this xor -region required"},
      {"region", Flag::kString, &o.Region,
       "Restrict pileup computation to the specified region. Format as <contig ID>:<1-based first pos>-<last "
       "pos>, <contig ID>:<1-based pos>, or just <contig ID>; this xor -bed required"},
      {"index", Flag::kString, &o.BamIndexPath, "Input BAM index path (.bai or .gbai; default <bam>.bai, then <bam>.gbai, if present)"},
      {"clip", Flag::kInt, &o.Clip, "Number of bases on end of each read to treat as minimum-quality"},
      {"cols", Flag::kString, &o.Cols,
       "Output TSV column sets. #CHROM/POS/REF(/ALT) are always present. Currently supported optional sets are "
       "'dpref', 'dpalt', 'enddists', 'quals', 'fraglens', 'strands', 'highq', and 'lowq'; default is "
       "\"dpref,highq,lowq\""},
      {"flag-exclude", Flag::kInt, &o.FlagExclude, "Reads with a FLAG bit intersecting this value are skipped"},
      {"format", Flag::kString, &format,
       "Output format; 'basestrand-tsv', 'basestrand-tsv-bgz', 'tsv', and 'tsv-bgz' supported"},
      {"mapq", Flag::kInt, &o.Mapq, "Reads with MAPQ below this level are skipped"},
      {"max-read-len", Flag::kInt, &o.MaxReadLen, "Upper bound on individual read length"},
      {"max-read-span", Flag::kInt, &o.MaxReadSpan, "Upper bound on size of reference-genome region a read maps to"},
      {"min-bag-depth", Flag::kInt, &o.MinBagDepth, "Lower bound on bag depth (DS aux tag value)"},
      {"min-base-qual", Flag::kInt, &o.MinBaseQual, "Lower bound on base quality in a single read"},
      {"out", Flag::kString, &out_prefix, "Output path prefix"},
      {"parallelism", Flag::kInt, &o.Parallelism, "Host threads for inflate, text rendering and output compression (0 = CPUs, at most 16); jobs run on the GPU"},
      {"per-strand", Flag::kBool, &o.PerStrand, "Generate two pairs of output files, one for each strand"},
      {"remove-sq", Flag::kBool, &o.RemoveSq, "Remove sequencing duplicates (no DL aux tag with value > 1)"},
      {"stitch", Flag::kBool, &o.Stitch, "Stitch read-pairs"},
      {"temp-dir", Flag::kString, &o.TempDir, "Accepted for compatibility; no temporary files are written"},
      {"device", Flag::kInt, &o.Device, "HIP device ordinal"},
      {"devices", Flag::kString, &devices, "Spread the jobs over this many devices (a number, mapped round-robin onto the GPUs present, or 'all'); default: only -device"},
  };
  auto usage = [&]() {
    printf("Usage: %s [OPTIONS] {b,p}ampath fapath\n", argv[0]);
    printf("Other options:\n");
    for (const Flag& f : flags) printf("  -%s\n    \t%s\n", f.name, f.usage);
  };

  int i = 1;
  for (; i < argc; i++) {
    std::string a = argv[i];
    if (a.size() < 2 || a[0] != '-') break;
    size_t dashes = 1;
    if (a[1] == '-') {
      dashes = 2;
      if (a.size() == 2) { i++; break; }  // "--" terminates the flags
    }
    std::string name = a.substr(dashes), value;
    if (name.empty() || name[0] == '-' || name[0] == '=') {
      fprintf(stderr, "bad flag syntax: %s\n", a.c_str());
      usage();
      return 2;
    }
    bool has_value = false;
    size_t eq = name.find('=', 1);
    if (eq != std::string::npos) {
      value = name.substr(eq + 1);
      name = name.substr(0, eq);
      has_value = true;
    }
    if (name == "help" || name == "h") { usage(); return 0; }
    const Flag* f = nullptr;
    for (const Flag& c : flags)
      if (name == c.name) f = &c;
    if (!f) {
      fprintf(stderr, "flag provided but not defined: -%s\n", name.c_str());
      usage();
      return 2;
    }
    if (f->type == Flag::kBool) {
      bool v = true;
      if (has_value && !ParseGoBool(value, &v)) {
        fprintf(stderr, "invalid boolean value \"%s\" for -%s: parse error\n", value.c_str(), name.c_str());
        usage();
        return 2;
      }
      *(bool*)f->dst = v;
      continue;
    }
    if (!has_value) {
      if (i + 1 >= argc) {
        fprintf(stderr, "flag needs an argument: -%s\n", name.c_str());
        usage();
        return 2;
      }
      value = argv[++i];
    }
    if (f->type == Flag::kInt) {
      if (!ParseGoInt(value, (int*)f->dst)) {
        fprintf(stderr, "invalid value \"%s\" for flag -%s: parse error\n", value.c_str(), name.c_str());
        usage();
        return 2;
      }
    } else {
      *(std::string*)f->dst = value;
    }
  }
  const int n_pos = argc - i;
  if (n_pos != 2) {
    std::string joined;
    for (int k = i; k < argc; k++) joined += (k > i ? " " : "") + std::string(argv[k]);
    if (n_pos < 2)
      fprintf(stderr,
              "Missing positional arguments ({b,p}ampath and fapath required); please check flag syntax: '%s'\n",
              joined.c_str());
    else
      fprintf(stderr,
              "Too many positional arguments (only {b,p}ampath and fapath expected); please check flag syntax: "
              "'%s'\n",
              joined.c_str());
    return 1;
  }
  if (!devices.empty()) {
    if (devices == "all") {
      o.Devices = -1;
    } else {
      char* end = nullptr;
      const long n = strtol(devices.c_str(), &end, 10);
      if (*end || n < 1 || n > 1024) {
        fprintf(stderr, "invalid value \"%s\" for flag -devices: a number from 1 to 1024, or all\n", devices.c_str());
        return 2;
      }
      o.Devices = (int)n;
    }
  }
  std::string err = Pileup(argv[i], argv[i + 1], format, out_prefix, o);
  if (!err.empty()) {
    fprintf(stderr, "panic: %s\n", err.c_str());  // log.Panicf (main.go:94-96)
    return 2;
  }
  return 0;
}

}  // namespace bio
