// pileup.cpp -- bio::Pileup, the host driver around the device hot path (reference:
// snp.Pileup pileup/snp/pileup.go:970-1130 and pileupSNPMain :812-968).
//
// Job structure.  The reference cuts the genome into 100 kb shards, gives each of
// `parallelism` goroutines a run of shards with a maxReadSpan halo and the BED Subset of
// exactly its range (pileup.go:862-909), and concatenates the per-job rows.  Counts are sums
// of commutative per-base updates, so a cut into position ranges with enough padding gives
// the rows of the single-job run (reference -parallelism 1); here a job is a range of one
// contig of at most JobSpan() positions, sized for the pipeline rather than for CPU cores (a
// job's text can only be rendered and written once it is closed: short jobs let the output start
// early and run beside the reading), and jobs run one after another on the device.  In the reference a pair whose
// two reads fall on either side of a job boundary is stitched in neither job, so its output
// depends mildly on -parallelism when flags are malformed; the padding used here (see step 6)
// removes that dependence.  `-region` keeps the reference's single unpadded shard
// (pileup.go:1091-1096): reads that start before the region start are never seen.
#include <unistd.h>

#include <algorithm>
#include <condition_variable>
#include <future>
#include <mutex>
#include <thread>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <deque>

#include "../par.h"
#include "bio_host.h"

namespace bio {

namespace {

constexpr int64_t kRowChunk = 1 << 20;       // rows per bp_rows copy
// With an index, stretches of the BAM that start more than this many positions before the
// next padded BED interval are skipped (the reference's position shards are this long, and
// it never opens a shard that misses the BED: pileup.go:1073-1084, pileup_support.go:71-85).
constexpr int64_t kSeekGap = 100000;

// Walks the padded BED intervals [start - pad, end + pad_end) in step with the sorted records.
class NeedCursor {
 public:
  NeedCursor(const BEDUnion& bed, int64_t pad, int64_t pad_end) : bed_(bed), pad_(pad), pad_end_(pad_end) {}
  // The first padded interval that ends after (ref, pos): false when there is none; else
  // (*nref, *nstart) is its start, which may lie at or before pos.
  bool Next(int ref, int64_t pos, int* nref, int64_t* nstart) {
    if (ref != ref_) { ref_ = ref; k_ = 0; }
    for (size_t r = (size_t)ref; r < bed_.id_map.size(); r++) {
      const std::vector<PosType>& e = bed_.id_map[r];
      size_t k = 0;
      if ((int)r == ref_) {
        while (k_ + 1 < e.size() && (int64_t)e[k_ + 1] + pad_end_ <= pos) k_ += 2;
        k = k_;
      }
      if (k + 1 < e.size()) {
        *nref = (int)r;
        *nstart = std::max<int64_t>(0, (int64_t)e[k] - pad_);
        return true;
      }
    }
    return false;
  }

 private:
  const BEDUnion& bed_;
  int64_t pad_, pad_end_;
  int ref_ = -1;
  size_t k_ = 0;
};

int64_t JobSpan() {
  // Positions per job.  BIO_PILEUP_JOB_SPAN exists for tests of the halo logic.
  const char* e = getenv("BIO_PILEUP_JOB_SPAN");
  if (e && *e) {
    long long v = atoll(e);
    if (v > 0) return v;
  }
  // (chr1 30x, whole run: 2^25 1.58-1.62 s, 2^24 1.58-1.64, 2^23 1.52-1.61, 2^22 1.47-1.57,
  // 2^21 1.52-1.58, 2^20 1.60-1.62: the output is the long pole and starts when the first job closes)
  return 1ll << 22;
}

// BIO_PILEUP_STATS=1 prints one line of counters and stage times to stderr at the end.
struct Stats {
  int64_t records_read = 0, records_kept = 0, seeks = 0, jobs = 0;
  double submit_s = 0, close_s = 0, close_wait_s = 0, total_s = 0;
};
double Now() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct Job {
  int ref;
  PosType start, end;  // positions [start, end) of contig `ref`
  int64_t seq = 0;     // its place in the output order
};

// What closing a job needs besides its handles.
struct JobRows {
  int64_t row_lo = 0, row_hi = 0;  // the handle's rows [row_lo, row_hi) lie in the job's own range
  bool device_text = false;        // the device renders the lines (bp_render_text)
};

struct Stream {  // one output stream = one strand pass (pileup.go:1118-1128)
  int strand_req = BP_STRAND_NONE;
  std::unique_ptr<RowWriter> writer;
  // -per-strand in ONE pass: a split-strand job (strand_req none, counts by strand) feeds both
  // files, `writer` the forward strand's and `writer_rev` the reverse strand's; each takes its
  // strand's share of the rows (Depth = the strand's counts summed: exact while no base fails a
  // quality rule, so only without -min-base-qual; and without -stitch, where a pair of reads on
  // different strands would be stitched here and piled up alone in each of the reference's passes)
  std::unique_ptr<RowWriter> writer_rev;
  bp_handle* h = nullptr;
};

// One strand's share of split-strand rows, as a -per-strand pass has them (format tsv: every count
// in slot 0): n rows of BP_ROW_U32 in place.
void StrandShare(uint32_t* vals, int64_t n, int s) {
  for (int64_t i = 0; i < n; i++) {
    uint32_t* r = vals + i * BP_ROW_U32;
    uint32_t d = 0;
    for (int b = 0; b < 5; b++) {
      const uint32_t c = r[1 + 2 * b + s];
      r[1 + 2 * b] = c;
      r[2 + 2 * b] = 0;
      d += c;
    }
    r[0] = d;
  }
}

std::string HandleErr(bp_handle* h) { return bp_last_error(h); }

// The records of one chunk of the inflated stream that one job wants: the bytes stay where the
// reader inflated them (pinned memory) and go to the device from there (bp_submit_bam).
struct Piece {
  std::shared_ptr<const void> hold;
  const uint8_t* data = nullptr;
  size_t bytes = 0;
  std::vector<uint32_t> off;
  // a run of records of a DEVICE chunk (BamReader::StartResident): entries [first, first + count) of
  // the scan's columns; the bytes are on the device and stay there (bp_submit_bam_device)
  const uint8_t* dev = nullptr;
  const bp_bam_columns* cols = nullptr;
  int64_t first = 0, count = 0;
  bool empty() const { return cols ? count == 0 : off.empty(); }
};

class Driver {
 public:
  Driver(const Opts& o, const BamHeader& hd, const BEDUnion& bed, int fmt_ignore_strand)
      : opts_(o), hd_(hd), bed_(bed), ignore_strand_(fmt_ignore_strand) {}

  std::vector<Stream> streams;
  bool per_read = false;  // a per-read column set is requested (pCtx.perReadNeeded)
  // basestrand-tsv without per-read columns: the rows become text on the device
  bool device_text = false;
  bool text_bgzf = false;  // ... and compresses them into BGZF members (-bgz formats, bp_render_fetch_bgzf)
  bool text_tsv = false;   // ... of the tsv format (two files) rather than basestrand-tsv
  int32_t text_cols = 0;   // its -cols bits
  const std::vector<const std::string*>* ref_seqs = nullptr;
  JobRows open_rows;  // of the open job

  void SetPadding(int64_t bed_pad) { bed_pad_ = bed_pad; }

  // -devices: job k runs on logical device k mod N, logical devices mapped round-robin onto the
  // physical ones - the reference's traverse.Each(parallelism, ...) (pileup.go:862-932) with
  // devices for goroutines; jobs are independent (halo-padded), nothing crosses between devices.
  std::vector<int> devices;  // physical ordinal of every logical device; empty: opts.Device only
  int DeviceFor(int64_t seq) const {
    return devices.empty() ? opts_.Device : devices[(size_t)(seq % (int64_t)devices.size())];
  }
  size_t n_closers() const { return devices.empty() ? 1 : devices.size(); }
  int64_t est_records_ = 0, est_bytes_ = 0;  // what the job before the open one held (OpenJob)

  std::string OpenJob(const Job& j) {
    BEDUnion part = bed_.Subset(j.ref, (PosType)std::max<int64_t>(0, (int64_t)j.start - bed_pad_), j.ref,
                                (PosType)std::min<int64_t>(kPosTypeMax, (int64_t)j.end + bed_pad_));
    std::vector<const char*> names;
    for (const auto& s : hd_.ref_names) names.push_back(s.c_str());
    for (Stream& s : streams) {
      bp_opts bo;
      memset(&bo, 0, sizeof(bo));
      bo.clip = opts_.Clip;
      bo.flag_exclude = opts_.FlagExclude;
      bo.mapq = opts_.Mapq;
      bo.max_read_len = opts_.MaxReadLen;
      bo.max_read_span = opts_.MaxReadSpan;
      bo.min_base_qual = opts_.MinBaseQual;
      bo.stitch = opts_.Stitch;
      bo.ignore_strand = s.writer_rev ? 0 : ignore_strand_;
      bo.strand_req = s.strand_req;
      bo.device = DeviceFor(j.seq);
      if (bp_create(&bo, (int32_t)hd_.ref_lens.size(), hd_.ref_lens.data(), names.data(), &s.h) != BP_OK)
        return HandleErr(nullptr);
      const std::vector<PosType>& e = part.id_map[(size_t)j.ref];
      if (bp_set_bed(s.h, j.ref, e.data(), (int32_t)e.size()) != BP_OK) return HandleErr(s.h);
      // a run's jobs are of a size: room for what the last one held, so that the job's buffers
      // are taken whole (from the library's cache of the last job's) instead of grown step by
      // step, a copy and a synchronisation each (stitched chr1 30x: half of the submitter's time)
      if (est_records_ > 0 && bp_reserve(s.h, est_records_ + est_records_ / 8 + 1024, est_bytes_) != BP_OK)
        return HandleErr(s.h);
    }
    // rows of the padding belong to the neighbouring jobs: BED positions of `part` before the
    // job's start and before its end
    const std::vector<PosType>& e = part.id_map[(size_t)j.ref];
    auto before = [&](int64_t x) {
      int64_t n = 0;
      for (size_t k = 0; k + 1 < e.size(); k += 2)
        n += std::max<int64_t>(0, std::min<int64_t>(e[k + 1], x) - e[k]);
      return n;
    };
    open_rows = JobRows();
    open_rows.row_lo = before(j.start);
    open_rows.row_hi = before(j.end);
    // The device writes the lines itself when every position of the job has a reference base
    // (otherwise the host path reports the missing one, output.go:158).
    const std::string* seq = ref_seqs && (size_t)j.ref < ref_seqs->size() ? (*ref_seqs)[(size_t)j.ref] : nullptr;
    if (device_text && seq && e.size() >= 2 && (size_t)e.back() <= seq->size()) {
      for (Stream& s : streams)
        if (bp_set_ref_seq8(s.h, j.ref, e.front(), (const uint8_t*)seq->data() + e.front(), (int64_t)e.back() - e.front()) != BP_OK)
          return HandleErr(s.h);
      open_rows.device_text = true;
    }
    return "";
  }

  std::string SubmitPiece(size_t stream, const Piece& p) {
    if (p.empty()) return "";
    Stream& s = streams[stream];
    if (p.cols) {
      if (bp_submit_bam_device(s.h, p.dev, p.bytes, p.cols, p.first, p.count) != BP_OK) return HandleErr(s.h);
      return "";
    }
    if (bp_submit_bam(s.h, p.data, p.bytes, p.off.data(), (int64_t)p.off.size()) != BP_OK) return HandleErr(s.h);
    return "";
  }

  // The handles of the open job leave the streams (the next job opens its own) and are closed
  // by its device's closer thread.
  std::vector<bp_handle*> TakeHandles() {
    std::vector<bp_handle*> hs;
    if (!streams.empty() && streams[0].h) {
      int64_t n = 0, b = 0;
      if (bp_job_size(streams[0].h, &n, &b) == BP_OK) { est_records_ = n; est_bytes_ = b; }
    }
    for (Stream& s : streams) {
      hs.push_back(s.h);
      s.h = nullptr;
    }
    return hs;
  }

  // finishRef + the output conversion for one job (pileup.go:934-966): rows to the writers, in
  // job order.  Frees the handles.
  // First half, concurrent across devices: the device computation of the job.
  std::string ComputeJob(std::vector<bp_handle*>& hs) {
    for (bp_handle* h : hs) {
      if (bp_finish(h) != BP_OK) return HandleErr(h);
      if (per_read && bp_per_read_build(h) != BP_OK) return HandleErr(h);
    }
    return "";
  }
  // Second half, in job order: the rows (or the device's text) to the writers.
  std::string OutputJob(const Job& j, std::vector<bp_handle*>& hs, const JobRows& jr) {
    return jr.device_text ? FinishJobText(hs, jr) : FinishJobRows(j, hs);
  }
  static void FreeHandles(std::vector<bp_handle*>& hs) {
    for (bp_handle*& h : hs) {
      if (h) bp_destroy(h);
      h = nullptr;
    }
  }

  // One chunk of a job's rows on the host (with its per-read lists when those are asked for).
  struct Chunk {
    std::vector<int32_t> ref, pos;
    std::vector<uint32_t> vals;
    std::vector<uint64_t> off;
    std::vector<bp_per_read> feats;
    int64_t lo = 0, hi = 0;  // rows [lo, hi) of the chunk lie in the job's own range
    std::string err;
  };
  void Fetch(bp_handle* h, const Job& j, int64_t first, int64_t m, Chunk* c) const {
    c->err.clear();
    c->ref.resize((size_t)m);
    c->pos.resize((size_t)m);
    c->vals.resize((size_t)m * BP_ROW_U32);
    if (bp_rows(h, first, m, c->ref.data(), c->pos.data(), c->vals.data()) != BP_OK) { c->err = HandleErr(h); return; }
    // rows of the padding belong to the neighbouring jobs
    c->lo = std::lower_bound(c->pos.begin(), c->pos.begin() + m, j.start) - c->pos.begin();
    c->hi = std::lower_bound(c->pos.begin(), c->pos.begin() + m, j.end) - c->pos.begin();
    if (!per_read || c->lo >= c->hi) return;
    c->off.resize((size_t)(4 * (c->hi - c->lo) + 1));
    if (bp_per_read_offsets(h, first + c->lo, c->hi - c->lo, c->off.data()) != BP_OK) { c->err = HandleErr(h); return; }
    c->feats.resize((size_t)(c->off.back() - c->off.front()));
    if (bp_per_read_features(h, c->off.front(), c->off.back() - c->off.front(), c->feats.data()) != BP_OK)
      c->err = HandleErr(h);
  }

  // The lines come from the device, a chunk of rows at a time; the sink writes one chunk while
  // the next is rendered and copied.
  std::string FinishJobText(std::vector<bp_handle*>& hs, const JobRows& jr) {
    for (size_t si = 0; si < streams.size(); si++) {
      bp_handle* h = hs[si];
      for (int64_t first = jr.row_lo; first < jr.row_hi; first += kRowChunk) {
        const int64_t m = std::min<int64_t>(kRowChunk, jr.row_hi - first);
        // basestrand-tsv: one file; tsv: the .ref.tsv and the .alt.tsv lines of the same rows
        // (one-pass -per-strand: of each strand's share of them)
        const bool split = streams[si].writer_rev != nullptr;
        for (int strand = 0; strand < (split ? 2 : 1); strand++) {
          RowWriter* w = strand ? streams[si].writer_rev.get() : streams[si].writer.get();
          const int32_t sel = split ? (strand ? BP_TEXT_STRAND_REV : BP_TEXT_STRAND_FWD) : 0;
          for (int which = 0; which < (text_tsv ? 2 : 1); which++) {
            const int32_t kind = text_tsv ? (which == 0 ? BP_TEXT_TSV_REF : BP_TEXT_TSV_ALT) : BP_TEXT_BASESTRAND;
            size_t len = 0;
            if (bp_render_text(h, kind | sel, text_cols, first, m, nullptr, 0, &len) != BP_OK) return HandleErr(h);
            TextString text = w->SpareText(which);
            std::string err;
            if (text_bgzf) {  // compressed where it was rendered: a fifth of the bytes cross the bus and reach the file
              size_t clen = 0;  // (first the length, then a string of just that size: growing one fills it)
              if (bp_render_fetch_bgzf(h, nullptr, 0, &clen) != BP_OK) return HandleErr(h);
              text.resize(clen);
              if (bp_render_fetch_bgzf(h, &text[0], text.size(), &clen) != BP_OK) return HandleErr(h);
              err = w->WriteBgzf(which, std::move(text));
            } else {
              text.resize(len);
              if (bp_render_fetch(h, &text[0], text.size()) != BP_OK) return HandleErr(h);
              err = w->WriteText(which, std::move(text));
            }
            if (!err.empty()) return err;
          }
        }
      }
    }
    return "";
  }

  std::string FinishJobRows(const Job& j, std::vector<bp_handle*>& hs) {
    // per-read lists can be long: smaller row chunks keep the host copies bounded
    const int64_t chunk = per_read ? (kRowChunk >> 6) : kRowChunk;
    Chunk buf[2];
    for (size_t si = 0; si < streams.size(); si++) {
      bp_handle* h = hs[si];
      const int64_t n = bp_num_rows(h);
      // the copy of chunk k+1 from the device runs while chunk k is rendered
      int cur = 0;
      if (n > 0) Fetch(h, j, 0, std::min(chunk, n), &buf[0]);
      for (int64_t first = 0; first < n; first += chunk, cur ^= 1) {
        Chunk& c = buf[cur];
        std::thread next;
        if (first + chunk < n)
          next = std::thread([&, first] { Fetch(h, j, first + chunk, std::min(chunk, n - first - chunk), &buf[cur ^ 1]); });
        std::string err = c.err;
        if (err.empty() && c.lo < c.hi && streams[si].writer_rev) {
          // one-pass -per-strand (never with per-read columns): each file its strand's share
          std::vector<uint32_t> share(c.vals.begin() + c.lo * BP_ROW_U32, c.vals.begin() + c.hi * BP_ROW_U32);
          StrandShare(share.data(), c.hi - c.lo, 1);
          err = streams[si].writer_rev->Write(c.ref.data() + c.lo, c.pos.data() + c.lo, share.data(), c.hi - c.lo, nullptr, nullptr);
          StrandShare(c.vals.data() + c.lo * BP_ROW_U32, c.hi - c.lo, 0);
        }
        if (err.empty() && c.lo < c.hi)
          err = streams[si].writer->Write(c.ref.data() + c.lo, c.pos.data() + c.lo, c.vals.data() + c.lo * BP_ROW_U32,
                                          c.hi - c.lo, per_read ? c.off.data() : nullptr,
                                          per_read ? c.feats.data() : nullptr);
        if (next.joinable()) next.join();
        if (!err.empty()) return err;
      }
    }
    return "";
  }

  void Abort() {
    for (Stream& s : streams) {
      if (s.h) bp_destroy(s.h);
      s.h = nullptr;
    }
  }

 private:
  const Opts& opts_;
  const BamHeader& hd_;
  const BEDUnion& bed_;
  int ignore_strand_;
  int64_t bed_pad_ = 0;
};

// Jobs are closed (device computation, rows to the host, rendering, file writes) on closer
// threads, one per device, while the main thread already reads the next job's records: the
// reference overlaps its jobs across goroutines (pileup.go:862) and pays for the output
// conversion serially at the end (output.go:143-148); here reading, computing and output overlap.
// A closer computes its job at once and then waits for its turn to write: output is in job order.
// One job can wait behind the one a closer is working on, so a device holds three jobs at most
// (the open one, one queued, one being closed).
struct CloseGate {  // shared by the closers of a run
  std::mutex mu;
  std::condition_variable cv;
  int64_t next_seq = 0;  // the job whose turn it is to write
  std::string err;       // first error of the run, in job order as far as the turns go
};

class Closer {
 public:
  Closer(Driver* d, CloseGate* g) : drv_(d), gate_(g), th_([this] { Run(); }) {}
  ~Closer() { Join(); }
  // Blocks while a job is already waiting behind the one being closed.
  std::string Push(const Job& j, std::vector<bp_handle*> hs, const JobRows& jr) {
    std::unique_lock<std::mutex> lk(mu_);
    cv_.wait(lk, [this] { return q_.size() < 2; });
    q_.push_back({j, std::move(hs), jr});
    cv_.notify_all();
    lk.unlock();
    std::unique_lock<std::mutex> gl(gate_->mu);
    return gate_->err;
  }
  void Join() {
    {
      std::unique_lock<std::mutex> lk(mu_);
      done_ = true;
      cv_.notify_all();
    }
    if (th_.joinable()) th_.join();
  }
  double busy_s() const { return busy_s_; }
  double part_s(int k) const { return part_s_[k]; }  // 0 compute (bp_finish), 1 output (text to the sink), 2 handles freed

 private:
  struct Item { Job job; std::vector<bp_handle*> hs; JobRows rows; };
  void Run() {
    for (;;) {
      Item* it;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [this] { return !q_.empty() || done_; });
        if (q_.empty()) return;
        it = &q_.front();  // stays in the queue while it is worked on: it counts against Push
      }
      const double t0 = Now();
      std::string e;
      {
        std::unique_lock<std::mutex> gl(gate_->mu);
        e = gate_->err;
      }
      if (e.empty()) e = drv_->ComputeJob(it->hs);
      const double t1 = Now();
      {
        std::unique_lock<std::mutex> gl(gate_->mu);
        gate_->cv.wait(gl, [&] { return gate_->next_seq == it->job.seq; });
        if (gate_->err.empty()) gate_->err = e;
        e = gate_->err;
      }
      const double t2 = Now();
      if (e.empty()) e = drv_->OutputJob(it->job, it->hs, it->rows);
      const double t3 = Now();
      Driver::FreeHandles(it->hs);
      part_s_[0] += t1 - t0; part_s_[1] += t3 - t2; part_s_[2] += Now() - t3;
      {
        std::unique_lock<std::mutex> gl(gate_->mu);
        if (gate_->err.empty()) gate_->err = e;
        gate_->next_seq++;
        gate_->cv.notify_all();
      }
      busy_s_ += Now() - t0;
      std::unique_lock<std::mutex> lk(mu_);
      q_.pop_front();
      cv_.notify_all();
    }
  }
  Driver* drv_;
  CloseGate* gate_;
  std::mutex mu_;
  std::condition_variable cv_;
  std::deque<Item> q_;
  bool done_ = false;
  double busy_s_ = 0, part_s_[3] = {0, 0, 0};
  std::thread th_;
};

// The open job's pieces go to the device (bp_submit_bam: host scan, copy, pack kernel) on a
// thread of their own, in order, while the main thread walks the next chunk.  One submitter per
// output stream (-per-strand has two handles a job, fed the same pieces side by side).
class Submitter {
 public:
  Submitter(Driver* d, size_t stream) : drv_(d), stream_(stream), th_([this] { Run(); }) {}
  ~Submitter() {
    {
      std::unique_lock<std::mutex> lk(mu_);
      done_ = true;
      cv_.notify_all();
    }
    if (th_.joinable()) th_.join();
  }
  // Blocks while three pieces are already waiting; returns the first error so far.
  std::string Push(Piece&& p) {
    std::unique_lock<std::mutex> lk(mu_);
    cv_.wait(lk, [this] { return q_.size() < 3; });
    if (err_.empty()) {
      q_.push_back(std::move(p));
      cv_.notify_all();
    }
    return err_;
  }
  // Every piece pushed so far has been submitted (or dropped after an error).
  std::string Drain() {
    std::unique_lock<std::mutex> lk(mu_);
    cv_.wait(lk, [this] { return q_.empty() && !working_; });
    return err_;
  }
  double busy_s() const { return busy_s_; }
  double release_s() const { return release_s_; }  // of busy_s: handing the pieces' buffers back
  int64_t pieces() const { return pieces_; }

 private:
  void Run() {
    for (;;) {
      Piece p;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [this] { return !q_.empty() || done_; });
        if (q_.empty()) return;
        p = std::move(q_.front());
        q_.pop_front();
        working_ = true;
        cv_.notify_all();
      }
      const double t0 = Now();
      std::string e = drv_->SubmitPiece(stream_, p);
      const double t1 = Now();
      p = Piece();  // the chunk's buffer goes back to the reader's pool
      busy_s_ += Now() - t0;
      release_s_ += Now() - t1;
      pieces_++;
      std::unique_lock<std::mutex> lk(mu_);
      working_ = false;
      if (!e.empty() && err_.empty()) {
        err_ = e;
        q_.clear();
      }
      cv_.notify_all();
    }
  }
  Driver* drv_;
  size_t stream_;
  std::mutex mu_;
  std::condition_variable cv_;
  std::deque<Piece> q_;
  bool done_ = false, working_ = false;
  std::string err_;
  double busy_s_ = 0, release_s_ = 0;
  int64_t pieces_ = 0;
  std::thread th_;
};

}  // namespace

std::string Pileup(const std::string& xampath, const std::string& fapath, const std::string& format,
                   const std::string& out_prefix, const Opts& opts) {
  const double t_entry = Now();
  // 1. options (pileup.go:974-1025)
  if (opts.Clip < 0 || opts.Clip * 2 >= opts.MaxReadLen) return "Pileup: invalid clip= argument";
  if (opts.MaxReadLen > opts.MaxReadSpan)
    return "Pileup: max-read-len= argument cannot be larger than max-read-span= argument";
  enum { kRio, kBsTsv, kBsTsvBgz, kTsv, kTsvBgz } fmt;
  if (format == "basestrand-rio") fmt = kRio;
  else if (format == "basestrand-tsv") fmt = kBsTsv;
  else if (format == "basestrand-tsv-bgz") fmt = kBsTsvBgz;
  else if (format == "tsv") fmt = kTsv;
  else if (format == "tsv-bgz") fmt = kTsvBgz;
  else return "Pileup: unrecognized format= argument";
  const int cols_default = kColDpRef | kColHighQ | kColLowQ;
  int cols = cols_default;
  if (!opts.Cols.empty()) {
    if (fmt == kRio) return "Pileup: -cols cannot be used with basestrand-rio output";
    std::string err = ParseCols(opts.Cols, cols_default, &cols);
    if (!err.empty()) return err;
  }

  // 2. BAM header, BED / region (pileup.go:1042-1105)
  if (opts.Region.empty() && opts.BedPath.empty()) return "Pileup: either -bed or -region is currently required";
  if (xampath.size() >= 4 && xampath.compare(xampath.size() - 4, 4, ".pam") == 0)
    return "Pileup: PAM input is not supported by this build (BAM only)";
  BamReader bam;
  bam.SetPinned(true);  // the device copies records straight out of the reader's buffers
  {
    std::string err = bam.Open(xampath);
    if (!err.empty()) return err;
  }
  const BamHeader& hd = bam.header();
  // The index is optional here (the reference requires one, bamprovider.go:76-100): -index,
  // else <bam>.bai, else <bam>.gbai if present; without one the BAM is read front to back.
  BamIndex index;
  {
    std::string ipath = opts.BamIndexPath;
    if (ipath.empty()) {
      if (access((xampath + ".bai").c_str(), R_OK) == 0) ipath = xampath + ".bai";
      else if (access((xampath + ".gbai").c_str(), R_OK) == 0) ipath = xampath + ".gbai";
    }
    if (!ipath.empty()) {
      std::string err = index.Load(ipath);
      if (!err.empty()) return err;
    }
  }
  BEDUnion bed;
  Entry region;
  if (!opts.Region.empty()) {
    std::string err = ParseRegionString(opts.Region, &region);
    if (!err.empty()) return err;
  }
  int region_ref = -1;
  if (!opts.BedPath.empty()) {
    if (!opts.Region.empty()) return "Pileup: -region and -bed flags can't be used together yet";
    std::string err = BEDUnion::FromPath(opts.BedPath, hd.ref_names, &bed);
    if (!err.empty()) return err;
  } else {
    for (size_t i = 0; i < hd.ref_names.size(); i++)
      if (hd.ref_names[i] == region.ref_name) { region_ref = (int)i; break; }
    if (region_ref < 0) return "Pileup: region= contig not in BAM/PAM";
    // A contig-only region is [0, 2^31-2) in the reference (bedunion.go:599-603), which then
    // walks off the end of the reference sequence at output time (output.go:158).  Clamp to
    // the contig length instead (documented divergence).
    if (region.end > hd.ref_lens[(size_t)region_ref]) region.end = hd.ref_lens[(size_t)region_ref];
    if (region.start0 >= region.end) return "Pileup: region= starts past the end of contig " + region.ref_name;
    std::string err = BEDUnion::FromEntries({region}, hd.ref_names, &bed);
    if (!err.empty()) return err;
  }

  // 3. reference sequences (pileup.go:1107-1114).  Loading them is a good part of a second for a
  //    human chromosome, and so is bringing up the HIP runtime: the two run side by side.
  Fasta fa;
  std::vector<const std::string*> ref_seqs;
  if (access(fapath.c_str(), R_OK) != 0) return "open " + fapath + ": no such file or directory";
  std::future<int> dev_init = std::async(std::launch::async, [&opts] { return bp_init(opts.Device); });
  {
    std::string err = LoadFaSeq8(fapath, &fa);
    std::string warnings;
    if (err.empty()) err = FaToStringSlice(fa, hd, &ref_seqs, &warnings);
    if (!err.empty()) {
      (void)dev_init.get();
      return err;
    }
    if (!warnings.empty()) fputs(warnings.c_str(), stderr);
  }

  // 4. pileupSNPMain set-up (pileup.go:812-835, :887-888, :944-966)
  const bool tsv_format = fmt == kTsv || fmt == kTsvBgz;
  if (opts.PerStrand && !tsv_format)
    return "pileupSNPMain: single-strand mode not supported with basestrand output (strands are already "
           "tracked separately)";
  if (opts.MinBaseQual < 0 || opts.MinBaseQual >= 96) return "newQualPassTable: minBaseQual too large";
  if (fmt == kRio)
    return "Pileup: basestrand-rio output is not supported by this build (the recordio container lives in "
           "grailbio/base, which the reference tree does not include); use basestrand-tsv";
  const bool per_read = (cols & (kColEndDists | kColQuals | kColFraglens | kColStrands)) != 0;  // pileup.go:888
  if (per_read && opts.Stitch)
    // the reference panics at the first pair it merges (pileup.go:422-443); refused up front
    return "stitched per-read features not yet supported";
  if (dev_init.get() != BP_OK) {
    // (the message is the initialising thread's: ask again on this one)
    (void)bp_init(opts.Device);
    return bp_last_error(nullptr);
  }
  // -devices N | all: N logical devices, mapped round-robin onto the physical ones
  std::vector<int> devices;
  if (opts.Devices != 0) {
    const int n_phys = bp_device_count();
    if (n_phys <= 0) return bp_last_error(nullptr);
    if (opts.Devices < -1 || opts.Devices > 1024) return "Pileup: invalid devices= argument";
    const int n = opts.Devices == -1 ? n_phys : opts.Devices;
    for (int k = 0; k < n; k++) devices.push_back(k % n_phys);
  }
  // From here on BGZF blocks are inflated on the device (what the reader has read ahead so far,
  // the header's blocks among it, went through the host threads; BIO_PILEUP_HOST_INFLATE=1 keeps it
  // that way, as before round 3).
  // ... and, where nothing on the host needs the records' bytes (no mate pairing by name, no aux-tag
  // filters, no region or index seeks, one device), the stream STAYS on the device: records are
  // found and their fixed fields read there (bp_bam_scan), 45 bytes of columns per record come to
  // the host instead of the record, and the pack kernel reads the records where the inflate kernel
  // left them (BIO_PILEUP_HOST_SCAN=1: the bytes come back and the host walks them).
  bool resident = false;
  if (const char* e = getenv("BIO_PILEUP_HOST_INFLATE"); !(e && *e && *e != '0')) {
    const char* hs = getenv("BIO_PILEUP_HOST_SCAN");
    bool one_device = true;
    for (int d : devices) one_device &= d == opts.Device;
    resident = !(hs && *hs && *hs != '0') && opts.MinBagDepth == 0 && !opts.RemoveSq && region_ref < 0 &&
               !index.loaded() && one_device;
    if (resident) {
      // (stitching pairs reads by name: the scan then gathers the names on the device and sends
      // them along with the columns, ~30 bytes a record)
      std::string err = bam.StartResident(opts.Device, opts.Stitch);
      if (!err.empty()) return err;
    } else {
      bam.SetDeviceInflate(opts.Device);
    }
  }
  SetHostThreads(opts.Parallelism);

  Driver drv(opts, hd, bed, tsv_format ? 1 : 0);
  drv.devices = devices;
  drv.per_read = per_read;
  // without per-read columns the rows become text on the device (BIO_PILEUP_HOST_TEXT=1 keeps
  // the host renderer, for tests that compare the two)
  drv.device_text = !per_read && !getenv("BIO_PILEUP_HOST_TEXT");
  drv.text_tsv = tsv_format;
  drv.text_cols = cols & (kColDpRef | kColDpAlt | kColHighQ | kColLowQ);
  drv.ref_seqs = &ref_seqs;
  // -per-strand: the reference runs pileupSNPMain twice, forward reads then reverse reads
  // (pileup.go:1118-1125).  Here one split-strand pass yields both files when that is exact (see
  // Stream::writer_rev; BIO_PILEUP_TWO_PASS=1 keeps two handles fed side by side, as before round 3);
  // otherwise two handles are fed the same records side by side.
  const char* two_pass_env = getenv("BIO_PILEUP_TWO_PASS");
  const bool one_pass = opts.PerStrand && opts.MinBaseQual == 0 && !opts.Stitch && !per_read &&
                        !(two_pass_env && *two_pass_env && *two_pass_env != '0');
  const int n_stream = opts.PerStrand && !one_pass ? 2 : 1;
  drv.streams.resize((size_t)n_stream);
  const bool bgz = fmt == kTsvBgz || fmt == kBsTsvBgz;
  // -bgz: the device compresses what it renders (BIO_PILEUP_HOST_DEFLATE=1: libdeflate/zlib level 6 on
  // the host's threads as before - files a fifth smaller, runs several times longer)
  drv.text_bgzf = bgz && drv.device_text && !getenv("BIO_PILEUP_HOST_DEFLATE");
  auto new_writer = [&](const std::string& path, std::string* err) {
    return tsv_format ? NewTsvWriter(path, cols, bgz, hd.ref_names, ref_seqs, err)
                      : NewBasestrandTsvWriter(path, cols, bgz, hd.ref_names, ref_seqs, err);
  };
  for (int s = 0; s < n_stream; s++) {
    Stream& st = drv.streams[(size_t)s];
    std::string main_path = out_prefix;
    if (opts.PerStrand) {
      if (!one_pass) st.strand_req = s == 0 ? BP_STRAND_FWD : BP_STRAND_REV;
      main_path += s == 0 ? ".strand.fwd" : ".strand.rev";
    }
    std::string err;
    st.writer = new_writer(main_path, &err);
    if (!err.empty()) return err;
    if (one_pass) {
      st.writer_rev = new_writer(out_prefix + ".strand.rev", &err);
      if (!err.empty()) return err;
    }
  }

  // 5. jobs: BED-covered ranges of each contig, at most JobSpan() positions each
  const int64_t job_span = std::max<int64_t>(JobSpan(), (int64_t)opts.MaxReadSpan + 1);
  std::vector<Job> jobs;
  for (size_t r = 0; r < bed.id_map.size(); r++) {
    const std::vector<PosType>& e = bed.id_map[r];
    if (e.size() < 2) continue;
    for (int64_t s = (int64_t)e.front() / job_span * job_span; s < e.back(); s += job_span) {
      const int64_t t = std::min<int64_t>(s + job_span, kPosTypeMax);
      if (bed.IntersectsByID((int)r, (PosType)s, (PosType)t))
        jobs.push_back({(int)r, (PosType)s, (PosType)t, (int64_t)jobs.size()});
    }
  }

  // 6. one pass over the BAM.  Job k is fed the records that start in
  //    [start_k - pad, end_k + pad_end) and computes the BED positions of that padded range;
  //    only its rows in [start_k, end_k) are written.  Without stitching pad = maxReadSpan
  //    on the left (gbam.Shard.PaddedStart, encoding/bam/shard.go:64-120) is all a position
  //    can see.  With stitching a read's mate, and the BED positions that decide whether the
  //    mate is kept (pileup.go:773-777), lie up to two spans away on either side; padding by
  //    that much makes every job see its pairs exactly as one whole-contig job would, so the
  //    output does not depend on where the cuts fall.  pend[k - ji] holds what job k has
  //    been given so far; only the front job has device handles open.
  const bool need_aux = opts.MinBagDepth != 0 || opts.RemoveSq;
  const int64_t pad = opts.Stitch ? 2 * (int64_t)opts.MaxReadSpan : (int64_t)opts.MaxReadSpan;
  const int64_t pad_end = opts.Stitch ? pad : 0;
  drv.SetPadding(opts.Stitch ? pad : 0);
  Stats st;
  const double t_start = Now();
  // The reader hands out the inflated stream a chunk at a time (all whole records of up to 16 MiB,
  // in pinned memory).  Nothing is copied on the host: a job gets, per chunk, the list of the
  // records it wants (a Piece), and the device takes the bytes from the chunk itself.  A chunk
  // that lies inside one job's range and inside scanned shards - nearly all of a whole-genome run
  // - is handed over whole on what the reader's walk saw in its fixed fields; other chunks go through
  // the record-by-record rules below.  pend[k - ji]: pieces job k has been given before it opens.
  std::deque<std::vector<Piece>> pend(1);
  std::vector<std::vector<uint32_t>> sel(1);  // of the chunk at hand: records for job ji + k
  size_t ji = 0;
  bool open = false;
  std::string err;
  CloseGate gate;
  std::vector<std::unique_ptr<Closer>> closers;
  for (size_t k = 0; k < drv.n_closers(); k++) closers.emplace_back(new Closer(&drv, &gate));
  std::vector<std::unique_ptr<Submitter>> submitters;
  for (size_t k = 0; k < drv.streams.size(); k++) submitters.emplace_back(new Submitter(&drv, k));
  auto push_piece = [&](Piece&& p) -> std::string {
    std::string e;
    for (size_t k = 0; k + 1 < submitters.size(); k++) {
      std::string ek = submitters[k]->Push(Piece(p));
      if (e.empty()) e = ek;
    }
    std::string el = submitters.back()->Push(std::move(p));
    return e.empty() ? el : e;
  };
  auto drain = [&]() -> std::string {
    std::string e;
    for (auto& sm : submitters) {
      std::string ek = sm->Drain();
      if (e.empty()) e = ek;
    }
    return e;
  };
  double open_s = 0;  // the main thread in OpenJob (handles created, BED and reference stretch uploaded)
  auto open_job = [&]() -> std::string {
    const double t_open = Now();
    std::string e = drv.OpenJob(jobs[ji]);
    open_s += Now() - t_open;
    open = e.empty();
    // what the job was given while the ones before it were open
    for (Piece& p : pend[0])
      if (e.empty()) e = push_piece(std::move(p));
    pend[0].clear();
    return e;
  };
  BamReader::Chunk ck;
  // The chunk's records chosen so far become pieces: the open job's go to the submitter.
  auto flush_sel = [&]() -> std::string {
    std::string e;
    for (size_t k = 0; k < sel.size(); k++) {
      if (sel[k].empty()) continue;
      Piece p;
      p.hold = ck.hold;
      p.data = ck.data;
      p.bytes = ck.bytes;
      p.off.swap(sel[k]);
      while (pend.size() <= k) pend.emplace_back();
      if (k == 0 && open) {
        if (e.empty()) e = push_piece(std::move(p));
      } else {
        pend[k].push_back(std::move(p));
      }
    }
    return e;
  };
  auto close_job = [&]() -> std::string {
    std::string e = flush_sel();
    if (!sel.empty()) sel.erase(sel.begin());
    if (sel.empty()) sel.emplace_back();
    double t0 = Now();
    if (e.empty()) e = drain();
    st.submit_s += Now() - t0;
    t0 = Now();
    if (e.empty()) e = closers[ji % closers.size()]->Push(jobs[ji], drv.TakeHandles(), drv.open_rows);
    st.close_wait_s += Now() - t0;
    st.jobs++;
    open = false;
    ji++;
    pend.pop_front();
    if (pend.empty()) pend.emplace_back();
    return e;
  };
  // jobs whose padded range ends at or before (ref, pos) are complete
  auto close_upto = [&](int32_t ref, int32_t pos) -> std::string {
    while (ji < jobs.size() && (jobs[ji].ref < ref || (jobs[ji].ref == ref && (int64_t)jobs[ji].end + pad_end <= pos))) {
      std::string e;
      if (!open && !(e = open_job()).empty()) return e;
      if (!(e = close_job()).empty()) return e;
    }
    return "";
  };
  // jobs behind the front one that also want a record at (ref, pos): those it lies in the padding of
  auto followers = [&](int32_t ref, int32_t pos) {
    size_t n = 0;
    for (size_t k = ji + 1; k < jobs.size() && jobs[k].ref == ref && (int64_t)jobs[k].start - pad <= pos; k++) n++;
    return n;
  };
  // Is the 100 kb position shard that starts at shard0 one the reference scans: does it meet a
  // padded BED interval [start - pad, end + pad_end)?
  auto shard_scanned = [&](int32_t ref, int64_t shard0) {
    return bed.IntersectsByID(ref, (PosType)std::max<int64_t>(0, shard0 - pad_end),
                              (PosType)std::min<int64_t>(kPosTypeMax, shard0 + kSeekGap + pad));
  };
  BamReader::Raw r;
  int32_t last_ref = -1, last_pos = -1;
  NeedCursor need(bed, pad, pad_end);
  int skip_ref = -1;  // after a seek: records before (skip_ref, skip_pos) are not wanted
  int64_t skip_pos = 0;
  bool no_records = false;
  if (index.loaded() && region_ref >= 0) {
    // bamIterator.reset (bamprovider.go:286-348): start at the region's first position
    uint64_t voff = 0;
    if (!index.Offset(region_ref, region.start0, &voff)) {
      no_records = true;  // no record at or after the region start
    } else if ((voff >> 16) > bam.BlockOffset()) {
      err = bam.Seek(voff);
      st.seeks++;
    }
  }
  bool done = false;
  while (err.empty() && !no_records && !done && ji < jobs.size()) {
    int rc = bam.NextChunk(&ck);
    if (rc < 0) { err = bam.err(); break; }
    if (rc == 0) break;
    const size_t n = ck.off.size();
    const BamReader::Chunk& cf = ck;  // what the reader's walk saw in the records' fixed fields
    // ---- the whole chunk at once
    bool whole = region_ref < 0 && !need_aux && cf.fields_ok && cf.sorted && cf.one_ref && cf.first_ref >= 0 &&
                 !(cf.first_ref < last_ref || (cf.first_ref == last_ref && cf.first_pos < last_pos)) &&
                 !(cf.first_ref < skip_ref || (cf.first_ref == skip_ref && cf.first_pos < skip_pos));
    if (whole)
      for (int64_t sh = (int64_t)cf.first_pos / kSeekGap * kSeekGap; sh <= cf.last_pos && whole; sh += kSeekGap)
        whole = shard_scanned(cf.first_ref, sh);
    if (whole) {
      // The chunk is sorted, lies on one contig and inside scanned shards, and every record's
      // fields fit: it is handed over in runs of records that the same jobs want, on what the
      // walk saw - a run ends where a job's padded range starts or ends, found by bisection
      // over the records' positions (a chunk is cut at most a few times; walking a 256 MiB
      // chunk record by record for the one cut in it was a tenth of a second).
      st.records_read += (int64_t)n;
      last_ref = cf.last_ref;
      last_pos = cf.last_pos;
      const int32_t ref = cf.first_ref;
      auto pos_of = [&](size_t i) {
        if (ck.cols) return ck.cols->pos[i];
        int32_t v;
        memcpy(&v, ck.data + ck.off[i] + 8, 4);
        return v;
      };
      size_t i = 0;
      while (i < n && err.empty()) {
        const int32_t p0 = pos_of(i);
        if (!(err = close_upto(ref, p0)).empty()) break;
        if (ji >= jobs.size() || jobs[ji].ref > ref) break;  // nothing behind wants the rest
        // (jobs[ji].ref == ref now: close_upto has closed the jobs of earlier contigs)
        // the next position at which the set of jobs that want a record changes
        int64_t cut = (int64_t)jobs[ji].end + pad_end;
        for (size_t k = ji + 1; k < jobs.size() && jobs[k].ref == ref; k++) {
          const int64_t sk = (int64_t)jobs[k].start - pad;
          if (sk > p0) { cut = std::min(cut, sk); break; }
        }
        size_t lo = i + 1, hi = n;  // first record at or behind the cut
        while (lo < hi) {
          const size_t mid = (lo + hi) / 2;
          if ((int64_t)pos_of(mid) >= cut) hi = mid; else lo = mid + 1;
        }
        const size_t j = lo;
        if (!open && !(err = open_job()).empty()) break;
        const size_t nf = followers(ref, p0);
        st.records_kept += (int64_t)(j - i);
        if (ck.cols) {
          // a device chunk: the run is a range of the scan's columns, nothing is copied
          for (size_t k = 0; k <= nf && err.empty(); k++) {
            Piece p;
            p.hold = ck.hold;
            p.dev = ck.dev;
            p.cols = ck.cols;
            p.bytes = ck.bytes;
            p.first = (int64_t)i;
            p.count = (int64_t)(j - i);
            while (pend.size() <= k) pend.emplace_back();
            if (k == 0) err = push_piece(std::move(p));  // (the front job is open)
            else pend[k].push_back(std::move(p));
          }
        } else {
          if (sel.size() < nf + 1) sel.resize(nf + 1);
          for (size_t k = 0; k <= nf; k++) sel[k].assign(ck.off.begin() + (ptrdiff_t)i, ck.off.begin() + (ptrdiff_t)j);
          err = flush_sel();
        }
        i = j;
      }
      continue;
    }
    // ---- record by record (a device chunk's bytes come to the host for it)
    if (!ck.data && !(err = bam.FetchHost(&ck)).empty()) break;
    ck.cols = nullptr;
    for (size_t i = 0; i < n && err.empty(); i++) {
      const uint8_t* rec = ck.data + ck.off[i];
      st.records_read++;
      if (!BamReader::Parse(rec, &r)) { err = "bam: record fields exceed block size"; break; }
      if (r.ref_id < 0) { done = true; break; }  // unplaced reads sort last
      if (r.ref_id < last_ref || (r.ref_id == last_ref && r.pos < last_pos)) {
        err = "Pileup: BAM is not coordinate-sorted at " + std::string(r.name, r.l_name);
        break;
      }
      last_ref = r.ref_id;
      last_pos = r.pos;
      if (region_ref >= 0) {  // iterator of the single region shard (bamprovider.go:417-435)
        if (r.ref_id < region_ref || (r.ref_id == region_ref && r.pos < region.start0)) continue;
        if (r.ref_id > region_ref || r.pos >= region.end) { done = true; break; }
      } else {
        // The reference scans whole 100 kb position shards and only those that meet the padded BED
        // (bamprovider.go:262-263, pileup_support.go:71-85): a record is scanned - and goes through
        // the read-length checks - iff its shard does.  The same rule here, index or not: both
        // ways the device is fed the same records and raises the same errors; the index only saves
        // reading the rest.
        if (r.ref_id < skip_ref || (r.ref_id == skip_ref && r.pos < skip_pos)) continue;
        const int64_t shard0 = (int64_t)r.pos / kSeekGap * kSeekGap;
        int nref = 0;
        int64_t nstart = 0;
        if (!need.Next(r.ref_id, shard0, &nref, &nstart)) { done = true; break; }  // past the last BED interval
        if (nref != r.ref_id || nstart >= shard0 + kSeekGap) {
          if (!index.loaded()) continue;
          const int64_t target = nstart / kSeekGap * kSeekGap;  // the first record of the next shard that is scanned
          uint64_t voff = 0;
          if (!index.Offset(nref, (int32_t)target, &voff)) { done = true; break; }  // no records from there on
          skip_ref = nref;
          skip_pos = target;
          // The reader stands behind this chunk.  A target further on than that lies behind every
          // record of the chunk too (the index points at or before the first record it is for):
          // the rest of the chunk is dropped with what the seek skips.
          if ((voff >> 16) > bam.BlockOffset()) {
            if (!(err = bam.Seek(voff)).empty()) break;
            st.seeks++;
            break;
          }
          continue;
        }
      }
      if (!(err = close_upto(r.ref_id, r.pos)).empty()) break;
      if (ji >= jobs.size()) { done = true; break; }
      if (jobs[ji].ref != r.ref_id) continue;  // contig without BED positions
      if (need_aux) {
        // Same read-level predicate the device applies first (pileup.go:691), so the aux tags
        // are only demanded of reads the reference would have asked (pileup.go:695-716).
        if ((opts.FlagExclude & r.flags) || opts.Mapq > (int)r.mapq || r.n_cigar == 0) continue;
        int64_t v = 0;
        if (opts.RemoveSq) {
          if (!AuxInt(r.aux, r.l_aux, "DL", &v)) { err = "sam.Record.LibraryBagSize: no DL aux tag on read " + std::string(r.name, r.l_name); break; }
          if (v < 2) continue;
        }
        if (opts.MinBagDepth != 0) {
          if (!AuxInt(r.aux, r.l_aux, "DS", &v)) { err = "sam.Record.BagSize: no DS aux tag on read " + std::string(r.name, r.l_name); break; }
          if (v < opts.MinBagDepth) continue;
        }
      }
      // The front job also takes reads ahead of its padded start: they cannot reach it, but
      // they go through the library's read checks as they do in whichever reference job
      // scans them.
      if (!open && !(err = open_job()).empty()) break;
      sel[0].push_back(ck.off[i]);
      st.records_kept++;
      for (size_t k = ji + 1; k < jobs.size() && jobs[k].ref == r.ref_id && (int64_t)jobs[k].start - pad <= r.pos; k++) {
        if (sel.size() <= k - ji) sel.resize(k - ji + 1);
        sel[k - ji].push_back(ck.off[i]);
      }
    }
    if (err.empty()) err = flush_sel();
  }
  for (auto& v : sel) v.clear();
  ck = BamReader::Chunk();
  // jobs after the last record still emit their (zero) rows (flushEmptyContigs,
  // pileup_support.go:112-123)
  while (err.empty() && ji < jobs.size()) {
    if (!open && !(err = open_job()).empty()) break;
    err = close_job();
  }
  double cpart[3] = {0, 0, 0};
  {
    const double t0 = Now();
    (void)drain();
    for (auto& c : closers) c->Join();
    st.close_wait_s += Now() - t0;
    for (auto& c : closers) st.close_s += c->busy_s();
    for (auto& c : closers)
      for (int k = 0; k < 3; k++) cpart[k] += c->part_s(k);
    if (err.empty()) err = gate.err;
  }
  if (!err.empty()) {
    drv.Abort();
    return err;
  }
  for (Stream& s : drv.streams) {
    std::string e = s.writer->Close();
    if (!e.empty()) return e;
    if (s.writer_rev && !(e = s.writer_rev->Close()).empty()) return e;
  }
  st.total_s = Now() - t_start;
  double sr = 0, si = 0, stk = 0, ssc = 0;
  bam.StageSeconds(&sr, &si, &stk, &ssc);
  if (const char* e = getenv("BIO_PILEUP_STATS"); e && *e && *e != '0')
    fprintf(stderr,
            "bio-pileup stats: records_read=%lld records_kept=%lld seeks=%lld jobs=%lld index=%d "
            "open_s=%.3f submit_s=%.3f (busy on its own thread, %lld pieces, %.3f s of it handing buffers back; the reader waited %.3f s for it) close_s=%.3f (on its own "
            "threads: compute %.3f, text out %.3f, handles freed %.3f; the reader waited %.3f s) read_s=%.3f (of which waiting for inflated blocks %.3f) device_inflate_s=%.3f stage_read_s=%.3f stage_inflate_s=%.3f (buffers %.3f) stage_scan_s=%.3f resident=%d (host walks %lld) total_s=%.3f (after %.3f s of set-up: header, BED, FASTA, device)\n",
            (long long)st.records_read, (long long)st.records_kept, (long long)st.seeks, (long long)st.jobs,
            index.loaded() ? 1 : 0, open_s, submitters[0]->busy_s(), (long long)submitters[0]->pieces(), submitters[0]->release_s(), st.submit_s, st.close_s, cpart[0], cpart[1], cpart[2], st.close_wait_s,
            st.total_s - st.submit_s - st.close_wait_s, bam.WaitSeconds(), bam.DeviceInflateSeconds(), sr, si, stk, ssc, resident ? 1 : 0, (long long)bam.ResidentFallbacks(), st.total_s, t_start - t_entry);
  return "";
}

}  // namespace bio
