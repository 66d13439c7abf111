// copy_crew.hpp -- host memory copies by several threads (csrc/host/stream_encoder.cpp: source frames into the pinned batch buffer;
// csrc/capi.hip: the staging copies of the host-pointer entry points).
#ifndef SVC_COPY_CREW_HPP
#define SVC_COPY_CREW_HPP

#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include <unistd.h>

namespace svc {

// One core moves 6 MB in ~0.55 ms, which is what a 1080p frame's share of PCIe takes in BOTH directions together, so a single copying
// thread -- not the link -- would bound a host-fed pipeline.  The calling thread takes the first share of the rows itself and waits for
// the others, so Copy() returns when the bytes are there.  One caller at a time (callers serialise on a mutex).
// fork(): threads do not survive into the child, so a crew inherited from the parent has no helpers -- the child (any process whose
// pid is not the constructing one) runs every job on the calling thread alone and never touches the inherited locks: not in Rows(), and
// not in the destructor either (a helper may have held mu_ at fork(), and joining a std::thread whose thread does not exist in this
// process waits forever; so does destroying a condition variable that was copied with waiters on it): the child leaves the state alone.
class CopyCrew {
 public:
  explicit CopyCrew(uint32_t helpers) : owner_(getpid()), s_(new State) {
    for (uint32_t i = 0; i < helpers; ++i) s_->threads.emplace_back([this, i] { Run(i); });
  }
  ~CopyCrew() {
    // a forked child destroying (or unwinding through) an inherited crew: nothing to stop, nothing to join, nothing to DESTROY -- glibc's
    // pthread_cond_destroy waits for the waiters the parent's helpers left in the copied condition variable, and the thread handles
    // name threads this process does not have: the state stays on the heap, untouched
    if (getpid() != owner_) return;
    { std::lock_guard<std::mutex> l(s_->mu); s_->stop = true; ++s_->generation; }
    s_->wake.notify_all();
    for (auto& t : s_->threads) t.join();
    delete s_;
  }
  CopyCrew(const CopyCrew&) = delete;
  CopyCrew& operator=(const CopyCrew&) = delete;
  // job(r0, r1) over [0, rows) cut into one share per thread (the caller's first); returns when every share is done.  `bytes` is what the
  // job moves in all: below 1 MB the caller runs it alone.  Shares must not overlap in what they write.
  template <typename Job>
  void Rows(uint32_t rows, size_t bytes, const Job& job) {
    if (getpid() != owner_) { if (rows) job(0u, rows); return; }  // a forked child: no helper threads here
    State& s = *s_;
    std::lock_guard<std::mutex> one_caller(s.caller);
    const uint32_t parts = (uint32_t)s.threads.size() + 1;
    if (parts == 1 || bytes < (1u << 20) || rows == 0) { if (rows) job(0u, rows); return; }
    {
      std::lock_guard<std::mutex> l(s.mu);
      s.job = [&job](uint32_t r0, uint32_t r1) { job(r0, r1); };
      s.rows = rows;
      s.left.store((uint32_t)s.threads.size(), std::memory_order_relaxed);
      ++s.generation;
    }
    s.wake.notify_all();
    // the helpers hold a reference to `job` until they have counted out: whatever the caller's share does (throwing included),
    // this frame must not unwind before they have
    struct WaitForHelpers {
      State* c;
      ~WaitForHelpers() {
        for (int spin = 0; spin < 4096 && c->left.load(std::memory_order_acquire) != 0; ++spin) std::this_thread::yield();
        if (c->left.load(std::memory_order_acquire) == 0) return;
        std::unique_lock<std::mutex> l(c->mu);  // a helper was descheduled: sleep instead of burning the core
        c->done.wait(l, [this] { return c->left.load(std::memory_order_acquire) == 0; });
      }
    } wait{&s};
    if (rows / parts) job(0u, rows / parts);
  }
  // one flat run of bytes, cut into 64 KiB rows for the crew
  void Copy(void* dst, const void* src, size_t bytes) {
    constexpr size_t kRow = 64u << 10;
    const uint32_t rows = (uint32_t)(bytes / kRow);
    if (rows) Copy(static_cast<uint8_t*>(dst), kRow, static_cast<const uint8_t*>(src), kRow, kRow, rows);
    if (bytes % kRow) std::memcpy(static_cast<uint8_t*>(dst) + (size_t)rows * kRow, static_cast<const uint8_t*>(src) + (size_t)rows * kRow, bytes % kRow);
  }
  // rows of row_bytes each, from src (pitch src_pitch) to dst (pitch dst_pitch)
  void Copy(uint8_t* dst, size_t dst_pitch, const uint8_t* src, size_t src_pitch, size_t row_bytes, uint32_t rows) {
    Rows(rows, (size_t)rows * row_bytes, [=](uint32_t r0, uint32_t r1) {
      if (dst_pitch == row_bytes && src_pitch == row_bytes) { std::memcpy(dst + (size_t)r0 * row_bytes, src + (size_t)r0 * row_bytes, (size_t)(r1 - r0) * row_bytes); return; }
      for (uint32_t y = r0; y < r1; ++y) std::memcpy(dst + (size_t)y * dst_pitch, src + (size_t)y * src_pitch, row_bytes);
    });
  }
  uint32_t threads() const { return (uint32_t)s_->threads.size() + 1; }  // the caller's included

 private:
  // everything the helpers touch lives on the heap, so that a forked child can leave it alone (see the destructor)
  struct State {
    std::vector<std::thread> threads;
    std::mutex caller, mu;
    std::condition_variable wake, done;
    uint64_t generation = 0;
    bool stop = false;
    std::function<void(uint32_t, uint32_t)> job;
    uint32_t rows = 0;
    std::atomic<uint32_t> left{0};
  };
  void Run(uint32_t index) {
    State& s = *s_;
    uint64_t seen = 0;
    for (;;) {
      std::unique_lock<std::mutex> l(s.mu);
      s.wake.wait(l, [&] { return s.generation != seen; });
      seen = s.generation;
      if (s.stop) return;
      const uint32_t rows = s.rows, parts = (uint32_t)s.threads.size() + 1;
      const std::function<void(uint32_t, uint32_t)>& job = s.job;  // stays put until every helper has counted itself out
      l.unlock();
      const uint32_t r0 = (uint32_t)((uint64_t)rows * (index + 1) / parts), r1 = (uint32_t)((uint64_t)rows * (index + 2) / parts);
      if (r1 > r0) job(r0, r1);
      if (s.left.fetch_sub(1, std::memory_order_acq_rel) == 1) {
        std::lock_guard<std::mutex> g(s.mu);  // the caller may be asleep on done (it checks left under mu)
        s.done.notify_one();
      }
    }
  }
  const pid_t owner_;
  State* s_;
};

}  // namespace svc

#endif  // SVC_COPY_CREW_HPP
