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
// pid is not the constructing one) runs every job on the calling thread alone and never touches the inherited locks.
class CopyCrew {
 public:
  explicit CopyCrew(uint32_t helpers) : owner_(getpid()) {
    for (uint32_t i = 0; i < helpers; ++i) threads_.emplace_back([this, i] { Run(i); });
  }
  ~CopyCrew() {
    { std::lock_guard<std::mutex> l(mu_); stop_ = true; ++generation_; }
    wake_.notify_all();
    for (auto& t : threads_) t.join();
  }
  // job(r0, r1) over [0, rows) cut into one share per thread (the caller's first); returns when every share is done.  `bytes` is what the
  // job moves in all: below 1 MB the caller runs it alone.  Shares must not overlap in what they write.
  template <typename Job>
  void Rows(uint32_t rows, size_t bytes, const Job& job) {
    if (getpid() != owner_) { if (rows) job(0u, rows); return; }  // a forked child: no helper threads here
    std::lock_guard<std::mutex> one_caller(caller_);
    const uint32_t parts = (uint32_t)threads_.size() + 1;
    if (parts == 1 || bytes < (1u << 20) || rows == 0) { if (rows) job(0u, rows); return; }
    {
      std::lock_guard<std::mutex> l(mu_);
      job_ = [&job](uint32_t r0, uint32_t r1) { job(r0, r1); };
      rows_ = rows;
      left_.store((uint32_t)threads_.size(), std::memory_order_relaxed);
      ++generation_;
    }
    wake_.notify_all();
    // the helpers hold a reference to `job` until they have counted out: whatever the caller's share does (throwing included),
    // this frame must not unwind before they have
    struct WaitForHelpers {
      CopyCrew* c;
      ~WaitForHelpers() {
        for (int spin = 0; spin < 4096 && c->left_.load(std::memory_order_acquire) != 0; ++spin) std::this_thread::yield();
        if (c->left_.load(std::memory_order_acquire) == 0) return;
        std::unique_lock<std::mutex> l(c->mu_);  // a helper was descheduled: sleep instead of burning the core
        c->done_.wait(l, [this] { return c->left_.load(std::memory_order_acquire) == 0; });
      }
    } wait{this};
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

 private:
  void Run(uint32_t index) {
    uint64_t seen = 0;
    for (;;) {
      std::unique_lock<std::mutex> l(mu_);
      wake_.wait(l, [&] { return generation_ != seen; });
      seen = generation_;
      if (stop_) return;
      const uint32_t rows = rows_, parts = (uint32_t)threads_.size() + 1;
      const std::function<void(uint32_t, uint32_t)>& job = job_;  // stays put until every helper has counted itself out
      l.unlock();
      const uint32_t r0 = (uint32_t)((uint64_t)rows * (index + 1) / parts), r1 = (uint32_t)((uint64_t)rows * (index + 2) / parts);
      if (r1 > r0) job(r0, r1);
      if (left_.fetch_sub(1, std::memory_order_acq_rel) == 1) {
        std::lock_guard<std::mutex> g(mu_);  // the caller may be asleep on done_ (it checks left_ under mu_)
        done_.notify_one();
      }
    }
  }
  std::vector<std::thread> threads_;
  std::mutex caller_, mu_;
  std::condition_variable wake_, done_;
  const pid_t owner_;
  uint64_t generation_ = 0;
  bool stop_ = false;
  std::function<void(uint32_t, uint32_t)> job_;
  uint32_t rows_ = 0;
  std::atomic<uint32_t> left_{0};
};

}  // namespace svc

#endif  // SVC_COPY_CREW_HPP
