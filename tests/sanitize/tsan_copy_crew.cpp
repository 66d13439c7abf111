// tests/sanitize/tsan_copy_crew.cpp -- csrc/host/copy_crew.hpp under ThreadSanitizer (and, built a second time, under ASan + UBSan):
// the crew that stages source frames (svc::StreamEncoder) and the host-pointer entry points' copies.  Pure host code: no device.
// Checked: every byte arrives for flat and pitched copies of awkward sizes (fewer rows than threads, sizes below the crew's threshold,
// rows that do not divide), callers from several threads at once serialise, a crew with no helpers copies inline, crews start and stop
// back to back.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

#include <atomic>
#include <chrono>
#include <stdexcept>
#include <sys/wait.h>
#include <unistd.h>

#include "host/copy_crew.hpp"

static int g_fail = 0;
#define CHECK(c) do { if (!(c)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); ++g_fail; } } while (0)

static void flat(svc::CopyCrew& crew, size_t bytes, uint32_t seed) {
  std::vector<uint8_t> src(bytes + 64), dst(bytes + 64, 0xEE);
  std::mt19937 g(seed);
  for (auto& b : src) b = (uint8_t)g();
  crew.Copy(dst.data() + 7, src.data() + 3, bytes);  // unaligned on purpose
  CHECK(std::memcmp(dst.data() + 7, src.data() + 3, bytes) == 0);
  CHECK(dst[6] == 0xEE && dst[7 + bytes] == 0xEE);  // nothing outside the run
}

static void pitched(svc::CopyCrew& crew, uint32_t rows, size_t row_bytes, size_t sp, size_t dp, uint32_t seed) {
  std::vector<uint8_t> src(sp * rows + 1), dst(dp * rows + 1, 0xEE);
  std::mt19937 g(seed);
  for (auto& b : src) b = (uint8_t)g();
  crew.Copy(dst.data(), dp, src.data(), sp, row_bytes, rows);
  for (uint32_t y = 0; y < rows; ++y) {
    CHECK(std::memcmp(dst.data() + y * dp, src.data() + y * sp, row_bytes) == 0);
    if (dp > row_bytes) CHECK(dst[y * dp + row_bytes] == 0xEE);  // the padding between rows stays
  }
}

int main() {
  {
    svc::CopyCrew crew(3);
    for (size_t bytes : {(size_t)0, (size_t)1, (size_t)65535, (size_t)65536, (size_t)(1u << 20) - 1, (size_t)(1u << 20), (size_t)(1u << 20) + 65537, (size_t)6220800,
                         (size_t)25067520})
      flat(crew, bytes, (uint32_t)bytes);
    pitched(crew, 1080, 1920 * 3, 1920 * 3, 1936 * 3, 1);  // a 1080p frame into a padded batch buffer
    pitched(crew, 2, 700000, 700001, 700003, 2);            // fewer rows than threads
    pitched(crew, 3, 400000, 400000, 400000, 3);
    pitched(crew, 5, 17, 19, 23, 4);                        // below the threshold: inline
    pitched(crew, 0, 100, 100, 100, 5);
    // several callers at once: each sees its own copy complete
    std::vector<std::thread> callers;
    for (int t = 0; t < 4; ++t)
      callers.emplace_back([&crew, t] {
        for (int i = 0; i < 6; ++i) flat(crew, (size_t)(2u << 20) + 4099 * t + i, 100 + 10 * t + i);
      });
    for (auto& c : callers) c.join();
  }
  {
    svc::CopyCrew alone(0);
    flat(alone, (size_t)(3u << 20) + 5, 77);
    pitched(alone, 100, 30000, 30001, 30002, 78);
  }
  for (int i = 0; i < 20; ++i) {  // start / stop back to back, with and without work in between
    svc::CopyCrew crew(1 + i % 4);
    if (i & 1) flat(crew, (size_t)(1u << 20) + i, 200 + i);
  }
  {  // a job that throws on the caller's share: the helpers finish against a live job, the exception reaches the caller, the crew still works
    svc::CopyCrew crew(3);
    std::atomic<uint32_t> rows_done{0};
    bool thrown = false;
    try {
      crew.Rows(64, (size_t)4 << 20, [&](uint32_t r0, uint32_t r1) {
        if (r0 == 0) throw std::runtime_error("caller's share");
        std::this_thread::sleep_for(std::chrono::milliseconds(20));  // still running when the caller's frame would unwind
        rows_done.fetch_add(r1 - r0);
      });
    } catch (const std::runtime_error&) { thrown = true; }
    CHECK(thrown && rows_done.load() == 48);
    flat(crew, (size_t)(2u << 20) + 11, 300);
  }
#if !defined(__SANITIZE_THREAD__)
  {  // fork(): the child inherits the crew object but none of its threads -- a large copy there must complete on the calling thread
    svc::CopyCrew crew(3);
    flat(crew, (size_t)(2u << 20), 400);
    std::fflush(stdout);
    pid_t pid = fork();
    if (pid == 0) {
      g_fail = 0;
      flat(crew, (size_t)(3u << 20) + 9, 401);
      _exit(g_fail ? 1 : 0);
    }
    int status = -1;
    CHECK(pid > 0 && waitpid(pid, &status, 0) == pid && WIFEXITED(status) && WEXITSTATUS(status) == 0);
    flat(crew, (size_t)(2u << 20) + 1, 402);  // the parent's crew is untouched
  }
  {  // ... and a child that lets an inherited crew go out of scope (unwinding, a unique_ptr owner) must not join threads it does not have, nor
     // take a lock a helper may have held at fork(): the destructor returns at once there.  alarm(): a hang fails the test instead of stalling it
    auto* crew = new svc::CopyCrew(3);
    flat(*crew, (size_t)(2u << 20), 410);
    std::fflush(stdout);
    pid_t pid = fork();
    if (pid == 0) {
      alarm(20);
      g_fail = 0;
      flat(*crew, (size_t)(2u << 20) + 5, 411);
      delete crew;  // the forked child's destructor
      _exit(g_fail ? 1 : 0);
    }
    int status = -1;
    CHECK(pid > 0 && waitpid(pid, &status, 0) == pid && WIFEXITED(status) && WEXITSTATUS(status) == 0);
    flat(*crew, (size_t)(2u << 20) + 1, 412);
    delete crew;  // the owner's: stops and joins
  }
#endif
  if (g_fail) return 1;
  std::puts("copy crew ok");
  return 0;
}
