// TEST SEAM for tests/dropin/ref_encoder_*: the reference seeds its RANSAC engine from std::random_device
// (libs/motion.cpp:186-187) and so does the product wrapper (csrc/host/motion_hip.cpp); a test that compares the
// encoder's output stream needs the draws to repeat.  This object seeds the MAIN thread's engine -- the thread
// Encoder::operator() runs on, apps/encoder.cpp:228 -- before main() starts.  Not part of the product, not linked by
// the build line INTEGRATION.md gives to maintainers.
#include <cstdlib>

void SvcSeedRansac(unsigned seed);  // include/svc/motion.hpp (an addition of this repo, not in the reference's header)

namespace {
struct SeedBeforeMain {
  SeedBeforeMain() {
    const char* s = std::getenv("SVC_TEST_RANSAC_SEED");
    SvcSeedRansac(s ? (unsigned)std::strtoul(s, nullptr, 0) : 12345u);
  }
} g_seed_before_main;
}  // namespace
