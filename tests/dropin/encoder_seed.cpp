// TEST SEAM for tests/dropin/ref_app_svc_encoder: fixes the seed of the batched Encoder (csrc/host/encoder_hip.cpp) before
// main() so that the output stream repeats (SVC_TEST_ENCODER_SEED, default 4242).  Not part of the product.
#include <cstdlib>

extern "C" void SvcEncoderSeed(unsigned long long seed);

namespace {
struct SeedBeforeMain {
  SeedBeforeMain() {
    const char* s = std::getenv("SVC_TEST_ENCODER_SEED");
    SvcEncoderSeed(s ? std::strtoull(s, nullptr, 0) : 4242ull);
  }
} g_seed_before_main;
}  // namespace
