// TEST TOOL: prints the sample indices the C++ RANSAC wrapper (csrc/host/motion_hip.cpp) draws for `calls` consecutive
// calls after SvcSeedRansac(seed): the same libstdc++ engine and distribution types, the same rejection loop
// (libs/motion.cpp:211-220 with the upper bound N - 1).  usage: ransac_draws seed N iters subset calls
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

int main(int argc, char** argv) {
  if (argc != 6) return 2;
  const unsigned seed = (unsigned)std::strtoul(argv[1], nullptr, 0), n = (unsigned)std::atoi(argv[2]), iters = (unsigned)std::atoi(argv[3]),
                 subset = (unsigned)std::atoi(argv[4]), calls = (unsigned)std::atoi(argv[5]);
  std::default_random_engine eng;
  eng.seed(seed);
  for (unsigned c = 0; c < calls; ++c) {
    std::uniform_int_distribution<unsigned> pick(0, n - 1);  // constructed per call, as the wrapper does
    for (unsigned it = 0; it < iters; ++it) {
      std::vector<unsigned> s(subset);
      for (unsigned i = 0; i < subset; ++i) {
        bool again;
        do {
          s[i] = pick(eng);
          again = false;
          for (unsigned j = 0; j < i; ++j) again = again || s[j] == s[i];
        } while (again);
        std::printf("%u ", s[i]);
      }
    }
    std::printf("\n");
  }
  return 0;
}
