// stream_fuzz.cpp -- random use of svc::StreamEncoder (include/svc/stream_encoder.hpp) against itself.
//
// The encoder's results are a function of the clip and the seed alone, "whatever the batch size" (stream_encoder.hpp): RANSAC draws and
// k-means seeds are indexed by the clip-wide frame.  So a clip must encode to the same bytes through ANY encoder configuration that differs
// only in how the work is cut and moved -- batch size, batches in flight, copy threads, pointer or Source entry point, a fresh encoder or one
// that has encoded other clips before (short last batches leave stale frames in its slots, a clip that ends on a batch boundary ends the
// loop another way, a two-frame clip is one short batch).  This program draws such configurations and clip orders at random and compares
// every encoded frame with the first result seen for (clip, output form); tests/test_gpu_stream.py ties one of those results to the
// resident path, and through it to the oracle.
//   stream_fuzz <w> <h> <levels> <dct_block> <encoders> <seed>      exit 0 = every result equal
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <tuple>
#include <vector>

#include "svc/stream_encoder.hpp"

namespace {

uint32_t Hash32(uint64_t x) {
  uint32_t v = (uint32_t)(x ^ (x >> 32));
  v ^= v >> 16; v *= 0x7FEB352Du;
  v ^= v >> 15; v *= 0x846CA68Bu;
  v ^= v >> 16;
  return v;
}

struct Rng {
  uint64_t s;
  uint32_t next() { s = s * 6364136223846793005ull + 1442695040888963407ull; return Hash32(s); }
  uint32_t below(uint32_t n) { return next() % n; }
};

// a textured background moving by (dx, dy) per frame with a block moving the other way on top: something for every stage to find
std::vector<uint8_t> MakeClip(uint32_t w, uint32_t h, uint32_t n, uint32_t id) {
  std::vector<uint8_t> c((size_t)w * h * 3 * n);
  const int dx = 1 + (int)(id % 3), dy = (int)(id % 2);
  for (uint32_t t = 0; t < n; ++t)
    for (uint32_t y = 0; y < h; ++y)
      for (uint32_t x = 0; x < w; ++x) {
        const int sx = (int)x + dx * (int)t, sy = (int)y + dy * (int)t;
        uint32_t v = (Hash32(((uint64_t)(sy / 4) << 20) ^ (uint64_t)(sx / 4) ^ ((uint64_t)id << 40)) & 0x7F) + 40 +
                     (Hash32(((uint64_t)sy << 20) ^ (uint64_t)sx ^ ((uint64_t)id << 44)) & 0x0F);
        const int bx = (int)(w / 3) - 2 * (int)t, by = (int)(h / 3) + (int)t;  // the foreground block
        if ((int)x >= bx && (int)x < bx + (int)w / 5 && (int)y >= by && (int)y < by + (int)h / 4)
          v = 200 + (Hash32(((uint64_t)(y - by) << 20) ^ (uint64_t)(x - bx)) & 0x1F);
        uint8_t* p = &c[(((size_t)t * h + y) * w + x) * 3];
        p[0] = (uint8_t)v; p[1] = (uint8_t)(v * 3 / 4 + 20); p[2] = (uint8_t)(255 - v);
      }
  return c;
}

struct Result {  // everything one encoded clip delivers, in clip order
  std::vector<float> mv, gm, coeffs;
  std::vector<uint32_t> types;
  std::vector<uint8_t> records, header;
  uint32_t frames = 0;
  bool operator==(const Result& o) const {
    auto eqf = [](const std::vector<float>& a, const std::vector<float>& b) {
      return a.size() == b.size() && (a.empty() || std::memcmp(a.data(), b.data(), a.size() * sizeof(float)) == 0);
    };
    return frames == o.frames && eqf(mv, o.mv) && eqf(gm, o.gm) && eqf(coeffs, o.coeffs) && types == o.types && records == o.records && header == o.header;
  }
};

}  // namespace

int main(int argc, char** argv) {
  if (argc != 7) { std::fprintf(stderr, "usage: stream_fuzz <w> <h> <levels> <dct_block> <encoders> <seed>\n"); return 2; }
  const uint32_t w = std::atoi(argv[1]), h = std::atoi(argv[2]), levels = std::atoi(argv[3]), dct = std::atoi(argv[4]);
  const uint32_t encoders = std::atoi(argv[5]);
  Rng rng{std::strtoull(argv[6], nullptr, 10) * 0x9E3779B97F4A7C15ull + 1};
  // clips of lengths around every batch boundary the draw below can produce, one of them the minimum of two frames
  const uint32_t lengths[] = {2, 3, 5, 8, 9, 13, 17};
  std::vector<std::vector<uint8_t>> clips;
  for (uint32_t i = 0; i < sizeof(lengths) / sizeof(lengths[0]); ++i) clips.push_back(MakeClip(w, h, lengths[i], i));
  std::map<std::tuple<uint32_t, int>, Result> golden;  // (clip, output form) -> the first result seen
  uint32_t compared = 0, encoded = 0;
  try {
    for (uint32_t e = 0; e < encoders; ++e) {
      svc::StreamEncoderConfig cfg;
      cfg.width = w; cfg.height = h; cfg.levels = levels; cfg.dct_block = dct;
      const int form = (int)rng.below(3);  // 0 planes, 1 records over the padded grid, 2 the reference encoder's stream
      cfg.wire = form != 0;
      cfg.reference_stream = form == 2;
      cfg.batch = 1 + rng.below(8);
      cfg.depth = 3 + rng.below(3);
      cfg.copy_threads = 1 + rng.below(4);
      cfg.seed = 12345;
      svc::StreamEncoder enc(cfg);
      const uint32_t uses = 1 + rng.below(5);
      std::string trace;
      for (uint32_t u = 0; u < uses; ++u) {
        const uint32_t ci = rng.below((uint32_t)clips.size()), n = lengths[ci];
        const bool by_source = rng.below(2) != 0;
        Result r;
        uint32_t next_frame = 1;
        bool order_ok = true, header_ok = true;
        auto sink = [&](const svc::EncodedBatch& b) {
          order_ok = order_ok && b.first_frame == next_frame && b.count >= 1 && b.count <= cfg.batch;
          header_ok = header_ok && ((b.header != nullptr) == (cfg.wire && b.first_frame == 1));
          next_frame += b.count;
          const size_t blocks = (size_t)b.mv_field_w * b.mv_field_h;
          r.mv.insert(r.mv.end(), b.mv_xy, b.mv_xy + b.count * blocks * 2);
          r.gm.insert(r.gm.end(), b.global_motion, b.global_motion + b.count * 2);
          r.types.insert(r.types.end(), b.block_types, b.block_types + b.count * blocks);
          if (b.coeffs) r.coeffs.insert(r.coeffs.end(), b.coeffs, b.coeffs + (size_t)b.count * 3 * b.padded_w * b.padded_h);
          if (b.records) r.records.insert(r.records.end(), b.records, b.records + (size_t)b.count * b.record_bytes);
          if (b.header) r.header.assign((const uint8_t*)b.header, (const uint8_t*)b.header + sizeof(*b.header));
          r.frames += b.count;
        };
        const size_t frame_bytes = (size_t)w * h * 3;
        if (by_source) {
          uint32_t i = 0;
          enc.Encode([&]() -> const uint8_t* { return i < n ? clips[ci].data() + (size_t)(i++) * frame_bytes : nullptr; }, n, sink);
        } else {
          enc.Encode(clips[ci].data(), n, sink);
        }
        ++encoded;
        trace += " clip" + std::to_string(ci) + (by_source ? "s" : "p");
        const char* bad = nullptr;
        if (r.frames != n - 1) bad = "frame count";
        else if (!order_ok) bad = "batch order";
        else if (!header_ok) bad = "header placement";
        else if (enc.last_stats().encoded_frames != n - 1) bad = "stats";
        else {
          auto key = std::make_tuple(ci, form);
          auto it = golden.find(key);
          if (it == golden.end()) golden.emplace(key, std::move(r));
          else { ++compared; if (!(it->second == r)) bad = "bytes differ from the first encoding of this clip"; }
        }
        if (bad) {
          std::printf("FAIL %s: form %d batch %u depth %u copy_threads %u, uses so far:%s\n", bad, form, cfg.batch, cfg.depth, cfg.copy_threads, trace.c_str());
          return 1;
        }
      }
      std::printf("ok   form %d batch %u depth %u copy_threads %u:%s\n", form, cfg.batch, cfg.depth, cfg.copy_threads, trace.c_str());
    }
  } catch (const std::exception& ex) {
    std::printf("FAIL exception: %s\n", ex.what());
    return 1;
  }
  // planes and the two record forms carry the same motion fields and region ids
  for (auto& kv : golden) {
    auto base = golden.find(std::make_tuple(std::get<0>(kv.first), 0));
    if (base != golden.end() && (base->second.mv != kv.second.mv || base->second.types != kv.second.types)) {
      std::printf("FAIL clip %u: output form %d disagrees with the planes form on motion field / region ids\n", std::get<0>(kv.first), std::get<1>(kv.first));
      return 1;
    }
  }
  std::printf("%u clips encoded by %u encoders, %u compared with the first encoding of the same clip and output form: all equal\n", encoded, encoders, compared);
  return 0;
}
