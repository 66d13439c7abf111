// stream_main.cpp -- a host application written against include/svc/stream_encoder.hpp only:
// reads a raw B,G,R clip, encodes it through svc::StreamEncoder, writes every output in clip
// order.  tests/test_gpu_stream.py compares the files with the resident Python path.
//   stream_main <clip.raw> <w> <h> <frames> <levels> <dct_block> <wire 0|1> <batch> <seed> <out_prefix>
// out_prefix "-": no output files, only the PCIe-inclusive rate (bench.py's end_to_end.stream_encoder_fps).
#include <chrono>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <string>
#include <vector>

#include "svc/stream_encoder.hpp"

int main(int argc, char** argv) {
  if (argc != 11) { std::fprintf(stderr, "usage: see the header comment\n"); return 2; }
  const uint32_t w = std::atoi(argv[2]), h = std::atoi(argv[3]), n = std::atoi(argv[4]);
  svc::StreamEncoderConfig cfg;
  cfg.width = w; cfg.height = h;
  cfg.levels = std::atoi(argv[5]);
  cfg.dct_block = std::atoi(argv[6]);
  cfg.wire = std::atoi(argv[7]) != 0;
  cfg.batch = std::atoi(argv[8]);
  cfg.seed = std::strtoull(argv[9], nullptr, 10);
  const std::string prefix = argv[10];

  std::vector<uint8_t> clip((size_t)w * h * 3 * n);
  FILE* f = std::fopen(argv[1], "rb");
  if (!f || std::fread(clip.data(), 1, clip.size(), f) != clip.size()) { std::fprintf(stderr, "cannot read %s\n", argv[1]); return 1; }
  std::fclose(f);

  const bool files = prefix != "-";
  const char* sink_path = files ? nullptr : "/dev/null";
  FILE* f_mv = std::fopen(files ? (prefix + ".mv").c_str() : sink_path, "wb");
  FILE* f_ty = std::fopen(files ? (prefix + ".types").c_str() : sink_path, "wb");
  FILE* f_gm = std::fopen(files ? (prefix + ".gm").c_str() : sink_path, "wb");
  FILE* f_big = std::fopen(files ? (prefix + ".big").c_str() : sink_path, "wb");
  if (!f_mv || !f_ty || !f_gm || !f_big) { std::fprintf(stderr, "cannot open outputs under %s\n", prefix.c_str()); return 1; }
  try {
    svc::StreamEncoder enc(cfg);
    uint32_t next = 1, total = 0;
    bool dump = files;
    // the documented lifetime: a delivered view stays valid until depth - 2 = 1 more batch has been delivered
    const float* held = nullptr;
    std::vector<float> held_copy;
    auto sink = [&](const svc::EncodedBatch& b) {
      if (b.first_frame != next) { std::fprintf(stderr, "batch out of order: %u, expected %u\n", b.first_frame, next); std::exit(1); }
      if (held && std::memcmp(held, held_copy.data(), held_copy.size() * sizeof(float)) != 0) {
        std::fprintf(stderr, "the previous batch's view changed before its lifetime ended\n"); std::exit(1);
      }
      held = b.mv_xy;
      held_copy.assign(b.mv_xy, b.mv_xy + (size_t)b.count * b.mv_field_w * b.mv_field_h * 2);
      if ((b.header != nullptr) != (cfg.wire && b.first_frame == 1)) { std::fprintf(stderr, "header on the wrong batch\n"); std::exit(1); }
      if (b.header && dump && files) {
        FILE* f_h = std::fopen((prefix + ".hdr").c_str(), "wb");
        if (f_h) { std::fwrite(b.header, sizeof(*b.header), 1, f_h); std::fclose(f_h); }
      }
      next += b.count; total += b.count;
      if (!dump) return;
      const size_t blocks = (size_t)b.mv_field_w * b.mv_field_h;
      std::fwrite(b.mv_xy, sizeof(float), b.count * blocks * 2, f_mv);
      std::fwrite(b.block_types, sizeof(uint32_t), b.count * blocks, f_ty);
      std::fwrite(b.global_motion, sizeof(float), b.count * 2, f_gm);
      if (b.coeffs) std::fwrite(b.coeffs, sizeof(float), (size_t)b.count * 3 * b.padded_w * b.padded_h, f_big);
      else std::fwrite(b.records, 1, (size_t)b.count * b.record_bytes, f_big);
    };
    enc.Encode(clip.data(), n, sink);
    if (total != n - 1) { std::fprintf(stderr, "%u encoded frames, expected %u\n", total, n - 1); return 1; }
    // the same clip again without the file writes, as often as it takes to fill a second: PCIe-inclusive rate of the schedule itself,
    // and where its time went (the encoder's own clocks, summed over the passes)
    dump = false;
    uint32_t passes = 0, frames = 0;
    svc::EncodeStats sum;
    double d2h_by_pass[64] = {};  // GB/s of each pass's D2H copies: does the link's rate move over the process's first second?
    const auto t0 = std::chrono::steady_clock::now();
    double s = 0;
    do {
      next = 1; total = 0; held = nullptr;
      enc.Encode(clip.data(), n, sink);
      const svc::EncodeStats& e = enc.last_stats();
      sum.batches += e.batches; sum.wall_ms += e.wall_ms; sum.staging_ms += e.staging_ms; sum.slot_wait_ms += e.slot_wait_ms;
      sum.deliver_wait_ms += e.deliver_wait_ms; sum.sink_ms += e.sink_ms; sum.h2d_ms += e.h2d_ms; sum.kernels_ms += e.kernels_ms;
      sum.d2h_ms += e.d2h_ms; sum.h2d_bytes += e.h2d_bytes; sum.d2h_bytes += e.d2h_bytes;
      sum.copy_threads = e.copy_threads; sum.host_cores = e.host_cores;
      if (passes < 64) d2h_by_pass[passes] = e.d2h_bytes / (e.d2h_ms * 1e6);
      ++passes; frames += total;
      s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    } while (s < 1.0 && passes < 64);
    total = n - 1;
    std::printf("d2h_GBps_by_pass");
    for (uint32_t q = 0; q < passes; ++q) std::printf(" %.1f", d2h_by_pass[q]);
    std::printf("\n");
    std::printf("%u encoded frames, %.0f frames/s PCIe-inclusive (second pass)\n", frames, frames / s);
    std::printf("phases {\"passes\": %u, \"batches\": %u, \"seconds\": %.4f, \"host_cores\": %u, \"copy_threads\": %u, "
                "\"host_ms_per_batch\": {\"staging\": %.3f, \"slot_wait\": %.3f, \"deliver_wait\": %.3f, \"sink\": %.3f, \"wall\": %.3f}, "
                "\"device_ms_per_batch\": {\"h2d\": %.3f, \"kernels\": %.3f, \"d2h\": %.3f}, \"h2d_GBps\": %.2f, \"d2h_GBps\": %.2f}\n",
                passes, sum.batches, s, sum.host_cores, sum.copy_threads, sum.staging_ms / sum.batches, sum.slot_wait_ms / sum.batches,
                sum.deliver_wait_ms / sum.batches, sum.sink_ms / sum.batches, sum.wall_ms / sum.batches, sum.h2d_ms / sum.batches,
                sum.kernels_ms / sum.batches, sum.d2h_ms / sum.batches, sum.h2d_bytes / (sum.h2d_ms * 1e6), sum.d2h_bytes / (sum.d2h_ms * 1e6));
    // ... and as ONE stream of the same length (the clip's frames cycled through the Source callback in a single Encode call): the passes
    // above fill and drain the pipeline once per 65-frame clip, a stream does so once -- what is left per batch is the schedule's steady state
    {
      const uint32_t n_long = frames + 1;
      const size_t frame_bytes = (size_t)w * h * 3;
      uint32_t i = 0;
      next = 1; total = 0; held = nullptr;
      const auto t1 = std::chrono::steady_clock::now();
      enc.Encode([&]() -> const uint8_t* { return i < n_long ? clip.data() + (size_t)(i++ % n) * frame_bytes : nullptr; }, n_long, sink);
      const double sl = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
      const svc::EncodeStats& e = enc.last_stats();
      if (total != n_long - 1) { std::fprintf(stderr, "%u encoded frames, expected %u\n", total, n_long - 1); return 1; }
      std::printf("long_stream {\"encoded_frames\": %u, \"frames_per_s\": %.0f, \"batches\": %u, \"seconds\": %.4f, "
                  "\"host_ms_per_batch\": {\"staging\": %.3f, \"slot_wait\": %.3f, \"deliver_wait\": %.3f, \"sink\": %.3f, \"wall\": %.3f}, "
                  "\"device_ms_per_batch\": {\"h2d\": %.3f, \"kernels\": %.3f, \"d2h\": %.3f}, \"h2d_GBps\": %.2f, \"d2h_GBps\": %.2f}\n",
                  total, total / sl, e.batches, sl, e.staging_ms / e.batches, e.slot_wait_ms / e.batches, e.deliver_wait_ms / e.batches,
                  e.sink_ms / e.batches, e.wall_ms / e.batches, e.h2d_ms / e.batches, e.kernels_ms / e.batches, e.d2h_ms / e.batches,
                  e.h2d_bytes / (e.h2d_ms * 1e6), e.d2h_bytes / (e.d2h_ms * 1e6));
      total = n - 1;
    }
  } catch (const std::exception& e) {
    std::fprintf(stderr, "%s\n", e.what());
    return 1;
  }
  std::fclose(f_mv); std::fclose(f_ty); std::fclose(f_gm); std::fclose(f_big);
  return 0;
}
