// asan_main.cpp -- the host side under AddressSanitizer + UndefinedBehaviorSanitizer (tests/sanitize/Makefile).
//
// Part 1 walks the oracle (test infrastructure) over exactly-sized heap buffers: an out-of-range window, a sample
// index past the field or a tile loop past a plane would be a sanitizer report.  Part 2 calls every entry point of
// the C ABI (include/svc_hip.h) and of the C++ layer with precondition violations and, where no GPU is visible,
// with valid arguments too: each must come back with a status (never a launch, never UB), and the message buffer,
// the staging logic in front of the device check and the shard planner run instrumented.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <vector>

#include "svc/clip_encoder.hpp"
#include "svc/motion.hpp"
#include "svc/stream_encoder.hpp"
#include "svc_clip.h"
#include "svc_hip.h"
#include "svc_oracle.h"

namespace {

int g_checks = 0;
#define CHECK(cond)                                                          \
  do {                                                                       \
    ++g_checks;                                                              \
    if (!(cond)) {                                                           \
      std::fprintf(stderr, "CHECK failed: %s (%s:%d)\n", #cond, __FILE__, __LINE__); \
      std::exit(2);                                                          \
    }                                                                        \
  } while (0)

uint32_t g_lcg = 12345;
uint32_t Rnd() { g_lcg = g_lcg * 1664525u + 1013904223u; return g_lcg >> 8; }

std::vector<uint8_t> Plane(uint32_t w, uint32_t h) {
  std::vector<uint8_t> p((size_t)w * h);
  for (auto& v : p) v = (uint8_t)Rnd();
  return p;
}

void OraclePart() {
  // motion search: every level count, a non-square block, ranges that clamp at every border
  const struct { uint32_t levels, w, h, r, bw, bh; } shapes[] = {
      {1, 48, 32, 8, 16, 16}, {2, 64, 64, 8, 16, 16}, {3, 64, 64, 8, 16, 16}, {4, 128, 64, 8, 16, 16},
      {1, 18, 20, 3, 6, 10},  {2, 32, 16, 30, 16, 8}, {3, 64, 32, 4, 8, 8}};
  for (const auto& s : shapes) {
    std::vector<std::vector<uint8_t>> t, a;
    std::vector<const uint8_t*> tp, ap;
    for (uint32_t l = 0; l < s.levels; ++l) {
      t.push_back(Plane(s.w >> l, s.h >> l));
      a.push_back(Plane(s.w >> l, s.h >> l));
    }
    for (uint32_t l = 0; l < s.levels; ++l) { tp.push_back(t[l].data()); ap.push_back(a[l].data()); }
    const size_t blocks = (size_t)(s.w / s.bw) * (s.h / s.bh);
    std::vector<svc_oracle_vec2f> mv(blocks);
    std::vector<float> mad(blocks);
    CHECK(svc_oracle_hbma(tp.data(), ap.data(), s.levels, s.w, s.h, s.r, s.bw, s.bh, mv.data(), mad.data()) == 0);
    if (s.levels == 4 && s.bw == 16)
      CHECK(svc_oracle_hbma16_sse2(tp.data(), ap.data(), s.w, s.h, s.r, mv.data(), mad.data()) == 0);
    svc_oracle_ebma(t[0].data(), a[0].data(), s.w, s.h, s.r, s.bw, s.bh, mv.data(), mad.data());
    svc_oracle_vec2f gm;
    float mm;
    svc_oracle_global_ebma(t[0].data(), a[0].data(), s.w, s.h, s.r < s.w && s.r < s.h ? s.r : 1, 0, &gm, &mm);
    svc_oracle_global_ebma(t[0].data(), a[0].data(), s.w, s.h, 2, 1, &gm, &mm);
    CHECK(gm.x == 0.0f && gm.y == 0.0f);  // the reference's literal loops never run
    svc_oracle_global_hbma(tp.data(), ap.data(), s.levels, s.w, s.h, 1u << (s.levels - 1), 0, &gm);
  }
  // RANSAC on a field of exactly n entries, samples in [0, n - 1]; both exits (consensus / none)
  for (uint32_t n : {3u, 64u, 1000u}) {
    for (uint32_t subset : {1u, 3u}) {
      for (int scatter = 0; scatter < 2; ++scatter) {
        std::vector<svc_oracle_vec2f> mv(n);
        for (uint32_t i = 0; i < n; ++i)
          mv[i] = scatter ? svc_oracle_vec2f{(float)(i * 50), (float)(i * 70)} : svc_oracle_vec2f{(float)(Rnd() % 3), 1.0f};
        svc_oracle_ransac_params p{subset, 7.5f, 0.99f, 0.5f};
        const uint32_t k = svc_oracle_ransac_iter_count(p);
        std::vector<uint32_t> samples((size_t)k * subset), inl(n);
        for (uint32_t it = 0; it < k; ++it)
          for (uint32_t i = 0; i < subset; ++i) samples[(size_t)it * subset + i] = (Rnd() % (n - subset + 1)) + i;
        float rmse = 0;
        svc_oracle_vec2f gm{0.25f, -0.75f};
        uint32_t cnt = 0;
        svc_oracle_ransac(mv.data(), n, p, samples.data(), k, &rmse, &gm, inl.data(), &cnt);
        CHECK(cnt <= n);
        std::vector<uint8_t> mask(n);
        svc_oracle_fg_mask(inl.data(), cnt, n, mask.data());
        svc_oracle_vec2f avg;
        svc_oracle_global_avg(mv.data(), n, &avg);
      }
    }
  }
  // segmentation glue, transform, quant, wire, decode on exactly-sized buffers
  {
    const uint32_t mfw = 12, mfh = 7, n = mfw * mfh;
    std::vector<uint8_t> mask(n);
    std::vector<svc_oracle_vec2f> mv(n);
    for (uint32_t i = 0; i < n; ++i) { mask[i] = (Rnd() % 3) != 0; mv[i] = {(float)(Rnd() % 7) - 3, (float)(Rnd() % 5)}; }
    std::vector<uint32_t> types(n);
    for (uint32_t conn : {4u, 8u})
      CHECK(svc_oracle_segment(mask.data(), mv.data(), mfw, mfh, 16, 16, 3, 3, 10, 3, 10, 1.0f, conn, 99, types.data()) == 0);
    const uint32_t w = mfw * 16, h = mfh * 16;
    std::vector<uint8_t> bgr((size_t)w * h * 3);
    for (auto& v : bgr) v = (uint8_t)Rnd();
    for (auto blk : {std::pair<uint32_t, uint32_t>{8, 8}, {16, 16}, {4, 4}, {16, 8}, {2, 16}}) {
      std::vector<double> p64((size_t)3 * w * h);
      std::vector<float> p32((size_t)3 * w * h);
      svc_oracle_dct_frame_f64(bgr.data(), w, h, blk.first, blk.second, p64.data());
      svc_oracle_dct_frame_f32(bgr.data(), w, h, blk.first, blk.second, p32.data());
      svc_oracle_quant_frame(p32.data(), w, h, 16, 16, types.data(), 1, 640);
      if (blk.first == blk.second) {
        std::vector<uint8_t> rec((size_t)(w / blk.first) * (h / blk.second) * (4 + 12 * blk.first * blk.second));
        const uint64_t got = svc_oracle_serialize_frame(p32.data(), (uint64_t)w * h, 3, types.data(), w, h, blk.first, blk.second, mfw, 16, 16, rec.data());
        CHECK(got == rec.size());
        std::vector<double> out((size_t)w * h * 3);
        svc_oracle_decode_frame(p32.data(), w, h, blk.first, blk.second, types.data(), 16, 16, 1, 640, 8, 8, 64, 32, out.data());
        std::vector<float> recf(out.begin(), out.end());
        (void)svc_oracle_sse_frame(bgr.data(), recf.data(), w, w - 3, h - 5);
      }
    }
    std::vector<float> c(1001);
    for (auto& v : c) v = (float)(Rnd() % 4000) - 2000.0f;
    svc_oracle_quant(c.data(), c.size(), 640);
    // the per-call statements (oracle/svc_imageops.c) and the CPU baseline's transform (oracle/svc_cpu_dct.c), exactly-sized buffers
    std::vector<uint8_t> yuv(bgr.size());
    svc_oracle_bgr2yuv(bgr.data(), w, h, yuv.data());
    std::vector<uint8_t> img(n), out8(n);
    for (auto& v : img) v = (Rnd() % 3) ? 255 : 0;
    for (uint32_t op = 0; op < 4; ++op) svc_oracle_morph_rect(img.data(), mfw, mfh, 3, 2, op, out8.data());
    svc_oracle_morph_rect(img.data(), mfw, mfh, 9, 9, 3, img.data());  // in place, element larger than the image
    std::vector<int32_t> lab(n);
    for (uint32_t conn : {4u, 8u}) CHECK(svc_oracle_connected_components(img.data(), mfw, mfh, conn, lab.data()) >= 1);
    std::vector<float> feats((size_t)n * 4, 0.0f);
    for (uint32_t i = 0; i < n; ++i) { feats[4 * i + 1] = (float)(Rnd() % 17) - 8; feats[4 * i + 2] = (float)((i % mfw) * 16); feats[4 * i + 3] = (float)((i / mfw) * 16); }
    double compact = 0;
    CHECK(svc_oracle_kmeans(feats.data(), n, 4, 10, 3, 10, 1.0f, 7, lab.data(), &compact) == 0 && compact >= 0);
    CHECK(svc_oracle_kmeans(feats.data(), n, 4, n, 1, 2, 1.0f, 7, lab.data(), nullptr) == (n <= 255 ? 0 : 1));  // as many clusters as points
    feats[5] = 0.5f;
    CHECK(svc_oracle_kmeans(feats.data(), n, 4, 3, 1, 2, 1.0f, 7, lab.data(), nullptr) == 1);  // not an integer: outside the definition
    for (uint32_t blk : {8u, 16u}) {
      std::vector<float> p32((size_t)3 * w * h);
      CHECK(svc_cpu_dct_frame_f32(bgr.data(), w, h, blk, p32.data()) == 0);
      svc_cpu_quant_frame_f32(p32.data(), w, h, 16, 16, types.data(), 3, 640);
    }
    std::vector<float> p32((size_t)3 * w * h);
    CHECK(svc_cpu_dct_frame_f32(bgr.data(), w, h, 4, p32.data()) == 1 && svc_cpu_dct_isa() >= 0);
  }
}

void AbiPart() {
  int ndev = -1;
  CHECK(svc_hip_device_count(&ndev) == SVC_OK && ndev >= 0);
  CHECK(svc_hip_device_count(nullptr) == SVC_ERR_INVALID_ARG && std::strlen(svc_hip_last_error()) > 0);
  CHECK(svc_hip_pyramid_bytes(1920, 1088, 3) == 2741760);
  CHECK(svc_hip_ransac_iter_count(svc_ransac_params{1, 7.5f, 0.99f, 0.5f}) == 7);
  CHECK(svc_hip_serialized_frame_bytes(1920, 1088, 8, 8) == (uint64_t)240 * 136 * 772);
  svc_wire_header hdr;
  CHECK(svc_hip_wire_header(300, 1920, 1080, 16, 16, 3, 8, 8, &hdr) == SVC_OK && hdr.frame_count == 299 && hdr.frame_excess_h == 8);
  CHECK(svc_hip_wire_header(300, 1920, 1080, 16, 16, 3, 8, 8, nullptr) == SVC_ERR_INVALID_ARG);

  const bool no_gpu = ndev == 0;
  std::vector<uint8_t> plane = Plane(64, 64), bgr((size_t)64 * 64 * 3);
  const uint8_t* pyr1[1] = {plane.data()};
  std::vector<float> mv(2 * 16), mad(16), planes((size_t)3 * 64 * 64);
  // the reference's asserts (libs/motion.cpp:417-433) as statuses, before any device work
  CHECK(svc_hip_hbma_host(pyr1, pyr1, 1, 64, 64, 8, 16, 24, mv.data(), mad.data(), 0) == SVC_ERR_INVALID_ARG);
  CHECK(svc_hip_hbma_host(pyr1, pyr1, 1, 64, 64, 8, 0, 16, mv.data(), mad.data(), 0) == SVC_ERR_INVALID_ARG);
  CHECK(svc_hip_hbma_host(nullptr, pyr1, 1, 64, 64, 8, 16, 16, mv.data(), mad.data(), 0) == SVC_ERR_INVALID_ARG);
  CHECK(svc_hip_hbma_host(pyr1, pyr1, 3, 64, 64, 2, 16, 16, mv.data(), mad.data(), 0) == SVC_ERR_INVALID_ARG);
  CHECK(svc_hip_ebma_host(plane.data(), plane.data(), 64, 60, 8, 16, 16, mv.data(), mad.data()) == SVC_ERR_INVALID_ARG);
  CHECK(svc_hip_dct_host(bgr.data(), 64, 64, 3, 4, planes.data()) == SVC_ERR_INVALID_ARG);
  CHECK(svc_hip_dct_host(bgr.data(), 64, 62, 8, 8, planes.data()) == SVC_ERR_INVALID_ARG);
  CHECK(svc_hip_quant_host(planes.data(), 16, 0) == SVC_ERR_INVALID_ARG);
  CHECK(svc_hip_global_ebma_host(plane.data(), plane.data(), 64, 64, 64, mv.data(), mad.data()) == SVC_ERR_INVALID_ARG);
  CHECK(svc_hip_global_hbma_host(pyr1, pyr1, 0, 64, 64, 8, mv.data()) == SVC_ERR_INVALID_ARG);
  uint32_t bad_sample[1] = {16}, inl[16], cnt = 0;
  float gm[2] = {0, 0}, rmse = 0;
  CHECK(svc_hip_ransac_host(mv.data(), 16, svc_ransac_params{1, 7.5f, 0.99f, 0.5f}, bad_sample, 1, gm, &rmse, inl, &cnt) == SVC_ERR_INVALID_ARG);
  // device-pointer entry points: null / misaligned / inconsistent arguments never reach a launch
  CHECK(svc_hip_hbma_pairs(nullptr, nullptr, 0, 1, 3, 64, 64, 8, 16, 16, nullptr, nullptr, 0, nullptr) == SVC_ERR_INVALID_ARG);
  CHECK(svc_hip_hbma_pairs(nullptr, nullptr, 0, 0, 3, 64, 64, 8, 16, 16, nullptr, nullptr, 0, nullptr) == SVC_OK);  // empty batch
  CHECK(svc_hip_dct_quant_frames(nullptr, 0, 1, 64, 64, 8, 8, nullptr, 16, 16, 1, 640, nullptr, nullptr) == SVC_ERR_INVALID_ARG);
  CHECK(svc_hip_dct_records_frames(nullptr, 0, 1, 64, 64, 8, nullptr, 16, 16, 1, 0, 64, nullptr, 0, nullptr) == SVC_ERR_INVALID_ARG);
  CHECK(svc_hip_global_ebma_pairs(nullptr, nullptr, 0, 1, 64, 64, 4, nullptr, 0, nullptr, nullptr, nullptr) == SVC_ERR_INVALID_ARG);
  CHECK(svc_hip_halo_shift(nullptr, nullptr, nullptr, 16, 2, 2, 0, nullptr) == SVC_ERR_INVALID_ARG);
  CHECK(svc_hip_halo_shift(nullptr, nullptr, nullptr, 16, 0, 1, 0, nullptr) == SVC_OK);  // a single rank has no neighbour
  CHECK(svc_hip_comm_create(nullptr, 0, 1, nullptr) == SVC_ERR_INVALID_ARG);
  if (no_gpu) {
    // valid arguments, no device: a loud status, no fallback
    CHECK(svc_hip_hbma_host(pyr1, pyr1, 1, 64, 64, 8, 16, 16, mv.data(), mad.data(), 0) == SVC_ERR_NO_DEVICE);
    CHECK(svc_hip_dct_host(bgr.data(), 64, 64, 8, 8, planes.data()) == SVC_ERR_NO_DEVICE);
    CHECK(svc_hip_global_avg_host(mv.data(), 16, gm) == SVC_ERR_NO_DEVICE);
    CHECK(std::strstr(svc_hip_last_error(), "no CPU path") != nullptr);
  }

  // C++ layer: the shard planner over every small clip / world, and constructors that must throw, not crash
  for (uint32_t frames = 1; frames <= 40; ++frames)
    for (uint32_t world = 1; world <= 9; ++world) {
      uint32_t covered = 0, pairs = 0, next = 0;
      for (uint32_t r = 0; r < world; ++r) {
        const svc::Shard s = svc::PlanShard(frames, world, r);
        CHECK(s.first_frame == next || s.frames == 0);
        next += s.frames; covered += s.frames; pairs += s.pairs;
        CHECK(s.frames == 0 || s.needs_halo == (s.first_frame > 0));
      }
      CHECK(covered == frames && pairs == frames - 1);
    }
  uint32_t a, b, c, d;
  CHECK(svc_clip_plan_shard(300, 8, 8, &a, &b, &c, &d) != 0 && std::strlen(svc_clip_last_error()) > 0);
  svc_clip* h = nullptr;
  svc_clip_config cc{};
  CHECK(svc_clip_create(&cc, &h) != 0 && h == nullptr);  // struct_size 0: a caller built against another layout
  CHECK(std::strstr(svc_clip_last_error(), "struct_size") != nullptr);
  cc.struct_size = sizeof(cc);
  cc.tuning = 1u << 20;
  CHECK(svc_clip_create(&cc, &h) != 0 && h == nullptr && std::strstr(svc_clip_last_error(), "tuning") != nullptr);
  cc.tuning = 0; cc.lat_depth = 4;
  CHECK(svc_clip_create(&cc, &h) != 0 && h == nullptr && std::strstr(svc_clip_last_error(), "lat_depth") != nullptr);
  cc.lat_depth = 0; cc.hbma_flags = 64;
  CHECK(svc_clip_create(&cc, &h) != 0 && h == nullptr && std::strstr(svc_clip_last_error(), "hbma_flags") != nullptr);
  cc.hbma_flags = 0;
  CHECK(svc_clip_create(&cc, &h) != 0 && h == nullptr);  // all-zero configuration: rejected by the encoder itself
  CHECK(svc_clip_create(nullptr, &h) != 0);
  if (no_gpu) {
    svc::ClipEncoderConfig k;
    k.width = 64; k.height = 48; k.clip_frames = 4;
    bool threw = false;
    try { svc::ClipEncoder enc(k); } catch (const std::runtime_error&) { threw = true; }
    CHECK(threw);
    svc::StreamEncoderConfig sc;
    sc.width = 64; sc.height = 48;
    threw = false;
    try { svc::StreamEncoder enc(sc); } catch (const std::runtime_error&) { threw = true; }
    CHECK(threw);
  }
  svc::StreamEncoderConfig shallow;
  shallow.width = 64; shallow.height = 48; shallow.depth = 2;
  bool threw = false;
  try { svc::StreamEncoder enc(shallow); } catch (const std::runtime_error&) { threw = true; }
  CHECK(threw);  // depth < 3 leaves nothing overlapped (stream_encoder.hpp)
  const Vec2f none = EstimateGlobalMotionAvg(nullptr, 0);
  CHECK(none.x == 0.0f && none.y == 0.0f);
}

}  // namespace

int main() {
  OraclePart();
  AbiPart();
  std::printf("sanitize: %d checks passed\n", g_checks);
  return 0;
}
