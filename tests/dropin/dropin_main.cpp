// dropin_main.cpp -- a caller written against the reference's API only.
//
// It includes "motion.hpp" and nothing from this repo.  build.py compiles it twice:
//   dropin_ref_hdr : -I/root/reference/libs  (the reference's OWN header, when present)
//   dropin_own_hdr : -Iinclude/svc           (this repo's re-declaration)
// and links both against libsvc_motion.so instead of the reference's `motion` library --
// what a maintainer does to switch apps/encoder.cpp over.  The calls
// mirror libs/encoder.cpp:472-498.
//
// usage: dropin <in.bin> <out.bin>
//   in : u32 levels, w, h, search_range, block_w, block_h, ransac_n; f32 thresh, p, w;
//        then tracked planes (level 0..L-1), then anchor planes.
//   out: f32 mv[blocks][2], f32 min_mad[blocks], f32 gm[2], f32 rmse, u32 n_inliers, u32 inliers[]
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "motion.hpp"

int main(int argc, char** argv) {
  if (argc != 3) return 2;
  FILE* f = std::fopen(argv[1], "rb");
  if (!f) return 3;
  uint hdr[7];
  float rp[3];
  if (std::fread(hdr, 4, 7, f) != 7 || std::fread(rp, 4, 3, f) != 3) return 4;
  const uint levels = hdr[0], w = hdr[1], h = hdr[2], range = hdr[3], bw = hdr[4], bh = hdr[5];
  std::vector<std::vector<uchar>> trk(levels), anc(levels);
  for (int side = 0; side < 2; ++side)
    for (uint l = 0; l < levels; ++l) {
      auto& p = side == 0 ? trk[l] : anc[l];
      p.resize(static_cast<size_t>(w >> l) * (h >> l));
      if (std::fread(p.data(), 1, p.size(), f) != p.size()) return 5;
    }
  std::fclose(f);
  std::vector<const uchar*> tp(levels), ap(levels);
  for (uint l = 0; l < levels; ++l) { tp[l] = trk[l].data(); ap[l] = anc[l].data(); }

  const uint blocks = (w / bw) * (h / bh);
  std::vector<Vec2f> mv(blocks);
  std::vector<float> mad(blocks);
#if defined(__SSE2__) && defined(DROPIN_USE_SSE2_ENTRY)
  EstimateMotionHierarchical16x16Sse2(tp.data(), ap.data(), w, h, range, mv.data(), mad.data());
#else
  EstimateMotionHierarchical(tp.data(), ap.data(), levels, w, h, range, bw, bh, mv.data(), mad.data());
#endif

  RansacParams params;
  params.subset_sz = hdr[6];
  params.inlier_thresh = rp[0];
  params.success_prob = rp[1];
  params.inlier_ratio = rp[2];
  Vec2f gm{0.f, 0.f};
  float rmse = 0.f;
  std::vector<uint> inliers;
  EstimateGlobalMotionRansac(mv.data(), static_cast<uint>(mv.size()), params, &rmse, &gm, &inliers);

  FILE* o = std::fopen(argv[2], "wb");
  if (!o) return 6;
  std::fwrite(mv.data(), sizeof(Vec2f), blocks, o);
  std::fwrite(mad.data(), 4, blocks, o);
  std::fwrite(&gm, sizeof(Vec2f), 1, o);
  std::fwrite(&rmse, 4, 1, o);
  const uint n = static_cast<uint>(inliers.size());
  std::fwrite(&n, 4, 1, o);
  std::fwrite(inliers.data(), 4, n, o);
#ifdef SVC_MOTION_HPP
  // Only with this repo's header: the additions that have no counterpart in the reference's
  // motion.hpp.  A seeded RANSAC run (must repeat exactly), then Dct + QuantizeDequantize on a
  // fixed 48 x 32 pattern.
  float rmse2[2];
  Vec2f gm2[2];
  std::vector<uint> inl2[2];
  for (int k = 0; k < 2; ++k) {
    SvcSeedRansac(12345u);
    gm2[k] = Vec2f{0.f, 0.f};
    EstimateGlobalMotionRansac(mv.data(), static_cast<uint>(mv.size()), params, &rmse2[k], &gm2[k], &inl2[k]);
  }
  const uint same = (gm2[0].x == gm2[1].x && gm2[0].y == gm2[1].y && rmse2[0] == rmse2[1] && inl2[0] == inl2[1]) ? 1u : 0u;
  std::fwrite(&same, 4, 1, o);
  const uint dw = 48, dh = 32;
  std::vector<uchar> bgr(dw * dh * 3);
  for (uint y = 0; y < dh; ++y)
    for (uint x = 0; x < dw; ++x)
      for (uint c = 0; c < 3; ++c) bgr[(y * dw + x) * 3 + c] = static_cast<uchar>((x * 7 + y * 13 + c * 29) & 255);
  std::vector<float> planes(3 * dw * dh);
  float* pp[3] = {planes.data(), planes.data() + dw * dh, planes.data() + 2 * dw * dh};
  Dct(bgr.data(), dw, dh, 8, 8, pp);
  std::fwrite(planes.data(), 4, planes.size(), o);
  QuantizeDequantize(planes.data(), planes.size(), 640);
  std::fwrite(planes.data(), 4, planes.size(), o);
#endif
  std::fclose(o);
  return 0;
}
