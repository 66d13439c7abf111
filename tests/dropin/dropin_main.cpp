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
  std::fclose(o);
  return 0;
}
