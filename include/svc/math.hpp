// math.hpp -- the vector PODs that appear in the boundary's signatures.
//
// Only what the hot path's signatures need: Vec2f / Vec2i / Vec2ui with the layout of
// reference libs/math.hpp:115-185 (two 4-byte members, no padding; Vec2f is what
// `Vec2f* motion_field` points at).  When a translation unit already includes the
// reference's own math.hpp, that definition is used instead (same names, same
// layout, same mangling), so this header stays out of its way.
#ifndef SVC_MATH_HPP
#define SVC_MATH_HPP

#include "types.hpp"

#ifndef SCALABLE_VIDEO_CODEC_MATH_HPP
struct Vec2i {
  int x;
  int y;
};

struct Vec2ui {
  uint x;
  uint y;
};

struct Vec2f {
  float x;
  float y;
};
#endif  // SCALABLE_VIDEO_CODEC_MATH_HPP

static_assert(sizeof(Vec2f) == 8, "Vec2f must be two packed floats");

#endif  // SVC_MATH_HPP
