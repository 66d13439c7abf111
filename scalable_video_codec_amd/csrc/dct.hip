// dct.hip -- per-tile 2-D DCT-II (+ optional quantise/dequantise) of BGR u8 frames.
//
// Replaces the reference's static Dct (libs/encoder.cpp:323-339: cv::split, then an
// in-place cv::dct on every block_w x block_h ROI of every plane) and, fused behind
// it, the decoder's quant lines (libs/decoder.cpp:130-144).  cv::dct(flags = 0) is
// the orthonormal DCT-II, Y = C X C^T.
//
// Bound: HBM (3 B/pixel in, 12 B/pixel out; measured 5.45 TB/s at N = 8, 87 % of the chip's
// copy rate) -- but only after the arithmetic was trimmed: f64 issues at 4 cycles per wave
// instruction on gfx950, and the first version (f64 butterflies, IEEE f32 divide + roundf in the
// quantiser, ~60 issue cycles per coefficient) was VALU-bound with the VALU ~100 % busy.
// float64 is still the right precision: rounding once to f32 puts every coefficient within
// 1/2 ulp(f32) of the exact value, well inside the 1e-4 * max(1, |ref|) parity bar that plain
// f32 accumulation misses for small AC coefficients of bright tiles; the even/odd split of
// the basis halves the multiplies, and the butterflies of the row pass run in f32 (exact).
//
// Shape.  The unit of work is a "segment column": 16 pixels wide x N rows tall x 3
// channels (two 8x8 tiles or one 16x16 tile per channel).  N consecutive lanes own
// it, one ROW each: a lane loads its 48 contiguous bytes (16 BGR pixels), splits
// the channels in registers (v_cvt_f32_ubyteN), runs the 1-D row transforms, and
// parks them in LDS; the same N lanes then each take COLUMNS, read them back
// transposed, run the column transforms and store f32.  A segment column never
// leaves its wave, so there is no workgroup barrier anywhere -- only wave-local
// LDS ordering -- and one channel's 1-2 KiB LDS slab is reused for the next.
// LDS pitches (144 B rows; 1152 / 2432 B slabs) make both the row writes
// (ds_write_b128) and the column reads (ds_read_b128 / ds_read_b64) conflict-free.
// Stores: N = 8, float2 per lane = 512 contiguous bytes per wave instruction; N = 16, one float per lane = 256 contiguous bytes -- which is
// NOT what holds N = 16 at 0.65 of the HBM peak against N = 8's 0.76: a 16 x 16 form with eight segment columns per wave, two adjacent
// columns per lane and float2 stores (512 bytes per instruction; dct16_wide_kernel, commit e2da6a1) measures level with this one at C5
// (1.502-1.513 against 1.507-1.516 ms serial, profiles/r06_ab_dct16_wide.txt) and was removed.  N = 16 is bound by its f64 arithmetic:
// 600 f64 instructions per lane and segment row (420 FMA + 90 MUL + 90 ADD) against 348 at N = 8, see DESIGN.md 4.2.
#include <algorithm>

#include "luma16.hpp"
#include "svc_common.hpp"

namespace svc {

#include "dct_tables.inc"

struct DctArgs {
  const uint8_t* bgr;
  uint64_t frame_stride;
  uint32_t w, h;
  uint32_t segs_per_band;    // W / 16
  uint32_t bands_per_frame;  // H / N
  uint32_t total_segcols;    // frames * bands * segs
  float* planes;
  // quant
  const uint32_t* types;
  uint32_t mv_bw, mv_bh, mfw, mv_blocks;
  float fg_step, bg_step;
  float fg_inv, bg_inv;  // RN(1 / step), computed on the host
  // wire output (records of libs/encoder.cpp:222-269 instead of planes)
  uint8_t* records;
  uint64_t records_stride;  // bytes per frame
  uint32_t emit_bands;      // tile rows emitted per frame (frame_h as passed / N)
  // LUMA: the frame's Y plane (level 0 of its packed pyramid) as a by-product of the pass over the BGR bytes
  uint8_t* luma;
  uint64_t luma_stride;     // bytes from one frame's level 0 to the next (the pyramid stride)
  // SPEC = 2 (redo the foreground): the MV blocks whose region id is not 0, as fg_list[0 .. *fg_count) = frame * mv_blocks + block
  const uint32_t* fg_list;
  const uint32_t* fg_count;
  uint32_t segs_per_block_x, segs_per_block;  // segment columns of one MV block: (mv_bw / 16) x (mv_bh / N)
};

template <int N> struct Basis;
template <> struct Basis<8> {
  static __device__ __forceinline__ double even(int k, int i) { return kDctEven8[k][i]; }
  static __device__ __forceinline__ double odd(int k, int i) { return kDctOdd8[k][i]; }
};
template <> struct Basis<16> {
  static __device__ __forceinline__ double even(int k, int i) { return kDctEven16[k][i]; }
  static __device__ __forceinline__ double odd(int k, int i) { return kDctOdd16[k][i]; }
};

template <int N, int L> struct RecTab;
#define SVC_RECTAB(N_, L_) \
  template <> struct RecTab<N_, L_> { \
    static __device__ __forceinline__ double at(int r, int i) { return kDctRec##N_##_L##L_[r][i]; } \
  }
SVC_RECTAB(8, 0); SVC_RECTAB(8, 1); SVC_RECTAB(8, 2);
SVC_RECTAB(16, 0); SVC_RECTAB(16, 1); SVC_RECTAB(16, 2); SVC_RECTAB(16, 3);
#undef SVC_RECTAB
template <int N> struct RecDc;
template <> struct RecDc<8> { static constexpr double v = kDctRec8_Dc; };
template <> struct RecDc<16> { static constexpr double v = kDctRec16_Dc; };

// N-point orthonormal DCT-II by the even/odd split of the basis, applied recursively: the odd rows
// of a level act on the differences x[i] - x[M-1-i], the even rows are a scaled M/2-point DCT of the
// sums (86 multiplies for 16 points instead of 128, 22 instead of 32 for 8).  x holds the M inputs
// of level L, y the N outputs: level L produces the rows k = 2^L * odd.
// T = int for the row pass: the inputs are bytes, so the sums and differences of every level are integers <= 4080 (byte
// extraction folds into the adds as SDWA operands); each operand of a multiply is widened once (v_cvt_f64_i32).
template <int N, int M, int L, typename T>
__device__ __forceinline__ void dct_level(const T* __restrict__ x, double* __restrict__ y) {
  if constexpr (M == 1) {
    y[0] = RecDc<N>::v * (double)x[0];
  } else {
    constexpr int H = M / 2;
    T s[H];
    double d[H];
#pragma unroll
    for (int i = 0; i < H; ++i) {
      s[i] = x[i] + x[M - 1 - i];
      d[i] = (double)(x[i] - x[M - 1 - i]);
    }
#pragma unroll
    for (int r = 0; r < H; ++r) {
      double o = RecTab<N, L>::at(r, 0) * d[0];
#pragma unroll
      for (int i = 1; i < H; ++i) o = __builtin_fma(RecTab<N, L>::at(r, i), d[i], o);
      y[(1 << L) * (2 * r + 1)] = o;
    }
    dct_level<N, H, L + 1, T>(s, y);
  }
}

template <int N, typename T>
__device__ __forceinline__ void dct1d(const T* __restrict__ x, double* __restrict__ y) {
  dct_level<N, N, 0, T>(x, y);
}

// libs/decoder.cpp:141-143: c /= step; c = std::round(c); c *= step  (all f32)
__device__ __forceinline__ float quant1(float c, float step) {
  float q = c / step;  // correctly rounded (hipcc default), as the CPU's divss
  q = roundf(q);
  return q * step;
}

// The same three lines at a fraction of the cost (the IEEE divide expansion + roundf are
// ~60 issue cycles per coefficient on gfx950 and made the fused kernel VALU-bound):
//  - division: q0 = c * inv, r = fma(-q0, step, c), q = fma(r, inv, q0) with inv = RN(1/step)
//    from the host is the correctly rounded quotient (Markstein's correction step) as long
//    as nothing under/overflows -- coefficients here are 0 or 1e-16 < |c| < 4100;
//  - std::round (half away from zero) == trunc(q + copysign(0.5 - 2^-25, q)) for every float.
// Both identities are checked bit-for-bit against the oracle by tests/test_gpu_dct_quant.py.
__device__ __forceinline__ float quant1_fast(float c, float step, float inv) {
  const float q0 = c * inv;
  const float r = __builtin_fmaf(-q0, step, c);
  float q = __builtin_fmaf(r, inv, q0);
  q = __builtin_truncf(q + __builtin_copysignf(0.49999997f, q));
  return q * step;
}

// Two coefficients of one tile at a time: the same five steps as v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32, which issue
// like one f32 instruction on gfx950 (copysign and trunc have no packed form) -- 9 instructions per pair instead of 14.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 quant2_fast(f32x2 c, float step, float inv) {
  const f32x2 s2 = {step, step}, i2 = {inv, inv};
  const f32x2 q0 = c * i2;
  const f32x2 r = __builtin_elementwise_fma(-q0, s2, c);
  f32x2 q = __builtin_elementwise_fma(r, i2, q0);
  const f32x2 h = {__builtin_copysignf(0.49999997f, q.x), __builtin_copysignf(0.49999997f, q.y)};
  q = q + h;
  q = f32x2{__builtin_truncf(q.x), __builtin_truncf(q.y)};
  return q * s2;
}

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

constexpr int kRowPitch = 144;                  // 16 f64 + 16 B pad
constexpr int kSlab8 = 8 * kRowPitch;           // 1152 B  (= 128 mod 256)
constexpr int kSlab16 = 16 * kRowPitch + 128;   // 2432 B  (= 128 mod 256)
// WIRE: once the last channel has left it, a wave's slabs (adjacent in LDS) become ONE linear staging buffer for the
// records of the wave's 64 / N segment columns (8 x 2 x 772 = 12 352 B for N = 8, 4 x 3076 = 12 304 B for N = 16) plus up
// to 12 bytes of alignment slack -- so the slab pitch grows to cover its share of that, still = 128 mod 256
constexpr int kSlabWire8 = 1664, kSlabWire16 = 3200;

typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));

// WIRE: instead of three coefficient planes the kernel emits the reference's serialised records (libs/encoder.cpp:222-269:
// u32 block type, then per channel N rows of N floats).  A record is 4 + 12 N^2 bytes -- 772 or 3076: no run of it is aligned
// to a cache line, and stores that leave as pieces of lines cost the transform a third of its time (2.20 ms against 1.6 ms
// with line-aligned 768-byte pseudo-records at C3, profiles/r04_ab_wire_aligned_experiment.txt; aligning only the 16-byte
// stores is slower still, r04_ab_wire_aligned_chunks.txt).  So a lane keeps the f32 results of all three channels in
// registers; when the last channel is through, the wave lays the COMPLETE records of its 64 / N segment columns -- one
// contiguous 12 KB stretch of the stream -- into its slabs in stream order and writes them out as 16-byte chunks at
// consecutive aligned addresses, 1 KiB per instruction: whole lines except at the stretch's two ends.  A wave whose segment
// columns do not form such a stretch (the clip's last wave, a frame boundary inside it, rows SerializeEncodedFrame does not
// visit) stores from the registers directly.  No extra HBM traffic versus planar output.
//
// LUMA (with WIRE, no QUANT: the reference encoder's real output is raw coefficients, libs/encoder.cpp:638-650): the lane that holds 16
// B,G,R pixels of a row for the transform also has everything cv::cvtColor needs for them (libs/encoder.cpp:468-469) -- Y is pointwise --
// so it stores the 16 luma bytes into level 0 of the frame's pyramid and the clip's BGR bytes are read ONCE per step instead of twice
// (luma pass + transform: 1.88 of 13.75 GB per step at C3).  The region id of a tile is not known yet when this runs (it needs the
// pyramid this kernel is producing): every record's type word is written as 0 = background (libs/codec.hpp:6) and
// wire_patch_types_kernel (wire.hip) stores the foreground ids once the segmentation has them.
//
// SPEC (planes + quant): the quantiser's step depends on the tile's region id (libs/decoder.cpp:130-135), which exists only after luma ->
// pyramid -> motion search -> RANSAC -> segmentation of the SAME frame -- that is why the two-pass step reads the BGR clip twice.  Speculation
// removes the second read: SPEC = 1 runs at the FRONT of the step (with LUMA), quantises EVERY tile as background (id 0: bg_step) and leaves the
// luma plane; once the ids exist, SPEC = 2 redoes exactly the tiles of foreground MV blocks (fg_list, built by fg_list_kernel) with fg_step:
// a fixed grid whose workgroups walk the list.  Same arithmetic, same bytes as SPEC = 0 with the ids up front.  By bytes alone (15 per
// foreground pixel redone against 3 per pixel saved) speculation would pay up to ~17 % foreground; MEASURED on MI355X it pays below ~2-3 % --
// the redo moves scattered 16-pixel pieces at a third of the streaming rate (profiles/r05_ab_speculative_quant.txt: at 13 % foreground the
// speculative step is 19 % slower).  svc_hip_count_foreground is how a caller decides (svc::ClipEncoder does, per chunk: C3 has 0.5 %).
template <int N, bool QUANT, bool WIRE, bool LUMA = false, int SPEC = 0>
__global__ __launch_bounds__(256) void dct_kernel(DctArgs a) {
  static_assert(!LUMA || (WIRE && !QUANT) || (QUANT && !WIRE && SPEC == 1), "the luma by-product rides on the raw-coefficient record emitter or on the speculative quantiser");
  static_assert(SPEC == 0 || (QUANT && !WIRE), "speculation is about the quantiser's step");
  constexpr int kSegPerWg = 256 / N;  // 512- and 1024-lane workgroups (longer runs per row) measured level or worse: profiles/r03_ab_dct_lanes.txt
  constexpr int kSlab = WIRE ? (N == 8 ? kSlabWire8 : kSlabWire16) : (N == 8 ? kSlab8 : kSlab16);
  __shared__ __attribute__((aligned(16))) uint8_t lds[kSegPerWg * kSlab];

  const uint32_t tid = threadIdx.x;
  const uint32_t sc_local = tid / N, j = tid % N;
  uint32_t n_units = 1, unit = 0;
  if (SPEC == 2) {  // workgroup-uniform trip count: the list's segment columns / segment columns per workgroup trip
    const uint32_t total = *a.fg_count * a.segs_per_block;
    n_units = (total + kSegPerWg - 1) / kSegPerWg;
    unit = blockIdx.x;
  }
  for (; unit < n_units; unit += SPEC == 2 ? gridDim.x : 1u) {
  uint32_t gsc, frame, band, seg;
  if (SPEC == 2) {
    gsc = unit * kSegPerWg + sc_local;  // index into the list's segment columns
    if (gsc >= *a.fg_count * a.segs_per_block) break;  // whole N-lane groups leave together (the last trip of the last workgroup)
    const uint32_t item = a.fg_list[gsc / a.segs_per_block], sub = gsc % a.segs_per_block;
    frame = item / a.mv_blocks;
    const uint32_t b = item - frame * a.mv_blocks, by = b / a.mfw, bx = b - by * a.mfw;
    band = by * (a.mv_bh / N) + sub / a.segs_per_block_x;
    seg = bx * a.segs_per_block_x + sub % a.segs_per_block_x;
  } else {
    // a workgroup's place in the clip follows the XCD it runs on: the 2 KiB row pieces that neighbouring
    // workgroups write land in the same L2 and leave it as longer runs (stores alone: 1.33 -> 1.25 ms)
    gsc = xcd_contiguous_block(blockIdx.x, gridDim.x) * kSegPerWg + sc_local;
    if (gsc >= a.total_segcols) return;  // whole N-lane groups leave together
    const uint32_t band_g = gsc / a.segs_per_band;
    seg = gsc - band_g * a.segs_per_band;
    frame = band_g / a.bands_per_frame;
    band = band_g - frame * a.bands_per_frame;
  }
  const uint32_t y_pix = band * N, x_pix = seg * 16;

  // 16 BGR pixels of row j of this segment column
  const uint8_t* src = a.bgr + (size_t)frame * a.frame_stride +
                       ((size_t)(y_pix + j) * a.w + x_pix) * 3;
  uint32_t wds[12];
  {
    const uint4* p = reinterpret_cast<const uint4*>(src);
    uint4 v0 = p[0], v1 = p[1], v2 = p[2];
    wds[0] = v0.x; wds[1] = v0.y; wds[2] = v0.z; wds[3] = v0.w;
    wds[4] = v1.x; wds[5] = v1.y; wds[6] = v1.z; wds[7] = v1.w;
    wds[8] = v2.x; wds[9] = v2.y; wds[10] = v2.z; wds[11] = v2.w;
  }

  if (LUMA) {
    uint32_t y16[4];
    luma16(wds, y16);
    *reinterpret_cast<uint4*>(a.luma + (size_t)frame * a.luma_stride + (size_t)(y_pix + j) * a.w + x_pix) = make_uint4(y16[0], y16[1], y16[2], y16[3]);
  }

  uint8_t* slab = lds + sc_local * kSlab;
  float* out_frame = a.planes + (size_t)frame * 3 * a.w * a.h;

  float step = 1.f, inv_step = 1.f;
  uint32_t t = SPEC == 2 ? 1u : 0u;  // SPEC 1: everything as background; SPEC 2: the list holds foreground blocks only
  if ((QUANT || WIRE) && !LUMA && SPEC == 0) {
    // tile type = type of the MV block that holds it (libs/encoder.cpp:243-249);
    // background (0, libs/codec.hpp:6) takes bg_step (libs/decoder.cpp:130-135)
    const uint32_t col = N == 8 ? x_pix + 2 * j : x_pix + j;
    t = a.types[(size_t)frame * a.mv_blocks + (y_pix / a.mv_bh) * a.mfw + col / a.mv_bw];
  }
  if (QUANT) {
    step = t == 0 ? a.bg_step : a.fg_step;
    inv_step = t == 0 ? a.bg_inv : a.fg_inv;
  }
  // records of this segment column: tile (band, 2 seg + q) for N = 8, (band, seg) for N = 16
  constexpr uint32_t kRec = 4 + 12 * N * N;
  const bool emit = WIRE && band < a.emit_bands;
  uint8_t* rec0 = nullptr;
  if (WIRE) {
    const uint32_t tiles_x = a.w / N, tile0 = band * tiles_x + (N == 8 ? 2 * seg : seg);
    rec0 = a.records + (size_t)frame * a.records_stride + (size_t)tile0 * kRec;
  }
  float keep[WIRE ? 3 : 1][16];  // WIRE: this lane's f32 results, all channels (N = 8: row v, columns 2j, 2j + 1 at [2v], [2v + 1])

#pragma unroll
  for (int c = 0; c < 3; ++c) {
    int x[16];
    double r[16];
#pragma unroll
    for (int p = 0; p < 16; ++p) x[p] = (int)((wds[(3 * p + c) >> 2] >> (8 * ((3 * p + c) & 3))) & 0xFFu);
    if (N == 8) {
      dct1d<8, int>(x, r);
      dct1d<8, int>(x + 8, r + 8);
    } else {
      dct1d<16, int>(x, r);
    }
    double2* row = reinterpret_cast<double2*>(slab + j * kRowPitch);
#pragma unroll
    for (int i = 0; i < 8; ++i) row[i] = make_double2(r[2 * i], r[2 * i + 1]);
    wave_lds_sync();

    float* plane = out_frame + (size_t)c * a.w * a.h;
    if (N == 8) {
      // lane j takes columns 2j, 2j+1 of the 16-wide slab (tile j >> 2)
      double ca[8], cb[8], ya[8], yb[8];
#pragma unroll
      for (int y = 0; y < 8; ++y) {
        double2 t = *reinterpret_cast<const double2*>(slab + y * kRowPitch + j * 16);
        ca[y] = t.x;
        cb[y] = t.y;
      }
      dct1d<8, double>(ca, ya);
      dct1d<8, double>(cb, yb);
      float* dst = plane + (size_t)y_pix * a.w + x_pix + 2 * j;
#pragma unroll
      for (int v = 0; v < 8; ++v) {
        float fa = (float)ya[v], fb = (float)yb[v];
        if (QUANT) {
          const f32x2 qq = quant2_fast(f32x2{fa, fb}, step, inv_step);
          fa = qq.x; fb = qq.y;
        }
        if (WIRE) { keep[WIRE ? c : 0][2 * v] = fa; keep[WIRE ? c : 0][2 * v + 1] = fb; }
        else *reinterpret_cast<float2*>(dst + (size_t)v * a.w) = make_float2(fa, fb);
      }
    } else {
      double cc[16], yy[16];
#pragma unroll
      for (int y = 0; y < 16; ++y)
        cc[y] = *reinterpret_cast<const double*>(slab + y * kRowPitch + j * 8);
      dct1d<16, double>(cc, yy);
      float* dst = plane + (size_t)y_pix * a.w + x_pix + j;
#pragma unroll
      for (int v = 0; v < 16; v += 2) {
        f32x2 f = {(float)yy[v], (float)yy[v + 1]};
        if (QUANT) f = quant2_fast(f, step, inv_step);
        if (WIRE) {
          keep[WIRE ? c : 0][v] = f.x;
          keep[WIRE ? c : 0][v + 1] = f.y;
        } else {
          dst[(size_t)v * a.w] = f.x;
          dst[(size_t)(v + 1) * a.w] = f.y;
        }
      }
    }
    wave_lds_sync();  // the slab is rewritten by the next channel (WIRE: by the record staging below)
  }

  if constexpr (WIRE) {
    constexpr uint32_t kGroupBytes = N == 8 ? 2 * kRec : kRec;  // the records of one segment column
    constexpr uint32_t kGroups = 64 / N;                         // segment columns per wave
    constexpr uint32_t kWaveBytes = kGroups * kGroupBytes;
    static_assert(kGroups * (uint32_t)kSlab >= kWaveBytes + 16, "a wave's slabs hold its stretch of the stream plus alignment slack");
    const uint32_t lane = tid & 63u, g = lane / N;
    // the records of consecutive segment columns are consecutive in the stream (across bands as well: tile index = band *
    // tiles_x + tile column), as long as they are in one frame and SerializeEncodedFrame visits their rows
    const uint32_t gsc_last = gsc - g + kGroups - 1;
    bool stretch = gsc_last < a.total_segcols;
    if (stretch) {
      const uint32_t bg_last = gsc_last / a.segs_per_band, bg_first = (gsc - g) / a.segs_per_band;
      const uint32_t f_last = bg_last / a.bands_per_frame;
      stretch = f_last == bg_first / a.bands_per_frame && bg_last - f_last * a.bands_per_frame < a.emit_bands;
    }
    const uint32_t q = N == 8 ? j >> 2 : 0u;        // tile of the segment column this lane's columns belong to
    const uint32_t col_off = N == 8 ? (j & 3u) * 8u : j * 4u;  // byte offset of the lane's column(s) inside a record row
    if (stretch) {  // wave-uniform
      uint8_t* wave_rec = rec0 - (size_t)g * kGroupBytes;                      // the same address in every lane
      const uint32_t delta = (uint32_t)(reinterpret_cast<uintptr_t>(wave_rec) & 15u);
      uint8_t* wbase = lds + (sc_local - g) * kSlab + delta;                   // LDS image of the stretch: same 16-byte phase as global
      uint8_t* mine = wbase + g * kGroupBytes + q * kRec;
      if (N == 8 ? (j & 3u) == 0 : j == 0) *reinterpret_cast<uint32_t*>(mine) = t;
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int v = 0; v < N; ++v) {
          uint32_t* d = reinterpret_cast<uint32_t*>(mine + 4 + c * (4 * N * N) + v * (4 * N) + col_off);
          if (N == 8) { d[0] = __float_as_uint(keep[c][2 * v]); d[1] = __float_as_uint(keep[c][2 * v + 1]); }
          else d[0] = __float_as_uint(keep[c][v]);
        }
      wave_lds_sync();
      const uint32_t head = (16u - delta) & 15u;                  // bytes up to the first 16-byte boundary (a multiple of 4)
      const uint32_t nfull = (kWaveBytes - head) / 16u, tail = (kWaveBytes - head) % 16u;
      for (uint32_t k = lane; k < nfull; k += 64u) {               // 1 KiB of consecutive aligned chunks per wave instruction
        const uint32_t o = head + 16u * k;
        *reinterpret_cast<uint4*>(wave_rec + o) = *reinterpret_cast<const uint4*>(wbase + o);
      }
      const uint32_t hd = head / 4u, ne = hd + tail / 4u;          // the dwords in front of the first and behind the last chunk
      if (lane < ne) {
        const uint32_t o = lane < hd ? 4u * lane : head + 16u * nfull + 4u * (lane - hd);
        *reinterpret_cast<uint32_t*>(wave_rec + o) = *reinterpret_cast<const uint32_t*>(wbase + o);
      }
    } else if (emit) {  // no stretch: this lane's values straight from its registers
      uint8_t* mine = rec0 + q * kRec;
      if (N == 8 ? (j & 3u) == 0 : j == 0) *reinterpret_cast<uint32_t*>(mine) = t;
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int v = 0; v < N; ++v) {
          uint32_t* d = reinterpret_cast<uint32_t*>(mine + 4 + c * (4 * N * N) + v * (4 * N) + col_off);
          if (N == 8) { d[0] = __float_as_uint(keep[c][2 * v]); d[1] = __float_as_uint(keep[c][2 * v + 1]); }
          else d[0] = __float_as_uint(keep[c][v]);
        }
    }
  }
  }  // the trips of SPEC = 2 (one trip otherwise); a slab is rewritten by the wave that owns it, behind its own wave_lds_sync
}

// The foreground MV blocks of a batch of frames as a list (order irrelevant: every entry is redone independently).
__global__ __launch_bounds__(256) void fg_list_kernel(const uint32_t* types, uint32_t total_blocks, uint32_t* list, uint32_t* count) {
  const uint32_t g = blockIdx.x * 256u + threadIdx.x;
  const bool fg = g < total_blocks && types[g] != 0;
  // one atomic per wave: the wave's foreground lanes take consecutive slots
  const uint64_t m = __builtin_amdgcn_ballot_w64(fg);
  if (m == 0) return;
  const uint32_t lane = threadIdx.x & 63u, n = (uint32_t)__builtin_popcountll(m);
  uint32_t base = 0;
  if (lane == (uint32_t)__builtin_ctzll(m)) base = atomicAdd(count, n);
  base = __builtin_amdgcn_readlane(base, __builtin_ctzll(m));
  if (fg) list[base + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull))] = g;
}

// ---- any transform block the reference's Validate admits (libs/encoder.cpp:62-142) ---------------
// static Dct (libs/encoder.cpp:323-339) loops over ANY block_w x block_h; Validate only asks that the
// transform block divides the MV block.  8x8 / 16x16 on frames 16 pixels wide are the kernels above;
// every other shape -- 2x2, 4x4, 16x8, 8x16, 32x32, 1xN ... up to 64x64 -- runs here, straight from the
// definition Y = Ch X Cw^T in f64 (rounded once to f32, like the fast kernels):
//   a workgroup owns a strip (bh rows x SW columns, SW a multiple of bw) of one frame; per channel the
//   strip goes to LDS as f32 (exact), a row pass multiplies every tile row by Cw^T into an f64 strip, a
//   column pass multiplies by Ch, quantises and stores lane-contiguous f32.  The two basis matrices are
//   evaluated once per workgroup (cospi on the exactly reduced angle) and kept in LDS; a workgroup walks
//   several strips, so that cost is amortised.  Not the roofline path: the fast kernels carry BASELINE's
//   configurations, this one carries the rest of the API.
struct DctGenArgs {
  const uint8_t* bgr;
  uint64_t frame_stride;
  uint32_t w, h, bw, bh;
  uint32_t sw;               // strip width, a multiple of bw
  uint32_t strips_per_band;  // ceil(w / sw)
  uint32_t bands_per_frame;  // h / bh
  uint32_t total_strips;
  float* planes;
  const uint32_t* types;
  uint32_t mv_bw, mv_bh, mfw, mv_blocks;
  float fg_step, bg_step, fg_inv, bg_inv;
  uint8_t* records;         // WIRE: serialised records (libs/encoder.cpp:222-269) instead of planes; square blocks only
  uint64_t records_stride;  // bytes per frame
  uint32_t emit_bands;      // tile rows SerializeEncodedFrame visits (ceil(emit_frame_h / bh))
};

// C[k][n] = s_k cos(pi (2n + 1) k / 2N), s_0 = sqrt(1/N), s_k = sqrt(2/N): the orthonormal DCT-II of
// cv::dct(flags = 0).  (2n + 1) k is reduced mod 4N in integers, so the argument of cospi is in [0, 2).
__device__ __forceinline__ double dct_basis(uint32_t k, uint32_t n, uint32_t N) {
  const uint32_t j = ((2 * n + 1) * k) % (4 * N);
  const double s = k == 0 ? 1.0 / (double)N : 2.0 / (double)N;
  return __builtin_sqrt(s) * cospi((double)j / (double)(2 * N));
}

// WIRE (square blocks): the coefficient of channel c at (v, u) of tile (band, tx) goes to dword 1 + c * bw * bh + v * bh + u
// of the tile's record, the tile's type word to dword 0 -- the bytes of the plane form followed by serialize_kernel
// (wire.hip), without the two extra passes over 25 MB per frame.
template <bool QUANT, bool WIRE>
__global__ __launch_bounds__(256) void dct_general_kernel(DctGenArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t gen_lds[];
  const uint32_t tid = threadIdx.x, bw = a.bw, bh = a.bh, sw = a.sw;
  const uint32_t pw = bw + 1, ph = bh + 1;  // odd-ish pitches: rows of a basis matrix fall on different banks
  double* cw = reinterpret_cast<double*>(gen_lds);  // [bw][pw]
  double* ch = cw + bw * pw;                        // [bh][ph]
  double* yrow = ch + bh * ph;                      // [bh][sw] row-pass results
  float* xin = reinterpret_cast<float*>(yrow + bh * sw);  // [bh][sw] pixels of one channel
  for (uint32_t i = tid; i < bw * bw; i += 256) cw[(i / bw) * pw + i % bw] = dct_basis(i / bw, i % bw, bw);
  for (uint32_t i = tid; i < bh * bh; i += 256) ch[(i / bh) * ph + i % bh] = dct_basis(i / bh, i % bh, bh);
  __syncthreads();
  for (uint32_t unit = blockIdx.x; unit < a.total_strips; unit += gridDim.x) {
    const uint32_t band_g = unit / a.strips_per_band, strip = unit - band_g * a.strips_per_band;
    const uint32_t frame = band_g / a.bands_per_frame, band = band_g - frame * a.bands_per_frame;
    const uint32_t y0 = band * bh, x0 = strip * sw;
    const uint32_t cols = a.w - x0 < sw ? a.w - x0 : sw;  // the last strip of a band may be narrower
    const uint32_t n = bh * cols;
    const uint8_t* src = a.bgr + (size_t)frame * a.frame_stride;
    float* out_frame = a.planes + (size_t)frame * 3 * a.w * a.h;
    for (uint32_t c = 0; c < 3; ++c) {
      for (uint32_t i = tid; i < n; i += 256) {
        const uint32_t r = i / cols, x = i - r * cols;
        xin[r * sw + x] = (float)src[((size_t)(y0 + r) * a.w + x0 + x) * 3 + c];
      }
      __syncthreads();
      for (uint32_t i = tid; i < n; i += 256) {  // rows: y[r][t * bw + u] = sum_n Cw[u][n] x[r][t * bw + n]
        const uint32_t r = i / cols, x = i - r * cols, u = x % bw;
        const float* xr = xin + r * sw + (x - u);
        const double* cr = cw + u * pw;
        double acc = 0.0;
        for (uint32_t k = 0; k < bw; ++k) acc = __builtin_fma(cr[k], (double)xr[k], acc);
        yrow[r * sw + x] = acc;
      }
      __syncthreads();
      float* plane = out_frame + (size_t)c * a.w * a.h;
      for (uint32_t i = tid; i < n; i += 256) {  // columns: Y[v][x] = sum_m Ch[v][m] y[m][x]
        const uint32_t v = i / cols, x = i - v * cols;
        const double* cr = ch + v * ph;
        double acc = 0.0;
        for (uint32_t m = 0; m < bh; ++m) acc = __builtin_fma(cr[m], yrow[m * sw + x], acc);
        float f = (float)acc;
        uint32_t t = 0;
        if (QUANT || WIRE)
          // tile type = type of the MV block that holds it (libs/encoder.cpp:243-249); background (0) takes bg_step
          t = a.types[(size_t)frame * a.mv_blocks + (y0 / a.mv_bh) * a.mfw + (x0 + x) / a.mv_bw];
        if (QUANT) f = quant1_fast(f, t == 0 ? a.bg_step : a.fg_step, t == 0 ? a.bg_inv : a.fg_inv);
        if (WIRE) {
          if (band < a.emit_bands) {
            const uint32_t u = x % bw, tile = band * (a.w / bw) + (x0 + x) / bw, rec_dw = 1 + 3 * bw * bh;
            uint32_t* rec = reinterpret_cast<uint32_t*>(a.records + (size_t)frame * a.records_stride) + (size_t)tile * rec_dw;
            rec[1 + c * bw * bh + v * bh + u] = __float_as_uint(f);
            if (c == 0 && v == 0 && u == 0) rec[0] = t;
          }
        } else {
          plane[(size_t)(y0 + v) * a.w + x0 + x] = f;
        }
      }
      __syncthreads();
    }
  }
}

constexpr uint32_t kDctGenMaxSide = 64;

static int launch_dct_general(const uint8_t* d_bgr, uint64_t frame_stride, uint32_t n_frames, uint32_t w, uint32_t h,
                              uint32_t bw, uint32_t bh, const uint32_t* d_types, uint32_t mv_bw, uint32_t mv_bh,
                              uint32_t fg_step, uint32_t bg_step, bool quant, float* d_planes, hipStream_t stream,
                              uint8_t* d_records = nullptr, uint64_t records_stride = 0, uint32_t emit_h = 0) {
  const bool wire = d_records != nullptr;
  if (wire && (bw != bh || w % bw != 0))
    return fail(SVC_ERR_UNSUPPORTED, "dct_records: the fused record emitter takes square transform blocks that divide the frame "
                                     "width; for %ux%u call svc_hip_dct[_quant]_frames, then svc_hip_serialize_frames", bw, bh);
  // cv::dct (libs/encoder.cpp:335) takes even sizes, and single rows / columns of even length
  if ((bw > 1 && bw % 2) || (bh > 1 && bh % 2) || (bw == 1 && bh == 1))
    return fail(SVC_ERR_INVALID_ARG, "dct: transform block %ux%u: cv::dct implements even sizes only", bw, bh);
  if (bw > kDctGenMaxSide || bh > kDctGenMaxSide)
    return fail(SVC_ERR_UNSUPPORTED, "dct: transform block %ux%u exceeds %ux%u", bw, bh, kDctGenMaxSide, kDctGenMaxSide);
  DctGenArgs a{};
  a.bgr = d_bgr; a.frame_stride = frame_stride;
  a.w = w; a.h = h; a.bw = bw; a.bh = bh;
  // strip: about 4096 samples, at least one tile, whole tiles
  uint32_t tiles = 4096 / (bw * bh);
  if (tiles < 1) tiles = 1;
  if (tiles > w / bw) tiles = w / bw;
  a.sw = tiles * bw;
  a.strips_per_band = div_up(w, a.sw);
  a.bands_per_frame = h / bh;
  const uint64_t total = (uint64_t)n_frames * a.strips_per_band * a.bands_per_frame;
  if (total == 0) return SVC_OK;
  if (total > 0x7FFFFFFFull) return fail(SVC_ERR_UNSUPPORTED, "dct: %llu strips exceed one launch", (unsigned long long)total);
  a.total_strips = (uint32_t)total;
  a.planes = d_planes;
  a.records = d_records;
  a.records_stride = records_stride;
  a.emit_bands = wire ? div_up(emit_h, bh) : 0;
  if (quant || wire) {
    a.types = d_types;
    a.mv_bw = mv_bw; a.mv_bh = mv_bh;
    a.mfw = w / mv_bw;
    a.mv_blocks = a.mfw * (h / mv_bh);
  }
  if (quant) {
    a.fg_step = (float)fg_step; a.bg_step = (float)bg_step;
    a.fg_inv = 1.0f / a.fg_step; a.bg_inv = 1.0f / a.bg_step;
  }
  const size_t lds = 8 * ((size_t)bw * (bw + 1) + (size_t)bh * (bh + 1) + (size_t)bh * a.sw) + 4 * (size_t)bh * a.sw;
  const uint32_t grid = (uint32_t)(total < 4096 ? total : 4096);
  if (wire) {
    if (quant) hipLaunchKernelGGL((dct_general_kernel<true, true>), dim3(grid), dim3(256), lds, stream, a);
    else hipLaunchKernelGGL((dct_general_kernel<false, true>), dim3(grid), dim3(256), lds, stream, a);
  } else {
    if (quant) hipLaunchKernelGGL((dct_general_kernel<true, false>), dim3(grid), dim3(256), lds, stream, a);
    else hipLaunchKernelGGL((dct_general_kernel<false, false>), dim3(grid), dim3(256), lds, stream, a);
  }
  return check_launch("dct_general_kernel");
}

int launch_dct(const uint8_t* d_bgr, uint64_t frame_stride, uint32_t n_frames, uint32_t w,
               uint32_t h, uint32_t bw, uint32_t bh, const uint32_t* d_types, uint32_t mv_bw,
               uint32_t mv_bh, uint32_t fg_step, uint32_t bg_step, bool quant, float* d_planes,
               hipStream_t stream, uint8_t* d_records, uint64_t records_stride, uint32_t emit_h, uint8_t* d_luma,
               uint64_t luma_stride) {
  const bool wire = d_records != nullptr;
  const bool fast = bw == bh && (bw == 8 || bw == 16) && w % 16 == 0;
  if (d_luma && !(fast && wire && !quant))
    return fail(SVC_ERR_UNSUPPORTED, "dct: the luma by-product needs the tuned record emitter (8x8 / 16x16 blocks, width a multiple of 16, no quant)");
  if (!fast)
    return launch_dct_general(d_bgr, frame_stride, n_frames, w, h, bw, bh, d_types, mv_bw, mv_bh, fg_step, bg_step, quant,
                              d_planes, stream, d_records, records_stride, emit_h);
  DctArgs a{};
  a.bgr = d_bgr;
  a.frame_stride = frame_stride;
  a.w = w; a.h = h;
  a.segs_per_band = w / 16;
  a.bands_per_frame = h / bh;
  const uint64_t total = (uint64_t)n_frames * a.segs_per_band * a.bands_per_frame;
  if (total == 0) return SVC_OK;
  if (total > 0x7FFFFFFFull) return fail(SVC_ERR_UNSUPPORTED, "dct: %llu segment columns exceed one launch", (unsigned long long)total);
  a.total_segcols = (uint32_t)total;
  a.planes = d_planes;
  a.records = d_records;
  a.records_stride = records_stride;
  a.emit_bands = wire ? div_up(emit_h, bh) : 0;
  a.luma = d_luma;
  a.luma_stride = luma_stride;
  if (quant || wire) {
    a.types = d_types;
    a.mv_bw = mv_bw; a.mv_bh = mv_bh;
    a.mfw = w / mv_bw;
    a.mv_blocks = a.mfw * (h / mv_bh);
  }
  if (quant) {
    a.fg_step = (float)fg_step;  // libs/decoder.cpp:141 divides a float by an unsigned
    a.bg_step = (float)bg_step;
    a.fg_inv = 1.0f / a.fg_step;
    a.bg_inv = 1.0f / a.bg_step;
  }
  const uint32_t seg_per_wg = 256 / bw;
  const dim3 grid(div_up(a.total_segcols, seg_per_wg)), block(256);
#define SVC_DCT_LAUNCH(N_, Q_, W_) hipLaunchKernelGGL((dct_kernel<N_, Q_, W_>), grid, block, 0, stream, a)
  if (d_luma) {
    if (bw == 8) hipLaunchKernelGGL((dct_kernel<8, false, true, true>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((dct_kernel<16, false, true, true>), grid, block, 0, stream, a);
  } else if (bw == 8) {
    if (wire) { if (quant) SVC_DCT_LAUNCH(8, true, true); else SVC_DCT_LAUNCH(8, false, true); }
    else { if (quant) SVC_DCT_LAUNCH(8, true, false); else SVC_DCT_LAUNCH(8, false, false); }
  } else {
    if (wire) { if (quant) SVC_DCT_LAUNCH(16, true, true); else SVC_DCT_LAUNCH(16, false, true); }
    else { if (quant) SVC_DCT_LAUNCH(16, true, false); else SVC_DCT_LAUNCH(16, false, false); }
  }
#undef SVC_DCT_LAUNCH
  return check_launch("dct_kernel");
}

// How many MV blocks of a batch are foreground (region id != 0): the feedback the driver's speculation policy runs on.
__global__ __launch_bounds__(256) void fg_count_kernel(const uint32_t* types, uint32_t total_blocks, uint32_t* count) {
  __shared__ uint32_t part[4];
  uint32_t n = 0;
  for (uint32_t g = blockIdx.x * 256u + threadIdx.x; g < total_blocks; g += gridDim.x * 256u) n += types[g] != 0 ? 1u : 0u;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) n += __shfl_down(n, o, 64);
  if ((threadIdx.x & 63u) == 0) part[threadIdx.x >> 6] = n;
  __syncthreads();
  // one atomic per workgroup with anything to add (a few hundred per launch: 38 k same-address atomics, one per wave, took 50 us at C3b)
  if (threadIdx.x == 0) {
    const uint32_t s = part[0] + part[1] + part[2] + part[3];
    if (s) atomicAdd(count, s);
  }
}

int launch_count_foreground(const uint32_t* d_types, uint64_t n, uint32_t* d_count, hipStream_t stream) {
  hipError_t e = hipMemsetAsync(d_count, 0, 4, stream);
  if (e != hipSuccess) return fail(SVC_ERR_HIP, "count_foreground: hipMemsetAsync: %s", hipGetErrorString(e));
  if (n == 0) return SVC_OK;
  if (n > 0x7FFFFFFFull) return fail(SVC_ERR_UNSUPPORTED, "count_foreground: too many MV blocks for one launch");
  hipLaunchKernelGGL(fg_count_kernel, dim3(std::min<uint32_t>(div_up((uint32_t)n, 1024), 512u)), dim3(256), 0, stream, d_types, (uint32_t)n, d_count);
  return check_launch("fg_count_kernel");
}

// ---- speculative quantisation: one pass over the BGR clip per step (see dct_kernel, SPEC) --------------------------------------
static bool spec_shape_ok(uint32_t w, uint32_t block) { return (block == 8 || block == 16) && w % 16 == 0; }

// planes quantised with bg_step EVERYWHERE (every tile taken for background) + the luma plane, at the front of a step
int launch_dct_quant_speculative(const uint8_t* d_bgr, uint64_t frame_stride, uint32_t n_frames, uint32_t w, uint32_t h, uint32_t block,
                                 uint32_t bg_step, float* d_planes, uint8_t* d_luma, uint64_t luma_stride, hipStream_t stream) {
  if (!spec_shape_ok(w, block))
    return fail(SVC_ERR_UNSUPPORTED, "dct_quant_luma: the speculative form needs the tuned transform (8x8 / 16x16 blocks, width a multiple of 16)");
  DctArgs a{};
  a.bgr = d_bgr; a.frame_stride = frame_stride;
  a.w = w; a.h = h;
  a.segs_per_band = w / 16;
  a.bands_per_frame = h / block;
  const uint64_t total = (uint64_t)n_frames * a.segs_per_band * a.bands_per_frame;
  if (total == 0) return SVC_OK;
  if (total > 0x7FFFFFFFull) return fail(SVC_ERR_UNSUPPORTED, "dct: %llu segment columns exceed one launch", (unsigned long long)total);
  a.total_segcols = (uint32_t)total;
  a.planes = d_planes;
  a.luma = d_luma; a.luma_stride = luma_stride;
  a.fg_step = a.bg_step = (float)bg_step;
  a.fg_inv = a.bg_inv = 1.0f / a.bg_step;
  const dim3 grid(div_up(a.total_segcols, 256 / block)), blk(256);
  if (block == 8) hipLaunchKernelGGL((dct_kernel<8, true, false, true, 1>), grid, blk, 0, stream, a);
  else hipLaunchKernelGGL((dct_kernel<16, true, false, true, 1>), grid, blk, 0, stream, a);
  return check_launch("dct_kernel<speculative>");
}

uint64_t dct_redo_workspace_bytes(uint32_t n_frames, uint32_t mv_blocks) { return 16 + 4ull * n_frames * mv_blocks; }

// the tiles of foreground MV blocks once more, with fg_step: list the blocks, then a fixed grid walks the list
int launch_dct_quant_redo_foreground(const uint8_t* d_bgr, uint64_t frame_stride, uint32_t n_frames, uint32_t w, uint32_t h, uint32_t block,
                                     const uint32_t* d_types, uint32_t mv_bw, uint32_t mv_bh, uint32_t fg_step, float* d_planes,
                                     uint8_t* d_ws, hipStream_t stream) {
  if (!spec_shape_ok(w, block) || mv_bw % 16 != 0 || mv_bh % block != 0)
    return fail(SVC_ERR_UNSUPPORTED, "dct_quant_redo: needs the tuned transform (8x8 / 16x16, width a multiple of 16) and MV blocks that are whole "
                                     "16-pixel segments wide and whole transform blocks tall");
  DctArgs a{};
  a.bgr = d_bgr; a.frame_stride = frame_stride;
  a.w = w; a.h = h;
  a.segs_per_band = w / 16;
  a.bands_per_frame = h / block;
  a.planes = d_planes;
  a.mv_bw = mv_bw; a.mv_bh = mv_bh;
  a.mfw = w / mv_bw;
  a.mv_blocks = a.mfw * (h / mv_bh);
  const uint64_t total_blocks = (uint64_t)n_frames * a.mv_blocks;
  if (total_blocks == 0) return SVC_OK;
  if (total_blocks * (mv_bw / 16) * (mv_bh / block) > 0x7FFFFFFFull) return fail(SVC_ERR_UNSUPPORTED, "dct_quant_redo: too many MV blocks for one launch");
  a.fg_step = a.bg_step = (float)fg_step;
  a.fg_inv = a.bg_inv = 1.0f / a.fg_step;
  uint32_t* count = reinterpret_cast<uint32_t*>(d_ws);
  uint32_t* list = reinterpret_cast<uint32_t*>(d_ws + 16);
  a.fg_count = count; a.fg_list = list;
  a.segs_per_block_x = mv_bw / 16;
  a.segs_per_block = a.segs_per_block_x * (mv_bh / block);
  hipError_t e = hipMemsetAsync(count, 0, 16, stream);
  if (e != hipSuccess) return fail(SVC_ERR_HIP, "dct_quant_redo: hipMemsetAsync: %s", hipGetErrorString(e));
  hipLaunchKernelGGL(fg_list_kernel, dim3(div_up((uint32_t)total_blocks, 256)), dim3(256), 0, stream, d_types, (uint32_t)total_blocks, list, count);
  int rc = check_launch("fg_list_kernel");
  if (rc) return rc;
  // 36 KB of LDS: four workgroups per CU are resident; the grid never outnumbers the work a list of EVERY block would hold
  const uint64_t worst = div_up((uint32_t)(total_blocks * a.segs_per_block), 256 / block);
  const dim3 grid((uint32_t)std::min<uint64_t>(worst, 2048)), blk(256);
  if (block == 8) hipLaunchKernelGGL((dct_kernel<8, true, false, false, 2>), grid, blk, 0, stream, a);
  else hipLaunchKernelGGL((dct_kernel<16, true, false, false, 2>), grid, blk, 0, stream, a);
  return check_launch("dct_kernel<redo foreground>");
}

// ---- cv::dct over a LIST of tiles of one f32 image, in place -------------------------------------
// The reference's Dct calls cv::dct(block, block) once per transform block of an f32 plane (libs/encoder.cpp:330-337).
// A host that keeps that control flow (compat/opencv2/: the per-tile calls are collected and flushed together) hands
// over the plane and the tiles' corners; this is dct_general_kernel's arithmetic (same basis, same f64 FMA chains in the
// same order, rounded once to f32) on f32 input, TPB tiles per workgroup trip.
struct DctTilesArgs {
  float* img;
  const uint32_t* xy;  // [n_tiles][2] corners, or null: the regular grid, row-major
  uint32_t w, h, bw, bh, n_tiles, grid_w, tpb;
};

__global__ __launch_bounds__(256) void dct_tiles_kernel(DctTilesArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t tiles_lds[];
  const uint32_t tid = threadIdx.x, bw = a.bw, bh = a.bh, area = bw * bh;
  const uint32_t pw = bw + 1, ph = bh + 1;
  double* cw = reinterpret_cast<double*>(tiles_lds);  // [bw][pw]
  double* ch = cw + bw * pw;                          // [bh][ph]
  double* yrow = ch + bh * ph;                        // [tpb][bh][bw] row-pass results
  float* xin = reinterpret_cast<float*>(yrow + (size_t)a.tpb * area);  // [tpb][bh][bw]
  for (uint32_t i = tid; i < bw * bw; i += 256) cw[(i / bw) * pw + i % bw] = dct_basis(i / bw, i % bw, bw);
  for (uint32_t i = tid; i < bh * bh; i += 256) ch[(i / bh) * ph + i % bh] = dct_basis(i / bh, i % bh, bh);
  __syncthreads();
  for (uint32_t base = blockIdx.x * a.tpb; base < a.n_tiles; base += gridDim.x * a.tpb) {
    const uint32_t nt = a.n_tiles - base < a.tpb ? a.n_tiles - base : a.tpb, ne = nt * area;
    auto origin = [&](uint32_t t, uint32_t& tx, uint32_t& ty) {
      const uint32_t g = base + t;
      if (a.xy) { tx = a.xy[2 * g]; ty = a.xy[2 * g + 1]; }
      else { ty = (g / a.grid_w) * bh; tx = (g % a.grid_w) * bw; }
    };
    for (uint32_t i = tid; i < ne; i += 256) {
      const uint32_t t = i / area, e = i - t * area, r = e / bw, u = e - r * bw;
      uint32_t tx, ty;
      origin(t, tx, ty);
      xin[i] = a.img[(size_t)(ty + r) * a.w + tx + u];
    }
    __syncthreads();
    for (uint32_t i = tid; i < ne; i += 256) {  // rows: y[r][u] = sum_k Cw[u][k] x[r][k]
      const uint32_t t = i / area, e = i - t * area, r = e / bw, u = e - r * bw;
      const float* xr = xin + t * area + r * bw;
      const double* cr = cw + u * pw;
      double acc = 0.0;
      for (uint32_t k = 0; k < bw; ++k) acc = __builtin_fma(cr[k], (double)xr[k], acc);
      yrow[i] = acc;
    }
    __syncthreads();
    for (uint32_t i = tid; i < ne; i += 256) {  // columns: Y[v][u] = sum_m Ch[v][m] y[m][u]
      const uint32_t t = i / area, e = i - t * area, v = e / bw, u = e - v * bw;
      const double* cr = ch + v * ph;
      const double* yc = yrow + t * area + u;
      double acc = 0.0;
      for (uint32_t m = 0; m < bh; ++m) acc = __builtin_fma(cr[m], yc[m * bw], acc);
      uint32_t tx, ty;
      origin(t, tx, ty);
      a.img[(size_t)(ty + v) * a.w + tx + u] = (float)acc;
    }
    __syncthreads();
  }
}

int launch_dct_tiles(float* d_img, uint32_t w, uint32_t h, uint32_t bw, uint32_t bh, const uint32_t* d_xy, uint32_t n_tiles,
                     hipStream_t stream) {
  if ((bw > 1 && bw % 2) || (bh > 1 && bh % 2) || (bw == 1 && bh == 1) || bw == 0 || bh == 0)
    return fail(SVC_ERR_INVALID_ARG, "dct: transform block %ux%u: cv::dct implements even sizes only", bw, bh);
  if (bw > kDctGenMaxSide || bh > kDctGenMaxSide)
    return fail(SVC_ERR_UNSUPPORTED, "dct: transform block %ux%u exceeds %ux%u", bw, bh, kDctGenMaxSide, kDctGenMaxSide);
  if (n_tiles == 0) return SVC_OK;
  DctTilesArgs a{};
  a.img = d_img; a.xy = d_xy;
  a.w = w; a.h = h; a.bw = bw; a.bh = bh; a.n_tiles = n_tiles;
  a.grid_w = w / bw;
  uint32_t tpb = 4096 / (bw * bh);
  a.tpb = tpb < 1 ? 1 : tpb > 64 ? 64 : tpb;
  const size_t lds = 8 * ((size_t)bw * (bw + 1) + (size_t)bh * (bh + 1)) + (size_t)a.tpb * bw * bh * 12;
  const uint32_t trips = div_up(n_tiles, a.tpb);
  hipLaunchKernelGGL(dct_tiles_kernel, dim3(trips < 2048 ? trips : 2048), dim3(256), lds, stream, a);
  return check_launch("dct_tiles_kernel");
}

// ---- standalone quantise/dequantise (libs/decoder.cpp:140-144) ----------------

__global__ __launch_bounds__(256) void quant_kernel(float* c, uint64_t n, float step) {
  const uint64_t stride = (uint64_t)gridDim.x * 256 * 4;
  for (uint64_t i = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 4 <= n && (reinterpret_cast<uintptr_t>(c + i) & 15) == 0) {
      float4 v = *reinterpret_cast<float4*>(c + i);
      v.x = quant1(v.x, step); v.y = quant1(v.y, step);
      v.z = quant1(v.z, step); v.w = quant1(v.w, step);
      *reinterpret_cast<float4*>(c + i) = v;
    } else {
      for (uint64_t k = i; k < n && k < i + 4; ++k) c[k] = quant1(c[k], step);
    }
  }
}

struct QuantFramesArgs {
  float* planes;
  const uint32_t* types;
  uint32_t w, h, mv_bw, mv_bh, mfw, mv_blocks;
  uint64_t total;  // frames * 3 * h * w
  float fg_step, bg_step;
};

__global__ __launch_bounds__(256) void quant_frames_kernel(QuantFramesArgs a) {
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  const uint64_t plane_sz = (uint64_t)a.w * a.h;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < a.total; i += stride) {
    const uint64_t pl = i / plane_sz, rem = i - pl * plane_sz;
    const uint32_t frame = (uint32_t)(pl / 3);
    const uint32_t y = (uint32_t)(rem / a.w), x = (uint32_t)(rem - (uint64_t)y * a.w);
    const uint32_t t = a.types[(size_t)frame * a.mv_blocks + (y / a.mv_bh) * a.mfw + x / a.mv_bw];
    a.planes[i] = quant1(a.planes[i], t == 0 ? a.bg_step : a.fg_step);
  }
}

int launch_quant(float* d_coeffs, uint64_t n, uint32_t step, hipStream_t stream) {
  if (n == 0) return SVC_OK;
  const uint64_t want = (n + 1023) / 1024;
  const uint32_t grid = (uint32_t)(want < 4096 ? want : 4096);
  hipLaunchKernelGGL(quant_kernel, dim3(grid), dim3(256), 0, stream, d_coeffs, n, (float)step);
  return check_launch("quant_kernel");
}

int launch_quant_frames(float* d_planes, uint32_t n_frames, uint32_t w, uint32_t h,
                        uint32_t mv_bw, uint32_t mv_bh, const uint32_t* d_types,
                        uint32_t fg_step, uint32_t bg_step, hipStream_t stream) {
  QuantFramesArgs a;
  a.planes = d_planes;
  a.types = d_types;
  a.w = w; a.h = h; a.mv_bw = mv_bw; a.mv_bh = mv_bh;
  a.mfw = w / mv_bw;
  a.mv_blocks = a.mfw * (h / mv_bh);
  a.total = (uint64_t)n_frames * 3 * w * h;
  a.fg_step = (float)fg_step;
  a.bg_step = (float)bg_step;
  if (a.total == 0) return SVC_OK;
  const uint64_t want = (a.total + 255) / 256;
  const uint32_t grid = (uint32_t)(want < 8192 ? want : 8192);
  hipLaunchKernelGGL(quant_frames_kernel, dim3(grid), dim3(256), 0, stream, a);
  return check_launch("quant_frames_kernel");
}

}  // namespace svc
