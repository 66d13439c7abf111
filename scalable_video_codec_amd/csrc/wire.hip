// wire.hip -- the encoder's output stream: Header + one serialised record per transform tile.
//
// Reference: Header, libs/codec.hpp:8-17 and libs/encoder.cpp:360-381;
// SerializeEncodedFrame, libs/encoder.cpp:222-269.  The serialiser here takes the reference's
// arguments with the reference's meaning, quirks included, so the bytes are identical:
//   - tiles are visited over frame_w x frame_h AS PASSED; the encoder passes the UNPADDED size
//     (:647-650) although the coefficient planes are padded, so for a padded height the last
//     tile row(s) are not emitted, and the passed width doubles as the ROW STRIDE (:258);
//   - a record = u32 type of the tile's MV block (:243-249) + per channel `transform_block_w`
//     rows of `transform_block_h` floats (:257-262: w and h swapped; harmless for square tiles).
// (The reference's decoder expects the PADDED tile counts, libs/decoder.cpp:185-186 -- pass the
// padded size to get a stream it can parse; INTEGRATION.md.)
// Pure data movement: one lane per output dword, fully coalesced stores.
#include "svc_common.hpp"

namespace svc {

struct WireArgs {
  const float* planes;      // [frames][3][plane_elems]
  const uint32_t* types;    // [frames][mv_blocks]
  uint32_t* out;            // frame f at out + f * out_stride_dw
  uint64_t plane_elems, out_stride_dw, total_dw;
  uint32_t frame_w, tbw, tbh, tiles_x, rec_dw, frame_dw, mfw, mv_bw, mv_bh, mv_blocks;
};

__global__ __launch_bounds__(256) void serialize_kernel(WireArgs a) {
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  for (uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x; g < a.total_dw; g += stride) {
    const uint32_t frame = (uint32_t)(g / a.frame_dw), r = (uint32_t)(g - (uint64_t)frame * a.frame_dw);
    const uint32_t tile = r / a.rec_dw, k = r - tile * a.rec_dw;
    const uint32_t ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
    const uint32_t tb_x = tx * a.tbw, tb_y = ty * a.tbh;
    uint32_t v;
    if (k == 0) {
      v = a.types[(size_t)frame * a.mv_blocks + (tb_y / a.mv_bh) * a.mfw + tb_x / a.mv_bw];
    } else {
      const uint32_t e = k - 1, area = a.tbw * a.tbh;
      const uint32_t c = e / area, q = e - c * area;
      const uint32_t row = q / a.tbh, col = q - row * a.tbh;  // `tbw` rows of `tbh` floats
      const float* ch = a.planes + ((size_t)frame * 3 + c) * a.plane_elems;
      v = __float_as_uint(ch[(size_t)(tb_y + row) * a.frame_w + tb_x + col]);
    }
    a.out[(size_t)frame * a.out_stride_dw + r] = v;
  }
}

int launch_serialize(const float* d_planes, uint64_t plane_elems, uint32_t n_frames, const uint32_t* d_types,
                     uint32_t mv_blocks, uint32_t frame_w, uint32_t frame_h, uint32_t tbw, uint32_t tbh,
                     uint32_t mfw, uint32_t mv_bw, uint32_t mv_bh, uint8_t* d_out, uint64_t out_stride,
                     hipStream_t stream) {
  WireArgs a;
  a.planes = d_planes;
  a.types = d_types;
  a.out = reinterpret_cast<uint32_t*>(d_out);
  a.plane_elems = plane_elems;
  a.out_stride_dw = out_stride / 4;
  a.frame_w = frame_w; a.tbw = tbw; a.tbh = tbh;
  a.tiles_x = div_up(frame_w, tbw);
  a.rec_dw = 1 + 3 * tbw * tbh;
  a.frame_dw = a.tiles_x * div_up(frame_h, tbh) * a.rec_dw;
  a.total_dw = (uint64_t)a.frame_dw * n_frames;
  a.mfw = mfw; a.mv_bw = mv_bw; a.mv_bh = mv_bh; a.mv_blocks = mv_blocks;
  if (a.total_dw == 0) return SVC_OK;
  const uint64_t want = (a.total_dw + 255) / 256;
  hipLaunchKernelGGL(serialize_kernel, dim3((uint32_t)(want < 16384 ? want : 16384)), dim3(256), 0, stream, a);
  return check_launch("serialize_kernel");
}

// ---- region ids into records that were emitted before they were known ------------------------------------------------------
// dct_kernel<N, false, true, LUMA = true> writes every record's type word as 0 (background): it runs BEFORE the frame's pyramid -- and so
// its motion field, global motion and region ids -- exists.  Once the segmentation has the ids, this kernel stores the type word of every
// tile whose MV block is foreground (libs/encoder.cpp:243-249); background tiles already hold their 0.  One lane per MV block: it reads
// the block's id (coalesced) and, if non-zero, writes the (mv_bw / tb) x (mv_bh / tb) type words of its tiles (4 at 16 / 8, 1 at 16 / 16).
struct PatchArgs {
  const uint32_t* types;  // [frames][mv_blocks]
  uint32_t* out;          // frame f at out + f * out_stride_dw
  uint64_t out_stride_dw;
  uint32_t mfw, mv_blocks, total_blocks, tiles_per_side_x, tiles_per_side_y, tiles_x, emit_tile_rows, rec_dw;
  uint32_t force_all;     // also store the zeros (records that were not emitted by the LUMA kernel)
};

__global__ __launch_bounds__(256) void wire_patch_types_kernel(PatchArgs a) {
  const uint32_t g = blockIdx.x * 256u + threadIdx.x;
  if (g >= a.total_blocks) return;
  const uint32_t t = a.types[g];
  if (t == 0 && !a.force_all) return;
  const uint32_t frame = g / a.mv_blocks, b = g - frame * a.mv_blocks;
  const uint32_t by = b / a.mfw, bx = b - by * a.mfw;
  uint32_t* out = a.out + (size_t)frame * a.out_stride_dw;
  for (uint32_t ty = by * a.tiles_per_side_y; ty < (by + 1) * a.tiles_per_side_y && ty < a.emit_tile_rows; ++ty)
    for (uint32_t tx = bx * a.tiles_per_side_x; tx < (bx + 1) * a.tiles_per_side_x; ++tx)
      out[(size_t)(ty * a.tiles_x + tx) * a.rec_dw] = t;
}

int launch_wire_patch_types(const uint32_t* d_types, uint32_t n_frames, uint32_t frame_w, uint32_t frame_h, uint32_t emit_h, uint32_t tb,
                            uint32_t mv_bw, uint32_t mv_bh, uint8_t* d_out, uint64_t out_stride, bool force_all, hipStream_t stream) {
  PatchArgs a;
  a.types = d_types;
  a.out = reinterpret_cast<uint32_t*>(d_out);
  a.out_stride_dw = out_stride / 4;
  a.mfw = frame_w / mv_bw;
  a.mv_blocks = a.mfw * (frame_h / mv_bh);
  const uint64_t total = (uint64_t)a.mv_blocks * n_frames;
  if (total == 0) return SVC_OK;
  if (total > 0x7FFFFFFFull) return fail(SVC_ERR_UNSUPPORTED, "wire_patch_types: %llu MV blocks exceed one launch", (unsigned long long)total);
  a.total_blocks = (uint32_t)total;
  a.tiles_per_side_x = mv_bw / tb; a.tiles_per_side_y = mv_bh / tb;
  a.tiles_x = frame_w / tb;
  a.emit_tile_rows = div_up(emit_h, tb);
  a.rec_dw = 1 + 3 * tb * tb;
  a.force_all = force_all ? 1u : 0u;
  hipLaunchKernelGGL(wire_patch_types_kernel, dim3(div_up(a.total_blocks, 256)), dim3(256), 0, stream, a);
  return check_launch("wire_patch_types_kernel");
}

}  // namespace svc
