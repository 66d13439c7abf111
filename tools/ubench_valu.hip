// ubench_valu.hip -- issue rate of the byte-SAD family and friends on gfx950.
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o /tmp/ubench && /tmp/ubench
// Each kernel runs ITERS x 32 independent-ish instructions per wave, 4 waves per SIMD on every CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define ITERS 2000

#define BODY32(STMT) STMT(0) STMT(1) STMT(2) STMT(3) STMT(4) STMT(5) STMT(6) STMT(7) \
  STMT(0) STMT(1) STMT(2) STMT(3) STMT(4) STMT(5) STMT(6) STMT(7) \
  STMT(0) STMT(1) STMT(2) STMT(3) STMT(4) STMT(5) STMT(6) STMT(7) \
  STMT(0) STMT(1) STMT(2) STMT(3) STMT(4) STMT(5) STMT(6) STMT(7)

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed) {
  uint32_t a[8], b = seed * 2654435761u + threadIdx.x, c = b ^ 0x9e3779b9u;
  uint64_t q[8];
  double d[8];
  float f[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = b + i * 77; q[i] = ((uint64_t)a[i] << 32) | (c + i); d[i] = (double)a[i]; f[i] = (float)(a[i] & 1023) + 1.5f; }
  for (int it = 0; it < ITERS; ++it) {
#define S_SAD(i) a[i] = __builtin_amdgcn_sad_u8(a[i], b, c);
#define S_SADHI(i) a[i] = __builtin_amdgcn_sad_hi_u8(a[i], b, c);
#define S_SAD16(i) a[i] = __builtin_amdgcn_sad_u16(a[i], b, c);
#define S_QSAD(i) q[i] = __builtin_amdgcn_qsad_pk_u16_u8(q[i], b, q[i]);
#define S_MQSAD(i) q[i] = __builtin_amdgcn_mqsad_pk_u16_u8(q[i], b, q[i]);
#define S_ALIGN(i) a[i] = __builtin_amdgcn_alignbyte(a[i], b, c);
#define S_PERM(i) a[i] = __builtin_amdgcn_perm(a[i], b, c);
#define S_ADD(i) a[i] = a[i] + b;
#define S_MIN3(i) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define S_LSHLOR(i) asm volatile("v_lshl_or_b32 %0, %0, 5, %1" : "+v"(a[i]) : "v"(b));
#define S_FMA64(i) d[i] = __builtin_fma(d[i], 1.0000001, 0.5);
#define S_ADD64(i) d[i] = d[i] + 0.5;
#define S_CVT64(i) d[i] = (double)f[i]; f[i] = f[i] + 1.0f;
#define S_FMA32(i) f[i] = __builtin_fmaf(f[i], 1.0000001f, 0.5f);
#define S_RCP(i) f[i] = __builtin_amdgcn_rcpf(f[i]);
#define S_DIV(i) f[i] = f[i] / 1.37f + 3.0f;
#define S_ROUND(i) f[i] = roundf(f[i] * 1.01f);
#define S_PKSUB(i) asm volatile("v_pk_sub_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define S_MADU24(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define S_DOT4(i) asm volatile("v_dot4_u32_u8 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
    if (OP == 0) { BODY32(S_SAD) }
    if (OP == 1) { BODY32(S_QSAD) }
    if (OP == 2) { BODY32(S_ALIGN) }
    if (OP == 3) { BODY32(S_ADD) }
    if (OP == 4) { BODY32(S_MIN3) }
    if (OP == 5) { BODY32(S_FMA64) }
    if (OP == 6) { BODY32(S_CVT64) }
    if (OP == 7) { BODY32(S_FMA32) }
    if (OP == 8) { BODY32(S_DIV) }
    if (OP == 9) { BODY32(S_ROUND) }
    if (OP == 10) { BODY32(S_PERM) }
    if (OP == 11) { BODY32(S_SAD16) }
    if (OP == 12) { BODY32(S_MQSAD) }
    if (OP == 13) { BODY32(S_LSHLOR) }
    if (OP == 14) { BODY32(S_ADD64) }
    if (OP == 15) { BODY32(S_PKSUB) }
    if (OP == 16) { BODY32(S_MADU24) }
    if (OP == 17) { BODY32(S_SADHI) }
    if (OP == 18) { BODY32(S_DOT4) }
    b += 1;
  }
  uint32_t r = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) r ^= a[i] ^ (uint32_t)q[i] ^ (uint32_t)(q[i] >> 32) ^ (uint32_t)d[i] ^ (uint32_t)f[i];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int OP> float run(const char* name, uint32_t* out, double ghz, int per_stmt) {
  const int blocks = 256 * 4;  // 4 workgroups of 4 waves per CU = 4 waves per SIMD
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 1u);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 2u);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // per SIMD: 4 waves x ITERS x 32 statements
  double stmts = 4.0 * ITERS * 32;
  double cyc = ms * 1e-3 * ghz * 1e9 / stmts;
  printf("%-28s %8.3f ms  %6.2f cycles per wave-statement (at %.2f GHz, %d instr/stmt)\n", name, ms, cyc, ghz, per_stmt);
  return ms;
}

int main() {
  uint32_t* out; hipMalloc(&out, 256 * 4 * 256 * 4);
  int khz = 0; hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
  double ghz = khz / 1e6; printf("clock attr %.3f GHz\n", ghz);
  run<3>("v_add_u32", out, ghz, 1);
  run<0>("v_sad_u8", out, ghz, 1);
  run<17>("v_sad_hi_u8", out, ghz, 1);
  run<11>("v_sad_u16", out, ghz, 1);
  run<1>("v_qsad_pk_u16_u8", out, ghz, 1);
  run<12>("v_mqsad_pk_u16_u8", out, ghz, 1);
  run<2>("v_alignbyte_b32", out, ghz, 1);
  run<10>("v_perm_b32", out, ghz, 1);
  run<4>("v_min3_u32", out, ghz, 1);
  run<13>("v_lshl_or_b32", out, ghz, 1);
  run<15>("v_pk_sub_u16", out, ghz, 1);
  run<16>("v_mad_u32_u24", out, ghz, 1);
  run<18>("v_dot4_u32_u8", out, ghz, 1);
  run<7>("v_fma_f32", out, ghz, 1);
  run<5>("v_fma_f64", out, ghz, 1);
  run<14>("v_add_f64", out, ghz, 1);
  run<6>("cvt_f64_f32 + add_f32", out, ghz, 2);
  run<8>("f32 IEEE div + add", out, ghz, 11);
  run<9>("roundf(mul)", out, ghz, 7);
  return 0;
}
