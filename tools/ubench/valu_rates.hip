// valu_rates.hip -- issue-rate microbenchmark for the VALU instruction candidates of the SSV cell update
// (gfx950).  Each kernel runs N dependent-free chains per lane; reports wave-instructions per cycle per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
#define ITERS 4096
#define CH 16

template <int KIND>
__global__ void __launch_bounds__(256) k(unsigned *out, unsigned seed) {
  unsigned a[CH];
  for (int c = 0; c < CH; ++c) a[c] = seed + threadIdx.x * 7 + c;
  unsigned b = seed * 3 + threadIdx.x;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      if (KIND == 0) a[c] = __builtin_bit_cast(unsigned, __builtin_elementwise_add_sat(__builtin_bit_cast(s16x2, a[c]), __builtin_bit_cast(s16x2, b)));
      if (KIND == 1) a[c] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, a[c]), __builtin_bit_cast(s16x2, b)));
      if (KIND == 2) a[c] = __builtin_bit_cast(unsigned, __builtin_bit_cast(h2, a[c]) + __builtin_bit_cast(h2, b));
      if (KIND == 3) a[c] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(h2, a[c]), __builtin_bit_cast(h2, b)));
      if (KIND == 4) a[c] = a[c] + b;
      if (KIND == 5) a[c] = (unsigned)max((int)a[c], (int)b);
      if (KIND == 6) a[c] = __builtin_bit_cast(unsigned, __builtin_bit_cast(float, a[c]) + __builtin_bit_cast(float, b));
      if (KIND == 7) a[c] = __builtin_bit_cast(unsigned, fmaxf(__builtin_bit_cast(float, a[c]), __builtin_bit_cast(float, b)));
      if (KIND == 8) a[c] = (unsigned)max(max((int)a[c], (int)b), (int)a[(c + 1) % CH]);          // v_max3_i32
      if (KIND == 9) a[c] = __builtin_bit_cast(unsigned, fmaxf(fmaxf(__builtin_bit_cast(float, a[c]), __builtin_bit_cast(float, b)), __builtin_bit_cast(float, a[(c + 1) % CH])));  // v_max3_f32
      if (KIND == 10) a[c] = (unsigned)(unsigned short)max((short)a[c], (short)b);  // v_max_i16
      if (KIND == 11) a[c] = __builtin_amdgcn_alignbit(a[c], b, 16);
      if (KIND == 12) a[c] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)a[c], 0x111, 0xf, 0xf, true);
      if (KIND == 13) a[c] = __builtin_amdgcn_perm(a[c], b, 0x05040100);
      if (KIND == 14) a[c] = __builtin_amdgcn_sad_u8(a[c], b, a[(c + 1) % CH]);
      if (KIND == 15) a[c] = a[c] + b + a[(c + 1) % CH];   // v_add3_u32
    }
  }
  unsigned r = 0;
  for (int c = 0; c < CH; ++c) r ^= a[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int KIND>
__global__ void __launch_bounds__(256) kpk32(unsigned *out, unsigned seed) {   // packed f32 pairs
  f2 a[CH / 2];
  for (int c = 0; c < CH / 2; ++c) a[c] = f2{(float)(seed + c), (float)(threadIdx.x)};
  f2 b = f2{1.5f, 0.25f};
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int c = 0; c < CH / 2; ++c) {
      if (KIND == 0) a[c] = a[c] + b;                                     // v_pk_add_f32
      if (KIND == 1) a[c] = __builtin_elementwise_max(a[c], b);           // v_pk_max? (may scalarise)
    }
  }
  float r = 0;
  for (int c = 0; c < CH / 2; ++c) r += a[c].x + a[c].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = __builtin_bit_cast(unsigned, r);
}

template <class F>
double timeit(F launch) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); hipDeviceSynchronize();
  hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int blocks = p.multiProcessorCount * 8;      // 8 blocks x 4 waves = 8 waves/SIMD
  unsigned *out; hipMalloc(&out, (size_t)blocks * 256 * 4);
  const double clk = p.clockRate * 1e3;              // Hz (nominal)
  const char *names[] = {"v_pk_add_i16 clamp", "v_pk_max_i16", "v_pk_add_f16", "v_pk_max_f16", "v_add_u32", "v_max_i32", "v_add_f32", "v_max_f32",
                         "v_max3_i32", "v_max3_f32", "v_max_i16", "v_alignbit_b32", "v_mov_b32_dpp row_shr", "v_perm_b32", "v_sad_u8", "v_add3_u32"};
  printf("device %s, %d CUs, nominal %.0f MHz\n", p.gcnArchName, p.multiProcessorCount, clk / 1e6);
#define RUN(K)                                                                                                   \
  {                                                                                                              \
    double ms = timeit([&] { hipLaunchKernelGGL(k<K>, dim3(blocks), dim3(256), 0, 0, out, 1u); });               \
    double winst = (double)blocks * 4 * ITERS * CH;                                                              \
    printf("%-26s %8.3f ms  %.3f wave-instr/clk/SIMD (nominal clk)  => %.2f cycles per wave64 instr\n", names[K], ms, \
           winst / (ms * 1e-3) / (p.multiProcessorCount * 4) / clk, 1.0 / (winst / (ms * 1e-3) / (p.multiProcessorCount * 4) / clk)); \
  }
  RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12) RUN(13) RUN(14) RUN(15)
  {
    double ms = timeit([&] { hipLaunchKernelGGL(kpk32<0>, dim3(blocks), dim3(256), 0, 0, out, 1u); });
    double winst = (double)blocks * 4 * ITERS * (CH / 2);
    printf("%-26s %8.3f ms  => %.2f cycles per wave64 instr (2 f32 per lane)\n", "v_pk_add_f32", ms, 1.0 / (winst / (ms * 1e-3) / (p.multiProcessorCount * 4) / clk));
    ms = timeit([&] { hipLaunchKernelGGL(kpk32<1>, dim3(blocks), dim3(256), 0, 0, out, 1u); });
    printf("%-26s %8.3f ms  => %.2f cycles per (pair of f32 max)\n", "pk max f32 (as compiled)", ms, 1.0 / (winst / (ms * 1e-3) / (p.multiProcessorCount * 4) / clk));
  }
  return 0;
}
