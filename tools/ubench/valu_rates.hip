// valu_rates.hip -- issue-rate microbenchmark for the VALU instruction candidates of the SSV cell update (gfx950).
// Every measured instruction is an `asm volatile` statement: 16 independent chains per lane (chain c takes chain c+1's register as its
// second operand), 4096 iterations, 8 wavefronts per SIMD -- the loop body is exactly 16 instructions of the named opcode and the
// compiler can neither fold nor merge them (round 1 wrote the operations in C with one loop-invariant operand, and the compiler
// folded "max(max(a, b), b)" and "a + b + b + ...": the rows below 4 cycles in profiles/r01_valu_rates.txt were artefacts).
// tools/ubench/check_isa.sh counts the opcodes of every loop without a GPU.  Reports nominal cycles per wave64 instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>

#define ITERS 4096
#define CH 16

#define OP2(name, text)                                                                                     \
  __global__ void __launch_bounds__(256) name(unsigned *out, unsigned seed) {                               \
    unsigned a[CH];                                                                                         \
    for (int c = 0; c < CH; ++c) a[c] = seed + threadIdx.x * 7 + c * 0x01010101u;                           \
    for (int it = 0; it < ITERS; ++it) {                                                                    \
      _Pragma("unroll") for (int c = 0; c < CH; ++c) asm volatile(text : "+v"(a[c]) : "v"(a[(c + 1) % CH])); \
    }                                                                                                       \
    unsigned r = 0;                                                                                         \
    for (int c = 0; c < CH; ++c) r ^= a[c];                                                                 \
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;                                                         \
  }
#define OP3(name, text)                                                                                     \
  __global__ void __launch_bounds__(256) name(unsigned *out, unsigned seed) {                               \
    unsigned a[CH];                                                                                         \
    for (int c = 0; c < CH; ++c) a[c] = seed + threadIdx.x * 7 + c * 0x01010101u;                           \
    for (int it = 0; it < ITERS; ++it) {                                                                    \
      _Pragma("unroll") for (int c = 0; c < CH; ++c) asm volatile(text : "+v"(a[c]) : "v"(a[(c + 1) % CH]), "v"(a[(c + 2) % CH])); \
    }                                                                                                       \
    unsigned r = 0;                                                                                         \
    for (int c = 0; c < CH; ++c) r ^= a[c];                                                                 \
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;                                                         \
  }

OP2(k_pk_add_i16, "v_pk_add_i16 %0, %0, %1 clamp")
OP2(k_pk_max_i16, "v_pk_max_i16 %0, %0, %1")
OP2(k_pk_add_f16, "v_pk_add_f16 %0, %0, %1")
OP2(k_pk_max_f16, "v_pk_max_f16 %0, %0, %1")
OP2(k_add_u32, "v_add_u32 %0, %0, %1")
OP2(k_max_i32, "v_max_i32 %0, %0, %1")
OP2(k_add_f32, "v_add_f32 %0, %0, %1")
OP2(k_mul_f32, "v_mul_f32 %0, %0, %1")
OP2(k_max_f32, "v_max_f32 %0, %0, %1")
OP3(k_fma_f32, "v_fma_f32 %0, %0, %1, %2")
OP3(k_max3_i32, "v_max3_i32 %0, %0, %1, %2")
OP2(k_max_i16, "v_max_i16 %0, %0, %1")
OP2(k_alignbit, "v_alignbit_b32 %0, %0, %1, 16")
OP2(k_dpp_row_shr, "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
OP3(k_perm, "v_perm_b32 %0, %0, %1, %2")
OP3(k_sad_u8, "v_sad_u8 %0, %0, %1, %2")
OP3(k_add3_u32, "v_add3_u32 %0, %0, %1, %2")
// round 3: candidates for a 1.5-op SSV row (f16 holds k/256, k = 0..256, exactly; clamp = [0, 1] = the byte arithmetic's floor and ceiling)
OP2(k_pk_add_f16_clamp, "v_pk_add_f16 %0, %0, %1 clamp")
OP3(k_pk_maximum3_f16, "v_pk_maximum3_f16 %0, %0, %1, %2")
OP3(k_pk_minimum3_f16, "v_pk_minimum3_f16 %0, %0, %1, %2")
OP3(k_maximum3_f32, "v_maximum3_f32 %0, %0, %1, %2")
OP3(k_max3_f32, "v_max3_f32 %0, %0, %1, %2")
OP3(k_max3_i16, "v_max3_i16 %0, %0, %1, %2")
OP2(k_pk_add_u16_clamp, "v_pk_add_u16 %0, %0, %1 clamp")
OP2(k_pk_sub_u16_clamp, "v_pk_sub_u16 %0, %0, %1 clamp")
OP2(k_pk_max_u16, "v_pk_max_u16 %0, %0, %1")
OP3(k_pk_mad_i16, "v_pk_mad_i16 %0, %0, %1, %2")
OP3(k_pk_fma_f16, "v_pk_fma_f16 %0, %0, %1, %2")
// the two row bodies side by side: 2 rows x (add, max) in packed i16 against 2 rows x add + one max3 in packed f16
#define MIX(name, body)                                                                                    \
  __global__ void __launch_bounds__(256) name(unsigned *out, unsigned seed) {                               \
    unsigned a[CH], m[CH];                                                                                  \
    for (int c = 0; c < CH; ++c) { a[c] = seed + threadIdx.x * 7 + c * 0x01010101u; m[c] = a[c] ^ 0x5a5au; } \
    for (int it = 0; it < ITERS / 2; ++it) {                                                                \
      _Pragma("unroll") for (int c = 0; c < CH; ++c) { unsigned t; asm volatile(body : "+v"(a[c]), "+v"(m[c]), "=&v"(t) : "v"(a[(c + 1) % CH])); } \
    }                                                                                                       \
    unsigned r = 0;                                                                                         \
    for (int c = 0; c < CH; ++c) r ^= a[c] ^ m[c];                                                          \
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;                                                         \
  }
MIX(k_mix_i16, "v_pk_add_i16 %2, %0, %3 clamp\n v_pk_max_i16 %1, %1, %2\n v_pk_add_i16 %0, %2, %3 clamp\n v_pk_max_i16 %1, %1, %0")
MIX(k_mix_f16, "v_pk_add_f16 %2, %0, %3 clamp\n v_pk_add_f16 %0, %2, %3 clamp\n v_pk_maximum3_f16 %1, %1, %2, %0")

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
  printf("device %s, %d CUs, nominal %.0f MHz; %d chains x %d iterations x 8 waves/SIMD, asm volatile (nothing folded: tools/ubench/check_isa.sh)\n",
         p.gcnArchName, p.multiProcessorCount, clk / 1e6, CH, ITERS);
#define RUN(kern, label)                                                                                         \
  {                                                                                                              \
    double ms = timeit([&] { hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 1u); });               \
    double winst = (double)blocks * 4 * ITERS * CH;                                                              \
    printf("%-26s %8.3f ms  => %.2f cycles per wave64 instruction per SIMD\n", label, ms,                        \
           1.0 / (winst / (ms * 1e-3) / (p.multiProcessorCount * 4) / clk));                                     \
  }
  RUN(k_pk_add_i16, "v_pk_add_i16 clamp") RUN(k_pk_max_i16, "v_pk_max_i16") RUN(k_pk_add_f16, "v_pk_add_f16") RUN(k_pk_max_f16, "v_pk_max_f16")
  RUN(k_add_u32, "v_add_u32") RUN(k_max_i32, "v_max_i32") RUN(k_add_f32, "v_add_f32") RUN(k_mul_f32, "v_mul_f32") RUN(k_max_f32, "v_max_f32")
  RUN(k_fma_f32, "v_fma_f32") RUN(k_max3_i32, "v_max3_i32") RUN(k_max_i16, "v_max_i16") RUN(k_alignbit, "v_alignbit_b32")
  RUN(k_dpp_row_shr, "v_mov_b32_dpp row_shr:1") RUN(k_perm, "v_perm_b32") RUN(k_sad_u8, "v_sad_u8") RUN(k_add3_u32, "v_add3_u32")
  RUN(k_pk_add_f16_clamp, "v_pk_add_f16 clamp") RUN(k_pk_maximum3_f16, "v_pk_maximum3_f16") RUN(k_pk_minimum3_f16, "v_pk_minimum3_f16")
  RUN(k_maximum3_f32, "v_maximum3_f32") RUN(k_max3_f32, "v_max3_f32") RUN(k_max3_i16, "v_max3_i16")
  RUN(k_pk_add_u16_clamp, "v_pk_add_u16 clamp") RUN(k_pk_sub_u16_clamp, "v_pk_sub_u16 clamp") RUN(k_pk_max_u16, "v_pk_max_u16")
  RUN(k_pk_mad_i16, "v_pk_mad_i16") RUN(k_pk_fma_f16, "v_pk_fma_f16")
#define RUNMIX(kern, label, ninst)                                                                               \
  {                                                                                                              \
    double ms = timeit([&] { hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 1u); });               \
    double rows = (double)blocks * 4 * (ITERS / 2) * CH * 2;                                                     \
    printf("%-26s %8.3f ms  => %.2f cycles per (register, row) per SIMD  (%d instructions per 2 rows)\n", label, ms, \
           1.0 / (rows / (ms * 1e-3) / (p.multiProcessorCount * 4) / clk), ninst);                               \
  }
  RUNMIX(k_mix_i16, "row body packed i16", 4) RUNMIX(k_mix_f16, "row body packed f16+max3", 3)
  return 0;
}
