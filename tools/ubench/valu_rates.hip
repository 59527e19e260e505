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
  return 0;
}
