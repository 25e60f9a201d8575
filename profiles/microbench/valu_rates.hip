// profiles/microbench/valu_rates.hip -- issue cost of the integer instructions the stage-A kernels are made of, on one SIMD:
// N dependent-free copies of one instruction per loop trip, 8 waves per SIMD, cycles per wave-instruction from the shader clock.
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip ; run: ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP 64
#define TRIPS 2000
template <int OP>
__global__ __launch_bounds__(256) void k(uint64_t* out, uint64_t seed) {
  uint64_t a = seed + threadIdx.x, b = seed * 3 + threadIdx.x, c = 0;
  uint32_t x = (uint32_t)a, y = (uint32_t)b, z = 0, s = (uint32_t)(seed & 31) | 1;
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int i = 0; i < TRIPS; ++i) {
#pragma unroll
    for (int r = 0; r < REP; ++r) {
      if (OP == 0) asm volatile("v_add_u32 %0, %1, %2" : "=v"(z) : "v"(x), "v"(y));
      if (OP == 1) asm volatile("v_lshlrev_b64 %0, %1, %2" : "=v"(c) : "v"(s), "v"(a));
      if (OP == 2) asm volatile("v_cmp_lt_u64 vcc, %0, %1" : : "v"(a), "v"(b) : "vcc");
      if (OP == 3) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(c) : "v"(x), "v"(y) : "vcc");
      if (OP == 4) asm volatile("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(c) : "v"(a), "v"(b));
      if (OP == 5) asm volatile("v_alignbit_b32 %0, %1, %2, %3" : "=v"(z) : "v"(x), "v"(y), "v"(s));
      if (OP == 6) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(z) : "v"(x), "v"(y) : "vcc");
      if (OP == 7) asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(z) : "v"(x), "v"(y));
      if (OP == 8) asm volatile("v_bfi_b32 %0, %1, %2, %3" : "=v"(z) : "v"(x), "v"(y), "v"(s));
      if (OP == 9) asm volatile("v_ffbl_b32 %0, %1" : "=v"(z) : "v"(x));
      if (OP == 10) asm volatile("v_bcnt_u32_b32 %0, %1, 0" : "=v"(z) : "v"(x));
      if (OP == 11) asm volatile("v_lshrrev_b64 %0, %1, %2" : "=v"(c) : "v"(s), "v"(a));
      if (OP == 12) asm volatile("v_cmp_eq_u64 vcc, %0, %1" : : "v"(a), "v"(b) : "vcc");
      if (OP == 13) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "=v"(z) : "v"(x));
      if (OP == 14) asm volatile("s_and_b64 vcc, vcc, exec" : : : "vcc");
      if (OP == 15) asm volatile("v_readlane_b32 s20, %0, 3" : : "v"(x) : "s20");
      if (OP == 16) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(x), "v"(y) : "vcc");
      if (OP == 17) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(z) : "v"(x), "v"(y), "v"(s));
      if (OP == 18) asm volatile("v_mul_hi_u32 %0, %1, %2" : "=v"(z) : "v"(x), "v"(y));
      if (OP == 19) asm volatile("v_mul_u32_u24 %0, %1, %2" : "=v"(z) : "v"(x), "v"(y));
      if (OP == 20) asm volatile("v_cndmask_b32 %0, %1, %2, s[20:21]" : "=v"(z) : "v"(x), "v"(y) : "s20", "s21");
      if (OP == 21) asm volatile("v_add_u32 %0, s20, %1" : "=v"(z) : "v"(x) : "s20");
      if (OP == 22) asm volatile("v_cmp_lt_u32 s[20:21], %0, %1" : : "v"(x), "v"(y) : "s20", "s21");
      if (OP == 23) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %2, %0, %1, vcc" : : "v"(x), "v"(y), "v"(z) : "vcc");
      if (OP == 24) asm volatile("v_add_co_u32 %0, vcc, %1, %2" : "=v"(z) : "v"(x), "v"(y) : "vcc");
      if (OP == 25) asm volatile("v_addc_co_u32 %0, vcc, %1, %2, vcc" : "=v"(z) : "v"(x), "v"(y) : "vcc");
      if (OP == 26) asm volatile("v_min_u32 %0, %1, %2" : "=v"(z) : "v"(x), "v"(y));
      if (OP == 27) asm volatile("v_bfe_u32 %0, %1, %2, 1" : "=v"(z) : "v"(x), "v"(s));
      if (OP == 28) asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(z) : "v"(x), "v"(y));
      if (OP == 29) asm volatile("v_cndmask_b32_e64 %0, %1, %2, vcc" : "=v"(z) : "v"(x), "v"(y) : "vcc");
      if (OP == 30) asm volatile("s_and_b64 s[20:21], s[22:23], exec" : : : "s20", "s21");
      if (OP == 31) asm volatile("v_lshl_or_b32 %0, %1, 3, %2" : "=v"(z) : "v"(x), "v"(y));
    }
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
  if (c + z == 0x123456789ULL) out[1] = c + z;
}
template <int OP> void run(const char* name, uint64_t* d) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<OP>, dim3(256 * 8), dim3(256), 0, 0, d, 12345ULL);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k<OP>, dim3(256 * 8), dim3(256), 0, 0, d, 12345ULL);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  uint64_t h = 0; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  // 8 waves per SIMD (2 blocks of 4 waves per SIMD... 8 blocks of 256 per CU): cycles per wave-instruction on one SIMD = clocks / (TRIPS * REP * 8)
  printf("%-28s %7.2f ns per wave-instruction and SIMD = %6.2f cycles at 2.4 GHz (kernel %.3f ms, s_memtime ticks %llu)\n", name, ms * 1e6 / ((double)TRIPS * REP * 8), ms * 1e6 / ((double)TRIPS * REP * 8) * 2.4, ms, (unsigned long long)h);
}
int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  uint64_t* d; hipMalloc(&d, 64);
  run<0>("v_add_u32", d); run<1>("v_lshlrev_b64", d); run<11>("v_lshrrev_b64", d); run<2>("v_cmp_lt_u64", d); run<12>("v_cmp_eq_u64", d); run<16>("v_cmp_lt_u32", d);
  run<3>("v_mad_u64_u32", d); run<4>("v_lshl_add_u64", d); run<5>("v_alignbit_b32", d); run<6>("v_cndmask_b32", d); run<7>("v_mul_lo_u32", d); run<18>("v_mul_hi_u32", d); run<19>("v_mul_u32_u24", d);
  run<8>("v_bfi_b32", d); run<9>("v_ffbl_b32", d); run<10>("v_bcnt_u32_b32", d); run<13>("v_mov_b32_dpp row_shr", d); run<17>("v_perm_b32", d); run<15>("v_readlane_b32", d);
  run<20>("v_cndmask_b32 (sgpr pair)", d); run<29>("v_cndmask_b32_e64 vcc", d); run<21>("v_add_u32 v, s, v", d); run<22>("v_cmp_lt_u32 -> sgpr pair", d); run<23>("v_cmp + v_cndmask (pair)", d);
  run<26>("v_min_u32", d); run<27>("v_bfe_u32", d); run<31>("v_lshl_or_b32", d);
  return 0;
}
