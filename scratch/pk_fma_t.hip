// Experiment for DESIGN section 4, trap 4: the instruction sequence hipcc 7.2 (SLP on) emits in conv_ring's
// backward-statistics epilogue, isolated (registers and order as in the kernel's ISA):
//   v_cndmask v170 / v171 (dz pair)  ->  v_pk_fma_f32 v[154:155], v[170:171], v[162:163], v[154:155]
//   op_sel:[0,1,0] op_sel_hi:[1,0,1]  ->  v_cndmask v162 ; s_or ; v_cndmask v163   (the y pair's registers re-used for
//   the next pair's dz two and three instructions later)
// with known inputs, 2 waves per SIMD. Variants: NOPS s_nop between the packed FMA and the first overwrite; OPSEL = 0:
// the same arithmetic with the y pair stored low-half-first and no op_sel; MFMA = 1: a burst of eight
// v_mfma_f32_16x16x32_bf16 issued by the same wave right before the sequence (the matrix pipe still busy).
// Prints how many results came out wrong per (quarter-wave, half).
// build: hipcc --offload-arch=gfx950 -O2 scratch/pk_fma_t.hip -o scratch/pk_fma_t
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int NOPS, int OPSEL, int MFMA, int OVERWRITE>
__global__ void k(unsigned* bad, int iters) {
  const int lane = threadIdx.x & 63;
  unsigned nb[2] = {0, 0};
  for (int it = 0; it < iters; ++it) {
    const float dz0 = (float)(lane + 1 + (it & 7)), dz1 = (float)(lane + 101), y0 = 2.f + (it & 3), y1 = 3.f, a0 = 0.5f, a1 = 0.25f;
    float r0, r1;
    asm volatile(
        "s_mov_b64 s[78:79], -1\n\t"
        "v_mov_b32 v34, %2\n\t"            // vals[0], vals[1] (what the masks select)
        "v_mov_b32 v35, %3\n\t"
        "v_mov_b32 v36, 0x4e6e6b28\n\t"    // vals[2], vals[3]: 1e9 - what lands in v162 / v163 afterwards
        "v_mov_b32 v37, 0x4e6e6b28\n\t"
        "v_mov_b32 v170, 1.0\n\t"          // mask values (> 0)
        "v_mov_b32 v171, 1.0\n\t"
        "v_mov_b32 v168, 1.0\n\t"
        "v_mov_b32 v169, 1.0\n\t"
        ".if %c9\n\t"
        "v_mov_b32 v162, %5\n\t"           // kernel order: v162 = y1 (high half of the bf16 pair), v163 = y0
        "v_mov_b32 v163, %4\n\t"
        ".else\n\t"
        "v_mov_b32 v162, %4\n\t"
        "v_mov_b32 v163, %5\n\t"
        ".endif\n\t"
        "v_mov_b32 v154, %6\n\t"
        "v_mov_b32 v155, %7\n\t"
        ".if %c10\n\t"
        "v_mov_b32 v120, 0\n\tv_mov_b32 v121, 0\n\tv_mov_b32 v122, 0\n\tv_mov_b32 v123, 0\n\t"
        "v_mov_b32 v124, 0\n\tv_mov_b32 v125, 0\n\tv_mov_b32 v126, 0\n\tv_mov_b32 v127, 0\n\t"
        "s_nop 3\n\t"
        ".rept 8\n\t"
        "v_mfma_f32_16x16x32_bf16 v[128:131], v[120:123], v[124:127], v[128:131]\n\t"
        ".endr\n\t"
        ".endif\n\t"
        "v_cmp_lt_f32_e32 vcc, 0, v171\n\t"
        "v_cmp_lt_f32_e64 s[0:1], 0, v170\n\t"
        "s_or_b64 vcc, s[78:79], vcc\n\t"
        "v_cndmask_b32_e32 v170, 0, v34, vcc\n\t"
        "s_or_b64 vcc, s[78:79], s[0:1]\n\t"
        "v_cndmask_b32_e32 v171, 0, v35, vcc\n\t"
        "v_cmp_lt_f32_e32 vcc, 0, v169\n\t"
        "v_cmp_lt_f32_e64 s[0:1], 0, v168\n\t"
        "s_or_b64 vcc, s[78:79], vcc\n\t"
        ".if %c9\n\t"
        "v_pk_fma_f32 v[154:155], v[170:171], v[162:163], v[154:155] op_sel:[0,1,0] op_sel_hi:[1,0,1]\n\t"
        ".else\n\t"
        "v_pk_fma_f32 v[154:155], v[170:171], v[162:163], v[154:155]\n\t"
        ".endif\n\t"
        ".rept %c8\n\t"
        "s_nop 0\n\t"
        ".endr\n\t"
        ".if %c11\n\t"
        "v_cndmask_b32_e32 v162, 0, v36, vcc\n\t"
        "s_or_b64 vcc, s[78:79], s[0:1]\n\t"
        "v_cndmask_b32_e32 v163, 0, v37, vcc\n\t"
        ".endif\n\t"
        "s_nop 7\n\t"
        "v_mov_b32 %0, v154\n\t"
        "v_mov_b32 %1, v155\n\t"
        : "=v"(r0), "=v"(r1)
        : "v"(dz0), "v"(dz1), "v"(y0), "v"(y1), "v"(a0), "v"(a1), "n"(NOPS), "n"(OPSEL), "n"(MFMA), "n"(OVERWRITE)
        : "v34", "v35", "v36", "v37", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130",
          "v131", "v154", "v155", "v162", "v163", "v168", "v169", "v170", "v171", "s0", "s1", "s78", "s79", "vcc", "memory");
    if (r0 != a0 + dz0 * y0) ++nb[0];
    if (r1 != a1 + dz1 * y1) ++nb[1];
  }
  if (nb[0]) atomicAdd(bad + (lane >> 4) * 2 + 0, nb[0]);
  if (nb[1]) atomicAdd(bad + (lane >> 4) * 2 + 1, nb[1]);
}

template <int NOPS, int OPSEL, int MFMA, int OVERWRITE>
void run(const char* name) {
  unsigned* d; (void)hipMalloc(&d, 8 * 4); (void)hipMemset(d, 0, 8 * 4);
  hipLaunchKernelGGL((k<NOPS, OPSEL, MFMA, OVERWRITE>), dim3(1024), dim3(512), 0, 0, d, 4000);
  (void)hipDeviceSynchronize();
  unsigned h[8]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-66s wrong (lo, hi) per quarter-wave: ", name);
  for (int q = 0; q < 4; ++q) printf("[%u, %u] ", h[q * 2], h[q * 2 + 1]);
  printf("\n");
  (void)hipFree(d);
}

int main() {
  run<0, 1, 0, 1>("op_sel, sources overwritten 1 and 3 instructions later");
  run<0, 1, 1, 1>("op_sel, overwritten, eight MFMAs of the same wave just before");
  run<0, 1, 0, 0>("op_sel, sources left alone");
  run<0, 1, 1, 0>("op_sel, sources left alone, MFMAs before");
  run<1, 1, 1, 1>("op_sel, overwritten, MFMAs before, 1 s_nop behind the packed FMA");
  run<2, 1, 1, 1>("op_sel, overwritten, MFMAs before, 2 s_nop");
  run<4, 1, 1, 1>("op_sel, overwritten, MFMAs before, 4 s_nop");
  run<0, 0, 0, 1>("no op_sel (pair stored low-first), overwritten");
  run<0, 0, 1, 1>("no op_sel, overwritten, MFMAs before");
  return 0;
}
