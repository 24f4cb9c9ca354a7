// Experiment for DESIGN "compiler trap (4)": is the store data of a ds_write_b128 safe against a VALU instruction that
// overwrites its registers a few instructions later while the LDS queue is busy (the sequence hipcc 7.2 emits in
// conv_ring's statistics epilogue when SLP packs the final shuffle adds)?
//   16 x ds_bpermute_b32 (queue the LDS unit)  ->  ds_write_b128 A, v[100:103]  (exec = lanes 0,16,32,48)
//   -> [NOPS x s_nop 0] -> v_pk_add_f32 v[100:101], ...  (full or same exec)  -> s_waitcnt lgkmcnt(0) -> read back.
// Prints, per variant, how many of ITER x waves stored values came back overwritten.
// build: hipcc --offload-arch=gfx950 -O2 scratch/lds_war_t.hip -o scratch/lds_war_t
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int NOPS, int NPERM, int FULLEXEC, int B128>
__global__ void k(unsigned* bad, int iters) {
  __shared__ float lds[64 * 4 * 8 + 64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  unsigned nbad = 0;
  for (int it = 0; it < iters; ++it) {
    const float base = (float)(it * 8 + wv + 1);
    const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)(lds + wv * 256 + (lane >> 4) * 4);
    const unsigned baddr = (unsigned)((lane ^ 8) << 2);
    float r0, r1, r2, r3;
    asm volatile(
        "v_mov_b32 v100, %4\n\t"
        "v_add_f32 v101, 1.0, %4\n\t"
        "v_add_f32 v102, 2.0, %4\n\t"
        "v_add_f32 v103, 3.0, %4\n\t"
        "v_mov_b32 v104, 0x4e6e6b28\n\t"      // 1e9: what the overwrite adds
        "v_mov_b32 v105, 0x4e6e6b28\n\t"
        "v_mov_b32 v106, 0\n\t"
        "v_mov_b32 v107, 0\n\t"
        ".rept %c7\n\t"
        "ds_bpermute_b32 v110, %6, v104\n\t"
        ".endr\n\t"
        "s_mov_b64 s[20:21], exec\n\t"
        "s_mov_b32 s22, 0x00010001\n\t"
        "s_mov_b32 s23, 0x00010001\n\t"
        "s_mov_b64 exec, s[22:23]\n\t"
        ".if %c9\n\t"
        "ds_write_b128 %5, v[100:103]\n\t"
        ".else\n\t"
        "ds_write_b64 %5, v[100:101]\n\t"
        ".endif\n\t"
        ".if %c8\n\t"
        "s_mov_b64 exec, s[20:21]\n\t"
        ".endif\n\t"
        ".rept %c10\n\t"
        "s_nop 0\n\t"
        ".endr\n\t"
        "v_pk_add_f32 v[100:101], v[104:105], v[106:107]\n\t"
        "s_mov_b64 exec, s[20:21]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "ds_read_b128 v[112:115], %5\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_mov_b32 %0, v112\n\t"
        "v_mov_b32 %1, v113\n\t"
        "v_mov_b32 %2, v114\n\t"
        "v_mov_b32 %3, v115\n\t"
        : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)
        : "v"(base), "v"(addr), "v"(baddr), "n"(NPERM), "n"(FULLEXEC), "n"(B128), "n"(NOPS)
        : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v110", "v112", "v113", "v114", "v115",
          "s20", "s21", "s22", "s23", "memory");
    if ((lane & 15) == 0) {
      if (r0 != base) nbad |= 1u << (0 + 4 * (lane >> 4));
      if (r1 != base + 1.f) nbad |= 1u << (1 + 4 * (lane >> 4));
      if (B128 && r2 != base + 2.f) nbad |= 1u << (2 + 4 * (lane >> 4));
      if (B128 && r3 != base + 3.f) nbad |= 1u << (3 + 4 * (lane >> 4));
      if (nbad) atomicAdd(bad + 16, 1u);
    }
    __syncthreads();
  }
  if (nbad) for (int b = 0; b < 16; ++b) if (nbad & (1u << b)) atomicAdd(bad + b, 1u);
}

template <int NOPS, int NPERM, int FULLEXEC, int B128>
void run(const char* name) {
  unsigned* d; hipMalloc(&d, 17 * 4); hipMemset(d, 0, 17 * 4);
  hipLaunchKernelGGL((k<NOPS, NPERM, FULLEXEC, B128>), dim3(1024), dim3(512), 0, 0, d, 2000);
  hipDeviceSynchronize();
  unsigned h[17]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-44s overwritten stores %u of %u;  lanes-with-bad per (lane16 x dword): ", name, h[16], 1024u * 8 * 2000 * 4);
  for (int b = 0; b < 16; ++b) printf("%u ", h[b]);
  printf("\n");
  hipFree(d);
}

int main() {
  run<0, 16, 1, 1>("b128, 16 bpermute queued, 0 nop, full exec");
  run<0, 16, 0, 1>("b128, 16 bpermute queued, 0 nop, same exec");
  run<1, 16, 1, 1>("b128, 16 bpermute queued, 1 nop");
  run<2, 16, 1, 1>("b128, 16 bpermute queued, 2 nop");
  run<4, 16, 1, 1>("b128, 16 bpermute queued, 4 nop");
  run<8, 16, 1, 1>("b128, 16 bpermute queued, 8 nop");
  run<0, 0, 1, 1>("b128, idle LDS queue, 0 nop");
  run<0, 16, 1, 0>("b64, 16 bpermute queued, 0 nop");
  run<0, 0, 1, 0>("b64, idle LDS queue, 0 nop");
  return 0;
}
