#!/bin/bash
# second round of experiment builds for DESIGN section 4 trap 4 (packed f32 ENABLED, SLP on): one statement of
# conv_ring's backward-statistics epilogue at a time is taken away from the vectoriser with a scalar inline-asm form
#   4: the sum(dz*y) accumulate (v_fma_f32)   5: the sum(dz) accumulate (v_add_f32)   6: the mask affine (v_fma_f32)
set -e
cd "$(dirname "$0")/.."
C=hrnet-hand-pose-estimation_amd/csrc
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -I include -I $C -Wno-unused-result"
for n in 4 5 6; do
  d=scratch/var_trap4_$n; mkdir -p $d
  cp $C/conv_ring.hip $d/conv_ring_exp.hip
  case $n in
    4) sed -i 's|              s2\[k\] = fmaf(dz, yv\[k\], s2\[k\]);|              asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s2[k]) : "v"(dz), "v"(yv[k]));|' $d/conv_ring_exp.hip ;;
    5) sed -i 's|              s1\[k\] += dz;|              asm volatile("v_add_f32 %0, %0, %1" : "+v"(s1[k]) : "v"(dz));|' $d/conv_ring_exp.hip ;;
    6) sed -i 's|              for (int k = 0; k < C::LANE_C; ++k) mv\[k\] = fmaf(yv\[k\], bsc\[k\], bsh\[k\]);|              for (int k = 0; k < C::LANE_C; ++k) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(mv[k]) : "v"(yv[k]), "v"(bsc[k]), "v"(bsh[k]));|' $d/conv_ring_exp.hip ;;
  esac
  diff <(cat $C/conv_ring.hip) $d/conv_ring_exp.hip | head -4
  /opt/rocm/bin/hipcc $F -c $d/conv_ring_exp.hip -o $d/conv_ring.o &
done
wait
for n in 4 5 6; do
  d=scratch/var_trap4_$n
  objs=$(ls $C/build/*.o | grep -v conv_ring.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/libhrnet_hip.so $objs $d/conv_ring.o
  echo built $d/libhrnet_hip.so
done
