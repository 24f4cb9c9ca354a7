"""Build a variant of libhrnet_hip.so with extra compile flags (measurement builds: -DHR_RING_STAMP ...).
usage: python scratch/build_variant.py NAME -DFLAG ...   ->  scratch/var_NAME/libhrnet_hip.so  (HRNET_HIP_LIB points the binding at it)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'hrnet-hand-pose-estimation_amd'))
import build as B
name, extra = sys.argv[1], sys.argv[2:]
d = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'var_' + name)
os.makedirs(d, exist_ok=True)
B.OBJ = os.path.join(d, 'build')
B.LIB = os.path.join(d, 'libhrnet_hip.so')
B.FLAGS = B.FLAGS + extra
print(B.build())
