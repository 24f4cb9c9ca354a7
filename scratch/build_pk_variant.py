"""variant of libhrnet_hip.so in which the listed sources keep packed-f32 VALU instructions (the shipped build takes
them away from every file: build.py). usage: python scratch/build_pk_variant.py NAME file.hip ...  -> scratch/var_NAME/"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'hrnet-hand-pose-estimation_amd'))
import build as B
name, keep = sys.argv[1], set(sys.argv[2:])
d = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'var_' + name)
os.makedirs(d, exist_ok=True)
B.OBJ = os.path.join(d, 'build')
B.LIB = os.path.join(d, 'libhrnet_hip.so')
nopk = ['-Xclang', '-target-feature', '-Xclang', '-packed-fp32-ops']
base = [f for f in B.FLAGS]
for i in range(len(base) - 3):
    if base[i:i + 4] == nopk:
        del base[i:i + 4]
        break
B.FLAGS = base
B.EXTRA = {s: nopk for s in B.SOURCES if s not in keep}
print(B.build())
