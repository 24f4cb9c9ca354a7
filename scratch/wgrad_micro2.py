import sys
sys.path.insert(0, '/root/repo/scratch')
import importlib, types
src = open('/root/repo/scratch/wgrad_micro.py').read()
src = src[:src.index("case(64, 64, 32, 32, 3, splits=(384")]
exec(src)
for c in [(64, 64, 32, 32, 3, 1), (64, 32, 64, 64, 3, 1), (64, 16, 128, 128, 3, 1), (64, 8, 256, 256, 3, 1), (64, 64, 64, 64, 3, 1),
          (64, 64, 256, 64, 1, 1), (64, 64, 32, 64, 3, 2), (64, 32, 64, 128, 3, 2), (64, 16, 128, 256, 3, 2)]:
    case(*c[:5], stride=c[5])
