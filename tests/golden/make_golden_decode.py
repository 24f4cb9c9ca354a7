"""Cross-check fixture for the expectation decode (build container only).

    python tests/golden/make_golden_decode.py        # writes tests/golden/decode_crosscheck.npz

The reference's decode calls un-vendored kornia (lib/utils/heatmap_decoding.py:100), which cannot run here.
The closest code the reference itself holds is integrate_tensor_2d
(lib/models/triangulation_model_utils/op.py:11-47): centre of mass with (softmax=True) a spatial softmax first,
or (softmax=False) ReLU + renormalisation. On non-negative maps that sum to one the latter is exactly the
expectation sum(x*h), sum(y*h). The function is executed from the reference file (its module cannot be imported
as a whole: relative imports of sibling files), only inputs and outputs are stored.
"""
import ast
import os

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get('HRNET_REFERENCE', '/root/reference')


def load_function(path, name):
    tree = ast.parse(open(path).read())
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == name][0]
    ns = {'torch': torch, 'nn': nn, 'np': np}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), path, 'exec'), ns)
    return ns[name]


def main():
    f = load_function(os.path.join(REF, 'lib/models/triangulation_model_utils/op.py'), 'integrate_tensor_2d')
    rng = np.random.RandomState(7)
    raw = rng.randn(3, 21, 24, 20).astype(np.float32) * 2
    pos = np.abs(rng.randn(3, 21, 24, 20)).astype(np.float32)
    pos /= pos.sum((2, 3), keepdims=True)
    c_soft, hm_soft = f(torch.from_numpy(raw), softmax=True)
    c_pos, _ = f(torch.from_numpy(pos), softmax=False)
    np.savez_compressed(os.path.join(HERE, 'decode_crosscheck.npz'), raw=raw, pos=pos,
                        coords_softmax=c_soft.numpy(), softmax_maps=hm_soft.numpy(), coords_normalised=c_pos.numpy())
    print('decode_crosscheck.npz', os.path.getsize(os.path.join(HERE, 'decode_crosscheck.npz')))


if __name__ == '__main__':
    main()
