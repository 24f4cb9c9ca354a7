"""Reference pins for core.inference.get_max_preds / get_final_preds (build container only).

    python tests/golden/make_golden_inference.py      # writes tests/golden/inference_preds.npz

The two functions are executed FROM the reference's own files: lib/core/inference.py:18-46 (get_max_preds),
:49-85 (get_final_preds) and the helpers they call in lib/utils/transforms.py:50-112 (transform_preds,
get_affine_transform, affine_transform, get_3rd_point, get_dir). The module cannot be imported as a whole - its
first import (utils.transforms) pulls in cv2, which this image does not have - so the function bodies are
compiled out of the reference files (ast, as tests/golden/make_golden_decode.py does) into a namespace that holds
numpy, math and a `cv2` object with ONE function: getAffineTransform(src, dst), restated here from OpenCV's
published definition (the 2x3 matrix M with M @ [x, y, 1] = dst for the three point pairs, solved in float64).

What that pins: get_max_preds is pure numpy -> fully pinned. get_final_preds is pinned through the reference's own
arg-max, quarter-pixel refinement and transform_preds code; the one call it cannot pin is cv2.getAffineTransform
(a 3-point linear solve; "parity unpinned" applies to that call only - DESIGN.md section 2).
Only inputs and outputs are stored.
"""
import ast
import math
import os
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get('HRNET_REFERENCE', '/root/reference')


def _get_affine_transform_3pt(src, dst):
    src = np.asarray(src, dtype=np.float64)
    dst = np.asarray(dst, dtype=np.float64)
    a = np.concatenate([src, np.ones((3, 1))], axis=1)
    return np.linalg.solve(a, dst).T            # 2 x 3, float64 like OpenCV's result


def load_functions(path, names, ns):
    tree = ast.parse(open(path).read())
    fns = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert len(fns) == len(names), (path, names)
    exec(compile(ast.Module(body=fns, type_ignores=[]), path, 'exec'), ns)
    return ns


def main():
    cv2 = types.SimpleNamespace(getAffineTransform=_get_affine_transform_3pt)
    ns = {'np': np, 'math': math, 'cv2': cv2}
    load_functions(os.path.join(REF, 'lib/utils/transforms.py'),
                   ('transform_preds', 'get_affine_transform', 'affine_transform', 'get_3rd_point', 'get_dir'), ns)
    load_functions(os.path.join(REF, 'lib/core/inference.py'), ('get_max_preds', 'get_final_preds'), ns)
    get_max_preds, get_final_preds = ns['get_max_preds'], ns['get_final_preds']

    rng = np.random.RandomState(11)
    out = {}
    # (1) get_max_preds: a square and a non-square batch; ties (first index wins), all-negative maps (zeroed
    # coordinates, max kept), an all-zero map (max 0 is NOT > 0 -> zeroed), the last pixel as the peak
    for tag, shape in (('sq', (3, 21, 16, 16)), ('rect', (2, 5, 12, 20))):
        hm = rng.randn(*shape).astype(np.float32)
        hm[0, 0] = -np.abs(hm[0, 0]) - 0.1
        hm[0, 1] = 0.0
        hm[0, 2, 3, 4] = hm[0, 2, 7, 9] = 9.0
        hm[1, 0, -1, -1] = 11.0
        hm[1, 1, 0, 0] = 11.0
        preds, maxvals = get_max_preds(hm)
        out['hm_' + tag], out['preds_' + tag], out['maxvals_' + tag] = hm, preds, maxvals
    # (2) get_final_preds with and without the quarter-pixel post-processing; peaks on the border rows / columns
    # (the `1 < px < W-1` rule skips them), zero gradients (sign 0), per-image centre / scale
    hm = np.abs(rng.randn(4, 21, 24, 18)).astype(np.float32)
    for n in range(4):
        for k in range(21):
            y, x = rng.randint(0, 24), rng.randint(0, 18)
            if k < 4:
                y, x = (0, 1, 23, 22)[k], (1, 0, 17, 16)[k]
            hm[n, k, y, x] = 8.0 + k
    hm[2, 5] = 0.0
    hm[2, 5, 10, 9] = 3.0                       # isolated peak: both differences are zero
    hm[3, 6] = -1.0                             # never positive: coordinates zeroed before the transform
    center = (rng.rand(4, 2) * 200 + 60).astype(np.float32)
    scale = np.stack([rng.rand(4) * 1.5 + 0.5] * 2, axis=1).astype(np.float32)
    scale[:, 1] *= 1.25
    out['fp_hm'], out['fp_center'], out['fp_scale'] = hm, center, scale
    for pp in (False, True):
        cfg = types.SimpleNamespace(TEST=types.SimpleNamespace(POST_PROCESS=pp))
        preds, maxvals = get_final_preds(cfg, hm.copy(), center, scale)
        out['fp_preds_pp%d' % int(pp)], out['fp_maxvals_pp%d' % int(pp)] = preds, maxvals
    path = os.path.join(HERE, 'inference_preds.npz')
    np.savez_compressed(path, **out)
    print('inference_preds.npz', os.path.getsize(path), sorted(out))


if __name__ == '__main__':
    main()
