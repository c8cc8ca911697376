#!/usr/bin/env python3
"""Pin f-3 (exact-match total / open / closed accuracy) against the reference's own ``acc()``
(``/root/reference/model/evaluate_score.py:216-251``): generate ground-truth / prediction json pairs, run the imported
reference function on them, store inputs + outputs as ``tests/golden/acc_golden.json``.
Shims (environment only): ``pycocotools`` and the reference's ``evaluation`` module (pycocoevalcap + Java) are stubbed in
``sys.modules`` -- ``acc()`` touches neither.   Re-run:  python tests/golden/make_acc_golden.py"""
import contextlib
import io
import json
import os
import random
import sys
import tempfile
import types

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    for name in ('pycocotools', 'pycocotools.coco', 'evaluation'):
        sys.modules[name] = types.ModuleType(name)
    sys.modules['pycocotools.coco'].COCO = object
    sys.modules['evaluation'].my_COCOEvalCap = object
    sys.path.insert(0, '/root/reference/model')
    import evaluate_score as R
    rng = random.Random(5)
    words = ['yes', 'no', 'the main image has an additional finding of pleural effusion than the reference image',
             'left lung', 'cardiomegaly', 'nothing has changed', 'mild', 'right lower lung zone', 'atelectasis']
    qs = ['is there any pneumonia ?', 'what has changed compared to the reference image ?', 'where is the opacity ?',
          'what abnormalities are seen ?', 'is the heart enlarged ?', 'what level is the edema ?']
    cases = []
    tmp = tempfile.mkdtemp()
    for c in range(4):
        n = [12, 40, 7, 25][c]
        gt, pr = [], []
        for i in range(n):
            q = rng.choice(qs)
            a = rng.choice(words[:2]) if q.startswith('is ') else rng.choice(words[2:])
            p = a if rng.random() < [0.5, 0.8, 0.3, 1.0][c] else rng.choice(words)
            gt.append({'image_id': str(i), 'question': q, 'caption': a, 'id': i})
            pr.append({'image_id': str(i), 'caption': p})
        gpath, ppath = os.path.join(tmp, 'gt%d.json' % c), os.path.join(tmp, 'pr%d.json' % c)
        json.dump({'annotations': gt, 'images': [{'id': str(i)} for i in range(n)]}, open(gpath, 'w'))
        json.dump(pr, open(ppath, 'w'))
        R.annotation_file = gpath
        with contextlib.redirect_stdout(io.StringIO()):
            total, opened, closed = R.acc(ppath)
        cases.append({'annotations': gt, 'predictions': pr, 'expected': [total, opened, closed]})
    json.dump(cases, open(os.path.join(HERE, 'acc_golden.json'), 'w'))
    print('wrote acc_golden.json', [c['expected'] for c in cases])


if __name__ == '__main__':
    main()
