"""Exact-match answer accuracy (SURVEY f-3): total / open / closed, as the reference's ``acc``
(``evaluate_score.py:216-251``): pairs are aligned by position, questions containing 'what has changed' (the
difference-captioning items) are skipped, an answer is "closed" when the ground truth is 'yes' or 'no', and a
prediction counts only if the decoded string equals the ground truth exactly.  Host-side string logic, as in the
reference; the language metrics (BLEU/METEOR/ROUGE/CIDEr, ``evaluation.py:9-76``) stay delegated to pycocoevalcap."""
import json


def acc_from_lists(gt_annotations, predictions):
    """``gt_annotations``: list of {'question': str, 'caption': str}; ``predictions``: list of {'caption': str}
    in the same order.  Returns (total, open, closed) accuracies; a class with no items gives float('nan')
    (the reference divides by zero there)."""
    if len(predictions) < len(gt_annotations):
        raise ValueError('acc: %d predictions for %d ground-truth items' % (len(predictions), len(gt_annotations)))
    total = opened = closed = 0
    total_ok = open_ok = closed_ok = 0
    for g, p in zip(gt_annotations, predictions):
        if 'what has changed' in g['question']:
            continue
        is_closed = g['caption'] in ('yes', 'no')
        hit = g['caption'] == p['caption']
        total += 1
        total_ok += hit
        if is_closed:
            closed += 1
            closed_ok += hit
        else:
            opened += 1
            open_ok += hit
    div = lambda a, b: a / b if b else float('nan')
    return div(total_ok, total), div(open_ok, opened), div(closed_ok, closed)


def acc(results_file, annotation_file):
    """File form: ``annotation_file`` = COCO-style json with an 'annotations' list, ``results_file`` = the json written
    by ``utils.coco_gen_format_save`` (list of {'image_id', 'caption'})."""
    with open(annotation_file) as f:
        gt = json.load(f)['annotations']
    with open(results_file) as f:
        pr = json.load(f)
    return acc_from_lists(gt, pr)
