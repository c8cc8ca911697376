"""Caption metrics by delegation -- counterpart of reference ``model/evaluation.py:9-76`` (``my_COCOEvalCap``: BLEU-1..4, METEOR,
ROUGE-L, CIDEr through ``pycocoevalcap``) and of the "best checkpoint by BLEU-1" rule (``train_mimic.py:364-383``).

The metric code itself is third party (``pycocoevalcap``; its PTB tokenizer and METEOR need Java) and outside the MI355X path:
when the package is importable the scores are computed with ITS scorers on the greedy-decoded sentences; when it is not (this
build image), ``caption_metrics`` returns None and the callers fall back to exact-match accuracy (``evaluate_score.acc``).
"""


def available():
    try:
        import pycocoevalcap.bleu.bleu    # noqa: F401
        return True
    except Exception:      # noqa: BLE001  (ImportError, or a broken install)
        return False


def _tokenize(res, gts):
    """PTB tokenisation like the reference when Java is there, lower-cased whitespace tokens otherwise."""
    try:
        from pycocoevalcap.tokenizer.ptbtokenizer import PTBTokenizer
        tok = PTBTokenizer()
        return (tok.tokenize({k: [{'caption': c} for c in v] for k, v in res.items()}),
                tok.tokenize({k: [{'caption': c} for c in v] for k, v in gts.items()}))
    except Exception:      # noqa: BLE001
        def norm(d):
            return {k: [' '.join(c.lower().split()) for c in v] for k, v in d.items()}
        return norm(res), norm(gts)


def caption_metrics(predictions, references):
    """``predictions``: {id: sentence}; ``references``: {id: sentence or [sentences]}.  Returns {'Bleu_1'.., 'ROUGE_L', 'CIDEr'
    [, 'METEOR']} computed by pycocoevalcap's scorers, or None when the package is absent."""
    if not available():
        return None
    from pycocoevalcap.bleu.bleu import Bleu
    from pycocoevalcap.cider.cider import Cider
    from pycocoevalcap.rouge.rouge import Rouge
    ids = [k for k in predictions if k in references]
    res = {k: [predictions[k]] for k in ids}
    gts = {k: (references[k] if isinstance(references[k], (list, tuple)) else [references[k]]) for k in ids}
    res, gts = _tokenize(res, gts)
    out = {}
    scorers = [(Bleu(4), ['Bleu_1', 'Bleu_2', 'Bleu_3', 'Bleu_4']), (Rouge(), 'ROUGE_L'), (Cider(), 'CIDEr')]
    try:
        from pycocoevalcap.meteor.meteor import Meteor
        scorers.insert(1, (Meteor(), 'METEOR'))
    except Exception:      # noqa: BLE001  (needs Java)
        pass
    for scorer, name in scorers:
        try:
            score, _ = scorer.compute_score(gts, res)
        except Exception:  # noqa: BLE001  (METEOR without a JVM)
            continue
        if isinstance(name, list):
            out.update({n: float(s) for n, s in zip(name, score)})
        else:
            out[name] = float(score)
    return out
