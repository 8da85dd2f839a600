from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


def _collect(loader, model, n_classes=10):
    if loader is None:
        return torch.zeros([0, n_classes]), torch.zeros([0], dtype=torch.int64)
    dev = next(model.parameters()).device
    probs, labels = [], []
    model.eval()
    with torch.no_grad():
        for x, y in loader:
            probs.append(F.softmax(model(x.to(dev)), dim=-1).cpu())
            labels.append(y.cpu())
    return torch.cat(probs), torch.cat(labels)


def _entropy(p):
    return -torch.where(p > 0, p * p.log(), torch.zeros_like(p)).sum(-1)


def _m_entropy(p, labels):
    """Reference m_entropy (SVC_MIA.py:12-22) as written: the columns named by ANY label in the
    batch are swapped to (1-p, log p) for every row; both log terms are log p (floored)."""
    floor = torch.tensor(1e-30).log()
    logp = torch.where(p > 0, p.log(), floor)
    mod_p, mod_logp = p.clone(), logp.clone()
    cols = labels.long()
    mod_p[:, cols] = (1 - p)[:, cols]
    mod_logp[:, cols] = logp[:, cols]
    return -(mod_p * mod_logp).sum(-1)


def _svc_acc(shadow_in, shadow_out, target_in, target_out):
    from sklearn.svm import SVC
    X = torch.cat([shadow_in, shadow_out]).numpy().reshape(len(shadow_in) + len(shadow_out), -1)
    Y = np.concatenate([np.ones(len(shadow_in)), np.zeros(len(shadow_out))])
    clf = SVC(C=3, gamma="auto", kernel="rbf").fit(X, Y)
    accs = []
    if len(target_in):
        accs.append(clf.predict(target_in.numpy().reshape(len(target_in), -1)).mean())
    if len(target_out):
        accs.append(1 - clf.predict(target_out.numpy().reshape(len(target_out), -1)).mean())
    return float(np.mean(accs))


def SVC_MIA(shadow_train, target_train, target_test, shadow_test, model):
    """dict(correctness, confidence, entropy, m_entropy, prob) of attack accuracies; the paper's
    MIA number is result['confidence'] * 100 with target_test = forget set."""
    sets = {k: _collect(l, model) for k, l in dict(st=shadow_train, so=shadow_test, tt=target_train,
                                                   to=target_test).items()}

    def feat(fn):
        return [fn(*sets[k]).float() for k in ("st", "so", "tt", "to")]

    features = {
        "correctness": lambda p, y: (p.argmax(1) == y).int() if len(p) else torch.zeros(0),
        "confidence": lambda p, y: p.gather(1, y[:, None].long()) if len(p) else torch.zeros(0, 1),
        "entropy": lambda p, y: _entropy(p),
        "m_entropy": lambda p, y: _m_entropy(p, y) if len(p) else torch.zeros(0),
        "prob": lambda p, y: p,
    }
    return {name: _svc_acc(*feat(fn)) for name, fn in features.items()}
