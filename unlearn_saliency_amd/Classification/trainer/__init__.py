"""Evaluation loop used for the UA/RA/TA accuracies.  (The reference's trainer package also
exports pre-training loops — out of scope, SURVEY.md §2 C8 — and a `train_with_rewind` that
does not exist upstream; neither is reproduced.)"""
from .val import validate
