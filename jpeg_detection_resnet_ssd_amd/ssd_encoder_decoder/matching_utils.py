"""Greedy bipartite and threshold ('multi') matching between ground-truth and anchor boxes -- host
numpy, same contracts as localisation_part/ssd_encoder_decoder/matching_utils.py:22-116."""
import numpy as np


def match_bipartite_greedy(weight_matrix):
    """For an (m ground truths, n anchors) weight matrix, repeatedly take the globally best remaining
    (ground truth, anchor) pair; returns the matched anchor index per ground truth.
    Ties resolve like np.argmax: lowest anchor index within a row, then lowest row."""
    w = np.array(weight_matrix, dtype=float, copy=True)
    m = w.shape[0]
    rows = np.arange(m)
    matches = np.zeros(m, dtype=int)
    for _ in range(m):
        best_anchor = np.argmax(w, axis=1)
        best_val = w[rows, best_anchor]
        gt = int(np.argmax(best_val))
        anchor = int(best_anchor[gt])
        matches[gt] = anchor
        w[gt, :] = 0
        w[:, anchor] = 0
    return matches


def match_multi(weight_matrix, threshold):
    """Every anchor is matched to its best ground truth if that weight reaches `threshold`.
    Returns (ground-truth indices, anchor indices)."""
    w = np.asarray(weight_matrix)
    cols = np.arange(w.shape[1])
    best_gt = np.argmax(w, axis=0)
    met = np.nonzero(w[best_gt, cols] >= threshold)[0]
    return best_gt[met], met
