"""The DOA-bin parity bar shared by the GPU tests and tools/fuzz_parity.py.

A GPU bin must equal the oracle's, except on frames whose pick the ORACLE ITSELF cannot pin: `mca_or_select_doa_fragile` says
whether perturbing the frame's normalised energies by EPS_TIE (1e-6; the fp32 paths agree with the fp64 oracle to ~1e-8 there)
could change a pick -- two peaks within EPS_TIE of each other, a first difference within EPS_TIE of zero feeding the sign /
median chain of SteeringBeamforming.cpp:159-165 at a position that reaches the picked value (a peak can then appear anywhere,
also at the map's edge, so no |delta bin| <= 1 clause applies to those), or a zero pick next to a candidate within EPS_TIE
of zero.  Such frames are counted and bounded (max_ties); every other difference fails."""
import numpy as np

from oracle import pyoracle as po

EPS_TIE = 1e-6


def classify_bins(gpu_bins, ora_bins, ora_energy, n_pairs):
    """gpu_bins, ora_bins: [F][S] (or [F]); ora_energy: [F][D] un-normalised oracle energies.
    Returns (ties, unclassified): lists of frame indices whose bins differ on fragile / on pinned oracle rows."""
    g = np.asarray(gpu_bins).reshape(len(gpu_bins), -1)
    o = np.asarray(ora_bins).reshape(len(ora_bins), -1)
    S = g.shape[1]
    ties, bad = [], []
    for t in np.unique(np.argwhere(g != o)[:, 0]):
        (ties if po.select_doa_fragile(ora_energy[t], n_pairs, S, EPS_TIE) else bad).append(int(t))
    return ties, bad


def assert_bins(gpu_bins, ora_bins, ora_energy, n_pairs, max_ties=0):
    """exact match, or a frame the oracle flags as fragile (counted, at most max_ties)."""
    ties, bad = classify_bins(gpu_bins, ora_bins, ora_energy, n_pairs)
    if bad:
        t = bad[0]
        raise AssertionError("DOA bin mismatch on %d frame(s) whose oracle pick is pinned, first: frame %d gpu %s oracle %s"
                             % (len(bad), t, np.asarray(gpu_bins)[t].tolist(), np.asarray(ora_bins)[t].tolist()))
    assert len(ties) <= max_ties, "%d fragile-frame differences (allowed %d)" % (len(ties), max_ties)
    return len(ties)
