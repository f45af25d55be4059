"""The DOA-bin parity bar shared by the GPU tests and tools/fuzz_parity.py.

A GPU bin must equal the oracle's, except on frames whose pick the ORACLE ITSELF cannot pin at the resolution of fp32 arithmetic:
`mca_or_select_doa_fragile` says whether perturbing the frame's normalised energies by eps could change a pick -- two peaks within
eps of each other, a first difference within eps of zero feeding the sign / median chain of SteeringBeamforming.cpp:159-165 at a
position that reaches the picked value (a peak can then appear anywhere, also at the map's edge, so no |delta bin| <= 1 clause
applies to those), or a zero pick next to a candidate within eps of zero.  Such frames are counted and bounded (max_ties); every
other difference fails.

eps is RELATIVE to the row (round 4): eps = EPS_TIE x max(1, max_d |En[d]|), En = (E + 15 P) / (30 P) (:155-156).  The reference's
normalisation does not bound En by 1 -- E reaches P K, so En reaches K / 30 = 17 -- and an fp32 value of 5 resolves 4.8e-7: with
an absolute 1e-6 the bar asked fp32 pipelines for differences of two units in the last place.  The one difference the 1 940
configurations of round 3 left unclassified (seed 60221, case 160, frame 54) is that: the oracle's En[339] - En[338] = -2.09e-6
at En ~ 5, and EVERY GPU mode sits 2.1e-6 from the oracle in that row -- fp32 -3.49e-6, fp16x3 -1.16e-6, adaptive 0 (a tie in
fp32: first maximum) -- profiles/r04_case160_dump.log.  1e-6 relative = 8 fp32 units in the last place of the row's peak."""
import numpy as np

from oracle import pyoracle as po

EPS_TIE = 1e-6


def row_eps(ora_energy_row, n_pairs, eps=EPS_TIE):
    """the perturbation the classifier tests a row with: eps relative to the row's largest normalised energy (never below eps)"""
    en = (np.asarray(ora_energy_row, dtype=np.float64) + 15.0 * n_pairs) / (30.0 * n_pairs)
    return eps * max(1.0, float(np.abs(en).max()))


def fragile(ora_energy_row, n_pairs, n_sources, eps=EPS_TIE):
    return po.select_doa_fragile(ora_energy_row, n_pairs, n_sources, row_eps(ora_energy_row, n_pairs, eps))


def classify_bins(gpu_bins, ora_bins, ora_energy, n_pairs):
    """gpu_bins, ora_bins: [F][S] (or [F]); ora_energy: [F][D] un-normalised oracle energies.
    Returns (ties, unclassified): lists of frame indices whose bins differ on fragile / on pinned oracle rows."""
    g = np.asarray(gpu_bins).reshape(len(gpu_bins), -1)
    o = np.asarray(ora_bins).reshape(len(ora_bins), -1)
    S = g.shape[1]
    ties, bad = [], []
    for t in np.unique(np.argwhere(g != o)[:, 0]):
        (ties if fragile(ora_energy[t], n_pairs, S) else bad).append(int(t))
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
