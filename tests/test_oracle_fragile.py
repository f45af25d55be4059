"""mca_or_select_doa_fragile -- the classifier behind the DOA-bin parity bar (tests/parity_helpers.py): known cases of each
kind (SteeringBeamforming.cpp:146-195)."""
import numpy as np

from oracle import pyoracle as po

D, P = 37, 6


def _map(peaks):
    x = np.linspace(-1, 1, D)
    En = 0.3 + 0.01 * x + sum(h * np.exp(-((x - c) / 0.12) ** 2) for c, h in peaks)     # (a slope: no first difference near zero)
    return -15.0 * P + 30.0 * P * En           # un-normalised energies, as the oracle returns them


def test_clean_peak_is_pinned():
    E = _map([(0.2, 0.4)])
    assert not po.select_doa_fragile(E, P, 1, 1e-6)
    assert not po.select_doa_fragile(E, P, 1, 1e-3)


def test_peak_tie_is_fragile():
    E = _map([(-0.5, 0.4), (0.5, 0.4)])
    En = (E + 15.0 * P) / (30.0 * P)
    i0, i1 = int(np.argmax(En[:D // 2])), D // 2 + int(np.argmax(En[D // 2:]))
    En[i1] = En[i0]                               # two equal peaks: the first one wins by position only
    E = En * 30.0 * P - 15.0 * P
    b = po.select_doa(E, P, 5.0, 1)[2]
    assert b[0] == i0 and po.select_doa_fragile(E, P, 1, 1e-6)
    E[b[0]] += 30.0 * P * 1e-3                   # lift the winner clear of the other by 1e-3
    assert not po.select_doa_fragile(E, P, 1, 1e-6)
    assert po.select_doa_fragile(E, P, 1, 1e-2)


def test_sign_chain_tie_at_the_edge_is_fragile():
    """the hand-analysed round-2 case: the map's last first difference is ~1e-7 from zero while the edge energy reaches the
    winning peak's value: a peak can appear at the last position under rounding"""
    E = _map([(0.0, 0.2)])
    En = (E + 15.0 * P) / (30.0 * P)
    En[-6:] = En.max() + 0.05 + 0.01 * np.arange(6)      # a rising shelf at the edge, higher than the peak ...
    En[-1] = En[-2] - 4.7e-7                     # ... whose last step is a rounding error wide: a peak appears at D - 2 or not
    E2 = En * 30.0 * P - 15.0 * P
    assert po.select_doa_fragile(E2, P, 1, 1e-6)
    En[-1] = En[-2] - 1e-3                       # a clear step: pinned again
    assert not po.select_doa_fragile(En * 30.0 * P - 15.0 * P, P, 1, 1e-6)


def test_second_source_tie_and_zero_pick():
    E = _map([(-0.5, 0.5), (0.1, 0.3), (0.6, 0.3)])
    En = (E + 15.0 * P) / (30.0 * P)
    j1, j2 = 15 + int(np.argmax(En[15:24])), 26 + int(np.argmax(En[26:33]))
    En[j2] = En[j1]                               # the 2nd and 3rd peaks tie: 1 source pinned, 2 sources fragile
    E = En * 30.0 * P - 15.0 * P
    assert not po.select_doa_fragile(E, P, 1, 1e-6)
    assert po.select_doa_fragile(E, P, 2, 1e-6)
    flat = np.full(D, -15.0 * P + 30.0 * P * 0.5)          # no peak at all: every first difference is zero
    assert po.select_doa_fragile(flat, P, 1, 1e-6)


def test_local_bar_is_the_absolute_bar_where_energies_are_small_and_relative_where_they_are_large():
    """mca_or_select_doa_fragile_local (round 5): eps x max(1, |values compared|).  A near-tie of 3e-6 between two peaks of
    normalised energy 5 is a tie (fp32 resolves 4.8e-7 there); the same 3e-6 between peaks of energy 0.6 is not, even when another
    part of the row is large -- round 4's row-relative bar called that one a tie as well."""
    from parity_helpers import row_eps
    P, D = 28, 361
    x = np.arange(D)

    def row(h1, h2):
        En = 0.1 + 1e-4 * x + h1 * np.exp(-0.5 * ((x - 100) / 6.0) ** 2) + (h2 - 0.015) * np.exp(-0.5 * ((x - 250) / 6.0) ** 2)   # (0.015: the slope's share at 250)
        return En * 30.0 * P - 15.0 * P
    hi = row(5.0, 5.0 - 3e-6)
    assert not po.select_doa_fragile(hi, P, 1, 1e-6)            # absolute: pinned
    assert po.select_doa_fragile_local(hi, P, 1, 1e-6)          # 3e-6 <= 1e-6 x 5
    lo = row(0.5, 0.5 - 3e-6)
    assert not po.select_doa_fragile_local(lo, P, 2, 1e-6) and not po.select_doa_fragile(lo, P, 2, 1e-6)
    spike = lo.copy()
    spike[5] = (17.0 * 30 - 15) * P                             # one large value on the slope: no candidate (median-3 removes a lone sign), but max |En| = 17
    assert row_eps(spike, P) > 1e-5
    assert po.select_doa_fragile(spike, P, 2, row_eps(spike, P))              # round 4's bar: the two 0.6 peaks tie (3e-6 < 1.7e-5)
    assert not po.select_doa_fragile_local(spike, P, 2, 1e-6)                 # this bar: pinned (3e-6 between values of 0.6)
    assert po.select_doa_fragile_local(row(0.5, 0.5 - 5e-7), P, 2, 1e-6)      # and 5e-7 is a tie under every bar
