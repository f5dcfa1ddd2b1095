#!/usr/bin/env python
"""LDS banks of the image-window reads in conv1's weight-gradient kernel (skinny_wgrad_kernel<7,3>, srl-zoo_amd/csrc/skinny.hip).

A lane of N-tile j reads tap k = (c, ky, kx) of the de-interleaved window at (2c + (kx & 1)) * PP + ky * XP + (kx >> 1) (+ a part
that is the same for all lanes); ds_read_b32 serves 32 lanes per LDS cycle from 32 banks (bank = dword address mod 32).
    python tools/tap_banks.py            # conflicts of the natural column order for the forward kernel's pitches (24, 900),
                                         # the pitches for which no bank holds more than 5 of the 147 taps, and the TAP7 table
                                         # (5 tiles x 32 lanes, 255 = unused) for XP = 23, PP = 861 — checked conflict-free
"""
import collections

KT, NT = 147, 5


def off(k, xp, pp):
    c, ky, kx = k // 49, (k // 7) % 7, k % 7
    return (c * 2 + (kx & 1)) * pp + ky * xp + (kx >> 1)


def natural_conflicts(xp, pp):
    extra = []
    for j in range(NT):
        banks = collections.Counter(off(k, xp, pp) % 32 for k in range(j * 32, min(KT, j * 32 + 32)))
        extra.append(max(banks.values()) - 1)
    return extra


def deal(xp, pp):
    by_bank = collections.defaultdict(list)
    for k in range(KT):
        by_bank[off(k, xp, pp) % 32].append(k)
    if max(len(v) for v in by_bank.values()) > NT:
        return None
    groups = [[] for _ in range(NT)]
    for _, ks in sorted(by_bank.items(), key=lambda x: -len(x[1])):
        order = sorted(range(NT), key=lambda g: len(groups[g]))
        for k, g in zip(ks, order):
            groups[g].append(k)
    table = []
    for g in groups:
        g = sorted(g)
        assert len(set(off(k, xp, pp) % 32 for k in g)) == len(g) <= 32
        table += g + [255] * (32 - len(g))
    assert sorted(t for t in table if t != 255) == list(range(KT))
    return table


if __name__ == "__main__":
    print("natural order, XP = 24, PP = 900: extra LDS cycles per lane group and tile:", natural_conflicts(24, 900))
    print("pitches with at most 5 taps per bank:", [(xp, pp) for xp in range(19, 41) for pp in range(37 * xp, 37 * xp + 33)
                                                   if deal(xp, pp)])
    t = deal(23, 861)
    for i in range(0, 160, 16):
        print("    " + ", ".join("%3d" % v for v in t[i:i + 16]) + ",")
