"""Host logic of the 64->64 convolution family: the virtual-grid programs built by csrc/conv64.hip::build_program
(dumped through srlz_conv64_debug_program, a host-only call) are interpreted here in numpy and must reproduce
torch's conv2d / conv_transpose2d and their data gradients for every layer shape of the network."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

# (hi, stride, pad, transposed) of every 64->64 layer: conv2, conv3, ConvT1..4 (reference models/models.py:54-78)
LAYERS = [(56, 1, 1, 0), (27, 2, 1, 0), (6, 2, 0, 1), (13, 2, 0, 1), (27, 2, 0, 1), (55, 2, 0, 1),
          (9, 1, 1, 0), (10, 2, 1, 0), (7, 2, 0, 1)]


def out_size(hi, s, p, t):
    return (hi - 1) * s - 2 * p + 3 if t else (hi + 2 * p - 3) // s + 1


def get_program(cabi, n, hi, s, p, t, backward):
    d = cabi.Conv64Desc(n, hi, hi, out_size(hi, s, p, t), out_size(hi, s, p, t), 3, s, p, t)
    buf = (ctypes.c_int * 64)()
    cnt = cabi.conv64_debug_program(d, backward, buf, 64)
    assert cnt == 48, cabi.error_text()
    v = list(buf)[:cnt]
    keys = ["N", "PH", "PW", "ss", "Hs", "Ws", "ds", "Hd", "Wd", "min_off", "span", "s2"]
    P = dict(zip(keys, v[:12]))
    P["taps"] = [tuple(v[12 + 4 * i: 16 + 4 * i]) for i in range(9)]  # (src, dst, off, w)
    return P


def interpret(P, src, Wg):
    """src [N,Hs,Ws,Cin], Wg [9][Cin][Cout] -> dst [N,Hd,Wd,Cout] following the program semantics."""
    N, PH, PW = P["N"], P["PH"], P["PW"]
    total = N * PH * PW
    cout = Wg.shape[2]
    dst = np.full((N, P["Hd"], P["Wd"], cout), np.nan, dtype=np.float64)
    q = np.arange(total)
    n, rem = q // (PH * PW), q % (PH * PW)
    a, b = rem // PW, rem % PW
    for d in sorted(set(t[1] for t in P["taps"])):
        acc = np.zeros((total, cout))
        for (c, dd, off, w) in P["taps"]:
            if dd != d:
                continue
            q2 = q + off
            ok = (q2 >= 0) & (q2 < total)
            q2c = np.clip(q2, 0, total - 1)
            n2, rem2 = q2c // (PH * PW), q2c % (PH * PW)
            y = (rem2 // PW) * P["ss"] + (c >> 1)
            x = (rem2 % PW) * P["ss"] + (c & 1)
            ok &= (y < P["Hs"]) & (x < P["Ws"])
            vals = np.where(ok[:, None], src[n2, np.minimum(y, P["Hs"] - 1), np.minimum(x, P["Ws"] - 1), :], 0.0)
            acc += vals @ Wg[w]
        oy, ox = a * P["ds"] + (d >> 1), b * P["ds"] + (d & 1)
        ok = (oy < P["Hd"]) & (ox < P["Wd"])
        assert not np.isfinite(dst[n[ok], oy[ok], ox[ok], 0]).any(), "an output position is produced twice"
        dst[n[ok], oy[ok], ox[ok], :] = acc[ok]
    assert np.isfinite(dst).all(), "some output position is never produced"
    return dst


@pytest.mark.parametrize("hi,s,p,t", LAYERS)
def test_forward_program_matches_torch(cabi, hi, s, p, t):
    rs = np.random.RandomState(hi * 10 + s)
    N, C = 2, 3
    x = rs.randn(N, hi, hi, C)
    Wg = rs.randn(9, C, C)  # [tap][cin][cout]
    P = get_program(cabi, N, hi, s, p, t, 0)
    got = interpret(P, x, Wg)
    xt = torch.from_numpy(x).permute(0, 3, 1, 2)
    if t:
        w = torch.from_numpy(Wg.reshape(3, 3, C, C)).permute(2, 3, 0, 1).contiguous()  # [ci,co,ky,kx]
        ref = F.conv_transpose2d(xt, w, stride=s, padding=p)
    else:
        w = torch.from_numpy(Wg.reshape(3, 3, C, C)).permute(3, 2, 0, 1).contiguous()  # [co,ci,ky,kx]
        ref = F.conv2d(xt, w, stride=s, padding=p)
    ref = ref.permute(0, 2, 3, 1).numpy()
    assert got.shape == ref.shape
    np.testing.assert_allclose(got, ref, rtol=1e-10, atol=1e-10)


@pytest.mark.parametrize("hi,s,p,t", LAYERS)
def test_backward_data_program_matches_autograd(cabi, hi, s, p, t):
    rs = np.random.RandomState(hi * 7 + s + 100)
    N, C = 2, 3
    ho = out_size(hi, s, p, t)
    x = torch.from_numpy(rs.randn(N, C, hi, hi)).requires_grad_(True)
    dy = rs.randn(N, ho, ho, C)
    Wg = rs.randn(9, C, C)  # forward slabs [tap][cin][cout]
    if t:
        w = torch.from_numpy(Wg.reshape(3, 3, C, C)).permute(2, 3, 0, 1).contiguous()
        y = F.conv_transpose2d(x, w, stride=s, padding=p)
    else:
        w = torch.from_numpy(Wg.reshape(3, 3, C, C)).permute(3, 2, 0, 1).contiguous()
        y = F.conv2d(x, w, stride=s, padding=p)
    y.backward(torch.from_numpy(dy).permute(0, 3, 1, 2))
    ref = x.grad.permute(0, 2, 3, 1).numpy()
    P = get_program(cabi, N, hi, s, p, t, 1)
    got = interpret(P, dy, Wg.transpose(0, 2, 1))  # data-gradient slabs map cout -> cin
    np.testing.assert_allclose(got, ref, rtol=1e-10, atol=1e-10)


@pytest.mark.parametrize("hi,s,p,t", LAYERS)
def test_program_tap_grouping(cabi, hi, s, p, t):
    """Stride-2 programs group taps {4,2,2,1} by class (the weight-gradient kernel relies on it)."""
    for backward in (0, 1):
        P = get_program(cabi, 1, hi, s, p, t, backward)
        key = [(c, d) for (c, d, _, _) in P["taps"]]
        if P["s2"]:
            assert key[0] == key[1] == key[2] == key[3] and key[4] == key[5] and key[6] == key[7]
            assert len(set(key)) == 4
        else:
            assert len(set(key)) == 1
        assert sorted(w for (_, _, _, w) in P["taps"]) == list(range(9))


@pytest.mark.parametrize("hi,s,p,t", [(56, 1, 1, 0), (6, 2, 0, 1), (13, 2, 0, 1), (27, 2, 0, 1), (55, 2, 0, 1), (9, 1, 1, 0)])
def test_wgrad_ring_schedule_never_aliases(cabi, hi, s, p, t):
    """conv64_wgrad_ring_kernel keeps source row q in LDS slot q & 255 and, per 64-position chunk, lands only the 64 new
    rows of the NEXT chunk after the current chunk's last MFMA.  Replay that schedule for the forward program of every
    single-source-class layer: whenever a tap reads slot (q0 + row + off) & 255 it must still hold row q0 + row + off."""
    P = get_program(cabi, 2, hi, s, p, t, 0)
    assert len(set(c for (c, _, _, _) in P["taps"])) == 1, "ring kernel is only used for single-source-class programs"
    RING, TK = 256, 64
    span, min_off = P["span"], P["min_off"]
    assert TK + span <= RING
    total = P["N"] * P["PH"] * P["PW"]
    nchunks = (total + TK - 1) // TK
    offs = sorted(set(off for (_, _, off, _) in P["taps"]))
    assert offs[0] == min_off and offs[-1] - offs[0] == span
    for (c_begin, c_end) in ((0, nchunks), (3, min(nchunks, 9))):  # a workgroup's contiguous chunk range
        slot = {}
        q0 = c_begin * TK
        for r in range(0, TK + span, 64):  # prologue: passes of 64 rows, may run past TK + span
            for q in range(q0 + min_off + r, q0 + min_off + r + 64):
                slot[q & (RING - 1)] = q
        for chunk in range(c_begin, c_end):
            q0 = chunk * TK
            for row in range(TK):
                for off in offs:
                    q = q0 + row + off
                    assert slot.get(q & (RING - 1)) == q, (chunk, row, off)
            if chunk + 1 < c_end:  # landed after the barrier that follows this chunk's MFMA loop
                for q in range(q0 + min_off + TK + span, q0 + min_off + TK + span + 64):
                    slot[q & (RING - 1)] = q


def test_programs_for_random_shapes(cabi):
    """Beyond the network's own layers: random sizes (odd / even, rectangular inputs are square here by ABI), both strides,
    padded and unpadded, plain and transposed — forward and data-gradient programs against torch, plus the structural
    invariants the kernels rely on (a zero row/column behind every negative offset, span within the LDS ring)."""
    rs = np.random.RandomState(2024)
    checked, declined = 0, []
    for _ in range(60):
        t = int(rs.randint(0, 2))
        s = int(rs.randint(1, 3))
        p = int(rs.randint(0, 2))
        hi = int(rs.randint(3, 20))
        if not t and hi + 2 * p < 3:
            continue
        ho = out_size(hi, s, p, t)
        if ho < 1:
            continue
        d = cabi.Conv64Desc(2, hi, hi, ho, ho, 3, s, p, t)
        buf = (ctypes.c_int * 64)()
        if cabi._lib.srlz_conv64_debug_program(ctypes.byref(d), 0, buf, 64) != 48 or \
                cabi._lib.srlz_conv64_debug_program(ctypes.byref(d), 1, buf, 64) != 48:
            # a geometry one of the two programs cannot express (e.g. the data gradient of an unpadded stride-1 conv needs
            # offsets of -2 rows): the C ABI rejects the descriptor loudly instead of computing something else
            declined.append((hi, s, p, t))
            continue
        N, C = 2, 2
        x = rs.randn(N, hi, hi, C)
        Wg = rs.randn(9, C, C)
        P = get_program(cabi, N, hi, s, p, t, 0)
        got = interpret(P, x, Wg)
        xt = torch.from_numpy(x).permute(0, 3, 1, 2).requires_grad_(True)
        if t:
            w = torch.from_numpy(Wg.reshape(3, 3, C, C)).permute(2, 3, 0, 1).contiguous()
            y = F.conv_transpose2d(xt, w, stride=s, padding=p)
        else:
            w = torch.from_numpy(Wg.reshape(3, 3, C, C)).permute(3, 2, 0, 1).contiguous()
            y = F.conv2d(xt, w, stride=s, padding=p)
        np.testing.assert_allclose(got, y.detach().permute(0, 2, 3, 1).numpy(), rtol=1e-10, atol=1e-10)
        dy = rs.randn(N, ho, ho, C)
        y.backward(torch.from_numpy(dy).permute(0, 3, 1, 2))
        Pb = get_program(cabi, N, hi, s, p, t, 1)
        gotb = interpret(Pb, dy, Wg.transpose(0, 2, 1))
        np.testing.assert_allclose(gotb, xt.grad.permute(0, 2, 3, 1).numpy(), rtol=1e-10, atol=1e-10)
        for prog in (P, Pb):
            assert prog["span"] + 64 <= 256 or len(set(c for (c, _, _, _) in prog["taps"])) > 1 or prog["PW"] > 95
            assert prog["min_off"] <= 0 <= prog["min_off"] + prog["span"]
        checked += 1
    assert checked >= 20, (checked, declined)
    assert all(not (s == 1 and p == 1 and not t) and not (s == 2 and p == 0 and t) and not (s == 2 and p == 1 and not t)
               for (_, s, p, t) in declined), "a geometry of the network's own layers was declined: %r" % declined


def test_conv1_weight_gradient_tap_table_is_a_conflict_free_deal():
    """TAP7 in csrc/skinny.hip (the column order of conv1's weight-gradient GEMM): every one of the 147 taps exactly once, and no two
    taps of an N-tile on the same LDS bank for the window pitches the kernel uses — the property the table exists for, checked on
    the table that is compiled (tools/tap_banks.py generated it)."""
    import os
    import re
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(os.path.dirname(here), "tools"))
    import tap_banks
    src = open(os.path.join(os.path.dirname(here), "srl-zoo_amd", "csrc", "skinny.hip")).read()
    xp, pp = (int(v) for v in re.search(r"constexpr int WG7_XP = (\d+), WG7_PP = (\d+);", src).groups())
    body = re.search(r"__constant__ unsigned char TAP7\[160\] = \{(.*?)\};", src, re.S).group(1)
    table = [int(v) for v in re.findall(r"\d+", body)]
    assert len(table) == 160
    assert sorted(t for t in table if t != 255) == list(range(147))
    for j in range(5):
        group = [t for t in table[32 * j:32 * j + 32] if t != 255]
        banks = [tap_banks.off(t, xp, pp) % 32 for t in group]
        assert len(set(banks)) == len(banks), (j, sorted(banks))
        assert table[32 * j] != 255  # unused lanes re-read the tile's first tap (a broadcast)
    assert table == tap_banks.deal(xp, pp)
    # the window fits its planes: 37 rows of the row pitch per (channel, column-parity) plane, one spare cell for dead elements
    assert 37 * xp <= pp - 1 and 6 * pp <= 5400
    # and the natural order does conflict, which is why the table exists
    assert max(tap_banks.natural_conflicts(24, 900)) >= 1
