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


RING_ROWS, RING_MIRROR = 246, 8  # csrc/conv64.hip (the ring shrank from 248 rows when the two row tables moved into its 80 KB)


@pytest.mark.parametrize("hi,s,p,t", [(56, 1, 1, 0), (6, 2, 0, 1), (13, 2, 0, 1), (27, 2, 0, 1), (55, 2, 0, 1), (9, 1, 1, 0)])
def test_wgrad_ring_schedule_never_aliases(cabi, hi, s, p, t):
    """conv64_wgrad_ring_kernel keeps source row q in LDS slot q mod RING_ROWS (the first RING_MIRROR slots a second time behind
    the ring, so a reader that starts at slot u goes on to u + 7 without wrapping) and, per 64-position chunk, lands only the 64 new
    rows of the NEXT chunk after the current chunk's last MFMA.  Replay that schedule for the forward program of every
    single-source-class layer: whenever a tap reads row q0 + row + off through its block's start slot it must find that row."""
    P = get_program(cabi, 2, hi, s, p, t, 0)
    assert len(set(c for (c, _, _, _) in P["taps"])) == 1, "ring kernel is only used for single-source-class programs"
    TK = 64
    span, min_off = P["span"], P["min_off"]
    assert TK + span + TK <= RING_ROWS, "the launch condition of the ring kernel"
    total = P["N"] * P["PH"] * P["PW"]
    nchunks = (total + TK - 1) // TK
    offs = sorted(set(off for (_, _, off, _) in P["taps"]))
    assert offs[0] == min_off and offs[-1] - offs[0] == span

    def land(lds, q):  # rows64_store<RINGED>
        r = q % RING_ROWS
        lds[r] = q
        if r < RING_MIRROR:
            lds[r + RING_ROWS] = q

    for (c_begin, c_end) in ((0, nchunks), (3, min(nchunks, 9))):  # a workgroup's contiguous chunk range
        lds = {}
        q0 = c_begin * TK
        for r in range(0, TK + span, 64):  # prologue: passes of 64 rows, may run past TK + span
            for q in range(q0 + min_off + r, q0 + min_off + r + 64):
                land(lds, q)
        for chunk in range(c_begin, c_end):
            q0 = chunk * TK
            for off in offs:
                for b in range(TK // 8):  # a block of 4 k-steps reads rows u .. u + 7 from ONE wrapped start slot
                    u = (q0 + off + 8 * b) % RING_ROWS
                    for k in range(8):
                        assert lds.get(u + k) == q0 + off + 8 * b + k, (chunk, off, b, k)
            if chunk + 1 < c_end:  # landed after the barrier that follows this chunk's MFMA loop
                for q in range(q0 + min_off + TK + span, q0 + min_off + TK + span + 64):
                    land(lds, q)


def _rowtab_entry(P, q):
    """rowtab_build / gtab_build of csrc/conv64.hip for grid position q of one BatchNorm group: pixel index of the row's class-(0,0)
    source pixel << 4 | bit k: source class k = (cy << 1) | cx lies inside the image; 0 = no image."""
    PHW = P["PH"] * P["PW"]
    n1 = (q + PHW) // PHW  # (shifted by one image, as the kernel does for the negative q of the first tile)
    rem = (q + PHW) - n1 * PHW
    a, b = rem // P["PW"], rem % P["PW"]
    y0, x0 = a * P["ss"], b * P["ss"]
    if not 0 <= n1 - 1 < P["N"]:
        return 0
    f = sum(1 << k for k in range(4) if y0 + (k >> 1) < P["Hs"] and x0 + (k & 1) < P["Ws"])
    return ((((n1 - 1) * P["Hs"] + y0) * P["Ws"] + x0) << 4) | f


@pytest.mark.parametrize("hi,s,p,t", LAYERS)
def test_row_tables_address_the_rows_the_program_means(cabi, hi, s, p, t):
    """The per-tile row tables (round 4): for every staged row of every 128-position tile and every source class of the program,
    the pixel the table-driven staging loads — ((entry >> 4) + cy * Ws + cx) masked by the entry's class bit — is the pixel the
    program's definition S_c(q) = src[n, a * ss + cy, b * ss + cx] names, and rows outside the image are masked off; the packed
    destination word (rowinfo) does the same for D_d(q).  Entries must fit their bit fields at the network's largest batch."""
    TM = 128
    for backward in (0, 1):
        P = get_program(cabi, 3, hi, s, p, t, backward)
        PH, PW, N = P["PH"], P["PW"], P["N"]
        total = N * PH * PW
        classes = sorted(set(c for (c, _, _, _) in P["taps"]))
        for q0 in range(0, total, TM):
            for R in range(TM + P["span"]):
                q = q0 + P["min_off"] + R
                e = _rowtab_entry(P, q)
                n, rem = divmod(q, PH * PW) if q >= 0 else (-1, 0)
                a, b = rem // PW, rem % PW
                for c in classes:
                    y, x = a * P["ss"] + (c >> 1), b * P["ss"] + (c & 1)
                    inside = 0 <= q < total and y < P["Hs"] and x < P["Ws"]
                    assert bool((e >> c) & 1) == inside, (backward, q, c)
                    if inside:
                        assert (e >> 4) + (c >> 1) * P["Ws"] + (c & 1) == (n * P["Hs"] + y) * P["Ws"] + x, (backward, q, c)
        # destination side: pixel index of the class-(0,0) output << 2 | row below exists | column to the right exists << 1
        for q in range(total):
            n, rem = divmod(q, PH * PW)
            ya, xb = (rem // PW) * P["ds"], (rem % PW) * P["ds"]
            ri = -1 if not (ya < P["Hd"] and xb < P["Wd"]) else \
                (((n * P["Hd"] + ya) * P["Wd"] + xb) << 2) | (1 if ya + 1 < P["Hd"] else 0) | (2 if xb + 1 < P["Wd"] else 0)
            for d in sorted(set(dd for (_, dd, _, _) in P["taps"])):
                need = (d >> 1) | ((d & 1) << 1)
                inside = ya + (d >> 1) < P["Hd"] and xb + (d & 1) < P["Wd"]
                assert (ri >= 0 and (ri & need) == need) == inside, (backward, q, d)
                if inside:
                    assert (ri >> 2) + (d >> 1) * P["Wd"] + (d & 1) == (n * P["Hd"] + ya + (d >> 1)) * P["Wd"] + xb + (d & 1)
    # bit budget at a large call (1149 images of the widest layer: the most one launch took until round 5; beyond their budget the
    # launchers reject a call): 28-bit pixel indices, 32-bit float offsets
    P = get_program(cabi, 1149, hi, s, p, t, 0)
    assert P["N"] * P["Hs"] * P["Ws"] < 1 << 28 and P["N"] * P["Hs"] * P["Ws"] * 64 < 1 << 32
    assert P["N"] * P["Hd"] * P["Wd"] < 1 << 29


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
