"""Per-kernel micro-benchmark at the BASELINE workload's shapes (one model call = 256 images).

    python tools/kbench.py [name-substring ...]

HIP-event timing on the launch stream, median of `reps` launches after warm-up, random data (never zeros: DVFS).
Prints achieved TFLOP/s (algorithmic FLOP) for the MFMA kernels and GB/s (algorithmic bytes) for the HBM-bound ones.
"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "srl-zoo_amd"), REPO]
import torch  # noqa: E402
from srlz import _cabi as C  # noqa: E402

DEV = "cuda"
N = int(os.environ.get("KB_N", "256"))


def timeit(fn, reps=12, warm=25):  # long warm-up: the clock governor needs ~10 ms of load to settle
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        e.synchronize()
        ts.append(s.elapsed_time(e))
    ts.sort()
    return ts[len(ts) // 2] * 1e-3, ts[0] * 1e-3


def report(name, sec, best, flop=None, bytes_=None):
    extra = ""
    if flop:
        extra += "  %7.1f TFLOP/s (best %.1f)" % (flop / sec / 1e12, flop / best / 1e12)
    if bytes_:
        extra += "  %7.0f GB/s" % (bytes_ / sec / 1e9)
    print("%-44s %9.1f us%s" % (name, sec * 1e6, extra), flush=True)


def rnd(*shape):
    return torch.randn(*shape, device=DEV)


def conv64_cases():
    # (label, hi, stride, pad, transposed)
    return [("conv2 3x3 s1 56x56", 56, 1, 1, 0), ("conv3 3x3 s2 27->14", 27, 2, 1, 0),
            ("convT1 6->13", 6, 2, 0, 1), ("convT2 13->27", 13, 2, 0, 1), ("convT3 27->55", 27, 2, 0, 1),
            ("convT4 55->111", 55, 2, 0, 1)]


def bench_conv64(sel):
    for label, hi, s, p, t in conv64_cases():
        ho = (hi - 1) * s - 2 * p + 3 if t else (hi + 2 * p - 3) // s + 1
        d = C.Conv64Desc(N, hi, hi, ho, ho, 3, s, p, t)
        flop = 2.0 * 9 * 64 * 64 * N * (hi * hi if t else ho * ho)
        x, dy = rnd(N, hi, hi, 64), rnd(N, ho, ho, 64)
        w = rnd(64, 64, 3, 3) * 0.05
        packs = torch.empty(2, C.conv64_packed_floats(), device=DEV)
        st = C.stream()
        C.conv64_pack_weights(C.ptr(w), C.ptr(packs[0]), C.ptr(packs[1]), d, st)
        y, dx = torch.empty(N, ho, ho, 64, device=DEV), torch.empty(N, hi, hi, 64, device=DEV)
        stats = torch.empty(C.conv64_fwd_tiles(d), 128, device=DEV)
        nb = C.conv64_bwd_weight_workspace(d)
        ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
        dw, db = torch.empty(64, 64, 3, 3, device=DEV), torch.empty(64, device=DEV)
        if sel("fwd"):
            report(label + " fwd", *timeit(lambda: C.conv64_fwd(C.ptr(x), C.ptr(packs[0]), None, C.ptr(y), C.ptr(stats), None, d, st)), flop=flop)
        if sel("dgrad"):
            report(label + " dgrad", *timeit(lambda: C.conv64_bwd_data(C.ptr(dy), C.ptr(packs[1]), C.ptr(dx), None, d, st)), flop=flop)
        if sel("wgrad"):
            report(label + " wgrad(+reduce)", *timeit(lambda: C.conv64_bwd_weight(C.ptr(x), C.ptr(dy), C.ptr(dw), C.ptr(db), None, None, C.ptr(ws), nb, d, st)), flop=flop)


def bench_skinny(sel):
    st = C.stream()
    d = C.SkinnyDesc(N, 3, 224, 224, 112, 112, 0)
    x, w = rnd(N, 3, 224, 224), rnd(64, 3, 7, 7) * 0.1
    y, dy = torch.empty(N, 112, 112, 64, device=DEV), rnd(N, 112, 112, 64)
    stats = torch.empty(C.skinny_tiles(d), 128, device=DEV)
    flop = 2.0 * 147 * 64 * N * 112 * 112
    if sel("conv1 fwd"):
        report("conv1 7x7 s2 fwd", *timeit(lambda: C.conv1_fwd(C.ptr(x), C.ptr(w), C.ptr(y), C.ptr(stats), d, st)), flop=flop)
    nb = C.skinny_bwd_weight_workspace(d)
    ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
    dw = torch.empty(64, 3, 7, 7, device=DEV)
    if sel("conv1 wgrad"):
        report("conv1 7x7 s2 wgrad(+reduce)", *timeit(lambda: C.conv1_bwd_weight(C.ptr(x), C.ptr(dy), C.ptr(dw), C.ptr(ws), nb, d, st)), flop=flop)
    d1 = C.SkinnyDesc(N, 3, 224, 224, 111, 111, 1)
    xf, wt, b = rnd(N, 111, 111, 64), rnd(64, 3, 4, 4) * 0.1, rnd(3)
    img, dimg = torch.empty(N, 3, 224, 224, device=DEV), rnd(N, 3, 224, 224)
    flop1 = 2.0 * 48 * 64 * N * 111 * 111
    if sel("convT5 fwd"):
        report("convT5 4x4 s2 fwd", *timeit(lambda: C.convT_out_fwd(C.ptr(xf), C.ptr(wt), C.ptr(b), C.ptr(img), None, d1, st)), flop=flop1,
               bytes_=4.0 * N * (111 * 111 * 64 + 3 * 224 * 224))
    bnp1 = torch.cat((torch.zeros(64), torch.ones(64), torch.ones(64), torch.zeros(64))).to(DEV)
    if sel("convT5 fwd"):
        report("convT5 4x4 s2 fwd (+BN/ReLU on load)", *timeit(lambda: C.convT_out_fwd(C.ptr(xf), C.ptr(wt), C.ptr(b), C.ptr(img), C.ptr(bnp1), d1, st)), flop=flop1,
               bytes_=4.0 * N * (111 * 111 * 64 + 3 * 224 * 224))
    dxf = torch.empty(N, 111, 111, 64, device=DEV)
    if sel("convT5 dgrad"):
        report("convT5 4x4 s2 dgrad", *timeit(lambda: C.convT_out_bwd_data(C.ptr(dimg), C.ptr(wt), C.ptr(dxf), None, None, None, d1, st)), flop=flop1,
               bytes_=4.0 * N * (111 * 111 * 64 + 3 * 224 * 224))
    nb1 = C.skinny_bwd_weight_workspace(d1)
    ws1 = torch.empty(nb1, dtype=torch.uint8, device=DEV)
    dwt, dbt = torch.empty(64, 3, 4, 4, device=DEV), torch.empty(3, device=DEV)
    if sel("convT5 wgrad"):
        report("convT5 4x4 s2 wgrad(+reduce+dbias)", *timeit(lambda: C.convT_out_bwd_weight(C.ptr(xf), C.ptr(dimg), C.ptr(dwt), C.ptr(dbt), None, C.ptr(ws1), nb1, d1, st)), flop=flop1,
               bytes_=4.0 * N * (111 * 111 * 64 + 3 * 224 * 224))


def bench_bn(sel):
    st = C.stream()
    nbw = C.bn_bwd_workspace(0)
    ws = torch.empty(nbw, dtype=torch.uint8, device=DEV)
    for label, h, pad in (("bn+relu+pool 112->56", 112, 1), ("bn+relu+pool 56->27", 56, 0)):
        hp = (h + 2 * pad - 3) // 2 + 1
        y, dp = rnd(N, h, h, 64), rnd(N, hp, hp, 64)
        bnp = torch.cat((torch.zeros(64), torch.ones(64), torch.ones(64), torch.zeros(64))).to(DEV)
        pooled, arg = torch.empty(N, hp, hp, 64, device=DEV), torch.empty(N, hp, hp, 64, dtype=torch.uint8, device=DEV)
        dy, dg, db = torch.empty_like(y), torch.empty(64, device=DEV), torch.empty(64, device=DEV)
        d = C.PoolDesc(N, h, h, hp, hp, pad, 0)
        if sel("pool fwd"):
            report(label + " fwd", *timeit(lambda: C.bn_relu_pool_fwd(C.ptr(y), C.ptr(bnp), C.ptr(pooled), C.ptr(arg), d, st)),
                   bytes_=4.0 * y.numel() + 5.0 * pooled.numel())
        C.bn_relu_pool_fwd(C.ptr(y), C.ptr(bnp), C.ptr(pooled), C.ptr(arg), d, st)
        if sel("pool bwd"):
            report(label + " bwd (reduce+apply)", *timeit(lambda: C.bn_relu_pool_bwd(C.ptr(y), C.ptr(bnp), C.ptr(arg), C.ptr(dp), C.ptr(pooled), C.ptr(dy), C.ptr(dg), C.ptr(db), 1, C.ptr(ws), nbw, d, st)),
                   bytes_=8.0 * y.numel() + 5.0 * pooled.numel())
    for label, h in (("bn+relu 111x111", 111), ("bn+relu 55x55", 55)):
        y, da = rnd(N, h, h, 64), rnd(N, h, h, 64)
        bnp = torch.cat((torch.zeros(64), torch.ones(64), torch.ones(64), torch.zeros(64))).to(DEV)
        a, dy = torch.empty_like(y), torch.empty_like(y)
        dg, db = torch.empty(64, device=DEV), torch.empty(64, device=DEV)
        if sel("relu fwd"):
            report(label + " fwd", *timeit(lambda: C.bn_relu_fwd(C.ptr(y), C.ptr(bnp), C.ptr(a), N * h * h, st)), bytes_=8.0 * y.numel())
        if sel("relu bwd"):
            report(label + " bwd (reduce+apply)", *timeit(lambda: C.bn_relu_bwd(C.ptr(y), C.ptr(bnp), C.ptr(da), C.ptr(dy), C.ptr(dg), C.ptr(db), 1, C.ptr(ws), nbw, N * h * h, 1, st)),
                   bytes_=12.0 * y.numel())


def bench_peak(sel):
    if not sel("mfma"):
        return
    cus = C.device_cus()
    for per_cu in (1, 2):
        blocks, iters = cus * per_cu, 20000
        out = torch.empty(blocks * 256, device=DEV)
        sec, best = timeit(lambda: C.debug_mfma_peak(C.ptr(out), blocks, iters, C.stream()), reps=5, warm=1)
        flop = blocks * 4.0 * 4 * iters * 4096
        report("fp32 MFMA 32x32x2 peak, %d WG/CU" % per_cu, sec, best, flop=flop)


def bench_mfma_valu(sel):
    """What does an instruction next to fp32 MFMAs cost?  us per launch of 256 x per_cu workgroups, 4 x 4000 MFMAs per wave."""
    if not sel("valu"):
        return
    cus = C.device_cus()
    names = ["v_fma_f32", "v_add_u32", "v_pk_fma_f32", "v_mov_b32"]
    iters = 4000
    out = torch.empty(cus * 2 * 512, device=DEV)
    for split in (0, 1):
        for per_cu in ((1, 2) if not split else (1,)):
            for kind in range(4):
                row = []
                for kv in (0, 1, 2, 4, 8, 16):
                    sec, best = timeit(lambda: C.debug_mfma_valu(C.ptr(out), cus * per_cu, iters, kv, kind, split, C.stream()), reps=3, warm=1)
                    row.append("%d: %.0f" % (kv, best * 1e6))
                print("mfma+valu %-12s %s %d WG/CU   us by instructions per MFMA  %s" % (
                    names[kind], "other wave of the SIMD" if split else "same wave", per_cu, "  ".join(row)))


def main():
    keys = [k.lower() for k in sys.argv[1:]]

    def group(name):
        def sel(sub):
            full = (name + " " + sub).lower()
            return not keys or any(k in full for k in keys)
        return sel
    print("N = %d images per call, device %s" % (N, torch.cuda.get_device_name(0)))
    bench_peak(group("peak"))
    bench_mfma_valu(group("mfma"))
    bench_conv64(group("conv64"))
    bench_skinny(group("skinny"))
    bench_bn(group("bn"))


if __name__ == "__main__":
    main()
