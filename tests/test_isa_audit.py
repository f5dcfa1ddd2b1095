"""Static guard for the compiler-level pathologies of profiles/NOTES.md 5.2, on the ISA hipcc generates for gfx950 (no GPU needed):
no kernel may serialise its global stores behind full memory waits, and the hot kernels must not spill (a spill reload is a
vector-memory load: waiting for it drains every store issued before it)."""
import os
import shutil
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
import isa_audit  # noqa: E402

HOT = ("conv64_fwd_kernel<4, false>", "conv64_fwd_kernel<4, true>", "conv64_wgrad_ring_kernel", "conv64_wgrad_gather_kernel",
       "conv64_dgrad_poolsum_kernel<1>", "conv64_dgrad_poolsum_kernel<2>",
       "conv64_wgrad_ring_s2_kernel", "conv64_gather_pipe_kernel",
       "skinny_conv_kernel<7, 3, false, false, float>",
       "skinny_conv_kernel<4, 0, true, false, float>", "skinny_wgrad_kernel<7, 3, true, float>",
       "skinny_wgrad_kernel<7, 3, true, unsigned char>", "skinny_wgrad_kernel<4, 0, false, float>",
       "convT_out_os_kernel<1, false, float, false>", "convT_out_os_kernel<1, true, float, false>",
       "convT_out_os_kernel<1, true, unsigned char, false>", "convT_out_os_bwd_kernel<1>", "convT_out_os_bwd_kernel<2>")


@pytest.mark.skipif(not os.path.exists(isa_audit.HIPCC) and shutil.which("hipcc") is None, reason="needs hipcc")
@pytest.mark.parametrize("source", ["conv64.hip", "skinny.hip", "convt_out.hip", "linear.hip"])
def test_no_serialised_stores_and_no_spills_in_hot_kernels(source):
    ks = list(isa_audit.kernels(isa_audit.disassemble(os.path.join(isa_audit.CSRC, source))))
    names = isa_audit.demangle([k for k, _ in ks])
    assert ks, "no kernels found in " + source
    seen = set()
    for (_, body), name in zip(ks, names):
        a = isa_audit.audit(body)
        assert a["store_wait_chain"] < 2, "%s: %d stores each wait for the previous one (s_waitcnt vmcnt(0) between them)" % (
            name, a["store_wait_chain"] + 1)
        for hot in HOT:
            if name.startswith(hot):
                seen.add(hot)
                assert a["scratch_reloads"] == 0, "%s spills (%d scratch reloads)" % (name, a["scratch_reloads"])
    if source != "linear.hip":
        where = lambda h: "conv64.hip" if h.startswith("conv64") else "convt_out.hip" if h.startswith("convT_out_os") else "skinny.hip"
        expected = [h for h in HOT if where(h) == source]
        assert set(expected) <= seen, "kernels renamed? missing %s" % sorted(set(expected) - seen)
