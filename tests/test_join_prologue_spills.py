"""Build-time guard against the wrong-code pattern behind the tracer's fault family (tools/repro/README.md, third case): a VGPR spill
store placed at the head of the block that joins a divergent `if`, AHEAD of the `s_or_b64 exec` that re-enables the lanes which skipped the
`if`, for a register defined before the `if` -- those lanes' value is never saved, and a later reload hands them whatever the slot held.  The
compiler emits it for the 4-wave HBM-scene kernel when a real call is added to start_path, at -O1 and -O3 alike; nothing announces it.  This
test disassembles every shipped translation unit (gfx950 code objects of lib/obj/*.o, as built by _build.build()) and asks for zero occurrences;
the scanner itself is pinned on the disassembly of the faulting kernel (tests/golden/join_prologue_spill_sample.s, cut from that build)."""
import importlib
import os
import re
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "tools", "repro"))
import scan_join_spills as S  # noqa: E402


def test_scanner_finds_the_known_bad_join():
    found = S.scan_text(open(os.path.join(HERE, "golden", "join_prologue_spill_sample.s")).read().splitlines(), quiet=True)
    assert [(a, s) for _, a, s in found] == [(0x278664, "scratch_store_dwordx2 off, v[38:39], off offset:1700")]
    # the same block with the store moved behind the exec restore (what correct code looks like) is clean
    lines = open(os.path.join(HERE, "golden", "join_prologue_spill_sample.s")).read().splitlines()
    i = next(k for k, ln in enumerate(lines) if "scratch_store_dwordx2 off, v[38:39], off offset:1700" in ln)
    assert "s_or_b64 exec, exec, s[4:5]" in lines[i + 1]
    addr = lambda ln: re.search(r"// ([0-9A-F]+):", ln).group(1)
    st, ex = lines[i], lines[i + 1]
    lines[i] = ex.replace("// " + addr(ex), "// " + addr(st))                                   # the exec restore first (4 bytes) ...
    lines[i + 1] = st.replace("// " + addr(st), "// %012X" % (int(addr(st), 16) + 4))           # ... then the spill store
    assert S.scan_text(lines, quiet=True) == []


@pytest.mark.parametrize("unit", ["gpt_capi", "gbdpt_capi", "poisson_capi"])
def test_no_shipped_kernel_spills_in_a_join_prologue(unit):
    b = importlib.import_module("gradientdomain-mitsuba_amd._build")
    b.build()
    obj = os.path.join(b.OBJDIR, unit + ".o")
    if not os.path.exists(obj):
        pytest.skip("lib/obj/ is not in this tree (the objects stay where the library was built)")
    found = S.scan(obj, quiet=True)
    assert found == [], "spill stores ahead of an exec restore in %s:\n%s" % (unit, "\n".join("%s  branch %#x  %s" % f for f in found))


def test_else_branch_copies_into_a_spill_slot_are_not_a_join_prologue():
    """A two-sided `if` whose value goes through a spill slot: the THEN lanes store before the flip to the ELSE lanes (s_andn2_saveexec), the ELSE lanes
    store in their own short branch, then exec is restored.  The branch over the THEN part lands on the flip, a few instructions ahead of an
    `s_or_b64 exec`: looks like the bad pattern, is correct code (met in round 4's build of k_render<1,0,2,1,0,1>).  The same stores with the flip taken
    out -- a store under the `if`'s mask at the head of the real join -- must still be reported."""
    def text(flip):
        body = ["0000000000000000 <k>:",
                "\tv_mov_b32_e32 v28, v1                                       // 000000000000: 00000000",
                "\ts_and_saveexec_b64 s[6:7], s[0:1]                          // 000000000004: 00000000",
                "\ts_cbranch_execz 2                                          // 000000000008: 00000000",
                "\tv_mov_b32_e32 v54, v2                                      // 00000000000C: 00000000",
                "\tscratch_store_dword off, v54, off offset:916               // 000000000010: 00000000",
                ("\ts_andn2_saveexec_b64 s[6:7], s[10:11]                     // 000000000014: 00000000" if flip else
                 "\ts_nop 0                                                   // 000000000014: 00000000"),
                "\ts_nop 0                                                    // 000000000018: 00000000",
                "\tscratch_store_dword off, v28, off offset:916               // 00000000001C: 00000000",
                "\ts_or_b64 exec, exec, s[6:7]                                // 000000000020: 00000000",
                "\ts_endpgm                                                   // 000000000024: 00000000"]
        return body
    assert S.scan_text(text(True), quiet=True) == []
    bad = S.scan_text(text(False), quiet=True)
    assert [(a, s) for _, a, s in bad] == [(0x8, "scratch_store_dword off, v28, off offset:916")]
