"""Static scan for the wrong-code pattern of tools/repro/README.md (third case): a VGPR spill store placed at the head of a block that joins a
divergent `if` -- the target of its `s_cbranch_execz` -- AHEAD of the `s_or_b64 exec, exec, sN` that re-enables the lanes which skipped the `if`,
for a register whose last definition lies BEFORE the `if` (so the lanes outside the `if` own a value too, and lose it).
    python tools/repro/scan_join_spills.py <code object or .so/.o with a gfx950 fatbin> [kernel-name-substring]
Prints every candidate: kernel, branch address, store, distance back to the register's last definition."""
import re, subprocess, sys, os, tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
BUNDLER = "/opt/rocm/lib/llvm/bin/clang-offload-bundler"
OBJCOPY = "/opt/rocm/lib/llvm/bin/llvm-objcopy"


def code_object(path):
    head = open(path, "rb").read(20)
    if head[:4] == b"\x7fELF" and head[18] == 0xE0:          # EM_AMDGPU
        return path
    tmp = tempfile.mkdtemp()
    fat = os.path.join(tmp, "f.fatbin"); co = os.path.join(tmp, "k.co")
    subprocess.check_call([OBJCOPY, "-O", "binary", "--only-section=.hip_fatbin", path, fat])
    subprocess.check_call([BUNDLER, "--type=o", "--unbundle", "--input=" + fat, "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
    return co


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def scan(co, want="", quiet=False):
    txt = subprocess.run([OBJDUMP, "-d", code_object(co)], capture_output=True, text=True, check=True).stdout.splitlines()
    assert sum(1 for ln in txt if "s_endpgm" in ln) > 0, "no gfx950 code in %s" % co
    return scan_text(txt, want, quiet)


def scan_text(txt, want="", quiet=False):
    """txt: lines of `llvm-objdump -d`.  Returns the list of candidates (kernel, branch address, store text)."""
    found = []
    kernels, cur = {}, None
    for ln in txt:
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", ln)
        if m: cur = m.group(1); kernels[cur] = []; continue
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*// ([0-9A-F]+):", ln)
        if m and cur: kernels[cur].append((int(m.group(3), 16), m.group(1), m.group(2)))
    total = 0
    for name, ins in kernels.items():
        if want and want not in name: continue
        index = {a: i for i, (a, _, _) in enumerate(ins)}
        for i, (a, op, args) in enumerate(ins):
            if op not in ("s_cbranch_execz", "s_cbranch_execnz"): continue
            try: off = int(args.split()[0])
            except ValueError: continue
            if off <= 0: continue
            t = a + 4 + 4 * off
            j = index.get(t)
            if j is None: continue
            end = next((q for q in range(j, min(j + 8, len(ins))) if ins[q][1].startswith("s_or_b64") and ins[q][2].startswith("exec, exec")), None)
            if end is None or op != "s_cbranch_execz": continue          # (only a block whose prologue does end in an exec restore is a join)
            # ... and whose prologue runs under the `if`'s mask all the way: a target that first flips to the ELSE lanes (s_andn2_saveexec / s_or_saveexec)
            # and then runs a short else-branch is no join -- stores in there are the else lanes' own copies into the slot, the then lanes made theirs
            # before the flip (met in round 4: k_render<1,0,2,1,0,1>, a two-sided phi through a spill slot; correct code)
            if any(ins[q][1].startswith(("s_andn2_saveexec", "s_or_saveexec", "s_and_saveexec", "s_xor_saveexec", "s_cbranch", "s_branch")) or
                   (ins[q][1].startswith(("s_mov_b64", "s_and_b64", "s_andn2_b64", "s_xor_b64")) and ins[q][2].startswith("exec")) for q in range(j, end)): continue
            k = j
            while k < end:
                if ins[k][1].startswith("scratch_store") and ", off" in ins[k][2]:
                    src = set()
                    for tok in re.split(r",\s*", ins[k][2]): src |= regs(tok)
                    # last definition of the stored register(s): walk back for an instruction whose FIRST operand writes it
                    d = None
                    for b in range(k - 1, max(k - 6000, -1), -1):
                        o, ar = ins[b][1], ins[b][2]
                        if o.startswith(("scratch_store", "global_store", "flat_store", "ds_write", "s_", "buffer_store")): continue
                        first = re.split(r",\s*", ar)[0] if ar else ""
                        if regs(first) & src: d = b; break
                    if d is not None and d < i:
                        total += 1
                        found.append((name, a, ins[k][1] + " " + ins[k][2]))
                        if not quiet: print("%s\n   branch %#x -> %#x | %s %s | last def %d instructions before the branch: %s %s" % (
                            name[:110], a, t, ins[k][1], ins[k][2], i - d, ins[d][1], ins[d][2][:60]))
                k += 1
    if not quiet: print("candidates: %d" % total)
    return found


if __name__ == "__main__":
    scan(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
